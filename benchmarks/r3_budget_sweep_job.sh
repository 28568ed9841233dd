#!/bin/bash
# asynchronous loop: time budget per launch (load balance between boards of very different descent depth)
O=gpurun_out/${OUT:-r3t}; mkdir -p $O
for cfg in "0 64 64" "500 4096 256" "1000 4096 128" "2000 4096 64" "4000 4096 32"; do
  set -- $cfg
  BOARDS=4096 PLAYOUTS=400 BUDGET=$1 MAXP=$2 FIX=0 ROUNDS=$3 ITERS=$((30000 / $3 * 1)) EVERY=$((30000 / $3 / 4)) timeout 300 python benchmarks/async_debug.py > $O/budget_$1.log 2>&1
  echo "budget $1 maxp $2:"; tail -2 $O/budget_$1.log | head -1 | cut -c1-420
done
