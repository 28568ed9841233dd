#!/bin/bash
# asynchronous loop, late-game regime: time budget per launch
O=gpurun_out/${OUT:-r3t}; mkdir -p $O
for b in ${BUDGETS:-1000 2000 3000}; do
  BOARDS=${BOARDS:-4096} PLAYOUTS=400 BUDGET=$b MAXP=4096 FIX=0 SKIP_ROUNDS=8960 ROUNDS=64 ITERS=$((6400000 / (b + 700) / 64)) EVERY=$((6400000 / (b + 700) / 64 / 2)) timeout 300 python benchmarks/async_debug.py > $O/budget_$b.log 2>&1
  echo "budget $b:"; grep '^{' $O/budget_$b.log | tail -1 | cut -c1-330
done
