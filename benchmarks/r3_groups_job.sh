#!/bin/bash
# asynchronous loop, late-game regime: board groups on separate HIP streams (their launches overlap)
O=gpurun_out/${OUT:-r3g}; mkdir -p $O
for cfg in ${CFGS:-"1 1000" "2 1000" "4 1000" "2 700"}; do
  set -- $cfg
  BOARDS=${BOARDS:-4096} PLAYOUTS=400 NGROUPS=$1 BUDGET=$2 MAXP=4096 FIX=0 SKIP_ROUNDS=8960 ROUNDS=64 ITERS=50 EVERY=25 timeout 300 python benchmarks/async_debug.py > $O/g$1_b$2.log 2>&1
  echo "groups $1 budget $2:"; grep '^{' $O/g$1_b$2.log | tail -1 | cut -c1-300
done
