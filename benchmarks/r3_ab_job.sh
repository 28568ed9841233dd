#!/bin/bash
# same-box A/B of library builds on the asynchronous loop's late-game regime: LIBS="path1 path2 ..."
O=gpurun_out/${OUT:-r3ab}; mkdir -p $O
for lib in $LIBS; do
  n=$(basename $lib .so)
  QZ_BENCH_LIB=$GRAFT_REPO_ROOT/$lib BOARDS=4096 PLAYOUTS=400 BUDGET=${BUDGET:-1000} MAXP=4096 FIX=0 SKIP_ROUNDS=${SKIP_ROUNDS:-12800} ROUNDS=64 ITERS=${ITERS:-48} EVERY=${EVERY:-16} timeout 300 python benchmarks/async_debug.py > $O/$n.log 2>&1
  echo "$n:"; grep '^{' $O/$n.log | tail -2 | cut -c1-330
done
