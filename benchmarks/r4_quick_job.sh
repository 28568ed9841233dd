#!/bin/bash
# quick check of kernel changes: the parity tests of the tree kernels, then the loop's interval statistics in the late-game
# regime for the product build and (W8=1) for the 8-wavefronts-per-SIMD build at twice the boards
O=gpurun_out/${OUT:-r4q}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_async_oracle.py tests/test_gpu_async.py tests/test_gpu_mcts.py -m gpu -x -q --timeout=600 2>&1 | tail -6 | tee $O/pytest.log
run() {  # name boards lib
  QZ_BENCH_LIB=$3 BOARDS=$2 PLAYOUTS=400 MAXP=4096 BUDGET=${BUDGET:-1000} FIX=0 MAXD=992 SKIP_ROUNDS=${SKIP:-6400} ITERS=${ITERS:-48} ROUNDS=64 EVERY=16 GRAPH=1 timeout 600 python benchmarks/async_debug.py > $O/async_$1.log 2>&1
  tail -4 $O/async_$1.log | cut -c1-330
}
run default 4096 ""
if [ -n "$W8" ]; then make -C tests/hip -s libqzero_hip_w8.so && run w8_8192 8192 $PWD/tests/hip/libqzero_hip_w8.so && run w8_4096 4096 $PWD/tests/hip/libqzero_hip_w8.so; fi
