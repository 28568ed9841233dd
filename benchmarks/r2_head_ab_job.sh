#!/bin/bash
# same-box A/B of a change of the evaluation kernels against tests/hip/libqzero_hip_prev.so (the previous build, made by hand from HEAD)
mkdir -p gpurun_out/nn
for rep in 1 2 3 4; do
  echo "prev: $(QZ_BENCH_LIB=$PWD/tests/hip/libqzero_hip_prev.so python benchmarks/conv_bench.py --what heads_staged --iters 40 2>&1 | grep 'head stage')" | tee -a gpurun_out/nn/head_ksplit_ab.txt
  echo "new:  $(python benchmarks/conv_bench.py --what heads_staged --iters 40 2>&1 | grep 'head stage')" | tee -a gpurun_out/nn/head_ksplit_ab.txt
done
