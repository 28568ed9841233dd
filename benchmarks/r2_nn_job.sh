#!/bin/bash
# after a change of the evaluation kernels: parity tests, trunk + heads timing, short bench
mkdir -p gpurun_out/nn
python -m pytest tests/test_gpu_conv.py tests/test_gpu_api.py tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/nn/pytest.log
for rep in 1 2 3; do python benchmarks/conv_bench.py --what heads_staged --iters 30 2>&1 | grep 'head stage' | tee -a gpurun_out/nn/conv_heads.txt; done
STEPS=6 bash benchmarks/r2_bench_short_job.sh 2>&1 | tee gpurun_out/nn/bench_short.txt
