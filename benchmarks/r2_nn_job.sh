#!/bin/bash
mkdir -p gpurun_out/nn
python -m pytest tests/test_gpu_conv.py tests/test_gpu_api.py tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/nn/pytest.log
python benchmarks/conv_bench.py --what heads_staged --iters 30 2>&1 | grep 'head stage' | tee gpurun_out/nn/conv_heads.txt
STEPS=6 bash benchmarks/r2_bench_short_job.sh 2>&1 | tee gpurun_out/nn/bench_short.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/nn/prof -- /usr/bin/python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --desync-plies 300 --no-cpu-baseline --no-c3 > $GRAFT_REPO_ROOT/gpurun_out/nn/prof.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/nn/prof -name '*kernel_trace.csv' -delete
head -8 $(find $GRAFT_REPO_ROOT/gpurun_out/nn/prof -name '*kernel_stats.csv' | head -1) | cut -c1-160
