#!/bin/bash
# quick check of pending kernel changes: conv parity + timing, MCTS parity (incl. records-vs-walk), short bench
mkdir -p gpurun_out/r2c
python -m pytest tests/test_gpu_conv.py tests/test_gpu_mcts.py tests/test_gpu_determinism.py -m gpu -q -x -s 2>&1 | grep -E "passed|failed|Error|assert|deepest" | tee gpurun_out/r2c/pytest.log
python benchmarks/conv_bench.py --what fused,heads_staged 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c/conv.log
python benchmarks/trunk_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c/trunk_stamps.txt
STEPS=6 bash benchmarks/r2_bench_short_job.sh
