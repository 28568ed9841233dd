#!/bin/bash
# quick check of pending kernel changes: MCTS / API parity, determinism, short bench with and without the fused descent
mkdir -p gpurun_out/r2c
python -m pytest tests/test_gpu_mcts.py tests/test_gpu_api.py tests/test_gpu_determinism.py tests/test_gpu_train.py -m gpu -q -x -s 2>&1 | grep -E "passed|failed|Error|assert|deepest" | tee gpurun_out/r2c/pytest.log
STEPS=6 bash benchmarks/r2_bench_short_job.sh
STEPS=6 EXTRA=--separate-descent bash benchmarks/r2_bench_short_job.sh
