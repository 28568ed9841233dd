#!/bin/bash
# k_select work: MCTS parity tests, a short bench, the s_memtime stamps of the diagnostic build
mkdir -p gpurun_out/r2k
python -m pytest tests/test_gpu_mcts.py tests/test_gpu_api.py tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r2k/pytest.log
tail -15 gpurun_out/r2k/pytest.log
STEPS=6 bash benchmarks/r2_bench_short_job.sh
cp gpurun_out/r2j/bench.json gpurun_out/r2k/bench.json
python benchmarks/select_stamps.py 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2k/select_stamps.txt
