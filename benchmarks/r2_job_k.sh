#!/bin/bash
# round-2 job K: replay records of k_select: MCTS parity tests, the bench, the stamps
mkdir -p gpurun_out/r2k
python -m pytest tests/test_gpu_mcts.py tests/test_gpu_api.py tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r2k/pytest.log
tail -15 gpurun_out/r2k/pytest.log
STEPS=6 bash benchmarks/r2_job_j.sh
cp gpurun_out/r2j/bench.json gpurun_out/r2k/bench.json
python benchmarks/select_stamps.py 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2k/select_stamps.txt
