#!/bin/bash
# round-2 job M: input stage of the trunk launch: parity, then the bench
mkdir -p gpurun_out/r2m
python -m pytest tests/test_gpu_conv.py tests/test_gpu_api.py tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -12 | tee gpurun_out/r2m/pytest.log
python benchmarks/conv_bench.py --what fused,heads_staged 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2m/conv.log
STEPS=6 bash benchmarks/r2_job_j.sh
cp gpurun_out/r2j/bench.json gpurun_out/r2m/bench.json
