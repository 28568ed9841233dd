#!/bin/bash
# round-2 job L: trunk kernel timing + parity
mkdir -p gpurun_out/r2l
python benchmarks/conv_bench.py --what fused,heads_staged 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2l/conv.log
python -m pytest tests/test_gpu_conv.py -m gpu -q -x 2>&1 | tail -4 | tee -a gpurun_out/r2l/conv.log
