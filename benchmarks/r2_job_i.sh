#!/bin/bash
# round-2 job I: head stage of the trunk launch: parity + timing
mkdir -p gpurun_out/r2i
python -m pytest tests/test_gpu_conv.py tests/test_gpu_determinism.py -m gpu -q -x -s 2>&1 | tail -25 > gpurun_out/r2i/pytest.log
tail -25 gpurun_out/r2i/pytest.log
python benchmarks/conv_bench.py 2>&1 | tail -30 | tee gpurun_out/r2i/conv_bench.log
