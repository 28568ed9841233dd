#!/bin/bash
# the new direct-oracle tests of the asynchronous loop + the whole GPU tier (first failure of each file stops it)
O=gpurun_out/${OUT:-r4b}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_async_oracle.py -m gpu -q --timeout=600 -s 2>&1 | tail -25 | tee $O/pytest_async_oracle.log
timeout 1200 python -m pytest tests -m gpu -x -q --timeout=900 --deselect tests/test_gpu_async_oracle.py 2>&1 | tail -8 | tee $O/pytest_gpu.log
