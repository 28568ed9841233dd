#!/bin/bash
mkdir -p gpurun_out/rows
python -m pytest tests/test_gpu_rules.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/rows/pytest_rules4.log
python benchmarks/insitu_rules_timing.py 2>&1 | grep variant | tee gpurun_out/rows/insitu4.txt
python benchmarks/rules_stamps.py 2 2>&1 | grep -v Warning | tee gpurun_out/rows/rules_stamps4.txt
for v in 3 5; do
  python benchmarks/movegen_bench.py --boards 4096 --variant $v --launches 100 2>&1 | grep '^{' | tee -a gpurun_out/rows/movegen_b4096_4.jsonl
done
