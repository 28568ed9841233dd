#!/bin/bash
# base paths on nine lanes per player (qz_path_rows.h): parity of every rules-kernel variant, then A/B against the
# one-lane search (variant 5) on the synthetic sets at 4,096 boards and on in-situ leaf batches
mkdir -p gpurun_out/rows
python -m pytest tests/test_gpu_rules.py -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/rows/pytest_rules.log
for v in 3 5; do
  python benchmarks/movegen_bench.py --boards 4096 --variant $v --launches 100 2>&1 | grep '^{' | tee -a gpurun_out/rows/movegen_b4096.jsonl
done
python benchmarks/insitu_rules_timing.py 2>&1 | grep variant | tee gpurun_out/rows/insitu.txt
