#!/bin/bash
# k_wave_rules after a change: parity of every rules-kernel variant, in-situ timing of variants 0 / 6 / 5, per-phase stamps, synthetic sets at 4,096 boards
mkdir -p gpurun_out/rows
python -m pytest tests/test_gpu_rules.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/rows/pytest_rules.log
python benchmarks/insitu_rules_timing.py 2>&1 | grep variant | tee gpurun_out/rows/insitu.txt
python benchmarks/rules_stamps.py 2 2>&1 | grep -v Warning | tee gpurun_out/rows/rules_stamps.txt
for v in 3 5; do
  python benchmarks/movegen_bench.py --boards 4096 --variant $v --launches 100 2>&1 | grep '^{' | tee -a gpurun_out/rows/movegen_b4096.jsonl
done
