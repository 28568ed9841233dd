#!/bin/bash
mkdir -p gpurun_out/rows
python -m pytest tests/test_gpu_rules.py -m gpu -q -x 2>&1 | tail -3 | tee gpurun_out/rows/pytest_rules3.log
python benchmarks/insitu_rules_timing.py 2>&1 | grep variant | tee gpurun_out/rows/insitu_nt.txt
python benchmarks/rules_stamps.py 2 2>&1 | grep -v Warning | tee gpurun_out/rows/rules_stamps3.txt
for v in 3 6; do
  python benchmarks/movegen_bench.py --boards 4096 --variant $v --launches 100 2>&1 | grep '^{' | tee -a gpurun_out/rows/movegen_b4096_nt.jsonl
done
for rep in 1 2; do
  QZ_BENCH_LIB=$PWD/tests/hip/libqzero_hip_prev.so python benchmarks/movegen_bench.py --launches 100 2>&1 | grep '^{' | sed 's/^{/{"lib": "prev", /' | tee -a gpurun_out/rows/c3_planjumps_ab.jsonl
  python benchmarks/movegen_bench.py --launches 100 2>&1 | grep '^{' | sed 's/^{/{"lib": "new", /' | tee -a gpurun_out/rows/c3_planjumps_ab.jsonl
done
