#!/bin/bash
# round-2 job P: rules op after batching the corner-table loads: parity, then timings
mkdir -p gpurun_out/r2p
timeout 900 python -m pytest tests/test_gpu_rules.py -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/r2p/pytest.log
timeout 300 python benchmarks/movegen_bench.py 2>/dev/null | cut -c1-200 | tee gpurun_out/r2p/movegen_c3.jsonl
timeout 300 python benchmarks/movegen_bench.py --boards 4096 2>/dev/null | cut -c1-200 | tee gpurun_out/r2p/movegen_b4096.jsonl
STEPS=4 bash benchmarks/r2_job_j.sh 2>&1 | tail -6
python - <<PY
import json
d=json.loads(open("gpurun_out/r2j/bench.json").read().strip().splitlines()[-1])
print("in situ rules op:", d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
PY
