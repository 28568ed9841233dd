#!/bin/bash
# round 4, first GPU call: the new direct-oracle tests of the asynchronous loop, the no-legal-move roots for the fixture, and a
# baseline bench line of the unchanged kernels on this box
O=gpurun_out/${OUT:-r4a}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_async_oracle.py -m gpu -x -q --timeout=600 -s 2>&1 | tail -25 | tee $O/pytest_async_oracle.log
timeout 200 python benchmarks/capture_no_move_roots.py --seconds 90 --out $O/no_move_roots_input.npy 2>&1 | tail -3 | tee $O/capture.log
timeout 300 python bench.py --steps 6 --warmup 2 > $O/bench_baseline.json 2> $O/bench_baseline.err; tail -c 600 $O/bench_baseline.err; python - <<PY
import json
d=json.loads(open("$O/bench_baseline.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","plies_per_s","playouts_per_s","ms_per_step")}, d["roofline"]["frac"], d.get("roofline_c3",{}).get("frac"))
PY
