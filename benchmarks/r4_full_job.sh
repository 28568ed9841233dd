#!/bin/bash
# the whole GPU tier, then the default bench line exactly as the driver runs it (timed), then smoke()
O=gpurun_out/${OUT:-r4full}; mkdir -p $O
make -C alphazero_quoridor_amd/csrc -s 2>&1 | grep -E "error"; make -C tests/hip -s 2>&1 | grep -E "error"; make -C oracle -s 2>&1 | grep -E "error"
ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | tail -15 | tee $O/pytest_gpu.log
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
echo "bench wall seconds: $(( $(date +%s) - t0 ))" | tee $O/bench_wall.txt
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("value","value_low","value_high","plies_per_s","playouts_per_s","ms_per_step","length_window_doubling_delta")})
print("roofline", d["roofline"]["frac"], d["roofline"]["avg_launch_us"], "c3", d.get("roofline_c3",{}).get("frac"), d.get("roofline_c3",{}).get("traffic_over_algorithmic"))
print("second", d.get("second_line_fix_terminal_sign"))
print("cpu", {k:d["cpu_baseline"].get(k) for k in ("value","cores","playouts_per_s_allcores")} if d.get("cpu_baseline") else None)
PY
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
