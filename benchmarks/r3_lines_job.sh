#!/bin/bash
# The A/B and labelled second lines of profiles/round3/SUMMARY.md section 1, one bench.py run each (no CPU baseline, no C3 leg)
O=gpurun_out/r3lines; mkdir -p $O
run() { n=$1; shift; timeout 400 python bench.py --no-cpu-baseline --no-c3 "$@" > $O/$n.json 2> $O/$n.err; python - <<PY
import json
try:
    d = json.load(open("$O/$n.json")); print("$n", round(d["plies_per_s"]), round(d["playouts_per_s"] / 1e6, 1), d.get("games_in_timed_region"), round(d["value"], 2), round(d["ms_per_step"] * d["steps"] / 1e3, 1))
except Exception as e:
    print("$n FAILED", e)
PY
}
run bench_no_memo --steps 6 --no-memo
run bench_lockstep --mode lockstep --steps 3
run bench_groups2 --steps 6 --groups 2
run bench_no_depth_limit --steps 6 --max-depth 0
run bench_c5_playouts800_1gpu --steps 6 --playouts 800
run bench_c2_playouts100 --steps 6 --playouts 100
run bench_fix_terminal_sign --steps 10 --fix-terminal-sign
run bench_fix_terminal_sign_fp16_NON_PARITY --steps 10 --fix-terminal-sign --nn-dtype fp16
run bench_fp16_throughput_mode_NON_PARITY --steps 6 --nn-dtype fp16
run bench_boards8192 --steps 6 --boards 8192
