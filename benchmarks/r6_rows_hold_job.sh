#!/bin/bash
# same-box A/B: k_rows' wavefronts placed BEFORE k_advance's (QZ_ADV_HOLD_US: k_advance's stream held back) x wavefronts in k_rows' grid x boards
#   CASES="name:boards:select_opts:rows_waves:hold_us ..."
O=gpurun_out/${OUT:-r6hold}; mkdir -p $O
for c in $CASES; do
  IFS=: read name boards so waves hold extra <<< "$c"
  QZ_ROWS_WAVES=$waves QZ_ADV_HOLD_US=$hold timeout 400 python bench.py --steps ${STEPS:-4} --warmup 2 --boards $boards --no-cpu-baseline --no-c3 --second-line-seconds 0 --select-opts $so ${extra//,/ } > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name", round(d["playouts_per_s"]/1e6,1), "M playouts/s; plies/s", round(d["plies_per_s"]), "ms/round", round(d["ms_per_round"],3), "advance us", round(d["roofline"]["avg_launch_us"]), "nn us", round(d["roofline_nn"]["avg_launch_us"]), "leaves/round", round(d["nn_evaluations_per_s"]*d["ms_per_round"]/1e3), "hit", round(d["memo_hit_rate"],4))
except Exception as e:
    print("$name FAILED", e, open("$O/bench_$name.err").read()[-600:])
PY
done
