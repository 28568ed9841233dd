#!/bin/bash
# the sign-fixed mode (NOT the reference's mcts.py:125; labelled second line of the bench) at several board counts / budgets: the
# stationary games/s estimate (boards / E[wall time of a game], committed sign-fixed length sample) and the network's throughput
O=gpurun_out/${OUT:-r5fix}; mkdir -p $O
for c in ${CASES:-4096:500 8192:500 16384:500 16384:1000 32768:1000}; do
  IFS=: read boards budget <<< "$c"
  timeout 500 python bench.py --steps 8 --warmup 4 --boards $boards --budget-us $budget --fix-terminal-sign --length-file profiles/round5/game_length_400playouts_sign_fixed.json \
    --rounds-per-step 512 --settle-rounds 5120 --no-cpu-baseline --no-c3 > $O/fix_${boards}_${budget}.json 2> $O/fix_${boards}_${budget}.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/fix_${boards}_${budget}.json").read().strip().splitlines()[-1])
    ss=d["games_per_s_steady_state"]
    print("$boards/$budget", "stationary games/s", round(d["value"],1), "[", round(d.get("value_low") or 0,1), round(d.get("value_high") or 0,1), "] raw", round(d["games_in_timed_region_per_s"],1), "evals/s", round(d["nn_evaluations_per_s"]/1e6,2), "M plies/s", round(d["plies_per_s"]), "open share", round(ss["open_phase_share_of_a_game"],3), "ms/round", round(d["ms_per_round"],3), "nn us", round(d["roofline_nn"]["avg_launch_us"]), "leaves", round(d["roofline_nn"]["leaves_per_launch"]))
except Exception as e:
    print("$boards/$budget FAILED", e, open("$O/fix_${boards}_${budget}.err").read()[-600:])
PY
done
