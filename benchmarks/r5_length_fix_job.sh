#!/bin/bash
# the game-length sample of the SIGN-FIXED mode (a won position backed up as +1, NOT the reference's mcts.py:125) behind the
# stationary estimate of bench.py's second line: BOARDS boards played continuously, Kaplan-Meier + exponential tail as for the headline
O=gpurun_out/${OUT:-r5lenfix}; mkdir -p $O
make -C alphazero_quoridor_amd/csrc -s 2>&1 | grep -E "error"
timeout $(( ${SECONDS_RUN:-420} + 200 )) python benchmarks/game_length.py --boards ${BOARDS:-2048} --playouts 400 --fix-sign 1 --budget-us ${BUDGET:-500} --seconds ${SECONDS_RUN:-420} --out $O/game_length_400playouts_sign_fixed.json > $O/game_length.log 2> $O/game_length_progress.txt
tail -3 $O/game_length_progress.txt | cut -c1-300
python - <<PY
import json
d=json.load(open("$O/game_length_400playouts_sign_fixed.json"))
print({k:d.get(k) for k in ("boards","seconds","plies_run","plies_per_s","games_finished","games_censored","games_dropped_as_censored","mean_plies_per_game","mean_open_plies_per_game","mean_ci95","restricted_mean","survival_at_T","T","window_doubling","finished_fraction_of_started","dropped_mean_ply")})
PY
