#!/bin/bash
# the game-length sample behind bench.py's steady-state games/s, on a window >= 2x round 3's: 1,024 boards played continuously
# (per-board speed decides how far a window reaches in a given time), dropped games recorded as right-censored observations
O=gpurun_out/${OUT:-r4len}; mkdir -p $O
make -C alphazero_quoridor_amd/csrc -s 2>&1 | grep -E "error"
timeout $(( ${SECONDS_RUN:-2700} + 300 )) python benchmarks/game_length.py --boards ${BOARDS:-1024} --playouts 400 --seconds ${SECONDS_RUN:-2700} --out $O/game_length_400playouts.json > $O/game_length.log 2> $O/game_length_progress.txt
tail -3 $O/game_length_progress.txt | cut -c1-300
python - <<PY
import json
d=json.load(open("$O/game_length_400playouts.json"))
print({k:d[k] for k in ("boards","seconds","plies_run","plies_per_s","games_finished","games_censored","games_dropped_as_censored","mean_plies_per_game","mean_ci95","restricted_mean","survival_at_T","T","window_doubling","finished_fraction_of_started","dropped_mean_ply")})
PY
