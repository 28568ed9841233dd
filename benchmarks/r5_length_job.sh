#!/bin/bash
# the game-length sample behind bench.py's steady-state games/s (reference-faithful mode) once more on an independent, longer run:
# 1,024 boards played continuously, dropped games as right-censored observations (round 4: 45 minutes, T = 408,516 plies)
O=gpurun_out/${OUT:-r5len}; mkdir -p $O
make -C alphazero_quoridor_amd/csrc -s 2>&1 | grep -E "error"
timeout $(( ${SECONDS_RUN:-4500} + 300 )) python benchmarks/game_length.py --boards ${BOARDS:-1024} --playouts 400 --seconds ${SECONDS_RUN:-4500} --out $O/game_length_400playouts.json > $O/game_length.log 2> $O/game_length_progress.txt
tail -2 $O/game_length_progress.txt | cut -c1-300
python - <<PY
import json
d=json.load(open("$O/game_length_400playouts.json"))
print({k:d[k] for k in ("boards","seconds","plies_run","plies_per_s","games_finished","games_censored","games_dropped_as_censored","mean_plies_per_game","mean_ci95","restricted_mean","survival_at_T","T","window_doubling","finished_fraction_of_started","dropped_mean_ply")})
PY
