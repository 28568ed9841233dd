#!/bin/bash
# the second line (terminal sign fixed) as bench.py's own --fix-terminal-sign headline, for several k_advance budgets: real finished games / wall time
O=gpurun_out/${OUT:-r4sec}; mkdir -p $O
make -C alphazero_quoridor_amd/csrc -s 2>&1 | grep -E "error"
for bud in 1000 500 300 150; do
  timeout 300 python - <<PY
import json, subprocess, sys
sys.argv = ["x"]
import importlib.util, os
spec = importlib.util.spec_from_file_location("bench", "bench.py"); bench = importlib.util.module_from_spec(spec); sys.modules["bench"] = bench; spec.loader.exec_module(bench)
import argparse, torch
from alphazero_quoridor_amd import dist as qdist
a = argparse.Namespace(seed=2026, boards=${BOARDS:-8192}, groups=1, playouts=400, max_playouts=4096, budget_us=1000, second_line_budget_us=$bud, desync_plies=700, desync_playouts=4,
                       second_line_warm_seconds=${WARM:-15.0}, second_line_seconds=${SECS:-6.0}, nn_dtype="fp32", bn="per_leaf", channels_last=1, library_trunk=False, select_opts=0, no_memo=False, max_depth=992)
r = bench.second_line(a, torch.device("cuda:0"), qdist)
print(json.dumps({k: r[k] for k in ("budget_us", "value", "games_finished", "seconds", "mean_plies_per_game", "plies_per_s", "playouts_per_s", "nn_evaluations_per_s", "memo_hit_rate", "board_seconds_per_open_ply", "nn_evaluations_per_game")}))
PY
done 2>&1 | grep "^{" | tee $O/second_line_budget_sweep.jsonl
