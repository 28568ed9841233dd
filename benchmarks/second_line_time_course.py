"""How long the second line (terminal sign fixed: bench.py's second_line) takes to become stationary: the same engine and phases, then
finished games, plies and network evaluations per window of WINDOW seconds for TOTAL seconds.  -> JSON on stdout.
Env: BOARDS (4096), BUDGET (500 us), WINDOW (10), TOTAL (300)."""
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["x"]
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
sys.modules["bench"] = bench
spec.loader.exec_module(bench)
import argparse  # noqa: E402

from alphazero_quoridor_amd import dist as qdist  # noqa: E402
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet  # noqa: E402

B = int(os.environ.get("BOARDS", 4096)); BUD = int(os.environ.get("BUDGET", 500)); WIN = float(os.environ.get("WINDOW", 10)); TOT = float(os.environ.get("TOTAL", 300))
a = argparse.Namespace(seed=2026, boards=B, groups=1, playouts=400, max_playouts=4096, budget_us=BUD, desync_plies=700, desync_playouts=4, nn_dtype="fp32", bn="per_leaf",
                       channels_last=1, library_trunk=False, select_opts=0, no_memo=False, max_depth=992)
dev = torch.device("cuda:0")
torch.manual_seed(a.seed)
net = PolicyValueNet(use_gpu=True, device=dev)
eng = bench.make_engine(a, net, dev, qdist.shard_seed(a.seed + 1, 0), True, boards=B)
kw = dict(max_playouts=a.max_playouts, budget_us=BUD)
lens = []


def harvest():
    n = 0
    for tb in eng.harvest():
        n += tb.n_games
        lens.extend(torch.bincount(tb.game.long(), minlength=tb.n_games).tolist())
    return n


eng.set_playouts(a.desync_playouts)
for _ in range(0, a.desync_plies * (a.desync_playouts + 1), 64):
    eng.run_rounds(64, **kw)
    harvest()
eng.set_playouts(a.playouts)
rows = []
t_start = time.perf_counter()
while time.perf_counter() - t_start < TOT:
    del lens[:]
    st0 = eng.stats()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    games = 0
    while time.perf_counter() - t0 < WIN:
        eng.run_rounds(64, **kw)
        games += harvest()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    st1 = eng.stats()
    d = {k: st1[k] - st0[k] for k in st1}
    rows.append({"t_end_s": round(time.perf_counter() - t_start, 1), "games_per_s": games / dt, "mean_plies_of_the_finished": float(np.mean(lens)) if lens else None,
                 "plies_per_s": d["plies_played"] / dt, "nn_evaluations_per_s": d["nn_evals"] / dt, "memo_hit_rate": d["memo_hits"] / max(d["playouts"], 1),
                 "open_round_share": d["open_rounds"] / max(d["rounds"] * B, 1)})
    sys.stderr.write(json.dumps(rows[-1]) + "\n")
eng.close()
print(json.dumps({"boards": B, "budget_us": BUD, "n_playout": 400, "fix_terminal_sign": True, "window_s": WIN, "windows": rows}))
