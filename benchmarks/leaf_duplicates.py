"""How often does a search evaluate a leaf board it (or an earlier search of the same game) has
already evaluated?  VERDICT r2 item 4: policy_value_fn on a batch of one is a pure function of
the 24-byte board, so a repeat is a recomputation.

Bench-like population (4,096 boards desynchronised with 700 four-playout plies), then PLIES plies
at 400 playouts; every playout step the leaf boards of the first SAMPLE boards are logged on the
device.  On the host, per sampled board: the fraction of non-terminal leaves whose 24-byte board
was seen earlier in the log (a cold per-game memo: the first logged ply has no history, so the
figure is a LOWER bound of the steady-state hit rate), by ply and by whether the mover has walls.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from alphazero_quoridor_amd.engine import SelfPlayEngine
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

BOARDS = int(os.environ.get("BOARDS", 4096))
SAMPLE = int(os.environ.get("SAMPLE", 256))
PLIES = int(os.environ.get("PLIES", 12))
PLAYOUTS = int(os.environ.get("PLAYOUTS", 400))
DESYNC = int(os.environ.get("DESYNC", 700))
OUT = os.environ.get("OUT", "gpurun_out/leaf_duplicates.json")

dev = torch.device("cuda:0")
torch.manual_seed(2026)
net = PolicyValueNet(use_gpu=True)
ev = net.evaluator("per_leaf")
eng = SelfPlayEngine(BOARDS, n_playout=PLAYOUTS, seed=2026, device=dev)
for _ in range(DESYNC):
    eng.run_playouts(ev, 4)
    eng.finish_move()
    eng.harvest()

st, term_ptr, n = eng.leaf_ref()
import ctypes as C


hip = C.CDLL("libamdhip64.so")
log = torch.zeros((PLIES * PLAYOUTS, 4, SAMPLE), dtype=torch.int64, device=dev)  # hb, vb, meta, term


def snapshot(k):
    s = torch.cuda.current_stream(dev).cuda_stream
    for j, p in enumerate((st.hbits, st.vbits, st.meta)):
        assert hip.hipMemcpyAsync(C.c_void_p(log[k, j].data_ptr()), C.c_void_p(p), C.c_size_t(SAMPLE * 8), 3, C.c_void_p(s)) == 0
    tb = torch.empty(SAMPLE, dtype=torch.uint8, device=dev)
    assert hip.hipMemcpyAsync(C.c_void_p(tb.data_ptr()), C.c_void_p(term_ptr), C.c_size_t(SAMPLE), 3, C.c_void_p(s)) == 0
    log[k, 3] = tb.to(torch.int64)


rb = eng.get_boards().meta.cpu().numpy()
rw1, rw2, rcur = (rb >> 16) & 0xFF, (rb >> 24) & 0xFF, (rb >> 32) & 0xFF
root_pop = {"root_mover_has_walls_frac": float(np.mean(np.where(rcur == 1, rw1, rw2) > 0)), "any_walls_left_frac": float(np.mean((rw1 + rw2) > 0)),
            "walls_left_mean": float(np.mean(rw1 + rw2))}
serial0 = eng.stats()["games_finished"]
k = 0
for ply in range(PLIES):
    for j in range(PLAYOUTS):
        eng.select(want_planes=False)
        snapshot(k)
        k += 1
        p, v = ev(None, leaf=eng.leaf_ref())
        eng.expand_backup(p, v, then_descend=False)
    eng.finish_move()
    eng.harvest()
torch.cuda.synchronize()
L = log.cpu().numpy()  # [T][4][SAMPLE]
T = L.shape[0]
res = {"boards": BOARDS, "sample": SAMPLE, "plies": PLIES, "playouts": PLAYOUTS, "desync_plies": DESYNC,
       "games_finished_during_log": eng.stats()["games_finished"] - serial0, "root_population_after_desync": root_pop}
hits_by_ply = np.zeros(PLIES)
n_by_ply = np.zeros(PLIES)
hits_w = [0, 0]
n_w = [0, 0]
distinct = []
for b in range(SAMPLE):
    seen = set()
    for t in range(T):
        if L[t, 3, b] != 0:
            continue  # terminal leaf / idle board: no evaluation
        key = (int(L[t, 0, b]), int(L[t, 1, b]), int(L[t, 2, b]))
        meta = key[2]
        w1, w2, cur = (meta >> 16) & 0xFF, (meta >> 24) & 0xFF, (meta >> 32) & 0xFF
        has_walls = 1 if (w1 if cur == 1 else w2) > 0 else 0
        ply = t // PLAYOUTS
        hit = key in seen
        seen.add(key)
        hits_by_ply[ply] += hit
        n_by_ply[ply] += 1
        hits_w[has_walls] += hit
        n_w[has_walls] += 1
    distinct.append(len(seen))
res["hit_rate_by_ply"] = (hits_by_ply / np.maximum(n_by_ply, 1)).round(4).tolist()
res["hit_rate_overall"] = float(hits_by_ply.sum() / max(n_by_ply.sum(), 1))
res["hit_rate_last_ply"] = float(hits_by_ply[-1] / max(n_by_ply[-1], 1))
res["mover_without_walls"] = {"evaluations": int(n_w[0]), "hit_rate": float(hits_w[0] / max(n_w[0], 1))}
res["mover_with_walls"] = {"evaluations": int(n_w[1]), "hit_rate": float(hits_w[1] / max(n_w[1], 1))}
res["distinct_boards_per_sampled_game"] = {"mean": float(np.mean(distinct)), "max": int(np.max(distinct)), "evaluations_per_game": PLIES * PLAYOUTS}
# per-board hit rate in the last ply, split by the phase of the ROOT (mover has walls or not at the ply's first leaf)
per_board_last = []
for b in range(SAMPLE):
    seen = set()
    h = n_ = 0
    for t in range(T):
        if L[t, 3, b] != 0:
            continue
        key = (int(L[t, 0, b]), int(L[t, 1, b]), int(L[t, 2, b]))
        if t // PLAYOUTS == PLIES - 1:
            h += key in seen
            n_ += 1
        seen.add(key)
    if n_:
        per_board_last.append(h / n_)
q = np.quantile(per_board_last, [0.0, 0.05, 0.25, 0.5, 0.75, 0.95, 1.0]).round(4).tolist()
res["per_board_hit_rate_last_ply_quantiles_0_5_25_50_75_95_100"] = q
res["boards_with_hit_rate_below_half_last_ply"] = int(np.sum(np.array(per_board_last) < 0.5))
os.makedirs(os.path.dirname(OUT) or ".", exist_ok=True)
with open(OUT, "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res))
