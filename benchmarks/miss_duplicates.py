"""How many leaves of a round's miss list are DUPLICATES of each other (the same 24-byte board missed by several boards in the same
round)?  The memo removes repeats across rounds; inside a round every board sends its own copy to the network.  Bench regime:
desync with 4-playout games, then 400-playout rounds."""
import json, os, sys
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from alphazero_quoridor_amd import _cabi
from alphazero_quoridor_amd.engine import SelfPlayEngine
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet
import ctypes as C
B = int(os.environ.get("BOARDS", 8192)); dev = torch.device("cuda:0"); torch.manual_seed(2026)
ev = PolicyValueNet(use_gpu=True).evaluator("per_leaf")
eng = SelfPlayEngine(B, n_playout=400, seed=3, device=dev, max_depth=992)
kw = dict(max_playouts=4096, budget_us=1000)
eng.set_playouts(4)
for _ in range(0, 700 * 5, 64):
    eng.run_rounds(ev, 64, **kw); eng.harvest()
eng.set_playouts(400)
out = []
for block in range(int(os.environ.get("BLOCKS", 6))):
    eng.run_rounds(ev, 512, **kw); eng.harvest()
    tot = dist = open_n = 0
    for _ in range(8):
        eng._memo_guard(ev)
        L = eng.L
        _cabi.check(L.qz_selfplay_advance(eng.h, 4096, 1000, 1, eng._s()))
        _cabi.check(L.qz_selfplay_leaf_rules(eng.h, eng._s()))
        _cabi.check(L.qz_selfplay_evaluate(eng.h, C.byref(ev.nn_weights()), eng._s()))
        packed, mask, p, v = eng.misses()
        _cabi.check(L.qz_selfplay_round_tail(eng.h, eng._s()))
        keys = set(x.tobytes() for x in packed)
        tot += len(packed); dist += len(keys)
        open_n += int(np.sum(np.where(packed["cur"] == 1, packed["w1"], packed["w2"]) > 0))
    st = eng.stats()
    out.append({"rounds": st["rounds"], "leaves_per_round": tot / 8, "distinct_per_round": dist / 8, "duplicate_frac": 1 - dist / max(tot, 1), "leaf_mover_has_walls_frac": open_n / max(tot, 1)})
    print(json.dumps(out[-1]), flush=True)
