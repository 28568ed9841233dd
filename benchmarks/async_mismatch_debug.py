"""first mismatch between lock-step and asynchronous games (debug helper of tests/test_gpu_async.py)"""
import os, sys, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _stubs import det_fill_state_dict
from alphazero_quoridor_amd.engine import SelfPlayEngine
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet
dev = torch.device("cuda:0")
net = PolicyValueNet(use_gpu=True, device=dev)
net.policy_value_net.load_state_dict(det_fill_state_dict(net.policy_value_net.state_dict(), 7))
ev = net.evaluator("per_leaf")
B, NP = 192, 24
CE = int(os.environ.get("COMPACT", 0)); MEMO = int(os.environ.get("MEMO", 1))
lock = SelfPlayEngine(B, n_playout=NP, seed=77, device=dev, fix_terminal_sign=True)
asyn = SelfPlayEngine(B, n_playout=NP, seed=77, device=dev, fix_terminal_sign=True, compact_edges=CE, memo=bool(MEMO), tree_pool_pages=int(os.environ.get("POOL", 0)))
def games(batches):
    out = {}
    for tb in batches:
        gid = tb.game.cpu().numpy(); slot = tb.slot.cpu().numpy(); packed = tb.boards.to_packed(); pi = tb.pi.cpu().numpy(); z = tb.z.cpu().numpy()
        for g in range(tb.n_games):
            sel = gid == g
            out.setdefault(int(slot[g]), []).append((packed[sel], pi[sel], z[sel]))
    return out
lb, ab = [], []
for _ in range(200):
    lock.play_ply(ev); tb = lock.harvest()
    if tb is not None: lb.append(tb)
n_lock = sum(t.n_games for t in lb)
r = 0
while sum(t.n_games for t in ab) < n_lock and r < 20000:
    asyn.run_rounds(ev, 8, max_playouts=NP + 8); r += 8
    tb = asyn.harvest()
    if tb is not None: ab.append(tb)
gl, ga = games(lb), games(ab)
bad = 0
for slot in sorted(gl):
    for i, (g1, g2) in enumerate(zip(gl[slot], ga.get(slot, []))):
        if g1[0].tobytes() == g2[0].tobytes() and g1[1].tobytes() == g2[1].tobytes() and g1[2].tobytes() == g2[2].tobytes():
            continue
        bad += 1
        if bad <= 4:
            n = min(len(g1[0]), len(g2[0]))
            for t in range(n):
                if g1[0][t].tobytes() != g2[0][t].tobytes() or g1[1][t].tobytes() != g2[1][t].tobytes():
                    print("slot", slot, "game", i, "len", len(g1[0]), len(g2[0]), "first differing ply", t, "boards equal", g1[0][t].tobytes() == g2[0][t].tobytes())
                    nz = np.nonzero((g1[1][t] != g2[1][t]))[0]
                    print("  board", g1[0][t], "differing pi entries", len(nz), [(int(a), float(g1[1][t][a]), float(g2[1][t][a])) for a in nz[:6]])
                    break
            else:
                print("slot", slot, "game", i, "prefix equal, lengths", len(g1[0]), len(g2[0]))
        break
print("games compared", sum(min(len(gl[s]), len(ga.get(s, []))) for s in gl), "slots with a mismatch", bad, "async stats", {k: v for k, v in asyn.stats().items() if k in ("node_overflow", "games_aborted", "runaway_descents", "max_edges", "max_depth")})
