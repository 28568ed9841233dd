"""Where does k_select's time go?  Runs the bench workload (4,096 boards, 400 playouts, desynchronised)
on the DIAGNOSTIC build of the library (tests/hip/libqzero_hip_stamps.so: k_select with s_memtime
stamps) and prints, for the last 64 launches, the slowest descent of each launch and what it did."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_quoridor_amd import _cabi  # noqa: E402

_cabi.LIB_PATH = os.path.join(ROOT, "tests", "hip", "libqzero_hip_stamps.so")  # before anything loads the library
from alphazero_quoridor_amd.engine import BoardGroups  # noqa: E402
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet  # noqa: E402

plies = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
torch.manual_seed(2026)
net = PolicyValueNet(use_gpu=True, device=dev)
eng = BoardGroups(4096, 1, lambda: net.evaluator("per_leaf", torch.float32, True), seed=2026, device=dev, n_playout=400, c_puct=5, temp=1.0,
                  is_selfplay=1)
for _ in range(700):
    eng.run_playouts(4)
    eng.finish_move()
    eng.harvest()
for _ in range(plies):
    eng.run_playouts()
    eng.finish_move()
    eng.harvest()
eng.run_playouts(320)  # stop inside a ply: the last 64 launches are ordinary playouts
torch.cuda.synchronize()
L = _cabi.load()
L.qzt_select_stamps_read.restype = C.c_int
L.qzt_select_stamps_read.argtypes = [C.c_void_p]
buf = np.zeros((64, 4096, 8), dtype=np.uint32)
assert L.qzt_select_stamps_read(buf.ctypes.data) == 0
tot = buf[:, :, 0].astype(np.int64)
names = ["total", "replay", "walk", "rounds", "walk_narrow", "walk_wide", "plen", "replayed"]
print("per launch: the slowest descent (s_memtime ticks) and what it did")
worst = tot.argmax(axis=1)
rows = buf[np.arange(64), worst]
for k in np.argsort(-rows[:, 0])[:12]:
    print("  launch %2d board %4d: " % (k, worst[k]) + "  ".join("%s %d" % (n, v) for n, v in zip(names, rows[k])))
print("mean over launches of the slowest: " + "  ".join("%s %.0f" % (n, v) for n, v in zip(names, rows.mean(axis=0))))
print("all descents: mean total %.0f ticks, mean plen %.1f; p99 total %.0f" % (tot.mean(), buf[:, :, 6].mean(), np.percentile(tot, 99)))
# per-unit costs from a regression over the slowest decile
sel = tot > np.percentile(tot, 90)
X = np.stack([buf[:, :, 3][sel], buf[:, :, 4][sel], buf[:, :, 5][sel], np.ones(sel.sum())], axis=1).astype(np.float64)
coef, *_ = np.linalg.lstsq(X, tot[sel].astype(np.float64), rcond=None)
print("ticks ~ %.0f per replay round + %.0f per narrow walk level + %.0f per wide walk level + %.0f" % tuple(coef))
# the board that is the slowest most often: its 64 consecutive descents (slot = playout counter & 63; the run stopped after
# playout 320 of a ply, so slots 0..63 are playouts 256..319 in order)
vals, counts = np.unique(worst, return_counts=True)
bsel = int(vals[counts.argmax()])
print("board %d, descents of playouts 256..319: (plen, replayed, rounds, walked narrow)" % bsel)
print(" ".join("(%d,%d,%d,%d)" % (buf[k, bsel, 6], buf[k, bsel, 7], buf[k, bsel, 3], buf[k, bsel, 4]) for k in range(64)))
