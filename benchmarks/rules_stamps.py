"""Where does k_wave_rules' time go on in-situ leaf batches?  Runs 4,096 boards for 300 four-playout plies on the
DIAGNOSTIC build of the library (tests/hip/libqzero_hip_rstamps.so: s_memtime / s_memrealtime stamps inside
k_wave_rules), then a few more playout steps, and reads the stamps of the LAST launch: per phase of a searching
wavefront (cycles), and where searching wavefronts and encoder tiles sit inside the launch (100 MHz real time)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_quoridor_amd import _cabi  # noqa: E402

_cabi.LIB_PATH = os.path.join(ROOT, "tests", "hip", "libqzero_hip_rstamps.so")  # before anything loads the library
from alphazero_quoridor_amd.engine import SelfPlayEngine  # noqa: E402
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(2026)
net = PolicyValueNet(use_gpu=True)
ev = net.evaluator("per_leaf")
eng = SelfPlayEngine(4096, n_playout=400, seed=1, device=dev)
for _ in range(300):
    eng.run_playouts(ev, 4)
    eng.finish_move()
    eng.harvest()
L = _cabi.load()
L.qzt_rules_stamps_read.restype = C.c_int
L.qzt_rules_stamps_read.argtypes = [C.c_void_p] * 3
names = ["load", "pre", "search", "convert", "post", "slots", "floods", "masks"]
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    for _ in range(5):
        eng.playout_step(ev, write_planes=True)  # like bench.py: the full op, planes included
    torch.cuda.synchronize()
    st = np.zeros((4096, 16), dtype=np.uint32)
    enc = np.zeros((512, 2), dtype=np.uint64)
    rt = np.zeros((4096, 2), dtype=np.uint64)
    assert L.qzt_rules_stamps_read(st.ctypes.data, enc.ctypes.data, rt.ctypes.data) == 0
    s = st[:, 15] == 1
    d = (st[s, 1:9].astype(np.int64) - st[s, 0:8].astype(np.int64)) & 0xFFFFFFFF
    tot = (st[s, 8].astype(np.int64) - st[s, 0].astype(np.int64)) & 0xFFFFFFFF
    t0 = min(int(rt[:, 0].min()), int(enc[:256, 0].min()))
    print("launch %d: %d searching wavefronts; s_memtime ticks per phase, mean (max):" % (rep, int(s.sum())))
    print("   " + "  ".join("%s %.0f (%d)" % (n, d[:, i].mean(), d[:, i].max()) for i, n in enumerate(names)))
    print("   total mean %.0f max %d ticks; path lengths mean %.1f max %d; flood items mean %.1f max %d" %
          (tot.mean(), tot.max(), st[s, 10:12].mean(), st[s, 10:12].max(), st[s, 12].mean(), st[s, 12].max()))
    w = int(np.argmax(tot))
    print("   slowest: " + "  ".join("%s %d" % (n, d[w, i]) for i, n in enumerate(names)) + "  L %d/%d items %d" % (st[s][w, 10], st[s][w, 11], st[s][w, 12]))
    rs, re_ = (rt[s, 0].astype(np.int64) - t0) / 100.0, (rt[s, 1].astype(np.int64) - t0) / 100.0
    es, ee = (enc[:256, 0].astype(np.int64) - t0) / 100.0, (enc[:256, 1].astype(np.int64) - t0) / 100.0
    allend = (rt[:, 1].astype(np.int64) - t0) / 100.0
    print("   real time (us from the first stamp): searching wavefronts start %.2f..%.2f end %.2f..%.2f (mean %.2f); encoder tiles start %.2f..%.2f end %.2f..%.2f; "
          "last short-cut wavefront ends %.2f" % (rs.min(), rs.max(), re_.min(), re_.max(), re_.mean(), es.min(), es.max(), ee.min(), ee.max(), allend[~s].max()))
