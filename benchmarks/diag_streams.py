"""Diagnostic: which stage of the evaluator is not stream-safe?  Runs the evaluator on 3 streams at
once, keeps every intermediate tensor, and reports the first stage whose output differs from the
single-stream run."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from _stubs import det_fill_state_dict
from synth import synth_positions
from alphazero_quoridor_amd.boards import DeviceBoards
from alphazero_quoridor_amd.policy_value_net import LeafEvaluator, PolicyValueNet

dev = torch.device("cuda:0")
pvn = PolicyValueNet(use_gpu=True, device=dev)
pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), 2024))
sizes = [int(a) for a in sys.argv[1:]] or [32, 39, 46]
dbs = [DeviceBoards.from_packed(synth_positions(n, seed=40 + i), dev) for i, n in enumerate(sizes)]

class Tap(LeafEvaluator):
    def __init__(self, *a, **k):
        super().__init__(*a, **k); self.taps = []
    def _first_layer_from_boards(self, leaf):
        x = super()._first_layer_from_boards(leaf); self.taps.append(("input_layer", x.clone())); return x
    def _trunk_mfma(self, x):
        x = super()._trunk_mfma(x); self.taps.append(("trunk", x.clone())); return x

for name, kw in (("fused", {}), ("layered", {"fused_trunk": False})):
    ev = Tap(pvn.policy_value_net, "per_leaf", channels_last=True, **kw)
    torch.cuda.synchronize()
    ref = []
    for db in dbs:
        ev.taps = []
        p, v = ev(None, leaf=(db.struct(), 0, db.n))
        ref.append(ev.taps + [("p", p), ("v", v)])
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in dbs]
    first_bad = {}
    for rep in range(60):
        got = []
        for db, st in zip(dbs, streams):
            with torch.cuda.stream(st):
                ev.taps = []
                p, v = ev(None, leaf=(db.struct(), 0, db.n))
                got.append(ev.taps + [("p", p), ("v", v)])
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(ref, got)):
            for (sa, ta), (sb, tb) in zip(a, b):
                if not torch.equal(ta, tb):
                    d = (ta - tb).abs()
                    nz = d.reshape(d.shape[0], -1).amax(dim=1).nonzero().flatten().tolist()
                    first_bad.setdefault(sa, []).append((rep, i, float(d.max()), nz[:6], len(nz)))
                    break
    print(name, "first differing stage -> (rep, input, max|d|, leaves, n_leaves):", {k: v[:4] for k, v in first_bad.items()}, flush=True)
