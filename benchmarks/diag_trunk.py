import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from _stubs import det_fill_state_dict
from alphazero_quoridor_amd import _cabi
from alphazero_quoridor_amd.policy_value_net import LeafEvaluator, PolicyValueNet
dev = torch.device("cuda:0")
pvn = PolicyValueNet(use_gpu=True, device=dev)
pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), 2024))
ev = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True)
L = _cabi.load()
g = torch.Generator().manual_seed(0)
for B in (1, 2, 64, 1024):
    x0 = torch.relu(torch.randn((B, 64, 9, 9), generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
    ev._trunk_mfma(x0.clone(memory_format=torch.preserve_format))
    w, gm, bt, sc = ev._trunk_args
    for nb in (1, 2, 5):
        outs = []
        for fused in (0, 1, 1):
            x = x0.clone(memory_format=torch.preserve_format)
            tmp = torch.empty_like(x)
            _cabi.check(L.qz_nn_trunk(x.data_ptr(), tmp.data_ptr(), B, nb, w, gm, bt, sc, 1e-5, fused, torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            outs.append(x)
        d = (outs[0] - outs[1]).abs()
        print("B=%d blocks=%d: fused vs layered max|d| %.3g (at %s), fused twice equal: %s, max|x| %.3g" %
              (B, nb, d.max().item(), tuple(int(v) for v in (d == d.max()).nonzero()[0]), torch.equal(outs[1], outs[2]), outs[0].abs().max().item()), flush=True)
