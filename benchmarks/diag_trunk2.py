import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from _stubs import det_fill_state_dict
from alphazero_quoridor_amd import _cabi
from alphazero_quoridor_amd.policy_value_net import LeafEvaluator, PolicyValueNet
dev = torch.device("cuda:0")
T = C.CDLL(os.path.join(ROOT, "tests", "hip", "libqz_testkernels.so"))
T.qzt_trunk_stamps.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
pvn = PolicyValueNet(use_gpu=True, device=dev)
pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), 2024))
ev = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True)
L = _cabi.load()
g = torch.Generator().manual_seed(0)
B = 1024
x0 = torch.relu(torch.randn((B, 64, 9, 9), generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
ev._trunk_mfma(x0.clone(memory_format=torch.preserve_format))
w, gm, bt, sc = ev._trunk_args
stamps = torch.zeros(B * 2 * 8, dtype=torch.int64, device=dev)
# one layer: fused kernel (no hand-over) vs the layer kernel
ref = torch.empty_like(x0)
w16, inv = ev._w16[0]
_cabi.check(L.qz_nn_conv3x3_norm(x0.data_ptr(), w16.data_ptr(), ev._layers[1][2].data_ptr(), ev._layers[1][3].data_ptr(), 0, ref.data_ptr(), B, inv, 1, 1e-5,
                                 torch.cuda.current_stream().cuda_stream))
for rep in range(3):
    x = x0.clone(memory_format=torch.preserve_format)
    T.qzt_trunk_stamps(x.data_ptr(), B, 1, w, gm, bt, sc, stamps.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    d = (x - ref).abs()
    bad = (d > 1e-4).nonzero()
    print("1 layer: max|d| %.3g, elements off by > 1e-4: %d, first: %s" % (d.max().item(), len(bad), bad[:5].tolist()), flush=True)
# two layers, to look at WHERE the wrong elements are
refs = []
for rep in range(3):
    x = x0.clone(memory_format=torch.preserve_format)
    T.qzt_trunk_stamps(x.data_ptr(), B, 2, w, gm, bt, sc, stamps.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    refs.append(x)
x = x0.clone(memory_format=torch.preserve_format); tmp = torch.empty_like(x)
_cabi.check(L.qz_nn_trunk(x.data_ptr(), tmp.data_ptr(), B, 1, w, gm, bt, sc, 1e-5, 0, torch.cuda.current_stream().cuda_stream)); torch.cuda.synchronize()
for r_ in refs:
    d = (r_ - x).abs()
    bad = (d > 1e-3).nonzero()
    leaves = sorted(set(bad[:, 0].tolist()))
    pos = sorted(set((int(a), int(b)) for a, b in bad[:, 2:].tolist()))
    print("2 layers: max|d| %.3g; bad elements %d in %d leaves; positions %s; channels %s" % (d.max().item(), len(bad), len(leaves), pos[:12], sorted(set(bad[:, 1].tolist()))[:16]), flush=True)
# anatomy of a bad leaf after ONE layer
for rep in range(12):
    x = x0.clone(memory_format=torch.preserve_format)
    T.qzt_trunk_stamps(x.data_ptr(), B, 1, w, gm, bt, sc, stamps.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    d = (x - ref)
    bad = (d.abs() > 1e-4).nonzero()
    if len(bad) == 0:
        continue
    leaf = int(bad[0, 0])
    dl = d[leaf].reshape(64, 81).cpu().numpy()       # [channel][position]
    xl = x[leaf].reshape(64, 81).cpu().numpy(); rl = ref[leaf].reshape(64, 81).cpu().numpy()
    chans = [c for c in range(64) if np.abs(dl[c]).max() > 1e-4]
    print("bad leaf %d (workgroup %d): channels %s" % (leaf, leaf, chans))
    for c in chans[:4]:
        pos = np.nonzero(np.abs(dl[c]) > 1e-4)[0]
        print("  ch %d: %d bad positions; got[:6] %s want[:6] %s" % (c, len(pos), np.round(xl[c][:6], 3).tolist(), np.round(rl[c][:6], 3).tolist()))
    sm_id = stamps.cpu().numpy().reshape(-1, 2, 8)[leaf, :, 7]
    print("  smid of its waves:", [hex(int(v)) for v in sm_id])
    break
