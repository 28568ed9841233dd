"""Diagnostic: which kernels of the library change their results when an MFMA-dense kernel runs on
another stream?  (k_head does; this checks the others.)"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from _stubs import det_fill_state_dict
from synth import synth_positions
from alphazero_quoridor_amd import _cabi, rules
from alphazero_quoridor_amd.boards import DeviceBoards
from alphazero_quoridor_amd.policy_value_net import LeafEvaluator, PolicyValueNet
dev = torch.device("cuda:0")
T = C.CDLL(os.path.join(ROOT, "tests", "hip", "libqz_testkernels.so"))
T.qzt_stress.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
sout = torch.zeros(2048 * 256, dtype=torch.float32, device=dev)
sin = torch.randn(2048 * 256 + 16 * 65536, dtype=torch.float32, device=dev)
pvn = PolicyValueNet(use_gpu=True, device=dev)
pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), 2024))
ev = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True)
N = 1024
db = DeviceBoards.from_packed(synth_positions(N, seed=3), dev)
g = torch.Generator().manual_seed(1)
x = torch.relu(torch.randn((N, 64, 9, 9), generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
gamma = torch.ones(64, device=dev); beta = torch.zeros(64, device=dev)
L = _cabi.load()
def v_input(st): return (ev._first_layer_from_boards((db.struct(), 0, db.n)),)
def v_rules(st): return rules.movegen_encode(db)
def v_norm(st):
    out = torch.empty_like(x)
    _cabi.check(L.qz_nn_instnorm_act_nhwc(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 0, out.data_ptr(), N, 64, 1, 1e-5, st.cuda_stream)); return (out,)
def v_trunk(st): return (ev._trunk_mfma(x.clone(memory_format=torch.preserve_format)),)
def v_head(st):
    hd = ev._head
    p = torch.empty((N, 140), dtype=torch.float32, device=dev); v = torch.empty(N, dtype=torch.float32, device=dev)
    _cabi.check(L.qz_nn_head(x.data_ptr(), N, hd[0].data_ptr(), hd[8].data_ptr(), hd[1].data_ptr(), hd[2].data_ptr(), hd[3].data_ptr(),
                             hd[4].data_ptr(), hd[5].data_ptr(), hd[6].data_ptr(), hd[7].data_ptr(), p.data_ptr(), v.data_ptr(), 1e-5, st.cuda_stream))
    return p, v
A, B = torch.cuda.Stream(), torch.cuda.Stream()
for name, fn in (("k_input_layer", v_input), ("rules op (k_wave_rules + encoder)", v_rules), ("k_instnorm_act_nhwc64", v_norm), ("k_trunk", v_trunk), ("k_head", v_head)):
    torch.cuda.synchronize()
    with torch.cuda.stream(A):
        ref = fn(A)
    torch.cuda.synchronize()
    bad = 0
    for rep in range(60):
        T.qzt_stress(sout.data_ptr(), sin.data_ptr(), 2048, 300, 1, 1, B.cuda_stream)
        with torch.cuda.stream(A):
            out = fn(A)
        T.qzt_stress(sout.data_ptr(), sin.data_ptr(), 2048, 300, 1, 1, B.cuda_stream)
        torch.cuda.synchronize()
        bad += int(not all(torch.equal(a, b) for a, b in zip(ref, out)))
    print("%-36s next to an MFMA loop: %d of 60 runs differ" % (name, bad), flush=True)
