"""Where does a wave of the fused trunk kernel spend its cycles?  Diagnostic build with s_memtime
stamps (tests/hip/qz_conv_stamps.hip): per wave, cycles in staging / MFMA loop / statistics / hand-over."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from _stubs import det_fill_state_dict
from alphazero_quoridor_amd.policy_value_net import LeafEvaluator, PolicyValueNet
dev = torch.device("cuda:0")
T = C.CDLL(os.path.join(ROOT, "tests", "hip", "libqz_testkernels.so"))
T.qzt_trunk_stamps.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
pvn = PolicyValueNet(use_gpu=True, device=dev)
pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), 2024))
ev = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator().manual_seed(0)
x0 = torch.relu(torch.randn((B, 64, 9, 9), generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
ev._trunk_mfma(x0.clone(memory_format=torch.preserve_format))
w, gm, bt, sc = ev._trunk_args
stamps = torch.zeros(B * 2 * 8, dtype=torch.int64, device=dev)
for rep in range(3):
    x = x0.clone(memory_format=torch.preserve_format)
    torch.cuda.synchronize()
    assert T.qzt_trunk_stamps(x.data_ptr(), B, 10, w, gm, bt, sc, stamps.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 2, 8).astype(np.float64)
ref = ev._trunk_mfma(x0.clone(memory_format=torch.preserve_format))
print("diagnostic build output equals the product kernel:", bool(torch.equal(ref, x)))
tot = s[:, :, 4]
print("waves %d; cycles per wave: total %.0f | staging %.0f | MFMA loops %.0f | statistics %.0f | hand-over %.0f   (10 layers)" %
      (s.shape[0] * 2, tot.mean(), s[:, :, 0].mean(), s[:, :, 1].mean(), s[:, :, 2].mean(), s[:, :, 3].mean()))
print("per layer: MFMA loop %.0f cycles (324 MFMAs = 10,368 MFMA-pipe cycles), statistics %.0f, hand-over %.0f" % (s[:, :, 1].mean() / 10, s[:, :, 2].mean() / 10, s[:, :, 3].mean() / 9))
mhz = (s[:, :, 4] / np.maximum(s[:, :, 6], 1.0)) * 100.0
print("in-kernel clock (s_memtime / s_memrealtime x 100 MHz): median %.0f MHz, p5 %.0f, p95 %.0f -> the dense fp16 MFMA peak at that clock is %.2f PFLOP/s"
      % (np.median(mhz), np.percentile(mhz, 5), np.percentile(mhz, 95), 2.5 * np.median(mhz) / 2400.0))
begin = s[:, 0, 5]
span = (begin.max() - begin.min() + tot.max())
print("kernel span %.0f cycles; workgroups %d; first-start spread %.0f" % (span, s.shape[0], begin.max() - begin.min()))
