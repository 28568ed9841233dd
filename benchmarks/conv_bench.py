"""Micro-benchmark of the trunk kernels at the bench's leaf-batch size: the layer kernel
(qz_nn_conv3x3_norm), the fused trunk (qz_nn_trunk fused=1) and the library path, HIP-event timed."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from _stubs import det_fill_state_dict
if os.environ.get("QZ_BENCH_LIB"):  # A/B of a differently built library: before anything loads it
    from alphazero_quoridor_amd import _cabi as _c
    _c.LIB_PATH = os.environ["QZ_BENCH_LIB"]
from alphazero_quoridor_amd.policy_value_net import LeafEvaluator, PolicyValueNet

ap = argparse.ArgumentParser()
ap.add_argument("--boards", type=int, default=4096)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--what", default="layered,fused,library,heads_staged,heads_separate")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
pvn = PolicyValueNet(use_gpu=True, device=dev)
pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), 2024))
g = torch.Generator().manual_seed(0)
x = torch.relu(torch.randn((a.boards, 64, 9, 9), generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
def timed(run, label):
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters): run()
    e.record(); torch.cuda.synchronize()
    print("%-60s %.1f us" % (label, s.elapsed_time(e) / a.iters * 1e3), flush=True)

for what in [w for w in a.what.split(",") if w.startswith("heads_")]:
    # trunk + heads from the first layer's output (no clone): the head stage inside the trunk launch + k_head_fc,
    # against the fused trunk + the separate fp32 head kernel
    ev = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True, fused_heads_stage=(what == "heads_staged"))
    if what == "heads_staged":
        timed(lambda: ev._trunk_heads_mfma(x), "trunk + head stage + k_head_fc (%d leaves)" % a.boards)
    else:
        xx = x.clone(memory_format=torch.preserve_format)
        def run2():
            t = ev._trunk_mfma(xx)   # in place: values drift towards a fixed point, timing is unaffected
            hd = ev._head
            from alphazero_quoridor_amd import _cabi
            p = torch.empty((a.boards, 140), dtype=torch.float32, device=dev); v = torch.empty(a.boards, dtype=torch.float32, device=dev)
            _cabi.check(_cabi.load().qz_nn_head(t.data_ptr(), a.boards, hd[0].data_ptr(), hd[8].data_ptr(), hd[1].data_ptr(), hd[2].data_ptr(),
                hd[3].data_ptr(), hd[4].data_ptr(), hd[5].data_ptr(), hd[6].data_ptr(), hd[7].data_ptr(), p.data_ptr(), v.data_ptr(), 1e-5,
                torch.cuda.current_stream(dev).cuda_stream))
        timed(run2, "fused trunk + separate k_head (%d leaves)" % a.boards)

for what in [w for w in a.what.split(",") if not w.startswith("heads_")]:
    ev = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True, fused_trunk=(what == "fused"), mfma_trunk=(what != "library"))
    def run():
        y = x.clone(memory_format=torch.preserve_format)
        if what == "library":
            li = 1
            for _ in range(5):
                z = ev._cbn(y, li); y = ev._cbn(z, li + 1, relu=True, residual=y); li += 2
            return y
        return ev._trunk_mfma(y)
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / a.iters
    flops = 10 * 2.0 * a.boards * 81 * 64 * 576
    print("%-8s trunk (10 layers, %d leaves): %.1f us  = %.1f us/layer, %.0f TFLOP/s fp32-equivalent (incl. one 85-MB clone)" % (what, a.boards, ms * 1e3, ms * 100, flops / ms / 1e9), flush=True)
