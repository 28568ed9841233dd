"""GPU parity of the split-fp16 MFMA trunk layer (csrc/qz_conv.hip) against plain PyTorch fp32:
conv3x3 + per-sample (training-mode, batch of one) BatchNorm + residual + ReLU."""
import numpy as np
import pytest
import numpy as np
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, w, gamma, beta, res, relu, eps=1e-5):
    """float64 reference on the CPU: F.conv2d, then every (sample, channel) plane normalised with its
    own mean / biased variance (BatchNorm2d.train() on a batch of one, policy_value_net.py:20-48)."""
    x, w = x.double().cpu(), w.double().cpu()
    y = F.conv2d(x, w, None, 1, 1)
    mean = y.mean(dim=(2, 3), keepdim=True)
    var = y.var(dim=(2, 3), unbiased=False, keepdim=True)
    y = (y - mean) / torch.sqrt(var + eps) * gamma.double().cpu().view(1, -1, 1, 1) + beta.double().cpu().view(1, -1, 1, 1)
    if res is not None:
        y = y + res.double().cpu()
    return F.relu(y) if relu else y


def _run(x, w, gamma, beta, res, relu):
    from alphazero_quoridor_amd import _cabi
    from alphazero_quoridor_amd.policy_value_net import LeafEvaluator

    w16, inv_scale = LeafEvaluator._split_weight(w)
    out = torch.empty_like(x, memory_format=torch.channels_last)
    _cabi.check(_cabi.load().qz_nn_conv3x3_norm(x.data_ptr(), w16.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                res.data_ptr() if res is not None else 0, out.data_ptr(), x.shape[0], inv_scale.data_ptr(),
                                                int(relu), 1e-5, torch.cuda.current_stream(x.device).cuda_stream))
    return out


def test_mfma_operand_maps_with_exact_integer_data(gpu_device):
    """Small-integer activations and weights are exact in fp16 and in the fp32 accumulator, so the
    un-normalised convolution must be reproduced EXACTLY: any wrong lane -> element map, tap offset
    or padding shows up as a bit difference.  (gamma = std, beta = mean would need the statistics;
    instead the kernel's normalisation is inverted on the host in float64.)"""
    g = torch.Generator().manual_seed(1)
    for B in (1, 2, 3, 7):
        x = torch.randint(-4, 5, (B, 64, 9, 9), generator=g).float()
        w = torch.randint(-3, 4, (64, 64, 3, 3), generator=g).float()  # asymmetric in every index
        xd = x.to(gpu_device).contiguous(memory_format=torch.channels_last)
        gamma = torch.ones(64, device=gpu_device)
        beta = torch.zeros(64, device=gpu_device)
        out = _run(xd, w.to(gpu_device), gamma, beta, None, relu=False).cpu().double()
        y = F.conv2d(x.double(), w.double(), None, 1, 1)
        mean = y.mean(dim=(2, 3), keepdim=True)
        std = torch.sqrt(y.var(dim=(2, 3), unbiased=False, keepdim=True) + 1e-5)
        back = out * std + mean  # undo the normalisation: integers again
        assert (back - y).abs().max().item() < 2e-3, B
        assert torch.equal(torch.round(back), y), B


def test_trunk_layer_matches_fp32_reference(gpu_device):
    """Random fp32 data with the network's magnitudes: |error| vs a float64 reference must be at
    the level of an fp32 convolution's own rounding noise (MIOpen's is measured next to it)."""
    g = torch.Generator().manual_seed(7)
    for B in (1, 2, 5, 64, 257):
        x = (torch.randn((B, 64, 9, 9), generator=g).clamp(min=-0.3) * 1.3).to(gpu_device).contiguous(memory_format=torch.channels_last)
        w = ((torch.rand((64, 64, 3, 3), generator=g) * 2 - 1) / 24.0).to(gpu_device)
        gamma = (torch.rand(64, generator=g) + 0.5).to(gpu_device)
        beta = torch.randn(64, generator=g).to(gpu_device)
        res = torch.randn((B, 64, 9, 9), generator=g).to(gpu_device).contiguous(memory_format=torch.channels_last)
        for use_res in (False, True):
            for relu in (False, True):
                want = _ref(x, w, gamma, beta, res if use_res else None, relu)
                got = _run(x, w, gamma, beta, res if use_res else None, relu)
                assert got.is_contiguous(memory_format=torch.channels_last)
                err = (got.cpu().double() - want).abs().max().item()
                # the library path on the same inputs
                y = F.conv2d(x, w.contiguous(memory_format=torch.channels_last), None, 1, 1)
                y = F.instance_norm(y, None, None, gamma, beta, True, 0.0, 1e-5)
                if use_res:
                    y = y + res
                if relu:
                    y = F.relu(y)
                lib = (y.cpu().double() - want).abs().max().item()
                if B == 257 and use_res and relu:
                    print("max |err| vs float64: split-fp16 MFMA %.3g, MIOpen fp32 %.3g" % (err, lib))
                assert err < 4e-6, (B, use_res, relu, err, lib)
    # in place (out aliases x) is allowed
    x2 = x.clone(memory_format=torch.preserve_format)
    from alphazero_quoridor_amd import _cabi
    from alphazero_quoridor_amd.policy_value_net import LeafEvaluator
    w16, inv_scale = LeafEvaluator._split_weight(w)
    _cabi.check(_cabi.load().qz_nn_conv3x3_norm(x2.data_ptr(), w16.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 0, x2.data_ptr(),
                                                x2.shape[0], inv_scale.data_ptr(), 1, 1e-5, torch.cuda.current_stream().cuda_stream))
    assert torch.equal(x2, _run(x, w, gamma, beta, None, True))


def test_evaluator_with_mfma_trunk_vs_float64_and_library_trunk(gpu_device, golden_dir):
    """The whole evaluator (first layer -> 10 MFMA trunk layers -> fused heads) on 4,096 leaves
    against the same evaluator on MIOpen's fp32 convolutions, and both against a float64 evaluation
    of the module on the CPU (256 leaves): the split-fp16 trunk must be as close to the exact
    result as the library's fp32 trunk is (the value head amplifies trunk noise ~10x)."""
    from _stubs import det_fill_state_dict
    from alphazero_quoridor_amd import rules
    from alphazero_quoridor_amd.boards import DeviceBoards
    from alphazero_quoridor_amd.policy_value_net import LeafEvaluator, PolicyValueNet
    from synth import synth_positions

    pvn = PolicyValueNet(use_gpu=True, device=gpu_device)
    pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), 2024))
    boards = synth_positions(4096, seed=5)
    planes = rules.encode(DeviceBoards.from_packed(boards, gpu_device))
    fast = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True, mfma_trunk=True)
    slow = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True, mfma_trunk=False)
    assert fast.mfma_trunk and fast._w16 is not None and not slow.mfma_trunk
    p1, v1 = fast(planes)
    p2, v2 = slow(planes)
    dp, dv = (p1 - p2).abs().max().item(), (v1 - v2).abs().max().item()
    cpu = PolicyValueNet(use_gpu=False)
    cpu.policy_value_net.load_state_dict(det_fill_state_dict(cpu.policy_value_net.state_dict(), 2024))
    cpu.policy_value_net.double()
    exact = LeafEvaluator(cpu.policy_value_net, "per_leaf", dtype=torch.float64)
    pe, ve = exact(planes[:256].cpu().double())
    e_fast = ((p1[:256].cpu().double() - pe).abs().max().item(), (v1[:256].cpu().double() - ve).abs().max().item())
    e_slow = ((p2[:256].cpu().double() - pe).abs().max().item(), (v2[:256].cpu().double() - ve).abs().max().item())
    print("evaluator vs float64 (256 leaves): MFMA trunk |dp| %.3g |dv| %.3g; MIOpen trunk |dp| %.3g |dv| %.3g; "
          "MFMA vs MIOpen on 4,096 leaves |dp| %.3g |dv| %.3g" % (e_fast + e_slow + (dp, dv)))
    assert e_fast[0] < 2e-6 and e_fast[1] < 1e-5
    assert e_fast[1] <= max(2.0 * e_slow[1], 6e-6)   # no worse than the library's fp32 path (within its own noise)
    assert dp < 5e-6 and dv < 2e-5


def test_fused_trunk_equals_layer_by_layer(gpu_device):
    """qz_nn_trunk: the persistent one-launch trunk (activations resident on the CU) against ten
    launches of the layer kernel: the same MFMA sequence per element; the per-leaf statistics are
    summed in another order (inside one wave instead of across two), so the results agree to fp32
    rounding, not bit for bit.  Batch sizes with and without a ragged last workgroup."""
    from _stubs import det_fill_state_dict
    from alphazero_quoridor_amd.policy_value_net import LeafEvaluator, PolicyValueNet

    pvn = PolicyValueNet(use_gpu=True, device=gpu_device)
    pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), 2024))
    fused = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True, fused_trunk=True)
    layered = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True, fused_trunk=False)
    g = torch.Generator().manual_seed(11)
    for B in (1, 2, 3, 64, 777, 4096):
        x = torch.relu(torch.randn((B, 64, 9, 9), generator=g)).to(gpu_device).contiguous(memory_format=torch.channels_last)
        a = fused._trunk_mfma(x.clone(memory_format=torch.preserve_format))
        b = layered._trunk_mfma(x.clone(memory_format=torch.preserve_format))
        err = (a - b).abs().max().item()
        assert torch.isfinite(a).all() and err < 2e-5 * max(1.0, b.abs().max().item()), (B, err)
    planes = (torch.rand((130, 26, 9, 9), generator=g) > 0.8).float().to(gpu_device)
    p1, v1 = fused(planes)
    p2, v2 = layered(planes)
    assert (p1 - p2).abs().max().item() < 2e-6 and (v1 - v2).abs().max().item() < 1e-5


def test_head_stage_of_the_trunk_launch_equals_the_separate_head_kernel(gpu_device):
    """qz_nn_trunk_heads: the merged 64 -> 6 head convolution + bn2 / bn3 + ReLU as the last stage of
    the fused trunk launch (MFMA on the activations still in LDS) and k_head_fc, against the fused
    trunk followed by the fp32 VALU head kernel (qz_nn_head): same function, p to 2e-6 and v to 1e-5
    (the tolerance of the evaluator against the reference).  Ragged batch sizes included; the input
    tensor must not be modified."""
    from _stubs import det_fill_state_dict
    from alphazero_quoridor_amd.policy_value_net import LeafEvaluator, PolicyValueNet

    pvn = PolicyValueNet(use_gpu=True, device=gpu_device)
    pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), 2024))
    staged = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True)
    separate = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True, fused_heads_stage=False)
    assert staged._w6_16 is not None and staged.fused_heads_stage
    g = torch.Generator().manual_seed(12)
    for B in (1, 5, 6, 7, 64, 1000, 4096):
        planes = (torch.rand((B, 26, 9, 9), generator=g) > 0.8).float().to(gpu_device)
        p1, v1 = staged(planes)
        p2, v2 = separate(planes)
        assert p1.shape == (B, 140) and v1.shape == (B,)
        dp, dv = (p1 - p2).abs().max().item(), (v1 - v2).abs().max().item()
        assert torch.isfinite(p1).all() and torch.isfinite(v1).all() and dp < 2e-6 and dv < 1e-5, (B, dp, dv)
        assert abs(p1.sum(dim=1) - 1).max().item() < 1e-5
    x = torch.relu(torch.randn((300, 64, 9, 9), generator=g)).to(gpu_device).contiguous(memory_format=torch.channels_last)
    x0 = x.clone(memory_format=torch.preserve_format)
    pa, va = staged._trunk_heads_mfma(x)
    assert torch.equal(x, x0)
    pb, vb = staged._trunk_heads_mfma(x)
    assert torch.equal(pa, pb) and torch.equal(va, vb)   # run to run: bit-identical


def test_input_stage_of_the_trunk_launch_equals_the_separate_input_kernel(gpu_device, golden_dir):
    """qz_nn_evaluate: the first layer computed from the packed boards inside the trunk launch (same
    additions in the same order as qz_nn_input_layer; the per-leaf statistics are summed in another
    order) against qz_nn_input_layer + qz_nn_trunk_heads, on fixture positions of all game phases,
    with terminal flags on some leaves and ragged batch sizes.  p to 2e-6, v to 1e-5."""
    from _stubs import det_fill_state_dict
    from alphazero_quoridor_amd.boards import DeviceBoards
    from alphazero_quoridor_amd.policy_value_net import LeafEvaluator, PolicyValueNet

    pvn = PolicyValueNet(use_gpu=True, device=gpu_device)
    pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), 2024))
    staged = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True)
    separate = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True, fused_input_stage=False)
    assert staged.fused_input_stage and staged.board_input_layer
    b = np.load(golden_dir + "/rules_positions.npz")["board"]
    rng = np.random.RandomState(3)
    for n in (1, 7, 130, 4096):
        sel = b[rng.randint(0, len(b), size=n)]
        db = DeviceBoards.from_packed(sel, gpu_device)
        term = torch.from_numpy((rng.rand(n) < 0.1).astype(np.uint8)).to(gpu_device)
        for tp in (0, term.data_ptr()):
            leaf = (db.struct(), tp, n)
            p1, v1 = staged(None, leaf=leaf)
            p2, v2 = separate(None, leaf=leaf)
            dp, dv = (p1 - p2).abs().max().item(), (v1 - v2).abs().max().item()
            assert torch.isfinite(p1).all() and torch.isfinite(v1).all() and dp < 2e-6 and dv < 1e-5, (n, bool(tp), dp, dv)
        pa, va = staged(None, leaf=leaf)
        assert torch.equal(pa, p1) and torch.equal(va, v1)   # run to run: bit-identical


def test_fp16_throughput_mode_stays_within_its_stated_bound(gpu_device, golden_dir):
    """The labelled NON-PARITY mode (LeafEvaluator nn_precision="fp16", bench.py --nn-dtype fp16): one MFMA per product on
    fp16 operands instead of three on split operands.  Against the reference's policy_value_fn on the 64 fixture states
    it must stay within 5e-3 on p and 3e-2 on v (the parity mode: 1e-5), and a 64-playout search with it must mostly
    visit what the parity evaluator's search visits (same boards, same seed): the printed figures are what the mode
    costs; nothing in the product selects it by default."""
    import sys, os
    sys.path.insert(0, golden_dir)
    from _stubs import det_fill_state_dict
    from synth import synth_positions
    from alphazero_quoridor_amd.boards import DeviceBoards
    from alphazero_quoridor_amd.engine import SelfPlayEngine
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

    d = np.load(golden_dir + "/net_fixture.npz")
    pvn = PolicyValueNet(use_gpu=True, device=gpu_device)
    pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), 2024))
    ev32 = pvn.evaluator("per_leaf", torch.float32, True)
    ev16 = pvn.evaluator("per_leaf", torch.float32, True, nn_precision="fp16")
    assert ev16 is not ev32 and ev16.nn_precision == "fp16" and ev16.engine_route_ok() and ev16.nn_weights().precision == 1
    db = DeviceBoards.from_packed(d["board"], gpu_device)
    p, v = ev16(None, leaf=(db.struct(), 0, db.n))
    p, v = p.cpu().numpy(), v.cpu().numpy()
    worst_p = worst_v = 0.0
    for i in range(64):
        acts = d["leaf_acts"][i]
        k = int((acts != 255).sum())
        worst_p = max(worst_p, float(np.abs(p[i][acts[:k]] - d["leaf_p"][i][:k]).max()))
        worst_v = max(worst_v, float(abs(v[i] - d["leaf_v"][i])))
    boards = synth_positions(128, seed=3, max_walls=10)
    engs = []
    for ev in (ev32, ev16):
        eng = SelfPlayEngine(128, n_playout=64, seed=1, device=gpu_device)
        eng.set_boards(DeviceBoards.from_packed(boards, gpu_device), reset_trees=True)
        eng.run_playouts_memo(ev)
        engs.append(eng)
    (pi_a, vis_a), (pi_b, vis_b) = engs[0].root_pi(), engs[1].root_pi()
    same_best = float((pi_a.argmax(dim=1) == pi_b.argmax(dim=1)).float().mean())
    l1 = float((pi_a - pi_b).abs().sum(dim=1).mean())
    for eng in engs:
        eng.close()
    print("fp16 throughput mode vs reference policy_value_fn: max |dp| %.3g, max |dv| %.3g; 64-playout searches: same most-visited move on "
          "%.0f %% of 128 boards, mean L1 distance of pi %.3f" % (worst_p, worst_v, 100 * same_best, l1))
    assert worst_p < 5e-3 and worst_v < 3e-2, (worst_p, worst_v)
    assert same_best > 0.6 and l1 < 0.8, (same_best, l1)
