"""Policy-value network for the self-play path, kept in PyTorch-ROCm.

Mirror of the reference's ``policy_value_net`` module surface (policy_value_net.py:110-200):
``PolicyValueNet(model_file=None, use_gpu=True)`` with ``policy_value``, ``policy_value_fn``,
``train_step``, ``get_policy_param``, ``save_model``.  The nn.Module keeps the reference's
parameter names so ``ckpt/<name>.pth`` state_dicts interchange.

What is new here is the batched leaf evaluator used by the engine (``LeafEvaluator``):
one forward on a device-resident [B,26,9,9] leaf batch with no host round trip, in one of
three BatchNorm modes:

  "per_leaf"  every sample normalised with its own statistics -- what the reference's
              policy_value_fn does (module left in train mode, batch of one:
              policy_value_net.py:154 with no .eval() anywhere).  Batch-invariant.  DEFAULT.
  "batch"     batch statistics, like the reference's policy_value (policy_value_net.py:127-143)
  "eval"      running statistics (standard inference; BN folded into the convolutions)
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

N_ACTIONS = 140
IN_PLANES = 26
WIDTH = 64
N_RES = 5
BN_EPS = 1e-5


class _BatchNormTrainF64Acc(torch.autograd.Function):
    """Training-mode batch normalisation whose REDUCTIONS run in float64 (the element-wise part stays
    fp32).  The library's GPU kernels accumulate the batch sums in fp32; the backward's
    dy - mean(dy) - xhat * mean(dy * xhat) cancels most of dy in this network, so fp32 sums cost
    three digits of the early layers' gradients after ten blocks (benchmarks/diag_train_grads.py:
    2e-3 of the largest element, against 2e-6 for PyTorch's CPU kernels -- which accumulate in
    double, and which the reference runs on).  With float64 sums the GPU's gradients are as close
    to the float64 values as the reference's CPU ones (tests/test_gpu_train.py)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        xd = x.double()
        mean = xd.mean(dim=(0, 2, 3))
        var = xd.var(dim=(0, 2, 3), unbiased=False)
        invstd = torch.rsqrt(var + eps)
        scale = (invstd * weight.double())[None, :, None, None]
        y = ((xd - mean[None, :, None, None]) * scale + bias.double()[None, :, None, None]).to(x.dtype)
        ctx.save_for_backward(x, weight, mean, invstd)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar):
        x, weight, mean, invstd = ctx.saved_tensors
        n = x.shape[0] * x.shape[2] * x.shape[3]
        dyd = dy.double()
        xhat = (x.double() - mean[None, :, None, None]) * invstd[None, :, None, None]
        dbeta = dyd.sum(dim=(0, 2, 3))
        dgamma = (dyd * xhat).sum(dim=(0, 2, 3))
        dx = (weight.double() * invstd)[None, :, None, None] * (dyd - (dbeta / n)[None, :, None, None] - xhat * (dgamma / n)[None, :, None, None])
        return dx.to(x.dtype), dgamma.to(weight.dtype), dbeta.to(weight.dtype), None


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters, buffers and state_dict keys) whose training-mode pass on the GPU
    accumulates in float64 (_BatchNormTrainF64Acc); everything else is the parent's."""

    def forward(self, x):
        if not (self.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and self.affine and self.track_running_stats
                and self.momentum is not None):
            return super().forward(x)
        y, mean, var = _BatchNormTrainF64Acc.apply(x, self.weight, self.bias, self.eps)
        with torch.no_grad():
            n = x.shape[0] * x.shape[2] * x.shape[3]
            self.num_batches_tracked += 1
            self.running_mean.mul_(1.0 - self.momentum).add_(mean.to(self.running_mean.dtype), alpha=self.momentum)
            self.running_var.mul_(1.0 - self.momentum).add_((var * (n / max(n - 1, 1))).to(self.running_var.dtype), alpha=self.momentum)
        return y


class BasicBlock(nn.Module):
    """conv-bn-relu-conv-bn-(+x)-relu (policy_value_net.py:20-48); names conv1/bn1/conv2/bn2."""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        y = y + (x if self.downsample is None else self.downsample(x))
        return self.relu(y)


class policy_value_net(nn.Module):  # noqa: N801 -- the reference's class name
    """26x9x9 -> (log-softmax over 140 actions, tanh value); policy_value_net.py:51-99."""

    def __init__(self, block=BasicBlock, inplanes=IN_PLANES, planes=WIDTH, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        for i in range(1, N_RES + 1):
            setattr(self, "res%d" % i, block(planes, planes))
        self.conv2 = nn.Conv2d(planes, 4, 3, stride, 1, bias=False)  # value head
        self.bn2 = BatchNorm2d(4)
        self.fc1 = nn.Linear(4 * 81, 128)
        self.fc2 = nn.Linear(128, 1)
        self.conv3 = nn.Conv2d(planes, 2, 3, stride, 1, bias=False)  # policy head
        self.bn3 = BatchNorm2d(2)
        self.fc3 = nn.Linear(2 * 81, N_ACTIONS)

    def trunk(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        for i in range(1, N_RES + 1):
            x = getattr(self, "res%d" % i)(x)
        return x

    def forward(self, x):
        t = self.trunk(x)
        v = self.relu(self.bn2(self.conv2(t))).reshape(-1, 4 * 81)
        v = torch.tanh(self.fc2(self.fc1(v)))  # no activation between fc1 and fc2 (:89-90)
        p = self.relu(self.bn3(self.conv3(t))).reshape(-1, 2 * 81)
        p = F.log_softmax(self.fc3(p), dim=1)
        return p, v


# --------------------------------------------------------------------------------------
class LeafEvaluator:
    """planes [B,26,9,9] float32 (device) -> (p [B,140] float32, v [B] float32), both
    contiguous and device resident: the `p`/`v` arguments of qz_mcts_expand_backup.

    p = exp(log_softmax(logits)) as in policy_value_net.py:155 (not renormalised over the
    legal moves: the expand kernel gathers it at the legal actions, policy_value_net.py:162).
    """

    def __init__(self, net: nn.Module, bn_mode="per_leaf", dtype=torch.float32, channels_last=False, fused_norm=True,
                 board_input_layer=True, fused_head=True, mfma_trunk=True, fused_trunk=True, fused_heads_stage=True, fused_input_stage=True,
                 nn_precision="fp32"):
        assert bn_mode in ("per_leaf", "batch", "eval")
        # "fp32" (default, the parity mode): every product of the trunk as three MFMAs on split fp16 operands, 1e-5 against the
        # reference.  "fp16": ONE MFMA per product on fp16 operands with fp32 accumulation -- the labelled THROUGHPUT mode of the
        # HIP evaluation from packed boards (~1e-3 on p / v); results are no longer the reference's to 1e-5
        assert nn_precision in ("fp32", "fp16")
        self.nn_precision = nn_precision
        self.fused_norm = fused_norm  # per_leaf on the GPU: use the one-pass HIP normalisation kernel
        # first layer straight from the packed boards (qz_nn_input_layer) when the caller hands them
        # over: fp32, channels-last, per-leaf or folded BatchNorm
        self.board_input_layer = board_input_layer and bn_mode in ("per_leaf", "eval") and dtype == torch.float32 \
            and channels_last and fused_norm
        self.accepts_leaf_boards = self.board_input_layer
        self._in_tables = None
        # both heads in one HIP kernel (qz_nn_head), same conditions: 104 us instead of 183 us for the
        # library convolution + zero-fill + normalisation + three GEMMs + softmax/exp/tanh + copies
        self.fused_head = fused_head and fused_norm and bn_mode in ("per_leaf", "eval") and dtype == torch.float32 \
            and channels_last
        self._head = None
        # trunk layers (conv3x3 64->64 + per-leaf norm + residual + ReLU) as ONE HIP kernel each on the fp16
        # matrix cores with split operands (fp32 accuracy; qz_nn_conv3x3_norm): fp32, channels-last, per-leaf
        self.mfma_trunk = mfma_trunk and fused_norm and bn_mode == "per_leaf" and dtype == torch.float32 and channels_last
        self._w16 = None
        self._w6_16 = None
        self._inv_scale = None
        self._trunk_args = None
        self._nn_weights = None
        self._dirty = False
        self.version = 0            # bumped by refresh(): engines flush their leaf-evaluation memo when it changes
        self.fused_trunk = fused_trunk  # all ten layers in ONE persistent launch (activations stay on the CU)
        self.fused_heads_stage = fused_heads_stage  # the head convolution as the last stage of the fused trunk launch
        self.fused_input_stage = fused_input_stage  # ... and the first layer (from the packed boards) as its first
        self.trunk_events = None   # bench.py: a list that receives (start, end) HIP events around trunk-layer launches
        self.net = net
        self.bn_mode = bn_mode
        self.dtype = dtype
        self.channels_last = channels_last
        self._layers = None
        self.refresh()

    # weights may change between self-play rounds (training).  Cached tensors are updated IN
    # PLACE so that a captured HIP graph keeps pointing at live data.
    def mark_dirty(self):
        """The module's weights changed; the copies are re-derived before the next use (ensure_fresh), not now: a
        policy update runs several optimiser steps before self-play resumes."""
        self._dirty = True

    def ensure_fresh(self):
        if self._dirty:
            self.refresh()

    def refresh(self):
        self._dirty = False
        self.version += 1
        n, dt = self.net, self.dtype
        mf = torch.channels_last if self.channels_last else torch.contiguous_format

        def conv_w(w):
            return w.detach().to(dt).contiguous(memory_format=mf).clone(memory_format=torch.preserve_format)

        names = [("conv1", "bn1")]
        for i in range(1, N_RES + 1):
            names += [("res%d.conv1" % i, "res%d.bn1" % i), ("res%d.conv2" % i, "res%d.bn2" % i)]
        names += [("conv2", "bn2"), ("conv3", "bn3")]
        mods = dict(n.named_modules())
        layers = []
        for cn, bn in names:
            conv, b = mods[cn], mods[bn]
            if self.bn_mode == "eval":
                scale = (b.weight / torch.sqrt(b.running_var + b.eps)).detach()
                w = conv_w(conv.weight.detach() * scale.view(-1, 1, 1, 1))
                bias = (b.bias - b.running_mean * scale).detach().to(dt).clone()
                layers.append([w, bias, None, None])
            else:
                layers.append([conv_w(conv.weight), None, b.weight.detach().to(dt).clone(), b.bias.detach().to(dt).clone()])
        # the two head convolutions (64->4 value, 64->2 policy) read the same trunk output: run
        # them as ONE 64->6 convolution + one normalisation pass and split afterwards
        hv, hp = layers[-2], layers[-1]
        layers.append([torch.cat([hv[0], hp[0]], 0).contiguous(memory_format=mf)] +
                      [None if a is None else torch.cat([a, b], 0) for a, b in zip(hv[1:], hp[1:])])
        fc = [[m.weight.detach().to(dt).clone(), m.bias.detach().to(dt).clone()] for m in (n.fc1, n.fc2, n.fc3)]
        if self.fused_head and layers[0][0].is_cuda:
            hw, hbias, hg, hb = layers[-1]
            head = [hw.permute(2, 3, 1, 0).reshape(9, 64, 6).contiguous(),          # [tap][cin][6]
                    (hbias if self.bn_mode == "eval" else hb).contiguous(),
                    fc[0][0].t().contiguous(), fc[0][1], fc[1][0].reshape(-1).contiguous(), fc[1][1],
                    fc[2][0].t().contiguous(), fc[2][1]]
            if self.bn_mode != "eval":
                head.append(hg.contiguous())
            if self._head is None:
                self._head = head
            else:
                for o, t in zip(self._head, head):
                    o.copy_(t)
        if self.mfma_trunk and layers[0][0].is_cuda:
            # 1 / scale of every layer's weight image lives in ONE device array (2 N_RES trunk layers + the head stage),
            # updated in place like the images themselves: kernels read it from memory, so a launch captured in a HIP
            # graph before a training step runs with the scales of the weights it finds
            if self._inv_scale is None:
                self._inv_scale = torch.ones(2 * N_RES + 1, dtype=torch.float32, device=layers[0][0].device)
            w16 = [self._split_weight(layers[i][0]) for i in range(1, 1 + 2 * N_RES)]
            if self._w16 is None:
                self._w16 = [(t, self._inv_scale[i:i + 1]) for i, (t, _) in enumerate(w16)]
            else:
                for (o, _), (t, _) in zip(self._w16, w16):
                    o.copy_(t)
            self._inv_scale[:2 * N_RES].copy_(torch.cat([sc for _, sc in w16]))
            if self.fused_head and self.fused_trunk:
                # the merged head convolution as one more stage of the fused trunk launch (qz_nn_trunk_heads):
                # the same split-fp16 B operand with 32 output columns, 6 of them real
                hw = layers[-1][0]
                w6 = self._split_weight(torch.cat([hw, torch.zeros((26,) + tuple(hw.shape[1:]), dtype=hw.dtype, device=hw.device)], 0))
                if self._w6_16 is None:
                    self._w6_16 = (w6[0], self._inv_scale[2 * N_RES:])
                else:
                    self._w6_16[0].copy_(w6[0])
                self._inv_scale[2 * N_RES:].copy_(w6[1])
        if self.board_input_layer and layers[0][0].is_cuda:
            tabs = self._input_tables(layers[0][0])
            if self._in_tables is None:
                self._in_tables = tabs
            else:
                for o, t in zip(self._in_tables, tabs):
                    o.copy_(t)
        if getattr(self, "_layers", None) is None:
            self._layers, self._fc = layers, fc
        else:
            for old, new in zip(self._layers + self._fc, layers + fc):
                for o, t in zip(old, new):
                    if o is not None:
                        o.copy_(t)

    @staticmethod
    def _split_weight(w):
        """conv weight [c_out,64,3,3] fp32 -> (fp16 [2,9,4,c_out,16] = [hi|lo][tap][c_in chunk][c_out][c_in in chunk] of
        w * scale, 1 / scale as a one-element DEVICE tensor): the B operand of qz_nn_conv3x3_norm (include/qz_abi.h).
        scale is the power of two that brings max |w| into [1, 2), so that the lo parts are fp16 normals.  Everything
        stays on the device (no host synchronisation: this runs after every optimiser step)."""
        W = w.detach().to(torch.float32).contiguous()
        mx = W.abs().max()
        _, ex = torch.frexp(mx)                      # mx = m * 2^ex, m in [0.5, 1): floor(log2(mx)) = ex - 1, exactly
        one = torch.ones((), dtype=torch.float32, device=W.device)
        live = mx > 0
        scale = torch.where(live, torch.ldexp(one, 1 - ex), one)
        inv = torch.where(live, torch.ldexp(one, ex - 1), one).reshape(1)
        ws = (W * scale).permute(2, 3, 1, 0).reshape(9, 4, 16, W.shape[0]).permute(0, 1, 3, 2).contiguous()  # [tap][chunk][c_out][16 c_in]
        hi = ws.to(torch.float16)
        lo = (ws - hi.to(torch.float32)).to(torch.float16)
        return torch.stack([hi, lo]).contiguous(), inv

    def _conv_norm_mfma(self, x, i, relu=True, residual=None):
        from . import _cabi
        w16, inv_scale = self._w16[i - 1]
        _, _, gamma, beta = self._layers[i]
        out = torch.empty_like(x, memory_format=torch.channels_last)
        ev = None
        if self.trunk_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        _cabi.check(_cabi.load().qz_nn_conv3x3_norm(
            x.data_ptr(), w16.data_ptr(), gamma.data_ptr(), beta.data_ptr(), residual.data_ptr() if residual is not None else 0,
            out.data_ptr(), x.shape[0], inv_scale.data_ptr(), int(relu), BN_EPS, torch.cuda.current_stream(x.device).cuda_stream))
        if ev is not None:
            ev[1].record()
            self.trunk_events.append(ev)
        return out

    def _trunk_heads_mfma(self, x):
        """Trunk + both heads from one library call, two launches (qz_nn_trunk_heads): x = the first layer's
        output -> (p, v); the trunk output never reaches HBM."""
        from . import _cabi
        w, g, b, sc = self._trunk_tables()
        hd = self._head
        B, dev = x.shape[0], x.device
        p = torch.empty((B, N_ACTIONS), dtype=torch.float32, device=dev)
        v = torch.empty(B, dtype=torch.float32, device=dev)
        feat = torch.empty((B, 6 * 81), dtype=torch.float32, device=dev)
        _cabi.check(_cabi.load().qz_nn_trunk_heads(
            x.data_ptr(), B, N_RES, w, g, b, sc, self._w6_16[0].data_ptr(), hd[8].data_ptr(), hd[1].data_ptr(),
            hd[2].data_ptr(), hd[3].data_ptr(), hd[4].data_ptr(), hd[5].data_ptr(), hd[6].data_ptr(), hd[7].data_ptr(),
            feat.data_ptr(), p.data_ptr(), v.data_ptr(), BN_EPS, torch.cuda.current_stream(dev).cuda_stream))
        return p, v

    def _evaluate_boards_mfma(self, leaf):
        """The whole evaluation from the packed leaf boards, two launches (qz_nn_evaluate_w): first layer from the
        boards, trunk and head convolution in one persistent launch, then the fully connected layers."""
        from . import _cabi
        import ctypes as C
        st, term_ptr, n = leaf
        dev = self._layers[0][2].device
        p = torch.empty((n, N_ACTIONS), dtype=torch.float32, device=dev)
        v = torch.empty(n, dtype=torch.float32, device=dev)
        feat = torch.empty((n, 6 * 81), dtype=torch.float32, device=dev)
        _cabi.check(_cabi.load().qz_nn_evaluate_w(C.byref(st), term_ptr or 0, n, C.byref(self.nn_weights()), feat.data_ptr(), p.data_ptr(), v.data_ptr(),
                                                  torch.cuda.current_stream(dev).cuda_stream))
        return p, v

    def _trunk_tables(self):
        import ctypes as C
        if getattr(self, "_trunk_args", None) is None:
            L = 2 * N_RES
            self._trunk_args = ((C.c_void_p * L)(*[self._w16[i][0].data_ptr() for i in range(L)]),
                                (C.c_void_p * L)(*[self._layers[i + 1][2].data_ptr() for i in range(L)]),
                                (C.c_void_p * L)(*[self._layers[i + 1][3].data_ptr() for i in range(L)]),
                                self._inv_scale.data_ptr())
        return self._trunk_args

    def engine_route_ok(self):
        """True if this evaluator is the two-launch HIP evaluation from packed boards (qz_nn_evaluate): the one the
        asynchronous self-play loop can run on its miss list (qz_selfplay_evaluate)."""
        return bool(self.board_input_layer and self._in_tables is not None and self.fused_input_stage and self.bn_mode == "per_leaf"
                    and self.mfma_trunk and self.fused_trunk and self.fused_head and self.fused_heads_stage and self._w16 is not None
                    and self._w6_16 is not None and self._head is not None and len(self._head) > 8)

    def nn_weights(self):
        """qz_nn_weights (include/qz_abi.h) over this evaluator's device tensors.  They are updated in place by
        refresh(), so the struct stays valid for the evaluator's lifetime."""
        from . import _cabi
        self.ensure_fresh()
        if self._nn_weights is None:
            assert self.engine_route_ok(), "this evaluator configuration has no one-call HIP route"
            w, g, b, sc = self._trunk_tables()
            hd = self._head
            _, _, gamma0, beta0 = self._layers[0]
            hot9, base0, wd = self._in_tables
            nw = _cabi.qz_nn_weights()
            nw.hot9, nw.base0, nw.wd, nw.gamma0, nw.beta0 = hot9.data_ptr(), base0.data_ptr(), wd.data_ptr(), gamma0.data_ptr(), beta0.data_ptr()
            nw.n_blocks = N_RES
            nw.w16, nw.gamma, nw.beta = C_addr(w), C_addr(g), C_addr(b)
            nw.inv_scale = sc
            nw.w6_16 = self._w6_16[0].data_ptr()
            nw.gamma6, nw.beta6 = hd[8].data_ptr(), hd[1].data_ptr()
            nw.w1t, nw.b1, nw.w2, nw.b2, nw.w3t, nw.b3 = (hd[i].data_ptr() for i in (2, 3, 4, 5, 6, 7))
            nw.eps = BN_EPS
            nw.precision = 1 if self.nn_precision == "fp16" else 0
            self._nn_weights = nw
        return self._nn_weights

    def _trunk_mfma(self, x):
        """All ten trunk layers from one library call (qz_nn_trunk); x is updated in place."""
        from . import _cabi
        w, g, b, sc = self._trunk_tables()
        # scratch of the layer-by-layer route: from the caching allocator every time (one evaluator may serve
        # several board groups on different streams at once, so nothing mutable is kept on the object)
        tmp = None if self.fused_trunk else torch.empty_like(x, memory_format=torch.channels_last)
        _cabi.check(_cabi.load().qz_nn_trunk(x.data_ptr(), tmp.data_ptr() if tmp is not None else 0, x.shape[0], N_RES, w, g, b, sc, BN_EPS,
                                             int(self.fused_trunk), torch.cuda.current_stream(x.device).cuda_stream))
        return x

    @staticmethod
    def _input_tables(w):
        """Tables of qz_nn_input_layer (include/qz_abi.h) from the first layer's weight
        [64,26,3,3]: the layer applied, in float64, to the few images state() is made of."""
        W = w.detach().to(torch.float64).contiguous()
        dev = W.device
        img = torch.zeros((22, 26, 9, 9), dtype=torch.float64, device=dev)
        for i in range(21):
            img[i, 5 + i] = 1.0              # an all-ones plane 5+i
        img[21, 0, :8, :8] = 1.0             # plane 0 with no wall on the board
        out = F.conv2d(img, W, None, 1, 1)   # [22,64,9,9]
        rep = torch.tensor([0, 4, 8], device=dev)
        hot9 = out[:21][:, :, rep][:, :, :, rep]                     # [21,64,3,3]: the nine border classes
        hot9 = hot9.permute(0, 2, 3, 1).reshape(21, 9, 64)
        base0 = out[21].permute(1, 2, 0).reshape(81, 64)
        kinds = torch.stack([W[:, 2] - W[:, 0], W[:, 1] - W[:, 0], W[:, 3], W[:, 4]])   # [4,64,3,3]
        # output (y,x) sees input pixel (y+dy, x+dx) through W[.., dy+1, dx+1]
        wd = kinds.permute(0, 2, 3, 1).reshape(4, 9, 64)
        return [t.to(torch.float32).contiguous() for t in (hot9, base0, wd)]

    def _first_layer_from_boards(self, leaf):
        from . import _cabi
        import ctypes as C
        st, term_ptr, n = leaf
        w, bias, gamma, beta = self._layers[0]
        dev = w.device
        out = torch.empty((n, 64, 9, 9), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        hot9, base0, wd = self._in_tables
        if self.bn_mode == "eval":
            g_ptr, b_ptr = 0, bias.data_ptr()
        else:
            g_ptr, b_ptr = gamma.data_ptr(), beta.data_ptr()
        _cabi.check(_cabi.load().qz_nn_input_layer(C.byref(st), term_ptr or 0, n, hot9.data_ptr(), base0.data_ptr(), wd.data_ptr(),
                                                   g_ptr, b_ptr, out.data_ptr(), BN_EPS, torch.cuda.current_stream(dev).cuda_stream))
        return out

    def _cbn(self, x, i, relu=True, residual=None):
        w, bias, gamma, beta = self._layers[i]
        if self.mfma_trunk and self._w16 is not None and 1 <= i <= 2 * N_RES and x.is_cuda and x.dtype == torch.float32 \
                and x.shape[1] == WIDTH and x.is_contiguous(memory_format=torch.channels_last) \
                and (residual is None or residual.is_contiguous(memory_format=torch.channels_last)):
            return self._conv_norm_mfma(x, i, relu, residual)
        y = F.conv2d(x, w, bias, 1, 1)
        if self.bn_mode == "per_leaf" and self.fused_norm and y.is_cuda and y.dtype == torch.float32 and y.dim() == 4 \
                and y.shape[1] > 1 and y.shape[1] <= 64 and y.is_contiguous(memory_format=torch.channels_last) \
                and (residual is None or residual.is_contiguous(memory_format=torch.channels_last)):
            from . import _cabi
            _cabi.check(_cabi.load().qz_nn_instnorm_act_nhwc(
                y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), residual.data_ptr() if residual is not None else 0,
                y.data_ptr(), y.shape[0], y.shape[1], int(relu), BN_EPS, torch.cuda.current_stream(y.device).cuda_stream))
            return y
        if self.bn_mode == "per_leaf" and self.fused_norm and y.is_cuda and y.dtype == torch.float32 \
                and y.is_contiguous() and (residual is None or residual.is_contiguous()):
            # one HIP pass: per-plane statistics + affine (+ residual) + ReLU (csrc/qz_nn.hip)
            from . import _cabi
            _cabi.check(_cabi.load().qz_nn_instnorm_act(
                y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), residual.data_ptr() if residual is not None else 0,
                y.data_ptr(), y.shape[0] * y.shape[1], y.shape[1], int(relu), BN_EPS,
                torch.cuda.current_stream(y.device).cuda_stream))
            return y
        if self.bn_mode == "per_leaf":
            # BatchNorm2d in training mode on a batch of one == per-sample statistics over
            # the 81 positions (biased variance), then the affine transform
            y = F.instance_norm(y, None, None, gamma, beta, True, 0.0, BN_EPS)
        elif self.bn_mode == "batch":
            y = F.batch_norm(y, None, None, gamma, beta, True, 0.0, BN_EPS)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y

    @torch.no_grad()
    def __call__(self, planes: torch.Tensor, leaf=None):
        """leaf = (qz_boards struct, terminal-flag device pointer or None, n): the boards `planes`
        was encoded from (SelfPlayEngine.leaf_ref()); with it the first layer is computed from
        the 24-byte boards and `planes` is not read."""
        self.ensure_fresh()
        if leaf is not None and self.board_input_layer and self._in_tables is not None:
            if self.fused_input_stage and self.bn_mode == "per_leaf" and self.mfma_trunk and self.fused_trunk and self.fused_head \
                    and self.fused_heads_stage and self._w16 is not None and self._w6_16 is not None and self._head is not None \
                    and len(self._head) > 8 and self.trunk_events is None:
                return self._evaluate_boards_mfma(leaf)
            x = self._first_layer_from_boards(leaf)
        else:
            x = planes.to(self.dtype)
            if self.channels_last:
                x = x.contiguous(memory_format=torch.channels_last)
            x = self._cbn(x, 0)
        li = 1
        mfma_ok = self.mfma_trunk and self._w16 is not None and self.trunk_events is None and x.is_cuda and x.dtype == torch.float32 \
            and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] == WIDTH
        if mfma_ok and self.fused_trunk and self.fused_head and self.fused_heads_stage and self._w6_16 is not None and self._head is not None \
                and len(self._head) > 8:
            return self._trunk_heads_mfma(x)
        if mfma_ok:
            x = self._trunk_mfma(x)  # ten launches, one library call; x (the first layer's output) is updated in place
            li += 2 * N_RES
        else:
            for _ in range(N_RES):
                y = self._cbn(x, li)
                x = self._cbn(y, li + 1, relu=True, residual=x)
                li += 2
        B = x.shape[0]
        if self.fused_head and self._head is not None and x.is_cuda and x.is_contiguous(memory_format=torch.channels_last):
            from . import _cabi
            hd = self._head
            p = torch.empty((B, N_ACTIONS), dtype=torch.float32, device=x.device)
            v = torch.empty(B, dtype=torch.float32, device=x.device)
            _cabi.check(_cabi.load().qz_nn_head(
                x.data_ptr(), B, hd[0].data_ptr(), hd[8].data_ptr() if len(hd) > 8 else 0, hd[1].data_ptr(), hd[2].data_ptr(),
                hd[3].data_ptr(), hd[4].data_ptr(), hd[5].data_ptr(), hd[6].data_ptr(), hd[7].data_ptr(), p.data_ptr(), v.data_ptr(),
                BN_EPS, torch.cuda.current_stream(x.device).cuda_stream))
            return p, v
        h = self._cbn(x, li + 2)  # merged value+policy head convolution (all norm modes are per channel)
        v = h[:, :4].reshape(B, 4 * 81)
        p = h[:, 4:].reshape(B, 2 * 81)
        (w1, b1), (w2, b2), (w3, b3) = self._fc
        v = torch.tanh(F.linear(F.linear(v, w1, b1), w2, b2)).reshape(B)
        p = torch.exp(F.log_softmax(F.linear(p, w3, b3).float(), dim=1))
        return p.contiguous(), v.float().contiguous()


# --------------------------------------------------------------------------------------
def C_addr(arr):
    import ctypes as C
    return C.addressof(arr)


def set_learning_rate(optimizer, lr):
    for group in optimizer.param_groups:
        group["lr"] = lr


class PolicyValueNet:
    """Drop-in for the reference's PolicyValueNet (policy_value_net.py:110-200).

    use_gpu=True requires a HIP device (as the reference's .cuda() does); use_gpu=False keeps
    the *network* on the CPU for checkpoint work and numerics tests -- game logic still runs
    on the GPU (the rules engine has no CPU path).
    """

    def __init__(self, model_file=None, use_gpu=True, bn_mode="per_leaf", device=None):
        self.use_gpu = use_gpu
        self.l2_const = 1e-4
        if device is None:
            device = "cuda:0" if use_gpu else "cpu"
        self.device = torch.device(device)
        self.policy_value_net = policy_value_net(BasicBlock, IN_PLANES, WIDTH).to(self.device)
        self.optimizer = torch.optim.Adam(self.policy_value_net.parameters(), weight_decay=self.l2_const)
        self.bn_mode = bn_mode
        if model_file:
            path = model_file if os.path.exists(str(model_file)) else "ckpt/%s.pth" % model_file
            self.policy_value_net.load_state_dict(torch.load(path, map_location=self.device))
        self._evaluators = {}

    # engine-facing ---------------------------------------------------------------
    def evaluator(self, bn_mode=None, dtype=torch.float32, channels_last=None, nn_precision="fp32") -> LeafEvaluator:
        if channels_last is None:
            # MIOpen's NHWC fp32 convolutions are ~20 % faster on MI355X; "batch" mode stays NCHW
            # (MIOpen's NHWC BatchNorm-training path crashed on this stack)
            channels_last = self.device.type == "cuda" and (bn_mode or self.bn_mode) != "batch"
        # one evaluator per configuration, all kept: engines hold on to theirs (cloned, re-laid-out
        # weights), and weights_changed() must reach every one of them after a training step
        key = (bn_mode or self.bn_mode, dtype, bool(channels_last), nn_precision)
        if key not in self._evaluators:
            self._evaluators[key] = LeafEvaluator(self.policy_value_net, key[0], dtype, channels_last, nn_precision=nn_precision)
        return self._evaluators[key]

    def weights_changed(self):
        """Call after the module's parameters changed (train_step does): every live evaluator
        re-derives its weight copies IN PLACE before its next use (captured HIP graphs keep pointing at live data; the
        per-layer scales live in device memory too), and engines flush their leaf-evaluation memos."""
        for ev in self._evaluators.values():
            ev.mark_dirty()

    # device-tensor API (what TrainPipeline uses: nothing goes through numpy) ---------------------
    def _as_states(self, state_batch):
        if isinstance(state_batch, torch.Tensor):
            return state_batch.to(self.device, torch.float32)
        return torch.as_tensor(np.asarray(state_batch), dtype=torch.float32, device=self.device)

    @torch.no_grad()
    def policy_value_t(self, states: torch.Tensor):
        """states float32 [B,26,9,9] on the device -> (act_probs [B,140], value [B,1]) device
        tensors.  Like the reference's policy_value (policy_value_net.py:127-143) the module is in
        train mode, so BatchNorm uses the batch's statistics (and updates its running ones)."""
        logp, v = self.policy_value_net(states)
        return torch.exp(logp), v

    def train_step_t(self, states, mcts_probs, winners, lr):
        """One optimiser step on device tensors (policy_value_net.py:166-192):
        loss = mse(v, z) - mean(sum(pi * log p)) (+ L2 through Adam's weight decay).  Returns
        (loss, entropy) as 0-dim device tensors: no host synchronisation here.  With
        torch.distributed initialised (one rank per GPU) the gradients are averaged over the ranks
        by ONE all-reduce of a flat 1.8-MB bucket (RCCL over xGMI) before the step, so replicas
        that started from the same weights stay identical."""
        self.optimizer.zero_grad(set_to_none=False)
        set_learning_rate(self.optimizer, lr)
        logp, v = self.policy_value_net(states)
        value_loss = F.mse_loss(v.view(-1), winners)
        policy_loss = -torch.mean(torch.sum(mcts_probs * logp, 1))
        loss = value_loss + policy_loss
        loss.backward()
        self._allreduce_grads()
        self.optimizer.step()
        with torch.no_grad():
            entropy = -torch.mean(torch.sum(torch.exp(logp) * logp, 1))
        self.weights_changed()
        return loss.detach(), entropy

    def _world(self):
        import torch.distributed as dist

        return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    @staticmethod
    def _collectives_on():
        from . import dist as qdist

        return qdist.collectives_on()

    def _allreduce_grads(self):
        world = self._world()
        if not self._collectives_on():
            return
        import torch.distributed as dist

        params = [p for p in self.policy_value_net.parameters() if p.grad is not None]
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        dist.all_reduce(flat)
        flat.div_(world)
        off = 0
        for p in params:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n

    def sync_from_rank0(self):
        """Every rank takes rank 0's weights, BatchNorm buffers and optimiser step count: replicas
        must start identical for the gradient all-reduce to keep them identical."""
        if not self._collectives_on():
            return
        import torch.distributed as dist

        with torch.no_grad():
            for t in list(self.policy_value_net.parameters()) + list(self.policy_value_net.buffers()):
                dist.broadcast(t, src=0)
        self.weights_changed()

    def average_buffers(self):
        """BatchNorm running statistics are fed by each rank's own minibatches: average them."""
        world = self._world()
        if not self._collectives_on():
            return
        import torch.distributed as dist

        with torch.no_grad():
            bufs = [b for b in self.policy_value_net.buffers() if b.dtype.is_floating_point]
            flat = torch.cat([b.reshape(-1) for b in bufs])
            dist.all_reduce(flat)
            flat.div_(world)
            off = 0
            for b in bufs:
                b.copy_(flat[off:off + b.numel()].view_as(b))
                off += b.numel()

    # reference API -----------------------------------------------------------------
    def policy_value(self, state_batch):
        """batch of states -> (act_probs float32 [B,140], value float32 [B,1]) as numpy
        (policy_value_net.py:127-143; module in train mode => batch statistics)."""
        probs, v = self.policy_value_t(self._as_states(state_batch))
        return probs.cpu().numpy(), v.cpu().numpy()

    def policy_value_fn(self, game):
        """game -> (iterable[(action, prob)], value) (policy_value_net.py:145-164)."""
        legal = game.actions()
        x = torch.as_tensor(np.ascontiguousarray(game.state()).reshape(1, 26, 9, 9), dtype=torch.float32,
                            device=self.device)
        with torch.no_grad():
            logp, v = self.policy_value_net(x)
        probs = np.exp(logp.cpu().numpy().reshape(-1))
        return zip(legal, probs[legal]), float(v.reshape(-1)[0])

    def train_step(self, state_batch, mcts_probs, winner_batch, lr):
        """The reference's signature (lists / arrays in, Python floats out: policy_value_net.py:
        166-192, whose `.data[0]` no longer works on a modern torch)."""
        pi = torch.as_tensor(np.asarray(mcts_probs), dtype=torch.float32, device=self.device)
        z = torch.as_tensor(np.asarray(winner_batch), dtype=torch.float32, device=self.device)
        loss, entropy = self.train_step_t(self._as_states(state_batch), pi, z, lr)
        return loss.item(), entropy.item()

    def get_policy_param(self):
        return self.policy_value_net.state_dict()

    def save_model(self, model_file):
        os.makedirs("ckpt", exist_ok=True)
        torch.save(self.policy_value_net.state_dict(), "ckpt/%s.pth" % model_file)
