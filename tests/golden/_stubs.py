"""Shared by gen_golden.py (reference side) and the tests: packed-board dtype, the
deterministic stub policies (pure Python ints, duck-typed on the reference's Quoridor
attribute names) and the deterministic weight filler for the network fixture.

The stub policies obey the reference's callback contract
``policy_value_function(game) -> (iterable[(action, prob)], value)``
(policy_value_net.py:145-164, pure_mcts.py:13-16); priors are np.float32 like the real
net's, the value is a Python float.
"""
from __future__ import annotations

import numpy as np

PACKED_DTYPE = np.dtype(
    [("hbits", "<u8"), ("vbits", "<u8"), ("p1", "i1"), ("p2", "i1"), ("w1", "u1"), ("w2", "u1"),
     ("cur", "u1"), ("pad", "u1", (3,))]
)

M32 = 0xFFFFFFFF


def pack_fields(inter, p1, p2, w1, w2, cur):
    rec = np.zeros((), dtype=PACKED_DTYPE)
    hb = vb = 0
    for ix in range(64):
        v = int(inter[ix])
        if v == 1:
            hb |= 1 << ix
        elif v == -1:
            vb |= 1 << ix
    rec["hbits"], rec["vbits"] = hb, vb
    rec["p1"], rec["p2"], rec["w1"], rec["w2"], rec["cur"] = int(p1), int(p2), int(w1), int(w2), int(cur)
    return rec


def fmix32(h):
    h &= M32
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M32
    h ^= h >> 16
    return h


def state_hash_fields(hb, vb, p1, p2, w1, w2, cur):
    words = [hb & M32, (hb >> 32) & M32, vb & M32, (vb >> 32) & M32,
             (p1 & 0xFF) | ((p2 & 0xFF) << 8) | ((w1 & 0xFF) << 16) | ((w2 & 0xFF) << 24), cur]
    h = 0x9E3779B9
    for w in words:
        h = fmix32(h ^ w)
    return h


def state_hash_game(game):
    """game: anything with the reference's attribute names (quoridor.py:34-56)."""
    hb = vb = 0
    inter = game._intersections
    for ix in range(64):
        v = int(inter[ix])
        if v == 1:
            hb |= 1 << ix
        elif v == -1:
            vb |= 1 << ix
    return state_hash_fields(hb, vb, int(game._positions[1]), int(game._positions[2]),
                             int(game._player1_walls_remaining), int(game._player2_walls_remaining),
                             int(game.current_player))


def hash_prior(h, a):
    r = fmix32(h ^ ((a * 0x9E3779B1 + 0x7F4A7C15) & M32))
    return np.float32((r >> 8) + 1) * np.float32(1.0 / 536870912.0)


def hash_value(h):
    r2 = fmix32(h ^ 0xA511E9B3)
    return float(((r2 >> 8) - 8388608) / 16777216.0)


def _legal(game):
    # the reference evaluates the policy on terminal leaves too (mcts.py:117) and its real
    # policy_value_fn crashes there on off-board winning jumps (IndexError);
    # the output is ignored for terminal leaves (mcts.py:121-125), so the stubs swallow it.
    try:
        return game.actions()
    except IndexError:
        return []


def hash_policy_py(game):
    legal = _legal(game)
    h = state_hash_game(game)
    return zip(legal, [hash_prior(h, a) for a in legal]), hash_value(h)


def uniform_policy_py(game):
    legal = _legal(game)
    n = len(legal)
    p = np.float32(1.0 / n) if n else np.float32(0)
    return zip(legal, [p] * n), 0.0


def det_fill_state_dict(sd, seed=2024):
    """Deterministic, well-conditioned weights for every key of the policy-value net's
    state_dict (names in SURVEY Appendix C); returns {key: torch.Tensor}."""
    import torch

    out = {}
    for i, (k, v) in enumerate(sd.items()):
        rs = np.random.RandomState(seed + i)
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_mean"):
            out[k] = torch.from_numpy((0.1 * rs.standard_normal(shape)).astype(np.float32))
        elif k.endswith("running_var"):
            out[k] = torch.from_numpy((1.0 + 0.2 * np.abs(rs.standard_normal(shape))).astype(np.float32))
        elif ".bn" in k or k.startswith("bn"):
            if k.endswith("weight"):
                out[k] = torch.from_numpy((1.0 + 0.1 * rs.standard_normal(shape)).astype(np.float32))
            else:
                out[k] = torch.from_numpy((0.1 * rs.standard_normal(shape)).astype(np.float32))
        elif k.endswith("bias"):
            out[k] = torch.from_numpy((0.05 * rs.standard_normal(shape)).astype(np.float32))
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            out[k] = torch.from_numpy((rs.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32))
    return out


def trained_like_state_dict(sd, seed=4711):
    """Weights with the spread a trained net shows, for the network parity fixtures: BatchNorm gammas log-uniform in
    [0.1, 3] and betas ~ N(0, 0.5), convolution / linear weights whose output channels differ in scale by up to 100x
    inside every layer (10^U(-1, 1) per channel on top of the He fill), biases ~ N(0, 0.2).  Deterministic in `seed`."""
    import torch

    out = {}
    for i, (k, v) in enumerate(sd.items()):
        rs = np.random.RandomState(seed + i)
        shape = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_mean"):
            out[k] = torch.from_numpy((0.3 * rs.standard_normal(shape)).astype(np.float32))
        elif k.endswith("running_var"):
            out[k] = torch.from_numpy((0.5 + np.abs(rs.standard_normal(shape))).astype(np.float32))
        elif ".bn" in k or k.startswith("bn"):
            if k.endswith("weight"):
                out[k] = torch.from_numpy(np.exp(rs.uniform(np.log(0.1), np.log(3.0), shape)).astype(np.float32))
            else:
                out[k] = torch.from_numpy((0.5 * rs.standard_normal(shape)).astype(np.float32))
        elif k.endswith("bias"):
            out[k] = torch.from_numpy((0.2 * rs.standard_normal(shape)).astype(np.float32))
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            w = rs.standard_normal(shape) * np.sqrt(2.0 / fan_in)
            ch = 10.0 ** rs.uniform(-1.0, 1.0, (shape[0],) + (1,) * (len(shape) - 1))
            if k.startswith("fc"):
                ch = ch * 0.12  # (nothing normalises the fully connected layers' outputs: keep tanh / softmax out of saturation)
            out[k] = torch.from_numpy((w * ch).astype(np.float32))
    return out
