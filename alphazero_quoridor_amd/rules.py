"""Batched rules kernels (thin wrappers over the C ABI): legal-move masks, state planes
and transitions for DeviceBoards.  Everything runs on the GPU through libqzero_hip.so."""
from __future__ import annotations

import torch

from . import _cabi
from .boards import DeviceBoards


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu(device="cuda:0"):
    if not torch.cuda.is_available():
        raise _cabi.QzError(_cabi.E_NO_DEVICE, "torch sees no HIP device; this package has no CPU path")
    return torch.device(device)


def movegen(boards: DeviceBoards) -> torch.Tensor:
    """Quoridor.actions() for every board -> int32 [n,5] 140-bit masks (quoridor.py:138-157)."""
    L = _cabi.load()
    mask = torch.empty((boards.n, 5), dtype=torch.int32, device=boards.device)
    with torch.cuda.device(boards.device):
        _cabi.check(L.qz_movegen(boards.byref(), boards.n, mask.data_ptr(), _stream(boards.device)))
    return mask


def encode(boards: DeviceBoards, out: torch.Tensor | None = None) -> torch.Tensor:
    """Quoridor.state() for every board -> float32 [n,26,9,9] (quoridor.py:58-131)."""
    L = _cabi.load()
    if out is None:
        out = torch.empty((boards.n, 26, 9, 9), dtype=torch.float32, device=boards.device)
    with torch.cuda.device(boards.device):
        _cabi.check(L.qz_encode(boards.byref(), boards.n, out.data_ptr(), _stream(boards.device)))
    return out


def rules_opts(variant=0, detour_pooled=None, detour_wave=None, enc_split_pct=None) -> _cabi.qz_rules_opts:
    """qz_rules_opts (include/qz_abi.h): which formulation of the rules op to run.  None = the
    library default; detour modes are 0 | 1 | 2."""
    return _cabi.qz_rules_opts(int(variant), 0 if detour_pooled is None else 1 + int(detour_pooled),
                               0 if detour_wave is None else 1 + int(detour_wave), int(enc_split_pct or 0))


def movegen_encode(boards: DeviceBoards, mask: torch.Tensor | None = None, planes: torch.Tensor | None = None, opts=None):
    """actions() + state() in one pass; `opts` = rules_opts(...) forces a kernel formulation."""
    import ctypes as C

    L = _cabi.load()
    if mask is None:
        mask = torch.empty((boards.n, 5), dtype=torch.int32, device=boards.device)
    if planes is None:
        planes = torch.empty((boards.n, 26, 9, 9), dtype=torch.float32, device=boards.device)
    with torch.cuda.device(boards.device):
        _cabi.check(L.qz_movegen_encode_opts(boards.byref(), boards.n, mask.data_ptr(), planes.data_ptr(),
                                             C.byref(opts) if opts is not None else None, _stream(boards.device)))
    return mask, planes


def step(boards: DeviceBoards, actions: torch.Tensor):
    """Quoridor.step() in place; returns (done uint8 [n], winner uint8 [n]) (quoridor.py:159-202)."""
    L = _cabi.load()
    actions = actions.to(device=boards.device, dtype=torch.uint8).contiguous()
    done = torch.empty(boards.n, dtype=torch.uint8, device=boards.device)
    winner = torch.empty(boards.n, dtype=torch.uint8, device=boards.device)
    with torch.cuda.device(boards.device):
        _cabi.check(L.qz_step(boards.byref(), actions.data_ptr(), boards.n, done.data_ptr(), winner.data_ptr(), _stream(boards.device)))
    return done, winner


# the reference's actions() order (quoridor.py:146-157, 423-428) as a permutation of action ids
ACTION_ORDER = list(range(12)) + [a for ix in range(64) for a in (12 + ix, 76 + ix)]


def mask_to_actions(mask5) -> list:
    """One 140-bit mask (5 x int32/uint32) -> the reference's ordered action list."""
    words = [int(w) & 0xFFFFFFFF for w in mask5]
    return [a for a in ACTION_ORDER if (words[a >> 5] >> (a & 31)) & 1]
