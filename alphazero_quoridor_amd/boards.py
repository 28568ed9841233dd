"""Device-resident board batches in the SoA layout of ``include/qz_abi.h`` (three u64
words per board: hbits, vbits, meta) held as torch int64 tensors."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _cabi


class DeviceBoards:
    def __init__(self, n: int, device):
        self.n = int(n)
        self.device = torch.device(device)
        self.hbits = torch.zeros(self.n, dtype=torch.int64, device=self.device)
        self.vbits = torch.zeros(self.n, dtype=torch.int64, device=self.device)
        self.meta = torch.zeros(self.n, dtype=torch.int64, device=self.device)

    @classmethod
    def from_packed(cls, packed: np.ndarray, device):
        hb, vb, meta = _cabi.packed_to_soa(packed)
        o = cls(len(hb), device)
        o.hbits.copy_(torch.from_numpy(hb))
        o.vbits.copy_(torch.from_numpy(vb))
        o.meta.copy_(torch.from_numpy(meta))
        return o

    def to_packed(self) -> np.ndarray:
        return _cabi.soa_to_packed(self.hbits.cpu().numpy(), self.vbits.cpu().numpy(), self.meta.cpu().numpy())

    def struct(self) -> _cabi.qz_boards:
        return _cabi.qz_boards(self.hbits.data_ptr(), self.vbits.data_ptr(), self.meta.data_ptr())

    def byref(self):
        self._s = self.struct()
        return C.byref(self._s)

    def __len__(self):
        return self.n


def opening_packed(n: int = 1) -> np.ndarray:
    """Quoridor.reset() (quoridor.py:26-56) as packed records."""
    out = np.zeros(n, dtype=_cabi.PACKED_DTYPE)
    out["p1"], out["p2"], out["w1"], out["w2"], out["cur"] = 4, 76, 10, 10, 1
    return out
