"""Multi-GPU glue: one process per GPU, every rank runs an independent SelfPlayEngine on its
own boards (no data-path collective); the only exchange is the all-gather of finished
(board, pi, z) tuples into every rank's replay buffer.  On ROCm the "nccl" backend is RCCL
over xGMI; the same code runs on "gloo" with CPU tensors (the world_size-2 CPU tests).

Wire format (588 B / ply): hbits u64 | vbits u64 | meta u64 | pi float32[140] | z float32.
Traffic is KB..MB per exchange -- far below one xGMI link (~153 GB/s) -- so the cost is
launch latency: one count all-gather + ONE padded payload all-gather per exchange.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

TUPLE_BYTES = 24 + 560 + 4


def pack_tuples(hbits, vbits, meta, pi, z) -> torch.Tensor:
    """-> uint8 [n, 588] on the inputs' device."""
    n = int(pi.shape[0])
    words = torch.stack([hbits, vbits, meta], dim=1).contiguous().view(torch.uint8).reshape(n, 24)
    return torch.cat([words, pi.contiguous().view(torch.uint8).reshape(n, 560),
                      z.contiguous().view(torch.uint8).reshape(n, 4)], dim=1).contiguous()


def unpack_tuples(buf: torch.Tensor):
    """uint8 [n,588] -> (hbits, vbits, meta int64 [n], pi float32 [n,140], z float32 [n])."""
    n = int(buf.shape[0])
    words = buf[:, :24].contiguous().view(torch.int64).reshape(n, 3)
    pi = buf[:, 24:584].contiguous().view(torch.float32).reshape(n, 140)
    z = buf[:, 584:588].contiguous().view(torch.float32).reshape(n)
    return words[:, 0].contiguous(), words[:, 1].contiguous(), words[:, 2].contiguous(), pi, z


def init_from_env(device_type="cuda"):
    """torchrun contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks: QZ_DIST_BACKEND=gloo + QZ_SHARE_DEVICE=1 let two ranks share ONE GPU (RCCL
    # refuses that), so the N>1 code path of bench.py can be exercised on a 1-GPU box
    backend = os.environ.get("QZ_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
    if os.environ.get("QZ_SHARE_DEVICE") == "1":
        local = 0
    # QZ_DIST_FORCE=1: create the process group (and run every collective of this package) even with ONE rank -- the
    # RCCL code path on a single GPU: init with device_id, uint8 / int64 all_gather_into_tensor, float32 all_reduce
    if (world > 1 or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if device_type == "cuda":
            torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def force_collectives() -> bool:
    return os.environ.get("QZ_DIST_FORCE") == "1"


def collectives_on(group=None) -> bool:
    """True if the package's collectives should really run: more than one rank, or one rank with QZ_DIST_FORCE=1."""
    if not dist.is_available() or not dist.is_initialized():
        return False
    return dist.get_world_size(group) > 1 or force_collectives()


class ExchangeLog:
    """What the path's only exchange costs ("games/s total and all-gather ms"): one entry per allgather_tuples
    call that really ran its collectives -- wall milliseconds from the count all-gather to the arrival of the payload (the
    device is synchronised at the end of a timed call: an exchange happens once per harvest, which synchronises anyway),
    bytes this rank contributed and received."""

    def __init__(self):
        self.ms, self.bytes_in, self.bytes_out = [], [], []

    def add(self, ms, n_in, n_out):
        self.ms.append(float(ms))
        self.bytes_in.append(int(n_in))
        self.bytes_out.append(int(n_out))

    def clear(self):
        del self.ms[:], self.bytes_in[:], self.bytes_out[:]

    def summary(self):
        n = len(self.ms)
        if n == 0:
            return {"calls": 0, "mean_ms": None, "max_ms": None, "mean_bytes_contributed": None, "mean_bytes_received": None}
        return {"calls": n, "mean_ms": sum(self.ms) / n, "max_ms": max(self.ms), "mean_bytes_contributed": sum(self.bytes_in) / n,
                "mean_bytes_received": sum(self.bytes_out) / n}


exchange_log = ExchangeLog()


def allgather_tuples(buf: torch.Tensor, group=None, n_games=None):
    """Every rank contributes uint8 [n_r, 588]; every rank gets the concatenation in rank
    order, uint8 [sum n_r, 588].  With `n_games` (this rank's finished games) the call returns
    (tuples, games summed over ranks) -- the same number on every rank, so loops that run
    "until N games" stay in lockstep and issue the same sequence of collectives.  Every call that runs
    its collectives is timed into `exchange_log`."""
    if not collectives_on(group):
        return buf if n_games is None else (buf, int(n_games))
    import time

    t0 = time.perf_counter()
    out = _allgather_tuples(buf, group, n_games)
    got = out[0] if n_games is not None else out
    if got.is_cuda:
        torch.cuda.synchronize(got.device)
    exchange_log.add((time.perf_counter() - t0) * 1e3, buf.numel(), got.numel())
    return out


def _allgather_tuples(buf, group, n_games):
    world = dist.get_world_size(group)
    dev = buf.device
    count = torch.tensor([buf.shape[0], int(n_games or 0)], dtype=torch.int64, device=dev)
    counts = torch.zeros(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, count, group=group)
    counts = counts.cpu().reshape(world, 2)
    games = int(counts[:, 1].sum())
    counts = counts[:, 0].tolist()
    mx = max(counts)
    if mx == 0:
        return buf if n_games is None else (buf, games)
    padded = torch.zeros((mx, TUPLE_BYTES), dtype=torch.uint8, device=dev)
    padded[: buf.shape[0]] = buf
    out = torch.empty((world * mx, TUPLE_BYTES), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, padded, group=group)
    out = out.reshape(world, mx, TUPLE_BYTES)
    out = torch.cat([out[r, : counts[r]] for r in range(world)], dim=0)
    return out if n_games is None else (out, games)


def shard_seed(seed: int, rank: int) -> int:
    """Independent Philox stream per rank (seed xor rank)."""
    return (int(seed) ^ (0x9E3779B97F4A7C15 * (rank + 1))) & 0xFFFFFFFFFFFFFFFF
