"""Replay buffer and replay-shard files for the self-play -> training hand-over.

The reference keeps ``deque(maxlen=buffer_size)`` of (float64 state [26,9,9], float64 pi [140],
float64 z) tuples on the host (train.py:23, 55-63) -- 72 KB per ply -- and ``random.sample``s
minibatches out of it (train.py:67).  Here the buffer is a device-resident ring of the packed
588-byte tuples the engine harvests (board 24 B | pi float32[140] | z float32): 150x smaller,
and a minibatch is gathered on the GPU and its states are re-encoded by the HIP encoder
(``qz_encode``, quoridor.py:58-131) right where the training step reads them.  Nothing of the
training path goes through numpy.

Replay shards (``*.qzr``) are the same tuples on disk, so that self-play jobs and training jobs
can run apart (the reference has no such format; train.py:109 only checkpoints the net):

    offset  size  field
    0       4     magic  b"QZRS"
    4       4     version (1)                     little endian throughout
    8       4     tuple_bytes (588)
    12      4     n_playout of the games (0 = unknown)
    16      8     n_tuples
    24      8     n_games (0 = unknown)
    32      4     crc32 of the payload
    36      28    reserved (0)
    64      n_tuples x 588   hbits u64 | vbits u64 | meta u64 | pi f32[140] | z f32
"""
from __future__ import annotations

import os
import random
import struct
import zlib

import numpy as np
import torch

from . import dist as qdist
from .boards import DeviceBoards

SHARD_MAGIC = b"QZRS"
SHARD_VERSION = 1
SHARD_HEADER = 64
TUPLE_BYTES = qdist.TUPLE_BYTES


# ------------------------------------------------------------------------------ shard files
def write_shard(path, packed, n_games=0, n_playout=0):
    """packed: uint8 [n, 588] (torch, any device, or numpy).  Atomic: writes path + '.tmp' then renames."""
    if isinstance(packed, torch.Tensor):
        packed = packed.detach().cpu().numpy()
    packed = np.ascontiguousarray(packed, dtype=np.uint8).reshape(-1, TUPLE_BYTES)
    payload = packed.tobytes()
    head = SHARD_MAGIC + struct.pack("<IIIQQI", SHARD_VERSION, TUPLE_BYTES, int(n_playout), packed.shape[0], int(n_games),
                                     zlib.crc32(payload) & 0xFFFFFFFF)
    head += b"\0" * (SHARD_HEADER - len(head))
    tmp = str(path) + ".tmp"
    with open(tmp, "wb") as f:
        f.write(head)
        f.write(payload)
    os.replace(tmp, path)
    return packed.shape[0]


def read_shard(path):
    """-> (uint8 numpy [n, 588], {"n_games", "n_playout", "version"}).  Raises ValueError on a
    foreign, truncated or corrupted file."""
    with open(path, "rb") as f:
        head = f.read(SHARD_HEADER)
        if len(head) != SHARD_HEADER or head[:4] != SHARD_MAGIC:
            raise ValueError("%s is not a replay shard (bad magic)" % path)
        version, tb, n_playout, n, n_games, crc = struct.unpack("<IIIQQI", head[4:36])
        if version != SHARD_VERSION or tb != TUPLE_BYTES:
            raise ValueError("%s: unsupported shard version %d / tuple size %d" % (path, version, tb))
        payload = f.read()
    if len(payload) != n * TUPLE_BYTES:
        raise ValueError("%s: truncated (%d of %d payload bytes)" % (path, len(payload), n * TUPLE_BYTES))
    if zlib.crc32(payload) & 0xFFFFFFFF != crc:
        raise ValueError("%s: payload checksum mismatch" % path)
    return np.frombuffer(payload, dtype=np.uint8).reshape(n, TUPLE_BYTES).copy(), {"n_games": n_games, "n_playout": n_playout, "version": version}


# ------------------------------------------------------------------------------ planes <-> boards
def planes_to_packed(states) -> np.ndarray:
    """Inverse of Quoridor.state() (quoridor.py:58-131) for live positions: float [n,26,9,9] ->
    packed board records, so that reference-shaped tuples can enter the packed buffer.  Pawn
    planes 3/4 are mover/opponent, planes 5..14 / 15..24 one-hot walls remaining (plane 14 / 24
    is both "10 left" and "0 left", quoridor.py:79-80: settled by the number of walls on the board), plane 25 = mover is player 2."""
    from . import _cabi

    st = np.asarray(states).reshape(-1, 26, 81)
    n = st.shape[0]
    out = np.zeros(n, dtype=_cabi.PACKED_DTYPE)
    grid = st.reshape(n, 26, 9, 9)
    v = (grid[:, 1, :8, :8] > 0.5).reshape(n, 64)
    h = (grid[:, 2, :8, :8] > 0.5).reshape(n, 64)
    w = (1 << np.arange(64, dtype=np.uint64))
    out["hbits"] = (h * w).sum(axis=1, dtype=np.uint64)
    out["vbits"] = (v * w).sum(axis=1, dtype=np.uint64)
    mover_is_p2 = st[:, 25, 0] > 0.5
    pm = st[:, 3].argmax(axis=1)
    po = st[:, 4].argmax(axis=1)
    wm = st[:, 5:15, 0].argmax(axis=1) + 1
    wo = st[:, 15:25, 0].argmax(axis=1) + 1
    # plane 14 / 24 means "10 left" or "0 left" (index -1 aliasing); the walls on the board settle it:
    # w_mover + w_opponent = 20 - placed.  With both sides on the aliased plane and 10 walls placed,
    # (10, 0) and (0, 10) have IDENTICAL planes, so either choice re-encodes to the same state.
    total = 20 - (h.sum(axis=1) + v.sum(axis=1))
    am, ao = wm == 10, wo == 10
    wm = np.where(am & ~ao, np.where(total - wo >= 10, 10, 0), wm)
    wo = np.where(ao & ~am, np.where(total - wm >= 10, 10, 0), wo)
    both = am & ao
    wm = np.where(both, np.where(total >= 10, 10, 0), wm)
    wo = np.where(both, np.where(total >= 20, 10, 0), wo)
    out["p1"] = np.where(mover_is_p2, po, pm)
    out["p2"] = np.where(mover_is_p2, pm, po)
    out["w1"] = np.where(mover_is_p2, wo, wm)
    out["w2"] = np.where(mover_is_p2, wm, wo)
    out["cur"] = np.where(mover_is_p2, 2, 1)
    return out


# ------------------------------------------------------------------------------ device ring
class ReplayBuffer:
    """deque(maxlen=capacity) of (board, pi, z) tuples, device resident (train.py:23).

    ``extend`` appends in order and overwrites the oldest entries once full; ``len`` and
    ``sample_indices`` follow the deque's logical order (index 0 = oldest), so
    ``random.sample(buffer, k)`` of the reference and ``sample(k)`` here draw the same positions
    for the same ``random`` state."""

    def __init__(self, capacity=10000, device="cuda:0"):
        self.capacity = int(capacity)
        self.device = torch.device(device)
        self.words = torch.zeros((self.capacity, 3), dtype=torch.int64, device=self.device)  # hbits, vbits, meta
        self.pi = torch.zeros((self.capacity, 140), dtype=torch.float32, device=self.device)
        self.z = torch.zeros(self.capacity, dtype=torch.float32, device=self.device)
        self._head = 0      # physical slot of the next write
        self._size = 0
        self.total_added = 0

    def __len__(self):
        return self._size

    @property
    def maxlen(self):
        return self.capacity

    def clear(self):
        self._head = self._size = 0

    # ---- writing
    def _put(self, words, pi, z):
        n = int(pi.shape[0])
        if n == 0:
            return
        if n > self.capacity:  # only the newest `capacity` survive, like the deque
            words, pi, z = words[-self.capacity:], pi[-self.capacity:], z[-self.capacity:]
            n = self.capacity
        slots = (torch.arange(n, device=self.device) + self._head) % self.capacity
        self.words[slots] = words.to(self.device)
        self.pi[slots] = pi.to(self.device, torch.float32)
        self.z[slots] = z.to(self.device, torch.float32)
        self._head = (self._head + n) % self.capacity
        self._size = min(self.capacity, self._size + n)
        self.total_added += n

    def extend(self, data):
        """data: TupleBatch | packed uint8 [n,588] (torch / numpy) | list of reference-shaped
        (state [26,9,9], pi [140], z) tuples (Quoridor.start_self_play's output)."""
        if hasattr(data, "boards") and hasattr(data, "pi"):  # TupleBatch
            b = data.boards
            self._put(torch.stack([b.hbits, b.vbits, b.meta], dim=1), data.pi, data.z)
        elif isinstance(data, (torch.Tensor, np.ndarray)):
            buf = torch.as_tensor(data)
            hb, vb, meta, pi, z = qdist.unpack_tuples(buf.reshape(-1, TUPLE_BYTES))
            self._put(torch.stack([hb, vb, meta], dim=1), pi, z)
        else:
            data = list(data)
            if not data:
                return
            from . import _cabi

            packed = planes_to_packed(np.stack([np.asarray(d[0]) for d in data]))
            hb, vb, meta = (torch.from_numpy(x) for x in _cabi.packed_to_soa(packed))
            pi = torch.from_numpy(np.stack([np.asarray(d[1], dtype=np.float32) for d in data]))
            z = torch.tensor([float(d[2]) for d in data], dtype=torch.float32)
            self._put(torch.stack([hb, vb, meta], dim=1), pi, z)

    # ---- reading
    def _physical(self, logical):
        start = (self._head - self._size) % self.capacity
        return (torch.as_tensor(logical, dtype=torch.int64, device=self.device) + start) % self.capacity

    def gather(self, logical_indices):
        """-> (DeviceBoards, pi float32 [k,140], z float32 [k]) for deque positions `logical_indices`."""
        slots = self._physical(logical_indices)
        w = self.words[slots]
        boards = DeviceBoards(len(slots), self.device)
        boards.hbits, boards.vbits, boards.meta = w[:, 0].contiguous(), w[:, 1].contiguous(), w[:, 2].contiguous()
        return boards, self.pi[slots].contiguous(), self.z[slots].contiguous()

    def sample_indices(self, k, rng=random):
        """random.sample(range(len(buffer)), k): the positions random.sample(buffer, k) would pick
        (train.py:67) -- same consumption of the `random` module's state."""
        return rng.sample(range(self._size), k)

    def sample(self, k, rng=random):
        """-> (states float32 [k,26,9,9] re-encoded on the GPU, pi [k,140], z [k]), all device tensors."""
        from . import rules

        boards, pi, z = self.gather(self.sample_indices(k, rng))
        return rules.encode(boards), pi, z

    def packed(self) -> torch.Tensor:
        """The whole buffer, oldest first, as uint8 [len, 588] on the device (shard / wire format)."""
        slots = self._physical(torch.arange(self._size, device=self.device))
        w = self.words[slots]
        return qdist.pack_tuples(w[:, 0].contiguous(), w[:, 1].contiguous(), w[:, 2].contiguous(), self.pi[slots], self.z[slots])

    def reference_tuples(self, logical_indices):
        """list[(float64 [26,9,9], float64 [140], float64)] -- the reference's data_buffer entries (quoridor.py:610,
        train.py:23) -- at deque positions `logical_indices`; states re-encoded on the device (qz_encode)."""
        from . import rules

        logical_indices = list(logical_indices)
        if not logical_indices:
            return []
        boards, pi, z = self.gather(logical_indices)
        st = rules.encode(boards).cpu().numpy().astype(np.float64)
        pi = pi.cpu().numpy().astype(np.float64)
        z = z.cpu().numpy().astype(np.float64)
        return [(st[i], pi[i], z[i]) for i in range(len(logical_indices))]

    def to_reference_tuples(self):
        """the whole buffer, oldest first, like list(the reference's data_buffer)"""
        return self.reference_tuples(range(self._size))

    # ---- the deque's sequence protocol, so that code written against the reference's buffer -- `random.sample(
    # self.data_buffer, self.batch_size)` (train.py:67), indexing, iteration -- runs unchanged.  Every entry comes back in
    # the reference's shape; TrainPipeline.policy_update itself uses sample() (same positions, no host round trip).
    # COST: one element = one device gather + one qz_encode launch + three device-to-host copies, so `random.sample(buffer,
    # 512)` -- which indexes element by element -- is ~1,500 host round trips.  Batched forms, one gather + one launch each:
    # buffer[[i, j, ...]] / buffer[slice], reference_sample(k) (= random.sample(buffer, k): same positions, same consumption of
    # the `random` module's state), sample(k) (device tensors, what TrainPipeline.policy_update uses), iteration (chunks of 256).
    def reference_sample(self, k, rng=random):
        """random.sample(buffer, k) in ONE device round trip: the same positions in the same order"""
        return self.reference_tuples(self.sample_indices(k, rng))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.reference_tuples(range(*i.indices(self._size)))
        if isinstance(i, (list, tuple, np.ndarray, torch.Tensor)):  # an index list: one gather for all of them
            idx = [int(j) for j in (i.tolist() if hasattr(i, "tolist") else i)]
            idx = [j + self._size if j < 0 else j for j in idx]
            if any(not 0 <= j < self._size for j in idx):
                raise IndexError("ReplayBuffer index out of range")
            return self.reference_tuples(idx)
        i = int(i)
        if i < 0:
            i += self._size
        if not 0 <= i < self._size:
            raise IndexError("ReplayBuffer index out of range")
        return self.reference_tuples([i])[0]

    def __iter__(self):
        n = self._size  # (like a deque, the buffer must not be mutated during iteration)
        for lo in range(0, n, 256):
            for t in self.reference_tuples(range(lo, min(lo + 256, n))):
                yield t

    # ---- shards
    def save_shard(self, path, n_games=0, n_playout=0):
        return write_shard(path, self.packed(), n_games=n_games, n_playout=n_playout)

    def load_shard(self, path):
        packed, meta = read_shard(path)
        self.extend(packed)
        return meta


import collections.abc as _abc

_abc.Sequence.register(ReplayBuffer)  # random.sample() insists on a Sequence


class ShardWriter:
    """Self-play side of the hand-over: collects harvested tuples and cuts a shard file every
    `tuples_per_shard` tuples: <dir>/<prefix>-r<rank>-<serial>.qzr (written atomically, so a
    training job can pick up complete files only)."""

    def __init__(self, directory, prefix="selfplay", rank=0, tuples_per_shard=65536, n_playout=0):
        self.dir, self.prefix, self.rank = str(directory), prefix, int(rank)
        self.tuples_per_shard, self.n_playout = int(tuples_per_shard), int(n_playout)
        self.serial = 0
        self._chunks, self._n, self._games = [], 0, 0
        os.makedirs(self.dir, exist_ok=True)
        self.paths = []

    def add(self, tuple_batch):
        b = tuple_batch.boards
        self._chunks.append(qdist.pack_tuples(b.hbits, b.vbits, b.meta, tuple_batch.pi, tuple_batch.z).cpu())
        self._n += len(tuple_batch)
        self._games += int(tuple_batch.n_games)
        if self._n >= self.tuples_per_shard:
            self.flush()

    def flush(self):
        if not self._n:
            return None
        path = os.path.join(self.dir, "%s-r%d-%06d.qzr" % (self.prefix, self.rank, self.serial))
        write_shard(path, torch.cat(self._chunks), n_games=self._games, n_playout=self.n_playout)
        self.paths.append(path)
        self.serial += 1
        self._chunks, self._n, self._games = [], 0, 0
        return path
