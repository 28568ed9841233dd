"""C3 microbenchmark: the fused move-generation + encoder kernel on 32,768
boards, three position sets generated ON THE GPU by random legal play with the rules kernels:
  S-open   the opening position (131 legal moves, 10/10 walls left)
  S-mid    k ~ U{0..20} plies of random legal play biased to walls, mover has >= 1 wall
  S-dense  16..19 walls on the board, mover has >= 1 wall
Prints one JSON line per set: average launch time over >= 50 launches (HIP events),
algorithmic GB/s (8,468 B/board) and the fraction of the 8 TB/s HBM peak.
`--only S-mid --launches 20` is what the rocprofv3 PMC passes run."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if os.environ.get("QZ_BENCH_LIB"):  # A/B of a differently built library (benchmarks only): before anything loads it
    from alphazero_quoridor_amd import _cabi as _c  # noqa: E402

    _c.LIB_PATH = os.environ["QZ_BENCH_LIB"]
from alphazero_quoridor_amd import rules  # noqa: E402
from alphazero_quoridor_amd.boards import DeviceBoards, opening_packed  # noqa: E402

BYTES = 24 + 20 + 26 * 81 * 4


def mask_bool(mask):
    bits = torch.arange(32, device=mask.device, dtype=torch.int64)
    m = (mask.to(torch.int64) & 0xFFFFFFFF).unsqueeze(-1) >> bits
    return (m & 1).reshape(mask.shape[0], 160)[:, :140].bool()


def random_play(n, dev, seed, min_plies, max_plies, wall_weight, need_walls_placed=None):
    """Random legal play on the GPU.  Every board plays its own number of plies."""
    g = torch.Generator(device=dev).manual_seed(seed)
    db = DeviceBoards.from_packed(opening_packed(n), dev)
    target = torch.randint(min_plies, max_plies + 1, (n,), device=dev, generator=g)
    w = torch.ones(140, device=dev)
    w[12:] = wall_weight
    for ply in range(max_plies):
        legal = mask_bool(rules.movegen(db)).float() * w
        legal[:, 0] += 1e-9  # keep multinomial happy on boards with no move (never stepped)
        a = torch.multinomial(legal, 1, generator=g).squeeze(1)
        meta = db.meta
        placed = 20 - ((meta >> 16) & 0xFF) - ((meta >> 24) & 0xFF)
        stop = ply >= target
        if need_walls_placed is not None:
            stop = placed >= need_walls_placed[0] + (target % (need_walls_placed[1] - need_walls_placed[0] + 1))
        over = ((meta & 0xFF).to(torch.int8).to(torch.int64) > 71) | (((meta >> 8) & 0xFF).to(torch.int8).to(torch.int64) < 9)
        a = torch.where(stop | over, torch.full_like(a, 255), a)
        rules.step(db, a)
    return db


def fix_mover_walls(db):
    """Make sure the side to move has at least one wall (gives it the opponent's if not)."""
    meta = db.meta
    cur = (meta >> 32) & 0xFF
    w1, w2 = (meta >> 16) & 0xFF, (meta >> 24) & 0xFF
    need1 = (cur == 1) & (w1 == 0)
    need2 = (cur == 2) & (w2 == 0)
    meta = torch.where(need1, meta + (1 << 16), meta)
    meta = torch.where(need2, meta + (1 << 24), meta)
    db.meta = meta.contiguous()
    return db


def live_only(db, n):
    meta = db.meta
    p1 = (meta & 0xFF).to(torch.int8).to(torch.int64)
    p2 = ((meta >> 8) & 0xFF).to(torch.int8).to(torch.int64)
    keep = torch.nonzero((p1 <= 71) & (p2 >= 9)).squeeze(1)
    idx = keep[torch.arange(n, device=meta.device) % keep.numel()]
    out = DeviceBoards(n, meta.device)
    out.hbits, out.vbits, out.meta = db.hbits[idx].contiguous(), db.vbits[idx].contiguous(), db.meta[idx].contiguous()
    return out


def position_set(name, n, dev):
    if name == "S-open":
        return DeviceBoards.from_packed(opening_packed(n), dev)
    if name == "S-mid":
        return live_only(fix_mover_walls(random_play(n, dev, 0x5EED, 0, 20, 8.0)), n)
    if name == "S-dense":
        return live_only(fix_mover_walls(random_play(n, dev, 0x5EED + 1, 40, 40, 50.0, need_walls_placed=(16, 19))), n)
    raise ValueError(name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=32768)
    ap.add_argument("--launches", type=int, default=60)
    ap.add_argument("--only", default="S-open,S-mid,S-dense")
    ap.add_argument("--variant", type=int, default=0, help="qz_rules_opts.variant: 0 by size (default), 2|3|4 wave-per-board, 8..32 pooled tile size")
    ap.add_argument("--enc-split", type=int, default=-1, help="A/B: percent of the encoder tiles launched beside the path search")
    ap.add_argument("--detour", type=int, default=-1, help="A/B: pool_k1 detour_mode, pooled + 3 * wave-per-board (0..8)")
    args = ap.parse_args()
    opts = rules.rules_opts(args.variant, None if args.detour < 0 else args.detour % 3, None if args.detour < 0 else args.detour // 3,
                            None if args.enc_split < 0 else max(args.enc_split, 1))
    dev = torch.device("cuda:0")
    n = args.boards
    mask = torch.empty((n, 5), dtype=torch.int32, device=dev)
    planes = torch.empty((n, 26, 9, 9), dtype=torch.float32, device=dev)
    for name in args.only.split(","):
        db = position_set(name, n, dev)
        for _ in range(5):
            rules.movegen_encode(db, mask, planes, opts=opts)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.launches)]
        for a, b in evs:
            a.record()
            rules.movegen_encode(db, mask, planes, opts=opts)
            b.record()
        torch.cuda.synchronize()
        us = float(np.mean([a.elapsed_time(b) for a, b in evs])) * 1e3
        legal = mask_bool(mask).sum(dim=1).float()
        meta = db.meta
        placed = (20 - ((meta >> 16) & 0xFF) - ((meta >> 24) & 0xFF)).float()
        gbs = n * BYTES / us / 1e3
        print(json.dumps({"set": name, "variant": args.variant, "enc_split": args.enc_split, "boards": n, "launches": args.launches, "avg_launch_us": us,
                          "algorithmic_GBps": gbs, "frac_of_8TBps": gbs / 8000.0, "boards_per_s": n / us * 1e6,
                          "mean_legal_actions": float(legal.mean()), "mean_walls_placed": float(placed.mean())}))


if __name__ == "__main__":
    main()
