"""Exploratory timings on one MI355X: leaf evaluator variants and the engine's step pieces.
Not the bench contract (see bench.py); prints a small table."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from alphazero_quoridor_amd.boards import DeviceBoards, opening_packed  # noqa: E402
from alphazero_quoridor_amd.engine import SelfPlayEngine  # noqa: E402
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet  # noqa: E402
from alphazero_quoridor_amd import rules  # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    net = PolicyValueNet(use_gpu=True)
    print("== leaf evaluator (ms per forward, 62.8 MFLOP/leaf)")
    for B in (1024, 4096, 16384):
        x = (torch.rand((B, 26, 9, 9), device=dev) > 0.8).float()
        for mode in ("per_leaf", "eval"):
            for dt, cl in ((torch.float32, False), (torch.float32, True), (torch.bfloat16, False), (torch.bfloat16, True)):
                try:
                    ev = net.evaluator(mode, dt, cl)
                    ms = timeit(lambda: ev(x), n=10)
                    print("B=%5d %-8s %-8s cl=%d  %8.3f ms  %7.1f TFLOP/s  %9.0f leaves/s" % (
                        B, mode, str(dt).split(".")[-1], cl, ms, B * 62.8e6 / ms / 1e9, B / ms * 1e3))
                except Exception as e:  # noqa: BLE001
                    print("B=%d %s %s cl=%d failed: %s" % (B, mode, dt, cl, str(e)[:100]))
    print("== rules kernels")
    from synth import synth_positions
    for name, kw in (("open", None), ("mid", dict(min_walls=0, max_walls=20, mover_has_walls=True)),
                     ("dense", dict(min_walls=16, max_walls=19, mover_has_walls=True))):
        for B in (4096, 32768):
            packed = opening_packed(B) if kw is None else np.resize(synth_positions(4096, seed=0x5EED, **kw), B)
            db = DeviceBoards.from_packed(packed, dev)
            mask = torch.empty((B, 5), dtype=torch.int32, device=dev)
            planes = torch.empty((B, 26, 9, 9), dtype=torch.float32, device=dev)
            t_f = timeit(lambda: rules.movegen_encode(db, mask, planes), n=30)
            t_e = timeit(lambda: rules.encode(db, planes), n=30)
            t_m = timeit(lambda: rules.movegen(db), n=30)
            print("%-5s B=%5d fused %7.1f us (%6.0f GB/s alg)  encode %7.1f us (%6.0f GB/s)  movegen %7.1f us" % (
                name, B, t_f * 1e3, B * 8468 / t_f / 1e6, t_e * 1e3, B * 8448 / t_e / 1e6, t_m * 1e3))
    print("== engine step pieces, B=4096, n_playout=400 config")
    B = 4096
    eng = SelfPlayEngine(B, n_playout=400, device=dev, seed=1)
    ev = net.evaluator("per_leaf", torch.float32, True)
    for _ in range(50):
        eng.playout_step(ev)
    p, v = ev(eng.planes)
    print("select+movegen+encode %.3f ms" % timeit(lambda: eng.select(), n=20))
    print("net                   %.3f ms" % timeit(lambda: ev(eng.planes), n=20))
    print("expand_backup         %.3f ms" % timeit(lambda: eng.expand_backup(p, v), n=20))
    print("full step eager       %.3f ms" % timeit(lambda: eng.playout_step(ev), n=50))
    try:
        eng.capture_steps(ev, 4, warmup=2)
        t = timeit(lambda: eng._graph.replay(), n=20) / 4
        print("full step graph(4)    %.3f ms" % t)
    except Exception as e:  # noqa: BLE001
        print("graph capture failed:", str(e)[:300])
    t0 = time.time()
    eng.finish_move()
    torch.cuda.synchronize()
    print("finish_move (first)   %.3f ms" % ((time.time() - t0) * 1e3))
    print(eng.stats())


if __name__ == "__main__":
    main()
