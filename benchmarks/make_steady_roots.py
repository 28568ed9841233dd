#!/usr/bin/env python3
"""tests/golden/steady_state_roots.npz: the root positions of a steady-state population of the engine -- the workload the GPU's
bench number is quoted on -- for cpu_baseline.by_phase (oracle/cpu_baseline.py --phase open | late).

    python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-c3 --second-line-seconds 0 --dump-roots roots_all.npz     (GPU box)
    python benchmarks/make_steady_roots.py roots_all.npz                                                                 (anywhere)

The dump holds every board's root after the bench's desync + settle + warm-up + timed rounds (13,312 boards, ~4,000 rounds at 400
playouts per move).  A seeded sample of 512 live roots keeps the population's phase mix; the file records it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def main():
    d = np.load(sys.argv[1])
    b = d["board"]
    live = np.array([not oracle.OracleGame.from_packed(r).has_a_winner()[0] for r in b])
    b = b[live]
    rng = np.random.RandomState(2026)
    pick = b[np.sort(rng.choice(len(b), size=512, replace=False))]
    mover_walls = np.where(pick["cur"] == 1, pick["w1"], pick["w2"])
    out = os.path.join(ROOT, "tests", "golden", "steady_state_roots.npz")
    np.savez_compressed(out, board=pick, population=len(b), boards_per_gpu=int(d["boards_per_gpu"]), n_playout=int(d["n_playout"]),
                        rounds_played=int(d["rounds_played"]), bench_seed=int(d["seed"]),
                        mover_has_walls_in_population=float(np.mean(np.where(b["cur"] == 1, b["w1"], b["w2"]) > 0)),
                        nobody_has_walls_in_population=float(np.mean((b["w1"] == 0) & (b["w2"] == 0))))
    print("%d of %d live roots; mover has walls: %d of the sample (%.3f of the population); nobody has walls: %.3f of the population; walls on the board: mean %.1f"
          % (len(pick), len(b), int((mover_walls > 0).sum()), float(np.mean(np.where(b["cur"] == 1, b["w1"], b["w2"]) > 0)),
             float(np.mean((b["w1"] == 0) & (b["w2"] == 0))), float(np.mean(20 - b["w1"].astype(int) - b["w2"].astype(int)))))


if __name__ == "__main__":
    main()
