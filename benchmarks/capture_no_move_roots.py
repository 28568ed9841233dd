"""Harvest root positions at which the engine dropped a game because the mover has NO legal move (qz_stats.aborted_no_move):
reference-faithful play at 4 playouts per move (bench.py's desynchronisation phase, where such roots turn up in ~0.6 % of the
games), collected from the engine's drop log (qz_engine_dropped_games).  The boards go to tests/golden/no_move_roots_input.npy;
tests/golden/gen_golden.py then asks the REFERENCE about each of them (actions() == [], choose_action prints "the board is
full") and writes the fixture the -m gpu test reads."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=4096)
    ap.add_argument("--playouts", type=int, default=4)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--want", type=int, default=64)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "no_move_roots_input.npy"))
    args = ap.parse_args()
    from alphazero_quoridor_amd.engine import SelfPlayEngine
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = PolicyValueNet(use_gpu=True, device=dev)
    ev = net.evaluator("per_leaf")
    eng = SelfPlayEngine(args.boards, n_playout=args.playouts, seed=1, device=dev, max_depth=992)
    seen, t0 = {}, time.time()
    while time.time() - t0 < args.seconds and len(seen) < args.want:
        eng.run_rounds(ev, 256, max_playouts=64, budget_us=1000)
        eng.harvest()
        packed, causes, plies, slots, total = eng.dropped_games()
        for rec, c, p in zip(packed, causes, plies):
            if c == "no_legal_move":
                seen.setdefault(rec.tobytes(), (rec, int(p)))
    st = eng.stats()
    recs = np.array([v[0] for v in seen.values()], dtype=packed.dtype) if seen else np.zeros(0, dtype=packed.dtype)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    np.save(args.out, recs)
    print("%d distinct no-legal-move roots (plies %s) in %.0f s: %d games finished, %d dropped without a move, %d for depth"
          % (len(recs), sorted(v[1] for v in seen.values())[:8], time.time() - t0, st["games_finished"], st["aborted_no_move"], st["aborted_depth"]))
    eng.close()


if __name__ == "__main__":
    main()
