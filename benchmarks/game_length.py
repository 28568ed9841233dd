"""How long are self-play games at a given playout count?  (bench.py's games/s depends on
it: steady-state games/s = plies/s / mean plies per game.)  Plays `--boards` games for up to
`--plies` plies and reports the length distribution of the FIRST generation of games (those
started at ply 0), plus how many of them are still running when the run stops."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from alphazero_quoridor_amd.engine import SelfPlayEngine  # noqa: E402
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=512)
    ap.add_argument("--playouts", type=int, default=400)
    ap.add_argument("--plies", type=int, default=1200)
    ap.add_argument("--seconds", type=float, default=600)
    ap.add_argument("--bn", default="per_leaf")
    ap.add_argument("--fix-sign", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(2026)
    torch.backends.cudnn.benchmark = True
    net = PolicyValueNet(use_gpu=True)
    ev = net.evaluator(args.bn)
    eng = SelfPlayEngine(args.boards, n_playout=args.playouts, seed=5, device=dev, fix_terminal_sign=bool(args.fix_sign),
                         max_plies=args.plies + 8)
    first = []
    lengths_all = []
    t0 = time.time()
    ply = 0
    for ply in range(1, args.plies + 1):
        eng.run_playouts(ev)
        eng.finish_move()
        tb = eng.harvest()
        if tb is not None:
            lens = np.bincount(tb.game.cpu().numpy()).tolist()
            lengths_all.extend(lens)
            first.extend(L for L in lens if L == ply)  # started at ply 0 <=> length == elapsed plies
        if time.time() - t0 > args.seconds:
            break
    st = eng.stats()
    fl = np.array(first)
    print(json.dumps({
        "boards": args.boards, "n_playout": args.playouts, "plies_run": ply, "seconds": time.time() - t0,
        "first_generation_finished": int(len(fl)), "first_generation_unfinished": int(args.boards - len(fl)),
        "first_gen_mean_len_finished": float(fl.mean()) if len(fl) else None,
        "first_gen_percentiles_10_50_90": [float(x) for x in np.percentile(fl, [10, 50, 90])] if len(fl) else None,
        "all_finished_games": len(lengths_all), "all_mean_len": float(np.mean(lengths_all)) if lengths_all else None,
        "stats": st, "fix_sign": args.fix_sign,
    }))


if __name__ == "__main__":
    main()
