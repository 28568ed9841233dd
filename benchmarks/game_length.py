"""How long are reference-faithful self-play games at a given playout count?  bench.py's
steady-state games/s is plies/s / E[plies per game], so this run measures E[plies per game]:
`--boards` boards play continuously (finished boards restart at once) for `--seconds`; every
game contributes either its length (finished) or the plies it had played when the run stopped
(right-censored: its start time, hence its censoring time, does not depend on its length).

Estimator: Kaplan-Meier survival S(t) in plies; E[L] = integral of S up to the longest observed
time T plus S(T)/lambda, lambda = the hazard measured over [T/2, T] (finish events / plies at
risk) -- an exponential tail beyond the observation window; the per-bin hazards are printed so
the assumption can be checked.  95 % interval: bootstrap over games.

    python benchmarks/game_length.py --boards 512 --playouts 400 --seconds 900 --out profiles/round3/game_length_400playouts.json
"""
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


def km_mean(times, events):
    """(E[L] with exponential tail, restricted mean, S(T), lambda_tail, T)."""
    times = np.asarray(times, dtype=np.float64)
    events = np.asarray(events, dtype=bool)
    order = np.argsort(times, kind="stable")
    t, e = times[order], events[order]
    n = len(t)
    at_risk = n - np.arange(n)
    # survival just after each observation (ties: events before censorings at equal times does not matter for the mean at this resolution)
    factors = np.where(e, 1.0 - 1.0 / at_risk, 1.0)
    S = np.cumprod(factors)
    T = t[-1]
    # integral of the step function S over [0, T]
    prev_t = np.concatenate([[0.0], t[:-1]])
    prev_S = np.concatenate([[1.0], S[:-1]])
    rmst = float(np.sum(prev_S * (t - prev_t)))
    lo = T / 2.0
    exposure = float(np.sum(np.clip(t, lo, T) - lo))        # plies at risk inside [T/2, T]
    tail_events = int(np.sum(e & (t > lo)))
    lam = tail_events / exposure if exposure > 0 and tail_events > 0 else None
    ST = float(S[-1]) if not e[-1] else float(S[-1])
    mean = rmst + (ST / lam if lam else 0.0)
    return mean, rmst, ST, lam, float(T)


def window_doubling(times, events):
    """E[length] estimated from the first HALF of the observation window (every observation truncated at T/2) against the whole
    window: how much the estimate still moves when the window doubles."""
    times = np.asarray(times, dtype=np.float64)
    events = np.asarray(events, dtype=bool)
    T = float(times.max())
    half_t = np.minimum(times, T / 2.0)
    half_e = events & (times <= T / 2.0)
    if half_e.sum() < 2:
        return None
    m_half = km_mean(half_t, half_e)[0]
    m_full = km_mean(times, events)[0]
    return {"mean_at_half_window": m_half, "mean_at_full_window": m_full, "delta": m_full / m_half - 1.0, "half_window_plies": T / 2.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=512)
    ap.add_argument("--playouts", type=int, default=400)
    ap.add_argument("--seconds", type=float, default=600)
    ap.add_argument("--max-plies", type=int, default=1000000)
    ap.add_argument("--bn", default="per_leaf")
    ap.add_argument("--fix-sign", type=int, default=0)
    ap.add_argument("--budget-us", type=int, default=1000)
    ap.add_argument("--traj-pages-per-board", type=int, default=200, help="trajectory pool (64-KB pages): games of > 100,000 plies must not be dropped for want of room")
    ap.add_argument("--max-depth", type=int, default=992, help="drop a game whose search descends deeper (the reference's RecursionError); 0 = never")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(2026)
    torch.backends.cudnn.benchmark = True
    net = PolicyValueNet(use_gpu=True)
    ev = net.evaluator(args.bn)
    B = args.boards
    eng = SelfPlayEngine(B, n_playout=args.playouts, seed=5, device=dev, fix_terminal_sign=bool(args.fix_sign), max_depth=args.max_depth,
                         traj_pool_pages=args.traj_pages_per_board * B)
    # the asynchronous self-play loop (bit-identical games to the lock-step engine, tests/test_gpu_async.py): boards
    # restart at once when their game is harvested.  Per finished game: its length and the plies of its OPEN phase (the
    # mover still has walls: the part of a game whose leaves are almost all new to the memo)
    finished, finished_open = [], []
    dropped, dropped_seen = [], 0   # (ply at which a game was dropped, cause): right-censored observations
    t0 = time.time()
    rounds = 0
    stream = torch.cuda.Stream(device=dev)
    t_print, p_print = t0, 0
    with torch.cuda.stream(stream):
        while time.time() - t0 < args.seconds:
            eng.run_rounds(ev, 64, max_playouts=4096, budget_us=args.budget_us)
            rounds += 64
            tb = eng.harvest()
            if tb is not None:
                gid = tb.game.long()
                meta = tb.boards.meta
                cur = (meta >> 32) & 0xFF
                walls = torch.where(cur == 1, (meta >> 16) & 0xFF, (meta >> 24) & 0xFF)
                finished.extend(torch.bincount(gid, minlength=tb.n_games).tolist())
                finished_open.extend(torch.bincount(gid, weights=(walls > 0).double(), minlength=tb.n_games).tolist())
            if rounds % 1024 == 0:  # games the engine dropped (depth limit, root without a move): censored at their ply, not ignored
                packed, causes, plies, slots, total = eng.dropped_games()
                new = min(total - dropped_seen, len(plies))
                if new > 0:
                    dropped.extend(zip(plies[-new:].tolist(), causes[-new:]))
                dropped_seen = total
            if time.time() - t_print > 60:
                st = eng.stats()
                sys.stderr.write("t %.0fs rounds %d plies/s %.0f games %d mean depth %.1f max depth %d deep descents %d (levels %d) max edges %d pages in use %d\n"
                                 % (time.time() - t0, rounds, (st["plies_played"] - p_print) / (time.time() - t_print), len(finished),
                                    st["descent_levels"] / max(st["playouts"], 1), st["max_depth"], st["deep_descents"], st["deep_levels"], st["max_edges"], st["tree_pages_in_use"]))
                t_print, p_print = time.time(), st["plies_played"]
        seconds = time.time() - t0
        st = eng.stats()
        ply = int(st["plies_played"] // B)
        age = eng.get_plies().long()
        censored = age[age > 0].cpu().tolist()
        packed, causes, plies, slots, total = eng.dropped_games()
        new = min(total - dropped_seen, len(plies))
        if new > 0:
            dropped.extend(zip(plies[-new:].tolist(), causes[-new:]))
    # a dropped game is an observation "longer than the ply it was dropped at" (depth drops come with long forced lines:
    # informative censoring, so the estimate still leans low -- said in the output)
    censored_dropped = [max(int(p), 1) for p, _ in dropped]
    times = np.array(finished + censored + censored_dropped, dtype=np.float64)
    events = np.array([True] * len(finished) + [False] * (len(censored) + len(censored_dropped)))
    mean, rmst, ST, lam, T = km_mean(times, events)
    rng = np.random.RandomState(0)
    boots = []
    for _ in range(300):
        idx = rng.randint(0, len(times), len(times))
        if events[idx].sum() < 2:
            continue
        boots.append(km_mean(times[idx], events[idx])[0])
    ci = [float(np.percentile(boots, 2.5)), float(np.percentile(boots, 97.5))] if boots else None
    edges = np.linspace(0, T, 9)
    hazards = []
    for a, b in zip(edges[:-1], edges[1:]):
        exposure = float(np.sum(np.clip(times, a, b) - a))
        n_ev = int(np.sum(events & (times > a) & (times <= b)))
        hazards.append({"plies": [float(a), float(b)], "events": n_ev, "plies_at_risk": exposure, "hazard_per_ply": n_ev / exposure if exposure else None})
    out = {
        "boards": B, "n_playout": args.playouts, "max_depth": args.max_depth, "fix_terminal_sign": bool(args.fix_sign), "plies_run": ply, "seconds": seconds,
        "plies_per_s": st["plies_played"] / seconds,
        "games_finished": len(finished), "games_censored": len(censored), "games_dropped_as_censored": len(censored_dropped),
        "dropped_by_cause": {c: sum(1 for _, cc in dropped if cc == c) for c in sorted(set(c for _, c in dropped))},
        "dropped_ply_percentiles_10_50_90": [float(x) for x in np.percentile(censored_dropped, [10, 50, 90])] if censored_dropped else None,
        "dropped_mean_ply": float(np.mean(censored_dropped)) if censored_dropped else None,
        "finished_fraction_of_started": len(finished) / max(len(finished) + len(censored_dropped), 1),
        "window_doubling": window_doubling(times, events),
        "finished_length_percentiles_10_50_90": [float(x) for x in np.percentile(finished, [10, 50, 90])] if finished else None,
        "finished_mean": float(np.mean(finished)) if finished else None,
        "mean_open_plies_per_game": float(np.mean(finished_open)) if finished_open else None,
        "open_plies_percentiles_10_50_90": [float(x) for x in np.percentile(finished_open, [10, 50, 90])] if finished_open else None,
        "engine": "asynchronous self-play loop (qz_selfplay_*), %d rounds" % rounds,
        "estimator": "Kaplan-Meier restricted mean up to T = %.0f plies (%.0f) + S(T) / lambda with S(T) = %.3f and the hazard over [T/2, T] "
                     "lambda = %s per ply (exponential tail)" % (T, rmst, ST, ("%.3g" % lam) if lam else "n/a"),
        "mean_plies_per_game": mean, "mean_ci95": ci, "restricted_mean": rmst, "survival_at_T": ST, "tail_hazard_per_ply": lam, "T": T,
        "hazard_by_bin": hazards,
        "stats": {k: st[k] for k in ("games_finished", "plies_played", "playouts", "leaf_terminal", "node_overflow", "games_aborted",
                                     "aborted_no_move", "aborted_max_plies", "aborted_pool", "aborted_depth", "max_depth", "tree_pages_peak", "tree_pages_total",
                                     "traj_pages_peak", "traj_pages_total", "max_edges", "runaway_descents", "compact_slices", "memo_inserts", "memo_hits", "nn_evals", "rounds")},
    }
    text = json.dumps(out)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
