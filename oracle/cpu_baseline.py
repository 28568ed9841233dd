"""CPU baseline leg of bench.py (test infrastructure, never on the product path): the oracle's
scalar C restatement of the reference's MCTS (one playout at a time, per-candidate double BFS)
with a batch-1 fp32 torch-CPU forward of the policy-value net per leaf, exactly the work
`MCTS._playout` does in the reference (mcts.py:103-127, policy_value_net.py:145-164).

    python -m oracle.cpu_baseline --seconds 8 --n-playout 400 [--seed 0] [--phase opening|open|late]

prints one JSON line {"playouts": n, "seconds": t, ...}.  bench.py starts one of these for the
1-core figure and os.cpu_count() of them side by side for the all-core figure (independent
searches, like the reference run as N processes).  One torch thread per process.

--phase opening: the first ply from the opening (131 legal moves), as in rounds 1-5.
--phase open | late: n_playout-playout searches from fresh trees at the root positions of a
STEADY-STATE population (tests/golden/steady_state_roots.npz: roots harvested from a sustained
run of the GPU engine, `python bench.py --dump-roots`), those whose mover still has walls /
the others -- the workload the GPU's number is quoted on, phase by phase.  Process `seed`
starts at position `seed` of its phase's list and walks on (wrapping), so that side-by-side
processes search different positions.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


STEADY_ROOTS = os.path.join(ROOT, "tests", "golden", "steady_state_roots.npz")


def run(seconds: float, n_playout: int, seed: int = 0, phase: str = "opening") -> dict:
    import numpy as np
    import torch

    import oracle
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

    torch.set_num_threads(1)
    torch.manual_seed(seed)
    net = PolicyValueNet(use_gpu=False)
    mod = net.policy_value_net  # train mode, batch of one: the reference's behaviour

    def policy(game, legal):
        x = torch.from_numpy(game.state().reshape(1, 26, 9, 9).astype(np.float32))
        with torch.no_grad():
            logp, v = mod(x)
        p = np.exp(logp.numpy().reshape(-1))
        return legal, p[legal], float(v.reshape(-1)[0])

    if phase == "opening":
        g = oracle.OracleGame()
        m = oracle.OracleMCTS(policy, c_puct=5, n_playout=n_playout)
        for _ in range(3):
            m.playout(g)
        n, t0 = 0, time.time()
        while time.time() - t0 < seconds:
            for _ in range(10):
                m.playout(g)
            n += 10
        return {"playouts": n, "seconds": time.time() - t0, "phase": phase, "positions": 1, "mean_root_moves": float(len(g.actions()))}
    d = np.load(STEADY_ROOTS)
    boards = d["board"]
    mover_walls = np.where(boards["cur"] == 1, boards["w1"], boards["w2"])
    boards = boards[mover_walls > 0] if phase == "open" else boards[mover_walls == 0]
    if len(boards) == 0:
        raise SystemExit("no %s-phase position in %s" % (phase, STEADY_ROOTS))
    # (one warm-up search outside the timed region: torch's first forward, the oracle's tables)
    oracle.OracleMCTS(policy, c_puct=5, n_playout=4).get_move_probs(oracle.OracleGame.from_packed(boards[seed % len(boards)]), 1.0)
    n, k, moves, t0 = 0, 0, [], time.time()
    while time.time() - t0 < seconds:
        g = oracle.OracleGame.from_packed(boards[(seed + k) % len(boards)])
        m = oracle.OracleMCTS(policy, c_puct=5, n_playout=n_playout)
        moves.append(len(g.actions()))
        done = 0
        while done < n_playout and time.time() - t0 < seconds:  # (a search that the clock cuts short still counts what it did)
            for _ in range(min(10, n_playout - done)):
                m.playout(g)
            done += min(10, n_playout - done)
        n += done
        k += 1
    return {"playouts": n, "seconds": time.time() - t0, "phase": phase, "positions": k, "mean_root_moves": float(np.mean(moves))}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--n-playout", type=int, default=400)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--phase", default="opening", choices=["opening", "open", "late"])
    a = ap.parse_args()
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    print(json.dumps(run(a.seconds, a.n_playout, a.seed, a.phase)))
