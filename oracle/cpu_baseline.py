"""CPU baseline leg of bench.py (test infrastructure, never on the product path): the oracle's
scalar C restatement of the reference's MCTS (one playout at a time, per-candidate double BFS)
with a batch-1 fp32 torch-CPU forward of the policy-value net per leaf, exactly the work
`MCTS._playout` does in the reference (mcts.py:103-127, policy_value_net.py:145-164).

    python -m oracle.cpu_baseline --seconds 8 --n-playout 400 [--seed 0]

prints one JSON line {"playouts": n, "seconds": t}.  bench.py starts one of these for the
1-core figure and os.cpu_count() of them side by side for the all-core figure (independent
searches, like the reference run as N processes).  One torch thread per process.
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


def run(seconds: float, n_playout: int, seed: int = 0) -> dict:
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

    g = oracle.OracleGame()
    m = oracle.OracleMCTS(policy, c_puct=5, n_playout=n_playout)
    for _ in range(3):
        m.playout(g)
    n, t0 = 0, time.time()
    while time.time() - t0 < seconds:
        for _ in range(10):
            m.playout(g)
        n += 10
    return {"playouts": n, "seconds": time.time() - t0}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--n-playout", type=int, default=400)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    print(json.dumps(run(a.seconds, a.n_playout, a.seed)))
