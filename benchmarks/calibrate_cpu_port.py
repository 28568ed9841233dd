#!/usr/bin/env python3
"""Calibration of bench.py's cpu_baseline (CPU baseline plan (2)): the REAL
reference (pure Python, /root/reference) and the oracle's C port timed on the same host, same
work: playouts of the first ply from the opening with a random-init policy_value_net evaluated
on the CPU, batch of one, one torch thread.  Runs only in the build container (the reference
cannot travel); the ratios (opening ply + late phase of a steady-state population) go to profiles/round6/cpu_calibration.json, which bench.py reads to
turn the port's number on the GPU box's host into an estimate of "the reference on that host".

    python benchmarks/calibrate_cpu_port.py [--seconds 30]
"""
import argparse
import contextlib
import copy
import io
import json
import os
import platform
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def time_reference_late(seconds, n_playout, boards):
    """the reference's MCTS.get_move_probs loop (mcts.py:135-139) on steady-state LATE roots (mover without walls), fresh tree per position"""
    sys.path.insert(0, os.environ.get("QZ_REFERENCE", "/root/reference"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    warnings.filterwarnings("ignore")
    import torch

    torch.set_num_threads(1)
    torch.manual_seed(0)
    from gen_golden import game_from_packed
    from mcts import MCTS
    from policy_value_net import PolicyValueNet

    pvn = PolicyValueNet(use_gpu=False)
    n, k, t0 = 0, 0, time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        while time.time() - t0 < seconds:
            game = game_from_packed(boards[k % len(boards)])
            m = MCTS(pvn.policy_value_fn, c_puct=5, n_playout=n_playout)
            done = 0
            try:
                while done < n_playout and time.time() - t0 < seconds:
                    m._playout(copy.deepcopy(game))
                    done += 1
            except IndexError:  # quirk Q5: the reference's own crash on an off-board winning jump
                pass
            n += done
            k += 1
    return n, time.time() - t0


def time_reference(seconds, n_playout):
    sys.path.insert(0, os.environ.get("QZ_REFERENCE", "/root/reference"))
    warnings.filterwarnings("ignore")
    import torch

    torch.set_num_threads(1)
    torch.manual_seed(0)
    from mcts import MCTS
    from policy_value_net import PolicyValueNet
    from quoridor import Quoridor

    pvn = PolicyValueNet(use_gpu=False)
    game = Quoridor()
    m = MCTS(pvn.policy_value_fn, c_puct=5, n_playout=n_playout)
    with contextlib.redirect_stdout(io.StringIO()):
        for _ in range(3):
            m._playout(copy.deepcopy(game))
        n, t0 = 0, time.time()
        while time.time() - t0 < seconds:
            m._playout(copy.deepcopy(game))  # mcts.py:135-137: every playout runs on a deepcopy
            n += 1
    return n, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--n-playout", type=int, default=400)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "round6", "cpu_calibration.json"))
    a = ap.parse_args()
    rn, rt = time_reference(a.seconds, a.n_playout)
    from oracle.cpu_baseline import run

    port = run(a.seconds, a.n_playout)
    cpu = ""
    try:
        cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    out = {
        "what": "playouts/s of the first ply from the opening (131 legal moves), random-init net on the CPU, batch 1, 1 torch thread, 1 process",
        "host_cpu": cpu, "host_cores": os.cpu_count(), "python": platform.python_version(),
        "reference_playouts": rn, "reference_seconds": rt, "reference_playouts_per_s": rn / rt,
        "port_playouts": port["playouts"], "port_seconds": port["seconds"], "port_playouts_per_s": port["playouts"] / port["seconds"],
    }
    out["port_over_reference"] = out["port_playouts_per_s"] / out["reference_playouts_per_s"]
    # the same on the LATE phase of a steady-state population (tests/golden/steady_state_roots.npz: roots whose mover has no wall left --
    # there the reference's actions() costs microseconds, not 97 ms, and the batch-1 network forward dominates both sides)
    import numpy as np

    roots = np.load(os.path.join(ROOT, "tests", "golden", "steady_state_roots.npz"))["board"]
    late = roots[np.where(roots["cur"] == 1, roots["w1"], roots["w2"]) == 0]
    ln, lt = time_reference_late(a.seconds, a.n_playout, late)
    port_late = run(a.seconds, a.n_playout, 0, "late")
    out.update({"late_what": "playouts/s of %d-playout searches from fresh trees on steady-state roots whose mover has no wall left (%d positions in the file)" % (a.n_playout, len(late)),
                "reference_late_playouts_per_s": ln / lt, "port_late_playouts_per_s": port_late["playouts"] / port_late["seconds"]})
    out["port_over_reference_late"] = out["port_late_playouts_per_s"] / out["reference_late_playouts_per_s"]
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
