#!/usr/bin/env python3
"""bench.py -- self-play games/sec on N MI355X GPUs (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W
    N > 1: either launched by torch.distributed.run (one rank per GPU, RCCL; RANK / WORLD_SIZE in
    the environment), or started plainly -- then this process only spawns the N ranks itself
    (python -m torch.distributed.run ... bench.py, before anything here touches a GPU), relays
    rank 0's JSON line and exits with the ranks' status.

Workload (config.workload): BASELINE.json configs[3] per GPU = 4096 concurrent boards,
n_playout=400, 9x9, reference defaults (10 walls, c_puct=5, temp=1, Dirichlet 0.3/0.25),
random-init policy_value_net evaluated in fp32 with the reference's per-leaf BatchNorm
statistics; weak scaling (4096 boards on every rank), finished tuples all-gathered every ply.

A STEP is one ply of every board: 400 playout steps (select -> movegen+encode -> net ->
expand/backup on the whole 4096-leaf batch) + finish_move + harvest (+ all-gather).  Nothing
inside a step is skipped or cached.

games/sec needs finished games, and a 400-playout game takes ~20 min of wall clock, so the
boards are first DESYNCHRONISED (untimed): they play `--desync-plies` plies with
`--desync-playouts` playouts per move so that the population is spread over all game phases
(continuous refill).  value = games that finished inside the K timed steps / their wall
time; plies/s, playouts/s and the mean length of the games seen are reported next to it
(plies/s is the robust statistic: game length varies 4-10x).

roofline: the move-generation + encoder op of every playout step (qz_mcts_leaf_inputs: the
single-launch k_wave_rules below 8,192 boards, the pooled pipeline k_pool_stage1 + k_pool_masks
from there on), timed with HIP events around every one of its invocations in
the timed region on the launch stream; algorithmic bytes =
8,468 B/board (24 B board + 20 B mask + 26*81*4 B planes) x 4096 boards.  `traffic` is the
PMC-measured HBM traffic of the same op at the same size (profiles/round1/pmc_traffic.json,
collected with rocprofv3 --pmc in separate passes), or null if that file is absent.
cpu_baseline: the CPU oracle (oracle/, scalar C port of the reference's algorithm, one
playout at a time, batch-1 network on the CPU like the reference) timed on this host.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_BOARD = 24 + 20 + 26 * 81 * 4


def cpu_baseline(seconds, mean_plies_per_game, n_playout):
    """Scalar port of the reference on one host core: playouts/s at n_playout=400 from the
    opening, batch-1 network forward on the CPU per leaf (policy_value_net.py:145-164)."""
    import oracle
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

    torch.set_num_threads(1)
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
    dt = time.time() - t0
    playouts_s = n / dt
    games_s = playouts_s / n_playout / max(mean_plies_per_game, 1.0)
    return {
        "value": games_s, "unit": "games/s", "cores": 1, "kind": "port",
        "sample": "%d playouts in %.1fs of the first ply at n_playout=%d from the opening (131 legal moves), oracle C port + "
                  "batch-1 fp32 torch-CPU forward per leaf; games/s = playouts/s / %d / %.0f plies per game (mean of the GPU run)"
                  % (n, dt, n_playout, n_playout, mean_plies_per_game),
        "playouts_per_s": playouts_s,
    }


def c3_microbench(dev, launches=60):
    """BASELINE configs[2] / SURVEY C3: the fused actions() + state() op on 32,768 mid-game boards
    (random legal play from the opening, mover has a wall left), outside the timed region.
    Reported next to `roofline` (which is the same op on the 4,096-leaf batches of the timed
    region) because at this size the pooled pipeline runs instead of the wave-per-board kernel."""
    sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
    from movegen_bench import position_set
    from alphazero_quoridor_amd import rules

    n = 32768
    db = position_set("S-mid", n, dev)
    mask = torch.empty((n, 5), dtype=torch.int32, device=dev)
    planes = torch.empty((n, 26, 9, 9), dtype=torch.float32, device=dev)
    for _ in range(5):
        rules.movegen_encode(db, mask, planes)
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
    for a, b in evs:
        a.record()
        rules.movegen_encode(db, mask, planes)
        b.record()
    torch.cuda.synchronize(dev)
    us = sum(a.elapsed_time(b) for a, b in evs) / launches * 1e3
    gbs = n * BYTES_PER_BOARD / us / 1e3
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "round1", "pmc_traffic_c3.json")) as f:
            traffic = json.load(f)["traffic_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return {"workload": "BASELINE configs[2] microbenchmark: 32,768 boards (S-mid: 0..20 plies of random legal play, mover has a wall), "
                        "actions() + state(), inputs resident in HBM, NOT part of the timed region",
            "kernel": "k_pool_paths_enc + k_pool_masks_enc (pooled pipeline, two launches)",
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
            "avg_launch_us": us, "launches": launches, "algorithmic_bytes_per_launch": n * BYTES_PER_BOARD}


def launch_ranks(n_gpus, argv):
    """`python bench.py --gpus N` without a launcher: become the launcher.  The ranks are CHILD
    processes (this process never initialises HIP and never re-execs); stdout of the job is the
    single JSON line rank 0 prints, everything else goes to stderr."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith("{") and line is None:
            line = out.rstrip("\n")
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks exited cleanly but printed no JSON line\n")
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--boards", type=int, default=4096)
    ap.add_argument("--playouts", type=int, default=400)
    ap.add_argument("--groups", type=int, default=1,
                    help="split the boards of a GPU into this many independent groups on their own HIP streams")
    ap.add_argument("--bn", default="per_leaf", choices=["per_leaf", "eval", "batch"])
    ap.add_argument("--nn-dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--channels-last", type=int, default=1)
    ap.add_argument("--desync-plies", type=int, default=700)
    ap.add_argument("--desync-playouts", type=int, default=4)
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fix-terminal-sign", action="store_true",
                    help="NOT the headline: back a winning move up as +1 (the reference backs it up as -1, mcts.py:125, which makes "
                         "searches avoid winning and games run for thousands of plies)")
    ap.add_argument("--no-c3", action="store_true", help="skip the 32,768-board microbenchmark line (roofline_c3)")
    args = ap.parse_args()
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    from alphazero_quoridor_amd import dist as qdist
    from alphazero_quoridor_amd.engine import BoardGroups
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

    rank, local, world = qdist.init_from_env("cuda")
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU path)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(args.seed)  # identical random-init weights on every rank
    net = PolicyValueNet(use_gpu=True, device=dev)
    dt = torch.float32 if args.nn_dtype == "fp32" else torch.bfloat16
    eng = BoardGroups(args.boards, args.groups, lambda: net.evaluator(args.bn, dt, bool(args.channels_last)),
                      seed=qdist.shard_seed(args.seed, rank), device=dev,
                      n_playout=args.playouts, c_puct=5, temp=1.0, is_selfplay=1, fix_terminal_sign=args.fix_terminal_sign)
    group_boards = args.boards // args.groups
    is_dist = world > 1

    def barrier():
        if is_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    lengths = []

    def end_of_ply():
        eng.finish_move()
        tbs = eng.harvest()
        n_games = 0
        for tb in tbs:
            n_games += tb.n_games
            gid = tb.game.cpu().numpy()
            assert gid.size and 0 <= int(gid.min()) and int(gid.max()) < tb.n_games, "corrupt game ids in a harvest"
            lengths.extend(np.bincount(gid, minlength=tb.n_games).tolist())
        if is_dist:  # the path's only exchange: finished tuples -> every rank's replay buffer
            eng.synchronize()
            bufs = [qdist.pack_tuples(tb.boards.hbits, tb.boards.vbits, tb.boards.meta, tb.pi, tb.z) for tb in tbs]
            bufs.append(torch.zeros((0, qdist.TUPLE_BYTES), dtype=torch.uint8, device=dev))
            qdist.allgather_tuples(torch.cat(bufs))
        return n_games

    # ---- desynchronise the games (untimed)
    t0 = time.time()
    for _ in range(args.desync_plies):
        eng.run_playouts(args.desync_playouts)
        end_of_ply()
    desync_s = time.time() - t0
    desync_games = len(lengths)

    # ---- warmup steps at the full playout count (untimed)
    for _ in range(args.warmup):
        eng.run_playouts()
        end_of_ply()

    # ---- timed region
    n_launch = args.steps * args.playouts
    evs = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.groups)]
           for _ in range(n_launch)]
    st0 = eng.stats()
    barrier()
    t0 = time.perf_counter()
    games = 0
    k = 0
    for _ in range(args.steps):
        for _ in range(args.playouts):
            eng.playout_step(events=evs[k], write_planes=True)
            k += 1
        games += end_of_ply()
    barrier()
    elapsed = time.perf_counter() - t0
    st1 = eng.stats()

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    tot = torch.tensor([games, st1["plies_played"] - st0["plies_played"], st1["playouts"] - st0["playouts"],
                        st1["leaf_terminal"] - st0["leaf_terminal"]], dtype=torch.float64, device=dev)
    if is_dist:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(tot, op=torch.distributed.ReduceOp.SUM)
    elapsed = float(el.item())
    games_all, plies_all, playouts_all, term_all = (float(x) for x in tot.tolist())

    n_evs = len(evs) * args.groups
    kern_ms = sum(a.elapsed_time(b) for row in evs for a, b in row) / n_evs
    achieved = group_boards * BYTES_PER_BOARD / (kern_ms * 1e-3) / 1e9
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "round1", "pmc_traffic.json")) as f:
            t = json.load(f)
        if int(t.get("boards", -1)) == group_boards:
            traffic = t["traffic_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

    if rank == 0:
        mean_len = float(np.mean(lengths)) if lengths else float("nan")
        out = {
            "metric": "self-play games/sec (9x9, n_playout=%d)" % args.playouts,
            "value": games_all / elapsed,
            "unit": "games/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 bitboards + f64 PUCT (rules/tree kernels); %s policy-value net" % args.nn_dtype,
            "data": "synthetic (random-init policy_value_net, seed %d; self-generated games)" % args.seed,
            "config": {
                "workload": "BASELINE configs[3] per GPU: %d concurrent boards/GPU, n_playout=%d, 9x9, 10 walls/player, "
                            "c_puct=5, temp=1.0, leaf batch=%d, finished tuples all-gathered every ply"
                            % (args.boards, args.playouts, args.boards),
                "boards_per_gpu": args.boards, "board_groups": args.groups, "fix_terminal_sign": bool(args.fix_terminal_sign), "n_playout": args.playouts, "bn_mode": args.bn,
                "nn_dtype": args.nn_dtype, "channels_last": bool(args.channels_last),
                "step": "one ply of every board (n_playout playout steps + finish_move + harvest)",
                "desync": "%d untimed plies at %d playouts/move (%.0fs, %d games finished)"
                          % (args.desync_plies, args.desync_playouts, desync_s, desync_games),
            },
            "games_in_timed_region": games_all,
            "plies_per_s": plies_all / elapsed,
            "playouts_per_s": playouts_all / elapsed,
            "leaf_evals_per_s": playouts_all / elapsed,
            "terminal_leaf_frac": term_all / max(playouts_all, 1.0),
            "mean_plies_per_game": mean_len,
            "mean_descent_depth": (st1["descent_levels"] - st0["descent_levels"]) / max(st1["playouts"] - st0["playouts"], 1),
            "games_per_s_from_plies": (plies_all / elapsed) / mean_len if lengths else None,
            "roofline": {
                "kernel": ("k_wave_rules (fused Quoridor.actions() + state() of the leaf batch: one wave per board + encoder groups, one launch)"
                           if group_boards < 8192 else
                           "k_pool_stage1 + k_pool_masks (Quoridor.actions() + state() of the leaf batch, pooled, two launches)"),
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "avg_launch_us": kern_ms * 1e3, "launches": n_evs,
                "algorithmic_bytes_per_launch": group_boards * BYTES_PER_BOARD,
            },
            "engine_stats": {k: st1[k] for k in ("node_overflow", "games_aborted", "arena_bytes", "max_nodes", "max_edges")},
        }
        if world == 1 and not args.no_c3:
            out["roofline_c3"] = c3_microbench(dev)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, mean_len if lengths else 600.0, args.playouts)
        print(json.dumps(out))
    eng.close()
    if is_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
