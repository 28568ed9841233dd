#!/usr/bin/env python3
"""bench.py -- self-play games/sec on N MI355X GPUs (BASELINE.json's metric).

    python bench.py --gpus N --steps K --warmup W
    N > 1: launched by torch.distributed.run (one rank per GPU, RCCL), or started plainly -- then this process only spawns
    the N ranks (before anything here touches a GPU), relays rank 0's JSON line and exits with the ranks' status.

Workload (config.workload): BASELINE.json configs[3] per GPU -- n_playout=400, 9x9, reference defaults (10 walls, c_puct=5,
temp=1, Dirichlet 0.3/0.25), random-init policy_value_net in fp32 with the reference's per-leaf BatchNorm statistics; weak
scaling, finished tuples all-gathered every step.  --boards concurrent boards per GPU (default 13,312: the chip holds 7,168
wavefronts of k_advance, seven per SIMD; a board leaves its launch when it meets a leaf for the network -- a seventh of them, those
whose mover still has walls, after one playout -- and its slot goes to the next board: with ONE deadline per launch (--select-opts 8)
the boards beyond the 7,168th take the slots in turn; configs[3] names 4,096 as the per-GPU minimum of concurrent boards and
`--boards 4096` runs exactly that; `--boards 10240 --budget-us 2400 --select-opts 0` is round 4's shape).

DEFAULT ROUTE (--mode async): the asynchronous self-play loop (qz_selfplay_*, include/qz_abi.h): every board runs its 400
playouts per move on its own clock; a leaf whose evaluation is in the leaf-evaluation memo is expanded from the memo, every
other leaf is evaluated by the network -- bit for bit the lock-step engine's search (tests/test_gpu_async*.py).  A STEP is
--rounds-per-step rounds; a round = k_advance (playouts until every board needs the network or its budget is used), the
network on the leaves the memo does not know with actions() of those leaves and k_moves beside it, the memo insert.  Then
finished games are harvested (+ all-gathered).  --mode lockstep is round 2's route (every leaf through the network), for A/B.

value            games_per_s_steady_state.value: boards / E[wall time of a game]; plies per phase of a game from the committed
                 length sample (benchmarks/game_length.py), cost per ply of each phase measured in the timed region.
                 A reference-faithful 400-playout game lasts tens of thousands of plies (the reference backs a won position up
                 with the wrong sign, mcts.py:119-125) and the length estimate has NOT converged with the observation window:
                 value_low / value_high bracket it, plies_per_s and playouts_per_s are the tracked scalars.
second_line_fix_terminal_sign   a short second run with the terminal sign fixed (NOT the reference's mcts.py:125; games of
                 ~300 plies): REAL finished games / wall time.  Labelled, never the headline.
roofline         the dominant kernel: k_advance (async) or the rules op (lockstep); HIP events on the launch stream.
roofline_rules / roofline_nn / roofline_c3   the rules op on the miss list, the network launches, the rules op on 32,768
                 mid-game boards (BASELINE configs[2]).
cpu_baseline     the CPU oracle (scalar C port of the reference's algorithm + batch-1 torch-CPU network) on this host.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONV_FLOPS = 2.0 * 81 * 9 * (10 * 64 * 64 + 64 * 6 + 26 * 64) + 2.0 * (324 * 128 + 128 + 162 * 140)  # fp32-equivalent FLOPs per leaf

sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
from bench_support import BYTES_PER_BOARD, HBM_PEAK_GBS, ClockSampler, _latest_profile, _load_json, c3_microbench, cpu_baseline, file_sha256, tree_sha, usable_cores  # noqa: E402,F401


def launch_ranks(n_gpus, argv):
    """`python bench.py --gpus N` without a launcher: become the launcher.  The ranks are CHILD processes (this process never
    initialises HIP and never re-execs); stdout of the job is the single JSON line rank 0 prints, everything else goes to stderr."""
    import socket

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


# ------------------------------------------------------------------------------ games/s from plies/s: the length sample
def _length_fields(d, length_file, rate):
    """What every estimate says about the length sample it rests on, and the BRACKET of the headline: the Kaplan-Meier
    restricted mean is a model-free lower bound of E[length] (-> value_high), the tail model with half the measured hazard an
    upper one of the same family (-> value_low); window_doubling = how much the estimate still moved when the window doubled."""
    rm, ST, lam = d.get("restricted_mean"), d.get("survival_at_T"), d.get("tail_hazard_per_ply")
    wd = d.get("window_doubling") or {}
    out = {"n_games_in_length_sample": d.get("games_finished"), "n_games_censored": d.get("games_censored"),
           "n_games_dropped_as_censored": d.get("games_dropped_as_censored"), "length_estimator": d.get("estimator"),
           "length_source": os.path.relpath(length_file, ROOT), "length_source_sha256": file_sha256(length_file), "restricted_mean_plies": rm, "observation_window_plies": d.get("T"),
           "survival_at_window": ST, "upper_bound": rate(float(rm)) if rm else None,
           "value_high": rate(float(rm)) if rm else None,
           "value_low": rate(float(rm) + 2.0 * float(ST) / float(lam)) if rm and ST is not None and lam else None,
           "length_window_doubling_delta": wd.get("delta"),
           "length_estimate_converged": (abs(wd["delta"]) < 0.10) if wd.get("delta") is not None else None}
    return out


def steady_state(plies_per_s, length_file):
    """plies/s / E[length of a game of this playout count], with a 95 % interval from the length sample."""
    d = _load_json(length_file or "")
    if not d or not d.get("mean_plies_per_game"):
        return None
    L = float(d["mean_plies_per_game"])
    lo, hi = (d.get("mean_ci95") or [None, None])[:2]
    out = {"value": plies_per_s / L, "unit": "games/s", "plies_per_s": plies_per_s, "mean_plies_per_game": L,
           "ci95": [plies_per_s / hi, plies_per_s / lo] if lo and hi else None}
    out.update(_length_fields(d, length_file, lambda length: plies_per_s / length))
    return out


def steady_state_two_phase(boards_all, open_plies, end_plies, open_board_s, end_board_s, length_file):
    """Games/s of a stationary population: boards / E[wall time of one game], E = plies of a game in each of its two phases
    (committed length sample) x board-seconds per ply of that phase (measured in the timed region).  The phases: the root's
    mover still has walls (almost every leaf is new: one network round trip per playout) and afterwards (a few thousand
    positions revisited: the memo answers).  Dropped games (depth limit, root without a move: the reference cannot finish them
    either) cost board time and yield no game: with the sample's finished fraction f and mean ply of a drop, the rate is
    f x boards / (f x E[time of a finished game] + (1 - f) x E[time of a dropped one])."""
    d = _load_json(length_file or "")
    if not d or not d.get("mean_plies_per_game") or not d.get("mean_open_plies_per_game") or open_plies <= 0 or end_plies <= 0:
        return None
    L, Lo = float(d["mean_plies_per_game"]), float(d["mean_open_plies_per_game"])
    c_open, c_end = open_board_s / open_plies, end_board_s / end_plies
    f = float(d.get("finished_fraction_of_started") or 1.0)
    drop_s = (Lo * c_open + max(float(d.get("dropped_mean_ply") or 0.0) - Lo, 0.0) * c_end) if f < 1.0 else 0.0
    lo, hi = (d.get("mean_ci95") or [None, None])[:2]

    def dur(length):
        return Lo * c_open + (length - Lo) * c_end

    def rate(length):
        return f * boards_all / (f * dur(length) + (1.0 - f) * drop_s)

    out = {"value": rate(L), "unit": "games/s",
           "estimator": "boards / (open plies x board-seconds per open ply + late plies x board-seconds per late ply), x the fraction of started games that finish",
           "mean_plies_per_game": L, "mean_open_plies_per_game": Lo, "board_seconds_per_open_ply": c_open, "board_seconds_per_late_ply": c_end,
           "seconds_per_game_per_board": dur(L), "open_phase_share_of_a_game": Lo * c_open / dur(L), "finished_fraction_of_started": f,
           "ci95": [rate(hi), rate(lo)] if lo and hi else None}
    out.update(_length_fields(d, length_file, rate))
    return out


# ------------------------------------------------------------------------------ pieces shared by the two routes
class Run:
    """The engine, the harvest (+ the path's only exchange: finished tuples -> every rank's buffer) and the phases' game lengths."""

    def __init__(self, args, eng, qdist, rank, local, world, dev):
        self.args, self.eng, self.qdist, self.rank, self.local, self.world, self.dev = args, eng, qdist, rank, local, world, dev
        self.lengths = {"desync": [], "warmup": [], "timed": []}
        self.phase = "desync"
        self.exch = None

    def barrier(self):
        if self.world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(self.dev)

    def harvest(self):
        tbs = self.eng.harvest()
        n_games = 0
        for tb in tbs:
            n_games += tb.n_games
            gid = tb.game.cpu().numpy()
            assert gid.size and 0 <= int(gid.min()) and int(gid.max()) < tb.n_games, "corrupt game ids in a harvest"
            self.lengths[self.phase].extend(np.bincount(gid, minlength=tb.n_games).tolist())
        if self.world > 1:
            self.eng.synchronize()
            bufs = [self.qdist.pack_tuples(tb.boards.hbits, tb.boards.vbits, tb.boards.meta, tb.pi, tb.z) for tb in tbs]
            bufs.append(torch.zeros((0, self.qdist.TUPLE_BYTES), dtype=torch.uint8, device=self.dev))
            self.qdist.allgather_tuples(torch.cat(bufs))
        return n_games

    def exchange_summary(self):
        """`allgather_ms` of an N > 1 line: this run's exchanges (dist.exchange_log), the slowest rank's mean and maximum."""
        if self.world == 1:
            return None
        d = self.qdist.exchange_log.summary()
        assert d["calls"] >= self.args.steps, "every step of an N > 1 run ends with one exchange"
        t = torch.tensor([d["mean_ms"] or 0.0, d["max_ms"] or 0.0, d["mean_bytes_received"] or 0.0], dtype=torch.float64, device=self.dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        self.exch = {"calls": d["calls"], "mean": float(t[0]), "max": float(t[1]), "unit": "ms", "bytes_received_per_call_mean": float(t[2]),
                "what": "dist.allgather_tuples per harvest (count all-gather + one padded payload all-gather of 588-byte tuples, then a device sync); "
                        "max over ranks of each rank's mean / maximum; inside the timed region and inside ms_per_step"}
        return self.exch

    def per_rank(self, elapsed, plies, playouts):
        """What EACH rank did, gathered on every rank: an N > 1 line shows N distinct devices each doing its share.
        On RCCL the PCI bus ids must differ (one process per GPU); under gloo two ranks may share a device (the CPU-box test shape)."""
        try:
            bus = torch.cuda.get_device_properties(self.dev)
            bus_id = "%04x:%02x:%02x" % (int(getattr(bus, "pci_domain_id", 0)), int(getattr(bus, "pci_bus_id", -1)), int(getattr(bus, "pci_device_id", 0)))
        except Exception:  # noqa: BLE001
            bus_id = "unknown"
        mine = {"rank": self.rank, "local_device": int(self.dev.index or 0), "pci_bus_id": bus_id, "plies": float(plies), "playouts": float(playouts),
                "ms_per_step": elapsed / self.args.steps * 1e3}
        if self.world == 1:
            return [mine]
        allr = [None] * self.world
        torch.distributed.all_gather_object(allr, mine)
        if torch.distributed.get_backend() == "nccl":
            ids = [r["pci_bus_id"] for r in allr]
            assert len(set(ids)) == self.world, "two ranks of an RCCL job on one device: %s" % ids
        return allr

    def reduce(self, elapsed, totals):
        el = torch.tensor([elapsed], dtype=torch.float64, device=self.dev)
        tot = torch.tensor(totals, dtype=torch.float64, device=self.dev)
        if self.world > 1:
            torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
            torch.distributed.all_reduce(tot, op=torch.distributed.ReduceOp.SUM)
        return float(el.item()), [float(x) for x in tot.tolist()]

    def base_line(self, elapsed, value, value_is, ss, games_all, plies_all, playouts_all, term_all, step_ms, desync_s, mode_cfg, workload_tail):
        a = self.args
        B = a.boards

        def len_stats(v):
            return {"n": len(v), "mean": float(np.mean(v)) if v else None, "median": float(np.median(v)) if v else None}

        name = "BASELINE configs[2] as an engine run" if B == 32768 else ("BASELINE configs[1]" if a.playouts == 100 else (
            "BASELINE configs[3] per GPU" if B == 4096 else "BASELINE configs[3]'s per-GPU workload with %d instead of 4,096 boards per GPU (the engine keeps seven "
            "k_advance wavefronts per SIMD busy, and the slots of boards that leave a launch early go to the boards beyond the 7,168th%s; same_loop_at_4096_boards is the literal board count)"
            % (B, ", one deadline per launch" if (a.select_opts & 8) else "")))
        label = ("NON-PARITY THROUGHPUT MODE (network products on fp16 operands, p / v ~1e-3 from the reference) -- " if a.nn_dtype == "fp16" else "") + \
                ("TERMINAL SIGN FIXED (not the reference's mcts.py:125) -- " if a.fix_terminal_sign else "")
        cfg = {"workload": label + "%s: %d concurrent boards/GPU, n_playout=%d, 9x9, 10 walls/player, c_puct=5, temp=1.0, %s" % (name, B, a.playouts, workload_tail),
               "boards_per_gpu": B, "board_groups": a.groups, "fix_terminal_sign": bool(a.fix_terminal_sign), "n_playout": a.playouts, "bn_mode": a.bn,
               "nn_dtype": a.nn_dtype, "max_depth": a.max_depth,
               "desync": "%d untimed plies at %d playouts/move, then %d untimed rounds at %d playouts/move (%.0fs, %d games finished)"
                         % (a.desync_plies, a.desync_playouts, a.settle_rounds if a.mode == "async" else 0, a.playouts, desync_s, len(self.lengths["desync"]))}
        cfg.update(mode_cfg)
        out = {"metric": "self-play games/sec (9x9, n_playout=%d)" % a.playouts, "value": value, "unit": "games/s", "n_gpus": self.world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "u64 bitboards + f64 PUCT (rules/tree kernels); %s policy-value net" % a.nn_dtype,
               "data": "synthetic (random-init policy_value_net, seed %d; self-generated games)" % a.seed, "config": cfg, "value_is": value_is,
               "tracked_scalar": "plies_per_s",
               "games_in_timed_region": games_all, "games_in_timed_region_per_s": games_all / elapsed, "games_per_s_steady_state": ss,
               "plies_per_s": plies_all / elapsed, "playouts_per_s": playouts_all / elapsed, "terminal_leaf_frac": term_all / max(playouts_all, 1.0),
               "game_lengths_seen": {"timed_region_%d_playouts" % a.playouts: len_stats(self.lengths["timed"] + self.lengths["warmup"]),
                                     "desync_phase_%d_playouts" % a.desync_playouts: len_stats(self.lengths["desync"]),
                                     "note": "games finishing in the timed region started in the desync phase; neither is the length of a %d-playout game" % a.playouts},
               "ms_per_step_series": [round(x, 1) for x in step_ms]}
        if ss:  # the bracket of the headline, on the top level where a reader of `value` finds it
            for k in ("value_low", "value_high", "length_window_doubling_delta", "length_estimate_converged"):
                out[k] = ss.get(k)
            if not ss.get("length_estimate_converged"):
                out["value_note"] = ("games/s in the reference-faithful mode is a BRACKET [value_low, value_high]: E[game length] kept growing with the observation "
                                     "window of the length sample; plies_per_s and playouts_per_s are the tracked scalars")
        return out

    def finish(self, out, st1, sampler, step_ms, ss, ss_simple=None):
        a = self.args
        if st1["games_aborted"]:
            sys.stderr.write("bench.py: NOTE %d games were dropped (depth > %d: %d, no legal move: %d, max_plies %d, pool %d)\n"
                             % (st1["games_aborted"], a.max_depth, st1["aborted_depth"], st1["aborted_no_move"], st1["aborted_max_plies"], st1["aborted_pool"]))
        assert st1["node_overflow"] == 0 and st1["runaway_descents"] == 0 and st1.get("miss_overflow", 0) == 0, "tree storage overflowed / corrupted during the run"
        out["clocks"] = sampler.summary() if sampler else None
        if self.world > 1:
            out["allgather_ms"] = self.exch
        out["provenance"] = tree_sha()
        if a.clock_log and sampler:
            with open(a.clock_log, "w") as f:
                json.dump({"step_ms": step_ms, "samples": sampler.samples}, f)
        if self.world == 1 and not a.no_c3:
            out["roofline_c3"] = c3_microbench(self.dev)
        if self.world == 1 and a.mode == "async" and not a.fix_terminal_sign and a.second_line_seconds > 0:
            self.eng.close()
            torch.cuda.empty_cache()
            if a.boards != 4096 and a.playouts == 400:
                out["same_loop_at_4096_boards"] = line_at_4096_boards(a, self.dev, self.qdist)
            out["second_line_fix_terminal_sign"] = second_line(a, self.dev, self.qdist)
            if not a.no_non_parity_line:  # SURVEY 7 hard part 1 "report both modes": the throughput-precision network, labelled, never the headline
                torch.cuda.empty_cache()
                out["second_line_NON_PARITY_fp16"] = second_line(a, self.dev, self.qdist, nn_precision="fp16")
        if not a.no_cpu_baseline and self.world == 1:
            self.eng.close()
            torch.cuda.empty_cache()
            src = ss or ss_simple
            if src:
                L_cpu, L_src = src["mean_plies_per_game"], src["length_source"]
            else:
                seen = self.lengths["timed"] + self.lengths["warmup"] + self.lengths["desync"]
                L_cpu = float(np.mean(seen)) if seen else None
                L_src = "NO length sample for n_playout=%d: mean length of the %d games finished in this run (mostly desync games)" % (a.playouts, len(seen))
            out["cpu_baseline"] = cpu_baseline(a.cpu_seconds, a.playouts, L_cpu, L_src, open_share=(out.get("open_phase") or {}).get("share_of_board_time"))
        out["summary"] = line_summary(out)
        print(json.dumps(out))


def round_schedule(rounds_per_step, event_every, graph_rounds=0):
    """The rounds of ONE step of the asynchronous loop as a list of ("timed", 1) / ("plain", n) items.  A "timed" round is
    issued as its four pieces with HIP events around them (what roofline / roofline_rules / roofline_nn are computed from);
    "plain" rounds go through run_rounds (graph replays when --graph-rounds).  Whatever the arguments, every step times at
    least ONE round: a bench line without `roofline` is unmeasured (round 4's GPU tier went red on exactly that:
    --event-every 64 with --rounds-per-step 48 timed nothing).  With a graph the first round of the step is the timed one
    (eager, piece by piece; run_rounds picks the graph of the miss-counter parity it finds)."""
    NR = int(rounds_per_step)
    if NR < 1:
        raise ValueError("--rounds-per-step must be >= 1")
    every = NR if graph_rounds else max(1, min(int(event_every), NR))
    out, done = [], 0
    while done < NR:
        out.append(("timed", 1))
        done += 1
        n = min(every - 1, NR - done)
        if n > 0:
            out.append(("plain", n))
            done += n
    return out


def hbm_line(kernel, us, nbytes, note, launches, traffic=None, src=None, src_file=None):
    if us is None or not launches:
        raise SystemExit("bench.py: no timed launch of '%s' -- a line without its roofline is unmeasured (round_schedule must time >= 1 round per step)" % kernel[:40])
    gbs = nbytes / (us * 1e-6) / 1e9
    return {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": src, "traffic_source_sha256": file_sha256(src_file) if src_file else None,
            "traffic_measured_on_these_kernels": ((_load_json(src_file) or {}).get("kernel_sources_sha256") == tree_sha()["kernel_sources_sha256"]) if src_file else None,
            "avg_launch_us": us, "launches_timed": launches, "algorithmic_bytes_per_launch": nbytes, "note": note}


def make_engine(args, net, dev, seed, fix_sign, boards=None, nn_precision=None, select_opts=None):
    from alphazero_quoridor_amd.engine import BoardGroups

    prec = nn_precision or ("fp16" if args.nn_dtype == "fp16" else "fp32")
    dt = torch.bfloat16 if args.nn_dtype == "bf16" else torch.float32
    if args.library_trunk:
        from alphazero_quoridor_amd.policy_value_net import LeafEvaluator
        lib_ev = LeafEvaluator(net.policy_value_net, args.bn, dt, bool(args.channels_last), mfma_trunk=False)
        make_ev = lambda: lib_ev  # noqa: E731
    else:
        make_ev = lambda: net.evaluator(args.bn, dt, bool(args.channels_last), nn_precision=prec)  # noqa: E731
    so = args.select_opts if select_opts is None else select_opts
    eng = BoardGroups(boards or args.boards, args.groups, make_ev, seed=seed, device=dev, n_playout=args.playouts, c_puct=5, temp=1.0, is_selfplay=1,
                      fix_terminal_sign=fix_sign, select_opts=so, memo=not args.no_memo, max_depth=args.max_depth)
    return eng


def line_summary(out):
    """The line's scalars once more, flat, as its LAST key: the line is ~18 KB and a record that keeps its head and its tail drops the
    middle -- where plies_per_s and playouts_per_s stand."""
    cb = out.get("cpu_baseline") or {}
    rf, c3, nn = out.get("roofline") or {}, out.get("roofline_c3") or {}, out.get("roofline_nn") or {}
    pps = out.get("playouts_per_s")
    g = lambda d, *ks: next((d.get(k) for k in ks if d.get(k) is not None), None)
    return {
        "plies_per_s": out.get("plies_per_s"), "playouts_per_s": pps, "ms_per_round": out.get("ms_per_round"), "ms_per_step": out.get("ms_per_step"),
        "games_per_s_value": out.get("value"), "games_per_s_low": out.get("value_low"), "games_per_s_high": out.get("value_high"),
        "roofline_frac": rf.get("frac"), "roofline_avg_launch_us": rf.get("avg_launch_us"), "roofline_traffic_over_algorithmic":
            (rf["traffic"] / rf["algorithmic_bytes_per_launch"]) if rf.get("traffic") and rf.get("algorithmic_bytes_per_launch") else None,
        "roofline_traffic_measured_on_these_kernels": rf.get("traffic_measured_on_these_kernels"),
        "roofline_c3_frac": c3.get("frac"), "roofline_c3_avg_launch_us": c3.get("avg_launch_us"), "roofline_nn_avg_launch_us": nn.get("avg_launch_us"),
        "leaves_per_round": nn.get("leaves_per_launch"), "memo_hit_rate": out.get("memo_hit_rate"),
        "open_phase_share_of_board_time": (out.get("open_phase") or {}).get("share_of_board_time"),
        "same_loop_at_4096_boards_playouts_per_s": (out.get("same_loop_at_4096_boards") or {}).get("playouts_per_s"),
        "second_line_fix_terminal_sign_games_per_s": (out.get("second_line_fix_terminal_sign") or {}).get("value"),
        "second_line_NON_PARITY_fp16_games_per_s": (out.get("second_line_NON_PARITY_fp16") or {}).get("value"),
        "cpu_port_playouts_per_s_allcores": cb.get("playouts_per_s_allcores"), "cpu_cores": cb.get("cores"),
        "cpu_reference_estimate_playouts_per_s_allcores": g(cb, "reference_playouts_per_s_allcores"),
        "gpu_over_cpu_port_playouts": (pps / cb["playouts_per_s_allcores"]) if pps and cb.get("playouts_per_s_allcores") else None,
        "kernel_sources_sha256": (out.get("provenance") or {}).get("kernel_sources_sha256"),
    }


def line_at_4096_boards(args, dev, qdist):
    """BASELINE configs[3] names 4,096 boards per GPU; the headline runs more (seven wavefronts of k_advance per SIMD + the slots of the boards that leave early).  The same
    loop at exactly 4,096 boards (k_advance<4>), same phases, a short timed region: so that both are in the driver's line."""
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

    torch.manual_seed(args.seed)
    net = PolicyValueNet(use_gpu=True, device=dev)
    eng = make_engine(args, net, dev, qdist.shard_seed(args.seed, 0), False, boards=4096, select_opts=0)  # (per-board budgets: every board has its slot)
    kw = dict(max_playouts=args.max_playouts, budget_us=1000)  # (4,096 boards, no board beyond the slots: 1,000 / 1,800 / 2,400 us = 214 / 205 / 181 M playouts/s)
    eng.set_playouts(args.desync_playouts)
    for _ in range(0, args.desync_plies * (args.desync_playouts + 1), 64):
        eng.run_rounds(64, **kw)
        eng.harvest()
    eng.set_playouts(args.playouts)
    for _ in range(0, args.settle_rounds + 1280, 64):
        eng.run_rounds(64, **kw)
        eng.harvest()
    st0 = eng.stats()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(0, 2560, 64):
        eng.run_rounds(64, **kw)
        eng.harvest()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    st1 = eng.stats()
    d = {k: st1[k] - st0[k] for k in st1}
    eng.close()
    return {"boards": 4096, "kernel": "k_advance<4> (97 registers, four wavefronts per SIMD)", "rounds": 2560, "seconds": dt, "ms_per_round": dt / 2560 * 1e3,
            "plies_per_s": d["plies_played"] / dt, "playouts_per_s": d["playouts"] / dt, "nn_evaluations_per_s": d["nn_evals"] / dt,
            "memo_hit_rate": d["memo_hits"] / max(d["playouts"], 1),
            "budget_us": 1000,
            "note": "same desync / settle phases as the headline + 1,280 warm-up rounds, then 2,560 rounds timed by this process; its own budget of 1,000 us"}


def second_line(args, dev, qdist, nn_precision="fp32"):
    """The target-reaching mode, timed by the same process: a winning move backed up as +1 (NOT the reference's mcts.py:125, which
    backs it up as -1 and makes searches avoid winning).  Games then last ~300 plies, so REAL finished games / wall time is a
    stationary quantity: boards desynchronised with short searches, warmed up for a few game lengths at the full playout count,
    then counted for >= --second-line-seconds.  Parity-precision network; everything else as in the headline."""
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

    torch.manual_seed(args.seed)
    net = PolicyValueNet(use_gpu=True, device=dev)
    # (round 5 sweep, stationary estimate on one box, boards / budget: 4,096 / 500 us 100 games/s, 8,192 / 500 116, 16,384 / 500 114,
    # 16,384 / 1,000 153, 32,768 / 1,000 135 -- profiles/round5/second_line_board_sweep.txt)
    B2 = args.second_line_boards
    eng = make_engine(args, net, dev, qdist.shard_seed(args.seed + 1, 0), True, boards=B2, nn_precision=nn_precision, select_opts=0)  # (the shape the sweep was run in)
    kw = dict(max_playouts=args.max_playouts, budget_us=args.second_line_budget_us)
    lens = []

    def harvest():
        n = 0
        for tb in eng.harvest():
            n += tb.n_games
            lens.extend(torch.bincount(tb.game.long(), minlength=tb.n_games).tolist())
        return n

    eng.set_playouts(args.desync_playouts)
    for _ in range(0, args.desync_plies * (args.desync_playouts + 1), 64):
        eng.run_rounds(64, **kw)
        harvest()
    eng.set_playouts(args.playouts)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.second_line_warm_seconds:  # a few game lengths at the full playout count
        eng.run_rounds(64, **kw)
        harvest()
    del lens[:]
    st0 = eng.stats()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    games = 0
    while time.perf_counter() - t0 < args.second_line_seconds:
        eng.run_rounds(64, **kw)
        games += harvest()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    st1 = eng.stats()
    d = {k: st1[k] - st0[k] for k in st1}
    eng.close()
    # The raw count is a TRANSIENT (the games that finish inside a short window are the ones the desync phase left close to their
    # end: followed for seven minutes the same engine sinks from 300 to 146 games/s, profiles/round4/second_line_time_course_4096boards.json).
    # `value` is therefore the stationary estimate, built like the headline's: boards / E[wall time of a game], plies per phase from the
    # committed SIGN-FIXED length sample (benchmarks/r5_length_fix_job.sh), board-seconds per ply of each phase measured here.
    rounds = max(d["rounds"], 1)
    open_board_s = d["open_rounds"] * dt / rounds
    end_board_s = max(B2 * rounds - d["open_rounds"], 0) * dt / rounds
    length_file = _latest_profile("game_length_%dplayouts_sign_fixed.json" % args.playouts)
    ss = steady_state_two_phase(B2, d["open_plies"], d["plies_played"] - d["open_plies"], open_board_s, end_board_s, length_file)
    raw = games / dt
    parity = nn_precision == "fp32"
    return {"label": ("NOT the headline and NOT the reference's arithmetic: terminal sign fixed (a won position backed up as +1; mcts.py:125 backs it up as -1)"
                      + ("" if parity else "; NON_PARITY network precision: ONE fp16 MFMA per product (p / v ~1e-3 from the reference instead of 1e-5)")),
            "value": ss["value"] if ss else raw, "unit": "games/s",
            "value_is": ("stationary estimate: boards / E[wall time of a game], plies per phase from %s, cost per ply of each phase measured in this run" % os.path.relpath(length_file, ROOT))
                        if ss else "NO committed sign-fixed length sample: the raw (transient) count",
            "value_transient": raw, "value_low": (ss or {}).get("value_low"), "value_high": (ss or {}).get("value_high"),
            "length_estimate_converged": (ss or {}).get("length_estimate_converged"), "games_per_s_steady_state": ss,
            "games_finished": games, "seconds": dt, "boards": B2, "n_playout": args.playouts,
            "mean_plies_per_game_of_the_finished": float(np.mean(lens)) if lens else None, "plies_per_s": d["plies_played"] / dt, "playouts_per_s": d["playouts"] / dt,
            "nn_evaluations_per_s": d["nn_evals"] / dt, "memo_hit_rate": d["memo_hits"] / max(d["playouts"], 1),
            "board_seconds_per_open_ply": open_board_s / max(d["open_plies"], 1),
            "nn_precision": "fp32 (parity: three fp16 MFMAs per product)" if parity else "fp16 operands, ONE MFMA per product (NON_PARITY)",
            "budget_us": args.second_line_budget_us, "nn_evaluations_per_game_transient": d["nn_evals"] / max(games, 1),
            "note": "network-bound: with the sign fixed a game spends most of its board time in the open phase (the mover has walls: nearly every leaf is new) at 400 "
                    "evaluations per ply; evaluations per game x games/s = the network's throughput",
            "measured": "after %d desync plies at %d playouts and %.0f s of warm-up at %d playouts; %.0f s counted"
                        % (args.desync_plies, args.desync_playouts, args.second_line_warm_seconds, args.playouts, dt)}


# ------------------------------------------------------------------------------ the asynchronous loop (default)
def run_async(R):
    import ctypes as C

    from alphazero_quoridor_amd import _cabi

    args, eng, dev, world, rank = R.args, R.eng, R.dev, R.world, R.rank
    B, G, NR = args.boards, args.groups, args.rounds_per_step
    kw = dict(max_playouts=args.max_playouts, budget_us=args.budget_us)
    t0 = time.time()
    if args.desync_plies > 0:  # desynchronise the games (untimed): short searches spread the boards over all game phases
        eng.set_playouts(args.desync_playouts)
        for _ in range(0, args.desync_plies * (args.desync_playouts + 1), 64):
            eng.run_rounds(64, **kw)
            R.harvest()
        eng.set_playouts(args.playouts)
        # ... then settle at the full playout count (untimed): the boards the short searches left in mid-game grow their trees and
        # the memo learns their neighbourhoods, as in any run longer than a few seconds (a 300-s soak: profiles/round3)
        for _ in range(0, args.settle_rounds, 64):
            eng.run_rounds(64, **kw)
            R.harvest()
    desync_s = time.time() - t0
    if args.graph_rounds:
        eng.capture_rounds(rounds=args.graph_rounds, **kw)
    R.phase = "warmup"
    for _ in range(args.warmup):
        eng.run_rounds(NR, **kw)
        R.harvest()
    # timed region.  Every `event_every`-th round of group 0 is issued as its four pieces with HIP events around them
    R.phase = "timed"
    eng0, ev0, st0_stream = eng.engines[0], eng.evaluators[0], eng.streams[0]
    L = _cabi.load()
    schedule = round_schedule(NR, args.event_every, args.graph_rounds)
    ev_every = NR if args.graph_rounds else max(1, min(args.event_every, NR))
    evs = []

    def timed_round():
        with torch.cuda.stream(st0_stream):
            eng0._memo_guard(ev0)
            s = eng0._s()
            pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
            w = ev0.nn_weights()
            for (a, b), call in zip(pairs, (lambda: L.qz_selfplay_advance(eng0.h, args.max_playouts, args.budget_us, 1, s), lambda: L.qz_selfplay_leaf_rules(eng0.h, s),
                                            lambda: L.qz_selfplay_evaluate(eng0.h, C.byref(w), s), lambda: L.qz_selfplay_round_tail(eng0.h, s))):
                a.record()
                _cabi.check(call())
                b.record()
            evs.append(pairs)

    sampler = ClockSampler(R.local) if rank == 0 else None
    st0 = eng.stats()
    R.qdist.exchange_log.clear()
    R.barrier()
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    games, step_ms = 0, []
    for _ in range(args.steps):
        ts = time.perf_counter()
        for kind, n in schedule:
            if kind == "timed":
                timed_round()
                for g in range(1, G):
                    with torch.cuda.stream(eng.streams[g]):
                        eng.engines[g].run_rounds(eng.evaluators[g], 1, **kw)
            else:
                eng.run_rounds(n, **kw)
        games += R.harvest()  # (synchronises with the device: the step's wall time)
        step_ms.append((time.perf_counter() - ts) * 1e3)
    R.barrier()
    elapsed = time.perf_counter() - t0
    if sampler:
        sampler.stop()
    st1 = eng.stats()
    d = {k: st1[k] - st0[k] for k in st1}
    exch = R.exchange_summary()
    ranks = R.per_rank(elapsed, d["plies_played"], d["playouts"])
    if args.dump_roots and rank == 0:  # the steady-state population's root positions (cpu_baseline's workload: tests/golden/steady_state_roots.npz)
        roots = np.concatenate([e.get_boards().to_packed() for e in eng.engines])
        np.savez_compressed(args.dump_roots, board=roots, boards_per_gpu=B, n_playout=args.playouts, rounds_played=int(st1["rounds"]), seed=args.seed)
    elapsed, (games_all, plies_all, playouts_all, term_all, evals_all, hits_all, open_plies_all, open_rounds_all) = R.reduce(
        elapsed, [games, d["plies_played"], d["playouts"], d["leaf_terminal"], d["nn_evals"], d["memo_hits"], d["open_plies"], d["open_rounds"]])
    if rank != 0:
        return
    rounds = args.steps * NR
    round_s = elapsed / rounds
    open_board_s = open_rounds_all * round_s  # a board is in the open phase in the launches k_advance counted (root's mover has walls)
    end_board_s = max(B * world * rounds - open_rounds_all, 0.0) * round_s
    length_file = args.length_file or (None if args.fix_terminal_sign else _latest_profile("game_length_%dplayouts.json" % args.playouts))
    ss = steady_state_two_phase(B * world, open_plies_all, plies_all - open_plies_all, open_board_s, end_board_s, length_file)
    ss_simple = steady_state(plies_all / elapsed, length_file)
    if not evs:
        raise SystemExit("bench.py: the timed region issued no timed round (rounds_per_step=%d, event_every=%d, graph_rounds=%d): no roofline, no line" % (NR, args.event_every, args.graph_rounds))
    adv_us, rules_us, nn_us, tail_us = (sum(p[i][0].elapsed_time(p[i][1]) for p in evs) / len(evs) * 1e3 for i in range(4))
    per_launch = {k: d[k] / (rounds * G) for k in ("edges_scanned", "edges_expanded", "descent_levels", "playouts", "memo_hits", "nn_evals")}
    # k_advance's algorithmic bytes: every level of a descent reads the node's edge records (32 B each) and its 12-byte record entry,
    # the backup rewrites 24 B per level, a new leaf probes one 512-byte memo bucket and its expansion writes 32 B per legal move
    adv_bytes = (per_launch["edges_scanned"] * 32 + per_launch["descent_levels"] * (12 + 24) + (per_launch["memo_hits"] + per_launch["nn_evals"]) * 512
                 + per_launch["edges_expanded"] * 32)
    t_adv_f = _latest_profile("pmc_traffic_advance.json")
    t_adv = _load_json(t_adv_f or "")
    miss_per_round = evals_all / max(rounds * G * world, 1)
    waves = (B // G + 1023) // 1024
    value_is = ("games_per_s_steady_state.value: boards / E[wall time of a game] (plies per phase from %s, cost per ply of each phase measured in the timed region)"
                % os.path.relpath(length_file, ROOT)) if ss else "games finished inside the timed region / wall time (no committed length sample with a phase split)"
    out = R.base_line(elapsed, ss["value"] if ss else games_all / elapsed, value_is, ss, games_all, plies_all, playouts_all, term_all, step_ms, desync_s,
                      {"mode": "async", "rounds_per_step": NR, "budget_us": args.budget_us, "max_playouts_per_round": args.max_playouts, "graph_rounds": args.graph_rounds,
                       "step": "%d rounds; a round = k_advance (every board: playouts until it needs the network, %d us budget) + network on the leaves the memo does "
                               "not know (actions() of those leaves and k_moves on a second stream beside the trunk) + memo insert; every %d-th round is issued piece by "
                               "piece with HIP events around the pieces" % (NR, args.budget_us, ev_every)},
                      "asynchronous self-play loop with leaf-evaluation memo, finished tuples all-gathered every step")
    out.update({
        "per_rank": ranks, "games_per_s_plies_over_mean_length": ss_simple, "nn_evaluations_per_s": evals_all / elapsed, "leaf_evals_per_s": evals_all / elapsed,
        "memo_hit_rate": hits_all / max(playouts_all, 1.0),
        "open_phase": {"plies": open_plies_all, "share_of_board_time": open_rounds_all / max(B * world * rounds, 1), "note": "root's mover still has walls"},
        "mean_descent_depth": d["descent_levels"] / max(d["playouts"], 1), "rounds": rounds, "ms_per_round": round_s * 1e3,
        "roofline": hbm_line(
            "k_moves + k_advance (one wavefront per board: %d boards per SIMD for at most seven wavefronts resident per SIMD; moves of the boards that finished "
            "their playouts, then descents -- recorded descents replayed 64 levels per round trip, the first unrecorded level selected by the same round --, "
            "memo probes, expansions and backups until the board needs the network or the budget is used)" % waves, adv_us, adv_bytes,
            "dependent-load latency and instruction issue, not bandwidth: a playout is a chain of memory round trips (record -> edge blocks -> memo bucket -> "
            "backup) of a single wavefront; %.0f playouts per launch, mean depth %.1f.  The launch lasts its time budget + the last playouts; see DESIGN 3.0"
            % (per_launch["playouts"], d["descent_levels"] / max(d["playouts"], 1)), len(evs),
            traffic=(t_adv or {}).get("traffic_bytes_per_launch"), src=("profiles: " + os.path.relpath(t_adv_f, ROOT)) if t_adv else None, src_file=t_adv_f if t_adv else None),
        "roofline_rules": hbm_line("k_wave_rules on the miss list (Quoridor.actions() of the leaves the memo does not know; legal sets only)", rules_us, miss_per_round * 44,
                                   "%.0f leaves per launch: a launch lasts as long as one board's dependent chain" % miss_per_round, len(evs)),
        "roofline_nn": None if nn_us is None else {
            "kernel": "k_trunk<true> + k_head_fc on the miss list (first layer from the packed boards, ten conv3x3 64->64 layers as implicit GEMMs on "
                      "v_mfma_f32_32x32x16_f16 with split fp16 operands, per-leaf normalisation, heads)",
            "bound": "mfma", "achieved": CONV_FLOPS * miss_per_round / (nn_us * 1e-6) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
            "frac": CONV_FLOPS * miss_per_round / (nn_us * 1e-6) / 1e12 / 2500.0, "traffic": None, "avg_launch_us": nn_us, "launches_timed": len(evs),
            "leaves_per_launch": miss_per_round,
            "note": "one leaf per 2-wave workgroup, 1,024 workgroups resident; three fp16 MFMAs per fp32-accurate product: 3.56x the useful flops counted here"},
        "round_tail_us": tail_us,
        "games_dropped_in_timed_region": {"total": d["games_aborted"], "depth_over_%d_levels" % args.max_depth: d["aborted_depth"], "no_legal_move": d["aborted_no_move"],
                                          "note": "the reference cannot finish these games either: a path longer than 992 levels overflows its recursive backup "
                                                  "(RecursionError, mcts.py:55-62), a root without a legal move crashes start_self_play (mcts.py:195-196; "
                                                  "tests/golden/no_move_roots.npz)"},
        "engine_stats": {kk: st1[kk] for kk in ("node_overflow", "games_aborted", "aborted_no_move", "aborted_max_plies", "aborted_pool", "aborted_depth", "nonfinite_values",
                                                "runaway_descents", "miss_overflow", "compact_slices", "arena_bytes", "max_nodes", "max_edges", "max_depth", "tree_pages_total",
                                                "tree_pages_peak", "traj_pages_total", "traj_pages_peak", "memo_inserts", "memo_locked")},
    })
    R.finish(out, st1, sampler, step_ms, ss, ss_simple)


# ------------------------------------------------------------------------------ the lock-step route (round 2; A/B)
def run_lockstep(R):
    args, eng, dev, world, rank = R.args, R.eng, R.dev, R.world, R.rank
    if args.rules_variant:
        from alphazero_quoridor_amd import rules as qrules

        for e in eng.engines:
            e.set_rules_opts(qrules.rules_opts(args.rules_variant))
    gb, G = args.boards // args.groups, args.groups
    write_planes = not args.no_planes
    planes_consumed = not getattr(eng.evaluators[0], "accepts_leaf_boards", False)
    if write_planes:
        for e in eng.engines:
            e.always_write_planes = True

    def ply(n=None):
        eng.run_playouts(n)
        eng.finish_move()
        return R.harvest()

    t0 = time.time()
    for _ in range(args.desync_plies):
        ply(args.desync_playouts)
    desync_s = time.time() - t0
    R.phase = "warmup"
    for _ in range(args.warmup):
        ply()
    R.phase = "timed"
    n_launch = args.steps * args.playouts

    def pair():
        return (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    evs = [[pair() for _ in range(G)] for _ in range(n_launch)]
    tevs = [[(pair(), pair()) for _ in range(G)] for _ in range(n_launch)]
    nevs = [[pair() for _ in range(G)] for _ in range(n_launch)]
    sampler = ClockSampler(R.local) if rank == 0 else None
    st0 = eng.stats()
    R.qdist.exchange_log.clear()
    R.barrier()
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    games, k, step_ms = 0, 0, []
    for _ in range(args.steps):
        ts = time.perf_counter()
        for j in range(args.playouts):  # (the expand / backup launch of every playout but the ply's last also runs the next playout's descent)
            eng.playout_step(events=evs[k], write_planes=write_planes, tree_events=tevs[k], nn_events=nevs[k], more=(j + 1 < args.playouts) and not args.separate_descent)
            k += 1
        eng.finish_move()
        games += R.harvest()
        step_ms.append((time.perf_counter() - ts) * 1e3)
    R.barrier()
    elapsed = time.perf_counter() - t0
    if sampler:
        sampler.stop()
    st1 = eng.stats()
    d = {kk: st1[kk] - st0[kk] for kk in st1}
    exch = R.exchange_summary()
    ranks = R.per_rank(elapsed, d["plies_played"], d["playouts"])
    elapsed, (games_all, plies_all, playouts_all, term_all) = R.reduce(elapsed, [games, d["plies_played"], d["playouts"], d["leaf_terminal"]])
    if rank != 0:
        return
    n_evs = n_launch * G

    def avg_us(rows):
        return sum(a.elapsed_time(b) for row in rows for a, b in row) / n_evs * 1e3

    rules_us, nn_us = avg_us(evs), avg_us(nevs)
    sel_us = avg_us([[g[0] for g in row] for row in tevs])
    exp_us = avg_us([[g[1] for g in row] for row in tevs])
    planes_written = bool(write_planes or planes_consumed)
    bpb = BYTES_PER_BOARD if planes_written else 44
    t = _load_json(_latest_profile("pmc_traffic.json") or "")
    traffic = t["traffic_bytes_per_launch"] if t and int(t.get("boards", -1)) == gb and bool(t.get("planes", True)) == planes_written else None
    # k_select: every level reads the node's edge records (32 B each); per board 24 B root board in, 37 B leaf record out, 4 B per level of
    # path.  k_expand_backup: per expansion a 560-B prior row in and k 32-B records out; per level 12 B read + 12 B written
    tree_bytes = (d["edges_scanned"] * 32 + d["descent_levels"] * 28 + d["playouts"] * (24 + 37 + 33) + d["edges_expanded"] * 32
                  + (d["playouts"] - d["leaf_terminal"]) * 568) / n_evs
    length_file = args.length_file or (None if args.fix_terminal_sign else _latest_profile("game_length_%dplayouts.json" % args.playouts))
    ss = steady_state(plies_all / elapsed, length_file)
    out = R.base_line(elapsed, games_all / elapsed, "games finished inside the timed region / wall time (the population was desynchronised with %d-playout games and is not "
                      "stationary) -- see games_per_s_steady_state" % args.desync_playouts, ss, games_all, plies_all, playouts_all, term_all, step_ms, desync_s,
                      {"mode": "lockstep", "channels_last": bool(args.channels_last), "rules_variant": args.rules_variant,
                       "step": "one ply of every board (n_playout playout steps + finish_move + harvest)"},
                      "lock-step engine: leaf batch=%d, every leaf through the network, finished tuples all-gathered every ply" % args.boards)
    out.update({
        "per_rank": ranks, "leaf_evals_per_s": playouts_all / elapsed, "mean_descent_depth": d["descent_levels"] / max(d["playouts"], 1),
        "roofline": dict(hbm_line("the rules op on the leaf batch (Quoridor.actions() + state(): k_wave_rules below 8,192 boards, the pooled two-launch pipeline above; "
                                  "qz_rules_opts.variant %d)" % args.rules_variant, rules_us, gb * bpb, "planes written: %s" % planes_written, n_evs, traffic=traffic),
                         planes_written=planes_written, planes_consumed_by_evaluator=bool(planes_consumed)),
        "roofline_tree": [hbm_line("k_expand_backup_select" if not args.separate_descent else "k_select + k_expand_backup", sel_us + exp_us, tree_bytes,
                                   "dependent-load latency: duration = the slowest of %d boards (mean depth %.1f, deepest descent %d levels)"
                                   % (gb, d["descent_levels"] / max(d["playouts"], 1), st1["max_depth"]), n_evs)],
        "roofline_nn": {"kernel": "k_trunk<true> + k_head_fc (the whole leaf evaluation in two launches)" if getattr(eng.evaluators[0], "mfma_trunk", False) and not args.library_trunk
                        else "evaluator on MIOpen fp32 convolutions (--library-trunk)", "bound": "mfma", "achieved": CONV_FLOPS * gb / (nn_us * 1e-6) / 1e12, "peak": 2500.0,
                        "unit": "TFLOP/s", "frac": CONV_FLOPS * gb / (nn_us * 1e-6) / 1e12 / 2500.0, "traffic": None, "avg_launch_us": nn_us, "launches_timed": n_evs},
        "engine_stats": {kk: st1[kk] for kk in ("node_overflow", "games_aborted", "aborted_no_move", "aborted_max_plies", "aborted_pool", "nonfinite_values", "arena_bytes",
                                                "max_nodes", "max_edges", "max_depth", "deep_descents", "deep_descents_cold", "deep_levels", "deep_levels_replayed",
                                                "tree_pages_total", "tree_pages_peak", "traj_pages_total", "traj_pages_peak")},
    })
    R.finish(out, st1, sampler, step_ms, ss)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--boards", type=int, default=13312, help="concurrent boards per GPU (7,168 = seven k_advance wavefronts per SIMD; 4,096 = BASELINE configs[3]'s number; the sweep behind the default: profiles/round4/SUMMARY.md section 2)")
    ap.add_argument("--playouts", type=int, default=400)
    ap.add_argument("--groups", type=int, default=1, help="split the boards of a GPU into this many independent groups on their own HIP streams")
    ap.add_argument("--mode", default="async", choices=["async", "lockstep"])
    ap.add_argument("--rounds-per-step", type=int, default=256, help="async: rounds of the loop per step")
    ap.add_argument("--budget-us", type=int, default=3000, help="async: wall-clock budget of a k_advance launch in microseconds (the sweep behind the default: profiles/round4/SUMMARY.md section 2)")
    ap.add_argument("--max-playouts", type=int, default=4096, help="async: playouts a board may start per round")
    ap.add_argument("--graph-rounds", type=int, default=0, help="async: capture this many (even) rounds per HIP graph (0 = eager launches, per-kernel events)")
    ap.add_argument("--event-every", type=int, default=64, help="async: every n-th round of group 0 (at least one per step) is issued in pieces with HIP events around them; those rounds run their pieces one after the other (cost of the timing: profiles/round4/SUMMARY.md section 1)")
    ap.add_argument("--max-depth", type=int, default=992, help="drop a game whose playout descends more than this many levels (0 = never): where the reference's "
                                                              "recursive backup (mcts.py:55-62) overflows Python's recursion limit and ends the run")
    ap.add_argument("--no-memo", action="store_true", help="async A/B: no leaf-evaluation memo (every leaf goes to the network)")
    ap.add_argument("--bn", default="per_leaf", choices=["per_leaf", "eval", "batch"])
    ap.add_argument("--nn-dtype", default="fp32", choices=["fp32", "bf16", "fp16"], help="fp32: the parity mode.  fp16: NON-PARITY throughput mode (async only; never the "
                                                                                         "headline).  bf16: library ops, lock-step mode only")
    ap.add_argument("--channels-last", type=int, default=1)
    ap.add_argument("--desync-plies", type=int, default=700)
    ap.add_argument("--desync-playouts", type=int, default=4)
    ap.add_argument("--settle-rounds", type=int, default=2560, help="async: untimed rounds at the full playout count between the desync phase and the warm-up steps")
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--cpu-seconds", type=float, default=16.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fix-terminal-sign", action="store_true", help="NOT the headline: back a winning move up as +1 (the reference backs it up as -1, mcts.py:125)")
    ap.add_argument("--second-line-seconds", type=float, default=5.0, help="async, 1 GPU: length of the labelled second run with the terminal sign fixed (0 = skip it)")
    ap.add_argument("--second-line-warm-seconds", type=float, default=12.0)
    ap.add_argument("--second-line-budget-us", type=int, default=1000, help="budget of a k_advance launch in the second line")
    ap.add_argument("--second-line-boards", type=int, default=16384, help="boards of the second line's engine")
    ap.add_argument("--no-non-parity-line", action="store_true", help="skip the labelled NON_PARITY second line (sign fixed + one fp16 MFMA per product)")
    ap.add_argument("--no-c3", action="store_true", help="skip the 32,768-board microbenchmark line (roofline_c3)")
    ap.add_argument("--no-planes", action="store_true", help="lockstep: the rules op only produces the legal sets (the evaluator reads the leaf boards)")
    ap.add_argument("--library-trunk", action="store_true", help="lockstep A/B: trunk convolutions through MIOpen instead of the split-fp16 MFMA kernel")
    ap.add_argument("--separate-descent", action="store_true", help="lockstep A/B: k_select as its own launch")
    ap.add_argument("--select-opts", type=int, default=8, help="switches of the descent / the launch (qz_config.select_opts); 8 = ONE deadline per k_advance launch, counted from its first "
                                                                 "wavefront, boards take the first slots in turn (round 5's default: 13,312 boards at 3,000 us = +6 %% over 10,240 boards with "
                                                                 "per-board budgets of 2,400 us on the same box; 0 = per-board budgets)")
    ap.add_argument("--rules-variant", type=int, default=0, help="lockstep A/B: qz_rules_opts.variant of the engines' leaf rules op")
    ap.add_argument("--length-file", default=None, help="game-length sample for games_per_s_steady_state (default: newest profiles/round*/game_length_<n>playouts.json)")
    ap.add_argument("--clock-log", default=None, help="write the clock / power samples of the timed region to this JSON file")
    ap.add_argument("--dump-roots", default=None, help="async: after the timed region write every board's root position to this .npz (the steady-state sample behind cpu_baseline.by_phase)")
    args = ap.parse_args()
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("QZ_BENCH_LIB"):  # A/B of a differently built library on the same box; never set by the driver
        from alphazero_quoridor_amd import _cabi

        _cabi.LIB_PATH = os.environ["QZ_BENCH_LIB"]
    from alphazero_quoridor_amd import dist as qdist
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

    rank, local, world = qdist.init_from_env("cuda")
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU path)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(args.seed)  # identical random-init weights on every rank
    net = PolicyValueNet(use_gpu=True, device=dev)
    eng = make_engine(args, net, dev, qdist.shard_seed(args.seed, rank), args.fix_terminal_sign)
    R = Run(args, eng, qdist, rank, local, world, dev)
    if args.mode == "async":
        assert args.nn_dtype in ("fp32", "fp16") and args.bn == "per_leaf" and not args.library_trunk, "the asynchronous loop runs the HIP evaluation (per-leaf BN)"
        run_async(R)
    else:
        run_lockstep(R)
    eng.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
