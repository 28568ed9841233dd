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
statistics; weak scaling (4096 boards on every rank), finished tuples all-gathered every step.

DEFAULT ROUTE (--mode async): the asynchronous self-play loop (qz_selfplay_*, include/qz_abi.h).
Every board runs its 400 playouts per move on its own clock; a leaf whose evaluation is in the
leaf-evaluation memo (policy_value_fn on a batch of one is a pure function of the 24-byte board)
is expanded from the memo, every other leaf is evaluated by the network as before, bit for bit
the same search as the lock-step engine (tests/test_gpu_async.py).  A STEP is --rounds-per-step
(256) rounds, a round = one pass of the hot path over all boards: k_advance (playouts until every
board needs the network or its time budget is used), the network on the leaves the memo does not
know -- with Quoridor.actions() of those leaves and k_moves (the moves of the boards that have
done their playouts) on a second stream beside it -- and the memo insert.  Then the
finished games are harvested (+ all-gathered).  Nothing is skipped: every one of the 400
playouts of every move descends, expands and backs up exactly as the reference does, every leaf
gets the reference's evaluation -- from the network the first time, from the memo afterwards.
--mode lockstep is round 2's route (every leaf through the network), kept for A/B.

games/sec needs finished games, and a reference-faithful 400-playout game lasts tens of thousands of plies
(median 5,000, mean ~80,000: the reference backs a won position up with the wrong sign, mcts.py:119-125), so the boards are
first DESYNCHRONISED (untimed): `--desync-plies` plies at `--desync-playouts` playouts per move
spread the population over all game phases (continuous refill).
  value                     = games_per_s_steady_state.value when a length sample of this playout
                              count is committed under profiles/ (benchmarks/game_length.py), else
                              the raw count below
  games_in_timed_region     = games that finished inside the K timed steps (raw count; the population
                              was desynchronised with short searches and is not yet stationary)
  games_per_s_steady_state  = what a long job converges to: boards / E[wall time of a game], with the
                              per-ply cost of the two phases of a game (mover still has walls: almost
                              every leaf is new and goes to the network; later: almost every leaf is in
                              the memo) measured in the timed region and the plies per phase from the
                              committed length sample
  plies_per_s, playouts_per_s are what GPU and CPU are compared on.

roofline       the dominant kernel of the route: k_advance (async: every descent, expansion and
               backup of a round) or the rules op (lockstep), HIP events around every launch in the
               timed region on the launch stream; algorithmic bytes from the engine's counters.
roofline_rules the rules op (Quoridor.actions() of the miss list) of every round, same method.
roofline_nn    the network launches (k_trunk + k_head_fc on the miss list).
roofline_c3    the rules op on 32,768 mid-game boards (BASELINE configs[2]), timed right after
               the timed region.
cpu_baseline   the CPU oracle (oracle/, scalar C port of the reference's algorithm, one playout
               at a time, batch-1 network on the CPU like the reference) on this host: one core
               and all cores (independent processes), plus the reference-on-this-host estimate
               through the port/reference ratio measured in the build container.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_BOARD = 24 + 20 + 26 * 81 * 4
PROFILES = os.path.join(ROOT, "profiles")


def _latest_profile(name):
    """profiles/round*/<name> of the newest round that has it."""
    hits = sorted(glob.glob(os.path.join(PROFILES, "round*", name)), key=lambda p: int(re.search(r"round(\d+)", p).group(1)))
    return hits[-1] if hits else None


def _load_json(path):
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError, TypeError):
        return None


# ------------------------------------------------------------------------------ clocks / power
class ClockSampler(threading.Thread):
    """Samples the GPU's shader clock, power and temperature while the timed region runs (sysfs
    when readable, else `rocm-smi --json` as a child process): evidence for the gap between the
    step time of a short profile and of a sustained run."""

    def __init__(self, index=0, period=1.0):
        super().__init__(daemon=True)
        self.index, self.period = index, period
        self.samples = []
        self._stop_ev = threading.Event()
        self.source = None
        self._sysfs = self._find_card(index)

    @staticmethod
    def _find_card(index):
        """/sys/class/drm/cardN/device of HIP device `index`, matched by PCI address (a box shows
        every GPU / partition of the host in sysfs, the process sees only its own)."""
        try:
            import ctypes

            hip = ctypes.CDLL("libamdhip64.so")
            buf = ctypes.create_string_buffer(64)
            if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) != 0:
                return None
            bdf = buf.value.decode().lower()
            for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
                if os.path.basename(os.path.realpath(card)).lower() == bdf and os.path.exists(os.path.join(card, "pp_dpm_sclk")):
                    return card
        except Exception:  # noqa: BLE001
            pass
        return None

    def _read_sysfs(self):
        d = self._sysfs
        out = {}
        with open(os.path.join(d, "pp_dpm_sclk")) as f:
            cur = [ln for ln in f.read().splitlines() if ln.rstrip().endswith("*")]
        m = re.search(r"(\d+)\s*[Mm][Hh]z", cur[0]) if cur else None
        if m:
            out["sclk_mhz"] = float(m.group(1))
        for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
            for key, fn, scale in (("power_w", "power1_average", 1e-6), ("power_w", "power1_input", 1e-6), ("temp_c", "temp1_input", 1e-3)):
                p = os.path.join(hw, fn)
                if key not in out and os.path.exists(p):
                    try:
                        out[key] = float(open(p).read().strip()) * scale
                    except (OSError, ValueError):
                        pass
        return out

    def _read_smi(self):
        r = subprocess.run(["rocm-smi", "-d", str(self.index), "--showclocks", "--showpower", "--showtemp", "--json"],
                           capture_output=True, text=True, timeout=20)
        card = next(iter(json.loads(r.stdout).values()))
        out = {}
        for k, v in card.items():
            kl = k.lower()
            m = re.search(r"([\d.]+)", str(v))
            if not m:
                continue
            if kl.startswith("sclk"):
                out["sclk_mhz"] = float(m.group(1))
            elif "power" in kl and "power_w" not in out:
                out["power_w"] = float(m.group(1))
            elif "temperature" in kl and ("junction" in kl or "temp_c" not in out):
                out["temp_c"] = float(m.group(1))
        return out

    def run(self):
        t0 = time.perf_counter()
        while not self._stop_ev.is_set():
            s = None
            for name, fn in (("sysfs", self._read_sysfs if self._sysfs else None), ("rocm-smi", self._read_smi)):
                if fn is None or (self.source not in (None, name)):
                    continue
                try:
                    s = fn()
                    if s:
                        self.source = name
                        break
                except Exception:  # noqa: BLE001 -- monitoring must never break the benchmark
                    s = None
            if s:
                s["t"] = time.perf_counter() - t0
                self.samples.append(s)
            elif self.source is None and time.perf_counter() - t0 > 30:
                return  # nothing readable on this box
            self._stop_ev.wait(self.period)

    def stop(self):
        self._stop_ev.set()
        self.join(timeout=30)

    def summary(self):
        if not self.samples:
            return {"source": None, "note": "no clock/power interface readable on this box"}
        out = {"source": self.source, "samples": len(self.samples), "sysfs_card": self._sysfs}
        for k in ("sclk_mhz", "power_w", "temp_c"):
            v = [s[k] for s in self.samples if k in s]
            if v:
                out[k] = {"min": min(v), "mean": sum(v) / len(v), "max": max(v), "first": v[0], "last": v[-1]}
        return out


# ------------------------------------------------------------------------------ CPU baseline
def usable_cores():
    """Cores this process may really use: the scheduler affinity mask, capped by the cgroup CPU
    quota (os.cpu_count() reports the host's 256 hardware threads on a box that grants far fewer)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


def cpu_baseline(seconds, n_playout, mean_plies_per_game, length_source):
    """The oracle port on this host: 1 process, then os.cpu_count() independent processes side
    by side (python -m oracle.cpu_baseline: one torch thread each).  playouts/s is the measured
    quantity; games/s divides it by n_playout and by the SAME plies-per-game the GPU's
    steady-state estimate uses."""
    cores = usable_cores()
    half = max(seconds / 2.0, 2.0)
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--seconds", "%.1f" % half, "--n-playout", str(n_playout)]

    def run_many(k):
        procs = [subprocess.Popen(cmd + ["--seed", str(i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env, cwd=ROOT)
                 for i in range(k)]
        res = []
        for p in procs:
            out = p.communicate()[0]
            lines = [ln for ln in out.splitlines() if ln.startswith("{")]
            if p.returncode == 0 and lines:
                res.append(json.loads(lines[-1]))
        return res

    one = run_many(1)
    many = run_many(cores)
    if not one or not many:
        return {"value": None, "unit": "games/s", "cores": cores, "kind": "port", "sample": "oracle.cpu_baseline failed to run"}
    pps1 = one[0]["playouts"] / one[0]["seconds"]
    ppsN = sum(r["playouts"] / r["seconds"] for r in many)
    cal = _load_json(_latest_profile("cpu_calibration.json") or "")
    out = {
        "value": (ppsN / n_playout / mean_plies_per_game) if mean_plies_per_game else None,
        "unit": "games/s", "cores": len(many), "host_hardware_threads": os.cpu_count(), "kind": "port",
        "sample": "%d + %d x %d playouts (%.0f s on 1 core, then %.0f s on %d cores as independent processes) of the first ply at "
                  "n_playout=%d from the opening (131 legal moves): oracle C port + batch-1 fp32 torch-CPU forward per leaf, 1 torch "
                  "thread per process; games/s = playouts/s / %d / %s plies per game (%s)"
                  % (one[0]["playouts"], len(many), int(np.mean([r["playouts"] for r in many])), half, half, len(many), n_playout,
                     n_playout, "%.0f" % mean_plies_per_game if mean_plies_per_game else "?", length_source),
        "compare_on": "playouts_per_s (length-independent)",
        "playouts_per_s_1core": pps1, "playouts_per_s_allcores": ppsN,
        "games_per_s_1core": (pps1 / n_playout / mean_plies_per_game) if mean_plies_per_game else None,
    }
    if cal and cal.get("port_over_reference"):
        r = float(cal["port_over_reference"])
        out["reference_estimate"] = {
            "playouts_per_s_1core": pps1 / r, "playouts_per_s_allcores": ppsN / r, "port_over_reference": r,
            "calibration": "pure-Python reference vs this C port on the build container's host (%s; %.2f vs %.1f playouts/s): "
                           "benchmarks/calibrate_cpu_port.py" % (cal.get("host_cpu", "?"), cal["reference_playouts_per_s"], cal["port_playouts_per_s"]),
        }
    return out


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
    t = _load_json(_latest_profile("pmc_traffic_c3.json") or "")
    return {"workload": "BASELINE configs[2] microbenchmark: 32,768 boards (S-mid: 0..20 plies of random legal play, mover has a wall), "
                        "actions() + state(), inputs resident in HBM, NOT part of the timed region",
            "kernel": "k_pool_paths_enc + k_pool_masks_enc (pooled pipeline, two launches)",
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": t.get("traffic_bytes_per_launch") if t else None,
            "avg_launch_us": us, "launches": launches, "algorithmic_bytes_per_launch": n * BYTES_PER_BOARD,
            "cache_note": "277 MB written per launch: larger than the 256-MiB Infinity Cache, the stores reach HBM"}


def launch_ranks(n_gpus, argv):
    """`python bench.py --gpus N` without a launcher: become the launcher.  The ranks are CHILD
    processes (this process never initialises HIP and never re-execs); stdout of the job is the
    single JSON line rank 0 prints, everything else goes to stderr."""
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


def steady_state(plies_per_s, length_file):
    """plies/s / E[length of a 400-playout game], with a 95 % interval from the length sample
    (Kaplan-Meier mean of benchmarks/game_length.py when games were still running at its end)."""
    d = _load_json(length_file or "")
    if not d or not d.get("mean_plies_per_game"):
        return None
    L = float(d["mean_plies_per_game"])
    lo, hi = (d.get("mean_ci95") or [None, None])[:2]
    return {
        "value": plies_per_s / L, "unit": "games/s", "plies_per_s": plies_per_s, "mean_plies_per_game": L,
        "ci95": [plies_per_s / hi, plies_per_s / lo] if lo and hi else None,
        "n_games_in_length_sample": d.get("games_finished"), "n_games_censored": d.get("games_censored"),
        "length_estimator": d.get("estimator"), "length_source": os.path.relpath(length_file, ROOT),
        # the model-free part: E[min(L, T)] <= E[L], so plies/s / restricted mean is an UPPER bound of games/s.  The hazard of
        # these games keeps falling with their age (heavy tail), so the exponential tail makes `value` lean high, not low
        "upper_bound": plies_per_s / float(d["restricted_mean"]) if d.get("restricted_mean") else None,
        "restricted_mean_plies": d.get("restricted_mean"), "observation_window_plies": d.get("T"), "survival_at_window": d.get("survival_at_T"),
    }


def steady_state_two_phase(boards_all, open_plies, end_plies, open_board_s, end_board_s, length_file):
    """Games/s of a stationary population: boards / E[wall time of one game], E = plies of a game in each of its two
    phases (committed length sample) x board-seconds per ply of that phase (measured in the timed region).  The phases:
    the root's mover still has walls (131 legal moves, almost every leaf is new: one network round trip per playout)
    and afterwards (a few thousand positions revisited for the rest of the game: the memo answers).  plies/s alone
    would weight the phases by the timed region's population, not by a game's."""
    d = _load_json(length_file or "")
    if not d or not d.get("mean_plies_per_game") or not d.get("mean_open_plies_per_game") or open_plies <= 0 or end_plies <= 0:
        return None
    L, Lo = float(d["mean_plies_per_game"]), float(d["mean_open_plies_per_game"])
    c_open, c_end = open_board_s / open_plies, end_board_s / end_plies
    dur = Lo * c_open + (L - Lo) * c_end
    lo, hi = (d.get("mean_ci95") or [None, None])[:2]

    def rate(length):
        return boards_all / (Lo * c_open + (length - Lo) * c_end)

    return {
        "value": boards_all / dur, "unit": "games/s", "estimator": "boards / (open plies x board-seconds per open ply + late plies x board-seconds per late ply)",
        "mean_plies_per_game": L, "mean_open_plies_per_game": Lo, "board_seconds_per_open_ply": c_open, "board_seconds_per_late_ply": c_end,
        "seconds_per_game_per_board": dur, "open_phase_share_of_a_game": Lo * c_open / dur,
        "ci95": [rate(hi), rate(lo)] if lo and hi else None,
        "n_games_in_length_sample": d.get("games_finished"), "n_games_censored": d.get("games_censored"),
        "length_estimator": d.get("estimator"), "length_source": os.path.relpath(length_file, ROOT),
        "upper_bound": rate(float(d["restricted_mean"])) if d.get("restricted_mean") else None,
        "restricted_mean_plies": d.get("restricted_mean"), "observation_window_plies": d.get("T"), "survival_at_window": d.get("survival_at_T"),
    }


def run_async(args, eng, net, rank, local, world, dev, qdist):
    """The default route: the asynchronous self-play loop (see the module docstring)."""
    import ctypes as C

    from alphazero_quoridor_amd import _cabi

    is_dist = world > 1
    B, G = args.boards, args.groups
    gb = B // G
    R = args.rounds_per_step
    kw = dict(max_playouts=args.max_playouts, budget_us=args.budget_us)

    def barrier():
        if is_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    lengths = {"desync": [], "warmup": [], "timed": []}
    phase = ["desync"]

    def end_of_step():
        """harvest every group's finished games (+ the path's only exchange: finished tuples -> every rank's buffer)"""
        tbs = eng.harvest()
        n_games = 0
        for tb in tbs:
            n_games += tb.n_games
            gid = tb.game.cpu().numpy()
            assert gid.size and 0 <= int(gid.min()) and int(gid.max()) < tb.n_games, "corrupt game ids in a harvest"
            lengths[phase[0]].extend(np.bincount(gid, minlength=tb.n_games).tolist())
        if is_dist:
            eng.synchronize()
            bufs = [qdist.pack_tuples(tb.boards.hbits, tb.boards.vbits, tb.boards.meta, tb.pi, tb.z) for tb in tbs]
            bufs.append(torch.zeros((0, qdist.TUPLE_BYTES), dtype=torch.uint8, device=dev))
            qdist.allgather_tuples(torch.cat(bufs))
        return n_games

    # ---- desynchronise the games (untimed): short searches spread the boards over all game phases
    t0 = time.time()
    if args.desync_plies > 0:
        eng.set_playouts(args.desync_playouts)
        # a ply of a board costs at most desync_playouts + 1 rounds (every leaf new), fewer once the memo answers
        for _ in range(0, args.desync_plies * (args.desync_playouts + 1), 64):
            eng.run_rounds(64, **kw)
            end_of_step()
        eng.set_playouts(args.playouts)
    desync_s = time.time() - t0
    if args.graph_rounds:
        eng.capture_rounds(rounds=args.graph_rounds, **kw)

    # ---- warmup steps at the full playout count (untimed)
    phase[0] = "warmup"
    for _ in range(args.warmup):
        eng.run_rounds(R, **kw)
        end_of_step()

    # ---- timed region.  Every `ev_every`-th round of group 0 is issued as its four pieces with HIP events around them
    phase[0] = "timed"
    eng0, ev0, st0_stream = eng.engines[0], eng.evaluators[0], eng.streams[0]
    L = _cabi.load()
    ev_every = max(1, args.event_every)
    names = ("advance", "rules", "nn", "tail")
    evs = []

    def ev_pair():
        return (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    def timed_round():
        """one round of group 0 through the split entry points, each piece bracketed by events on the launch stream"""
        with torch.cuda.stream(st0_stream):
            eng0._memo_guard(ev0)
            s = eng0._s()
            pairs = [ev_pair() for _ in names]
            w = ev0.nn_weights()
            for (a, b), call in zip(pairs, (lambda: L.qz_selfplay_advance(eng0.h, args.max_playouts, args.budget_us, 1, s),
                                            lambda: L.qz_selfplay_leaf_rules(eng0.h, s),
                                            lambda: L.qz_selfplay_evaluate(eng0.h, C.byref(w), s),
                                            lambda: L.qz_selfplay_round_tail(eng0.h, s))):
                a.record()
                _cabi.check(call())
                b.record()
            evs.append(pairs)

    sampler = ClockSampler(local) if rank == 0 else None
    st0 = eng.stats()
    barrier()
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    games = 0
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        done = 0
        while done < R:
            if args.graph_rounds == 0 and ev_every <= R:
                timed_round()                      # group 0, one round, with events
                for g in range(1, G):              # the other groups' matching round
                    with torch.cuda.stream(eng.streams[g]):
                        eng.engines[g].run_rounds(eng.evaluators[g], 1, **kw)
                done += 1
            n = min(ev_every - 1 if args.graph_rounds == 0 else R - done, R - done)
            if n > 0:
                eng.run_rounds(n, **kw)
                done += n
        games += end_of_step()  # harvest synchronises with the device (qz_harvest_counts), so this is the step's wall time
        step_ms.append((time.perf_counter() - ts) * 1e3)
    barrier()
    elapsed = time.perf_counter() - t0
    if sampler:
        sampler.stop()
    st1 = eng.stats()
    d = {k: st1[k] - st0[k] for k in st1}

    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    tot = torch.tensor([games, d["plies_played"], d["playouts"], d["leaf_terminal"], d["nn_evals"], d["memo_hits"], d["open_plies"], d["open_rounds"]],
                       dtype=torch.float64, device=dev)
    if is_dist:
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(tot, op=torch.distributed.ReduceOp.SUM)
    elapsed = float(el.item())
    games_all, plies_all, playouts_all, term_all, evals_all, hits_all, open_plies_all, open_rounds_all = (float(x) for x in tot.tolist())
    if rank != 0:
        return
    rounds = args.steps * R
    round_s = elapsed / rounds
    # board-seconds per phase: a board is in the open phase in the launches k_advance counted (root's mover has walls)
    open_board_s = open_rounds_all * round_s
    end_board_s = max(B * world * rounds - open_rounds_all, 0.0) * round_s
    length_file = args.length_file or _latest_profile("game_length_%dplayouts.json" % args.playouts)
    if args.fix_terminal_sign:
        length_file = args.length_file  # the committed sample is for the reference-faithful sign
    ss = steady_state_two_phase(B * world, open_plies_all, plies_all - open_plies_all, open_board_s, end_board_s, length_file)
    ss_simple = steady_state(plies_all / elapsed, length_file)

    def avg_us(i):
        return sum(p[i][0].elapsed_time(p[i][1]) for p in evs) / max(len(evs), 1) * 1e3

    adv_us, rules_us, nn_us, tail_us = (avg_us(i) for i in range(4)) if evs else (None,) * 4
    launches = rounds * G * world
    per_launch = {k: d[k] / (rounds * G) for k in ("edges_scanned", "edges_expanded", "descent_levels", "playouts", "memo_hits", "nn_evals")}
    # k_advance's algorithmic bytes: every level of a descent reads the node's edge records (32 B each) and its 12-byte
    # record entry, the backup rewrites 24 B per level, a new leaf probes one 512-byte memo bucket and its expansion
    # writes 32 B per legal move
    adv_bytes = (per_launch["edges_scanned"] * 32 + per_launch["descent_levels"] * (12 + 24) + (per_launch["memo_hits"] + per_launch["nn_evals"]) * 512
                 + per_launch["edges_expanded"] * 32)
    t_adv = _load_json(_latest_profile("pmc_traffic_advance.json") or "")

    def line(kernel, us, nbytes, note, traffic=None, src=None):
        if us is None:
            return None
        gbs = nbytes / (us * 1e-6) / 1e9
        return {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": src, "avg_launch_us": us, "launches_timed": len(evs), "algorithmic_bytes_per_launch": nbytes, "note": note}

    def len_stats(v):
        return {"n": len(v), "mean": float(np.mean(v)) if v else None, "median": float(np.median(v)) if v else None}

    miss_per_round = evals_all / max(rounds * G * world, 1)
    conv_flops = 2.0 * 81 * 9 * (10 * 64 * 64 + 64 * 6 + 26 * 64) + 2.0 * (324 * 128 + 128 + 162 * 140)
    out = {
        "metric": "self-play games/sec (9x9, n_playout=%d)" % args.playouts,
        "value": ss["value"] if ss else games_all / elapsed,
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
            "workload": ("NON-PARITY THROUGHPUT MODE (network products on fp16 operands, p / v ~1e-3 from the reference; %s) -- "
                         % ("terminal sign fixed" if args.fix_terminal_sign else "reference-faithful terminal sign") if args.nn_dtype == "fp16" else "") +
                        "%s: %d concurrent boards/GPU, n_playout=%d, 9x9, 10 walls/player, c_puct=5, temp=1.0, asynchronous self-play loop with "
                        "leaf-evaluation memo, finished tuples all-gathered every step"
                        % ("BASELINE configs[2] as an engine run" if B == 32768 else ("BASELINE configs[1]" if args.playouts == 100 else "BASELINE configs[3] per GPU"),
                           B, args.playouts),
            "mode": "async", "boards_per_gpu": B, "board_groups": G, "fix_terminal_sign": bool(args.fix_terminal_sign), "n_playout": args.playouts,
            "bn_mode": args.bn, "nn_dtype": args.nn_dtype,
            "step": "%d rounds; a round = k_advance (every board: playouts until it needs the network, %d us budget) + network on the leaves the memo "
                    "does not know (actions() of those leaves and k_moves, the moves of the boards that finished their playouts, on a second stream "
                    "beside the trunk) + memo insert; every %d-th round is issued piece by piece (k_moves + k_advance, actions(), network, tail) "
                    "with HIP events around the pieces" % (R, args.budget_us, ev_every),
            "max_depth": args.max_depth, "rounds_per_step": R, "budget_us": args.budget_us, "max_playouts_per_round": args.max_playouts, "graph_rounds": args.graph_rounds,
            "desync": "%d untimed plies at %d playouts/move (%.0fs, %d games finished)" % (args.desync_plies, args.desync_playouts, desync_s, len(lengths["desync"])),
        },
        "value_is": ("games_per_s_steady_state.value: boards / E[wall time of a game] (plies per phase from %s, cost per ply of each phase measured in "
                     "the timed region)" % os.path.relpath(length_file, ROOT)) if ss else
                    "games finished inside the timed region / wall time (no committed length sample with a phase split for this playout count)",
        "games_in_timed_region": games_all,
        "games_in_timed_region_per_s": games_all / elapsed,
        "games_per_s_steady_state": ss,
        "games_per_s_plies_over_mean_length": ss_simple,
        "plies_per_s": plies_all / elapsed,
        "playouts_per_s": playouts_all / elapsed,
        "nn_evaluations_per_s": evals_all / elapsed,
        "memo_hit_rate": hits_all / max(playouts_all, 1.0),
        "leaf_evals_per_s": evals_all / elapsed,
        "terminal_leaf_frac": term_all / max(playouts_all, 1.0),
        "open_phase": {"plies": open_plies_all, "share_of_board_time": open_rounds_all / max(B * world * rounds, 1), "note": "root's mover still has walls"},
        "game_lengths_seen": {"timed_region_%d_playouts" % args.playouts: len_stats(lengths["timed"] + lengths["warmup"]),
                              "desync_phase_%d_playouts" % args.desync_playouts: len_stats(lengths["desync"]),
                              "note": "games finishing in the timed region started in the desync phase; neither is the length of a %d-playout game" % args.playouts},
        "mean_descent_depth": d["descent_levels"] / max(d["playouts"], 1),
        "rounds": rounds, "ms_per_round": round_s * 1e3,
        "ms_per_step_series": [round(x, 1) for x in step_ms],
        "roofline": line("k_moves + k_advance (one wavefront per board: moves of the boards that finished their playouts, then descents (recorded descents "
                         "replayed 64 levels per round trip), memo probes, expansions and backups until the board needs the network or the budget is used)",
                         adv_us, adv_bytes,
                         "dependent-load latency and instruction issue, not bandwidth: a playout is a chain of ~10 memory round trips (record -> edge "
                         "blocks -> ... -> memo bucket -> backup) of a single wavefront with ~1,300 instructions between them (the PUCT expression in "
                         "float64), four wavefronts per SIMD which slow each other by 15 %% (SQ counters: a wavefront issues in 39 %% of its cycles, waits for "
                         "memory in 44 %%: profiles/round3/pmc_sq_async_late_game.json); %.0f playouts per launch, mean depth %.1f.  The launch lasts its time budget + the last "
                         "playouts (subtree copies stop at the budget and resume in the next launch); see DESIGN 3.0"
                         % (per_launch["playouts"], d["descent_levels"] / max(d["playouts"], 1)),
                         traffic=(t_adv or {}).get("traffic_bytes_per_launch"), src=("profiles: " + os.path.relpath(_latest_profile("pmc_traffic_advance.json"), ROOT)) if t_adv else None),
        "roofline_rules": line("k_wave_rules on the miss list (Quoridor.actions() of the leaves the memo does not know; legal sets only)", rules_us, miss_per_round * 44,
                               "%.0f leaves per launch: a launch lasts as long as one board's dependent chain" % miss_per_round),
        "roofline_nn": None if nn_us is None else {
            "kernel": "k_trunk<true> + k_head_fc on the miss list (first layer from the packed boards, ten conv3x3 64->64 layers as implicit GEMMs on "
                      "v_mfma_f32_32x32x16_f16 with split fp16 operands, per-leaf normalisation, heads)",
            "bound": "mfma", "achieved": conv_flops * miss_per_round / (nn_us * 1e-6) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
            "frac": conv_flops * miss_per_round / (nn_us * 1e-6) / 1e12 / 2500.0, "traffic": None, "avg_launch_us": nn_us, "launches_timed": len(evs),
            "leaves_per_launch": miss_per_round,
            "note": "one leaf per 2-wave workgroup, 1,024 workgroups resident: with %.0f leaves per launch every SIMD issues about one leaf's 28.5 k "
                    "MFMAs (three products per fp32-accurate product: 3.56x the useful flops counted here), a second pass above 1,024 leaves; at 4,096 "
                    "leaves per launch (lock-step mode, the opening phase) the same kernel reaches 0.135" % miss_per_round},
        "round_tail_us": tail_us,
        "games_dropped_in_timed_region": {"total": d["games_aborted"], "depth_over_%d_levels" % args.max_depth: d["aborted_depth"], "no_legal_move": d["aborted_no_move"],
                                          "note": "the reference cannot finish these games either: a path longer than 992 levels overflows its recursive backup "
                                                  "(RecursionError, mcts.py:55-62), a root without a legal move crashes start_self_play (mcts.py:195-196)"},
        "engine_stats": {kk: st1[kk] for kk in ("node_overflow", "games_aborted", "aborted_no_move", "aborted_max_plies", "aborted_pool", "aborted_depth", "nonfinite_values",
                                                "runaway_descents", "compact_slices", "arena_bytes", "max_nodes", "max_edges", "max_depth", "tree_pages_total", "tree_pages_peak",
                                                "traj_pages_total", "traj_pages_peak", "memo_inserts", "memo_locked")},
        "clocks": sampler.summary() if sampler else None,
    }
    if st1["games_aborted"]:
        sys.stderr.write("bench.py: NOTE %d games were dropped (depth > %d: %d, no legal move: %d, max_plies %d, pool %d)\n"
                         % (st1["games_aborted"], args.max_depth, st1["aborted_depth"], st1["aborted_no_move"], st1["aborted_max_plies"], st1["aborted_pool"]))
    assert st1["node_overflow"] == 0 and st1["runaway_descents"] == 0, "tree storage overflowed / corrupted during the run"
    if args.clock_log and sampler:
        with open(args.clock_log, "w") as f:
            json.dump({"step_ms": step_ms, "samples": sampler.samples}, f)
    if world == 1 and not args.no_c3:
        out["roofline_c3"] = c3_microbench(dev)
    if not args.no_cpu_baseline and world == 1:
        eng.close()
        torch.cuda.empty_cache()
        src = ss or ss_simple
        if src:
            L_cpu, L_src = src["mean_plies_per_game"], src["length_source"]
        else:
            seen = lengths["timed"] + lengths["warmup"] + lengths["desync"]
            L_cpu = float(np.mean(seen)) if seen else None
            L_src = "NO length sample for n_playout=%d: mean length of the %d games finished in this run (mostly desync games)" % (args.playouts, len(seen))
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.playouts, L_cpu, L_src)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--boards", type=int, default=4096)
    ap.add_argument("--playouts", type=int, default=400)
    ap.add_argument("--groups", type=int, default=1,
                    help="split the boards of a GPU into this many independent groups on their own HIP streams")
    ap.add_argument("--mode", default="async", choices=["async", "lockstep"],
                    help="async (default): the asynchronous self-play loop with the leaf-evaluation memo; lockstep: round 2's route, every leaf through the network")
    ap.add_argument("--rounds-per-step", type=int, default=256, help="async: rounds of the loop per step")
    ap.add_argument("--budget-us", type=int, default=1000, help="async: wall-clock budget of a k_advance launch")
    ap.add_argument("--max-playouts", type=int, default=4096, help="async: playouts a board may start per round")
    ap.add_argument("--graph-rounds", type=int, default=0, help="async: capture this many (even) rounds per HIP graph (0 = eager launches, per-kernel events)")
    ap.add_argument("--event-every", type=int, default=8, help="async: every n-th round of group 0 is issued in pieces with HIP events around them")
    ap.add_argument("--max-depth", type=int, default=992,
                    help="drop a game whose playout descends more than this many levels (0 = never).  992 = where the reference's recursive backup "
                         "(mcts.py:55-62) overflows Python's recursion limit under `python train.py` and ends the run with a RecursionError")
    ap.add_argument("--no-memo", action="store_true", help="async A/B: no leaf-evaluation memo (every leaf goes to the network)")
    ap.add_argument("--bn", default="per_leaf", choices=["per_leaf", "eval", "batch"])
    ap.add_argument("--nn-dtype", default="fp32", choices=["fp32", "bf16", "fp16"],
                    help="fp32 (default): the parity mode.  fp16: NON-PARITY throughput mode of the HIP evaluation (one MFMA per product on fp16 operands, "
                         "p / v within ~1e-3 of the reference instead of 1e-5), async mode only; never the headline.  bf16: library ops, lock-step mode only")
    ap.add_argument("--channels-last", type=int, default=1)
    ap.add_argument("--desync-plies", type=int, default=700)
    ap.add_argument("--desync-playouts", type=int, default=4)
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--cpu-seconds", type=float, default=16.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fix-terminal-sign", action="store_true",
                    help="NOT the headline: back a winning move up as +1 (the reference backs it up as -1, mcts.py:125, which makes "
                         "searches avoid winning and games run for thousands of plies)")
    ap.add_argument("--no-c3", action="store_true", help="skip the 32,768-board microbenchmark line (roofline_c3)")
    ap.add_argument("--no-planes", action="store_true",
                    help="product default of the engine route (the evaluator reads the leaf boards, state() is never materialised); "
                         "the rules op then only produces the legal sets and `roofline` is computed on 44 B/board")
    ap.add_argument("--library-trunk", action="store_true",
                    help="NOT the default: run the trunk convolutions through MIOpen (fp32 implicit GEMM) + the separate normalisation "
                         "kernel instead of the split-fp16 MFMA kernel (qz_nn_conv3x3_norm), for A/B runs")
    ap.add_argument("--separate-descent", action="store_true", help="A/B: k_select as its own launch (default: fused into the previous playout's expand / backup launch)")
    ap.add_argument("--select-opts", type=int, default=0, help="A/B switches of k_select (qz_config.select_opts)")
    ap.add_argument("--rules-variant", type=int, default=0,
                    help="A/B: qz_rules_opts.variant of the engines' leaf rules op (0 = the library's choice by batch size; include/qz_abi.h)")
    ap.add_argument("--length-file", default=None,
                    help="game-length sample for games_per_s_steady_state (default: newest profiles/round*/game_length_<n>playouts.json)")
    ap.add_argument("--clock-log", default=None, help="write the clock / power samples of the timed region to this JSON file")
    args = ap.parse_args()
    if args.gpus > 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    if os.environ.get("QZ_BENCH_LIB"):  # A/B of a differently built library on the same box; never set by the driver
        from alphazero_quoridor_amd import _cabi

        _cabi.LIB_PATH = os.environ["QZ_BENCH_LIB"]
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
    dt = torch.bfloat16 if args.nn_dtype == "bf16" else torch.float32
    prec = "fp16" if args.nn_dtype == "fp16" else "fp32"
    if args.library_trunk:
        from alphazero_quoridor_amd.policy_value_net import LeafEvaluator
        lib_ev = LeafEvaluator(net.policy_value_net, args.bn, dt, bool(args.channels_last), mfma_trunk=False)
        make_ev = lambda: lib_ev  # noqa: E731
    else:
        make_ev = lambda: net.evaluator(args.bn, dt, bool(args.channels_last), nn_precision=prec)  # noqa: E731
    eng = BoardGroups(args.boards, args.groups, make_ev,
                      seed=qdist.shard_seed(args.seed, rank), device=dev,
                      n_playout=args.playouts, c_puct=5, temp=1.0, is_selfplay=1, fix_terminal_sign=args.fix_terminal_sign,
                      select_opts=args.select_opts, memo=not args.no_memo, max_depth=args.max_depth)
    if args.mode == "async":
        assert args.nn_dtype in ("fp32", "fp16") and args.bn == "per_leaf" and not args.library_trunk, "the asynchronous loop runs the HIP evaluation (per-leaf BN)"
        run_async(args, eng, net, rank, local, world, dev, qdist)
        eng.close()
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if args.rules_variant:
        from alphazero_quoridor_amd import rules as qrules

        for e in eng.engines:
            e.set_rules_opts(qrules.rules_opts(args.rules_variant))
    group_boards = args.boards // args.groups
    is_dist = world > 1
    write_planes = not args.no_planes
    planes_consumed = not getattr(eng.evaluators[0], "accepts_leaf_boards", False)
    if write_planes:  # the untimed phases launch the rules op the way the timed region does (a rocprofv3 trace of a run
        for e in eng.engines:  # then averages over one kind of launch)
            e.always_write_planes = True

    def barrier():
        if is_dist:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    lengths = {"desync": [], "warmup": [], "timed": []}
    phase = ["desync"]

    def end_of_ply():
        eng.finish_move()
        tbs = eng.harvest()
        n_games = 0
        for tb in tbs:
            n_games += tb.n_games
            gid = tb.game.cpu().numpy()
            assert gid.size and 0 <= int(gid.min()) and int(gid.max()) < tb.n_games, "corrupt game ids in a harvest"
            lengths[phase[0]].extend(np.bincount(gid, minlength=tb.n_games).tolist())
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

    # ---- warmup steps at the full playout count (untimed)
    phase[0] = "warmup"
    for _ in range(args.warmup):
        eng.run_playouts()
        end_of_ply()

    # ---- timed region
    phase[0] = "timed"
    n_launch = args.steps * args.playouts

    def ev_pair():
        return (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    evs = [[ev_pair() for _ in range(args.groups)] for _ in range(n_launch)]
    tevs = [[(ev_pair(), ev_pair()) for _ in range(args.groups)] for _ in range(n_launch)]
    nevs = [[ev_pair() for _ in range(args.groups)] for _ in range(n_launch)]
    sampler = ClockSampler(local) if rank == 0 else None
    st0 = eng.stats()
    barrier()
    if sampler:
        sampler.start()
    t0 = time.perf_counter()
    games = 0
    k = 0
    step_ms = []
    ev0 = eng.evaluators[0]
    for _ in range(args.steps):
        ts = time.perf_counter()
        for j in range(args.playouts):
            # (the expand / backup launch of every playout but the ply's last also runs the next playout's descent)
            eng.playout_step(events=evs[k], write_planes=write_planes, tree_events=tevs[k], nn_events=nevs[k],
                             more=(j + 1 < args.playouts) and not args.separate_descent)
            k += 1
        games += end_of_ply()  # harvest synchronises with the device (qz_harvest_counts), so this is the ply's wall time
        step_ms.append((time.perf_counter() - ts) * 1e3)
    barrier()
    elapsed = time.perf_counter() - t0
    if sampler:
        sampler.stop()
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
    sel_ms = sum(g[0][0].elapsed_time(g[0][1]) for row in tevs for g in row) / n_evs
    exp_ms = sum(g[1][0].elapsed_time(g[1][1]) for row in tevs for g in row) / n_evs
    planes_written = bool(write_planes or planes_consumed)
    bytes_per_board = BYTES_PER_BOARD if planes_written else 44
    achieved = group_boards * bytes_per_board / (kern_ms * 1e-3) / 1e9
    traffic = None
    t = _load_json(_latest_profile("pmc_traffic.json") or "")
    if t and int(t.get("boards", -1)) == group_boards and bool(t.get("planes", True)) == planes_written:
        traffic = t["traffic_bytes_per_launch"]

    if rank == 0:
        d = {kk: st1[kk] - st0[kk] for kk in ("playouts", "leaf_terminal", "descent_levels", "edges_scanned", "edges_expanded")}
        launches = n_launch * args.groups
        # k_select: every level reads the node's edge records (32 B each); per board 24 B root board in,
        # 37 B leaf record out, 4 B per level of path.  k_expand_backup: per expansion a 560-B prior row in and
        # k 32-B records out (+ 8 B in the parent edge); per level 12 B read + 12 B written; per board mask + value in
        sel_bytes = (d["edges_scanned"] * 32 + d["descent_levels"] * 4 + d["playouts"] * (24 + 37)) / launches
        exp_bytes = (d["edges_expanded"] * 32 + (d["playouts"] - d["leaf_terminal"]) * (560 + 8) + d["descent_levels"] * 24
                     + d["playouts"] * (20 + 4 + 9)) / launches
        tt = _load_json(_latest_profile("pmc_traffic_tree.json") or "") or {}

        def tree_line(kernel, ms, nbytes, note):
            gbs = nbytes / (ms * 1e-3) / 1e9
            return {"kernel": kernel, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                    "traffic": (tt.get(kernel) or {}).get("traffic_bytes_per_launch"), "avg_launch_us": ms * 1e3, "launches": launches,
                    "algorithmic_bytes_per_launch": nbytes, "note": note}

        mean_depth = d["descent_levels"] / max(d["playouts"], 1)
        length_file = args.length_file or _latest_profile("game_length_%dplayouts.json" % args.playouts)
        if args.fix_terminal_sign:
            length_file = args.length_file  # the committed sample is for the reference-faithful sign
        ss = steady_state(plies_all / elapsed, length_file)

        def len_stats(v):
            return {"n": len(v), "mean": float(np.mean(v)) if v else None, "median": float(np.median(v)) if v else None}

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
                "workload": "%s: %d concurrent boards/GPU, n_playout=%d, 9x9, 10 walls/player, "
                            "c_puct=5, temp=1.0, leaf batch=%d, finished tuples all-gathered every ply"
                            % ("BASELINE configs[2] as an engine run" if args.boards == 32768 else
                               ("BASELINE configs[1]" if args.playouts == 100 else "BASELINE configs[3] per GPU"),
                               args.boards, args.playouts, args.boards),
                "boards_per_gpu": args.boards, "board_groups": args.groups, "fix_terminal_sign": bool(args.fix_terminal_sign),
                "n_playout": args.playouts, "bn_mode": args.bn,
                "nn_dtype": args.nn_dtype, "channels_last": bool(args.channels_last),
                "step": "one ply of every board (n_playout playout steps + finish_move + harvest)",
                "desync": "%d untimed plies at %d playouts/move (%.0fs, %d games finished)"
                          % (args.desync_plies, args.desync_playouts, desync_s, len(lengths["desync"])),
            },
            "value_is": "games finished inside the timed region / wall time (Poisson count of %d; the population was desynchronised with "
                        "%d-playout games and is not stationary for %d-playout games) -- see games_per_s_steady_state"
                        % (int(games_all), args.desync_playouts, args.playouts),
            "games_in_timed_region": games_all,
            "games_per_s_steady_state": ss,
            "plies_per_s": plies_all / elapsed,
            "playouts_per_s": playouts_all / elapsed,
            "leaf_evals_per_s": playouts_all / elapsed,
            "terminal_leaf_frac": term_all / max(playouts_all, 1.0),
            "game_lengths_seen": {"timed_region_%d_playouts" % args.playouts: len_stats(lengths["timed"] + lengths["warmup"]),
                                  "desync_phase_%d_playouts" % args.desync_playouts: len_stats(lengths["desync"]),
                                  "note": "games finishing in the timed region started in the desync phase; neither is the length of a "
                                          "%d-playout game" % args.playouts},
            "mean_descent_depth": mean_depth,
            "ms_per_step_series": [round(x, 1) for x in step_ms],
            "roofline": {
                "kernel": ("k_wave_rules (fused Quoridor.actions() + state() of the leaf batch: one wave per board, base paths on nine lanes per player, + encoder groups with streaming stores, one launch)"
                           if (group_boards < 8192 and args.rules_variant in (0, 3, 6)) or args.rules_variant in (3, 6) else
                           ("k_pool_paths_enc + k_pool_masks_enc (Quoridor.actions() + state() of the leaf batch, pooled, two launches)"
                            if args.rules_variant in (0,) or args.rules_variant >= 8 else "k_wave_rules, qz_rules_opts.variant %d" % args.rules_variant)),
                "rules_variant": args.rules_variant,
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "avg_launch_us": kern_ms * 1e3, "launches": n_evs,
                "algorithmic_bytes_per_launch": group_boards * bytes_per_board,
                "planes_written": planes_written, "planes_consumed_by_evaluator": bool(planes_consumed),
                "cache_note": (("%.1f MB written per launch: fits the 256-MiB Infinity Cache, so the stores need not reach HBM before they are "
                                "overwritten; `traffic` is what the PMC counters saw at the memory controllers"
                                if group_boards * bytes_per_board < 256 * 2**20 else
                                "%.1f MB written per launch: larger than the 256-MiB Infinity Cache, the stores reach HBM")
                               % (group_boards * bytes_per_board / 1e6)),
            },
            "roofline_tree": ([
                tree_line("k_select", sel_ms, sel_bytes, "dependent-load latency: duration = the slowest of %d descents (mean depth %.1f, deepest %d levels); "
                          "recorded descents are re-evaluated 64 levels per round (16 records per board, translated across re-roots), "
                          "levels never walked before cost one memory round trip each; bytes = edge records scanned"
                          % (group_boards, mean_depth, st1["max_depth"])),
                tree_line("k_expand_backup", exp_ms, exp_bytes, "one expansion (<= 131 records) + lane-parallel backup per board"),
            ] if args.separate_descent else [
                tree_line("k_expand_backup_select", sel_ms + exp_ms, sel_bytes + exp_bytes,
                          "playout i's expansion + backup and playout i+1's descent of the same board in one launch, same wavefront (the records the "
                          "backup touched are still in the XCD's L2 for the descent; the ply's first descent runs alone and is in the average).  "
                          "Dependent-load latency: duration = the slowest of %d boards (mean depth %.1f, deepest descent %d levels); recorded "
                          "descents are re-evaluated 64 levels per round (16 records per board, translated across re-roots); bytes = edge records "
                          "scanned + written" % (group_boards, mean_depth, st1["max_depth"])),
            ]),
            "roofline_nn": None,
            "engine_stats": {kk: st1[kk] for kk in ("node_overflow", "games_aborted", "aborted_no_move", "aborted_max_plies", "aborted_pool",
                                                    "nonfinite_values", "arena_bytes", "max_nodes", "max_edges", "max_depth", "deep_descents", "deep_descents_cold",
                                                    "deep_levels", "deep_levels_replayed", "tree_pages_total",
                                                    "tree_pages_peak", "traj_pages_total", "traj_pages_peak")},
            "clocks": sampler.summary() if sampler else None,
        }
        nn_ms = sum(a.elapsed_time(b) for row in nevs for a, b in row) / n_evs
        if getattr(ev0, "mfma_trunk", False) and not args.library_trunk:
            # the convolutions' own multiply-adds (what an fp32 kernel would do): 10 trunk layers 64->64, the merged head
            # convolution 64->6, the first layer 26->64; + the three fully connected layers
            conv_flops = 2.0 * group_boards * 81 * 9 * (10 * 64 * 64 + 64 * 6 + 26 * 64)
            fc_flops = 2.0 * group_boards * (324 * 128 + 128 + 162 * 140)
            trunk_mfma = 3.0 * (96.0 / 81.0) * 2.0 * group_boards * 81 * 9 * (10 * 64 * 64) + 3.0 * (96.0 / 81.0) * 2.0 * group_boards * 81 * 9 * 64 * 32
            out["roofline_nn"] = {
                "kernel": "k_trunk<true> + k_head_fc (the whole leaf evaluation in two launches: first layer from the packed boards, ten "
                          "conv3x3 64->64 layers as implicit GEMMs on v_mfma_f32_32x32x16_f16 with split fp16 operands, per-leaf "
                          "normalisation / residual / ReLU, merged head convolution -- activations never leave the CU -- then fc1/fc2/tanh, "
                          "fc3/softmax)",
                "bound": "mfma", "achieved": (conv_flops + fc_flops) / (nn_ms * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                "frac": (conv_flops + fc_flops) / (nn_ms * 1e-3) / 1e12 / 2500.0, "traffic": None,
                "avg_launch_us": nn_ms * 1e3, "launches": n_evs, "algorithmic_flops_per_launch": conv_flops + fc_flops,
                "executed_mfma_tflops": trunk_mfma / (nn_ms * 1e-3) / 1e12,
                "note": "achieved = fp32-equivalent FLOPs of the network / time of both launches (HIP events on the launch stream, every "
                        "playout step); peak = dense fp16 MFMA at the 2.4 GHz boost clock (MI355X_MICROARCH.md).  fp32 accuracy costs three "
                        "fp16 MFMAs per product (hi*hi + hi*lo + lo*hi) and 81 of 96 tile rows are live, so the matrix pipe executes 3.56x "
                        "the algorithmic FLOPs of the convolutions (executed_mfma_tflops); the f32-input MFMA peak this replaces is 157 "
                        "TFLOP/s.  Inside this kernel the chip runs at ~1.7 GHz (power), profiles/round2/trunk_stamps.txt",
            }
        else:
            out["roofline_nn"] = {"kernel": "evaluator on MIOpen fp32 convolutions (--library-trunk)", "bound": "mfma", "avg_launch_us": nn_ms * 1e3,
                                  "launches": n_evs, "achieved": None, "peak": None, "unit": "TFLOP/s", "frac": None, "traffic": None}
        if st1["games_aborted"]:
            sys.stderr.write("bench.py: WARNING %d games were dropped (no_move %d, max_plies %d, pool %d)\n"
                             % (st1["games_aborted"], st1["aborted_no_move"], st1["aborted_max_plies"], st1["aborted_pool"]))
        if args.clock_log and sampler:
            with open(args.clock_log, "w") as f:
                json.dump({"step_ms": step_ms, "samples": sampler.samples}, f)
        if world == 1 and not args.no_c3:
            out["roofline_c3"] = c3_microbench(dev)
        if not args.no_cpu_baseline and world == 1:
            eng.close()
            torch.cuda.empty_cache()
            if ss:
                L_cpu, L_src = ss["mean_plies_per_game"], ss["length_source"]
            else:  # no length sample for this playout count: fall back to the games seen in this run, and say so
                seen = lengths["timed"] + lengths["warmup"] + lengths["desync"]
                L_cpu = float(np.mean(seen)) if seen else None
                L_src = "NO length sample for n_playout=%d: mean length of the %d games finished in this run (mostly desync games)" % (args.playouts, len(seen))
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds, args.playouts, L_cpu, L_src)
        print(json.dumps(out))
    eng.close()
    if is_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
