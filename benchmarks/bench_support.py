"""Support pieces of bench.py that are not about the hot path: the clock / power sampler of the timed region, the CPU baseline leg
(oracle port on the host's cores) and the C3 microbenchmark of the rules op.  bench.py imports them."""
from __future__ import annotations

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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


def file_sha256(path):
    """sha256 of a committed file a bench line rests on (length samples, PMC traffic files): the line names what it read."""
    import hashlib

    try:
        with open(path, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()
    except (OSError, TypeError):
        return None


def tree_sha():
    """the git commit of this tree (None on a box without .git: gpurun snapshots do not carry it) + a hash of the kernel sources,
    which travels everywhere: a profile file recorded with the same `kernel_sources_sha256` was measured on these kernels."""
    import hashlib

    try:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
    except (OSError, subprocess.SubprocessError):
        commit = None
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "alphazero_quoridor_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return {"git_commit": commit, "kernel_sources_sha256": h.hexdigest()}


# ------------------------------------------------------------------------------ clocks / power
class ClockSampler(threading.Thread):
    """Samples shader clock, power and temperature while the timed region runs (sysfs when readable, else `rocm-smi --json`
    as a child process)."""

    def __init__(self, index=0, period=1.0):
        super().__init__(daemon=True)
        self.index, self.period, self.samples, self.source = index, period, [], None
        self._stop_ev = threading.Event()
        self._sysfs = self._find_card(index)

    @staticmethod
    def _find_card(index):
        """/sys/class/drm/cardN/device of HIP device `index`, matched by PCI address."""
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
        d, out = self._sysfs, {}
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
        r = subprocess.run(["rocm-smi", "-d", str(self.index), "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        out = {}
        for k, v in next(iter(json.loads(r.stdout).values())).items():
            m = re.search(r"([\d.]+)", str(v))
            kl = k.lower()
            if m and kl.startswith("sclk"):
                out["sclk_mhz"] = float(m.group(1))
            elif m and "power" in kl and "power_w" not in out:
                out["power_w"] = float(m.group(1))
            elif m and "temperature" in kl and ("junction" in kl or "temp_c" not in out):
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
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            elif int(txt[0]) > 0:
                n = min(n, max(1, int(txt[0]) // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


STEADY_ROOTS = os.path.join(ROOT, "tests", "golden", "steady_state_roots.npz")


def cpu_baseline(seconds, n_playout, mean_plies_per_game, length_source, open_share=None):
    """The oracle port on this host: 1 process, then one process per usable core side by side (python -m oracle.cpu_baseline:
    one torch thread each), ON THE WORKLOAD THE GPU NUMBER IS QUOTED ON: the searches of a steady-state population's root
    positions (tests/golden/steady_state_roots.npz: roots harvested from a sustained run of this engine), split by phase -- roots
    whose mover still has walls (`open`: 100+ legal moves, a double BFS per candidate wall: the slow class of the reference's
    actions(), quoridor.py:138-157) and the others (`late`) -- and weighted by the share of its board time the GPU run spent in
    each phase (`open_share` = the line's open_phase.share_of_board_time).  The first ply from the opening (131 legal moves: the
    figure of rounds 1-5) stays in by_phase as `opening_ply`.  playouts/s is the measured quantity; games/s divides it by
    n_playout and by the SAME plies per game the GPU's steady-state estimate uses."""
    cores = usable_cores()
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    have_roots = os.path.exists(STEADY_ROOTS)
    phases = ["opening"] + (["open", "late"] if have_roots else [])
    per = max(seconds / (2.0 * len(phases)), 2.0)

    def run_many(k, phase):
        cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--seconds", "%.1f" % per, "--n-playout", str(n_playout), "--phase", phase]
        procs = [subprocess.Popen(cmd + ["--seed", str(i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env, cwd=ROOT) for i in range(k)]
        res = []
        for p in procs:
            lines = [ln for ln in p.communicate()[0].splitlines() if ln.startswith("{")]
            if p.returncode == 0 and lines:
                res.append(json.loads(lines[-1]))
        return res

    by_phase = {}
    for ph in phases:
        one, many = run_many(1, ph), run_many(cores, ph)
        if not one or not many:
            return {"value": None, "unit": "games/s", "cores": cores, "kind": "port", "sample": "oracle.cpu_baseline --phase %s failed to run" % ph}
        by_phase["opening_ply" if ph == "opening" else "steady_" + ph] = {
            "playouts_per_s_1core": one[0]["playouts"] / one[0]["seconds"], "playouts_per_s_allcores": sum(r["playouts"] / r["seconds"] for r in many),
            "processes": len(many), "playouts_timed": int(sum(r["playouts"] for r in many)), "positions": many[0].get("positions"),
            "mean_legal_moves_at_the_roots": many[0].get("mean_root_moves"), "seconds_per_process": per}
    n_proc = by_phase["opening_ply"]["processes"]
    if have_roots and open_share is not None:
        w_open = min(max(float(open_share), 0.0), 1.0)
        pps1 = w_open * by_phase["steady_open"]["playouts_per_s_1core"] + (1.0 - w_open) * by_phase["steady_late"]["playouts_per_s_1core"]
        ppsN = w_open * by_phase["steady_open"]["playouts_per_s_allcores"] + (1.0 - w_open) * by_phase["steady_late"]["playouts_per_s_allcores"]
        basis = ("steady-state roots (tests/golden/steady_state_roots.npz, sha256 %s...): %.3f x the open-phase rate + %.3f x the late-phase rate -- the shares of its "
                 "board time the GPU run spent in each phase (open_phase.share_of_board_time)" % ((file_sha256(STEADY_ROOTS) or "?")[:12], w_open, 1.0 - w_open))
    else:
        pps1, ppsN = by_phase["opening_ply"]["playouts_per_s_1core"], by_phase["opening_ply"]["playouts_per_s_allcores"]
        basis = "the first ply from the opening only (no steady-state root sample / no phase split in this run)"
    L = mean_plies_per_game
    out = {
        "value": (ppsN / n_playout / L) if L else None, "unit": "games/s", "cores": n_proc, "host_hardware_threads": os.cpu_count(), "kind": "port",
        "sample": "oracle C port + batch-1 fp32 torch-CPU forward per leaf, 1 torch thread per process, %d-playout searches from fresh trees: %.0f s on 1 core, then %.0f s on "
                  "%d cores as independent processes, per phase (%s); weighted: %s; games/s = playouts/s / %d / %s plies per game (%s)"
                  % (n_playout, per, per, n_proc, ", ".join(by_phase), basis, n_playout, "%.0f" % L if L else "?", length_source),
        "compare_on": "playouts_per_s (length-independent)", "playouts_per_s_1core": pps1, "playouts_per_s_allcores": ppsN,
        "games_per_s_1core": (pps1 / n_playout / L) if L else None,
    }
    # (flat copies of the per-phase rates: a record that keeps an object's scalars and drops its nested objects still shows them)
    for ph, v in by_phase.items():
        out["playouts_per_s_allcores_" + ph] = v["playouts_per_s_allcores"]
    out.update({"open_share_used": open_share if have_roots else None, "by_phase": by_phase})
    cal = _load_json(_latest_profile("cpu_calibration.json") or "")
    if cal and cal.get("port_over_reference"):
        r_open = float(cal["port_over_reference"])
        r_late = float(cal.get("port_over_reference_late") or r_open)
        est = {"port_over_reference_opening": r_open, "port_over_reference_late": r_late,
               "calibration": "pure-Python reference vs this C port on the build container's host (%s): benchmarks/calibrate_cpu_port.py; the opening ratio is applied to "
                              "the open phase, the late-game ratio to the late phase" % cal.get("host_cpu", "?")}
        if have_roots and open_share is not None:
            w = min(max(float(open_share), 0.0), 1.0)
            est["playouts_per_s_1core"] = w * by_phase["steady_open"]["playouts_per_s_1core"] / r_open + (1.0 - w) * by_phase["steady_late"]["playouts_per_s_1core"] / r_late
            est["playouts_per_s_allcores"] = w * by_phase["steady_open"]["playouts_per_s_allcores"] / r_open + (1.0 - w) * by_phase["steady_late"]["playouts_per_s_allcores"] / r_late
        else:
            est["playouts_per_s_1core"], est["playouts_per_s_allcores"] = pps1 / r_open, ppsN / r_open
        out["reference_playouts_per_s_allcores"] = est["playouts_per_s_allcores"]
        out["reference_estimate"] = est
    return out


def c3_microbench(dev, launches=60):
    """BASELINE configs[2] / SURVEY C3: the fused actions() + state() op on 32,768 mid-game boards (random legal play from the
    opening, mover has a wall left), outside the timed region: the pooled two-launch pipeline."""
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
    tf = _latest_profile("pmc_traffic_c3.json")
    t = _load_json(tf or "")
    return {"workload": "BASELINE configs[2] microbenchmark: 32,768 boards (S-mid: 0..20 plies of random legal play, mover has a wall), actions() + state(), "
                        "inputs resident in HBM, NOT part of the timed region",
            "kernel": "k_pool_paths_enc + k_pool_masks_enc (pooled pipeline, two launches, 184-byte hand-off record per board)",
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": t.get("traffic_bytes_per_launch") if t else None, "traffic_source": ("profiles: " + os.path.relpath(tf, ROOT)) if t else None,
            "traffic_over_algorithmic": (t["traffic_bytes_per_launch"] / (n * BYTES_PER_BOARD)) if t else None,
            "avg_launch_us": us, "launches": launches, "algorithmic_bytes_per_launch": n * BYTES_PER_BOARD,
            "cache_note": "277 MB written per launch: larger than the 256-MiB Infinity Cache, the stores reach HBM"}


