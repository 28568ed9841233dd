"""GPU tier, LAST file in collection order (the name sorts after every kernel-parity file): bench.py as a subprocess -- the N > 1
path on one GPU, the self-launch, the contract keys of the N = 1 line.  These tests assert on the FORMAT of the bench line;
round 4's driver run died on one of them (test 11 of 106 under `-x`) before a single kernel-parity file had been collected, so
they live here: a formatting regression can no longer hide kernel parity."""
import pytest

pytestmark = pytest.mark.gpu


def test_bench_multi_rank_path_on_one_gpu(gpu_device):
    """bench.py's N>1 path (one engine per rank, per-ply all-gather of finished tuples, MAX /
    SUM reductions, one JSON line from rank 0) with two ranks sharing this GPU over gloo
    (RCCL refuses two ranks on one device; the driver's real runs use nccl)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, QZ_DIST_BACKEND="gloo", QZ_SHARE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
         "--boards", "256", "--playouts", "16", "--desync-plies", "200", "--rounds-per-step", "48", "--no-cpu-baseline"],
        env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2 and d["config"]["mode"] == "async"
    assert d["playouts_per_s"] > 0 and d["roofline"]["achieved"] > 0 and d["rounds"] == 2 * 48
    assert d["roofline"]["launches_timed"] == 2  # --rounds-per-step 48 below the default --event-every 64: one timed round per step (round 4: none, and no line)
    ag = d["allgather_ms"]  # the exchange is timed
    assert ag["calls"] >= 2 and 0 < ag["mean"] <= ag["max"] and ag["unit"] == "ms"
    # the line shows what EACH rank did -- rank, device, PCI bus id, plies, playouts, ms per step -- and the
    # ranks' shares add up to the aggregate (on RCCL the bus ids must differ: bench.py asserts it; here two gloo ranks share the GPU)
    pr = d["per_rank"]
    assert len(pr) == 2 and sorted(r["rank"] for r in pr) == [0, 1] and all(r["plies"] > 0 and r["playouts"] > 0 and r["ms_per_step"] > 0 for r in pr)
    assert sum(r["plies"] for r in pr) == pytest.approx(d["plies_per_s"] * d["ms_per_step"] * 2 / 1e3, rel=1e-6)
    assert sum(r["playouts"] for r in pr) == pytest.approx(d["playouts_per_s"] * d["ms_per_step"] * 2 / 1e3, rel=1e-6)
    assert all(":" in r["pci_bus_id"] for r in pr) and d["provenance"]["kernel_sources_sha256"]
    # both ranks' work is in the aggregate: every round advances every board of both ranks, a move needs 16 playouts
    plies = d["plies_per_s"] * d["ms_per_step"] * 2 / 1e3
    assert 2 * 256 * 2 <= plies and d["playouts_per_s"] / d["plies_per_s"] == pytest.approx(16, rel=0.2)
    assert d["engine_stats"]["node_overflow"] == 0 and d["engine_stats"]["runaway_descents"] == 0


def test_bench_launches_its_own_ranks(gpu_device):
    """`python bench.py --gpus 2` with NO launcher around it (the driver's command shape): the
    process spawns its two ranks itself and relays rank 0's single JSON line with n_gpus = 2.
    Same gloo / shared-device hooks as above (one GPU on this box)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(QZ_DIST_BACKEND="gloo", QZ_SHARE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--boards", "256",
                        "--playouts", "16", "--desync-plies", "100", "--no-cpu-baseline", "--mode", "lockstep"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]   # stdout is the JSON line and nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2 and d["value"] >= 0
    assert abs(d["plies_per_s"] * d["ms_per_step"] * 2 / 1e3 - 2 * 256 * 2) < 2 * 256 * 2 * 0.05


def test_bench_single_gpu_line_carries_the_contract(gpu_device):
    """`python bench.py` (N=1, tiny workload): ONE JSON line with the driver's contract keys, the
    roofline of the rules op in the timed region, the 32,768-board microbenchmark (roofline_c3) and
    the CPU baseline; no expansion skipped."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def run(*extra):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--boards", "256",
                            "--playouts", "16", "--desync-plies", "100", "--cpu-seconds", "2"] + list(extra),
                           capture_output=True, text=True, timeout=900, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        return json.loads(lines[0])

    # the default route: the asynchronous self-play loop
    a = run("--rounds-per-step", "64", "--second-line-boards", "256", "--second-line-warm-seconds", "3", "--second-line-seconds", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in a, k
    assert a["config"]["mode"] == "async" and a["n_gpus"] == 1 and a["steps"] == 2 and a["vs_baseline"] is None and a["unit"] == "games/s"
    assert "workload" in a["config"] and "model" not in a["config"] and a["rounds"] == 2 * 64
    for rf in (a["roofline"], a["roofline_rules"], a["roofline_c3"]):
        assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    assert a["roofline_nn"]["bound"] == "mfma" and a["roofline_nn"]["avg_launch_us"] > 0 and "k_advance" in a["roofline"]["kernel"]
    assert 0 <= a["memo_hit_rate"] <= 1 and a["nn_evaluations_per_s"] > 0 and a["playouts_per_s"] > a["nn_evaluations_per_s"]
    assert a["engine_stats"]["node_overflow"] == 0 and a["engine_stats"]["runaway_descents"] == 0
    cb = a["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0
    # the CPU leg runs the workload the GPU number is quoted on -- steady-state roots by phase, weighted by the
    # line's own open-phase share of board time -- next to the opening ply of rounds 1-5
    assert set(cb["by_phase"]) == {"opening_ply", "steady_open", "steady_late"} and all(v["playouts_per_s_allcores"] > 0 for v in cb["by_phase"].values())
    assert cb["open_share_used"] == pytest.approx(a["open_phase"]["share_of_board_time"]) and cb["by_phase"]["steady_open"]["mean_legal_moves_at_the_roots"] > 20 > cb["by_phase"]["steady_late"]["mean_legal_moves_at_the_roots"]
    w = cb["open_share_used"]
    assert cb["playouts_per_s_allcores"] == pytest.approx(w * cb["by_phase"]["steady_open"]["playouts_per_s_allcores"] + (1 - w) * cb["by_phase"]["steady_late"]["playouts_per_s_allcores"])
    # flat copies of the per-phase rates (a record that keeps an object's scalars and drops its nested objects still shows them), and
    # the line's scalars once more as its LAST key (the line is ~18 KB; a record that keeps head and tail drops plies_per_s in the middle)
    for ph, v in cb["by_phase"].items():
        assert cb["playouts_per_s_allcores_" + ph] == v["playouts_per_s_allcores"]
    assert cb["reference_playouts_per_s_allcores"] == cb["reference_estimate"]["playouts_per_s_allcores"]
    assert list(a)[-1] == "summary"
    sm = a["summary"]
    assert sm["plies_per_s"] == a["plies_per_s"] and sm["playouts_per_s"] == a["playouts_per_s"] and sm["roofline_frac"] == a["roofline"]["frac"]
    assert sm["roofline_c3_frac"] == a["roofline_c3"]["frac"] and sm["cpu_port_playouts_per_s_allcores"] == cb["playouts_per_s_allcores"]
    assert sm["gpu_over_cpu_port_playouts"] == pytest.approx(a["playouts_per_s"] / cb["playouts_per_s_allcores"]) and all(not isinstance(v, (dict, list)) for v in sm.values())
    assert len(a["per_rank"]) == 1 and a["per_rank"][0]["rank"] == 0 and a["provenance"]["kernel_sources_sha256"]
    assert "games_per_s_steady_state" in a and "games_in_timed_region" in a and len(a["ms_per_step_series"]) == 2
    # the headline names its tracked scalar; the second line's `value` is the stationary estimate from the
    # committed sign-fixed length sample with the raw count beside it; the throughput-precision network has a labelled line of its own
    assert a["tracked_scalar"] == "plies_per_s" and a["plies_per_s"] > 0
    for key, parity in (("second_line_fix_terminal_sign", True), ("second_line_NON_PARITY_fp16", False)):
        s2 = a[key]
        # (this run plays 16 playouts per move: no committed length sample for that count, so `value` falls back to the raw count and
        # says so; with the default 400 playouts it is the stationary estimate -- the estimator itself: tests/test_host_logic.py)
        assert s2["unit"] == "games/s" and s2["value_transient"] >= 0 and s2["value"] == s2["value_transient"] and "NO committed" in s2["value_is"]
        assert s2["games_per_s_steady_state"] is None and ("NON_PARITY" in s2["label"]) == (not parity) and "NOT the headline" in s2["label"]
        assert s2["boards"] == 256 and s2["plies_per_s"] > 0 and s2["nn_evaluations_per_s"] > 0
    # with the rounds captured in HIP graphs the step's first round is still issued piece by piece: the line keeps its roofline
    g = run("--rounds-per-step", "32", "--graph-rounds", "8", "--no-c3", "--no-cpu-baseline", "--second-line-seconds", "0")
    assert g["config"]["graph_rounds"] == 8 and g["roofline"]["launches_timed"] == 2 and 0 < g["roofline"]["frac"] < 1 and g["rounds"] == 2 * 32
    # round 2's route, kept for A/B
    d = run("--mode", "lockstep")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "games/s" and "workload" in d["config"] and "model" not in d["config"]
    for rf in (d["roofline"], d["roofline_c3"]):
        assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0 < rf["frac"] < 1
    assert d["roofline_c3"]["algorithmic_bytes_per_launch"] == 32768 * 8468
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert cb["playouts_per_s_allcores"] >= 0.5 * cb["playouts_per_s_1core"] > 0 and cb["compare_on"].startswith("playouts_per_s")
    assert d["engine_stats"]["node_overflow"] == 0
    assert [t["kernel"] for t in d["roofline_tree"]] == ["k_expand_backup_select"] and all(t["achieved"] > 0 for t in d["roofline_tree"])
    assert "games_per_s_steady_state" in d and "game_lengths_seen" in d and len(d["ms_per_step_series"]) == 2
    assert d["roofline"]["planes_written"] is True and d["roofline"]["planes_consumed_by_evaluator"] is False
