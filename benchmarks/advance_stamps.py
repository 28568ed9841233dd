"""Where does a wavefront's time go in k_advance?  Diagnostic build with s_memtime stamps (tests/hip/libqzero_hip_astamps.so):
cycles per phase per board, accumulated over the launches of the late-game regime of a run from the opening."""
import ctypes as C, json, os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from alphazero_quoridor_amd import _cabi
_cabi.LIB_PATH = os.path.join(ROOT, "tests", "hip", "libqzero_hip_astamps.so")  # before anything loads the library
from alphazero_quoridor_amd.engine import SelfPlayEngine
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet
B = int(os.environ.get("BOARDS", 4096)); NP = int(os.environ.get("PLAYOUTS", 400)); MP = int(os.environ.get("MAXP", 64)); BUD = int(os.environ.get("BUDGET", 0))
WARM = int(os.environ.get("WARM_ROUNDS", 12800)); MEAS = int(os.environ.get("MEAS_ROUNDS", 1280))
dev = torch.device("cuda:0"); torch.manual_seed(0)
ev = PolicyValueNet(use_gpu=True).evaluator("per_leaf")
eng = SelfPlayEngine(B, n_playout=NP, seed=77, device=dev)
L = _cabi.load()
buf = np.zeros((4096, 16), dtype=np.uint64)
for i in range(WARM // 64):
    eng.run_rounds(ev, 64, max_playouts=MP, budget_us=BUD); eng.harvest()
torch.cuda.synchronize()
assert L.qzt_advance_stamps_read(buf.ctypes.data_as(C.c_void_p), 1) == 0
buf2 = np.zeros((4096, 4), dtype=np.uint64)
assert L.qzt_advance_stamps2_read(buf2.ctypes.data_as(C.c_void_p), 1) == 0
buf3 = np.zeros((4096, 4), dtype=np.uint64)
assert L.qzt_advance_stamps3_read(buf3.ctypes.data_as(C.c_void_p), 1) == 0
st0 = eng.stats(); t0 = time.time()
for i in range(MEAS // 64):
    eng.run_rounds(ev, 64, max_playouts=MP, budget_us=BUD); eng.harvest()
torch.cuda.synchronize(); dt = time.time() - t0
assert L.qzt_advance_stamps_read(buf.ctypes.data_as(C.c_void_p), 0) == 0
assert L.qzt_advance_stamps2_read(buf2.ctypes.data_as(C.c_void_p), 0) == 0
assert L.qzt_advance_stamps3_read(buf3.ctypes.data_as(C.c_void_p), 0) == 0
st1 = eng.stats()
# (the accumulators are unsigned 64-bit sums of stamp differences: read as two's complement, so that a board whose few
# differences came out slightly negative counts as ~0 and not as 1.8e19)
a = buf[:min(B, 4096)].view(np.int64).astype(np.float64)
tot = a[:, 10].sum()
names = ["prologue", "expand", "backup", "move", "descent", "probe_hit", "probe_miss", "epilogue"]
po = a[:, 8].sum()
out = {"boards": B, "rounds": MEAS, "wall_s": dt, "playouts": int(st1["playouts"] - st0["playouts"]), "playouts_per_s": (st1["playouts"] - st0["playouts"]) / dt,
       "stamped_playouts": int(po), "mean_depth": a[:, 9].sum() / max(po, 1), "launches_per_board": a[:, 11].mean(),
       "cycles_per_launch_per_wave": tot / a[:, 11].sum(), "cycles_per_playout": tot / max(po, 1),
       "share": {n: a[:, k].sum() / tot for k, n in enumerate(names)},
       "cycles_per_playout_by_phase": {n: a[:, k].sum() / max(po, 1) for k, n in enumerate(names)}}
lv = float(a[:, 9].sum()); r2 = buf2[:min(B, 4096)].view(np.int64).astype(np.float64).sum(axis=0)
out["levels_replayed_frac"] = r2[0] / max(lv, 1); out["replay_rounds_per_playout"] = r2[1] / max(po, 1); out["replay_rounds_failed_frac"] = r2[2] / max(r2[1], 1)
out["levels_per_replay_round"] = r2[0] / max(r2[1], 1)
r3 = buf3[:min(B, 4096)].view(np.int64).astype(np.float64).sum(axis=0)
out["descent_cycles_per_playout"] = {"replay_rounds": r3[0] / max(po, 1), "walk_and_setup": r3[1] / max(po, 1), "record_commit": r3[2] / max(po, 1)}
out["cycles_per_replay_round"] = r3[0] / max(r2[1], 1); out["cycles_per_walked_level"] = r3[1] / max(lv - r2[0], 1)
out["levels_selected_by_a_replay_lane_per_playout"] = r3[3] / max(po, 1); out["levels_not_confirmed_per_playout"] = (lv - r2[0]) / max(po, 1)
q = [0.5, 0.9, 0.99, 1.0]
out["longest_single_descent_cycles_quantiles_over_boards_50_90_99_100"] = np.quantile(a[:, 13], q).tolist()
out["longest_launch_cycles_quantiles"] = np.quantile(a[:, 14], q).tolist()
# launches (board x round) that ran 15 % over their budget: the launch ends with its last wave
n_over = a[:, 12].sum()
out["overrunning_launches"] = {"per_round": n_over / MEAS, "frac_of_board_launches": n_over / a[:, 11].sum(), "mean_cycles": a[:, 15].sum() / max(n_over, 1),
                               "mean_prologue_cycles": r2[3] / max(n_over, 1),
                               "boards_with_any": int((a[:, 12] > 0).sum())}
print(json.dumps(out, indent=1))
