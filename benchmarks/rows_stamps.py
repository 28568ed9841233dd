"""Where does a wavefront's time go in k_rows?  Diagnostic build with s_memtime stamps (tests/hip/libqzero_hip_rowstamps.so): cycles per
code section, summed over the wavefronts and launches of a run on late-game boards (nobody has a wall left), real network."""
import ctypes as C, json, os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from alphazero_quoridor_amd import _cabi
_cabi.LIB_PATH = os.path.join(ROOT, "tests", "hip", "libqzero_hip_rowstamps.so")
from alphazero_quoridor_amd.boards import DeviceBoards
from alphazero_quoridor_amd.engine import SelfPlayEngine
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet
from synth import synth_positions
B = int(os.environ.get("BOARDS", 13312)); NP = 400; BUD = int(os.environ.get("BUDGET", 3000))
WARM = int(os.environ.get("WARM_ROUNDS", 1500)); MEAS = int(os.environ.get("MEAS_ROUNDS", 256))
dev = torch.device("cuda:0"); torch.manual_seed(0)
ev = PolicyValueNet(use_gpu=True).evaluator("per_leaf")
b = synth_positions(B, seed=5, max_walls=14); b["w1"] = 0; b["w2"] = 0
eng = SelfPlayEngine(B, n_playout=NP, seed=77, device=dev, select_opts=40, max_depth=992)
eng.set_boards(DeviceBoards.from_packed(b, dev), reset_trees=True)
L = _cabi.load()
buf = np.zeros(24, dtype=np.uint64)
for i in range(WARM // 64):
    eng.run_rounds(ev, 64, max_playouts=4096, budget_us=BUD); eng.harvest()
torch.cuda.synchronize()
assert L.qzt_rows_stamps_read(buf.ctypes.data_as(C.c_void_p), 1) == 0
st0 = eng.stats(); t0 = time.time()
for i in range(MEAS // 64):
    eng.run_rounds(ev, 64, max_playouts=4096, budget_us=BUD); eng.harvest()
torch.cuda.synchronize(); dt = time.time() - t0
assert L.qzt_rows_stamps_read(buf.ctypes.data_as(C.c_void_p), 0) == 0
st1 = eng.stats(); d = {k: st1[k] - st0[k] for k in st1}
a = buf.astype(np.float64)
names = ["leaf handling (expand / backup / miss) + prologue", "loop head + descent set-up", "replay rounds", "walked levels", "leaf board + probe issue", "record commit",
         "counters + terminal backups", "probe wait + compare"]
tot = a[:8].sum() + a[13:17].sum(); waves = a[12]; po = d["playouts"]
names = names + [None] * 5 + ["hit: payload", "hit: expansion", "hit: note", "hit: backup"]
print(json.dumps({"boards": B, "rounds": MEAS, "playouts_per_s": po / dt, "mean_depth": d["descent_levels"] / max(po, 1), "hit_rate": d["memo_hits"] / max(po, 1),
                  "wave_launches": waves, "cycles_per_wave_launch": tot / max(waves, 1), "playouts_per_wave_launch": po / max(waves, 1),
                  "cycles_per_playout_of_a_row (wave cycles x 4 rows / playouts)": 4 * tot / max(po, 1), "wave_cycles_per_wave_iteration": tot / max(a[10], 1),
                  "replay_round_sections_per_wave_iteration": a[8] / max(a[10], 1), "walk_sections_per_wave_iteration": a[9] / max(a[10], 1),
                  "share": {n: a[k] / tot for k, n in enumerate(names) if n}, "cycles_per_replay_round_section": a[2] / max(a[8], 1), "cycles_per_walk_section": a[3] / max(a[9], 1)}, indent=1))
