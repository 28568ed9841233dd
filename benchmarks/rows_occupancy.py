"""How many wavefronts of k_rows does the chip hold at once?  Engines of late-game boards only (nobody has a wall: every board is
k_rows'), budget 3,000 us: a launch that lasts ~2 x the budget did not fit."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from alphazero_quoridor_amd.boards import DeviceBoards
from alphazero_quoridor_amd.engine import SelfPlayEngine
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet
from synth import synth_positions
dev = torch.device("cuda:0"); torch.manual_seed(0)
ev = PolicyValueNet(use_gpu=True).evaluator("per_leaf")
for B in [int(x) for x in os.environ.get("BOARDS", "12288,14336,16384,20480").split(",")]:
    b = synth_positions(B, seed=5, max_walls=14); b["w1"] = 0; b["w2"] = 0
    eng = SelfPlayEngine(B, n_playout=400, seed=77, device=dev, select_opts=int(os.environ.get("SELECT_OPTS", 40)), max_depth=992)
    eng.set_boards(DeviceBoards.from_packed(b, dev), reset_trees=True)
    for i in range(4):
        eng.run_rounds(ev, 64, max_playouts=4096, budget_us=3000); eng.harvest()
    torch.cuda.synchronize(); st0 = eng.stats(); t0 = time.time()
    for i in range(2):
        eng.run_rounds(ev, 64, max_playouts=4096, budget_us=3000); eng.harvest()
    torch.cuda.synchronize(); dt = time.time() - t0; st1 = eng.stats()
    print(json.dumps({"boards": B, "ms_per_round": 1e3 * dt / 128, "playouts_per_s": (st1["playouts"] - st0["playouts"]) / dt, "evals_per_round": (st1["nn_evals"] - st0["nn_evals"]) / 128}), flush=True)
    eng.close(); del eng; torch.cuda.empty_cache()
