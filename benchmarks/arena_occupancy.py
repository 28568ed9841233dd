"""Peak tree sizes (nodes / edges) over a long bench-like run with generous arenas: sizes the default
node_cap / edge_cap of qz_engine_create so that no expansion is ever skipped (qz_stats.node_overflow == 0)."""
import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from alphazero_quoridor_amd.engine import SelfPlayEngine
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet
n_playout = int(sys.argv[1]) if len(sys.argv) > 1 else 400
plies = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0"); torch.manual_seed(2026); torch.backends.cudnn.benchmark = True
net = PolicyValueNet(use_gpu=True); ev = net.evaluator("per_leaf")
eng = SelfPlayEngine(2048, n_playout=n_playout, seed=1, device=dev, tree_pool_pages=2048 * 128)
for _ in range(700):
    eng.run_playouts(ev, 4); eng.finish_move(); eng.harvest()
pk_n = pk_e = 0
for ply in range(plies):
    eng.run_playouts(ev)
    st = eng.stats()
    pk_n, pk_e = max(pk_n, st["max_nodes"]), max(pk_e, st["max_edges"])
    eng.finish_move(); eng.harvest()
    if ply % 10 == 9:
        print(json.dumps({"n_playout": n_playout, "plies": ply + 1, "peak_nodes": pk_n, "peak_edges": pk_e,
                          "peak_nodes_per_playout": pk_n / n_playout, "overflow": st["node_overflow"],
                          "tree_pages_peak_per_board": st["tree_pages_peak"] / 2048, "traj_pages_in_use_per_board": st["traj_pages_in_use"] / 2048}), flush=True)
