"""Times qz_nn_input_layer on the leaf boards of a bench-like engine (4,096 boards)."""
import os, sys, torch, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from alphazero_quoridor_amd.engine import SelfPlayEngine
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet
dev = torch.device("cuda:0"); torch.manual_seed(2026); torch.backends.cudnn.benchmark = True
net = PolicyValueNet(use_gpu=True); ev = net.evaluator("per_leaf")
eng = SelfPlayEngine(4096, n_playout=400, seed=1, device=dev)
for _ in range(300):
    eng.run_playouts(ev, 4); eng.finish_move(); eng.harvest()
eng.select()
leaf = eng.leaf_ref()
for _ in range(5): ev._first_layer_from_boards(leaf)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): ev._first_layer_from_boards(leaf)
b.record(); torch.cuda.synchronize()
print("qz_nn_input_layer (incl. output allocation): %.1f us" % (a.elapsed_time(b) / 100 * 1e3))
