import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from alphazero_quoridor_amd import _cabi
from alphazero_quoridor_amd.engine import SelfPlayEngine
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet
dev = torch.device("cuda:0"); torch.manual_seed(2026); torch.backends.cudnn.benchmark = True
net = PolicyValueNet(use_gpu=True); ev = net.evaluator("per_leaf")
eng = SelfPlayEngine(4096, n_playout=400, seed=1, device=dev)
for _ in range(300):
    eng.run_playouts(ev, 4); eng.finish_move(); eng.harvest()
import ctypes as C
L = _cabi.load()
for variant in (0, 6, 5, 0, 6):  # k_wave_rules: default (1 board per wavefront, base paths on nine lanes per player), 6 = the same with streaming stores, 5 = one search per lane
    L.qz_engine_set_rules_opts(eng.h, C.byref(_cabi.qz_rules_opts(variant, 0, 0, 0)))
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(200)]
    for i in range(200): eng.playout_step(ev, events=evs[i], write_planes=True)  # like bench.py: the full op, planes included
    torch.cuda.synchronize()
    print("variant", variant, "avg rules-op us", sum(a.elapsed_time(b) for a, b in evs) / 200 * 1e3)
