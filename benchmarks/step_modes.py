"""Sustained ply time (400 playout steps) at B=4096 in three host modes: eager, eager with the
event pair bench.py records around the rules kernel, HIP-graph replay (8 steps per graph)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alphazero_quoridor_amd.engine import SelfPlayEngine
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

dev = torch.device("cuda:0")
torch.manual_seed(2026)
torch.backends.cudnn.benchmark = True
net = PolicyValueNet(use_gpu=True)
ev = net.evaluator("per_leaf")
eng = SelfPlayEngine(4096, n_playout=400, seed=1, device=dev)
for _ in range(300):  # desync like bench.py
    eng.run_playouts(ev, 4); eng.finish_move(); eng.harvest()
def ply(mode):
    torch.cuda.synchronize(); t = time.perf_counter()
    if mode == "graph":
        eng.run_playouts(ev, 400)
    elif mode == "events":
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(400)]
        for i in range(400): eng.playout_step(ev, events=evs[i])
    else:
        for i in range(400): eng.playout_step(ev)
    eng.finish_move(); eng.harvest(); torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3
for mode in ("eager", "events", "eager", "events"):
    print(mode, ["%.0f" % ply(mode) for _ in range(3)])
eng.capture_steps(ev, 8, warmup=2)
print("graph", ["%.0f" % ply("graph") for _ in range(4)])
print("cpu count", os.cpu_count())
