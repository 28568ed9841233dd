"""Does splitting the boards of one GPU into two groups on two HIP streams hide the tree kernels
(k_select tail, rules op, backup) behind the other group's network evaluation?
Prints playouts/s for 1x4096, 2x2048 and 2x4096 boards."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from alphazero_quoridor_amd.engine import SelfPlayEngine
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

dev = torch.device("cuda:0"); torch.manual_seed(2026); torch.backends.cudnn.benchmark = True
net = PolicyValueNet(use_gpu=True)

def run(n_groups, boards, steps=300, desync=300):
    engs = [SelfPlayEngine(boards, n_playout=400, seed=1 + g, device=dev) for g in range(n_groups)]
    evs = [net.evaluator("per_leaf") for _ in range(n_groups)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_groups)]
    for g in range(n_groups):
        with torch.cuda.stream(streams[g]):
            for _ in range(desync):
                engs[g].run_playouts(evs[g], 4); engs[g].finish_move(); engs[g].harvest()
    torch.cuda.synchronize()
    best = 0.0
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            for g in range(n_groups):
                with torch.cuda.stream(streams[g]):
                    engs[g].playout_step(evs[g])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = max(best, steps * n_groups * boards / dt)
    print(f"{n_groups} x {boards} boards: {best:,.0f} playouts/s  ({dt / steps * 1e3:.3f} ms per round of steps)", flush=True)
    del engs, evs
    torch.cuda.empty_cache()

run(1, 4096)
run(2, 2048)
run(2, 4096)
run(1, 8192)
