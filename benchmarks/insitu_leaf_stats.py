"""What do the leaf boards of a bench-like run look like?  Fraction terminal, fraction whose mover
has no wall left (no path search / floods needed), walls on the board, legal-move counts."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from alphazero_quoridor_amd.engine import SelfPlayEngine
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet
dev = torch.device("cuda:0"); torch.manual_seed(2026); torch.backends.cudnn.benchmark = True
net = PolicyValueNet(use_gpu=True); ev = net.evaluator("per_leaf")
eng = SelfPlayEngine(4096, n_playout=400, seed=1, device=dev)
for _ in range(700):
    eng.run_playouts(ev, 4); eng.finish_move(); eng.harvest()
for _ in range(100):
    eng.playout_step(ev)
acc = []
for _ in range(20):
    db = eng.select_boards()
    meta = db.meta
    w1 = (meta >> 16) & 0xFF; w2 = (meta >> 24) & 0xFF; cur = (meta >> 32) & 0xFF
    mover = torch.where(cur == 1, w1, w2)
    term = eng.leaf_term.bool()
    mask = eng.leaf_mask.to(torch.int64) & 0xFFFFFFFF
    bits = torch.arange(32, device=dev)
    legal = ((mask.unsqueeze(-1) >> bits) & 1).sum(dim=(1, 2))
    acc.append([term.float().mean().item(), ((mover == 0) & ~term).float().mean().item(), (20 - w1 - w2).float().mean().item(),
                legal.float().mean().item()])
    p, v = ev(eng.select()); eng.expand_backup(p, v)
a = torch.tensor(acc).mean(0).tolist()
print("terminal %.3f  mover_has_no_wall %.3f  walls_on_board %.1f  legal_moves %.1f" % tuple(a))
