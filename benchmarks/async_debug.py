"""debug driver of the asynchronous loop: rounds with timing and counters"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from alphazero_quoridor_amd.engine import SelfPlayEngine
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet
B = int(os.environ.get("BOARDS", 192)); NP = int(os.environ.get("PLAYOUTS", 24)); MP = int(os.environ.get("MAXP", 32))
FIX = int(os.environ.get("FIX", 1)); ITERS = int(os.environ.get("ITERS", 200)); R = int(os.environ.get("ROUNDS", 8)); BUD = int(os.environ.get("BUDGET", 0))
dev = torch.device("cuda:0"); torch.manual_seed(0)
ev = PolicyValueNet(use_gpu=True).evaluator("per_leaf")
eng = SelfPlayEngine(B, n_playout=NP, seed=77, device=dev, fix_terminal_sign=bool(FIX))
t0 = time.time(); games = 0
for i in range(ITERS):
    eng.run_rounds(ev, R, max_playouts=MP, budget_us=BUD)
    torch.cuda.synchronize()
    tb = eng.harvest()
    if tb is not None: games += tb.n_games
    if i % max(1, ITERS // 20) == 0:
        st = eng.stats()
        print("it %d t %.2fs rounds %d playouts %d plies %d games %d hits %d evals %d waiting %d aborted %d ovf %d" % (i, time.time() - t0, st["rounds"], st["playouts"], st["plies_played"], games, st["memo_hits"], st["nn_evals"], st["waiting_boards"], st["games_aborted"], st["node_overflow"]), flush=True)
print("done", time.time() - t0, flush=True)
