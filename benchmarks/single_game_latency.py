"""Latency of the reference-shaped single-game route: MCTSPlayer.choose_action on one board,
n_playout=400 (what a user of the reference's train.py / game loop sees per move)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from alphazero_quoridor_amd.quoridor import Quoridor
from alphazero_quoridor_amd.mcts import MCTSPlayer
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet
torch.manual_seed(0); torch.backends.cudnn.benchmark = True
net = PolicyValueNet(use_gpu=True)
game = Quoridor()
player = MCTSPlayer(net.policy_value_fn, c_puct=5, n_playout=400, is_selfplay=1)
player.choose_action(game, temp=1.0, return_prob=1)  # warm-up (MIOpen find, allocations)
game = Quoridor(); player.reset_player()
t0 = time.perf_counter(); n = 0
for _ in range(10):
    move, pi = player.choose_action(game, temp=1.0, return_prob=1)
    game.step(move); n += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("single game, n_playout=400: %.1f ms per move (%.0f playouts/s)" % (dt / n * 1e3, 400 * n / dt))
