"""Pure-MCTS rollout player and the win-rate gate (reference: pure_mcts.py, train.py:30-31,108).

The reference's pure MCTS is the same tree search with two substitutions (pure_mcts.py:13-16,
59-103): priors are uniform over the legal moves and a leaf is valued by ONE random rollout --
uniformly random legal moves until somebody wins, at most 1,000 iterations, +1 / -1 / 0 from
the point of view of the side to move at the leaf.  Here that is an *evaluator* for the same
``SelfPlayEngine`` the network player uses:

    RolloutEvaluator(leaf boards) -> (p = 1/k on the k legal moves, v = rollout outcome)

with the rollouts of a whole leaf batch running on the GPU (``qz_rollout``: the engine's own
move-generation kernels + a pick-and-step kernel per iteration, Philox random stream).  Terminal
leaves never reach it (the engine backs them up itself, with the reference's sign convention).

``evaluate_against_pure_mcts`` plays the evaluation match of the AlphaZero pipeline (the
reference left the call commented out, train.py:108): every game of the match side by side, the
net player as player 1 in half of them.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _cabi, rules
from .boards import DeviceBoards, opening_packed
from .engine import SelfPlayEngine


def rollout(boards: DeviceBoards, limit=1000, seed=0) -> torch.Tensor:
    """MCTS._evaluate_rollout (pure_mcts.py:81-103) for every board, IN PLACE (the boards are
    played out): int8 [n] on the device: +1 the side to move at the start won, -1 it lost, 0
    nobody won within `limit` iterations."""
    L = _cabi.load()
    n = boards.n
    value = torch.zeros(n, dtype=torch.int8, device=boards.device)
    scratch = torch.empty(int(L.qz_rollout_scratch_bytes(n)), dtype=torch.uint8, device=boards.device)
    with torch.cuda.device(boards.device):
        _cabi.check(L.qz_rollout(boards.byref(), n, int(limit), int(seed) & 0xFFFFFFFFFFFFFFFF, value.data_ptr(), scratch.data_ptr(),
                                 torch.cuda.current_stream(boards.device).cuda_stream))
    return value


class RolloutEvaluator:
    """policy_value_fn + _evaluate_rollout of pure_mcts.py as a leaf evaluator for the engine:
    uniform priors over the leaf's legal moves, value = one random rollout per leaf.  The engine
    hands it a COPY of the leaf boards (the rollout plays them out in place) and their legal sets."""

    takes_leaf_copy = True

    def __init__(self, limit=1000, seed=0):
        self.limit = int(limit)
        self.seed = int(seed)
        self.calls = 0

    @torch.no_grad()
    def from_boards(self, leaf: DeviceBoards, leaf_mask: torch.Tensor):
        n, dev = leaf.n, leaf.device
        bits = (leaf_mask.view(torch.int32).unsqueeze(2) >> torch.arange(32, device=dev, dtype=torch.int32)) & 1
        legal = bits.reshape(n, 160)[:, :140].to(torch.float32)
        k = legal.sum(dim=1, keepdim=True).clamp(min=1.0)
        p = (legal / k).contiguous()                     # np.ones(k) / k on the legal moves (pure_mcts.py:13-16)
        self.calls += 1
        v = rollout(leaf, self.limit, seed=self.seed + 0x9E3779B97F4A7C15 * self.calls).to(torch.float32)
        return p, v.contiguous()


# ------------------------------------------------------------------------------ reference-shaped API
def rollout_policy_fn(game):
    """pure_mcts.py:7-10: a random score per legal action (the rollout plays the arg-max)."""
    acts = game.actions()
    return zip(acts, np.random.rand(len(acts)))


def policy_value_fn(game):
    """pure_mcts.py:13-16: uniform priors, value 0."""
    acts = game.actions()
    return zip(acts, np.ones(len(acts)) / len(acts)), 0


class MCTS(object):
    """Pure MCTS on one board (pure_mcts.py:58-117): `n_playout` playouts with uniform priors
    and rollout values, then the most visited root child."""

    def __init__(self, policy_value_fn=policy_value_fn, c_puct=5, n_playout=10000, seed=0):
        self._c_puct, self._n_playout = c_puct, n_playout
        self._engine = SelfPlayEngine(1, n_playout=n_playout, c_puct=c_puct, temp=1.0, is_selfplay=0, seed=seed, traj_pool_pages=1)
        self._evaluator = RolloutEvaluator(seed=seed)

    def get_move(self, game):
        from .quoridor import Quoridor

        packed = game.packed() if hasattr(game, "packed") else Quoridor.packed(game)
        self._engine.set_boards(DeviceBoards.from_packed(packed, self._engine.device), reset_trees=False)
        self._engine.run_playouts(self._evaluator, self._n_playout)
        visits = self._engine.root_children()[0].cpu().numpy()[0]
        return most_visited(visits[None])[0]

    def update_with_move(self, last_move):
        mv = 255 if (last_move is None or last_move < 0 or last_move >= 140) else int(last_move)
        self._engine.update_with_move(torch.tensor([mv], dtype=torch.uint8))

    def __str__(self):
        return "MCTS"


class MCTSPlayer(object):
    """pure_mcts.py:120-142."""

    def __init__(self, c_puct=5, n_playout=50, seed=0):
        self.mcts = MCTS(policy_value_fn, c_puct, n_playout, seed=seed)

    def set_player_ind(self, p):
        self.player = p

    def reset_player(self):
        self.mcts.update_with_move(-1)

    def choose_action(self, game):
        if len(game.actions()) > 0:
            move = self.mcts.get_move(game)
            self.mcts.update_with_move(-1)
            return move
        print("WARNING: the board is full")

    def __str__(self):
        return "MCTS {}".format(getattr(self, "player", "?"))


def most_visited(visits: np.ndarray) -> np.ndarray:
    """visits [n,140] (-1 = not a child) -> the most visited child per row; ties go to the first
    in the reference's actions() order, like max() over the children dict (pure_mcts.py:107)."""
    order = np.array(rules.ACTION_ORDER)
    v = visits[:, order]
    return order[np.argmax(v, axis=1)]  # numpy's argmax returns the first maximum


# ------------------------------------------------------------------------------ the evaluation match
def evaluate_against_pure_mcts(pvn, n_games=10, n_playout=400, c_puct=5, pure_n_playout=1000, seed=0, temp=1e-3, max_plies=2000,
                               rollout_limit=1000, bn_mode=None):
    """n_games games between the net's MCTS player (MCTSPlayer(policy_value_fn, c_puct, n_playout),
    is_selfplay=0: a fresh tree per move, move ~ pi at temp 1e-3, mcts.py:183-187) and the pure-MCTS
    player (most visited child of `pure_n_playout` rollout playouts), all side by side; the net
    player starts in the even games.  -> {"wins", "losses", "ties", "plies"} from the net player's
    point of view; a game still running after max_plies counts as a tie."""
    dev = pvn.device
    n = int(n_games)
    boards = DeviceBoards.from_packed(opening_packed(n), dev)
    az = SelfPlayEngine(n, n_playout=n_playout, c_puct=c_puct, temp=1.0, is_selfplay=0, seed=seed, device=dev, traj_pool_pages=1)
    pure = SelfPlayEngine(n, n_playout=pure_n_playout, c_puct=c_puct, temp=1.0, is_selfplay=0, seed=seed + 1, device=dev, traj_pool_pages=1)
    ev_net = pvn.evaluator(bn_mode)
    ev_roll = RolloutEvaluator(limit=rollout_limit, seed=seed + 2)
    rng = np.random.RandomState(seed)
    net_is_p1 = np.arange(n) % 2 == 0
    winner = np.zeros(n, dtype=np.int64)
    plies = 0
    try:
        while plies < max_plies and (winner == 0).any():
            packed = boards.to_packed()
            cur = packed["cur"].astype(np.int64)
            net_to_move = (cur == 1) == net_is_p1
            live = winner == 0
            moves = np.full(n, 255, dtype=np.uint8)
            for eng, ev, mine, n_po in ((az, ev_net, live & net_to_move, n_playout), (pure, ev_roll, live & ~net_to_move, pure_n_playout)):
                if not mine.any():
                    continue
                # boards this player does not move (or finished ones) are parked on the opening: their search is ignored
                load = packed.copy()
                load[~mine] = opening_packed(1)[0]
                eng.set_boards(DeviceBoards.from_packed(load, dev), reset_trees=True)
                eng.run_playouts(ev, n_po)
                visits = eng.root_children()[0].cpu().numpy()
                if eng is az:  # move ~ softmax(log(visits + 1e-10) / temp) (mcts.py:141-144, 186)
                    for b in np.nonzero(mine)[0]:
                        acts = [a for a in rules.ACTION_ORDER if visits[b, a] >= 0]
                        x = np.log(visits[b, acts].astype(np.float64) + 1e-10) / temp
                        pr = np.exp(x - x.max())
                        moves[b] = acts[rng.choice(len(acts), p=pr / pr.sum())]
                else:
                    moves[mine] = most_visited(visits)[mine]
            done, win = rules.step(boards, torch.from_numpy(moves))
            done, win = done.cpu().numpy(), win.cpu().numpy()
            winner = np.where((winner == 0) & (done != 0), win, winner)
            plies += 1
    finally:
        az.close()
        pure.close()
    net_won = ((winner == 1) & net_is_p1) | ((winner == 2) & ~net_is_p1)
    return {"wins": int(net_won.sum()), "losses": int(((winner != 0) & ~net_won).sum()), "ties": int((winner == 0).sum()), "plies": plies}
