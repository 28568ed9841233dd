"""``MCTS`` / ``MCTSPlayer`` with the reference's surface (mcts.py:83-199), searching on the GPU.

The tree lives in the engine's HBM arena (``SelfPlayEngine`` with one board); ``_playout``
runs the select and expand/backup kernels.  Two leaf-evaluation routes:

* device route -- when the policy callback is ``PolicyValueNet.policy_value_fn`` of this
  package, the leaf planes written by the select kernel go straight into the network and the
  priors straight into the expand kernel; nothing crosses PCIe per playout;
* callback route -- any other ``policy_value_function(game) -> (iterable[(action, prob)],
  value)`` (the reference's contract, policy_value_net.py:145-164 / pure_mcts.py:13-16): the
  leaf board is handed to the callback as a ``Quoridor`` object.  Unlike the reference
  (mcts.py:117) the callback is not invoked on finished games, where its result is ignored
  anyway and where the reference's own callback crashes.

Visit counts -> pi and move sampling use host numpy exactly as written in the reference
(mcts.py:141-144, 181-187), so with equal visit counts and an equal ``np.random`` state the
single-game front end reproduces the reference's moves bit for bit.  The batched path
(``TrainPipeline`` / ``SelfPlayEngine``) does both on the device instead.
"""
from __future__ import annotations

import numpy as np
import torch

from . import rules
from .boards import DeviceBoards
from .engine import SelfPlayEngine
from .quoridor import Quoridor, _device


def softmax(x):
    probs = np.exp(x - np.max(x))
    probs /= np.sum(probs)
    return probs


class _NodeView(object):
    """Read-only snapshot of a tree node, shaped like the reference's TreeNode
    (mcts.py:12-80) for callers that inspect ``mcts._root``."""

    def __init__(self, n_visits=0, Q=0.0, P=1.0, children=None):
        self._parent = None
        self._children = children or {}
        self._n_visits = n_visits
        self._Q = Q
        self._u = 0
        self._P = P

    def is_leaf(self):
        return self._children == {}

    def is_root(self):
        return self._parent is None


class MCTS(object):
    def __init__(self, policy_value_fn, c_puct=5, n_playout=1800, is_selfplay=1, fix_terminal_sign=False,
                 node_cap=0, edge_cap=0, use_graph=True):
        self._policy = policy_value_fn
        self._c_puct = c_puct
        self._n_playout = n_playout
        self._engine = SelfPlayEngine(1, n_playout=n_playout, c_puct=c_puct, temp=1.0, is_selfplay=is_selfplay,
                                      device=_device(), fix_terminal_sign=fix_terminal_sign, node_cap=node_cap,
                                      edge_cap=edge_cap, traj_pool_pages=1)
        # device route only: replay the playout step (select -> rules -> net -> expand/backup, ~25
        # launches on one board) as a HIP graph; one board is launch-bound, not GPU-bound
        self._use_graph = bool(use_graph)
        self._graph_tried = False
        owner = getattr(policy_value_fn, "__self__", None)
        self._evaluator = None
        if owner is not None and hasattr(owner, "evaluator") and getattr(policy_value_fn, "__name__", "") == "policy_value_fn" \
                and getattr(owner, "device", torch.device("cpu")).type == "cuda":
            self._evaluator = owner.evaluator()

    # ------------------------------------------------------------------ playouts
    def _set_root_board(self, game):
        packed = game.packed() if hasattr(game, "packed") else Quoridor.packed(game)
        self._engine.set_boards(DeviceBoards.from_packed(packed, self._engine.device), reset_trees=False)

    def _playout_on_root(self):
        e = self._engine
        if self._evaluator is not None:
            e.playout_step(self._evaluator)
            return
        leaf = e.select_boards()
        p = np.zeros((1, 140), dtype=np.float32)
        v = np.zeros(1, dtype=np.float32)
        if int(e.leaf_term.cpu().numpy()[0]) == 0:
            action_probs, value = self._policy(Quoridor.from_packed(leaf.to_packed()))
            for a, pr in action_probs:
                p[0, int(a)] = pr
            v[0] = float(value)
        e.expand_backup(torch.from_numpy(p).to(e.device), torch.from_numpy(v).to(e.device))

    def _playout(self, game):
        """One simulation from `game` (mcts.py:103-127)."""
        self._set_root_board(game)
        self._playout_on_root()

    def get_move_probs(self, game, temp=1e-3):
        """n_playout simulations, then (acts, probs) over the root's children in the
        reference's insertion order (mcts.py:129-144)."""
        self._set_root_board(game)
        if self._evaluator is not None:
            todo = self._n_playout
            if self._use_graph and not self._graph_tried and todo >= 16:
                self._graph_tried = True
                try:
                    todo -= self._engine.capture_steps(self._evaluator, steps_per_graph=8, warmup=3)
                except Exception:  # capture unsupported for some op on this stack: stay eager
                    self._engine._graph = None
            self._engine.run_playouts(self._evaluator, todo)
        else:
            for _ in range(self._n_playout):
                self._playout_on_root()
        visits = self._engine.root_children()[0].cpu().numpy()[0]
        acts = tuple(a for a in rules.ACTION_ORDER if visits[a] >= 0)
        counts = np.array([visits[a] for a in acts])
        act_probs = softmax(1.0 / temp * np.log(counts + 1e-10))
        return acts, act_probs

    def update_with_move(self, last_move):
        """Re-root at the child for `last_move`, or start a fresh tree (mcts.py:146-151)."""
        mv = 255 if (last_move is None or last_move < 0 or last_move >= 140) else int(last_move)
        self._engine.update_with_move(torch.tensor([mv], dtype=torch.uint8))

    @property
    def _root(self):
        visits, q, prior, root_n = (t.cpu().numpy()[0] for t in self._engine.root_children())
        children = {a: _NodeView(int(visits[a]), float(q[a]), np.float32(prior[a]))
                    for a in rules.ACTION_ORDER if visits[a] >= 0}
        return _NodeView(int(root_n), 0.0, 1.0, children)

    def __str__(self):
        return "MCTS"


class MCTSPlayer(object):
    def __init__(self, policy_value_function, c_puct=5, n_playout=2000, is_selfplay=0, **engine_kwargs):
        self.mcts = MCTS(policy_value_function, c_puct, n_playout, **engine_kwargs)
        self._is_selfplay = is_selfplay

    def set_player_ind(self, p):
        self.player = p

    def reset_player(self):
        self.mcts.update_with_move(-1)

    def choose_action(self, game, temp=1e-3, return_prob=0):
        """mcts.py:172-196."""
        sensible_moves = game.actions()
        move_probs = np.zeros(140)
        if len(sensible_moves) > 0:
            acts, probs = self.mcts.get_move_probs(game, temp)
            move_probs[list(acts)] = probs
            if self._is_selfplay:
                move = np.random.choice(acts, p=0.75 * probs + 0.25 * np.random.dirichlet(0.3 * np.ones(len(probs))))
                self.mcts.update_with_move(move)
            else:
                move = np.random.choice(acts, p=probs)
                self.mcts.update_with_move(-1)
            if return_prob:
                return move, move_probs
            return move
        print("WARNING: the board is full")

    def __str__(self):
        return "MCTS {}".format(self.player)
