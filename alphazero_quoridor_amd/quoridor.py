"""``Quoridor`` -- single-game front end with the reference's class surface
(quoridor.py:5-610), computing on the GPU.

The game state lives in plain Python attributes with the reference's names
(``_positions``, ``_intersections``, ``_player{1,2}_walls_remaining``, ``current_player``)
so existing callers can read and poke them; every rule evaluation packs the state into the
24-byte board record and runs the HIP kernels through the C ABI on a batch of one:

    actions()  -> qz_movegen   (quoridor.py:138-157)
    state()    -> qz_encode    (quoridor.py:58-131)
    step()     -> qz_step      (quoridor.py:159-186)

For throughput use ``SelfPlayEngine`` / ``TrainPipeline.collect_selfplay_data`` (thousands
of boards per launch); this class is the drop-in for code written against the reference.
Quirks kept on purpose: directed row-0 edges, unconditional diagonal jumps,
off-board winning jumps, no player rotation on a terminal move, ``clone()`` returns a fresh
game.  Unlike the reference, ``step()`` does not waste a second ``actions()`` call unless
``safe=True``, and ``actions()`` / ``state()`` on a finished game do not crash.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from . import _cabi, rules
from .boards import DeviceBoards


def _device():
    if not torch.cuda.is_available():
        raise _cabi.QzError(_cabi.E_NO_DEVICE, "Quoridor needs a HIP device: the rules engine has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


class Quoridor(object):
    HORIZONTAL = 1
    VERTICAL = -1

    def __init__(self, safe=False):
        self.safe = safe
        self.action_space = 140
        self.n_players = 2
        self.players = [1, 2]
        self.reset()

    def load(self, p1, p2):
        self.player1 = p1
        self.player2 = p2

    def get_current_player(self):
        return self.current_player

    def reset(self):
        self.current_player = 1
        self.last_player = -1
        self.tiles = np.zeros(81)
        self._positions = {1: 4, 2: 76}
        self._DIRECTIONS = {"N": 0, "S": 1, "E": 2, "W": 3, "NN": 4, "SS": 5, "EE": 6, "WW": 7,
                            "NE": 8, "NW": 9, "SE": 10, "SW": 11}
        self.N_DIRECTIONS = 12
        self.N_TILES = 81
        self.N_ROWS = 9
        self.N_INTERSECTIONS = 64
        self._intersections = np.zeros(64)
        self._player1_walls_remaining = 10
        self._player2_walls_remaining = 10

    # ---------------------------------------------------------------- packed <-> attributes
    def packed(self) -> np.ndarray:
        rec = np.zeros(1, dtype=_cabi.PACKED_DTYPE)
        hb = vb = 0
        for ix, v in enumerate(np.asarray(self._intersections).tolist()):
            if v == 1:
                hb |= 1 << ix
            elif v == -1:
                vb |= 1 << ix
        rec["hbits"], rec["vbits"] = hb, vb
        rec["p1"], rec["p2"] = int(self._positions[1]), int(self._positions[2])
        rec["w1"], rec["w2"] = int(self._player1_walls_remaining), int(self._player2_walls_remaining)
        rec["cur"] = int(self.current_player)
        return rec

    def _load_packed(self, rec):
        rec = np.asarray(rec, dtype=_cabi.PACKED_DTYPE).reshape(-1)[0]
        hb, vb = int(rec["hbits"]), int(rec["vbits"])
        inter = np.zeros(64)
        for ix in range(64):
            if (hb >> ix) & 1:
                inter[ix] = 1
            elif (vb >> ix) & 1:
                inter[ix] = -1
        self._intersections = inter
        self._positions = {1: int(rec["p1"]), 2: int(rec["p2"])}
        self._player1_walls_remaining = int(rec["w1"])
        self._player2_walls_remaining = int(rec["w2"])
        cur = int(rec["cur"])
        if cur != self.current_player:
            self.last_player = self.current_player
        self.current_player = cur

    @classmethod
    def from_packed(cls, rec, safe=False):
        g = cls(safe=safe)
        g._load_packed(rec)
        g.last_player = 2 if g.current_player == 1 else 1
        return g

    def _boards(self):
        return DeviceBoards.from_packed(self.packed(), _device())

    # ---------------------------------------------------------------- reference API
    def state(self):
        """float64 [26,9,9] (quoridor.py:58-131)."""
        return rules.encode(self._boards())[0].cpu().numpy().astype(np.float64)

    def load_state(self, state):
        raise NotImplementedError("load_state is a stub in the reference too (quoridor.py:133-136)")

    def actions(self):
        """Ordered list of legal action ids (quoridor.py:138-157)."""
        if self.has_a_winner()[0]:
            return []
        return rules.mask_to_actions(rules.movegen(self._boards())[0].cpu().numpy())

    def step(self, action):
        """Apply `action`; returns done (quoridor.py:159-186)."""
        action = int(action)
        if self.safe:
            self.valid_actions = self.actions()
            if action not in self.valid_actions:
                raise ValueError("Invalid Action: {action}".format(action=action))
        if not 0 <= action < 140:
            raise ValueError("Invalid Pawn Action: {action}".format(action=action))
        b = self._boards()
        done, _ = rules.step(b, torch.tensor([action], dtype=torch.uint8))
        self._load_packed(b.to_packed())
        done = bool(done.cpu().numpy()[0])
        if done:
            print("game over !winner is player" + str(self.has_a_winner()[1]))
        return done

    def game_end(self):
        pass

    def has_a_winner(self):
        """(game_over, winner) -- player 2 is tested first (quoridor.py:193-202)."""
        if self._positions[2] < 9:
            return True, 2
        if self._positions[1] > 71:
            return True, 1
        return False, None

    def rotate_players(self):
        self.last_player = self.current_player
        self.current_player = 2 if self.current_player == 1 else 1

    def add_wall(self, wall, orientation):
        self._intersections[wall] = orientation

    def clone(self):
        return Quoridor()  # a fresh game, as in the reference (quoridor.py:570-571)

    def print_board(self):
        inter = np.asarray(self._intersections).reshape(8, 8)
        cells = [["-"] * 9 for _ in range(9)]
        for mark, p in (("X", self._positions[1]), ("O", self._positions[2])):
            if 0 <= p <= 80:
                cells[p // 9][p % 9] = mark
        for r in range(8, -1, -1):
            print("".join("%-4s" % c for c in cells[r]))
            if r > 0:
                row = inter[r - 1]
                print("  " + "".join("%-4s" % ("h" if v == 1 else ("v" if v == -1 else "")) for v in row))

    def start_self_play(self, player, is_shown=0, temp=1e-3):
        """One self-play game with `player` on both sides (quoridor.py:573-610).
        Returns (winner, zip(states, mcts_probs, winners_z))."""
        self.reset()
        states, mcts_probs, current_players = [], [], []
        while True:
            tic = time.time()
            move, move_probs = player.choose_action(self, temp=temp, return_prob=1)
            print("player %s  chosed move : %s ,prob: %.3f  spend: %.2f seconds"
                  % (self.current_player, move, move_probs[move], time.time() - tic))
            states.append(self.state())
            mcts_probs.append(move_probs)
            current_players.append(self.current_player)
            self.step(move)
            end, winner = self.has_a_winner()
            if end:
                who = np.array(current_players)
                winners_z = np.zeros(len(current_players))
                winners_z[who == winner] = 1.0
                winners_z[who != winner] = -1.0
                player.reset_player()
                if is_shown:
                    print("Game end. Winner is player:", winner)
                return winner, zip(states, mcts_probs, winners_z)
