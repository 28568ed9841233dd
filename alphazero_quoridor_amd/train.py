"""``TrainPipeline`` with the reference's attribute names and defaults (train.py:12-116);
``collect_selfplay_data`` is the accelerated path.

Where the reference plays one game at a time through ``Quoridor.start_self_play``
(train.py:55-63), this pipeline keeps ``n_boards`` games in flight inside a
``SelfPlayEngine`` and returns as soon as ``n_games`` of them have finished; the other
boards keep their trees and positions for the next call (continuous refill).  The replay
buffer receives the reference's tuple format: (float64 state [26,9,9], float64 pi [140],
float64 z).  With torch.distributed initialised (one process per GPU) every rank's finished
tuples are all-gathered into every rank's buffer.
"""
from __future__ import annotations

import random
from collections import deque

import numpy as np
import torch

from . import dist as qdist
from .boards import DeviceBoards
from .engine import BoardGroups, TupleBatch
from .mcts import MCTSPlayer
from .policy_value_net import PolicyValueNet
from .quoridor import Quoridor


class TrainPipeline(object):
    def __init__(self, init_model=None, n_boards=1024, device=None, seed=0, bn_mode="per_leaf",
                 nn_dtype=torch.float32, use_graph=False, n_groups=1):
        self.game = Quoridor()
        # the reference's hyper-parameters, same names and values (train.py:17-31)
        self.learn_rate = 2e-3
        self.lr_multiplier = 1.0
        self.temp = 1.0
        self.n_playout = 400
        self.c_puct = 5
        self.buffer_size = 10000
        self.batch_size = 128
        self.data_buffer = deque(maxlen=self.buffer_size)
        self.play_batch_size = 1
        self.epochs = 5
        self.kl_targ = 0.02
        self.check_freq = 50
        self.game_batch_num = 1500
        self.best_win_ratio = 0.0
        self.pure_mcts_playout_num = 1000
        self.policy_value_net = PolicyValueNet(model_file=init_model, bn_mode=bn_mode, device=device)
        self._mcts_player = None
        # engine knobs (new)
        self.n_boards = n_boards
        self.seed = seed
        self.bn_mode = bn_mode
        self.nn_dtype = nn_dtype
        self.use_graph = use_graph
        self.n_groups = n_groups  # board groups on separate HIP streams (engine.BoardGroups)
        self.episode_len = 0
        self._engine = None

    @property
    def mcts_player(self):
        """Single-game player with the reference's wiring (train.py:37-38); built on demand."""
        if self._mcts_player is None:
            self._mcts_player = MCTSPlayer(self.policy_value_net.policy_value_fn, c_puct=self.c_puct,
                                           n_playout=self.n_playout, is_selfplay=1)
        return self._mcts_player

    def engine(self) -> BoardGroups:
        if self._engine is None:
            rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
            net = self.policy_value_net
            self._engine = BoardGroups(self.n_boards, self.n_groups, lambda: net.evaluator(self.bn_mode, self.nn_dtype),
                                       seed=qdist.shard_seed(self.seed, rank), device=net.device,
                                       n_playout=self.n_playout, c_puct=self.c_puct, temp=self.temp, is_selfplay=1)
            if self.use_graph and self.n_groups == 1:
                with torch.cuda.stream(self._engine.streams[0]):
                    self._engine.engines[0].capture_steps(self._engine.evaluators[0], 1, warmup=2)
        return self._engine

    def _is_dist(self):
        return torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1

    def _extend_buffer(self, tbs):
        """Finished tuples of this ply -> replay buffer; returns the number of games they hold.
        Multi-GPU: ONE all-gather per ply on every rank, whether or not this rank finished a
        game, and the game count is the sum over ranks (identical everywhere), so all ranks
        leave collect_selfplay_data after the same ply."""
        n_games = sum(tb.n_games for tb in tbs)
        if self._is_dist():
            dev = self.policy_value_net.device
            bufs = [qdist.pack_tuples(tb.boards.hbits, tb.boards.vbits, tb.boards.meta, tb.pi, tb.z) for tb in tbs]
            bufs.append(torch.zeros((0, qdist.TUPLE_BYTES), dtype=torch.uint8, device=dev))
            buf, n_games = qdist.allgather_tuples(torch.cat(bufs), n_games=n_games)
            if buf.shape[0] == 0:
                return n_games
            hb, vb, meta, pi, z = qdist.unpack_tuples(buf)
            boards = DeviceBoards(len(hb), hb.device)
            boards.hbits, boards.vbits, boards.meta = hb, vb, meta
            tbs = [TupleBatch(boards, pi, z, None, n_games)]
        for tb in tbs:
            self.data_buffer.extend(tb.to_reference_tuples())
        return n_games

    def collect_selfplay_data(self, n_games=1):
        """Generate self-play data until `n_games` more games are complete (train.py:55-63)."""
        eng = self.engine()
        eng.join_main()  # self-play streams start after whatever updated the weights
        got = 0
        while got < n_games:
            eng.play_ply()
            tbs = eng.harvest()
            for tb in tbs:
                last = int(tb.game.max().item())
                self.episode_len = int((tb.game == last).sum().item())
            eng.synchronize()
            got += self._extend_buffer(tbs)

    def policy_update(self):
        """KL-adaptive policy/value update (train.py:65-92)."""
        mini_batch = random.sample(self.data_buffer, self.batch_size)
        state_batch = [d[0] for d in mini_batch]
        mcts_probs_batch = [d[1] for d in mini_batch]
        winner_batch = [d[2] for d in mini_batch]
        old_probs, old_v = self.policy_value_net.policy_value(state_batch)
        for _ in range(self.epochs):
            loss, entropy = self.policy_value_net.train_step(state_batch, mcts_probs_batch, winner_batch,
                                                             self.learn_rate * self.lr_multiplier)
            new_probs, new_v = self.policy_value_net.policy_value(state_batch)
            kl = np.mean(np.sum(old_probs * (np.log(old_probs + 1e-10) - np.log(new_probs + 1e-10)), axis=1))
            if kl > self.kl_targ * 4:
                break
        if kl > self.kl_targ * 2 and self.lr_multiplier > 0.1:
            self.lr_multiplier /= 1.5
        elif kl < self.kl_targ / 2 and self.lr_multiplier < 10:
            self.lr_multiplier *= 1.5
        z = np.array(winner_batch)
        ev_old = 1 - np.var(z - old_v.flatten()) / np.var(z)
        ev_new = 1 - np.var(z - new_v.flatten()) / np.var(z)
        print("kl:{:.5f},lr_multiplier:{:.3f},loss:{},entropy:{},explained_var_old:{:.3f},explained_var_new:{:.3f}"
              .format(kl, self.lr_multiplier, loss, entropy, ev_old, ev_new))
        return loss, entropy

    def run(self):
        try:
            for i in range(self.game_batch_num):
                self.collect_selfplay_data(self.play_batch_size)
                print("batch i:{}, episode_len:{}".format(i + 1, self.episode_len))
                if len(self.data_buffer) > self.batch_size:
                    loss, entropy = self.policy_update()
                    print("LOSS:", loss)
                    with open("loss.txt", "a") as f:
                        f.writelines(str(loss) + "\n")
                if (i + 1) % self.check_freq == 0:
                    print("current self-play batch: {}".format(i + 1))
                    self.policy_value_net.save_model("current_policy")
        except KeyboardInterrupt:
            print("\n\rquit")


if __name__ == "__main__":
    TrainPipeline().run()
