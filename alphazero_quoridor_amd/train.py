"""``TrainPipeline`` with the reference's attribute names and defaults (train.py:12-116).

What the reference does one game at a time on the host, this pipeline keeps on the GPU:

* ``collect_selfplay_data``: ``n_boards`` games in flight inside a ``SelfPlayEngine``; returns as
  soon as ``n_games`` more of them have finished, the others keep their trees and positions
  (continuous refill).  Finished (board, pi, z) tuples go straight into a device-resident
  ``ReplayBuffer`` (588 B per ply; the reference's deque holds 72 KB of float64 per ply).
* ``policy_update``: the minibatch is gathered and re-encoded on the device, old/new policies,
  KL, explained variance are device tensors; per epoch ONE scalar crosses to the host (the KL
  that decides whether to stop early, train.py:79-80).
* multi-GPU (one process per GPU, torch.distributed / RCCL): every rank plays its own boards,
  finished tuples are all-gathered into every rank's buffer, all ranks start from rank 0's
  weights and average their gradients, so the replicas stay one model; rank 0 alone writes
  checkpoints and ``loss.txt``.
* ``policy_evaluate``: the win-rate gate against the pure-MCTS rollout player (train.py:30-31 and
  the commented call at :108; pure_mcts.py), all games of the match played side by side.
"""
from __future__ import annotations

import random

import numpy as np
import torch

from . import dist as qdist
from .engine import REFERENCE_MAX_DEPTH, BoardGroups
from .mcts import MCTSPlayer
from .policy_value_net import PolicyValueNet
from .quoridor import Quoridor
from .replay import ReplayBuffer, ShardWriter


class TrainPipeline(object):
    def __init__(self, init_model=None, n_boards=1024, device=None, seed=0, bn_mode="per_leaf",
                 nn_dtype=torch.float32, use_graph=False, n_groups=1, shard_dir=None, max_depth=REFERENCE_MAX_DEPTH):
        self.game = Quoridor()
        # the reference's hyper-parameters, same names and values (train.py:17-31)
        self.learn_rate = 2e-3
        self.lr_multiplier = 1.0
        self.temp = 1.0
        self.n_playout = 400
        self.c_puct = 5
        self.buffer_size = 10000
        self.batch_size = 128
        self.play_batch_size = 1
        self.epochs = 5
        self.kl_targ = 0.02
        self.check_freq = 50
        self.game_batch_num = 1500
        self.best_win_ratio = 0.0
        self.pure_mcts_playout_num = 1000
        self.pure_mcts_rollout_limit = 1000   # pure_mcts.py:81 `limit`
        self.policy_value_net = PolicyValueNet(model_file=init_model, bn_mode=bn_mode, device=device)
        self.data_buffer = ReplayBuffer(self.buffer_size, self.policy_value_net.device)
        self._mcts_player = None
        # engine knobs (new)
        self.n_boards = n_boards
        self.seed = seed
        self.bn_mode = bn_mode
        self.nn_dtype = nn_dtype
        self.use_graph = use_graph
        self.n_groups = n_groups  # board groups on separate HIP streams (engine.BoardGroups)
        self.async_loop = True        # self-play through the asynchronous loop (False: the lock-step engine, one ply of every board per harvest)
        self.rounds_per_harvest = 64  # rounds of the loop between two harvests (+ all-gathers)
        self.budget_us = 1000         # wall-clock budget of a k_advance launch.  (bench.py runs 3,000 us with one deadline per launch: the optimum for 13,312 boards deep
                                      # in random-network games, where nearly every leaf is in the memo; a board whose mover still has walls makes
                                      # ONE playout per round, so a population of short games is better served by short rounds)
        # a game whose search descends deeper than this is DROPPED (not in the replay data; counted and logged by
        # collect_selfplay_data: `games_dropped`): the reference's recursive backup raises RecursionError there
        # (mcts.py:55-62, Python's recursion limit) and its whole run ends.  An explicit choice, because it shapes the data:
        # the dropped games are the long ones (1.5 % of reference-faithful random-net games).  max_depth=0 plays every game on,
        # at the price of launches that last as long as one board's 10,000-level descent.
        self.max_depth = int(max_depth)
        self.games_dropped = {"depth": 0, "no_legal_move": 0, "other": 0}   # since the engine was created
        self.episode_len = 0
        self._engine = None
        self.shards = ShardWriter(shard_dir, rank=self._rank(), n_playout=self.n_playout) if shard_dir else None
        self.policy_value_net.sync_from_rank0()
        self.last_update = {}

    # ------------------------------------------------------------------ plumbing
    @staticmethod
    def _rank():
        return torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0

    @staticmethod
    def _is_dist():
        return torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1

    @property
    def mcts_player(self):
        """Single-game player with the reference's wiring (train.py:37-38); built on demand."""
        if self._mcts_player is None:
            self._mcts_player = MCTSPlayer(self.policy_value_net.policy_value_fn, c_puct=self.c_puct,
                                           n_playout=self.n_playout, is_selfplay=1)
        return self._mcts_player

    def engine(self) -> BoardGroups:
        if self._engine is None:
            net = self.policy_value_net
            if len(self.data_buffer) == 0 and self.data_buffer.capacity != self.buffer_size:
                self.data_buffer = ReplayBuffer(self.buffer_size, net.device)  # buffer_size was changed after __init__, like the other knobs
            self._engine = BoardGroups(self.n_boards, self.n_groups, lambda: net.evaluator(self.bn_mode, self.nn_dtype),
                                       seed=qdist.shard_seed(self.seed, self._rank()), device=net.device,
                                       n_playout=self.n_playout, c_puct=self.c_puct, temp=self.temp, is_selfplay=1, max_depth=self.max_depth)
            if self.use_graph and self.n_groups == 1:
                with torch.cuda.stream(self._engine.streams[0]):
                    e0, ev0 = self._engine.engines[0], self._engine.evaluators[0]
                    if self._use_async(self._engine):
                        # the asynchronous loop replays whole rounds; its warm-up rounds are ordinary playouts of the first
                        # move (nothing extra is spent on a root: the loop counts playouts per board)
                        e0.capture_rounds(ev0, rounds=16, max_playouts=4096, budget_us=self.budget_us)
                    else:
                        e0.capture_steps(ev0, 1, warmup=2)
                        e0.reset()  # the warm-up's two playouts must not count towards the first move's n_playout
        return self._engine

    # ------------------------------------------------------------------ self-play -> replay buffer
    def _extend_buffer(self, tbs):
        """Finished tuples of this ply -> replay buffer (device to device); returns the number of
        games they hold.  Multi-GPU: ONE all-gather per ply on every rank, whether or not this rank
        finished a game, and the game count is the sum over ranks (identical everywhere), so all
        ranks leave collect_selfplay_data after the same ply."""
        n_games = sum(tb.n_games for tb in tbs)
        if self.shards is not None:
            for tb in tbs:
                self.shards.add(tb)
        if not self._is_dist():
            for tb in tbs:
                self.data_buffer.extend(tb)
            return n_games
        dev = self.policy_value_net.device
        bufs = [qdist.pack_tuples(tb.boards.hbits, tb.boards.vbits, tb.boards.meta, tb.pi, tb.z) for tb in tbs]
        bufs.append(torch.zeros((0, qdist.TUPLE_BYTES), dtype=torch.uint8, device=dev))
        buf, n_games = qdist.allgather_tuples(torch.cat(bufs), n_games=n_games)
        if buf.shape[0]:
            self.data_buffer.extend(buf)
        return n_games

    def collect_selfplay_data(self, n_games=1):
        """Generate self-play data until `n_games` more games are complete (train.py:55-63)."""
        eng = self.engine()
        eng.join_main()  # self-play streams start after whatever updated the weights
        got = 0
        # the asynchronous loop (boards on their own clocks, leaf-evaluation memo: engine.SelfPlayEngine.selfplay_round)
        # whenever the evaluator is the HIP evaluation it can run on its miss list; the same games either way
        use_async = self._use_async(eng)
        while got < n_games:
            if use_async:
                eng.run_rounds(self.rounds_per_harvest, max_playouts=4096, budget_us=self.budget_us)
            else:
                eng.play_ply()
            tbs = eng.harvest()
            for tb in tbs:
                last = int(tb.game.max().item())
                self.episode_len = int((tb.game == last).sum().item())
            eng.synchronize()
            got += self._extend_buffer(tbs)
        self._log_dropped(eng)

    def _use_async(self, eng):
        return self.async_loop and all(getattr(ev, "engine_route_ok", lambda: False)() for ev in eng.evaluators)

    def _log_dropped(self, eng):
        """Games the engine dropped since the last call, by cause (the reference would have crashed on them): they are
        missing from the replay data, so say so."""
        st = eng.stats()
        now = {"depth": st["aborted_depth"], "no_legal_move": st["aborted_no_move"],
               "other": st["games_aborted"] - st["aborted_depth"] - st["aborted_no_move"]}
        new = {k: now[k] - self.games_dropped[k] for k in now}
        self.games_dropped = now
        if any(new.values()) and self._rank() == 0:
            print("self-play dropped {} game(s): {} deeper than max_depth={} (the reference's RecursionError), {} without a legal "
                  "move (the reference's 'the board is full' crash), {} other".format(sum(new.values()), new["depth"], self.max_depth,
                                                                                     new["no_legal_move"], new["other"]))

    # ------------------------------------------------------------------ training
    def _mean_over_ranks(self, x: torch.Tensor) -> torch.Tensor:
        if self._is_dist():
            torch.distributed.all_reduce(x)
            x = x / torch.distributed.get_world_size()
        return x

    def policy_update(self):
        """One KL-controlled update of the policy-value net on a replay minibatch (train.py:65-92):
        up to `epochs` optimiser steps on the same minibatch, stopped early when the policy moved
        more than 4 kl_targ away from where it started; afterwards the learning-rate multiplier is
        lowered (KL > 2 kl_targ) or raised (KL < kl_targ / 2) by 1.5 within [0.1, 10]."""
        net = self.policy_value_net
        states, pi, z = self.data_buffer.sample(self.batch_size)  # random.sample positions, states re-encoded on the GPU
        old_probs, old_v = net.policy_value_t(states)
        old_logp = torch.log(old_probs + 1e-10)
        loss = entropy = None
        new_v = old_v
        kl = 0.0
        steps = 0
        for _ in range(self.epochs):
            loss, entropy = net.train_step_t(states, pi, z, self.learn_rate * self.lr_multiplier)
            new_probs, new_v = net.policy_value_t(states)
            kl_t = torch.mean(torch.sum(old_probs * (old_logp - torch.log(new_probs + 1e-10)), dim=1))
            kl = float(self._mean_over_ranks(kl_t))  # the one host read per epoch; identical on all ranks
            steps += 1
            if kl > self.kl_targ * 4:
                break
        if kl > self.kl_targ * 2 and self.lr_multiplier > 0.1:
            self.lr_multiplier /= 1.5
        elif kl < self.kl_targ / 2 and self.lr_multiplier < 10:
            self.lr_multiplier *= 1.5
        net.average_buffers()
        var_z = torch.var(z, unbiased=False)
        ev_old = 1 - torch.var(z - old_v.flatten(), unbiased=False) / var_z
        ev_new = 1 - torch.var(z - new_v.flatten(), unbiased=False) / var_z
        loss, entropy = float(loss), float(entropy)
        self.last_update = {"kl": kl, "lr_multiplier": self.lr_multiplier, "loss": loss, "entropy": entropy, "optimizer_steps": steps,
                            "explained_var_old": float(ev_old), "explained_var_new": float(ev_new)}
        if self._rank() == 0:
            print("kl:{kl:.5f},lr_multiplier:{lr_multiplier:.3f},loss:{loss},entropy:{entropy},"
                  "explained_var_old:{explained_var_old:.3f},explained_var_new:{explained_var_new:.3f}".format(**self.last_update))
        return loss, entropy

    # ------------------------------------------------------------------ evaluation gate
    def policy_evaluate(self, n_games=10, max_plies=2000):
        """Win ratio of the current net's MCTS player against the pure-MCTS rollout player with
        `pure_mcts_playout_num` playouts (train.py:30-31, :108), alternating who starts; all
        games run side by side on the GPU (pure_mcts.evaluate_against_pure_mcts)."""
        from .pure_mcts import evaluate_against_pure_mcts

        res = evaluate_against_pure_mcts(self.policy_value_net, n_games=n_games, n_playout=self.n_playout, c_puct=self.c_puct,
                                         pure_n_playout=self.pure_mcts_playout_num, seed=self.seed, max_plies=max_plies,
                                         rollout_limit=self.pure_mcts_rollout_limit, bn_mode=self.bn_mode)
        win_ratio = (res["wins"] + 0.5 * res["ties"]) / n_games
        if self._rank() == 0:
            print("num_playouts:{}, win: {}, lose: {}, tie:{}".format(self.pure_mcts_playout_num, res["wins"], res["losses"], res["ties"]))
        return win_ratio

    # ------------------------------------------------------------------ the loop
    def run(self, evaluate=False):
        """train.py:94-111.  `evaluate=True` also runs the win-rate gate every check_freq batches
        (commented out in the reference) and keeps ckpt/best_policy.pth."""
        rank0 = self._rank() == 0
        try:
            for i in range(self.game_batch_num):
                self.collect_selfplay_data(self.play_batch_size)
                if rank0:
                    print("batch i:{}, episode_len:{}".format(i + 1, self.episode_len))
                if len(self.data_buffer) > self.batch_size:
                    loss, _ = self.policy_update()
                    if rank0:
                        print("LOSS:", loss)
                        with open("loss.txt", "a") as f:
                            f.write(str(loss) + "\n")
                if (i + 1) % self.check_freq == 0:
                    if rank0:
                        print("current self-play batch: {}".format(i + 1))
                        self.policy_value_net.save_model("current_policy")
                    if evaluate:
                        ratio = self.policy_evaluate()
                        if ratio > self.best_win_ratio and rank0:
                            self.best_win_ratio = ratio
                            self.policy_value_net.save_model("best_policy")
        except KeyboardInterrupt:
            print("\n\rquit")
        finally:
            if self.shards is not None:
                self.shards.flush()


if __name__ == "__main__":
    TrainPipeline().run()
