"""Interactive front end: play Quoridor against the AlphaZero MCTS player or the pure-MCTS player
(reference: game.py:47-170).

The reference's game loop unpacks ``done, winner = game.step(action)`` although its
``Quoridor.step`` returns only ``done`` (quoridor.py:159-186), so its GUI raises TypeError on the
first move.  ``step_result`` is the adapter a fixed game.py needs: it keeps the drop-in
``step()`` return shape and adds the winner from ``has_a_winner()``.  ``play`` is the loop of
game.py:86-164 without the pygame drawing (pygame is not part of this image); a GUI only has to
call ``ManualPygameAgent.receive_action`` from its click handler and render ``game`` in ``on_move``.

    python -m alphazero_quoridor_amd.game --player_type 2 --computer_type 1   # human vs AlphaZero MCTS
    python -m alphazero_quoridor_amd.game --player_type 2 --computer_type 2   # human vs pure MCTS
"""
from __future__ import annotations

import argparse
import time

from .agents import ManualCLIAgent


def step_result(game, action):
    """game.step(action) -> (done, winner): the tuple game.py:117,133,154 expects."""
    done = game.step(action)
    if isinstance(done, tuple):  # an environment that already returns the pair
        return done
    end, winner = game.has_a_winner()
    return bool(done or end), winner


def play(game, players, on_move=None, max_plies=None, log=print):
    """players: {1: agent, 2: agent}; every agent has choose_action(game) -> action id.
    -> (winner or None, list of (player, action))."""
    history = []
    done, winner = game.has_a_winner()
    while not done and (max_plies is None or len(history) < max_plies):
        me = game.current_player
        tic = time.time()
        action = players[me].choose_action(game)
        if action not in game.actions():
            raise ValueError("player %s (%s) chose the illegal action %r" % (me, getattr(players[me], "name", players[me]), action))
        log("player %s chose action %s, spent %.2f seconds" % (me, action, time.time() - tic))
        done, winner = step_result(game, action)
        history.append((me, int(action)))
        if on_move is not None:
            on_move(game, me, action)
    if done:
        log("game over! winner is player:%s" % winner)
    return (winner if done else None), history


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--player_type", type=int, default=1, help="1: human vs human, 2: human vs computer")
    ap.add_argument("--computer_type", type=int, default=0, help="1: AlphaZero MCTS, 2: pure MCTS")
    ap.add_argument("--model", default=None, help="ckpt/<name>.pth to load (default: random weights, like the reference)")
    ap.add_argument("--n_playout", type=int, default=30)
    args = ap.parse_args(argv)
    from .quoridor import Quoridor

    game = Quoridor()
    players = {1: ManualCLIAgent("Kurumi"), 2: ManualCLIAgent("Cryer")}
    if args.player_type == 2:
        if args.computer_type == 1:
            from .mcts import MCTSPlayer
            from .policy_value_net import PolicyValueNet

            players[2] = MCTSPlayer(PolicyValueNet(model_file=args.model).policy_value_fn, c_puct=5, n_playout=args.n_playout, is_selfplay=0)
        elif args.computer_type == 2:
            from .pure_mcts import MCTSPlayer as PureMCTSPlayer

            players[2] = PureMCTSPlayer(c_puct=5, n_playout=max(args.n_playout, 50))
        else:
            raise SystemExit("Set computer type to 1 or 2 for choosing computer!")
    t0 = time.time()
    play(game, players)
    print("total time :", time.time() - t0)


if __name__ == "__main__":
    main()
