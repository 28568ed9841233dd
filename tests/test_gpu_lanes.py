"""GPU parity of the other mapping of the asynchronous loop's kernel, for the boards on which neither player has a wall left:
k_rows (csrc/qz_rows.h, qz_config.select_opts bit 5: SIXTEEN LANES per board, four boards per wavefront -- k_advance's algorithm with
what was wave-uniform held row-uniform) -- against oracle.OracleMCTS (the C restatement of mcts.py:12-151) and against the
wavefront-per-board kernel k_advance, which is pinned on the reference's fixtures by test_gpu_async_oracle.py.

(Round 6 also built k_lanes -- ONE LANE per board, the backup of playout i folded into the descent of playout i + 1 -- and ran this
whole file on it, green; it measured four times slower than k_advance and was removed: profiles/round6/SUMMARY.md 1, the source in
profiles/round6/ab/qz_lanes.h.removed.txt.)

The tests pin root visits, float64 Q and float32 P after every ply, bit for bit, in every launch regime (one playout per launch,
free-running, 1-us budgets that cut every launch short, budgets in between), with stub policies that put terminal leaves inside the
trees, and with the real network through the memo."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

from test_gpu_async_oracle import ORDER, make_engine, stub_round  # noqa: E402

ROWS = 32   # qz_config.select_opts bit 5: k_rows
MAPPINGS = [ROWS]


def _late_boards(n, seed, near_goal=False):
    """live boards on which nobody has a wall left; near_goal: pawns within three rows of their goals (terminal leaves inside the trees)"""
    import oracle
    from synth import synth_positions

    b = synth_positions(2 * n, seed=seed, max_walls=14)
    b["w1"] = 0
    b["w2"] = 0
    if near_goal:
        rng = np.random.RandomState(seed)
        b["p1"] = rng.randint(45, 72, size=len(b))   # player 1 walks north to row 8
        b["p2"] = rng.randint(9, 36, size=len(b))    # player 2 south to row 0
    keep = []
    for r in b:
        g = oracle.OracleGame.from_packed(r)
        if not g.has_a_winner()[0] and len(g.actions()) > 0 and int(r["p1"]) != int(r["p2"]):
            keep.append(r)
    return np.array(keep[:n], dtype=b.dtype)


def _search_plies(boards, name, n_playout, max_playouts, budget_us, memo, select_opts, plies, fix_sign=False, c_puct=5.0):
    """`plies` searches with forced moves (the most visited, first on ties) on the device and in the oracle: bit-equal after every ply"""
    import oracle

    B = len(boards)
    eng = make_engine(boards, n_playout, c_puct=c_puct, memo=memo, select_opts=select_opts, seed=11, fix_terminal_sign=fix_sign)
    trees = [oracle.OracleMCTS(name, c_puct=c_puct, n_playout=n_playout, fix_terminal_sign=fix_sign) for _ in range(B)]
    games = [oracle.OracleGame.from_packed(b) for b in boards]
    alive = np.ones(B, dtype=bool)
    try:
        for ply in range(plies):
            target = eng.stats()["playouts"] + int(alive.sum()) * n_playout
            for _ in range(200000):
                stub_round(eng, name, max_playouts, budget_us)
                st = eng.stats()
                if st["waiting_boards"] == 0 and st["playouts"] >= target:
                    break
            assert st["playouts"] == target, (st["playouts"], target)
            visits, q, prior, root_n = (t.cpu().numpy() for t in eng.root_children())
            forced = np.full(B, 255, dtype=np.uint8)
            for j in range(B):
                if not alive[j]:
                    continue
                acts, ov, _ = trees[j].get_move_probs(games[j], 1.0)
                _, _, q2, p2 = trees[j].root_children()
                assert [a for a in ORDER if visits[j, a] >= 0] == acts, (ply, j)
                assert np.array_equal(visits[j, acts], ov), (ply, j, visits[j, acts], ov)
                assert np.array_equal(q[j, acts], q2), (ply, j, q[j, acts], q2)
                assert np.array_equal(prior[j, acts], p2), (ply, j)
                assert root_n[j] == trees[j].root_visits(), (ply, j)
                mv = acts[int(np.argmax(ov))]
                forced[j] = mv
                trees[j].update_with_move(mv)
                if games[j].step(mv):
                    alive[j] = False
            eng.finish_move(torch.from_numpy(forced))  # (no harvest: a finished game's slot stays idle instead of restarting from the opening)
        st = eng.stats()
        assert st["node_overflow"] == 0 and st["miss_overflow"] == 0 and st["runaway_descents"] == 0, st
        return st
    finally:
        eng.close()


# (memo, playouts a board may start per launch, budget of a launch in us)
LANE_REGIMES = [(True, 1, 0), (True, 4096, 0), (False, 4096, 0), (True, 4096, 1), (True, 3, 0), (True, 4096, 40)]


@pytest.mark.parametrize("mapping", MAPPINGS)
@pytest.mark.parametrize("memo,max_playouts,budget_us", LANE_REGIMES)
def test_lane_kernel_search_equals_the_oracle(gpu_device, memo, max_playouts, budget_us, mapping):
    """200 late-game boards (no walls left: every board is k_rows'), hash stub policy, 60 playouts per move, four plies with
    the kept subtrees carried over (in-place re-roots): visits, Q, P, root visits bit-equal with oracle.OracleMCTS."""
    boards = _late_boards(200, seed=5)
    st = _search_plies(boards, "hash", 60, max_playouts, budget_us, memo, mapping, plies=4)
    assert (st["memo_hits"] > 0) == memo
    print("select_opts %d, %d boards x 4 plies x 60 playouts (memo %s, %d per launch, %d us): %d evaluations, %d memo hits, deepest %d"
          % (mapping, len(boards), memo, max_playouts, budget_us, st["nn_evals"], st["memo_hits"], st["max_depth"]))


@pytest.mark.parametrize("mapping", MAPPINGS)
@pytest.mark.parametrize("fix_sign", [False, True])
def test_lane_kernel_terminal_leaves_inside_the_trees(gpu_device, fix_sign, mapping):
    """pawns close to their goal rows: winning moves are in reach of the search, so terminal leaves are backed up with the
    reference's sign (mcts.py:119-126: +1 for the side that did NOT move) or the fixed one; 150 playouts, three plies, both
    kernels side by side (select_opts 0: k_advance) with identical counters."""
    boards = _late_boards(160, seed=9, near_goal=True)
    a = _search_plies(boards, "hash", 150, 4096, 0, True, mapping, plies=3, fix_sign=fix_sign)
    b = _search_plies(boards, "hash", 150, 4096, 0, True, 0, plies=3, fix_sign=fix_sign)
    assert a["leaf_terminal"] > 0, a
    for k in ("playouts", "leaf_terminal", "descent_levels", "edges_expanded", "max_depth", "edges_scanned"):
        assert a[k] == b[k], (k, a[k], b[k])
    print("terminal leaves inside the trees (sign fixed: %s): %d of %d playouts ended on one; counters equal k_advance's" % (fix_sign, a["leaf_terminal"], a["playouts"]))


@pytest.mark.parametrize("mapping", MAPPINGS)
def test_lane_kernel_uniform_policy_and_deep_trees(gpu_device, mapping):
    """the uniform stub (pure_mcts.py:13-16: equal priors, value 0 -- every comparison a tie broken by the first maximum) at 400
    playouts, c_puct 2.5"""
    boards = _late_boards(64, seed=21)
    st = _search_plies(boards, "uniform", 400, 4096, 0, True, mapping, plies=2, c_puct=2.5)
    print("uniform stub, 400 playouts: deepest descent %d levels" % st["max_depth"])


@pytest.mark.parametrize("mapping", MAPPINGS)
def test_boards_cross_from_the_wavefront_kernel_to_the_lane_kernel(gpu_device, mapping):
    """Games the loop plays ON ITS OWN (moves sampled on the device, subtrees kept in place or compacted) from positions where
    the players hold one or two walls between them: the boards start on k_advance and move to k_rows with the ply that places the
    last wall, their trees and pending state as they are.  Every harvested game replays in oracle.OracleMCTS ply by ply (pi pins
    every visit count); games restarted from the opening stay on k_advance."""
    import oracle
    from synth import synth_positions

    NP = 24
    b = synth_positions(400, seed=33, max_walls=14)
    rng = np.random.RandomState(4)
    b["w1"] = rng.randint(0, 2, size=len(b))
    b["w2"] = rng.randint(0, 2, size=len(b))
    keep = [r for r in b if not oracle.OracleGame.from_packed(r).has_a_winner()[0] and len(oracle.OracleGame.from_packed(r).actions()) > 0]
    boards = np.array(keep[:96], dtype=b.dtype)
    B = len(boards)
    eng = make_engine(boards, NP, seed=5, fix_terminal_sign=True, select_opts=mapping, is_selfplay=1)
    batches = []
    try:
        rounds = 0
        while sum(t.n_games for t in batches) < 60 and rounds < 40000:
            for _ in range(8):
                stub_round(eng, "hash", NP + 8, 0, auto_finish=True)
            rounds += 8
            tb = eng.harvest()
            if tb is not None:
                batches.append(tb)
        st = eng.stats()
        assert st["node_overflow"] == 0 and st["runaway_descents"] == 0 and st["miss_overflow"] == 0, st
    finally:
        eng.close()
    games = plies = late_plies = 0
    for tb in batches:
        gid = tb.game.cpu().numpy()
        packed = tb.boards.to_packed()
        pi = tb.pi.cpu().numpy()
        for g in range(tb.n_games):
            rows = np.nonzero(gid == g)[0]
            og = oracle.OracleGame.from_packed(packed[rows[0]])
            tree = oracle.OracleMCTS("hash", c_puct=5, n_playout=NP, fix_terminal_sign=True)
            for t, r in enumerate(rows):
                assert og.packed().tobytes() == packed[r].tobytes(), (g, t)
                acts, visits, probs = tree.get_move_probs(og, 1.0)
                want = np.zeros(140)
                want[acts] = probs
                assert np.allclose(pi[r], want, rtol=0, atol=1e-6), (g, t, visits)
                late_plies += int(packed[r]["w1"] == 0 and packed[r]["w2"] == 0)
                if t + 1 < len(rows):
                    nxt = oracle.OracleGame.from_packed(packed[rows[t + 1]])
                    mv = None
                    for a in acts:
                        g2 = og.copy()
                        if not g2.step(a) and g2.packed().tobytes() == nxt.packed().tobytes():
                            mv = a
                            break
                    assert mv is not None and pi[r][mv] > 0, (g, t)
                    tree.update_with_move(mv)
                    assert og.step(mv) is False
                    plies += 1
            games += 1
    assert games >= 60 and late_plies > 200, (games, plies, late_plies)
    print("%d games / %d plies replayed in the oracle, %d plies of them on boards without walls (k_rows)" % (games, plies, late_plies))


@pytest.mark.parametrize("mapping", MAPPINGS)
def test_lane_kernel_real_network_search_equals_the_oracle(gpu_device, mapping):
    """test_gpu_async_oracle's real-network search (64 late-game boards, 400 playouts, three plies, evaluations collected from the
    miss lists and fed to oracle.OracleMCTS) on k_rows."""
    import test_gpu_async_oracle as T

    T.test_real_network_search_equals_the_oracle_fed_with_the_miss_list_evaluations(gpu_device, mapping)
