"""GPU parity of the asynchronous self-play loop DIRECTLY against the reference fixtures and the CPU oracle.

tests/test_gpu_async.py compares the loop (k_moves / k_advance / k_round_tail) with the lock-step engine -- HIP against
HIP.  Here the same kernels are driven piece by piece through the C ABI

    qz_selfplay_advance -> qz_selfplay_leaf_rules -> qz_selfplay_misses + evaluations written by the CALLER -> qz_selfplay_round_tail

with the stub policies of tests/golden/_stubs.py (hash / uniform: the policies the reference-generated fixtures were
recorded with) or with the real network, and what comes out is compared with

  * tests/golden/mcts_stub.npz: 360 searches of the reference's MCTS (mcts.py:103-144): root visits, Q, P bit for bit, pi 1e-12;
  * tests/golden/episodes_stub.npz: complete reference games (quoridor.py:573-610): per-ply pi, carried root visits, z, tuples;
  * oracle.OracleMCTS (the C restatement of mcts.py) replaying whole games the loop played ON ITS OWN (moves sampled and
    re-rooted on the device: in place, compacting, compacting in 64-edge slices under a 1-us budget; memo on and off);
  * oracle.OracleMCTS fed with the REAL network's (p, v) as collected from the miss lists: 400-playout searches on late-game
    boards, visits and Q bit-equal over three plies.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

ORDER = list(range(12)) + [a for ix in range(64) for a in (12 + ix, 76 + ix)]


def stub_policy(name, packed, mask):
    """dense p [n,140] / v [n] of the fixture stubs for the boards of a miss list (mask = their legal sets)"""
    import oracle

    n = len(packed)
    if name == "hash":
        p, v = oracle.hash_policy_arrays(packed)
        return np.ascontiguousarray(p), np.ascontiguousarray(v)
    m = mask.view(np.uint32)
    cnt = np.zeros(n, dtype=np.int64)
    for w in range(5):
        cnt += np.array([bin(int(x)).count("1") for x in m[:, w]], dtype=np.int64)
    p = np.zeros((n, 140), dtype=np.float32)
    nz = cnt > 0
    p[nz] = (1.0 / cnt[nz]).astype(np.float32)[:, None]  # (float)(1.0 / n_legal), pure_mcts.py:13-16
    return p, np.zeros(n, dtype=np.float32)


def stub_round(eng, name, max_playouts, budget_us=0, auto_finish=False, seen=None):
    """one round of the loop with the CALLER as the evaluator; -> number of leaves it evaluated"""
    from alphazero_quoridor_amd import _cabi

    L = eng.L
    _cabi.check(L.qz_selfplay_advance(eng.h, int(max_playouts), int(budget_us), int(auto_finish), eng._s()))
    _cabi.check(L.qz_selfplay_leaf_rules(eng.h, eng._s()))
    packed, mask, _, _ = eng.misses()
    if len(packed):
        p, v = stub_policy(name, packed, mask)
        eng.set_miss_outputs(torch.from_numpy(p), torch.from_numpy(v))
        if seen is not None:
            seen.update(b.tobytes() for b in packed)
    _cabi.check(L.qz_selfplay_round_tail(eng.h, eng._s()))
    return len(packed)


def make_engine(boards, n_playout, **kw):
    from alphazero_quoridor_amd.boards import DeviceBoards
    from alphazero_quoridor_amd.engine import SelfPlayEngine

    kw.setdefault("c_puct", 5.0)
    kw.setdefault("temp", 1.0)
    eng = SelfPlayEngine(len(boards), n_playout=n_playout, device="cuda:0", **kw)
    eng.set_boards(DeviceBoards.from_packed(boards, eng.device), reset_trees=True)
    return eng


def search_to_completion(eng, name, n, max_playouts, budget_us, limit=100000):
    """stub rounds until every board has done its n playouts and none waits"""
    B = eng.n_boards
    rounds = evals = 0
    while True:
        evals += stub_round(eng, name, max_playouts, budget_us)
        rounds += 1
        st = eng.stats()
        if st["waiting_boards"] == 0 and st["playouts"] >= B * n:
            break
        assert rounds < limit, st
    return rounds, evals


# the loop's regimes: (memo, playouts a board may start per launch, wall-clock budget of a launch in us, select_opts)
# select_opts 4 = the 64-register build of k_advance (eight wavefronts per SIMD: what engines above 4,096 boards, i.e. the bench, run)
# 8 = one deadline per launch (counted from its first wavefront; budgets of 100 us and more) and boards taking the first slots in turn
REGIMES = [(True, 1, 0, 0), (True, 4096, 0, 0), (False, 4096, 0, 0), (True, 4096, 1, 0), (True, 4096, 0, 4), (True, 4096, 1, 4),
           (True, 4096, 120, 12), (True, 4096, 100, 8)]


@pytest.mark.parametrize("memo,max_playouts,budget_us,select_opts", REGIMES)
def test_async_route_reproduces_the_reference_search_fixture(gpu_device, golden_dir, memo, max_playouts, budget_us, select_opts):
    """mcts_stub.npz (360 searches recorded from the reference's MCTS.get_move_probs with stub policies, 8..400 playouts,
    c_puct 2.5 / 5, temperatures 1 / 0.5 / 1e-3, terminal leaves inside the trees) through k_advance: lock-step cadence,
    free-running (a board does all its playouts in as few launches as its misses allow), without the memo, and with a
    1-us budget (one playout per launch, every launch cut short).  Visits, float64 Q, float32 P bit for bit; pi 1e-12."""
    d = np.load(golden_dir + "/mcts_stub.npz")
    groups = {}
    for i in range(len(d["board"])):
        key = (str(d["policy"][i]), int(d["n_playout"][i]), float(d["c_puct"][i]), float(d["temp"][i]))
        groups.setdefault(key, []).append(i)
    checked = hits = 0
    for (pol, n, c_puct, temp), idx in groups.items():
        eng = make_engine(d["board"][idx], n, c_puct=c_puct, temp=temp, memo=memo, select_opts=select_opts)
        try:
            search_to_completion(eng, pol, n, max_playouts, budget_us)
            visits, q, prior, root_n = (t.cpu().numpy() for t in eng.root_children())
            pi, _ = eng.root_pi()
            pi = pi.cpu().numpy()
            st = eng.stats()
            assert st["playouts"] == len(idx) * n and st["node_overflow"] == 0 and st["miss_overflow"] == 0, st
            hits += st["memo_hits"]
            for j, i in enumerate(idx):
                k = int(d["k"][i])
                acts = d["acts"][i][:k].astype(int)
                assert [a for a in ORDER if visits[j, a] >= 0] == acts.tolist(), (pol, n, i)
                assert np.array_equal(visits[j, acts], d["visits"][i][:k]), (pol, n, i)
                assert np.array_equal(q[j, acts], d["q"][i][:k]), (pol, n, i)
                assert np.array_equal(prior[j, acts], d["p"][i][:k]), (pol, n, i)
                assert root_n[j] == d["root_visits"][i]
                assert np.allclose(pi[j, acts], d["probs"][i][:k], rtol=0, atol=1e-12)
                checked += 1
        finally:
            eng.close()
    assert checked == len(d["board"])
    assert (hits > 0) == memo, hits
    print("360 reference searches through k_advance (memo %s, %d playouts per launch, budget %d us): %d memo hits" % (memo, max_playouts, budget_us, hits))


@pytest.mark.parametrize("memo,max_playouts,budget_us", [(True, 4096, 0), (False, 1, 0), (True, 4096, 1)])
def test_async_route_replays_the_reference_episodes(gpu_device, golden_dir, memo, max_playouts, budget_us):
    """episodes_stub.npz: complete games of the reference's start_self_play (stub policy; its sampled moves are forced
    here).  Every ply's search runs through k_advance (boards stop at n_playout: auto_finish off), the move is
    qz_mcts_finish_move with the reference's move: per-ply pi (1e-12 in float64, 1e-6 as recorded), the root visits the
    kept subtree carries into the next ply, z and the harvested (board, pi, z) tuples."""
    from alphazero_quoridor_amd.boards import opening_packed

    d = np.load(golden_dir + "/episodes_stub.npz")
    order = sorted(range(int(d["n"])), key=lambda e: len(d["e%d_moves" % e]))[:4]
    for e in order:
        key = lambda k: d["e%d_%s" % (e, k)]  # noqa: E731
        moves, pis, z = key("moves"), key("pis"), key("z")
        n = int(key("n_playout"))
        pol = str(key("policy"))
        eng = make_engine(opening_packed(1), n, is_selfplay=1, max_plies=len(moves) + 1, memo=memo)
        try:
            for t in range(len(moves)):
                assert eng.get_boards().to_packed()[0].tobytes() == key("boards")[t].tobytes()
                for _ in range(10 * n + 10):
                    stub_round(eng, pol, max_playouts, budget_us)
                    st = eng.stats()
                    if st["playouts"] >= (t + 1) * n and st["waiting_boards"] == 0:
                        break
                assert st["playouts"] == (t + 1) * n, (st["playouts"], t, n)
                pi64, _ = eng.root_pi()
                assert np.allclose(pi64.cpu().numpy()[0], pis[t], rtol=0, atol=1e-12), (e, t)
                mv, pi32 = eng.finish_move(torch.tensor([int(moves[t])], dtype=torch.uint8))
                assert int(mv.cpu()[0]) == int(moves[t])
                assert np.allclose(pi32.cpu().numpy()[0], pis[t], rtol=0, atol=1e-6)
                if t < len(moves) - 1:
                    assert int(eng.root_children()[3].cpu()[0]) == int(key("root_n")[t]), (e, t)
            assert eng.pending() == (1, len(moves))
            tb = eng.harvest()
            assert tb.n_games == 1 and len(tb) == len(moves)
            assert np.array_equal(tb.z.cpu().numpy().astype(np.float64), z)
            assert np.array_equal(tb.boards.to_packed().view(np.uint64), key("boards").view(np.uint64))
            assert np.allclose(tb.pi.cpu().numpy(), pis, rtol=0, atol=1e-6)
        finally:
            eng.close()


def _move_between(og, cur_rec, nxt_rec):
    """the legal action that turns oracle game `og` (= packed board cur_rec) into the packed board nxt_rec (None: none does)"""
    dh, dv = int(cur_rec["hbits"]) ^ int(nxt_rec["hbits"]), int(cur_rec["vbits"]) ^ int(nxt_rec["vbits"])
    if dh:
        cand = [12 + dh.bit_length() - 1]
    elif dv:
        cand = [76 + dv.bit_length() - 1]
    else:
        cand = list(range(12))
    legal = set(og.actions())
    for a in cand:
        if a in legal:
            g = og.copy()
            g.step(a)
            if g.packed().tobytes() == nxt_rec.tobytes():
                return a
    return None


@pytest.mark.parametrize("name,compact_edges,budget_us,pool_pages,memo,select_opts", [
    ("hash", 0, 0, 0, True, 0),          # this small engine's default threshold (one page): compacting and in-place moves mixed
    ("hash", 0, 300, 64 * 60, True, 0),  # a large pool: moves in place, under a wall-clock budget
    ("hash", -1, 1, 0, True, 0),         # every move copies its subtree; 1-us budget: the copies proceed in 64-edge slices
    ("uniform", 0, 0, 0, False, 0),      # no memo: every leaf through the caller
    ("hash", 0, 0, 0, True, 4),          # the 64-register build of k_advance (what the bench's 8,192-board engine runs)
    ("hash", -1, 1, 0, True, 4),
    ("hash", -1, 150, 0, True, 12),      # one deadline per launch, boards rotate through the first slots
])
def test_games_the_loop_plays_on_its_own_replay_in_the_oracle(gpu_device, name, compact_edges, budget_us, pool_pages, memo, select_opts):
    """The free-running loop (auto_finish: k_moves samples, records, steps and re-roots on the device; k_advance resumes
    sliced subtree copies; the memo answers repeated leaves) with the caller as the evaluator, short games (terminal sign
    fixed, 24 playouts).  Every harvested game is then replayed ply by ply in oracle.OracleMCTS (mcts.py:103-151 restated
    in C) with the same stub policy and the moves the device sampled: at every ply the oracle's pi -- i.e. its root visit
    counts after n_playout playouts on the carried subtree -- must equal the pi the device recorded (float32: 1e-6; at
    temp = 1 pi is visits / sum, so every count is pinned), the recorded boards must follow from the oracle's game by a
    legal move, and z must follow quoridor.py:599-602."""
    import oracle
    from alphazero_quoridor_amd.boards import opening_packed

    B, NP = 64, 24
    eng = make_engine(opening_packed(B), NP, seed=5, fix_terminal_sign=True, compact_edges=compact_edges, tree_pool_pages=pool_pages, memo=memo, select_opts=select_opts)
    batches = []
    try:
        rounds = 0
        while sum(t.n_games for t in batches) < 40 and rounds < 60000:
            for _ in range(8):
                stub_round(eng, name, NP + 8, budget_us, auto_finish=True)
            rounds += 8
            tb = eng.harvest()
            if tb is not None:
                batches.append(tb)
        st = eng.stats()
        # (a root without a legal move ends a game in the reference too -- test_roots_without_a_legal_move_are_dropped_and_counted;
        # such games are dropped, never harvested: every harvested game below still replays in the oracle)
        assert st["node_overflow"] == 0 and st["runaway_descents"] == 0 and st["miss_overflow"] == 0, st
        assert st["games_aborted"] == st["aborted_no_move"], st
        assert (st["memo_hits"] > 0) == memo
        if budget_us == 1:
            assert st["compact_slices"] > 0, st
    finally:
        eng.close()
    games = plies = 0
    for tb in batches:
        gid = tb.game.cpu().numpy()
        packed = tb.boards.to_packed()
        pi = tb.pi.cpu().numpy()
        z = tb.z.cpu().numpy()
        for g in range(tb.n_games):
            rows = np.nonzero(gid == g)[0]
            og = oracle.OracleGame()
            tree = oracle.OracleMCTS(name, c_puct=5, n_playout=NP, fix_terminal_sign=True)
            for t, r in enumerate(rows):
                assert og.packed().tobytes() == packed[r].tobytes(), (g, t)
                acts, visits, probs = tree.get_move_probs(og, 1.0)
                want = np.zeros(140)
                want[acts] = probs
                assert np.allclose(pi[r], want, rtol=0, atol=1e-6), (g, t, visits)
                if t + 1 < len(rows):
                    mv = _move_between(og, packed[r], packed[rows[t + 1]])
                    assert mv is not None and pi[r][mv] > 0, (g, t)
                    tree.update_with_move(mv)  # mcts.py:146-151
                    assert og.step(mv) is False
                    plies += 1
            # the last recorded position's mover made the winning move (the reference does not rotate on it): z = +1 for it
            w = og.get_current_player()
            assert any(_wins(og, a) for a in og.actions())
            assert np.array_equal(z[rows], np.where(packed[rows]["cur"] == w, 1.0, -1.0))
            games += 1
    assert games >= 40 and plies > 1000
    print("%d games / %d plies the loop played on its own replayed in the oracle (%s, compact_edges %d, budget %d us, memo %s): "
          "%d in-loop evaluations, %d memo hits, %d sliced copies" % (games, plies, name, compact_edges, budget_us, memo, st["nn_evals"], st["memo_hits"], st["compact_slices"]))


def _wins(og, a):
    g = og.copy()
    return bool(g.step(a))


@pytest.mark.parametrize("select_opts", [0, 4])
def test_real_network_search_equals_the_oracle_fed_with_the_miss_list_evaluations(gpu_device, select_opts):
    """The kernel the bench times, with the real network and the reference's terminal sign: 64 late-game boards (movers
    without walls: the memo's regime), 400 playouts per move, three plies.  Every (board -> p, v) the network produced is
    collected from the miss lists; oracle.OracleMCTS with THAT table as its policy must arrive at bit-equal root visits and
    float64 Q (priors bit-equal) after every ply -- the memo, the replayed descents and the backups of k_advance against
    the pointer-tree restatement of mcts.py:103-151."""
    import ctypes as C

    import oracle
    from _stubs import det_fill_state_dict
    from alphazero_quoridor_amd import _cabi
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet
    from synth import synth_positions

    net = PolicyValueNet(use_gpu=True, device=gpu_device)
    net.policy_value_net.load_state_dict(det_fill_state_dict(net.policy_value_net.state_dict(), 2024))
    ev = net.evaluator("per_leaf")
    assert ev.engine_route_ok()
    B, NP = 64, 400
    boards = synth_positions(B, seed=77, max_walls=10)
    boards["w1"] = 0
    boards["w2"] = 0
    boards = boards[[not oracle.OracleGame.from_packed(b).has_a_winner()[0] for b in boards]]
    B = len(boards)
    eng = make_engine(boards, NP, seed=3, select_opts=select_opts)  # (4: the 64-register build of k_advance, the bench's)
    L = eng.L
    table = {}

    def policy(g, legal):
        p, v = table[g.packed().tobytes()]
        return legal, p[legal], float(v)

    trees = [oracle.OracleMCTS(policy, c_puct=5, n_playout=NP) for _ in range(B)]
    games = [oracle.OracleGame.from_packed(b) for b in boards]
    alive = np.ones(B, dtype=bool)
    try:
        for ply in range(3):
            target = eng.stats()["playouts"] + int(alive.sum()) * NP
            for _ in range(100000):
                eng._memo_guard(ev)
                _cabi.check(L.qz_selfplay_advance(eng.h, 4096, 200, 0, eng._s()))
                _cabi.check(L.qz_selfplay_leaf_rules(eng.h, eng._s()))
                _cabi.check(L.qz_selfplay_evaluate(eng.h, C.byref(ev.nn_weights()), eng._s()))
                packed, mask, p, v = eng.misses()
                for i in range(len(packed)):
                    table[packed[i].tobytes()] = (p[i].copy(), v[i])
                _cabi.check(L.qz_selfplay_round_tail(eng.h, eng._s()))
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
                a2, v2, q2, p2 = trees[j].root_children()
                assert [a for a in ORDER if visits[j, a] >= 0] == acts, (ply, j)
                if not acts:  # a root without a legal move (the reference crashes there): the engine drops the game
                    alive[j] = False
                    continue
                assert np.array_equal(visits[j, acts], ov), (ply, j, visits[j, acts], ov)
                assert np.array_equal(q[j, acts], q2), (ply, j)
                assert np.array_equal(prior[j, acts], p2), (ply, j)
                assert root_n[j] == trees[j].root_visits()
                mv = acts[int(np.argmax(ov))]  # the most visited move (first on ties) on both sides
                forced[j] = mv
                trees[j].update_with_move(mv)
                if games[j].step(mv):
                    alive[j] = False
            eng.finish_move(torch.from_numpy(forced))
            eng.harvest()
        st = eng.stats()
        assert st["node_overflow"] == 0 and st["miss_overflow"] == 0 and st["memo_hits"] > st["nn_evals"], st
        print("3 plies x 400 playouts on %d late-game boards: %d network evaluations, %d memo hits, deepest descent %d levels; "
              "visits and Q bit-equal with the oracle" % (B, st["nn_evals"], st["memo_hits"], st["max_depth"]))
    finally:
        eng.close()


@pytest.mark.parametrize("select_opts", [0, 4, 32])
def test_reference_search_with_its_own_network_through_the_loop(gpu_device, golden_dir, select_opts):
    """tests/golden/real_net_search.npz (gen_golden.py:gen_real_net_search): 80 searches of the REFERENCE's MCTS
    at 400 playouts with the reference's own PolicyValueNet as the policy -- under torch >= 0.4 its leaf value is a 0-dim float32
    tensor (policy_value_net.py:163), TreeNode._Q turns into one (mcts.py:53) and Q + u is compared in float32, where the build
    follows the code as written for torch 0.3 (float64).  The loop -- k_advance's two builds, and k_rows for the 64 boards without
    walls (select_opts 32) -- is fed, through the miss list, EXACTLY the (p, v) the reference's network returned for each board
    (a leaf the reference never evaluated would be a search that went another way: KeyError), memo on: root visit counts and
    root visits must equal the reference's on all 80.  (The oracle passes the same check on the CPU: tests/test_oracle_golden.py.)"""
    from alphazero_quoridor_amd import _cabi
    from test_oracle_golden import real_net_tables

    d = np.load(golden_dir + "/real_net_search.npz")
    tabs = real_net_tables(d)
    n, NP = len(d["board"]), int(d["n_playout"])
    table = {}
    for t in tabs:
        for k, (p, v, _) in t.items():
            if k in table:  # the evaluation is a pure function of the board: searches that meet the same board got the same answer
                assert np.array_equal(table[k][0], p) and table[k][1] == v
            table[k] = (p, v)
    eng = make_engine(d["board"], NP, c_puct=float(d["c_puct"]), select_opts=select_opts)
    L = eng.L
    try:
        for _ in range(100000):
            _cabi.check(L.qz_selfplay_advance(eng.h, 4096, 0, 0, eng._s()))
            _cabi.check(L.qz_selfplay_leaf_rules(eng.h, eng._s()))
            packed, mask, _, _ = eng.misses()
            if len(packed):
                p = np.stack([table[b.tobytes()][0] for b in packed])
                v = np.array([table[b.tobytes()][1] for b in packed], dtype=np.float32)
                eng.set_miss_outputs(torch.from_numpy(p), torch.from_numpy(v))
            _cabi.check(L.qz_selfplay_round_tail(eng.h, eng._s()))
            st = eng.stats()
            if st["waiting_boards"] == 0 and st["playouts"] >= n * NP:
                break
        assert st["playouts"] == n * NP and st["node_overflow"] == 0 and st["miss_overflow"] == 0, st
        visits, q, prior, root_n = (t.cpu().numpy() for t in eng.root_children())
        for i in range(n):
            k = int(d["k"][i])
            acts = d["acts"][i][:k].astype(int)
            assert [a for a in ORDER if visits[i, a] >= 0] == acts.tolist(), i
            assert np.array_equal(visits[i, acts], d["visits"][i][:k]), (i, visits[i, acts], d["visits"][i][:k])
            assert root_n[i] == d["root_visits"][i]
            assert np.abs(q[i, acts] - d["q32"][i][:k]).max() < 1e-5  # the reference's float32 running means against float64 ones
        print("80 searches of the reference with its own network through the loop (select_opts %d): root visits identical; %d evaluations, %d memo hits"
              % (select_opts, st["nn_evals"], st["memo_hits"]))
    finally:
        eng.close()


def test_a_leftover_of_eager_rounds_does_not_break_a_captured_graph(gpu_device):
    """a HIP graph of rounds bakes in which of the two miss counters its first round uses.  An odd number of
    eager rounds between two replays (a leftover of run_rounds, the n_playout + 1 rounds of run_playouts_memo) must not
    make the next replay append to a counter nobody cleared: capture_rounds(4) then run_rounds(5) twice equals eager."""
    from _stubs import det_fill_state_dict
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet
    from synth import synth_positions

    net = PolicyValueNet(use_gpu=True, device=gpu_device)
    net.policy_value_net.load_state_dict(det_fill_state_dict(net.policy_value_net.state_dict(), 6))
    ev = net.evaluator("per_leaf")
    boards = synth_positions(512, seed=13, max_walls=12, mover_has_walls=True)  # opening-phase boards: nearly every board misses every round
    # (no memo: the test compares two engines after the same NUMBER OF ROUNDS, and with the memo the round in which a board gets an
    # evaluation can depend on timing -- two inserters of one bucket in one tail: the loser skips, it is a cache -- though never
    # WHAT the board computes; the miss counters this test is about do not care)
    g = make_engine(boards, 16, seed=2, memo=False)
    e = make_engine(boards, 16, seed=2, memo=False)
    try:
        n0 = g.capture_rounds(ev, rounds=4, max_playouts=4, warmup=2)
        e.run_rounds(ev, n0, max_playouts=4)   # the eager rounds capture_rounds has run (warm-up + one between its two captures)
        for _ in range(2):
            g.run_rounds(ev, 5, max_playouts=4)   # 4 replayed + 1 eager: the engine is left on the other counter
            e.run_rounds(ev, 5, max_playouts=4)
        assert g.graph_replays == 2               # ... and the second call replayed the OTHER parity's graph
        g.run_rounds(ev, 9, max_playouts=4)       # 2 x 4 replayed + 1 eager
        e.run_rounds(ev, 9, max_playouts=4)
        g.run_rounds(ev, 4, max_playouts=4)       # n == the graph's rounds on an odd leftover (TrainPipeline's shape): still a replay
        e.run_rounds(ev, 4, max_playouts=4)
        assert g.graph_replays == 5
        for x, y in zip(g.root_children(), e.root_children()):
            assert torch.equal(x, y)
        assert torch.equal(g.get_boards().meta, e.get_boards().meta)
        sg, se = g.stats(), e.stats()
        assert sg["miss_overflow"] == 0 and sg["playouts"] == se["playouts"] and sg["nn_evals"] == se["nn_evals"] and sg["plies_played"] == se["plies_played"]
        # a reset between the pieces of a round leaves no stale counter behind
        from alphazero_quoridor_amd import _cabi
        # an engine whose boards play their moves on the device refuses launches WITHOUT the moves' kernel: a board with a
        # pending subtree copy sits out of k_advance and would never be continued
        with pytest.raises(_cabi.QzError):
            _cabi.check(g.L.qz_selfplay_advance(g.h, 4, 0, 0, g._s()))
        with pytest.raises(_cabi.QzError):
            g.selfplay_round(ev, 4, 0, auto_finish=False)
        _cabi.check(g.L.qz_selfplay_advance(g.h, 4, 0, 1, g._s()))
        g.reset()
        assert g.round_parity() == 0
        g.run_rounds(ev, 6, max_playouts=4)
        assert g.stats()["miss_overflow"] == 0 and g.stats()["waiting_boards"] <= g.n_boards
    finally:
        g.close()
        e.close()


@pytest.mark.parametrize("route", ["lockstep", "async"])
def test_roots_without_a_legal_move_are_dropped_and_counted(gpu_device, golden_dir, route):
    """tests/golden/no_move_roots.npz: 65 root positions a 4-playout self-play run dropped under aborted_no_move.  The
    REFERENCE returns [] from actions() on every one, prints "WARNING: the board is full" and returns None from
    choose_action (mcts.py:195-196), and start_self_play's unpack of that None raises (quoridor.py:587): its run ends
    (fixture columns, gen_golden.py:gen_no_move).  Here: qz_movegen gives an empty legal set on each (so does the oracle);
    an engine whose roots are those boards, interleaved with live middle-game boards, searches them (the root can never be
    expanded), then drops EXACTLY those games -- counted under aborted_no_move, listed in the drop log with the root
    position, nothing harvested from them -- restarts the slots from the opening, and plays the live boards on."""
    import oracle
    from alphazero_quoridor_amd import rules
    from alphazero_quoridor_amd.boards import DeviceBoards, opening_packed
    from synth import synth_positions

    d = np.load(golden_dir + "/no_move_roots.npz")
    stuck = d["board"]
    n = len(stuck)
    assert n >= 20 and (d["n_actions"] == 0).all() and d["prints_board_is_full"].all() and d["unpack_raises_typeerror"].all()
    mask = rules.movegen(DeviceBoards.from_packed(stuck, gpu_device)).cpu().numpy()
    omask, status = oracle.movegen_batch(stuck)
    assert not mask.any() and not omask.any() and (status >= 0).all()
    live = synth_positions(n, seed=31, max_walls=10, mover_has_walls=True)
    live = live[[len(oracle.OracleGame.from_packed(b).actions()) > 0 and not oracle.OracleGame.from_packed(b).has_a_winner()[0] for b in live]]
    boards = np.concatenate([stuck, live])
    is_stuck = np.arange(len(boards)) < n
    NP = 6
    eng = make_engine(boards, NP, seed=8)
    try:
        if route == "lockstep":
            for _ in range(NP):
                leaf = eng.select_boards()
                p, v = stub_policy("hash", leaf.to_packed(), eng.leaf_mask.cpu().numpy())
                eng.expand_backup(torch.from_numpy(p).to(eng.device), torch.from_numpy(v).to(eng.device))
            visits, _, _, root_n = (t.cpu().numpy() for t in eng.root_children())
            assert (root_n == NP).all() and (visits[is_stuck] < 0).all() and (visits[~is_stuck] >= 0).any(axis=1).all()
            moves, pi = eng.finish_move()
            moves = moves.cpu().numpy()
            assert (moves[is_stuck] == 255).all() and (moves[~is_stuck] != 255).all() and not pi.cpu().numpy()[is_stuck].any()
        else:
            for _ in range(10 * NP):
                stub_round(eng, "hash", NP + 2, 0, auto_finish=True)
                st = eng.stats()
                if st["aborted_no_move"] >= n and st["plies_played"] >= len(boards) - n:
                    break
        st = eng.stats()
        assert st["aborted_no_move"] == n and st["games_aborted"] == n, st
        assert st["plies_played"] >= len(boards) - n  # the live boards made their moves
        packed, causes, plies, slots, total = eng.dropped_games()
        assert total == n and set(causes) == {"no_legal_move"} and (plies == 0).all()
        assert sorted(slots.tolist()) == list(range(n))
        for rec, slot in zip(packed, slots):
            assert rec.tobytes() == stuck[slot].tobytes()
        # nothing of the dropped games is ever harvested
        tb = eng.harvest()
        assert tb is None or (tb.slot.cpu().numpy() >= n).all()
        if route == "lockstep":  # the move's launch pair hands a dropped board's pages back and restarts its slot from the opening
            now = eng.get_boards().to_packed()
            assert all(now[j].tobytes() == opening_packed(1)[0].tobytes() for j in range(n))
            assert all(now[j].tobytes() != boards[j].tobytes() for j in range(n, len(boards)))
    finally:
        eng.close()
