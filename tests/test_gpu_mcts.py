"""GPU parity: the HIP tree kernels (select / expand / backup / re-root / finish_move /
harvest, through the C ABI) vs the golden MCTS fixtures recorded from the real reference and
vs the CPU oracle on the same seeded inputs.

Visit counts, float64 Q and float32 priors are compared bit for bit.  pi goes through
device exp/log (vs numpy's), so it is compared at 1e-12 (float64 path) / 1e-6 (float32 copy
kept in the trajectory)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ORDER = list(range(12)) + [a for ix in range(64) for a in (12 + ix, 76 + ix)]


def stub_policy(name, leaf_packed, leaf_mask):
    """Dense [B,140] priors + [B] values of the test stubs for a batch of leaf boards."""
    import oracle

    B = len(leaf_packed)
    if name == "hash":
        p, v = oracle.hash_policy_arrays(leaf_packed)
        return np.ascontiguousarray(p), np.ascontiguousarray(v)
    m = leaf_mask.view(np.uint32)
    cnt = np.zeros(B, dtype=np.int64)
    for w in range(5):
        cnt += np.array([bin(int(x)).count("1") for x in m[:, w]])
    p = np.zeros((B, 140), dtype=np.float32)
    nz = cnt > 0
    p[nz] = (1.0 / cnt[nz]).astype(np.float32)[:, None]
    return p, np.zeros(B, dtype=np.float32)


def run_playouts(eng, name, n):
    for _ in range(n):
        leaf = eng.select_boards()
        p, v = stub_policy(name, leaf.to_packed(), eng.leaf_mask.cpu().numpy())
        eng.expand_backup(torch.from_numpy(p).to(eng.device), torch.from_numpy(v).to(eng.device))


def make_engine(boards, n_playout, c_puct=5.0, temp=1.0, **kw):
    from alphazero_quoridor_amd.boards import DeviceBoards
    from alphazero_quoridor_amd.engine import SelfPlayEngine

    eng = SelfPlayEngine(len(boards), n_playout=n_playout, c_puct=c_puct, temp=temp, device="cuda:0", **kw)
    eng.set_boards(DeviceBoards.from_packed(boards, eng.device), reset_trees=True)
    return eng


def test_visit_counts_match_reference_fixture(gpu_device, golden_dir):
    d = np.load(golden_dir + "/mcts_stub.npz")
    groups = {}
    for i in range(len(d["board"])):
        key = (str(d["policy"][i]), int(d["n_playout"][i]), float(d["c_puct"][i]), float(d["temp"][i]))
        groups.setdefault(key, []).append(i)
    checked = 0
    for (pol, n, c_puct, temp), idx in groups.items():
        eng = make_engine(d["board"][idx], n, c_puct=c_puct, temp=temp)
        run_playouts(eng, pol, n)
        visits, q, prior, root_n = (t.cpu().numpy() for t in eng.root_children())
        pi, _ = eng.root_pi()
        pi = pi.cpu().numpy()
        for j, i in enumerate(idx):
            k = int(d["k"][i])
            acts = d["acts"][i][:k].astype(int)
            assert [a for a in ORDER if visits[j, a] >= 0] == acts.tolist()
            assert np.array_equal(visits[j, acts], d["visits"][i][:k]), (pol, n, i)
            assert np.array_equal(q[j, acts], d["q"][i][:k]), (pol, n, i)
            assert np.array_equal(prior[j, acts], d["p"][i][:k])
            assert root_n[j] == d["root_visits"][i]
            assert np.allclose(pi[j, acts], d["probs"][i][:k], rtol=0, atol=1e-12)
            checked += 1
        eng.close()
    assert checked == len(d["board"])


def test_tree_vs_oracle_with_reuse(gpu_device):
    """256 seeded positions, 3 plies of 48 playouts with subtree reuse, hash stub: every
    root statistic equals the pointer-based oracle's after every ply."""
    import oracle
    from synth import synth_positions

    boards = synth_positions(256, seed=4242)
    n = 48
    eng = make_engine(boards, n)
    trees = [oracle.OracleMCTS("hash", c_puct=5, n_playout=n) for _ in boards]
    games = [oracle.OracleGame.from_packed(b) for b in boards]
    alive = np.ones(len(boards), dtype=bool)
    for ply in range(3):
        run_playouts(eng, "hash", n)
        visits, q, prior, root_n = (t.cpu().numpy() for t in eng.root_children())
        forced = np.full(len(boards), 255, dtype=np.uint8)
        for j in range(len(boards)):
            if not alive[j]:
                continue
            acts, ov, _ = trees[j].get_move_probs(games[j], 1.0)
            a2, v2, q2, p2 = trees[j].root_children()
            assert [a for a in ORDER if visits[j, a] >= 0] == acts, (ply, j)
            assert np.array_equal(visits[j, acts], ov) and np.array_equal(q[j, acts], q2)
            assert root_n[j] == trees[j].root_visits()
            # play the most visited move (first on ties) on both sides
            mv = acts[int(np.argmax(ov))]
            forced[j] = mv
            trees[j].update_with_move(mv)
            if games[j].step(mv):
                alive[j] = False
        moves, _ = eng.finish_move(torch.from_numpy(forced))
        assert np.array_equal(moves.cpu().numpy()[alive | (forced != 255)], forced[alive | (forced != 255)])
        got = eng.get_boards().to_packed()
        for j in range(len(boards)):
            if alive[j]:
                assert got[j].tobytes() == games[j].packed().tobytes()
    st = eng.stats()
    assert st["node_overflow"] == 0
    eng.close()


def test_episode_traces_match_reference(gpu_device, golden_dir):
    """Full self-play games recorded from the reference (stub policy, its sampled moves
    replayed as forced moves): per-ply pi, root visit carry-over, z and the harvested
    tuples."""
    d = np.load(golden_dir + "/episodes_stub.npz")
    from alphazero_quoridor_amd.boards import opening_packed

    order = sorted(range(int(d["n"])), key=lambda e: len(d["e%d_moves" % e]))[:4]
    for e in order:
        key = lambda k: d["e%d_%s" % (e, k)]  # noqa: E731
        moves, pis, players, z = key("moves"), key("pis"), key("players"), key("z")
        n = int(key("n_playout"))
        eng = make_engine(opening_packed(1), n, is_selfplay=1, max_plies=len(moves) + 1)
        for t in range(len(moves)):
            assert eng.get_boards().to_packed()[0].tobytes() == key("boards")[t].tobytes()
            run_playouts(eng, str(key("policy")), n)
            pi64, _ = eng.root_pi()
            assert np.allclose(pi64.cpu().numpy()[0], pis[t], rtol=0, atol=1e-12), (e, t)
            mv, pi32 = eng.finish_move(torch.tensor([int(moves[t])], dtype=torch.uint8))
            assert int(mv.cpu()[0]) == int(moves[t])
            assert np.allclose(pi32.cpu().numpy()[0], pis[t], rtol=0, atol=1e-6)
            if t < len(moves) - 1:
                assert int(eng.root_children()[3].cpu()[0]) == int(key("root_n")[t])
        games, plies = eng.pending()
        assert (games, plies) == (1, len(moves))
        tb = eng.harvest()
        assert tb.n_games == 1 and len(tb) == len(moves)
        assert np.array_equal(tb.z.cpu().numpy().astype(np.float64), z)
        assert np.array_equal(tb.boards.to_packed().view(np.uint64), key("boards").view(np.uint64))
        assert np.allclose(tb.pi.cpu().numpy(), pis, rtol=0, atol=1e-6)
        st = tb.states().cpu().numpy()
        assert np.array_equal(np.packbits(st[0].astype(np.uint8).reshape(-1)), key("first_state_bits"))
        assert np.array_equal(np.packbits(st[-1].astype(np.uint8).reshape(-1)), key("last_state_bits"))
        # the slot restarted from the opening with a fresh tree
        assert eng.get_boards().to_packed()[0].tobytes() == opening_packed(1)[0].tobytes()
        assert eng.pending() == (0, 0) and eng.stats()["games_finished"] == 1
        eng.close()


def test_terminal_sign_flag(gpu_device):
    import oracle

    g = oracle.OracleGame.from_fields(np.zeros(64), 67, 40, 0, 0, 1)
    b = np.array([g.packed()])
    for fix, qexp in ((False, -1.0), (True, 1.0)):
        eng = make_engine(b, 60, fix_terminal_sign=fix)
        run_playouts(eng, "uniform", 60)
        visits, q, _, _ = (t.cpu().numpy() for t in eng.root_children())
        assert q[0, 0] == qexp
        assert (visits[0, 0] == visits[0][visits[0] >= 0].min()) == (not fix)
        assert eng.stats()["leaf_terminal"] > 0
        eng.close()


def test_arena_overflow_is_safe(gpu_device):
    """Tiny arenas: expansions are skipped (counted), nothing is corrupted, play continues."""
    from alphazero_quoridor_amd.boards import opening_packed

    eng = make_engine(opening_packed(8), 40, node_cap=6, edge_cap=700)
    run_playouts(eng, "hash", 40)
    st = eng.stats()
    assert st["node_overflow"] > 0 and st["playouts"] == 8 * 40
    visits, _, _, root_n = (t.cpu().numpy() for t in eng.root_children())
    assert (root_n == 40).all()
    assert (np.where(visits >= 0, visits, 0).sum(axis=1) == 39).all()  # first playout expands the root
    eng.finish_move()
    run_playouts(eng, "hash", 10)
    eng.close()


def test_sampled_moves_follow_noisy_pi(gpu_device):
    """mcts.py:181: move ~ 0.75*pi + 0.25*Dirichlet(0.3).  E[p] = 0.75*pi + 0.25/k."""
    from alphazero_quoridor_amd.boards import opening_packed

    B = 8192
    eng = make_engine(opening_packed(B), 40, seed=123)
    run_playouts(eng, "hash", 40)
    pi, visits = eng.root_pi()
    pi = pi.cpu().numpy()[0]
    moves, _ = eng.finish_move()
    moves = moves.cpu().numpy()
    legal = np.nonzero(visits.cpu().numpy()[0] >= 0)[0]
    k = len(legal)
    expect = 0.75 * pi[legal] + 0.25 / k
    counts = np.array([(moves == a).sum() for a in legal])
    assert counts.sum() == B
    chi2 = ((counts - B * expect) ** 2 / (B * expect)).sum()
    assert chi2 < k + 6 * np.sqrt(2 * k), chi2  # ~6 sigma of a chi-square with k-1 dof
    # a second engine with another seed samples differently, the same seed identically
    eng2 = make_engine(opening_packed(B), 40, seed=123)
    run_playouts(eng2, "hash", 40)
    assert np.array_equal(eng2.finish_move()[0].cpu().numpy(), moves)
    eng3 = make_engine(opening_packed(B), 40, seed=124)
    run_playouts(eng3, "hash", 40)
    assert not np.array_equal(eng3.finish_move()[0].cpu().numpy(), moves)
    for e in (eng, eng2, eng3):
        e.close()


def test_continuous_selfplay_tuples_are_consistent(gpu_device):
    """Free-running self-play (device sampling, continuous refill): every harvested game is a
    legal move sequence from the opening whose last move wins, z follows quoridor.py:599-602."""
    import oracle
    from alphazero_quoridor_amd.boards import opening_packed

    B = 64
    eng = make_engine(opening_packed(B), 3, seed=7, fix_terminal_sign=True)  # fixed sign => short games
    games = 0
    for ply in range(400):
        run_playouts(eng, "hash", 3)
        eng.finish_move()
        tb = eng.harvest()
        if tb is None:
            continue
        packed = tb.boards.to_packed()
        gid = tb.game.cpu().numpy()
        z = tb.z.cpu().numpy()
        pi = tb.pi.cpu().numpy()
        for g in range(tb.n_games):
            rows = np.nonzero(gid == g)[0]
            assert rows.tolist() == list(range(rows[0], rows[-1] + 1))
            seq = packed[rows]
            assert seq[0].tobytes() == opening_packed(1)[0].tobytes()
            og = oracle.OracleGame()
            for t in range(len(rows)):
                assert og.packed().tobytes() == seq[t].tobytes()
                legal = og.actions()
                assert abs(pi[rows[t]].sum() - 1.0) < 1e-5
                assert set(np.nonzero(pi[rows[t]])[0]) <= set(legal)
                if t + 1 < len(rows):
                    nxt = [a for a in legal if _after(og, a) == seq[t + 1].tobytes()]
                    assert len(nxt) >= 1
                    og.step(nxt[0])
            wins = [a for a in og.actions() if _wins(og, a)]
            assert wins, "last recorded position has a winning move"
            w = og.get_current_player()  # the mover of the last ply won
            movers = seq["cur"]
            assert np.array_equal(z[rows], np.where(movers == w, 1.0, -1.0))
            games += 1
        if games >= 12:
            break
    assert games >= 12
    st = eng.stats()
    assert st["games_finished"] == games and st["games_aborted"] == 0
    eng.close()


def _after(og, a):
    c = og.copy()
    c.step(a)
    return c.packed().tobytes()


def _wins(og, a):
    c = og.copy()
    return c.step(a)
