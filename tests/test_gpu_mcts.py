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


def test_tree_pool_exhaustion_is_safe(gpu_device):
    """A shared tree pool that is far too small: expansions are skipped and subtrees truncated
    (both counted), nothing is corrupted, pages are recycled and play continues."""
    from alphazero_quoridor_amd.boards import opening_packed

    B = 96
    eng = make_engine(opening_packed(B), 40, tree_pool_pages=100, seed=3)  # ~1 page per board, 2 needed to re-root
    for _ in range(6):
        run_playouts(eng, "hash", 40)
        visits, _, _, root_n = (t.cpu().numpy() for t in eng.root_children())
        assert (np.where(visits >= 0, visits, 0).sum(axis=1) <= root_n).all()
        eng.finish_move()
    st = eng.stats()
    assert st["node_overflow"] > 0 and st["playouts"] == B * 240
    assert st["tree_pages_total"] == 100 and st["tree_pages_peak"] <= 100 and st["tree_pages_in_use"] <= 100
    eng.reset()
    assert eng.stats()["tree_pages_in_use"] == 0  # every page came back
    eng.close()


def test_pages_are_recycled_and_pool_accounting_balances(gpu_device):
    """After any number of plies the pages in use are exactly the pages the live trees map."""
    from alphazero_quoridor_amd.boards import opening_packed

    B = 64
    eng = make_engine(opening_packed(B), 24, seed=5, fix_terminal_sign=True)
    peak = 0
    for ply in range(60):
        run_playouts(eng, "hash", 24)
        eng.finish_move()
        eng.harvest()
        st = eng.stats()
        # a re-rooted tree owns ceil(n_edges / 2048) pages, allowing for the skipped page tails
        assert st["tree_pages_in_use"] <= B * (st["max_edges"] // 2048 + 2)
        peak = max(peak, st["tree_pages_in_use"])
    st = eng.stats()
    assert st["node_overflow"] == 0 and st["tree_pages_peak"] >= peak
    eng.reset()
    st = eng.stats()
    assert st["tree_pages_in_use"] == 0 and st["traj_pages_in_use"] == 0
    eng.close()


def test_games_cross_trajectory_pages_and_come_back_intact(gpu_device):
    """Games are never dropped for length: a game's variable-length records run over many
    trajectory pages (256-dword pages here, so that every game crosses dozens of them; 64 KB in
    production) and the harvested tuples equal what finish_move reported ply by ply (board before
    the move, float32 pi)."""
    from alphazero_quoridor_amd.boards import opening_packed

    B = 16
    eng = make_engine(opening_packed(B), 3, seed=11, fix_terminal_sign=True, traj_page_dwords=256, traj_pool_pages=B * 200)
    log = [[] for _ in range(B)]
    games, longest = 0, 0
    for ply in range(600):
        before = eng.get_boards().to_packed()
        run_playouts(eng, "hash", 3)
        moves, pi = eng.finish_move()
        moves, pi = moves.cpu().numpy(), pi.cpu().numpy()
        mid = eng.get_boards().to_packed()
        for b in range(B):
            assert moves[b] != 255
            log[b].append((before[b].tobytes(), pi[b].copy()))
        fin = [b for b in range(B) if mid[b]["p1"] >= 72 or mid[b]["p2"] <= 8]   # has_a_winner (quoridor.py:193-202)
        tb = eng.harvest()
        assert (tb is None) == (not fin)
        if tb is None:
            continue
        packed, tpi, gid = tb.boards.to_packed(), tb.pi.cpu().numpy(), tb.game.cpu().numpy()
        assert len(fin) == tb.n_games
        for g, b in enumerate(fin):  # games come in board order
            rows = np.nonzero(gid == g)[0]
            assert len(rows) == len(log[b])
            for t, r in enumerate(rows):
                assert packed[r].tobytes() == log[b][t][0] and np.array_equal(tpi[r], log[b][t][1])
            longest = max(longest, len(rows))
            log[b] = []
            games += 1
        if games >= 10 and longest >= 60:
            break
    assert games >= 10 and longest >= 60, (games, longest)
    st = eng.stats()
    assert st["games_aborted"] == 0 and st["traj_pages_peak"] >= 3 * B
    eng.close()


def test_max_plies_and_trajectory_pool_aborts_are_counted_by_cause(gpu_device):
    from alphazero_quoridor_amd.boards import opening_packed

    eng = make_engine(opening_packed(4), 2, seed=1, max_plies=5)
    for _ in range(8):
        run_playouts(eng, "uniform", 2)
        eng.finish_move()
    st = eng.stats()
    assert st["aborted_max_plies"] == 4 and st["aborted_pool"] == 0 and st["aborted_no_move"] == 0 and st["games_aborted"] == 4
    assert st["plies_played"] == 4 * 7  # 5 plies, the dropped attempt, 2 plies of the restarted game
    eng.close()
    eng = make_engine(opening_packed(4), 2, seed=1, traj_pool_pages=2)  # two boards never get a page
    run_playouts(eng, "uniform", 2)
    eng.finish_move()
    st = eng.stats()
    assert st["aborted_pool"] == 2 and st["plies_played"] == 2 and st["games_aborted"] == 2
    eng.close()


def test_illegal_forced_move_is_an_error_not_a_substitution(gpu_device):
    from alphazero_quoridor_amd import _cabi
    from alphazero_quoridor_amd.boards import opening_packed

    eng = make_engine(opening_packed(2), 8)
    run_playouts(eng, "hash", 8)
    before = eng.get_boards().to_packed()
    with pytest.raises(_cabi.QzError):
        eng.finish_move(forced=torch.tensor([1, 0], dtype=torch.uint8))  # action 1 (south) is illegal for P1 at the opening
    after = eng.get_boards().to_packed()
    assert after[0].tobytes() == before[0].tobytes()          # the board with the bad move did not move
    assert after[1].tobytes() != before[1].tobytes()          # the other one did
    assert eng.moves.cpu().numpy().tolist() == [255, 0]
    assert eng.stats()["bad_forced_moves"] == 1
    eng.close()


def test_nonfinite_network_outputs_do_not_corrupt_the_tree(gpu_device):
    """A diverged network (NaN priors / values): every PUCT comparison is false, Python's max()
    returns the first child (mcts.py:42) -- so does the descent, in bounds, and it is counted."""
    from alphazero_quoridor_amd.boards import opening_packed

    B = 16
    eng = make_engine(opening_packed(B), 12)
    for i in range(12):
        eng.select_boards()
        p = torch.full((B, 140), float("nan"), dtype=torch.float32, device=eng.device)
        v = torch.full((B,), float("nan"), dtype=torch.float32, device=eng.device)
        eng.expand_backup(p, v)
    visits, _, _, root_n = (t.cpu().numpy() for t in eng.root_children())
    assert (root_n == 12).all()
    assert (visits[:, 0] == 11).all()  # always the first child in actions() order (action 0)
    st = eng.stats()
    assert st["nonfinite_values"] > 0 and st["node_overflow"] == 0
    eng.finish_move()
    eng.close()


def test_32768_boards_at_400_playouts_fit_one_gpu(gpu_device):
    """BASELINE configs[2]: the engine for 32,768 concurrent boards x n_playout=400 is created
    with the default pools (round 1 needed 393 GB for it) and plays plies with the real net;
    nothing overflows, nothing is dropped."""
    from alphazero_quoridor_amd.engine import SelfPlayEngine
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

    torch.manual_seed(0)
    eng = SelfPlayEngine(32768, n_playout=400, seed=9, device=gpu_device)
    st = eng.stats()
    assert st["arena_bytes"] < 200e9, st["arena_bytes"]
    ev = PolicyValueNet(use_gpu=True, device=gpu_device).evaluator("per_leaf")
    for ply in range(2):
        eng.run_playouts(ev, 400 if ply == 0 else 60)
        visits, _, _, root_n = eng.root_children()
        assert int(root_n.min()) >= 60 and bool((visits.clamp(min=0).sum(dim=1) < root_n).all())
        eng.finish_move()
        eng.harvest()
    st = eng.stats()
    assert st["node_overflow"] == 0 and st["games_aborted"] == 0 and st["plies_played"] == 2 * 32768
    assert st["playouts"] == 32768 * 460 and st["tree_pages_peak"] <= st["tree_pages_total"]
    # the leaf batch of the next playout, in situ at this size: the rules kernels the library can pick from agree on
    # every live leaf (pooled pipeline = its choice here, k_wave_rules with base paths on nine lanes per player, the same
    # with one search per lane), and a sample matches the oracle
    import oracle
    from alphazero_quoridor_amd import rules
    from alphazero_quoridor_amd.boards import DeviceBoards

    eng.run_playouts(ev, 30)
    leaf = eng.select_boards().to_packed()
    live = leaf[(leaf["p1"] <= 71) & (leaf["p2"] >= 9)]
    assert len(live) > 30000
    db = DeviceBoards.from_packed(live, gpu_device)
    mask, planes = rules.movegen_encode(db)
    for variant in (3, 5):
        m, p = rules.movegen_encode(db, opts=rules.rules_opts(variant))
        assert torch.equal(m, mask) and torch.equal(p, planes), variant
    omask, status = oracle.movegen_batch(live[::64])
    assert (status >= 0).all() and np.array_equal(mask.cpu().numpy().view(np.uint32)[::64], omask)
    assert np.array_equal(planes.cpu().numpy()[::64], oracle.encode_batch(live[::64]))
    # ... and the asynchronous loop at this size (32,768 wavefronts per launch, the miss list's rules op on a device-side
    # count, memo tables of 32,768 boards): moves are played, nothing overflows, the memo answers
    eng.finish_move()
    eng.harvest()
    eng.run_rounds(ev, 24, max_playouts=32, budget_us=1000)
    st = eng.stats()
    assert st["node_overflow"] == 0 and st["runaway_descents"] == 0 and st["games_aborted"] == 0 and st["rounds"] == 24
    assert st["nn_evals"] > 0 and st["memo_inserts"] > 0 and st["playouts"] > 32768 * 490
    # ... and IN SITU (configs[2] on the asynchronous loop): four more rounds issued piece by piece; the miss
    # list of each -- the leaves the memo did not know, whose legal sets the loop's own rules op has just produced from a
    # device-side count -- against the oracle (every 8th leaf), and the evaluations the loop stored for them against the
    # full-batch evaluation of the same boards (the purity the memo rests on)
    import ctypes as C

    from alphazero_quoridor_amd import _cabi

    checked = 0
    for it in range(4):
        eng._memo_guard(ev)
        _cabi.check(eng.L.qz_selfplay_advance(eng.h, 32, 1000, 1, eng._s()))
        _cabi.check(eng.L.qz_selfplay_leaf_rules(eng.h, eng._s()))
        _cabi.check(eng.L.qz_selfplay_evaluate(eng.h, C.byref(ev.nn_weights()), eng._s()))
        packed, mmask, mp, mv = eng.misses()
        assert eng.stats()["waiting_boards"] == len(packed)
        if len(packed):
            omask, status = oracle.movegen_batch(packed[::8])
            assert (status >= 0).all() and np.array_equal(mmask[::8], omask), it
            sub = packed[::32]
            sdb = DeviceBoards.from_packed(sub, gpu_device)
            pr, vr = ev(None, leaf=(sdb.struct(), 0, len(sub)))
            assert np.array_equal(mp[::32], pr.cpu().numpy()) and np.array_equal(mv[::32], vr.cpu().numpy()), it
            checked += len(packed[::8])
        _cabi.check(eng.L.qz_selfplay_round_tail(eng.h, eng._s()))
    st = eng.stats()
    assert checked > 1000 and st["node_overflow"] == 0 and st["miss_overflow"] == 0 and st["runaway_descents"] == 0 and st["rounds"] == 28
    print("32,768 boards x n_playout=400 on the asynchronous loop: %d miss-list leaves in situ against the oracle, %d playouts, %d evaluations, %d memo hits"
          % (checked, st["playouts"], st["nn_evals"], st["memo_hits"]))
    eng.close()


def test_descent_records_do_not_change_the_search(gpu_device):
    """k_select follows recorded descents (replay rounds, hops between 16 records, records renamed across
    re-roots, value-based eviction); none of it may change what the walk would have done.  Two engines on
    the same late-game boards (no walls left: narrow nodes, deep lines, the case the records exist for),
    same seed, real network: one with the records, one walking every level (select_opts bit 0).  Root
    statistics, sampled moves and pi must be identical, bit for bit, ply after ply, while the trees grow
    past 64 levels."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from _stubs import det_fill_state_dict
    from synth import synth_positions
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

    boards = synth_positions(256, seed=5, max_walls=8)
    boards["w1"] = 0
    boards["w2"] = 0
    net = PolicyValueNet(use_gpu=True, device=gpu_device)
    net.policy_value_net.load_state_dict(det_fill_state_dict(net.policy_value_net.state_dict(), 2024))
    ev = net.evaluator("per_leaf")
    a = make_engine(boards, 64, seed=3)
    b = make_engine(boards, 64, seed=3, select_opts=1)
    try:
        for ply in range(48):
            for eng in (a, b):
                eng.run_playouts(ev, 64)
            ra, rb = a.root_children(), b.root_children()
            for x, y in zip(ra, rb):
                assert torch.equal(x, y), ply
            (ma, pa), (mb, pb) = a.finish_move(), b.finish_move()
            assert torch.equal(ma, mb) and torch.equal(pa, pb), ply
            a.harvest()
            b.harvest()
        sa, sb = a.stats(), b.stats()
        assert sa["max_depth"] == sb["max_depth"] and sa["descent_levels"] == sb["descent_levels"] and sa["edges_expanded"] == sb["edges_expanded"]
        print("deepest descent %d levels; levels of >= 256-level descents replayed: %d of %d" % (sa["max_depth"], sa["deep_levels_replayed"], sa["deep_levels"]))
        assert sa["max_depth"] >= 64, sa["max_depth"]
    finally:
        a.close()
        b.close()
