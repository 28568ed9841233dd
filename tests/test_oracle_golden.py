"""CPU tier: pins the oracle (oracle/, plain-C restatement) against the golden fixtures that
tests/golden/gen_golden.py recorded from the REAL reference.  No GPU, no reference needed.
Also checks the bitboard formulation used by the HIP kernels (tests/hostcheck, a g++ build of
alphazero_quoridor_amd/csrc/qz_rules.h) against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = os.path.join(HERE, "golden")


def load(name):
    return np.load(os.path.join(G, name))


# ------------------------------------------------------------------ rules: F1 / F2 / BFS
def test_get_intersections_all_tiles():
    d = load("rules_intersections.npz")
    for w, exp in zip(d["walls"], d["out"]):
        for t in range(81):
            got = oracle.get_intersections(w, t)
            assert [got["NW"], got["NE"], got["SE"], got["SW"]] == exp[t].tolist(), (t, w.tolist())


def test_valid_pawn_actions_every_pair():
    d = load("rules_pawn.npz")
    L = oracle.lib()
    out = (C.c_int * 12)()
    n_cases = 0
    for w, exp in zip(d["walls"], d["out"]):
        a = (C.c_int8 * 64)(*[int(x) for x in w])
        for loc in range(81):
            for opp in range(81):
                if loc == opp:
                    continue
                for player in (1, 2):
                    n = L.qzo_valid_pawn_actions(a, loc, opp, player, out)
                    e = exp[loc, opp, player - 1]
                    k = int((e >= 0).sum())
                    assert n == k and list(out[:n]) == e[:k].tolist(), (loc, opp, player)
                    n_cases += 1
    assert n_cases == 32 * 81 * 80 * 2


def test_bfs_to_goal():
    d = load("rules_bfs.npz")
    for i in range(len(d["walls"])):
        w, p1, p2 = d["walls"][i], int(d["p1"][i]), int(d["p2"][i])
        assert oracle.bfs_to_goal(w, 8, p1, p2, 1) == bool(d["reach1"][i])
        assert oracle.bfs_to_goal(w, 0, p2, p1, 2) == bool(d["reach2"][i])


# ------------------------------------------------------------------ positions: F3 / F4 / F5
@pytest.fixture(scope="module")
def positions():
    return load("rules_positions.npz")


def test_actions_ordered_lists(positions):
    d = positions
    mask, status = oracle.movegen_batch(d["board"])
    assert len(d["board"]) > 15000
    for i in range(len(d["board"])):
        exp = d["actions"][i][: d["n_actions"][i]].tolist()
        assert status[i] == len(exp)
        assert oracle.mask_to_actions(mask[i]) == exp, i
    # a few through the scalar API too (ordered list straight from qzo_actions)
    for i in range(0, len(d["board"]), 97):
        g = oracle.OracleGame.from_packed(d["board"][i])
        assert g.actions() == d["actions"][i][: d["n_actions"][i]].tolist()


def test_opening_position():
    g = oracle.OracleGame()
    a = g.actions()
    assert len(a) == 131 and a[:10] == [0, 2, 3, 12, 76, 13, 77, 14, 78, 15]
    s = g.state()
    assert s.reshape(26, 81).sum(axis=1).tolist() == [64, 0, 0, 1, 1] + [0] * 9 + [81] + [0] * 9 + [81, 0]


def test_state_planes(positions):
    d = positions
    planes = oracle.encode_batch(d["board"])
    bits = np.packbits(planes.astype(np.uint8).reshape(len(planes), -1), axis=1)
    assert np.array_equal(bits, d["state_bits"])


def test_step_transitions(positions):
    for d in (positions, load("rules_steps.npz")):
        ok = d["action"] < 140
        nb, done, winner = oracle.step_batch(d["board"][ok], d["action"][ok])
        assert np.array_equal(nb.view(np.uint64), d["next_board"][ok].view(np.uint64))
        assert np.array_equal(done, d["done"][ok])
        assert np.array_equal(winner, d["winner"][ok])


def test_offboard_winning_jumps(positions):
    """NN from row 7 over a pawn on row 8 / SS from row 1 leave the board and win."""
    d = positions
    nb = d["next_board"]
    off = (nb["p1"] > 80) | (nb["p2"] < 0)
    assert off.sum() >= 18
    assert (d["done"][off] == 1).all()
    assert ((d["winner"][off] == 1) == (nb["p1"][off] > 80)).all()


# ------------------------------------------------------------------ bitboard formulation (hostcheck)
@pytest.fixture(scope="module")
def hostcheck():
    src = os.path.join(HERE, "hostcheck", "hostcheck.cpp")
    so = os.path.join(HERE, "hostcheck", "libqz_hostcheck.so")
    hdr = os.path.join(HERE, "..", "alphazero_quoridor_amd", "csrc", "qz_rules.h")
    hdr2 = os.path.join(HERE, "..", "alphazero_quoridor_amd", "csrc", "qz_movegen_pool.h")
    hdr3 = os.path.join(HERE, "..", "alphazero_quoridor_amd", "csrc", "qz_path_rows.h")
    if not os.path.exists(so) or max(os.path.getmtime(p) for p in (src, hdr, hdr2, hdr3)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-o", so, src])
    return C.CDLL(so)


def _soa(boards):
    w = np.ascontiguousarray(boards).view(np.uint64).reshape(-1, 3)
    return [np.ascontiguousarray(w[:, k]) for k in range(3)]


def test_bitboard_movegen_equals_oracle(hostcheck, positions):
    import sys
    sys.path.insert(0, G)
    from synth import synth_positions

    boards = np.concatenate([positions["board"], synth_positions(6000, seed=77)])
    hb, vb, meta = _soa(boards)
    n = len(boards)
    omask, status = oracle.movegen_batch(boards)
    assert (status >= 0).all()
    for mode in (0, 1):  # 0 = path-cut pruning (what the kernel does), 1 = brute force
        m = np.zeros((n, 5), dtype=np.uint32)
        floods = C.c_int64(0)
        hostcheck.hc_movegen(hb.ctypes.data_as(C.c_void_p), vb.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p),
                             n, m.ctypes.data_as(C.c_void_p), mode, C.byref(floods))
        assert np.array_equal(m, omask), "mode %d" % mode
    # the pooled kernel's phase functions (qz_movegen_pool.h), tiles of 32 / 16 / 7 boards
    ref_planes = oracle.encode_batch(boards).reshape(n, 2106)
    for tile in (32, 16, 7):
        m = np.zeros((n, 5), dtype=np.uint32)
        pl = np.zeros((n, 2106), dtype=np.float32)
        floods = C.c_int64(0)
        hostcheck.hc_movegen_pool(hb.ctypes.data_as(C.c_void_p), vb.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p),
                                  n, tile, m.ctypes.data_as(C.c_void_p), pl.ctypes.data_as(C.c_void_p), C.byref(floods))
        assert np.array_equal(m, omask), "pool tile %d" % tile
        assert np.array_equal(pl, ref_planes), "pool planes tile %d" % tile
        assert floods.value < 40 * n  # path-cut pruning keeps the floods far below 256 per board
    hostcheck.hc_p2_mismatches.restype = C.c_long
    assert hostcheck.hc_p2_mismatches() == 0  # per-board need masks == per-slot cut tests
    # group detours (pool_k1 detour_mode 1 / 2): same masks, far fewer floods
    base_floods = floods.value
    for mode in (1, 2):
        hostcheck.hc_set_detour(mode)
        m = np.zeros((n, 5), dtype=np.uint32)
        floods = C.c_int64(0)
        hostcheck.hc_movegen_pool(hb.ctypes.data_as(C.c_void_p), vb.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p),
                                  n, 24, m.ctypes.data_as(C.c_void_p), None, C.byref(floods))
        assert np.array_equal(m, omask), "detour mode %d" % mode
        assert floods.value < base_floods * (0.75 if mode == 1 else 0.5)
    hostcheck.hc_set_detour(0)
    # the nine-rows formulation of the base-path search (qz_path_rows.h: what k_wave_rules runs on nine
    # lanes per player) in place of the one-lane search: same masks, with and without group detours
    for mode in (0, 1):
        hostcheck.hc_set_finder(1)
        hostcheck.hc_set_detour(mode)
        m = np.zeros((n, 5), dtype=np.uint32)
        hostcheck.hc_movegen_pool(hb.ctypes.data_as(C.c_void_p), vb.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p),
                                  n, 16, m.ctypes.data_as(C.c_void_p), None, None)
        hostcheck.hc_set_finder(0)
        hostcheck.hc_set_detour(0)
        assert np.array_equal(m, omask), "rows finder, detour mode %d" % mode
    # the hand-off of the device's pooled pipeline (launch 1 -> 184-byte PoolHand record -> launch 2): what the second launch
    # rebuilds from the record -- blocked sets, slot tests, path edge sets, path tiles, jump positions, the srcpos table, every
    # suffix set -- equals what pool_k1 computes directly, field by field, and the tile run on the rebuilt records (suffix
    # sets taken from the tile sequence: pool_p3_seq) gives the oracle's masks, with and without group detours
    hostcheck.hc_handoff_mismatches.restype = C.c_long
    for mode in (0, 1, 2):
        hostcheck.hc_set_handoff(1)
        hostcheck.hc_set_detour(mode)
        m = np.zeros((n, 5), dtype=np.uint32)
        hostcheck.hc_movegen_pool(hb.ctypes.data_as(C.c_void_p), vb.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p),
                                  n, 24, m.ctypes.data_as(C.c_void_p), None, None)
        hostcheck.hc_set_handoff(0)
        hostcheck.hc_set_detour(0)
        assert np.array_equal(m, omask), "hand-off record, detour mode %d" % mode
        assert hostcheck.hc_handoff_mismatches() == 0
    hostcheck.hc_cut_row_mismatches.restype = C.c_long
    assert hostcheck.hc_cut_row_mismatches() == 0  # cut masks in rows == path_cut_masks() on every path the rows finder found
    hostcheck.hc_plan_rows_mismatches.restype = C.c_long
    assert hostcheck.hc_plan_rows_mismatches(hb.ctypes.data_as(C.c_void_p), vb.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p), n) == 0
    hostcheck.hc_blocked_rows_mismatches.restype = C.c_long
    assert hostcheck.hc_blocked_rows_mismatches(hb.ctypes.data_as(C.c_void_p), vb.ctypes.data_as(C.c_void_p), n) == 0
    # ordered list from the mask through order_index (the expand kernel's slot rule)
    out = (C.c_int * 140)()
    for i in range(0, n, 53):
        k = hostcheck.hc_ordered(omask[i].ctypes.data_as(C.c_void_p), out)
        assert list(out[:k]) == oracle.mask_to_actions(omask[i])


def test_bitboard_corners_pawn_reach(hostcheck):
    d = load("rules_intersections.npz")
    for w, exp in zip(d["walls"], d["out"]):
        if (np.abs(w) == 1).all():
            pass
        hb = sum(1 << i for i in range(64) if w[i] == 1)
        vb = sum(1 << i for i in range(64) if w[i] == -1)
        for t in range(81):
            got = [hostcheck.hc_corner(C.c_uint64(hb), C.c_uint64(vb), t, k) for k in range(4)]
            assert got == exp[t].tolist(), (t,)
    d = load("rules_pawn.npz")
    hostcheck.hc_pawn_actions.restype = C.c_uint32
    for w, exp in zip(d["walls"][:8], d["out"][:8]):
        hb = sum(1 << i for i in range(64) if w[i] == 1)
        vb = sum(1 << i for i in range(64) if w[i] == -1)
        for loc in range(81):
            for opp in range(81):
                if loc == opp:
                    continue
                for player in (1, 2):
                    e = exp[loc, opp, player - 1]
                    e = e[e >= 0].tolist()
                    m = hostcheck.hc_pawn_actions(C.c_uint64(hb), C.c_uint64(vb), loc, opp, player)
                    got = [a for a in range(12) if (m >> a) & 1]
                    assert got == e and e == sorted(e), (loc, opp, player)
    d = load("rules_bfs.npz")
    for i in range(len(d["walls"])):
        w, p1, p2 = d["walls"][i], int(d["p1"][i]), int(d["p2"][i])
        hb = sum(1 << k for k in range(64) if w[k] == 1)
        vb = sum(1 << k for k in range(64) if w[k] == -1)
        assert hostcheck.hc_reach(C.c_uint64(hb), C.c_uint64(vb), p1, p2, 1) == int(d["reach1"][i])
        assert hostcheck.hc_reach(C.c_uint64(hb), C.c_uint64(vb), p1, p2, 2) == int(d["reach2"][i])


# ------------------------------------------------------------------ MCTS: F6
def test_mcts_visit_counts_and_q():
    d = load("mcts_stub.npz")
    assert len(d["board"]) >= 300
    for i in range(len(d["board"])):
        g = oracle.OracleGame.from_packed(d["board"][i])
        m = oracle.OracleMCTS(str(d["policy"][i]), c_puct=float(d["c_puct"][i]), n_playout=int(d["n_playout"][i]))
        acts, visits, probs = m.get_move_probs(g, float(d["temp"][i]))
        k = int(d["k"][i])
        a2, v2, q2, p2 = m.root_children()
        assert acts == d["acts"][i][:k].tolist()
        assert np.array_equal(visits, d["visits"][i][:k])
        assert np.array_equal(q2, d["q"][i][:k])          # float64 Q, bit for bit
        assert np.array_equal(p2, d["p"][i][:k])          # float32 priors
        assert m.root_visits() == d["root_visits"][i]
        assert m.node_count() == d["nodes"][i] and m.max_depth() == d["depth"][i]
        assert np.allclose(probs, d["probs"][i][:k], rtol=0, atol=1e-12)  # libm vs numpy exp/log


def test_mcts_terminal_sign_bug_reproduced():
    """mcts.py:125 + quoridor.py:176-181: a one-move win is backed up as -1 for the mover."""
    g = oracle.OracleGame.from_fields(np.zeros(64), 67, 40, 0, 0, 1)  # P1 one step from row 8
    m = oracle.OracleMCTS("uniform", c_puct=5, n_playout=60)
    acts, visits, _ = m.get_move_probs(g, 1.0)
    _, _, q, _ = m.root_children()
    i = acts.index(0)
    assert q[i] == -1.0 and visits[i] == visits.min()
    m2 = oracle.OracleMCTS("uniform", c_puct=5, n_playout=60, fix_terminal_sign=True)
    acts2, visits2, _ = m2.get_move_probs(g, 1.0)
    assert m2.root_children()[2][acts2.index(0)] == 1.0 and visits2[acts2.index(0)] == visits2.max()


# ------------------------------------------------------------------ episodes: F7
def test_episode_traces():
    d = load("episodes_stub.npz")
    for e in range(int(d["n"])):
        key = lambda k: d["e%d_%s" % (e, k)]  # noqa: E731
        moves, pis, players, z = key("moves"), key("pis"), key("players"), key("z")
        m = oracle.OracleMCTS(str(key("policy")), c_puct=5, n_playout=int(key("n_playout")))
        g = oracle.OracleGame()
        for t in range(len(moves)):
            assert g.packed().tobytes() == key("boards")[t].tobytes()
            assert g.get_current_player() == players[t]
            acts, visits, probs = m.get_move_probs(g, 1.0)
            pi = np.zeros(140)
            pi[acts] = probs
            assert np.allclose(pi, pis[t], rtol=0, atol=1e-12), (e, t)
            m.update_with_move(int(moves[t]))  # subtree reuse (mcts.py:182)
            assert m.root_visits() == key("root_n")[t]
            done = g.step(int(moves[t]))
            assert done == (t == len(moves) - 1)
        end, winner = g.has_a_winner()
        assert end and winner == int(key("winner"))
        assert np.array_equal(np.where(players == winner, 1.0, -1.0), z)  # quoridor.py:599-602


def test_roots_without_a_legal_move():
    """no_move_roots.npz: root positions at which a GPU self-play run dropped the game under aborted_no_move; the REAL
    reference was asked about each (gen_golden.py:gen_no_move): actions() == [], MCTSPlayer.choose_action prints "the board
    is full" and returns None (mcts.py:195-196), start_self_play's unpack raises (quoridor.py:587).  The oracle agrees:
    no action, no pawn move, game not over; MCTS on such a root never expands it (root stays a leaf with n_playout visits)."""
    d = load("no_move_roots.npz")
    assert len(d["board"]) >= 20
    assert (d["n_actions"] == 0).all() and d["prints_board_is_full"].all() and d["returns_none"].all() and d["unpack_raises_typeerror"].all()
    assert not d["game_over"].any() and (d["n_pawn_actions"] == 0).all()
    mask, status = oracle.movegen_batch(d["board"])
    assert (status >= 0).all() and not mask.any()
    for rec in d["board"]:
        g = oracle.OracleGame.from_packed(rec)
        assert g.actions() == [] and not g.has_a_winner()[0]
        cur = g.get_current_player()
        p = g.positions
        assert oracle.valid_pawn_actions(g.inter, p[cur], p[3 - cur], cur) == []
        t = oracle.OracleMCTS("uniform", c_puct=5, n_playout=4)
        acts, visits, probs = t.get_move_probs(g, 1.0)
        assert acts == [] and t.root_visits() == 4 and t.node_count() == 0


def real_net_tables(d):
    """real_net_search.npz -> per search {packed board bytes: (dense p[140], v)}: what the REFERENCE's network returned during it"""
    out = []
    t_k, t_off, t_acts, t_p, t_v, t_board = (d[k] for k in ("t_k", "t_off", "t_acts", "t_p", "t_v", "t_board"))  # (an NpzFile decompresses on every access)
    aoff = np.concatenate([[0], np.cumsum(t_k)]).astype(np.int64)
    for i in range(len(t_off) - 1):
        tab = {}
        for t in range(int(t_off[i]), int(t_off[i + 1])):
            acts = t_acts[aoff[t]:aoff[t + 1]].astype(int)
            p = np.zeros(140, dtype=np.float32)
            p[acts] = t_p[aoff[t]:aoff[t + 1]]
            tab[t_board[t].tobytes()] = (p, np.float32(t_v[t]), acts.tolist())
        out.append(tab)
    return out


def test_real_network_search_of_the_reference():
    """tests/golden/real_net_search.npz (gen_golden.py:gen_real_net_search): 80 searches of the reference's MCTS at 400
    playouts with the reference's OWN PolicyValueNet as the policy -- there `value.data[0][0]` (policy_value_net.py:163) is a
    0-dim float32 tensor under torch >= 0.4, TreeNode._Q (mcts.py:53) turns into a float32 tensor and Q + u (mcts.py:70) is
    compared in float32; the build follows the code as written for torch 0.3 (Q a Python float: float64).  Fed with exactly the
    evaluations the reference's network produced, the oracle must visit the root's children exactly as often: the float32 / float64
    difference of Q never changes a selection on these 80 searches (64 late-game, 8 mover out of walls, 8 with walls in hand).  If
    a case ever differs, the engines need a q_dtype compatibility switch -- this test is the tripwire."""
    d = load("real_net_search.npz")
    tabs = real_net_tables(d)
    n, diff, qerr = len(d["board"]), 0, 0.0
    assert n >= 80 and (d["kind"] == 0).sum() >= 8 and (d["kind"] == 2).sum() >= 64
    for i in range(n):
        tab = tabs[i]

        def policy(g, legal, tab=tab):
            p, v, acts = tab[g.packed().tobytes()]
            assert acts == list(legal)  # the reference's actions() order = the oracle's
            return legal, p[legal], float(v)

        m = oracle.OracleMCTS(policy, c_puct=float(d["c_puct"]), n_playout=int(d["n_playout"]))
        acts, visits, _ = m.get_move_probs(oracle.OracleGame.from_packed(d["board"][i]), 1.0)
        k = int(d["k"][i])
        assert acts == d["acts"][i][:k].astype(int).tolist(), i
        diff += int(not np.array_equal(visits, d["visits"][i][:k]))
        assert m.root_visits() == int(d["root_visits"][i])
        qerr = max(qerr, float(np.abs(np.asarray(m.root_children()[2]) - d["q32"][i][:k]).max()))
    assert diff == 0, "%d of %d searches differ from the reference's (float32 tensor Q): add a q_dtype switch" % (diff, n)
    assert qerr < 1e-5, qerr  # float32 running means against float64 ones
    print("%d reference searches with the reference's network: root visits identical; max |Q64 - Q32| = %.2e" % (n, qerr))


def test_oracle_under_address_and_ub_sanitizers():
    """`make -C oracle libqz_oracle_asan.so` (gcc -fsanitize=address,undefined) really runs: the
    rules fixtures, a slice of the position fixtures, MCTS searches with subtree reuse and the
    episode traces go through the sanitizer build in a child interpreter with libasan preloaded.
    Any out-of-bounds access, use-after-free or undefined behaviour in the C restatement aborts it."""
    import subprocess
    import sys

    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "libqz_oracle_asan.so"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import oracle
assert oracle._LIB_PATH.endswith("libqz_oracle_asan.so")
d = np.load(%r + "/rules_positions.npz")
b = d["board"][::9]
live = d["n_actions"][::9] > 0
mask, status = oracle.movegen_batch(b)
for i in np.nonzero(live)[0][:600]:
    assert oracle.mask_to_actions(mask[i]) == d["actions"][::9][i][: d["n_actions"][::9][i]].tolist()
ok = (b["p1"] >= 0) & (b["p1"] <= 80) & (b["p2"] >= 0) & (b["p2"] <= 80)
oracle.encode_batch(b[ok])
nb, done, win = oracle.step_batch(b[live], d["action"][::9][live])
assert np.array_equal(done, d["done"][::9][live])
m = oracle.OracleMCTS("hash", c_puct=5, n_playout=120)
g = oracle.OracleGame()
for ply in range(6):
    acts, visits, _ = m.get_move_probs(g, 1.0)
    mv = acts[int(np.argmax(visits))]
    m.update_with_move(mv)
    g.step(mv)
m2 = oracle.OracleMCTS("uniform", c_puct=5, n_playout=60, fix_terminal_sign=True)
m2.get_move_probs(oracle.OracleGame.from_packed(b[live][5]), 1.0)
del m, m2
print("sanitizer run ok")
""" % (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests", "golden"))
    env = dict(os.environ, QZ_ORACLE_SANITIZE="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "sanitizer run ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
