"""GPU parity AT THE BENCH'S OWN SHAPES ("What's missing" 3 and 4; "Next round" 2 and 7).

  * bench.py's default engine -- 13,312 boards on the 7,168 wavefront slots of k_advance<7>, the real network, one 3,000-us
    deadline per launch (and round 4's shape: 10,240 boards, per-board budgets of 2,400 us) -- against the oracle: every (board -> p, v) the network produced for the subtrees of 64 sampled boards is taken
    from the miss lists (qz_selfplay_misses) and handed to oracle.OracleMCTS (the C restatement of mcts.py:103-151) as its
    policy; three plies of 400 playouts, root visits / float64 Q / float32 P bit-equal.
  * SURVEY 4 T2 at its prescribed size: 10^6 positions through the library's default kernel choice at 32,768 boards per
    launch (BASELINE configs[2]'s batch) against the C oracle (quoridor.py:138-157 actions(), :58-131 state()) on every
    core of the box: every mask, every plane.
"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))

ORDER = list(range(12)) + [a for ix in range(64) for a in (12 + ix, 76 + ix)]


def _usable_cores():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


def oracle_masks_planes(boards, pool, want_planes=True, chunk=1024):
    """oracle.movegen_batch / encode_batch of `boards` on every core: the C calls release the GIL (ctypes), so threads do."""
    import oracle

    parts = [boards[i:i + chunk] for i in range(0, len(boards), chunk)]

    def one(part):
        m, st = oracle.movegen_batch(part)
        return m, st, (oracle.encode_batch(part) if want_planes else None)

    res = list(pool.map(one, parts))
    mask = np.concatenate([r[0] for r in res])
    status = np.concatenate([r[1] for r in res])
    planes = np.concatenate([r[2] for r in res]) if want_planes else None
    return mask, status, planes


@pytest.mark.parametrize("B,BUDGET,select_opts", [(13312, 3000, 8), (10240, 2400, 0)])
def test_bench_shape_real_net_equals_the_oracle(gpu_device, B, BUDGET, select_opts):
    """`python bench.py`'s engine, round 5's default shape (13,312 boards, ONE deadline of 3,000 us per launch, boards taking the
    first slots in turn: select_opts 8) and round 4's (10,240 boards, per-board budgets of 2,400 us): more boards than the 8,192
    wavefront slots of k_advance (slots are handed from boards that leave a launch to the boards beyond the 7,168th),
    n_playout=400, the real network in parity precision, the bench's playout cap.  85 % late-game boards (the mover has no wall left: the memo's regime, 30 playouts
    per launch) and 15 % whose mover has walls (one network round trip per playout), as in a sustained run.  64 boards are
    followed in the oracle: mcts.py's pointer tree fed with the evaluations the ENGINE'S network produced for their subtrees
    (collected from the miss lists; a leaf of board j carries a superset of j's walls, which is what the collector filters on).
    After each of three plies: legal children in actions() order, visit counts, float64 Q and float32 priors bit-equal."""
    import ctypes as C

    import oracle
    from _stubs import det_fill_state_dict
    from alphazero_quoridor_amd import _cabi
    from alphazero_quoridor_amd.boards import DeviceBoards
    from alphazero_quoridor_amd.engine import SelfPlayEngine
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet
    from synth import synth_positions

    NP = 400
    net = PolicyValueNet(use_gpu=True, device=gpu_device)
    net.policy_value_net.load_state_dict(det_fill_state_dict(net.policy_value_net.state_dict(), 2024))
    ev = net.evaluator("per_leaf")
    assert ev.engine_route_ok()
    n_open = B * 15 // 100
    late = synth_positions(B - n_open, seed=501, min_walls=3, max_walls=12)
    late["w1"] = 0
    late["w2"] = 0
    opn = synth_positions(n_open, seed=502, min_walls=3, max_walls=14, mover_has_walls=True)
    boards = np.concatenate([late, opn])
    rng = np.random.RandomState(7)
    boards = boards[rng.permutation(B)]  # the two kinds interleaved over the slots
    assert all(not oracle.OracleGame.from_packed(b).has_a_winner()[0] for b in boards[::97])
    mover_walls = np.where(boards["cur"] == 1, boards["w1"], boards["w2"])
    samp = np.concatenate([np.nonzero(mover_walls == 0)[0][:: (B - n_open) // 48][:48], np.nonzero(mover_walls > 0)[0][:: n_open // 16][:16]])
    # a sampled board must be the only board with its wall set (the collector's filter): true for random sets of >= 3 walls
    S = len(samp)
    assert S == 64
    s_hb, s_vb = boards["hbits"][samp].copy(), boards["vbits"][samp].copy()

    eng = SelfPlayEngine(B, n_playout=NP, c_puct=5.0, temp=1.0, seed=3, device=gpu_device, max_depth=992, select_opts=select_opts)
    eng.set_boards(DeviceBoards.from_packed(boards, eng.device), reset_trees=True)
    L = eng.L
    table = {}

    def collect(packed, p, v):
        if not len(packed):
            return
        hb, vb = packed["hbits"], packed["vbits"]
        keep = np.zeros(len(packed), dtype=bool)
        for k in range(S):
            keep |= ((hb & s_hb[k]) == s_hb[k]) & ((vb & s_vb[k]) == s_vb[k])
        for i in np.nonzero(keep)[0]:
            table[packed[i].tobytes()] = (p[i].copy(), v[i])

    def policy(g, legal):
        p, v = table[g.packed().tobytes()]  # KeyError = the oracle's search met a leaf the engine's never evaluated
        return legal, p[legal], float(v)

    trees = [oracle.OracleMCTS(policy, c_puct=5, n_playout=NP) for _ in range(S)]
    games = [oracle.OracleGame.from_packed(boards[j]) for j in samp]
    alive = np.ones(S, dtype=bool)
    t0 = time.time()
    rounds = 0
    try:
        for ply in range(3):
            st = eng.stats()
            moved0, target = st["plies_played"], st["playouts"] + B * NP
            last, idle = -1, 0
            for _ in range(20000):
                eng._memo_guard(ev)
                _cabi.check(L.qz_selfplay_advance(eng.h, 4096, BUDGET, 0, eng._s()))
                _cabi.check(L.qz_selfplay_leaf_rules(eng.h, eng._s()))
                _cabi.check(L.qz_selfplay_evaluate(eng.h, C.byref(ev.nn_weights()), eng._s()))
                packed, mask, p, v = eng.misses()
                collect(packed, p, v)
                _cabi.check(L.qz_selfplay_round_tail(eng.h, eng._s()))
                rounds += 1
                st = eng.stats()  # every board does exactly its NP playouts per ply (the host plays the moves: auto_finish = 0)
                if st["waiting_boards"] == 0 and st["playouts"] >= target:
                    break
                idle = idle + 1 if st["playouts"] == last else 0
                last = st["playouts"]
                assert idle < 8, ("the loop stopped making progress", ply, st["playouts"], target, st["waiting_boards"])
            assert st["playouts"] == target, (ply, st["playouts"], target)
            visits, q, prior, root_n = (t.cpu().numpy() for t in eng.root_children())
            assert (root_n >= NP).all()
            vis_dev = torch.from_numpy(visits).to(gpu_device)
            forced = torch.argmax(vis_dev, dim=1).to(torch.uint8).cpu().numpy()  # every other board: a most visited child
            forced[(visits < 0).all(axis=1)] = 255                                 # (a root without a legal move: the engine drops that game)
            for k, j in enumerate(samp):
                if not alive[k]:
                    continue
                acts, ov, _ = trees[k].get_move_probs(games[k], 1.0)
                a2, v2, q2, p2 = trees[k].root_children()
                assert [a for a in ORDER if visits[j, a] >= 0] == acts, (ply, j)
                if not acts:
                    alive[k] = False
                    continue
                assert np.array_equal(visits[j, acts], ov), (ply, j, visits[j, acts], ov)
                assert np.array_equal(q[j, acts], q2), (ply, j)
                assert np.array_equal(prior[j, acts], p2), (ply, j)
                assert root_n[j] == trees[k].root_visits()
                mv = acts[int(np.argmax(ov))]
                forced[j] = mv
                trees[k].update_with_move(mv)
                if games[k].step(mv):
                    alive[k] = False
            eng.finish_move(torch.from_numpy(forced))
            eng.harvest()
            assert eng.stats()["plies_played"] - moved0 >= B - 64
        st = eng.stats()
        assert st["node_overflow"] == 0 and st["miss_overflow"] == 0 and st["runaway_descents"] == 0 and st["memo_hits"] > st["nn_evals"], st
        assert alive.sum() >= 32
        print("bench shape (select_opts %d): %d boards x 3 plies x %d playouts, %d rounds at a %d-us budget in %.0f s: %d network evaluations, %d memo hits, deepest "
              "descent %d levels; %d boards followed in the oracle (%d evaluations of their subtrees collected): visits, Q, P bit-equal"
              % (select_opts, B, NP, rounds, BUDGET, time.time() - t0, st["nn_evals"], st["memo_hits"], st["max_depth"], S, len(table)))
    finally:
        eng.close()


def test_t2_one_million_positions_vs_the_oracle(gpu_device):
    """>= 10^6 positions, HIP against the oracle, masks (Quoridor.actions(), quoridor.py:138-157,420-528) and
    planes (Quoridor.state(), :58-131), every one of them.  31 launches of 32,768 boards through the library's default kernel
    choice (the pooled two-launch pipeline at this size): ten synthetic sets (no walls / few / dense walls, adjacent pawns,
    movers with and without walls) and twenty-one sets reached by random legal play on the GPU (0..80 plies, three wall
    weights).  The oracle runs on every core the process may use (ctypes releases the GIL)."""
    from alphazero_quoridor_amd import rules
    from alphazero_quoridor_amd.boards import DeviceBoards
    from movegen_bench import live_only, random_play
    from synth import synth_positions

    n = 32768
    cores = _usable_cores()
    sets = []
    synth_kw = [dict(min_walls=0, max_walls=0), dict(min_walls=0, max_walls=0, adjacent_frac=1.0), dict(min_walls=0, max_walls=4),
                dict(min_walls=14, max_walls=20), dict(min_walls=16, max_walls=20, adjacent_frac=1.0, mover_has_walls=True),
                dict(min_walls=18, max_walls=20, mover_has_walls=True), dict(min_walls=4, max_walls=12, adjacent_frac=0.6),
                dict(min_walls=0, max_walls=20, mover_has_walls=True), dict(min_walls=8, max_walls=16, mover_has_walls=False, adjacent_frac=0.5),
                dict(min_walls=0, max_walls=20)]
    for k, kw in enumerate(synth_kw):
        sets.append(("synth %s" % kw, lambda k=k, kw=kw: DeviceBoards.from_packed(synth_positions(n, seed=9000 + k, **kw), gpu_device)))
    for k, (plies, ww) in enumerate([(p, w) for p in (6, 12, 20, 30, 40, 60, 80) for w in (0.3, 2.0, 12.0)]):
        sets.append(("random play <= %d plies, wall weight %g" % (plies, ww),
                     lambda k=k, plies=plies, ww=ww: live_only(random_play(n, gpu_device, 77000 + k, 0, plies, ww), n)))
    total = bad_mask = bad_planes = 0
    t_gpu = t_cpu = 0.0
    legal_sum = 0
    t_all = time.time()
    with ThreadPoolExecutor(cores) as pool:
        for name, make in sets:
            db = make()
            packed = db.to_packed()
            torch.cuda.synchronize()
            t0 = time.time()
            mask, planes = rules.movegen_encode(db)   # opts=None: the library's own kernel choice
            torch.cuda.synchronize()
            t_gpu += time.time() - t0
            t0 = time.time()
            omask, status, oplanes = oracle_masks_planes(packed, pool)
            t_cpu += time.time() - t0
            assert (status >= 0).all(), name
            m = mask.cpu().numpy().view(np.uint32)
            bm = int((m != omask).any(axis=1).sum())
            bp = int((planes.cpu().numpy().reshape(n, -1) != oplanes.reshape(n, -1)).any(axis=1).sum())
            assert bm == 0 and bp == 0, "%s: %d masks and %d plane sets of %d differ from the oracle" % (name, bm, bp, n)
            bad_mask += bm
            bad_planes += bp
            total += n
            legal_sum += int(np.unpackbits(m.view(np.uint8), axis=1).sum())
            del mask, planes, oplanes
    assert total >= 1000000 and bad_mask == 0 and bad_planes == 0
    print("T2: %d positions in %d launches of %d, %d distinct position sets: 0 mask and 0 plane mismatches against the oracle "
          "(mean %.1f legal actions; GPU %.2f s incl. the first launch, oracle %.0f s on %d threads, whole test %.0f s)"
          % (total, len(sets), n, len(sets), legal_sum / total, t_gpu, t_cpu, cores, time.time() - t_all))
