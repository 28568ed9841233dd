"""GPU parity of the asynchronous self-play loop (qz_selfplay_*, include/qz_abi.h) and of the
leaf-evaluation memo.

The loop changes WHEN things happen (every board runs on its own clock, the network only sees
the leaves the memo does not know), never WHAT happens to a board: per board the operations of
MCTS.get_move_probs / choose_action / start_self_play (mcts.py:129-187, quoridor.py:582-610) run
in the lock-step engine's order.  So the tests compare it with the lock-step engine (itself pinned
on the reference fixtures and the oracle by test_gpu_mcts.py / test_gpu_api.py) BIT FOR BIT: root
statistics, pi, sampled moves, harvested tuples; and the miss list in situ with the oracle (legal
sets) and with the full-batch evaluation (p, v)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def _net(gpu_device, seed=2024):
    from _stubs import det_fill_state_dict
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

    net = PolicyValueNet(use_gpu=True, device=gpu_device)
    net.policy_value_net.load_state_dict(det_fill_state_dict(net.policy_value_net.state_dict(), seed))
    return net


def _engine(boards, n_playout, **kw):
    from alphazero_quoridor_amd.boards import DeviceBoards
    from alphazero_quoridor_amd.engine import SelfPlayEngine

    eng = SelfPlayEngine(len(boards), n_playout=n_playout, c_puct=5.0, temp=1.0, device="cuda:0", **kw)
    eng.set_boards(DeviceBoards.from_packed(boards, eng.device), reset_trees=True)
    return eng


def _mixed_boards(n, seed):
    """late-game boards (no walls left: the memo's home ground) and boards with walls, half and half"""
    from synth import synth_positions

    late = synth_positions(n // 2, seed=seed, max_walls=8)
    late["w1"] = 0
    late["w2"] = 0
    early = synth_positions(n - n // 2, seed=seed + 1, max_walls=12, mover_has_walls=True)
    return np.concatenate([late, early])


def test_memo_on_and_off_give_the_same_search(gpu_device):
    """two engines on the same boards, same seed, real network, lock-step cadence
    (every board starts one playout per round, the host plays the moves): A = the plain engine (every
    leaf evaluated, k_select / k_expand_backup), B = the asynchronous loop's kernels with the memo,
    C = the same without a memo.  Root visits, Q, priors, pi and sampled moves bit-identical for 10
    plies; B must have answered most late-game leaves from the memo."""
    boards = _mixed_boards(256, seed=21)
    NP = 48
    ev = _net(gpu_device).evaluator("per_leaf")
    assert ev.engine_route_ok()
    a = _engine(boards, NP, seed=9)
    b = _engine(boards, NP, seed=9)
    c = _engine(boards, NP, seed=9, memo=False)
    try:
        for ply in range(10):
            a.run_playouts(ev, NP)
            b.run_playouts_memo(ev)
            c.run_playouts_memo(ev)
            ra, rb, rc = a.root_children(), b.root_children(), c.root_children()
            for x, y, z in zip(ra, rb, rc):
                assert torch.equal(x, y) and torch.equal(x, z), ply
            (ma, pa), (mb, pb), (mc, pc) = a.finish_move(), b.finish_move(), c.finish_move()
            assert torch.equal(ma, mb) and torch.equal(pa, pb) and torch.equal(ma, mc) and torch.equal(pa, pc), ply
            ha, hb, hc = a.harvest(), b.harvest(), c.harvest()
            assert (ha is None) == (hb is None) == (hc is None)
            if ha is not None:
                assert torch.equal(ha.pi, hb.pi) and torch.equal(ha.z, hb.z) and torch.equal(ha.boards.meta, hb.boards.meta)
        sa, sb, sc = a.stats(), b.stats(), c.stats()
        for k in ("playouts", "leaf_terminal", "descent_levels", "edges_expanded", "max_depth", "plies_played"):
            assert sa[k] == sb[k] == sc[k], k
        assert sb["node_overflow"] == 0 and sb["games_aborted"] == 0 and sb["runaway_descents"] == 0
        assert sc["memo_hits"] == 0 and sc["nn_evals"] + sc["leaf_terminal"] == sc["playouts"]
        assert sb["memo_hits"] + sb["nn_evals"] + sb["leaf_terminal"] == sb["playouts"]
        print("memo: %d hits, %d evaluations, %d inserts (%d skipped on a locked bucket) in %d playouts"
              % (sb["memo_hits"], sb["nn_evals"], sb["memo_inserts"], sb["memo_locked"], sb["playouts"]))
        assert sb["memo_hits"] > 0.3 * sb["playouts"], sb
    finally:
        a.close()
        b.close()
        c.close()


def _games_by_slot(batches):
    """{slot: [game, ...]}, game = (boards packed bytes, pi bytes, z bytes) in the order the slot played them"""
    out = {}
    for tb in batches:
        gid = tb.game.cpu().numpy()
        slot = tb.slot.cpu().numpy()
        packed = tb.boards.to_packed()
        pi = tb.pi.cpu().numpy()
        z = tb.z.cpu().numpy()
        for g in range(tb.n_games):
            sel = gid == g
            out.setdefault(int(slot[g]), []).append((packed[sel].tobytes(), pi[sel].tobytes(), z[sel].tobytes()))
    return out


@pytest.mark.parametrize("compact_edges,budget_us,pool_pages,select_opts", [(0, 0, 0, 0), (-1, 0, 0, 0), (0, 300, 192 * 60, 0), (-1, 1, 0, 0), (-1, 150, 0, 8), (0, 0, 0, 32), (-1, 1, 0, 32)])
def test_asynchronous_games_equal_lockstep_games(gpu_device, compact_edges, budget_us, pool_pages, select_opts):
    """Complete self-play games (Dirichlet noise, sampled moves, subtree reuse, continuous refill) from the
    asynchronous loop -- boards on their own clocks, several playouts and whole moves per launch, memo on --
    against the lock-step engine with the same seed: for every board slot the games come out in the same
    order with identical (board, pi, z) tuples.  Short games (terminal sign fixed, 24 playouts).  The three
    ways a move treats the kept subtree: the default threshold of this small engine (one page: compacting
    moves, deferred to the start of a launch, and in-place moves mixed), always copied (-1, what the lock-step
    engine does), and a large pool whose threshold (20 pages) these trees rarely reach: moves in place, under a
    wall-clock budget per launch.  Last: always copied under a budget of 1 us -- one playout per launch, and every subtree
    copy stops after its first window of 64 edges and goes on, window by window, in the board's next launches
    (qz_stats.compact_slices).  The fifth case: one deadline per launch (select_opts 8); the last two: the boards without walls on
    k_rows (select_opts 32, csrc/qz_rows.h) beside k_advance's launch for the others.  Afterwards the engine refuses lock-step
    calls until it is reset."""
    from alphazero_quoridor_amd import _cabi
    from alphazero_quoridor_amd.engine import SelfPlayEngine

    B, NP = 192, 24
    ev = _net(gpu_device, 7).evaluator("per_leaf")
    lock = SelfPlayEngine(B, n_playout=NP, seed=77, device=gpu_device, fix_terminal_sign=True)
    asyn = SelfPlayEngine(B, n_playout=NP, seed=77, device=gpu_device, fix_terminal_sign=True, compact_edges=compact_edges,
                          tree_pool_pages=pool_pages, select_opts=select_opts)
    try:
        lb, ab = [], []
        for _ in range(260):
            lock.play_ply(ev)
            tb = lock.harvest()
            if tb is not None:
                lb.append(tb)
        n_lock = sum(t.n_games for t in lb)
        assert n_lock >= B // 2, n_lock
        rounds = 0
        while sum(t.n_games for t in ab) < n_lock and rounds < 40000:
            asyn.run_rounds(ev, 8, max_playouts=NP + 8, budget_us=budget_us)
            rounds += 8
            tb = asyn.harvest()
            if tb is not None:
                ab.append(tb)
        gl, ga = _games_by_slot(lb), _games_by_slot(ab)
        compared = 0
        for slot, games in gl.items():
            other = ga.get(slot, [])
            k = min(len(games), len(other))
            for i in range(k):
                assert games[i] == other[i], (slot, i)
            compared += k
        st = asyn.stats()
        print("%d games compared tuple by tuple; asynchronous run: %d rounds, %d playouts, %d memo hits, %d evaluations"
              % (compared, st["rounds"], st["playouts"], st["memo_hits"], st["nn_evals"]))
        assert compared >= n_lock // 2, (compared, n_lock)
        assert st["node_overflow"] == 0 and st["runaway_descents"] == 0
        if budget_us == 1:
            assert st["compact_slices"] > 0, st
            print("subtree copies suspended and resumed: %d" % st["compact_slices"])
        else:
            assert st["compact_slices"] == 0 or budget_us > 0, st
        with pytest.raises(_cabi.QzError, match="asynchronous self-play"):
            asyn.select(want_planes=False)
        asyn.reset()
        asyn.select(want_planes=False)
    finally:
        lock.close()
        asyn.close()


@pytest.mark.parametrize("select_opts", [0, 8, 40])
def test_more_boards_than_wavefront_slots_play_the_lockstep_games(gpu_device, select_opts):
    """The bench's shape: MORE boards (8,704) than the chip holds wavefronts of k_advance (7,168: seven per SIMD), one wavefront per workgroup --
    the boards behind the 7,168th start when a board that needs the network has left -- against the lock-step engine with the
    same seed and board count: every slot's games in the same order with identical (board, pi, z) tuples.  select_opts 8: one
    deadline per launch, boards rotating through the first slots; 40: the boards without walls on k_rows as well.  Short games
    (terminal sign fixed, 12 playouts)."""
    from alphazero_quoridor_amd.engine import SelfPlayEngine

    B, NP = 8704, 12
    ev = _net(gpu_device, 7).evaluator("per_leaf")
    lock = SelfPlayEngine(B, n_playout=NP, seed=41, device=gpu_device, fix_terminal_sign=True)
    asyn = SelfPlayEngine(B, n_playout=NP, seed=41, device=gpu_device, fix_terminal_sign=True, select_opts=select_opts)
    try:
        lb, ab = [], []
        for _ in range(150):
            lock.play_ply(ev)
            tb = lock.harvest()
            if tb is not None:
                lb.append(tb)
        n_lock = sum(t.n_games for t in lb)
        assert n_lock >= B // 8, n_lock
        rounds = 0
        while sum(t.n_games for t in ab) < n_lock and rounds < 20000:
            asyn.run_rounds(ev, 16, max_playouts=NP + 8, budget_us=200)
            rounds += 16
            tb = asyn.harvest()
            if tb is not None:
                ab.append(tb)
        gl, ga = _games_by_slot(lb), _games_by_slot(ab)
        compared = late = 0
        for slot, games in gl.items():
            other = ga.get(slot, [])
            k = min(len(games), len(other))
            for i in range(k):
                assert games[i] == other[i], (slot, i)
            compared += k
            late += k if slot >= 8192 else 0
        st = asyn.stats()
        print("%d games compared tuple by tuple (%d of them on boards behind the 8,192nd); asynchronous run: %d rounds, %d playouts, %d memo hits"
              % (compared, late, st["rounds"], st["playouts"], st["memo_hits"]))
        assert compared >= n_lock // 2 and late > 0, (compared, late, n_lock)
        assert st["node_overflow"] == 0 and st["runaway_descents"] == 0 and st["miss_overflow"] == 0
    finally:
        lock.close()
        asyn.close()


def test_miss_list_in_situ_against_oracle_and_full_batch_evaluation(gpu_device):
    """The pieces of a round, one by one, on a mid-run engine: after qz_selfplay_advance the miss list holds
    exactly the boards that wait (one slot each, no duplicates of a board slot); qz_selfplay_leaf_rules gives
    the oracle's legal sets; qz_selfplay_evaluate gives, bit for bit, what the full-batch qz_nn_evaluate
    computes for the same boards (the evaluation is a pure function of the board: the memo's premise)."""
    import ctypes as C

    import oracle
    from alphazero_quoridor_amd import _cabi
    from alphazero_quoridor_amd.boards import DeviceBoards
    from alphazero_quoridor_amd.engine import SelfPlayEngine

    B, NP = 1024, 32
    ev = _net(gpu_device, 11).evaluator("per_leaf")
    eng = SelfPlayEngine(B, n_playout=NP, seed=4, device=gpu_device)
    L = eng.L
    try:
        eng.run_rounds(ev, 40, max_playouts=8)
        checked = 0
        for it in range(6):
            eng._memo_guard(ev)
            _cabi.check(L.qz_selfplay_advance(eng.h, 8, 0, 1, eng._s()))
            _cabi.check(L.qz_selfplay_leaf_rules(eng.h, eng._s()))
            _cabi.check(L.qz_selfplay_evaluate(eng.h, C.byref(ev.nn_weights()), eng._s()))
            packed, mask, p, v = eng.misses()
            n = len(packed)
            st = eng.stats()
            assert st["waiting_boards"] == n, (st["waiting_boards"], n)
            if n:
                omask, status = oracle.movegen_batch(packed)
                assert (status >= 0).all() and np.array_equal(mask, omask)
                db = DeviceBoards.from_packed(packed, gpu_device)
                pr, vr = ev(None, leaf=(db.struct(), 0, n))
                assert np.array_equal(p, pr.cpu().numpy()) and np.array_equal(v, vr.cpu().numpy())
                checked += n
            _cabi.check(L.qz_selfplay_round_tail(eng.h, eng._s()))
        print("miss-list leaves checked against the oracle and the full-batch evaluation: %d" % checked)
        assert checked > 500
        st = eng.stats()
        assert st["memo_inserts"] + st["memo_locked"] <= st["nn_evals"] and st["memo_inserts"] > 0
    finally:
        eng.close()


def test_memo_is_flushed_when_the_weights_change(gpu_device):
    """Evaluations stored under one set of weights must never answer for another: after a weight change
    (PolicyValueNet.weights_changed -> LeafEvaluator.refresh -> qz_memo_flush) the memo engine still equals
    the plain engine evaluated with the NEW weights, also through a captured HIP graph of rounds (the per-layer
    scales live in device memory: ADVICE r2)."""
    boards = _mixed_boards(128, seed=5)
    NP = 32
    net = _net(gpu_device, 3)
    ev = net.evaluator("per_leaf")
    a = _engine(boards, NP, seed=1)
    b = _engine(boards, NP, seed=1)
    try:
        for phase in range(3):
            if phase:
                with torch.no_grad():
                    for prm in net.policy_value_net.parameters():
                        prm.mul_(1.0 + 0.9 * phase).add_(0.01 * phase)  # crosses powers of two of max |w|
                net.weights_changed()
            for ply in range(3):
                a.run_playouts(ev, NP)
                b.run_playouts_memo(ev)
                for x, y in zip(a.root_children(), b.root_children()):
                    assert torch.equal(x, y), (phase, ply)
                (ma, pa), (mb, pb) = a.finish_move(), b.finish_move()
                assert torch.equal(ma, mb) and torch.equal(pa, pb), (phase, ply)
                a.harvest()
                b.harvest()
        assert b.stats()["memo_hits"] > 0
    finally:
        a.close()
        b.close()


def test_unexpanded_root_probes_the_memo(gpu_device):
    """a root that is not expanded yet -- the first playout of every new game / fresh-root restart -- is
    itself the leaf of its descent, and k_advance's memo probe (issued from select_core's leaf hook) must be issued for it
    too.  64 boards each on a mid-game position, the opening and a position without walls in hand: round 1 sends every root to the network (one evaluation
    per board, nothing is in the memo), the tail stores them; after reset + the same boards, the first playout of every
    board must be a memo HIT, no leaf goes to the network, and the expanded roots equal the first run's bit for bit."""
    from synth import synth_positions

    mid = synth_positions(1, seed=21, max_walls=10, mover_has_walls=True)
    start = np.zeros(1, dtype=mid.dtype)
    start["p1"], start["p2"], start["w1"], start["w2"], start["cur"] = 4, 76, 10, 10, 1
    late = synth_positions(1, seed=22, max_walls=8)
    late["w1"] = 0
    late["w2"] = 0     # (the memo's small table: the mover has no wall left)
    boards = np.concatenate([np.repeat(mid, 64), np.repeat(start, 64), np.repeat(late, 64)])
    ev = _net(gpu_device, 4).evaluator("per_leaf")
    e = _engine(boards, 8, seed=1)
    try:
        e.run_rounds(ev, 1, max_playouts=1, auto_finish=False)   # every root: probe (miss) -> network -> memo insert
        e.run_rounds(ev, 1, max_playouts=1, auto_finish=False)   # ... consumed: the roots are expanded, one more playout each
        st1 = e.stats()
        assert st1["nn_evals"] >= 192 and st1["memo_inserts"] >= 3
        first = [x.clone() for x in e.root_children()]
        from alphazero_quoridor_amd.boards import DeviceBoards
        e.set_boards(DeviceBoards.from_packed(boards, e.device), reset_trees=True)
        e.run_rounds(ev, 1, max_playouts=1, auto_finish=False)
        st2 = e.stats()
        assert st2["memo_hits"] - st1["memo_hits"] == 192, "the probe of an unexpanded root must find what the first run stored"
        assert st2["nn_evals"] == st1["nn_evals"]
        _, _, prior, _ = e.root_children()
        assert torch.equal(prior, first[2])  # the root's priors from the memo are the network's bits
    finally:
        e.close()


def test_captured_rounds_follow_the_weights(gpu_device):
    """capture_rounds -> weight change -> replay: the replayed launches read the new weight images AND the new
    per-layer scales (device memory), so the graph run equals an eager run after the same change."""
    boards = _mixed_boards(128, seed=8)
    NP = 16
    net = _net(gpu_device, 5)
    ev = net.evaluator("per_leaf")
    g = _engine(boards, NP, seed=2)
    e = _engine(boards, NP, seed=2)
    try:
        n0 = g.capture_rounds(ev, rounds=4, max_playouts=4, warmup=2)
        e.run_rounds(ev, n0, max_playouts=4)   # the eager rounds capture_rounds has run (warm-up + one between its two captures)
        with torch.no_grad():
            for prm in net.policy_value_net.parameters():
                prm.mul_(2.7)
        net.weights_changed()
        g.run_rounds(ev, 40, max_playouts=4)
        e.run_rounds(ev, 40, max_playouts=4)  # no graph on this engine: eager
        for x, y in zip(g.root_children(), e.root_children()):
            assert torch.equal(x, y)
        assert torch.equal(g.get_boards().meta, e.get_boards().meta)
        sg, se = g.stats(), e.stats()
        assert sg["playouts"] == se["playouts"] and sg["plies_played"] == se["plies_played"] and sg["plies_played"] > 0
    finally:
        g.close()
        e.close()


def test_configs4_800_playouts_per_move(gpu_device):
    """BASELINE configs[4] per GPU: n_playout=800 (the reference's README goal; train.py:20 defaults to 400).  Tree
    pools, descent records and the memo are sized from n_playout: 1,024 boards, real net.  Two plies in the lock-step
    cadence on two routes -- the plain engine (every leaf through the network) and the asynchronous loop's kernels with
    the memo -- must agree bit for bit on root visits, Q, priors, pi and sampled moves; 256 of the leaf boards in situ
    against the oracle; then the free-running asynchronous loop at
    this playout count keeps the tree invariants (nothing overflows, nothing is dropped, no runaway descent)."""
    import oracle
    from alphazero_quoridor_amd.engine import SelfPlayEngine

    B, NP = 1024, 800
    ev = _net(gpu_device, 12).evaluator("per_leaf")
    a = SelfPlayEngine(B, n_playout=NP, seed=31, device=gpu_device)
    b = SelfPlayEngine(B, n_playout=NP, seed=31, device=gpu_device)
    try:
        rng = np.random.RandomState(5)
        for ply in range(2):
            a.run_playouts(ev, NP)
            b.run_playouts_memo(ev)
            ra, rb = a.root_children(), b.root_children()
            for x, y in zip(ra, rb):
                assert torch.equal(x, y), ply
            visits, _, _, root_n = ra
            assert int(root_n.min()) >= NP and bool((visits.clamp(min=0).sum(dim=1) == root_n - 1).all())
            if ply == 1:  # in situ: the leaves of the next playout against the oracle (legal sets, terminal flags)
                leaf = a.select_boards().to_packed()
                mask = a.leaf_mask.cpu().numpy().view(np.uint32)
                term = a.leaf_term.cpu().numpy()
                idx = rng.choice(B, 256, replace=False)
                won = (leaf["p1"] >= 72) | (leaf["p2"] <= 8)
                assert np.array_equal(term[idx] != 0, won[idx])
                live = idx[term[idx] == 0]
                omask, status = oracle.movegen_batch(leaf[live])
                assert (status >= 0).all() and np.array_equal(mask[live], omask)
            (ma, pa), (mb, pb) = a.finish_move(), b.finish_move()
            assert torch.equal(ma, mb) and torch.equal(pa, pb), ply
            a.harvest()
            b.harvest()
        sa, sb = a.stats(), b.stats()
        for st in (sa, sb):
            assert st["node_overflow"] == 0 and st["games_aborted"] == 0 and st["nonfinite_values"] == 0 and st["runaway_descents"] == 0
        assert sa["playouts"] == sb["playouts"] == 2 * B * NP and sa["max_depth"] == sb["max_depth"]
        # the free-running loop: whole moves inside the launches, in-place re-roots, budget
        b.run_rounds(ev, 1200, max_playouts=256, budget_us=500)
        st = b.stats()
        assert st["node_overflow"] == 0 and st["games_aborted"] == 0 and st["runaway_descents"] == 0 and st["plies_played"] > 2 * B
        assert st["memo_hits"] + st["nn_evals"] + st["leaf_terminal"] >= st["playouts"]  # (leaves waiting for the network are counted at the miss)
        print("n_playout=800: %d plies, deepest descent %d levels, %d evaluations, %d memo hits, tree pages peak %d of %d"
              % (st["plies_played"], st["max_depth"], st["nn_evals"], st["memo_hits"], st["tree_pages_peak"], st["tree_pages_total"]))
    finally:
        a.close()
        b.close()


_RCCL_ONE_RANK = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["QZ_ROOT"]); sys.path.insert(0, os.path.join(os.environ["QZ_ROOT"], "tests", "golden"))
import torch.distributed as dist
from alphazero_quoridor_amd import dist as qdist
from alphazero_quoridor_amd.engine import SelfPlayEngine
from alphazero_quoridor_amd.policy_value_net import PolicyValueNet
rank, local, world = qdist.init_from_env("cuda")
assert world == 1 and dist.is_initialized() and dist.get_backend() == "nccl", (world, dist.is_initialized())
dev = torch.device("cuda", local)
torch.manual_seed(0)
net = PolicyValueNet(use_gpu=True, device=dev)
ev = net.evaluator("per_leaf")
# (1) the path's only exchange on RCCL: a real harvested batch, uint8 payload + int64 counts
eng = SelfPlayEngine(256, n_playout=8, seed=3, device=dev, fix_terminal_sign=True)
tbs = []
for _ in range(400):
    eng.run_rounds(ev, 8, max_playouts=16)
    tb = eng.harvest()
    if tb is not None:
        tbs.append(tb)
    if sum(t.n_games for t in tbs) >= 4:
        break
assert tbs, "no game finished"
buf = torch.cat([qdist.pack_tuples(t.boards.hbits, t.boards.vbits, t.boards.meta, t.pi, t.z) for t in tbs])
out, games = qdist.allgather_tuples(buf, n_games=sum(t.n_games for t in tbs))
assert out.dtype == torch.uint8 and out.shape == buf.shape and torch.equal(out, buf) and games == sum(t.n_games for t in tbs)
hb, vb, meta, pi, z = qdist.unpack_tuples(out)
assert torch.equal(pi, torch.cat([t.pi for t in tbs])) and torch.equal(meta, torch.cat([t.boards.meta for t in tbs]))
# (2) the training-side collectives: gradient all-reduce (flat float32 bucket), weight broadcast, buffer averaging
st = tbs[0].states()[:32]
n = st.shape[0]
pi_t = tbs[0].pi[:n]
z_t = tbs[0].z[:n]
net.optimizer.zero_grad(set_to_none=False)
logp, v = net.policy_value_net(st)
loss = torch.nn.functional.mse_loss(v.view(-1), z_t) - torch.mean(torch.sum(pi_t * logp, 1))
loss.backward()
before = [p.grad.clone() for p in net.policy_value_net.parameters() if p.grad is not None]
net._allreduce_grads()
after = [p.grad for p in net.policy_value_net.parameters() if p.grad is not None]
assert all(torch.equal(x, y) for x, y in zip(before, after)), "a 1-rank all-reduce + division by 1 changed a gradient"
w0 = [p.detach().clone() for p in net.policy_value_net.parameters()]
b0 = [b.detach().clone() for b in net.policy_value_net.buffers()]
net.sync_from_rank0()
net.average_buffers()
assert all(torch.equal(x, y) for x, y in zip(w0, net.policy_value_net.parameters()))
assert all(torch.equal(x, y) for x, y in zip(b0, net.policy_value_net.buffers()))
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("RCCL one-rank ok: %d tuples all-gathered, %d gradient tensors all-reduced" % (out.shape[0], len(after)))
'''


def test_collectives_on_rccl_with_one_rank(gpu_device):
    """Every collective of the package on the backend the multi-GPU job uses ("nccl" = RCCL on ROCm), with the ONE rank a
    1-GPU box allows (QZ_DIST_FORCE=1 makes the package create the process group and run the collectives at world size
    1): process-group init with device_id, uint8 / int64 all_gather_into_tensor of a really harvested tuple batch
    (byte-equal to the input), float32 all_reduce of the gradient bucket (gradients unchanged), broadcast of weights and
    averaging of buffers (unchanged).  In a child process: the test session itself never creates a process group."""
    import socket
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, QZ_ROOT=root, QZ_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("QZ_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK], env=env, capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout[-2000:])
    assert r.returncode == 0, r.stderr[-4000:]
    assert "RCCL one-rank ok" in r.stdout


def test_games_deeper_than_the_reference_can_recurse_are_dropped_and_counted(gpu_device):
    """qz_config.max_depth: the reference's recursive backup (mcts.py:55-62) ends the run with a RecursionError once a
    path has more than ~992 levels; the engine drops such a game and counts it.  With a small limit on late-game boards
    (deep forced lines) both engines -- lock-step and asynchronous -- drop games, count them under aborted_depth, restart
    the boards and keep going; no descent is ever longer than limit + 1 levels; and a limit that is never reached
    changes nothing (same harvested tuples as without a limit)."""
    from alphazero_quoridor_amd.engine import SelfPlayEngine
    from synth import synth_positions

    boards = synth_positions(256, seed=9, max_walls=8)
    boards["w1"] = 0
    boards["w2"] = 0
    ev = _net(gpu_device, 4).evaluator("per_leaf")
    NP, LIMIT = 48, 10
    lock = _engine(boards, NP, seed=2, max_depth=LIMIT)
    asyn = _engine(boards, NP, seed=2, max_depth=LIMIT)
    free = _engine(boards, NP, seed=2, max_depth=0)
    big = _engine(boards, NP, seed=2, max_depth=100000)
    try:
        for _ in range(30):
            lock.play_ply(ev)
            lock.harvest()
        asyn.run_rounds(ev, 400, max_playouts=NP, budget_us=0)
        asyn.harvest()
        for eng in (lock, asyn):
            st = eng.stats()
            assert st["aborted_depth"] > 0 and st["games_aborted"] >= st["aborted_depth"] and st["max_depth"] <= LIMIT + 1, st
            assert st["node_overflow"] == 0 and st["runaway_descents"] == 0 and st["plies_played"] > 0
        fb, bb = [], []
        for _ in range(60):
            for eng, acc in ((free, fb), (big, bb)):
                eng.run_rounds(ev, 8, max_playouts=NP)
                tb = eng.harvest()
                if tb is not None:
                    acc.append(tb)
        assert free.stats()["aborted_depth"] == 0 == big.stats()["aborted_depth"]
        gf, gb = _games_by_slot(fb), _games_by_slot(bb)
        assert gf.keys() == gb.keys() and all(gf[k] == gb[k] for k in gf)
        print("depth limit %d: lock-step dropped %d games, asynchronous %d" % (LIMIT, lock.stats()["aborted_depth"], asyn.stats()["aborted_depth"]))
    finally:
        for eng in (lock, asyn, free, big):
            eng.close()
