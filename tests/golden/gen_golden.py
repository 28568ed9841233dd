#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Runs ONLY in the build container (needs /root/reference, which does not exist on the
GPU box).  It imports the reference's quoridor.py / mcts.py / policy_value_net.py,
drives them on seeded inputs and stores inputs + expected outputs as small .npz
files.  No reference source is copied: fixtures are data.

    python tests/golden/gen_golden.py [--only rules,pawn,positions,steps,mcts,episodes,net,train,rollouts,no_move,real_net_search]

The oracle (oracle/) and the HIP path are both checked against these files.
"""
from __future__ import annotations

import argparse
import contextlib
import io
import os
import random
import sys
import time
from multiprocessing import get_context

import numpy as np

REF = os.environ.get("QZ_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

from _stubs import PACKED_DTYPE, det_fill_state_dict, hash_policy_py, pack_fields, uniform_policy_py  # noqa: E402

MASK32 = 0xFFFFFFFF


def _ref():
    import quoridor  # noqa

    return quoridor


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def pack_game(g):
    return pack_fields(
        g._intersections, g._positions[1], g._positions[2], g._player1_walls_remaining,
        g._player2_walls_remaining, g.current_player,
    )


def set_game(g, inter, p1, p2, w1, w2, cur):
    g._intersections = np.array(inter, dtype=np.float64)
    g._positions = {1: int(p1), 2: int(p2)}
    g._player1_walls_remaining = int(w1)
    g._player2_walls_remaining = int(w2)
    g.current_player = int(cur)
    g.last_player = 2 if cur == 1 else 1


def game_from_packed(rec):
    q = _ref()
    g = q.Quoridor()
    inter = np.zeros(64)
    for ix in range(64):
        if (int(rec["hbits"]) >> ix) & 1:
            inter[ix] = 1
        if (int(rec["vbits"]) >> ix) & 1:
            inter[ix] = -1
    set_game(g, inter, rec["p1"], rec["p2"], rec["w1"], rec["w2"], rec["cur"])
    return g


def pad_actions(a):
    out = np.full(140, 255, dtype=np.uint8)
    out[: len(a)] = a
    return out


def random_walls(rng, k):
    """k walls placed with the static (overlap) rule only -- no path check."""
    inter = np.zeros(64, dtype=np.int8)
    tries = 0
    placed = 0
    while placed < k and tries < 1000:
        tries += 1
        ix = rng.randrange(64)
        o = rng.choice((1, -1))
        if inter[ix] != 0:
            continue
        if o == 1:
            if ix % 8 != 0 and inter[ix - 1] == 1:
                continue
            if ix % 8 != 7 and inter[ix + 1] == 1:
                continue
        else:
            if ix // 8 != 0 and inter[ix - 8] == -1:
                continue
            if ix // 8 != 7 and inter[ix + 8] == -1:
                continue
        inter[ix] = o
        placed += 1
    return inter


# --------------------------------------------------------------------------- F1 / F2
def gen_rules(out_dir):
    q = _ref()
    g = q.Quoridor()
    rng = random.Random(101)
    walls = [np.zeros(64, dtype=np.int8), np.ones(64, dtype=np.int8), -np.ones(64, dtype=np.int8)]
    for k in [1, 2, 3, 5, 8, 10, 12, 14, 16, 18, 20, 20, 24, 30, 40]:
        for _ in range(4):
            walls.append(random_walls(rng, k))
    walls = np.stack(walls)
    inter_out = np.zeros((len(walls), 81, 4), dtype=np.int8)
    for wi, w in enumerate(walls):
        wf = w.astype(np.float64)
        for t in range(81):
            d = g._get_intersections(wf, t)
            inter_out[wi, t] = [d["NW"], d["NE"], d["SE"], d["SW"]]
    np.savez_compressed(os.path.join(out_dir, "rules_intersections.npz"), walls=walls, out=inter_out)

    pw = walls[:3].tolist() + [walls[i] for i in range(3, len(walls), 2)]
    pw = np.stack([np.asarray(x, dtype=np.int8) for x in pw])[:32]
    pawn = np.full((len(pw), 81, 81, 2, 6), -1, dtype=np.int8)
    for wi, w in enumerate(pw):
        wf = w.astype(np.float64)
        for loc in range(81):
            for opp in range(81):
                if loc == opp:
                    continue
                for player in (1, 2):
                    v = g._valid_pawn_actions(wf, loc, opp, player)
                    pawn[wi, loc, opp, player - 1, : len(v)] = v
    np.savez_compressed(os.path.join(out_dir, "rules_pawn.npz"), walls=pw, out=pawn)

    # BFS reachability on random wall sets (incl. blocked ones)
    cases = []
    for _ in range(1500):
        w = random_walls(rng, rng.randrange(0, 28))
        p1 = rng.randrange(0, 72)
        p2 = rng.randrange(9, 81)
        if p1 == p2:
            continue
        wf = w.astype(np.float64)
        b1 = g._bfs_to_goal(wf, 8, p1, p2, player=1)
        b2 = g._bfs_to_goal(wf, 0, p2, p1, player=2)
        cases.append((w, p1, p2, b1, b2))
    np.savez_compressed(
        os.path.join(out_dir, "rules_bfs.npz"),
        walls=np.stack([c[0] for c in cases]),
        p1=np.array([c[1] for c in cases], dtype=np.int8),
        p2=np.array([c[2] for c in cases], dtype=np.int8),
        reach1=np.array([c[3] for c in cases], dtype=np.uint8),
        reach2=np.array([c[4] for c in cases], dtype=np.uint8),
    )
    print("rules: %d wall arrays, %d bfs cases" % (len(walls), len(cases)))


# --------------------------------------------------------------------------- F3 / F4 / F5
def _play_worker(seed):
    """One game of biased random legal play; returns recorded positions."""
    q = _ref()
    rng = random.Random(seed)
    g = q.Quoridor()
    recs = []
    wall_bias = rng.choice((0.25, 0.5, 0.8))
    for ply in range(500):
        over, _ = g.has_a_winner()
        if over:
            break
        acts = g.actions()
        mover_walls = g._player1_walls_remaining if g.current_player == 1 else g._player2_walls_remaining
        keep = mover_walls > 0 or rng.random() < 0.15
        pawn = [a for a in acts if a < 12]
        wall = [a for a in acts if a >= 12]
        if wall and (rng.random() < wall_bias or not pawn):
            a = rng.choice(wall)
        else:
            a = rng.choice(pawn)
        before = pack_game(g)
        st = g.state() if keep else None
        if mover_walls > 0:
            # skip the reference's wasted actions() inside step() is impossible; pay for it
            with quiet():
                done = g.step(a)
        else:
            with quiet():
                done = g.step(a)
        after = pack_game(g)
        over2, winner = g.has_a_winner()
        if keep:
            recs.append((before, pad_actions(acts), len(acts), a, after, int(done), int(winner or 0),
                         np.packbits(st.astype(np.uint8).reshape(-1))))
    return recs


def _synthetic_worker(seed):
    """Hand-built / synthetic positions: random walls + pawns anywhere legal for a live game."""
    q = _ref()
    rng = random.Random(seed)
    g = q.Quoridor()
    recs = []
    for i in range(40):
        k = rng.choice((0, 1, 2, 4, 8, 12, 16, 17, 18, 19, 20))
        inter = random_walls(rng, k)
        mode = rng.random()
        p1 = rng.randrange(0, 72)
        p2 = rng.randrange(9, 81)
        if mode < 0.4:  # adjacent pawns
            d = rng.choice((9, -9, 1, -1))
            p2 = p1 + d
            if not (9 <= p2 <= 80) or (abs(d) == 1 and p1 // 9 != p2 // 9):
                continue
        if p1 == p2:
            continue
        used = int(np.count_nonzero(inter))
        w1 = rng.randrange(0, 11)
        w2 = max(0, min(10, 20 - used - w1)) if rng.random() < 0.7 else rng.randrange(0, 11)
        cur = rng.choice((1, 2))
        if (cur == 1 and w1 == 0) or (cur == 2 and w2 == 0):
            if rng.random() < 0.7:
                if cur == 1:
                    w1 = rng.randrange(1, 11)
                else:
                    w2 = rng.randrange(1, 11)
        set_game(g, inter, p1, p2, w1, w2, cur)
        try:
            acts = g.actions()
        except IndexError:
            continue
        before = pack_game(g)
        st = g.state()
        a = rng.choice(acts) if acts else 255
        after, done, winner = before, 0, 0
        if acts:
            with quiet():
                done = g.step(a)
            after = pack_game(g)
            _, winner = g.has_a_winner()
        recs.append((before, pad_actions(acts), len(acts), a, after, int(done), int(winner or 0),
                     np.packbits(st.astype(np.uint8).reshape(-1))))
    return recs


def _edge_positions():
    """Quirk fixtures: off-board winning jumps, row-0 walls, corners."""
    q = _ref()
    g = q.Quoridor()
    recs = []

    def rec(inter, p1, p2, w1, w2, cur, force_action=None):
        set_game(g, inter, p1, p2, w1, w2, cur)
        acts = g.actions()
        before = pack_game(g)
        st = g.state()
        todo = acts if force_action is None else [force_action]
        for a in todo:
            set_game(g, inter, p1, p2, w1, w2, cur)
            with quiet():
                done = g.step(a)
            after = pack_game(g)
            _, winner = g.has_a_winner()
            recs.append((before, pad_actions(acts), len(acts), a, after, int(done), int(winner or 0),
                         np.packbits(st.astype(np.uint8).reshape(-1))))

    z = np.zeros(64, dtype=np.int8)
    # Q4: P1 on row 7 with P2 directly north on row 8 -> NN leaves the board and wins
    for c in range(9):
        rec(z, 63 + c, 72 + c, 0, 0, 1)
        rec(z, 63 + c, 72 + c, 3, 3, 1, force_action=4)
    # Q4: P2 on row 1 with P1 directly south on row 0 -> SS leaves the board and wins
    for c in range(9):
        rec(z, c, 9 + c, 0, 0, 2)
        rec(z, c, 9 + c, 2, 2, 2, force_action=5)
    # ordinary wins
    rec(z, 67, 40, 0, 0, 1)
    rec(z, 40, 13, 0, 0, 2)
    # row-0 walls (Q1) in front of the start tile, corners
    for ix in range(8):
        for o in (1, -1):
            w = z.copy()
            w[ix] = o
            rec(w, 4, 76, 0, 0, 1)
            rec(w, ix, 76, 0, 0, 1)
            rec(w, ix + 1, 76, 0, 0, 1)
    for p1, p2 in ((0, 80), (8, 72), (0, 9), (8, 17), (71, 80), (63, 72), (1, 0 + 9), (7, 8 + 9)):
        rec(z, p1, p2, 0, 0, 1)
        rec(z, p1, p2, 0, 0, 2)
    return recs


def gen_positions(out_dir, n_games, n_syn, procs):
    t0 = time.time()
    ctx = get_context("fork")
    with ctx.Pool(procs) as pool:
        play = pool.map(_play_worker, [1000 + i for i in range(n_games)], chunksize=1)
        syn = pool.map(_synthetic_worker, [5000 + i for i in range(n_syn)], chunksize=1)
    recs = [r for g in play for r in g] + [r for g in syn for r in g] + _edge_positions()
    np.savez_compressed(
        os.path.join(out_dir, "rules_positions.npz"),
        board=np.array([r[0] for r in recs], dtype=PACKED_DTYPE),
        actions=np.stack([r[1] for r in recs]),
        n_actions=np.array([r[2] for r in recs], dtype=np.int16),
        action=np.array([r[3] for r in recs], dtype=np.uint8),
        next_board=np.array([r[4] for r in recs], dtype=PACKED_DTYPE),
        done=np.array([r[5] for r in recs], dtype=np.uint8),
        winner=np.array([r[6] for r in recs], dtype=np.uint8),
        state_bits=np.stack([r[7] for r in recs]),
    )
    print("positions: %d records (%d from play, %d synthetic) in %.0fs"
          % (len(recs), sum(len(g) for g in play), sum(len(g) for g in syn), time.time() - t0))


# --------------------------------------------------------------------------- F4 (all actions)
def _all_steps_worker(rec):
    g = game_from_packed(rec)
    acts = g.actions()
    out = []
    for a in acts:
        g2 = game_from_packed(rec)
        with quiet():
            done = g2.step(a)
        _, winner = g2.has_a_winner()
        out.append((rec, a, pack_game(g2), int(done), int(winner or 0)))
    return out


def gen_steps(out_dir, procs, n_pos=96):
    pos = np.load(os.path.join(out_dir, "rules_positions.npz"))
    boards = pos["board"]
    rng = random.Random(7)
    idx = rng.sample(range(len(boards)), n_pos)
    ctx = get_context("fork")
    with ctx.Pool(procs) as pool:
        res = pool.map(_all_steps_worker, [boards[i] for i in idx], chunksize=1)
    rows = [r for x in res for r in x]
    np.savez_compressed(
        os.path.join(out_dir, "rules_steps.npz"),
        board=np.array([r[0] for r in rows], dtype=PACKED_DTYPE),
        action=np.array([r[1] for r in rows], dtype=np.uint8),
        next_board=np.array([r[2] for r in rows], dtype=PACKED_DTYPE),
        done=np.array([r[3] for r in rows], dtype=np.uint8),
        winner=np.array([r[4] for r in rows], dtype=np.uint8),
    )
    print("steps: %d transitions from %d positions" % (len(rows), n_pos))


# --------------------------------------------------------------------------- F6
def _mcts_worker(job):
    rec, policy_name, n_playout, c_puct, temp = job
    import mcts as ref_mcts

    g = game_from_packed(rec)
    pol = hash_policy_py if policy_name == "hash" else uniform_policy_py
    m = ref_mcts.MCTS(pol, c_puct=c_puct, n_playout=n_playout)
    with quiet():
        acts, probs = m.get_move_probs(g, temp=temp)
    ch = m._root._children
    visits = np.array([ch[a]._n_visits for a in acts], dtype=np.int32)
    qs = np.array([float(ch[a]._Q) for a in acts], dtype=np.float64)
    ps = np.array([np.float32(ch[a]._P) for a in acts], dtype=np.float32)

    def count(n, d):
        c = 1 if n._children else 0
        md = d
        for k in n._children.values():
            cc, dd = count(k, d + 1)
            c += cc
            md = max(md, dd)
        return c, md

    nodes, depth = count(m._root, 0)
    k = len(acts)
    pad = lambda x, dt: np.concatenate([np.asarray(x, dtype=dt), np.zeros(140 - k, dtype=dt)])  # noqa: E731
    return dict(
        board=rec, policy=policy_name, n_playout=n_playout, c_puct=c_puct, temp=temp, k=k,
        acts=pad(acts, np.int16), visits=pad(visits, np.int32), q=pad(qs, np.float64),
        p=pad(ps, np.float32), probs=pad(probs, np.float64), root_visits=m._root._n_visits,
        nodes=nodes, depth=depth,
    )


def gen_mcts(out_dir, procs):
    pos = np.load(os.path.join(out_dir, "rules_positions.npz"))
    boards, nact, done = pos["board"], pos["n_actions"], pos["done"]
    rng = random.Random(11)
    opening = pack_game(_ref().Quoridor())
    walls_idx = [i for i in range(len(boards)) if nact[i] > 20]
    nowall_idx = [i for i in range(len(boards)) if 0 < nact[i] <= 8]
    nearwin_idx = [i for i in range(len(boards)) if done[i] == 1]  # the recorded move ended the game
    chosen = [opening] + [boards[i] for i in rng.sample(walls_idx, 7)]
    chosen_fast = [boards[i] for i in rng.sample(nowall_idx, 16)] + [boards[i] for i in nearwin_idx[:24]]
    jobs = []
    for b in chosen:
        for pol in ("hash", "uniform"):
            for n in (8, 50, 200):
                if pol == "uniform" and n == 200:
                    continue
                jobs.append((b, pol, n, 5, 1.0))
    for b in chosen_fast:
        for pol in ("hash", "uniform"):
            for n in (8, 50, 400):
                jobs.append((b, pol, n, 5, 1.0))
        jobs.append((b, "hash", 100, 5, 1e-3))
        jobs.append((b, "hash", 60, 2.5, 0.5))
    jobs.sort(key=lambda j: -j[2] * (10 if int(j[0]["w1"]) + int(j[0]["w2"]) > 0 else 1))
    t0 = time.time()
    ctx = get_context("fork")
    with ctx.Pool(procs) as pool:
        res = pool.map(_mcts_worker, jobs, chunksize=1)
    np.savez_compressed(
        os.path.join(out_dir, "mcts_stub.npz"),
        board=np.array([r["board"] for r in res], dtype=PACKED_DTYPE),
        policy=np.array([r["policy"] for r in res]),
        n_playout=np.array([r["n_playout"] for r in res], dtype=np.int32),
        c_puct=np.array([r["c_puct"] for r in res], dtype=np.float64),
        temp=np.array([r["temp"] for r in res], dtype=np.float64),
        k=np.array([r["k"] for r in res], dtype=np.int32),
        acts=np.stack([r["acts"] for r in res]),
        visits=np.stack([r["visits"] for r in res]),
        q=np.stack([r["q"] for r in res]),
        p=np.stack([r["p"] for r in res]),
        probs=np.stack([r["probs"] for r in res]),
        root_visits=np.array([r["root_visits"] for r in res], dtype=np.int32),
        nodes=np.array([r["nodes"] for r in res], dtype=np.int32),
        depth=np.array([r["depth"] for r in res], dtype=np.int32),
    )
    print("mcts: %d searches in %.0fs" % (len(res), time.time() - t0))


# --------------------------------------------------------------------------- F7
def _episode_worker(job):
    seed, n_playout, policy_name = job
    import mcts as ref_mcts

    q = _ref()
    np.random.seed(seed)
    pol = hash_policy_py if policy_name == "hash" else uniform_policy_py
    player = ref_mcts.MCTSPlayer(pol, c_puct=5, n_playout=n_playout, is_selfplay=1)
    g = q.Quoridor()
    # re-state of quoridor.py:573-610's loop so the per-ply root statistics can be captured too;
    # the reference's own start_self_play is run below on the same seed and must agree.
    boards, moves, pis, players, root_n = [], [], [], [], []
    try:
        with quiet():
            g.reset()
            while True:
                boards.append(pack_game(g))
                players.append(g.current_player)
                move, probs = player.choose_action(g, temp=1.0, return_prob=1)
                root_n.append(player.mcts._root._n_visits)
                moves.append(move)
                pis.append(probs)
                g.step(move)
                end, winner = g.has_a_winner()
                if end:
                    player.reset_player()
                    break
                if len(moves) >= 3000:
                    return None
    except IndexError:
        return None  # reference crash on an off-board terminal leaf
    # now the reference's own episode loop on the same seed
    np.random.seed(seed)
    player2 = ref_mcts.MCTSPlayer(pol, c_puct=5, n_playout=n_playout, is_selfplay=1)
    with quiet():
        w2, data = q.Quoridor().start_self_play(player2, temp=1.0)
    data = list(data)
    assert w2 == winner and len(data) == len(moves)
    z = np.array([d[2] for d in data], dtype=np.float64)
    for i, d in enumerate(data):
        assert np.array_equal(d[1], pis[i])
    first_state_bits = np.packbits(data[0][0].astype(np.uint8).reshape(-1))
    last_state_bits = np.packbits(data[-1][0].astype(np.uint8).reshape(-1))
    return dict(seed=seed, n_playout=n_playout, policy=policy_name, winner=winner,
                boards=np.array(boards, dtype=PACKED_DTYPE), moves=np.array(moves, dtype=np.uint8),
                pis=np.stack(pis), players=np.array(players, dtype=np.uint8), z=z,
                root_n=np.array(root_n, dtype=np.int32), first_state_bits=first_state_bits,
                last_state_bits=last_state_bits)


def gen_episodes(out_dir, procs):
    jobs = [(21, 4, "hash"), (22, 6, "hash"), (23, 3, "uniform"), (24, 8, "hash"), (25, 5, "hash"),
            (26, 2, "hash"), (27, 4, "uniform"), (28, 6, "hash")]
    t0 = time.time()
    ctx = get_context("fork")
    with ctx.Pool(procs) as pool:
        res = pool.map(_episode_worker, jobs, chunksize=1)
    crashed = sum(r is None for r in res)
    res = [r for r in res if r is not None]
    out = {"n": np.array(len(res)), "crashed": np.array(crashed)}
    for i, r in enumerate(res):
        for k, v in r.items():
            out["e%d_%s" % (i, k)] = np.asarray(v)
    np.savez_compressed(os.path.join(out_dir, "episodes_stub.npz"), **out)
    print("episodes: %d (crashed %d), lengths %s in %.0fs"
          % (len(res), crashed, [len(r["moves"]) for r in res], time.time() - t0))


# --------------------------------------------------------------------------- F8
def gen_net(out_dir):
    import torch
    import warnings

    warnings.filterwarnings("ignore")
    torch.set_num_threads(4)
    from policy_value_net import PolicyValueNet

    pos = np.load(os.path.join(out_dir, "rules_positions.npz"))
    boards, nact = pos["board"], pos["n_actions"]
    rng = random.Random(5)
    live = [i for i in range(len(boards)) if nact[i] > 0 and 0 <= boards[i]["p1"] <= 71 and 9 <= boards[i]["p2"] <= 80]
    idx = rng.sample(live, 64)
    games = [game_from_packed(boards[i]) for i in idx]
    states = np.stack([g.state() for g in games]).astype(np.float32)

    def fresh():
        pvn = PolicyValueNet(use_gpu=False)
        sd = pvn.policy_value_net.state_dict()
        pvn.policy_value_net.load_state_dict(det_fill_state_dict(sd, seed=2024))
        return pvn

    # (a) eval mode, batched (the build's default serving mode)
    pvn = fresh()
    pvn.policy_value_net.eval()
    with torch.no_grad():
        logp, v = pvn.policy_value_net(torch.from_numpy(states))
    eval_logp, eval_v = logp.numpy().copy(), v.numpy().copy()
    # (b) reference policy_value(): module left in train mode, BN uses batch statistics
    pvn = fresh()
    train_p, train_v = pvn.policy_value(states)
    # (c) reference policy_value_fn(): train mode, batch of one
    leaf_acts, leaf_p, leaf_v = [], [], []
    for g in games:  # all 64 states through the per-leaf API (round 1 stored 16)
        pvn = fresh()
        ap, val = pvn.policy_value_fn(g)
        ap = list(ap)
        a = np.full(140, 255, dtype=np.uint8)
        p = np.zeros(140, dtype=np.float32)
        a[: len(ap)] = [x[0] for x in ap]
        p[: len(ap)] = [x[1] for x in ap]
        leaf_acts.append(a)
        leaf_p.append(p)
        leaf_v.append(float(val))
    np.savez_compressed(
        os.path.join(out_dir, "net_fixture.npz"),
        board=boards[idx], states=np.packbits(states.astype(np.uint8).reshape(64, -1), axis=1),
        fill_seed=np.array(2024), eval_logp=eval_logp, eval_v=eval_v, train_p=train_p, train_v=train_v,
        leaf_acts=np.stack(leaf_acts), leaf_p=np.stack(leaf_p), leaf_v=np.array(leaf_v, dtype=np.float32),
    )
    print("net: 64-state batch, eval + train-batch + 64 per-leaf outputs")


def gen_net_more(out_dir):
    """F8b: the per-leaf outputs of the reference's policy_value_fn (policy_value_net.py:145-164) on the same 64 states
    for two more weight sets: `trained_like` (_stubs.trained_like_state_dict: BatchNorm gammas in [0.1, 3], output
    channels of every layer spread over 100x) and `after_steps` (the det-fill weights after the three optimiser steps
    of the reference's own train_step on F9's minibatch; the full state_dict is stored, it cannot be regenerated
    without the reference)."""
    import torch
    import warnings

    warnings.filterwarnings("ignore")
    torch.set_num_threads(4)
    from policy_value_net import PolicyValueNet
    from _stubs import trained_like_state_dict

    fx = np.load(os.path.join(out_dir, "net_fixture.npz"))
    boards = fx["board"]
    games = [game_from_packed(b) for b in boards]

    def per_leaf(pvn):
        acts, ps, vs = [], [], []
        for g in games:
            ap, val = pvn.policy_value_fn(g)
            ap = list(ap)
            a = np.full(140, 255, dtype=np.uint8)
            p = np.zeros(140, dtype=np.float32)
            a[: len(ap)] = [x[0] for x in ap]
            p[: len(ap)] = [x[1] for x in ap]
            acts.append(a)
            ps.append(p)
            vs.append(float(val))
        return np.stack(acts), np.stack(ps), np.array(vs, dtype=np.float32)

    out = {"board": boards}
    # (1) trained-like
    pvn = PolicyValueNet(use_gpu=False)
    pvn.policy_value_net.load_state_dict(trained_like_state_dict(pvn.policy_value_net.state_dict(), seed=4711))
    out["trained_like_seed"] = np.array(4711)
    out["trained_like_acts"], out["trained_like_p"], out["trained_like_v"] = per_leaf(pvn)
    # (2) after the three optimiser steps of F9 (the reference's own train_step; IndexError after optimizer.step())
    tf = np.load(os.path.join(out_dir, "train_fixture.npz"))
    tgames = [game_from_packed(b) for b in tf["board"]]
    states = np.stack([g.state() for g in tgames])
    pi, z = tf["pi"].astype(np.float64), tf["z"].astype(np.float64)
    pvn = PolicyValueNet(use_gpu=False)
    pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), seed=2024))
    for lr in tf["lr"]:
        try:
            pvn.train_step(list(states), list(pi), list(z), float(lr))
            raise SystemExit("the reference's train_step returned")
        except IndexError:
            pass
    sd = pvn.policy_value_net.state_dict()
    for k, ref_sum in zip(tf["keys"], tf["sum"]):  # the same weights F9 recorded
        assert abs(float(sd[str(k)].double().sum()) - float(ref_sum)) <= 1e-9 * max(1.0, abs(float(ref_sum))), k
    for k, v in sd.items():
        out["after_steps_w_" + k.replace(".", "__")] = v.numpy()
    out["after_steps_acts"], out["after_steps_p"], out["after_steps_v"] = per_leaf(pvn)
    np.savez_compressed(os.path.join(out_dir, "net_fixture_more.npz"), **out)
    print("net (more weight sets): trained-like + after three optimiser steps, 64 per-leaf outputs each; |v| range",
          float(np.abs(out["trained_like_v"]).max()), float(np.abs(out["after_steps_v"]).max()))


# --------------------------------------------------------------------------- F9
def gen_train(out_dir):
    """Three consecutive PolicyValueNet.train_step calls of the REAL reference module on a
    128-tuple minibatch (policy_value_net.py:166-192).  Under a modern torch the reference's
    method raises IndexError on its last line (`loss.data[0]` on a 0-dim tensor) AFTER
    optimizer.step(), so the post-step weights come from the reference's own code; loss and
    entropy are recomputed here with the same three expressions (value_loss + policy_loss,
    entropy) on an identically initialised twin, whose weights are asserted bit-identical to the
    reference's after every step."""
    import torch
    import torch.nn.functional as F
    import warnings

    warnings.filterwarnings("ignore")
    torch.set_num_threads(4)
    from policy_value_net import PolicyValueNet, set_learning_rate

    pos = np.load(os.path.join(out_dir, "rules_positions.npz"))
    boards, nact, acts = pos["board"], pos["n_actions"], pos["actions"]
    rng = random.Random(9)
    live = [i for i in range(len(boards)) if nact[i] > 0 and 0 <= boards[i]["p1"] <= 71 and 9 <= boards[i]["p2"] <= 80]
    idx = rng.sample(live, 128)
    games = [game_from_packed(boards[i]) for i in idx]
    states = np.stack([g.state() for g in games])  # float64 like the replay buffer's tuples
    rs = np.random.RandomState(77)
    pi = np.zeros((128, 140), dtype=np.float64)
    for j, i in enumerate(idx):
        a = acts[i][: nact[i]].astype(np.int64)
        pi[j, a] = rs.dirichlet(0.5 * np.ones(len(a)))
    pi = pi.astype(np.float32).astype(np.float64)   # what the engine's float32 trajectories hold
    z = rs.choice([-1.0, 1.0], size=128)
    lrs = [2e-3, 2e-3 * 1.5, 2e-3 / 1.5]

    def fresh():
        pvn = PolicyValueNet(use_gpu=False)
        pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), seed=2024))
        return pvn

    ref, twin = fresh(), fresh()
    losses, entropies = [], []
    grad_keys = ["conv1.weight", "res1.conv1.weight", "res5.conv2.weight", "res3.bn1.weight", "conv2.weight", "conv3.weight",
                 "fc1.weight", "fc2.weight", "fc3.weight", "fc3.bias", "bn1.bias"]
    grads0 = {}
    for lr in lrs:
        try:
            ref.train_step(list(states), list(pi), list(z), lr)
            raise SystemExit("the reference's train_step returned: torch is old enough, record its outputs directly")
        except IndexError:
            pass  # `loss.data[0]`: raised after optimizer.step()
        if not grads0:  # the gradients of the reference's own backward() of step 0 (same weights on every machine)
            named = dict(ref.policy_value_net.named_parameters())
            grads0 = {k: named[k].grad.detach().clone().numpy() for k in grad_keys}
        sb, pb, wb = (torch.FloatTensor(np.asarray(x)) for x in (states, pi, z))
        twin.optimizer.zero_grad()
        set_learning_rate(twin.optimizer, lr)
        logp, value = twin.policy_value_net(sb)
        loss = F.mse_loss(value.view(-1), wb) + (-torch.mean(torch.sum(pb * logp, 1)))
        loss.backward()
        twin.optimizer.step()
        entropy = -torch.mean(torch.sum(torch.exp(logp) * logp, 1))
        losses.append(float(loss.item()))
        entropies.append(float(entropy.item()))
        a, b = ref.policy_value_net.state_dict(), twin.policy_value_net.state_dict()
        assert all(torch.equal(a[k], b[k]) for k in a), "restated step diverged from the reference's own"
    sd = ref.policy_value_net.state_dict()
    keep = ["fc2.weight", "fc2.bias", "bn1.weight", "bn1.bias", "conv3.weight", "conv2.weight", "bn1.running_mean",
            "res5.bn2.running_var", "fc3.bias"]
    out = {"board": boards[idx], "pi": pi.astype(np.float32), "z": z.astype(np.float32), "lr": np.array(lrs),
           "loss": np.array(losses), "entropy": np.array(entropies), "fill_seed": np.array(2024),
           "keys": np.array(list(sd.keys())),
           "sum": np.array([float(v.double().sum()) for v in sd.values()]),
           "abs_sum": np.array([float(v.double().abs().sum()) for v in sd.values()])}
    for k in keep:
        out["w_" + k.replace(".", "_")] = sd[k].numpy()
    for k, g in grads0.items():
        out["g0_" + k.replace(".", "_")] = g
    # the same gradients in float64 (the reference's module, weights and loss expressions, double precision): the yardstick
    # for how far an fp32 backward -- the reference's own on the CPU, ours on the GPU -- is from the exact value
    ref64 = fresh()
    ref64.policy_value_net.double()
    sb, pb, wb = (torch.from_numpy(np.asarray(x, dtype=np.float64)) for x in (states, pi, z))
    logp, value = ref64.policy_value_net(sb)
    (F.mse_loss(value.view(-1), wb) + (-torch.mean(torch.sum(pb * logp, 1)))).backward()
    named64 = dict(ref64.policy_value_net.named_parameters())
    for k in grads0:
        out["g64_" + k.replace(".", "_")] = named64[k].grad.detach().numpy()
    # old/new outputs of policy_value (batch statistics) on the same minibatch after the three steps
    p_after, v_after = ref.policy_value(states)
    out["p_after"], out["v_after"] = p_after, v_after
    np.savez_compressed(os.path.join(out_dir, "train_fixture.npz"), **out)
    print("train: 3 reference train_step calls on 128 tuples: loss", losses, "entropy", entropies)


# --------------------------------------------------------------------------- F10
def _rollout_worker(job):
    rec, seed, n = job
    import pure_mcts  # the reference's (sys.path[0] is the reference)

    np.random.seed(seed)
    m = pure_mcts.MCTS(pure_mcts.policy_value_fn, 5, 10)
    out = [0, 0, 0]
    with quiet():
        for _ in range(n):
            g = game_from_packed(rec)
            v = m._evaluate_rollout(g)  # pure_mcts.py:81-103, limit=1000
            out[{1: 0, -1: 1, 0: 2}[int(v)]] += 1
    return out


def gen_rollouts(out_dir, procs, per_position=320):
    """Outcome frequencies of the REAL reference's random rollouts (pure_mcts.MCTS._evaluate_rollout,
    limit 1000) from 10 late-middle-game positions: counts of +1 / -1 / 0 from the point of view of
    the side to move.  The HIP rollouts use another random stream, so the comparison is statistical."""
    pos = np.load(os.path.join(out_dir, "rules_positions.npz"))
    boards, nact = pos["board"], pos["n_actions"]
    rng = random.Random(21)
    cand = [i for i in range(len(boards)) if nact[i] > 0 and 0 <= boards[i]["p1"] <= 71 and 9 <= boards[i]["p2"] <= 80
            and int(boards[i]["w1"]) + int(boards[i]["w2"]) <= 6]
    idx = rng.sample(cand, 10)
    jobs = [(boards[i], 1000 + 17 * j + k, per_position // 4) for j, i in enumerate(idx) for k in range(4)]
    t0 = time.time()
    with get_context("fork").Pool(procs) as pool:
        res = pool.map(_rollout_worker, jobs)
    counts = np.array(res).reshape(10, 4, 3).sum(axis=1)
    np.savez_compressed(os.path.join(out_dir, "rollout_fixture.npz"), board=boards[idx], counts=counts, limit=np.array(1000))
    print("rollouts: 10 positions x %d reference rollouts in %.0fs:" % (per_position, time.time() - t0), counts.tolist())


# --------------------------------------------------------------------------- F11
def gen_no_move(out_dir):
    """Roots WITHOUT a legal move.  no_move_roots_input.npy holds root positions at which the engine dropped a game under
    qz_stats.aborted_no_move (harvested from a 4-playout self-play run on the GPU by benchmarks/capture_no_move_roots.py).
    The REFERENCE is asked about each of them: Quoridor.actions() must be [] (quoridor.py:138-157), and
    MCTSPlayer.choose_action (mcts.py:172-196) must take its `else` branch -- print "WARNING: the board is full" and return
    None, which start_self_play's `move, move_probs = ...` (quoridor.py:587) cannot unpack: a TypeError ends the
    reference's self-play there.  Stored with them: what the reference's own helper reports for the pawn
    (_valid_pawn_actions on the root) and whether the game is over (it is not)."""
    import mcts as ref_mcts

    q = _ref()
    boards = np.load(os.path.join(out_dir, "no_move_roots_input.npy"))
    n = len(boards)
    n_act = np.zeros(n, dtype=np.int32)
    n_pawn = np.zeros(n, dtype=np.int32)
    warned = np.zeros(n, dtype=bool)
    returned_none = np.zeros(n, dtype=bool)
    unpack_raises = np.zeros(n, dtype=bool)
    over = np.zeros(n, dtype=bool)
    for i, rec in enumerate(boards):
        g = game_from_packed(rec)
        over[i] = bool(g.has_a_winner()[0])
        acts = g.actions()
        n_act[i] = len(acts)
        cur = g.current_player
        n_pawn[i] = len(g._valid_pawn_actions(g._intersections, g._positions[cur], g._positions[3 - cur], cur))
        player = ref_mcts.MCTSPlayer(uniform_policy_py, c_puct=5, n_playout=4, is_selfplay=1)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            r = player.choose_action(g, temp=1.0, return_prob=1)
        warned[i] = "the board is full" in buf.getvalue()
        returned_none[i] = r is None
        try:
            move, move_probs = r  # quoridor.py:587
        except TypeError:
            unpack_raises[i] = True
    np.savez_compressed(os.path.join(out_dir, "no_move_roots.npz"), board=boards, n_actions=n_act, n_pawn_actions=n_pawn,
                        game_over=over, prints_board_is_full=warned, returns_none=returned_none, unpack_raises_typeerror=unpack_raises)
    print("no_move: %d roots: reference actions() empty on %d, 'the board is full' on %d, None returned on %d, unpack TypeError on %d, "
          "game over on %d; movers with walls left: %d" % (n, int((n_act == 0).sum()), int(warned.sum()), int(returned_none.sum()),
                                                          int(unpack_raises.sum()), int(over.sum()),
                                                          int(sum(int(b["w1"] if b["cur"] == 1 else b["w2"]) > 0 for b in boards))))


# --------------------------------------------------------------------------- F9: the reference's search WITH ITS OWN NETWORK
def _real_net_worker(job):
    """One get_move_probs of the reference's MCTS (mcts.py:129-144) with the reference's PolicyValueNet(use_gpu=False) as
    the policy, weights from det_fill_state_dict(seed).  Under torch >= 0.4 `value.data[0][0]` (policy_value_net.py:163) is a
    0-dim float32 TENSOR, so TreeNode._Q becomes a float32 tensor after its first update (mcts.py:53) and Q + u is
    compared in float32 -- the SURVEY 8(c) hazard.  Everything the network returned during the search is recorded (board ->
    legal actions, priors, value), so that a search fed with exactly these evaluations can be compared with this one."""
    rec, n_playout, seed = job
    import warnings

    import torch

    warnings.filterwarnings("ignore")
    torch.set_num_threads(1)
    import mcts as ref_mcts
    from policy_value_net import PolicyValueNet

    pvn = PolicyValueNet(use_gpu=False)
    pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), seed=seed))
    table = {}

    def policy(game):
        ap, val = pvn.policy_value_fn(game)
        ap = list(ap)
        key = pack_game(game).tobytes()
        v32 = np.float32(float(val))
        assert float(v32) == float(val)  # (a float32 tensor: nothing is lost)
        ent = (np.array([a for a, _ in ap], dtype=np.uint8), np.array([q for _, q in ap], dtype=np.float32), v32)
        old = table.get(key)
        if old is not None:  # the evaluation is a pure function of the board (batch of one, train-mode BN): the memo's premise
            assert np.array_equal(old[0], ent[0]) and np.array_equal(old[1], ent[1]) and old[2] == ent[2]
        table[key] = ent
        return ap, val

    g = game_from_packed(rec)
    m = ref_mcts.MCTS(policy, c_puct=5, n_playout=n_playout)
    try:
        with quiet():
            acts, probs = m.get_move_probs(g, temp=1.0)
    except IndexError:  # quirk Q5: the reference's own crash on an off-board winning jump inside a playout
        return None
    ch = m._root._children
    visits = np.array([ch[a]._n_visits for a in acts], dtype=np.int32)
    qs = np.array([float(ch[a]._Q) for a in acts], dtype=np.float64)
    q_is_tensor = any(isinstance(ch[a]._Q, torch.Tensor) for a in acts)
    k = len(acts)
    pad = lambda x, dt: np.concatenate([np.asarray(x, dtype=dt), np.zeros(140 - k, dtype=dt)])  # noqa: E731
    keys = list(table.keys())
    return dict(
        board=rec, k=k, acts=pad(acts, np.int16), visits=pad(visits, np.int32), q32=pad(qs, np.float64), root_visits=m._root._n_visits,
        q_is_tensor=q_is_tensor,
        t_board=np.frombuffer(b"".join(keys), dtype=PACKED_DTYPE).copy(),
        t_k=np.array([len(table[x][0]) for x in keys], dtype=np.int32),
        t_acts=np.concatenate([table[x][0] for x in keys]), t_p=np.concatenate([table[x][1] for x in keys]),
        t_v=np.array([table[x][2] for x in keys], dtype=np.float32),
    )


def gen_real_net_search(out_dir, procs, n_playout=400, seed=2024):
    """real_net_search.npz: >= 64 searches of the reference with the reference's network: 64 late-game
    positions (nobody has a wall left: the regime of the leaf-evaluation memo and of k_lanes), 8 where only the mover is out of
    walls, 8 with walls in hand (140-wide nodes).  Stored: root visits / actions / root visit count, Q as the reference held it
    (float32 tensors, widened), and the evaluation table of every search as ragged arrays."""
    from synth import synth_positions

    q = _ref()

    def live(rec):
        g = game_from_packed(rec)
        return not g.has_a_winner()[0] and int(rec["p1"]) != int(rec["p2"]) and len(g.actions()) > 0

    late = synth_positions(200, seed=77, max_walls=14)
    late["w1"] = 0
    late["w2"] = 0
    mover_out = synth_positions(60, seed=78, max_walls=12)
    for r in mover_out:
        if int(r["cur"]) == 1:
            r["w1"], r["w2"] = 0, max(1, int(r["w2"]))
        else:
            r["w2"], r["w1"] = 0, max(1, int(r["w1"]))
    walls = synth_positions(60, seed=79, max_walls=10, mover_has_walls=True)
    jobs = []
    for group, want in ((walls, 8), (mover_out, 8), (late, 64)):
        jobs.append([(r.copy(), n_playout, seed) for r in group if live(r)][: want + 6])  # (a few spares for Q5 crashes)
    t0 = time.time()
    ctx = get_context("fork")
    res, kinds = [], []
    with ctx.Pool(procs) as pool:
        for kind, (grp, want) in enumerate(zip(jobs, (8, 8, 64))):
            out = [r for r in pool.map(_real_net_worker, grp, chunksize=1) if r is not None][:want]
            assert len(out) == want, (kind, len(out))
            res += out
            kinds += [kind] * want
    assert all(r["q_is_tensor"] for r in res)
    toff = np.concatenate([[0], np.cumsum([len(r["t_board"]) for r in res])]).astype(np.int64)
    np.savez_compressed(
        os.path.join(out_dir, "real_net_search.npz"),
        board=np.array([r["board"] for r in res], dtype=PACKED_DTYPE), kind=np.array(kinds, dtype=np.int8),
        n_playout=np.array(n_playout), c_puct=np.array(5.0), fill_seed=np.array(seed),
        k=np.array([r["k"] for r in res], dtype=np.int32), acts=np.stack([r["acts"] for r in res]),
        visits=np.stack([r["visits"] for r in res]), q32=np.stack([r["q32"] for r in res]),
        root_visits=np.array([r["root_visits"] for r in res], dtype=np.int32),
        t_off=toff, t_board=np.concatenate([r["t_board"] for r in res]), t_k=np.concatenate([r["t_k"] for r in res]),
        t_acts=np.concatenate([r["t_acts"] for r in res]), t_p=np.concatenate([r["t_p"] for r in res]),
        t_v=np.concatenate([r["t_v"] for r in res]),
    )
    print("real_net_search: %d searches x %d playouts of the reference with its own network (%d evaluations recorded) in %.0fs"
          % (len(res), n_playout, int(toff[-1]), time.time() - t0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="rules,positions,steps,mcts,episodes,net,train,net_more,rollouts,no_move")
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--games", type=int, default=160)
    ap.add_argument("--synthetic", type=int, default=120)
    args = ap.parse_args()
    only = set(args.only.split(","))
    if "rules" in only:
        gen_rules(HERE)
    if "positions" in only:
        gen_positions(HERE, args.games, args.synthetic, args.procs)
    if "steps" in only:
        gen_steps(HERE, args.procs)
    if "mcts" in only:
        gen_mcts(HERE, args.procs)
    if "episodes" in only:
        gen_episodes(HERE, args.procs)
    if "net" in only:
        gen_net(HERE)
    if "train" in only:
        gen_train(HERE)
    if "net_more" in only:
        gen_net_more(HERE)
    if "rollouts" in only:
        gen_rollouts(HERE, args.procs)
    if "no_move" in only:
        gen_no_move(HERE)
    if "real_net_search" in only:  # (not in the default list: ~10 minutes of the reference on 8 cores)
        gen_real_net_search(HERE, args.procs)


if __name__ == "__main__":
    main()
