"""GPU tier: the drop-in module surface (quoridor.Quoridor / mcts.MCTSPlayer /
policy_value_net.PolicyValueNet / train.TrainPipeline mirrors) against fixtures recorded from
the real reference.  These read like tests the reference would have had."""
import contextlib
import copy
import io

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def test_quoridor_surface(gpu_device, golden_dir):
    from alphazero_quoridor_amd.quoridor import Quoridor

    g = Quoridor()
    assert g.action_space == 140 and g.players == [1, 2] and g.get_current_player() == 1
    assert g._positions == {1: 4, 2: 76} and g._player1_walls_remaining == 10
    assert Quoridor.HORIZONTAL == 1 and Quoridor.VERTICAL == -1
    a = g.actions()
    assert len(a) == 131 and a[:10] == [0, 2, 3, 12, 76, 13, 77, 14, 78, 15]
    s = g.state()
    assert s.dtype == np.float64 and s.shape == (26, 9, 9)
    d = np.load(golden_dir + "/rules_positions.npz")
    for i in range(0, len(d["board"]), 401):
        q = Quoridor.from_packed(d["board"][i])
        assert q.actions() == d["actions"][i][: d["n_actions"][i]].tolist()
        assert np.array_equal(np.packbits(q.state().astype(np.uint8).reshape(-1)), d["state_bits"][i])
        if d["action"][i] < 140:
            q2 = copy.deepcopy(q)
            with quiet():
                done = q2.step(int(d["action"][i]))
            assert done == bool(d["done"][i])
            assert q2.packed()[0].tobytes() == d["next_board"][i].tobytes()
            assert q2.has_a_winner() == (bool(d["done"][i]), int(d["winner"][i]) or None)
            assert q.packed()[0].tobytes() == d["board"][i].tobytes()  # deepcopy really copied
    # safe=True raises on an illegal action (quoridor.py:167-169); clone() is a fresh game
    s = Quoridor(safe=True)
    with pytest.raises(ValueError):
        s.step(1)  # S from row 0
    assert s.clone()._positions == {1: 4, 2: 76}
    s.add_wall(3, 1)
    assert s._intersections[3] == 1 and 0 not in s.actions()  # H wall at ix 3 removes N from tile 4


def test_mcts_callback_route_matches_reference(gpu_device, golden_dir):
    """MCTS with an arbitrary policy_value_function (the reference's callback contract)."""
    from _stubs import hash_policy_py, uniform_policy_py
    from alphazero_quoridor_amd.mcts import MCTS
    from alphazero_quoridor_amd.quoridor import Quoridor

    d = np.load(golden_dir + "/mcts_stub.npz")
    idx = [i for i in range(len(d["board"])) if d["n_playout"][i] <= 50][::9][:12]
    assert len(idx) >= 8
    for i in idx:
        pol = hash_policy_py if str(d["policy"][i]) == "hash" else uniform_policy_py
        m = MCTS(pol, c_puct=float(d["c_puct"][i]), n_playout=int(d["n_playout"][i]))
        acts, probs = m.get_move_probs(Quoridor.from_packed(d["board"][i]), temp=float(d["temp"][i]))
        k = int(d["k"][i])
        assert list(acts) == d["acts"][i][:k].tolist()
        assert np.array_equal(probs, d["probs"][i][:k])  # host numpy softmax: bit for bit
        root = m._root
        assert root._n_visits == d["root_visits"][i]
        assert [root._children[a]._n_visits for a in acts] == d["visits"][i][:k].tolist()
        assert [root._children[a]._Q for a in acts] == d["q"][i][:k].tolist()


def test_start_self_play_reproduces_reference_episode(gpu_device, golden_dir):
    """Same np.random seed, same stub policy => the same game as the reference, move for
    move, with identical pi and z (Quoridor.start_self_play, quoridor.py:573-610)."""
    from _stubs import hash_policy_py, uniform_policy_py
    from alphazero_quoridor_amd.mcts import MCTSPlayer
    from alphazero_quoridor_amd.quoridor import Quoridor

    d = np.load(golden_dir + "/episodes_stub.npz")
    e = min(range(int(d["n"])), key=lambda e: len(d["e%d_moves" % e]) * int(d["e%d_n_playout" % e]))
    key = lambda k: d["e%d_%s" % (e, k)]  # noqa: E731
    pol = hash_policy_py if str(key("policy")) == "hash" else uniform_policy_py
    np.random.seed(int(key("seed")))
    player = MCTSPlayer(pol, c_puct=5, n_playout=int(key("n_playout")), is_selfplay=1)
    with quiet():
        winner, data = Quoridor().start_self_play(player, temp=1.0)
    data = list(data)
    assert winner == int(key("winner")) and len(data) == len(key("moves"))
    assert np.array_equal(np.stack([x[1] for x in data]), key("pis"))
    assert np.array_equal(np.array([x[2] for x in data]), key("z"))
    assert np.array_equal(np.packbits(data[0][0].astype(np.uint8).reshape(-1)), key("first_state_bits"))
    assert np.array_equal(np.packbits(data[-1][0].astype(np.uint8).reshape(-1)), key("last_state_bits"))
    assert data[0][0].dtype == np.float64 and data[0][1].dtype == np.float64


def _fixture_net(device):
    from _stubs import det_fill_state_dict
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

    pvn = PolicyValueNet(use_gpu=device.type == "cuda", device=device)
    pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), 2024))
    return pvn


def test_network_outputs_match_reference_fixture(gpu_device, golden_dir):
    """'within 1e-5 on the policy-value outputs for identical leaf batches' (fp32)."""
    from alphazero_quoridor_amd.boards import DeviceBoards
    from alphazero_quoridor_amd import rules
    from alphazero_quoridor_amd.quoridor import Quoridor

    TOL = 1e-5
    d = np.load(golden_dir + "/net_fixture.npz")
    pvn = _fixture_net(gpu_device)
    planes = rules.encode(DeviceBoards.from_packed(d["board"], gpu_device))  # leaf batch from the HIP encoder
    assert np.array_equal(np.packbits(planes.cpu().numpy().astype(np.uint8).reshape(64, -1), axis=1), d["states"])
    p, v = pvn.evaluator("eval")(planes)
    assert np.abs(p.cpu().numpy() - np.exp(d["eval_logp"])).max() < TOL
    assert np.abs(v.cpu().numpy() - d["eval_v"].reshape(-1)).max() < TOL
    p, v = pvn.evaluator("batch")(planes)
    assert np.abs(p.cpu().numpy() - d["train_p"]).max() < TOL and np.abs(v.cpu().numpy() - d["train_v"].reshape(-1)).max() < TOL
    pc, vc = pvn.evaluator("per_leaf", torch.float32, True)(planes)  # channels-last route, first layer from the planes
    p, v = pvn.evaluator("per_leaf", torch.float32, False)(planes)  # NCHW route
    assert (pc - p).abs().max().item() < TOL and (vc - v).abs().max().item() < TOL
    p, v, pc, vc = p.cpu().numpy(), v.cpu().numpy(), pc.cpu().numpy(), vc.cpu().numpy()
    for i in range(64):
        acts = d["leaf_acts"][i]
        k = int((acts != 255).sum())
        assert np.abs(pc[i][acts[:k]] - d["leaf_p"][i][:k]).max() < TOL and abs(vc[i] - d["leaf_v"][i]) < TOL
    for i in range(16):
        acts = d["leaf_acts"][i]
        k = int((acts != 255).sum())
        assert np.abs(p[i][acts[:k]] - d["leaf_p"][i][:k]).max() < TOL and abs(v[i] - d["leaf_v"][i]) < TOL
        # the reference-shaped single-leaf API (module in train mode, batch of one)
        ap, val = _fixture_net(gpu_device).policy_value_fn(Quoridor.from_packed(d["board"][i]))
        ap = list(ap)
        assert [a for a, _ in ap] == acts[:k].tolist()
        assert np.abs(np.array([x for _, x in ap]) - d["leaf_p"][i][:k]).max() < TOL and abs(val - d["leaf_v"][i]) < TOL
    # reference-shaped batched API
    pvn2 = _fixture_net(gpu_device)
    st = planes.cpu().numpy()
    ap, vv = pvn2.policy_value(st)
    assert np.abs(ap - d["train_p"]).max() < TOL and np.abs(vv - d["train_v"]).max() < TOL
    # per-leaf statistics are batch-invariant: the 4096-board engine sees the same numbers
    big = planes.repeat(8, 1, 1, 1)
    pb, vb = pvn.evaluator("per_leaf")(big)
    p1, v1 = pvn.evaluator("per_leaf")(planes[:1])
    assert np.abs(pb[0].cpu().numpy() - p1[0].cpu().numpy()).max() < 1e-6


def test_engine_route_evaluator_matches_reference_fixture_directly(gpu_device, golden_dir):
    """The evaluator configuration bench.py times -- the two-launch HIP evaluation from the packed boards
    (qz_nn_evaluate: first layer from the boards, ten trunk layers and the head convolution on the matrix
    cores with split fp16 operands, then the fully connected layers) -- against the outputs of the REAL
    reference's policy_value_fn (policy_value_net.py:145-164) on all 64 fixture states: 1e-5 on p and v, no
    intermediate route."""
    from alphazero_quoridor_amd.boards import DeviceBoards

    TOL = 1e-5
    d = np.load(golden_dir + "/net_fixture.npz")
    assert d["leaf_p"].shape[0] == 64
    pvn = _fixture_net(gpu_device)
    ev = pvn.evaluator("per_leaf", torch.float32, True)
    assert ev.accepts_leaf_boards and ev.board_input_layer and ev.fused_head and ev.fused_norm
    assert ev.mfma_trunk and ev.engine_route_ok()
    db = DeviceBoards.from_packed(d["board"], gpu_device)
    worst_p = worst_v = 0.0
    for rep in (1, 64):  # the 64 states once, and inside a 4,096-leaf batch (per-leaf statistics are batch-invariant)
        big = DeviceBoards(64 * rep, gpu_device)
        big.hbits, big.vbits, big.meta = db.hbits.repeat(rep), db.vbits.repeat(rep), db.meta.repeat(rep)
        p, v = ev(None, leaf=(big.struct(), 0, big.n))
        p, v = p.cpu().numpy(), v.cpu().numpy()
        for j in range(big.n):
            i = j % 64
            acts = d["leaf_acts"][i]
            k = int((acts != 255).sum())
            worst_p = max(worst_p, float(np.abs(p[j][acts[:k]] - d["leaf_p"][i][:k]).max()))
            worst_v = max(worst_v, float(abs(v[j] - d["leaf_v"][i])))
    print("engine-route evaluator vs reference policy_value_fn: max |dp| %.3g, max |dv| %.3g" % (worst_p, worst_v))
    assert worst_p < TOL and worst_v < TOL, (worst_p, worst_v)


def test_engine_route_evaluator_on_two_more_weight_sets(gpu_device, golden_dir):
    """one deterministic fill pinned a18-a20 with a 1.7x margin.  Two more weight sets against the REAL
    reference's policy_value_fn on the same 64 states (tests/golden/gen_golden.py:gen_net_more): `trained_like`
    (BatchNorm gammas in [0.1, 3], output channels of every layer spread over 100x: small channels push the lo halves
    of the split operands towards fp16 subnormals) and `after_steps` (the weights after the three optimiser steps of
    the reference's own train_step).  1e-5 on p and v through qz_nn_evaluate, the worst differences printed."""
    from _stubs import trained_like_state_dict
    from alphazero_quoridor_amd.boards import DeviceBoards
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

    TOL = 1e-5
    d = np.load(golden_dir + "/net_fixture_more.npz")
    db = DeviceBoards.from_packed(d["board"], gpu_device)
    for name in ("trained_like", "after_steps"):
        pvn = PolicyValueNet(use_gpu=True, device=gpu_device)
        sd = pvn.policy_value_net.state_dict()
        if name == "trained_like":
            new = trained_like_state_dict(sd, seed=int(d["trained_like_seed"]))
        else:
            new = {k: torch.from_numpy(d["after_steps_w_" + k.replace(".", "__")]) for k in sd}
        pvn.policy_value_net.load_state_dict(new)
        ev = pvn.evaluator("per_leaf", torch.float32, True)
        assert ev.mfma_trunk and ev.engine_route_ok()
        worst_p = worst_v = 0.0
        for rep in (1, 16):
            big = DeviceBoards(64 * rep, gpu_device)
            big.hbits, big.vbits, big.meta = db.hbits.repeat(rep), db.vbits.repeat(rep), db.meta.repeat(rep)
            p, v = ev(None, leaf=(big.struct(), 0, big.n))
            p, v = p.cpu().numpy(), v.cpu().numpy()
            for j in range(big.n):
                i = j % 64
                acts = d[name + "_acts"][i]
                k = int((acts != 255).sum())
                worst_p = max(worst_p, float(np.abs(p[j][acts[:k]] - d[name + "_p"][i][:k]).max()))
                worst_v = max(worst_v, float(abs(v[j] - d[name + "_v"][i])))
        print("%s weights, engine-route evaluator vs reference policy_value_fn: max |dp| %.3g, max |dv| %.3g" % (name, worst_p, worst_v))
        assert worst_p < TOL and worst_v < TOL, (name, worst_p, worst_v)


def test_config1_4096_boards_100_playouts_real_net_with_insitu_oracle_samples(gpu_device):
    """BASELINE configs[1]: 4,096 concurrent boards, n_playout=100, leaf batch 4,096, real
    (random-init) net on the engine route.  The boards are first spread over all game phases
    (short searches), then three full plies run; every 25 playout steps 256 of the CURRENT leaf
    boards (late-game positions, most movers out of walls) are checked against the oracle: legal
    masks and state planes bit for bit, terminal flags, and after every ply the tree invariants."""
    import oracle
    from alphazero_quoridor_amd.engine import SelfPlayEngine
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

    B, NP = 4096, 100
    torch.manual_seed(1)
    ev = PolicyValueNet(use_gpu=True, device=gpu_device).evaluator("per_leaf")
    assert ev.accepts_leaf_boards
    eng = SelfPlayEngine(B, n_playout=NP, seed=5, device=gpu_device)
    for _ in range(160):
        eng.run_playouts(ev, 2)
        eng.finish_move()
        eng.harvest()
    rng = np.random.RandomState(3)
    checked = walls_on_board = movers_without_walls = 0
    for ply in range(3):
        for step in range(NP):
            if step % 25 == 0:
                leaf = eng.select_boards().to_packed()          # the descent alone does not change the tree:
                planes = eng.select(want_mask=True).cpu().numpy()  # ... so this finds the same leaves
                mask = eng.leaf_mask.cpu().numpy().view(np.uint32)
                term = eng.leaf_term.cpu().numpy()
                idx = rng.choice(B, 256, replace=False)
                won = (leaf["p1"] >= 72) | (leaf["p2"] <= 8)
                assert np.array_equal(term[idx] != 0, won[idx])
                live = idx[term[idx] == 0]
                omask, status = oracle.movegen_batch(leaf[live])
                assert (status >= 0).all() and np.array_equal(mask[live], omask)
                assert np.array_equal(planes[live], oracle.encode_batch(leaf[live]))
                assert not planes[idx[term[idx] != 0]].any() and not mask[idx[term[idx] != 0]].any()
                checked += len(live)
                walls_on_board += int((20 - leaf["w1"][live].astype(int) - leaf["w2"][live].astype(int)).sum())
                movers_without_walls += int((np.where(leaf["cur"][live] == 1, leaf["w1"][live], leaf["w2"][live]) == 0).sum())
            eng.playout_step(ev)
        visits, _, _, root_n = eng.root_children()
        assert int(root_n.min()) >= NP
        assert bool((visits.clamp(min=0).sum(dim=1) == root_n - 1).all())  # every playout but the expanding one went through a child
        eng.finish_move()
        eng.harvest()
    st = eng.stats()
    assert st["node_overflow"] == 0 and st["games_aborted"] == 0 and st["nonfinite_values"] == 0
    assert st["playouts"] == B * (160 * 2 + 3 * NP)
    print("in-situ leaves checked: %d, mean walls on board %.1f, movers without walls %.0f%%"
          % (checked, walls_on_board / checked, 100.0 * movers_without_walls / checked))
    assert checked > 2000
    eng.close()


def test_device_route_equals_callback_route(gpu_device):
    """MCTS driven by PolicyValueNet.policy_value_fn: leaf planes -> net -> expand on the
    device gives the same tree as calling the Python callback per leaf."""
    from alphazero_quoridor_amd.mcts import MCTS
    from alphazero_quoridor_amd.quoridor import Quoridor

    pvn = _fixture_net(gpu_device)
    g = Quoridor()
    with quiet():
        for a in (0, 1, 20, 90):
            g.step(a)
    fast = MCTS(pvn.policy_value_fn, c_puct=5, n_playout=24)
    assert fast._evaluator is not None
    slow = MCTS(lambda game: pvn.policy_value_fn(game), c_puct=5, n_playout=24)
    assert slow._evaluator is None
    a1, p1 = fast.get_move_probs(g, temp=1.0)
    a2, p2 = slow.get_move_probs(g, temp=1.0)
    assert a1 == a2 and np.array_equal(p1, p2)
    # the device route replayed its playout steps as a HIP graph (3 warm-up + 2 x 8 replayed + 5 eager)
    assert fast._engine._graph is not None and fast._engine._graph_steps == 8
    a3, p3 = MCTS(pvn.policy_value_fn, c_puct=5, n_playout=24, use_graph=False).get_move_probs(g, temp=1.0)
    assert a3 == a1 and np.array_equal(p3, p1)


def test_train_pipeline_collects_reference_shaped_tuples(gpu_device):
    from alphazero_quoridor_amd.train import TrainPipeline

    torch.manual_seed(0)
    tp = TrainPipeline(n_boards=64, seed=3)
    assert (tp.n_playout, tp.c_puct, tp.temp, tp.buffer_size, tp.batch_size, tp.play_batch_size) == (400, 5, 1.0, 10000, 128, 1)
    tp.n_playout = 2
    tp.collect_selfplay_data(1)
    assert len(tp.data_buffer) >= tp.episode_len > 0
    assert tp.data_buffer.pi.is_cuda and tp.data_buffer.maxlen == 10000   # the replay ring lives on the device
    s, pi, z = tp.data_buffer.to_reference_tuples()[0]                   # ... and still yields the reference's tuple format
    assert s.shape == (26, 9, 9) and s.dtype == np.float64 and pi.shape == (140,) and pi.dtype == np.float64
    assert z in (1.0, -1.0) and abs(pi.sum() - 1.0) < 1e-5
    assert s.reshape(26, 81).sum(axis=1).tolist() == [64, 0, 0, 1, 1] + [0] * 9 + [81] + [0] * 9 + [81, 0]
    st = tp.engine().stats()
    assert st["games_finished"] >= 1 and st["node_overflow"] == 0


def test_fused_per_leaf_normalisation_kernel(gpu_device):
    """qz_nn_instnorm_act == BatchNorm2d(training) on a batch of one, for every sample
    (+ residual, + ReLU), incl. tiles that are not a multiple of 64 planes."""
    import torch.nn.functional as F
    from alphazero_quoridor_amd import _cabi

    L = _cabi.load()
    g = torch.Generator(device="cpu").manual_seed(3)
    for B, C in ((1, 64), (3, 2), (5, 4), (37, 64), (256, 64)):
        x = torch.randn((B, C, 9, 9), generator=g).to(gpu_device) * 3 + 1
        res = torch.randn((B, C, 9, 9), generator=g).to(gpu_device)
        gamma = (torch.rand(C, generator=g) + 0.5).to(gpu_device)
        beta = torch.randn(C, generator=g).to(gpu_device)
        for use_res in (False, True):
            for relu in (0, 1):
                ref = torch.stack([F.batch_norm(x[i:i + 1], None, None, gamma, beta, True, 0.0, 1e-5)[0] for i in range(B)])
                if use_res:
                    ref = ref + res
                if relu:
                    ref = F.relu(ref)
                out = torch.empty_like(x)
                _cabi.check(L.qz_nn_instnorm_act(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), res.data_ptr() if use_res else 0,
                                                 out.data_ptr(), B * C, C, relu, 1e-5, torch.cuda.current_stream().cuda_stream))
                assert (out - ref).abs().max().item() < 2e-5, (B, C, use_res, relu)
                inplace = x.clone()  # out may alias x
                _cabi.check(L.qz_nn_instnorm_act(inplace.data_ptr(), gamma.data_ptr(), beta.data_ptr(), res.data_ptr() if use_res else 0,
                                                 inplace.data_ptr(), B * C, C, relu, 1e-5, torch.cuda.current_stream().cuda_stream))
                assert torch.equal(inplace, out)
    # channels-last variant: same maths on NHWC memory
    for B, C in ((1, 64), (3, 6), (11, 6), (37, 64), (130, 64), (5, 2)):
        x = (torch.randn((B, C, 9, 9), generator=g) * 2 - 0.5).to(gpu_device).contiguous(memory_format=torch.channels_last)
        res = torch.randn((B, C, 9, 9), generator=g).to(gpu_device).contiguous(memory_format=torch.channels_last)
        gamma = (torch.rand(C, generator=g) + 0.5).to(gpu_device)
        beta = torch.randn(C, generator=g).to(gpu_device)
        for use_res in (False, True):
            for relu in (0, 1):
                xn = x.contiguous()  # reference on plain NCHW copies
                ref = torch.stack([F.batch_norm(xn[i:i + 1], None, None, gamma, beta, True, 0.0, 1e-5)[0] for i in range(B)])
                if use_res:
                    ref = ref + res.contiguous()
                if relu:
                    ref = F.relu(ref)
                out = x.clone(memory_format=torch.preserve_format)
                assert out.is_contiguous(memory_format=torch.channels_last)
                _cabi.check(L.qz_nn_instnorm_act_nhwc(out.data_ptr(), gamma.data_ptr(), beta.data_ptr(), res.data_ptr() if use_res else 0,
                                                      out.data_ptr(), B, C, relu, 1e-5, torch.cuda.current_stream().cuda_stream))
                assert (out - ref).abs().max().item() < 2e-5, ("nhwc", B, C, use_res, relu)
    # the evaluator with and without the fused kernel agrees
    pvn = _fixture_net(gpu_device)
    from alphazero_quoridor_amd.policy_value_net import LeafEvaluator
    xs = (torch.rand((130, 26, 9, 9), generator=g) > 0.8).float().to(gpu_device)
    p1, v1 = LeafEvaluator(pvn.policy_value_net, "per_leaf", fused_norm=True)(xs)
    p2, v2 = LeafEvaluator(pvn.policy_value_net, "per_leaf", fused_norm=False)(xs)
    assert (p1 - p2).abs().max().item() < 1e-5 and (v1 - v2).abs().max().item() < 1e-5
    p3, v3 = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True)(xs)  # NHWC end to end
    assert (p3 - p2).abs().max().item() < 1e-5 and (v3 - v2).abs().max().item() < 1e-5


def test_selfplay_then_policy_update_closes_the_loop(gpu_device):
    """collect_selfplay_data -> policy_update (train.py:65-92) -> the engine's cached evaluator
    weights follow the optimiser step (refreshed in place) -> self-play continues."""
    from alphazero_quoridor_amd.train import TrainPipeline

    torch.manual_seed(1)
    tp = TrainPipeline(n_boards=128, seed=9, n_groups=2)  # two board groups on two HIP streams
    tp.n_playout = 2
    tp.batch_size = 32
    tp.epochs = 2
    with quiet():
        tp.collect_selfplay_data(2)
    assert len(tp.data_buffer) > tp.batch_size
    ev = tp.engine().evaluators[0]
    x = (torch.rand((8, 26, 9, 9), device=gpu_device) > 0.8).float()
    before = ev(x)[0].clone()
    with quiet():
        loss, entropy = tp.policy_update()
    assert np.isfinite(loss) and np.isfinite(entropy) and 0.1 <= tp.lr_multiplier <= 10
    after = ev(x)[0]
    assert not torch.equal(before, after), "the evaluator must see the updated weights"
    # module forward (train-mode BN on a batch of one per sample) == evaluator, after the update
    with torch.no_grad():
        ref = torch.cat([torch.exp(tp.policy_value_net.policy_value_net(x[i:i + 1])[0]) for i in range(8)])
    assert (ref - after).abs().max().item() < 1e-5
    # the tables of the board input layer follow the weights as well (rebuilt in place)
    fresh = ev._input_tables(ev._layers[0][0])
    assert all(torch.equal(a, b) for a, b in zip(ev._in_tables, fresh))
    n0 = len(tp.data_buffer)
    with quiet():
        tp.collect_selfplay_data(1)
    assert len(tp.data_buffer) > n0 or len(tp.data_buffer) == tp.buffer_size


def test_board_groups_equal_independent_engines(gpu_device):
    """engine.BoardGroups (groups on their own HIP streams, results published to the caller's
    stream) must produce exactly what the same engines produce one after the other on one stream:
    moves, pi, harvested tuples."""
    from alphazero_quoridor_amd.engine import BoardGroups, SelfPlayEngine

    ev = _fixture_net(gpu_device).evaluator("per_leaf")
    kw = dict(n_playout=6, c_puct=5, temp=1.0, is_selfplay=1)
    grp = BoardGroups(96, 3, lambda: ev, seed=11, device=gpu_device, **kw)
    solo = [SelfPlayEngine(32, seed=BoardGroups.group_seed(11, g), device=gpu_device, **kw) for g in range(3)]
    assert BoardGroups.group_seed(11, 0) == 11 and len({BoardGroups.group_seed(11, g) for g in range(3)}) == 3
    n_tuples = 0
    for ply in range(150):
        got = grp.play_ply()
        tbs = grp.harvest()
        exp_tbs = []
        for g, eng in enumerate(solo):
            moves, pi = eng.play_ply(ev)
            assert torch.equal(got[g][0], moves), (ply, g)
            assert torch.equal(got[g][1], pi), (ply, g)
            tb = eng.harvest()
            if tb is not None:
                exp_tbs.append(tb)
        assert len(tbs) == len(exp_tbs)
        for a, b in zip(tbs, exp_tbs):
            assert a.n_games == b.n_games and torch.equal(a.pi, b.pi) and torch.equal(a.z, b.z)
            assert torch.equal(a.boards.meta, b.boards.meta) and torch.equal(a.game, b.game)
            n_tuples += len(a)
    st = grp.stats()
    assert st["playouts"] == sum(e.stats()["playouts"] for e in solo)
    grp.close()
    for e in solo:
        e.close()


def test_input_layer_from_boards_equals_conv_of_planes(gpu_device, golden_dir):
    """qz_nn_input_layer: relu(norm(conv1(state(board)))) computed from the 24-byte boards must
    equal the same layer applied to the encoder's planes (fp32 summation order aside), for live
    and terminal leaves, with per-leaf and with folded BatchNorm; end to end the evaluator's
    (p, v) stay within the 1e-5 of the parity contract."""
    from alphazero_quoridor_amd import rules
    from alphazero_quoridor_amd.boards import DeviceBoards
    from alphazero_quoridor_amd.policy_value_net import LeafEvaluator

    pos = np.load(str(golden_dir) + "/rules_positions.npz")
    packed = pos["board"][::7][:3000]
    db = DeviceBoards.from_packed(packed, gpu_device)
    n = db.n
    g = torch.Generator(device="cpu").manual_seed(5)
    term = (torch.rand(n, generator=g) < 0.05).to(torch.uint8).to(gpu_device)
    planes = rules.encode(db)
    planes[term.bool()] = 0.0  # a terminal leaf has all-zero planes (mcts.py:118-125 never evaluates it)
    pvn = _fixture_net(gpu_device)
    bn = pvn.policy_value_net.bn1  # make the folded-BN mode non-trivial
    with torch.no_grad():
        bn.running_mean.copy_(torch.linspace(-0.3, 0.4, 64))
        bn.running_var.copy_(torch.linspace(0.5, 2.0, 64))
    leaf = (db.struct(), term.data_ptr(), n)
    for mode in ("per_leaf", "eval"):
        ev = LeafEvaluator(pvn.policy_value_net, mode, channels_last=True)
        assert ev.accepts_leaf_boards
        x = planes.contiguous(memory_format=torch.channels_last)
        ref = ev._cbn(x, 0)                       # MIOpen convolution + fused normalisation on the planes
        got = ev._first_layer_from_boards(leaf)
        assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
        err = (got - ref).abs().max().item()
        assert err < 2e-5, (mode, err)
        p0, v0 = ev(planes)
        p1, v1 = ev(planes, leaf=leaf)
        assert (p0 - p1).abs().max().item() < 1e-5 and (v0 - v1).abs().max().item() < 1e-5, mode
    # the engine hands its leaf boards to such an evaluator by itself
    from alphazero_quoridor_amd.engine import SelfPlayEngine
    ev = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True)
    ev_planes = LeafEvaluator(pvn.policy_value_net, "per_leaf", channels_last=True, board_input_layer=False)
    eng = SelfPlayEngine(64, n_playout=16, seed=4, device=gpu_device)
    for _ in range(30):
        eng.playout_step(ev)
    pl = eng.select(want_mask=True)
    pa, va = ev(pl, leaf=eng.leaf_ref())
    pb, vb = ev_planes(pl)
    assert (pa - pb).abs().max().item() < 1e-5 and (va - vb).abs().max().item() < 1e-5
    eng.close()


def test_fused_head_kernel_equals_library_head(gpu_device):
    """qz_nn_head (conv2+conv3 merged, bn2/bn3 per leaf or folded, fc1/fc2/tanh, fc3/softmax) ==
    the same head through MIOpen / rocBLAS / torch ops, for ragged batch sizes (3 leaves per
    workgroup) and both BatchNorm modes."""
    from alphazero_quoridor_amd.policy_value_net import LeafEvaluator

    pvn = _fixture_net(gpu_device)
    with torch.no_grad():
        for bn in (pvn.policy_value_net.bn2, pvn.policy_value_net.bn3):
            bn.running_mean.copy_(torch.linspace(-0.2, 0.3, bn.num_features))
            bn.running_var.copy_(torch.linspace(0.6, 1.7, bn.num_features))
    g = torch.Generator(device="cpu").manual_seed(9)
    for mode in ("per_leaf", "eval"):
        fused = LeafEvaluator(pvn.policy_value_net, mode, channels_last=True)
        plain = LeafEvaluator(pvn.policy_value_net, mode, channels_last=True, fused_head=False)
        assert not plain.fused_head
        assert fused.fused_head and fused._head is not None
        for B in (1, 2, 3, 4, 7, 256, 1000):
            xs = (torch.rand((B, 26, 9, 9), generator=g) > 0.8).float().to(gpu_device)
            p1, v1 = fused(xs)
            p2, v2 = plain(xs)
            assert p1.shape == (B, 140) and v1.shape == (B,)
            assert (p1 - p2).abs().max().item() < 1e-5, (mode, B)
            assert (v1 - v2).abs().max().item() < 1e-5, (mode, B)
            assert (p1.sum(dim=1) - 1).abs().max().item() < 1e-5
