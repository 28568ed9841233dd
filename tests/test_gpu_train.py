"""GPU tier for the rows next to the hot path: training step on device tensors
against the reference-generated fixture, device-resident replay + shard hand-over, pure-MCTS
rollouts (statistical parity with the reference's outcome frequencies) and the win-rate gate."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def test_train_step_on_device_matches_reference_fixture(gpu_device, golden_dir):
    """policy_value_net.py:166-192: three optimiser steps (Adam, weight decay 1e-4, three learning
    rates) on the reference's own 128-tuple minibatch, states re-encoded from the packed boards by
    the HIP encoder.  Step 0 (same weights on both sides) is the tight comparison: loss and entropy
    to 2e-5, and the GRADIENTS measured against their float64 value (fixture g64_*): no further from it
    than 4x the reference's own fp32 backward on the CPU is (fixture g0_*).  Adam's first steps are +-lr * sign-like, so an element whose gradient
    is at the rounding-noise level lands lr apart on the two machines and the later steps drift a
    little: 5e-5 / 2e-4 relative on the loss.  Post-step weights: the bulk must agree to 5e-5, no
    element may differ by more than one lr; the outputs three steps later are close, not equal.
    (The batch-norm reductions of the training pass run in float64 on the GPU, like PyTorch's CPU
    kernels do: with the library's fp32 sums the early layers' gradients are 2e-3 off.)"""
    from _stubs import det_fill_state_dict
    from alphazero_quoridor_amd import rules
    from alphazero_quoridor_amd.boards import DeviceBoards
    from alphazero_quoridor_amd.policy_value_net import PolicyValueNet

    d = np.load(golden_dir + "/train_fixture.npz")
    pvn = PolicyValueNet(use_gpu=True, device=gpu_device)
    pvn.policy_value_net.load_state_dict(det_fill_state_dict(pvn.policy_value_net.state_dict(), int(d["fill_seed"])))
    states = rules.encode(DeviceBoards.from_packed(d["board"], gpu_device))
    pi, z = torch.from_numpy(d["pi"]).to(gpu_device), torch.from_numpy(d["z"]).to(gpu_device)
    for i, lr in enumerate(d["lr"]):
        loss, ent = pvn.train_step_t(states, pi, z, float(lr))
        assert loss.is_cuda and loss.dim() == 0
        print("step %d: loss %.7f (ref %.7f)  entropy %.7f (ref %.7f)" % (i, float(loss), d["loss"][i], float(ent), d["entropy"][i]))
        tol = (2e-6, 5e-5, 2e-4)[i]
        assert abs(float(loss) - d["loss"][i]) < tol * abs(d["loss"][i]) and abs(float(ent) - d["entropy"][i]) < tol * 4.5
        if i == 0:
            # gradients of step 0 against the float64 value (fixture g64_*): the GPU's fp32 backward (MIOpen) may be no further
            # from it than 4x the reference's own fp32 backward on the CPU is (fixture g0_*), floor 2e-5 of the tensor's largest element
            named = dict(pvn.policy_value_net.named_parameters())
            worst = []
            for gk in [k for k in d.files if k.startswith("g0_")]:
                name = [n for n in named if n.replace(".", "_") == gk[3:]][0]
                g, ref, exact = named[name].grad.cpu().numpy().astype(np.float64), d[gk].astype(np.float64), d["g64_" + gk[3:]]
                sc = np.abs(exact).max()
                e_gpu, e_ref = np.abs(g - exact).max() / sc, np.abs(ref - exact).max() / sc
                print("grad %-20s |d|/max|g|: GPU fp32 vs float64 %.2e   reference CPU fp32 vs float64 %.2e" % (name, e_gpu, e_ref))
                worst.append((name, e_gpu, e_ref))
            for name, e_gpu, e_ref in worst:
                assert e_gpu < max(4.0 * e_ref, 2e-5), (name, e_gpu, e_ref)
    sd = pvn.get_policy_param()
    lr_max = float(max(d["lr"]))
    for k in ("fc2.weight", "bn1.weight", "conv3.weight", "conv2.weight"):
        diff = np.abs(sd[k].cpu().numpy() - d["w_" + k.replace(".", "_")]).reshape(-1)
        print(k, "median |dw| %.3g  p99 %.3g  max %.3g" % (np.median(diff), np.percentile(diff, 99), diff.max()))
        assert np.median(diff) < 5e-5 and diff.max() <= 1.05 * lr_max, k   # (an element whose gradient is rounding noise moves by +-lr either way)
    p, v = pvn.policy_value_t(states)
    # three sign-like Adam steps later the two machines' nets are close, not equal (v = tanh of a 128-term sum
    # whose weights differ by up to 3 lr each: single leaves move by tenths)
    dp, dv = np.abs(p.cpu().numpy() - d["p_after"]), np.abs(v.cpu().numpy() - d["v_after"])
    print("after 3 steps: |dp| median %.2e max %.2e   |dv| median %.2e max %.2e" % (np.median(dp), dp.max(), np.median(dv), dv.max()))
    assert dp.max() < 0.02 and np.median(dv) < 0.01 and dv.max() < 0.2


def test_replay_buffer_samples_reencoded_states_on_the_device(gpu_device, golden_dir, tmp_path):
    import random

    import oracle
    from alphazero_quoridor_amd import dist as qdist
    from alphazero_quoridor_amd import replay
    from alphazero_quoridor_amd.boards import DeviceBoards

    d = np.load(golden_dir + "/rules_positions.npz")
    b = d["board"]
    b = b[(b["p1"] >= 0) & (b["p1"] <= 71) & (b["p2"] >= 9) & (b["p2"] <= 80)][:700]
    db = DeviceBoards.from_packed(b, gpu_device)
    g = torch.Generator().manual_seed(0)
    pi = torch.softmax(torch.randn(700, 140, generator=g), dim=1).to(gpu_device)
    z = torch.sign(torch.randn(700, generator=g)).to(gpu_device)
    buf = replay.ReplayBuffer(capacity=512, device=gpu_device)
    buf.extend(qdist.pack_tuples(db.hbits, db.vbits, db.meta, pi, z))   # 700 into 512: the oldest 188 fall out
    assert len(buf) == 512
    random.seed(11)
    idx = random.sample(range(512), 128)
    random.seed(11)
    states, spi, sz = buf.sample(128)
    assert states.is_cuda and states.shape == (128, 26, 9, 9) and states.dtype == torch.float32
    src = [188 + i for i in idx]
    assert np.array_equal(states.cpu().numpy(), oracle.encode_batch(b[src]))
    assert torch.equal(spi, pi[src]) and torch.equal(sz, z[src])
    buf.save_shard(tmp_path / "x.qzr", n_games=5, n_playout=400)
    again = replay.ReplayBuffer(capacity=512, device=gpu_device)
    again.load_shard(tmp_path / "x.qzr")
    assert torch.equal(again.packed(), buf.packed())


def test_selfplay_shards_feed_a_separate_training_job(gpu_device, tmp_path):
    """Self-play job: TrainPipeline(shard_dir=...) cuts *.qzr files from the harvested tuples;
    training job: a fresh pipeline loads them and updates its net without playing a game."""
    import glob

    from alphazero_quoridor_amd.train import TrainPipeline

    torch.manual_seed(3)
    play = TrainPipeline(n_boards=64, seed=4, shard_dir=str(tmp_path))
    play.n_playout = 2
    with quiet():
        play.collect_selfplay_data(3)
    play.shards.flush()
    files = sorted(glob.glob(str(tmp_path / "*.qzr")))
    assert files
    learn = TrainPipeline(n_boards=64, seed=5)
    for f in files:
        learn.data_buffer.load_shard(f)
    assert len(learn.data_buffer) == len(play.data_buffer) and torch.equal(learn.data_buffer.packed(), play.data_buffer.packed())
    learn.batch_size, learn.epochs = 32, 2
    before = [p.detach().clone() for p in learn.policy_value_net.policy_value_net.parameters()]
    with quiet():
        loss, entropy = learn.policy_update()
    assert np.isfinite(loss) and np.isfinite(entropy) and learn.last_update["optimizer_steps"] >= 1
    assert any(not torch.equal(a, b) for a, b in zip(before, learn.policy_value_net.policy_value_net.parameters()))


def test_rollout_outcome_frequencies_match_the_reference(gpu_device, golden_dir):
    """pure_mcts.MCTS._evaluate_rollout (pure_mcts.py:81-103): 10 positions x 320 rollouts of the
    real reference vs 8,192 HIP rollouts per position (another random stream): the frequencies of
    +1 / -1 / 0 agree within 4.5 standard errors of the reference sample (+0.01)."""
    from alphazero_quoridor_amd.boards import DeviceBoards
    from alphazero_quoridor_amd.pure_mcts import rollout

    d = np.load(golden_dir + "/rollout_fixture.npz")
    N = 8192
    for j in range(len(d["board"])):
        ref = d["counts"][j].astype(np.float64)
        n_ref = ref.sum()
        boards = DeviceBoards.from_packed(np.repeat(d["board"][j:j + 1], N), gpu_device)
        v = rollout(boards, limit=int(d["limit"]), seed=100 + j).cpu().numpy()
        got = np.array([(v == 1).sum(), (v == -1).sum(), (v == 0).sum()], dtype=np.float64) / N
        p = ref / n_ref
        se = np.sqrt(np.maximum(p * (1 - p), 1e-4) / n_ref + got * (1 - got) / N)
        print("position %d: reference %s  hip %s" % (j, np.round(p, 3).tolist(), np.round(got, 3).tolist()))
        assert (np.abs(got - p) < 4.5 * se + 0.01).all(), (j, p, got)
        # every finished rollout really ended on a won board, every unfinished one did not
        packed = boards.to_packed()
        won = (packed["p1"] >= 72) | (packed["p2"] <= 8)
        assert np.array_equal(won, v != 0)


def test_rollout_edge_cases(gpu_device, golden_dir):
    from alphazero_quoridor_amd.boards import DeviceBoards, opening_packed
    from alphazero_quoridor_amd.pure_mcts import rollout

    # limit 1: the reference checks has_a_winner, then stops at i == limit-1 without moving (pure_mcts.py:88-90)
    b = DeviceBoards.from_packed(opening_packed(300), gpu_device)
    assert not rollout(b, limit=1).any() and b.to_packed().tobytes() == opening_packed(300).tobytes()
    # finished games return at once: +1 when the winner is the side "to move" (players are not rotated on a winning move)
    d = np.load(golden_dir + "/rules_positions.npz")
    nb = d["next_board"][d["done"] != 0][:50]
    assert len(nb) > 5
    db = DeviceBoards.from_packed(nb, gpu_device)
    v = rollout(db, limit=1000).cpu().numpy()
    win = np.where(nb["p2"] <= 8, 2, 1)
    assert np.array_equal(v, np.where(win == nb["cur"], 1, -1)) and db.to_packed().tobytes() == nb.tobytes()
    # same seed, same outcome; another seed, another stream
    x = np.repeat(d["board"][100:101], 512)
    a1 = rollout(DeviceBoards.from_packed(x, gpu_device), seed=5).cpu().numpy()
    a2 = rollout(DeviceBoards.from_packed(x, gpu_device), seed=5).cpu().numpy()
    a3 = rollout(DeviceBoards.from_packed(x, gpu_device), seed=6).cpu().numpy()
    assert np.array_equal(a1, a2) and not np.array_equal(a1, a3)


def test_pure_mcts_player_and_win_rate_gate(gpu_device):
    """pure_mcts.MCTSPlayer (uniform priors + rollout values, most visited child) and
    TrainPipeline.policy_evaluate (train.py:30-31, :108) end to end on a tiny budget."""
    from alphazero_quoridor_amd import pure_mcts
    from alphazero_quoridor_amd.quoridor import Quoridor
    from alphazero_quoridor_amd.train import TrainPipeline

    g = Quoridor()
    player = pure_mcts.MCTSPlayer(c_puct=5, n_playout=40, seed=2)
    with quiet():
        move = player.choose_action(g)
    assert move in g.actions()
    eng = player.mcts._engine
    st = eng.stats()
    assert st["playouts"] == 40 and st["node_overflow"] == 0
    # uniform priors over the 131 opening moves, as float32(1/131)
    eng.set_boards(eng.get_boards(), reset_trees=True)
    eng.run_playouts(player.mcts._evaluator, 3)
    prior = eng.root_children()[2].cpu().numpy()[0]
    assert np.count_nonzero(prior) == 131 and np.abs(prior[prior > 0] - 1.0 / 131.0).max() < 1e-9
    torch.manual_seed(0)
    tp = TrainPipeline(n_boards=8, seed=1)
    tp.n_playout, tp.pure_mcts_playout_num, tp.pure_mcts_rollout_limit = 6, 12, 40
    with quiet():
        ratio = tp.policy_evaluate(n_games=4, max_plies=120)
    assert 0.0 <= ratio <= 1.0


def test_interactive_loop_on_the_real_game(gpu_device):
    """game.play (the fixed game.py loop) with the drop-in Quoridor: a scripted player against the
    pure-MCTS player for a few plies, then step_result on a winning move (golden transition)."""
    from alphazero_quoridor_amd import pure_mcts
    from alphazero_quoridor_amd.agents import HistoricalAgent
    from alphazero_quoridor_amd.game import play, step_result
    from alphazero_quoridor_amd.quoridor import Quoridor

    class FirstLegal:
        def choose_action(self, game):
            return game.actions()[0]

    g = Quoridor()
    with quiet():
        winner, hist = play(g, {1: FirstLegal(), 2: pure_mcts.MCTSPlayer(n_playout=12, seed=3)}, max_plies=5, log=lambda *_: None)
    assert winner is None and [p for p, _ in hist] == [1, 2, 1, 2, 1] and g.current_player == 2
    assert HistoricalAgent("rec", [5, 6]).choose_action(g) == 5
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rules_positions.npz"))
    i = int(np.nonzero(d["done"])[0][0])
    won = Quoridor.from_packed(d["board"][i])
    with quiet():
        assert step_result(won, int(d["action"][i])) == (True, int(d["winner"][i]))
