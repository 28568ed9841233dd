"""GPU parity: HIP rules kernels (through the C ABI) vs the CPU oracle and the golden
fixtures recorded from the real reference.  Bit-exact: integer / bit work."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load_positions(golden_dir):
    d = np.load(golden_dir + "/rules_positions.npz")
    return d


def test_movegen_matches_golden_and_oracle(gpu_device, golden_dir):
    import oracle
    from alphazero_quoridor_amd import rules
    from alphazero_quoridor_amd.boards import DeviceBoards

    d = _load_positions(golden_dir)
    live = d["winner"] * 0 == 0  # every recorded position is a live board before its move
    boards = d["board"][live]
    db = DeviceBoards.from_packed(boards, gpu_device)
    mask = rules.movegen(db).cpu().numpy().view(np.uint32)
    # (1) golden: ordered action lists recorded from quoridor.Quoridor.actions()
    bad = 0
    for i in range(len(boards)):
        exp = list(d["actions"][i][: d["n_actions"][i]])
        got = rules.mask_to_actions(mask[i])
        if exp != got:
            bad += 1
    assert bad == 0, "%d of %d masks differ from the reference" % (bad, len(boards))
    # (2) oracle on the same inputs
    omask, status = oracle.movegen_batch(boards)
    assert (status >= 0).all()
    assert np.array_equal(omask, mask)


def test_encode_matches_golden_and_oracle(gpu_device, golden_dir):
    import oracle
    from alphazero_quoridor_amd import rules
    from alphazero_quoridor_amd.boards import DeviceBoards

    d = _load_positions(golden_dir)
    boards = d["board"]
    db = DeviceBoards.from_packed(boards, gpu_device)
    planes = rules.encode(db).cpu().numpy()
    assert set(np.unique(planes)) <= {0.0, 1.0}
    bits = np.packbits(planes.astype(np.uint8).reshape(len(boards), -1), axis=1)
    assert np.array_equal(bits, d["state_bits"])
    assert np.array_equal(planes, oracle.encode_batch(boards))


def test_fused_movegen_encode_equals_separate(gpu_device, golden_dir):
    from alphazero_quoridor_amd import rules
    from alphazero_quoridor_amd.boards import DeviceBoards

    d = _load_positions(golden_dir)
    boards = d["board"][:4099]  # not a multiple of the 4-board workgroup
    db = DeviceBoards.from_packed(boards, gpu_device)
    m1 = rules.movegen(db)
    p1 = rules.encode(db)
    m2, p2 = rules.movegen_encode(db)
    assert torch.equal(m1, m2) and torch.equal(p1, p2)


def test_step_matches_golden(gpu_device, golden_dir):
    import oracle
    from alphazero_quoridor_amd import rules
    from alphazero_quoridor_amd.boards import DeviceBoards

    for name in ("rules_positions.npz", "rules_steps.npz"):
        d = np.load(golden_dir + "/" + name)
        ok = d["action"] < 140
        boards, action = d["board"][ok], d["action"][ok]
        db = DeviceBoards.from_packed(boards, gpu_device)
        done, winner = rules.step(db, torch.from_numpy(action))
        got = db.to_packed()
        exp = d["next_board"][ok]
        assert np.array_equal(got.view(np.uint64), exp.view(np.uint64)), name
        assert np.array_equal(done.cpu().numpy(), d["done"][ok]), name
        assert np.array_equal(winner.cpu().numpy(), d["winner"][ok]), name
        ob, odone, owin = oracle.step_batch(boards, action)
        assert np.array_equal(ob.view(np.uint64), got.view(np.uint64))


def test_random_positions_vs_oracle(gpu_device):
    """Seeded synthetic positions (dense walls, adjacent pawns) at a size the oracle does in
    seconds; masks and planes must be identical."""
    import oracle
    from alphazero_quoridor_amd import _cabi, rules
    from alphazero_quoridor_amd.boards import DeviceBoards
    from synth import synth_positions

    boards = synth_positions(20000, seed=0x5EED)
    db = DeviceBoards.from_packed(boards, gpu_device)
    mask, planes = rules.movegen_encode(db)
    omask, status = oracle.movegen_batch(boards)
    assert (status >= 0).all()
    assert np.array_equal(mask.cpu().numpy().view(np.uint32), omask)
    assert np.array_equal(planes.cpu().numpy(), oracle.encode_batch(boards))


def test_edge_sizes(gpu_device):
    from alphazero_quoridor_amd import rules
    from alphazero_quoridor_amd.boards import DeviceBoards, opening_packed

    # empty batch
    db = DeviceBoards.from_packed(opening_packed(0), gpu_device)
    assert rules.movegen(db).shape == (0, 5)
    # single board: the opening has 131 legal actions, first ten as in SURVEY A.4
    db = DeviceBoards.from_packed(opening_packed(1), gpu_device)
    acts = rules.mask_to_actions(rules.movegen(db).cpu().numpy()[0])
    assert len(acts) == 131 and acts[:10] == [0, 2, 3, 12, 76, 13, 77, 14, 78, 15]
    planes = rules.encode(db).cpu().numpy()[0]
    assert planes.reshape(26, 81).sum(axis=1).tolist() == [64, 0, 0, 1, 1] + [0] * 9 + [81] + [0] * 9 + [81, 0]


def test_device_sqrt_is_correctly_rounded(gpu_device):
    """PUCT uses np.sqrt(parent visits) in float64 (mcts.py:69)."""
    from alphazero_quoridor_amd import _cabi

    n = 1 << 20
    out = torch.empty(n, dtype=torch.float64, device=gpu_device)
    _cabi.check(_cabi.load().qz_selftest_sqrt(out.data_ptr(), n, torch.cuda.current_stream().cuda_stream))
    assert np.array_equal(out.cpu().numpy(), np.sqrt(np.arange(n, dtype=np.float64)))


def _first_kernel(db, want_mask=True, want_planes=True):
    """The test-only first-generation kernel (tests/hip/libqz_testkernels.so): an independent HIP
    implementation of actions() + state(), not part of the product library."""
    import ctypes as C

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hip", "libqz_testkernels.so")
    assert os.path.exists(path), "build it with `make -C tests/hip` (or __graft_entry__.build())"
    T = C.CDLL(path)
    T.qzt_movegen_encode_v1.restype = C.c_int
    T.qzt_movegen_encode_v1.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4
    mask = torch.empty((db.n, 5), dtype=torch.int32, device=db.device) if want_mask else None
    planes = torch.empty((db.n, 26, 9, 9), dtype=torch.float32, device=db.device) if want_planes else None
    rc = T.qzt_movegen_encode_v1(db.hbits.data_ptr(), db.vbits.data_ptr(), db.meta.data_ptr(), db.n,
                                 mask.data_ptr() if want_mask else None, planes.data_ptr() if want_planes else None, None,
                                 torch.cuda.current_stream(db.device).cuda_stream)
    assert rc == 0
    return mask, planes


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32])
def test_every_movegen_kernel_variant_matches_the_oracle(gpu_device, golden_dir, variant):
    """The library picks k_wave_rules (3) for small batches and the pooled pipeline (tile sizes
    8..32) for large ones (0 at this batch size); 1 is the first kernel of round 1 (test-only library).  Force each on
    the same inputs through qz_rules_opts (odd batch size; terminal flags are exercised through
    the engine tests)."""
    import oracle
    from alphazero_quoridor_amd import rules
    from alphazero_quoridor_amd.boards import DeviceBoards
    from synth import synth_positions

    d = np.load(golden_dir + "/rules_positions.npz")
    boards = np.concatenate([d["board"][::3], synth_positions(3001, seed=31337)])
    omask, status = oracle.movegen_batch(boards)
    oplanes = oracle.encode_batch(boards)
    db = DeviceBoards.from_packed(boards, gpu_device)
    if variant == 1:
        mask, planes = _first_kernel(db)
        m_only, _ = _first_kernel(db, want_planes=False)
        _, p_only = _first_kernel(db, want_mask=False)
    else:
        opts = rules.rules_opts(variant)
        mask, planes = rules.movegen_encode(db, opts=opts)
        m_only, p_only = rules.movegen(db), rules.encode(db)
    assert np.array_equal(mask.cpu().numpy().view(np.uint32), omask)
    assert np.array_equal(planes.cpu().numpy(), oplanes)
    assert np.array_equal(m_only.cpu().numpy().view(np.uint32), omask)
    assert np.array_equal(p_only.cpu().numpy(), oplanes)


def test_c3_size_kernel_families_agree_and_match_oracle_sample(gpu_device):
    """BASELINE configs[2] size: 32,768 boards reached by random legal play on the GPU.  At this
    size the oracle is too slow for every board, so: (a) the three kernel families (pooled
    pipeline = default here, wave-per-board, first kernel) and the three group-detour modes must
    agree bit for bit on every mask and plane; (b) a strided sample of 2,048 boards is checked
    against the oracle; (c) the op is idempotent."""
    import sys

    import oracle
    from alphazero_quoridor_amd import rules

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "benchmarks"))
    from movegen_bench import position_set

    n = 32768
    for name in ("S-mid", "S-dense"):
        db = position_set(name, n, gpu_device)
        mask, planes = rules.movegen_encode(db)          # pooled pipeline, one detour group (the default)
        mask2, planes2 = rules.movegen_encode(db)
        assert torch.equal(mask, mask2) and torch.equal(planes, planes2)
        for split in (35, 70, 1, 100):   # wherever the encoder tiles sit in the two launches' grids: nothing changes
            m, p = rules.movegen_encode(db, opts=rules.rules_opts(0, enc_split_pct=split))
            assert torch.equal(m, mask) and torch.equal(p, planes), (name, split)
        mo = rules.movegen(db)                            # legal sets only: the second launch has no encoder tiles in front of its mask groups (they wait on the flags)
        assert torch.equal(mo, mask), name
        for mode in (0, 2):                               # no detours / three detour groups
            m, _ = rules.movegen_encode(db, opts=rules.rules_opts(0, detour_pooled=mode))
            assert torch.equal(m, mask), (name, mode)
        m, p = rules.movegen_encode(db, opts=rules.rules_opts(3))   # wave-per-board kernel
        assert torch.equal(m, mask) and torch.equal(p, planes), name
        for mode in (1, 2):
            m, _ = rules.movegen_encode(db, opts=rules.rules_opts(3, detour_wave=mode))
            assert torch.equal(m, mask), (name, "wave", mode)
        m, p = _first_kernel(db)                          # first kernel of the repo (test-only library)
        assert torch.equal(m, mask) and torch.equal(p, planes), (name, "first kernel")
        sample = db.to_packed()[::16]
        omask, status = oracle.movegen_batch(sample)
        assert (status >= 0).all()
        assert np.array_equal(mask.cpu().numpy().view(np.uint32)[::16], omask), name
        assert np.array_equal(planes.cpu().numpy()[::16], oracle.encode_batch(sample)), name
