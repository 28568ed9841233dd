"""ctypes binding of ``libqzero_hip.so`` (C ABI: ``include/qz_abi.h``).

The library is built in-tree by ``make -C alphazero_quoridor_amd/csrc`` (hipcc,
``--offload-arch=gfx950``) and loaded from ``alphazero_quoridor_amd/libqzero_hip.so``.
There is no CPU fallback anywhere in this package: a missing library raises
``QzLibraryError`` at load time and a missing GPU makes every compute entry point
return ``QZ_E_NO_DEVICE`` (raised as ``QzError``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqzero_hip.so")
CSRC = os.path.join(_HERE, "csrc")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "qz_abi.h")

ABI_VERSION = 7
N_ACTIONS = 140
PLANES = 26 * 81
MASK_WORDS = 5
NO_MOVE = 255

E_INVALID, E_NO_DEVICE, E_HIP, E_OOM, E_STATE = -1, -2, -3, -4, -5


class QzError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("libqzero_hip error %d: %s" % (code, text))
        self.code = code


class QzLibraryError(RuntimeError):
    pass


class qz_boards(C.Structure):
    _fields_ = [("hbits", C.c_void_p), ("vbits", C.c_void_p), ("meta", C.c_void_p)]


class qz_rules_opts(C.Structure):
    _fields_ = [("variant", C.c_int32), ("detour_pooled", C.c_int32), ("detour_wave", C.c_int32), ("enc_split_pct", C.c_int32)]


class qz_config(C.Structure):
    _fields_ = [
        ("n_boards", C.c_int32),
        ("n_playout", C.c_int32),
        ("c_puct", C.c_float),
        ("temp", C.c_float),
        ("dirichlet_alpha", C.c_float),
        ("noise_frac", C.c_float),
        ("seed", C.c_uint64),
        ("device", C.c_int32),
        ("is_selfplay", C.c_int32),
        ("fix_terminal_sign", C.c_int32),
        ("node_cap", C.c_int32),
        ("edge_cap", C.c_int32),
        ("max_plies", C.c_int32),
        ("tree_pool_pages", C.c_int32),
        ("traj_pool_pages", C.c_int32),
        ("traj_page_dwords", C.c_int32),
        ("rules", qz_rules_opts),
        ("select_opts", C.c_int32),
        ("memo_small_log2", C.c_int32),
        ("memo_big_log2", C.c_int32),
        ("compact_edges", C.c_int32),
        ("max_depth", C.c_int32),
    ]


class qz_dropped_game(C.Structure):
    _fields_ = [("hbits", C.c_uint64), ("vbits", C.c_uint64), ("meta", C.c_uint64), ("cause", C.c_int32), ("ply", C.c_int32),
                ("board", C.c_int32), ("reserved", C.c_int32)]


DROP_CAUSES = {1: "no_legal_move", 2: "depth", 3: "max_plies", 4: "trajectory_pool"}


class qz_stats(C.Structure):
    _fields_ = [
        ("games_finished", C.c_int64),
        ("plies_played", C.c_int64),
        ("playouts", C.c_int64),
        ("leaf_terminal", C.c_int64),
        ("node_overflow", C.c_int64),
        ("games_aborted", C.c_int64),
        ("pending_games", C.c_int64),
        ("pending_plies", C.c_int64),
        ("arena_bytes", C.c_int64),
        ("descent_levels", C.c_int64),
        ("max_nodes", C.c_int64),
        ("max_edges", C.c_int64),
        ("aborted_no_move", C.c_int64),
        ("aborted_max_plies", C.c_int64),
        ("aborted_pool", C.c_int64),
        ("bad_forced_moves", C.c_int64),
        ("nonfinite_values", C.c_int64),
        ("tree_pages_total", C.c_int64),
        ("tree_pages_in_use", C.c_int64),
        ("tree_pages_peak", C.c_int64),
        ("traj_pages_total", C.c_int64),
        ("traj_pages_in_use", C.c_int64),
        ("traj_pages_peak", C.c_int64),
        ("edges_scanned", C.c_int64),
        ("edges_expanded", C.c_int64),
        ("max_depth", C.c_int64),
        ("deep_descents", C.c_int64),
        ("deep_descents_cold", C.c_int64),
        ("deep_levels", C.c_int64),
        ("deep_levels_replayed", C.c_int64),
        ("rounds", C.c_int64),
        ("memo_hits", C.c_int64),
        ("nn_evals", C.c_int64),
        ("memo_inserts", C.c_int64),
        ("memo_locked", C.c_int64),
        ("open_rounds", C.c_int64),
        ("open_plies", C.c_int64),
        ("waiting_boards", C.c_int64),
        ("aborted_depth", C.c_int64),
        ("runaway_descents", C.c_int64),
        ("compact_slices", C.c_int64),
        ("miss_overflow", C.c_int64),
    ]


class qz_nn_weights(C.Structure):
    _fields_ = [
        ("hot9", C.c_void_p), ("base0", C.c_void_p), ("wd", C.c_void_p), ("gamma0", C.c_void_p), ("beta0", C.c_void_p),
        ("n_blocks", C.c_int32),
        ("w16", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),   # host arrays of device pointers
        ("inv_scale", C.c_void_p),                                          # device array
        ("w6_16", C.c_void_p),
        ("gamma6", C.c_void_p), ("beta6", C.c_void_p), ("w1t", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
        ("w3t", C.c_void_p), ("b3", C.c_void_p),
        ("eps", C.c_float),
        ("precision", C.c_int32),
    ]


def build(force: bool = False) -> str:
    """Compile libqzero_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in ("qz_kernels.hip", "qz_abi.hip", "qz_nn.hip", "qz_conv.hip", "qz_rules.h", "qz_movegen_pool.h", "qz_path_rows.h", "qz_device.h")] + [HEADER]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", CSRC, "-s"])
    return LIB_PATH


_P = C.c_void_p
_SIGNATURES = {
    "qz_version": (C.c_int, []),
    "qz_last_error": (C.c_char_p, []),
    "qz_device_count": (C.c_int, []),
    "qz_movegen": (C.c_int, [C.POINTER(qz_boards), C.c_int, _P, _P]),
    "qz_encode": (C.c_int, [C.POINTER(qz_boards), C.c_int, _P, _P]),
    "qz_movegen_encode": (C.c_int, [C.POINTER(qz_boards), C.c_int, _P, _P, _P]),
    "qz_movegen_encode_opts": (C.c_int, [C.POINTER(qz_boards), C.c_int, _P, _P, C.POINTER(qz_rules_opts), _P]),
    "qz_step": (C.c_int, [C.POINTER(qz_boards), _P, C.c_int, _P, _P, _P]),
    "qz_rollout_scratch_bytes": (C.c_int64, [C.c_int]),
    "qz_rollout": (C.c_int, [C.POINTER(qz_boards), C.c_int, C.c_int, C.c_uint64, _P, _P, _P]),
    "qz_engine_create": (C.c_int, [C.POINTER(qz_config), C.POINTER(_P)]),
    "qz_engine_destroy": (C.c_int, [_P]),
    "qz_engine_reset": (C.c_int, [_P, _P]),
    "qz_engine_set_boards": (C.c_int, [_P, C.POINTER(qz_boards), C.c_int, _P]),
    "qz_engine_get_boards": (C.c_int, [_P, C.POINTER(qz_boards), _P]),
    "qz_engine_set_temp": (C.c_int, [_P, C.c_float]),
    "qz_engine_get_plies": (C.c_int, [_P, _P, _P]),
    "qz_engine_dropped_games": (C.c_int, [_P, C.POINTER(qz_dropped_game), C.c_int, C.POINTER(C.c_int64), _P]),
    "qz_engine_set_rules_opts": (C.c_int, [_P, C.POINTER(qz_rules_opts)]),
    "qz_engine_set_playouts": (C.c_int, [_P, C.c_int]),
    "qz_mcts_select": (C.c_int, [_P, _P, _P, _P, _P]),
    "qz_mcts_descend": (C.c_int, [_P, _P]),
    "qz_mcts_leaf_inputs": (C.c_int, [_P, _P, _P, _P, _P]),
    "qz_mcts_select_boards": (C.c_int, [_P, C.POINTER(qz_boards), _P, _P, _P]),
    "qz_mcts_expand_backup": (C.c_int, [_P, _P, _P, _P]),
    "qz_mcts_expand_backup_descend": (C.c_int, [_P, _P, _P, _P]),
    "qz_mcts_root_pi": (C.c_int, [_P, _P, _P, _P]),
    "qz_mcts_root_children": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "qz_mcts_update_with_move": (C.c_int, [_P, _P, _P]),
    "qz_mcts_finish_move": (C.c_int, [_P, _P, _P, _P, _P]),
    "qz_harvest_counts": (C.c_int, [_P, C.POINTER(C.c_int64 * 2), _P]),
    "qz_harvest": (C.c_int, [_P, C.POINTER(qz_boards), _P, _P, _P, _P, C.c_int64, _P]),
    "qz_engine_stats": (C.c_int, [_P, C.POINTER(qz_stats), _P]),
    "qz_nn_instnorm_act": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_float, _P]),
    "qz_nn_instnorm_act_nhwc": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_float, _P]),
    "qz_nn_input_layer": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, C.c_float, _P]),
    "qz_engine_leaf_boards": (C.c_int, [_P, _P, _P]),
    "qz_nn_conv3x3_norm": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, _P, C.c_int, C.c_float, _P]),
    "qz_nn_trunk": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P, _P, _P, _P, C.c_float, C.c_int, _P]),
    "qz_nn_trunk_heads": (C.c_int, [_P, C.c_int64, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, _P]),
    "qz_nn_evaluate": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                 _P, C.c_float, _P]),
    "qz_nn_evaluate_w": (C.c_int, [_P, _P, C.c_int64, C.POINTER(qz_nn_weights), _P, _P, _P, _P]),
    "qz_selfplay_advance": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "qz_selfplay_leaf_rules": (C.c_int, [_P, _P]),
    "qz_selfplay_evaluate": (C.c_int, [_P, C.POINTER(qz_nn_weights), _P]),
    "qz_selfplay_round_tail": (C.c_int, [_P, _P]),
    "qz_selfplay_round": (C.c_int, [_P, C.POINTER(qz_nn_weights), C.c_int, C.c_int, C.c_int, _P]),
    "qz_selfplay_parity": (C.c_int, [_P]),
    "qz_selfplay_misses": (C.c_int, [_P, C.POINTER(qz_boards), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "qz_memo_flush": (C.c_int, [_P, _P]),
    "qz_nn_head": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, _P]),
    "qz_selftest_sqrt": (C.c_int, [_P, C.c_int, _P]),
}

_lib = None


def load():
    """Load the HIP library; never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QzLibraryError(
            "%s is missing. Build it with `make -C %s` (hipcc, gfx950) or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback." % (LIB_PATH, CSRC)
        )
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise QzLibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.qz_version() != ABI_VERSION:
        raise QzLibraryError("ABI version mismatch: library %d, binding %d (rebuild: make -C %s)" % (L.qz_version(), ABI_VERSION, CSRC))
    _lib = L
    return L


def check(rc: int):
    if rc != 0:
        raise QzError(rc, load().qz_last_error().decode("utf-8", "replace"))


def exported_symbols():
    return sorted(_SIGNATURES)


# packed 24-byte board record used on the host side (AoS view of the three SoA words)
PACKED_DTYPE = np.dtype(
    [("hbits", "<u8"), ("vbits", "<u8"), ("p1", "i1"), ("p2", "i1"), ("w1", "u1"), ("w2", "u1"),
     ("cur", "u1"), ("pad", "u1", (3,))]
)


def packed_to_soa(packed: np.ndarray):
    """[n] PACKED_DTYPE -> three contiguous int64 arrays (bit patterns of the u64 words)."""
    packed = np.ascontiguousarray(packed, dtype=PACKED_DTYPE).reshape(-1)
    w = packed.view(np.uint64).reshape(-1, 3)
    return (np.ascontiguousarray(w[:, 0]).view(np.int64), np.ascontiguousarray(w[:, 1]).view(np.int64),
            np.ascontiguousarray(w[:, 2]).view(np.int64))


def soa_to_packed(hb: np.ndarray, vb: np.ndarray, meta: np.ndarray) -> np.ndarray:
    w = np.stack([np.asarray(hb).view(np.uint64), np.asarray(vb).view(np.uint64), np.asarray(meta).view(np.uint64)], axis=1)
    return np.ascontiguousarray(w).view(PACKED_DTYPE).reshape(-1)
