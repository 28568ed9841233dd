"""SelfPlayEngine -- thousands of concurrent Quoridor games, one MCTS tree each, on one GPU.

Host-side driver of the kernels behind ``include/qz_abi.h``.  One playout *step* advances
every board by one playout (mcts.py:103-127):

    qz_mcts_select          descend by PUCT, actions() + state() on the leaf   (HIP)
    leaf evaluator          policy-value net on the [B,26,9,9] leaf batch      (PyTorch-ROCm)
    qz_mcts_expand_backup   expand with the priors, back the value up          (HIP)

so the leaf batch is exactly B, and each tree sees its playouts strictly one after the other
(the reference's sequential semantics; no virtual loss).  ``n_playout`` steps make one ply;
``finish_move`` then turns root visits into pi, samples the move, records (board, pi),
re-roots the tree and steps the real board; finished games are harvested as
(board, pi, z) tuples and their slots restart immediately (continuous refill).

All launches go on torch's current stream, so a whole step can be captured into a HIP graph
(``capture_steps``) and replayed with one host call.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _cabi
from .boards import DeviceBoards


@dataclass
class TupleBatch:
    """Finished-game tuples in the compact wire format (588 B / ply): the board before the
    move (re-encode with ``states()`` to get the reference's float planes), pi, z."""

    boards: DeviceBoards
    pi: torch.Tensor      # [n,140] float32
    z: torch.Tensor       # [n] float32 (+1 recorded mover won / -1)
    game: torch.Tensor    # [n] int32, id of the game inside this batch
    n_games: int
    slot: torch.Tensor = None   # [n_games] int32: the engine's board every game was played on

    def __len__(self):
        return int(self.pi.shape[0])

    def states(self) -> torch.Tensor:
        from . import rules

        return rules.encode(self.boards)

    def to_reference_tuples(self):
        """list[(float64[26,9,9], float64[140], float64)] exactly like
        Quoridor.start_self_play's zip (quoridor.py:610) / TrainPipeline.data_buffer."""
        if len(self) == 0:
            return []
        st = self.states().cpu().numpy().astype(np.float64)
        pi = self.pi.cpu().numpy().astype(np.float64)
        z = self.z.cpu().numpy().astype(np.float64)
        return [(st[i], pi[i], z[i]) for i in range(len(self))]


# Under `python train.py` the reference's recursive backup (TreeNode.update_recursive, mcts.py:55-62) overflows Python's
# default recursion limit of 1,000 when a playout's path has more than 992 levels: the run ends with a RecursionError.
REFERENCE_MAX_DEPTH = 992

_hip = None


def _dev_copy(dst: torch.Tensor, src_ptr: int, nbytes: int):
    """engine-owned device memory -> a torch tensor, on torch's current stream (hipMemcpyAsync, device to device)"""
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
    rc = _hip.hipMemcpyAsync(C.c_void_p(dst.data_ptr()), C.c_void_p(src_ptr), C.c_size_t(nbytes), 3,
                             C.c_void_p(torch.cuda.current_stream(dst.device).cuda_stream))
    if rc != 0:
        raise _cabi.QzError(_cabi.E_HIP, "hipMemcpyAsync failed: %d" % rc)


class SelfPlayEngine:
    def __init__(self, n_boards, n_playout=400, c_puct=5.0, temp=1.0, is_selfplay=1, seed=0, device="cuda:0",
                 fix_terminal_sign=False, node_cap=0, edge_cap=0, max_plies=0, dirichlet_alpha=0.3, noise_frac=0.25,
                 tree_pool_pages=0, traj_pool_pages=0, traj_page_dwords=0, rules_opts=None, select_opts=0, memo=True,
                 memo_small_log2=0, memo_big_log2=0, compact_edges=0, max_depth=0):
        if not torch.cuda.is_available():
            raise _cabi.QzError(_cabi.E_NO_DEVICE, "no HIP device: the engine has no CPU path")
        self.L = _cabi.load()
        self.device = torch.device(device)
        self.n_boards = int(n_boards)
        self.n_playout = int(n_playout)
        cfg = _cabi.qz_config()
        cfg.n_boards = self.n_boards
        cfg.n_playout = self.n_playout
        cfg.c_puct = float(c_puct)
        cfg.temp = float(temp)
        cfg.dirichlet_alpha = float(dirichlet_alpha)
        cfg.noise_frac = float(noise_frac)
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.device = self.device.index if self.device.index is not None else 0
        cfg.is_selfplay = int(is_selfplay)
        cfg.fix_terminal_sign = int(bool(fix_terminal_sign))
        cfg.node_cap, cfg.edge_cap, cfg.max_plies = int(node_cap), int(edge_cap), int(max_plies)
        cfg.tree_pool_pages, cfg.traj_pool_pages = int(tree_pool_pages), int(traj_pool_pages)
        cfg.traj_page_dwords = int(traj_page_dwords)
        cfg.select_opts = int(select_opts)
        # leaf-evaluation memo of the asynchronous self-play loop (include/qz_abi.h): log2 of the bucket counts, 0 = auto
        cfg.memo_small_log2 = int(memo_small_log2) if memo else -1
        cfg.memo_big_log2 = int(memo_big_log2) if memo else -1
        cfg.max_depth = int(max_depth)  # drop a game whose descent exceeds this (0 = never; REFERENCE_MAX_DEPTH mirrors the reference's RecursionError)
        cfg.compact_edges = int(compact_edges)  # 0 = default; < 0: every move of the asynchronous loop copies its subtree
        if rules_opts is not None:
            cfg.rules = rules_opts
        self.cfg = cfg
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            torch.cuda.init()
            _cabi.check(self.L.qz_engine_create(C.byref(cfg), C.byref(h)))
        self.h = h
        B = self.n_boards
        dev = self.device
        # caller-owned I/O buffers of the step (device resident, reused every step); the planes
        # (34.5 MB per 4,096 boards) exist only if somebody asks for them: an evaluator that takes
        # the leaf boards (qz_nn_input_layer) never reads them
        self._planes = None
        self.leaf_mask = torch.zeros((B, 5), dtype=torch.int32, device=dev)
        self.leaf_term = torch.zeros(B, dtype=torch.uint8, device=dev)
        self.moves = torch.full((B,), 255, dtype=torch.uint8, device=dev)
        self.pi = torch.zeros((B, 140), dtype=torch.float32, device=dev)
        self._graph = None
        self._graph_steps = 0
        # None: the rules op writes state() only when the evaluator reads it (one that takes the leaf boards does not);
        # True: always (bench.py: every launch of a run then moves the op's full algorithmic bytes, timed or not)
        self.always_write_planes = None
        self._leaf_ref = None
        self._descended = False  # the last expand / backup launch also ran the next playout's descent
        self._memo_version = None   # (evaluator id, evaluator.version) the memo's contents belong to
        self._round_graph = None    # captured rounds of the asynchronous loop: ({parity: graph}, rounds, key)
        self._graph_warned = False
        self.graph_replays = 0      # replays of captured round graphs (a test asserts that a captured graph is actually used)

    @property
    def planes(self):
        if self._planes is None:
            self._planes = torch.zeros((self.n_boards, 26, 9, 9), dtype=torch.float32, device=self.device)
        return self._planes

    # ------------------------------------------------------------------ plumbing
    def _s(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def close(self):
        if getattr(self, "h", None):
            self._graph = None
            with torch.cuda.device(self.device):
                self.L.qz_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self._descended = False
        _cabi.check(self.L.qz_engine_reset(self.h, self._s()))

    def set_boards(self, boards: DeviceBoards, reset_trees=True):
        assert boards.n == self.n_boards
        self._descended = False
        _cabi.check(self.L.qz_engine_set_boards(self.h, boards.byref(), int(reset_trees), self._s()))

    def get_boards(self) -> DeviceBoards:
        out = DeviceBoards(self.n_boards, self.device)
        _cabi.check(self.L.qz_engine_get_boards(self.h, out.byref(), self._s()))
        return out

    def get_plies(self) -> torch.Tensor:
        """int32 [B]: moves played so far in every board's current game"""
        out = torch.empty(self.n_boards, dtype=torch.int32, device=self.device)
        _cabi.check(self.L.qz_engine_get_plies(self.h, out.data_ptr(), self._s()))
        return out

    def dropped_games(self, cap=4096):
        """The games dropped last (qz_engine_dropped_games) -> (packed root boards [k], causes list[str], plies int32 [k],
        board slots int32 [k], total drops so far).  Synchronises."""
        buf = (_cabi.qz_dropped_game * int(cap))()
        total = C.c_int64()
        k = self.L.qz_engine_dropped_games(self.h, buf, int(cap), C.byref(total), self._s())
        if k < 0:
            _cabi.check(k)
        hb = np.array([buf[i].hbits for i in range(k)], dtype=np.uint64)
        vb = np.array([buf[i].vbits for i in range(k)], dtype=np.uint64)
        meta = np.array([buf[i].meta for i in range(k)], dtype=np.uint64)
        packed = _cabi.soa_to_packed(hb.view(np.int64), vb.view(np.int64), meta.view(np.int64))
        return (packed, [_cabi.DROP_CAUSES[buf[i].cause] for i in range(k)], np.array([buf[i].ply for i in range(k)], dtype=np.int32),
                np.array([buf[i].board for i in range(k)], dtype=np.int32), int(total.value))

    def set_temp(self, temp: float):
        _cabi.check(self.L.qz_engine_set_temp(self.h, float(temp)))

    def set_playouts(self, n_playout: int):
        """n_playout for later moves of the asynchronous loop (captured round graphs stay valid: the kernels read it
        from their arguments at launch... so a captured graph keeps the OLD value: re-capture after changing it)."""
        _cabi.check(self.L.qz_engine_set_playouts(self.h, int(n_playout)))
        self.n_playout = int(n_playout)
        self._round_graph = None

    def set_rules_opts(self, opts=None):
        """qz_rules_opts for this engine's leaf rules op (None = library defaults)."""
        _cabi.check(self.L.qz_engine_set_rules_opts(self.h, C.byref(opts) if opts is not None else None))

    # ------------------------------------------------------------------ one playout step
    def select(self, want_mask=False, events=None, want_planes=True, tree_events=None):
        """-> leaf planes [B,26,9,9] (None with want_planes=False: actions() only, the leaf boards
        stay in the engine for qz_nn_input_layer).  want_mask also copies the leaf masks /
        terminal flags into self.leaf_mask / self.leaf_term.  events=(start, stop): torch events
        recorded around the rules-op launch only (bench.py's roofline measurement);
        tree_events=(start, stop): around the descent (k_select)."""
        mp = self.leaf_mask.data_ptr() if want_mask else 0
        tp = self.leaf_term.data_ptr() if want_mask else 0
        pp = self.planes.data_ptr() if want_planes else 0
        descended, self._descended = self._descended, False  # the previous step's fused launch has already walked the trees
        if events is None and tree_events is None and not descended:
            _cabi.check(self.L.qz_mcts_select(self.h, pp, mp, tp, self._s()))
        else:
            if tree_events is not None:
                tree_events[0].record()
            if not descended:
                _cabi.check(self.L.qz_mcts_descend(self.h, self._s()))
            if tree_events is not None:
                tree_events[1].record()
            if events is not None:
                events[0].record()
            _cabi.check(self.L.qz_mcts_leaf_inputs(self.h, pp, mp, tp, self._s()))
            if events is not None:
                events[1].record()
        return self._planes if want_planes else None

    def select_boards(self) -> DeviceBoards:
        """-> leaf boards (for host-side policy callbacks); fills leaf_mask / leaf_term."""
        out = DeviceBoards(self.n_boards, self.device)
        _cabi.check(self.L.qz_mcts_select_boards(self.h, out.byref(), self.leaf_mask.data_ptr(),
                                                 self.leaf_term.data_ptr(), self._s()))
        return out

    def expand_backup(self, p: torch.Tensor, v: torch.Tensor, events=None, then_descend=False):
        """then_descend: also run the NEXT playout's descent in the same launch (qz_mcts_expand_backup_descend);
        the next select() then only runs the rules op.  Only between two playouts of the same roots."""
        assert p.dtype == torch.float32 and v.dtype == torch.float32 and p.is_contiguous() and v.is_contiguous()
        assert p.shape == (self.n_boards, 140) and v.numel() == self.n_boards
        if events is not None:
            events[0].record()
        if then_descend:
            _cabi.check(self.L.qz_mcts_expand_backup_descend(self.h, p.data_ptr(), v.data_ptr(), self._s()))
        else:
            _cabi.check(self.L.qz_mcts_expand_backup(self.h, p.data_ptr(), v.data_ptr(), self._s()))
        self._descended = bool(then_descend)
        if events is not None:
            events[1].record()

    def leaf_ref(self):
        """(qz_boards struct, terminal-flag pointer, n) of the engine's current leaf boards: what a
        LeafEvaluator needs to compute its first layer straight from the boards
        (qz_nn_input_layer) instead of from the float planes."""
        if self._leaf_ref is None:
            st = _cabi.qz_boards()
            term = C.c_void_p()
            _cabi.check(self.L.qz_engine_leaf_boards(self.h, C.byref(st), C.byref(term)))
            self._leaf_ref = (st, term.value, self.n_boards)
        return self._leaf_ref

    def playout_step(self, evaluator, events=None, write_planes=None, tree_events=None, nn_events=None, more=False):
        """One playout of every board.  An evaluator that computes its first layer from the leaf
        boards never reads state(), so the rules op then only produces the legal sets
        (write_planes=True forces the planes anyway: bench.py's roofline of the full op).
        tree_events = ((start, stop) around k_select, (start, stop) around k_expand_backup);
        nn_events = (start, stop) around the evaluator's launches.  more=True: another playout of the same
        roots follows, so its descent runs in this step's expand / backup launch."""
        if getattr(evaluator, "takes_leaf_copy", False):  # e.g. pure_mcts.RolloutEvaluator: plays its copy of the leaves out
            assert not self._descended
            leaf = self.select_boards()
            p, v = evaluator.from_boards(leaf, self.leaf_mask)
            self.expand_backup(p, v)
            return
        takes_boards = getattr(evaluator, "accepts_leaf_boards", False)
        if write_planes is None:
            write_planes = self.always_write_planes
        want_planes = (not takes_boards) if write_planes is None else bool(write_planes) or not takes_boards
        planes = self.select(events=events, want_planes=want_planes, tree_events=None if tree_events is None else tree_events[0])
        if nn_events is not None:
            nn_events[0].record()
        if takes_boards:
            p, v = evaluator(planes, leaf=self.leaf_ref())
        else:
            p, v = evaluator(planes)
        if nn_events is not None:
            nn_events[1].record()
        self.expand_backup(p, v, events=None if tree_events is None else tree_events[1], then_descend=more)

    def capture_steps(self, evaluator, steps_per_graph=1, warmup=3):
        """Capture `steps_per_graph` playout steps (select -> net -> expand/backup) into one
        HIP graph.  The evaluator must be allocation-stable (static shapes)."""
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.playout_step(evaluator)
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                for i in range(steps_per_graph):
                    self.playout_step(evaluator, more=i + 1 < steps_per_graph)
        except Exception:
            # nothing of the capture has executed: the host-side "the descent already ran" flag must not survive it,
            # or the next eager select() would skip its descent and expand the warm-up's stale leaf a second time
            self._graph, self._graph_steps = None, 0
            raise
        finally:
            self._descended = False  # (the last captured step uses more=False; a failed capture ran nothing)
        self._graph, self._graph_steps = g, int(steps_per_graph)
        return warmup  # playouts already spent on the current roots

    def run_playouts(self, evaluator, n=None):
        """MCTS.get_move_probs's loop (mcts.py:135-139) for every board."""
        n = self.n_playout if n is None else int(n)
        done = 0
        if self._graph is not None:
            while n - done >= self._graph_steps:
                self._graph.replay()
                done += self._graph_steps
        for i in range(n - done):
            self.playout_step(evaluator, more=i + 1 < n - done)

    # ------------------------------------------------------------------ asynchronous self-play (qz_selfplay_*)
    def _memo_guard(self, evaluator):
        """The memo holds evaluations of ONE set of weights: flush it when the evaluator or its weights changed."""
        evaluator.ensure_fresh()
        key = (id(evaluator), evaluator.version)
        if key != self._memo_version:
            if self._memo_version is not None:
                _cabi.check(self.L.qz_memo_flush(self.h, self._s()))
            self._memo_version = key

    def selfplay_round(self, evaluator, max_playouts=64, budget_us=0, auto_finish=True):
        """One ROUND of the asynchronous loop (include/qz_abi.h): every board runs playouts on its own -- leaves whose
        evaluation is in the memo and terminal leaves are resolved in place, a board that has done n_playout playouts
        plays its move and goes on -- until it meets a leaf that needs the network; those leaves are evaluated as one
        compacted batch and stored in the memo (the legal sets of those leaves and the finished boards' moves are computed on
        the engine's second stream beside the network).  Nothing synchronises with the host.  With auto_finish the boards
        are on their own clocks from then on: select() / expand_backup() / finish_move() / update_with_move() raise
        QzError until reset() or set_boards(reset_trees=True)."""
        self._memo_guard(evaluator)
        self._descended = False
        _cabi.check(self.L.qz_selfplay_round(self.h, C.byref(evaluator.nn_weights()), int(max_playouts), int(budget_us), int(bool(auto_finish)),
                                             self._s()))

    def capture_rounds(self, evaluator, rounds=16, max_playouts=64, budget_us=0, auto_finish=True, warmup=2):
        """Capture `rounds` (even: the two miss counters alternate) rounds into HIP graphs for run_rounds -- ONE GRAPH PER
        MISS-COUNTER PARITY: a graph bakes in which of the two counters its first round appends to (a kernel argument), and a
        caller may arrive on either (an odd leftover of the last call, run_playouts_memo, the split entry points).  Round 4
        repaired the parity with one eager round, which left `n - 1 < rounds` for the graph whenever n == rounds --
        TrainPipeline's exact shape -- so the graph was silently never replayed."""
        assert rounds % 2 == 0
        self._memo_guard(evaluator)
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            for _ in range(2 * ((warmup + 1) // 2)):
                self.selfplay_round(evaluator, max_playouts, budget_us, auto_finish)
        torch.cuda.current_stream(self.device).wait_stream(s)
        graphs = {}
        for k in range(2):
            torch.cuda.synchronize(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(rounds):
                    self.selfplay_round(evaluator, max_playouts, budget_us, auto_finish)
            graphs[self.round_parity()] = g  # (a capture executes nothing: the parity is the one the graph's first round will find)
            if k == 0:  # one eager round puts the engine on the other counter for the second capture
                self.selfplay_round(evaluator, max_playouts, budget_us, auto_finish)
        assert len(graphs) == 2, "qz_selfplay_parity did not alternate between two eager rounds"
        self._round_graph = (graphs, int(rounds), (id(evaluator), int(max_playouts), int(budget_us), bool(auto_finish)))
        return 2 * ((warmup + 1) // 2) + 1  # eager rounds this call has run

    def run_rounds(self, evaluator, n, max_playouts=64, budget_us=0, auto_finish=True):
        """n rounds; whole multiples of a captured graph (capture_rounds, same arguments) are replayed -- the graph of the
        miss-counter parity the engine is on (an even number of rounds per replay keeps it there)."""
        n = int(n)
        done = 0
        g = self._round_graph
        if g is not None and n >= g[1]:
            if g[2] == (id(evaluator), int(max_playouts), int(budget_us), bool(auto_finish)):
                self._memo_guard(evaluator)
                while n - done >= g[1]:
                    g[0][self.round_parity()].replay()
                    self.graph_replays += 1
                    done += g[1]
            elif not self._graph_warned:
                self._graph_warned = True
                import warnings
                warnings.warn("run_rounds: the captured graph was made with other arguments (evaluator / max_playouts / budget_us / auto_finish): running eager")
        for _ in range(n - done):
            self.selfplay_round(evaluator, max_playouts, budget_us, auto_finish)

    def round_parity(self) -> int:
        """which of the two miss counters the next round uses (qz_selfplay_parity)"""
        return int(self.L.qz_selfplay_parity(self.h))

    def run_playouts_memo(self, evaluator, n=None):
        """MCTS.get_move_probs's loop for every board in the LOCK-STEP cadence of run_playouts (every board starts one
        playout per round; the host plays the move), but through the asynchronous loop's machinery: the network only
        sees the leaves the memo does not know.  n + 1 rounds: the last one only consumes evaluations."""
        n = self.n_playout if n is None else int(n)
        assert n == self.n_playout, "the device-side playout budget is the engine's n_playout"
        for _ in range(n + 1):
            self.selfplay_round(evaluator, max_playouts=1, auto_finish=False)

    def misses(self):
        """The miss list of the round in progress (between qz_selfplay_advance and qz_selfplay_round_tail; after a
        complete round: of the round just finished, count already cleared) as host arrays -- test / inspection helper.
        -> (packed boards [n], mask5 uint32 [n,5], p float32 [n,140], v float32 [n]).  Synchronises."""
        st = _cabi.qz_boards()
        n_dev, mk, pp, vv = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _cabi.check(self.L.qz_selfplay_misses(self.h, C.byref(st), C.byref(n_dev), C.byref(mk), C.byref(pp), C.byref(vv)))
        cnt = torch.zeros(1, dtype=torch.int32, device=self.device)
        _dev_copy(cnt, n_dev.value, 4)
        n = int(cnt.item())
        hb = torch.empty(n, dtype=torch.int64, device=self.device)
        vb, meta = torch.empty_like(hb), torch.empty_like(hb)
        mask = torch.empty((n, 5), dtype=torch.int32, device=self.device)
        p = torch.empty((n, 140), dtype=torch.float32, device=self.device)
        v = torch.empty(n, dtype=torch.float32, device=self.device)
        if n:
            for t, ptr in ((hb, st.hbits), (vb, st.vbits), (meta, st.meta), (mask, mk.value), (p, pp.value), (v, vv.value)):
                _dev_copy(t, ptr, t.numel() * t.element_size())
        torch.cuda.synchronize(self.device)
        return (_cabi.soa_to_packed(hb.cpu().numpy(), vb.cpu().numpy(), meta.cpu().numpy()), mask.cpu().numpy().view(np.uint32),
                p.cpu().numpy(), v.cpu().numpy())

    def set_miss_outputs(self, p: torch.Tensor, v: torch.Tensor):
        """Evaluations of the miss list's first len(v) leaves from ANOTHER evaluator (qz_selfplay_misses: "for callers
        that evaluate the list themselves"): p float32 [n,140] (the policy at every action id; only legal entries are
        read), v float32 [n], written where qz_selfplay_evaluate would put them.  Between qz_selfplay_leaf_rules and
        qz_selfplay_round_tail, on the current stream."""
        n = int(v.numel())
        assert p.dtype == torch.float32 and v.dtype == torch.float32 and p.shape == (n, 140) and n <= self.n_boards
        if n == 0:
            return
        pp, vv = C.c_void_p(), C.c_void_p()
        _cabi.check(self.L.qz_selfplay_misses(self.h, None, None, None, C.byref(pp), C.byref(vv)))
        self._miss_src = (p.to(self.device).contiguous(), v.to(self.device).contiguous())  # (alive until the copies have run)
        global _hip
        if _hip is None:
            _hip = C.CDLL("libamdhip64.so")
        for src, dst in zip(self._miss_src, (pp.value, vv.value)):
            rc = _hip.hipMemcpyAsync(C.c_void_p(dst), C.c_void_p(src.data_ptr()), C.c_size_t(src.numel() * 4), 3, C.c_void_p(self._s()))
            if rc != 0:
                raise _cabi.QzError(_cabi.E_HIP, "hipMemcpyAsync failed: %d" % rc)

    # ------------------------------------------------------------------ end of a ply
    def finish_move(self, forced=None):
        """-> (moves uint8 [B] (255 = board idle), pi float32 [B,140])."""
        self._descended = False  # (a descent run ahead of this move would describe the old roots)
        fp = 0
        if forced is not None:
            self._forced = forced.to(device=self.device, dtype=torch.uint8).contiguous()
            fp = self._forced.data_ptr()
        bad0 = self.stats()["bad_forced_moves"] if forced is not None else 0
        _cabi.check(self.L.qz_mcts_finish_move(self.h, fp, self.pi.data_ptr(), self.moves.data_ptr(), self._s()))
        if forced is not None:  # forced moves come from host-side drivers (tests, single-game API): a sync is fine
            bad = self.stats()["bad_forced_moves"] - bad0
            if bad:
                raise _cabi.QzError(_cabi.E_INVALID, "%d forced move(s) are not children of their root: those boards did not move" % bad)
        return self.moves, self.pi

    def update_with_move(self, moves: torch.Tensor):
        self._descended = False
        self._upd = moves.to(device=self.device, dtype=torch.uint8).contiguous()
        _cabi.check(self.L.qz_mcts_update_with_move(self.h, self._upd.data_ptr(), self._s()))

    def root_pi(self):
        pi = torch.empty((self.n_boards, 140), dtype=torch.float64, device=self.device)
        visits = torch.empty((self.n_boards, 140), dtype=torch.int32, device=self.device)
        _cabi.check(self.L.qz_mcts_root_pi(self.h, pi.data_ptr(), visits.data_ptr(), self._s()))
        return pi, visits

    def root_children(self):
        B, dev = self.n_boards, self.device
        visits = torch.empty((B, 140), dtype=torch.int32, device=dev)
        q = torch.empty((B, 140), dtype=torch.float64, device=dev)
        prior = torch.empty((B, 140), dtype=torch.float32, device=dev)
        root_n = torch.empty(B, dtype=torch.int32, device=dev)
        _cabi.check(self.L.qz_mcts_root_children(self.h, visits.data_ptr(), q.data_ptr(), prior.data_ptr(),
                                                 root_n.data_ptr(), self._s()))
        return visits, q, prior, root_n

    def pending(self):
        """(finished games, their plies) waiting for harvest.  Synchronises the stream."""
        c = (C.c_int64 * 2)()
        _cabi.check(self.L.qz_harvest_counts(self.h, C.byref(c), self._s()))
        return int(c[0]), int(c[1])

    def harvest(self):
        """Collect every finished game as tuples and restart those boards.  -> TupleBatch | None"""
        games, plies = self.pending()
        if games == 0:
            return None
        tb = DeviceBoards(plies, self.device)
        pi = torch.empty((plies, 140), dtype=torch.float32, device=self.device)
        z = torch.empty(plies, dtype=torch.float32, device=self.device)
        gid = torch.empty(plies, dtype=torch.int32, device=self.device)
        slot = torch.empty(games, dtype=torch.int32, device=self.device)
        _cabi.check(self.L.qz_harvest(self.h, tb.byref(), pi.data_ptr(), z.data_ptr(), gid.data_ptr(), slot.data_ptr(), plies, self._s()))
        return TupleBatch(tb, pi, z, gid, games, slot)

    def stats(self) -> dict:
        st = _cabi.qz_stats()
        _cabi.check(self.L.qz_engine_stats(self.h, C.byref(st), self._s()))
        return {k: int(getattr(st, k)) for k, _ in st._fields_}

    # ------------------------------------------------------------------ whole plies / games
    def play_ply(self, evaluator, n_playout=None, forced=None):
        self.run_playouts(evaluator, n_playout)
        return self.finish_move(forced)

    def collect_games(self, evaluator, n_games, max_plies=None):
        """Play until at least `n_games` complete games were harvested. -> list[TupleBatch]"""
        out, got, plies = [], 0, 0
        while got < n_games:
            self.play_ply(evaluator)
            plies += 1
            tb = self.harvest()
            if tb is not None:
                out.append(tb)
                got += tb.n_games
            if max_plies is not None and plies >= max_plies:
                break
        return out


class BoardGroups:
    """The boards of one GPU split into `n_groups` independent ``SelfPlayEngine``s, each on its
    own HIP stream with its own leaf evaluator.  A playout step is select -> rules -> net ->
    expand/backup, strictly dependent inside one group; with two groups the tree kernels of one
    (their duration is the deepest descent / longest path search of the group: latency, not
    bandwidth) run underneath the other's convolutions.  Measured on MI355X, 400 playouts,
    per-leaf BN fp32: 1x4096 boards 1.32 M playouts/s, 2x2048 1.39 M, 2x4096 1.48 M
    (benchmarks/two_group_overlap.py).

    Groups never exchange anything; every board keeps its own Philox stream
    (seed -> group seed -> board), so results do not depend on how the streams interleave."""

    def __init__(self, n_boards, n_groups, make_evaluator, seed=0, device="cuda:0", **engine_kw):
        n_boards, n_groups = int(n_boards), int(n_groups)
        if n_groups < 1 or n_boards % n_groups:
            raise ValueError("n_boards=%d is not a multiple of n_groups=%d" % (n_boards, n_groups))
        self.device = torch.device(device)
        self.n_boards, self.n_groups = n_boards, n_groups
        self.engines = [SelfPlayEngine(n_boards // n_groups, seed=self.group_seed(seed, g), device=self.device, **engine_kw)
                        for g in range(n_groups)]
        self.evaluators = [make_evaluator() for _ in range(n_groups)]
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(n_groups)]
        main = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(main)
        self.n_playout = self.engines[0].n_playout

    @staticmethod
    def group_seed(seed, g):
        # additive, with a different constant than dist.shard_seed's xor, so (rank, group) pairs
        # never collide; group 0 keeps the seed (one group == a plain SelfPlayEngine)
        return (int(seed) + 0xD1B54A32D192ED03 * g) & 0xFFFFFFFFFFFFFFFF

    def _each(self):
        for g in range(self.n_groups):
            with torch.cuda.stream(self.streams[g]):
                yield g, self.engines[g], self.evaluators[g]

    def playout_step(self, events=None, write_planes=None, tree_events=None, nn_events=None, more=False):
        """One playout of every board; `events` = one (start, end) pair per group around the
        group's rules-op launch, `tree_events` = one pair of pairs per group (select, expand/backup),
        `nn_events` = one pair per group around the evaluator."""
        for g, eng, ev in self._each():
            eng.playout_step(ev, events=None if events is None else events[g], write_planes=write_planes,
                             tree_events=None if tree_events is None else tree_events[g],
                             nn_events=None if nn_events is None else nn_events[g], more=more)

    def run_playouts(self, n=None):
        n = self.n_playout if n is None else int(n)
        if self.n_groups == 1:  # may replay a captured HIP graph
            for _, eng, ev in self._each():
                eng.run_playouts(ev, n)
            return
        for i in range(n):
            self.playout_step(more=i + 1 < n)

    # Results handed to the caller were produced on a group stream: the caller's stream is made to
    # wait for it, and the tensors are marked as used there, before anything is returned.
    def _publish(self, g, tensors):
        main = torch.cuda.current_stream(self.device)
        main.wait_stream(self.streams[g])
        for t in tensors:
            if t is not None and t.is_cuda:
                t.record_stream(main)

    def finish_move(self):
        out = []
        for g, eng, _ in self._each():
            out.append(eng.finish_move())
        for g, (moves, pi) in enumerate(out):
            self._publish(g, (moves, pi))
        return out

    def harvest(self):
        """-> list[TupleBatch] (one per group that had finished games)"""
        out = []
        for g, eng, _ in self._each():
            tb = eng.harvest()
            if tb is not None:
                out.append((g, tb))
        for g, tb in out:
            self._publish(g, (tb.boards.hbits, tb.boards.vbits, tb.boards.meta, tb.pi, tb.z, tb.game))
        return [tb for _, tb in out]

    def play_ply(self, n_playout=None):
        self.run_playouts(n_playout)
        return self.finish_move()

    # counters are summed over the groups; these fields are maxima / high-water marks of ONE engine
    _MAX_FIELDS = ("max_depth", "max_nodes", "max_edges", "rounds")

    def stats(self) -> dict:
        tot = {}
        for _, eng, _ in self._each():
            for k, v in eng.stats().items():
                tot[k] = max(tot.get(k, 0), v) if k in self._MAX_FIELDS else tot.get(k, 0) + v
        return tot

    # ------------------------------------------------------------------ asynchronous self-play
    def set_playouts(self, n_playout):
        for eng in self.engines:
            eng.set_playouts(n_playout)
        self.n_playout = int(n_playout)

    def capture_rounds(self, rounds=16, **kw):
        for _, eng, ev in self._each():
            eng.capture_rounds(ev, rounds=rounds, **kw)

    def run_rounds(self, n, chunk=16, **kw):
        """n rounds of every group; the groups' launches are issued `chunk` rounds at a time in turn, each on its own
        stream, so that one group's tree kernel runs under the other's network launches and launch tails."""
        n, chunk = int(n), int(chunk)
        for k in range(0, n, chunk):
            for _, eng, ev in self._each():
                eng.run_rounds(ev, min(chunk, n - k), **kw)

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def join_main(self):
        """Order the group streams after the caller's stream (e.g. after a training step changed
        the weights the evaluators read)."""
        main = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(main)

    def close(self):
        self.synchronize()
        for eng in self.engines:
            eng.close()
