/*
 * qz_abi.h -- C ABI of libqzero_hip.so, the MI355X (gfx950) self-play engine.
 *
 * Drop-in boundary for the self-play data-generation path of cryer/AlphaZero_Quoridor.
 * The reference has no FFI of its own (it is pure Python); the boundary it offers is its
 * module surface.  Each entry point below names the reference interface it replaces
 * (file:line into the reference repo) -- the Python mirror in
 * alphazero_quoridor_amd/{quoridor,mcts,policy_value_net,train}.py binds these through
 * ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - plain C types only; every pointer marked [dev] is device memory on the engine's GPU
 *     (e.g. torch.Tensor.data_ptr()), [host] is host memory;
 *   - return 0 on success, a negative QZ_E_* code on failure (qz_last_error() has text);
 *     nothing throws across the ABI;
 *   - the caller owns every buffer it passes; the library neither frees nor keeps them
 *     past the call's stream order; tree arenas / trajectories are engine-owned and freed
 *     by qz_engine_destroy();
 *   - every launch goes on the hipStream_t passed as `void* stream` (NULL = default
 *     stream) with no implicit synchronisation unless the entry point says "sync";
 *   - there is NO CPU fallback: without a HIP device every compute entry point fails with
 *     QZ_E_NO_DEVICE.
 */
#ifndef QZ_ABI_H
#define QZ_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QZ_ABI_VERSION 7
#define QZ_N_ACTIONS 140            /* quoridor.py:12  action_space = 140            */
#define QZ_PLANES (26 * 81)         /* quoridor.py:58-131  26x9x9 state tensor       */
#define QZ_MASK_WORDS 5             /* 140-bit legal mask, bit a of word a/32        */
#define QZ_NO_MOVE 255              /* "no forced move" / mcts.py:169 update_with_move(-1) */

#define QZ_E_INVALID (-1)
#define QZ_E_NO_DEVICE (-2)
#define QZ_E_HIP (-3)
#define QZ_E_OOM (-4)
#define QZ_E_STATE (-5)

/*
 * Boards in HBM: structure-of-arrays, three u64 per board (24 B):
 *   hbits[b]  bit ix <=> _intersections[ix] == +1 (horizontal)      quoridor.py:49-53
 *   vbits[b]  bit ix <=> _intersections[ix] == -1 (vertical)
 *   meta[b]   byte0 = _positions[1] (int8), byte1 = _positions[2] (int8; off-board wins
 *             leave 0..80: quoridor.py:226-229), byte2/3 = _player{1,2}_walls_remaining,
 *             byte4 = current_player (1|2), bytes 5..7 = 0          quoridor.py:27-56
 */
typedef struct {
    uint64_t* hbits; /* [dev] [n] */
    uint64_t* vbits; /* [dev] [n] */
    uint64_t* meta;  /* [dev] [n] */
} qz_boards;

/* ------------------------------------------------------------------ library */
int qz_version(void);
const char* qz_last_error(void);  /* thread-local text of the last failure */
int qz_device_count(void);        /* number of HIP devices visible (0 => nothing can run) */

/* ------------------------------------------------- stateless rules kernels */
/* Quoridor.actions() (quoridor.py:138-157, 420-528) for n live boards:
 * mask5[n][5] <- 140-bit legal set.  The reference's ordered list is the mask read in
 * the order pawn codes 0..11 ascending, then for ix in 0..63: 12+ix, 76+ix. */
int qz_movegen(const qz_boards* boards, int n, uint32_t* mask5 /*[dev]*/, void* stream);
/* Quoridor.state() (quoridor.py:58-131): planes[n][26][9][9] float32 */
int qz_encode(const qz_boards* boards, int n, float* planes /*[dev]*/, void* stream);
/* both in one pass over the boards (the engine's leaf kernel) */
int qz_movegen_encode(const qz_boards* boards, int n, uint32_t* mask5 /*[dev]*/,
                      float* planes /*[dev]*/, void* stream);
/* Which formulation of the rules op a call / an engine uses (tuning and A/B runs; all zero =
 * the defaults).  Per call and per engine: the library keeps no mutable global state.
 *   variant        0 = pick by batch size (k_wave_rules below 8,192 boards, pooled pipeline from
 *                  there on), 2 | 3 | 4 = k_wave_rules with 2 | 1 | 4 boards per wavefront (3 = what 0
 *                  picks for small batches: base paths searched on nine lanes per player, planes as
 *                  streaming stores), 5 = 3 with one base-path search per lane, 6 = 3 with ordinary
 *                  stores (parity / A-B partners), 8 | 12 | 16 | 24 | 32 = pooled pipeline with that
 *                  many boards per mask workgroup
 *   detour_pooled  group-detour mode of the pooled pipeline: 0 = default (one group), else 1 + mode
 *   detour_wave    ... of k_wave_rules: 0 = default (off), else 1 + mode (mode 0 | 1 | 2)
 *   enc_split_pct  0 = default (50): percent of the encoder groups beside the path groups (first launch of the pooled pipeline) */
typedef struct {
    int32_t variant, detour_pooled, detour_wave, enc_split_pct;
} qz_rules_opts;
/* qz_movegen / qz_encode / qz_movegen_encode with explicit options: mask5 or planes may be NULL
 * (not both), opts may be NULL (defaults). */
int qz_movegen_encode_opts(const qz_boards* boards, int n, uint32_t* mask5 /*[dev] or NULL*/, float* planes /*[dev] or NULL*/,
                           const qz_rules_opts* opts, void* stream);
/* Quoridor.step() + has_a_winner() (quoridor.py:159-186, 193-202, 217-269), in place.
 * done[n] <- 1 if the move ended the game; winner[n] <- 0 | 1 | 2 */
int qz_step(qz_boards* boards, const uint8_t* action /*[dev]*/, int n, uint8_t* done /*[dev]*/,
            uint8_t* winner /*[dev]*/, void* stream);

/* MCTS._evaluate_rollout (pure_mcts.py:81-103) for n boards at once, IN PLACE: uniformly random
 * legal moves (rollout_policy_fn, pure_mcts.py:7-10) until somebody has won or `limit` (1000 in
 * the reference) iterations have passed.  value[n] int8 <- +1 if the winner is the side that
 * was to move when the rollout began, -1 if the other, 0 if nobody won within the limit.  Each
 * iteration is one move-generation launch (actions() of every live board) + one pick-and-step
 * launch; the loop ends early once every board is done (checked every 16 iterations).  Random
 * stream: Philox keyed by `seed`, counter (board, iteration).  scratch: caller-provided device
 * memory of qz_rollout_scratch_bytes(n) bytes.  SYNC. */
int64_t qz_rollout_scratch_bytes(int n);
int qz_rollout(qz_boards* boards, int n, int limit, uint64_t seed, int8_t* value /*[dev]*/, void* scratch /*[dev]*/, void* stream);

/* ---------------------------------------------------------------- engine */
typedef struct qz_engine qz_engine;

typedef struct {
    int32_t n_boards;          /* concurrent games, one search tree each               */
    int32_t n_playout;         /* mcts.py:89   playouts per move (train.py:20 = 400)   */
    float c_puct;              /* mcts.py:89   (train.py:21 = 5)                       */
    float temp;                /* mcts.py:129  (train.py:19 = 1.0)                     */
    float dirichlet_alpha;     /* mcts.py:181  0.3                                     */
    float noise_frac;          /* mcts.py:181  0.25                                    */
    uint64_t seed;             /* Philox key; the reference uses global numpy RNG      */
    int32_t device;            /* HIP device ordinal                                   */
    int32_t is_selfplay;       /* mcts.py:159  1: noise + subtree reuse; 0: reset tree */
    int32_t fix_terminal_sign; /* 0 = reproduce mcts.py:125 (winning edge backed up -1) */
    /* Memory.  Trees and trajectories live in two pools of 64-KB pages shared by all boards
     * (a tree page = 2,048 edge records of 32 B; a trajectory page holds ~100-1,000 plies), so
     * a board only occupies what its tree / game really needs: 32,768 boards x n_playout=400
     * fit one 288-GB MI355X.  A pool that runs dry never corrupts anything: the expansion is
     * skipped (node_overflow) / the game is dropped (aborted_pool), and both are counted. */
    int32_t node_cap;          /* most expanded nodes per tree (0 = no limit)                    */
    int32_t edge_cap;          /* most edges per tree (0 = the page table's reach: 262,144)      */
    int32_t max_plies;         /* drop a game after this many plies (0 = no limit; the reference has none) */
    int32_t tree_pool_pages;   /* 0 = auto: n_boards x max(4, ceil(250 x n_playout / 2,048)), >= 264 */
    int32_t traj_pool_pages;   /* 0 = auto: 64 per board (4 MB) up to 8,192 boards, 16 beyond    */
    int32_t traj_page_dwords;  /* 0 = 16,384 (64 KB); >= 256.  Small pages only make sense in tests */
    qz_rules_opts rules;       /* formulation of the leaf rules op (all zero = defaults)        */
    int32_t select_opts;       /* A/B switches of the descent kernel (0 = defaults): bit 0 = walk every level (no replay of
                                  recorded descents; same results, the test partner of the records), bit 1 = no readlane
                                  scan for nodes with <= 8 children, bit 2 = the build of the asynchronous loop's kernel
                                  for large engines (seven wavefronts per SIMD: what engines above 4,096 boards run) whatever the engine's size,
                                  bit 3 = qz_selfplay_advance's budget (if >= 100 us) counts from the launch's FIRST wavefront -- one
                                  deadline for all boards -- and the boards take the first slots in turn: for engines of more boards
                                  than the chip holds wavefronts of it (7,168), where a board may get its slot in the middle of a launch,
                                  bit 4 = reserved (round 6's one-lane-per-board prototype, k_lanes: parity-green, 4x slower than
                                  k_advance, removed -- profiles/round6/SUMMARY.md 1; ignored),
                                  bit 5 = the boards on which NEITHER player has a wall left are played by k_rows (csrc/qz_rows.h: SIXTEEN
                                  LANES per board, four boards per wavefront, boards from a queue) beside k_advance's launch for the
                                  others -- same search results bit for bit
                                  (tests/test_gpu_lanes.py); measured on a par with k_advance inside the launch at 1.5 x the boards and
                                  6 % behind per round (profiles/round6/SUMMARY.md): off by default */
    /* Leaf-evaluation memo (qz_selfplay_*): log2 of the number of buckets of its two tables; 0 = auto (16,384 small entries
     * -- at most 8 GB -- and 512 big entries per board, rounded up to a power of two), < 0 = no memo (every leaf goes to the network).
     * small: leaves whose mover has no wall left, 4 entries of 128 B per bucket; big: all others, 2 x 640 B. */
    int32_t memo_small_log2, memo_big_log2;
    /* qz_selfplay_*: update_with_move (mcts.py:146-151) keeps the chosen child's subtree WHERE IT IS while the tree's
     * allocation cursor is below this many edges (the rest of the old tree stays behind as garbage), and copies it into
     * fresh pages -- returning the old ones to the pool -- once the cursor has passed it.  0 = a third of a board's share
     * of the tree pool, at most 49,152 (24 pages of 64 KB); < 0 = every move copies, like qz_mcts_finish_move.  With node_cap / edge_cap set every move copies
     * (the caps count live nodes).  Same search either way. */
    int32_t compact_edges;
    /* Drop a game (qz_stats.aborted_depth) as soon as one of its playouts descends more than this many levels; 0 = never.
     * The reference backs a playout up by recursion (TreeNode.update_recursive, mcts.py:55-62: one Python frame per node
     * of the path) and never raises the interpreter's recursion limit of 1,000: under `python train.py` a path longer
     * than 992 levels ends the whole run with a RecursionError.  992 mirrors that envelope; reference-faithful games can
     * grow forced lines of tens of thousands of levels (the inverted terminal value makes searches avoid winning). */
    int32_t max_depth;
} qz_config;

typedef struct {
    int64_t games_finished;    /* complete games harvested so far                      */
    int64_t plies_played;      /* real moves played (finish_move calls x live boards)  */
    int64_t playouts;          /* leaf selections                                      */
    int64_t leaf_terminal;     /* playouts that ended on a terminal leaf               */
    int64_t node_overflow;     /* expansions skipped / subtrees truncated: arena full  */
    int64_t games_aborted;     /* dropped games: sum of the aborted_* causes below */
    int64_t pending_games;     /* finished, not yet harvested                          */
    int64_t pending_plies;
    int64_t arena_bytes;       /* device bytes owned by the engine                     */
    int64_t descent_levels;    /* tree levels walked by all playouts (mean depth = / playouts) */
    int64_t max_nodes;         /* largest tree right now, in nodes / in edges (arena occupancy   */
    int64_t max_edges;         /* against qz_config.node_cap / edge_cap)                         */
    int64_t aborted_no_move;   /* no legal move at the root (the reference crashes there, mcts.py:195) */
    int64_t aborted_max_plies; /* qz_config.max_plies reached, or a game outgrew its trajectory page table */
    int64_t aborted_pool;      /* trajectory pool empty                                           */
    int64_t bad_forced_moves;  /* forced moves that were not children of the root: those boards did NOT move */
    int64_t nonfinite_values;  /* descents that met a NaN PUCT value (diverged network); first child taken like Python's max() */
    int64_t tree_pages_total, tree_pages_in_use, tree_pages_peak;
    int64_t traj_pages_total, traj_pages_in_use, traj_pages_peak;
    int64_t edges_scanned;     /* 32-byte edge records read by the descents (k_select's algorithmic bytes / 32) */
    int64_t edges_expanded;    /* edge records created by expansions                                    */
    int64_t max_depth;         /* deepest descent so far (tree levels); descents beyond 2,048 levels back up by walking parent links */
    /* the descents of >= 256 levels (the ones that set the select kernel's duration): how many, how many of them
     * found a descent record shorter than half their length, their levels, and the levels the replay confirmed */
    int64_t deep_descents, deep_descents_cold, deep_levels, deep_levels_replayed;
    /* asynchronous self-play (qz_selfplay_*) */
    int64_t rounds;            /* qz_selfplay_advance launches                                            */
    int64_t memo_hits;         /* leaves expanded from the memo (no network evaluation)                   */
    int64_t nn_evals;          /* leaves sent to the network (memo misses)                                */
    int64_t memo_inserts;      /* evaluations stored in the memo                                          */
    int64_t memo_locked;       /* ... not stored because another insert held the bucket                   */
    int64_t open_rounds;       /* board-launches / plies whose root's mover still had walls (the phase of a game in which  */
    int64_t open_plies;        /* almost every leaf is new; the rest of a game revisits a few thousand boards)             */
    int64_t waiting_boards;    /* boards waiting for the network right now                                */
    int64_t aborted_depth;     /* games dropped because a descent exceeded qz_config.max_depth (the reference's RecursionError) */
    int64_t runaway_descents;  /* descents cut off because they were deeper than a tree has edges (a cycle = corrupted tree
                                  storage): must be 0; the guard exists so that such a bug cannot hang the GPU   */
    int64_t compact_slices;    /* subtree copies of the asynchronous loop that stopped at their launch's budget and went on
                                  in the board's next launch (a 1,000-level line copies one level per memory round trip)  */
    int64_t miss_overflow;     /* leaves that found the miss list full (only a stale miss counter can do that): must be 0; the
                                  guard exists so that such a bug cannot write past the list                    */
} qz_stats;

/* MCTSPlayer.__init__ / MCTS.__init__ (mcts.py:89-100, 159-161) for n_boards trees +
 * Quoridor.reset() (quoridor.py:26-56) for n_boards games. */
int qz_engine_create(const qz_config* cfg, qz_engine** out);
int qz_engine_destroy(qz_engine* e);
/* Quoridor.reset() on every board + fresh roots (MCTSPlayer.reset_player, mcts.py:168-169) */
int qz_engine_reset(qz_engine* e, void* stream);
/* load / read the root boards (the `game` argument of choose_action, mcts.py:172).
 * reset_trees != 0 also drops every search tree. */
int qz_engine_set_boards(qz_engine* e, const qz_boards* src, int reset_trees, void* stream);
int qz_engine_get_boards(qz_engine* e, const qz_boards* dst, void* stream);
/* plies[n_boards] int32 <- moves played so far in every board's current game (inspection: length statistics of games
 * that are still running) */
int qz_engine_get_plies(qz_engine* e, int32_t* plies /*[dev]*/, void* stream);
/* The games the engine dropped last (qz_stats.games_aborted counts them, by cause): the root position at which the game
 * could not go on, why, after how many plies, on which board slot.  The reference cannot play these games on either:
 * QZ_DROP_NO_MOVE = Quoridor.actions() of the root is empty -- MCTSPlayer.choose_action prints "WARNING: the board is full"
 * and returns None (mcts.py:195-196), start_self_play's unpack of it raises TypeError (quoridor.py:587);
 * QZ_DROP_DEPTH = a playout descended more than qz_config.max_depth levels (TreeNode.update_recursive's RecursionError,
 * mcts.py:55-62).  out[0..return value) <- the newest min(cap, 4096, *total) drops, oldest first; *total = drops so far.
 * Synchronises the stream (inspection / logging, not the hot path). */
enum { QZ_DROP_NO_MOVE = 1, QZ_DROP_DEPTH = 2, QZ_DROP_MAX_PLIES = 3, QZ_DROP_POOL = 4 };
typedef struct {
    uint64_t hbits, vbits, meta; /* the root board, one record of qz_boards */
    int32_t cause;               /* QZ_DROP_* */
    int32_t ply;                 /* moves played in the game before it was dropped */
    int32_t board;               /* engine board slot */
    int32_t reserved;
} qz_dropped_game;
int qz_engine_dropped_games(qz_engine* e, qz_dropped_game* out /*[host] cap*/, int cap, int64_t* total /*[host] or NULL*/, void* stream);
/* the `temp` argument of get_move_probs / choose_action (mcts.py:129,172) for later calls */
int qz_engine_set_temp(qz_engine* e, float temp);
/* the n_playout argument of MCTSPlayer (mcts.py:159) for later moves of the asynchronous loop (qz_selfplay_*): a board
 * plays its move once it has done this many playouts on its root.  >= 1.  (The lock-step entry points take their
 * playout count from the number of calls.) */
int qz_engine_set_playouts(qz_engine* e, int n_playout);
/* change the formulation of this engine's leaf rules op (qz_config.rules) for later calls */
int qz_engine_set_rules_opts(qz_engine* e, const qz_rules_opts* opts);

/* MCTS._playout, first half (mcts.py:107-117): for every board descend from the root by
 * PUCT (TreeNode.select / get_value, mcts.py:37-42, 64-70) applying Quoridor.step() to a
 * scratch copy, then run actions() + state() on the leaf.
 *   leaf_planes[n][26][9][9] <- network input for the leaf (zeros for terminal leaves)
 *   leaf_mask5[n][5]         <- legal set of the leaf (may be NULL)
 *   leaf_terminal[n]         <- 1 if the leaf is a finished game (may be NULL) */
int qz_mcts_select(qz_engine* e, float* leaf_planes /*[dev]*/, uint32_t* leaf_mask5 /*[dev]*/,
                   uint8_t* leaf_terminal /*[dev]*/, void* stream);
/* the two halves of qz_mcts_select as separate launches (bench.py brackets the second one
 * with HIP events): descend = the PUCT walk only; leaf_inputs = actions() + state() of the
 * leaves found by the last descend (the fused move-generation + encoder kernel). */
int qz_mcts_descend(qz_engine* e, void* stream);
int qz_mcts_leaf_inputs(qz_engine* e, float* leaf_planes /*[dev]*/, uint32_t* leaf_mask5 /*[dev]*/,
                        uint8_t* leaf_terminal /*[dev]*/, void* stream);
/* same, but hands back the leaf boards instead of the planes (for host-side policy
 * callbacks: the `game` passed to policy_value_function, mcts.py:117) */
int qz_mcts_select_boards(qz_engine* e, const qz_boards* leaf_out, uint32_t* leaf_mask5 /*[dev]*/,
                          uint8_t* leaf_terminal /*[dev]*/, void* stream);
/* MCTS._playout, second half (mcts.py:119-127): TreeNode.expand (mcts.py:27-35) with
 * priors p[n][140] (= exp(log_softmax), gathered at the legal moves, NOT renormalised:
 * policy_value_net.py:155-162) and update_recursive(-v) (mcts.py:44-62).  Terminal leaves
 * ignore p/v and use +-1 (mcts.py:125). */
int qz_mcts_expand_backup(qz_engine* e, const float* p /*[dev]*/, const float* v /*[dev]*/,
                          void* stream);
/* qz_mcts_expand_backup of this playout + qz_mcts_descend of the NEXT one in one launch (the loop of
 * MCTS.get_move_probs, mcts.py:135-139, runs them back to back on the same tree anyway): the edge
 * records the backup has just touched are still in the cache the descent reads through.  Follow with
 * qz_mcts_leaf_inputs, exactly as after qz_mcts_descend.  Same results as the two separate calls. */
int qz_mcts_expand_backup_descend(qz_engine* e, const float* p /*[dev]*/, const float* v /*[dev]*/, void* stream);
/* MCTS.get_move_probs tail (mcts.py:141-144): pi[n][140] float64 <- softmax(log(N+1e-10)/temp)
 * scattered to action ids (mcts.py:174-177); visits[n][140] int32 (may be NULL) */
int qz_mcts_root_pi(qz_engine* e, double* pi /*[dev]*/, int32_t* visits /*[dev]*/, void* stream);
/* test/inspection: per-action root-child statistics, -1 visits for non-children */
int qz_mcts_root_children(qz_engine* e, int32_t* visits /*[dev][n][140]*/,
                          double* q /*[dev][n][140]*/, float* prior /*[dev][n][140]*/,
                          int32_t* root_visits /*[dev][n]*/, void* stream);
/* MCTS.update_with_move (mcts.py:146-151) only: re-root (keep the subtree) per board;
 * QZ_NO_MOVE = fresh root.  Boards are not stepped. */
int qz_mcts_update_with_move(qz_engine* e, const uint8_t* moves /*[dev]*/, void* stream);
/* MCTSPlayer.choose_action tail + one iteration of Quoridor.start_self_play's loop
 * (mcts.py:174-187, quoridor.py:585-602): pi from root visits; move ~ 0.75*pi +
 * 0.25*Dirichlet(alpha) (or forced_move[b] != QZ_NO_MOVE); record (board, pi) in the
 * board's trajectory; update_with_move; Quoridor.step(move); finished games are flagged
 * for qz_harvest.  pi_out[n][140] float32 / move_out[n] may be NULL.  A forced move that is not
 * a child of the root leaves that board untouched (move_out = QZ_NO_MOVE) and counts in
 * qz_stats.bad_forced_moves -- the reference would raise KeyError in update_with_move's lookup. */
int qz_mcts_finish_move(qz_engine* e, const uint8_t* forced_move /*[dev]*/, float* pi_out /*[dev]*/,
                        uint8_t* move_out /*[dev]*/, void* stream);

/* finished games waiting for harvest: counts[0] = games, counts[1] = plies.  SYNC. */
int qz_harvest_counts(qz_engine* e, int64_t counts[2] /*[host]*/, void* stream);
/* Quoridor.start_self_play tail (quoridor.py:596-610) for every finished game:
 * one tuple per recorded ply, games in board order, plies in play order:
 *   t_boards  <- board before the move (re-encode with qz_encode to get `state`)
 *   t_pi      [cap][140] float32
 *   t_z       [cap] float32: +1 if the recorded mover won else -1   (quoridor.py:599-602)
 *   t_game    [cap] int32: running game id local to this call (may be NULL)
 *   g_board   [counts[0]] int32: the board (slot of the engine) every harvested game was played on (may be NULL)
 * then the boards are reset (quoridor.py:578) with fresh trees (mcts.py:168-169).
 * `cap` must be >= counts[1] from qz_harvest_counts. */
int qz_harvest(qz_engine* e, const qz_boards* t_boards, float* t_pi /*[dev]*/, float* t_z /*[dev]*/,
               int32_t* t_game /*[dev]*/, int32_t* g_board /*[dev]*/, int64_t cap, void* stream);

int qz_engine_stats(qz_engine* e, qz_stats* out /*[host]*/, void* stream); /* SYNC */

/* ------------------------------------------------------------- leaf-evaluator glue
 * The network stays in PyTorch-ROCm; this is the one normalisation the reference's leaf
 * evaluation needs that stock PyTorch does badly: BatchNorm2d in TRAINING mode on a batch of
 * ONE (policy_value_net.py:154 -- the module is never put in eval mode), i.e. every
 * (sample, channel) 9x9 plane uses its own mean / biased variance:
 *   out = act(gamma[c]*(x - mean_plane)/sqrt(var_plane + eps) + beta[c] [+ residual])
 * x/out/residual: float32 NCHW [n_planes/channels, channels, 9, 9] contiguous; out may alias x. */
int qz_nn_instnorm_act(const float* x /*[dev]*/, const float* gamma /*[dev][channels]*/,
                       const float* beta /*[dev][channels]*/, const float* residual /*[dev] or NULL*/,
                       float* out /*[dev]*/, int64_t n_planes, int channels, int relu, float eps, void* stream);

/* the same on channels-last memory: x/out/residual are the NHWC storage [n_samples][81][channels]
 * of a logical [n_samples, channels, 9, 9] tensor; channels <= 64 */
int qz_nn_instnorm_act_nhwc(const float* x /*[dev]*/, const float* gamma /*[dev]*/, const float* beta /*[dev]*/,
                            const float* residual /*[dev] or NULL*/, float* out /*[dev]*/, int64_t n_samples,
                            int channels, int relu, float eps, void* stream);

/* First layer of the policy-value net straight from the boards: out = relu(norm(conv1(state(board))))
 * without materialising state() (replaces encode + F.conv2d(conv1) + bn1 + relu of
 * policy_value_net.py:53-60,73 on the reference's leaf evaluation; same values up to fp32 summation
 * order).  Tables, built from conv1's weight W[64][26][3][3] by the caller (policy_value_net.py,
 * LeafEvaluator.refresh): hot9 [21][9][64] = 3x3 convolution of an all-ones plane 5+i at the nine
 * border classes (row class * 3 + column class, class = 0 first / 1 inner / 2 last); base0 [81][64] =
 * convolution of plane 0 with no wall on the board; wd [4][9][64] = per-tap weights of one extra
 * pixel: H wall (W2 - W0), V wall (W1 - W0), mover pawn (W3), other pawn (W4), tap = 3*ky + kx.
 * gamma != NULL: per-leaf normalisation over the 81 positions with gamma / beta (BatchNorm in
 * training mode on a batch of one); gamma == NULL: beta is a per-channel bias.  terminal[i] != 0
 * marks a leaf whose planes are all zero (may be NULL).  out: NHWC storage [n][81][64]. */
int qz_nn_input_layer(const qz_boards* boards /*[dev]*/, const uint8_t* terminal /*[dev] or NULL*/, int64_t n,
                      const float* hot9 /*[dev]*/, const float* base0 /*[dev]*/, const float* wd /*[dev]*/,
                      const float* gamma /*[dev] or NULL*/, const float* beta /*[dev]*/, float* out /*[dev]*/, float eps,
                      void* stream);
/* Both heads of the policy-value net in one pass (policy_value_net.py:64-70,88-93,155): trunk
 * output t (NHWC [n][81][64]) -> p [n][140] = exp(log_softmax(fc3(.))), v [n] = tanh(fc2(fc1(.))).
 * w6k [9][64][6]: the 3x3 weights of conv2 (channels 0..3) and conv3 (4..5), tap-major; gamma6 / beta6
 * [6]: bn2 + bn3 per-leaf normalisation (gamma6 == NULL: beta6 is a bias, folded BatchNorm);
 * w1t [324][128], w3t [162][140]: fc1 / fc3 weights transposed; b1 [128], w2 [128], b2 [1], b3 [140]. */
int qz_nn_head(const float* t /*[dev]*/, int64_t n, const float* w6k, const float* gamma6 /*or NULL*/, const float* beta6,
               const float* w1t, const float* b1, const float* w2, const float* b2, const float* w3t, const float* b3,
               float* p_out /*[dev] n*140*/, float* v_out /*[dev] n*/, float eps, void* stream);
/* One trunk layer of the policy-value net on the matrix cores (policy_value_net.py:20-48: conv3x3
 * 64 -> 64, pad 1, no bias; BatchNorm in training mode on a batch of one = statistics over the 81
 * positions of every (leaf, channel); + residual; ReLU), channels-last fp32 in and out:
 *   out = act(gamma * (conv(x, W) - mean) / sqrt(var + eps) + beta [+ residual])
 * x / residual / out: NHWC storage [n][81][64]; out may alias x.  fp32 accuracy on the fp16 MFMA
 * pipe by operand splitting (csrc/qz_conv.hip).  w16: the weight prepared by the caller as
 * fp16 [2][9][4][64][16] = [hi | lo part][tap 3 ky + kx][16-channel chunk of c_in][c_out][c_in in chunk]
 * of W * scale (scale = a power of two that makes the lo parts fp16 normals), hi = fp16(W scale),
 * lo = fp16(W scale - hi); inv_scale [dev]: ONE float in device memory = 1 / scale (device memory, so that a launch
 * captured in a HIP graph follows the weights when the caller re-derives them in place after training). */
int qz_nn_conv3x3_norm(const float* x /*[dev]*/, const void* w16 /*[dev]*/, const float* gamma /*[dev][64]*/,
                       const float* beta /*[dev][64]*/, const float* residual /*[dev] or NULL*/, float* out /*[dev]*/, int64_t n,
                       const float* inv_scale /*[dev]*/, int relu, float eps, void* stream);
/* The whole residual trunk (policy_value_net.py:75-83: n_blocks x [conv-bn-relu-conv-bn-(+x)-relu])
 * from ONE call; x is updated in place.  fused != 0 (n_blocks <= 8): ONE persistent launch in which
 * a workgroup keeps its leaves' activations in LDS / registers across all layers -- HBM sees the
 * input once and the output once; tmp is not used (may be NULL).  fused == 0: 2 n_blocks launches
 * of the layer kernel above through the scratch tensor tmp (same size as x).  Same values either
 * way.  w16 / gamma / beta: [2 n_blocks] HOST arrays of device pointers, layer order res1.conv1, res1.conv2,
 * res2.conv1, ...; inv_scale: DEVICE array of 2 n_blocks floats (see qz_nn_conv3x3_norm).  fused != 0 with
 * n_blocks > 8 runs layer by layer and needs tmp. */
int qz_nn_trunk(float* x /*[dev] in/out*/, float* tmp /*[dev] or NULL*/, int64_t n, int n_blocks, const void* const* w16 /*[host]*/,
                const float* const* gamma /*[host]*/, const float* const* beta /*[host]*/, const float* inv_scale /*[dev]*/,
                float eps, int fused, void* stream);
/* Trunk + both heads (policy_value_net.py:75-93,155) from ONE call, two launches: the fused trunk
 * launch of qz_nn_trunk with one more stage -- the merged 64 -> 6 head convolution + bn2 / bn3 per
 * leaf + ReLU on the last layer's activations while they are still in LDS -- writes 486 features
 * per leaf into feat (scratch, [n][486] floats, c * 81 + pos, value channels first), then fc1 /
 * fc2 / tanh and fc3 / softmax read them.  x [n][81][64] (the first layer's output) is only read;
 * the trunk output never reaches HBM.  w6_16: the merged head weight [6][64][3][3] prepared like
 * w16 but with 32 output columns: fp16 [2][9][4][32][16], columns 6..31 zero; the other arguments
 * as in qz_nn_trunk / qz_nn_head; inv_scale: DEVICE array of 2 n_blocks + 1 floats, the last one belongs to w6_16.
 * Per-leaf normalisation only (gamma6 must not be NULL). */
int qz_nn_trunk_heads(const float* x /*[dev]*/, int64_t n, int n_blocks, const void* const* w16 /*[host]*/, const float* const* gamma /*[host]*/,
                      const float* const* beta /*[host]*/, const float* inv_scale /*[dev] 2 n_blocks + 1*/, const void* w6_16 /*[dev]*/,
                      const float* gamma6, const float* beta6, const float* w1t, const float* b1, const float* w2, const float* b2,
                      const float* w3t, const float* b3, float* feat /*[dev] n*486 scratch*/, float* p_out /*[dev] n*140*/,
                      float* v_out /*[dev] n*/, float eps, void* stream);
/* The whole leaf evaluation policy_value_fn's network part (policy_value_net.py:145-164 without the
 * host round trips): packed boards -> p [n][140], v [n], two launches.  The fused trunk launch of
 * qz_nn_trunk_heads with one more stage in front: the first layer conv1(state(board)) + bn1 per leaf +
 * ReLU is computed from the 24-byte boards and the three tables of qz_nn_input_layer inside the
 * workgroup that owns the leaf, so no activation tensor is read from or written to HBM at all
 * (boards in, 486 head features out).  terminal: NULL or the leaf flags (a flagged leaf gets the
 * all-zero first-layer pre-activation, like qz_nn_input_layer).  The other arguments as in
 * qz_nn_input_layer / qz_nn_trunk_heads. */
int qz_nn_evaluate(const qz_boards* boards /*[dev] arrays*/, const uint8_t* terminal /*[dev] or NULL*/, int64_t n, const float* hot9,
                   const float* base0, const float* wd, const float* gamma0, const float* beta0, int n_blocks,
                   const void* const* w16 /*[host]*/, const float* const* gamma /*[host]*/, const float* const* beta /*[host]*/,
                   const float* inv_scale /*[dev] 2 n_blocks + 1*/, const void* w6_16, const float* gamma6, const float* beta6,
                   const float* w1t, const float* b1, const float* w2, const float* b2, const float* w3t, const float* b3,
                   float* feat /*[dev] n*486 scratch*/, float* p_out /*[dev] n*140*/, float* v_out /*[dev] n*/, float eps, void* stream);
/* The same arguments as one struct: what the asynchronous self-play loop (qz_selfplay_round) needs to evaluate its miss list. */
typedef struct {
    const float *hot9, *base0, *wd, *gamma0, *beta0;   /* first layer (qz_nn_input_layer)                */
    int32_t n_blocks;                                  /* residual blocks (<= 8)                         */
    const void* const* w16;                            /* [host] 2 n_blocks device pointers              */
    const float* const* gamma;                         /* [host]                                         */
    const float* const* beta;                          /* [host]                                         */
    const float* inv_scale;                            /* [dev] 2 n_blocks + 1 floats                    */
    const void* w6_16;
    const float *gamma6, *beta6, *w1t, *b1, *w2, *b2, *w3t, *b3;
    float eps;
    int32_t precision;   /* 0 = every product as three MFMAs on split fp16 operands: fp32 accuracy, 1e-5 against the reference
                            (the parity mode); 1 = one MFMA per product on fp16 operands with fp32 accumulation: the labelled
                            THROUGHPUT mode (~1e-3 on p / v), a third of the matrix work.  qz_nn_evaluate always runs 0. */
} qz_nn_weights;
/* the engine's current leaf boards (what qz_mcts_select just produced) and their terminal flags,
 * as device pointers owned by the engine: input of qz_nn_input_layer */
int qz_engine_leaf_boards(qz_engine* e, qz_boards* boards_out, const uint8_t** terminal_out);

/* qz_nn_evaluate with the weights as one struct; honours qz_nn_weights.precision. */
int qz_nn_evaluate_w(const qz_boards* boards /*[dev] arrays*/, const uint8_t* terminal /*[dev] or NULL*/, int64_t n, const qz_nn_weights* w,
                     float* feat /*[dev] n*486 scratch*/, float* p_out /*[dev] n*140*/, float* v_out /*[dev] n*/, void* stream);

/* ------------------------------------------------------------- asynchronous self-play
 * The loop of Quoridor.start_self_play / MCTSPlayer.choose_action / MCTS.get_move_probs (quoridor.py:582-593,
 * mcts.py:129-144, 172-187) for every board ON ITS OWN CLOCK.  A ROUND is
 *     qz_selfplay_advance     every board: consume the evaluation it was waiting for (TreeNode.expand +
 *                             update_recursive, mcts.py:27-62), then playouts (mcts.py:103-127) until one meets a leaf
 *                             that needs the network; terminal leaves and leaves whose evaluation is in the MEMO
 *                             (policy_value_fn on a batch of one is a pure function of the 24-byte board:
 *                             policy_value_net.py:145-164) are resolved in place.  After n_playout playouts the board
 *                             plays its move (everything qz_mcts_finish_move does) and goes on from the new root.
 *                             The leaves that need the network are compacted into the engine's MISS LIST.
 *     qz_selfplay_leaf_rules  Quoridor.actions() of the miss list (the same kernel as qz_mcts_leaf_inputs)
 *     qz_selfplay_evaluate    the network on the miss list (qz_nn_evaluate on the first *n boards)
 *     qz_selfplay_round_tail  store the evaluations in the memo; hand replaced trees back to the pool
 * and qz_selfplay_round is the four in one call, with what does not depend on the network moved beside it: the rules
 * op and the moves of the boards that have done their playouts run on a second stream of the engine while the trunk
 * has the matrix cores (the round's tail waits for both).  Nothing synchronises with the host; the whole round can be
 * captured in a HIP graph (the miss count stays on the device; the second stream joins the capture through events).  Per board the operations and their order are those of the lock-step
 * entry points above, so trees, pi, sampled moves and harvested tuples are bit-identical to a lock-step run of the
 * same seed; only the interleaving between boards differs.  Finished games wait for qz_harvest as before.
 *   max_playouts  playouts a board may START per round (1 = the lock-step cadence: n_playout + 1 rounds per move)
 *   budget_us     a board starts no new playout once this much wall time of the launch has passed (0 = no limit)
 *   auto_finish   0: boards stop at n_playout and the host calls qz_mcts_finish_move (once no board is waiting:
 *                 qz_stats.waiting_boards == 0).  Non-zero: from then on the engine's boards are on their own clocks -- one
 *                 may be waiting for an evaluation or hold a move whose subtree copy is left for its next launch -- and
 *                 the lock-step tree entry points (qz_mcts_descend / select* / expand_backup* / update_with_move /
 *                 finish_move) return QZ_E_INVALID until qz_engine_reset or qz_engine_set_boards(reset_trees).  So does
 *                 qz_selfplay_advance / qz_selfplay_round with auto_finish = 0 on such an engine: a board may be sitting
 *                 out of k_advance with a subtree copy that only the moves' launch (auto_finish) continues, and would
 *                 sit out for ever */
int qz_selfplay_advance(qz_engine* e, int max_playouts, int budget_us, int auto_finish, void* stream);
int qz_selfplay_leaf_rules(qz_engine* e, void* stream);
int qz_selfplay_evaluate(qz_engine* e, const qz_nn_weights* w, void* stream);
int qz_selfplay_round_tail(qz_engine* e, void* stream);
int qz_selfplay_round(qz_engine* e, const qz_nn_weights* w, int max_playouts, int budget_us, int auto_finish, void* stream);
/* which of the engine's two miss counters the NEXT qz_selfplay_advance uses (0 | 1; qz_selfplay_round_tail flips it,
 * qz_engine_reset / qz_engine_set_boards(reset_trees) set it to 0).  A HIP graph captured over whole rounds bakes the
 * counter's address in: replay it only while this value is what it was at capture time (SelfPlayEngine.capture_rounds
 * keeps one graph per value). */
int qz_selfplay_parity(qz_engine* e);
/* the miss list of the round in progress (between qz_selfplay_advance and qz_selfplay_round_tail), engine-owned
 * device memory: boards, *n_dev = how many, their legal sets / network outputs once the two calls above have run.
 * For callers that evaluate the list themselves (another network) and for the tests. */
int qz_selfplay_misses(qz_engine* e, qz_boards* boards_out, const int32_t** n_dev_out, uint32_t** mask5_out /*[n][5]*/,
                       float** p_out /*[n][140]*/, float** v_out /*[n]*/);
/* the weights changed (training step, checkpoint load): every stored evaluation is dead.  O(1): bumps the epoch the
 * memo's keys carry. */
int qz_memo_flush(qz_engine* e, void* stream);

/* self-test hook for the GPU tests: out[i] <- device sqrt((double)i), i < n.  The PUCT term
 * uses np.sqrt(parent visits) in float64 (mcts.py:69); the test checks the device result is
 * correctly rounded. */
int qz_selftest_sqrt(double* out /*[dev]*/, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
