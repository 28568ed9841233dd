// qz_device.h -- device-side view of the engine: board SoA, paged tree storage, paged
// trajectories.  Plain pointers + sizes, passed to kernels by value.
//
// HBM layout (B = n_boards; everything is allocated once by qz_engine_create):
//   root_{hb,vb,meta}[B], leaf_{hb,vb,meta}[B]      24 B/board each, SoA (include/qz_abi.h)
//   leaf_mask[B][5], leaf_pedge[B], leaf_term[B]
//
//   Trees: ONE pool of 64-KB pages (QZ_PAGE_EDGES = 2,048 edge records of 32 B) shared by all
//   boards, handed out by a free stack.  A tree addresses its edges by a LOGICAL index
//   0 .. n_edges-1 that its page table (tree_ptab, QZ_TREE_PT entries) maps to pool pages:
//       physical = tree_ptab[slot][e >> 11] * 2048 + (e & 2047)
//   so a board only owns the pages its tree really fills (round 1 reserved 2 x 136,896 edges
//   per board = 8.8 MB and used 0.5-2 MB of them; with the pool 32,768 boards x n_playout=400
//   fit one MI355X).  One edge record per child TreeNode of the reference:
//        Q f64 (_Q, mcts.py:23) | N u32 (_n_visits, :22) | P f32 (_P, :25) | pedge u32 (physical
//        index of the edge this node hangs under, QZ_NONE for the root's children: the
//        _parent link, :20) | coff u32 + cne u8 (logical offset / count of the child's own edge
//        block once expanded, cne == 0 <=> TreeNode.is_leaf(): the descent needs ONE dependent
//        HBM round trip per level) | act u8
//     A node's edges are consecutive, in the reference's actions() order (dict insertion
//     order), and never straddle a page (the allocation cursor skips to the next page instead),
//     so a wave translates a block's base once (wave-uniform) and lanes add their k.
//     There is no separate node array: everything the reference keeps in a TreeNode lives in
//     the edge that points to it, plus root_eoff/root_ne/root_N per board for the root.
//   Subtree reuse copies the kept subtree breadth-first into FRESH pages (table half h^1), then
//   the old half's pages go back to the stack (k_release, a push-only kernel: kernels that pop
//   never push, so the stack needs no ABA protection).
//   path_edges[B][QZ_PATH_RECS + 1][QZ_PATH_CAP] u32: physical edges of recorded descents (replay) and of the last one (parallel backup), root first
//
//   Trajectories: a second pool of 64-KB pages of dwords; one variable-length record per ply
//       [0] ne | [1] 0 | [2..7] board (hbits, vbits, meta) | [8..8+ne) pi f32 of the root's
//       children in actions() order | ceil(ne/4) dwords of action ids (u8 each)
//   = 60 B for a late-game ply with 5 legal moves, 688 B at most (round 1: 584 B for every
//   ply, 4,096 plies reserved per board, longer games dropped).  Records never straddle a
//   page; a page that cannot take the next record ends with the marker 0xFFFFFFFF.
#pragma once
#include <stdint.h>

#define QZ_N_ACT 140
#define QZ_PLANES_N 2106
#define QZ_NONE 0xFFFFFFFFu
#define QZ_NO_MOVE_U8 255
#ifndef QZ_PATH_RECS
#define QZ_PATH_RECS 16    // descent records per board (k_select), a power of two
#endif
#define QZ_PATH_CAP 2048   // levels of a descent that are recorded (late-game lines are forced and hundreds of plies deep)

#define QZ_PAGE_SHIFT 11
#define QZ_PAGE_EDGES (1u << QZ_PAGE_SHIFT)  // 2,048 x 32 B = 64 KB
#define QZ_TREE_PT 128                       // page-table entries per tree: 262,144 logical edges
#define QZ_TPAGE_DWORDS 16384                // default trajectory page: 64 KB (EngineDev.traj_page_dwords)
#define QZ_TRAJ_PT 256                       // page-table entries per game: >= 24k plies, ~140k typical
#define QZ_TRAJ_HDR 8u
#define QZ_TRAJ_SKIP 0xFFFFFFFFu
#define QZ_DROP_LOG 4096u                    // dropped games remembered (root board, cause, ply, board slot): a ring

enum { QZ_PLAYING = 0, QZ_FINISHED = 1, QZ_ABORTED = 2 };
enum {
    QZ_C_GAMES = 0,
    QZ_C_PLIES,
    QZ_C_ABORT_NO_MOVE,     // no legal move at the root
    QZ_C_ABORT_MAX_PLIES,   // qz_config.max_plies reached / trajectory page table full
    QZ_C_ABORT_POOL,        // trajectory pool exhausted
    QZ_C_PENDING_GAMES,
    QZ_C_PENDING_PLIES,
    QZ_C_BAD_FORCED,        // forced moves that were not children of the root (sticky error flag)
    QZ_C_DEEP_DESCENTS,     // descents of >= 256 levels ...
    QZ_C_DEEP_COLD,         // ... whose board's previous descent record was shorter than half of that
    QZ_C_DEEP_LEVELS,       // ... their levels
    QZ_C_DEEP_REPLAYED,     // ... of which the replay of the recorded descent confirmed this many
    QZ_C_ROUNDS,            // k_advance launches (asynchronous self-play)
    QZ_C_MEMO_INSERTS,      // evaluations stored in the memo
    QZ_C_MEMO_LOCKED,       // ... skipped because another wave held the bucket's lock
    QZ_C_ABORT_DEPTH,       // games dropped because a descent was deeper than qz_config.max_depth (the reference's RecursionError)
    QZ_C_RUNAWAY,           // descents cut off because they were deeper than a tree can be (corrupted storage; must stay 0)
    QZ_C_COMPACT_SLICES,    // subtree copies that stopped at their launch's budget and went on in the next launch
    QZ_C_DROPS_LOGGED,      // dropped games written to drop_log so far (the log keeps the last QZ_DROP_LOG of them)
    QZ_C_MISS_OVERFLOW,     // leaves that found the miss list full (a stale counter: must stay 0); their boards redo the descent next launch
    QZ_C_COUNT
    // behind the QZ_C_COUNT words: QZ_C_SPREAD words each for the memo's inserts and lock skips -- k_round_tail's thousands of
    // wavefronts bump the one of their index instead of ONE address (same-address atomics queue at ~10-20 ns each: 4,700 inserts
    // a round were ~50 of the tail's 90 us); qz_engine_stats adds them to memo_inserts / memo_locked
};
// pool bookkeeping words (int): free-stack tops and low-water marks
enum { QZ_P_TREE_TOP = 0, QZ_P_TREE_LOW, QZ_P_TRAJ_TOP, QZ_P_TRAJ_LOW, QZ_P_COUNT };

// (dword 0-1 Q, 2 N, 3 P, 4 coff, 5 act | cne << 8 | rid << 16, 6 pedge, 7 spare: what a descent needs of a PARENT -- N, child block,
// child count -- is one 16-byte load at dword 2, of a chosen child -- child block, action, child count -- one 8-byte load at dword 4)
struct Edge {
    double Q;
    uint32_t N;
    float P;
    uint32_t coff;
    uint8_t act, cne;
    uint16_t rid;    // 1 + the descent record (k_select) that went through this edge last; 0 = none.  A hint: re-verified
    uint32_t pedge;
    uint32_t spare;
};
static_assert(sizeof(Edge) == 32, "edge record must be 32 bytes");
typedef uint32_t qz_u32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));  // 16 bytes at an 8-byte boundary: one global_load_dwordx4
typedef uint32_t qz_u32x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef uint32_t qz_u32x3_a16 __attribute__((ext_vector_type(3), aligned(16)));  // 12 bytes: one global_store_dwordx3

// Which formulation of the rules op (actions() + state()) a call uses: per engine / per call,
// never process-global.  variant: 0 = by batch size (k_wave_rules below 8,192 boards, pooled
// pipeline above), 2 | 3 | 4 = k_wave_rules with 2 | 1 | 4 boards per wavefront, 5 | 6 = 3 with one
// base-path search per lane | with ordinary stores for the planes (A/B partners of 3), 8 | 12 | 16 |
// 24 | 32 = pooled pipeline with that many boards per mask workgroup.
constexpr int QZ_C_SPREAD = 64;
constexpr int QZ_C_TOTAL = QZ_C_COUNT + 2 * QZ_C_SPREAD;

struct RulesOpts {
    int variant = 0;
    int detour_pooled = 1, detour_wave = 0;  // pool_k1's detour_mode per kernel family
    int enc_split_pct = 50;                  // share of the encoder tiles beside the path groups
};

// Leaf-evaluation memo.  policy_value_fn on a batch of one (policy_value_net.py:145-164, BatchNorm in training mode)
// is a pure function of the 24-byte board, and a long game revisits the same few thousand boards millions of times
// (profiles/round3/leaf_duplicates_*.json), so (legal set, priors at the pawn moves / all moves, value) are kept per
// board, keyed on ALL 24 bytes + the weight epoch: a hit returns exactly the bits the evaluation would produce.
//   small table: leaves whose mover has no wall left (<= 12 pawn moves): buckets of 4 entries x 32 dwords (512 B, one
//                coalesced load per probe):  [0..5] hb, vb, meta | epoch << 48   [6] v   [7] legal pawn bits
//                [8..19] p of pawn codes 0..11   [31] of entry 0: the bucket's insert lock
//   big table:   every other live leaf: buckets of 2 entries x 160 dwords (640 B): [0..5] key  [6] v  [8..12] mask5
//                [15] of entry 0: lock   [16..155] p[140]
// Probes (k_advance / k_rows) and inserts (k_round_tail) never run at the same time (same stream, different launches), so
// readers need no protocol.  Inserters: a bucket takes ONE insert per round -- its lock word holds the number of the last
// round in which a wavefront wrote it (QZ_C_ROUNDS | 2^31: never 0), taken by compare-and-swap from the value the bucket was read
// with; nobody unlocks (memo_insert).  Consequences, all of them about WHEN a board gets an answer, never about what it
// computes: (a) two different boards hashing to one bucket in the same round -- the loser is skipped (qz_stats.memo_locked) and
// evaluated again the next time it is met; WHICH of the two loses is a race, so run-level counters (nn_evals, memo_hits, the
// round a board gets its answer in) differ between otherwise identical runs -- search results, games and tuples do not
// (tests/test_gpu_async.py compares tuples, not counters; the one test that compares two engines round for round runs without the
// memo); (b) round numbers come from k_advance's launches: two tails without an advance between them (the split entry points
// called out of order) carry the same number, and the second one's inserts into buckets the first wrote are skipped.
#define QZ_MEMO_S_DW 32
#define QZ_MEMO_S_WAYS 4
#define QZ_MEMO_B_DW 160
#define QZ_MEMO_B_WAYS 2
struct MemoDev {
    uint32_t* small;      // [small_buckets][4][32]
    uint32_t* big;        // [big_buckets][2][160]
    uint32_t small_mask;  // buckets - 1 (a power of two); tables == nullptr: memo off
    uint32_t big_mask;
    uint32_t* epoch;      // [1] device word: entries of other epochs are dead (qz_memo_flush: the weights changed)
};

struct EngineDev {
    int n_boards, node_cap, edge_cap, max_plies;
    int n_playout;
    int max_depth;       // drop a game whose descent is longer than this many levels (0: never)
    int compact_edges;   // asynchronous loop: moves keep the subtree in place while the tree's cursor is below this (<= 0: every move copies)
    int tree_pool_pages, traj_pool_pages;
    uint32_t traj_page_dwords;
    float c_puct, temp, dirichlet_alpha, noise_frac;
    uint64_t seed;
    int is_selfplay, fix_terminal_sign;
    int select_opts;
    // boards
    uint64_t *root_hb, *root_vb, *root_meta;
    uint64_t *leaf_hb, *leaf_vb, *leaf_meta;
    uint32_t* leaf_mask;
    uint32_t* leaf_pedge;
    uint8_t* leaf_term;
    // trees
    Edge* edge_pool;          // [tree_pool_pages][QZ_PAGE_EDGES]
    uint32_t* tree_ptab;      // [2][B][QZ_TREE_PT]
    uint32_t* tree_npages;    // [2][B]
    uint32_t* free_tree;      // [tree_pool_pages] stack of free pages
    uint32_t* traj_pool;      // [traj_pool_pages][traj_page_dwords]
    uint32_t* traj_ptab;      // [B][QZ_TRAJ_PT]
    uint32_t *traj_npages, *traj_cursor;  // [B]
    uint32_t* free_traj;      // [traj_pool_pages]
    int* pool_words;          // QZ_P_COUNT
    // k_select's descent records: QZ_PATH_RECS root-to-leaf paths per board + (last slot) the descent of this step
    uint32_t* path_edges;             // [B][QZ_PATH_RECS + 1][QZ_PATH_CAP] chosen physical edge per level, root first
    unsigned long long* path_blocks;  // [B][QZ_PATH_RECS + 1][QZ_PATH_CAP] (first physical edge << 8 | edge count) of the node at that level
    uint32_t* path_len;               // [B] length of the last descent (may exceed QZ_PATH_CAP: then the backup walks parent links)
    uint32_t *rec_len, *rec_stamp;    // [B][QZ_PATH_RECS] recorded levels; clock value when the record last confirmed levels
    uint32_t *rec_last, *rec_clock;   // [B] the record of the previous descent; descents so far
    uint8_t* tree_half;
    uint8_t* release;         // [B] bit0: the other table half holds pages to give back
    uint32_t *n_nodes, *n_edges, *root_N, *root_eoff, *root_ne;
    // games
    uint32_t *ply, *game_serial, *harvest_off, *harvest_gid;
    uint8_t *status, *winner;
    unsigned long long* counters;  // QZ_C_COUNT (touched once per ply / harvest)
    unsigned long long* drop_log;  // [QZ_DROP_LOG][4] root hbits, vbits, meta, (cause | ply << 8 | board slot << 40) of the games dropped last
    // per-board counters for the per-playout statistics: a shared atomic would serialise all
    // boards of a step on one address (~88 atomics/us on MI355X); summed by qz_engine_stats
    uint32_t *bc_playouts, *bc_terminal, *bc_overflow, *bc_nonfinite, *bc_maxdepth;
    unsigned long long *bc_levels, *bc_scanned, *bc_expanded;  // tree levels walked, edge records read by k_select, edges created
    // asynchronous self-play (k_advance): every board runs playouts on its own until it meets a leaf that needs the
    // network (a memo miss); those leaves are compacted into the miss list, evaluated as one batch, and consumed by the
    // next launch.  A board that has done n_playout playouts plays its move in the same launch.
    MemoDev memo;
    uint32_t* pl_done;        // [B] playouts done on the current root
    uint32_t* pend_slot;      // [B] miss-list slot of the leaf this board waits for, QZ_NONE = none
    uint32_t* compact_at;     // [B] allocation cursor at which the board's next move compacts (compact_edges, or twice the tree's size after its last compaction)
    uint32_t* compact_state;  // [B][8] a subtree copy that stopped at its launch's budget (wave_reroot): in progress, scan position, cursor, pages, nodes, cut, pool empty, root offset
    uint32_t* reroot_pend;    // [B] the subtree copy of the last move, left for the next k_advance launch: 0 none, 1 fresh root, e + 2 keep edge e
    int* miss_count;          // [8]: [0], [1] slots used by the misses of even / odd rounds; [2] rounds finished (k_round_tail); [4..5] (64 bits) the
                              // s_memrealtime stamp << 20 | round number of the running launch's first wavefront (select_opts bit 3)
    uint64_t *miss_hb, *miss_vb, *miss_meta;  // [B] the leaves awaiting evaluation, compacted
    uint32_t* miss_mask;      // [B][5] their legal sets (rules op on the miss list)
    float *miss_p, *miss_v;   // [B][140], [B] the network's output per slot
    uint32_t *bc_memo_hits, *bc_evals;  // [B] leaves answered by the memo / sent to the network
    uint32_t *bc_open_rounds, *bc_open_plies;  // [B] launches / plies with a root whose mover still has walls
    uint32_t* rows_list;      // [2 + B] k_rows' boards of this round, compacted by k_rows_scout: [0] count, [1] the queue's cursor (both cleared in front of every scout), then the boards
};

#if defined(__HIPCC__)
// A tree's page table, held by the wave: lane l keeps entry l in a register (pt0); entries 64..127 -- a tree of more than
// 131,072 edge records, which a search of a few hundred playouts per move reaches only past its compaction threshold --
// are read from the table in memory where they are needed (round 3 kept them in a second register per lane: one of the
// registers k_advance does not have at seven or eight wavefronts per SIMD).
struct TreeView {
    Edge* pool;
    uint32_t* ptab;   // this tree's table in HBM (QZ_TREE_PT entries)
    uint32_t pt0;
};
__device__ __forceinline__ size_t tree_slot(const EngineDev& E, int b, uint32_t half) {
    return (size_t)half * (size_t)E.n_boards + (size_t)b;
}
__device__ __forceinline__ TreeView tree_view(const EngineDev& E, int b, uint32_t half, int lane) {
    TreeView t;
    t.pool = E.edge_pool;
    t.ptab = E.tree_ptab + tree_slot(E, b, half) * QZ_TREE_PT;
    t.pt0 = t.ptab[lane];
    return t;
}
// an entry beyond the register copy, from memory (written by this wave's lane 0 in tree_alloc: device-scope load, past the L1)
__device__ __forceinline__ uint32_t tree_ptab_high(const TreeView& t, uint32_t pg) {
    return __hip_atomic_load(t.ptab + pg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// physical index of logical edge e; e must be wave-uniform
__device__ __forceinline__ uint32_t tree_phys(const TreeView& t, uint32_t e) {
    const uint32_t pg = (uint32_t)__builtin_amdgcn_readfirstlane((int)(e >> QZ_PAGE_SHIFT));
    uint32_t page;
    if (pg < 64u) page = (uint32_t)__builtin_amdgcn_readlane((int)t.pt0, (int)pg);
    else page = (uint32_t)__builtin_amdgcn_readfirstlane((int)tree_ptab_high(t, pg));
    return (page << QZ_PAGE_SHIFT) | (e & (QZ_PAGE_EDGES - 1u));
}
// the same for a per-lane e (ALL lanes must call it: the page table is read from other lanes' registers)
__device__ __forceinline__ uint32_t tree_phys_lanes(const TreeView& t, uint32_t e) {
    const uint32_t pg = e >> QZ_PAGE_SHIFT;
    // (ds_bpermute directly: __shfl derives the lane's own index first and keeps it in a register of its own)
    uint32_t page = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((pg & 63u) << 2), (int)t.pt0);
    if (__ballot(pg >= 64u)) {  // wave-uniform, rare
        if (pg >= 64u) page = tree_ptab_high(t, pg < (uint32_t)QZ_TREE_PT ? pg : 0u);
    }
    return (page << QZ_PAGE_SHIFT) | (e & (QZ_PAGE_EDGES - 1u));
}
#endif
