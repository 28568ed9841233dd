// qz_device.h -- device-side view of the engine: board SoA, per-board tree arenas,
// trajectories.  Plain pointers + sizes, passed to kernels by value.
//
// HBM layout (B = n_boards; everything is allocated once by qz_engine_create):
//   root_{hb,vb,meta}[B], leaf_{hb,vb,meta}[B]      24 B/board each, SoA (include/qz_abi.h)
//   leaf_mask[B][5], leaf_{pnode,pedge}[B], leaf_term[B]
//   tree arenas, double buffered (half h in {0,1}), slot = h*B + b:
//     nodes [2][B][node_cap]  16 B   {edge_off, n_edges, parent_node, parent_edge}
//     edges [2][B][edge_cap]  32 B   one record per child TreeNode of the reference:
//        Q f64 (_Q, mcts.py:23) | N u32 (_n_visits, :22) | P f32 (_P, :25) | child u32 (node id
//        once expanded, 0 = leaf) | coff u32 + cne u8 (edge offset / count of the child's own
//        edges, so the descent needs ONE dependent HBM round trip per level) | act u8
//     A node's edges are consecutive and in the reference's actions() order (dict insertion
//     order).  Records rather than one array per field: a level of the descent then touches one
//     2-4 KB span of one page instead of seven arrays megabytes apart (the first layout was
//     bound by page-table walks: 126 us per select at depth 10).
//   path_edges[B][QZ_PATH_CAP] u32: the edges of the last descent, root first (parallel backup)
//   traj_board[B][max_plies][3] u64, traj_pi[B][max_plies][140] f32
#pragma once
#include <stdint.h>

#define QZ_N_ACT 140
#define QZ_PLANES_N 2106
#define QZ_NONE 0xFFFFFFFFu
#define QZ_NO_MOVE_U8 255
#define QZ_PATH_CAP 256

enum { QZ_PLAYING = 0, QZ_FINISHED = 1 };
enum {
    QZ_C_GAMES = 0,
    QZ_C_PLIES,
    QZ_C_PLAYOUTS,
    QZ_C_LEAF_TERMINAL,
    QZ_C_OVERFLOW,
    QZ_C_ABORTED,
    QZ_C_PENDING_GAMES,
    QZ_C_PENDING_PLIES,
    QZ_C_COUNT
};

struct Node {
    uint32_t edge_off, n_edges, parent_node, parent_edge;
};
struct Edge {
    double Q;
    uint32_t N;
    float P;
    uint32_t child;
    uint32_t coff;
    uint8_t act, cne;
    uint16_t pad16;
    uint32_t pad32;
};
static_assert(sizeof(Edge) == 32, "edge record must be 32 bytes");

struct EngineDev {
    int n_boards, node_cap, edge_cap, max_plies;
    float c_puct, temp, dirichlet_alpha, noise_frac;
    uint64_t seed;
    int is_selfplay, fix_terminal_sign;
    // boards
    uint64_t *root_hb, *root_vb, *root_meta;
    uint64_t *leaf_hb, *leaf_vb, *leaf_meta;
    uint32_t* leaf_mask;
    uint32_t *leaf_pnode, *leaf_pedge;
    uint8_t* leaf_term;
    // trees
    Node* nodes;
    Edge* edges;
    uint32_t *path_edges, *path_len;
    uint8_t* tree_half;
    uint32_t *n_nodes, *n_edges, *root_N;
    // games
    uint32_t *ply, *game_serial, *harvest_off, *harvest_gid;
    uint8_t *status, *winner;
    uint64_t* traj_board;
    float* traj_pi;
    unsigned long long* counters;  // QZ_C_COUNT (touched once per ply / harvest)
    // per-board counters for the per-playout statistics: a shared atomic would serialise all
    // boards of a step on one address (~88 atomics/us on MI355X); summed by qz_engine_stats
    uint32_t *bc_playouts, *bc_terminal, *bc_overflow;
    unsigned long long* bc_levels;
};

struct TreeView {
    Node* nodes;
    Edge* e;
};

#if defined(__HIPCC__)
__device__ __forceinline__ TreeView tree_view(const EngineDev& E, int b, uint32_t half) {
    size_t slot = (size_t)half * (size_t)E.n_boards + (size_t)b;
    TreeView t;
    t.nodes = E.nodes + slot * (size_t)E.node_cap;
    t.e = E.edges + slot * (size_t)E.edge_cap;
    return t;
}
#endif
