// qz_device.h -- device-side view of the engine: board SoA, per-board tree arenas,
// trajectories.  Plain pointers + sizes, passed to kernels by value.
//
// HBM layout (B = n_boards; everything is allocated once by qz_engine_create):
//   root_{hb,vb,meta}[B], leaf_{hb,vb,meta}[B]      24 B/board each, SoA (include/qz_abi.h)
//   leaf_mask[B][5], leaf_{pnode,pedge}[B], leaf_term[B]
//   tree arenas, double buffered (half h in {0,1}), slot = h*B + b:
//     nodes [2][B][node_cap]  16 B   {edge_off, n_edges, parent_node, parent_edge}
//     eN    [2][B][edge_cap]  u32    visit count of the child TreeNode     (mcts.py:22)
//     eQ    [2][B][edge_cap]  f64    its Q                                  (mcts.py:23)
//     eP    [2][B][edge_cap]  f32    its prior                              (mcts.py:25)
//     eChild[2][B][edge_cap]  u32    node id of the child once expanded, 0 = leaf
//     eAct  [2][B][edge_cap]  u8     action id; a node's edges are stored in the
//                                    reference's actions() order (dict insertion order)
//   traj_board[B][max_plies][3] u64, traj_pi[B][max_plies][140] f32
#pragma once
#include <stdint.h>

#define QZ_N_ACT 140
#define QZ_PLANES_N 2106
#define QZ_NONE 0xFFFFFFFFu
#define QZ_NO_MOVE_U8 255

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
    uint32_t* eN;
    double* eQ;
    float* eP;
    uint32_t* eChild;
    uint8_t* eAct;
    uint8_t* tree_half;
    uint32_t *n_nodes, *n_edges, *root_N;
    // games
    uint32_t *ply, *game_serial, *harvest_off, *harvest_gid;
    uint8_t *status, *winner;
    uint64_t* traj_board;
    float* traj_pi;
    unsigned long long* counters;  // QZ_C_COUNT
};

struct TreeView {
    Node* nodes;
    uint32_t* eN;
    double* eQ;
    float* eP;
    uint32_t* eChild;
    uint8_t* eAct;
};

#if defined(__HIPCC__)
__device__ __forceinline__ TreeView tree_view(const EngineDev& E, int b, uint32_t half) {
    size_t slot = (size_t)half * (size_t)E.n_boards + (size_t)b;
    TreeView t;
    t.nodes = E.nodes + slot * (size_t)E.node_cap;
    size_t eo = slot * (size_t)E.edge_cap;
    t.eN = E.eN + eo;
    t.eQ = E.eQ + eo;
    t.eP = E.eP + eo;
    t.eChild = E.eChild + eo;
    t.eAct = E.eAct + eo;
    return t;
}
#endif
