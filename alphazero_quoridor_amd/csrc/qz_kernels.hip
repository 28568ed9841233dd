// qz_kernels.hip -- CDNA4 (gfx950) kernels of the Quoridor self-play engine.
//
// Tree kernels (select / expand / backup / finish_move / harvest): ONE 64-lane wavefront per
// board, 4 boards per 256-thread workgroup; board scalars are wave-uniform (SGPRs), the lanes
// are the <=131 edges of a tree node (three rounds).  Cross-lane traffic is ballots, mbcnt
// ranks and shuffle reductions.
// Rules kernels (actions() + state()), chosen by batch size in qzl::movegen_encode:
//   k_wave_rules                          < 8,192 boards: one launch, a wavefront per board
//   k_pool_paths_enc + k_pool_masks_enc   pooled pipeline, where every phase maps lanes to the
//                                         unit it has many of (qz_movegen_pool.h)
//   encoder groups                        ride in both: a tile of 16 boards -> one bit stream in
//                                         LDS -> 16-byte stores
// The first wave-per-board kernel (k_movegen_encode) is kept for A/B runs and as an independent
// implementation in the parity tests (qz_debug_set_movegen_variant).
// No MFMA anywhere: integer / indexing work.
//
// Reference semantics: see qz_rules.h (rules) and the per-kernel comments (mcts.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qz_rules.h"
#include "qz_movegen_pool.h"
#include "qz_device.h"

using namespace qz;

namespace {

constexpr int WPB = 4;  // waves (= boards) per workgroup
constexpr int TPB = 64 * WPB;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint32_t rfl(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint64_t rfl64(uint64_t x) {
    return (uint64_t)rfl((uint32_t)x) | ((uint64_t)rfl((uint32_t)(x >> 32)) << 32);
}
__device__ __forceinline__ uint32_t rdl(uint32_t x, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)x, l); }
__device__ __forceinline__ BB bb_rdl(BB a, int l) { return BB{rdl(a.w0, l), rdl(a.w1, l), rdl(a.w2, l)}; }
__device__ __forceinline__ PathEdges path_rdl(const PathEdges& p, int l) {
    PathEdges r;
    r.pn = bb_rdl(p.pn, l);
    r.ps = bb_rdl(p.ps, l);
    r.pe = bb_rdl(p.pe, l);
    r.pw = bb_rdl(p.pw, l);
    r.jump = rdl(p.jump ? 1u : 0u, l) != 0u;
    r.found = rdl(p.found ? 1u : 0u, l) != 0u;
    return r;
}
// number of set bits of a ballot below this lane
__device__ __forceinline__ int rank_below(uint64_t m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ Board load_board(const uint64_t* hb, const uint64_t* vb, const uint64_t* meta, int b) {
    return unpack(rfl64(hb[b]), rfl64(vb[b]), rfl64(meta[b]));
}

// ============================================================================ rules kernels

struct MoveShared {
    BB layers[WPB][2][84];     // BFS layers of the two base floods (lanes 0 / 1 of each wave)
    uint8_t items[WPB][256];   // work list: ix | horizontal<<6 | (player-1)<<7
    uint8_t res[WPB][256];     // flood results per work item
};

// Quoridor.actions() for one board per wave; returns the three legal sets.
__device__ __forceinline__ void wave_movegen(const Board& bd, MoveShared& sm, int wave, int lane, uint32_t& pawn,
                                             uint64_t& lh, uint64_t& lv) {
    MoveCtx c = make_ctx(bd);
    pawn = c.pawn;
    lh = 0;
    lv = 0;
    if (!c.walls) return;  // quoridor.py:149-156: no wall actions without walls (wave-uniform)

    // one concrete path per player on the current walls: lanes 0 and 1 in parallel
    PathEdges mine;
    mine.pn = mine.ps = mine.pe = mine.pw = bb_zero();
    mine.jump = false;
    mine.found = false;
    if (lane < 2) mine = base_path(c, lane + 1, &sm.layers[wave][lane][0]);
    PathEdges path1 = path_rdl(mine, 0), path2 = path_rdl(mine, 1);
    if (!(path1.found && path2.found)) return;  // somebody is already cut off: every wall "blocks"

    // lane = slot ix; round H then round V share the same lane
    const int ix = lane;
    bool stH = (c.sh >> ix) & 1ull, stV = (c.sv >> ix) & 1ull;
    Blk dH = candidate_delta(ix, true), dV = candidate_delta(ix, false);
    bool nH1 = stH && needs_check(c, path1, 1, ix, dH);
    bool nH2 = stH && needs_check(c, path2, 2, ix, dH);
    bool nV1 = stV && needs_check(c, path1, 1, ix, dV);
    bool nV2 = stV && needs_check(c, path2, 2, ix, dV);
    uint64_t mH1 = __ballot(nH1), mH2 = __ballot(nH2), mV1 = __ballot(nV1), mV2 = __ballot(nV2);
    int o1 = __popcll(mH1), o2 = o1 + __popcll(mH2), o3 = o2 + __popcll(mV1), total = o3 + __popcll(mV2);
    int sH1 = rank_below(mH1), sH2 = o1 + rank_below(mH2), sV1 = o2 + rank_below(mV1), sV2 = o3 + rank_below(mV2);
    if (total > 0) {
        if (nH1) sm.items[wave][sH1] = (uint8_t)(ix | 0x40);
        if (nH2) sm.items[wave][sH2] = (uint8_t)(ix | 0x40 | 0x80);
        if (nV1) sm.items[wave][sV1] = (uint8_t)(ix);
        if (nV2) sm.items[wave][sV2] = (uint8_t)(ix | 0x80);
        wave_sync();
        for (int base = 0; base < total; base += 64) {  // wave-uniform trip count
            int j = base + lane;
            if (j < total) {
                int it = sm.items[wave][j];
                int cix = it & 63;
                bool hz = (it & 0x40) != 0;
                int p = (it & 0x80) ? 2 : 1;
                Blk d = candidate_delta(cix, hz);
                sm.res[wave][j] = candidate_reaches(c, p, cix, hz, d) ? 1 : 0;
            }
        }
        wave_sync();
    }
    bool okH = stH, okV = stV;
    if (nH1) okH = okH && sm.res[wave][sH1];
    if (nH2) okH = okH && sm.res[wave][sH2];
    if (nV1) okV = okV && sm.res[wave][sV1];
    if (nV2) okV = okV && sm.res[wave][sV2];
    lh = __ballot(okH);
    lv = __ballot(okV);
}

__device__ __forceinline__ void store_mask(uint32_t* mask5, int b, int lane, uint32_t pawn, uint64_t lh, uint64_t lv) {
    // 140 bits: [pawn 12][H 64][V 64]
    if (lane < 5) {
        uint32_t w;
        switch (lane) {
            case 0: w = pawn | (uint32_t)(lh << 12); break;
            case 1: w = (uint32_t)(lh >> 20); break;
            case 2: w = (uint32_t)(lh >> 52) | (uint32_t)(lv << 12); break;
            case 3: w = (uint32_t)(lv >> 20); break;
            default: w = (uint32_t)(lv >> 52); break;
        }
        mask5[(size_t)b * 5 + lane] = w;
    }
}

// Quoridor.state(): 2,106 floats per board, written as 1,053 coalesced 8-byte stores
__device__ __forceinline__ void wave_encode(const Board& bd, float* planes, int b, int lane, bool zero) {
    float2* out = reinterpret_cast<float2*>(planes + (size_t)b * QZ_PLANES_N);
#pragma unroll 1
    for (int q = lane; q < QZ_PLANES_N / 2; q += 64) {
        float2 v;
        if (zero) {
            v.x = 0.f;
            v.y = 0.f;
        } else {
            v.x = plane_value(bd, 2 * q);
            v.y = plane_value(bd, 2 * q + 1);
        }
        out[q] = v;
    }
}

template <bool DO_MASK, bool DO_PLANES>
__global__ __launch_bounds__(TPB) void k_movegen_encode(const uint64_t* __restrict__ hb, const uint64_t* __restrict__ vb,
                                                        const uint64_t* __restrict__ meta, int n,
                                                        uint32_t* __restrict__ mask5, float* __restrict__ planes,
                                                        const uint8_t* __restrict__ terminal) {
    __shared__ MoveShared sm;
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= n) return;  // whole wave leaves; no workgroup barrier is used below
    Board bd = load_board(hb, vb, meta, b);
    bool term = terminal ? (rfl(terminal[b]) != 0u) : false;
    if (DO_MASK) {
        uint32_t pawn = 0;
        uint64_t lh = 0, lv = 0;
        if (!term) wave_movegen(bd, sm, wave, lane, pawn, lh, lv);
        store_mask(mask5, b, lane, pawn, lh, lv);
    }
    if (DO_PLANES) wave_encode(bd, planes, b, lane, term);
}

// ---------------------------------------------------------------------------- pooled kernels
// Quoridor.actions() + state() as two launches over an HBM scratch area (qz_movegen_pool.h):
//   k_pool_paths_enc  path groups (lane = (board, player): board context + one shortest base
//                     path per player -> scratch; long dependent chains, no LDS: latency-bound)
//                     + ~35 % of the encoder groups
//   k_pool_masks_enc  two kinds of workgroups in ONE grid so that they overlap on the CUs:
//                     * mask groups: a tile of NB boards: slot tests -> pooled work list ->
//                       floods -> 140-bit masks.  Issue-bound.
//                     * encoder groups: a tile of NBE boards -> 2,106-bit bitmaps in LDS ->
//                       26x9x9 planes with 16-byte stores.  HBM-bound, independent of the masks.
template <int NBE>
struct EncShared {
    EncCtx ec[NBE];
    uint32_t bm[(NBE + 1) * POOL_BM_WORDS];      // word-aligned bitmaps, +1 all-zero pad board
    uint32_t st[POOL_STREAM_WORDS(NBE) + 1];      // the tile as one bit stream (2,106 bits per board)
    __attribute__((aligned(16))) float tbl[16][4];  // nibble -> four floats
};
template <int NBE>
__device__ __forceinline__ void encoder_group(EncShared<NBE>& sm, const uint64_t* __restrict__ hb, const uint64_t* __restrict__ vb,
                                              const uint64_t* __restrict__ meta, int n, const uint8_t* __restrict__ terminal,
                                              float* __restrict__ planes, int b0, int tid) {
    const int nb = (n - b0) < NBE ? (n - b0) : NBE;
    if (tid < nb) {
        Board bd = unpack(hb[b0 + tid], vb[b0 + tid], meta[b0 + tid]);
        enc_ctx_build(sm.ec[tid], bd, terminal ? (terminal[b0 + tid] != 0) : false);
    }
    if (tid >= 64 && tid < 128) {  // a 256 B table: entry e lives in banks 4e..4e+3, so the b128 reads below never conflict
        const int e = (tid - 64) >> 2, j = tid & 3;
        sm.tbl[e][j] = (float)((e >> j) & 1);
    }
    __syncthreads();
    for (int w = tid; w < (nb + 1) * POOL_BM_WORDS; w += 256) {  // lane = one 32-bit word of a bitmap
        int bd = w / POOL_BM_WORDS, k = w - bd * POOL_BM_WORDS;
        sm.bm[w] = bd < nb ? pool_bitmap_word(sm.ec[bd], k) : 0u;
    }
    __syncthreads();
    const int nsw = POOL_STREAM_WORDS(nb);
    for (int w = tid; w < nsw; w += 256) sm.st[w] = pool_stream_word(sm.bm, w);
    __syncthreads();
    // lane = 16 bytes of output.  q advances by 256, so the nibble's shift is lane-invariant and
    // its word index advances by 32: per store one LDS word read, one bit-field extract, one
    // b128 table read.
    float4* out = reinterpret_cast<float4*>(planes + (size_t)b0 * QZ_PLANES_N);
    const int nf = nb * QZ_PLANES_N, nq = nf >> 2;
    const uint32_t sh = (uint32_t)(tid & 7) << 2;
    const float4* tbl = reinterpret_cast<const float4*>(sm.tbl);
#pragma unroll 4
    for (int q = tid; q < nq; q += 256) out[q] = tbl[(sm.st[q >> 3] >> sh) & 15u];
    if ((nf & 3) && tid == 0) {  // odd number of boards in the last tile: 2 floats left
        const int bit = nf - 2;
        const uint32_t two = (sm.st[bit >> 5] >> (bit & 31)) & 3u;
        planes[(size_t)b0 * QZ_PLANES_N + nf - 2] = (float)(two & 1u);
        planes[(size_t)b0 * QZ_PLANES_N + nf - 1] = (float)(two >> 1);
    }
}
template <int NBE>
__global__ __launch_bounds__(256) void k_pool_paths_enc(const uint64_t* __restrict__ hb, const uint64_t* __restrict__ vb,
                                                        const uint64_t* __restrict__ meta, int n, const uint8_t* __restrict__ terminal,
                                                        PoolBoard* __restrict__ recs, PathTab* __restrict__ tabs, int n_path_groups,
                                                        float* __restrict__ planes, int detour_mode) {
    __shared__ EncShared<NBE> sm;
    const int tid = (int)threadIdx.x;
    if ((int)blockIdx.x < n_path_groups) {
        // lane = (board, player): board context + one shortest base path per player -> scratch.
        // Long dependent chains, no LDS: latency-bound, ~1 wave per SIMD chip-wide.
        const int task = (int)blockIdx.x * 256 + tid;
        const int b = task >> 1, p = (task & 1) + 1;
        if (b >= n) return;
        Board bd = unpack(hb[b], vb[b], meta[b]);
        bool term = terminal ? (terminal[b] != 0) : false;
        pool_k1(bd, term, true, p, recs[b], tabs[(size_t)b * 2 + (p - 1)], detour_mode);
        return;
    }
    encoder_group<NBE>(sm, hb, vb, meta, n, terminal, planes, ((int)blockIdx.x - n_path_groups) * NBE, tid);
}

// ---------------------------------------------------------------------------- small batches
// k_wave_rules: for batches too small to saturate the chip the critical path matters, not the
// instruction count: ONE launch, no hand-off through HBM.  Move-generation groups give every
// board a wavefront that runs the same phase functions as the pooled pipeline on LDS-resident
// records (lanes 0/1: base paths; lane = slot: cut tests; lane = work item: floods); encoder
// groups (encoder_group) run beside them in the same grid.
template <int G>
struct WaveBoardShared {
    PoolBoard ctx[G];
    PathTab tab[G][2];
    uint16_t items[G * 256];
};
template <int NBE, int G>
union WaveRulesShared {
    WaveBoardShared<G> w[WPB];
    EncShared<NBE> enc;
};

// G boards per wavefront: the 2G base-path searches of a wave run side by side on 2G lanes (the
// search is one long dependent chain, so it costs a wave the same whether 2 or 8 lanes are
// live), and the work items of the G boards share the flood passes.
template <int NBE, int G>
__global__ __launch_bounds__(256) void k_wave_rules(const uint64_t* __restrict__ hb, const uint64_t* __restrict__ vb,
                                                    const uint64_t* __restrict__ meta, int n, const uint8_t* __restrict__ terminal,
                                                    uint32_t* __restrict__ mask5, float* __restrict__ planes, int n_mg_groups,
                                                    int detour_mode) {
    __shared__ WaveRulesShared<NBE, G> sm;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x < n_mg_groups) {
        const int bw = ((int)blockIdx.x * WPB + wave) * G;  // first board of this wave
        if (bw >= n) return;  // whole wave leaves; only wave-level synchronisation below
        const int ng = (n - bw) < G ? (n - bw) : G;
        if (G == 1) {
            // wave-uniform short cut: a terminal board has no moves, a mover without walls only pawn
            // moves -- one lane, no records, no work list
            const Board bd = unpack(hb[bw], vb[bw], meta[bw]);
            const bool term = terminal ? (terminal[bw] != 0) : false;
            if (term || (bd.cur == 1 ? bd.w1 : bd.w2) <= 0) {
                if (lane == 0) {
                    const int loc = bd.cur == 1 ? bd.p1 : bd.p2, opp = bd.cur == 1 ? bd.p2 : bd.p1;
                    mask5[(size_t)bw * 5] = term ? 0u : pawn_actions_tab(bd.hb, bd.vb, loc, opp, bd.cur);
                }
                if (lane >= 1 && lane < 5) mask5[(size_t)bw * 5 + lane] = 0u;
                return;
            }
        }
        WaveBoardShared<G>& ws = sm.w[wave];
        if (lane < 2 * ng) {
            const int g = lane >> 1, b = bw + g;
            Board bd = unpack(hb[b], vb[b], meta[b]);
            const bool term = terminal ? (terminal[b] != 0) : false;
            pool_k1(bd, term, true, (lane & 1) + 1, ws.ctx[g], ws.tab[g][lane & 1], detour_mode);
        }
        wave_sync();
        int total = 0;
        for (int g = 0; g < ng; g++) {  // wave-uniform
            const uint32_t m = pool_p2(ws.ctx[g], lane);
#pragma unroll
            for (int bit = 0; bit < 4; bit++) {
                const bool need = (m >> bit) & 1u;
                const uint64_t bal = __ballot(need);
                if (need) ws.items[total + rank_below(bal)] = (uint16_t)pool_item(g, lane, bit < 2, (bit & 1) + 1);
                total += __popcll(bal);
            }
        }
        wave_sync();
        for (int base = 0; base < total; base += 64) {  // wave-uniform trip count
            const int j = base + lane;
            if (j < total) {
                const uint32_t item = ws.items[j];
                const int g = (int)(item >> 8), ix = (int)(item & 63u);
                const bool ok = pool_p3(ws.ctx[g], item, ws.tab[g][(item & 0x80u) ? 1 : 0]);
                if (!ok) atomicOr(&ws.ctx[g].blocked[((item & 0x40u) ? 0 : 2) + (ix >> 5)], 1u << (ix & 31));
            }
        }
        wave_sync();
        if (lane < ng) {
            uint32_t m5[5];
            pool_p4(ws.ctx[lane], m5);
#pragma unroll
            for (int w = 0; w < 5; w++) mask5[(size_t)(bw + lane) * 5 + w] = m5[w];
        }
        return;
    }
    encoder_group<NBE>(sm.enc, hb, vb, meta, n, terminal, planes, ((int)blockIdx.x - n_mg_groups) * NBE, tid);
}

template <int NB>
struct MasksShared {
    PoolBoard ctx[NB];
    uint32_t srcpos[NB * 2][21];  // PathTab.srcpos of both players (84 B each), staged from scratch
    uint16_t items[NB * 256];     // every (slot, orientation, player) of every board at worst
    uint32_t n_items;
};
template <int NB, int NBE>
union MasksEncShared {
    MasksShared<NB> m;
    EncShared<NBE> enc;
};

// Second launch of the pooled pipeline: mask groups (issue-bound: slot tests, floods) and
// encoder groups (HBM-bound) are independent of each other, so they share one grid and
// overlap on the CUs.
template <int NB, int NBE>
__global__ __launch_bounds__(256) void k_pool_masks_enc(const PoolBoard* __restrict__ recs, const PathTab* __restrict__ tabs, int n,
                                                        uint32_t* __restrict__ mask5, int n_mask_groups, int enc_tile0,
                                                        const uint64_t* __restrict__ hb, const uint64_t* __restrict__ vb,
                                                        const uint64_t* __restrict__ meta, const uint8_t* __restrict__ terminal,
                                                        float* __restrict__ planes) {
    __shared__ MasksEncShared<NB, NBE> smu;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    if ((int)blockIdx.x >= n_mask_groups) {
        encoder_group<NBE>(smu.enc, hb, vb, meta, n, terminal, planes, (enc_tile0 + (int)blockIdx.x - n_mask_groups) * NBE, tid);
        return;
    }
    MasksShared<NB>& sm = smu.m;
    const int b0 = (int)blockIdx.x * NB;
    const int nb = (n - b0) < NB ? (n - b0) : NB;
    if (tid == 0) sm.n_items = 0u;
    {  // stage the tile's board records in LDS (coalesced dword copy)
        const uint32_t* src = reinterpret_cast<const uint32_t*>(recs + b0);
        uint32_t* dst = reinterpret_cast<uint32_t*>(sm.ctx);
        const int nw = nb * (int)(sizeof(PoolBoard) / 4);
        for (int i = tid; i < nw; i += 256) dst[i] = src[i];
        for (int i = tid; i < nb * 2 * 21; i += 256) {
            int t = i / 21, k = i - t * 21;
            sm.srcpos[t][k] = reinterpret_cast<const uint32_t*>(tabs[(size_t)b0 * 2 + t].srcpos)[k];
        }
    }
    __syncthreads();
    // P2: lane = (board, slot); the 64 lanes of a wave share a board
    for (int base = 0; base < nb * 64; base += 256) {
        int task = base + tid;
        int bd = task >> 6, ix = task & 63;
        uint32_t m = (task < nb * 64) ? pool_p2(sm.ctx[bd], ix) : 0u;
#pragma unroll
        for (int bit = 0; bit < 4; bit++) {
            bool need = (m >> bit) & 1u;
            uint64_t bal = __ballot(need);
            if (bal != 0ull) {  // wave-uniform
                uint32_t pos = 0u;
                if (lane == 0) pos = atomicAdd(&sm.n_items, (uint32_t)__popcll(bal));
                pos = rfl(pos);
                if (need) sm.items[pos + (uint32_t)rank_below(bal)] = (uint16_t)pool_item(bd, ix, bit < 2, (bit & 1) + 1);
            }
        }
    }
    __syncthreads();
    // P3: lane = work item
    const uint32_t ni = sm.n_items;
    for (uint32_t it = (uint32_t)tid; it < ni; it += 256u) {
        uint32_t item = sm.items[it];
        int bd = (int)(item >> 8), ix = (int)(item & 63u);
        const int side = (item & 0x80u) ? 1 : 0;
        bool ok = pool_p3(sm.ctx[bd], item, reinterpret_cast<const uint8_t*>(sm.srcpos[bd * 2 + side]),
                          tabs[(size_t)(b0 + bd) * 2 + side].suffix);
        if (!ok) atomicOr(&sm.ctx[bd].blocked[((item & 0x40u) ? 0 : 2) + (ix >> 5)], 1u << (ix & 31));
    }
    __syncthreads();
    // P4: legal sets -> 140-bit masks
    if (tid < nb) {
        uint32_t m5[5];
        pool_p4(sm.ctx[tid], m5);
#pragma unroll
        for (int w = 0; w < 5; w++) mask5[(size_t)(b0 + tid) * 5 + w] = m5[w];
    }
}

// Quoridor.step() + has_a_winner(): one thread per board, fully coalesced SoA traffic
__global__ __launch_bounds__(256) void k_step(uint64_t* hb, uint64_t* vb, uint64_t* meta, const uint8_t* action, int n,
                                              uint8_t* done, uint8_t* winner) {
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    Board b = unpack(hb[i], vb[i], meta[i]);
    int a = action[i];
    bool d = false;
    if (a < QZ_N_ACT) d = apply_action(b, a);
    hb[i] = b.hb;
    vb[i] = b.vb;
    meta[i] = pack_meta(b);
    if (done) done[i] = d ? 1 : 0;
    if (winner) winner[i] = (uint8_t)winner_of(b);
}

// ============================================================================ tree kernels

// wave-wide argmax with Python max() tie-breaking (first in order = smallest k)
__device__ __forceinline__ void wave_argmax(double& v, int& k) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        double ov = __shfl_xor(v, off, 64);
        int ok = __shfl_xor(k, off, 64);
        if (ov > v || (ov == v && ok < k)) {
            v = ov;
            k = ok;
        }
    }
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
// inclusive prefix sum across the wave
__device__ __forceinline__ double wave_scan(double v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        double o = __shfl_up(v, off, 64);
        if (lane >= off) v += o;
    }
    return v;
}

// MCTS._playout descent (mcts.py:107-113) + TreeNode.select/get_value (mcts.py:37-42, 64-70)
__global__ __launch_bounds__(TPB) void k_select(EngineDev E) {
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    Board bd = load_board(E.root_hb, E.root_vb, E.root_meta, b);
    TreeView T = tree_view(E, b, rfl(E.tree_half[b]));
    uint32_t n_nodes = rfl(E.n_nodes[b]);
    uint32_t pnode = QZ_NONE, pedge = QZ_NONE;
    uint32_t parentN = rfl(E.root_N[b]);
    bool done = false;
    bool live = rfl(E.status[b]) == QZ_PLAYING;
    uint32_t plen = 0u;
    if (live && n_nodes > 0) {
        uint32_t node = 0;
        Node root = T.nodes[0];
        uint32_t eoff = rfl(root.edge_off);
        int ne = (int)rfl(root.n_edges);
        uint32_t* path = E.path_edges + (size_t)b * QZ_PATH_CAP;
        for (int depth = 0; depth < 100000; depth++) {
            double sq = sqrt((double)parentN);  // np.sqrt(self._parent._n_visits), float64
            double best = -__builtin_inf();
            int bestk = 0x7fffffff;
            // everything the descent needs about the winning edge rides along with the
            // candidates, so the next level costs one dependent round trip, not three
            uint32_t mN = 0u, mChild = 0u, mCOff = 0u, mMisc = 0u;
            for (int k = lane; k < ne; k += 64) {
                const Edge ed = T.e[eoff + (uint32_t)k];           // one 32-byte record per lane
                uint32_t N = ed.N;
                float cp = E.c_puct * ed.P;                         // c_puct * self._P in float32
                double u = (double)cp * sq / (double)(1u + N);      // mcts.py:69
                double val = ed.Q + u;                              // mcts.py:70
                uint32_t ch = ed.child, co = ed.coff;
                uint32_t misc = (uint32_t)ed.act | ((uint32_t)ed.cne << 8);
                if (val > best) {
                    best = val;
                    bestk = k;
                    mN = N;
                    mChild = ch;
                    mCOff = co;
                    mMisc = misc;
                }
            }
            wave_argmax(best, bestk);
            const int kk = (int)rfl((uint32_t)bestk);
            const int wl = kk & 63;  // the winning edge is the winning lane's own best candidate
            uint32_t e = eoff + (uint32_t)kk;
            uint32_t misc = rdl(mMisc, wl);
            int a = (int)(misc & 0xFFu);
            uint32_t child = rdl(mChild, wl);
            uint32_t childN = rdl(mN, wl);
            done = apply_action(bd, a);  // game.step(action), mcts.py:113
            pnode = node;
            pedge = e;
            if (lane == 0 && plen < (uint32_t)QZ_PATH_CAP) path[plen] = e;
            plen++;
            if (child == 0u) break;  // TreeNode.is_leaf(): never expanded (or terminal)
            node = child;
            parentN = childN;
            eoff = rdl(mCOff, wl);
            ne = (int)((misc >> 8) & 0xFFu);
        }
    }
    if (lane == 0) {
        E.leaf_hb[b] = bd.hb;
        E.leaf_vb[b] = bd.vb;
        E.leaf_meta[b] = pack_meta(bd);
        E.leaf_pnode[b] = pnode;
        E.leaf_pedge[b] = pedge;
        E.path_len[b] = plen;
        // 0 live leaf; 1 terminal & winner == current_player; 2 terminal & winner != current_player;
        // 3 board not playing (finished, waiting for harvest): ignored by expand_backup
        uint8_t t = 0;
        if (!live) t = 3;
        else if (done) t = (winner_of(bd) == bd.cur) ? 1 : 2;
        E.leaf_term[b] = t;
    }
}

// TreeNode.expand (mcts.py:27-35) + update_recursive (mcts.py:44-62)
__global__ __launch_bounds__(TPB) void k_expand_backup(EngineDev E, const float* __restrict__ p, const float* __restrict__ v) {
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    uint32_t term = rfl(E.leaf_term[b]);
    if (term == 3u) return;
    TreeView T = tree_view(E, b, rfl(E.tree_half[b]));
    uint32_t pnode = rfl(E.leaf_pnode[b]), pedge = rfl(E.leaf_pedge[b]);
    double leaf_value;
    if (term == 0u) {
        leaf_value = (double)v[b];
        uint32_t m0 = rfl(E.leaf_mask[(size_t)b * 5 + 0]), m1 = rfl(E.leaf_mask[(size_t)b * 5 + 1]),
                 m2 = rfl(E.leaf_mask[(size_t)b * 5 + 2]), m3 = rfl(E.leaf_mask[(size_t)b * 5 + 3]),
                 m4 = rfl(E.leaf_mask[(size_t)b * 5 + 4]);
        uint32_t pawn = m0 & 0xFFFu;
        uint64_t lh = ((uint64_t)m0 >> 12) | ((uint64_t)m1 << 20) | ((uint64_t)(m2 & 0xFFFu) << 52);
        uint64_t lv = ((uint64_t)m2 >> 12) | ((uint64_t)m3 << 20) | ((uint64_t)(m4 & 0xFFFu) << 52);
        int k = __popc(pawn) + __popcll(lh) + __popcll(lv);
        uint32_t nn = rfl(E.n_nodes[b]), neu = rfl(E.n_edges[b]);
        if (k > 0) {
            if (nn < (uint32_t)E.node_cap && neu + (uint32_t)k <= (uint32_t)E.edge_cap) {
                for (int a = lane; a < QZ_N_ACT; a += 64) {
                    uint32_t w = a < 32 ? m0 : (a < 64 ? m1 : (a < 96 ? m2 : (a < 128 ? m3 : m4)));
                    if ((w >> (a & 31)) & 1u) {
                        uint32_t e = neu + (uint32_t)order_index(pawn, lh, lv, a);
                        Edge ed;
                        ed.Q = 0.0;
                        ed.N = 0u;
                        ed.P = p[(size_t)b * QZ_N_ACT + a];
                        ed.child = 0u;
                        ed.coff = 0u;
                        ed.act = (uint8_t)a;
                        ed.cne = 0;
                        ed.pad16 = 0;
                        ed.pad32 = 0u;
                        T.e[e] = ed;
                    }
                }
                if (lane == 0) {
                    Node nd;
                    nd.edge_off = neu;
                    nd.n_edges = (uint32_t)k;
                    nd.parent_node = pnode;
                    nd.parent_edge = pedge;
                    T.nodes[nn] = nd;
                    if (pnode != QZ_NONE) {
                        T.e[pedge].child = nn;
                        T.e[pedge].coff = neu;
                        T.e[pedge].cne = (uint8_t)k;
                    }
                    E.n_nodes[b] = nn + 1u;
                    E.n_edges[b] = neu + (uint32_t)k;
                }
            } else if (lane == 0) {
                E.bc_overflow[b] += 1u;
            }
        }
    } else {
        // mcts.py:125: +1 if winner == current_player else -1 (always +1 in practice: the
        // reference does not rotate players on a terminal move)
        leaf_value = (term == 1u) ? 1.0 : -1.0;
        if (E.fix_terminal_sign) leaf_value = -leaf_value;
    }
    // node.update_recursive(-leaf_value) (mcts.py:44-62, 127): the leaf edge gets -leaf_value, its
    // parent +leaf_value, ... up to the root.  The descent recorded its edges, so all levels are
    // updated in parallel (lane = level); a path longer than the record falls back to walking
    // the parent pointers.
    const uint32_t plen = rfl(E.path_len[b]);
    if (plen <= (uint32_t)QZ_PATH_CAP) {
        const uint32_t* path = E.path_edges + (size_t)b * QZ_PATH_CAP;
        for (uint32_t i = (uint32_t)lane; i < plen; i += 64u) {
            uint32_t pe = path[i];
            double val = ((plen - 1u - i) & 1u) ? leaf_value : -leaf_value;
            uint32_t N = T.e[pe].N + 1u;  // mcts.py:51
            double Q = T.e[pe].Q;
            Q += 1.0 * (val - Q) / (double)N;  // mcts.py:53
            T.e[pe].N = N;
            T.e[pe].Q = Q;
        }
    } else if (lane == 0) {
        double val = -leaf_value;
        uint32_t pn = pnode, pe = pedge;
        while (pn != QZ_NONE) {
            uint32_t N = T.e[pe].N + 1u;
            double Q = T.e[pe].Q;
            Q += 1.0 * (val - Q) / (double)N;
            T.e[pe].N = N;
            T.e[pe].Q = Q;
            val = -val;  // mcts.py:61
            Node nd = T.nodes[pn];
            pe = nd.parent_edge;
            pn = nd.parent_node;
        }
    }
    if (lane == 0) {
        E.root_N[b] = E.root_N[b] + 1u;  // the root is updated too
        E.bc_playouts[b] += 1u;
        E.bc_levels[b] += (unsigned long long)plen;
        if (term != 0u) E.bc_terminal[b] += 1u;
    }
}

// softmax(1/temp * log(visits + 1e-10)) over the root's children (mcts.py:6-9, 141-144).
// Returns this lane's probabilities for edges lane, lane+64, lane+128 in pr[3].
__device__ __forceinline__ void root_pi(const TreeView& T, const Node& root, double inv_temp, int lane, double pr[3]) {
    int ne = (int)root.n_edges;
    double x[3];
    double mx = -__builtin_inf();
#pragma unroll
    for (int r = 0; r < 3; r++) {
        int k = lane + 64 * r;
        x[r] = -__builtin_inf();
        if (k < ne) {
            x[r] = inv_temp * log((double)T.e[root.edge_off + k].N + 1e-10);
            mx = fmax(mx, x[r]);
        }
    }
    mx = wave_max(mx);
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        int k = lane + 64 * r;
        pr[r] = (k < ne) ? exp(x[r] - mx) : 0.0;
        s += pr[r];
    }
    s = wave_sum(s);
#pragma unroll
    for (int r = 0; r < 3; r++) pr[r] = pr[r] / s;
}

__global__ __launch_bounds__(TPB) void k_root_pi(EngineDev E, double* __restrict__ pi, int32_t* __restrict__ visits) {
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    for (int a = lane; a < QZ_N_ACT; a += 64) {
        if (pi) pi[(size_t)b * QZ_N_ACT + a] = 0.0;
        if (visits) visits[(size_t)b * QZ_N_ACT + a] = -1;
    }
    if (rfl(E.n_nodes[b]) == 0u) return;
    TreeView T = tree_view(E, b, rfl(E.tree_half[b]));
    Node root = T.nodes[0];
    double pr[3];
    root_pi(T, root, 1.0 / (double)E.temp, lane, pr);
    wave_sync();
#pragma unroll
    for (int r = 0; r < 3; r++) {
        int k = lane + 64 * r;
        if (k < (int)root.n_edges) {
            int a = T.e[root.edge_off + k].act;
            if (pi) pi[(size_t)b * QZ_N_ACT + a] = pr[r];
            if (visits) visits[(size_t)b * QZ_N_ACT + a] = (int32_t)T.e[root.edge_off + k].N;
        }
    }
}

__global__ __launch_bounds__(TPB) void k_root_children(EngineDev E, int32_t* visits, double* q, float* prior, int32_t* root_visits) {
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    for (int a = lane; a < QZ_N_ACT; a += 64) {
        if (visits) visits[(size_t)b * QZ_N_ACT + a] = -1;
        if (q) q[(size_t)b * QZ_N_ACT + a] = 0.0;
        if (prior) prior[(size_t)b * QZ_N_ACT + a] = 0.f;
    }
    if (root_visits && lane == 0) root_visits[b] = (int32_t)E.root_N[b];
    if (rfl(E.n_nodes[b]) == 0u) return;
    TreeView T = tree_view(E, b, rfl(E.tree_half[b]));
    Node root = T.nodes[0];
    wave_sync();
    for (int k = lane; k < (int)root.n_edges; k += 64) {
        uint32_t e = root.edge_off + k;
        int a = T.e[e].act;
        if (visits) visits[(size_t)b * QZ_N_ACT + a] = (int32_t)T.e[e].N;
        if (q) q[(size_t)b * QZ_N_ACT + a] = T.e[e].Q;
        if (prior) prior[(size_t)b * QZ_N_ACT + a] = T.e[e].P;
    }
}

// MCTS.update_with_move (mcts.py:146-151): keep the chosen child's subtree by copying it,
// breadth first, into the other arena half (new root = node 0).  `edge` is the root edge of
// the move or QZ_NONE for a fresh root.
__device__ __forceinline__ void wave_reroot(EngineDev& E, int b, int lane, uint32_t edge) {
    uint32_t half = rfl(E.tree_half[b]);
    TreeView S = tree_view(E, b, half);
    uint32_t child = 0u, childN = 0u;
    if (edge != QZ_NONE) {
        child = rfl(S.e[edge].child);
        childN = rfl(S.e[edge].N);
    }
    uint32_t new_nodes = 0u, new_edges = 0u;
    if (child != 0u) {
        TreeView D = tree_view(E, b, half ^ 1u);
        if (lane == 0) {
            Node r;
            r.edge_off = child;  // temporarily: id of the source node
            r.n_edges = S.nodes[child].n_edges;
            r.parent_node = QZ_NONE;
            r.parent_edge = QZ_NONE;
            D.nodes[0] = r;
        }
        new_nodes = 1u;
        wave_sync();
        for (uint32_t i = 0; i < new_nodes; i++) {
            uint32_t old = rfl(D.nodes[i].edge_off);
            Node on = S.nodes[old];
            uint32_t soff = rfl(on.edge_off);
            int ne = (int)rfl(on.n_edges);
            uint32_t doff = new_edges;
            for (int base = 0; base < ne; base += 64) {
                int k = base + lane;
                bool act = k < ne;
                uint32_t c = 0u;
                if (act) c = S.e[soff + k].child;
                bool has = act && c != 0u;
                uint64_t m = __ballot(has);
                uint32_t nid = new_nodes + (uint32_t)rank_below(m);
                if (act) {
                    Edge ed = S.e[soff + k];
                    ed.child = has ? nid : 0u;
                    ed.coff = 0u;  // fixed up when the child itself is copied
                    ed.cne = 0;
                    D.e[doff + k] = ed;
                    if (has) {
                        Node cn;
                        cn.edge_off = c;  // source id, fixed up when the node is visited
                        cn.n_edges = S.nodes[c].n_edges;
                        cn.parent_node = i;
                        cn.parent_edge = doff + (uint32_t)k;
                        D.nodes[nid] = cn;
                    }
                }
                new_nodes += (uint32_t)__popcll(m);
            }
            if (lane == 0) {
                D.nodes[i].edge_off = doff;
                if (i > 0u) {
                    uint32_t pe = D.nodes[i].parent_edge;
                    D.e[pe].coff = doff;
                    D.e[pe].cne = (uint8_t)ne;
                }
            }
            new_edges += (uint32_t)ne;
            wave_sync();
        }
        if (lane == 0) E.tree_half[b] = (uint8_t)(half ^ 1u);
    }
    if (lane == 0) {
        E.n_nodes[b] = new_nodes;
        E.n_edges[b] = new_edges;
        E.root_N[b] = childN;
    }
}

__device__ __forceinline__ void reset_board_state(EngineDev& E, int b) {  // lane 0 only
    Board o = opening();
    E.root_hb[b] = o.hb;
    E.root_vb[b] = o.vb;
    E.root_meta[b] = pack_meta(o);
    E.n_nodes[b] = 0u;
    E.n_edges[b] = 0u;
    E.root_N[b] = 0u;
    E.ply[b] = 0u;
    E.status[b] = QZ_PLAYING;
    E.winner[b] = 0;
    E.game_serial[b] = E.game_serial[b] + 1u;
}

__global__ __launch_bounds__(256) void k_reset(EngineDev E, int reset_boards) {
    int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= E.n_boards) return;
    if (reset_boards) {
        reset_board_state(E, b);
    } else {
        E.n_nodes[b] = 0u;
        E.n_edges[b] = 0u;
        E.root_N[b] = 0u;
    }
}

__global__ __launch_bounds__(TPB) void k_update_with_move(EngineDev E, const uint8_t* __restrict__ moves) {
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    int mv = (int)rfl(moves[b]);
    uint32_t edge = QZ_NONE;
    if (mv < QZ_N_ACT && rfl(E.n_nodes[b]) > 0u) {
        TreeView T = tree_view(E, b, rfl(E.tree_half[b]));
        Node root = T.nodes[0];
        for (int base = 0; base < (int)root.n_edges; base += 64) {
            int k = base + lane;
            bool hit = k < (int)root.n_edges && T.e[root.edge_off + k].act == mv;
            uint64_t m = __ballot(hit);
            if (m) edge = root.edge_off + (uint32_t)base + (uint32_t)(__ffsll((unsigned long long)m) - 1);
        }
    }
    wave_reroot(E, b, lane, edge);
}

// ---------------------------------------------------------------------------- sampling
// Philox4x32-10, keyed by the engine seed; the counter names (board, game, ply, stream, draw)
struct Philox {
    uint32_t k0, k1;
    __device__ uint4 operator()(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) const {
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; r++) {
            uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b,
                     n3 = (uint32_t)p0;
            c0 = n0;
            c1 = n1;
            c2 = n2;
            c3 = n3;
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
    }
};
__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) {  // (0,1)
    uint64_t x = ((uint64_t)hi << 21) ^ (uint64_t)(lo >> 11);
    return ((double)(x & ((1ull << 53) - 1)) + 0.5) * (1.0 / 9007199254740992.0);
}
// Gamma(alpha, 1) for alpha < 1: Marsaglia-Tsang on alpha+1, then the U^(1/alpha) boost
__device__ double gamma_small(const Philox& ph, uint32_t c0, uint32_t c1, uint32_t c2, double alpha) {
    double d = alpha + 1.0 - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (uint32_t t = 0; t < 64; t++) {
        uint4 r = ph(c0, c1, c2, 2u * t);
        uint4 s = ph(c0, c1, c2, 2u * t + 1u);
        double u1 = u01(r.x, r.y), u2 = u01(r.z, r.w);
        double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        double vv = 1.0 + c * x;
        if (vv <= 0.0) continue;
        vv = vv * vv * vv;
        double u = u01(s.x, s.y);
        if (u < 1.0 - 0.0331 * x * x * x * x || log(u) < 0.5 * x * x + d * (1.0 - vv + log(vv))) {
            double ub = u01(s.z, s.w);
            return d * vv * pow(ub, 1.0 / alpha);
        }
    }
    return alpha;
}

// MCTSPlayer.choose_action tail + one iteration of start_self_play (mcts.py:174-187,
// quoridor.py:585-602)
__global__ __launch_bounds__(TPB) void k_finish_move(EngineDev E, const uint8_t* __restrict__ forced, float* __restrict__ pi_out,
                                                     uint8_t* __restrict__ move_out) {
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    if (move_out && lane == 0) move_out[b] = QZ_NO_MOVE_U8;
    if (rfl(E.status[b]) != QZ_PLAYING) return;
    Board bd = load_board(E.root_hb, E.root_vb, E.root_meta, b);
    uint32_t ply = rfl(E.ply[b]);
    uint32_t n_nodes = rfl(E.n_nodes[b]);
    if (n_nodes == 0u || ply >= (uint32_t)E.max_plies) {
        // no legal move at the root (the reference prints "board is full" and crashes in
        // start_self_play's unpack, mcts.py:195-196) or the trajectory is full: drop the game
        if (lane == 0) {
            atomicAdd((unsigned long long*)&E.counters[QZ_C_ABORTED], 1ull);
            reset_board_state(E, b);
        }
        return;
    }
    TreeView T = tree_view(E, b, rfl(E.tree_half[b]));
    Node root = T.nodes[0];
    uint32_t eoff = rfl(root.edge_off);
    int ne = (int)rfl(root.n_edges);
    double pr[3];
    root_pi(T, root, 1.0 / (double)E.temp, lane, pr);

    // record (board, pi) BEFORE the move (quoridor.py:589-591)
    float* tp = E.traj_pi + ((size_t)b * E.max_plies + ply) * QZ_N_ACT;
    for (int a = lane; a < QZ_N_ACT; a += 64) {
        tp[a] = 0.f;
        if (pi_out) pi_out[(size_t)b * QZ_N_ACT + a] = 0.f;
    }
    wave_sync();
    int act[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        int k = lane + 64 * r;
        act[r] = -1;
        if (k < ne) {
            act[r] = T.e[eoff + k].act;
            tp[act[r]] = (float)pr[r];
            if (pi_out) pi_out[(size_t)b * QZ_N_ACT + act[r]] = (float)pr[r];
        }
    }
    if (lane == 0) {
        uint64_t* tb = E.traj_board + ((size_t)b * E.max_plies + ply) * 3;
        tb[0] = bd.hb;
        tb[1] = bd.vb;
        tb[2] = pack_meta(bd);
    }

    // the move
    int chosen_k = -1;
    int fm = forced ? (int)rfl(forced[b]) : QZ_NO_MOVE_U8;
    if (fm < QZ_N_ACT) {
#pragma unroll
        for (int r = 0; r < 3; r++) {
            uint64_t m = __ballot(act[r] == fm);
            if (m) chosen_k = 64 * r + (__ffsll((unsigned long long)m) - 1);
        }
        if (chosen_k < 0) chosen_k = 0;  // illegal forced move: fall back to the first child
    } else {
        Philox ph{(uint32_t)E.seed, (uint32_t)(E.seed >> 32)};
        uint32_t serial = rfl(E.game_serial[b]);
        double w[3];
        if (E.is_selfplay) {
            // 0.75*probs + 0.25*Dirichlet(alpha * ones(k)) (mcts.py:181)
            double g[3], gs = 0.0;
#pragma unroll
            for (int r = 0; r < 3; r++) {
                int k = lane + 64 * r;
                g[r] = (k < ne) ? gamma_small(ph, (uint32_t)b, serial, (ply << 8) | (uint32_t)k, (double)E.dirichlet_alpha) : 0.0;
                gs += g[r];
            }
            gs = wave_sum(gs);
#pragma unroll
            for (int r = 0; r < 3; r++) w[r] = (1.0 - (double)E.noise_frac) * pr[r] + (double)E.noise_frac * (g[r] / gs);
        } else {
#pragma unroll
            for (int r = 0; r < 3; r++) w[r] = pr[r];
        }
        // np.random.choice(acts, p=w): cdf = cumsum(w); idx = searchsorted(cdf/cdf[-1], u, 'right')
        double carry = 0.0, cdf[3];
#pragma unroll
        for (int r = 0; r < 3; r++) {
            cdf[r] = carry + wave_scan(w[r], lane);
            carry = __shfl(cdf[r], 63, 64);
        }
        uint4 ur = ph((uint32_t)b, serial, (ply << 8) | 0xFFu, 0xC401CEu);
        double target = u01(ur.x, ur.y) * carry;
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            int k = lane + 64 * r;
            cnt += __popcll(__ballot(k < ne && cdf[r] <= target));
        }
        chosen_k = cnt < ne ? cnt : ne - 1;
    }
    uint32_t edge = eoff + (uint32_t)chosen_k;
    int mv = (int)rfl(T.e[edge].act);
    if (move_out && lane == 0) move_out[b] = (uint8_t)mv;

    // update_with_move(move) in self-play, update_with_move(-1) otherwise (mcts.py:182,187)
    wave_reroot(E, b, lane, E.is_selfplay ? edge : QZ_NONE);

    // self.step(move); has_a_winner() (quoridor.py:593-596)
    bool done = apply_action(bd, mv);
    if (lane == 0) {
        E.root_hb[b] = bd.hb;
        E.root_vb[b] = bd.vb;
        E.root_meta[b] = pack_meta(bd);
        E.ply[b] = ply + 1u;
        atomicAdd((unsigned long long*)&E.counters[QZ_C_PLIES], 1ull);
        if (done) {
            E.status[b] = QZ_FINISHED;
            E.winner[b] = (uint8_t)winner_of(bd);
            atomicAdd((unsigned long long*)&E.counters[QZ_C_PENDING_GAMES], 1ull);
            atomicAdd((unsigned long long*)&E.counters[QZ_C_PENDING_PLIES], (unsigned long long)(ply + 1u));
        }
    }
}

// ---------------------------------------------------------------------------- harvest
// exclusive prefix over finished boards (board order) -> tuple offsets / game ids
__global__ __launch_bounds__(1024) void k_harvest_scan(EngineDev E) {
    __shared__ uint32_t s_p[1024], s_g[1024];
    __shared__ uint32_t base_p, base_g;
    if (threadIdx.x == 0) {
        base_p = 0;
        base_g = 0;
    }
    __syncthreads();
    for (int start = 0; start < E.n_boards; start += 1024) {
        int b = start + (int)threadIdx.x;
        bool fin = b < E.n_boards && E.status[b] == QZ_FINISHED;
        uint32_t np = fin ? E.ply[b] : 0u, ng = fin ? 1u : 0u;
        s_p[threadIdx.x] = np;
        s_g[threadIdx.x] = ng;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            uint32_t ap = 0, ag = 0;
            if ((int)threadIdx.x >= off) {
                ap = s_p[threadIdx.x - off];
                ag = s_g[threadIdx.x - off];
            }
            __syncthreads();
            s_p[threadIdx.x] += ap;
            s_g[threadIdx.x] += ag;
            __syncthreads();
        }
        if (fin) {
            E.harvest_off[b] = base_p + s_p[threadIdx.x] - np;
            E.harvest_gid[b] = base_g + s_g[threadIdx.x] - ng;
        }
        __syncthreads();
        if (threadIdx.x == 1023) {
            base_p += s_p[1023];
            base_g += s_g[1023];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        E.counters[QZ_C_GAMES] += base_g;
        E.counters[QZ_C_PENDING_GAMES] = 0;
        E.counters[QZ_C_PENDING_PLIES] = 0;
    }
}

// quoridor.py:596-610: z = +1 where the recorded mover is the winner, else -1; then reset
__global__ __launch_bounds__(TPB) void k_harvest_copy(EngineDev E, uint64_t* t_hb, uint64_t* t_vb, uint64_t* t_meta,
                                                      float* __restrict__ t_pi, float* __restrict__ t_z,
                                                      int32_t* __restrict__ t_game, long long cap) {
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    const int b = (int)blockIdx.x * WPB + wave;
    if (b >= E.n_boards) return;
    if (rfl(E.status[b]) != QZ_FINISHED) return;
    uint32_t n = rfl(E.ply[b]), off = rfl(E.harvest_off[b]), gid = rfl(E.harvest_gid[b]);
    int win = (int)rfl(E.winner[b]);
    for (uint32_t i = 0; i < n; i++) {
        long long o = (long long)off + i;
        if (o >= cap) break;
        const uint64_t* tb = E.traj_board + ((size_t)b * E.max_plies + i) * 3;
        const float* tp = E.traj_pi + ((size_t)b * E.max_plies + i) * QZ_N_ACT;
        for (int a = lane; a < QZ_N_ACT; a += 64) t_pi[(size_t)o * QZ_N_ACT + a] = tp[a];
        if (lane == 0) {
            uint64_t m = tb[2];
            t_hb[o] = tb[0];
            t_vb[o] = tb[1];
            t_meta[o] = m;
            int mover = (int)((m >> 32) & 0xFF);
            t_z[o] = (mover == win) ? 1.0f : -1.0f;
            if (t_game) t_game[o] = (int32_t)gid;
        }
    }
    wave_sync();
    if (lane == 0) reset_board_state(E, b);
}

__global__ void k_sqrt_table(double* out, int n) {  // self-test helper: device sqrt(double(i))
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) out[i] = sqrt((double)i);
}

}  // namespace

// ============================================================================ launchers
namespace qzl {

static inline dim3 wave_grid(int n) { return dim3((unsigned)((n + WPB - 1) / WPB)); }

int g_enc_split_pct = 70;
int g_detour_pooled = 1, g_detour_wave = 0;  // pool_k1's detour_mode per kernel family
int g_movegen_variant = 0;  // 0 = by batch size; 1 = first wave-per-board kernel (A/B); 2/3/4 = k_wave_rules with 2/1/4 boards per wave; 8..32 = pooled, forced tile

constexpr int NBE = 16;  // boards per encoder group

template <int NB>
static void launch_masks_enc(const PoolBoard* recs, const PathTab* tabs, int n, uint32_t* mask5, const uint64_t* hb,
                             const uint64_t* vb, const uint64_t* meta, const uint8_t* terminal, float* planes, int enc_tile0,
                             int n_enc_groups, hipStream_t s) {
    const int n_mask_groups = mask5 ? (n + NB - 1) / NB : 0;
    if (n_mask_groups + n_enc_groups == 0) return;
    hipLaunchKernelGGL((k_pool_masks_enc<NB, NBE>), dim3((unsigned)(n_mask_groups + n_enc_groups)), dim3(256), 0, s, recs, tabs, n,
                       mask5, n_mask_groups, enc_tile0, hb, vb, meta, terminal, planes);
}

size_t movegen_scratch_bytes(int n) { return (size_t)n * (sizeof(PoolBoard) + 2 * sizeof(PathTab)); }

hipError_t movegen_encode(const uint64_t* hb, const uint64_t* vb, const uint64_t* meta, int n, uint32_t* mask5,
                          float* planes, const uint8_t* terminal, void* scratch, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    // Small batches are latency-bound: one launch, a wavefront per board, no hand-off through
    // HBM (k_wave_rules).  From ~8k boards on the chip is saturated and the pooled two-launch
    // pipeline, which packs lanes better, wins.
    if (g_movegen_variant == 2 || g_movegen_variant == 3 || g_movegen_variant == 4 || (g_movegen_variant == 0 && n < 8192)) {
        const int n_enc_groups = planes ? (n + NBE - 1) / NBE : 0;
        // boards per wavefront: on bench trees (late-game boards, many without walls left) one board per
        // wavefront measured 29.1 us vs 33.0 (two) / 34.6 (four) at 4,096 boards; on the synthetic
        // S-mid set two were slightly ahead (40.4 vs 42.9 us).  The in-situ number decides.
        const int G = g_movegen_variant == 2 ? 2 : (g_movegen_variant == 4 ? 4 : 1);
        const int n_mg_groups = mask5 ? (n + WPB * G - 1) / (WPB * G) : 0;
        dim3 grid((unsigned)(n_mg_groups + n_enc_groups));
        if (G == 1) hipLaunchKernelGGL((k_wave_rules<NBE, 1>), grid, dim3(256), 0, s, hb, vb, meta, n, terminal, mask5, planes, n_mg_groups, g_detour_wave);
        else if (G == 4) hipLaunchKernelGGL((k_wave_rules<NBE, 4>), grid, dim3(256), 0, s, hb, vb, meta, n, terminal, mask5, planes, n_mg_groups, g_detour_wave);
        else hipLaunchKernelGGL((k_wave_rules<NBE, 2>), grid, dim3(256), 0, s, hb, vb, meta, n, terminal, mask5, planes, n_mg_groups, g_detour_wave);
        return hipGetLastError();
    }
    if (g_movegen_variant == 1) {  // the first kernel of this repo, kept for A/B runs
        if (mask5 && planes)
            hipLaunchKernelGGL((k_movegen_encode<true, true>), wave_grid(n), dim3(TPB), 0, s, hb, vb, meta, n, mask5, planes, terminal);
        else if (mask5)
            hipLaunchKernelGGL((k_movegen_encode<true, false>), wave_grid(n), dim3(TPB), 0, s, hb, vb, meta, n, mask5, planes, terminal);
        else
            hipLaunchKernelGGL((k_movegen_encode<false, true>), wave_grid(n), dim3(TPB), 0, s, hb, vb, meta, n, mask5, planes, terminal);
        return hipGetLastError();
    }
    PoolBoard* recs = reinterpret_cast<PoolBoard*>(scratch);
    PathTab* tabs = reinterpret_cast<PathTab*>(recs + n);
    // encoder tiles are split over the two launches: 70 % ride beside the path search (a
    // latency-bound dependent chain of ~31 us that leaves issue slots and the memory pipe idle),
    // the rest beside the mask groups (~22 us alone).  Sweep at 32,768 boards, S-mid: 35 % 80 us,
    // 50 % 80, 60 % 77, 70 % 75, 80 % 78, 100 % 82.
    const int enc_total = planes ? (n + NBE - 1) / NBE : 0;
    const int enc_a = mask5 ? (enc_total * g_enc_split_pct) / 100 : 0;
    if (mask5) {
        const int n_path_groups = (2 * n + 255) / 256;
        hipLaunchKernelGGL((k_pool_paths_enc<NBE>), dim3((unsigned)(n_path_groups + enc_a)), dim3(256), 0, s, hb, vb, meta, n,
                           terminal, recs, tabs, n_path_groups, planes, g_detour_pooled);
    }
    int nbt = g_movegen_variant >= 8 ? g_movegen_variant : (n >= 16384 ? 24 : (n >= 8192 ? 16 : 8));
    const int enc_b = enc_total - enc_a;
    if (nbt >= 32) launch_masks_enc<32>(recs, tabs, n, mask5, hb, vb, meta, terminal, planes, enc_a, enc_b, s);
    else if (nbt >= 24) launch_masks_enc<24>(recs, tabs, n, mask5, hb, vb, meta, terminal, planes, enc_a, enc_b, s);
    else if (nbt >= 16) launch_masks_enc<16>(recs, tabs, n, mask5, hb, vb, meta, terminal, planes, enc_a, enc_b, s);
    else if (nbt >= 12) launch_masks_enc<12>(recs, tabs, n, mask5, hb, vb, meta, terminal, planes, enc_a, enc_b, s);
    else launch_masks_enc<8>(recs, tabs, n, mask5, hb, vb, meta, terminal, planes, enc_a, enc_b, s);
    return hipGetLastError();
}
hipError_t step(uint64_t* hb, uint64_t* vb, uint64_t* meta, const uint8_t* action, int n, uint8_t* done, uint8_t* winner,
                hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_step, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, hb, vb, meta, action, n, done, winner);
    return hipGetLastError();
}
hipError_t select(const EngineDev& E, hipStream_t s) {
    hipLaunchKernelGGL(k_select, wave_grid(E.n_boards), dim3(TPB), 0, s, E);
    return hipGetLastError();
}
hipError_t expand_backup(const EngineDev& E, const float* p, const float* v, hipStream_t s) {
    hipLaunchKernelGGL(k_expand_backup, wave_grid(E.n_boards), dim3(TPB), 0, s, E, p, v);
    return hipGetLastError();
}
hipError_t root_pi(const EngineDev& E, double* pi, int32_t* visits, hipStream_t s) {
    hipLaunchKernelGGL(k_root_pi, wave_grid(E.n_boards), dim3(TPB), 0, s, E, pi, visits);
    return hipGetLastError();
}
hipError_t root_children(const EngineDev& E, int32_t* visits, double* q, float* prior, int32_t* root_visits, hipStream_t s) {
    hipLaunchKernelGGL(k_root_children, wave_grid(E.n_boards), dim3(TPB), 0, s, E, visits, q, prior, root_visits);
    return hipGetLastError();
}
hipError_t update_with_move(const EngineDev& E, const uint8_t* moves, hipStream_t s) {
    hipLaunchKernelGGL(k_update_with_move, wave_grid(E.n_boards), dim3(TPB), 0, s, E, moves);
    return hipGetLastError();
}
hipError_t finish_move(const EngineDev& E, const uint8_t* forced, float* pi_out, uint8_t* move_out, hipStream_t s) {
    hipLaunchKernelGGL(k_finish_move, wave_grid(E.n_boards), dim3(TPB), 0, s, E, forced, pi_out, move_out);
    return hipGetLastError();
}
hipError_t reset(const EngineDev& E, int reset_boards, hipStream_t s) {
    hipLaunchKernelGGL(k_reset, dim3((unsigned)((E.n_boards + 255) / 256)), dim3(256), 0, s, E, reset_boards);
    return hipGetLastError();
}
hipError_t harvest(const EngineDev& E, uint64_t* t_hb, uint64_t* t_vb, uint64_t* t_meta, float* t_pi, float* t_z,
                   int32_t* t_game, long long cap, hipStream_t s) {
    hipLaunchKernelGGL(k_harvest_scan, dim3(1), dim3(1024), 0, s, E);
    hipLaunchKernelGGL(k_harvest_copy, wave_grid(E.n_boards), dim3(TPB), 0, s, E, t_hb, t_vb, t_meta, t_pi, t_z, t_game, cap);
    return hipGetLastError();
}
void set_movegen_variant(int v) {
    if (v >= 300 && v < 309) {  // A/B knob: detour_mode of the pooled (v % 3) and the wave-per-board kernel (v / 3)
        g_detour_pooled = (v - 300) % 3;
        g_detour_wave = (v - 300) / 3;
        return;
    }
    if (v >= 100 && v <= 200) {  // A/B knob: share of the encoder tiles that ride beside the path search
        g_enc_split_pct = v - 100;
        return;
    }
    g_movegen_variant = v;
}
hipError_t sqrt_table(double* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_sqrt_table, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n);
    return hipGetLastError();
}

}  // namespace qzl
