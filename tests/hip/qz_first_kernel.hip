// qz_first_kernel.hip -- TEST-ONLY: the first wave-per-board actions() + state() kernel of this
// repo (round 1), kept out of the product library as an independent HIP implementation for the
// parity tests (tests/test_gpu_rules.py compares it, the shipped kernels and the CPU oracle).
// Built by `make -C tests/hip` into tests/hip/libqz_testkernels.so; nothing under
// alphazero_quoridor_amd/ loads it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../alphazero_quoridor_amd/csrc/qz_rules.h"

#define QZ_PLANES_N 2106
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
}  // namespace

// mask5 / planes may be NULL (not both); terminal may be NULL.  Returns a hipError_t.
extern "C" int qzt_movegen_encode_v1(const uint64_t* hb, const uint64_t* vb, const uint64_t* meta, int n, uint32_t* mask5,
                                     float* planes, const uint8_t* terminal, void* stream) {
    if (n <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)((n + WPB - 1) / WPB));
    if (mask5 && planes)
        hipLaunchKernelGGL((k_movegen_encode<true, true>), grid, dim3(TPB), 0, s, hb, vb, meta, n, mask5, planes, terminal);
    else if (mask5)
        hipLaunchKernelGGL((k_movegen_encode<true, false>), grid, dim3(TPB), 0, s, hb, vb, meta, n, mask5, planes, terminal);
    else
        hipLaunchKernelGGL((k_movegen_encode<false, true>), grid, dim3(TPB), 0, s, hb, vb, meta, n, mask5, planes, terminal);
    return (int)hipGetLastError();
}
