// Census of resident wavefronts on an MI355X: how many wavefronts of a given register / LDS footprint are on the chip AT ONCE, where
// they sit (XCC / SE / CU / SIMD from HW_ID), and how late after the launch's first wavefront each one starts.  occupancy_probe.hip
// infers residency from the length of a launch whose wavefronts spin on vector FMAs; this one counts (a live counter + the start /
// end times of every wavefront) and offers three ways of spending the time, because a wavefront that is resident but not yet issued
// its first instruction looks, to a launch-length probe, like one that is not resident:
//   mode 0: spin on independent vector FMAs (occupancy_probe's loop), mode 1: s_sleep between looks at the clock,
//   mode 2: a dependent chain of global loads (the shape of the tree kernels: a wavefront waits for memory most of its time).
//   hipcc --offload-arch=gfx950 -O2 -o residency_census residency_census.hip && ./residency_census
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
struct Rec { unsigned t0, t1, hw, xcc; };
template <int VG, int LDSB, int TPBK, int MODE>
__global__ __launch_bounds__(TPBK) void census(unsigned long long ticks, float* out, Rec* rec, unsigned* live, const unsigned* chain, unsigned chain_mask) {
    __shared__ char lds[LDSB > 0 ? LDSB : 1];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(4)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(20)" : "=s"(xcc));
    const bool first = (threadIdx.x & 63) == 0;
    if (first) {
        const unsigned n = atomicAdd(live, 1u) + 1;
        atomicMax(live + 1, n);
    }
    float acc[VG];
#pragma unroll
    for (int i = 0; i < VG; i++) acc[i] = (float)(threadIdx.x + i);
    if (LDSB > 0) lds[threadIdx.x] = (char)threadIdx.x;
    unsigned p = (blockIdx.x * 2654435761u + threadIdx.x) & chain_mask;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < VG; i++) acc[i] = acc[i] * 1.0001f + 0.5f;
        } else {
#pragma unroll
            for (int i = 0; i < VG; i++) asm volatile("" : "+v"(acc[i]));
            if (MODE == 1) __builtin_amdgcn_s_sleep(64);
            else {
                p = chain[p] & chain_mask;   // one dependent trip per look at the clock
                acc[0] += (float)p;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    float s = LDSB > 0 ? (float)lds[threadIdx.x] : 0.f;
#pragma unroll
    for (int i = 0; i < VG; i++) s += acc[i];
    if (s == 12345.678f) out[0] = s;
    if (first) {
        atomicSub(live, 1u);
        const unsigned w = blockIdx.x * (TPBK / 64) + threadIdx.x / 64;
        rec[w] = Rec{(unsigned)t0, (unsigned)t1, hw, xcc};
    }
}
static float* g_out;
static Rec* g_rec;
static unsigned* g_live;
static unsigned* g_chain;
static const unsigned CHAIN = 1u << 24;   // 64 MB of indices: the trips go to HBM / the far caches
template <int VG, int LDSB, int TPBK, int MODE>
void run(const char* name, std::initializer_list<int> grids) {
    hipFuncAttributes a;
    hipFuncGetAttributes(&a, (const void*)census<VG, LDSB, TPBK, MODE>);
    printf("%s | mode %d (%s), %d threads per workgroup, numRegs %d, LDS %zu B\n", name, MODE, MODE == 0 ? "vector FMAs" : MODE == 1 ? "s_sleep" : "dependent loads", TPBK, a.numRegs, (size_t)a.sharedSizeBytes);
    for (int n : grids) {
        std::vector<Rec> rec(n);
        hipMemset(g_live, 0, 8);
        hipLaunchKernelGGL((census<VG, LDSB, TPBK, MODE>), dim3(256), dim3(TPBK), 0, 0, 1000ull, g_out, g_rec, g_live, g_chain, CHAIN - 1);  // warm
        hipDeviceSynchronize();
        hipMemset(g_live, 0, 8);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((census<VG, LDSB, TPBK, MODE>), dim3(n * 64 / TPBK), dim3(TPBK), 0, 0, 100000ull, g_out, g_rec, g_live, g_chain, CHAIN - 1);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned live[2];
        hipMemcpy(live, g_live, 8, hipMemcpyDeviceToHost);
        hipMemcpy(rec.data(), g_rec, sizeof(Rec) * n, hipMemcpyDeviceToHost);
        unsigned tmin = ~0u;
        for (auto& r : rec) tmin = std::min(tmin, r.t0);
        std::vector<unsigned> starts;
        for (auto& r : rec) starts.push_back(r.t0 - tmin);
        std::sort(starts.begin(), starts.end());
        // the most wavefronts any one SIMD held at once, and how many SIMDs were used (key: xcc, se, sh, cu, simd)
        std::map<unsigned, std::vector<std::pair<unsigned, int>>> ev;
        for (auto& r : rec) {
            const unsigned key = ((r.xcc & 15) << 16) | (r.hw & 0xFFF0);   // HW_ID without the wave slot
            ev[key].push_back({r.t0 - tmin, +1});
            ev[key].push_back({r.t1 - tmin, -1});
        }
        int worst = 0, least = 1 << 30;
        for (auto& kv : ev) {
            auto& v = kv.second;
            std::sort(v.begin(), v.end());
            int cur = 0, mx = 0;
            for (auto& e : v) { cur += e.second; mx = std::max(mx, cur); }
            worst = std::max(worst, mx);
            least = std::min(least, mx);
        }
        printf("   %5d wavefronts: launch %.2f ms, most alive at once %u, SIMDs used %zu (most on one SIMD %d, fewest %d), start after the first wavefront: median %.1f us, 90%% %.1f us, last %.1f us\n",
               n, ms, live[1], ev.size(), worst, least, starts[n / 2] / 100.0, starts[(size_t)(n * 0.9)] / 100.0, starts[n - 1] / 100.0);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
}
int main() {
    hipMalloc(&g_out, 4);
    hipMalloc(&g_rec, sizeof(Rec) * 16384);
    hipMalloc(&g_live, 8);
    hipMalloc(&g_chain, sizeof(unsigned) * CHAIN);
    {
        std::vector<unsigned> c(CHAIN);
        unsigned x = 12345;
        for (unsigned i = 0; i < CHAIN; i++) { x = x * 1664525u + 1013904223u; c[i] = x >> 8; }
        hipMemcpy(g_chain, c.data(), sizeof(unsigned) * CHAIN, hipMemcpyHostToDevice);
    }
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
    run<40, 0, 64, 0>("~48 registers", {4096, 5120, 6144, 8192});
    run<40, 0, 64, 1>("~48 registers", {4096, 5120, 6144, 8192, 9216});
    run<40, 0, 64, 2>("~48 registers", {4096, 5120, 6144, 8192, 9216});
    run<40, 4096, 64, 1>("~48 registers, 4 KB", {4096, 6144, 8192, 9216});
    run<40, 4096, 64, 2>("~48 registers, 4 KB", {4096, 6144, 8192, 9216});
    run<52, 4096, 64, 0>("~56 registers, 4 KB", {3072, 4096, 8192});
    run<52, 4096, 64, 1>("~56 registers, 4 KB", {3072, 4096, 8192, 9216});
    run<52, 4096, 64, 2>("~56 registers, 4 KB", {3072, 4096, 8192, 9216});
    run<58, 4096, 64, 2>("~64 registers, 4 KB", {4096, 8192, 9216});
    run<100, 4096, 64, 0>("~104 registers, 4 KB", {3072, 4096, 5120});
    run<100, 4096, 64, 1>("~104 registers, 4 KB", {3072, 4096, 5120});
    run<100, 4096, 64, 2>("~104 registers, 4 KB", {3072, 4096, 5120});
    run<120, 4096, 64, 2>("~128 registers, 4 KB", {3072, 4096, 5120});
    run<40, 8192, 256, 1>("~48 registers, 8 KB per workgroup", {4096, 8192, 9216});
    run<40, 8192, 256, 0>("~48 registers, 8 KB per workgroup", {4096, 8192});
    return 0;
}
