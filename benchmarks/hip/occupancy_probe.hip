// NOT a residency probe (it was written as one, and round 6's first reading of it was wrong -- see residency_census.hip, which counts).
// Every workgroup spins for 1 ms on independent packed FMAs and reads its start time AFTER ~60 vector instructions of set-up; a launch
// of N wavefronts that lasts ~2 ms was read as "N do not fit".  What it measures instead: the SIMD's arbiter issues the OLDEST ready
// wavefront first, so once four to five wavefronts of back-to-back FMAs saturate a SIMD's vector pipe, a younger RESIDENT wavefront
// does not get through its set-up -- does not even read its start time -- until the older ones have left.  The census (start time
// read by the wavefront's first instruction, a live counter, HW_ID) shows all 8,192 wavefronts of <= 64 registers resident within 3 us
// of each other, in the same loop.  Kept for that lesson: a wavefront that is resident is not a wavefront that runs.
//   hipcc --offload-arch=gfx950 -O2 -o occupancy_probe occupancy_probe.hip && ./occupancy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int VG, int LDSB, int TPBK = 64>
__global__ __launch_bounds__(TPBK) void spin(unsigned long long ticks, float* out) {
    __shared__ char lds[LDSB > 0 ? LDSB : 1];
    float acc[VG];
#pragma unroll
    for (int i = 0; i < VG; i++) acc[i] = (float)(threadIdx.x + i);
    if (LDSB > 0) lds[threadIdx.x] = (char)threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < VG; i++) acc[i] = acc[i] * 1.0001f + 0.5f;
    }
    float s = LDSB > 0 ? (float)lds[threadIdx.x] : 0.f;
#pragma unroll
    for (int i = 0; i < VG; i++) s += acc[i];
    if (s == 12345.678f) out[0] = s;
}
template <int VG, int LDSB, int TPBK = 64>
void probe(const char* name, float* out) {
    hipFuncAttributes a;
    hipFuncGetAttributes(&a, (const void*)spin<VG, LDSB, TPBK>);
    printf("%s: %d threads per workgroup, numRegs %d, LDS %zu B; wavefronts in the grid : ms:", name, TPBK, a.numRegs, (size_t)a.sharedSizeBytes);
    for (int n : {2048, 3072, 3328, 3584, 4096, 5120, 6144, 8192, 8448}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((spin<VG, LDSB, TPBK>), dim3(n * 64 / TPBK), dim3(TPBK), 0, 0, 100000ull, out);  // warm
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((spin<VG, LDSB, TPBK>), dim3(n * 64 / TPBK), dim3(TPBK), 0, 0, 100000ull, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf(" %d:%.2f", n, ms);
    }
    printf("\n");
}
int main() {
    float* out;
    hipMalloc(&out, 4);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, regsPerBlock %d\n", p.name, p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock);
    probe<40, 0>("~64 VGPR, no LDS", out);
    probe<40, 4096>("~64 VGPR, 4 KB LDS", out);
    probe<40, 2048>("~64 VGPR, 2 KB LDS", out);
    probe<100, 0>("~128 VGPR, no LDS", out);
    probe<100, 3840>("~128 VGPR, 3.75 KB LDS", out);
    probe<100, 2048>("~128 VGPR, 2 KB LDS", out);
    probe<72, 0>("~96 VGPR, no LDS", out);
    probe<40, 4096, 128>("~64 VGPR, 4 KB LDS per workgroup", out);
    probe<40, 8192, 128>("~64 VGPR, 8 KB LDS per workgroup", out);
    probe<40, 8192, 256>("~64 VGPR, 8 KB LDS per workgroup", out);
    probe<40, 0, 256>("~64 VGPR, no LDS", out);
    probe<100, 4096, 128>("~128 VGPR, 4 KB LDS per workgroup", out);
    probe<100, 4096, 256>("~128 VGPR, 4 KB LDS per workgroup", out);
    probe<20, 0, 64>("~32 VGPR, no LDS", out);
    probe<46, 4096>("~48 VGPR, 4 KB LDS", out);
    probe<52, 4096>("~56 VGPR, 4 KB LDS", out);
    probe<58, 4096>("~60 VGPR, 4 KB LDS", out);
    probe<62, 4096>("~64 VGPR, 4 KB LDS", out);
    probe<66, 4096>("~68 VGPR, 4 KB LDS", out);
    probe<70, 4096>("~72 VGPR, 4 KB LDS", out);
    return 0;
}
