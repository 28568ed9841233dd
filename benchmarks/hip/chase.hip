// Dependent-load latency under the asynchronous loop's occupancy (one wave per board, 16 waves per CU): each wave follows a
// chain of 2-KB nodes (64 lanes x 32 B, like a node's edge records); the next node comes from the data just loaded.
//   scope 0: the chain wanders over the whole buffer (tree pages of a shared pool, scattered)
//   scope 1: every wave stays inside its own contiguous slice of the buffer (a per-board arena)
// Prints ns per dependent step.  Build: hipcc -O3 --offload-arch=gfx950 chase.hip -o chase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void k_fill(uint4* buf, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = mix((uint32_t)i * 2654435761u + 17u);
        buf[i] = make_uint4(h, mix(h), mix(h + 1), mix(h + 2));
    }
}
__global__ __launch_bounds__(256) void k_chase(const uint4* buf, uint32_t nodes, uint32_t per_wave, int scope, int steps, int active, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t cur = scope ? wave * per_wave : mix(wave) % nodes;
    uint32_t acc = 0;
    for (int s = 0; s < steps; s++) {
        uint4 a = make_uint4(0, 0, 0, 0), b = a;
        if (lane < active) {  // 32 B per lane: 2 x uint4, lanes contiguous (active = 8: a narrow node, 256 B; 64: a wide one, 2 KB)
            a = buf[(size_t)cur * 128 + lane * 2];
            b = buf[(size_t)cur * 128 + lane * 2 + 1];
        }
        acc += a.y ^ b.z;
        const uint32_t r = __builtin_amdgcn_readfirstlane(a.x ^ (uint32_t)s * 0x9e3779b9u);
        cur = scope ? wave * per_wave + mix(r) % per_wave : mix(r) % nodes;
    }
    if (lane == 0) out[wave] = acc + cur;
}
int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 4096;
    const int steps = argc > 2 ? atoi(argv[2]) : 2000;
    const int active = argc > 3 ? atoi(argv[3]) : 8;
    uint32_t* out; CK(hipMalloc(&out, waves * 4));
    const size_t sizes_mb[] = {64, 2048, 8192};
    for (size_t mb : sizes_mb) {
        const size_t bytes = mb << 20;
        uint4* buf; CK(hipMalloc(&buf, bytes));
        k_fill<<<4096, 256>>>(buf, bytes / 16);
        CK(hipDeviceSynchronize());
        const uint32_t nodes = (uint32_t)(bytes / 2048), per_wave = nodes / waves;
        for (int scope = 0; scope < 2; scope++) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            k_chase<<<waves / 4, 256>>>(buf, nodes, per_wave, scope, 200, active, out);
            CK(hipEventRecord(e0));
            k_chase<<<waves / 4, 256>>>(buf, nodes, per_wave, scope, steps, active, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("{\"buffer_mb\": %zu, \"scope\": \"%s\", \"waves\": %d, \"lanes_loading\": %d, \"slice_kb\": %u, \"ns_per_dependent_step\": %.1f}\n", mb, scope ? "own slice per wave" : "whole buffer",
                   waves, active, scope ? per_wave * 2 : 0, ms * 1e6 / steps);
        }
        CK(hipFree(buf));
    }
    return 0;
}
