// qz_stress_kernels.hip -- TEST-ONLY: synthetic neighbours for the stream-safety diagnostics
// (benchmarks/diag_head.py): kernels that exercise ONE hardware resource each, to find out what a
// co-resident workgroup must be doing for another kernel's results to change.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// mode bit 0: MFMA loop, bit 1: LDS traffic (ds_read_b128 / ds_write_b128 over `lds_bytes`), bit 2: global loads
template <int LDS_BYTES>
__global__ __launch_bounds__(256) void k_stress(float* out, const float* in, int iters, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    half8* v = reinterpret_cast<half8*>(lds);
    const int tid = threadIdx.x;
    const int nvec = LDS_BYTES / 16;
    half8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = (_Float16)(0.001f * (tid + j)); b[j] = (_Float16)(0.002f * (tid - j)); }
    if (mode & 2) for (int i = tid; i < nvec; i += 256) v[i] = a;
    __syncthreads();
    floatx16 acc;
    for (int j = 0; j < 16; j++) acc[j] = 0.f;
    float g = 0.f;
    for (int it = 0; it < iters; it++) {
        if (mode & 2) { a = v[(tid * 9 + it * 7) % nvec]; }
        if (mode & 4) g += in[(size_t)blockIdx.x * 256 + tid + (size_t)(it & 15) * 65536];
        if (mode & 1) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc, 0, 0, 0);
        }
        if ((mode & 2) && (it & 7) == 7) { __syncthreads(); v[(tid + it) % nvec] = b; __syncthreads(); }
    }
    float s = g;
    for (int j = 0; j < 16; j++) s += acc[j];
    out[(size_t)blockIdx.x * 256 + tid] = s + (float)a[0];
}

extern "C" int qzt_stress(float* out, const float* in, int blocks, int iters, int mode, int lds_kb, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (lds_kb >= 48) hipLaunchKernelGGL((k_stress<49280>), dim3(blocks), dim3(256), 0, s, out, in, iters, mode);
    else if (lds_kb >= 16) hipLaunchKernelGGL((k_stress<16384>), dim3(blocks), dim3(256), 0, s, out, in, iters, mode);
    else hipLaunchKernelGGL((k_stress<1024>), dim3(blocks), dim3(256), 0, s, out, in, iters, mode);
    return (int)hipGetLastError();
}

// Victim kernels with a known answer: each thread iterates a deterministic recurrence with ONE kind of
// instruction; the host compares against a run without neighbours.
typedef float f2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(192) void k_victim(float* out, int iters, int kind) {
    const int tid = blockIdx.x * 192 + threadIdx.x;
    float x = 0.001f * (float)(tid % 977) + 0.5f, y = 0.25f;
    f2_t a = {x, y}, b = {1.0001f, 0.9999f}, c = {0.0003f, -0.0002f};
    __shared__ float lds[192 * 4];
    for (int it = 0; it < iters; it++) {
        if (kind == 0) {  // v_pk_fma_f32
            a = __builtin_elementwise_fma(a, b, c);
        } else if (kind == 1) {  // v_fma_f32
            a[0] = __builtin_fmaf(a[0], b[0], c[0]);
            a[1] = __builtin_fmaf(a[1], b[1], c[1]);
        } else if (kind == 2) {  // transcendental
            a[0] = __expf(-a[0] * 0.01f) + 0.3f;
            a[1] = tanhf(a[1]) + 0.2f;
        } else if (kind == 3) {  // LDS round trip
            lds[threadIdx.x * 4] = a[0];
            __syncthreads();
            a[0] = lds[((threadIdx.x + 1) % 192) * 4] * 0.999f + 0.001f;
            __syncthreads();
        } else {  // cross-lane (ds_bpermute)
            a[0] = __shfl_xor(a[0], 1 + (it & 31)) * 0.999f + a[1] * 0.001f;
        }
    }
    out[tid] = a[0] + a[1];
}
extern "C" int qzt_victim(float* out, int blocks, int iters, int kind, void* stream) {
    hipLaunchKernelGGL(k_victim, dim3(blocks), dim3(192), 0, (hipStream_t)stream, out, iters, kind);
    return (int)hipGetLastError();
}

