// Aggressor variants for the packed-fp32 hunt (round 5; tools/repro_pk_aggressors.py): small kernels that can share a CU with
// fbank_kernel, launched on a caller-provided stream. KIND bits: 1 = f16 MFMAs, 2 = ds_read_b128 operand fetches, 4 = bf16 MFMAs
// instead of f16, 8 = a barrier every 16 steps, 16 = packed-fp32 VALU work (v_pk_fma_f32) instead of MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/pk_aggr.hip -o tools/micro/pk_aggr.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

template <int KIND, int LDS_KB>
__global__ __launch_bounds__(256, 2) void aggr_kernel(const uint4* __restrict__ src, int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_KB * 1024];
    constexpr int NV = LDS_KB * 64;                               // uint4 slots
    for (int j = threadIdx.x; j < NV; j += 256) reinterpret_cast<uint4*>(smem)[j] = src[(j + blockIdx.x * 97) & 4095];
    __syncthreads();
    floatx16 acc[4] = {{0}, {0}, {0}, {0}};
    uint4 a[2], b[2];
    a[0] = a[1] = src[threadIdx.x & 4095]; b[0] = b[1] = src[(threadIdx.x * 5 + 64) & 4095];
    floatx2 pk = {1.0f, 0.5f};
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND & 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                a[u] = reinterpret_cast<const uint4*>(smem)[(threadIdx.x + 256 * (2 * it + u)) & (NV - 1)];
                b[u] = reinterpret_cast<const uint4*>(smem)[(threadIdx.x * 5 + 64 + 256 * (2 * it + u)) & (NV - 1)];
            }
        }
        if constexpr (KIND & 1) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    if constexpr (KIND & 4)
                        acc[2 * u + w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[u]), __builtin_bit_cast(bf16x8, b[w]), acc[2 * u + w], 0, 0, 0);
                    else
                        acc[2 * u + w] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[u]), __builtin_bit_cast(f16x8, b[w]), acc[2 * u + w], 0, 0, 0);
                }
        }
        if constexpr (KIND & 16) {
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pk) : "v"(pk));
        }
        if constexpr (!(KIND & 2)) { asm volatile("" : "+v"(a[0].x), "+v"(b[0].x)); }
        if constexpr (KIND & 8) { if ((it & 15) == 15) __syncthreads(); }
    }
    float t = pk[0] + pk[1] + __builtin_bit_cast(float, a[0].x ^ b[1].y);
    for (int k = 0; k < 4; ++k) t += acc[k][0] + acc[k][7];
    if (t == 123.456f) sink[0] = t;
}

template <int KIND, int LDS_KB> static void go(hipStream_t s, const uint4* src, int blocks, int iters, float* sink) {
    hipLaunchKernelGGL((aggr_kernel<KIND, LDS_KB>), dim3(blocks), dim3(256), 0, s, src, iters, sink);
}
extern "C" int pk_aggr_launch(void* stream, int kind, int lds_kb, const void* src, int blocks, int iters, float* sink) {
    hipStream_t s = (hipStream_t)stream;
    const uint4* p = (const uint4*)src;
#define CASE(K, L) if (kind == K && lds_kb == L) { go<K, L>(s, p, blocks, iters, sink); return (int)hipGetLastError(); }
    CASE(11, 64) CASE(9, 64) CASE(10, 64) CASE(3, 64) CASE(1, 64) CASE(15, 64) CASE(11, 16) CASE(9, 16) CASE(26, 64) CASE(24, 16)
#undef CASE
    return -1;
}
