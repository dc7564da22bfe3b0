// Energy roofline of the matrix pipe for THIS instruction mix (measurement infrastructure, round 5; not product code).
// Register-resident v_mfma_f32_32x32x16_f16 (or _bf16) chains in exactly the f16x2 GEMM's product pattern -- 16 accumulator
// tiles, per 16-deep step lo*hi, hi*lo, hi*hi over 4 x 4 operand fragments -- with NO LDS and NO memory traffic in the loop:
// what the part sustains on random operand planes under its power / clock management is the ceiling any f16x2 kernel can
// approach, and the number `roofline.power_limited_peak` quotes (tools/bench_w4.py drives this and samples MHz / W).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/mfma_peak.hip -o tools/micro/mfma_peak.so
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int BF16> __device__ __forceinline__ void mf(floatx16& acc, const f16x8& a, const f16x8& b) {
    if constexpr (BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// planes: [4 kinds (A hi, A lo, W hi, W lo)][4 tiles][64 lanes] x 16 B, the same for every wave (what matters is the bit
// pattern entering the multipliers, not where it came from)
template <int BF16>
__global__ __launch_bounds__(256, 1) void mfma_peak_kernel(const uint4* __restrict__ planes, int iters, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    f16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ah[i] = __builtin_bit_cast(f16x8, planes[(0 * 4 + i) * 64 + lane]);
        al[i] = __builtin_bit_cast(f16x8, planes[(1 * 4 + i) * 64 + lane]);
        bh[i] = __builtin_bit_cast(f16x8, planes[(2 * 4 + i) * 64 + lane]);
        bl[i] = __builtin_bit_cast(f16x8, planes[(3 * 4 + i) * 64 + lane]);
    }
    floatx16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mf<BF16>(acc[i][j], al[i], bh[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mf<BF16>(acc[i][j], ah[i], bl[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mf<BF16>(acc[i][j], ah[i], bh[j]);
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 123.456f) out[0] = t;
}

extern "C" {
// runs `launches` launches of `iters` steps on `blocks` workgroups of 4 waves; returns the mean ms per launch (< 0: HIP error).
// flops per launch = blocks * 4 * iters * 48 * 32768
float mfma_peak_run(const void* planes_dev, int bf16, int blocks, int iters, int launches, float* scratch_dev) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1.f;
    auto go = [&] {
        if (bf16) hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(blocks), dim3(256), 0, 0, (const uint4*)planes_dev, iters, scratch_dev);
        else hipLaunchKernelGGL(mfma_peak_kernel<0>, dim3(blocks), dim3(256), 0, 0, (const uint4*)planes_dev, iters, scratch_dev);
    };
    go();
    hipEventRecord(a, 0);
    for (int i = 0; i < launches; ++i) go();
    hipEventRecord(b, 0);
    if (hipEventSynchronize(b) != hipSuccess) return -2.f;
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    return ms / launches;
}
}
