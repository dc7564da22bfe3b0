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

// ---- memory-path probes (bench.py "mem_probe"): what a box's memory system delivers to plain kernels, outside the bench clock.
// MODE 0: streaming read of `n16` 16-byte words (grid-stride, four loads in flight per thread, xor-folded so nothing is dropped);
// MODE 1: streaming write; MODE 2: every workgroup re-reads ONE 1-MiB window (per workgroup pair of the same XCD slot) `reps` times --
// L2-resident after the first pass, the pattern of a GEMM workgroup re-streaming its W panel.
template <int MODE>
__global__ __launch_bounds__(256) void mem_probe_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, int reps, unsigned* sink) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
    uint4 acc = make_uint4(0, 0, 0, 0);
    if constexpr (MODE == 0) {
        size_t i = tid;
        for (; i + 3 * nt < n16; i += 4 * nt) {
            const uint4 a = src[i], b = src[i + nt], c = src[i + 2 * nt], d = src[i + 3 * nt];
            acc.x ^= a.x ^ b.x ^ c.x ^ d.x; acc.y ^= a.y ^ b.y ^ c.y ^ d.y; acc.z ^= a.z ^ b.z ^ c.z ^ d.z; acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
        }
        for (; i < n16; i += nt) { const uint4 a = src[i]; acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w; }
    } else if constexpr (MODE == 1) {
        const uint4 v = make_uint4((unsigned)tid, 1u, 2u, 3u);
        for (size_t i = tid; i < n16; i += nt) dst[i] = v;
    } else {
        // window w = blockIdx.x % 64 (64 windows of 1 MiB = 65536 words): blocks that share a window mostly share an XCD's L2
        const uint4* win = src + (size_t)(blockIdx.x & 63) * 65536;
        for (int r = 0; r < reps; ++r)
            for (int i = threadIdx.x; i < 65536; i += 1024) {
                const uint4 a = win[i], b = win[i + 256], c = win[i + 512], d = win[i + 768];
                acc.x ^= a.x ^ b.x ^ c.x ^ d.x; acc.y ^= a.y ^ b.y ^ c.y ^ d.y; acc.z ^= a.z ^ b.z ^ c.z ^ d.z; acc.w ^= a.w ^ b.w ^ c.w ^ d.w;
            }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u && MODE != 1) *sink = 1u;       // (keeps the loads alive; practically never true)
}

extern "C" {
// mode 0 / 1 / 2 as above; returns the best ms of `launches` launches (< 0: HIP error). bytes moved per launch: modes 0, 1: 16 * n16;
// mode 2: blocks * reps * 1 MiB (from L2 after the first pass)
float mem_probe_run(const void* src, void* dst, size_t n16, int mode, int blocks, int reps, int launches, unsigned* sink_dev) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1.f;
    float best = 1e30f;
    for (int l = 0; l < launches + 1; ++l) {
        (void)hipEventRecord(a, 0);
        if (mode == 0) hipLaunchKernelGGL(mem_probe_kernel<0>, dim3(blocks), dim3(256), 0, 0, (const uint4*)src, (uint4*)dst, n16, reps, sink_dev);
        else if (mode == 1) hipLaunchKernelGGL(mem_probe_kernel<1>, dim3(blocks), dim3(256), 0, 0, (const uint4*)src, (uint4*)dst, n16, reps, sink_dev);
        else hipLaunchKernelGGL(mem_probe_kernel<2>, dim3(blocks), dim3(256), 0, 0, (const uint4*)src, (uint4*)dst, n16, reps, sink_dev);
        (void)hipEventRecord(b, 0);
        if (hipEventSynchronize(b) != hipSuccess) return -2.f;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, a, b);
        if (l > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return best;
}

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
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < launches; ++i) go();
    (void)hipEventRecord(b, 0);
    if (hipEventSynchronize(b) != hipSuccess) return -2.f;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return ms / launches;
}
}
