// Micro-benchmark: how fast can ONE CU pull L2-resident data (a) into LDS by LDS-DMA (global_load_lds_dwordx4, the path every
// f16x2 GEMM of this library stages its operands through) and (b) into registers by plain global_load_dwordx4?
// Each workgroup streams `iters` x (waves x 16 KB) from a small buffer that every workgroup shares (L2 hits after the first
// touch), one workgroup per CU. Prints GB/s per CU and bytes per clock at the measured time (clock from hipDeviceProp).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ldsdma_rate.hip -o gpurun_out/ldsdma_rate && gpurun_out/ldsdma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void* g, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}

// MODE 0: LDS-DMA, 16 pieces (16 KB) per wave in flight, vmcnt(0) between batches; MODE 1: the same bytes with plain loads into
// registers (16 x dwordx4 per lane in flight); MODE 2: LDS-DMA with rows of 64 B at a 1-KB stride (the GEMMs' access pattern:
// 16 rows x 64 B per piece) instead of one contiguous KB per piece
template <int MODE>
__global__ __launch_bounds__(1024) void stream_kernel(const char* __restrict__ src, size_t span, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem + wave * 16384);
    float acc = 0.f;
    size_t off = ((size_t)blockIdx.x * 4096 + (size_t)wave * 16384) % span;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            uint4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const uint4*>(src + (off + i * 1024 + lane * 16) % span);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += __uint_as_float(v[i].x ^ v[i].w);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const size_t o = MODE == 0 ? (off + i * 1024 + lane * 16) % span
                                           : (off + (size_t)(i * 16 + (lane >> 2)) * 1024 + (lane & 3) * 16) % span;
                glds16(src + o, lds0 + i * 1024);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        off = (off + (size_t)nw * 16384) % span;
    }
    if (MODE != 1) acc = reinterpret_cast<float*>(smem)[threadIdx.x];
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE>
static void run(const char* name, const char* src, size_t span, int waves, int iters, int ncu, float* sink, double clk_ghz) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int lds = MODE == 1 ? 0 : waves * 16384;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    hipLaunchKernelGGL(stream_kernel<MODE>, dim3(ncu), dim3(waves * 64), lds, 0, src, span, 4, sink);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(stream_kernel<MODE>, dim3(ncu), dim3(waves * 64), lds, 0, src, span, iters, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    const double bytes_per_cu = (double)iters * waves * 16384;
    const double gbs = bytes_per_cu / (best * 1e-3) / 1e9;
    printf("{\"path\": \"%s\", \"waves_per_cu\": %d, \"span_mb\": %.1f, \"GBps_per_cu\": %.1f, \"TBps_chip\": %.2f, \"bytes_per_clk_per_cu_at_%.1fGHz\": %.1f}\n",
           name, waves, span / 1048576.0, gbs, gbs * ncu / 1e3, clk_ghz, gbs / clk_ghz);
}

int main() {
    hipDeviceProp_t pr;
    CHECK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount;
    const double clk = pr.clockRate / 1e6;
    char* src; float* sink;
    const size_t cap = 512u << 20;
    CHECK(hipMalloc(&src, cap + (1 << 20))); CHECK(hipMemset(src, 1, cap + (1 << 20))); CHECK(hipMalloc(&sink, 64));
    const size_t spans[3] = {2u << 20, 64u << 20, 512u << 20};        // L2-resident, Infinity-Cache-resident, HBM
    for (size_t span : spans) {
        for (int waves : {1, 2, 4, 8}) {
            const int iters = 2000 / waves;
            run<0>("lds_dma_contiguous_1KB_pieces", src, span, waves, iters, ncu, sink, clk);
            run<2>("lds_dma_16rows_x_64B_pieces", src, span, waves, iters, ncu, sink, clk);
            run<1>("global_load_dwordx4_to_vgpr", src, span, waves, iters, ncu, sink, clk);
        }
    }
    return 0;
}
