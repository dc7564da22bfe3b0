// Does an LDS-DMA write (global_load_lds_dwordx4, M0-addressed) of one workgroup ever land in the LDS of ANOTHER workgroup on the CU?
// Stream A: `dma_kernel` -- 256 threads, DYN bytes of dynamic LDS, every wave DMAs a constant pattern over the workgroup's whole
// allocation, over and over.  Stream B: `victim_kernel` -- 256 threads, 14 KB of static LDS filled with per-word sentinels, re-read in a
// loop; every word that changed is reported (offset, value found) and repaired.  Both run concurrently (two HIP streams, one process).
// build: hipcc --offload-arch=gfx950 -O2 -o tools/micro/lds_dma_canary.bin tools/micro/lds_dma_canary.hip
// usage: lds_dma_canary.bin SECONDS DYN_BYTES [dynamic|static|read128|mfma|read128_mfma] [xchg]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

template <bool STATIC64>
__global__ __launch_bounds__(256, 2) void dma_kernel(const uint4* __restrict__ src, int dyn_bytes, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
    __shared__ __attribute__((aligned(16))) unsigned char ssm[STATIC64 ? 65536 : 16];
    unsigned char* smem = STATIC64 ? ssm : dsm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (unsigned)wave * 1024);
    const int pieces = dyn_bytes / 4096;                    // per wave: pieces of 1 KB, 4 KB apart
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < pieces; ++i) glds16(src + lane, lds0 + (unsigned)i * 4096);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (reinterpret_cast<unsigned*>(smem)[threadIdx.x] == 0x12345678u) sink[0] = 1;
}

// aggressors without DMA: ds_read_b128 sweeps over the workgroup's LDS, optionally feeding f16 MFMAs (the inner loop of the 128 x 128
// f16x2 GEMM block minus its operand staging)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <bool MFMA, bool READ>
__global__ __launch_bounds__(256, 2) void read_kernel(int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
    for (int j = threadIdx.x; j < 65536 / 16; j += 256) reinterpret_cast<uint4*>(smem)[j] = make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);
    __syncthreads();
    floatx16 acc0 = {0}, acc1 = {0};
    uint4 x = make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u), y = x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 4
        for (int j = 0; j < 16; ++j) {
            if (READ) {
                x = reinterpret_cast<const uint4*>(smem)[(threadIdx.x + 256 * j) & 4095];
                y = reinterpret_cast<const uint4*>(smem)[(threadIdx.x * 5 + 256 * j + 64) & 4095];
            }
            if (MFMA) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, y), __builtin_bit_cast(f16x8, x), acc1, 0, 0, 0);
            } else {
                acc0[0] += __builtin_bit_cast(float, x.x ^ y.w);
            }
        }
    }
    if (acc0[0] + acc1[3] == 123.456f) sink[0] = acc0[1];
}

// a victim that does what fbank_kernel's FFT exchange does: every wave writes float2 per lane into its private LDS region with one
// lane map and reads it back with another (ds_write_b64 / ds_read_b64, no barrier but the wave's own program order), and checks it
__global__ __launch_bounds__(256, 4) void xchg_victim_kernel(unsigned* report, int iters) {
    __shared__ float2 zs[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float2* z = zs[wave];
    for (int it = 0; it < iters; ++it) {
        const float base = (float)((blockIdx.x * 131 + it * 7 + wave) & 1023);
        bool bad = false;
        float2 got = make_float2(0.f, 0.f);
        int where = 0;
        // the four exchanges of fbank_kernel's 256-point FFT, in its order: write map / read map pairs
#pragma unroll
        for (int stage = 0; stage < 4; ++stage) {
            const float b2 = base + 1024.f * (float)stage;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int wi = stage == 0 ? 64 * r + lane
                             : stage == 1 ? 64 * (lane >> 4) + 16 * r + (lane & 15)
                             : stage == 2 ? 64 * (lane >> 4) + 16 * ((lane >> 2) & 3) + 4 * r + (lane & 3)
                                          : 64 * r + 16 * (lane & 3) + 4 * ((lane >> 2) & 3) + (lane >> 4);
                z[wi] = make_float2(b2 + (float)wi, -(b2 + (float)wi));
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ri = stage == 0 ? 64 * (lane >> 4) + 16 * r + (lane & 15)
                             : stage == 1 ? 64 * (lane >> 4) + 16 * ((lane >> 2) & 3) + 4 * r + (lane & 3)
                             : stage == 2 ? 4 * lane + r
                                          : (r & 1 ? (256 - (lane + 64 * (r >> 1))) & 255 : lane + 64 * (r >> 1));
                const float2 v = z[ri];
                if (v.x != b2 + (float)ri || v.y != -(b2 + (float)ri)) { bad = true; got = v; where = ri + 1000 * stage; }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (bad) {
            const unsigned k = atomicAdd(report, 1u);
            if (k < 64) { report[4 + 4 * k] = where; report[5 + 4 * k] = __builtin_bit_cast(unsigned, got.x); report[6 + 4 * k] = blockIdx.x; report[7 + 4 * k] = it; }
        }
    }
}

constexpr int VW = 3584;                                     // 14 KB of words
__global__ __launch_bounds__(256, 4) void victim_kernel(unsigned* report, int iters) {
    __shared__ unsigned buf[VW];
    const unsigned tag = 0x5A000000u | ((blockIdx.x & 0xFFF) << 12);
    for (int j = threadIdx.x; j < VW; j += 256) buf[j] = tag | j;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        for (int j = threadIdx.x; j < VW; j += 256) {
            const unsigned v = buf[j];
            if (v != (tag | j)) {
                const unsigned k = atomicAdd(report, 1u);
                if (k < 64) { report[4 + 4 * k] = j; report[5 + 4 * k] = v; report[6 + 4 * k] = blockIdx.x; report[7 + 4 * k] = it; }
                buf[j] = tag | j;
            }
        }
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 5.0;
    const int dyn = argc > 2 ? atoi(argv[2]) : 65536;
    const bool stat = argc > 3 && !strcmp(argv[3], "static");
    const char* mode = argc > 3 ? argv[3] : "dynamic";      // dynamic | static (DMA aggressor) | read128 | mfma | read128_mfma
    const bool xchg = argc > 4 && !strcmp(argv[4], "xchg"); // victim: exchange pattern instead of the read-only sentinel words
    uint4* src; unsigned *sink, *report;
    CK(hipMalloc(&src, 4096)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&report, 4 * (4 + 4 * 64)));
    std::vector<unsigned> pat(1024, 0xD0D0D0D0u);
    CK(hipMemcpy(src, pat.data(), 4096, hipMemcpyHostToDevice));
    CK(hipMemset(report, 0, 4 * (4 + 4 * 64)));
    hipStream_t a, b;
    CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b));
    if (!stat) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
    const auto t0 = std::chrono::steady_clock::now();
    long la = 0, lb = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int r = 0; r < 8; ++r) {
            if (!strcmp(mode, "read128")) hipLaunchKernelGGL((read_kernel<false, true>), dim3(128), dim3(256), 0, a, 200, reinterpret_cast<float*>(sink));
            else if (!strcmp(mode, "mfma")) hipLaunchKernelGGL((read_kernel<true, false>), dim3(128), dim3(256), 0, a, 200, reinterpret_cast<float*>(sink));
            else if (!strcmp(mode, "read128_mfma")) hipLaunchKernelGGL((read_kernel<true, true>), dim3(128), dim3(256), 0, a, 200, reinterpret_cast<float*>(sink));
            else if (stat) hipLaunchKernelGGL(dma_kernel<true>, dim3(128), dim3(256), 0, a, src, 65536, 40, sink);
            else hipLaunchKernelGGL(dma_kernel<false>, dim3(128), dim3(256), dyn, a, src, dyn, 40, sink);
            if (xchg) hipLaunchKernelGGL(xchg_victim_kernel, dim3(512), dim3(256), 0, b, report, 400);
            else hipLaunchKernelGGL(victim_kernel, dim3(512), dim3(256), 0, b, report, 200);
            ++la; ++lb;
        }
        CK(hipStreamSynchronize(a)); CK(hipStreamSynchronize(b));
    }
    std::vector<unsigned> h(4 + 4 * 64);
    CK(hipMemcpy(h.data(), report, 4 * h.size(), hipMemcpyDeviceToHost));
    printf("{\"aggressor\": \"%s\", \"victim\": \"%s\", \"dma_lds_bytes\": %d, \"dma_launches\": %ld, \"victim_launches\": %ld, \"victim_words_changed\": %u, \"first\": [",
           mode, xchg ? "exchange (ds_write_b64 / ds_read_b64 through the wave's region)" : "read-only sentinel words", stat ? 65536 : dyn, la, lb, h[0]);
    for (unsigned k = 0; k < h[0] && k < 12; ++k)
        printf("%s{\"word\": %u, \"found\": \"0x%08X\", \"block\": %u, \"iter\": %u}", k ? ", " : "", h[4 + 4 * k], h[5 + 4 * k], h[6 + 4 * k], h[7 + 4 * k]);
    printf("]}\n");
    return 0;
}
