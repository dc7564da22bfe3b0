// Stand-alone form of DESIGN 4's observation: does packed-fp32 VALU arithmetic of one kernel change its results while waves of an
// f16-MFMA kernel (v_mfma_f32_32x32x16_f16 fed by ds_read_b128, two 256-thread workgroups per CU -- the inner loop of this library's
// 128 x 128 f16x2 GEMM block) run on the same CUs from a second HIP stream?  Victims: tools/micro/pk/victim.inc, once with packed-fp32
// instructions, once built without them.  build: tools/micro/pk/build.sh   usage: pk_next_to_mfma.bin SECONDS
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
extern "C" __global__ void victim_pk(unsigned* report, int iters, int rounds);
extern "C" __global__ void victim_nopk(unsigned* report, int iters, int rounds);

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void mfma_aggressor(const uint4* __restrict__ src, int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
    for (int j = threadIdx.x; j < 4096; j += 256) reinterpret_cast<uint4*>(smem)[j] = src[(j + blockIdx.x * 97) & 4095];
    __syncthreads();
    floatx16 acc[4] = {{0}, {0}, {0}, {0}};
    for (int it = 0; it < iters; ++it) {
        uint4 a[2], b[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            a[u] = reinterpret_cast<const uint4*>(smem)[(threadIdx.x + 256 * (2 * it + u)) & 4095];
            b[u] = reinterpret_cast<const uint4*>(smem)[(threadIdx.x * 5 + 64 + 256 * (2 * it + u)) & 4095];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int w = 0; w < 2; ++w)
                acc[2 * u + w] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[u]), __builtin_bit_cast(f16x8, b[w]), acc[2 * u + w], 0, 0, 0);
        if ((it & 15) == 15) __syncthreads();
    }
    float t = 0.f;
    for (int k = 0; k < 4; ++k) t += acc[k][0] + acc[k][7];
    if (t == 123.456f) sink[0] = t;
}

// Round 5: the aggressor that disturbs the REAL fbank_kernel in ~90 % of its calls (tools/repro_pk_aggressors.py): the same MFMAs on
// REGISTER operands (no LDS traffic at all) with a workgroup barrier every 16 steps, 16 KB of LDS so that it shares a CU with anything
__global__ __launch_bounds__(256, 2) void mfma_aggressor_regs(const uint4* __restrict__ src, int iters, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[16384];
    for (int j = threadIdx.x; j < 1024; j += 256) reinterpret_cast<uint4*>(smem)[j] = src[(j + blockIdx.x * 97) & 4095];
    __syncthreads();
    floatx16 acc[4] = {{0}, {0}, {0}, {0}};
    uint4 a[2], b[2];
    a[0] = a[1] = src[threadIdx.x & 4095]; b[0] = b[1] = src[(threadIdx.x * 5 + 64) & 4095];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int w = 0; w < 2; ++w)
                acc[2 * u + w] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[u]), __builtin_bit_cast(f16x8, b[w]), acc[2 * u + w], 0, 0, 0);
        asm volatile("" : "+v"(a[0].x), "+v"(b[0].x));
        if ((it & 15) == 15) __syncthreads();
    }
    float t = reinterpret_cast<const float*>(smem)[threadIdx.x];
    for (int k = 0; k < 4; ++k) t += acc[k][0] + acc[k][7];
    if (t == 123.456f) sink[0] = t;
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 4.0;
    uint4* src; float* sink; unsigned* report;
    CK(hipMalloc(&src, 65536)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&report, 64));
    std::vector<unsigned short> h(32768);
    unsigned s = 12345u;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (unsigned short)(0x3000u + ((s >> 16) & 0x0FFFu)); }   // fp16 values in [0.125, 2): busy operands
    CK(hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice));
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    // round 5: is the disturbance local to a CU? streams restricted to CU sets (hipExtStreamCreateWithCUMask; bits 0..127 / 128..255 are
    // disjoint halves with 16 CUs of every XCD each, tools/exp_cumask.py): mode 5 = victim and aggressor on DISJOINT halves, 6 = both on the same half
    hipStream_t lo_a, lo_b, hi_b;
    {
        uint32_t lo[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0}, hi[8] = {0, 0, 0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        CK(hipExtStreamCreateWithCUMask(&lo_a, 8, lo)); CK(hipExtStreamCreateWithCUMask(&lo_b, 8, lo)); CK(hipExtStreamCreateWithCUMask(&hi_b, 8, hi));
    }
    for (int mode = 0; mode < 7; ++mode) {       // 0: packed-fp32 victim beside the aggressor, 1: the same victim alone, 2: no-packed victim beside the aggressor;
                                                 // 3 / 4: packed / no-packed victim beside the register-operand aggressor (round 5)
        CK(hipMemset(report, 0, 64));
        const auto t0 = std::chrono::steady_clock::now();
        long launches = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
            for (int r = 0; r < 8; ++r) {
                if (mode == 0 || mode == 2) hipLaunchKernelGGL(mfma_aggressor, dim3(256), dim3(256), 0, sa, src, 2000, sink);
                if (mode == 3 || mode == 4) hipLaunchKernelGGL(mfma_aggressor_regs, dim3(512), dim3(256), 0, sa, src, 2000, sink);
                if (mode >= 5) {
                    hipLaunchKernelGGL(mfma_aggressor_regs, dim3(512), dim3(256), 0, lo_a, src, 2000, sink);
                    hipLaunchKernelGGL(victim_pk, dim3(1024), dim3(256), 0, mode == 5 ? hi_b : lo_b, report, 64, 8);
                    ++launches;
                    continue;
                }
                if (mode == 2 || mode == 4) hipLaunchKernelGGL(victim_nopk, dim3(1024), dim3(256), 0, sb, report, 64, 8);
                else hipLaunchKernelGGL(victim_pk, dim3(1024), dim3(256), 0, sb, report, 64, 8);
                ++launches;
            }
            CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
            CK(hipStreamSynchronize(lo_a)); CK(hipStreamSynchronize(lo_b)); CK(hipStreamSynchronize(hi_b));
        }
        unsigned bad = 0;
        CK(hipMemcpy(&bad, report, 4, hipMemcpyDeviceToHost));
        printf("{\"victim\": \"%s\", \"aggressor\": \"%s\", \"victim_launches\": %ld, \"double_evaluations\": %.3g, \"disagreements\": %u}\n",
               (mode == 2 || mode == 4) ? "scalar fp32 (built without packed-fp32 instructions)" : "packed fp32",
               mode == 1 ? "none" : (mode == 5 ? "register-operand MFMAs + barriers on the OTHER half of the CUs (disjoint CU masks)" : mode == 6 ? "register-operand MFMAs + barriers on the SAME half of the CUs (equal CU masks)" :
                                     mode >= 3 ? "f16 MFMA 32x32x16 on register operands + a barrier every 16 steps, 16 KB LDS" : "f16 MFMA 32x32x16 + ds_read_b128, 2 workgroups per CU"),
               launches, (double)launches * 1024 * 256 * 64, bad);
    }
    return 0;
}
