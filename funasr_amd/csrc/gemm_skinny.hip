// Exact-fp32 GEMM for SMALL M (the streaming step: M = streams x 15 window rows, or x <= 20 token rows):
// C = epilogue(A[M,K] * W[N,K]^T), same contract as gemm_f32.hip.
//
// With M << 128 the 128 x 128 tile kernel burns a full tile of MFMA work per 16 useful rows and exposes one long
// serial K loop on a handful of CUs (27-110 us per GEMM regardless of M). Here the problem is treated as what it is:
// a weight-streaming pass. v_mfma_f32_16x16x4_f32 (exact f32, 16-row tiles), one wave = 16 rows x 32 columns, no LDS
// and no barriers: every lane fetches its own operands as 16-B loads (A: row l&15, 4 consecutive k; W: column l&15,
// the same 4 k -- the MFMA k index is only a pairing, so the four floats of a load feed four MFMAs), weights are read
// exactly once per 16-row tile, and K is split over blockIdx.y so that even a 15-row problem spreads its weight
// stream over ~100+ CUs. Split-K partials are summed in a fixed order by a second tiny kernel that also applies the
// epilogue (deterministic: hipGraph replay == eager launch, bit for bit).
#include "common.h"

namespace pf {

namespace {

constexpr int SK_BN = 128;   // columns per block (4 waves x 32)

// part == nullptr: single K slice, epilogue in-kernel. Otherwise raw partial sums to part[slice][M][N].
// RM = 16-row MFMA tiles per wave (1, 2 or 4): more rows per wave re-use each weight load; a row's own fma chain
// (k order, K slicing) is the same for every RM, so the choice changes speed only, never a bit of the result.
template <int RM>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p, int k_per_slice, float* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.z * (16 * RM);
    const int n0 = blockIdx.x * SK_BN + wave * 32;
    if (n0 >= p.N) return;
    const int kb = blockIdx.y * k_per_slice;
    int ke = kb + k_per_slice;
    ke = ke < p.K ? ke : p.K;

    const float* ap[RM];
#pragma unroll
    for (int i = 0; i < RM; ++i) {
        int arow = m0 + i * 16 + i16;
        arow = arow < p.M ? arow : p.M - 1;
        ap[i] = p.A + (size_t)arow * p.lda + kq * 4;
    }
    const float* wp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int col = n0 + j * 16 + i16;
        col = col < p.N ? col : p.N - 1;
        wp[j] = p.W + (size_t)col * p.ldw + kq * 4;
    }
    floatx4 acc[RM][2];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 2
    for (int k = kb; k < ke; k += 16) {
        float4 a[RM];
#pragma unroll
        for (int i = 0; i < RM; ++i) a[i] = *reinterpret_cast<const float4*>(ap[i] + k);
        const float4 b0 = *reinterpret_cast<const float4*>(wp[0] + k);
        const float4 b1 = *reinterpret_cast<const float4*>(wp[1] + k);
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b0.x, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b1.x, acc[i][1], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b0.y, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b1.y, acc[i][1], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b0.z, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b1.z, acc[i][1], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b0.w, acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b1.w, acc[i][1], 0, 0, 0);
        }
    }

    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + j * 16 + i16;
        if (col >= p.N) continue;
        const float bv = (!part && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < RM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + i * 16 + kq * 4 + r;
                if (row >= p.M) continue;
                if (part) {
                    part[((size_t)blockIdx.y * p.M + row) * p.N + col] = acc[i][j][r];
                } else {
                    float v = acc[i][j][r] + bv;
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (p.R1) v = v + p.R1[(size_t)row * p.ldr1 + col];
                    if (p.R2) v = p.R2[(size_t)row * p.ldr2 + col] + v;
                    p.C[(size_t)row * p.ldc + col] = v;
                }
            }
        }
    }
}

// fixed-order sum of the K slices + epilogue
__global__ __launch_bounds__(256) void gemm_skinny_reduce_kernel(GemmArgs p, const float* __restrict__ part, int slices) {
    const size_t total = (size_t)p.M * p.N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int row = (int)(i / p.N), col = (int)(i % p.N);
        float v = part[i];
        for (int s = 1; s < slices; ++s) v += part[(size_t)s * total + i];
        if (p.bias) v += p.bias[col];
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.R1) v = v + p.R1[(size_t)row * p.ldr1 + col];
        if (p.R2) v = p.R2[(size_t)row * p.ldr2 + col] + v;
        p.C[(size_t)row * p.ldc + col] = v;
    }
}

struct Scratch {
    float* p = nullptr;
    size_t cap = 0;
};
Scratch g_part;   // split-K partials; grows outside of graph capture (first eager call with a shape)

}  // namespace

bool gemm_skinny_applicable(const GemmArgs& a) { return a.amax_val == nullptr && a.K % 16 == 0; }

int launch_gemm_skinny(const GemmArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 16 == 0, "gemm_skinny: K must be a multiple of 16");
    PF_REQUIRE(a.lda % 4 == 0 && a.ldw % 4 == 0, "gemm_skinny: row strides must be multiples of 4 floats");
    PF_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0 && a.C, "gemm_skinny: operands must be 16-B aligned");
    const int RM = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
    const int nN = ceil_div(a.N, SK_BN), nM = ceil_div(a.M, 16 * RM);
    // K slicing is a function of K ONLY (a row's summation order must not depend on how many other rows are in the
    // batch): 128 k per slice -> 4 slices for K = 512, 16 for K = 2048; even a 15-row problem then streams its
    // weights from >= 16-64 workgroups
    const int kps = 128;
    const int slices = ceil_div(a.K, kps);
    dim3 grid(nN, slices, nM), block(256);
    auto launch = [&](float* part) {
        if (RM == 1) hipLaunchKernelGGL(gemm_skinny_kernel<1>, grid, block, 0, stream, a, kps, part);
        else if (RM == 2) hipLaunchKernelGGL(gemm_skinny_kernel<2>, grid, block, 0, stream, a, kps, part);
        else hipLaunchKernelGGL(gemm_skinny_kernel<4>, grid, block, 0, stream, a, kps, part);
    };
    if (slices == 1) {
        launch(nullptr);
        PF_HIP_TRY(hipGetLastError());
        return 0;
    }
    const size_t need = sizeof(float) * (size_t)slices * a.M * a.N;
    if (need > g_part.cap) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &cs);
        PF_REQUIRE(cs == hipStreamCaptureStatusNone, "gemm_skinny: split-K scratch must be sized by an eager call first");
        PF_HIP_TRY(hipDeviceSynchronize());
        if (g_part.p) PF_HIP_TRY(hipFree(g_part.p));
        g_part.p = nullptr; g_part.cap = 0;
        PF_HIP_TRY(hipMalloc((void**)&g_part.p, need + need / 4));
        g_part.cap = need + need / 4;
    }
    launch(g_part.p);
    const size_t total = (size_t)a.M * a.N;
    const int rb = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
    hipLaunchKernelGGL(gemm_skinny_reduce_kernel, dim3(rb), dim3(256), 0, stream, a, (const float*)g_part.p, slices);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pf
