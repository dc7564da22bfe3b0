// Exact-fp32 GEMM for SMALL M (the streaming step: M = streams x 15 window rows, or x <= 20 token rows):
// C = epilogue(A[M,K] * W[N,K]^T), same contract as gemm_f32.hip.
//
// With M << 128 the 128 x 128 tile kernel burns a full tile of MFMA work per 16 useful rows and exposes one long
// serial K loop on a handful of CUs (27-110 us per GEMM regardless of M). Here the problem is treated as what it is:
// a weight-streaming pass. v_mfma_f32_16x16x4_f32 (exact f32, 16-row tiles); one workgroup = 16*RM rows x 32 columns,
// its 4 waves split K four ways (intra-workgroup split-K): no LDS staging and no barriers in the main loop, every
// lane fetches its own operands as 16-B loads (A: row l&15, 4 consecutive k; W: column l&15, the same 4 k -- the MFMA
// k index is only a pairing, so the four floats of a load feed four MFMAs). The four partial tiles meet in LDS and are
// summed in a FIXED order (wave 0 + 1 + 2 + 3) by the epilogue, which also applies bias / ReLU / residuals and writes
// 128-B row segments: one launch per GEMM (the streaming step is launch-latency bound), deterministic, and a row's
// fma chain depends on K only -- never on how many rows or streams are in the batch (hipGraph replay == eager, a
// stream's result is bitwise independent of its neighbours).
#include "common.h"

namespace pf {

namespace {

constexpr int SK_BN = 32;    // columns per workgroup
constexpr int SK_LD = 33;    // padded row of the LDS partial tiles

// RM = 16-row MFMA tiles per workgroup in M (1, 2 or 4): more rows re-use each weight load; a row's own fma chain is
// the same for every RM, so the choice changes speed only, never a bit of the result.
template <int RM>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs p) {
    __shared__ float red[4][RM * 16][SK_LD];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * (16 * RM);
    const int n0 = blockIdx.x * SK_BN;
    // this wave's quarter of K (in 16-wide steps)
    const int steps = p.K >> 4;
    const int spw = (steps + 3) >> 2;
    const int kb = wave * spw * 16;
    int ke = kb + spw * 16;
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

#pragma unroll 4
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
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][i * 16 + kq * 4 + r][j * 16 + i16] = acc[i][j][r];
    __syncthreads();

    // fixed-order sum of the four K quarters + epilogue; consecutive threads -> consecutive columns of a row
#pragma unroll
    for (int e = 0; e < RM * 2; ++e) {
        const int t = threadIdx.x + 256 * e;
        const int lr = t >> 5, lc = t & 31;
        const int row = m0 + lr, col = n0 + lc;
        if (row >= p.M || col >= p.N) continue;
        float v = ((red[0][lr][lc] + red[1][lr][lc]) + red[2][lr][lc]) + red[3][lr][lc];
        if (p.bias) v += p.bias[col];
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.R1) v = v + p.R1[(size_t)row * p.ldr1 + col];
        if (p.R2) v = p.R2[(size_t)row * p.ldr2 + col] + v;
        p.C[(size_t)row * p.ldc + col] = v;
    }
}

}  // namespace

bool gemm_skinny_applicable(const GemmArgs& a) { return a.amax_val == nullptr && a.K % 16 == 0; }

int launch_gemm_skinny(const GemmArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 16 == 0, "gemm_skinny: K must be a multiple of 16");
    PF_REQUIRE(a.lda % 4 == 0 && a.ldw % 4 == 0, "gemm_skinny: row strides must be multiples of 4 floats");
    PF_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0 && a.C, "gemm_skinny: operands must be 16-B aligned");
    const int RM = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
    dim3 grid(ceil_div(a.N, SK_BN), ceil_div(a.M, 16 * RM)), block(256);
    if (RM == 1) hipLaunchKernelGGL(gemm_skinny_kernel<1>, grid, block, 0, stream, a);
    else if (RM == 2) hipLaunchKernelGGL(gemm_skinny_kernel<2>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(gemm_skinny_kernel<4>, grid, block, 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pf
