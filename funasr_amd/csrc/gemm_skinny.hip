// Exact-fp32 GEMM for SMALL M (the streaming step: M = streams x 15 window rows, or x <= 20 token rows):
// C = epilogue(A[M,K] * W[N,K]^T), same contract as gemm_f32.hip.
//
// With M << 128 the 128 x 128 tile kernel burns a full tile of MFMA work per 16 useful rows and exposes one long
// serial K loop on a handful of CUs (27-110 us per GEMM regardless of M). Here the problem is treated as what it is:
// a weight-streaming pass. v_mfma_f32_16x16x4_f32 (exact f32, 16-row tiles); one workgroup = 16*RM rows x 16 (or 32) columns,
// its 16 waves split K sixteen ways (intra-workgroup split-K): no LDS staging and no barriers in the main loop, every
// lane fetches its own operands as 16-B loads (A: row l&15, 4 consecutive k; W: column l&15, the same 4 k -- the MFMA
// k index is only a pairing, so the four floats of a load feed four MFMAs). The sixteen partial tiles meet in LDS and are
// summed in a FIXED order (wave 0 + 1 + ... + 15) by the epilogue, which also applies bias / ReLU / residuals and writes
// 64-B row segments: one launch per GEMM (the streaming step is launch-latency bound), deterministic, and a row's
// fma chain depends on K only -- never on how many rows or streams are in the batch (hipGraph replay == eager, a
// stream's result is bitwise independent of its neighbours).
#include "common.h"
#include <type_traits>

namespace pf {

namespace {

constexpr int SK_KW = 16;    // waves per workgroup = K slices (a constant: a row's summation order must not depend on M)

// RM = 16-row MFMA tiles per workgroup in M (1, 2 or 4), CN = 16-column tiles per workgroup (1: most workgroups, the
// latency regime of a few rows; 2: each A load feeds two column tiles, the many-stream regime). A row's own fma chain is
// the same for every RM / CN, so the choice changes speed only, never a bit of the result (tested).
// Measured and dropped: four waves with four slices each in four accumulator sets (a quarter of the LDS partials; same
// summation tree): S = 8 streams 8.2 vs 7.0 ms per step, S = 64 12.0 vs 11.0 -- the sixteen-wave form keeps more loads in flight.
template <int RM, int CN>
__global__ __launch_bounds__(SK_KW * 64) void gemm_skinny_kernel(GemmArgs p) {
    constexpr int LD = CN * 16 + 1;                                        // padded row of the LDS partial tiles
    extern __shared__ __attribute__((aligned(16))) float red[];          // [SK_KW][RM * 16][LD]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * (16 * RM);
    const int n0 = blockIdx.x * (16 * CN);
    // this wave's slice of K (in 16-wide steps): with K = 512 two steps, i.e. ONE round of loads in flight per wave -- the
    // streaming step is a chain of dependent launches, so a GEMM's time is its longest chain of memory latencies
    const int steps = p.K >> 4;
    const int spw = (steps + SK_KW - 1) / SK_KW;
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
    const float* wp[CN];
#pragma unroll
    for (int j = 0; j < CN; ++j) {
        int col = n0 + j * 16 + i16;
        col = col < p.N ? col : p.N - 1;
        wp[j] = p.W + (size_t)col * p.ldw + kq * 4;
    }
    floatx4 acc[RM][CN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < CN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    // U consecutive 16-wide k steps: all their loads first, then their MFMAs in ascending k (the summation order of the plain
    // loop). Written out because the compiler does not unroll a runtime-trip-count loop around MFMAs (convergent), and an
    // un-unrolled loop pays one memory latency per step -- eight in a row for a K = 2048 slice.
    auto steps_of = [&](auto U_, int k0) {
        constexpr int U = decltype(U_)::value;
        float4 a[U][RM], b[U][CN];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < RM; ++i) a[u][i] = *reinterpret_cast<const float4*>(ap[i] + k0 + 16 * u);
#pragma unroll
            for (int j = 0; j < CN; ++j) b[u][j] = *reinterpret_cast<const float4*>(wp[j] + k0 + 16 * u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < CN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].x, b[u][j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < CN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].y, b[u][j].y, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < CN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].z, b[u][j].z, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < CN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i].w, b[u][j].w, acc[i][j], 0, 0, 0);
        }
    };
    const int nit = ke > kb ? (ke - kb) >> 4 : 0;
    int it = 0;
    for (; it + 4 <= nit; it += 4) steps_of(std::integral_constant<int, 4>{}, kb + 16 * it);
    if (it + 2 <= nit) { steps_of(std::integral_constant<int, 2>{}, kb + 16 * it); it += 2; }
    if (it < nit) steps_of(std::integral_constant<int, 1>{}, kb + 16 * it);

    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    float* mine = red + (size_t)wave * (RM * 16 * LD);
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < CN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(i * 16 + kq * 4 + r) * LD + j * 16 + i16] = acc[i][j][r];
    __syncthreads();

    // fixed-order sum of the K slices (wave 0 + 1 + ... + 15) + epilogue; consecutive threads -> consecutive columns of a row
    for (int t = threadIdx.x; t < RM * CN * 256; t += SK_KW * 64) {
        const int lr = t / (16 * CN), lc = t % (16 * CN);
        const int row = m0 + lr, cc = n0 + lc;
        if (row < p.M && cc < p.N) {
            float v = red[lr * LD + lc];
#pragma unroll
            for (int w = 1; w < SK_KW; ++w) v += red[(size_t)w * (RM * 16 * LD) + lr * LD + lc];
            if (p.bias) v += p.bias[cc];
            if (p.relu) v = fmaxf(v, 0.f);
            if (p.R1) v = v + p.R1[(size_t)row * p.ldr1 + cc];
            if (p.R2) v = p.R2[(size_t)row * p.ldr2 + cc] + v;
            p.C[(size_t)row * p.ldc + cc] = v;
        }
    }
}

template <int RM, int CN>
int launch_skinny(const GemmArgs& a, hipStream_t stream) {
    constexpr int lds = SK_KW * RM * 16 * (CN * 16 + 1) * (int)sizeof(float);
    static bool configured = false;
    if (!configured) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<RM, CN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured = true;
    }
    dim3 grid(ceil_div(a.N, 16 * CN), ceil_div(a.M, 16 * RM)), block(SK_KW * 64);
    hipLaunchKernelGGL((gemm_skinny_kernel<RM, CN>), grid, block, lds, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

bool gemm_skinny_applicable(const GemmArgs& a) { return a.amax_val == nullptr && a.K % 16 == 0; }

int launch_gemm_skinny(const GemmArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 16 == 0, "gemm_skinny: K must be a multiple of 16");
    PF_REQUIRE(a.lda % 4 == 0 && a.ldw % 4 == 0, "gemm_skinny: row strides must be multiples of 4 floats");
    PF_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0 && a.C, "gemm_skinny: operands must be 16-B aligned");
    if (a.M <= 16) return launch_skinny<1, 1>(a, stream);
    if (a.M <= 32) return launch_skinny<2, 1>(a, stream);
    return launch_skinny<4, 2>(a, stream);
}

}  // namespace pf
