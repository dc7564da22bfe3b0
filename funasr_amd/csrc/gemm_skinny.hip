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
// LNIN: C = W LayerNorm(A) + bias without a LayerNorm launch and without normalising A (GemmArgs.ln_stats_in; the algebra is at
// the statistics' loads below). The row statistics come from the producing GEMM's per-16-column partial sums (ln_stats_out), so
// no workgroup re-reads a row to recompute them -- round 2 folded the LayerNorm into the fetch WITH a recomputation per column
// workgroup and lost 1 ms per step to it. mean = S / K, var = merged M2 / K (centred block partials merged by Chan's formula since round 5; the stand-alone kernel is two-pass:
// results differ in the last bits, fp32-class either way), and every order of summation depends on K only -- never on M, RM or
// CN -- so a stream's bits stay independent of its neighbours.
// NW = waves per workgroup. 16: one workgroup owns a tile and all sixteen K slices. 4 (GemmArgs.ws_part): FOUR workgroups share
// a tile, workgroup blockIdx.z takes slices 4 z .. 4 z + 3 -- for the long-K, N = 512 projections of a step (w_2: 4 MB of
// weights behind 32 column tiles = 32 CUs at ~25 GB/s each, 10-18 us in the chain) this spreads the weight stream over 128
// CUs. The sixteen slice tiles go to a workspace; the workgroup that arrives last at the tile's counter folds them in the order
// of the one-workgroup form (slice 0 + 1 + .. + 15) and runs the epilogue: the bits do not depend on the form. MEASURED SLOWER (17.9 vs
// 10.4 us for w_2: the device-scope fence of the hand-over -- the four workgroups may sit on different XCDs -- costs more than the
// spread saves) and therefore off by default (stream option "wide_k"); kept with its tests as the record of that experiment.
template <int RM, int CN, int LNIN, int NW>      // LNIN: 0 plain, 1 LayerNorm form with gamma applied in the loop, 2 with gamma a stored by the producer
__device__ __forceinline__ void skinny_body(const GemmArgs& p) {
    constexpr int LD = CN * 16 + 1;                                        // padded row of the LDS partial tiles
    extern __shared__ __attribute__((aligned(16))) float red[];          // [SK_KW][RM * 16][LD]  (NW == 16)
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;                                       // wave of the workgroup
    const int wave = NW == SK_KW ? wv : (int)blockIdx.z * NW + wv;         // K slice
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

    // one 16-wide k step: the four floats of a load feed four MFMAs, ascending k
    auto mfma_step = [&](const float4 (&a)[RM], const float4 (&b)[CN]) {
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < CN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < CN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < CN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < CN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
    };
    const int nit = ke > kb ? (ke - kb) >> 4 : 0;
    int it = 0;

    // LNIN (GemmArgs.ln_stats_in): C = W LayerNorm(a) + bias without touching a: with (mean, rstd) of the row,
    //   W (gamma (a - mean) rstd + beta) = rstd (W (gamma a) - mean c1) + c2,   c1 = W gamma,  c2 = W beta + bias
    // (c1, c2: per-weight constants, launch_ln_consts). So the loop multiplies the A fragments by gamma (or not at all when the
    // producer stored gamma a already: ln_g == nullptr) and the row statistics are needed by the EPILOGUE only: their block
    // partials are fetched first thing and reduced after the last MFMA -- no latency of theirs lies on the path of the
    // operands (the first form applied (a - mean) rstd gamma + beta on the fetch: +1.8 us at K = 512, +5 us at K = 2048, two
    // rounds of loads). Partials: wave w holds rows w, w + NW, .. (<= 2 loads per lane and row: K <= 2048), butterfly 32 .. 1 in
    // an order that depends on K only.
    constexpr int RW = LNIN ? 16 * RM / NW : 1;            // rows per wave
    __shared__ float2 ln_ms[LNIN ? 16 * RM : 1];
    float2 st0[RW], st1[RW];
    if constexpr (LNIN) {
        const int nblk = p.K >> 4;
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            int arow = m0 + wv + NW * i;
            arow = arow < p.M ? arow : p.M - 1;
            const float2* st = reinterpret_cast<const float2*>(p.ln_stats_in) + (size_t)arow * nblk;
            st0[i] = lane < nblk ? st[lane] : make_float2(0.f, 0.f);
            st1[i] = lane + 64 < nblk ? st[lane + 64] : make_float2(0.f, 0.f);
        }
    }

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
        float4 gq[LNIN == 1 ? U : 1];
        if constexpr (LNIN == 1) {
            {
#pragma unroll
                for (int u = 0; u < U; ++u) gq[u] = *reinterpret_cast<const float4*>(p.ln_g + k0 + 16 * u + kq * 4);
            }
        }
        // the scheduler otherwise sinks loads between the MFMAs to save registers (two steps in flight: four dependent
        // latencies for a K = 2048 slice); the registers are there (4 waves per SIMD whatever the count: 128 each)
        if constexpr (RM * CN <= 2) __builtin_amdgcn_sched_barrier(0);      // (the 64-row tile keeps the compiler's rolling schedule)
        if constexpr (LNIN == 1) {
            {                                              // a <- gamma a; one rounding, the same in every tile shape
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float4 g4 = gq[u];
#pragma unroll
                    for (int i = 0; i < RM; ++i) { a[u][i].x *= g4.x; a[u][i].y *= g4.y; a[u][i].z *= g4.z; a[u][i].w *= g4.w; }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) mfma_step(a[u], b[u]);
    };
    if constexpr (RM * CN == 1 || (RM * CN == 2 && LNIN != 1))   // (a K = 2048 slice of the 16- and 32-row tiles in ONE round of loads)
        for (; it + 8 <= nit; it += 8) steps_of(std::integral_constant<int, 8>{}, kb + 16 * it);
    if constexpr (!(LNIN && RM * CN >= 8))       // (the LayerNorm form of the 64-row tile has no registers for four steps in flight)
        for (; it + 4 <= nit; it += 4) steps_of(std::integral_constant<int, 4>{}, kb + 16 * it);
    for (; (LNIN && RM * CN >= 8) && it + 2 <= nit; it += 2) steps_of(std::integral_constant<int, 2>{}, kb + 16 * it);
    if (it + 2 <= nit) { steps_of(std::integral_constant<int, 2>{}, kb + 16 * it); it += 2; }
    for (; it < nit; ++it) steps_of(std::integral_constant<int, 1>{}, kb + 16 * it);

    if constexpr (LNIN) {
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            // block partials are (sum, M2 = sum of squared deviations from the BLOCK mean) of 16 columns; merged by Chan's formula:
            // M2 = sum_b [M2_b + 16 (mean_b - mean)^2]. (Until round 5 the partials were (sum, sum of squares) and the variance
            // Q / K - mean^2, which cancels catastrophically for rows with |mean| >> std)
            const int nblk = p.K >> 4;
            float sx = st0[i].x + st1[i].x;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sx += __shfl_xor(sx, o, 64);
            const float mean = sx / (float)p.K;
            float m2 = 0.f;
            if (lane < nblk) { const float d = st0[i].x * 0.0625f - mean; m2 += st0[i].y + 16.f * d * d; }
            if (lane + 64 < nblk) { const float d = st1[i].x * 0.0625f - mean; m2 += st1[i].y + 16.f * d * d; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 64);
            const float var = m2 / (float)p.K;
            if (lane == 0) ln_ms[wv + NW * i] = make_float2(mean, 1.0f / sqrtf(var + p.ln_eps));
        }
        // (visible to the epilogue's threads through the barrier that also publishes the slice tiles)
    }
    // bias / ReLU / addends / store (+ the LayerNorm partials) of one finished output; consecutive threads -> consecutive
    // columns of a row
    auto finish = [&](float v, int lr, int lc) {
        const int row = m0 + lr, cc = n0 + lc;
        if (row < p.M && cc < p.N) {
            if constexpr (LNIN) {
#pragma clang fp contract(off)
                const float2 ms = ln_ms[lr];
                v = ms.y * (v - ms.x * p.ln_c1[cc]) + p.ln_c2[cc];          // (c2 holds the bias)
            } else {
                if (p.bias) v += p.bias[cc];
            }
            if (p.relu) v = fmaxf(v, 0.f);
            if (p.R1) v = v + p.R1[(size_t)row * p.ldr1 + cc];
            if (p.R2) v = p.R2[(size_t)row * p.ldr2 + cc] + v;
            // (out_gamma: the consumer's gamma applied to the stored value; the partials below stay those of v)
            p.C[(size_t)row * p.ldc + cc] = p.out_gamma ? v * p.out_gamma[cc] : v;
            if (p.ln_stats_out) {
                // (sum, sum of squares) over this row's 16-column block: the 16 threads of the block are 16 consecutive,
                // 16-aligned lanes and all active (N % 16 == 0); xor butterfly 8, 4, 2, 1
                float sx = v;
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) sx += __shfl_xor(sx, o, 16);
                const float dv = v - sx * 0.0625f;               // deviation from the block mean: (sum, M2) partials, merged by Chan's formula
                float m2 = dv * dv;
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 16);
                if ((lc & 15) == 0)
                    reinterpret_cast<float2*>(p.ln_stats_out)[(size_t)row * (p.N >> 4) + (cc >> 4)] = make_float2(sx, m2);
            }
        }
    };
    // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    if constexpr (NW == SK_KW) {
        float* mine = red + (size_t)wave * (RM * 16 * LD);
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int j = 0; j < CN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) mine[(i * 16 + kq * 4 + r) * LD + j * 16 + i16] = acc[i][j][r];
        __syncthreads();
        // fixed-order sum of the K slices (wave 0 + 1 + ... + 15) + epilogue
        for (int t = threadIdx.x; t < RM * CN * 256; t += SK_KW * 64) {
            const int lr = t / (16 * CN), lc = t % (16 * CN);
            float v = red[lr * LD + lc];
#pragma unroll
            for (int w = 1; w < SK_KW; ++w) v += red[(size_t)w * (RM * 16 * LD) + lr * LD + lc];
            finish(v, lr, lc);
        }
    } else {
        static_assert(NW == SK_KW || CN == 1, "the four-workgroup form is built for 16-column tiles");
        constexpr int TILE = RM * 256;                                     // floats of a slice tile
        const int tile = blockIdx.y * gridDim.x + blockIdx.x;
        float* part = p.ws_part + (size_t)tile * SK_KW * TILE;
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(size_t)wave * TILE + (i * 16 + kq * 4 + r) * 16 + i16] = acc[i][0][r];
        __shared__ int arrived;
        __threadfence();                                                   // this thread's tile rows are visible device-wide ...
        __syncthreads();                                                   // ... for every thread of the workgroup
        if (threadIdx.x == 0) arrived = atomicAdd(p.ws_count + tile, 1);
        __syncthreads();
        if (arrived != SK_KW / NW - 1) return;                             // (uniform) not the last workgroup of the tile
        __threadfence();                                                   // the other workgroups' rows, not stale lines
        if (threadIdx.x == 0) p.ws_count[tile] = 0;                        // ready for the next launch (graph replays)
        for (int t = threadIdx.x; t < TILE; t += NW * 64) {
            float v = part[t];
#pragma unroll
            for (int w = 1; w < SK_KW; ++w) v += part[(size_t)w * TILE + t];
            finish(v, t >> 4, t & 15);
        }
    }
}

template <int RM, int CN, int LNIN>
__global__ __launch_bounds__(SK_KW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_skinny_kernel(GemmArgs p) {
    skinny_body<RM, CN, LNIN, SK_KW>(p);
}
template <int RM, int LNIN>
__global__ __launch_bounds__(256) void gemm_skinny_ws_kernel(GemmArgs p) { skinny_body<RM, 1, LNIN, 4>(p); }

// the same GEMM for up to 16 weight matrices on ONE A operand (blockIdx.z picks W / bias / C): the streaming decoder's sixteen
// key/value projections of the step's encoder rows do not depend on the token chain, so they leave it as one launch
template <int RM, int CN>
__global__ __launch_bounds__(SK_KW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void gemm_skinny_batch_kernel(GemmArgs p, GemmBatch t) {
    p.W = t.W[blockIdx.z]; p.bias = t.bias[blockIdx.z]; p.C = t.C[blockIdx.z];
    skinny_body<RM, CN, 0, SK_KW>(p);
}

template <int RM, int CN>
int launch_skinny_batch_t(const GemmArgs& a, const GemmBatch& t, hipStream_t stream) {
    constexpr int lds = SK_KW * RM * 16 * (CN * 16 + 1) * (int)sizeof(float);
    static PerDeviceOnce configured;
    if (!configured.done()) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_batch_kernel<RM, CN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured.mark();
    }
    dim3 grid(ceil_div(a.N, 16 * CN), ceil_div(a.M, 16 * RM), t.n), block(SK_KW * 64);
    hipLaunchKernelGGL((gemm_skinny_batch_kernel<RM, CN>), grid, block, lds, stream, a, t);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

template <int RM, int CN, int LNIN>
int launch_skinny_t(const GemmArgs& a, hipStream_t stream) {
    constexpr int lds = SK_KW * RM * 16 * (CN * 16 + 1) * (int)sizeof(float);
    static PerDeviceOnce configured;
    if (!configured.done()) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<RM, CN, LNIN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured.mark();
    }
    dim3 grid(ceil_div(a.N, 16 * CN), ceil_div(a.M, 16 * RM)), block(SK_KW * 64);
    hipLaunchKernelGGL((gemm_skinny_kernel<RM, CN, LNIN>), grid, block, lds, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}
template <int RM>
int launch_skinny_ws(const GemmArgs& a, hipStream_t stream) {
    dim3 grid(ceil_div(a.N, 16), ceil_div(a.M, 16 * RM), 4), block(256);
    if (a.ln_stats_in && a.ln_g) hipLaunchKernelGGL((gemm_skinny_ws_kernel<RM, 1>), grid, block, 0, stream, a);
    else if (a.ln_stats_in) hipLaunchKernelGGL((gemm_skinny_ws_kernel<RM, 2>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((gemm_skinny_ws_kernel<RM, 0>), grid, block, 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}
template <int RM, int CN>
int launch_skinny(const GemmArgs& a, hipStream_t stream) {
    if (a.ln_stats_in) return a.ln_g ? launch_skinny_t<RM, CN, 1>(a, stream) : launch_skinny_t<RM, CN, 2>(a, stream);
    return launch_skinny_t<RM, CN, 0>(a, stream);
}

}  // namespace

bool gemm_skinny_applicable(const GemmArgs& a) { return a.amax_val == nullptr && a.K % 16 == 0; }

int launch_gemm_skinny(const GemmArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 16 == 0, "gemm_skinny: K must be a multiple of 16");
    PF_REQUIRE(a.lda % 4 == 0 && a.ldw % 4 == 0, "gemm_skinny: row strides must be multiples of 4 floats");
    PF_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0 && a.C, "gemm_skinny: operands must be 16-B aligned");
    if (a.ln_stats_out) PF_REQUIRE(a.N % 16 == 0 && ((uintptr_t)a.ln_stats_out & 7) == 0, "gemm_skinny: LayerNorm partials need N % 16 == 0");
    if (a.ln_stats_in) PF_REQUIRE(a.ln_c1 && a.ln_c2 && ((uintptr_t)a.ln_stats_in & 7) == 0 && ((uintptr_t)a.ln_g & 15) == 0 && a.K <= 2048,
                                  "gemm_skinny: the LayerNorm form needs the block partials, the constants c1 / c2 (launch_ln_consts) and K <= 2048");
    if (a.out_gamma) PF_REQUIRE(a.ln_stats_out, "gemm_skinny: out_gamma goes with ln_stats_out");
    if (a.ws_part) {
        // four workgroups per tile (see skinny_body): the caller hands a workspace of gemm_skinny_ws_floats() floats and as many
        // zeroed counters as tiles; same bits as the one-workgroup form
        PF_REQUIRE(a.ws_count && a.M <= 32 && ((uintptr_t)a.ws_part & 15) == 0, "gemm_skinny: the four-workgroup form is for <= 32 rows");
        return a.M <= 16 ? launch_skinny_ws<1>(a, stream) : launch_skinny_ws<2>(a, stream);
    }
    if (a.M <= 16) return launch_skinny<1, 1>(a, stream);
    if (a.M <= 32) return launch_skinny<2, 1>(a, stream);
    return launch_skinny<4, 2>(a, stream);
}

int launch_gemm_skinny_batch(const GemmArgs& a, const GemmBatch& t, hipStream_t stream) {
    PF_REQUIRE(t.n >= 1 && t.n <= 16, "gemm_skinny_batch: 1 .. 16 weight matrices");
    PF_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.K % 16 == 0 && a.lda % 4 == 0 && a.ldw % 4 == 0 && ((uintptr_t)a.A & 15) == 0,
               "gemm_skinny_batch: K % 16, strides % 4, aligned A");
    PF_REQUIRE(!a.ln_stats_in && !a.ln_stats_out && !a.amax_val && !a.R1 && !a.R2, "gemm_skinny_batch: plain epilogue (bias, relu) only");
    for (int i = 0; i < t.n; ++i) PF_REQUIRE(t.W[i] && t.C[i] && ((uintptr_t)t.W[i] & 15) == 0, "gemm_skinny_batch: null or unaligned operand");
    if (a.M <= 16) return launch_skinny_batch_t<1, 1>(a, t, stream);
    if (a.M <= 32) return launch_skinny_batch_t<2, 1>(a, t, stream);
    return launch_skinny_batch_t<4, 2>(a, t, stream);
}

}  // namespace pf
