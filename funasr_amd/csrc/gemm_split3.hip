// fp32-accurate GEMM on the bf16 matrix cores: C = epilogue(A[M,K] * W[N,K]^T) with both operands held as THREE bf16
// planes (x = hi + mid + lo exactly: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); 8 + 8 + 8 significand
// bits cover fp32's 24) and six v_mfma_f32_32x32x16_bf16 products per operand pair:
//     hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid      (dropped: mid*lo, lo*mid <= 2^-24 |a||w| each,
//                                                              lo*lo <= 2^-32: the size of fp32's own product rounding)
// Every bf16 x bf16 product is exact in fp32 and the sums are carried in fp32, so the result has fp32-class error
// (measured against float64 in tests/test_kernels_gpu.py next to the v_mfma_f32_32x32x2_f32 kernel) while the
// matrix pipe runs at the bf16 rate: 2.5 PFLOP/s / 6 products = 417 TFLOP/s of fp32-equivalent work per GPU against
// 157 TFLOP/s for the fp32 MFMA. Same call sites as gemm_f32.hip (the dense Linear layers of the SAN-M blocks:
// funasr/models/sanm/attention.py:256,306, funasr/models/transformer/positionwise_feed_forward.py:32).
//
// Design (gfx950).
//   * 256 x 128 x 32 block tile, 8 waves in a 4 x 2 grid, each wave owns 2 x 2 MFMA tiles of 32 x 32; per 16-deep
//     k-step a wave issues 12 ds_read_b128 and 24 MFMAs (768 matrix-pipe cycles): matrix-pipe bound by construction.
//   * The six planes of a K tile (3 x 256 + 3 x 128 rows of 64 B) go HBM -> LDS with global_load_lds_dwordx4,
//     72 pieces of 1 KB, 9 per wave, double buffered (2 x 72 KB of LDS, one workgroup = 2 waves per SIMD per CU).
//   * LDS image: rows of 32 bf16 (64 B); 16-B chunk c of row r is stored at chunk c ^ ((r >> 2) & 3), applied to the
//     per-lane DMA source address and again on the read: the 16-lane groups of a ds_read_b128 hit 16 distinct slots.
//   * XCD-aware block order as in gemm_f32.hip.
// The planes are produced by the kernels that write the activations (LayerNorm, the relu epilogue of this kernel,
// split3_kernel below) and once per weight at load time.
#include "common.h"

namespace pf {

namespace {

constexpr int BM = 256, BN = 128;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KS = k extent of one LDS stage (32: 64-B rows, one workgroup per CU; 16: 32-B rows, two workgroups per CU)
template <int KS> struct Geo {
    static constexpr int ROWB = KS * 2;                         // bytes per LDS row of one plane
    static constexpr int CPR = ROWB / 16;                       // 16-B chunks per row
    static constexpr int RPB = 256 / ROWB;                      // rows per 256-B bank row
    static constexpr int RPP = 1024 / ROWB;                     // rows per 1-KB DMA piece
    static constexpr int A_PLANE_B = BM * ROWB, B_PLANE_B = BN * ROWB;
    static constexpr int STAGE_B = 3 * (A_PLANE_B + B_PLANE_B);
    static constexpr int NPIECE = STAGE_B / 1024;               // 72 / 36
    static constexpr int A_PIECES = 3 * BM / RPP;               // pieces of the A planes (48 / 24)
    static constexpr int PPW = (NPIECE + 7) / 8;                // pieces per wave (9 / 5, the last partly idle for KS = 16)
    static constexpr int STEPS = KS / 16;
};

template <int MODE, bool OUT3, int KS>
__global__ __launch_bounds__(512, KS == 16 ? 4 : 2) void gemm_split3_kernel(Gemm3Args p, int nM, int nN) {
    typedef Geo<KS> G;
    constexpr int ROWB = G::ROWB, CPR = G::CPR, RPP = G::RPP, PPW = G::PPW, STAGE_B = G::STAGE_B;
    constexpr int A_PLANE_B = G::A_PLANE_B, B_PLANE_B = G::B_PLANE_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * STAGE_B

    const int L = blockIdx.x;
    const int xcd = L & 7, j8 = L >> 3;
    const int mblk = (j8 / nN) * 8 + xcd, nblk = j8 % nN;
    if (mblk >= nM) return;
    const int m0 = mblk * BM, n0 = nblk * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int hh = lane >> 5, idx = lane & 31;

    // ---- DMA sources: a stage is NPIECE pieces of 1 KB (RPP rows of one plane), A planes first, then the W planes,
    //      laid out linearly in LDS; wave w issues pieces w, w + 8, ...; lane l of a piece lands at row l / CPR,
    //      physical chunk l % CPR, and fetches the logical chunk that the read-side swizzle expects there
    const unsigned short* src[PPW];
    {
        const int prow = lane / CPR;
        const int chunk = (lane % CPR) ^ ((prow / G::RPB) & (CPR - 1));
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave + 8 * i;
            if (q < G::A_PIECES) {
                int row = m0 + (q % (BM / RPP)) * RPP + prow;
                row = row < p.M ? row : p.M - 1;
                src[i] = p.A + (size_t)(q / (BM / RPP)) * p.a_plane + (size_t)row * p.lda + chunk * 8;
            } else {
                const int qq = (q < G::NPIECE ? q : G::NPIECE - 1) - G::A_PIECES;
                int col = n0 + (qq % (BN / RPP)) * RPP + prow;
                col = col < p.N ? col : p.N - 1;
                src[i] = p.W + (size_t)(qq / (BN / RPP)) * p.w_plane + (size_t)col * p.ldw + chunk * 8;
            }
        }
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (unsigned)wave * 1024);
    const bool last_piece_ok = wave + 8 * (PPW - 1) < G::NPIECE;           // wave-uniform
    auto piece = [&](int i, int buf, int kt) {
        if (i == PPW - 1 && !last_piece_ok) return;
        glds16(src[i] + kt * KS, lds0 + (unsigned)buf * STAGE_B + (unsigned)i * 8192);
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    // per-lane read offsets (bytes): row idx of a 32-row MFMA operand, chunk (2 s + h | h) ^ swizzle(row)
    const int f = (idx / G::RPB) & (CPR - 1);
    const int aoff = (wr * 64 + idx) * ROWB;
    const int boff = 3 * A_PLANE_B + (wc * 64 + idx) * ROWB;
    int coff[G::STEPS];
#pragma unroll
    for (int st = 0; st < G::STEPS; ++st) coff[st] = ((G::STEPS * st + hh) ^ f) * 16;

    const int nk = p.K / KS;
#pragma unroll
    for (int i = 0; i < PPW; ++i) piece(i, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        glds_wait_all();
        __syncthreads();
        // the DMA pieces of K tile kt+1 are issued one per group of four MFMAs: each piece costs the wave tens of issue
        // cycles, which the matrix pipe covers when they sit between MFMAs instead of in front of them
        const bool nxt = kt + 1 < nk;
        const int nb = (kt + 1) & 1;
        const unsigned char* sb = smem + (kt & 1) * STAGE_B;
#define PF_PIECE(I) do { if ((I) < PPW && nxt) piece((I) < PPW ? (I) : 0, nb, kt + 1); } while (0)
#define PF_PROD(PA, PB)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                    \
        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA], b[jj][PB], acc[i][jj], 0, 0, 0)
#define PF_LOAD(S)                                                                                                    \
    _Pragma("unroll") for (int pl = 0; pl < 3; ++pl) {                                                                \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                 \
            a[i][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + pl * A_PLANE_B + aoff + i * 32 * ROWB + coff[S]));   \
        _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                              \
            b[jj][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sb + pl * B_PLANE_B + boff + jj * 32 * ROWB + coff[S])); \
    }
        bf16x8 a[2][3], b[2][3];
        PF_LOAD(0)
        PF_PROD(1, 1); PF_PIECE(0);
        PF_PROD(0, 2); PF_PIECE(1);
        PF_PROD(2, 0); PF_PIECE(2);
        PF_PROD(0, 1); PF_PIECE(3);
        PF_PROD(1, 0); PF_PIECE(4);
        PF_PROD(0, 0); PF_PIECE(5);
        if constexpr (G::STEPS == 2) {
            PF_LOAD(G::STEPS - 1)
            PF_PROD(1, 1); PF_PIECE(6);
            PF_PROD(0, 2); PF_PIECE(7);
            PF_PROD(2, 0); PF_PIECE(8);
            PF_PROD(0, 1);
            PF_PROD(1, 0);
            PF_PROD(0, 0);
        }
#undef PF_PROD
#undef PF_LOAD
#undef PF_PIECE
    }

    // ---- epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)):
    //      through a wave-private LDS slab so that every global access is a 16-B piece of a 256-B row segment
    constexpr bool HAS_R1 = (MODE & 1) != 0, HAS_R2 = (MODE & 2) != 0;
    constexpr int ELD = 68;
    __syncthreads();
    float* slab = reinterpret_cast<float*>(smem) + wave * (32 * ELD);
    const int c4 = lane & 15, rsub = lane >> 4;
    const int col = n0 + wc * 64 + c4 * 4;
    const bool colok = col + 3 < p.N;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && colok) bias4 = *reinterpret_cast<const float4*>(p.bias + col);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[((r & 3) + 8 * (r >> 2) + 4 * hh) * ELD + jj * 32 + idx] = acc[i][jj][r];
        float4 v[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) v[it] = *reinterpret_cast<const float4*>(slab + (it * 4 + rsub) * ELD + c4 * 4);
        const int row0 = m0 + wr * 64 + i * 32 + rsub;
        if (!colok) continue;
        float4 r1[8], r2[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = row0 + it * 4;
            const int rr = row < p.M ? row : p.M - 1;
            if constexpr (HAS_R1) r1[it] = *reinterpret_cast<const float4*>(p.R1 + (size_t)rr * p.ldr1 + col);
            if constexpr (HAS_R2) r2[it] = *reinterpret_cast<const float4*>(p.R2 + (size_t)rr * p.ldr2 + col);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = row0 + it * 4;
            float o[4] = {v[it].x + bias4.x, v[it].y + bias4.y, v[it].z + bias4.z, v[it].w + bias4.w};
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
            }
            if constexpr (HAS_R1) { o[0] = o[0] + r1[it].x; o[1] = o[1] + r1[it].y; o[2] = o[2] + r1[it].z; o[3] = o[3] + r1[it].w; }
            if constexpr (HAS_R2) { o[0] = r2[it].x + o[0]; o[1] = r2[it].y + o[1]; o[2] = r2[it].z + o[2]; o[3] = r2[it].w + o[3]; }
            if (row >= p.M) continue;
            if constexpr (OUT3) {
                store_split3x4(p.C3 + (size_t)row * p.ldc3 + col, p.c_plane, o);
            } else {
                *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + col) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

// fp32 [M, N] (row stride ldx) -> three bf16 planes [M, ldy] (plane stride `plane` elements); columns N..ldy are zero
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, int ldx, unsigned short* __restrict__ y,
                                                     int ldy, size_t plane, int M, int N) {
    const int c4n = ldy >> 2;
    const size_t total = (size_t)M * c4n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int row = (int)(i / c4n), c = (int)(i % c4n) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (c + 3 < N) {
            const float4 t = *reinterpret_cast<const float4*>(x + (size_t)row * ldx + c);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c + e < N) v[e] = x[(size_t)row * ldx + c + e];
        }
        store_split3x4(y + (size_t)row * ldy + c, plane, v);
    }
}

template <int MODE, bool OUT3, int KS>
int launch_ks(const Gemm3Args& a, int nM, int nN, hipStream_t stream) {
    constexpr int LDS_B = 2 * Geo<KS>::STAGE_B;
    static PerDeviceOnce configured;
    if (!configured.done()) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split3_kernel<MODE, OUT3, KS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B));
        configured.mark();
    }
    const int nMpad = (nM + 7) / 8 * 8;
    hipLaunchKernelGGL((gemm_split3_kernel<MODE, OUT3, KS>), dim3((unsigned)nMpad * nN), dim3(512), LDS_B, stream, a, nM, nN);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}
// KS = 16 (32-B rows, two workgroups per CU so that one's epilogue / barrier sits under the other's MFMAs) measured
// 165 TF-equivalent against 180 for KS = 32 on the encoder shapes: the half-line DMA rows and twice the barriers cost
// more than the overlap returns
template <int MODE, bool OUT3>
int launch_one(const Gemm3Args& a, int nM, int nN, hipStream_t stream) {
    return launch_ks<MODE, OUT3, 32>(a, nM, nN, stream);
}

}  // namespace

int launch_split3(const float* x, int ldx, unsigned short* y, int ldy, size_t plane, int M, int N, hipStream_t stream) {
    PF_REQUIRE(M > 0 && N > 0 && ldy >= N && ldy % 4 == 0, "split3: ldy must cover N and be a multiple of 4");
    PF_REQUIRE(ldx % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0 && plane % 4 == 0, "split3: alignment");
    const size_t total = (size_t)M * (ldy >> 2);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(split3_kernel, dim3(blocks), dim3(256), 0, stream, x, ldx, y, ldy, plane, M, N);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_gemm_split3(const Gemm3Args& a, hipStream_t stream) {
    PF_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm_split3: empty problem");
    PF_REQUIRE(a.K % 32 == 0, "gemm_split3: K must be a multiple of 32 (pad the planes with zeros)");
    PF_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.a_plane % 8 == 0 && a.w_plane % 8 == 0, "gemm_split3: operand strides % 8");
    PF_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm_split3: operands must be 16-B aligned");
    PF_REQUIRE(a.N % 4 == 0, "gemm_split3: N % 4");
    if (a.C3) PF_REQUIRE(a.ldc3 % 4 == 0 && a.c_plane % 4 == 0 && ((uintptr_t)a.C3 & 7) == 0, "gemm_split3: plane output alignment");
    else PF_REQUIRE(a.C && a.ldc % 4 == 0 && ((uintptr_t)a.C & 15) == 0, "gemm_split3: output alignment");
    if (a.bias) PF_REQUIRE(((uintptr_t)a.bias & 15) == 0, "gemm_split3: bias alignment");
    if (a.R1) PF_REQUIRE(a.ldr1 % 4 == 0 && ((uintptr_t)a.R1 & 15) == 0, "gemm_split3: R1 alignment");
    if (a.R2) PF_REQUIRE(a.ldr2 % 4 == 0 && ((uintptr_t)a.R2 & 15) == 0, "gemm_split3: R2 alignment");
    const int nM = ceil_div(a.M, BM), nN = ceil_div(a.N, BN);
    const int mode = (a.R1 ? 1 : 0) | (a.R2 ? 2 : 0);
    if (a.C3) {
        PF_REQUIRE(mode == 0, "gemm_split3: the plane output has no residual form");
        return launch_one<0, true>(a, nM, nN, stream);
    }
    switch (mode) {
        case 0: return launch_one<0, false>(a, nM, nN, stream);
        case 1: return launch_one<1, false>(a, nM, nN, stream);
        case 2: return launch_one<2, false>(a, nM, nN, stream);
        default: return launch_one<3, false>(a, nM, nN, stream);
    }
}

}  // namespace pf
