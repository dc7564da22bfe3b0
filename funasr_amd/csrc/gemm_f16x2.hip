// fp32-accurate GEMM on the fp16 matrix cores with THREE products per result instead of the six of gemm_split3.hip:
// C = epilogue(A[M,K] * W[N,K]^T) with both operands held as TWO fp16 planes of the pre-scaled tensor
//     x * 2^e = hi + lo,   hi = f16(x * 2^e),  lo = f16(x * 2^e - hi)        (11 + 11 significand bits + lo's sign)
// and v_mfma_f32_32x32x16_f16 products  hi*hi + hi*lo + lo*hi  into ONE fp32 accumulator (each f16 x f16 product is
// exact in fp32). What is dropped or rounded away is <= 2^-24 |a||w| per term (lo*lo, and lo's own rounding): the size
// of fp32's product rounding, like the terms gemm_split3.hip drops. fp16 has 5 exponent bits, so the planes carry a
// per-tensor power-of-two scale 2^e chosen from an a-priori bound (weights: max |w| at load; activations: the LayerNorm
// / attention / ReLU bounds engine.hip derives from the parameters), which keeps every hi below 65504 and puts the
// typical lo far above fp16's subnormal step 2^-24; the epilogue undoes both scales with one exact multiplication.
// Matrix-pipe ceiling 2.5 PFLOP/s / 3 = 833 TFLOP/s of fp32-equivalent work. Same call sites as gemm_f32.hip /
// gemm_split3.hip (funasr/models/sanm/attention.py:256,306, funasr/models/transformer/positionwise_feed_forward.py:32).
//
// Design (gfx950): as gemm_split3.hip -- 8 waves in a 4 x 2 grid, 32-deep K stages of 64-B LDS rows moved HBM -> LDS by
// asm-issued global_load_lds_dwordx4 pieces (double buffered, one piece per group of MFMAs), chunk swizzle
// c ^ ((r >> 2) & 3), XCD-aware block order, LDS-slab float4 epilogue -- in two block shapes:
//   256 x 128 (wave = 2 x 2 MFMA tiles):  8 ds_read_b128 + 12 MFMA per 16-deep step, 48-KB stages
//   256 x 256 (wave = 2 x 4 MFMA tiles): 12 ds_read_b128 + 24 MFMA per 16-deep step, 64-KB stages, 2/3 of the L2 -> LDS
//                                         bytes per flop: the shape for N >= 1024
#include "common.h"
#include <cstdlib>
#include <type_traits>
#include "gemm_f16x2_epilogue.h"

namespace pf {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// wave grid WGM (M) x 2 (N); a wave owns WM x WN tiles of 32 x 32. KS_: depth of a K stage -- 32 (64-B LDS rows, two
// stages) or 16 (32-B rows, a ring of three stages: the shape for TWO workgroups per CU, see gemm_f16x2_kernel)
template <int WM, int WN, int WGM = 4, int KS_ = 32> struct Geo2 {
    static constexpr int NW = WGM * 2;                          // waves per workgroup
    static constexpr int BM = WGM * WM * 32, BN = 2 * WN * 32;
    static constexpr int KS = KS_, ROWB = 2 * KS, CPR = ROWB / 16, RPP = 1024 / ROWB;
    static constexpr int SWZ = CPR == 4 ? 2 : 3;                // rows per 256 B of LDS = 1 << SWZ: the swizzle changes that often
    static constexpr int A_PLANE_B = BM * ROWB, B_PLANE_B = BN * ROWB;
    static constexpr int STAGE_B = 2 * (A_PLANE_B + B_PLANE_B);
    // stages: two 32-deep ones; 16-deep: a ring of three (the two-workgroup shape) or, with eight waves, as many as the CU's
    // 160 KB hold (five 32-KB stages of the 256 x 256 block: four of them in flight, see the deep-ring loop)
    static constexpr int NSTG = KS == 16 ? (WGM == 4 ? (163840 / STAGE_B > 6 ? 6 : 163840 / STAGE_B) : 3) : 2;
    static constexpr int NPIECE = STAGE_B / 1024;               // 48 / 64 (24 in the 16-deep 128 x 256 shape)
    static constexpr int A_PIECES = 2 * BM / RPP;
    static constexpr int PPW = NPIECE / NW;                     // 6 / 8
    static constexpr int ELD = WN * 32 + 4;                     // epilogue slab row (floats)
    static constexpr int SLAB_B = NW * 32 * ELD * 4;
    static constexpr int LDS_B = NSTG * STAGE_B > SLAB_B ? NSTG * STAGE_B : SLAB_B;
};
template <int N> __device__ __forceinline__ void glds_wait_but() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// ABL (measurement only, tools/bench_gemm2.py): 1 = no global stores in the epilogue, 2 = no epilogue, 3 = no operand DMA
// KS_ = 16 with WGM = 2 (four waves, 128 x 256 block, 72 KB of LDS): TWO workgroups per CU. With one 512-thread workgroup per
// CU every CU's waves meet at the same barriers, load their first stages together and drain their epilogues together -- the
// matrix pipes idle through each prologue and epilogue (a quarter of a K = 512 GEMM). Two independent four-wave workgroups
// per CU (one wave of each per SIMD, the same 256 VGPRs per wave) run out of phase: one's barrier waits, first-stage loads
// and epilogue stores lie under the other's MFMAs. Price: 1.5x the L2 -> LDS bytes per flop of the 256 x 256 block, and one
// barrier per 16-deep step (24 MFMAs per wave) instead of per 32-deep stage; the stage ring is three deep so that a stage's
// DMA has two whole steps to land. Same products, same k order per output element as the other shapes: bitwise equal (tested).
// one block tile: rows m0 .., columns n0 .. of the problem (`nblk` = the column block's index, for the arg-max form's partials)
template <int WM, int WN, int MODE, int OUT, int ABL, int SCHED, int WGM, int KS_>
__device__ __forceinline__ void gemm2_tile_body(Gemm2Args& p, const int m0, const int n0, const int nblk, unsigned char* smem) {
    typedef Geo2<WM, WN, WGM, KS_> G;
    constexpr int BM = G::BM, BN = G::BN, KS = G::KS, ROWB = G::ROWB, CPR = G::CPR, RPP = G::RPP, PPW = G::PPW;
    constexpr int STAGE_B = G::STAGE_B, A_PLANE_B = G::A_PLANE_B, B_PLANE_B = G::B_PLANE_B;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int hh = lane >> 5, idx = lane & 31;

    // ---- DMA sources: a stage is NPIECE pieces of 1 KB (16 rows of one plane), the A planes first, then the W planes,
    //      linear in LDS; wave w issues pieces w, w + 8, ...; lane l lands at row l / 4, physical chunk l % 4 and
    //      fetches the logical chunk the read-side swizzle expects there
    const unsigned short* src[PPW];
    {
        const int prow = lane / CPR;
        const int chunk = (lane % CPR) ^ ((prow >> G::SWZ) & (CPR - 1));
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int q = wave + G::NW * i;
            if (q < G::A_PIECES) {
                int row = m0 + (q % (BM / RPP)) * RPP + prow;
                row = row < p.M ? row : p.M - 1;
                src[i] = p.A + (size_t)(q / (BM / RPP)) * p.a_plane + (size_t)row * p.lda + chunk * 8;
            } else {
                const int qq = q - G::A_PIECES;
                int col = n0 + (qq % (BN / RPP)) * RPP + prow;
                col = col < p.N ? col : p.N - 1;
                src[i] = p.W + (size_t)(qq / (BN / RPP)) * p.w_plane + (size_t)col * p.ldw + chunk * 8;
            }
        }
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (unsigned)wave * 1024);
    // elements between the K stages of an operand row: KS in the row-major plane layout [rows, K]; rows * KS in the K-blocked
    // layout [K / KS][rows][KS] (a_kstep / w_kstep > 0), where a piece's 16 rows are ONE contiguous KB = eight whole 128-B lines
    // (row-major, a piece touches 64 B of sixteen lines and the other halves are fetched again by the next stage)
    const size_t a_step = p.a_kstep > 0 ? (size_t)p.a_kstep : (size_t)KS, w_step = p.w_kstep > 0 ? (size_t)p.w_kstep : (size_t)KS;
    auto piece = [&](int i, int buf, int kt) {
        if constexpr (ABL == 3) return;
        glds16(src[i] + kt * (i < G::A_PIECES / G::NW ? a_step : w_step), lds0 + (unsigned)buf * STAGE_B + (unsigned)i * (G::NW * 1024));
    };

    floatx16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int jj = 0; jj < WN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    const int f = (idx >> G::SWZ) & (CPR - 1);
    const int aoff = (wr * (WM * 32) + idx) * ROWB;
    const int boff = 2 * A_PLANE_B + (wc * (WN * 32) + idx) * ROWB;
    int coff[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) coff[st] = ((2 * st + hh) ^ f) * 16;

    const int nk = p.K / KS;
    // SCHED 3 (A/B, round 5): the plain order + static priority for the second-dispatched half of the workgroup (waves 4-7 lose the
    // issue arbitration against their older SIMD partners on every segment; MI355X guide, "two waves per SIMD", item 4)
    if constexpr (SCHED == 3) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
    if constexpr (KS_ == 16 && WGM == 4) {
        // Deep ring with EXACT waits (round 4). Five 32-KB buffers of 16-deep stages; at step kt stage kt is in registers, stage
        // kt + 1 is read from LDS into the other fragment set, stages kt + 2 .. kt + 5 are in flight: four stages = 128 KB against
        // the ONE 64-KB stage of the two-stage loop below, whose cover (64 KB / 1.4 us of loaded round trip = 46 GB/s per CU) is
        // less than the block's own appetite at the full matrix rate. Four stages in flight cannot be waited for with
        // counted vmcnt -- LDS-DMA pieces of one wave retire out of issue order when their sources differ (round 3's
        // form of this loop did exactly that and is unsafe) -- so every stage has ONE owner: wave pair g = wave >> 1 issues all 32
        // pieces of the stages s with s % 4 == g and is the only pair that waits for them, with vmcnt(0), when that stage is the
        // only thing it has in flight (its next stage is issued right after the barrier that publishes this one). The pairs of a
        // SIMD's two waves differ, so an owner's issue burst runs under its SIMD partner's MFMAs. Same products in the same k order
        // as every other shape: bitwise equal (tested).
        constexpr int R = G::NSTG;
        static_assert(R == 5 && G::NPIECE == 32 && G::A_PIECES == 16 && RPP == 32, "deep ring: five 32-KB stages of 32-row pieces");
        const int grp = wave >> 1, hw = wave & 1;
        const int prow = lane / CPR;
        const unsigned chunkb = (unsigned)(((lane % CPR) ^ ((prow >> G::SWZ) & (CPR - 1))) * 16);
        const unsigned ldsb = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
        const size_t a_step2 = (p.a_kstep > 0 ? (size_t)p.a_kstep : (size_t)KS) * 2, w_step2 = (p.w_kstep > 0 ? (size_t)p.w_kstep : (size_t)KS) * 2;
        // this wave's piece i (0..15) of stage kt -> buffer `buf`: piece q = hw + 2 i of the stage (A plane 0 rows 32 q.., A plane 1,
        // W plane 0, W plane 1; 8 pieces each)
        auto issue = [&](int kt, int buf, int i) {
            if constexpr (ABL == 3) return;
            const unsigned dst = ldsb + (unsigned)buf * STAGE_B + (unsigned)(hw + 2 * i) * 1024;
            if (i < 8) {
                int row = m0 + (hw + 2 * (i & 3)) * RPP + prow;
                row = row < p.M ? row : p.M - 1;
                glds16s(reinterpret_cast<const char*>(p.A) + (size_t)(i >> 2) * p.a_plane * 2 + (size_t)kt * a_step2,
                        (unsigned)row * (unsigned)p.lda * 2u + chunkb, dst);
            } else {
                int col = n0 + (hw + 2 * (i & 3)) * RPP + prow;
                col = col < p.N ? col : p.N - 1;
                glds16s(reinterpret_cast<const char*>(p.W) + (size_t)((i - 8) >> 2) * p.w_plane * 2 + (size_t)kt * w_step2,
                        (unsigned)col * (unsigned)p.ldw * 2u + chunkb, dst);
            }
        };
        auto issue_stage = [&](int kt, int buf) {
#pragma unroll
            for (int i = 0; i < 16; ++i) issue(kt, buf, i);
        };
        auto frags = [&](f16x8 (&a)[WM][2], f16x8 (&b)[WN][2], int buf) {
            const unsigned char* sb = smem + buf * STAGE_B;
#pragma unroll
            for (int pl = 1; pl >= 0; --pl) {      // the lo planes first: the first product is lo * hi
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    a[i][pl] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(sb + pl * A_PLANE_B + aoff + i * 32 * ROWB + coff[0]));
            }
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int jj = 0; jj < WN; ++jj)
                    b[jj][pl] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(sb + pl * B_PLANE_B + boff + jj * 32 * ROWB + coff[0]));
            }
        };
        // prologue: pair g fills buffer g with stage g; stage 0 is published; pair 0 (nothing in flight any more) sends stage 4
        if (grp < nk) issue_stage(grp, grp);
        if (grp == 0) glds_wait_all();
        __syncthreads();
        if (grp == 0 && 4 < nk) issue_stage(4, 4);
        f16x8 a0[WM][2], b0[WN][2], a1[WM][2], b1[WN][2];
        frags(a0, b0, 0);
        int buf = 0;                               // kt % R
        // `more`: a stage follows (compile-time, so that the fragment reads and the MFMAs share one basic block and the
        // compiler's lgkmcnt waits stay exact -- behind a branch join it waits for the just-issued reads before the first MFMA)
        auto step = [&](auto more, int kt, f16x8 (&a)[WM][2], f16x8 (&b)[WN][2], f16x8 (&an)[WM][2], f16x8 (&bn)[WN][2]) {
            const int nbuf = buf + 1 == R ? 0 : buf + 1;
            if constexpr (decltype(more)::value) {
                const bool own = grp == ((kt + 1) & 3);                // owner of stage kt + 1 (and of stage kt + 5)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's fragment reads of stage kt are complete
                if (own) glds_wait_all();                              // stage kt + 1: the only thing this pair has in flight
                __syncthreads();
                frags(an, bn, nbuf);
                if (own && kt + R < nk) issue_stage(kt + R, buf);      // into the buffer stage kt just left
                __builtin_amdgcn_sched_barrier(0);     // the reads go out before the MFMAs (a step's worth of time to land)
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int jj = 0; jj < WN; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[jj][0], acc[i][jj], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int jj = 0; jj < WN; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[jj][1], acc[i][jj], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int jj = 0; jj < WN; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[jj][0], acc[i][jj], 0, 0, 0);
            buf = nbuf;
        };
        typedef std::integral_constant<bool, true> More;
        typedef std::integral_constant<bool, false> Last;
        for (int kt = 0; kt + 2 < nk; kt += 2) {   // K % 32 == 0: an even number of 16-deep steps
            step(More{}, kt, a0, b0, a1, b1);
            step(More{}, kt + 1, a1, b1, a0, b0);
        }
        step(More{}, nk - 2, a0, b0, a1, b1);
        step(Last{}, nk - 1, a1, b1, a0, b0);
    } else if constexpr (KS_ == 16) {
        // ring of three 16-deep stages: stage kt + 2 is issued while stage kt is multiplied. MEASUREMENT HOOK ONLY (tile 5, never
        // chosen by shape): the counted wait below assumes that a wave's LDS-DMA pieces retire in issue order, so that "at most
        // PPW outstanding" means its pieces of stage kt have landed. Round 4 showed that assumption false on this hardware when
        // the pieces come from different sources (gemm_f16x2_ffn.hip header: a few stale workgroups per thousand at 50-block
        // depth); the shapes the engine picks wait with vmcnt(0) or by buffer ownership (the deep ring below).
#pragma unroll
        for (int i = 0; i < PPW; ++i) piece(i, 0, 0);
        if (nk > 1) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) piece(i, 1, 1);
        }
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) glds_wait_but<PPW>(); else glds_wait_all();
            __syncthreads();
            const unsigned char* sb = smem + (kt % 3) * STAGE_B;
            f16x8 a[WM][2], b[WN][2];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    a[i][pl] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(sb + pl * A_PLANE_B + aoff + i * 32 * ROWB + coff[0]));
#pragma unroll
                for (int jj = 0; jj < WN; ++jj)
                    b[jj][pl] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(sb + pl * B_PLANE_B + boff + jj * 32 * ROWB + coff[0]));
            }
            if (kt + 2 < nk) {
#pragma unroll
                for (int i = 0; i < PPW; ++i) piece(i, (kt + 2) % 3, kt + 2);
            }
#define PF_PROD16(PA, PB)                                                                                             \
    _Pragma("unroll") for (int i = 0; i < WM; ++i) _Pragma("unroll") for (int jj = 0; jj < WN; ++jj)                  \
        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][PA], b[jj][PB], acc[i][jj], 0, 0, 0)
            PF_PROD16(1, 0);
            PF_PROD16(0, 1);
            PF_PROD16(0, 0);
#undef PF_PROD16
        }
    } else {
#pragma unroll
    for (int i = 0; i < PPW; ++i) piece(i, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        glds_wait_all();
        __syncthreads();
        const bool nxt = kt + 1 < nk;
        const int nb = (kt + 1) & 1;
        const unsigned char* sb = smem + (kt & 1) * STAGE_B;
#define PF_PIECE(I) do { if ((I) < PPW && nxt) piece((I) < PPW ? (I) : 0, nb, kt + 1); } while (0)
#define PF_PROD(PA, PB)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < WM; ++i) _Pragma("unroll") for (int jj = 0; jj < WN; ++jj)                  \
        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][PA], b[jj][PB], acc[i][jj], 0, 0, 0)
#define PF_LOAD(S)                                                                                                    \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                                                                \
        _Pragma("unroll") for (int i = 0; i < WM; ++i)                                                                \
            a[i][pl] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(sb + pl * A_PLANE_B + aoff + i * 32 * ROWB + coff[S]));   \
        _Pragma("unroll") for (int jj = 0; jj < WN; ++jj)                                                             \
            b[jj][pl] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(sb + pl * B_PLANE_B + boff + jj * 32 * ROWB + coff[S])); \
    }
        if constexpr (SCHED == 0 || SCHED == 3) {
            f16x8 a[WM][2], b[WN][2];
            // the two small products first, hi*hi last (fixed order: results do not depend on the block shape's schedule)
            PF_LOAD(0)
            PF_PROD(1, 0); PF_PIECE(0); PF_PIECE(1);
            PF_PROD(0, 1); PF_PIECE(2); PF_PIECE(3);
            PF_PROD(0, 0); PF_PIECE(4);
            PF_LOAD(1)
            PF_PROD(1, 0); PF_PIECE(5); PF_PIECE(6);
            PF_PROD(0, 1); PF_PIECE(7);
            PF_PROD(0, 0);
        } else {
            // both k-steps' fragments are requested before the first MFMA: one LDS-latency bubble per stage instead of two
            f16x8 a[WM][2], b[WN][2], a1[WM][2], b1[WN][2];
            PF_LOAD(0)
            {
#define a a1
#define b b1
                PF_LOAD(1)
#undef a
#undef b
            }
            if constexpr (SCHED == 2) { PF_PIECE(0); PF_PIECE(1); PF_PIECE(2); PF_PIECE(3); }
            PF_PROD(1, 0); if constexpr (SCHED == 1) { PF_PIECE(0); PF_PIECE(1); } else { PF_PIECE(4); PF_PIECE(5); }
            PF_PROD(0, 1); if constexpr (SCHED == 1) { PF_PIECE(2); PF_PIECE(3); } else { PF_PIECE(6); PF_PIECE(7); }
            PF_PROD(0, 0); if constexpr (SCHED == 1) { PF_PIECE(4); }
#define a a1
#define b b1
            PF_PROD(1, 0); if constexpr (SCHED == 1) { PF_PIECE(5); PF_PIECE(6); }
            PF_PROD(0, 1); if constexpr (SCHED == 1) { PF_PIECE(7); }
            PF_PROD(0, 0);
#undef a
#undef b
        }
#undef PF_PROD
#undef PF_LOAD
#undef PF_PIECE
    }
    }

    gemm2_epilogue<WM, WN, G::NW, MODE, OUT, ABL>(p, acc, smem, m0, n0, nblk, wave, wr, wc, lane);
}
template <int WM, int WN, int MODE, int OUT, int ABL = 0, int SCHED = 0, int WGM = 4, int KS_ = 32>   // OUT: 0 fp32 C, 1 two fp16 planes, 2 the QKV form (Gemm2Args)
__global__ __launch_bounds__(WGM * 128, 2) void gemm_f16x2_kernel(Gemm2Args p, int nM, int nN) {
    typedef Geo2<WM, WN, WGM, KS_> G;
    constexpr int BM = G::BM, BN = G::BN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int L = blockIdx.x;
    const int xcd = L & 7, j8 = L >> 3;
    const int mblk = (j8 / nN) * 8 + xcd, nblk = j8 % nN;
    if (mblk >= nM) return;
    const int m0 = mblk * BM, n0 = nblk * BN;
    if constexpr (OUT == 0 && MODE == 0) {
        // split-K form: slice blockIdx.y multiplies columns [y K, (y + 1) K) of both operands (p.K is the slice length, the row
        // strides are the full ones) into its own [M, N] partial
        if (p.kslices > 1) {
            const size_t z = blockIdx.y;
            p.A += z * (size_t)p.K;
            p.W += z * (size_t)p.K;
            p.C += z * (size_t)p.M * (size_t)p.ldc;
        }
    }

    gemm2_tile_body<WM, WN, MODE, OUT, ABL, SCHED, WGM, KS_>(p, m0, n0, nblk, smem);
}

// (Round 5 built a de-phased block order on top of gemm2_tile_body -- every second workgroup of the first round computes a
// 256 x 128 half of its tile and the other halves close the list, so that only half of the CUs run their epilogue's store burst
// at a time. Bitwise equal, and SLOWER in both of its forms: w_1 planes 205 -> 223 us, the step 51.0 -> 52.6 ms
// (profiles/r05x_ab_dephased_first_order.txt, r05y_ab_dephased_second_order.txt), although de-phasing by a plain DELAY
// recovers 12-15 us per launch (profiles/r05r_dephase_experiment.jsonl; that measurement build is gone too): the narrow tiles and
// the changed order cost more than the bursts. Removed.)

// fp32 [M, N] (row stride ldx) * scale -> two fp16 planes [M, ldy] (`plane` elements apart); columns N..ldy are zero
__global__ __launch_bounds__(256) void split2_kernel(const float* __restrict__ x, int ldx, unsigned short* __restrict__ y,
                                                     int ldy, size_t plane, int M, int N, float scale,
                                                     const float* __restrict__ scale_dev, int seq_out, int seq_in) {
    if (scale_dev) scale *= *scale_dev;
    const int c4n = ldy >> 2;
    const size_t total = (size_t)M * c4n;
    // threads 2k / 2k + 1 hold neighbouring 4-column groups of one row and run the same trip count when ldy % 8 == 0
    const bool wide_st = ldy % 8 == 0 && plane % 8 == 0 && ((uintptr_t)y & 15) == 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int row = (int)(i / c4n), c = (int)(i % c4n) * 4;
        int xrow = row;
        bool pad = false;
        if (seq_out > 0) {
            const int b = row / seq_out, t = row % seq_out;
            pad = t >= seq_in;
            xrow = b * seq_in + t;
        }
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (pad) {
        } else if (c + 3 < N) {
            const float4 t = *reinterpret_cast<const float4*>(x + (size_t)xrow * ldx + c);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c + e < N) v[e] = x[(size_t)xrow * ldx + c + e];
        }
        if (wide_st) store_split2x4_pair(y + (size_t)row * ldy + c, plane, v, scale, threadIdx.x);
        else store_split2x4(y + (size_t)row * ldy + c, plane, v, scale);
    }
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(out, __builtin_bit_cast(unsigned, m));   // non-negative floats order like their bits
}

// one wave per row n: bound_n = in_bound * sum_k |W[n, k]| + |bias[n]|; *out = max_n bound_n
__global__ __launch_bounds__(256) void rowl1_bound_kernel(const float* __restrict__ W, int rows, int cols, int ld,
                                                          const float* __restrict__ bias, float in_bound, unsigned* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float a = 0.f;
    for (int k = lane; k < cols; k += 64) a += fabsf(W[(size_t)row * ld + k]);
    a = wave_sum(a);
    // rounded up a little: the fp32 sum above is not an upper bound to the last bit
    const float bnd = (in_bound * a) * 1.0001f + (bias ? fabsf(bias[row]) : 0.f);
    if (lane == 0) atomicMax(out, __builtin_bit_cast(unsigned, bnd));
}

// split-K form, second launch: C = epilogue(sum over slices of part[z]) with the tile kernel's epilogue order (bias, relu, + R1,
// R2 +); slices are added in slice order by one thread per float4 (deterministic). R1 / R2 may alias C.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int slices, size_t slice_stride, int M, int N,
                                                            const float* __restrict__ bias, int relu, const float* R1, int ldr1,
                                                            const float* R2, int ldr2, float* C, int ldc, unsigned short* C2, int ldc2,
                                                            size_t c_plane, float cscale) {
    const int n4 = N >> 2;
    const size_t total = (size_t)M * n4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int row = (int)(i / n4), col = (int)(i % n4) * 4;
        const float* src = part + (size_t)row * N + col;
        float4 v = *reinterpret_cast<const float4*>(src);
        for (int z = 1; z < slices; ++z) {
            const float4 t = *reinterpret_cast<const float4*>(src + (size_t)z * slice_stride);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (bias) {
            const float4 b = *reinterpret_cast<const float4*>(bias + col);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (R1) {
            const float4 r = *reinterpret_cast<const float4*>(R1 + (size_t)row * ldr1 + col);
            v.x = v.x + r.x; v.y = v.y + r.y; v.z = v.z + r.z; v.w = v.w + r.w;
        }
        if (R2) {
            const float4 r = *reinterpret_cast<const float4*>(R2 + (size_t)row * ldr2 + col);
            v.x = r.x + v.x; v.y = r.y + v.y; v.z = r.z + v.z; v.w = r.w + v.w;
        }
        if (C2) {                                   // the planes of the result * cscale (the next GEMM's operand)
            const float o[4] = {v.x, v.y, v.z, v.w};
            store_split2x4(C2 + (size_t)row * ldc2 + col, c_plane, o, cscale);
        } else {
            *reinterpret_cast<float4*>(C + (size_t)row * ldc + col) = v;
        }
    }
}

template <int WM, int WN, int MODE, int OUT, int ABL = 0, int SCHED = 0, int WGM = 4, int KS_ = 32>
int launch_tile(const Gemm2Args& a, hipStream_t stream) {
    typedef Geo2<WM, WN, WGM, KS_> G;
    static PerDeviceOnce configured;
    if (!configured.done()) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_kernel<WM, WN, MODE, OUT, ABL, SCHED, WGM, KS_>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_B));
        configured.mark();
    }
    const int nM = ceil_div(a.M, G::BM), nN = ceil_div(a.N, G::BN);
    const int nMpad = (nM + 7) / 8 * 8;
    const unsigned gy = (OUT == 0 && MODE == 0 && a.kslices > 1) ? (unsigned)a.kslices : 1u;
    hipLaunchKernelGGL((gemm_f16x2_kernel<WM, WN, MODE, OUT, ABL, SCHED, WGM, KS_>), dim3((unsigned)nMpad * nN, gy), dim3(WGM * 128), G::LDS_B, stream, a, nM, nN);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}
#if defined(PF_MEASUREMENT_KERNELS)
// NOTE (round 4): the 128 x 256 shape below (tile 5) stages its operands behind COUNTED vmcnt waits (glds_wait_but). LDS-DMA pieces of
// one wave were found to retire out of issue order when their sources differ (gemm_f16x2_ffn.hip header): it passes the warm kernel
// tests but is NOT safe inside a long pipeline -- a measurement hook only. The deep ring (tile 6) was rebuilt on exact waits with one
// owner wave pair per stage.
// the 128 x 256 four-wave shape, two workgroups per CU (tile 5; N % 256 == 0)
template <int MODE, int OUT>
int launch_pair(const Gemm2Args& a, hipStream_t stream) { return launch_tile<2, 4, MODE, OUT, 0, 0, 2, 16>(a, stream); }
// the 256 x 256 eight-wave shape over a deep ring of 16-deep stages (tile 6; N % 256 == 0)
template <int MODE, int OUT>
int launch_ring(const Gemm2Args& a, hipStream_t stream) { return launch_tile<2, 4, MODE, OUT, 0, 0, 4, 16>(a, stream); }
#endif
template <int MODE, int OUT>
int launch_one(const Gemm2Args& a, hipStream_t stream) {
    // block shape by N only (never by the batch's M: a clip's result must not depend on what else is in the batch --
    // and does not anyway: both shapes issue the same products in the same k order)
    if constexpr (OUT == 1) {
        if (a.tile == 3) return launch_tile<2, 2, MODE, 1, 0, 0, 2>(a, stream);   // 128 x 128, 4 waves (plane output)
    }
    if constexpr (OUT == 0) {
        if (a.tile == 3) return launch_tile<2, 2, MODE, 0, 0, 0, 2>(a, stream);   // 128 x 128, 4 waves, two workgroups per CU
    }
#if defined(PF_MEASUREMENT_KERNELS)
    if constexpr (OUT == 0) {              // measurement hooks (tools/abl_gemm2.py)
        if (a.tile >= 256) return (a.tile >> 8) == 1 ? launch_tile<2, 4, MODE, 0, 0, 1>(a, stream) : launch_tile<2, 4, MODE, 0, 0, 2>(a, stream);
    }
    if constexpr (MODE == 0 && OUT == 0) {
        // ablation builds of the 128 x 128 four-wave shape (tools/repro_inproc.py: which part of it disturbs a co-resident
        // frontend): 0x82 = no epilogue, 0x83 = no operand DMA
        if (a.tile == 0x82) return launch_tile<2, 2, 0, 0, 2, 0, 2>(a, stream);
        if (a.tile == 0x83) return launch_tile<2, 2, 0, 0, 3, 0, 2>(a, stream);
        if (a.tile >= 16) {                 // ablation builds of the wide tile: bits 4.. = ABL
            switch (a.tile >> 4) {
                case 1: return launch_tile<2, 4, 0, 0, 1>(a, stream);
                case 2: return launch_tile<2, 4, 0, 0, 2>(a, stream);
                case 3: return launch_tile<2, 4, 0, 0, 3>(a, stream);
                default: return launch_tile<2, 4, 0, 0, 4>(a, stream);
            }
        }
    }
#endif
    // 256 x 256 blocks or 256 x 128 ones (twice as many, each 0.57 of the time: tools/abl_gemm2.py), whichever needs less
    // time in whole rounds over the CUs: 290 wide blocks on 256 CUs are two rounds, 580 narrow ones three half-rounds.
    // The choice moves no result: both shapes issue the same products in the same k order per output element (bitwise
    // equal, tested), so a clip's result still does not depend on what else is in the batch.
    const int n_cu = device_cu_count();
    const long m_tiles = ceil_div(a.M, 256);
    const long wide_blocks = m_tiles * (a.N / 256), narrow_blocks = m_tiles * ceil_div(a.N, 128);
    const double cost_wide = (double)((wide_blocks + n_cu - 1) / n_cu), cost_narrow = 0.57 * (double)((narrow_blocks + n_cu - 1) / n_cu);
    const bool wide = a.tile == 2 || a.tile == 8 || a.tile == 9 || (a.tile == 0 && a.N % 256 == 0 && cost_wide <= cost_narrow);
    if constexpr (OUT == 0) {
        // fp32 output: the 128 x 128 four-wave shape (two workgroups per CU, 0.55 of a 256 x 256 block's time per round of
        // 2 n_cu blocks: tools/bench_r03.py `dec`) wins where the larger shapes leave CUs idle -- the decoder's token-side GEMMs
        // (M ~ 11 k rows: 82 vs 86-89 us) and short ragged batches. Bitwise equal to the other shapes like them.
        const long small_blocks = (long)ceil_div(a.M, 128) * ceil_div(a.N, 128);
        const double cost_small = 0.55 * (double)((small_blocks + 2 * n_cu - 1) / (2 * n_cu));
        if (a.tile == 0 && cost_small < (wide ? cost_wide : cost_narrow)) return launch_tile<2, 2, MODE, 0, 0, 0, 2>(a, stream);
    }
    if constexpr (OUT == 1) {
        // plane output (w_1): the 128 x 128 shape only where the 256-row shapes leave most of the chip idle (the streaming step, short
        // batches: M = 960 17 vs 27 us, 1920 21 vs 31, 3840 33 vs 34 -- profiles/r03y_small_ring_microbench.jsonl, fp32-output form)
        if (a.tile == 0 && 2 * narrow_blocks <= n_cu) return launch_tile<2, 2, MODE, 1, 0, 0, 2>(a, stream);
    }
    // SCHED 2 on the 256 x 256 shape (both k-steps' fragments requested up front, DMA pieces early): 0-10 % faster there
    // The wide shape's schedule. Round 2 chose SCHED 2 (both k-steps' fragments requested before the first MFMA, the DMA pieces
    // early: "0-10 % faster"); on round 5's tree and boxes the plain order (SCHED 0: a k-step's fragments, its three products with
    // the pieces between them, then the next k-step's) is 6-15 % FASTER per launch and 1.4 % per step
    // (profiles/r05s_ab_wide_tile_sched0.txt: w_1 planes 230 -> 211 us, QKV form 185.5 -> 174.5, w_2 212 -> 191; same bits).
    // tile 8 = SCHED 2 stays reachable for A/B runs.
#if defined(PF_MEASUREMENT_KERNELS)
    if (wide && a.tile == 8) return launch_tile<2, 4, MODE, OUT, 0, 2>(a, stream);
    if (wide && a.tile == 9) return launch_tile<2, 4, MODE, OUT, 0, 3>(a, stream);
#endif
    return wide ? launch_tile<2, 4, MODE, OUT, 0, 0>(a, stream) : launch_tile<2, 2, MODE, OUT>(a, stream);
}

}  // namespace

__global__ void pow2_scale_kernel(const float* __restrict__ amax, float* __restrict__ sc) {
    const float a = *amax;
    int e = a > 0.f ? (int)floorf(log2f(32768.f / a)) : 0;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    sc[0] = ldexpf(1.f, e);
    sc[1] = ldexpf(1.f, -e);
}

int launch_pow2_scale(const float* amax_dev, float* sc, hipStream_t stream) {
    hipLaunchKernelGGL(pow2_scale_kernel, dim3(1), dim3(1), 0, stream, amax_dev, sc);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

__global__ void kv_scales_kernel(const float* __restrict__ amax, const float* __restrict__ lb, int n, float* __restrict__ out) {
    const int i = blockIdx.x * 64 + threadIdx.x;          // one (layer, k | v) pair per thread
    if (i >= 2 * n) return;
    const float bound = *amax * lb[2 * i] + lb[2 * i + 1];
    int e = bound > 0.f ? (int)floorf(log2f(32768.f / bound)) : 0;
    e = e > 60 ? 60 : (e < -60 ? -60 : e);
    out[4 * (i >> 1) + (i & 1)] = ldexpf(1.f, e);
    out[4 * (i >> 1) + 2 + (i & 1)] = ldexpf(1.f, -e);
}

int launch_kv_scales(const float* amax_dev, const float* l1_bmax_dev, int n_layers, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(kv_scales_kernel, dim3(ceil_div(2 * n_layers, 64)), dim3(64), 0, stream, amax_dev, l1_bmax_dev, n_layers, out);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_split2(const float* x, int ldx, unsigned short* y, int ldy, size_t plane, int M, int N, float scale,
                  hipStream_t stream, const float* scale_dev, int seq_out, int seq_in) {
    PF_REQUIRE(M > 0 && N > 0 && ldy >= N && ldy % 4 == 0, "split2: ldy must cover N and be a multiple of 4");
    PF_REQUIRE(ldx % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0 && plane % 4 == 0, "split2: alignment");
    const size_t total = (size_t)M * (ldy >> 2);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(split2_kernel, dim3(blocks), dim3(256), 0, stream, x, ldx, y, ldy, plane, M, N, scale, scale_dev, seq_out, seq_in);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_absmax(const float* x, size_t n, float* out_dev, hipStream_t stream) {
    PF_HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(float), stream));
    const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, x, n, reinterpret_cast<unsigned*>(out_dev));
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_rowl1_bound(const float* W, int rows, int cols, int ld, const float* bias, float in_bound, float* out_dev,
                       hipStream_t stream) {
    PF_HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(float), stream));
    hipLaunchKernelGGL(rowl1_bound_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, W, rows, cols, ld, bias, in_bound,
                       reinterpret_cast<unsigned*>(out_dev));
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

// QKV / KV form: the persistent 256 x 128 shape (gemm_f16x2_ps.hip) takes it where whole rounds of 256 x 256 blocks fit the row count
// badly (the last round less than ~ half full: SenseVoice's 128 x 10 s batch, M = 22 528 -> 528 blocks = 2.06 rounds): its 256 x 128
// tiles quantise twice as finely. Same-call A/B inside the engine (profiles/r06i_ab_qkv_on_ps.txt): SenseVoiceSmall 29 412 -> 29 852
// audio-s/s; the headline shape (768 blocks = 3 whole rounds) is left alone (50.59 / 50.68 ms with the form forced on). Same bits either
// way. PF_QKV_PS=0 / 1 (environment) forces never / always for A/B runs.
static bool qkv_form_prefers_ps(const Gemm2Args& a) {
    static const int mode = [] { const char* e = getenv("PF_QKV_PS"); return e ? atoi(e) : 2; }();
    if (mode == 0 || !gemm_f16x2_ps_ok(a)) return false;
    if (mode == 1) return true;
    const int n_cu = device_cu_count();
    const long blocks = (long)ceil_div(a.M, 256) * (a.N / 256);
    if (blocks < n_cu) return false;            // (small problems: the 128 x 128 shapes below)
    const double rounds = (double)blocks / n_cu, whole = (double)((blocks + n_cu - 1) / n_cu);
    return whole / rounds > 1.15;
}

int gemm_f16x2_argmax_parts(int M, int N) { (void)M; return 2 * ceil_div(N, 256); }

int launch_gemm_f16x2(const Gemm2Args& a, hipStream_t stream) {
    PF_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm_f16x2: empty problem");
    if (a.ksplit > 1) {
        PF_REQUIRE((a.C || a.C2 || a.ln_g) && a.part && !a.amax_val && a.qkv_D <= 0 && a.a_kstep <= 0 && a.w_kstep <= 0 &&
                   a.K % (32 * a.ksplit) == 0 && a.N % 4 == 0 && ((uintptr_t)a.part & 15) == 0,
                   "gemm_f16x2: the split-K form needs K % (32 ksplit) == 0 and a partial buffer [ksplit][M][N]");
        if (a.C2) PF_REQUIRE(!a.ln_g && a.ldc2 % 4 == 0 && a.c_plane % 4 == 0 && ((uintptr_t)a.C2 & 7) == 0, "gemm_f16x2: split-K plane output alignment");
        else PF_REQUIRE((a.ln_g && !a.C) || (a.ldc % 4 == 0 && ((uintptr_t)a.C & 15) == 0), "gemm_f16x2: split-K output alignment");
        PF_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.a_plane % 8 == 0 && a.w_plane % 8 == 0 && ((uintptr_t)a.A & 15) == 0 &&
                   ((uintptr_t)a.W & 15) == 0 && (a.K / a.ksplit) % 8 == 0, "gemm_f16x2: operand alignment");
        if (a.bias) PF_REQUIRE(((uintptr_t)a.bias & 15) == 0, "gemm_f16x2: bias alignment");
        if (a.R1) PF_REQUIRE(a.ldr1 % 4 == 0 && ((uintptr_t)a.R1 & 15) == 0, "gemm_f16x2: R1 alignment");
        if (a.R2) PF_REQUIRE(a.ldr2 % 4 == 0 && ((uintptr_t)a.R2 & 15) == 0, "gemm_f16x2: R2 alignment");
        Gemm2Args p = a;
        p.ksplit = 0; p.kslices = a.ksplit; p.K = a.K / a.ksplit;
        p.C = a.part; p.ldc = a.N; p.C2 = nullptr; p.bias = nullptr; p.relu = 0; p.R1 = nullptr; p.R2 = nullptr; p.tile = 0;
        int rc = launch_tile<2, 2, 0, 0, 0, 0, 2>(p, stream);            // 128 x 128 blocks, two workgroups per CU
        if (rc) return rc;
        if (a.ln_g) {
            PF_REQUIRE(a.ln_b && a.ln_y, "gemm_f16x2: the split-K + LayerNorm form needs beta and an output");
            return launch_splitk_reduce_ln(a.part, a.ksplit, (size_t)a.M * a.N, a.M, a.N, a.bias, a.relu, a.R1, a.ldr1, a.R2, a.ldr2, a.C, a.ldc, a.ln_g, a.ln_b,
                                           a.ln_eps, a.ln_y, a.ln_ldy, a.ln_out, a.ln_plane, a.ln_oscale, stream);
        }
        const size_t total = (size_t)a.M * (a.N >> 2);
        const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, a.part, a.ksplit, (size_t)a.M * a.N, a.M, a.N,
                           a.bias, a.relu, a.R1, a.ldr1, a.R2, a.ldr2, a.C, a.ldc, a.C2, a.ldc2, a.c_plane, a.cscale);
        PF_HIP_TRY(hipGetLastError());
        return 0;
    }
    if (a.amax_val) {
        PF_REQUIRE(a.amax_idx && a.amax_ld >= gemm_f16x2_argmax_parts(a.M, a.N) && !a.R1 && !a.R2 && !a.relu && a.qkv_D <= 0,
                   "gemm_f16x2: the arg-max form takes bias only and needs amax_ld >= 2 ceil(N / 256)");
        PF_REQUIRE(a.K % 32 == 0 && a.lda % 8 == 0 && a.ldw % 8 == 0 && ((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0,
                   "gemm_f16x2: operand alignment");
        return launch_tile<2, 4, 0, 3, 0, 0>(a, stream);
    }
    PF_REQUIRE(a.K % 32 == 0, "gemm_f16x2: K must be a multiple of 32 (pad the planes with zeros)");
    PF_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.a_plane % 8 == 0 && a.w_plane % 8 == 0, "gemm_f16x2: operand strides % 8");
    PF_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm_f16x2: operands must be 16-B aligned");
    PF_REQUIRE(a.N % 4 == 0, "gemm_f16x2: N % 4");
    if (a.C2 && a.qkv_D <= 0) PF_REQUIRE(a.ldc2 % 4 == 0 && a.c_plane % 4 == 0 && ((uintptr_t)a.C2 & 7) == 0, "gemm_f16x2: plane output alignment");
    else if (a.qkv_D <= 0 || a.C) PF_REQUIRE(a.C && a.ldc % 4 == 0 && ((uintptr_t)a.C & 15) == 0, "gemm_f16x2: output alignment");
    if (a.bias) PF_REQUIRE(((uintptr_t)a.bias & 15) == 0, "gemm_f16x2: bias alignment");
    if (a.R1) PF_REQUIRE(a.ldr1 % 4 == 0 && ((uintptr_t)a.R1 & 15) == 0, "gemm_f16x2: R1 alignment");
    if (a.R2) PF_REQUIRE(a.ldr2 % 4 == 0 && ((uintptr_t)a.R2 & 15) == 0, "gemm_f16x2: R2 alignment");
    const int mode = (a.R1 ? 1 : 0) | (a.R2 ? 2 : 0);
    if (a.qkv_D > 0) {
        PF_REQUIRE(mode == 0 && !a.relu && a.N == (a.kv_form ? 2 : 3) * a.qkv_D && a.qkv_D % 256 == 0 && a.M % 16 == 0,
                   "gemm_f16x2: the QKV / KV form needs N == 3 D (2 D), D % 256 == 0, M % 16 == 0, no residuals");
        PF_REQUIRE((a.kv_form || (a.Qp && a.C)) && a.Kp && a.VT && a.ldvt % 8 == 0 && a.vt_plane % 8 == 0 && a.qk_plane % 8 == 0 &&
                   ((uintptr_t)a.Qp & 15) == 0 && ((uintptr_t)a.Kp & 15) == 0 && ((uintptr_t)a.VT & 15) == 0,
                   "gemm_f16x2: QKV outputs");
    }
    // the four-wave 256 x 256 shape (gemm_f16x2_w4.hip): tile 7; bits 4.. = its measurement builds
    // the persistent wave-specialised shape (gemm_f16x2_ps.hip): tile 10; shapes it does not take are chosen by shape instead
    if ((a.tile & 15) == 10 || (a.tile & 15) == 12) {         // 10: the persistent shape; 12: its finisher form (measurement library only)
        if (gemm_f16x2_ps_ok(a)) return launch_gemm_f16x2_ps(a, stream);
        Gemm2Args b = a;
        b.tile = 0;
        return launch_gemm_f16x2(b, stream);
    }
    if ((a.tile & 15) == 7) {
        if (gemm_f16x2_w4_ok(a)) return launch_gemm_f16x2_w4(a, a.tile >> 4, stream);
        Gemm2Args b = a;                    // a shape the four-wave block does not take (N % 256 != 0, K < 64): chosen by shape instead
        b.tile = 0;
        return launch_gemm_f16x2(b, stream);
    }
    // Tile ids of the product: 0 by shape, 1 / 2 the eight-wave 256 x 128 / 256 x 256 shapes, 3 the four-wave 128 x 128 shape, 7 the
    // four-wave 256 x 256 shape, 10 the persistent 256 x 128 shape. The measured-and-off shapes -- 5 (128 x 256, two workgroups per CU,
    // counted waits), 6 (deep ring), 8 / 9 (round 2's k-step order, static priorities), the ablation builds -- are in the measurement
    // library only (`make measure`; their numbers: profiles/r03y*, r04_ffn*, r05s, r05w, docs/history).
#if !defined(PF_MEASUREMENT_KERNELS)
    PF_REQUIRE(a.tile == 0 || a.tile == 1 || a.tile == 2 || a.tile == 3 || a.tile == 7 || (a.tile & 15) == 10,
               "gemm_f16x2: this tile id is a measurement shape (libparaformer_hip_measure.so, make measure)");
#endif
    if (a.qkv_D > 0) {
#if defined(PF_MEASUREMENT_KERNELS)
        if (a.tile == 5) return launch_pair<0, 2>(a, stream);
        if (a.tile == 6) return launch_ring<0, 2>(a, stream);
#endif
        if (a.tile == 3) return launch_tile<2, 2, 0, 2, 0, 0, 2>(a, stream);     // 128 x 128, four waves, two workgroups per CU
        if (a.tile == 0 && qkv_form_prefers_ps(a)) return launch_gemm_f16x2_ps(a, stream);
        if (a.tile == 0) {
            // Shape by the row count, in units of one round of 256 x 256 blocks (tools/bench_r03.py `qkvsplit`, N = 1536):
            //   256 x 256 only: whole rounds -- 528 blocks (M = 22 528) cost 3, not 2.06;
            //   128 x 128 (four waves, two workgroups per CU, out of phase): 0.6 per round of 2 n_cu blocks, and fractional
            //     rounds cost fractions -- 134 us instead of 154 at M = 22 528, 212 instead of 178 at M = 32 768;
            //   head / tail split: the whole rounds as 256 x 256 blocks, the rows behind them as 128 x 128 blocks in a second
            //     launch (+0.65: a launch boundary and one small round) -- 198 us instead of 218 at M = 33 536.
            // Row ranges are independent and all shapes give the same bits (tested), so the choice may depend on M.
            const int n_cu = device_cu_count();
            const int nN = a.N / 256, m_tiles = ceil_div(a.M, 256);
            const long blocks = (long)m_tiles * nN, full = blocks / n_cu, rem = blocks % n_cu;
            const double cost_wide = (double)(full + (rem ? 1 : 0));
            const double r_small = (double)ceil_div(a.M, 128) * (a.N / 128) / (2.0 * n_cu);
            const double cost_small = 0.6 * (r_small > 1.0 ? r_small : 1.0);
            const int head_tiles = (int)(full * n_cu / nN);
            const int tail_rows = a.M - head_tiles * 256;
            double cost_split = 1e30;
            if (full >= 2 && rem > 0 && tail_rows > 0 && (long)ceil_div(tail_rows, 128) * (a.N / 128) <= 2L * n_cu) cost_split = (double)full + 0.65;
            if (cost_small < cost_wide && cost_small <= cost_split) return launch_tile<2, 2, 0, 2, 0, 0, 2>(a, stream);
            if (cost_split < cost_wide) {
                Gemm2Args h = a, t = a;
                const size_t r0 = (size_t)head_tiles * 256;
                h.M = (int)r0;
                t.M = tail_rows;
                t.A = a.A + r0 * a.lda;
                if (a.C) t.C = a.C + r0 * a.ldc;
                if (a.Qp) t.Qp = a.Qp + r0 * a.qkv_D;
                t.Kp = a.Kp + r0 * a.qkv_D;
                t.VT = a.VT + r0;
                const int rc = launch_tile<2, 4, 0, 2, 0, 0>(h, stream);
                return rc ? rc : launch_tile<2, 2, 0, 2, 0, 0, 2>(t, stream);
            }
        }
#if defined(PF_MEASUREMENT_KERNELS)
        if (a.tile == 8) return launch_tile<2, 4, 0, 2, 0, 2>(a, stream);       // (A/B: round 2's schedule)
        if (a.tile == 9) return launch_tile<2, 4, 0, 2, 0, 3>(a, stream);
#endif
        return launch_tile<2, 4, 0, 2, 0, 0>(a, stream);
    }
    if (a.C2) {
        PF_REQUIRE(mode == 0, "gemm_f16x2: the plane output has no residual form");
#if defined(PF_MEASUREMENT_KERNELS)
        if (a.tile == 5 && a.N % 256 == 0) return launch_pair<0, 1>(a, stream);
        if (a.tile == 6 && a.N % 256 == 0) return launch_ring<0, 1>(a, stream);
#endif
        return launch_one<0, 1>(a, stream);
    }
#if defined(PF_MEASUREMENT_KERNELS)
    if (a.tile == 5 && a.N % 256 == 0) {
        switch (mode) {
            case 0: return launch_pair<0, 0>(a, stream);
            case 1: return launch_pair<1, 0>(a, stream);
            case 2: return launch_pair<2, 0>(a, stream);
            default: return launch_pair<3, 0>(a, stream);
        }
    }
    if (a.tile == 6 && a.N % 256 == 0) {
        switch (mode) {
            case 0: return launch_ring<0, 0>(a, stream);
            case 1: return launch_ring<1, 0>(a, stream);
            case 2: return launch_ring<2, 0>(a, stream);
            default: return launch_ring<3, 0>(a, stream);
        }
    }
#endif
    switch (mode) {
        case 0: return launch_one<0, 0>(a, stream);
        case 1: return launch_one<1, 0>(a, stream);
        case 2: return launch_one<2, 0>(a, stream);
        default: return launch_one<3, 0>(a, stream);
    }
}

}  // namespace pf
