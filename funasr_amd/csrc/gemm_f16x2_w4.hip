// The f16x2 tile GEMM (gemm_f16x2.hip: C = epilogue(A[M,K] * W[N,K]^T), both operands as two fp16 planes, three
// v_mfma_f32_32x32x16_f16 products per operand pair into one fp32 accumulator) as a FOUR-wave 256 x 256 block: one wave per
// SIMD, each with the SIMD's whole 512-entry register file -- a 128 x 128 output quadrant = 16 accumulator tiles = all 256
// AGPRs, and TWO 16-deep k-steps' operand fragments (2 x 64 VGPRs) alive at a time. Same call sites as the other shapes
// (funasr/models/transformer/positionwise_feed_forward.py:14-34, funasr/models/sanm/attention.py:256,306); the same products
// in the same k order per output element and the shared epilogue (gemm_f16x2_epilogue.h), so results are BITWISE those of
// every other shape (tested: tests/test_kernels_f16x2_gpu.py, tools/bench_w4.py parity). Two wave grids over the same K loop:
//   2 x 2: the 256 x 256 tile form (gemm_f16x2.hip's epilogue: fp32 / planes / QKV form), Gemm2Args.tile 7;
//   1 x 4: the 128 x 512 full-row form (gemm_f16x2_row.hip's epilogue: residuals, FSMN memory block, LayerNorm, planes; run once
//          per 64-row half of the block), GemmRowArgs.block_rows 130.
//
// Why another shape (round 5). The eight-wave shapes read ALL fragments of a 32-deep stage right behind the stage's barrier,
// with both waves of every SIMD waiting for them at the same time: 192 ds_read_b128 (768 LDS cycles) per 3072 cycles of
// matrix work run uncovered -- the loop alone reaches 0.62 of the matrix rate (tools/abl_gemm2.py). Here
//   * LDS reads per MFMA drop by a third (16 ds_read_b128 per 48 MFMAs against 12 per 24) and the NEXT k-step's fragments
//     are read under the current k-step's MFMAs, one read per two MFMAs, placed by hand (every instruction of the K loop is
//     issued from volatile asm in source order: with one wave per SIMD nothing else fills the matrix pipe's shadow). The loop
//     alone (no DMA, no epilogue; M = 32768, w_1) runs 92 us on zero operands = 0.90 of the 2.5 PFLOP/s peak, and 124-139 us
//     on random operands = the power-limited rate of this instruction mix (tools/micro/mfma_peak.hip: 1647 TFLOP/s at
//     1.78 GHz / 1240 W; profiles/r05a_w4_first_run.jsonl);
//   * operand staging as in the eight-wave shapes: two 64-KB stages (32 deep, 64-B LDS rows, chunk swizzle c ^ ((r >> 2) & 3)),
//     every wave issues 16 of a stage's 64 one-KB LDS-DMA pieces, one per three MFMA gaps, during the k-step in which the
//     buffer falls free, and waits for them -- vmcnt(0): the only exact wait on LDS-DMA (DESIGN, round 4) -- a k-step later;
//     ONE barrier per stage, in the middle of it (it publishes the next stage and frees the buffer just read);
//   * measured and dropped on the way (profiles/r05a..c): a ring of four / five 16-deep stages with one owner wave per stage
//     (exact waits with three / four stages in flight). Its cover was never the problem -- four and five stages took the same
//     time -- its cost was the owner's issue burst (32 pieces in one k-step from ONE wave: whatever does not fit an MFMA's
//     shadow idles that SIMD's matrix pipe, and the other three waves wait for the owner at the step's barrier: +50 us on a
//     92-us loop) and the 32-B row slices of a 16-deep stage (twice the address-coalescer work per byte: -12 us with a
//     contiguous pattern). Uniform issue by all waves is what the exact wait allows only with ONE stage in flight.
#include "common.h"
#include "gemm_f16x2_epilogue.h"
#include "gemm_f16x2_row_epilogue.h"

namespace pf {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// wave grid GM x GN (GM GN == 4), every wave a 128 x 128 quadrant. A 32-deep stage = [A hi | A lo | W hi | W lo] of 64-B rows
template <int GM_, int GN_> struct W4Geo {
    static constexpr int GM = GM_, GN = GN_;
    static constexpr int BM = GM * 128, BN = GN * 128;
    static constexpr int A_PLANE_B = BM * 64, W_PLANE_B = BN * 64;
    static constexpr int STAGE_B = 2 * (A_PLANE_B + W_PLANE_B);      // 64 KB (2 x 2) / 80 KB (1 x 4)
    static constexpr int NPW = STAGE_B / 1024 / 4;                   // 1-KB pieces per wave and stage: 16 / 20
    static constexpr int TA = BM / 64, TW = BN / 64;                 // this wave's pieces per A / W plane: 4 / 2 and 4 / 8
    static constexpr int LDS_B = 2 * STAGE_B;
    static_assert(GM * GN == 4 && NPW == 2 * (TA + TW), "four waves");
};

struct W4Frags { f16x8 al[4], ah[4], bh[4], bl[4]; };      // A lo / hi tiles (rows), W hi / lo tiles (columns)

__device__ __forceinline__ void w4_mfma(floatx16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
template <int OFF> __device__ __forceinline__ void w4_read(f16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// every fragment read of this wave has returned; the operands tie the wait to the registers it covers
__device__ __forceinline__ void w4_reads_done(W4Frags& f) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.al[0]), "+v"(f.al[1]), "+v"(f.al[2]), "+v"(f.al[3]), "+v"(f.ah[0]), "+v"(f.ah[1]), "+v"(f.ah[2]), "+v"(f.ah[3]),
                   "+v"(f.bh[0]), "+v"(f.bh[1]), "+v"(f.bh[2]), "+v"(f.bh[3]), "+v"(f.bl[0]), "+v"(f.bl[1]), "+v"(f.bl[2]), "+v"(f.bl[3]));
}
// the compiler does not see an MFMA in w4_mfma: the wait states between the last product (or the zero fill) and the next
// reader of an accumulator are spent here (16 passes of a 32 x 32 x 16 product + the write-back)
__device__ __forceinline__ void w4_settle(floatx16 (&acc)[4][4]) {
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                 : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]),
                   "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]), "+a"(acc[3][3]));
}
// one 1-KB LDS-DMA piece: uniform 64-bit base + 32-bit per-lane byte offset, destination lds_buf + LOFF. No branch, no compare:
// a piece costs the issuing wave ~17 cycles of scalar work and ~19 of VMEM issue that one MFMA's shadow has to hold
// (profiles/r05h_w4_piece_cost_probes.jsonl), so the stages that issue nothing are a separate copy of the loop body.
template <int LOFF, bool NT> __device__ __forceinline__ void w4_piece(const char* sbase, unsigned voff, unsigned lds_buf) {
    if constexpr (NT)
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(voff), "s"(sbase), "s"(lds_buf), "n"(LOFF) : "memory", "scc");
    else
        asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_buf), "n"(LOFF) : "memory", "scc");
    // (m0 is reserved: the compiler keeps nothing in it)
}

// The K loop of both forms: acc += A[m0 .., :] W[n0 .., :]^T over p.K (a multiple of 32, >= 64), this wave's quadrant (wr, wc).
// NODMA (timing only): 1 = no LDS-DMA pieces and no waits for them (the MFMA + fragment-read loop alone)
template <class G, int NODMA, bool A_NT>
__device__ __forceinline__ void w4_kloop(const unsigned short* A, int lda, size_t a_plane, const unsigned short* W, int ldw, size_t w_plane,
                                         int M, int N, int K, int m0, int n0, unsigned char* smem, int wave, int wr, int wc, int lane,
                                         floatx16 (&acc)[4][4]) {
    const int hh = lane >> 5, idx = lane & 31;
    // ---- DMA sources. A stage is STAGE_B / 1024 pieces of 16 rows x 64 B of one plane (lane l -> row l / 4, physical chunk l % 4
    //      <- the logical chunk the read-side swizzle expects there), linear in LDS; wave w issues pieces w + 4 i: the A planes'
    //      row groups 16 (w + 4 t) .. first, then the W planes'
    unsigned va[G::TA], vw[G::TW];
    {
        const int prow = lane >> 2;
        const unsigned chunkb = (unsigned)(((lane & 3) ^ ((prow >> 2) & 3)) * 16);
#pragma unroll
        for (int t = 0; t < G::TA; ++t) {
            int row = m0 + 16 * (wave + 4 * t) + prow;
            row = row < M ? row : M - 1;
            va[t] = (unsigned)row * (unsigned)lda * 2u + chunkb;
        }
#pragma unroll
        for (int t = 0; t < G::TW; ++t) {
            int col = n0 + 16 * (wave + 4 * t) + prow;
            col = col < N ? col : N - 1;
            vw[t] = (unsigned)col * (unsigned)ldw * 2u + chunkb;
        }
    }
    const char* const a_hi = reinterpret_cast<const char*>(A);
    const char* const a_lo = a_hi + a_plane * 2;
    const char* const w_hi = reinterpret_cast<const char*>(W);
    const char* const w_lo = w_hi + w_plane * 2;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const unsigned ldsw = lds0 + (unsigned)wave * 1024;
    // this wave's piece I (0 .. NPW - 1) of stage `js` -> the stage buffer at byte address `buf` (+ wave KB)
    auto piece = [&](auto I, int js, unsigned buf) {
        constexpr int i = decltype(I)::value;
        if constexpr (NODMA == 1) return;
        const size_t ko = (size_t)js * 64;
        if constexpr (i < G::TA) w4_piece<4096 * i, A_NT>(a_hi + ko, va[i], buf);
        else if constexpr (i < 2 * G::TA) w4_piece<G::A_PLANE_B + 4096 * (i - G::TA), A_NT>(a_lo + ko, va[i - G::TA], buf);
        else if constexpr (i < 2 * G::TA + G::TW) w4_piece<2 * G::A_PLANE_B + 4096 * (i - 2 * G::TA), false>(w_hi + ko, vw[i - 2 * G::TA], buf);
        else w4_piece<2 * G::A_PLANE_B + G::W_PLANE_B + 4096 * (i - 2 * G::TA - G::TW), false>(w_lo + ko, vw[i - 2 * G::TA - G::TW], buf);
    };

    // ---- fragment addresses: lane (idx, hh) reads row idx of a 32-row tile, logical chunk 2 st + hh of k-step st
    const unsigned fsw = (unsigned)((idx >> 2) & 3);
    const unsigned fa0 = lds0 + (unsigned)((wr * 128 + idx) * 64);
    const unsigned fb0 = lds0 + 2 * G::A_PLANE_B + (unsigned)((wc * 128 + idx) * 64);
    // read R (0..15) of a k-step: A lo tiles, W hi tiles (the first product's operands), A hi tiles, W lo tiles
    auto frag_read = [&](auto Rr, W4Frags& f, unsigned fa, unsigned fb) {
        constexpr int r = decltype(Rr)::value;
        if constexpr (r < 4) w4_read<G::A_PLANE_B + (r & 3) * 2048>(f.al[r & 3], fa);
        else if constexpr (r < 8) w4_read<(r & 3) * 2048>(f.bh[r & 3], fb);
        else if constexpr (r < 12) w4_read<(r & 3) * 2048>(f.ah[r & 3], fa);
        else w4_read<G::W_PLANE_B + (r & 3) * 2048>(f.bl[r & 3], fb);
    };
    auto coff = [&](int st) { return (unsigned)(((2 * st + hh) ^ fsw) * 16); };

    const int ns = K / 32;                        // stages (>= 2)

    // ---- prologue: stage 0 lands and is published; its first k-step is read; stage 1 is on its way
    [&]<int... I>(std::integer_sequence<int, I...>) { (piece(std::integral_constant<int, I>{}, 0, ldsw), ...); }(std::make_integer_sequence<int, G::NPW>{});
    if constexpr (NODMA == 0) glds_wait_all();
    __builtin_amdgcn_s_barrier();
    W4Frags f0, f1;
    [&]<int... I>(std::integer_sequence<int, I...>) { (frag_read(std::integral_constant<int, I>{}, f0, fa0 + coff(0), fb0 + coff(0)), ...); }(std::make_integer_sequence<int, 16>{});
    [&]<int... I>(std::integer_sequence<int, I...>) { (piece(std::integral_constant<int, I>{}, 1, ldsw + G::STAGE_B), ...); }(std::make_integer_sequence<int, G::NPW>{});
    w4_reads_done(f0);

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    w4_settle(acc);

    // ---- one 16-deep k-step: 48 MFMAs on `x`; the next k-step's fragments are read into `y` from (fa, fb), one read per two
    //      MFMAs; ISSUE: this wave's pieces of stage `js` go out, spread evenly over the 48 MFMA gaps
    auto kstep = [&](auto Issue, W4Frags& x, W4Frags& y, unsigned fa, unsigned fb, int js, unsigned buf) {
        constexpr bool ISSUE = decltype(Issue)::value;
        [&]<int... Gp>(std::integer_sequence<int, Gp...>) {
            ([&] {
                constexpr int g = Gp, P = g >> 4, t = g & 15, i = t >> 2, j = t & 3;
                // the two small products first, hi * hi last -- the order of every other shape
                if constexpr (P == 0) w4_mfma(acc[i][j], x.al[i], x.bh[j]);
                else if constexpr (P == 1) w4_mfma(acc[i][j], x.ah[i], x.bl[j]);
                else w4_mfma(acc[i][j], x.ah[i], x.bh[j]);
                if constexpr ((g & 1) == 0 && g < 32) frag_read(std::integral_constant<int, (g >> 1)>{}, y, fa, fb);
                // piece q goes behind MFMA (48 q) / NPW + 1: g carries piece q iff that holds for q = ((g - 1) NPW + 47) / 48
                if constexpr (ISSUE && g >= 1) {
                    constexpr int q = ((g - 1) * G::NPW + 47) / 48;
                    if constexpr (q < G::NPW && (48 * q) / G::NPW + 1 == g) piece(std::integral_constant<int, q>{}, js, buf);
                }
            }(), ...);
        }(std::make_integer_sequence<int, 48>{});
        w4_reads_done(y);
    };
    // stage j (buffer j & 1): first k-step from f0 while its second k-step is read into f1; everything in flight (stage j + 1)
    // lands; the barrier publishes stage j + 1 and frees buffer j & 1 (its last reads have returned); second k-step from f1
    // while stage j + 1's first k-step is read into f0 and stage j + 2 goes into the freed buffer
    auto stage = [&](auto Issue, int j) {
        const unsigned cur = (unsigned)(j & 1) * G::STAGE_B, nxt = G::STAGE_B - cur;
        kstep(std::false_type{}, f0, f1, fa0 + cur + coff(1), fb0 + cur + coff(1), 0, 0u);
        if constexpr (NODMA == 0) glds_wait_all();
        __builtin_amdgcn_s_barrier();
        kstep(Issue, f1, f0, fa0 + nxt + coff(0), fb0 + nxt + coff(0), j + 2, ldsw + cur);
    };
    for (int j = 0; j < ns - 2; ++j) stage(std::true_type{}, j);
    stage(std::false_type{}, ns - 2);
    stage(std::false_type{}, ns - 1);             // (its second k-step reads one k-step past the end into f0: never multiplied)
    w4_settle(acc);
}

// EABL: the shared epilogue's measurement switch (0 product, 1 no global stores, 2 no epilogue)
template <int MODE, int OUT, int EABL, int NODMA>
__global__ __launch_bounds__(256, 1) void gemm_f16x2_w4_kernel(Gemm2Args p, int nM, int nN) {
    typedef W4Geo<2, 2> G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int L = blockIdx.x;
    const int xcd = L & 7, j8 = L >> 3;
    const int mblk = (j8 / nN) * 8 + xcd, nblk = j8 % nN;
    if (mblk >= nM) return;
    const int m0 = mblk * G::BM, n0 = nblk * G::BN;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    floatx16 acc[4][4];
    w4_kloop<G, NODMA, false>(p.A, p.lda, p.a_plane, p.W, p.ldw, p.w_plane, p.M, p.N, p.K, m0, n0, smem, wave, wr, wc, lane, acc);
    gemm2_epilogue<4, 4, 4, MODE, OUT, EABL>(p, acc, smem, m0, n0, nblk, wave, wr, wc, lane);
}

#if defined(PF_MEASUREMENT_KERNELS)
// full-row form (N == 512): one workgroup = 128 complete rows, wave w = columns 128 w ..; the row epilogue runs once per 64-row half
template <int MODE, bool LN, bool A_NT>
__global__ __launch_bounds__(256, 1) void gemm_f16x2_w4_row_kernel(GemmRowArgs p) {
    typedef W4Geo<1, 4> G;
    static_assert(G::LDS_B == RW_LDS_B && G::BM == RW_BM && G::BN == RW_BN, "the row epilogue's LDS plan");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m0 = blockIdx.x * G::BM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    floatx16 acc[4][4];
    w4_kloop<G, 0, A_NT>(p.A, p.lda, p.a_plane, p.W, p.ldw, p.w_plane, p.M, G::BN, p.K, m0, 0, smem, wave, 0, wave, lane, acc);
    gemm2_row_epilogue<MODE, LN, 256, 4, 0>(p, acc, smem, m0, tid, wave, 0, wave, lane, true);
    gemm2_row_epilogue<MODE, LN, 256, 4, 2>(p, acc, smem, m0, tid, wave, 1, wave, lane, false);
}

#endif

template <int MODE, int OUT, int EABL = 0, int NODMA = 0>
int launch_w4(const Gemm2Args& a, hipStream_t stream) {
    typedef W4Geo<2, 2> G;
    static PerDeviceOnce configured;
    if (!configured.done()) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_w4_kernel<MODE, OUT, EABL, NODMA>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_B));
        configured.mark();
    }
    const int nM = ceil_div(a.M, G::BM), nN = ceil_div(a.N, G::BN);
    const int nMpad = (nM + 7) / 8 * 8;
    hipLaunchKernelGGL((gemm_f16x2_w4_kernel<MODE, OUT, EABL, NODMA>), dim3((unsigned)nMpad * nN), dim3(256), G::LDS_B, stream, a, nM, nN);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

#if defined(PF_MEASUREMENT_KERNELS)
template <int MODE, bool LN, bool A_NT>
int launch_w4_row_t(const GemmRowArgs& a, hipStream_t stream) {
    typedef W4Geo<1, 4> G;
    static PerDeviceOnce configured;
    if (!configured.done()) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_w4_row_kernel<MODE, LN, A_NT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_B));
        configured.mark();
    }
    hipLaunchKernelGGL((gemm_f16x2_w4_row_kernel<MODE, LN, A_NT>), dim3((unsigned)ceil_div(a.M, G::BM)), dim3(256), G::LDS_B, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}
template <int MODE, bool LN>
int launch_w4_row_m(const GemmRowArgs& a, hipStream_t stream) {
    return (a.a_nt & 1) ? launch_w4_row_t<MODE, LN, true>(a, stream) : launch_w4_row_t<MODE, LN, false>(a, stream);
}

#endif

}  // namespace

bool gemm_f16x2_w4_ok(const Gemm2Args& a) {
    // rows / columns are clamped in the DMA sources and masked in the epilogue like the other shapes; 32-bit per-lane byte offsets
    return a.K % 32 == 0 && a.K >= 64 && a.N % 256 == 0 && a.kslices <= 1 && a.ksplit <= 1 && !a.amax_val && a.a_kstep <= 0 && a.w_kstep <= 0 &&
           (size_t)a.M * (size_t)a.lda * 2 < (1ull << 32) && (size_t)a.N * (size_t)a.ldw * 2 < (1ull << 32);
}

// abl (measurement only): 0 product, 1 no global stores, 2 no epilogue, 3 no epilogue and no operand DMA (the loop alone)
int launch_gemm_f16x2_w4(const Gemm2Args& a, int abl, hipStream_t stream) {
    PF_REQUIRE(gemm_f16x2_w4_ok(a), "gemm_f16x2 (four-wave shape): needs K % 32 == 0, K >= 64, N % 256 == 0 and operands below 4 GB");
    const int mode = (a.R1 ? 1 : 0) | (a.R2 ? 2 : 0);
    if (a.qkv_D > 0) return launch_w4<0, 2>(a, stream);
    if (a.C2) {
        PF_REQUIRE(mode == 0, "gemm_f16x2: the plane output has no residual form");
#if defined(PF_MEASUREMENT_KERNELS)
        if (abl == 1) return launch_w4<0, 1, 1>(a, stream);
#endif
        return launch_w4<0, 1>(a, stream);
    }
#if defined(PF_MEASUREMENT_KERNELS)
    if (abl == 1) return launch_w4<0, 0, 1>(a, stream);
    if (abl == 2) return launch_w4<0, 0, 2>(a, stream);
    if (abl == 3) return launch_w4<0, 0, 2, 1>(a, stream);
#else
    PF_REQUIRE(abl == 0, "gemm_f16x2 (four-wave shape): the ablation builds are in the measurement library (make measure)");
#endif
    switch (mode) {
        case 0: return launch_w4<0, 0>(a, stream);
        case 1: return launch_w4<1, 0>(a, stream);
        case 2: return launch_w4<2, 0>(a, stream);
        default: return launch_w4<3, 0>(a, stream);
    }
}

bool gemm_f16x2_w4_row_ok(const GemmRowArgs& a) {
    return a.N == 512 && a.K % 32 == 0 && a.K >= 64 && (size_t)a.M * (size_t)a.lda * 2 < (1ull << 32) && (size_t)a.ldw * 1024 < (1ull << 32);
}

// the four-wave full-row form (measurement library only); the caller (launch_gemm_f16x2_row) has checked the arguments
#if defined(PF_MEASUREMENT_KERNELS)
int launch_gemm_f16x2_w4_row(const GemmRowArgs& a, hipStream_t stream) {
    PF_REQUIRE(gemm_f16x2_w4_row_ok(a), "gemm_f16x2_row (four-wave shape): needs N == 512, K % 32 == 0, K >= 64 and operands below 4 GB");
    const bool ln = a.ln_g != nullptr;
    if (a.fs_v) return a.R2 ? launch_w4_row_m<6, true>(a, stream) : launch_w4_row_m<4, true>(a, stream);
    const int mode = (a.R1 ? 1 : 0) | (a.R2 ? 2 : 0);
    switch (mode * 2 + (ln ? 1 : 0)) {
        case 0: return launch_w4_row_m<0, false>(a, stream);
        case 1: return launch_w4_row_m<0, true>(a, stream);
        case 2: return launch_w4_row_m<1, false>(a, stream);
        case 3: return launch_w4_row_m<1, true>(a, stream);
        case 4: return launch_w4_row_m<2, false>(a, stream);
        case 5: return launch_w4_row_m<2, true>(a, stream);
        case 6: return launch_w4_row_m<3, false>(a, stream);
        default: return launch_w4_row_m<3, true>(a, stream);
    }
}

#else
int launch_gemm_f16x2_w4_row(const GemmRowArgs&, hipStream_t) {
    set_error("gemm_f16x2_row (four-wave shape): measured and off -- in the measurement library only (make measure)");
    return -1;
}
#endif

}  // namespace pf
