// The f16x2 tile GEMM (gemm_f16x2.hip: C = epilogue(A[M,K] * W[N,K]^T), both operands as two fp16 planes, three
// v_mfma_f32_32x32x16_f16 products per operand pair into one fp32 accumulator) as a FOUR-wave 256 x 256 block: one wave per
// SIMD, each with the SIMD's whole 512-entry register file -- a 128 x 128 output quadrant = 16 accumulator tiles = all 256
// AGPRs, and TWO 16-deep k-steps' operand fragments (2 x 64 VGPRs) alive at a time. Same call sites as the other shapes
// (funasr/models/transformer/positionwise_feed_forward.py:14-34, funasr/models/sanm/attention.py:256,306); the same products
// in the same k order per output element and the shared epilogue (gemm_f16x2_epilogue.h), so results are BITWISE those of
// every other shape (tested: tests/test_kernels_f16x2_gpu.py, tools/bench_w4.py parity).
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

namespace pf {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int W4_PLANE_B = 256 * 64;              // one plane of a 32-deep stage: 256 rows x 64 B
constexpr int W4_STAGE_B = 4 * W4_PLANE_B;        // 64 KB: [A hi | A lo | W hi | W lo]
constexpr int W4_LDS_B = 2 * W4_STAGE_B;          // 128 KB
static_assert(4 * 32 * (4 * 32 + 4) * 4 <= W4_LDS_B, "epilogue slabs alias the stages");

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
// one 1-KB LDS-DMA piece: uniform 64-bit base + 32-bit per-lane byte offset, destination lds_buf + LOFF. `own` (uniform) == 0
// skips it: the branch lives INSIDE the statement, so the K loop stays one basic block for the compiler -- with a C++ branch
// per step variant the register allocator shuffled and spilled the 256 accumulation registers at every join (1500 dwords of
// scratch in the first build of this file).
// TIMING ONLY: the same statement with a plain load into a scratch register (KIND 4) or with no memory instruction at all (KIND 5):
// what of a piece's cost is the LDS-DMA instruction, what the scalar work around it
template <int LOFF, int KIND> __device__ __forceinline__ void w4_piece_probe(const char* sbase, unsigned voff, unsigned lds_buf, int own) {
    uint4 sink;
    if constexpr (KIND == 4)
        asm volatile("s_cmp_lg_u32 %5, 0\n\ts_cbranch_scc0 .Lw4_skipp_%=\n\ts_add_u32 m0, %3, %4\n\ts_nop 0\n\tglobal_load_dwordx4 %0, %1, %2\n.Lw4_skipp_%=:"
                     : "=v"(sink) : "v"(voff), "s"(sbase), "s"(lds_buf), "n"(LOFF), "s"(own) : "memory", "scc");
    else
        asm volatile("s_cmp_lg_u32 %4, 0\n\ts_cbranch_scc0 .Lw4_skipp_%=\n\ts_add_u32 m0, %2, %3\n\ts_nop 0\n.Lw4_skipp_%=:"
                     : : "v"(voff), "s"(sbase), "s"(lds_buf), "n"(LOFF), "s"(own) : "memory", "scc");
}
template <int LOFF> __device__ __forceinline__ void w4_piece(const char* sbase, unsigned voff, unsigned lds_buf, int own) {
    asm volatile("s_cmp_lg_u32 %4, 0\n\ts_cbranch_scc0 .Lw4_skip_%=\n\ts_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n.Lw4_skip_%=:"
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_buf), "n"(LOFF), "s"(own)
                 : "memory", "scc");   // (m0 is reserved: the compiler keeps nothing in it)
}
// EABL: the shared epilogue's measurement switch (0 product, 1 no global stores, 2 no epilogue); NODMA: no LDS-DMA pieces and
// no waits for them (the MFMA + fragment-read loop alone: tools/bench_w4.py)
template <int MODE, int OUT, int EABL, int NODMA, int PSP = 3>
__global__ __launch_bounds__(256, 1) void gemm_f16x2_w4_kernel(Gemm2Args p, int nM, int nN) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int L = blockIdx.x;
    const int xcd = L & 7, j8 = L >> 3;
    const int mblk = (j8 / nN) * 8 + xcd, nblk = j8 % nN;
    if (mblk >= nM) return;
    const int m0 = mblk * 256, n0 = nblk * 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int hh = lane >> 5, idx = lane & 31;

    // ---- DMA sources. A stage is 64 pieces of 1 KB = 16 rows x 64 B of one plane (lane l -> row l / 4, physical chunk l % 4
    //      <- the logical chunk the read-side swizzle expects there); wave w issues pieces w + 4 i, i = 0..15: plane i / 4,
    //      rows 16 (w + 4 (i % 4)) ..
    unsigned va[4], vw[4];
    {
        const int prow = lane >> 2;
        const unsigned chunkb = (unsigned)(((lane & 3) ^ ((prow >> 2) & 3)) * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int row = m0 + 16 * (wave + 4 * t) + prow;
            row = row < p.M ? row : p.M - 1;
            va[t] = (unsigned)row * (unsigned)p.lda * 2u + chunkb;
            int col = n0 + 16 * (wave + 4 * t) + prow;
            col = col < p.N ? col : p.N - 1;
            vw[t] = (unsigned)col * (unsigned)p.ldw * 2u + chunkb;
        }
    }
    const char* const a_hi = reinterpret_cast<const char*>(p.A);
    const char* const a_lo = a_hi + p.a_plane * 2;
    const char* const w_hi = reinterpret_cast<const char*>(p.W);
    const char* const w_lo = w_hi + p.w_plane * 2;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const unsigned ldsw = lds0 + (unsigned)wave * 1024;
    // this wave's piece I (0..15) of stage `js` -> the stage buffer at byte address `buf` (+ wave KB)
    auto piece = [&](auto I, int js, unsigned buf, int own) {
        constexpr int i = decltype(I)::value;
        if constexpr (NODMA == 1) return;
        const size_t ko = (size_t)js * 64;
        constexpr int LOFF = ((i >> 2) * 16 + 4 * (i & 3)) * 1024;
        if constexpr (NODMA >= 4) {
            if constexpr ((i >> 2) == 0) w4_piece_probe<LOFF, NODMA>(a_hi + ko, va[i & 3], buf, own);
            else if constexpr ((i >> 2) == 1) w4_piece_probe<LOFF, NODMA>(a_lo + ko, va[i & 3], buf, own);
            else if constexpr ((i >> 2) == 2) w4_piece_probe<LOFF, NODMA>(w_hi + ko, vw[i & 3], buf, own);
            else w4_piece_probe<LOFF, NODMA>(w_lo + ko, vw[i & 3], buf, own);
        } else if constexpr ((i >> 2) == 0) w4_piece<LOFF>(a_hi + ko, va[i & 3], buf, own);
        else if constexpr ((i >> 2) == 1) w4_piece<LOFF>(a_lo + ko, va[i & 3], buf, own);
        else if constexpr ((i >> 2) == 2) w4_piece<LOFF>(w_hi + ko, vw[i & 3], buf, own);
        else w4_piece<LOFF>(w_lo + ko, vw[i & 3], buf, own);
    };

    // ---- fragment addresses: lane (idx, hh) reads row idx of a 32-row tile, logical chunk 2 st + hh of k-step st
    const unsigned fsw = (unsigned)((idx >> 2) & 3);
    const unsigned fa0 = lds0 + (unsigned)((wr * 128 + idx) * 64);
    const unsigned fb0 = lds0 + 2 * W4_PLANE_B + (unsigned)((wc * 128 + idx) * 64);
    // read R (0..15) of a k-step: A lo tiles, W hi tiles (the first product's operands), A hi tiles, W lo tiles
    auto frag_read = [&](auto Rr, W4Frags& f, unsigned fa, unsigned fb) {
        constexpr int r = decltype(Rr)::value;
        if constexpr (r < 4) w4_read<W4_PLANE_B + (r & 3) * 2048>(f.al[r & 3], fa);
        else if constexpr (r < 8) w4_read<(r & 3) * 2048>(f.bh[r & 3], fb);
        else if constexpr (r < 12) w4_read<(r & 3) * 2048>(f.ah[r & 3], fa);
        else w4_read<W4_PLANE_B + (r & 3) * 2048>(f.bl[r & 3], fb);
    };
    auto coff = [&](int st) { return (unsigned)(((2 * st + hh) ^ fsw) * 16); };

    const int ns = p.K / 32;                      // stages

    // ---- prologue: stage 0 lands and is published; its first k-step is read; stage 1 is on its way
    [&]<int... I>(std::integer_sequence<int, I...>) { (piece(std::integral_constant<int, I>{}, 0, ldsw, 1), ...); }(std::make_integer_sequence<int, 16>{});
    if constexpr (NODMA == 0) glds_wait_all();
    __builtin_amdgcn_s_barrier();
    W4Frags f0, f1;
    [&]<int... I>(std::integer_sequence<int, I...>) { (frag_read(std::integral_constant<int, I>{}, f0, fa0 + coff(0), fb0 + coff(0)), ...); }(std::make_integer_sequence<int, 16>{});
    {
        const int own = __builtin_amdgcn_readfirstlane(1 < ns ? 1 : 0);
        [&]<int... I>(std::integer_sequence<int, I...>) { (piece(std::integral_constant<int, I>{}, 1, ldsw + W4_STAGE_B, own), ...); }(std::make_integer_sequence<int, 16>{});
    }
    w4_reads_done(f0);

    floatx16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    w4_settle(acc);

    // ---- one 16-deep k-step: 48 MFMAs on `x`; the next k-step's fragments are read into `y` from (fa, fb), one read per two
    //      MFMAs; ISSUE: this wave's 16 pieces of stage `js` go out, one per three MFMA gaps, if `own`
    auto kstep = [&](auto Issue, W4Frags& x, W4Frags& y, unsigned fa, unsigned fb, int js, unsigned buf, int own) {
        constexpr bool ISSUE = decltype(Issue)::value;
        [&]<int... G>(std::integer_sequence<int, G...>) {
            ([&] {
                constexpr int g = G, P = g >> 4, t = g & 15, i = t >> 2, j = t & 3;
                // the two small products first, hi * hi last -- the order of every other shape
                if constexpr (P == 0) w4_mfma(acc[i][j], x.al[i], x.bh[j]);
                else if constexpr (P == 1) w4_mfma(acc[i][j], x.ah[i], x.bl[j]);
                else w4_mfma(acc[i][j], x.ah[i], x.bh[j]);
                if constexpr ((g & 1) == 0 && g < 32) frag_read(std::integral_constant<int, (g >> 1)>{}, y, fa, fb);
                if constexpr (ISSUE && g % PSP == (PSP == 1 ? 0 : 1) && g / PSP < 16) piece(std::integral_constant<int, g / PSP>{}, js, buf, own);
            }(), ...);
        }(std::make_integer_sequence<int, 48>{});
        w4_reads_done(y);
    };
    // stage j (buffer j & 1): first k-step from f0 while its second k-step is read into f1; everything in flight (stage j + 1)
    // lands; the barrier publishes stage j + 1 and frees buffer j & 1 (its last reads have returned); second k-step from f1
    // while stage j + 1's first k-step is read into f0 and stage j + 2 goes into the freed buffer
    for (int j = 0; j < ns; ++j) {
        const unsigned cur = (unsigned)(j & 1) * W4_STAGE_B, nxt = W4_STAGE_B - cur;
        kstep(std::false_type{}, f0, f1, fa0 + cur + coff(1), fb0 + cur + coff(1), 0, 0u, 0);
        if constexpr (NODMA == 0) glds_wait_all();
        __builtin_amdgcn_s_barrier();
        const int own = __builtin_amdgcn_readfirstlane(j + 2 < ns ? 1 : 0);
        kstep(std::true_type{}, f1, f0, fa0 + nxt + coff(0), fb0 + nxt + coff(0), j + 2, ldsw + cur, own);
    }
    w4_settle(acc);
    gemm2_epilogue<4, 4, 4, MODE, OUT, EABL>(p, acc, smem, m0, n0, nblk, wave, wr, wc, lane);
}

template <int MODE, int OUT, int EABL = 0, int NODMA = 0, int PSP = 3>
int launch_w4(const Gemm2Args& a, hipStream_t stream) {
    static bool configured = false;
    if (!configured) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_w4_kernel<MODE, OUT, EABL, NODMA, PSP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_B));
        configured = true;
    }
    const int nM = ceil_div(a.M, 256), nN = ceil_div(a.N, 256);
    const int nMpad = (nM + 7) / 8 * 8;
    hipLaunchKernelGGL((gemm_f16x2_w4_kernel<MODE, OUT, EABL, NODMA, PSP>), dim3((unsigned)nMpad * nN), dim3(256), W4_LDS_B, stream, a, nM, nN);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

bool gemm_f16x2_w4_ok(const Gemm2Args& a) {
    // rows / columns are clamped in the DMA sources and masked in the epilogue like the other shapes; 32-bit per-lane byte offsets
    return a.K % 32 == 0 && a.N % 256 == 0 && a.kslices <= 1 && a.ksplit <= 1 && !a.amax_val && a.a_kstep <= 0 && a.w_kstep <= 0 &&
           (size_t)a.M * (size_t)a.lda * 2 < (1ull << 32) && (size_t)a.N * (size_t)a.ldw * 2 < (1ull << 32);
}

// abl (measurement only): 0 product, 1 no global stores, 2 no epilogue, 3 no epilogue and no operand DMA (the loop alone)
int launch_gemm_f16x2_w4(const Gemm2Args& a, int abl, hipStream_t stream) {
    PF_REQUIRE(gemm_f16x2_w4_ok(a), "gemm_f16x2 (four-wave shape): needs K % 32 == 0, N % 256 == 0 and operands below 4 GB");
    const int mode = (a.R1 ? 1 : 0) | (a.R2 ? 2 : 0);
    if (a.qkv_D > 0) return launch_w4<0, 2>(a, stream);
    if (a.C2) {
        PF_REQUIRE(mode == 0, "gemm_f16x2: the plane output has no residual form");
        return launch_w4<0, 1>(a, stream);
    }
    if (abl == 1) return launch_w4<0, 0, 1>(a, stream);
    if (abl == 2) return launch_w4<0, 0, 2>(a, stream);
    if (abl == 3) return launch_w4<0, 0, 2, 1>(a, stream);
    if (abl == 4) return launch_w4<0, 0, 2, 3>(a, stream);        // no epilogue; pieces issued, never waited for (timing only)
    if (abl == 5) return launch_w4<0, 0, 2, 4>(a, stream);        // no epilogue; plain loads into a scratch register instead of LDS-DMA (timing only)
    if (abl == 6) return launch_w4<0, 0, 2, 5>(a, stream);        // no epilogue; only the scalar work of a piece (timing only)
    switch (mode) {
        case 0: return launch_w4<0, 0>(a, stream);
        case 1: return launch_w4<1, 0>(a, stream);
        case 2: return launch_w4<2, 0>(a, stream);
        default: return launch_w4<3, 0>(a, stream);
    }
}

}  // namespace pf
