// The f16x2 tile GEMM (gemm_f16x2.hip: C = epilogue(A[M,K] * W[N,K]^T), both operands as two fp16 planes, three
// v_mfma_f32_32x32x16_f16 products per operand pair into one fp32 accumulator) as a FOUR-wave 256 x 256 block: one wave per
// SIMD, each with the SIMD's whole 512-entry register file -- a 128 x 128 output quadrant = 16 accumulator tiles = all 256
// AGPRs, and both 16-deep k-steps' operand fragments (2 x 64 VGPRs) in flight. Same call sites as the other shapes
// (funasr/models/transformer/positionwise_feed_forward.py:14-34, funasr/models/sanm/attention.py:256,306); the same products
// in the same k order per output element and the shared epilogue (gemm_f16x2_epilogue.h), so results are BITWISE those of
// every other shape (tested: tests/test_kernels_f16x2_gpu.py).
//
// Why another shape (round 5). The eight-wave shapes read ALL fragments of a 32-deep stage right behind the stage's barrier,
// with both waves of every SIMD waiting for them at the same time: 192 ds_read_b128 (768 LDS cycles) per 3072 cycles of
// matrix work run uncovered, and the two 64-KB stages leave ONE stage (about one loaded L2 round trip) of DMA cover -- the
// loop alone reaches 0.62 of the matrix rate (tools/abl_gemm2.py), the kernel 0.33. Here
//   * LDS reads per MFMA drop by a third (16 ds_read_b128 per 48 MFMAs against 12 per 24) and the NEXT step's fragments
//     are read under the current step's MFMAs, one read per two MFMAs, placed by hand (every instruction of the K loop is
//     issued from volatile asm in source order: with one wave per SIMD nothing else fills the matrix pipe's shadow);
//   * a ring of five 32-KB stages (16 deep; the whole LDS), four of them in flight: a stage has three to four steps
//     (2-3 us) to land (the first build of this file had four stages and two to three steps: the loop + DMA ran 180 us
//     against 132 us for the loop alone at M = 32768, w_1);
//   * EXACT waits: LDS-DMA pieces of one wave were seen to retire out of issue order when their sources differ (DESIGN,
//     round 4), so counted vmcnt waits are not used. Stage s belongs to wave s % 4, which issues all 32 pieces of it -- one
//     piece per MFMA gap -- and is the only wave that waits for it (vmcnt(0), three steps later, when it has nothing else
//     in flight);
//   * one barrier among four waves per step instead of one among eight per stage.
//
// LDS (KS = 16 layout of gemm_f16x2.hip): a stage = [A hi | A lo | W hi | W lo] x 256 rows x 32 B, moved by 1-KB pieces of
// 32 rows (lane l -> row l / 2, physical chunk l % 2 holds logical chunk (l % 2) ^ ((row >> 3) & 1): conflict-free
// ds_read_b128 of the 32 x 16 fragments). The epilogue slabs (4 x 32 x 132 floats) alias the ring.
#include "common.h"
#include "gemm_f16x2_epilogue.h"

namespace pf {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int W4_PLANE_B = 256 * 32;              // one plane of a stage: 256 rows x 32 B
constexpr int W4_STAGE_B = 4 * W4_PLANE_B;        // 32 KB
constexpr int W4_RING = 5;
constexpr int W4_LDS_B = W4_RING * W4_STAGE_B;    // 160 KB: the whole LDS of a CU
static_assert(4 * 32 * (4 * 32 + 4) * 4 <= W4_LDS_B, "epilogue slabs alias the ring");

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
template <int LOFF> __device__ __forceinline__ void w4_piece(const char* sbase, unsigned voff, unsigned lds_buf, int own) {
    asm volatile("s_cmp_lg_u32 %4, 0\n\ts_cbranch_scc0 .Lw4_skip_%=\n\ts_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n.Lw4_skip_%=:"
                 :
                 : "v"(voff), "s"(sbase), "s"(lds_buf), "n"(LOFF), "s"(own)
                 : "memory", "scc");   // (m0 is reserved: the compiler keeps nothing in it)
}
// this wave waits for everything it has in flight iff `doit` (uniform) != 0 -- again without a branch the compiler sees
__device__ __forceinline__ void w4_wait_dma_if(bool cond) {
    const int doit = __builtin_amdgcn_readfirstlane(cond ? 1 : 0);
    asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc0 .Lw4_nowait_%=\n\ts_waitcnt vmcnt(0)\n.Lw4_nowait_%=:" : : "s"(doit) : "memory", "scc");
}

// EABL: the shared epilogue's measurement switch (0 product, 1 no global stores, 2 no epilogue); NODMA: no LDS-DMA pieces and
// no waits for them (the MFMA + fragment-read loop alone: tools/bench_w4.py)
template <int MODE, int OUT, int EABL, int NODMA>
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

    // ---- DMA sources: piece j of a plane = rows 32 j .. 32 j + 31 of the block
    unsigned va[8], vw[8];
    {
        const int prow = lane >> 1;
        const unsigned chunkb = (unsigned)(((lane & 1) ^ ((prow >> 3) & 1)) * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int row = m0 + 32 * j + prow;
            row = row < p.M ? row : p.M - 1;
            va[j] = (unsigned)row * (unsigned)p.lda * 2u + chunkb;
            int col = n0 + 32 * j + prow;
            col = col < p.N ? col : p.N - 1;
            vw[j] = (unsigned)col * (unsigned)p.ldw * 2u + chunkb;
            if constexpr (NODMA == 2) {
                // TIMING ONLY (wrong results): every piece reads ONE contiguous KB, the access pattern a K-blocked operand layout
                // [K / 16][rows][16] would give -- same bytes, same L2 footprint per stage, 8 whole lines per piece instead of 32 quarter lines
                int r0 = m0 + 32 * j; r0 = r0 < p.M - 40 ? r0 : (p.M > 40 ? p.M - 40 : 0);
                int c0 = n0 + 32 * j; c0 = c0 < p.N - 40 ? c0 : (p.N > 40 ? p.N - 40 : 0);
                va[j] = (unsigned)r0 * (unsigned)p.lda * 2u + (unsigned)lane * 16u;
                vw[j] = (unsigned)c0 * (unsigned)p.ldw * 2u + (unsigned)lane * 16u;
            }
        }
    }
    const char* const a_hi = reinterpret_cast<const char*>(p.A);
    const char* const a_lo = a_hi + p.a_plane * 2;
    const char* const w_hi = reinterpret_cast<const char*>(p.W);
    const char* const w_lo = w_hi + p.w_plane * 2;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    // piece I (0..31) of stage `kt` -> ring buffer at byte address `buf`
    auto piece = [&](auto I, int kt, unsigned buf, int own) {
        constexpr int i = decltype(I)::value;
        if constexpr (NODMA == 1) return;
        const size_t ko = (size_t)kt * (NODMA == 2 ? 1024 : 32);
        if constexpr (i < 8) w4_piece<i * 1024>(a_hi + ko, va[i & 7], buf, own);
        else if constexpr (i < 16) w4_piece<i * 1024>(a_lo + ko, va[i & 7], buf, own);
        else if constexpr (i < 24) w4_piece<i * 1024>(w_hi + ko, vw[i & 7], buf, own);
        else w4_piece<i * 1024>(w_lo + ko, vw[i & 7], buf, own);
    };

    // ---- fragment addresses: lane (idx, hh) reads row idx of a 32-row tile, logical chunk hh
    const unsigned coff = (unsigned)((hh ^ ((idx >> 3) & 1)) * 16);
    const unsigned fa0 = lds0 + (unsigned)((wr * 128 + idx) * 32) + coff;
    const unsigned fb0 = lds0 + 2 * W4_PLANE_B + (unsigned)((wc * 128 + idx) * 32) + coff;
    // read R (0..15) of a step: A lo tiles, W hi tiles (the first product's operands), A hi tiles, W lo tiles
    auto frag_read = [&](auto Rr, W4Frags& f, unsigned fa, unsigned fb) {
        constexpr int r = decltype(Rr)::value;
        if constexpr (r < 4) w4_read<W4_PLANE_B + (r & 3) * 1024>(f.al[r & 3], fa);
        else if constexpr (r < 8) w4_read<(r & 3) * 1024>(f.bh[r & 3], fb);
        else if constexpr (r < 12) w4_read<(r & 3) * 1024>(f.ah[r & 3], fa);
        else w4_read<W4_PLANE_B + (r & 3) * 1024>(f.bl[r & 3], fb);
    };

    const int nk = p.K / 16;                      // K % 64 == 0 (launcher)

    // ---- prologue: wave w fills buffer w with stage w; stage 0 is published; wave 0 (nothing in flight any more) sends stage 4;
    //      stage 0 is read; stage 1 is published
    {
        const unsigned buf = lds0 + (unsigned)wave * W4_STAGE_B;
        const int own = __builtin_amdgcn_readfirstlane(wave < nk ? 1 : 0);
        [&]<int... I>(std::integer_sequence<int, I...>) { (piece(std::integral_constant<int, I>{}, wave, buf, own), ...); }(std::make_integer_sequence<int, 32>{});
    }
    if constexpr (NODMA != 1) w4_wait_dma_if(wave == 0);
    __builtin_amdgcn_s_barrier();
    {
        const int own = __builtin_amdgcn_readfirstlane((wave == 0 && 4 < nk) ? 1 : 0);
        [&]<int... I>(std::integer_sequence<int, I...>) { (piece(std::integral_constant<int, I>{}, 4, lds0 + 4u * W4_STAGE_B, own), ...); }(std::make_integer_sequence<int, 32>{});
    }
    W4Frags f0, f1;
    [&]<int... I>(std::integer_sequence<int, I...>) { (frag_read(std::integral_constant<int, I>{}, f0, fa0, fb0), ...); }(std::make_integer_sequence<int, 16>{});
    w4_reads_done(f0);
    if constexpr (NODMA != 1) w4_wait_dma_if(wave == 1);
    __builtin_amdgcn_s_barrier();

    floatx16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    w4_settle(acc);

    // ---- one 16-deep step. C = kt % 4 (compile time): stage kt is in `x`; stage kt + 1 (ring buffer `rb` = (kt + 1) % 5) is read
    //      into `y` under the MFMAs, one read per two MFMAs; wave (C + 1) % 4 -- the owner of stage kt + 5, which waited for its
    //      previous stage at the end of the last step -- issues that stage into the buffer stage kt just left (`ib` = kt % 5),
    //      one piece per MFMA gap; at the end the owner of stage kt + 2 waits for it (the only thing it has in flight) and the
    //      barrier publishes it. Four stages in flight: a stage has three to four steps to land.
    int ib = 0, rb = 1;                            // kt % 5, (kt + 1) % 5
    auto step = [&](auto Cc, int kt, W4Frags& x, W4Frags& y) {
        constexpr int C = decltype(Cc)::value;
        const unsigned nb = (unsigned)rb * W4_STAGE_B;
        const unsigned fa = fa0 + nb, fb = fb0 + nb;
        const unsigned buf = lds0 + (unsigned)ib * W4_STAGE_B;
        const int own = __builtin_amdgcn_readfirstlane((wave == ((C + 1) & 3) && kt + 5 < nk) ? 1 : 0);
        [&]<int... G>(std::integer_sequence<int, G...>) {
            ([&] {
                constexpr int g = G, P = g >> 4, t = g & 15, i = t >> 2, j = t & 3;
                // the two small products first, hi * hi last -- the order of every other shape
                if constexpr (P == 0) w4_mfma(acc[i][j], x.al[i], x.bh[j]);
                else if constexpr (P == 1) w4_mfma(acc[i][j], x.ah[i], x.bl[j]);
                else w4_mfma(acc[i][j], x.ah[i], x.bh[j]);
                if constexpr ((g & 1) == 0 && g < 32) frag_read(std::integral_constant<int, (g >> 1)>{}, y, fa, fb);
                if constexpr (g < 32) piece(std::integral_constant<int, g>{}, kt + 5, buf, own);
            }(), ...);
        }(std::make_integer_sequence<int, 48>{});
        w4_reads_done(y);
        if constexpr (NODMA != 1) w4_wait_dma_if(wave == ((C + 2) & 3));      // stage kt + 2: the only one this wave has in flight
        __builtin_amdgcn_s_barrier();
        ib = rb;
        rb = rb == W4_RING - 1 ? 0 : rb + 1;
    };
    for (int kt0 = 0; kt0 < nk; kt0 += 4) {
        step(std::integral_constant<int, 0>{}, kt0, f0, f1);
        step(std::integral_constant<int, 1>{}, kt0 + 1, f1, f0);
        step(std::integral_constant<int, 2>{}, kt0 + 2, f0, f1);
        step(std::integral_constant<int, 3>{}, kt0 + 3, f1, f0);
    }
    w4_settle(acc);
    gemm2_epilogue<4, 4, 4, MODE, OUT, EABL>(p, acc, smem, m0, n0, nblk, wave, wr, wc, lane);
}

template <int MODE, int OUT, int EABL = 0, int NODMA = 0>
int launch_w4(const Gemm2Args& a, hipStream_t stream) {
    static bool configured = false;
    if (!configured) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_w4_kernel<MODE, OUT, EABL, NODMA>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_B));
        configured = true;
    }
    const int nM = ceil_div(a.M, 256), nN = ceil_div(a.N, 256);
    const int nMpad = (nM + 7) / 8 * 8;
    hipLaunchKernelGGL((gemm_f16x2_w4_kernel<MODE, OUT, EABL, NODMA>), dim3((unsigned)nMpad * nN), dim3(256), W4_LDS_B, stream, a, nM, nN);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

bool gemm_f16x2_w4_ok(const Gemm2Args& a) {
    // rows / columns are clamped in the DMA sources and masked in the epilogue like the other shapes; the ring wants whole
    // super-steps of four 16-deep stages and 32-bit per-lane byte offsets
    return a.K % 64 == 0 && a.N % 256 == 0 && a.kslices <= 1 && a.ksplit <= 1 && !a.amax_val && a.a_kstep <= 0 && a.w_kstep <= 0 &&
           (size_t)a.M * (size_t)a.lda * 2 < (1ull << 32) && (size_t)a.N * (size_t)a.ldw * 2 < (1ull << 32);
}

// abl (measurement only): 0 product, 1 no global stores, 2 no epilogue, 3 no epilogue and no operand DMA (the loop alone)
int launch_gemm_f16x2_w4(const Gemm2Args& a, int abl, hipStream_t stream) {
    PF_REQUIRE(gemm_f16x2_w4_ok(a), "gemm_f16x2 (four-wave shape): needs K % 64 == 0, N % 256 == 0 and operands below 4 GB");
    const int mode = (a.R1 ? 1 : 0) | (a.R2 ? 2 : 0);
    if (a.qkv_D > 0) return launch_w4<0, 2>(a, stream);
    if (a.C2) {
        PF_REQUIRE(mode == 0, "gemm_f16x2: the plane output has no residual form");
        return launch_w4<0, 1>(a, stream);
    }
    if (abl == 1) return launch_w4<0, 0, 1>(a, stream);
    if (abl == 2) return launch_w4<0, 0, 2>(a, stream);
    if (abl == 3) return launch_w4<0, 0, 2, 1>(a, stream);
    if (abl == 4) return launch_w4<0, 0, 2, 2>(a, stream);     // no epilogue, contiguous-KB DMA pattern (timing only)
    switch (mode) {
        case 0: return launch_w4<0, 0>(a, stream);
        case 1: return launch_w4<1, 0>(a, stream);
        case 2: return launch_w4<2, 0>(a, stream);
        default: return launch_w4<3, 0>(a, stream);
    }
}

}  // namespace pf
