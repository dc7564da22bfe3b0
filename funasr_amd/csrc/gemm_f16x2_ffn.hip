// The position-wise feed-forward of a SAN-M encoder block in ONE launch (mode f16x2):
//     x = x + w_2(relu(w_1 xn))     funasr/models/transformer/positionwise_feed_forward.py:14-34, sanm/encoder.py:141-146
//     next block's norm1(x)         funasr/models/sanm/encoder.py:96-98, transformer/layer_norm.py:13-38
// with xn = norm2(x) as two fp16 planes (written by linear_out's full-row epilogue), both weight matrices as two fp16 planes,
// three v_mfma_f32_32x32x16_f16 products per operand pair (gemm_f16x2.hip), and the [M, 2048] hidden activations NEVER in
// memory: the pair gemm_f16x2 (w_1, plane output) -> gemm_f16x2_row (w_2) writes and re-reads 536 MB of hidden planes per block
// at M = 32768 (a fifth of the step's HBM traffic), pays two prologues, two epilogue drains and two launches per block.
//
// Shape of the computation. One workgroup = 128 rows x the full 512-wide output, four waves, ONE wave per SIMD with the whole
// 512-entry register file: wave w owns rows 32 w .. 32 w + 31 and holds their 32 x 512 outputs as 16 accumulator tiles (256
// registers). The hidden dimension is streamed in chunks of 128 columns; per chunk
//     S^T[128 hidden, 32 rows] = W1[chunk, :] xn^T            "swapped" product: A operand = weight rows, B = activation rows
//     h = relu(S^T oscale1 + b1) * 2^eh -> hi / lo fp16       per lane, in registers
//     Y^T[512, 32 rows]      += W2[:, chunk] h                 B operand = h, STRAIGHT from S^T's accumulator registers
// The swapped form is what lets h stay in registers (the flash-attention idiom of attention_f16x2.hip with W1 as "K", W2 as
// "V^T" and no softmax): a lane of the 32 x 32 C/D layout holds, for ONE activation row, the hidden units
// {0..3, 8..11, 16..19, 24..27} + 4 (lane >> 5) of a 32-unit tile, but a B operand wants 8 CONSECUTIVE k per lane. So the
// A-operand lanes of the first product read W1's rows in the order "bits 2 and 3 of the row index swapped": accumulator
// register r of lane half hh then holds hidden unit 16 (r >> 3) + 8 hh + (r & 7) -- registers 0..7 / 8..15 are exactly the two
// B fragments of the tile's two 16-deep k-steps, in W2's natural column order (no repacked weights; the LDS bank pattern of a
// read group is unchanged because the swap permutes rows inside the group).
//
// Arithmetic: per output element the products, their order (lo*hi, hi*lo, hi*hi per 16-deep k-step, k ascending) and every
// epilogue expression are those of the two-kernel pair; the LayerNorm statistics are summed in layernorm_kernel's order
// (common.h ln_*). The hidden planes are rounded exactly as w_1's plane epilogue rounds them. What may differ from the pair is
// the matrix core's internal summation inside one 16-deep product when the operand roles are swapped; tests compare with the
// pair (bitwise where the hardware allows, else <= 2^-22 relative) and with float64.
//
// Data movement. Everything arrives by asm-issued LDS-DMA (global_load_lds_dwordx4, 1-KB pieces of 16 rows x 64 B, the chunk
// swizzle c ^ ((row >> 2) & 3) of gemm_f16x2.hip) into a ring of four 32-KB slots:
//     first product, per 32-deep K stage: [xn hi | xn lo | W1 hi | W1 lo] x 128 rows x 64 B = one slot, 16 stages per chunk
//     second product, per 32 hidden units: W2 hi plane 512 rows x 64 B = one slot, W2 lo plane = the next slot, 4 pairs per chunk
// The xn tile (256 KB for 128 rows: more than the CU's LDS) is re-streamed from L2 for every chunk: 768 KB of L2 -> LDS per
// chunk and workgroup, 3.1 GB per launch at M = 32768 against 2.3 GB for the two-kernel pair -- and 0.27 GB of HBM traffic
// (xn planes + residual in, residual + next planes out, 8 MB of weights) against 1.07 GB.
//
// STATE (round 4, measured; profiles/r04_ffn_*): correct -- bitwise the two-kernel pair on the fp32 stream AND the LayerNorm
// planes at every row count, in place, inside the 50-block encoder -- but NOT faster: 435-460 us against 455-480 us for the
// pair at M = 32768 (same box), so the engine keeps the pair (encoder option "ffn_fused" = 0) and this kernel is opt-in. Why:
//   * the kernel is bound by L2 -> LDS latency x bytes in flight, not by the matrix pipe: with the LDS-DMA pieces removed it runs
//     265-297 us (the MFMA + LDS-read floor; 187 us at 100 % of the matrix rate), with ONLY the pieces and barriers 396 us. One
//     wave sustains ~12 GB/s of LDS-DMA whatever the source (16 KB per ~1.4 us round trip; tools/micro/ldsdma_rate.hip: 12 / 22 / 41
//     / 75 GB/s per CU with 1 / 2 / 4 / 8 waves of 16 pieces each, L2-, Infinity-Cache- and HBM-resident spans alike), i.e.
//     throughput = bytes in flight / 1.4 us, and 160 KB of LDS hold at most three 32-KB slots in flight next to the one being read.
//   * three slots in flight need COUNTED waits ("all but my newest N pieces have landed"), and those are WRONG on this hardware:
//     the LDS-DMA pieces of one wave do not retire in issue order when their sources differ (L2 hit / Infinity Cache / HBM), so
//     s_waitcnt vmcnt(N) can return while an OLDER piece is still in flight and the stage is multiplied with the ring buffer's
//     previous contents -- plausible numbers, a few corrupted workgroups per thousand, invisible to warm micro-tests and found
//     by the 50-block encoder (ABL 12 keeps that schedule for the record: 410 us, wrong results). The same hazard sits in the
//     opt-in deep-ring shapes of gemm_f16x2.hip (tile 5 / 6). Exact waits -- s_waitcnt vmcnt(0) on everything the wave has in
//     flight, as every default kernel of this library does -- leave ONE stage of cover, which ties the pair.
//   * what would be needed: one owner wave per ring buffer (exact waits with three slots in flight) without per-piece branches --
//     tried: 950 basic blocks, 590 us -- or operands that bypass LDS-DMA (xn fragments by plain loads: no registers left), or an
//     interleaving of the two products across chunks (uniform 1 KB of DMA per MFMA; needs ~270 VGPRs). Not done.
#include "common.h"
#include <type_traits>

namespace pf {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int FF_BM = 128, FF_D = 512, FF_HC = 128, FF_KS = 32;
constexpr int FF_SLOT_B = 32768, FF_RING = 4, FF_PLANE_B = 8192;    // a 128-row plane of a 32-deep stage: 128 x 64 B
constexpr int FF_B1_OFF = FF_RING * FF_SLOT_B;                       // b1 [F] floats behind the ring
constexpr int FF_MAX_F = 6144;                                       // 128 KB + 24 KB <= 160 KB
constexpr int FF_ELD = 132;                                          // epilogue slab row (floats): 128 columns + 4
constexpr int FF_SLAB_B = 4 * 32 * FF_ELD * 4;                       // 67584: aliases the ring
static_assert(FF_SLAB_B <= FF_RING * FF_SLOT_B, "epilogue slabs alias the ring");

template <int N> __device__ __forceinline__ void ff_wait_but_() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

// Both products are issued from inline asm: the first with "+v" accumulators (S^T, 64 arch VGPRs), the second with "+a"
// accumulators (Y^T, all 256 accumulation registers). hipcc picks ONE register class for the C/D operands of every MFMA
// builtin of a kernel (AGPRs once the kernel may use more than 256 registers); 64 + 256 accumulators do not fit 256 AGPRs and
// it then shuttles tiles between the two files around every product (4 v_accvgpr moves per MFMA in the ISA). The asm
// statements are volatile, so their order -- and the placement of the LDS-DMA pieces and fragment reads between them -- is
// the source order: with one wave per SIMD nothing else fills the matrix pipe's shadow, so every filler is placed by hand.
// The compiler does not see an MFMA in these statements; the software-visible hazards are covered by ff_settle*() (wait
// states between the last product and the first VALU read of an accumulator, and behind the zero fill).
__device__ __forceinline__ void mfma_v(floatx16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_a(floatx16& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void ff_settle(floatx16 (&S)[4]) {
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(S[0]), "+v"(S[1]), "+v"(S[2]), "+v"(S[3]));
}
__device__ __forceinline__ void ff_settle_a(floatx16& y) { asm volatile("s_nop 15\n\ts_nop 15" : "+a"(y)); }
__device__ __forceinline__ f16x8 lds_frag(const unsigned char* p) { return __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(p)); }

// ABL (measurement only, tools/bench_ffn.py): 1 = no LDS-DMA (operands are whatever the LDS holds), 2 = no first product,
// 3 = no second product, 4 = no fragment reads in the loops, 5 = no xn pieces, 6 = no weight pieces, 7 = no W1 pieces, 8 = no W2 pieces
// (5..8: timing only), 12 = COUNTED vmcnt waits (three slots in flight). Counted waits are WRONG on this hardware: LDS-DMA pieces
// of one wave do not retire in issue order when their sources differ (L2 hit / Infinity Cache / HBM) -- s_waitcnt vmcnt(N) then
// returns while an OLDER piece is still in flight and the stage is multiplied with the ring buffer's previous contents. Warm
// micro-tests pass; a 50-block encoder run shows a few corrupted workgroups per thousand (tools/_dbg_ffn.py, r04). Exact waits
// (vmcnt(0) on everything the wave has in flight) are the only safe form.
template <bool LN, int ABL>
__global__ __launch_bounds__(256, 1) void ffn_f16x2_kernel(FfnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int m0 = blockIdx.x * FF_BM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, idx = lane & 31;

    // ---- b1 -> LDS (read back as broadcast float4s when a tile of S^T is finished)
    float* b1s = reinterpret_cast<float*>(smem + FF_B1_OFF);
    for (int i = tid; i < p.F; i += 256) b1s[i] = p.b1[i];
    glds_wait_all();                                   // plain loads and LDS-DMA pieces share vmcnt: start the counted waits clean

    // ---- DMA sources: per lane a byte offset (row l / 4 of a 16-row piece, physical chunk l % 4 <- the logical chunk the
    //      read-side swizzle expects there), per piece a uniform base
    const int prow = lane >> 2;
    const unsigned chunkb = (unsigned)(((lane & 3) ^ ((prow >> 2) & 3)) * 16);
    int rx0 = (ABL == 9 ? 0 : m0) + 16 * wave + prow, rx1 = rx0 + 64;       // ABL 9: every workgroup streams the SAME xn tile (L2-resident)
    rx0 = rx0 < p.M ? rx0 : p.M - 1;
    rx1 = rx1 < p.M ? rx1 : p.M - 1;
    const unsigned ox0 = (unsigned)rx0 * (unsigned)p.ldx * 2u + chunkb, ox1 = (unsigned)rx1 * (unsigned)p.ldx * 2u + chunkb;
    const unsigned ow1 = (unsigned)(16 * wave + prow) * (unsigned)p.ldw1 * 2u + chunkb;
    const unsigned ow2 = (unsigned)(16 * wave + prow) * (unsigned)p.ldw2 * 2u + chunkb;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (unsigned)wave * 1024);
    const int nchunk = p.F / FF_HC;
    const char* const bx = reinterpret_cast<const char*>(p.X2);
    const char* const bw1 = reinterpret_cast<const char*>(p.W1);
    const char* const bw2 = reinterpret_cast<const char*>(p.W2);
    // A slot is 32 pieces of 1 KB; this wave issues pieces wave + 4 i (i = 0..7), which land at (wave + 4 i) KB of the slot.
    // Slot t of chunk c (buffer t & 3): t < 16: stage t of the first product [xn hi 0..7 | xn lo 8..15 | W1 hi 16..23 | W1 lo
    // 24..31]; 16 + 2 j + pl: plane pl of W2[:, chunk c, hidden 32 j .. 32 j + 31] (512 rows x 64 B); t >= 24: the next chunk
    // (past the last chunk the same addresses are fetched again: harmless, drained before the epilogue; keeps the counted waits uniform)
    // The uniform part of a piece's address is built INSIDE the loops from a handful of SGPR values: `opaque` hides the small
    // per-piece constant from loop-invariant code motion -- otherwise hipcc precomputes ~100 distinct 64-bit bases in front of
    // the chunk loop, spills the SGPRs into VGPR lanes and the VGPRs into scratch.
    auto opaque = [](unsigned v) { asm volatile("" : "+s"(v)); return v; };
    const size_t xpl = p.x_plane * 2, w1pl = p.w1_plane * 2, w2pl = p.w2_plane * 2;
    const size_t w1half = (size_t)64 * p.ldw1 * 2, w2blk = (size_t)64 * p.ldw2 * 2;
    // bytes between the 32-deep K stages of an operand row: 64 in the row-major plane layout [rows, K]; rows * 64 in the K-blocked
    // layout [K / 32][rows][32] (FfnArgs.*_kstep > 0), where a 16-row piece is ONE contiguous KB = eight whole 128-B lines
    const size_t xks = (p.x_kstep > 0 ? (size_t)p.x_kstep : FF_KS) * 2, w1ks = (p.w1_kstep > 0 ? (size_t)p.w1_kstep : FF_KS) * 2,
                 w2ks = (p.w2_kstep > 0 ? (size_t)p.w2_kstep : FF_KS) * 2;
    // stage `kstage` (0..15, may be a runtime value) of the first product of chunk c -> buffer `buf`; piece i of this wave
    auto piece_g1 = [&](int buf, int kstage, int c, int i) {        // buf, i: compile-time
        if constexpr (ABL == 1) return;
        const unsigned dst = lds0 + (unsigned)buf * FF_SLOT_B + (unsigned)i * 4096;
        const unsigned ks = opaque((unsigned)kstage);
        if (i < 4) { if constexpr (ABL != 5) glds16s(bx + (i & 2 ? xpl : 0) + (size_t)ks * xks, i & 1 ? ox1 : ox0, dst); }
        else if constexpr (ABL != 6 && ABL != 7) glds16s(bw1 + (size_t)c * FF_HC * p.ldw1 * 2 + (i & 1 ? w1half : 0) + (i & 2 ? w1pl : 0) + (size_t)ks * w1ks, ow1, dst);
    };
    // plane pl of W2[:, chunk c, hidden 32 j .. 32 j + 31] -> buffer `buf`; piece i of this wave (rows 64 i + 16 wave ..)
    auto piece_g2 = [&](int buf, int pl, int j, int c, int i) {     // all but c: compile-time
        if constexpr (ABL == 1 || ABL == 6 || ABL == 8) return;
        const unsigned dst = lds0 + (unsigned)buf * FF_SLOT_B + (unsigned)i * 4096;
        const unsigned ks = opaque((unsigned)j);
        glds16s(bw2 + (pl ? w2pl : 0) + ((size_t)c * (FF_HC / FF_KS) + ks) * w2ks + (size_t)opaque(i) * w2blk, ow2, dst);
    };

    // ---- fragment addressing (64-B LDS rows; logical 16-B chunk 2 st + hh of row R sits at chunk ^ ((R >> 2) & 3))
    const int sw = (idx & 0x13) | ((idx & 4) << 1) | ((idx & 8) >> 1);          // bits 2 and 3 of the row index swapped
    const int fa = (sw >> 2) & 3, fb = (idx >> 2) & 3;
    // Twelve LDS base addresses per lane -- {xn row, W1 row, W2 row} x k-step x ring half -- so that every fragment read is
    // `base + 16-bit immediate`: offsets inside the 128-KB ring do not fit the DS offset field, and left to itself hipcc forms a
    // new address register per read, hoists them out of the chunk loop and spills them (a reload's vmcnt(0) drains the DMA ring).
    unsigned ax[2][2], aw1[2][2], aw2[2][2];            // [k-step][ring half: buffers 0, 1 / buffers 2, 3]
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ax[st][h] = lds_addr_of(smem) + h * 65536 + (32 * wave + idx) * 64 + ((2 * st + hh) ^ fb) * 16;          // xn row in a plane
            aw1[st][h] = lds_addr_of(smem) + h * 65536 + 2 * FF_PLANE_B + sw * 64 + ((2 * st + hh) ^ fa) * 16;       // W1 tile t: + t * 2048
            aw2[st][h] = lds_addr_of(smem) + h * 65536 + idx * 64 + ((2 * st + hh) ^ fb) * 16;                       // W2 tile o: + o * 2048
            asm volatile("" : "+v"(ax[st][h]), "+v"(aw1[st][h]), "+v"(aw2[st][h]));
        }
    typedef __attribute__((address_space(3))) const uint4* lds_u4;
    auto lds_at = [](unsigned base, int off) { return __builtin_bit_cast(f16x8, *(lds_u4)(uintptr_t)(base + (unsigned)off)); };

    // fragments of one 16-deep k-step of the first product: xn (B operand) hi / lo, W1 tiles 0..3 (A operand) hi / lo
    struct F1 { f16x8 x[2]; f16x8 w[4][2]; };
    auto read_f1 = [&](F1& f, int slot, int st) {       // slot, st: compile-time
        if constexpr (ABL == 4) return;
        const int h = (slot & 3) >> 1, sb = (slot & 1) * FF_SLOT_B;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            f.x[pl] = lds_at(ax[st][h], sb + pl * FF_PLANE_B);
#pragma unroll
            for (int t = 0; t < 4; ++t) f.w[t][pl] = lds_at(aw1[st][h], sb + pl * FF_PLANE_B + t * 2048);
        }
    };
    // fragments of one group of the second product: W2 tiles 2 g, 2 g + 1 (A operand), both 16-deep k-steps, hi / lo
    struct F2 { f16x8 w[2][2][2]; };                    // [k-step][tile][plane]
    auto read_f2 = [&](F2& f, int j, int g) {           // compile-time; the hi plane's slot is buffer 0 or 2, the lo plane's the next
        if constexpr (ABL == 4) return;
        const int h = (j & 1);
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int oo = 0; oo < 2; ++oo)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) f.w[st][oo][pl] = lds_at(aw2[st][h], pl * FF_SLOT_B + (2 * g + oo) * 2048);
    };

    floatx16 Y[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Y[o][r] = 0.f;
        ff_settle_a(Y[o]);
    }
    const float osc1 = p.oscale1, hsc = p.hscale;

    // h of tile j of the chunk: bias, ReLU, plane scale, two-plane split -> the B fragments of the tile's two 16-deep k-steps
    struct HF { f16x8 h[2][2]; };                       // [k-step][plane]
    floatx16 S[4];
    auto make_h = [&](HF& f, int c, int j) {
        const float* bj = b1s + c * FF_HC + 32 * j + 8 * hh;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float4 b0 = *reinterpret_cast<const float4*>(bj + 16 * g), b1 = *reinterpret_cast<const float4*>(bj + 16 * g + 4);
            const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaxf(S[j][8 * g + e] * osc1 + bv[e], 0.f) * hsc;
            uint4 h4, l4;
            split2_pk(o[0], o[1], h4.x, l4.x);
            split2_pk(o[2], o[3], h4.y, l4.y);
            split2_pk(o[4], o[5], h4.z, l4.z);
            split2_pk(o[6], o[7], h4.w, l4.w);
            f.h[g][0] = __builtin_bit_cast(f16x8, h4);
            f.h[g][1] = __builtin_bit_cast(f16x8, l4);
        }
    };

    // ---- prologue: slots 0..3 of chunk 0 in flight, slot 0 published, its first k-step's fragments requested
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) piece_g1(t, t, 0, i);
    if constexpr (ABL != 12) glds_wait_all(); else ff_wait_but_<24>();
    __syncthreads();
    F1 fA, fB;
    F2 gA, gB;
    read_f1(fA, 0, 0);

    // One stage of the first product: stage kt = 4 q + U of chunk c, in ring buffer U. Invariant at the top: slot kt is
    // published, fA holds its k-step 0, slots .. kt + 3 are issued. The buffer is refilled with slot kt + 4: a later stage of
    // this product (q < 3) or, in the last round (LASTQ), slot 16 + U = plane U & 1 of the W2 pair U >> 1.
    auto stage1 = [&](auto U_, auto LASTQ_, int q, int c) {
        constexpr int U = decltype(U_)::value;
        constexpr bool LASTQ = decltype(LASTQ_)::value;
        constexpr bool LAST = LASTQ && U == 3;             // stage 15: publishes the first W2 pair instead of another stage
        auto refill = [&](int i) {
            if constexpr (LASTQ) piece_g2(U, U & 1, U >> 1, c, i);
            else piece_g1(U, 4 * q + U + 4, c, i);
        };
        read_f1(fB, U, 1);
        if constexpr (ABL != 2) {
#pragma unroll
            for (int t = 0; t < 4; ++t) mfma_v(S[t], fA.w[t][0], fA.x[1]);   // xn lo * W hi
#pragma unroll
            for (int t = 0; t < 4; ++t) mfma_v(S[t], fA.w[t][1], fA.x[0]);   // xn hi * W lo
#pragma unroll
            for (int t = 0; t < 4; ++t) mfma_v(S[t], fA.w[t][0], fA.x[0]);   // xn hi * W hi
        }
        // every read of this slot is complete (fB is about to be multiplied): publish the next slot, then refill this buffer and
        // request the next k-step's fragments under the MFMAs below
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (ABL != 12) glds_wait_all(); else if constexpr (!LAST) ff_wait_but_<16>(); else ff_wait_but_<8>();
        __syncthreads();
        if constexpr (!LAST) read_f1(fA, (U + 1) & 3, 0); else read_f2(gA, 0, 0);
        if constexpr (ABL != 2) {
            mfma_v(S[0], fB.w[0][0], fB.x[1]); refill(0);
            mfma_v(S[1], fB.w[1][0], fB.x[1]); refill(1);
            mfma_v(S[2], fB.w[2][0], fB.x[1]); refill(2);
            mfma_v(S[3], fB.w[3][0], fB.x[1]); refill(3);
            mfma_v(S[0], fB.w[0][1], fB.x[0]); refill(4);
            mfma_v(S[1], fB.w[1][1], fB.x[0]); refill(5);
            mfma_v(S[2], fB.w[2][1], fB.x[0]); refill(6);
            mfma_v(S[3], fB.w[3][1], fB.x[0]); refill(7);
#pragma unroll
            for (int t = 0; t < 4; ++t) mfma_v(S[t], fB.w[t][0], fB.x[0]);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) refill(i);
        }
    };
    typedef std::integral_constant<bool, true> True_;
    typedef std::integral_constant<bool, false> False_;

    for (int c = 0; c < nchunk; ++c) {
        const int cn = c + 1 < nchunk ? c + 1 : c;       // past the last chunk the refills fetch it again (drained before the epilogue)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) S[t][r] = 0.f;
        ff_settle(S);

        // ------------------------------------------------------------------ first product: 16 stages of 32 over K = 512
        for (int q = 0; q < 3; ++q) {
            stage1(std::integral_constant<int, 0>{}, False_{}, q, c);
            stage1(std::integral_constant<int, 1>{}, False_{}, q, c);
            stage1(std::integral_constant<int, 2>{}, False_{}, q, c);
            stage1(std::integral_constant<int, 3>{}, False_{}, q, c);
        }
        stage1(std::integral_constant<int, 0>{}, True_{}, 3, c);
        stage1(std::integral_constant<int, 1>{}, True_{}, 3, c);
        stage1(std::integral_constant<int, 2>{}, True_{}, 3, c);
        stage1(std::integral_constant<int, 3>{}, True_{}, 3, c);
        ff_settle(S);

        // ------------------------------------------------------------------ second product: 4 steps of 32 hidden units, each 8
        // groups of 2 output tiles (12 MFMAs). Invariant at the top of step j: the W2 pair j (buffers 2 (j & 1), + 1) is
        // published, gA holds group 0, hf holds the step's B fragments; the barrier that publishes the next pair sits in front of
        // the LAST group's MFMAs, which cover the next step's first fragment reads and the DMA issue of the pair after it: pair
        // j + 2 of this chunk (j < 2) or stages 2 (j - 2), 2 (j - 2) + 1 of the next chunk's first product.
        HF hf;
        make_h(hf, c, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            HF hn;
            if (j < 3) make_h(hn, c, j + 1);
            auto refill = [&](int half, int i) {         // the two slots freed by this step: buffers 2 (j & 1) + half
                const int buf = 2 * (j & 1) + half;
                if (j < 2) piece_g2(buf, half, j + 2, c, i);
                else piece_g1(buf, 2 * (j - 2) + half, cn, i);
            };
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                F2& cur = (g & 1) ? gB : gA;
                F2& nxt = (g & 1) ? gA : gB;
                if (g < 7) read_f2(nxt, j, g + 1);
                else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    // next to be read: the pair j + 1 (j = 3: the next chunk's stage 0; its stage 1 may stay in flight)
                    if (j < 3 || ABL != 12) glds_wait_all(); else ff_wait_but_<8>();
                    __syncthreads();
                    if (j < 3) read_f2(gA, j + 1, 0); else read_f1(fA, 0, 0);
                }
                if constexpr (ABL != 3) {
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        mfma_a(Y[2 * g], cur.w[st][0][0], hf.h[st][1]);          // h lo * W hi
                        if (g == 7) { refill(0, 4 * st); refill(0, 4 * st + 1); }
                        mfma_a(Y[2 * g + 1], cur.w[st][1][0], hf.h[st][1]);
                        if (g == 7) { refill(0, 4 * st + 2); refill(0, 4 * st + 3); }
                        mfma_a(Y[2 * g], cur.w[st][0][1], hf.h[st][0]);          // h hi * W lo
                        if (g == 7) { refill(1, 4 * st); refill(1, 4 * st + 1); }
                        mfma_a(Y[2 * g + 1], cur.w[st][1][1], hf.h[st][0]);
                        if (g == 7) { refill(1, 4 * st + 2); refill(1, 4 * st + 3); }
                        mfma_a(Y[2 * g], cur.w[st][0][0], hf.h[st][0]);          // h hi * W hi
                        mfma_a(Y[2 * g + 1], cur.w[st][1][0], hf.h[st][0]);
                    }
                } else if (g == 7) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { refill(0, i); refill(1, i); }
                }
            }
            if (j < 3) hf = hn;
        }
    }
#pragma unroll
    for (int o = 0; o < 16; ++o) ff_settle_a(Y[o]);
    glds_wait_all();                                    // the refills issued past the last chunk

    // ---- epilogue. Y^T (C/D layout: this lane = row 32 wave + idx, register r of tile o = column 32 o + (r & 3) + 8 (r >> 2)
    //      + 4 hh) -> wave-private slab, 128 columns at a time -> row-major float4 pieces (half-wave h: rows 16 h .. 16 h + 15 of
    //      the wave's 32, lane c4 = idx: columns 4 c4 ..) -> bias, residual, the fp32 stream; the finished values stay in registers
    __syncthreads();                                    // the slabs alias the ring
    float* slab = reinterpret_cast<float*>(smem) + wave * (32 * FF_ELD);
    const int c4 = idx, rsub = hh;
    const float osc2 = p.oscale2;
    float4 ov[4][16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int oo = 0; oo < 4; ++oo)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                *reinterpret_cast<float4*>(slab + idx * FF_ELD + oo * 32 + 8 * r4 + 4 * hh) =
                    make_float4(Y[4 * q + oo][4 * r4], Y[4 * q + oo][4 * r4 + 1], Y[4 * q + oo][4 * r4 + 2], Y[4 * q + oo][4 * r4 + 3]);
        const int col = q * 128 + c4 * 4;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.b2) bias4 = *reinterpret_cast<const float4*>(p.b2 + col);
        const int row0 = m0 + wave * 32 + rsub * 16;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            float4 v[8], r2[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = *reinterpret_cast<const float4*>(slab + (rsub * 16 + h2 * 8 + t) * FF_ELD + c4 * 4);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int row = row0 + h2 * 8 + t;
                r2[t] = *reinterpret_cast<const float4*>(p.R + (size_t)(row < p.M ? row : p.M - 1) * p.ldr + col);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int row = row0 + h2 * 8 + t;
                float o[4] = {v[t].x * osc2 + bias4.x, v[t].y * osc2 + bias4.y, v[t].z * osc2 + bias4.z, v[t].w * osc2 + bias4.w};
                o[0] = r2[t].x + o[0]; o[1] = r2[t].y + o[1]; o[2] = r2[t].z + o[2]; o[3] = r2[t].w + o[3];
                const float4 o4 = make_float4(o[0], o[1], o[2], o[3]);
                ov[q][h2 * 8 + t] = o4;
                if (p.C && row < p.M) *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + col) = o4;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (!LN) return;

    // ---- LayerNorm over the 512 columns of every row, in layernorm_kernel's order: chunk index = 32 q + c4; lane l of the
    //      stand-alone kernel adds chunks l and l + 64, then the 64-lane xor butterfly 32, 16 .. 1 -- here (q0 + q2) + (q1 + q3)
    //      per c4, then the butterfly over c4 inside the half-wave that owns the row
    float4 g4[4], b4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        g4[q] = *reinterpret_cast<const float4*>(p.ln_g + q * 128 + c4 * 4);
        b4[q] = *reinterpret_cast<const float4*>(p.ln_b + q * 128 + c4 * 4);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        float s = (ln_sum4(ov[0][t]) + ln_sum4(ov[2][t])) + (ln_sum4(ov[1][t]) + ln_sum4(ov[3][t]));
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        const float mean = ln_mean(s, FF_D);
        float qd = (ln_sqdev4(ov[0][t], mean) + ln_sqdev4(ov[2][t], mean)) + (ln_sqdev4(ov[1][t], mean) + ln_sqdev4(ov[3][t], mean));
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) qd += __shfl_xor(qd, o, 64);
        const float rstd = ln_rstd(qd, FF_D, p.ln_eps);
        const int row = m0 + wave * 32 + rsub * 16 + t;
        if (row >= p.M) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 y = ln_apply4(ov[q][t], mean, rstd, g4[q], b4[q]);
            const int col = q * 128 + c4 * 4;
            if (p.Y2) {
                const float yv[4] = {y.x, y.y, y.z, y.w};
                store_split2x4_pair(p.Y2 + (size_t)row * p.ldy2 + col, p.y_plane, yv, p.yscale, lane);
            } else {
                *reinterpret_cast<float4*>(p.Yf + (size_t)row * p.ldyf + col) = y;
            }
        }
    }
}

template <bool LN, int ABL>
int launch_ffn_t(const FfnArgs& a, hipStream_t stream) {
    const int lds = FF_B1_OFF + a.F * 4;
    static int configured = 0;
    if (configured < lds) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_f16x2_kernel<LN, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured = lds;
    }
    hipLaunchKernelGGL((ffn_f16x2_kernel<LN, ABL>), dim3((unsigned)ceil_div(a.M, FF_BM)), dim3(256), lds, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace

bool ffn_f16x2_applicable(int D, int F) { return D == FF_D && F > 0 && F % FF_HC == 0 && F <= FF_MAX_F; }

int launch_ffn_f16x2(const FfnArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.M > 0 && a.D == FF_D && a.F > 0 && a.F % FF_HC == 0 && a.F <= FF_MAX_F, "ffn_f16x2: d_model must be 512, ffn_dim a multiple of 128 (<= 6144)");
    PF_REQUIRE(a.ldx % 8 == 0 && a.ldw1 % 8 == 0 && a.ldw2 % 8 == 0 && a.x_plane % 8 == 0 && a.w1_plane % 8 == 0 && a.w2_plane % 8 == 0,
               "ffn_f16x2: operand strides % 8");
    PF_REQUIRE(((uintptr_t)a.X2 & 15) == 0 && ((uintptr_t)a.W1 & 15) == 0 && ((uintptr_t)a.W2 & 15) == 0, "ffn_f16x2: operands must be 16-B aligned");
    PF_REQUIRE(a.b1 && a.R && a.ldr % 4 == 0 && ((uintptr_t)a.R & 15) == 0, "ffn_f16x2: b1 and the residual stream are required");
    if (a.b2) PF_REQUIRE(((uintptr_t)a.b2 & 15) == 0, "ffn_f16x2: b2 alignment");
    if (a.C) PF_REQUIRE(a.ldc % 4 == 0 && ((uintptr_t)a.C & 15) == 0, "ffn_f16x2: C alignment");
    const bool ln = a.ln_g != nullptr;
    if (ln) {
        PF_REQUIRE(a.ln_b && ((uintptr_t)a.ln_g & 15) == 0 && ((uintptr_t)a.ln_b & 15) == 0, "ffn_f16x2: LayerNorm parameters");
        PF_REQUIRE((a.Y2 != nullptr) != (a.Yf != nullptr), "ffn_f16x2: the LayerNorm form writes planes (Y2) or fp32 (Yf)");
        if (a.Y2) PF_REQUIRE(a.ldy2 % 8 == 0 && a.y_plane % 8 == 0 && ((uintptr_t)a.Y2 & 15) == 0, "ffn_f16x2: plane output alignment");
        else PF_REQUIRE(a.ldyf % 4 == 0 && ((uintptr_t)a.Yf & 15) == 0, "ffn_f16x2: fp32 LayerNorm output alignment");
        switch (a.abl) {
            case 1: return launch_ffn_t<true, 1>(a, stream);
            case 2: return launch_ffn_t<true, 2>(a, stream);
            case 3: return launch_ffn_t<true, 3>(a, stream);
            case 4: return launch_ffn_t<true, 4>(a, stream);
            case 5: return launch_ffn_t<true, 5>(a, stream);
            case 6: return launch_ffn_t<true, 6>(a, stream);
            case 7: return launch_ffn_t<true, 7>(a, stream);
            case 8: return launch_ffn_t<true, 8>(a, stream);
            case 9: return launch_ffn_t<true, 9>(a, stream);
            case 12: return launch_ffn_t<true, 12>(a, stream);
            default: return launch_ffn_t<true, 0>(a, stream);
        }
    }
    PF_REQUIRE(a.C, "ffn_f16x2: nothing to write");
    if (a.abl == 12) return launch_ffn_t<false, 12>(a, stream);
    return launch_ffn_t<false, 0>(a, stream);
}

}  // namespace pf
