// The f16x2 tile GEMM (gemm_f16x2.hip: C = epilogue(A[M,K] * W[N,K]^T), both operands as two fp16 planes, three
// v_mfma_f32_32x32x16_f16 products per operand pair into one fp32 accumulator) as a PERSISTENT, WAVE-SPECIALISED kernel (round 6).
// Same call sites as the other shapes (funasr/models/transformer/positionwise_feed_forward.py:14-34 w_1 / w_2,
// funasr/models/sanm/attention.py:256 linear_q_k_v); the same products in the same k order per output element and the same
// epilogue expressions, so results are BITWISE those of every other shape (tests/test_kernels_f16x2_gpu.py).
//
// Why. Round 5's dissection of the 256 x 256 shapes: the K loop runs at the power-limited matrix rate, and everything
// else is serial with it in every workgroup at the same time -- operand-DMA issue from the MFMA waves (+25 us of 216 on w_1), the
// epilogue through an LDS slab that aliases the stage buffers (+20), and a chip-wide store burst that the next tile's first stage
// wait has to sit out, because on gfx950 stores and LDS-DMA loads share vmcnt (+35). This kernel takes those three apart:
//   * one workgroup per CU for the whole launch, walking a static list of 256 x 128 tiles (XCD-aware order); eight waves:
//     waves 0-3 (one per SIMD) are MFMA waves, waves 4-7 (their SIMD partners) are LOADER waves;
//   * the MFMA waves never issue an LDS-DMA piece and never wait on vmcnt: their matrix pipe sees MFMAs, fragment ds_reads and,
//     between tiles, the epilogue -- whose global stores go out straight from the accumulator registers and are left in flight
//     while the next tile's K loop runs (nothing ever waits for them; a wave's 32 stores per tile fit the 6-bit counter);
//   * the product is computed TRANSPOSED (W fragment as the first MFMA operand), so a lane holds 4 consecutive output COLUMNS of
//     one row per accumulator quad: fp32 output = 16-B pieces, plane output = 16-B pieces after one v_permlane32_swap per dword
//     with the lane that holds the neighbouring 4 columns -- no LDS slab, so the stage ring is never interrupted;
//   * the loader waves own the ring of three 48-KB stages: stage x is issued whole (48 one-KB pieces) by loader x & 3 right after
//     the barrier that frees its buffer and waited for with vmcnt(0) -- exact, it is the only thing that wave has in flight --
//     before the barrier that publishes it. Two stages (96 KB) in flight against the one 64-KB stage of the other shapes, and
//     the next tile's first stages are in flight during the current tile's epilogue;
//   * ONE s_barrier per stage for all eight waves, in the middle of the stage (it publishes stage g + 1 and frees buffer g % 3).
// Price: 256 x 128 tiles move 1.5x the L2 -> LDS bytes per flop of a 256 x 256 tile, and a wave's 64 x 128 quadrant needs 12
// ds_read_b128 per 24 MFMAs (the eight-wave shape's ratio), placed by hand one per two MFMAs like gemm_f16x2_w4.hip.
#include "common.h"

namespace pf {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct PsGeo {
    static constexpr int BM = 256, BN = 128;
    static constexpr int A_PLANE_B = BM * 64, W_PLANE_B = BN * 64;          // 32-deep stage: 64-B LDS rows
    static constexpr int STAGE_B = 2 * (A_PLANE_B + W_PLANE_B);             // 48 KB
    static constexpr int NSTG = 3;
    static constexpr int LDS_B = NSTG * STAGE_B;                            // 144 KB
    static constexpr int A_PIECES = BM / 16, W_PIECES = BN / 16;            // 1-KB pieces (16 rows x 64 B) per plane
};

// this wave's fragments of one 16-deep k-step: W hi / lo tiles (4 x 32 columns), A hi / lo tiles (2 x 32 rows)
struct PsFrags { f16x8 wh[4], wl[4], ah[2], al[2]; };

__device__ __forceinline__ void ps_mfma(floatx16& acc, const f16x8& w, const f16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}
// a tile's very first products: C = 0 (no zero fill of 128 accumulator registers -- the compiler's fill was a reload of a spilled
// zero vector, i.e. a VMEM load that would wait behind the previous tile's stores)
__device__ __forceinline__ void ps_mfma0(floatx16& acc, const f16x8& w, const f16x8& a) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(w), "v"(a));
}
template <int OFF> __device__ __forceinline__ void ps_read(f16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void ps_reads_done(PsFrags& f) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f.wh[0]), "+v"(f.wh[1]), "+v"(f.wh[2]), "+v"(f.wh[3]), "+v"(f.wl[0]), "+v"(f.wl[1]), "+v"(f.wl[2]), "+v"(f.wl[3]),
                   "+v"(f.ah[0]), "+v"(f.ah[1]), "+v"(f.al[0]), "+v"(f.al[1]));
}
// the compiler does not see an MFMA in ps_mfma: the wait states between the last product (or the zero fill) and the next reader
// / writer of an accumulator are spent here
__device__ __forceinline__ void ps_settle(floatx16 (&acc)[4][2]) {
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                 : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[3][0]), "+a"(acc[3][1]));
}

// v_permlane32_swap: swaps the upper 32 lanes of `a` with the lower 32 lanes of `b` (gfx950)
__device__ __forceinline__ void ps_swap32(unsigned& a, unsigned& b) {
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// the static tile list of workgroup `wg` of `nwg` (nwg % 8 == 0): XCD x = wg & 7 owns the row blocks x, x + 8, ... and walks
// all column blocks of a row block back to back (the A panel stays in that XCD's L2); its `nwg / 8` workgroups take the list's
// entries slot, slot + nwg / 8, ...
struct PsTiles {
    int nM, nN, xcd, slot, per, count;      // count: tiles of this workgroup
    __device__ __forceinline__ void init(int M, int N, int wg, int nwg) {
        nM = (M + PsGeo::BM - 1) / PsGeo::BM; nN = N / PsGeo::BN;
        xcd = wg & 7; slot = wg >> 3; per = nwg >> 3;
        const int mine = nM > xcd ? (nM - xcd + 7) / 8 : 0;          // row blocks of this XCD
        const int total = mine * nN;
        count = total > slot ? (total - slot + per - 1) / per : 0;
    }
    __device__ __forceinline__ void at(int i, int& m0, int& n0) const {
        const int j = slot + i * per;
        m0 = ((j / nN) * 8 + xcd) * PsGeo::BM;
        n0 = (j % nN) * PsGeo::BN;
    }
};

// OUT: 0 fp32 C (+ R1 / R2 by MODE bits), 1 two fp16 planes of result * cscale, 2 the QKV / KV form (Gemm2Args)
template <int MODE, int OUT, bool RELU>
__global__ __launch_bounds__(512, 1) void gemm_f16x2_ps_kernel(Gemm2Args p) {
    typedef PsGeo G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    PsTiles tiles;
    tiles.init(p.M, p.N, (int)blockIdx.x, (int)gridDim.x);
    if (tiles.count == 0) return;
    const int nk = p.K / 32;
    const int total_stages = tiles.count * nk;

    if (wave >= 4) {
        // ======================================================================================== loader waves
        const int L = wave - 4;
        const int prow = lane >> 2;
        const unsigned chunkb = (unsigned)(((lane & 3) ^ ((prow >> 2) & 3)) * 16);
        const unsigned voff_a = (unsigned)prow * (unsigned)p.lda * 2u + chunkb;
        const unsigned voff_w = (unsigned)prow * (unsigned)p.ldw * 2u + chunkb;
        const char* const a_hi = reinterpret_cast<const char*>(p.A);
        const char* const w_hi = reinterpret_cast<const char*>(p.W);
        const size_t a_plane_b = p.a_plane * 2, w_plane_b = p.w_plane * 2;
        const int last_piece_row = p.M - 16;                                         // M % 16 == 0: a piece is valid or wholly past M
        // stage x (global index over this workgroup's tiles) -> buffer x % 3
        auto issue = [&](int x) {
            const int ti = x / nk, s = x - ti * nk;
            int m0, n0;
            tiles.at(ti, m0, n0);
            const unsigned dst = lds0 + (unsigned)(x % G::NSTG) * G::STAGE_B;
            const size_t ko = (size_t)s * 64;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll 4
                for (int q = 0; q < G::A_PIECES; ++q) {
                    int r0 = m0 + 16 * q;
                    r0 = r0 <= last_piece_row ? r0 : last_piece_row;
                    glds16s(a_hi + pl * a_plane_b + (size_t)r0 * p.lda * 2 + ko, voff_a, dst + pl * G::A_PLANE_B + q * 1024);
                }
            }
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll 4
                for (int q = 0; q < G::W_PIECES; ++q)
                    glds16s(w_hi + pl * w_plane_b + (size_t)(n0 + 16 * q) * p.ldw * 2 + ko, voff_w,
                            dst + 2 * G::A_PLANE_B + pl * G::W_PLANE_B + q * 1024);
            }
        };
        // stages 0 and 1 before the first barrier; then, around barrier g (g = -1 .. total - 1): the owner of stage g + 1 waits
        // for it before the barrier (the only thing it has in flight), the owner of stage g + 3 issues it after the barrier
        // (buffer g % 3 was read for the last time before it)
        if (L == 0) issue(0);
        if (L == 1 && total_stages > 1) issue(1);
        for (int g = -1; g < total_stages; ++g) {
            if (((g + 1) & 3) == L) glds_wait_all();
            __builtin_amdgcn_s_barrier();
            if (g + 3 < total_stages && ((g + 3) & 3) == L) issue(g + 3);
        }
        return;
    }

    // ============================================================================================ MFMA waves
    const int hh = lane >> 5, idx = lane & 31;
    // fragment addresses: lane (idx, hh) reads row idx of a 32-row tile, logical chunk 2 st + hh of k-step st
    const unsigned fsw = (unsigned)((idx >> 2) & 3);
    const unsigned fa0 = lds0 + (unsigned)((wave * 64 + idx) * 64);                 // this wave's 64 A rows
    const unsigned fw0 = lds0 + 2 * G::A_PLANE_B + (unsigned)(idx * 64);            // all 128 W rows
    auto coff = [&](int st) { return (unsigned)(((2 * st + hh) ^ fsw) * 16); };
    // read R (0..11) of a k-step: A lo tiles, W hi tiles (the first product's operands), A hi tiles, W lo tiles
    auto frag_read = [&](auto Rr, PsFrags& f, unsigned fa, unsigned fw) {
        constexpr int r = decltype(Rr)::value;
        if constexpr (r < 2) ps_read<G::A_PLANE_B + r * 2048>(f.al[r], fa);
        else if constexpr (r < 6) ps_read<(r - 2) * 2048>(f.wh[r - 2], fw);
        else if constexpr (r < 8) ps_read<(r - 6) * 2048>(f.ah[r - 6], fa);
        else ps_read<G::W_PLANE_B + (r - 8) * 2048>(f.wl[r - 8], fw);
    };
    PsFrags f0, f1;
    __builtin_amdgcn_s_barrier();                   // barrier -1: stage 0 has landed
    int buf = 0;
    const float oscale = p.oscale_dev ? p.oscale * *p.oscale_dev : p.oscale;
    // one tile, start to finish. A generic lambda instantiated per operand order: the accumulators are LOCAL to an instantiation
    // (a tile's first products write them with C = 0, nothing is carried between tiles), so the two orders never meet in a phi
    // and neither epilogue's registers leak into the other's K loop.
    // Swap (the V tiles of the QKV form): the A fragment is the first MFMA operand -- lane idx = column, registers = rows, the
    // layout whose register octets ARE the V^T pieces attention_f16x2.hip reads; same products, same sums
    auto run_tile = [&](auto Swap, const int m0, const int n0, const int seg) {
        constexpr bool SW = decltype(Swap)::value;
        floatx16 acc[4][2];                         // [column tile][row tile]; !SW: lane idx = row, registers = columns (D^T); SW: the reverse
        // one 16-deep k-step: 24 MFMAs on `x`; the next k-step's fragments are read into `y`, one read per two MFMAs
        auto kstep = [&](auto First, PsFrags& x, PsFrags& y, unsigned fa, unsigned fw) {
            [&]<int... Gp>(std::integer_sequence<int, Gp...>) {
                ([&] {
                    constexpr int g = Gp, P = g >> 3, t = g & 7, tn = t >> 1, tm = t & 1;
                    // the two small products first, hi * hi last -- the order of every other shape
                    const f16x8& w = P == 1 ? x.wl[tn] : x.wh[tn];
                    const f16x8& a = P == 0 ? x.al[tm] : x.ah[tm];
                    if constexpr (P == 0 && decltype(First)::value) { if constexpr (SW) ps_mfma0(acc[tn][tm], a, w); else ps_mfma0(acc[tn][tm], w, a); }
                    else { if constexpr (SW) ps_mfma(acc[tn][tm], a, w); else ps_mfma(acc[tn][tm], w, a); }
                    if constexpr ((g & 1) == 0) frag_read(std::integral_constant<int, (g >> 1)>{}, y, fa, fw);
                }(), ...);
            }(std::make_integer_sequence<int, 24>{});
            ps_reads_done(y);
        };
        // the tile's first fragments (its stage 0 was published by the previous tile's last barrier). Read here, not under the
        // previous tile's last k-step: 48 registers that would otherwise live across the epilogue
        {
            const unsigned cur0 = (unsigned)buf * G::STAGE_B;
            [&]<int... I>(std::integer_sequence<int, I...>) { (frag_read(std::integral_constant<int, I>{}, f0, fa0 + cur0 + coff(0), fw0 + cur0 + coff(0)), ...); }(std::make_integer_sequence<int, 12>{});
            ps_reads_done(f0);
        }
        auto stage = [&](auto First) {
            const unsigned cur = (unsigned)buf * G::STAGE_B;
            const int nb = buf + 1 == G::NSTG ? 0 : buf + 1;
            const unsigned nxt = (unsigned)nb * G::STAGE_B;
            kstep(First, f0, f1, fa0 + cur + coff(1), fw0 + cur + coff(1));
            __builtin_amdgcn_s_barrier();           // stage g + 1 is published; buffer `buf` has been read for the last time
            kstep(std::false_type{}, f1, f0, fa0 + nxt + coff(0), fw0 + nxt + coff(0));   // (in a tile's last stage: fragments nobody multiplies; the next tile reads its own)
            buf = nb;
        };
        stage(std::true_type{});                    // the tile's first products write the accumulators (C = 0)
        for (int s = 1; s < nk; ++s) stage(std::false_type{});
        ps_settle(acc);
        if constexpr (SW) {
                // ---- V tile. acc[tn][tm][r]: column d = n0 + 32 tn + idx, row m0 + 64 wave + 32 tm + 8 (r >> 2) + 4 hh + (r & 3).
            //      Registers 8 G .. 8 G + 7 = rows {0..3, 8..11} + 4 hh of the 16-row group G: one 16-B V^T piece per plane
            //      (the piece layout of gemm_f16x2_epilogue.h / attention_f16x2.hip). fp32 V (the FSMN memory block reads it):
            //      one dword per register, 32 lanes = 128 contiguous bytes of a row.
            float v_mul = p.v_mul;
            if (p.kv_mul_dev) v_mul *= p.kv_mul_dev[1];
            // (opaque copies: everything derived from them is recomputed per tile instead of being hoisted out of the tile loop
            // into ~200 registers that live across the K loop)
            int ldc_ = p.ldc, ldvt_ = p.ldvt, idx_ = idx, hh_ = hh;
            asm volatile("" : "+s"(ldc_), "+s"(ldvt_), "+v"(idx_), "+v"(hh_));
            const int vc0 = n0 - (p.kv_form ? 1 : 2) * p.qkv_D;                 // first column of the tile inside v
            const int mw = m0 + wave * 64;
            unsigned short* const vt0 = p.VT + (size_t)(vc0 + idx_) * ldvt_ + mw + 8 * hh_;
            float* const c0 = p.C ? p.C + (size_t)(mw + 4 * hh_) * ldc_ + vc0 + idx_ : nullptr;
            float bv[4];
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) bv[tn] = p.bias ? p.bias[n0 + 32 * tn + idx_] : 0.f;
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    const floatx16& a = acc[tn][tm];
#pragma unroll
                    for (int Gq = 0; Gq < 2; ++Gq) {
                        if (mw + 32 * tm + 16 * Gq >= p.M) continue;                // M % 16 == 0: a 16-row group is valid or wholly past M
                        float t[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) t[j] = (a[8 * Gq + j] * oscale + bv[tn]) * v_mul;
                        uint4 h, l;
                        split2_pk(t[0], t[1], h.x, l.x);
                        split2_pk(t[2], t[3], h.y, l.y);
                        split2_pk(t[4], t[5], h.z, l.z);
                        split2_pk(t[6], t[7], h.w, l.w);
                        unsigned short* vp = vt0 + (size_t)(32 * tn) * ldvt_ + 32 * tm + 16 * Gq;
                        *reinterpret_cast<uint4*>(vp) = h;
                        *reinterpret_cast<uint4*>(vp + p.vt_plane) = l;
                        if (c0) {
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                c0[(size_t)(32 * tm + 16 * Gq + 8 * (j >> 2) + (j & 3)) * ldc_ + 32 * tn] = a[8 * Gq + j] * oscale + bv[tn];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);      // one accumulator tile at a time (16 registers out of the accumulator file, not 128)
                }
            }
        } else {

        // ---- epilogue, straight from the accumulators. acc[tn][tm][r]: row m0 + 64 wave + 32 tm + idx,
        //      column n0 + 32 tn + 8 (r >> 2) + 4 hh + (r & 3): four consecutive columns per quad q = r >> 2.
        //      Two phases. Phase 1 finishes every value IN REGISTERS (bias, scale, ReLU, residual loads, plane split); phase 2 is
        //      32 stores and nothing else. vmcnt completes in issue order, so a load issued behind a store would wait for that
        //      store's acknowledgement: with all loads of a tile ahead of all of its stores the only thing a load ever waits
        //      behind is the PREVIOUS tile's stores, a whole K loop old.
        // (opaque per-tile copies of the lane coordinates: what is derived from them is recomputed per tile, not hoisted out of the
        // tile loop into registers that live across the K loop)
        int idx_ = idx, hh_ = hh;
        asm volatile("" : "+v"(idx_), "+v"(hh_));
        const int rbase = m0 + wave * 64 + idx_;
        // (results overwrite the accumulator elements they came from: 128 live registers, not 256)
        constexpr bool HAS_R1 = OUT == 0 && (MODE & 1) != 0, HAS_R2 = OUT == 0 && (MODE & 2) != 0;
        // units u = 0..7 of (column tile u >> 1, quads 2 (u & 1), 2 (u & 1) + 1): unit u + 1's loads are in flight under unit u's arithmetic
        float4 bias4[2][2], r1[2][2][2], r2[2][2][2];
        // plane outputs: the multiplier of the planes and where they go (QKV form: q / k planes, row stride qkv_D)
        float pscale = p.cscale;
        unsigned short* pdst = p.C2;
        int pld = p.ldc2, pcol0 = n0;
        size_t pplane = p.c_plane;
        if constexpr (OUT == 2) {
            pscale = seg == 0 ? p.q_mul : (p.kv_mul_dev ? p.k_mul * p.kv_mul_dev[0] : p.k_mul);
            pdst = seg == 0 ? p.Qp : p.Kp;
            pld = p.qkv_D; pplane = p.qk_plane;
            pcol0 = n0 - (n0 / p.qkv_D) * p.qkv_D;
        }
        auto load_unit = [&](auto U) {
            constexpr int u = decltype(U)::value, tn = u >> 1;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int q = 2 * (u & 1) + k;
                const int col = n0 + 32 * tn + 8 * q + 4 * hh_;
                bias4[u & 1][k] = p.bias ? *reinterpret_cast<const float4*>(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    const int row = rbase + 32 * tm;
                    const int rr = row < p.M ? row : p.M - 1;
                    if constexpr (HAS_R1) r1[u & 1][k][tm] = *reinterpret_cast<const float4*>(p.R1 + (size_t)rr * p.ldr1 + col);
                    if constexpr (HAS_R2) r2[u & 1][k][tm] = *reinterpret_cast<const float4*>(p.R2 + (size_t)rr * p.ldr2 + col);
                }
            }
        };
        auto finish_unit = [&](auto U) {
            constexpr int u = decltype(U)::value, tn = u >> 1;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int q = 2 * (u & 1) + k;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    floatx16& a = acc[tn][tm];
                    const float4 b4 = bias4[u & 1][k];
                    float o[4] = {a[4 * q + 0] * oscale + b4.x, a[4 * q + 1] * oscale + b4.y, a[4 * q + 2] * oscale + b4.z, a[4 * q + 3] * oscale + b4.w};
                    if constexpr (RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
                    }
                    if constexpr (OUT == 0) {
                        if constexpr (HAS_R1) { const float4 r = r1[u & 1][k][tm]; o[0] = o[0] + r.x; o[1] = o[1] + r.y; o[2] = o[2] + r.z; o[3] = o[3] + r.w; }
                        if constexpr (HAS_R2) { const float4 r = r2[u & 1][k][tm]; o[0] = r.x + o[0]; o[1] = r.y + o[1]; o[2] = r.z + o[2]; o[3] = r.w + o[3]; }
                        a[4 * q + 0] = o[0]; a[4 * q + 1] = o[1]; a[4 * q + 2] = o[2]; a[4 * q + 3] = o[3];
                    } else {
                        // two planes of o * cscale: this lane's 4 columns and the 4 of the lane 32 away (the other half of the same
                        // row's 8-column group): the lower lane ends up with the hi plane's 8 columns, the upper lane with the lo plane's
                        unsigned h0, l0, h1, l1;
                        split2_pk(o[0] * pscale, o[1] * pscale, h0, l0);
                        split2_pk(o[2] * pscale, o[3] * pscale, h1, l1);
                        ps_swap32(h0, l0);      // lower lanes: l0 <- the partner's h0; upper lanes: h0 <- the partner's l0
                        ps_swap32(h1, l1);
                        a[4 * q + 0] = __builtin_bit_cast(float, h0); a[4 * q + 1] = __builtin_bit_cast(float, h1);
                        a[4 * q + 2] = __builtin_bit_cast(float, l0); a[4 * q + 3] = __builtin_bit_cast(float, l1);
                    }
                }
            }
        };
        load_unit(std::integral_constant<int, 0>{});
        [&]<int... U>(std::integer_sequence<int, U...>) {
            ([&] {
                if constexpr (U + 1 < 8) load_unit(std::integral_constant<int, U + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                finish_unit(std::integral_constant<int, U>{});
                // the finished values go back to the accumulator file, not to 32 more VGPRs
                if constexpr ((U & 1) == 1) asm volatile("" : "+a"(acc[U >> 1][0]), "+a"(acc[U >> 1][1]));
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 8>{});
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) {
                    const int row = rbase + 32 * tm;
                    if (row >= p.M) continue;
                    const floatx16& a = acc[tn][tm];
                    const float4 v4 = make_float4(a[4 * q + 0], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
                    if constexpr (OUT == 0) {
                        *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + n0 + 32 * tn + 8 * q + 4 * hh_) = v4;
                    } else {
                        // hh = 0: hi plane, columns c8 .. c8 + 7; hh = 1: lo plane, the same columns
                        unsigned short* dst = pdst + (size_t)row * pld + pcol0 + 32 * tn + 8 * q + (hh_ ? pplane : (size_t)0);
                        *reinterpret_cast<float4*>(dst) = v4;
                    }
                }
        }
    };
    for (int ti = 0; ti < tiles.count; ++ti) {
        int m0, n0;
        tiles.at(ti, m0, n0);
        // QKV / KV form: this tile's 128 columns lie inside one of q | k | v (qkv_D % 128 == 0): 0 q, 1 k, 2 v
        const int seg = OUT == 2 ? n0 / p.qkv_D + (p.kv_form ? 1 : 0) : 0;
        if constexpr (OUT == 2) {
            if (seg == 2) { run_tile(std::true_type{}, m0, n0, seg); continue; }
        }
        run_tile(std::false_type{}, m0, n0, seg);
    }
}

template <int MODE, int OUT, bool RELU>
int launch_ps_r(const Gemm2Args& a, hipStream_t stream) {
    static PerDeviceOnce configured;
    if (!configured.done()) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_ps_kernel<MODE, OUT, RELU>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PsGeo::LDS_B));
        configured.mark();
    }
    const int n_cu = device_cu_count() / 8 * 8;
    const long tiles = (long)ceil_div(a.M, PsGeo::BM) * (a.N / PsGeo::BN);
    int grid = n_cu;
    if (tiles < grid) grid = (int)((tiles + 7) / 8 * 8);
    hipLaunchKernelGGL((gemm_f16x2_ps_kernel<MODE, OUT, RELU>), dim3((unsigned)grid), dim3(512), PsGeo::LDS_B, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

template <int MODE, int OUT>
int launch_ps(const Gemm2Args& a, hipStream_t stream) {
    return a.relu ? launch_ps_r<MODE, OUT, true>(a, stream) : launch_ps_r<MODE, OUT, false>(a, stream);
}

}  // namespace

bool gemm_f16x2_ps_ok(const Gemm2Args& a) {
    return a.K % 32 == 0 && a.K >= 64 && a.N % 128 == 0 && a.M % 16 == 0 && a.M >= 16 && a.kslices <= 1 && a.ksplit <= 1 && !a.amax_val &&
           (a.qkv_D <= 0 || (a.qkv_D % 128 == 0 && !a.R1 && !a.R2 && !a.relu && ((uintptr_t)a.VT & 15) == 0 && a.ldvt % 8 == 0 && a.vt_plane % 8 == 0 && a.qk_plane % 8 == 0)) && a.a_kstep <= 0 && a.w_kstep <= 0 && (size_t)a.lda * 32 < (1ull << 32) && (size_t)a.ldw * 32 < (1ull << 32) && a.N % 8 == 0 &&
           !(a.R1 && a.R2) && (!a.C2 || (a.ldc2 % 8 == 0 && a.c_plane % 8 == 0 && ((uintptr_t)a.C2 & 15) == 0 && !a.R1 && !a.R2));
}

int launch_gemm_f16x2_ps(const Gemm2Args& a, hipStream_t stream) {
    PF_REQUIRE(gemm_f16x2_ps_ok(a), "gemm_f16x2 (persistent shape): needs K % 32 == 0, K >= 64, N % 128 == 0, M % 16 == 0, fp32 or plane output");
    if (a.qkv_D > 0) return launch_ps_r<0, 2, false>(a, stream);
    if (a.C2) return launch_ps<0, 1>(a, stream);
    const int mode = (a.R1 ? 1 : 0) | (a.R2 ? 2 : 0);
    switch (mode) {
        case 0: return launch_ps<0, 0>(a, stream);
        case 1: return launch_ps<1, 0>(a, stream);
        default: return launch_ps<2, 0>(a, stream);      // (both residuals at once: refused by gemm_f16x2_ps_ok -- no call site has them)
    }
}

}  // namespace pf
