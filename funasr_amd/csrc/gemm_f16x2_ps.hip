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
//   * no LDS slab: a store instruction's 32 lanes are 32 consecutive columns of a row (the C/D layout of the 32 x 32 MFMA with the A
//     fragment as first operand), fp32 output = dword stores of 128 contiguous bytes per row, plane output = one DPP exchange
//     between neighbouring lanes and dword stores of two fp16 (64 contiguous bytes per row and plane) -- the stage ring is never
//     interrupted by an epilogue;
//   * the loader waves own the ring of three 48-KB stages: loaders 0 / 1 the even stages, 2 / 3 the odd ones, one operand plane
//     each (24 one-KB pieces per loader and stage), issued right after the barrier that frees the stage's buffer and waited for
//     with vmcnt(0) -- exact, it is the only thing that wave has in flight -- before the barrier that publishes it. Two stages
//     (96 KB) in flight against the one 64-KB stage of the other shapes, and the next tile's first stages are in flight during the
//     current tile's epilogue;
//   * ONE s_barrier per stage for all eight waves, in the middle of the stage (it publishes stage g + 1 and frees buffer g % 3).
// Price: 256 x 128 tiles move 1.5x the L2 -> LDS bytes per flop of a 256 x 256 tile, and a wave's 64 x 128 quadrant needs 12
// ds_read_b128 per 24 MFMAs (the eight-wave shape's ratio), placed by hand one per two MFMAs like gemm_f16x2_w4.hip.
#include "common.h"

namespace pf {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float ps_f4 __attribute__((ext_vector_type(4)));           // (a native vector: HIP's float4 is a struct, no register constraint takes it)

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

// one 1-KB LDS-DMA piece: uniform 64-bit base + 32-bit per-lane byte offset, destination lds_buf + LOFF (m0 is reserved: the
// compiler keeps nothing in it). Three instructions.
template <int LOFF> __device__ __forceinline__ void ps_piece(const char* sbase, unsigned voff, unsigned lds_buf) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_buf), "n"(LOFF) : "memory", "scc");
}

// v_permlane32_swap: swaps the upper 32 lanes of `a` with the lower 32 lanes of `b` (gfx950)
// (the builtin, not inline asm: the instruction needs wait states after a VALU write of its operands, which the compiler's hazard
// recogniser only inserts for instructions it can see -- the asm form returned garbage right behind v_accvgpr_read)
__device__ __forceinline__ void ps_swap32(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
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
    // measurement switches (tools/bench_ps.py; Gemm2Args.tile = 10 + 16 abl): bit 0 no operand DMA, bit 1 no epilogue, bit 2 no stores,
    // bit 3 stores into an L2-resident window, bit 4 staggered workgroup starts
    const int abl = p.tile >> 4;
    if (abl & 16) {
        // measurement: workgroups start up to one tile time apart (16 phases), so that their epilogues' store phases do not coincide
        const long long t0 = __builtin_readcyclecounter();
        const long long d = (long long)(((int)blockIdx.x >> 3) & 15) * (p.K / 32) * 140;     // 16 x 140 cycles ~ one 32-deep stage
        while (__builtin_readcyclecounter() - t0 < d) __builtin_amdgcn_s_sleep(16);
    }

    if (wave >= 4) {
        // ======================================================================================== loader waves
        // Loaders 0 / 1 own the EVEN stages, 2 / 3 the ODD ones; inside a pair, loader (L & 1) moves plane (L & 1) of both operands:
        // 16 A pieces + 8 W pieces = 24 KB per loader and stage. A stage is issued right after the barrier that frees its buffer and
        // waited for with vmcnt(0) before the barrier that publishes it -- exact: a loader never has more than that one stage in
        // flight (its previous one was published two barriers earlier). Issue is three instructions per piece (m0, wait state,
        // load): uniform 64-bit base per (plane, stage) in SGPRs, the piece's rows in a per-lane byte offset that is computed
        // once per tile (A) / once per launch (W) and kept in this wave's otherwise idle registers. (The first form of this
        // loader -- one wave per stage, address arithmetic per piece -- needed 2.3 us to issue a 48-KB stage and made the whole
        // kernel DMA-issue bound: w_1 316 us, profiles/r06b_ps_first_run.jsonl.)
        const int L = wave - 4, par = L >> 1, pl = L & 1;
        const int prow = lane >> 2;
        const unsigned chunkb = (unsigned)(((lane & 3) ^ ((prow >> 2) & 3)) * 16);
        const char* const a_pl = reinterpret_cast<const char*>(p.A) + (size_t)pl * p.a_plane * 2;
        const char* const w_pl = reinterpret_cast<const char*>(p.W) + (size_t)pl * p.w_plane * 2;
        unsigned va[G::A_PIECES], vw[G::W_PIECES];
#pragma unroll
        for (int q = 0; q < G::W_PIECES; ++q) vw[q] = (unsigned)(16 * q + prow) * (unsigned)p.ldw * 2u + chunkb;
        const unsigned lds_a = lds0 + (unsigned)pl * G::A_PLANE_B, lds_w = lds0 + 2 * G::A_PLANE_B + (unsigned)pl * G::W_PLANE_B;
        // my next stage: global index x = tile ti, stage s of the tile; (m0, n0) and the A offsets follow the tile
        int x = par, ti = 0, s = par;
        while (s >= nk) { s -= nk; ++ti; }
        int cur_ti = -1;
        const char* a_tile = a_pl;
        const char* w_tile = w_pl;
        auto issue = [&]() {
            if (ti != cur_ti) {
                int m0, n0;
                tiles.at(ti, m0, n0);
                cur_ti = ti;
                a_tile = a_pl + (size_t)m0 * p.lda * 2;
                w_tile = w_pl + (size_t)n0 * p.ldw * 2;
                // rows past M (M % 16 == 0: a piece is valid or wholly past M) re-read the last valid piece; their products are never stored
                const int last = p.M - 16 - m0;
#pragma unroll
                for (int q = 0; q < G::A_PIECES; ++q) va[q] = (unsigned)((16 * q <= last ? 16 * q : last) + prow) * (unsigned)p.lda * 2u + chunkb;
            }
            const unsigned buf = (unsigned)(x % G::NSTG) * G::STAGE_B;
            const char* const ab = a_tile + (size_t)s * 64;
            const char* const wb = w_tile + (size_t)s * 64;
            [&]<int... Q>(std::integer_sequence<int, Q...>) { (ps_piece<Q * 1024>(ab, va[Q], lds_a + buf), ...); }(std::make_integer_sequence<int, G::A_PIECES>{});
            [&]<int... Q>(std::integer_sequence<int, Q...>) { (ps_piece<Q * 1024>(wb, vw[Q], lds_w + buf), ...); }(std::make_integer_sequence<int, G::W_PIECES>{});
            x += 2; s += 2;
            while (s >= nk) { s -= nk; ++ti; }
        };
        // stages 0 (pair 0) and 1 (pair 1) before the first barrier; then, around barrier g (g = -1 .. total - 1): the pair of stage
        // g + 1 waits for it before the barrier, the pair of stage g + 3 issues it after the barrier (buffer g % 3 was read for
        // the last time before it)
        const bool dma = (abl & 1) == 0;
        if (x < total_stages && dma) issue();
        for (int g = -1; g < total_stages; ++g) {
            if (((g + 1) & 1) == par) glds_wait_all();
            __builtin_amdgcn_s_barrier();
            if (((g + 3) & 1) == par && g + 3 < total_stages && dma) issue();
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
    constexpr bool HAS_R1 = OUT == 0 && (MODE & 1) != 0, HAS_R2 = OUT == 0 && (MODE & 2) != 0;
    for (int ti = 0; ti < tiles.count; ++ti) {
        int m0, n0;
        tiles.at(ti, m0, n0);
        // accumulators [column tile][row tile], local to the tile (its first products write them with C = 0). The A fragment is the
        // FIRST MFMA operand: lane idx = output COLUMN n0 + 32 tn + idx, registers = rows 32 tm + 8 (r >> 2) + 4 hh + (r & 3)
        floatx16 acc[4][2];
        // one 16-deep k-step: 24 MFMAs on `x`; the next k-step's fragments are read into `y`, one read per two MFMAs
        auto kstep = [&](auto First, PsFrags& x, PsFrags& y, unsigned fa, unsigned fw) {
            [&]<int... Gp>(std::integer_sequence<int, Gp...>) {
                ([&] {
                    constexpr int g = Gp, P = g >> 3, t = g & 7, tn = t >> 1, tm = t & 1;
                    // the two small products first, hi * hi last -- the order of every other shape
                    const f16x8& w = P == 1 ? x.wl[tn] : x.wh[tn];
                    const f16x8& a = P == 0 ? x.al[tm] : x.ah[tm];
                    if constexpr (P == 0 && decltype(First)::value) ps_mfma0(acc[tn][tm], a, w);
                    else ps_mfma(acc[tn][tm], a, w);
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
        if (abl & 2) {                              // (the accumulators still count as used: the products are volatile asm)
            if (acc[0][0][0] == 123.456f && p.C) p.C[0] = acc[3][1][15];
            continue;
        }

        // ---- epilogue, straight from the accumulators: no LDS slab, so the stage ring is never interrupted and the loaders keep
        //      filling it under the epilogue. A store instruction's 32 lanes are 32 consecutive COLUMNS of a row (the two half-waves:
        //      rows 4 apart): fp32 = 128 contiguous bytes per row and instruction; planes: lanes 2k / 2k + 1 trade one value (a DPP
        //      quad permute) so that the even lane holds columns (c, c + 1) of row r and the odd lane those of row r + 1 -- one dword
        //      of two fp16 per plane, 64 contiguous bytes per row and instruction. (The first form of this kernel computed the
        //      TRANSPOSED product, lane = row, and stored 16-B pieces of 32 different rows per instruction: one cache line per lane,
        //      and the loaders' LDS-DMA queued behind it -- profiles/r06c_ps_fast_loaders.jsonl: epilogue 69 us of a 279-us w_1.)
        //      Phase 1 finishes every value IN REGISTERS (the finished words overwrite the accumulator elements they came from);
        //      phase 2 is stores and nothing else, left in flight under the next tile's K loop: vmcnt completes in issue order, so
        //      with all loads of a tile ahead of all of its stores a load only ever waits behind the PREVIOUS tile's stores.
        // (opaque per-tile copies of the lane coordinates: what is derived from them is recomputed per tile, not hoisted out of the
        // tile loop into registers that live across the K loop)
        int idx_ = idx, hh_ = hh;
        asm volatile("" : "+v"(idx_), "+v"(hh_));
        const int odd = idx_ & 1;
        const int rowl = wave * 64 + 4 * hh_;                                  // + 32 tm + 8 (r >> 2) + (r & 3): row inside the tile
        const int seg = OUT == 2 ? n0 / p.qkv_D + (p.kv_form ? 1 : 0) : 0;     // QKV / KV form: 0 q, 1 k, 2 v (qkv_D % 128 == 0)
        if constexpr (OUT == 2) {
            if (seg == 2) {
                // ---- V tile: registers 8 G .. 8 G + 7 = rows {0..3, 8..11} + 4 hh of the 16-row group G: one 16-B V^T piece per plane
                //      (the piece layout of gemm_f16x2_epilogue.h / attention_f16x2.hip); fp32 V (the FSMN memory block reads it): one
                //      dword per register
                float v_mul = p.v_mul;
                if (p.kv_mul_dev) v_mul *= p.kv_mul_dev[1];
                int ldc_ = p.ldc, ldvt_ = p.ldvt;
                asm volatile("" : "+s"(ldc_), "+s"(ldvt_));
                const int vc0 = n0 - (p.kv_form ? 1 : 2) * p.qkv_D;             // first column of the tile inside v
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
                            if (mw + 32 * tm + 16 * Gq >= p.M) continue;        // M % 16 == 0: a 16-row group is valid or wholly past M
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
                continue;
            }
        }
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
        // Units = PAIRS of column tiles: u = 2 p + tm holds acc[2 p][tm] and acc[2 p + 1][tm] (64 columns x 32 rows). One
        // v_permlane32_swap per register pair turns (columns 32 x rows {hh = 0 | hh = 1}) x 2 tiles into
        //     A'[r] = rows 8 (r >> 2) + (r & 3)     , lower half-wave: tile 2 p, upper half-wave: tile 2 p + 1   (64 consecutive columns)
        //     B'[r] = rows 8 (r >> 2) + (r & 3) + 4 , the same columns
        // so that a store instruction covers ONE row: 256 contiguous bytes of fp32, or (after the DPP exchange between neighbouring
        // lanes) two rows x 128 contiguous bytes per plane -- whole cache lines instead of 64-B halves of four of them.
        // Unit u + 1's residual loads are in flight under unit u's arithmetic.
        const int h2 = hh_;                                                     // half-wave = which tile of the pair
        const int lc = 32 * h2 + idx_;                                          // this lane's column inside the pair's 64
        float bv[2];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) bv[pp] = p.bias ? p.bias[n0 + 64 * pp + lc] : 0.f;
        float r1[2][32], r2[2][32];
        // (uniform 64-bit bases in SGPRs + ONE per-lane offset: 128 per-lane addresses would cost 256 registers. M % 16 == 0 and the
        // rows of a register octet lie in one 16-row group, so row validity is uniform too)
        int ldr1_ = p.ldr1, ldr2_ = p.ldr2;
        asm volatile("" : "+s"(ldr1_), "+s"(ldr2_));
        const int mw = m0 + wave * 64;
        const int mws = (abl & 8) ? wave * 64 : mw;                             // measurement: every tile stores into the first 256 rows (an L2-resident window)
        auto load_unit = [&](auto U) {
            constexpr int u = decltype(U)::value, pp = u >> 1, tm = u & 1;
            if constexpr (HAS_R1 || HAS_R2) {
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const int r = k & 15, rl = 32 * tm + 8 * (r >> 2) + (r & 3) + 4 * (k >> 4);
                    const bool ok = mw + 32 * tm + 16 * (r >> 3) < p.M;
                    if constexpr (HAS_R1) r1[u & 1][k] = ok ? (p.R1 + (size_t)(mws + rl) * ldr1_ + n0 + 64 * pp)[lc] : 0.f;
                    if constexpr (HAS_R2) r2[u & 1][k] = ok ? (p.R2 + (size_t)(mws + rl) * ldr2_ + n0 + 64 * pp)[lc] : 0.f;
                }
            }
        };
        auto finish_unit = [&](auto U) {
            constexpr int u = decltype(U)::value, pp = u >> 1, tm = u & 1;
            floatx16& A = acc[2 * pp][tm];
            floatx16& B = acc[2 * pp + 1][tm];
            float o[32];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float fa = A[r], fb = B[r];   // (copies: __builtin_bit_cast of a vector ELEMENT lvalue reads element 0)
                unsigned x = __builtin_bit_cast(unsigned, fa), y = __builtin_bit_cast(unsigned, fb);
                ps_swap32(x, y);                    // x = A'[r], y = B'[r]
                o[r] = __builtin_bit_cast(float, x);
                o[16 + r] = __builtin_bit_cast(float, y);
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                o[k] = o[k] * oscale + bv[pp];
                if constexpr (RELU) o[k] = fmaxf(o[k], 0.f);
                if constexpr (HAS_R1) o[k] = o[k] + r1[u & 1][k];
                if constexpr (HAS_R2) o[k] = r2[u & 1][k] + o[k];
            }
            if constexpr (OUT == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { A[r] = o[r]; B[r] = o[16 + r]; }
            } else {
                // rows (r, r + 1), r even: the even lane keeps row r and gets the odd lane's value of it (column c + 1); the odd lane keeps
                // row r + 1 and gets the even lane's (column c - 1): (hi, lo) words of two adjacent columns of ONE row per lane
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    // (the UNSCALED values are exchanged and the plane multiplier is applied inside the split, the expression of
                    // store_split2x4_pair: the compiler fuses `o * scale - hi` into one fma there, so with a multiplier that is not a
                    // power of two -- q's d_k^-1/2 -- scaling before the exchange gives other low-plane bits than the other shapes)
                    const float y0 = o[2 * k], y1 = o[2 * k + 1];
                    const float got = __builtin_bit_cast(float, dpp_swap1(__builtin_bit_cast(unsigned, odd ? y0 : y1)));
                    const float ca = odd ? got : y0, cb = odd ? y1 : got;
                    unsigned h, l;
                    split2_pk(ca * pscale, cb * pscale, h, l);
                    floatx16& D = k < 8 ? A : B;
                    D[2 * (k & 7)] = __builtin_bit_cast(float, h);
                    D[2 * (k & 7) + 1] = __builtin_bit_cast(float, l);
                }
            }
        };
        load_unit(std::integral_constant<int, 0>{});
        [&]<int... U>(std::integer_sequence<int, U...>) {
            ([&] {
                if constexpr (U + 1 < 4) load_unit(std::integral_constant<int, U + 1>{});
                __builtin_amdgcn_sched_barrier(0);
                finish_unit(std::integral_constant<int, U>{});
                // the finished words go back to the accumulator file, not to 32 more VGPRs
                asm volatile("" : "+a"(acc[2 * (U >> 1)][U & 1]), "+a"(acc[2 * (U >> 1) + 1][U & 1]));
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 4>{});
        if (abl & 4) {                              // measurement: everything but the stores
            if (acc[0][0][0] == 123.456f && p.C) p.C[0] = acc[3][1][15];
            continue;
        }
        int ldo_ = OUT == 0 ? p.ldc : pld;
        asm volatile("" : "+s"(ldo_));
        const size_t so = OUT == 0 ? (size_t)lc : (size_t)odd * ldo_ + (lc & ~1);   // this lane inside a register's row (pair)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int half = 0; half < 2; ++half) {                           // A' (rows + 0), B' (rows + 4)
                    const floatx16& a = acc[2 * pp + half][tm];
                    if constexpr (OUT == 0) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int rl = 32 * tm + 8 * (r >> 2) + (r & 3) + 4 * half;
                            if (mw + 32 * tm + 16 * (r >> 3) < p.M) (p.C + (size_t)(mws + rl) * ldo_ + n0 + 64 * pp)[so] = a[r];
                        }
                    } else {
                        // this lane's row of the pair (r, r + 1) is r + odd; its two columns start at the even one
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int rl = 32 * tm + 8 * (k >> 1) + 2 * (k & 1) + 4 * half;   // rows 0, 2, 8, 10, 16, 18, 24, 26 of the tile (+ 4 half + odd)
                            if (mw + 32 * tm + 16 * (k >> 2) < p.M) {
                                unsigned short* const d = pdst + (size_t)(mws + rl) * ldo_ + pcol0 + 64 * pp;
                                *reinterpret_cast<float*>(d + so) = a[2 * k];
                                *reinterpret_cast<float*>(d + pplane + so) = a[2 * k + 1];
                            }
                        }
                    }
                }
    }
}

#if defined(PF_MEASUREMENT_KERNELS)
// ================================================================================================ finisher form (planes / QKV)
// Round 6, second step. The dissection of the kernel above (profiles/r06g_ps_dissection.jsonl; w_1, M = 32768, random planes):
// K loop 132 us; + operand DMA by the loader waves 156; + epilogue arithmetic in the MFMA waves 175; + their stores 236 -- the
// store phase is the chip's write bandwidth (268 MB at 4.4 TB/s) with every MFMA wave blocked behind it, because a wave's stores
// are accepted at the rate the memory system drains them and all workgroups reach their epilogue together. What can absorb a
// tile's 128 KB of results while the NEXT tile is multiplied is the register file of the four auxiliary waves (256 registers each,
// ~40 used). So here the MFMA waves do no epilogue at all:
//   * at the end of a tile an MFMA wave dumps its RAW accumulators (128 registers) into LDS -- the stage buffer its last barrier
//     freed (48 KB) plus the 16 KB behind the ring: 64 KB = half a tile, so two rounds with a barrier pair each -- and starts the
//     next tile; its SIMD partner (wave + 4) reads the image back lane for lane into its own accumulator file;
//   * the partner finishes the tile while the next one is multiplied: one quarter (scale, bias, ReLU, plane split, 32 stores)
//     per stage interval, its VALU work in the shadow of the MFMA wave's matrix instructions;
//   * the auxiliary waves still own the stage ring, now ONE wave per stage (stage x: aux x & 3, all 48 pieces, three instructions
//     each): a wave has a stage in flight during two of every four intervals and finishes results only in the other two, so the
//     vmcnt(0) that publishes a stage never waits for a store younger than two intervals (stores and LDS-DMA share the counter).
// Same products, same k order, same epilogue expressions: bitwise the results of every other shape.
template <int OUT, bool RELU>   // OUT: 1 two fp16 planes of result * cscale, 2 the QKV / KV form
__global__ __launch_bounds__(512, 1) void gemm_f16x2_psf_kernel(Gemm2Args p) {
    typedef PsGeo G;
    constexpr unsigned SPARE = G::NSTG * G::STAGE_B;                        // 16 KB behind the ring
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
    const int abl = p.tile >> 4;            // measurement switches: bit 0 no operand DMA, bit 1 no hand-over / epilogue, bit 2 no stores
    // this lane's 16 bytes of a 1-KB dump row; MFMA wave w (and its partner w + 4) use 12 KB of the freed stage buffer + 4 KB of the spare
    const int pw = wave & 3;
    const unsigned dump_ring = (unsigned)pw * 12288u + (unsigned)lane * 16u, dump_spare = lds0 + SPARE + (unsigned)pw * 4096u + (unsigned)lane * 16u;

    if (wave >= 4) {
        // ======================================================================================== auxiliary waves
        const int L = wave - 4;
        const int prow = lane >> 2;
        const unsigned chunkb = (unsigned)(((lane & 3) ^ ((prow >> 2) & 3)) * 16);
        const char* const a_base = reinterpret_cast<const char*>(p.A);
        const char* const w_base = reinterpret_cast<const char*>(p.W);
        const size_t a_plane_b = p.a_plane * 2, w_plane_b = p.w_plane * 2;
        unsigned va[G::A_PIECES], vw[G::W_PIECES];
#pragma unroll
        for (int q = 0; q < G::W_PIECES; ++q) vw[q] = (unsigned)(16 * q + prow) * (unsigned)p.ldw * 2u + chunkb;
        // ---- loader state: my next stage (global index x = tile dti, stage ds)
        int x = L, dti = 0, ds = L;
        while (ds >= nk) { ds -= nk; ++dti; }
        int cur_ti = -1;
        const char* a_tile = a_base;
        const char* w_tile = w_base;
        bool inflight = false;
        const bool dma = (abl & 1) == 0;
        auto issue_next = [&]() __attribute__((always_inline)) {
            if (dma) {
                if (dti != cur_ti) {
                    int m0, n0;
                    tiles.at(dti, m0, n0);
                    cur_ti = dti;
                    a_tile = a_base + (size_t)m0 * p.lda * 2;
                    w_tile = w_base + (size_t)n0 * p.ldw * 2;
                    const int last = p.M - 16 - m0;                 // rows past M re-read the last valid piece (never stored)
#pragma unroll
                    for (int q = 0; q < G::A_PIECES; ++q) va[q] = (unsigned)((16 * q <= last ? 16 * q : last) + prow) * (unsigned)p.lda * 2u + chunkb;
                }
                const unsigned buf = lds0 + (unsigned)(x % G::NSTG) * G::STAGE_B;
                const char* const ab = a_tile + (size_t)ds * 64;
                const char* const wb = w_tile + (size_t)ds * 64;
                [&]<int... Q>(std::integer_sequence<int, Q...>) { (ps_piece<Q * 1024>(ab, va[Q], buf), ...); }(std::make_integer_sequence<int, G::A_PIECES>{});
                [&]<int... Q>(std::integer_sequence<int, Q...>) { (ps_piece<G::A_PLANE_B + Q * 1024>(ab + a_plane_b, va[Q], buf), ...); }(std::make_integer_sequence<int, G::A_PIECES>{});
                [&]<int... Q>(std::integer_sequence<int, Q...>) { (ps_piece<2 * G::A_PLANE_B + Q * 1024>(wb, vw[Q], buf), ...); }(std::make_integer_sequence<int, G::W_PIECES>{});
                [&]<int... Q>(std::integer_sequence<int, Q...>) { (ps_piece<2 * G::A_PLANE_B + G::W_PLANE_B + Q * 1024>(wb + w_plane_b, vw[Q], buf), ...); }(std::make_integer_sequence<int, G::W_PIECES>{});
            }
            inflight = true;
            x += 4; ds += 4;
            while (ds >= nk) { ds -= nk; ++dti; }
        };

        // ---- finisher state: the partner's raw accumulators of one tile (accumulator file), its coordinates, how much is left to do
        floatx16 img[4][2];
        int fm0 = 0, fn0 = 0, units_left = 0;
        const float oscale = p.oscale_dev ? p.oscale * *p.oscale_dev : p.oscale;
        const int hh = lane >> 5, idx = lane & 31, odd = lane & 1;
        // one quarter of a tile: arithmetic + stores. Plane tiles: unit U = 2 pp + tm = the column-tile pair (2 pp, 2 pp + 1) of row tile tm,
        // half-waves swapped so that a store covers whole rows (see the kernel above); V tiles: unit U = column tile U, both row tiles
        // (the argument struct through an opaque pointer: its fields are s_loaded where a quarter needs them instead of ~40 of them
        // living in SGPRs across the whole kernel -- the K-loop side needs the scalar file too)
        // (not &p: taking the parameter's address makes the compiler keep a private copy of the whole struct)
        typedef const Gemm2Args __attribute__((address_space(4)))* KernArgs;
#if defined(__HIP_DEVICE_COMPILE__)
        KernArgs pq = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
#else
        KernArgs pq = nullptr;
#endif
        asm volatile("" : "+s"(pq));
        auto unit = [&](auto Uc) __attribute__((always_inline)) {
            constexpr int U = decltype(Uc)::value;
            const auto& q = *pq;
            const int m0 = fm0, n0 = fn0;
            const int mw = m0 + L * 64;
            const int seg = OUT == 2 ? n0 / q.qkv_D + (q.kv_form ? 1 : 0) : 0;
            const bool stores = (abl & 4) == 0;
            if constexpr (OUT == 2) {
                if (seg == 2) {
                    constexpr int tn = U;
                    float v_mul = q.v_mul;
                    if (q.kv_mul_dev) v_mul *= q.kv_mul_dev[1];
                    const int vc0 = n0 - (q.kv_form ? 1 : 2) * q.qkv_D;
                    const int ldvt = q.ldvt, ldc = q.ldc;
                    unsigned short* const vt0 = q.VT + (size_t)(vc0 + 32 * tn + idx) * ldvt + mw + 8 * hh;
                    float* const c0 = q.C ? q.C + (size_t)(mw + 4 * hh) * ldc + vc0 + 32 * tn + idx : nullptr;
                    const size_t vt_plane = q.vt_plane;
                    const float bv = q.bias ? q.bias[n0 + 32 * tn + idx] : 0.f;
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm) {
                        const floatx16& a = img[tn][tm];
#pragma unroll
                        for (int Gq = 0; Gq < 2; ++Gq) {
                            if (mw + 32 * tm + 16 * Gq >= q.M) continue;
                            float t[8];
#pragma unroll
                            for (int j = 0; j < 8; ++j) t[j] = (a[8 * Gq + j] * oscale + bv) * v_mul;
                            uint4 h, l;
                            split2_pk(t[0], t[1], h.x, l.x);
                            split2_pk(t[2], t[3], h.y, l.y);
                            split2_pk(t[4], t[5], h.z, l.z);
                            split2_pk(t[6], t[7], h.w, l.w);
                            if (stores) {
                                unsigned short* vp = vt0 + 32 * tm + 16 * Gq;
                                *reinterpret_cast<uint4*>(vp) = h;
                                *reinterpret_cast<uint4*>(vp + vt_plane) = l;
                                if (c0) {
#pragma unroll
                                    for (int j = 0; j < 8; ++j)
                                        c0[(size_t)(32 * tm + 16 * Gq + 8 * (j >> 2) + (j & 3)) * ldc] = a[8 * Gq + j] * oscale + bv;
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    return;
                }
            }
            constexpr int pp = U >> 1, tm = U & 1;
            float pscale = q.cscale;
            unsigned short* pdst = q.C2;
            int pld = q.ldc2, pcol0 = n0;
            size_t pplane = q.c_plane;
            if constexpr (OUT == 2) {
                pscale = seg == 0 ? q.q_mul : (q.kv_mul_dev ? q.k_mul * q.kv_mul_dev[0] : q.k_mul);
                pdst = seg == 0 ? q.Qp : q.Kp;
                pld = q.qkv_D; pplane = q.qk_plane;
                pcol0 = n0 - (n0 / q.qkv_D) * q.qkv_D;
            }
            const int lc = 32 * hh + idx;                                   // this lane's column inside the pair's 64
            const float bv = q.bias ? q.bias[n0 + 64 * pp + lc] : 0.f;
            const floatx16& A = img[2 * pp][tm];
            const floatx16& B = img[2 * pp + 1][tm];
            // this lane's first element: row (mw + 32 tm + odd), its even column of the pair; rows advance by 2 pld elements
            unsigned short* const d0 = pdst + (size_t)(mw + 32 * tm + odd) * pld + pcol0 + 64 * pp + (lc & ~1);
            const size_t row2 = (size_t)2 * pld;
            const int M = q.M;
#pragma unroll
            for (int k = 0; k < 8; ++k) {                                   // register pair (2 k, 2 k + 1): rows rl, rl + 1 (A') and + 4 (B')
                float lo_[2], hi_[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float fa = A[2 * k + e], fb = B[2 * k + e];
                    unsigned xx = __builtin_bit_cast(unsigned, fa), yy = __builtin_bit_cast(unsigned, fb);
                    ps_swap32(xx, yy);                                       // xx = A'[r] (rows + 0), yy = B'[r] (rows + 4)
                    lo_[e] = __builtin_bit_cast(float, xx) * oscale + bv;
                    hi_[e] = __builtin_bit_cast(float, yy) * oscale + bv;
                    if constexpr (RELU) { lo_[e] = fmaxf(lo_[e], 0.f); hi_[e] = fmaxf(hi_[e], 0.f); }
                }
                if (mw + 32 * tm + 16 * (k >> 2) < M) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const float y0 = half ? hi_[0] : lo_[0], y1 = half ? hi_[1] : lo_[1];
                        const float got = __builtin_bit_cast(float, dpp_swap1(__builtin_bit_cast(unsigned, odd ? y0 : y1)));
                        const float ca = odd ? got : y0, cb = odd ? y1 : got;
                        unsigned h, l;
                        split2_pk(ca * pscale, cb * pscale, h, l);                  // (multiplier inside the split: see the kernel above)
                        if (stores) {
                            const int rl2 = 4 * (k >> 1) + (k & 1);          // (rows 0, 2, 8, 10, 16, 18, 24, 26) / 2
                            unsigned short* const d = d0 + (size_t)(rl2 + 2 * half) * row2;
                            *reinterpret_cast<unsigned*>(d) = h;
                            *reinterpret_cast<unsigned*>(d + pplane) = l;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto do_unit = [&](int u) __attribute__((always_inline)) {
            if (u == 0) unit(std::integral_constant<int, 0>{});
            else if (u == 1) unit(std::integral_constant<int, 1>{});
            else if (u == 2) unit(std::integral_constant<int, 2>{});
            else unit(std::integral_constant<int, 3>{});
        };
        // the partner's dump -> my accumulator file, lane for lane: round R = accumulator tiles 4 R .. 4 R + 3, 16 reads of 16 B
        auto take = [&](auto Rc, unsigned ring) __attribute__((always_inline)) {
            constexpr int R = decltype(Rc)::value;
            const unsigned ring_ = ring, spare_ = dump_spare;
            [&, ring_, spare_]<int... J>(std::integer_sequence<int, J...>) {
                ([&, ring_, spare_] {
                    constexpr int tt = 4 * R + (J >> 2), r0 = 4 * (J & 3);
                    ps_f4 v;
                    if constexpr (J < 12) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(ring_), "n"(J * 1024));
                    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(spare_), "n"((J - 12) * 1024));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));
                    floatx16& a = img[tt >> 1][tt & 1];
                    a[r0] = v[0]; a[r0 + 1] = v[1]; a[r0 + 2] = v[2]; a[r0 + 3] = v[3];
                }(), ...);
            }(std::make_integer_sequence<int, 16>{});
        };

        // stages 0 and 1 before the first barrier, stage 2 right after it; then around barrier g (the one in the middle of stage g):
        // the owner of stage g + 1 waits for it before the barrier, the owner of stage g + 3 issues it after the barrier -- except
        // behind a tile's LAST stage, whose buffer is the hand-over window first.
        // (ONE lexical call site of the quarter-tile code: it is ~400 instructions per quarter, and an un-inlined copy would force the
        // whole argument struct into memory. The loop runs one virtual tile past the end -- no barriers, no DMA -- to finish the last
        // tile's results.)
        if (L < 2 && x < total_stages) issue_next();
        if (L == 0) { glds_wait_all(); inflight = false; }
        __builtin_amdgcn_s_barrier();                                       // barrier -1
        if (L == 2 && x < total_stages) issue_next();
        int g = 0;
        for (int ti = 0; ti <= tiles.count; ++ti) {
            const bool real = ti < tiles.count;
            bool deferred = false;
            for (int s = 0; s < nk; ++s) {
                if (real) {
                    if (inflight && ((g + 1) & 3) == L) { glds_wait_all(); inflight = false; }
                    __builtin_amdgcn_s_barrier();                           // barrier g
                    if (((g + 3) & 3) == L && g + 3 < total_stages) {
                        if (s != nk - 1) issue_next(); else deferred = true;
                    }
                    ++g;
                }
                // a quarter of the previous tile, in an interval without a stage of mine in flight (or, running out of intervals, anyway:
                // the image must be free at the hand-over)
                if (units_left > 0 && (!inflight || nk - 1 - s < units_left)) { do_unit(4 - units_left); --units_left; }
            }
            if (!real || (abl & 2)) { if (deferred) issue_next(); continue; }
            const unsigned ring = lds0 + (unsigned)((g - 1) % G::NSTG) * G::STAGE_B + dump_ring;
            __builtin_amdgcn_s_barrier();                                   // X1: the partner has dumped accumulator tiles 0..3
            take(std::integral_constant<int, 0>{}, ring);
            __builtin_amdgcn_s_barrier();                                   // X2
            __builtin_amdgcn_s_barrier();                                   // X3: tiles 4..7 are in the window
            take(std::integral_constant<int, 1>{}, ring);
            __builtin_amdgcn_s_barrier();                                   // X4: the window is a stage buffer again
            if (deferred) issue_next();
            tiles.at(ti, fm0, fn0);
            units_left = 4;
        }
        return;
    }

    // ============================================================================================ MFMA waves
    const int hh = lane >> 5, idx = lane & 31;
    const unsigned fsw = (unsigned)((idx >> 2) & 3);
    const unsigned fa0 = lds0 + (unsigned)((wave * 64 + idx) * 64);
    const unsigned fw0 = lds0 + 2 * G::A_PLANE_B + (unsigned)(idx * 64);
    auto coff = [&](int st) { return (unsigned)(((2 * st + hh) ^ fsw) * 16); };
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
    for (int ti = 0; ti < tiles.count; ++ti) {
        floatx16 acc[4][2];
        auto kstep = [&](auto First, PsFrags& x, PsFrags& y, unsigned fa, unsigned fw) {
            [&]<int... Gp>(std::integer_sequence<int, Gp...>) {
                ([&] {
                    constexpr int g = Gp, P = g >> 3, t = g & 7, tn = t >> 1, tm = t & 1;
                    const f16x8& w = P == 1 ? x.wl[tn] : x.wh[tn];
                    const f16x8& a = P == 0 ? x.al[tm] : x.ah[tm];
                    if constexpr (P == 0 && decltype(First)::value) ps_mfma0(acc[tn][tm], a, w);
                    else ps_mfma(acc[tn][tm], a, w);
                    if constexpr ((g & 1) == 0) frag_read(std::integral_constant<int, (g >> 1)>{}, y, fa, fw);
                }(), ...);
            }(std::make_integer_sequence<int, 24>{});
            ps_reads_done(y);
        };
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
            __builtin_amdgcn_s_barrier();
            kstep(std::false_type{}, f1, f0, fa0 + nxt + coff(0), fw0 + nxt + coff(0));
            buf = nb;
        };
        stage(std::true_type{});
        for (int s = 1; s < nk; ++s) stage(std::false_type{});
        ps_settle(acc);
        if (abl & 2) {
            if (acc[0][0][0] == 123.456f && p.C) p.C[0] = acc[3][1][15];
            continue;
        }
        // ---- hand the raw accumulators to the partner: the buffer of the tile's last stage (read for the last time before its
        //      barrier; nothing is issued into it before X4) + the spare 16 KB, two rounds of four accumulator tiles
        const unsigned ring = lds0 + (unsigned)(buf == 0 ? G::NSTG - 1 : buf - 1) * G::STAGE_B + dump_ring;
        auto dump = [&](auto Rc) {
            constexpr int R = decltype(Rc)::value;
            const unsigned ring_ = ring, spare_ = dump_spare;
            [&, ring_, spare_]<int... J>(std::integer_sequence<int, J...>) {
                ([&, ring_, spare_] {
                    constexpr int tt = 4 * R + (J >> 2), r0 = 4 * (J & 3);
                    const floatx16& a = acc[tt >> 1][tt & 1];
                    const ps_f4 v = {a[r0], a[r0 + 1], a[r0 + 2], a[r0 + 3]};
                    if constexpr (J < 12) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(ring_), "v"(v), "n"(J * 1024) : "memory");
                    else asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(spare_), "v"(v), "n"((J - 12) * 1024) : "memory");
                }(), ...);
            }(std::make_integer_sequence<int, 16>{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        dump(std::integral_constant<int, 0>{});
        __builtin_amdgcn_s_barrier();               // X1
        __builtin_amdgcn_s_barrier();               // X2: the partner has read round 0
        dump(std::integral_constant<int, 1>{});
        __builtin_amdgcn_s_barrier();               // X3
        __builtin_amdgcn_s_barrier();               // X4
    }
}

template <int OUT, bool RELU>
int launch_psf(const Gemm2Args& a, hipStream_t stream) {
    constexpr int LDS = PsGeo::LDS_B + 16384;       // the ring + the spare 16 KB = the CU's whole LDS
    static PerDeviceOnce configured;
    if (!configured.done()) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_psf_kernel<OUT, RELU>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        configured.mark();
    }
    const int n_cu = device_cu_count() / 8 * 8;
    const long tiles = (long)ceil_div(a.M, PsGeo::BM) * (a.N / PsGeo::BN);
    int grid = n_cu;
    if (tiles < grid) grid = (int)((tiles + 7) / 8 * 8);
    hipLaunchKernelGGL((gemm_f16x2_psf_kernel<OUT, RELU>), dim3((unsigned)grid), dim3(512), LDS, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

#endif  // PF_MEASUREMENT_KERNELS

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
#if defined(PF_MEASUREMENT_KERNELS)
    // tile 12 (measurement library only, `make measure`): the finisher form for plane / QKV outputs -- measured and off
    // (profiles/r06h_ps_finisher_form.jsonl: w_1 330 us against 251 for the form above and 211 for the eight-wave 256 x 256 shape)
    const bool fin = (a.tile & 15) == 12 && a.K % 128 == 0;       // (>= 4 stage intervals per quarter-tile round)
    if (fin && a.qkv_D > 0) return launch_psf<2, false>(a, stream);
    if (fin && a.C2) return a.relu ? launch_psf<1, true>(a, stream) : launch_psf<1, false>(a, stream);
#endif
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
