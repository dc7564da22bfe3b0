// Flash-style self-attention with fp32-class results on the fp16 matrix cores (d_k = 128): both products run as THREE
// v_mfma_f32_32x32x16_f16 per operand pair on two-plane fp16 split operands (x * 2^e = hi + lo, gemm_f16x2.hip) with
// fp32 accumulators and fp32 softmax statistics. Its operands arrive READY: the QKV projection's epilogue
// (gemm_f16x2.hip, QKV form) writes Q (pre-multiplied by d_k^-0.5 like the reference) and K as row-major planes and V
// as TRANSPOSED planes V^T[d][row], so no per-tile conversion pass is left in this kernel -- K and V^T tiles go HBM -> LDS
// by asm-issued global_load_lds_dwordx4 one tile ahead and are consumed by ds_read_b128 as MFMA A operands.
//
// Reference semantics: funasr/models/sanm/attention.py:270-306,322-327 (scores, key mask -inf, softmax, mask 0, .V).
//
// Row layout: sequence b occupies rows [b Tp, b Tp + Tp), Tp % 16 == 0 (the encoder pads T; rows >= len hold finite
// don't-care values, masked here) -- or, packed, the rows koffs / qoffs name (the encoder keeps them 16-aligned so that a
// sequence's key tiles, hence its bits, do not depend on where it lies). V^T columns are rows with bits 2 and 3 of the
// index swapped, so that the 8 keys whose
// scores one lane's S^T accumulator registers hold ({0..3, 8..11} + 4 (lane >> 5) of a 16-key step) are one 16-B chunk.
//
// One workgroup = 8 waves = 256 queries of one (sequence, head); a wave owns 32 queries x d_k, a lane one query.
// Per 32-key tile and wave: S^T[key][q] = K Q^T (A = K planes from LDS, B = Q planes in registers): 24 MFMAs; online
// softmax per lane; O^T[d][q] += V^T[d][key] P^T[key][q] (A = V^T planes from LDS, B = P split in registers): 24 MFMAs.
// 48 MFMAs per tile against 96 in attention_split3.hip, and no split pass.
#include "common.h"

namespace pf {

namespace {

constexpr int DK = 128, KT = 32;
constexpr int KP_B = KT * DK * 2;                        // one K plane tile: 32 keys x 256 B
constexpr int VP_B = DK * KT * 2;                        // one V^T plane tile: 128 d x 64 B
constexpr int STAGE_B = 2 * KP_B + 2 * VP_B;             // 32 KB
constexpr int OLD = DK + 4;                              // epilogue slab row (floats)
constexpr int SLAB_B = 8 * 32 * OLD * 4;                 // 135168
constexpr int LDS_BYTES = SLAB_B > 3 * STAGE_B ? SLAB_B : 3 * STAGE_B;
constexpr float P_SCALE = 1024.f;                        // probabilities are split as p * 2^10 (p <= 1)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Schedules (Attn2Args.variant; all three are tested against float64 attention, tests/test_kernels_f16x2_gpu.py):
// 3 (the engine's choice), LAZY: the running maximum of the online softmax is only moved when a tile's maximum exceeds it by
//   more than LAZY_TAU -- the probabilities are then at most e^TAU instead of 1 (p * 2^10 stays far inside fp16) and the 64
//   accumulator multiplications per tile are skipped whenever no query of the wave moved its maximum; exponentials in base 2
//   (log2 e folded into the score scale), no key-mask compares on tiles whose keys are all valid. 110 us against 122 / 124 for
//   variants 1 / 0 at B = 64, T = 512 (profiles/r02l_bench_attn2.json).
// 1, PIPE: software-pipelined -- the S^T products of tile t+1 are issued BEFORE the softmax of tile t, in one scheduling
//   region, so the matrix pipe works under the softmax's VALU (three LDS stages instead of two); bitwise equal to variant 0.
// 0: the plain schedule.
// (Measured in round 2 and removed: a four-wave workgroup with two workgroups per CU, 127 us; LAZY + PIPE, 114 us.)
constexpr float LAZY_TAU = 3.0f;
template <int PIPE, bool LAZY = false>
__global__ __launch_bounds__(512, 2) void attention_f16x2_kernel(Attn2Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NSTAGE = PIPE ? 3 : 2;
    constexpr int QB = 256;                                  // queries per workgroup: 8 waves x 32
    constexpr int NV = 1;                                    // DMA pieces per wave and plane tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, idx = lane & 31;
    // workgroup -> (sequence, head, query block). XCD-aware order (xcd_nqb > 0, 1-D grid): the hardware deals workgroup L to
    // XCD L % 8, so the xcd_nqb query blocks of one (sequence, head) get consecutive slots of ONE XCD and fetch that pair's
    // K / V^T tiles through the same L2 (the plain 3-D order puts neighbouring query blocks on different XCDs: every tile
    // came from HBM once per query block, 1.7x the compulsory bytes at T = 512)
    int b, head, qblk;
    if (p.xcd_nqb > 0) {
        const int L = blockIdx.x, j = L >> 3;
        const int pair = (j / p.xcd_nqb) * 8 + (L & 7);
        if (pair >= p.B * p.H) return;
        qblk = j % p.xcd_nqb; head = pair % p.H; b = pair / p.H;
    } else {
        b = blockIdx.z; head = blockIdx.y; qblk = blockIdx.x;
    }
    // cross-attention: Q / O rows per sequence differ from K's; packed queries: sequence b owns rows [qoffs[b], qoffs[b+1])
    const int Tq = p.qoffs ? p.qoffs[b + 1] - p.qoffs[b] : (p.Tq > 0 ? p.Tq : p.Tp);
    if (qblk * QB >= Tq) return;                         // (uniform per workgroup) nothing to do for this query block
    const int q = qblk * QB + wave * 32 + idx;
    const int qc = q < Tq ? q : Tq - 1;
    const int klen = p.klens[b];
    // keys: rows [b Tp, b Tp + klen), or with koffs any rows [koffs[b], koffs[b] + klen): tiles then start at the 16-row
    // group below the first key (the V^T column order is per 16-row group of the GLOBAL row index) and the kskip rows in
    // front -- the tail of the neighbouring sequence, finite values -- are masked like the rows behind the last key
    const int kstart = p.koffs ? p.koffs[b] : b * p.Tp;
    const size_t row0 = (size_t)(kstart & ~15);
    const int kskip = kstart & 15, kend = kskip + klen;
    const size_t qrow0 = p.qoffs ? (size_t)p.qoffs[b] : (size_t)b * Tq;

    // ---- Q planes: step s covers d in [16 s, 16 s + 16); half-wave h holds the 8 d's of chunk 2 s + h
    f16x8 qf[2][8];
    {
        const unsigned short* qp = p.Q + (qrow0 + qc) * p.ldq + head * DK + hh * 8;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int s = 0; s < 8; ++s)
                qf[pl][s] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(qp + pl * p.q_plane + 16 * s));
    }

    floatx16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // ---- DMA of one tile: 32 pieces of 1 KB; wave w issues K plane 0 / 1 rows 4w..4w+3 and V^T plane 0 / 1 rows
    //      16w..16w+15. K row r (256 B): chunk c at c ^ (r & 15); V^T row d (64 B): chunk c at c ^ ((d >> 2) & 3)
    const unsigned short* ksrc[NV];
    const unsigned short* vsrc[NV];
    unsigned lds_w[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int vw = wave * NV + v;                        // virtual wave 0..7
        const int kr = vw * 4 + (lane >> 4);
        ksrc[v] = p.K + (row0 + kr) * p.ldk + head * DK + (((lane & 15) ^ (kr & 15)) * 8);
        const int dr = vw * 16 + (lane >> 2);
        vsrc[v] = p.VT + (size_t)(head * DK + dr) * p.ldvt + row0 + (((lane & 3) ^ ((dr >> 2) & 3)) * 8);
        lds_w[v] = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (unsigned)vw * 1024);
    }
    auto stage = [&](int kt) {
        const size_t ko = (size_t)kt * KT * p.ldk;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const unsigned base = lds_w[v] + (unsigned)(kt % NSTAGE) * STAGE_B;
            glds16(ksrc[v] + ko, base);
            glds16(ksrc[v] + p.k_plane + ko, base + KP_B);
            glds16(vsrc[v] + kt * KT, base + 2 * KP_B);
            glds16(vsrc[v] + p.vt_plane + kt * KT, base + 2 * KP_B + VP_B);
        }
    };

    const float sscale = p.sscale_dev ? p.sscale * *p.sscale_dev : p.sscale;   // 2^-(e_q + e_k)
    // LAZY variants work in the base-2 domain: log2(e) is folded into the score scale and exp is a bare v_exp_f32
    const float sscale2 = LAZY ? sscale * 1.4426950408889634f : sscale;
#define PF_EXP(x_) (LAZY ? __builtin_amdgcn_exp2f(x_) : __expf(x_))
    constexpr float PSC = LAZY ? 1.f : P_SCALE;              // (a multiplication by the literal 1 is folded away)
    const int ntiles = (kend + KT - 1) / KT;

    // S^T tile (32 keys x 32 queries): small products into sa, hi*hi into sb (no dependent MFMA pairs)
#define PF_QK(KT_, SA, SB)                                                                                            \
    {                                                                                                                 \
        const unsigned char* kp = smem + ((KT_) % NSTAGE) * STAGE_B + idx * 256;                                      \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) { SA[r] = 0.f; SB[r] = 0.f; }                                  \
        _Pragma("unroll") for (int st = 0; st < 8; ++st) {                                                            \
            const int co = ((2 * st + hh) ^ (idx & 15)) * 16;                                                         \
            const f16x8 kh = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(kp + co));                     \
            const f16x8 kl = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(kp + KP_B + co));              \
            SA = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qf[0][st], SA, 0, 0, 0);                                  \
            SB = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qf[0][st], SB, 0, 0, 0);                                  \
            SA = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qf[1][st], SA, 0, 0, 0);                                  \
        }                                                                                                             \
    }
    // online softmax for query (lane & 31); this lane holds keys k0 + (r&3) + 8(r>>2) + 4h of the tile; s <- p
#define PF_SOFTMAX(K0_, SA, SB)                                                                                       \
    {                                                                                                                 \
        float mx = -INFINITY;                                                                                         \
        if (LAZY && (K0_) >= kskip && (K0_) + KT <= kend) {      /* (wave-uniform) every key of the tile is valid */   \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                          \
                s[r] = (SA[r] + SB[r]) * sscale2;                                                                     \
                mx = fmaxf(mx, s[r]);                                                                                 \
            }                                                                                                         \
        } else {                                                                                                      \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                          \
                const int key = (K0_) + (r & 3) + 8 * (r >> 2) + 4 * hh;                                              \
                s[r] = (unsigned)(key - kskip) < (unsigned)klen ? (SA[r] + SB[r]) * sscale2 : -INFINITY;              \
                mx = fmaxf(mx, s[r]);                                                                                 \
            }                                                                                                         \
        }                                                                                                             \
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                                                                       \
        float m_new = fmaxf(m_run, mx);                                                                               \
        bool moved = true;                                                                                            \
        if constexpr (LAZY) {                       /* keep a stale maximum while the true one is < e^TAU above it */  \
            moved = m_new > m_run + LAZY_TAU * 1.4426950408889634f;                                                                         \
            m_new = moved ? m_new : m_run;                                                                            \
        }                                                                                                             \
        const float alpha = moved ? PF_EXP(m_run - m_new) : 1.f;                                                      \
        float psum = 0.f;                                                                                             \
        const float m_off = LAZY ? m_new - 10.f : m_new;        /* LAZY: p comes out as p * 2^10, the split's scale */  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                              \
            s[r] = PF_EXP(s[r] - m_off);                                                                              \
            psum += s[r];                                                                                             \
        }                                                                                                             \
        psum += __shfl_xor(psum, 32, 64);                                                                             \
        l_run = l_run * alpha + psum;                                                                                 \
        m_run = m_new;                                                                                                \
        if (!LAZY || __any(moved)) {                                                                                  \
            _Pragma("unroll") for (int d = 0; d < 4; ++d) _Pragma("unroll") for (int r = 0; r < 16; ++r) o[d][r] *= alpha; \
        }                                                                                                             \
    }
    // O^T += V^T P^T. step st uses this lane's registers r in [8st, 8st+8): keys 16st + 4h + {0..3, 8..11}
#define PF_PV(KT_)                                                                                                    \
    {                                                                                                                 \
        const unsigned char* vp = smem + ((KT_) % NSTAGE) * STAGE_B + 2 * KP_B + idx * 64;                            \
        _Pragma("unroll") for (int st = 0; st < 2; ++st) {                                                            \
            uint4 ph, pl;                                                                                             \
            split2_pk(s[8 * st + 0] * PSC, s[8 * st + 1] * PSC, ph.x, pl.x);                                  \
            split2_pk(s[8 * st + 2] * PSC, s[8 * st + 3] * PSC, ph.y, pl.y);                                  \
            split2_pk(s[8 * st + 4] * PSC, s[8 * st + 5] * PSC, ph.z, pl.z);                                  \
            split2_pk(s[8 * st + 6] * PSC, s[8 * st + 7] * PSC, ph.w, pl.w);                                  \
            const f16x8 Ph = __builtin_bit_cast(f16x8, ph), Pl = __builtin_bit_cast(f16x8, pl);                       \
            const int co = ((2 * st + hh) ^ ((idx >> 2) & 3)) * 16;                                                   \
            f16x8 vf[4][2];                                                                                           \
            _Pragma("unroll") for (int d = 0; d < 4; ++d) _Pragma("unroll") for (int pn = 0; pn < 2; ++pn)            \
                vf[d][pn] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(vp + pn * VP_B + d * 32 * 64 + co)); \
            _Pragma("unroll") for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[d][1], Ph, o[d], 0, 0, 0); \
            _Pragma("unroll") for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[d][0], Pl, o[d], 0, 0, 0); \
            _Pragma("unroll") for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[d][0], Ph, o[d], 0, 0, 0); \
        }                                                                                                             \
    }

    float s[16];
    const bool active = qblk * QB + wave * 32 < Tq;          // (wave-uniform)
    if constexpr (PIPE == 0) {
        stage(0);
        for (int kt = 0; kt < ntiles; ++kt) {
            glds_wait_all();
            __syncthreads();                    // tile kt landed; every wave is done with tile kt-1 (the other buffer)
            if (kt + 1 < ntiles) stage(kt + 1);
            // a wave whose 32 queries all lie behind the sequence's last query (T = 171 -> 176 rows in a 256-query block: waves 6, 7;
            // the decoder's ~170 tokens per clip likewise) only stages tiles and keeps the barriers: its matrix-pipe time goes to the
            // other workgroup of the CU
            if (active) {
                floatx16 sa, sb;
                PF_QK(kt, sa, sb)
                PF_SOFTMAX(kt * KT, sa, sb)
                PF_PV(kt)
            }
        }
    } else {
        floatx16 ca, cb, na, nb;
        stage(0);
        if (ntiles > 1) stage(1);
        glds_wait_all();
        __syncthreads();
        PF_QK(0, ca, cb)
        for (int kt = 0; kt < ntiles; ++kt) {
            // tile kt + 1 (issued one iteration ago) must have landed for S_next; its buffer and tile kt's are live, the third
            // buffer (tile kt - 1, fully consumed before this barrier) takes tile kt + 2
            if (kt > 0) { glds_wait_all(); __syncthreads(); }
            if (kt + 2 < ntiles) stage(kt + 2);
            // branch-free (one scheduling region): the last iteration recomputes its own tile's scores and drops them
            const int kn = kt + 1 < ntiles ? kt + 1 : kt;
            PF_QK(kn, na, nb)
            PF_SOFTMAX(kt * KT, ca, cb)
            // interleave: one MFMA, then a few of the softmax's VALU ops, so the matrix pipe runs under them
#pragma unroll
            for (int g = 0; g < 24; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
            }
            PF_PV(kt)
            ca = na; cb = nb;
        }
    }
#undef PF_QK
#undef PF_EXP
#undef PF_SOFTMAX
#undef PF_PV

    // ---- epilogue: O^T (lane = query, registers = d) -> wave-private slab [32 q][128 d] -> row-wise 16-B plane pieces
    __syncthreads();
    float* slab = reinterpret_cast<float*>(smem) + wave * (32 * OLD);
    {
        // oscale = 2^(e_ctx - e_v) / P_SCALE; in the LAZY variants l_run was summed in the p * 2^10 units as well
        const float inv = (LAZY ? p.oscale * P_SCALE : p.oscale) / l_run;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 t = make_float4(o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
                *reinterpret_cast<float4*>(slab + idx * OLD + d * 32 + 8 * g + 4 * hh) = t;
            }
    }
    // same wave wrote and reads its slab: no workgroup barrier needed, only the LDS counter (the compiler waits)
    const int qw0 = qblk * QB + wave * 32;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int rr = it * 4 + (lane >> 4), c8 = (lane & 15) * 8;
        const float4 a = *reinterpret_cast<const float4*>(slab + rr * OLD + c8);
        const float4 c = *reinterpret_cast<const float4*>(slab + rr * OLD + c8 + 4);
        if (qw0 + rr < Tq) {
            uint4 h, l;
            split2_pk(a.x, a.y, h.x, l.x);
            split2_pk(a.z, a.w, h.y, l.y);
            split2_pk(c.x, c.y, h.z, l.z);
            split2_pk(c.z, c.w, h.w, l.w);
            unsigned short* op = p.O + (qrow0 + qw0 + rr) * p.ldo + head * DK + c8;
            *reinterpret_cast<uint4*>(op) = h;
            *reinterpret_cast<uint4*>(op + p.o_plane) = l;
        }
    }
}

}  // namespace

int launch_attention_f16x2(const Attn2Args& a, hipStream_t stream) {
    PF_REQUIRE(a.B > 0 && a.H > 0 && (a.koffs || (a.Tp > 0 && a.Tp % 16 == 0)), "attention_f16x2: rows per sequence must be a positive multiple of 16");
    PF_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldvt % 8 == 0 && a.ldo % 8 == 0 && a.q_plane % 8 == 0 &&
               a.k_plane % 8 == 0 && a.vt_plane % 8 == 0 && a.o_plane % 8 == 0, "attention_f16x2: strides % 8");
    PF_REQUIRE(((uintptr_t)a.Q & 15) == 0 && ((uintptr_t)a.K & 15) == 0 && ((uintptr_t)a.VT & 15) == 0 && ((uintptr_t)a.O & 15) == 0,
               "attention_f16x2: 16-B alignment");
    static PerDeviceOnce configured;
    if (!configured.done()) {
#if defined(PF_MEASUREMENT_KERNELS)
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f16x2_kernel<0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f16x2_kernel<1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
#endif
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_f16x2_kernel<0, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        configured.mark();
    }
    const int rows = a.Tq > 0 ? a.Tq : a.Tp;
    Attn2Args k = a;
    // variant 3 (lazy rescale) is the product's schedule; 0 / 1 are in the measurement library only (`make measure`; profiles/r02l_bench_attn2.json)
#if defined(PF_MEASUREMENT_KERNELS)
    PF_REQUIRE(a.variant == 0 || a.variant == 1 || a.variant == 3, "attention_f16x2: variant must be 0, 1 or 3");
#else
    PF_REQUIRE(a.variant == 3, "attention_f16x2: variant must be 3 (variants 0 / 1: the measurement library, make measure)");
#endif
    const int nqb = ceil_div(rows, 256);
    dim3 grid(nqb, a.H, a.B);
    // more than one query block per (sequence, head): XCD-aware 1-D order (xcd_nqb < 0 keeps the plain order: measurement hook)
    k.xcd_nqb = (a.xcd_nqb >= 0 && nqb > 1) ? nqb : 0;
    if (k.xcd_nqb > 0) grid = dim3((unsigned)(ceil_div(a.B * a.H, 8) * 8 * nqb), 1, 1);
    if (a.variant == 3)
        hipLaunchKernelGGL((attention_f16x2_kernel<0, true>), grid, dim3(512), LDS_BYTES, stream, k);
#if defined(PF_MEASUREMENT_KERNELS)
    else if (a.variant == 1)
        hipLaunchKernelGGL(attention_f16x2_kernel<1>, grid, dim3(512), LDS_BYTES, stream, k);
    else
        hipLaunchKernelGGL(attention_f16x2_kernel<0>, grid, dim3(512), LDS_BYTES, stream, k);
#endif
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pf
