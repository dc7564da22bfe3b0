// The full-row form of gemm_f16x2_row.hip with a block of 32 WM rows (WM = 3: 96 rows, WM = 4: 128 rows) instead of 128 only.
// The row kernel's one block per CU is the quantum of its load balance: M = 22 528 rows (SenseVoiceSmall, 128 x 10 s) are 176
// blocks of 128 rows on 256 CUs -- 0.69 of a round, 80 CUs idle -- but 235 blocks of 96 rows: every CU busy for three quarters
// of the time. Same projections and epilogues as gemm_f16x2_row.hip:
//     x = residual + (dropout(linear_out(ctx)) + fsmn_memory)       funasr/models/sanm/encoder.py:120-137
//     x = residual + feed_forward(norm2(x));  next block: norm1(x)   funasr/models/sanm/encoder.py:141-146, :96-98
//     LayerNorm                                                       funasr/models/transformer/layer_norm.py:13-38
//     fsmn_memory = mask (conv_k11(pad(mask v)) + mask v)             funasr/models/sanm/attention.py:216-239
//
// Design (gfx950): 8 waves as 1 (M) x 8 (N): a wave owns ALL 32 WM rows x 64 columns = WM x 2 MFMA tiles of 32 x 32 (WM = 4:
// 12 ds_read_b128 + 24 MFMAs per 16-deep step like the 2 x 4 grid; WM = 3: 10 + 18), 32-deep K stages of 64-B LDS rows --
// 2 planes x (32 WM + 512) rows, double buffered (WM = 3: 152 KB) -- by asm-issued global_load_lds_dwordx4 pieces (the A
// planes are 4 WM pieces: waves 0 .. 4 WM - 9 carry two of them). Epilogue: per 32-row tile the waves 2 c, 2 c + 1 put their
// 32 x 64 halves side by side in one LDS slab (columns 128 c ..), then wave 2 c + q finishes rows 16 q .. 16 q + 15 of the tile
// with the lane layout of the other row kernel (half-wave = 8 rows, lane = 4 columns of the 128): so bias, addends, FSMN
// window, LayerNorm statistics ((c 0 + c 2) + (c 1 + c 3), then the xor butterfly over the 32 lanes) and the plane stores
// are the SAME operations in the same order -- the result is bitwise the other row kernel's (tested), so the launcher may
// pick the block height by the batch's row count.
#include "common.h"

namespace pf {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int R8_BN = 512, R8_KS = 32, R8_ROWB = 64;
constexpr int R8_B_PLANE_B = R8_BN * R8_ROWB;                        // 32 KB
constexpr int R8_ELD = 132;                                          // slab row (floats): 128 columns + 4
constexpr int R8_SLAB_B = 4 * 32 * R8_ELD * 4;                       // four wave pairs x 32 rows: 67 584 B
constexpr int R8_FS_KS = 11, R8_FS_LP = 5;
constexpr int R8_FSW_OFF_B = R8_SLAB_B;                              // FSMN taps [11][512] floats behind the slabs

template <int WM> struct R8Geo {
    static constexpr int BM = 32 * WM;
    static constexpr int A_PLANE_B = BM * R8_ROWB;
    static constexpr int STAGE_B = 2 * (A_PLANE_B + R8_B_PLANE_B);
    static constexpr int LDS_B = 2 * STAGE_B;
    static constexpr int NA = 4 * WM;                                // A pieces of 1 KB (16 rows of one plane) per stage
    static constexpr int P_FLOATS = BM * 128;                        // statistics exchange [row][wave pair][lane]
    static_assert(LDS_B <= 163840, "two stages must fit the CU's LDS");
    static_assert(R8_FSW_OFF_B + R8_FS_KS * R8_BN * 4 <= LDS_B && (P_FLOATS + 2 * BM) * 4 <= LDS_B, "epilogue LDS");
    static_assert(NA > 8 && NA <= 16, "a wave carries one or two A pieces");
};

// MODE bit 0: R1 addend, bit 1: R2 addend, bit 2: FSMN memory of p.fs_v as first addend (excludes bit 0); LN: LayerNorm epilogue
template <int WM, int MODE, bool LN, bool A_NT>
__global__ __launch_bounds__(512, 2) void gemm_f16x2_row8_kernel(GemmRowArgs p) {
    typedef R8Geo<WM> G;
    constexpr int BM = G::BM, A_PLANE_B = G::A_PLANE_B, STAGE_B = G::STAGE_B, NA = G::NA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int m0 = blockIdx.x * BM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pc = wave >> 1, q = wave & 1;               // wave pair = 128-column group, half of a 32-row tile in the epilogue
    const int hh = lane >> 5, idx = lane & 31;

    // ---- DMA sources. Stage = [A plane 0 | A plane 1 | W plane 0 | W plane 1], pieces of 1 KB = 16 rows of one plane, linear.
    //      A piece a (plane a / (2 WM), rows 16 (a % (2 WM)) ..): wave w carries a = w and, if it exists, a = w + 8.
    //      W piece v = w + 8 j (plane v / 32, columns 16 (v % 32) ..), j = 0 .. 7. Lane l lands at row l / 4, physical
    //      chunk l % 4 and fetches the logical chunk the read-side swizzle expects there.
    const bool has_a1 = wave + 8 < NA;
    const unsigned short* src_a[2];
    const unsigned short* src_w[8];
    {
        const int prow = lane >> 2;
        const int chunk = (lane & 3) ^ ((prow >> 2) & 3);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int a = (wave + 8 * j) < NA ? wave + 8 * j : wave;
            int row = m0 + (a % (2 * WM)) * 16 + prow;
            row = row < p.M ? row : p.M - 1;
            src_a[j] = p.A + (size_t)(a / (2 * WM)) * p.a_plane + (size_t)row * p.lda + chunk * 8;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int v = wave + 8 * j;
            const int col = (v & 31) * 16 + prow;
            src_w[j] = p.W + (size_t)(v >> 5) * p.w_plane + (size_t)col * p.ldw + chunk * 8;
        }
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    auto piece_a = [&](int j, int buf, int kt) {
        const unsigned dst = lds0 + (unsigned)buf * STAGE_B + (unsigned)(wave + 8 * j) * 1024;
        if (A_NT) glds16_nt(src_a[j] + kt * R8_KS, dst);
        else glds16(src_a[j] + kt * R8_KS, dst);
    };
    auto piece_w = [&](int j, int buf, int kt) {
        glds16(src_w[j] + kt * R8_KS, lds0 + (unsigned)buf * STAGE_B + 2 * A_PLANE_B + (unsigned)(wave + 8 * j) * 1024);
    };

    floatx16 acc[WM][2];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    const int f = (idx >> 2) & 3;
    const int aoff = idx * R8_ROWB;
    const int boff = 2 * A_PLANE_B + (wave * 64 + idx) * R8_ROWB;
    int coff[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) coff[st] = ((2 * st + hh) ^ f) * 16;

    const int nk = p.K / R8_KS;
    piece_a(0, 0, 0);
    if (has_a1) piece_a(1, 0, 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) piece_w(j, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        glds_wait_all();
        __syncthreads();
        const bool nxt = kt + 1 < nk;
        const int nb = (kt + 1) & 1;
        const unsigned char* sb = smem + (kt & 1) * STAGE_B;
#define R8_PROD(AF, BF, PA, PB)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < WM; ++i) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                   \
        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AF[i][PA], BF[jj][PB], acc[i][jj], 0, 0, 0)
#define R8_LOAD(AF, BF, S)                                                                                            \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                                                                \
        _Pragma("unroll") for (int i = 0; i < WM; ++i)                                                                \
            AF[i][pl] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(sb + pl * A_PLANE_B + aoff + i * 32 * R8_ROWB + coff[S]));     \
        _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                              \
            BF[jj][pl] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(sb + pl * R8_B_PLANE_B + boff + jj * 32 * R8_ROWB + coff[S])); \
    }
        // both k-steps' fragments are requested before the first MFMA, the next stage's DMA pieces early; per k-step the two
        // small products first, hi * hi last: the order of gemm_f16x2_kernel and of the other row kernel
        f16x8 a0[WM][2], b0[2][2], a1[WM][2], b1[2][2];
        R8_LOAD(a0, b0, 0)
        R8_LOAD(a1, b1, 1)
        if (nxt) { piece_a(0, nb, kt + 1); if (has_a1) piece_a(1, nb, kt + 1); piece_w(0, nb, kt + 1); piece_w(1, nb, kt + 1); }
        R8_PROD(a0, b0, 1, 0); if (nxt) { piece_w(2, nb, kt + 1); piece_w(3, nb, kt + 1); }
        R8_PROD(a0, b0, 0, 1); if (nxt) { piece_w(4, nb, kt + 1); piece_w(5, nb, kt + 1); }
        R8_PROD(a0, b0, 0, 0); if (nxt) { piece_w(6, nb, kt + 1); piece_w(7, nb, kt + 1); }
        R8_PROD(a1, b1, 1, 0);
        R8_PROD(a1, b1, 0, 1);
        R8_PROD(a1, b1, 0, 0);
#undef R8_PROD
#undef R8_LOAD
    }

    // ---- epilogue, part 1, one 32-row tile at a time: accumulators (C/D layout of the 32 x 32 MFMA: col = lane & 31, row =
    //      (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) -> the wave pair's slab [32][128] -> float4 pieces of rows: wave 2 c + q
    //      takes rows 16 q .. 16 q + 15 of the tile, its half-wave h rows 8 h .. 8 h + 7 of those, lane c4 the columns
    //      4 c4 .. 4 c4 + 3 of the pair's 128; the finished values stay in registers
    constexpr bool HAS_R1 = (MODE & 1) != 0, HAS_R2 = (MODE & 2) != 0, FSMN = (MODE & 4) != 0;
    static_assert(!(HAS_R1 && FSMN), "the FSMN form computes the first addend");
    float* smf = reinterpret_cast<float*>(smem);
    float* slab = smf + pc * (32 * R8_ELD);
    const float* fsw = reinterpret_cast<const float*>(smem + R8_FSW_OFF_B);      // [tap][512]
    const int c4 = idx, rsub = hh;
    const int col = pc * 128 + c4 * 4;
    const float oscale = p.oscale_dev ? p.oscale * *p.oscale_dev : p.oscale;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + col);
    float4 ov[WM][8];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        __syncthreads();                       // the stage buffers (i = 0) / the previous tile's slab reads are done
        if constexpr (FSMN) {
            if (i == 0) {
                // taps [512][11] -> LDS [11][512]: thread = channel (consecutive threads, consecutive LDS words)
                float* dst = reinterpret_cast<float*>(smem + R8_FSW_OFF_B);
#pragma unroll
                for (int j = 0; j < R8_FS_KS; ++j) dst[j * R8_BN + tid] = p.fs_w[(size_t)tid * R8_FS_KS + j];
            }
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[((r & 3) + 8 * (r >> 2) + 4 * hh) * R8_ELD + q * 64 + jj * 32 + idx] = acc[i][jj][r];
        __syncthreads();
        const int rl0 = q * 16 + rsub * 8;                 // first of this half-wave's 8 rows inside the tile
        const int row0 = m0 + i * 32 + rl0;
        if constexpr (FSMN) {
            // the 8 rows lie inside one 16-row group of one sequence: valid v rows [lo, hi); everything else counts as zero
            // input and output rows >= hi get no memory. Four output rows at a time over a sliding window of 14 v rows.
            const int grp = row0 >> 4;
            const bool gok = row0 < p.M;
            const int lo = gok ? p.fs_lo[grp] : 0, hi = gok ? p.fs_hi[grp] : 0;
            float4 win[8 + R8_FS_KS - 1];
            auto load_row = [&](int k) {
                const int vr = row0 - R8_FS_LP + k;
                const bool ok = vr >= lo && vr < hi;
                const float4 t = *reinterpret_cast<const float4*>(p.fs_v + (size_t)(ok ? vr : (lo < hi ? lo : 0)) * p.ldfv + col);
                win[k] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            };
#pragma unroll
            for (int k = 0; k < R8_FS_KS - 1; ++k) load_row(k);
#pragma unroll
            for (int h4 = 0; h4 < 2; ++h4) {
                float4 r2[4], fa[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    load_row(R8_FS_KS - 1 + h4 * 4 + t);
                    fa[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (HAS_R2) {
                        const int row = row0 + h4 * 4 + t;
                        r2[t] = *reinterpret_cast<const float4*>(p.R2 + (size_t)(row < p.M ? row : p.M - 1) * p.ldr2 + col);
                    }
                }
#pragma unroll
                for (int j = 0; j < R8_FS_KS; ++j) {
                    const float4 wj = *reinterpret_cast<const float4*>(fsw + j * R8_BN + col);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float4 x = win[h4 * 4 + t + j];
                        fa[t].x = fmaf(wj.x, x.x, fa[t].x); fa[t].y = fmaf(wj.y, x.y, fa[t].y);
                        fa[t].z = fmaf(wj.z, x.z, fa[t].z); fa[t].w = fmaf(wj.w, x.w, fa[t].w);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = row0 + h4 * 4 + t;
                    const float4 c = win[h4 * 4 + t + R8_FS_LP];                  // the (masked) input row itself
                    float4 mem = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < hi) mem = make_float4(fa[t].x + c.x, fa[t].y + c.y, fa[t].z + c.z, fa[t].w + c.w);
                    const float4 vt = *reinterpret_cast<const float4*>(slab + (rl0 + h4 * 4 + t) * R8_ELD + c4 * 4);
                    float o[4] = {vt.x * oscale + bias4.x, vt.y * oscale + bias4.y, vt.z * oscale + bias4.z,
                                  vt.w * oscale + bias4.w};
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
                    }
                    o[0] = o[0] + mem.x; o[1] = o[1] + mem.y; o[2] = o[2] + mem.z; o[3] = o[3] + mem.w;
                    if constexpr (HAS_R2) { o[0] = r2[t].x + o[0]; o[1] = r2[t].y + o[1]; o[2] = r2[t].z + o[2]; o[3] = r2[t].w + o[3]; }
                    const float4 o4 = make_float4(o[0], o[1], o[2], o[3]);
                    ov[i][h4 * 4 + t] = o4;
                    if (p.C && row < p.M) *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + col) = o4;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            float4 v[8], r1[8], r2[8];
#pragma unroll
            for (int t = 0; t < 8; ++t)
                v[t] = *reinterpret_cast<const float4*>(slab + (rl0 + t) * R8_ELD + c4 * 4);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int row = row0 + t;
                const int rr = row < p.M ? row : p.M - 1;
                if constexpr (HAS_R1) r1[t] = *reinterpret_cast<const float4*>(p.R1 + (size_t)rr * p.ldr1 + col);
                if constexpr (HAS_R2) r2[t] = *reinterpret_cast<const float4*>(p.R2 + (size_t)rr * p.ldr2 + col);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int row = row0 + t;
                float o[4] = {v[t].x * oscale + bias4.x, v[t].y * oscale + bias4.y, v[t].z * oscale + bias4.z,
                              v[t].w * oscale + bias4.w};
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
                }
                if constexpr (HAS_R1) { o[0] = o[0] + r1[t].x; o[1] = o[1] + r1[t].y; o[2] = o[2] + r1[t].z; o[3] = o[3] + r1[t].w; }
                if constexpr (HAS_R2) { o[0] = r2[t].x + o[0]; o[1] = r2[t].y + o[1]; o[2] = r2[t].z + o[2]; o[3] = r2[t].w + o[3]; }
                const float4 o4 = make_float4(o[0], o[1], o[2], o[3]);
                ov[i][t] = o4;
                if (p.C && row < p.M) *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + col) = o4;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (!LN) return;

    // ---- part 2: LayerNorm over the 512 columns of every row. The stand-alone kernel gives lane l the chunks l and l + 64 of
    //      a row, adds the two, then runs the 64-lane xor butterfly: chunk index = 32 pc + c4 here, so the first two levels
    //      are (pc 0 + pc 2) + (pc 1 + pc 3) per c4 and the rest a butterfly over c4. Per-chunk partials go through LDS
    //      ([row][pc][c4]); wave w reduces the rows BM / 8 * w .. of the block, two at a time (one per half-wave).
    float* P = smf;                                   // aliases the slabs: every wave is past its slab reads after the barrier
    float* ST = smf + G::P_FLOATS;                    // mean[BM], rstd[BM]
    const int lrow0 = q * 16 + rsub * 8;              // + 32 i + t: this lane's rows inside the block
    auto reduce_rows = [&](bool second) {
#pragma unroll
        for (int j = 0; j < BM / 16; ++j) {
            const int rl = wave * (BM / 8) + rsub * (BM / 16) + j;
            const float* pr = P + (size_t)rl * 128 + c4;
            float v = (pr[0] + pr[64]) + (pr[32] + pr[96]);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            const float res = second ? ln_rstd(v, R8_BN, p.ln_eps) : ln_mean(v, R8_BN);
            if (c4 == 0) ST[(second ? BM : 0) + rl] = res;
        }
    };
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int t = 0; t < 8; ++t)
            P[(size_t)(lrow0 + 32 * i + t) * 128 + pc * 32 + c4] = ln_sum4(ov[i][t]);
    __syncthreads();
    reduce_rows(false);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float mean = ST[lrow0 + 32 * i + t];
            P[(size_t)(lrow0 + 32 * i + t) * 128 + pc * 32 + c4] = ln_sqdev4(ov[i][t], mean);
        }
    __syncthreads();
    reduce_rows(true);
    __syncthreads();
    const float4 g4 = *reinterpret_cast<const float4*>(p.ln_g + col);
    const float4 b4 = *reinterpret_cast<const float4*>(p.ln_b + col);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int rl = lrow0 + 32 * i + t;
            const int row = m0 + rl;
            if (row >= p.M) continue;
            const float4 y = ln_apply4(ov[i][t], ST[rl], ST[BM + rl], g4, b4);
            if (p.Y2) {
                const float yv[4] = {y.x, y.y, y.z, y.w};
                store_split2x4_pair(p.Y2 + (size_t)row * p.ldy2 + col, p.y_plane, yv, p.yscale, lane);
            } else {
                *reinterpret_cast<float4*>(p.Yf + (size_t)row * p.ldyf + col) = y;
            }
        }
}

template <int WM, int MODE, bool LN, bool A_NT>
int launch_row8_t(const GemmRowArgs& a, hipStream_t stream) {
    typedef R8Geo<WM> G;
    static PerDeviceOnce configured;
    if (!configured.done()) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_row8_kernel<WM, MODE, LN, A_NT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_B));
        configured.mark();
    }
    hipLaunchKernelGGL((gemm_f16x2_row8_kernel<WM, MODE, LN, A_NT>), dim3((unsigned)ceil_div(a.M, G::BM)), dim3(512), G::LDS_B, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}
template <int WM, int MODE, bool LN>
int launch_row8_m(const GemmRowArgs& a, hipStream_t stream) {
    return (a.a_nt & 1) ? launch_row8_t<WM, MODE, LN, true>(a, stream) : launch_row8_t<WM, MODE, LN, false>(a, stream);
}
template <int WM>
int launch_row8_w(const GemmRowArgs& a, hipStream_t stream) {
    const bool ln = a.ln_g != nullptr;
    if (a.fs_v) return a.R2 ? launch_row8_m<WM, 6, true>(a, stream) : launch_row8_m<WM, 4, true>(a, stream);
    const int mode = (a.R1 ? 1 : 0) | (a.R2 ? 2 : 0);
    // the forms the engine issues, and the plain ones for the tests: (R2 | R1 + R2 | none) x LayerNorm, R2 without
    switch (mode * 2 + (ln ? 1 : 0)) {
        case 0: return launch_row8_m<WM, 0, false>(a, stream);
        case 1: return launch_row8_m<WM, 0, true>(a, stream);
        case 3: return launch_row8_m<WM, 1, true>(a, stream);
        case 4: return launch_row8_m<WM, 2, false>(a, stream);
        case 5: return launch_row8_m<WM, 2, true>(a, stream);
        case 7: return launch_row8_m<WM, 3, true>(a, stream);
        default: return -3;                            // not built in this block height: the caller takes the 128-row kernel
    }
}

}  // namespace

/* rows per block 96 or 128 (bm); -3: this form is not built in that height. Argument checks are launch_gemm_f16x2_row's. */
int launch_gemm_f16x2_row8(const GemmRowArgs& a, int bm, hipStream_t stream) {
    if (bm == 96) return launch_row8_w<3>(a, stream);
#if defined(PF_MEASUREMENT_KERNELS)
    if (bm == 128) return launch_row8_w<4>(a, stream);               // (the 128-row height of this wave grid: measured and off)
#endif
    return -3;
}

}  // namespace pf
