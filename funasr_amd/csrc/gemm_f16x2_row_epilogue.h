// Epilogue of the full-row f16x2 GEMMs (gemm_f16x2_row.hip: eight waves 2 x 4; gemm_f16x2_w4.hip: four waves 1 x 4, called once
// per 64-row half): a wave's 64 x 128 accumulators -> bias / ReLU / FSMN memory block / residual adds -> the fp32 residual
// stream -> LayerNorm over the 512-column row -> the two fp16 planes the next GEMM reads. Shared so that both wave grids
// evaluate exactly the same expressions in the same order (bitwise equal, tested). The code was the tail of
// gemm_f16x2_row_kernel until round 5 and is unchanged except for what the parameters below replace:
//   NT      threads in the workgroup (512 / 256): the FSMN taps are staged by NT threads, the slabs belong to NT / 64 waves
//   AM, I0  the accumulator array has AM row tiles; this call finishes tiles I0, I0 + 1 (rows 64 wr .. 64 wr + 63 of the block)
//   wave    slab owner (0 .. NT / 64 - 1); wr / wc: the row half / column quarter of the block this call covers
//   stage_taps: copy the FSMN taps into LDS (first call of a workgroup only)
#pragma once
#include "common.h"

namespace pf {

constexpr int RW_BM = 128, RW_BN = 512, RW_KS = 32, RW_ROWB = 64;
constexpr int RW_A_PLANE_B = RW_BM * RW_ROWB;                       // 8 KB
constexpr int RW_B_PLANE_B = RW_BN * RW_ROWB;                       // 32 KB
constexpr int RW_STAGE_B = 2 * (RW_A_PLANE_B + RW_B_PLANE_B);       // 80 KB
constexpr int RW_PPW = RW_STAGE_B / 1024 / 8;                       // 10 pieces per wave and stage
constexpr int RW_ELD = 132;                                         // slab row (floats): 128 columns + 4
constexpr int RW_SLAB_B = 8 * 32 * RW_ELD * 4;                      // 135168
constexpr int RW_LDS_B = 2 * RW_STAGE_B;                            // 163840 = the CU's whole LDS
constexpr int RW_P_FLOATS = RW_BM * 4 * 32;                         // statistics exchange [row][wave column][lane]: 64 KB
constexpr int RW_FS_KS = 11, RW_FS_LP = 5;                          // FSMN taps / left padding (the offline encoder's kernel 11)
constexpr int RW_FSW_OFF_B = RW_SLAB_B;                             // FSMN taps [11][512] floats behind the slabs: 22 KB
static_assert(RW_SLAB_B <= RW_LDS_B && (RW_P_FLOATS + 2 * RW_BM) * 4 <= RW_LDS_B, "epilogue LDS");
static_assert(RW_FSW_OFF_B + RW_FS_KS * RW_BN * 4 <= RW_LDS_B, "FSMN taps do not fit behind the slabs");

template <int MODE, bool LN, int NT, int AM, int I0>
__device__ __forceinline__ void gemm2_row_epilogue(const GemmRowArgs& p, floatx16 (&accf)[AM][4], unsigned char* smem, int m0, int tid,
                                                   int wave, int wr, int wc, int lane, bool stage_taps) {
    constexpr int WM = 2, WN = 4;
    const int hh = lane >> 5, idx = lane & 31;
    floatx16 (&acc_)[AM][4] = accf;
    // measurement switches of the FSMN form (a_nt bits 4..6, measurement library only; profiles/r06ab): 16 no v-row loads, 32 no
    // residual loads, 64 no global stores -- the results are then wrong by construction
#if defined(PF_MEASUREMENT_KERNELS)
    const int abl = p.a_nt;
#else
    constexpr int abl = 0;
#endif
#define acc(i, jj) acc_[I0 + (i)][jj]
    // ---- epilogue, part 1: accumulators (C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) +
    //      4 (lane >> 5)) -> wave-private slab -> float4 pieces of rows: half-wave h takes rows 16 h .. 16 h + 15 of the
    //      32-row tile, lane c4 its columns 4 c4 .. 4 c4 + 3 of the wave's 128; the finished values stay in registers
    constexpr bool HAS_R1 = (MODE & 1) != 0, HAS_R2 = (MODE & 2) != 0, FSMN = (MODE & 4) != 0;
    static_assert(!(HAS_R1 && FSMN), "the FSMN form computes the first addend");
    __syncthreads();
    float* smf = reinterpret_cast<float*>(smem);
    float* slab = smf + wave * (32 * RW_ELD);
    const float* fsw = reinterpret_cast<const float*>(smem + RW_FSW_OFF_B);      // [tap][512]
    if constexpr (FSMN) {
        // taps [512][11] -> LDS [11][512]: thread = channel (consecutive threads, consecutive LDS words)
        if (stage_taps) {
            float* dst = reinterpret_cast<float*>(smem + RW_FSW_OFF_B);
#pragma unroll
            for (int c = tid; c < RW_BN; c += NT)
#pragma unroll
                for (int j = 0; j < RW_FS_KS; ++j) dst[j * RW_BN + c] = p.fs_w[(size_t)c * RW_FS_KS + j];
        }
        __syncthreads();
    }
    const int c4 = idx, rsub = hh;
    const int col = wc * 128 + c4 * 4;
    const float oscale = p.oscale_dev ? p.oscale * *p.oscale_dev : p.oscale;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + col);
    float4 ov[WM][16];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int jj = 0; jj < WN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[((r & 3) + 8 * (r >> 2) + 4 * hh) * RW_ELD + jj * 32 + idx] = acc(i, jj)[r];
        const int row0 = m0 + wr * 64 + i * 32 + rsub * 16;
        if constexpr (FSMN) {
            // this half-wave's 16 rows are one 16-row group of one sequence: valid v rows [lo, hi) (sequence start .. start +
            // len), everything else -- other sequences, padding rows, rows outside the batch -- counts as zero input, and
            // output rows >= hi get no memory. Four output rows at a time over a sliding window of 14 v rows (26 per tile).
            const int grp = row0 >> 4;
            const bool gok = row0 < p.M;
            const int lo = gok ? p.fs_lo[grp] : 0, hi = gok ? p.fs_hi[grp] : 0;
            float4 win[16 + RW_FS_KS - 1];
            auto load_row = [&](int k) {
                const int vr = row0 - RW_FS_LP + k;
                const bool ok = vr >= lo && vr < hi;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!(abl & 16)) t = *reinterpret_cast<const float4*>(p.fs_v + (size_t)(ok ? vr : (lo < hi ? lo : 0)) * p.ldfv + col);
                win[k] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            };
#pragma unroll
            for (int k = 0; k < RW_FS_KS - 1; ++k) load_row(k);
#pragma unroll
            for (int h4 = 0; h4 < 4; ++h4) {
                float4 r2[4], fa[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    load_row(RW_FS_KS - 1 + h4 * 4 + t);
                    fa[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (HAS_R2) {
                        const int row = row0 + h4 * 4 + t;
                        r2[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (!(abl & 32)) r2[t] = *reinterpret_cast<const float4*>(p.R2 + (size_t)(row < p.M ? row : p.M - 1) * p.ldr2 + col);
                    }
                }
#pragma unroll
                for (int j = 0; j < RW_FS_KS; ++j) {
                    const float4 wj = *reinterpret_cast<const float4*>(fsw + j * RW_BN + col);
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
                    const float4 c = win[h4 * 4 + t + RW_FS_LP];                  // the (masked) input row itself
                    float4 mem = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < hi) mem = make_float4(fa[t].x + c.x, fa[t].y + c.y, fa[t].z + c.z, fa[t].w + c.w);
                    const float4 vt = *reinterpret_cast<const float4*>(slab + (rsub * 16 + h4 * 4 + t) * RW_ELD + c4 * 4);
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
                    if (p.C && row < p.M && !(abl & 64)) *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + col) = o4;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            continue;
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            float4 v[8], r1[8], r2[8];
#pragma unroll
            for (int t = 0; t < 8; ++t)
                v[t] = *reinterpret_cast<const float4*>(slab + (rsub * 16 + h2 * 8 + t) * RW_ELD + c4 * 4);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int row = row0 + h2 * 8 + t;
                const int rr = row < p.M ? row : p.M - 1;
                if constexpr (HAS_R1) r1[t] = *reinterpret_cast<const float4*>(p.R1 + (size_t)rr * p.ldr1 + col);
                if constexpr (HAS_R2) r2[t] = *reinterpret_cast<const float4*>(p.R2 + (size_t)rr * p.ldr2 + col);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int row = row0 + h2 * 8 + t;
                float o[4] = {v[t].x * oscale + bias4.x, v[t].y * oscale + bias4.y, v[t].z * oscale + bias4.z,
                              v[t].w * oscale + bias4.w};
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
                }
                if constexpr (HAS_R1) { o[0] = o[0] + r1[t].x; o[1] = o[1] + r1[t].y; o[2] = o[2] + r1[t].z; o[3] = o[3] + r1[t].w; }
                if constexpr (HAS_R2) { o[0] = r2[t].x + o[0]; o[1] = r2[t].y + o[1]; o[2] = r2[t].z + o[2]; o[3] = r2[t].w + o[3]; }
                const float4 o4 = make_float4(o[0], o[1], o[2], o[3]);
                ov[i][h2 * 8 + t] = o4;
                if (p.C && row < p.M) *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + col) = o4;
            }
            // keep the eight-row groups apart: without the fence the scheduler hoists every slab read of the tile above the
            // arithmetic and spills (209 VGPRs in the form without addends)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (!LN) return;

    // ---- part 2: LayerNorm over the 512 columns of every row. The stand-alone kernel gives lane l the chunks l and l + 64
    //      of a row, adds the two, then runs the 64-lane xor butterfly: chunk index = 32 wc + c4 here, so the first two
    //      levels are (wc 0 + wc 2) + (wc 1 + wc 3) per c4 and the rest a butterfly over c4. Per-chunk partials go through
    //      LDS ([row][wc][c4]); wave (wr, wc) reduces rows 16 wc .. 16 wc + 15 of its row half.
    float* P = smf;                                   // aliases the slabs: every wave is past its slab reads
    float* ST = smf + RW_P_FLOATS;                    // mean[128], rstd[128]
    const int lrow0 = wr * 64 + rsub * 16;            // + 32 i + it: this lane's rows inside the tile
    auto reduce_rows = [&](bool second) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int rl = wr * 64 + wc * 16 + rsub * 8 + j;
            const float* pr = P + (size_t)rl * 128 + c4;
            float v = (pr[0] + pr[64]) + (pr[32] + pr[96]);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            const float res = second ? ln_rstd(v, RW_BN, p.ln_eps) : ln_mean(v, RW_BN);
            if (c4 == 0) ST[(second ? RW_BM : 0) + rl] = res;
        }
    };
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int it = 0; it < 16; ++it)
            P[(size_t)(lrow0 + 32 * i + it) * 128 + wc * 32 + c4] = ln_sum4(ov[i][it]);
    __syncthreads();
    reduce_rows(false);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const float mean = ST[lrow0 + 32 * i + it];
            P[(size_t)(lrow0 + 32 * i + it) * 128 + wc * 32 + c4] = ln_sqdev4(ov[i][it], mean);
        }
    __syncthreads();
    reduce_rows(true);
    __syncthreads();
    const float4 g4 = *reinterpret_cast<const float4*>(p.ln_g + col);
    const float4 b4 = *reinterpret_cast<const float4*>(p.ln_b + col);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int rl = lrow0 + 32 * i + it;
            const int row = m0 + rl;
            if (row >= p.M || (abl & 64)) continue;
            const float4 y = ln_apply4(ov[i][it], ST[rl], ST[RW_BM + rl], g4, b4);
            if (p.Y2) {
                const float yv[4] = {y.x, y.y, y.z, y.w};
                store_split2x4_pair(p.Y2 + (size_t)row * p.ldy2 + col, p.y_plane, yv, p.yscale, lane);
            } else {
                *reinterpret_cast<float4*>(p.Yf + (size_t)row * p.ldyf + col) = y;
            }
        }
#undef acc
}

}  // namespace pf
