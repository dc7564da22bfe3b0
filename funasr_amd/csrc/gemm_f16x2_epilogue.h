// Epilogue of the f16x2 tile GEMMs (gemm_f16x2.hip, gemm_f16x2_w4.hip): one wave's WM x WN accumulator tiles of 32 x 32 ->
// fp32 rows / two fp16 planes / the QKV form / the fused row arg-max. Shared so that every block shape evaluates exactly the
// same expressions in the same order per output element (the shapes are bitwise equal, tested). The code was the tail of
// gemm_f16x2_kernel until round 5 and is unchanged.
//   NW   waves per workgroup (each owns a 32 x ELD float slab at `smem`)
//   wr / wc: the wave's row / column position in the workgroup's wave grid; m0 / n0: the block's first row / column
#pragma once
#include "common.h"

namespace pf {

template <int WM, int WN, int NW, int MODE, int OUT, int ABL>
__device__ __forceinline__ void gemm2_epilogue(const Gemm2Args& p, floatx16 (&acc)[WM][WN], unsigned char* smem, int m0, int n0, int nblk,
                                               int wave, int wr, int wc, int lane) {
    const int hh = lane >> 5, idx = lane & 31;
    constexpr int ELD_ = WN * 32 + 4;
    // ---- epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)):
    //      through a wave-private LDS slab so that every global access is a 16-B piece of a contiguous row segment
    if constexpr (ABL == 2) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int jj = 0; jj < WN; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][jj][r];
        if (t == 123.456f) p.C[0] = t;
        return;
    }
    if constexpr (OUT == 3) {
        // fused row arg-max over this wave's WN * 32 columns, straight from the accumulators (C/D layout: col = lane & 31,
        // row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)); ties go to the lowest column like torch.argmax
        const float osc = p.oscale_dev ? p.oscale * *p.oscale_dev : p.oscale;
        float bv[WN];
#pragma unroll
        for (int jj = 0; jj < WN; ++jj) {
            const int col = n0 + wc * (WN * 32) + jj * 32 + idx;
            bv[jj] = (p.bias && col < p.N) ? p.bias[col] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * (WM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                float best = -INFINITY;
                int besti = 0x7fffffff;
#pragma unroll
                for (int jj = 0; jj < WN; ++jj) {
                    const int col = n0 + wc * (WN * 32) + jj * 32 + idx;
                    if (col < p.N) {
                        const float v = acc[i][jj][r] * osc + bv[jj];
                        if (v > best) { best = v; besti = col; }
                    }
                }
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) {   // stays inside the 32-lane half (same row)
                    const float ov = __shfl_xor(best, o, 64);
                    const int oi = __shfl_xor(besti, o, 64);
                    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
                }
                if (idx == 0 && row < p.M) {
                    const size_t o = (size_t)row * p.amax_ld + 2 * nblk + wc;
                    p.amax_val[o] = best;
                    p.amax_idx[o] = besti;
                }
            }
        }
        return;
    }
    constexpr bool HAS_R1 = (MODE & 1) != 0, HAS_R2 = (MODE & 2) != 0;
    constexpr int ELD = ELD_;
    constexpr int LPR = WN * 8;              // lanes per slab row (float4 each)
    constexpr int RPS = 64 / LPR;            // rows per pass
    constexpr int NPASS = 32 / RPS;
    __syncthreads();
    float* slab = reinterpret_cast<float*>(smem) + wave * (32 * ELD);
    const int c4 = lane % LPR, rsub = lane / LPR;
    const int col = n0 + wc * (WN * 32) + c4 * 4;
    const bool colok = col + 3 < p.N;
    // plane output: 16-B stores by lane pairs when a pair's 8 columns are always valid together and 16-B aligned
    const bool wide_st = OUT == 1 && p.N % 8 == 0 && p.ldc2 % 8 == 0 && p.c_plane % 8 == 0 && ((uintptr_t)p.C2 & 15) == 0;
    const float oscale = p.oscale_dev ? p.oscale * *p.oscale_dev : p.oscale;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && colok) bias4 = *reinterpret_cast<const float4*>(p.bias + col);
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int jj = 0; jj < WN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[((r & 3) + 8 * (r >> 2) + 4 * hh) * ELD + jj * 32 + idx] = acc[i][jj][r];
        const int row0 = m0 + wr * (WM * 32) + i * 32 + rsub;
        if (!colok) continue;
        // QKV form: this block's 256 columns lie inside one of q | k | v (qkv_D % 256 == 0)
        const int seg = OUT == 2 ? n0 / p.qkv_D + (p.kv_form ? 1 : 0) : 0;      // 0 q, 1 k, 2 v
        const int scol = OUT == 2 ? col - (n0 / p.qkv_D) * p.qkv_D : col;
        float k_mul = p.k_mul, v_mul = p.v_mul;
        if constexpr (OUT == 2) {
            if (p.kv_mul_dev) { k_mul *= p.kv_mul_dev[0]; v_mul *= p.kv_mul_dev[1]; }
        }
#pragma unroll
        for (int h2 = 0; h2 < NPASS / 8; ++h2) {
            float4 v[8], r1[8], r2[8];
#pragma unroll
            for (int it = 0; it < 8; ++it)
                v[it] = *reinterpret_cast<const float4*>(slab + ((h2 * 8 + it) * RPS + rsub) * ELD + c4 * 4);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = row0 + (h2 * 8 + it) * RPS;
                const int rr = row < p.M ? row : p.M - 1;
                if constexpr (HAS_R1) r1[it] = *reinterpret_cast<const float4*>(p.R1 + (size_t)rr * p.ldr1 + col);
                if constexpr (HAS_R2) r2[it] = *reinterpret_cast<const float4*>(p.R2 + (size_t)rr * p.ldr2 + col);
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = row0 + (h2 * 8 + it) * RPS;
                float o[4] = {v[it].x * oscale + bias4.x, v[it].y * oscale + bias4.y, v[it].z * oscale + bias4.z,
                              v[it].w * oscale + bias4.w};
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
                }
                if constexpr (HAS_R1) { o[0] = o[0] + r1[it].x; o[1] = o[1] + r1[it].y; o[2] = o[2] + r1[it].z; o[3] = o[3] + r1[it].w; }
                if constexpr (HAS_R2) { o[0] = r2[it].x + o[0]; o[1] = r2[it].y + o[1]; o[2] = r2[it].z + o[2]; o[3] = r2[it].w + o[3]; }
                if (row >= p.M) continue;
                if constexpr (ABL == 1) { if (o[0] == 123.456f) p.C[0] = o[1] + o[2] + o[3]; continue; }
                if constexpr (OUT == 2) {           // (qkv_D % 256 == 0: a lane pair's 8 columns lie in one segment)
                    if (seg == 0) store_split2x4_pair(p.Qp + (size_t)row * p.qkv_D + scol, p.qk_plane, o, p.q_mul, lane);
                    else if (seg == 1) store_split2x4_pair(p.Kp + (size_t)row * p.qkv_D + scol, p.qk_plane, o, k_mul, lane);
                    else if (p.C) *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + scol) = make_float4(o[0], o[1], o[2], o[3]);
                } else if constexpr (OUT == 1) {
                    if (wide_st) store_split2x4_pair(p.C2 + (size_t)row * p.ldc2 + col, p.c_plane, o, p.cscale, lane);
                    else store_split2x4(p.C2 + (size_t)row * p.ldc2 + col, p.c_plane, o, p.cscale);
                } else if constexpr (ABL == 4) {
                    // same store instructions, but into a 64-KB window that stays in L2: issue cost without the HBM drain
                    *reinterpret_cast<float4*>(p.C + (size_t)(row & 63) * 256 + (col & 255)) = make_float4(o[0], o[1], o[2], o[3]);
                } else {
                    *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + col) = make_float4(o[0], o[1], o[2], o[3]);
                }
            }
        }
        if constexpr (OUT == 2) {
            // V^T planes: the slab read column-wise. A 16-B piece = one d (column), 8 rows {0..3, 8..11} + 4 half of the
            // 16-row group G -- the rows whose scores one attention lane holds per 16-key step (attention_f16x2.hip).
            // Raw accumulators are still in the slab: bias and scales are applied again here.
            if (seg == 2) {
                const int vc0 = n0 - (p.kv_form ? 1 : 2) * p.qkv_D + wc * (WN * 32);
#pragma unroll
                for (int ps = 0; ps < WN * 32 * 4 / 64; ++ps) {
                    const int piece = ps * 64 + lane;
                    const int nl = piece >> 2, G = (piece >> 1) & 1, half = piece & 1;
                    const int rb = 16 * G + 4 * half;
                    const float bv = p.bias ? p.bias[n0 + wc * (WN * 32) + nl] : 0.f;
                    float t[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        t[j] = (slab[(rb + (j & 3) + 8 * (j >> 2)) * ELD + nl] * oscale + bv) * v_mul;
                    const int mb = m0 + wr * (WM * 32) + i * 32 + 16 * G;
                    if (mb < p.M) {
                        uint4 h, l;
                        split2_pk(t[0], t[1], h.x, l.x);
                        split2_pk(t[2], t[3], h.y, l.y);
                        split2_pk(t[4], t[5], h.z, l.z);
                        split2_pk(t[6], t[7], h.w, l.w);
                        unsigned short* vp = p.VT + (size_t)(vc0 + nl) * p.ldvt + mb + 8 * half;
                        *reinterpret_cast<uint4*>(vp) = h;
                        *reinterpret_cast<uint4*>(vp + p.vt_plane) = l;
                    }
                }
            }
        }
    }
}

}  // namespace pf
