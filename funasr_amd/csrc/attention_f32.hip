// Flash-style scaled-dot-product attention in exact fp32 on the CDNA4 matrix cores (d_k = 128).
//
// Replaces the materialised  softmax((q * d_k^-0.5) k^T masked_fill -inf) masked_fill 0  @ v  of
//   funasr/models/sanm/attention.py:270-306,322-327   (encoder self-attention, SAN-M)
//   funasr/models/sanm/attention.py:760-813           (decoder cross-attention over the encoder memory)
// The T x T score matrix (256 MB per layer at B=64, T=500) never reaches HBM.
//
// Mapping. One workgroup = 4 waves = 128 queries of one (sequence, head); each wave owns 32 queries and the
// whole d_k = 128. Both MFMA products are issued "swapped" so that a lane always holds ONE query
// (q = lane & 31) and the keys / output channels run over its accumulator registers:
//     S^T[key][q] = sum_d K[key][d] * Q[q][d]      A = K tile (LDS), B = Q (64 registers per lane)
//     O^T[d][q]   = sum_key V[key][d] * P[q][key]  A = V tile (LDS), B = P (= the S^T accumulator registers)
// The 32x32x2 MFMA's k index is only a pairing between A and B, so the half-wave h = lane >> 5 takes
// d in [64h, 64h+64) for S^T and, for O^T, exactly the key its own S^T register r already holds
// (key = (r&3) + 8(r>>2) + 4h): P feeds the second product straight from registers, the online-softmax row
// statistics are per lane (one xor-32 shuffle joins the two halves) and no LDS transpose is needed.
// K rows are padded to 132 floats so the ds_read_b128 operand fetch is bank-conflict free; V rows are
// read 32 consecutive floats per half-wave.
#include "common.h"

namespace pf {

namespace {

constexpr int DK = 128;        // head dim
constexpr int KT = 32;         // keys per tile
constexpr int KLD = DK + 4;    // padded K row (floats)

__global__ __launch_bounds__(256, 2) void attention_f32_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[KT * KLD + KT * DK];
    float* Ks = smem;
    float* Vs = smem + KT * KLD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, idx = lane & 31;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + idx;
    const int qc = q < p.Tq ? q : p.Tq - 1;
    // keys: [0, n1) from source 1 (K, V; rows b*Tk + t), then n2 keys from the optional source 2 (K2, V2; rows
    // b*T2 + t) -- the streaming step attends over [cached K/V ring | this chunk's K/V] without staging them
    // into one buffer (funasr/models/sanm/attention.py:343-347 `torch.cat((cache["k"], k_h))`)
    const int n1 = p.K2 ? p.n1_dev[b * p.n1_stride] : p.klens[b];
    const int klen = p.K2 ? n1 + p.n2 : n1;

    // Q fragment: this lane's query row, d in [64h, 64h + 64), pre-scaled like the reference (q * d_k^-0.5)
    float qreg[64];
    {
        const float* qp = p.Q + ((size_t)b * p.Tq + qc) * p.ldq + head * DK + hh * 64;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(qp + 4 * i);
            qreg[4 * i + 0] = t.x * p.scale;
            qreg[4 * i + 1] = t.y * p.scale;
            qreg[4 * i + 2] = t.z * p.scale;
            qreg[4 * i + 3] = t.w * p.scale;
        }
    }

    floatx16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int lc4 = tid & 31, lr = tid >> 5;   // tile loader: 32 float4 per row, 8 rows per pass
    const float* kbase = p.K + (size_t)b * p.Tk * p.ldk + head * DK + lc4 * 4;
    const float* vbase = p.V + (size_t)b * p.Tk * p.ldv + head * DK + lc4 * 4;
    const float* kbase2 = p.K2 ? p.K2 + (size_t)b * p.T2 * p.ldk2 + head * DK + lc4 * 4 : nullptr;
    const float* vbase2 = p.K2 ? p.V2 + (size_t)b * p.T2 * p.ldv2 + head * DK + lc4 * 4 : nullptr;

    const int ntiles = (klen + KT - 1) / KT;
    for (int kt = 0; kt < ntiles; ++kt) {
        const int k0 = kt * KT;
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lr + 8 * i;
            int kr = k0 + r;
            kr = kr < klen ? kr : klen - 1;          // rows past the last valid key only feed masked (-inf) scores
            const float* ksrc;
            const float* vsrc;
            if (kr < n1) {
                ksrc = kbase + (size_t)kr * p.ldk;
                vsrc = vbase + (size_t)kr * p.ldv;
            } else {
                ksrc = kbase2 + (size_t)(kr - n1) * p.ldk2;
                vsrc = vbase2 + (size_t)(kr - n1) * p.ldv2;
            }
            const float4 kv = *reinterpret_cast<const float4*>(ksrc);
            const float4 vv = *reinterpret_cast<const float4*>(vsrc);
            *reinterpret_cast<float4*>(&Ks[r * KLD + lc4 * 4]) = kv;
            *reinterpret_cast<float4*>(&Vs[r * DK + lc4 * 4]) = vv;
        }
        __syncthreads();

        // ---- S^T tile (32 keys x 32 queries)
        floatx16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kp = &Ks[idx * KLD + hh * 64];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 kf = *reinterpret_cast<const float4*>(kp + 4 * i);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qreg[4 * i + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qreg[4 * i + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qreg[4 * i + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qreg[4 * i + 3], s, 0, 0, 0);
        }

        // ---- online softmax for query (lane & 31); this lane holds keys k0 + (r&3) + 8(r>>2) + 4h
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= klen) s[r] = -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);          // finite: every tile has >= 1 valid key
        const float alpha = expf(m_run - m_new);       // exp(-inf) = 0 on the first tile
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = expf(s[r] - m_new);
            psum += s[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

        // ---- O^T += V^T P^T
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int krow = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float* vp = &Vs[krow * DK + idx];
#pragma unroll
            for (int d = 0; d < 4; ++d)
                o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32 * d], s[r], o[d], 0, 0, 0);
        }
    }

    if (q < p.Tq) {
        const float inv = 1.0f / l_run;
        const size_t orow = ((size_t)b * p.Tq + q) * p.ldo + head * DK;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float t[4] = {o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv};
                const size_t off = orow + d * 32 + 8 * g + 4 * hh;
                if (p.O3) store_split3x4(p.O3 + off, p.o_plane, t);      // the out-projection's operand planes (bf16x3 mode)
                else *reinterpret_cast<float4*>(p.O + off) = make_float4(t[0], t[1], t[2], t[3]);
            }
    }
}

// ---- offline variant: K/V tiles go HBM -> LDS with global_load_lds_dwordx4 (no staging registers), double buffered:
// the DMA of tile t+1 is in flight under the 128 MFMAs of tile t, one barrier per tile. K rows are 512 B; the DMA
// writes lane-linear, so the 16-B chunk permutation that makes the ds_read_b128 operand fetch conflict-free
// (chunk c of key r at c ^ (r & 15)) is applied to the per-lane SOURCE address and again on the read.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
constexpr int TILE_F = KT * DK;                 // floats per K (or V) tile: 16 KB
constexpr int STAGE_F = 2 * TILE_F;             // K tile + V tile

__global__ __launch_bounds__(256, 2) void attention_f32_dma_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE_F];     // 64 KB, the only LDS object

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, idx = lane & 31;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + idx;
    const int qc = q < p.Tq ? q : p.Tq - 1;
    const int klen = p.klens[b];

    float qreg[64];
    {
        const float* qp = p.Q + ((size_t)b * p.Tq + qc) * p.ldq + head * DK + hh * 64;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(qp + 4 * i);
            qreg[4 * i + 0] = t.x * p.scale;
            qreg[4 * i + 1] = t.y * p.scale;
            qreg[4 * i + 2] = t.z * p.scale;
            qreg[4 * i + 3] = t.w * p.scale;
        }
    }

    floatx16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // DMA pieces: a piece = 2 keys x 512 B; wave w stages pieces 4w .. 4w+3 of both tiles
    const float* kbase = p.K + (size_t)b * p.Tk * p.ldk + head * DK;
    const float* vbase = p.V + (size_t)b * p.Tk * p.ldv + head * DK;
    const int cp = lane & 31;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (unsigned)wave * 4 * 2 * DK * 4);
    auto stage = [&](int buf, int k0) {
        // issued from inline asm (common.h: glds16): the compiler must not drain the prefetch before the MFMAs
        const unsigned base = lds_base + (unsigned)buf * (STAGE_F * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (wave * 4 + i) * 2 + (lane >> 5);          // key row inside the tile
            int kr = k0 + r;
            kr = kr < klen ? kr : klen - 1;                           // rows past the last valid key feed masked scores only
            const float* ks = kbase + (size_t)kr * p.ldk + ((cp ^ (r & 15)) * 4);
            const float* vs = vbase + (size_t)kr * p.ldv + cp * 4;
            glds16(ks, base + i * 2 * DK * 4);
            glds16(vs, base + TILE_F * 4 + i * 2 * DK * 4);
        }
    };

    const int ntiles = (klen + KT - 1) / KT;
    stage(0, 0);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int k0 = kt * KT;
        glds_wait_all();                      // this wave's pieces of tile kt have landed ...
        __syncthreads();                      // ... and everybody else's; the other buffer is free again
        if (kt + 1 < ntiles) stage((kt + 1) & 1, k0 + KT);
        const float* Ks = smem + (kt & 1) * STAGE_F;
        const float* Vs = Ks + TILE_F;

        // ---- S^T tile (32 keys x 32 queries); this lane's K row idx, d chunks [16h, 16h+16) permuted by idx & 15
        floatx16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const float* kp = Ks + idx * DK;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float4 kf = *reinterpret_cast<const float4*>(kp + (((hh * 16 + i) ^ (idx & 15)) * 4));
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qreg[4 * i + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qreg[4 * i + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qreg[4 * i + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qreg[4 * i + 3], s, 0, 0, 0);
        }

        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (key >= klen) s[r] = -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = expf(s[r] - m_new);
            psum += s[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int krow = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float* vp = Vs + krow * DK + idx;
#pragma unroll
            for (int d = 0; d < 4; ++d)
                o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32 * d], s[r], o[d], 0, 0, 0);
        }
    }

    if (q < p.Tq) {
        const float inv = 1.0f / l_run;
        const size_t orow = ((size_t)b * p.Tq + q) * p.ldo + head * DK;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float t[4] = {o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv};
                const size_t off = orow + d * 32 + 8 * g + 4 * hh;
                if (p.O3) store_split3x4(p.O3 + off, p.o_plane, t);      // the out-projection's operand planes (bf16x3 mode)
                else *reinterpret_cast<float4*>(p.O + off) = make_float4(t[0], t[1], t[2], t[3]);
            }
    }
}

// ---- few-query variant: the streaming step attends 15 window rows (or <= 24 token rows) of one stream over a few dozen
// keys, one launch in a chain of dependent few-microsecond launches. The 128-query kernels above spend ~4 us of dependent
// v_mfma_f32_32x32x2_f32 per 32-key tile on one SIMD whatever the number of queries; here one workgroup = 16 queries of one
// (stream, head), its four waves split the KEYS (16 per wave and 64-key round), every operand goes global -> registers in
// one round of loads (no LDS staging, no barrier before the end), v_mfma_f32_16x16x4_f32:
//   S^T[key][q] = K Q^T : A = K (row = key, k = d), B = Q^T -- a lane's 16-B loads along d feed four MFMAs (the MFMA k
//                 index is only a pairing); the lane then holds S^T[4 kq + r][q = lane & 15];
//   O^T[d][q]  += V^T P : k slot kq of step s pairs key 4 kq + s on both operands, so B is the lane's own register s.
// Each wave keeps online-softmax statistics over its keys; the four partial results meet in LDS and are combined in a
// fixed order (flash-decoding merge). Exact fp32 products and accumulation, libm expf.
// AttnArgs.fs_*: eight consecutive rows of one stream's FSMN memory block, thread = 4 channels -- fsmn_kernel<11, 5> of
// rowwise.hip term by term (same taps order, same fma chain: the bits of the stand-alone launch), every row valid
__device__ __forceinline__ void fewq_fsmn_rows(const AttnArgs& p, int b, int t0, int c4) {
    constexpr int KS = 11, LP = 5, TT = 8;
    float4 w[KS];
    {
        const float* wp = p.fs_w + (size_t)c4 * 4 * KS;
#pragma unroll
        for (int j = 0; j < KS; ++j) { w[j].x = wp[j]; w[j].y = wp[KS + j]; w[j].z = wp[2 * KS + j]; w[j].w = wp[3 * KS + j]; }
    }
    const size_t base = (size_t)b * p.fs_T;
    float4 win[KS + TT - 1];
#pragma unroll
    for (int i = 0; i < KS + TT - 1; ++i) {
        const int tt = t0 - LP + i;
        win[i] = (tt >= 0 && tt < p.fs_T) ? *reinterpret_cast<const float4*>(p.fs_in + (base + tt) * p.fs_ldin + c4 * 4)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        if (t0 + i < p.fs_T) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                acc.x = fmaf(w[j].x, win[i + j].x, acc.x);
                acc.y = fmaf(w[j].y, win[i + j].y, acc.y);
                acc.z = fmaf(w[j].z, win[i + j].z, acc.z);
                acc.w = fmaf(w[j].w, win[i + j].w, acc.w);
            }
            const float4 c = win[i + LP];
            *reinterpret_cast<float4*>(p.fs_out + (base + t0 + i) * p.fs_ldo + c4 * 4) =
                make_float4(acc.x + c.x, acc.y + c.y, acc.z + c.z, acc.w + c.w);
        }
    }
}

__global__ __launch_bounds__(256) void attention_f32_fewq_kernel(AttnArgs p) {
    if (blockIdx.y >= p.H) {                                  // the FSMN workgroups (wave-uniform: no barrier is skipped by part of a block)
        const int t0 = ((blockIdx.y - p.H) * gridDim.x + blockIdx.x) * 8;
        if (t0 < p.fs_T && threadIdx.x * 4 < p.H * DK) fewq_fsmn_rows(p, blockIdx.z, t0, threadIdx.x);
        return;
    }
    __shared__ float red_o[4][16][DK + 1];
    __shared__ float red_m[4][16], red_l[4][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int b = blockIdx.z, head = blockIdx.y, q0 = blockIdx.x * 16;
    const int n1 = p.K2 ? p.n1_dev[b * p.n1_stride] : p.klens[b];
    const int klen = p.K2 ? n1 + p.n2 : n1;

    // the ring append (below) copies rows of (K2, V2) that nothing in this kernel produces: fetch them now, store them at the end
    const bool do_app = p.app_rows > 0 && gridDim.x == 1 && p.K2;
    const int app_cap = (p.app_mod > 0 && n1 > 0) ? p.app_mod : p.Tk;  // (AttnArgs.app_mod: an encoder ring after its first append)
    const int app_skip = p.app_rows > app_cap ? p.app_rows - app_cap : 0;   // only the newest `cap` rows can survive
    const int app_n = (p.app_rows - app_skip) * 64;                    // 32 float4 of K and 32 of V per row
    float4 appv[4];
    if (do_app) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = tid + 256 * e;
            if (i < app_n) {
                const int r = i >> 6, c = i & 63, c4 = (c & 31) * 4;
                const size_t src = (size_t)b * p.T2 + p.app_r0 + app_skip + r;
                appv[e] = c < 32 ? *reinterpret_cast<const float4*>(p.K2 + src * p.ldk2 + head * DK + c4)
                                 : *reinterpret_cast<const float4*>(p.V2 + src * p.ldv2 + head * DK + c4);
            }
        }
    }

    float4 qf[8];
    {
        int qrow = q0 + i16;
        qrow = qrow < p.Tq ? qrow : p.Tq - 1;
        const float* qp = p.Q + ((size_t)b * p.Tq + qrow) * p.ldq + head * DK + kq * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 t = *reinterpret_cast<const float4*>(qp + 16 * j);
            qf[j] = make_float4(t.x * p.scale, t.y * p.scale, t.z * p.scale, t.w * p.scale);   // q * d_k^-0.5 like the reference
        }
    }
    // row pointers of key `key` (clamped to the last valid one: those scores are masked) in the one or two sources
    auto krow = [&](int key) -> const float* {
        key = key < klen ? key : klen - 1;
        return key < n1 ? p.K + ((size_t)b * p.Tk + key) * p.ldk : p.K2 + ((size_t)b * p.T2 + (key - n1)) * p.ldk2;
    };
    auto vrow = [&](int key) -> const float* {
        key = key < klen ? key : klen - 1;
        return key < n1 ? p.V + ((size_t)b * p.Tk + key) * p.ldv : p.V2 + ((size_t)b * p.T2 + (key - n1)) * p.ldv2;
    };

    floatx4 o[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) o[dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    for (int k0 = 0; k0 < klen; k0 += 64) {
        const int kw0 = k0 + 16 * wave;
        if (kw0 >= klen) break;                           // (wave-uniform) no key of this round for this wave
        float4 kf[8];
        {
            const float* kp = krow(kw0 + i16) + head * DK + kq * 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) kf[j] = *reinterpret_cast<const float4*>(kp + 16 * j);
        }
        float vf[4][8];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* vp = vrow(kw0 + 4 * kq + s) + head * DK + i16;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) vf[s][dt] = vp[16 * dt];
        }
        floatx4 sc = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j].x, qf[j].x, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j].y, qf[j].y, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j].z, qf[j].z, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j].w, qf[j].w, sc, 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (kw0 + 4 * kq + r >= klen) sc[r] = -INFINITY;
            mx = fmaxf(mx, sc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));           // finite: key kw0 is valid
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);          // exp(-inf) = 0 on the first round
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[r] = expf(sc[r] - m_new);
            psum += sc[r];
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[s][dt], sc[s], o[dt], 0, 0, 0);
    }

    // lane holds O^T[d = 16 dt + 4 kq + r][q = i16] of its wave's keys
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red_o[wave][i16][16 * dt + 4 * kq + r] = o[dt][r];
    if (kq == 0) { red_m[wave][i16] = m_run; red_l[wave][i16] = l_run; }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int t = tid + 256 * e;                      // 16 queries x 32 float4
        const int qq = t >> 5, d = (t & 31) * 4;
        if (q0 + qq >= p.Tq) continue;
        float m = fmaxf(fmaxf(red_m[0][qq], red_m[1][qq]), fmaxf(red_m[2][qq], red_m[3][qq]));
        float l = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {                     // fixed order; a wave without keys has m = -inf, weight 0
            const float wgt = expf(red_m[w][qq] - m);
            l += red_l[w][qq] * wgt;
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] += red_o[w][qq][d + u] * wgt;
        }
        const float o[4] = {acc[0] / l, acc[1] / l, acc[2] / l, acc[3] / l};
        const size_t row = (size_t)b * p.Tq + q0 + qq;
        if (p.O2) store_split2x4(p.O2 + row * p.ldo2 + head * DK + d, p.o2_plane, o, p.o2_scale);
        else *reinterpret_cast<float4*>(p.O + row * p.ldo + head * DK + d) = make_float4(o[0], o[1], o[2], o[3]);
    }
    // ring append (attention.py:343-361: the cache keeps the last rows): every wave of this -- the only -- workgroup of
    // (stream, head) passed the barrier above, i.e. has its ring rows in registers; other heads own other columns
    if (do_app && (!p.app_gate || p.app_gate[b] >= 1)) {
        const int cap = app_cap;
        const int wp = p.app_wp[b * p.app_wp_stride];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = tid + 256 * e;
            if (i < app_n) {
                const int r = i >> 6, c = i & 63, c4 = (c & 31) * 4;
                const size_t dst = (size_t)b * p.Tk + (wp + app_skip + r) % cap;
                float* base = c < 32 ? const_cast<float*>(p.K) + dst * p.ldk : const_cast<float*>(p.V) + dst * p.ldv;
                *reinterpret_cast<float4*>(base + head * DK + c4) = appv[e];
            }
        }
    }
}

}  // namespace

int launch_attention_f32(const AttnArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.B > 0 && a.H > 0 && a.Tq > 0 && a.Tk > 0, "attention: empty problem");
    PF_REQUIRE(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.ldo % 4 == 0, "attention: strides % 4");
    PF_REQUIRE(a.O || a.O3 || a.O2, "attention: null output");
    if (a.O3) PF_REQUIRE(a.o_plane % 4 == 0 && ((uintptr_t)a.O3 & 7) == 0, "attention: plane output alignment");
    if (attention_takes_fewq(a)) {
        const int gx = ceil_div(a.Tq, 16);
        int fs_y = 0;
        if (a.fs_in) {
            PF_REQUIRE(a.fs_w && a.fs_out && a.fs_T > 0 && a.fs_ldin % 4 == 0 && a.fs_ldo % 4 == 0 && a.H * DK <= 1024,
                       "attention: FSMN rider needs taps, an output, strides % 4 and <= 1024 channels");
            fs_y = ceil_div(ceil_div(a.fs_T, 8), gx);
        }
        if (a.O2) PF_REQUIRE(a.ldo2 % 4 == 0 && a.o2_plane % 4 == 0 && ((uintptr_t)a.O2 & 7) == 0, "attention: plane output alignment");
        hipLaunchKernelGGL(attention_f32_fewq_kernel, dim3(gx, a.H + fs_y, a.B), dim3(256), 0, stream, a);
        PF_HIP_TRY(hipGetLastError());
        return 0;
    }
    PF_REQUIRE(!a.fs_in && !a.O2, "attention: the FSMN rider and the two-plane output exist in the few-query kernel only");
    dim3 grid(ceil_div(a.Tq, 128), a.H, a.B);
    // the K/V ring form of the streaming step (two sources, a few dozen keys) keeps the register-staged loader; the
    // offline form (one source) takes the LDS-DMA double-buffered kernel. Both do the same arithmetic in the same order.
    const bool dma_ok = a.K2 == nullptr && ((uintptr_t)a.K & 15) == 0 && ((uintptr_t)a.V & 15) == 0;
    if (dma_ok) hipLaunchKernelGGL(attention_f32_dma_kernel, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(attention_f32_kernel, grid, dim3(256), 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pf
