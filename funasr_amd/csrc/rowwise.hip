// HBM-bound row-wise kernels of the SAN-M encoder/decoder: LayerNorm, input scale + sinusoidal PE, and the
// FSMN memory block (depthwise conv along time).
//
// Reference semantics:
//   LayerNorm            funasr/models/transformer/layer_norm.py:13-38 (nn.LayerNorm over the last dim, eps 1e-12;
//                        SenseVoice's copy funasr/models/sense_voice/model.py:300-323 uses eps 1e-5)
//   scale + PE           funasr/models/sanm/encoder.py:409 (x * sqrt(d_model)) and
//                        funasr/models/transformer/embedding.py:422-432 (x + PE, positions 1-based)
//   FSMN (encoder)       funasr/models/sanm/attention.py:216-239  mask*(conv_k(pad(mask*v)) + mask*v)
//   FSMN (decoder)       funasr/models/sanm/attention.py:583-631 + residual at paraformer/decoder.py:106-107
//
// All three are pure streaming ops: one coalesced 16-B load / store per lane, wave-shuffle reductions,
// nothing is reshaped into a GEMM.
#include "common.h"

namespace pf {

namespace {

__device__ __forceinline__ float4 load4(const float* base, size_t off, bool bf16) {
    if (!bf16) return *reinterpret_cast<const float4*>(base + off);
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off);
    float4 v;
    v.x = bf16_to_f32((unsigned short)(u.x & 0xffffu)); v.y = bf16_to_f32((unsigned short)(u.x >> 16));
    v.z = bf16_to_f32((unsigned short)(u.y & 0xffffu)); v.w = bf16_to_f32((unsigned short)(u.y >> 16));
    return v;
}

// One wave per row. NV = float4 chunks per lane (D <= 256 * NV).
// OUT: 0 fp32, 1 bf16, 2 three bf16 planes (split3) `plane` elements apart, 3 two fp16 planes of result * oscale
// seq_out > 0: output row r = b * seq_out + t reads input row b * seq_in + t (drops a padded layout's extra rows)
template <int NV, int OUT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        int ldy, int M, int D, int Dpad, float eps, int in_bf16,
                                                        size_t plane, float oscale, int seq_out, int seq_in,
                                                        const int* __restrict__ out_map) {
    const int lane = threadIdx.x & 63;
    const int xr0 = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (xr0 >= M) return;
    const int row = out_map ? out_map[xr0] : xr0;        // output row (packed layout -> the caller's rows)
    if (row < 0) return;
    const int xrow = out_map ? xr0 : (seq_out > 0 ? (row / seq_out) * seq_in + row % seq_out : row);
    const int nchunk = D >> 2;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) {
            v[j] = load4(x, (size_t)xrow * ldx + 4 * c, in_bf16 != 0);
            s += ln_sum4(v[j]);
        } else {
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    s = wave_sum(s);
    const float mean = ln_mean(s, D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) q += ln_sqdev4(v[j], mean);
    }
    q = wave_sum(q);
    const float rstd = ln_rstd(q, D, eps);
    float* yr = y + (size_t)row * ldy;
    unsigned short* yb = reinterpret_cast<unsigned short*>(y) + (size_t)row * ldy;     // bf16 view (ldy in elements)
    // plane output: lanes 2k / 2k + 1 hold neighbouring chunks of the row and pass the Dpad test together when Dpad % 8 == 0
    const bool wide_st = OUT == 3 && Dpad % 8 == 0 && ldy % 8 == 0 && plane % 8 == 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane + 64 * j;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nchunk) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * c);
            const float4 b = *reinterpret_cast<const float4*>(beta + 4 * c);
            o = ln_apply4(v[j], mean, rstd, g, b);
        }
        if (4 * c < Dpad) {
            if constexpr (OUT == 3) {
                const float ov[4] = {o.x, o.y, o.z, o.w};
                if (wide_st) store_split2x4_pair(yb + 4 * c, plane, ov, oscale, lane);
                else store_split2x4(yb + 4 * c, plane, ov, oscale);
            } else if constexpr (OUT == 2) {
                const float ov[4] = {o.x, o.y, o.z, o.w};
                store_split3x4(yb + 4 * c, plane, ov);
            } else if constexpr (OUT == 1) {
                uint2 pk;
                pk.x = (unsigned)f32_to_bf16(o.x) | ((unsigned)f32_to_bf16(o.y) << 16);
                pk.y = (unsigned)f32_to_bf16(o.z) | ((unsigned)f32_to_bf16(o.w) << 16);
                *reinterpret_cast<uint2*>(yb + 4 * c) = pk;
            } else {
                *reinterpret_cast<float4*>(yr + 4 * c) = o;
            }
        }
    }
}

// The second launch of a split-K GEMM (gemm_f16x2.hip) and the LayerNorm that follows it as one: a row's slices are added in
// slice order, + bias, R2 + (the epilogue order of splitk_reduce_kernel), the finished row goes to C (fp32) and -- normalised by
// the statements of layernorm_kernel above, same lane-to-chunk map, same reductions: the bits of the two launches -- to y.
// One wave per row; OUT as above (0 fp32, 3 two fp16 planes of result * oscale).
template <int NV, int OUT>
__global__ __launch_bounds__(256) void splitk_reduce_ln_kernel(const float* __restrict__ part, int slices, size_t slice_stride, int M, int D,
                                                               const float* __restrict__ bias, int relu, const float* R1, int ldr1, const float* R2,
                                                               int ldr2, float* C, int ldc, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, float* __restrict__ y, int ldy,
                                                               size_t plane, float oscale) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nchunk = D >> 2;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) {
            const float* src = part + (size_t)row * D + 4 * c;
            float4 t = *reinterpret_cast<const float4*>(src);
            for (int z = 1; z < slices; ++z) {
                const float4 u = *reinterpret_cast<const float4*>(src + (size_t)z * slice_stride);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            if (bias) {
                const float4 b = *reinterpret_cast<const float4*>(bias + 4 * c);
                t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
            }
            if (relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
            if (R1) {
                const float4 r = *reinterpret_cast<const float4*>(R1 + (size_t)row * ldr1 + 4 * c);
                t.x = t.x + r.x; t.y = t.y + r.y; t.z = t.z + r.z; t.w = t.w + r.w;
            }
            if (R2) {
                const float4 r = *reinterpret_cast<const float4*>(R2 + (size_t)row * ldr2 + 4 * c);
                t.x = r.x + t.x; t.y = r.y + t.y; t.z = r.z + t.z; t.w = r.w + t.w;
            }
            if (C) *reinterpret_cast<float4*>(C + (size_t)row * ldc + 4 * c) = t;
            v[j] = t;
            s += ln_sum4(t);
        } else {
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    s = wave_sum(s);
    const float mean = ln_mean(s, D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane + 64 * j;
        if (c < nchunk) q += ln_sqdev4(v[j], mean);
    }
    q = wave_sum(q);
    const float rstd = ln_rstd(q, D, eps);
    float* yr = y + (size_t)row * ldy;
    unsigned short* yb = reinterpret_cast<unsigned short*>(y) + (size_t)row * ldy;
    const bool wide_st = OUT == 3 && D % 8 == 0 && ldy % 8 == 0 && plane % 8 == 0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane + 64 * j;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < nchunk) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * c);
            const float4 b = *reinterpret_cast<const float4*>(beta + 4 * c);
            o = ln_apply4(v[j], mean, rstd, g, b);
        }
        if (4 * c < D) {
            if constexpr (OUT == 3) {
                const float ov[4] = {o.x, o.y, o.z, o.w};
                if (wide_st) store_split2x4_pair(yb + 4 * c, plane, ov, oscale, lane);
                else store_split2x4(yb + 4 * c, plane, ov, oscale);
            } else {
                *reinterpret_cast<float4*>(yr + 4 * c) = o;
            }
        }
    }
}

// constants of the LayerNorm form of the small-M GEMM (gemm_skinny.hip): one wave per weight row n
__global__ __launch_bounds__(256) void ln_consts_kernel(const float* __restrict__ W, int ldw, int N, int K, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ bias,
                                                        float* __restrict__ c1, float* __restrict__ c2) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const float* w = W + (size_t)n * ldw;
    float s1 = 0.f, s2 = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const float4 wv = *reinterpret_cast<const float4*>(w + k);
        const float4 g = *reinterpret_cast<const float4*>(gamma + k);
        const float4 b = *reinterpret_cast<const float4*>(beta + k);
        s1 += (wv.x * g.x + wv.y * g.y) + (wv.z * g.z + wv.w * g.w);
        s2 += (wv.x * b.x + wv.y * b.y) + (wv.z * b.z + wv.w * b.w);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) { c1[n] = s1; c2[n] = s2 + (bias ? bias[n] : 0.f); }
}

__global__ __launch_bounds__(256) void scale_add_pe_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ pe, float* __restrict__ y,
                                                           int T, int D4, float scale, size_t total4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t per_seq = (size_t)T * D4;
    for (; i < total4; i += stride) {
        const size_t r = i % per_seq;   // (t, d4) inside one sequence
        const float4 a = reinterpret_cast<const float4*>(x)[i];
        const float4 p = reinterpret_cast<const float4*>(pe)[r];
        float4 o;   // mul then add, two roundings like the reference (x * scale, then + pe)
        o.x = __fadd_rn(__fmul_rn(a.x, scale), p.x);
        o.y = __fadd_rn(__fmul_rn(a.y, scale), p.y);
        o.z = __fadd_rn(__fmul_rn(a.z, scale), p.z);
        o.w = __fadd_rn(__fmul_rn(a.w, scale), p.w);
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

// row-wise log-softmax over a vocabulary (decoder / CTC scores for the beam search, paraformer/model.py:345,
// transformer/scorers/ctc.py:46): one wave per row, three sweeps (max, sum of exp, write), fp32, libm expf / logf
__global__ __launch_bounds__(256) void log_softmax_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                          int M, int N) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    float mx = -INFINITY;
    for (int j = lane; j < N; j += 64) mx = fmaxf(mx, xr[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < N; j += 64) sum += expf(xr[j] - mx);
    sum = wave_sum(sum);
    const float lse = mx + logf(sum);
    float* yr = y + (size_t)row * ldy;
    for (int j = lane; j < N; j += 64) yr[j] = xr[j] - lse;
}

// the same into a padded layout: y is [B, Tp, D], rows t >= T are zero
__global__ __launch_bounds__(256) void scale_add_pe_pad_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                               float* __restrict__ y, int T, int Tp, int D4, float scale,
                                                               size_t total4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t per_seq = (size_t)Tp * D4;
    for (; i < total4; i += stride) {
        const size_t bq = i / per_seq, r = i % per_seq;
        const int t = (int)(r / D4);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < T) {
            const float4 a = reinterpret_cast<const float4*>(x)[bq * (size_t)T * D4 + r];
            const float4 p = reinterpret_cast<const float4*>(pe)[r];
            o.x = __fadd_rn(__fmul_rn(a.x, scale), p.x);
            o.y = __fadd_rn(__fmul_rn(a.y, scale), p.y);
            o.z = __fadd_rn(__fmul_rn(a.z, scale), p.z);
            o.w = __fadd_rn(__fmul_rn(a.w, scale), p.w);
        }
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

// packed row layout (one sequence after the other, no padding rows): row r takes the caller's row map[r] of [B, T, D]
__global__ __launch_bounds__(256) void scale_add_pe_rows_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                                float* __restrict__ y, const int* __restrict__ map, int T,
                                                                int D4, float scale, size_t total4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < total4; i += stride) {
        const int r = (int)(i / D4), c = (int)(i % D4);
        const int src = map[r];
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (src >= 0) {
            const float4 a = reinterpret_cast<const float4*>(x)[(size_t)src * D4 + c];
            const float4 p = reinterpret_cast<const float4*>(pe)[(size_t)(src % T) * D4 + c];
            o.x = __fadd_rn(__fmul_rn(a.x, scale), p.x);
            o.y = __fadd_rn(__fmul_rn(a.y, scale), p.y);
            o.z = __fadd_rn(__fmul_rn(a.z, scale), p.z);
            o.w = __fadd_rn(__fmul_rn(a.w, scale), p.w);
        }
        reinterpret_cast<float4*>(y)[i] = o;
    }
}

// FSMN memory block. Thread = 4 channels; block = (C/4 threads) x FSMN_TT consecutive frames of one sequence,
// a register sliding window of KS + TT - 1 masked input rows (each input row is fetched once per block).
constexpr int FSMN_TT = 8;
template <int KS, int LP>
__global__ __launch_bounds__(256) void fsmn_kernel(FsmnArgs p) {
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * FSMN_TT;
    const int c4 = threadIdx.x;
    if (c4 * 4 >= p.C) return;
    const int len = p.lens[b];
    const size_t base = p.offs ? (size_t)p.offs[b] : (size_t)b * p.T;
    // packed layout: sequence b owns rows [offs[b], offs[b + 1]) -- its len valid rows and possibly padding rows behind
    // them (the encoder keeps the first one: the predictor reads it); rows past that do not exist
    const int rows = p.offs ? p.offs[b + 1] - p.offs[b] : p.T;
    if (t0 >= rows) return;
    float4 w[KS];
    {
        const float* wp = p.w + (size_t)c4 * 4 * KS;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            w[j].x = wp[j];
            w[j].y = wp[KS + j];
            w[j].z = wp[2 * KS + j];
            w[j].w = wp[3 * KS + j];
        }
    }
    float4 win[KS + FSMN_TT - 1];
#pragma unroll
    for (int i = 0; i < KS + FSMN_TT - 1; ++i) {
        const int tt = t0 - LP + i;
        if (tt >= 0 && tt < p.T && tt < len)
            win[i] = load4(p.in, (base + tt) * p.ldin + c4 * 4, p.in_bf16 != 0);
        else
            win[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < FSMN_TT; ++i) {
        const int t = t0 + i;
        if (t < rows) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < len) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    acc.x = fmaf(w[j].x, win[i + j].x, acc.x);
                    acc.y = fmaf(w[j].y, win[i + j].y, acc.y);
                    acc.z = fmaf(w[j].z, win[i + j].z, acc.z);
                    acc.w = fmaf(w[j].w, win[i + j].w, acc.w);
                }
                const float4 c = win[i + LP];   // the (masked) input row itself
                o.x = acc.x + c.x; o.y = acc.y + c.y; o.z = acc.z + c.z; o.w = acc.w + c.w;
            }
            const size_t row = base + t;
            if (p.R) {
                const float4 r = *reinterpret_cast<const float4*>(p.R + row * p.ldr + c4 * 4);
                o.x = r.x + o.x; o.y = r.y + o.y; o.z = r.z + o.z; o.w = r.w + o.w;
            }
            *reinterpret_cast<float4*>(p.out + row * p.ldo + c4 * 4) = o;
        }
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        uint2 pk;
        pk.x = (unsigned)f32_to_bf16(v.x) | ((unsigned)f32_to_bf16(v.y) << 16);
        pk.y = (unsigned)f32_to_bf16(v.z) | ((unsigned)f32_to_bf16(v.w) << 16);
        reinterpret_cast<uint2*>(y)[i] = pk;
    }
}

}  // namespace

int launch_cast_bf16(const float* x, unsigned short* y, size_t n, hipStream_t stream) {
    PF_REQUIRE(n % 4 == 0, "cast_bf16: n % 4");
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, stream, x, y, n4);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_layernorm(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy, int M,
                     int D, int Dpad, float eps, hipStream_t stream, int out_mode, int in_bf16, size_t plane,
                     float oscale, int seq_out, int seq_in, const int* out_map) {
    PF_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 2048, "layernorm: D must be a multiple of 4 and <= 2048");
    PF_REQUIRE(Dpad >= D && Dpad % 4 == 0 && Dpad <= 2048 && ldy >= Dpad, "layernorm: bad Dpad/ldy");
    PF_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "layernorm: strides must be multiples of 4");
    PF_REQUIRE(((uintptr_t)y & 15) == 0, "layernorm: output must be 16-B aligned");
    dim3 grid(ceil_div(M, 4)), block(256);
    const int nv = ceil_div(Dpad / 4, 64);
#define PF_LN(NV_)                                                                                                 \
    do {                                                                                                          \
        if (out_mode == 3) hipLaunchKernelGGL((layernorm_kernel<NV_, 3>), grid, block, 0, stream, x, ldx, gamma, beta, y, ldy, M, D, Dpad, eps, in_bf16, plane, oscale, seq_out, seq_in, out_map); \
        else if (out_mode == 2) hipLaunchKernelGGL((layernorm_kernel<NV_, 2>), grid, block, 0, stream, x, ldx, gamma, beta, y, ldy, M, D, Dpad, eps, in_bf16, plane, oscale, seq_out, seq_in, out_map); \
        else if (out_mode == 1) hipLaunchKernelGGL((layernorm_kernel<NV_, 1>), grid, block, 0, stream, x, ldx, gamma, beta, y, ldy, M, D, Dpad, eps, in_bf16, plane, oscale, seq_out, seq_in, out_map); \
        else hipLaunchKernelGGL((layernorm_kernel<NV_, 0>), grid, block, 0, stream, x, ldx, gamma, beta, y, ldy, M, D, Dpad, eps, in_bf16, plane, oscale, seq_out, seq_in, out_map);         \
    } while (0)
    if (nv <= 2) PF_LN(2);
    else if (nv <= 3) PF_LN(3);
    else PF_LN(8);
#undef PF_LN
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_splitk_reduce_ln(const float* part, int slices, size_t slice_stride, int M, int D, const float* bias, int relu, const float* R1, int ldr1,
                            const float* R2, int ldr2, float* C, int ldc, const float* gamma, const float* beta, float eps, float* y, int ldy,
                            int out_mode, size_t plane, float oscale, hipStream_t stream) {
    PF_REQUIRE(M > 0 && D > 0 && D % 4 == 0 && D <= 2048 && slices >= 1 && part && y && gamma && beta, "splitk_reduce_ln: D % 4, D <= 2048");
    PF_REQUIRE((!C || ldc % 4 == 0) && ldy % 4 == 0 && (!R2 || ldr2 % 4 == 0) && (!R1 || (ldr1 % 4 == 0 && ((uintptr_t)R1 & 15) == 0)) &&
                   slice_stride % 4 == 0 && (out_mode == 0 || out_mode == 3),
               "splitk_reduce_ln: strides % 4; fp32 or two-plane output");
    PF_REQUIRE(((uintptr_t)part & 15) == 0 && ((uintptr_t)C & 15) == 0 && ((uintptr_t)y & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0) &&
               (!R2 || ((uintptr_t)R2 & 15) == 0), "splitk_reduce_ln: operands must be 16-B aligned");
    dim3 grid(ceil_div(M, 4)), block(256);
    // (NV as launch_layernorm picks it -- 2 up to 512 columns, 8 above 768: the same lane-to-chunk map, hence the same bits)
#define PF_RLN(NV_, OUT_) hipLaunchKernelGGL((splitk_reduce_ln_kernel<NV_, OUT_>), grid, block, 0, stream, part, slices, slice_stride, M, D, bias, relu, \
                                             R1, ldr1, R2, ldr2, C, ldc, gamma, beta, eps, y, ldy, plane, oscale)
    const int nv = ceil_div(D / 4, 64);
    PF_REQUIRE(nv <= 2 || nv > 3, "splitk_reduce_ln: widths of 513 .. 768 columns are not instantiated");
    if (nv <= 2) { if (out_mode == 3) PF_RLN(2, 3); else PF_RLN(2, 0); }
    else { if (out_mode == 3) PF_RLN(8, 3); else PF_RLN(8, 0); }
#undef PF_RLN
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_ln_consts(const float* W, int ldw, int N, int K, const float* gamma, const float* beta, const float* bias, float* c1, float* c2,
                     hipStream_t stream) {
    PF_REQUIRE(W && gamma && beta && c1 && c2 && N > 0 && K > 0 && K % 4 == 0 && ldw % 4 == 0 && ((uintptr_t)W & 15) == 0 &&
               ((uintptr_t)gamma & 15) == 0 && ((uintptr_t)beta & 15) == 0, "ln_consts: K % 4, 16-B aligned operands");
    hipLaunchKernelGGL(ln_consts_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, stream, W, ldw, N, K, gamma, beta, bias, c1, c2);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void scale_cols_kernel(float* __restrict__ x, int ld, int M, int N4, float sc) {
    const size_t total = (size_t)M * N4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        float4* p = reinterpret_cast<float4*>(x + (i / N4) * (size_t)ld) + i % N4;
        float4 v = *p;
        v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        *p = v;
    }
}
// x[r, 0..N) *= sc for M rows of stride ld (N, ld multiples of 4, x 16-B aligned)
int launch_scale_cols(float* x, int ld, int M, int N, float sc, hipStream_t stream) {
    PF_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && ld % 4 == 0 && ((uintptr_t)x & 15) == 0, "scale_cols: alignment");
    const size_t total = (size_t)M * (N / 4);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(scale_cols_kernel, dim3(blocks), dim3(256), 0, stream, x, ld, M, N / 4, sc);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void scatter_i32_kernel(const int* __restrict__ in, const int* __restrict__ map, int* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[map[i]] = in[i];
}
int launch_scatter_i32(const int* in, const int* map, int* out, int n, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(scatter_i32_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, in, map, out, n);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_log_softmax(const float* x, int ldx, float* y, int ldy, int M, int N, hipStream_t stream) {
    PF_REQUIRE(M > 0 && N > 0 && ldx >= N && ldy >= N, "log_softmax: bad shape");
    hipLaunchKernelGGL(log_softmax_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, stream, x, ldx, y, ldy, M, N);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_scale_add_pe(const float* x, const float* pe, float* y, int B, int T, int D, float scale,
                        hipStream_t stream, int Tp) {
    PF_REQUIRE(B > 0 && T > 0 && D % 4 == 0, "scale_add_pe: D must be a multiple of 4");
    if (Tp > T) {
        const size_t tot = (size_t)B * Tp * (D / 4);
        const int nb = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
        hipLaunchKernelGGL(scale_add_pe_pad_kernel, dim3(nb), dim3(256), 0, stream, x, pe, y, T, Tp, D / 4, scale, tot);
        PF_HIP_TRY(hipGetLastError());
        return 0;
    }
    const size_t total4 = (size_t)B * T * (D / 4);
    const int blocks = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(scale_add_pe_kernel, dim3(blocks), dim3(256), 0, stream, x, pe, y, T, D / 4, scale, total4);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_scale_add_pe_rows(const float* x, const float* pe, float* y, const int* map, int M, int T, int D, float scale,
                             hipStream_t stream) {
    PF_REQUIRE(M > 0 && T > 0 && D % 4 == 0 && map, "scale_add_pe_rows: bad arguments");
    const size_t tot = (size_t)M * (D / 4);
    const int nb = (int)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096);
    hipLaunchKernelGGL(scale_add_pe_rows_kernel, dim3(nb), dim3(256), 0, stream, x, pe, y, map, T, D / 4, scale, tot);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_fsmn(const FsmnArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.B > 0 && a.T > 0 && a.C % 4 == 0 && a.C <= 1024, "fsmn: C must be a multiple of 4 and <= 1024");
    PF_REQUIRE(a.ldin % 4 == 0 && a.ldo % 4 == 0 && (a.R == nullptr || a.ldr % 4 == 0), "fsmn: strides % 4");
    dim3 grid(ceil_div(a.T, FSMN_TT), a.B), block(a.C / 4);
    if (a.K == 11 && a.left_pad == 5)
        hipLaunchKernelGGL((fsmn_kernel<11, 5>), grid, block, 0, stream, a);
    else if (a.K == 11 && a.left_pad == 10)
        hipLaunchKernelGGL((fsmn_kernel<11, 10>), grid, block, 0, stream, a);
    else if (a.K == 21 && a.left_pad == 10)          // SeACo's bias decoder (seaco_paraformer/template.yaml: kernel_size 21)
        hipLaunchKernelGGL((fsmn_kernel<21, 10>), grid, block, 0, stream, a);
    else {
        set_error("fsmn: built for kernel_size 11 with left padding 5 (offline) or 10 (sanm_shfit 5) and kernel_size 21 "
                  "with left padding 10");
        return -1;
    }
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pf
