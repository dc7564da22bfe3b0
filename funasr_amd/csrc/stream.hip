// Kernels of the streaming (chunked, stateful) Paraformer step. All of them are tiny and latency-bound: the whole
// step is meant to be captured once in a hipGraph and replayed, so every step-varying scalar lives in device
// memory (StreamDev) and is advanced by a kernel instead of being passed as a launch argument.
//
// Reference semantics:
//   window = [last 5 rows of the previous window | x * sqrt(d) + PE(start_idx ..)]   funasr/models/scama/encoder.py:480-503,
//                                                                                    funasr/models/transformer/embedding.py:468-482
//   K/V caches ("last look_back * chunk rows")                                       funasr/models/sanm/attention.py:343-361, :829-841
//   sequential integrate-and-fire with carried remainder                              funasr/models/paraformer/cif_predictor.py:343-392
//   causal decoder FSMN with carried left context                                     funasr/models/sanm/attention.py:606-625
#include "stream.h"

namespace pf {

namespace {

// ------------------------------------------------------------------------------------------------- window / PE
__global__ __launch_bounds__(256) void stream_embed_kernel(StreamEmbedArgs p) {
    const int D4 = p.Din >> 2;
    const int W = p.tail ? p.keep : p.keep + p.n;
    const int total = p.S * W * D4;
    const int start = p.st->start_idx;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int c4 = i % D4, row = (i / D4) % W, s = i / (D4 * W);
        float4 v;
        if (p.tail) {
            // tail chunk: the cached window itself is scaled a second time, no new position (encoder.py:496-502)
            const float4 c = reinterpret_cast<const float4*>(p.cache_feats)[(s * p.keep + row) * D4 + c4];
            v.x = __fmul_rn(c.x, p.scale); v.y = __fmul_rn(c.y, p.scale);
            v.z = __fmul_rn(c.z, p.scale); v.w = __fmul_rn(c.w, p.scale);
        } else if (row < p.keep) {
            v = reinterpret_cast<const float4*>(p.cache_feats)[(s * p.keep + row) * D4 + c4];
        } else {
            const int t = row - p.keep;
            int pos = start + t;
            pos = pos < p.pe_rows ? pos : p.pe_rows - 1;
            const float4 a = reinterpret_cast<const float4*>(p.feats)[(s * p.n + t) * D4 + c4];
            const float4 e = reinterpret_cast<const float4*>(p.pe)[pos * D4 + c4];
            v.x = __fadd_rn(__fmul_rn(a.x, p.scale), e.x); v.y = __fadd_rn(__fmul_rn(a.y, p.scale), e.y);
            v.z = __fadd_rn(__fmul_rn(a.z, p.scale), e.z); v.w = __fadd_rn(__fmul_rn(a.w, p.scale), e.w);
        }
        reinterpret_cast<float4*>(p.win)[(s * W + row) * D4 + c4] = v;
    }
}

// second pass (separate launch: every window row must have been read first): cache <- last `keep` rows of the window
__global__ __launch_bounds__(256) void stream_keep_kernel(const float* win, float* cache, int S, int W, int keep, int D4) {
    const int total = S * keep * D4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int c4 = i % D4, row = (i / D4) % keep, s = i / (D4 * keep);
        reinterpret_cast<float4*>(cache)[i] = reinterpret_cast<const float4*>(win)[(s * W + (W - keep) + row) * D4 + c4];
    }
}

// ---------------------------------------------------------------------------------------------------- K/V rings
__global__ __launch_bounds__(256) void ring_append_kernel(RingAppendArgs p) {
    const int C4 = p.cols >> 2;
    const int cap = (p.mod > 0 && p.st && p.st->enc_valid > 0) ? p.mod : p.cap;   // (RingAppendArgs.mod: after the first append)
    const int skip = p.rows > cap ? p.rows - cap : 0;          // only the newest `cap` rows can survive
    const int rows = p.rows - skip;
    const int total = p.S * rows * C4;
    const float* src_l = p.src + (size_t)blockIdx.y * p.src_layer;
    float* ring_l = p.ring + (size_t)blockIdx.y * p.ring_layer;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int c4 = i % C4, r = (i / C4) % rows, s = i / (C4 * rows);
        int wp;
        if (p.st) wp = p.st->enc_wp;
        else {
            if (p.gate_dev[s] < 1) continue;
            wp = p.wp_dev[s];
        }
        const int dst = (wp + skip + r) % cap;
        const float4 v = *reinterpret_cast<const float4*>(src_l + (size_t)(s * p.src_T + p.r0 + skip + r) * p.ldsrc + c4 * 4);
        reinterpret_cast<float4*>(ring_l)[((size_t)s * p.cap + dst) * C4 + c4] = v;
    }
}

// RingFoldArgs: runs after a step's appends and before its advance. R0 = rows of the untrimmed first append (global indices 0 .. R0-1,
// stored linearly); this step appended new_rows more at (R0 + r) mod `mod`. Of the old rows the last max(0, mod - new_rows) survive the
// trim; those with index >= mod move to index % mod (their targets hold rows that the trim drops, never this step's new rows)
__global__ __launch_bounds__(256) void ring_fold_kernel(RingFoldArgs p) {
    const int R0 = p.st->enc_valid;
    if (R0 <= p.mod) return;
    const int lo_keep = R0 + p.new_rows - p.mod;
    const int lo = lo_keep > p.mod ? lo_keep : p.mod;
    const int rows = R0 - lo;
    if (rows <= 0) return;
    const int C4 = p.cols >> 2;
    float* ring_l = p.ring + (size_t)blockIdx.y * p.ring_layer;
    const int total = p.S * rows * C4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int c4 = i % C4, r = (i / C4) % rows, s = i / (C4 * rows);
        const int src = lo + r, dst = src % p.mod;
        reinterpret_cast<float4*>(ring_l)[((size_t)s * p.cap + dst) * C4 + c4] =
            reinterpret_cast<const float4*>(ring_l)[((size_t)s * p.cap + src) * C4 + c4];
    }
}

__global__ void stream_advance_enc_kernel(StreamAdvanceArgs p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        StreamDev s = *p.st;
        s.start_idx += p.n_frames;
        if (p.enc_cap > 0 && p.enc_mod > 0) {
            // the first chunk's cache stays untrimmed, every later one is trimmed to enc_mod rows (sanm/attention.py:353-361)
            const bool first = s.enc_valid == 0;
            const int v = s.enc_valid + p.enc_rows;
            s.enc_valid = first ? (p.enc_rows < p.enc_cap ? p.enc_rows : p.enc_cap) : (v < p.enc_mod ? v : p.enc_mod);
            s.enc_wp = first ? p.enc_rows % p.enc_mod : (s.enc_wp + p.enc_rows) % p.enc_mod;
        } else if (p.enc_cap > 0) {
            const int v = s.enc_valid + p.enc_rows;
            s.enc_valid = v < p.enc_cap ? v : p.enc_cap;
            s.enc_wp = (s.enc_wp + p.enc_rows) % p.enc_cap;
        }
        s.step += 1;
        *p.st = s;
    }
}

__global__ void stream_advance_dec_kernel(StreamAdvanceArgs p) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= p.S || p.dec_cap <= 0) return;
    if (p.gate[s] < 1) return;                    // the reference skips the decoder when no token fired (:589-590)
    const int v = p.dec_valid[s] + p.dec_rows;
    p.dec_valid[s] = v < p.dec_cap ? v : p.dec_cap;
    p.dec_wp[s] = (p.dec_wp[s] + p.dec_rows) % p.dec_cap;
}

// ------------------------------------------------------------------------------------------- CIF, one chunk
// One wave per stream; lane l owns channels [8l, 8l+8) (+512 per extra pass). The integrate/fire decisions are
// wave-uniform scalars evaluated in float32 exactly in the reference's order (cif_predictor.py:360-383):
//   if alpha + integrate < thr:  integrate += alpha; frames += alpha * h
//   else: frames += (thr - integrate) * h; emit(frames); integrate += alpha; integrate -= thr; frames = integrate * h
template <int CPL>   // channels per lane
__global__ __launch_bounds__(64) void cif_chunk_kernel(CifChunkArgs p) {
    const int s = blockIdx.x, lane = threadIdx.x;
    const int c0 = lane * CPL;
    float frames[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) frames[j] = 0.f;
    float integrate = 0.f;
    int n = 0;
    const int T = 1 + p.W + (p.is_final ? 1 : 0);
    for (int t = 0; t < T; ++t) {
        float a;
        float h[CPL];
        if (t == 0) {
            a = p.cif_alpha[s];
#pragma unroll
            for (int j = 0; j < CPL; ++j) h[j] = p.cif_hidden[(size_t)s * p.D + c0 + j];
        } else if (t <= p.W) {
            const int w = t - 1;
            a = (w >= p.lo && w < p.hi) ? p.alphas[(size_t)s * p.ld_alpha + w] : 0.f;
#pragma unroll
            for (int j = 0; j < CPL; ++j) h[j] = p.hidden[((size_t)s * p.W + w) * p.D + c0 + j];
        } else {
            a = p.tail_threshold;
#pragma unroll
            for (int j = 0; j < CPL; ++j) h[j] = 0.f;
        }
        if (__fadd_rn(a, integrate) < p.threshold) {
            integrate = __fadd_rn(integrate, a);
#pragma unroll
            for (int j = 0; j < CPL; ++j) frames[j] = __fadd_rn(frames[j], __fmul_rn(a, h[j]));
        } else {
            const float rest = __fsub_rn(p.threshold, integrate);
#pragma unroll
            for (int j = 0; j < CPL; ++j) frames[j] = __fadd_rn(frames[j], __fmul_rn(rest, h[j]));
            if (n < p.Nmax) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) p.embeds[((size_t)s * p.Nmax + n) * p.D + c0 + j] = frames[j];
            }
            ++n;
            integrate = __fadd_rn(integrate, a);
            integrate = __fsub_rn(integrate, p.threshold);
#pragma unroll
            for (int j = 0; j < CPL; ++j) frames[j] = __fmul_rn(integrate, h[j]);
        }
    }
    for (int k = n < p.Nmax ? n : p.Nmax; k < p.Nmax; ++k)
#pragma unroll
        for (int j = 0; j < CPL; ++j) p.embeds[((size_t)s * p.Nmax + k) * p.D + c0 + j] = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j)
        p.cif_hidden[(size_t)s * p.D + c0 + j] = integrate > 0.f ? __fdiv_rn(frames[j], integrate) : frames[j];
    if (lane == 0) {
        p.cif_alpha[s] = integrate;
        p.n_fired[s] = n < p.Nmax ? n : p.Nmax;
    }
}

// -------------------------------------------------------------------------- decoder FSMN with carried context
// thread = (stream, 4 channels); sequence = [state (K-1 rows) | this chunk's n tokens]; out[k] = resid[k] +
// (sum_j w[j] * seq[k + j] + in[k]); new state = last K-1 rows of the sequence. Rows >= n are padding.
template <int KS, int NMAX>
__global__ __launch_bounds__(128) void dec_fsmn_chunk_kernel(DecFsmnChunkArgs p) {
    const int s = blockIdx.y;
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (c4 * 4 >= p.C) return;
    int n = p.n_valid[s];
    n = n < p.N ? n : p.N;
    float4 w[KS];
    {
        const float* wp = p.w + (size_t)c4 * 4 * KS;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            w[j].x = wp[j]; w[j].y = wp[KS + j]; w[j].z = wp[2 * KS + j]; w[j].w = wp[3 * KS + j];
        }
    }
    float4 seq[KS - 1 + NMAX];
#pragma unroll
    for (int i = 0; i < KS - 1; ++i)
        seq[i] = *reinterpret_cast<const float4*>(p.state + ((size_t)s * (KS - 1) + i) * p.C + c4 * 4);
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        seq[KS - 1 + k] = (k < p.N) ? *reinterpret_cast<const float4*>(p.in + ((size_t)s * p.N + k) * p.C + c4 * 4)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        if (k < p.N) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                acc.x = fmaf(w[j].x, seq[k + j].x, acc.x);
                acc.y = fmaf(w[j].y, seq[k + j].y, acc.y);
                acc.z = fmaf(w[j].z, seq[k + j].z, acc.z);
                acc.w = fmaf(w[j].w, seq[k + j].w, acc.w);
            }
            const float4 x = seq[KS - 1 + k];
            const float4 r = *reinterpret_cast<const float4*>(p.resid + ((size_t)s * p.N + k) * p.C + c4 * 4);
            float4 o;
            o.x = r.x + (acc.x + x.x); o.y = r.y + (acc.y + x.y); o.z = r.z + (acc.z + x.z); o.w = r.w + (acc.w + x.w);
            *reinterpret_cast<float4*>(p.out + ((size_t)s * p.N + k) * p.C + c4 * 4) = o;
        }
    }
    // new state: rows [n, n + KS - 1) of the sequence (n is wave-divergent only across streams = blocks)
#pragma unroll
    for (int i = 0; i < KS - 1; ++i) {
        float4 v = seq[i];
#pragma unroll
        for (int k = 1; k <= NMAX; ++k)
            if (n == k) v = seq[i + k];
        *reinterpret_cast<float4*>(p.state + ((size_t)s * (KS - 1) + i) * p.C + c4 * 4) = v;
    }
}

// The same block for at most 24 token rows (the 600 ms geometry), built for the latency of ONE stream: the kernel above walks a
// stream's 24 rows x 11 taps in two waves (12-22 us in the step's chain); here 512 threads = (4 channels) x (4 groups of 6 rows),
// every thread holds a 16-row window, the taps come in as eleven 16-B loads. Term by term the arithmetic of the kernel above
// (same bits). LN: the LayerNorms on either side ride along (DecFsmnChunkArgs.ln_*): norm2 applied to `in` on the fetch from the
// block partials w_2's epilogue left, the partials of the output rows left for norm3 on the fetch of the query projection.
template <int KS, bool LN>
__global__ __launch_bounds__(512) void dec_fsmn_chunk24_kernel(DecFsmnChunkArgs p) {
    constexpr int NMAX = 24, G = 4, R = NMAX / G, WIN = KS - 1 + R;
    const int s = blockIdx.x;
    const int c4 = threadIdx.x & 127, grp = threadIdx.x >> 7;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool active = c4 * 4 < p.C;
    __shared__ float2 ms[NMAX];
    if constexpr (LN) {
        if (p.ln_stats_in) {
            // (mean, rstd) per token row: wave w reduces rows w, w + 8, w + 16 (one load per lane and row), butterfly 32 .. 1 --
            // the reduction of gemm_skinny.hip's consumer
            const int nblk = p.C >> 4;
            // (sum, M2 about the block mean) partials of 16 channels (C <= 512: at most 32 blocks, one per lane), Chan's merge -- as
            // gemm_skinny.hip's consumer
            float2 bp[3];
            float sx[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int r = wave + 8 * i;
                bp[i] = make_float2(0.f, 0.f);
                if (r < p.N && lane < nblk) bp[i] = (reinterpret_cast<const float2*>(p.ln_stats_in) + ((size_t)s * p.N + r) * nblk)[lane];
                sx[i] = bp[i].x;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sx[i] += __shfl_xor(sx[i], o, 64);
                const float mean = sx[i] / (float)p.C;
                float m2 = 0.f;
                if (lane < nblk) { const float d = bp[i].x * 0.0625f - mean; m2 = bp[i].y + 16.f * d * d; }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 64);
                const float var = m2 / (float)p.C;
                if (lane == 0 && wave + 8 * i < p.N) ms[wave + 8 * i] = make_float2(mean, 1.0f / sqrtf(var + p.ln_eps));
            }
        }
    }
    int n = p.n_valid[s];
    n = n < p.N ? n : p.N;
    const int cc = active ? c4 * 4 : 0;
    // row i of the sequence [state (KS - 1 rows) | this chunk's rows], before the LayerNorm
    auto seq_raw = [&](int i) -> float4 {
        if (i < KS - 1) return *reinterpret_cast<const float4*>(p.state + ((size_t)s * (KS - 1) + i) * p.C + cc);
        const int k = i - (KS - 1);
        return k < p.N ? *reinterpret_cast<const float4*>(p.in + ((size_t)s * p.N + k) * p.C + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    const int k0 = grp * R;
    float4 win[WIN], res[R], nst[3];
#pragma unroll
    for (int i = 0; i < WIN; ++i) win[i] = seq_raw(k0 + i);
#pragma unroll
    for (int i = 0; i < R; ++i)
        res[i] = k0 + i < p.N ? *reinterpret_cast<const float4*>(p.resid + ((size_t)s * p.N + k0 + i) * p.C + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
    // the new state = rows [n, n + KS - 1) of the sequence: this group fetches rows grp, grp + 4, grp + 8 of it now and writes
    // them after every window of the workgroup has been read (the state is updated in place)
#pragma unroll
    for (int i = 0; i < 3; ++i) nst[i] = grp + G * i < KS - 1 ? seq_raw(n + grp + G * i) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 w[KS];
    {
        float4 t[KS];                                       // 4 channels x KS taps, contiguous
        const float4* wp = reinterpret_cast<const float4*>(p.w + (size_t)cc * KS);
#pragma unroll
        for (int j = 0; j < KS; ++j) t[j] = wp[j];
        const float* f = reinterpret_cast<const float*>(t);
#pragma unroll
        for (int j = 0; j < KS; ++j) { w[j].x = f[j]; w[j].y = f[KS + j]; w[j].z = f[2 * KS + j]; w[j].w = f[3 * KS + j]; }
    }
    // ms[]; and the state is updated in place: every load of it has RETURNED before any thread passes the barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (LN) {
        if (p.ln_stats_in) {
            const float4 g4 = *reinterpret_cast<const float4*>(p.ln_g + cc);
            const float4 b4 = *reinterpret_cast<const float4*>(p.ln_b + cc);
            auto norm = [&](float4& v, int i) {             // i = sequence index; state rows were normalised when they were token rows
#pragma clang fp contract(off)
                const int k = i - (KS - 1);
                if (k >= 0 && k < p.N) {
                    const float2 t = ms[k];
                    v.x = (v.x - t.x) * t.y * g4.x + b4.x;
                    v.y = (v.y - t.x) * t.y * g4.y + b4.y;
                    v.z = (v.z - t.x) * t.y * g4.z + b4.z;
                    v.w = (v.w - t.x) * t.y * g4.w + b4.w;
                }
            };
#pragma unroll
            for (int i = 0; i < WIN; ++i) norm(win[i], k0 + i);
#pragma unroll
            for (int i = 0; i < 3; ++i) norm(nst[i], n + grp + G * i);
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int k = k0 + i;
        if (k < p.N) {                                      // (uniform per group of two waves)
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                acc.x = fmaf(w[j].x, win[i + j].x, acc.x);
                acc.y = fmaf(w[j].y, win[i + j].y, acc.y);
                acc.z = fmaf(w[j].z, win[i + j].z, acc.z);
                acc.w = fmaf(w[j].w, win[i + j].w, acc.w);
            }
            const float4 x = win[KS - 1 + i];
            const float4 r = res[i];
            float4 o;
            o.x = r.x + (acc.x + x.x); o.y = r.y + (acc.y + x.y); o.z = r.z + (acc.z + x.z); o.w = r.w + (acc.w + x.w);
            if (active) *reinterpret_cast<float4*>(p.out + ((size_t)s * p.N + k) * p.C + cc) = o;
            if constexpr (LN) {
                if (p.ln_stats_out) {
                    // block partials of the output row: 16 channels = 4 consecutive lanes (C % 16 == 0: all four active or none)
                    float sx = (o.x + o.y) + (o.z + o.w);
                    sx += __shfl_xor(sx, 1, 64);
                    sx += __shfl_xor(sx, 2, 64);
                    const float mb = sx * 0.0625f;                // block mean; partials = (sum, sum of squared deviations from it)
                    const float dx = o.x - mb, dy = o.y - mb, dz = o.z - mb, dw = o.w - mb;
                    float m2 = (dx * dx + dy * dy) + (dz * dz + dw * dw);
                    m2 += __shfl_xor(m2, 1, 64);
                    m2 += __shfl_xor(m2, 2, 64);
                    if (active && (c4 & 3) == 0)
                        reinterpret_cast<float2*>(p.ln_stats_out)[((size_t)s * p.N + k) * (p.C >> 4) + (c4 >> 2)] = make_float2(sx, m2);
                }
            }
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (grp + G * i < KS - 1)
                *reinterpret_cast<float4*>(p.state + ((size_t)s * (KS - 1) + grp + G * i) * p.C + cc) = nst[i];
    }
}

__global__ void fill_int_kernel(int* p, int n, int v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

int launch_stream_embed(const StreamEmbedArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.Din % 4 == 0 && a.S > 0 && a.keep >= 0, "stream_embed: bad shape");
    const int W = a.tail ? a.keep : a.keep + a.n;
    PF_REQUIRE(W > 0, "stream_embed: empty window");
    const int total = a.S * W * (a.Din / 4);
    hipLaunchKernelGGL(stream_embed_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, stream, a);
    if (a.keep > 0 && W >= a.keep) {
        const int t2 = a.S * a.keep * (a.Din / 4);
        hipLaunchKernelGGL(stream_keep_kernel, dim3(ceil_div(t2, 256)), dim3(256), 0, stream, a.win, a.cache_feats, a.S, W,
                           a.keep, a.Din / 4);
    }
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_ring_append(const RingAppendArgs& a, hipStream_t stream) {
    if (a.rows <= 0 || a.cap <= 0) return 0;
    PF_REQUIRE(a.cols % 4 == 0 && a.ldsrc % 4 == 0, "ring_append: cols % 4");
    const int rows = a.rows > a.cap ? a.cap : a.rows;
    const int total = a.S * rows * (a.cols / 4);
    hipLaunchKernelGGL(ring_append_kernel, dim3(ceil_div(total, 256), a.n_layers > 1 ? a.n_layers : 1), dim3(256), 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_ring_fold(const RingFoldArgs& a, hipStream_t stream) {
    if (a.mod <= 0 || a.cap <= a.mod) return 0;
    PF_REQUIRE(a.cols % 4 == 0, "ring_fold: cols % 4");
    const int total = a.S * (a.cap - a.mod) * (a.cols / 4);
    hipLaunchKernelGGL(ring_fold_kernel, dim3(ceil_div(total, 256), a.n_layers > 1 ? a.n_layers : 1), dim3(256), 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_stream_advance_enc(const StreamAdvanceArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(stream_advance_enc_kernel, dim3(1), dim3(64), 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}
int launch_stream_advance_dec(const StreamAdvanceArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(stream_advance_dec_kernel, dim3(ceil_div(a.S, 64)), dim3(64), 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_cif_chunk(const CifChunkArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.D == 512, "cif_chunk: d_model 512 is built (8 channels per lane)");
    hipLaunchKernelGGL(cif_chunk_kernel<8>, dim3(a.S), dim3(64), 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_dec_fsmn_chunk(const DecFsmnChunkArgs& a, hipStream_t stream) {
    // the token rows of a step live in registers beside the carried context: the kernel is instantiated for 24 (the 600 ms
    // geometry: <= 17 fires per step), 48 and 96 token rows (e.g. chunk_size [0, 20, 10]: <= 43)
    PF_REQUIRE(a.C % 4 == 0 && a.N <= 96, "dec_fsmn_chunk: at most 96 token rows per step");
    dim3 grid(ceil_div(a.C / 4, 128), a.S), block(128);
    if (a.ln_stats_in || a.ln_stats_out)
        PF_REQUIRE(a.C % 16 == 0 && a.N <= 24 && a.C <= 512 && (!a.ln_stats_in || (a.ln_g && a.ln_b)),
                   "dec_fsmn_chunk: the LayerNorm-carrying form needs C % 16 == 0, C <= 512, <= 24 token rows, gamma and beta");
    if (a.N <= 24 && a.C <= 512 && a.C % 4 == 0 && ((uintptr_t)a.w & 15) == 0) {
        if (a.ln_stats_in || a.ln_stats_out) hipLaunchKernelGGL((dec_fsmn_chunk24_kernel<11, true>), dim3(a.S), dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((dec_fsmn_chunk24_kernel<11, false>), dim3(a.S), dim3(512), 0, stream, a);
    } else if (a.N <= 24) hipLaunchKernelGGL((dec_fsmn_chunk_kernel<11, 24>), grid, block, 0, stream, a);
    else if (a.N <= 48) hipLaunchKernelGGL((dec_fsmn_chunk_kernel<11, 48>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((dec_fsmn_chunk_kernel<11, 96>), grid, block, 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_fill_int(int* p, int n, int value, hipStream_t stream) {
    hipLaunchKernelGGL(fill_int_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, p, n, value);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pf
