// Scaled-dot-product attention for small heads (d_k <= 64, any multiple of 4) and short sequences: the shape of the
// CT-Transformer punctuation model (funasr/models/ct_transformer/template.yaml: d_model 256, 8 heads of d_k = 32, a few
// dozen to ~220 tokens, batch 1), where a 128-wide MFMA tile per head has nothing to chew on. Same semantics as
// attention_f32.hip (funasr/models/sanm/attention.py:270-306: (q * d_k^-0.5) . k, key mask -inf, softmax, . V), plain fp32
// FMAs: one wave per (sequence, head, query); lane l scores keys l, l + 64, ...; softmax statistics by wave shuffles;
// each lane accumulates p . V over its keys and the d_k partial sums are reduced across the wave.
#include "common.h"

namespace pf {

namespace {

constexpr int SMALL_MAX_DK = 64;
constexpr int SMALL_MAX_KEYS_PER_LANE = 16;       // Tk <= 1024

__global__ __launch_bounds__(256) void attention_small_kernel(AttnArgs p, int dk) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int head = blockIdx.y, b = blockIdx.z;
    if (q >= p.Tq) return;
    const int klen = p.klens[b];
    // which keys this query may see (key padding; SANMVadEncoder: causal, or the VAD corner)
    int kend = klen, corner = 0x7fffffff;
    if (p.mask_mode == 1) kend = klen < q + 1 ? klen : q + 1;
    else if (p.mask_mode == 2) {
        const int vp = p.vad_pos[b];
        if (vp > 0 && vp < p.Tk && q < vp - 1) corner = vp;
    }
    auto visible = [&](int key) { return key < kend && key < corner; };
    const float* qp = p.Q + ((size_t)b * p.Tq + q) * p.ldq + head * dk;
    const float* kb = p.K + (size_t)b * p.Tk * p.ldk + head * dk;
    const float* vb = p.V + (size_t)b * p.Tk * p.ldv + head * dk;
    float qv[SMALL_MAX_DK];
#pragma unroll
    for (int d = 0; d < SMALL_MAX_DK; d += 4) {
        if (d < dk) {
            const float4 t = *reinterpret_cast<const float4*>(qp + d);
            qv[d] = t.x * p.scale; qv[d + 1] = t.y * p.scale; qv[d + 2] = t.z * p.scale; qv[d + 3] = t.w * p.scale;
        }
    }
    float s[SMALL_MAX_KEYS_PER_LANE];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < SMALL_MAX_KEYS_PER_LANE; ++j) {
        const int key = lane + 64 * j;
        float acc = -INFINITY;
        if (visible(key)) {
            const float* kp = kb + (size_t)key * p.ldk;
            acc = 0.f;
#pragma unroll
            for (int d = 0; d < SMALL_MAX_DK; d += 4) {
                if (d < dk) {
                    const float4 t = *reinterpret_cast<const float4*>(kp + d);
                    acc = fmaf(qv[d], t.x, acc); acc = fmaf(qv[d + 1], t.y, acc);
                    acc = fmaf(qv[d + 2], t.z, acc); acc = fmaf(qv[d + 3], t.w, acc);
                }
            }
        }
        s[j] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < SMALL_MAX_KEYS_PER_LANE; ++j) {
        s[j] = visible(lane + 64 * j) ? expf(s[j] - mx) : 0.f;
        sum += s[j];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    float* op = p.O + ((size_t)b * p.Tq + q) * p.ldo + head * dk;
    for (int d0 = 0; d0 < dk; d0 += 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < SMALL_MAX_KEYS_PER_LANE; ++j) {
            const int key = lane + 64 * j;
            if (visible(key)) {
                const float4 t = *reinterpret_cast<const float4*>(vb + (size_t)key * p.ldv + d0);
                acc.x = fmaf(s[j], t.x, acc.x); acc.y = fmaf(s[j], t.y, acc.y);
                acc.z = fmaf(s[j], t.z, acc.z); acc.w = fmaf(s[j], t.w, acc.w);
            }
        }
        acc.x = wave_sum(acc.x); acc.y = wave_sum(acc.y); acc.z = wave_sum(acc.z); acc.w = wave_sum(acc.w);
        if (lane == 0) *reinterpret_cast<float4*>(op + d0) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, int ld, const int* __restrict__ ids,
                                                          float* __restrict__ out, int n, int D4, int rows) {
    const size_t total = (size_t)n * D4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / D4), c = (int)(i % D4);
        int id = ids[r];
        id = id < 0 ? 0 : (id >= rows ? rows - 1 : id);
        reinterpret_cast<float4*>(out)[i] = *reinterpret_cast<const float4*>(table + (size_t)id * ld + 4 * c);
    }
}

}  // namespace

bool attention_small_applicable(const AttnArgs& a, int dk) {
    return dk <= SMALL_MAX_DK && dk % 4 == 0 && a.Tk <= 64 * SMALL_MAX_KEYS_PER_LANE && a.K2 == nullptr && a.O3 == nullptr;
}

// ---- attention-score filter of SeACo (seaco_paraformer/model.py:323-349 over decoder.py:485-513 `forward_asf6` and
//      attention.py:760-784 with ret_attn): the softmax attention matrix of ONE sequence, summed over heads and queries.
// one wave per (query, head): scores over all keys (lane-per-key), softmax, masked keys 0 -> P[(h * N + q) * T + k]
__global__ __launch_bounds__(64) void asf_probs_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K, int ldk,
                                                       float* __restrict__ P, int N, int T, int klen, int dk, float scale) {
    __shared__ float qs[128];
    const int q = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    for (int d = lane; d < dk; d += 64) qs[d] = Q[(size_t)q * ldq + h * dk + d] * scale;
    __syncthreads();
    float* pr = P + ((size_t)h * N + q) * T;
    float mx = -INFINITY;
    for (int k = lane; k < T; k += 64) {
        float sc = -INFINITY;
        if (k < klen) {
            const float* kp = K + (size_t)k * ldk + h * dk;
            sc = 0.f;
            for (int d = 0; d < dk; d += 4) {
                const float4 kv = *reinterpret_cast<const float4*>(kp + d);
                sc = fmaf(qs[d], kv.x, sc); sc = fmaf(qs[d + 1], kv.y, sc); sc = fmaf(qs[d + 2], kv.z, sc); sc = fmaf(qs[d + 3], kv.w, sc);
            }
        }
        pr[k] = sc;
        mx = fmaxf(mx, sc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int k = lane; k < T; k += 64) sum += (k < klen) ? expf(pr[k] - mx) : 0.f;
    sum = wave_sum(sum);
    for (int k = lane; k < T; k += 64) pr[k] = (k < klen) ? expf(pr[k] - mx) / sum : 0.f;
}
// out[k] = sum_q (sum_h P[h][q][k]): attn[0].sum(0).sum(0)
__global__ __launch_bounds__(256) void asf_colsum_kernel(const float* __restrict__ P, int H, int N, int T, float* __restrict__ out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= T) return;
    float acc = 0.f;
    for (int q = 0; q < N; ++q) {
        float t = 0.f;
        for (int h = 0; h < H; ++h) t += P[((size_t)h * N + q) * T + k];
        acc += t;
    }
    out[k] = acc;
}

int launch_asf_scores(const float* Q, int ldq, const float* K, int ldk, float* P_scratch, float* out, int H, int dk, int N, int T,
                      int klen, float scale, hipStream_t stream) {
    PF_REQUIRE(H > 0 && N > 0 && T > 0 && klen >= 1 && klen <= T && dk % 4 == 0 && dk <= 128 && ldk % 4 == 0, "asf_scores: bad shape");
    hipLaunchKernelGGL(asf_probs_kernel, dim3(N, H), dim3(64), 0, stream, Q, ldq, K, ldk, P_scratch, N, T, klen, dk, scale);
    hipLaunchKernelGGL(asf_colsum_kernel, dim3(ceil_div(T, 256)), dim3(256), 0, stream, P_scratch, H, N, T, out);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_attention_small(const AttnArgs& a, int dk, hipStream_t stream) {
    PF_REQUIRE(a.B > 0 && a.H > 0 && a.Tq > 0 && a.Tk > 0, "attention: empty problem");
    PF_REQUIRE(attention_small_applicable(a, dk), "attention_small: d_k <= 64 (multiple of 4), Tk <= 1024, one key/value source");
    PF_REQUIRE(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.ldo % 4 == 0 && a.O, "attention_small: strides % 4");
    hipLaunchKernelGGL(attention_small_kernel, dim3(ceil_div(a.Tq, 4), a.H, a.B), dim3(256), 0, stream, a, dk);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_gather_rows(const float* table, int ld, int rows, const int* ids, float* out, int n, int D, hipStream_t stream) {
    PF_REQUIRE(n > 0 && D > 0 && D % 4 == 0 && ld % 4 == 0 && rows > 0, "gather_rows: D and ld must be multiples of 4");
    const size_t total = (size_t)n * (D / 4);
    const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, stream, table, ld, ids, out, n, D / 4, rows);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pf
