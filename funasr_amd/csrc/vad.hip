// Row-wise kernels of the FSMN-VAD network (funasr/models/fsmn_vad_streaming/encoder.py): the uni-directional FSMN memory
// block with its left-context cache, the softmax that is reduced straight to the silence posterior, and the per-frame
// energy in decibel the decision logic needs beside it. All three are HBM-bound streaming kernels; the dense layers of
// the network go through the GEMM kernels.
//
// Reference semantics:
//   FSMNBlock.forward  encoder.py:131-161   out[t] = x[t] + sum_k w[c][k] * z[t + k * lstride],  z = [cache | x]
//                                           (depthwise Conv2d, kernel [lorder, 1], no bias; the new cache is the last
//                                           (lorder - 1) * lstride rows of z; without a cache the left context is zero)
//   FSMN.forward       encoder.py:354-378   softmax over the output pdfs; GetFrameState (model.py:783-795) only ever
//                                           reads the summed posterior of the silence pdfs
//   ComputeDecibel     model.py:513-530     10 * log10(sum(frame^2) + 1e-6) over the raw samples of each 25 ms frame
#include "common.h"

namespace pf {

namespace {

constexpr int VAD_TT = 8;            // frames per workgroup

// block = (C / 4) x VAD_TT threads: thread (c4, tt) computes 4 channels of frame t0 + tt
__global__ __launch_bounds__(256) void vad_fsmn_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                       const float* __restrict__ cache, float* __restrict__ y, int ldy,
                                                       int T, int C, int L, int S) {
    const int c4 = threadIdx.x, tt = threadIdx.y;
    const int b = blockIdx.y, t = blockIdx.x * VAD_TT + tt;
    if (t >= T || c4 * 4 >= C) return;
    const int ctx = (L - 1) * S;
    const float* xb = x + (size_t)b * T * ldx;
    const float* cb = cache ? cache + (size_t)b * ctx * C : nullptr;
    float4 acc = *reinterpret_cast<const float4*>(xb + (size_t)t * ldx + c4 * 4);
    const float* wp = w + (size_t)c4 * 4 * L;
#pragma unroll 4
    for (int k = 0; k < L; ++k) {
        const int zi = t + k * S;                    // row of z = [cache (ctx rows) | x (T rows)]
        float4 v;
        if (zi >= ctx) v = *reinterpret_cast<const float4*>(xb + (size_t)(zi - ctx) * ldx + c4 * 4);
        else if (cb) v = *reinterpret_cast<const float4*>(cb + (size_t)zi * C + c4 * 4);
        else continue;
        acc.x = fmaf(wp[k], v.x, acc.x);
        acc.y = fmaf(wp[L + k], v.y, acc.y);
        acc.z = fmaf(wp[2 * L + k], v.z, acc.z);
        acc.w = fmaf(wp[3 * L + k], v.w, acc.w);
    }
    *reinterpret_cast<float4*>(y + ((size_t)b * T + t) * ldy + c4 * 4) = acc;
}

// new cache = last ctx rows of z = [old cache | x]; out-of-place (cache_out != cache_in) so that no row is read after
// it was overwritten
__global__ __launch_bounds__(256) void vad_cache_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ cin,
                                                        float* __restrict__ cout, int T, int C, int ctx) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;         // over ctx * C / 4
    const int c4n = C / 4;
    if (i >= ctx * c4n) return;
    const int r = i / c4n, c4 = i % c4n;
    const int zi = T + r;                                   // row of z (ctx + T rows) that becomes cache row r
    float4 v;
    if (zi >= ctx) v = *reinterpret_cast<const float4*>(x + ((size_t)b * T + (zi - ctx)) * ldx + c4 * 4);
    else v = *reinterpret_cast<const float4*>(cin + ((size_t)b * ctx + zi) * C + c4 * 4);
    *reinterpret_cast<float4*>(cout + ((size_t)b * ctx + r) * C + c4 * 4) = v;
}

// one wave per row: softmax over N <= 512 outputs, reduced to the summed posterior of the listed pdfs
struct SilIds { int n; int id[8]; };
__global__ __launch_bounds__(256) void vad_softmax_sil_kernel(const float* __restrict__ x, int ldx, int M, int N, SilIds ids,
                                                              float* __restrict__ p_sil, float* __restrict__ probs, int ldp) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    float v[8];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = lane + 64 * j;
        v[j] = c < N ? xr[c] : -INFINITY;
        mx = fmaxf(mx, v[j]);
    }
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = expf(v[j] - mx);
        s += v[j];
    }
    s = wave_sum(s);
    if (probs) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = lane + 64 * j;
            if (c < N) probs[(size_t)row * ldp + c] = v[j] / s;
        }
    }
    if (lane == 0) {
        float acc = 0.f;
        for (int i = 0; i < ids.n; ++i) acc += expf(xr[ids.id[i]] - mx) / s;
        p_sil[row] = acc;
    }
}

// one wave per frame
__global__ __launch_bounds__(256) void frame_decibel_kernel(const float* __restrict__ wav, int n_frames, int flen, int shift,
                                                            float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= n_frames) return;
    const float* p = wav + (size_t)f * shift;
    float s = 0.f;
    for (int i = lane; i < flen; i += 64) s = fmaf(p[i], p[i], s);
    s = wave_sum(s);
    if (lane == 0) out[f] = 10.0f * log10f(s + 0.000001f);
}

}  // namespace

int launch_vad_fsmn(const float* x, int ldx, const float* w, const float* cache_in, float* cache_out, float* y, int ldy,
                    int B, int T, int C, int L, int S, hipStream_t stream) {
    PF_REQUIRE(B > 0 && T > 0 && C % 4 == 0 && C <= 128 && L >= 1 && S >= 1, "vad_fsmn: C % 4 == 0, C <= 128");
    PF_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "vad_fsmn: strides % 4");
    PF_REQUIRE((cache_in == nullptr) == (cache_out == nullptr) && (cache_in == nullptr || cache_in != cache_out),
               "vad_fsmn: cache update is out of place");
    hipLaunchKernelGGL(vad_fsmn_kernel, dim3(ceil_div(T, VAD_TT), B), dim3(C / 4, VAD_TT), 0, stream, x, ldx, w, cache_in, y,
                       ldy, T, C, L, S);
    const int ctx = (L - 1) * S;
    if (cache_in && ctx > 0)
        hipLaunchKernelGGL(vad_cache_kernel, dim3(ceil_div(ctx * C / 4, 256), B), dim3(256), 0, stream, x, ldx, cache_in,
                           cache_out, T, C, ctx);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_vad_softmax_sil(const float* x, int ldx, int M, int N, const int* ids, int n_ids, float* p_sil, float* probs,
                           int ldp, hipStream_t stream) {
    PF_REQUIRE(M > 0 && N > 0 && N <= 512 && n_ids >= 1 && n_ids <= 8, "vad_softmax: N <= 512, 1..8 silence pdfs");
    SilIds s{};
    s.n = n_ids;
    for (int i = 0; i < n_ids; ++i) {
        PF_REQUIRE(ids[i] >= 0 && ids[i] < N, "vad_softmax: silence pdf id out of range");
        s.id[i] = ids[i];
    }
    hipLaunchKernelGGL(vad_softmax_sil_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, stream, x, ldx, M, N, s, p_sil, probs, ldp);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_frame_decibel(const float* wav, int n_frames, int flen, int shift, float* out, hipStream_t stream) {
    PF_REQUIRE(n_frames > 0 && flen > 0 && shift > 0, "frame_decibel: empty");
    hipLaunchKernelGGL(frame_decibel_kernel, dim3(ceil_div(n_frames, 4)), dim3(256), 0, stream, wav, n_frames, flen, shift, out);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pf
