// Feature normalisation between the frontend and the encoder: the reference's `normalize` modules (normalize_classes),
// applied by Paraformer.encode (funasr/models/paraformer/model.py:305-306) and SenseVoiceSmall (sense_voice/model.py:836-837).
//   UtteranceMVN  funasr/models/normalize/utterance_mvn.py:51-96   per-utterance mean (and variance) over the valid frames
//   GlobalMVN     funasr/models/normalize/global_mvn.py:66-92      corpus statistics from a stats file
// Both work in place on x [B, T, D] (D contiguous), HBM-bound: a workgroup owns 64 columns of one utterance, a wave reads
// 256-B row segments. The published recipes set `normalize: null`; these kernels exist so that a config which names one runs.
#include "common.h"

namespace pf {
namespace {

// The reference's arithmetic, step by step (utterance_mvn.py:71-96), including what it does to the padded rows:
//   x[pad] = 0; mean = sum_t x / len
//   norm_means:             x -= mean on EVERY row (padded rows become -mean)
//     and norm_vars:        var = sum over all T rows of x^2 / len (the padded rows' mean^2 included); std = max(sqrt(var), eps);
//                           x = x / sqrt(std)                                    [sqrt of the std: utterance_mvn.py:87]
//   norm_vars only:         y = x - mean with pads zero; var = sum y^2 / len; x /= max(sqrt(var), eps)
// Sums are accumulated in float64 and rounded once (the reference's float32 tree sum is within a few ulp of that).
__global__ __launch_bounds__(256) void utterance_mvn_kernel(float* __restrict__ x, const int* __restrict__ lens, int T, int D,
                                                            int norm_means, int norm_vars, float eps) {
    __shared__ double part[4][64];
    __shared__ float stat[64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool live = c < D;
    const int len = min(max(lens[b], 0), T);
    float* xb = x + (size_t)b * T * D;
    const float flen = (float)lens[b];
    double s = 0.0;
    if (live) for (int t = wave; t < len; t += 4) s += (double)xb[(size_t)t * D + c];
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0) stat[lane] = __fdiv_rn((float)(part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]), flen);
    __syncthreads();
    const float mean = stat[lane];
    if (!norm_vars) {
        if (live) for (int t = wave; t < T; t += 4) {
            const size_t i = (size_t)t * D + c;
            const float v = t < len ? xb[i] : 0.f;
            xb[i] = norm_means ? __fsub_rn(v, mean) : v;
        }
        return;
    }
    double q = 0.0;
    if (live) {
        if (norm_means) {
            for (int t = wave; t < T; t += 4) {
                const float v = __fsub_rn(t < len ? xb[(size_t)t * D + c] : 0.f, mean);
                q += (double)__fmul_rn(v, v);
            }
        } else {
            for (int t = wave; t < len; t += 4) {
                const float v = __fsub_rn(xb[(size_t)t * D + c], mean);
                q += (double)__fmul_rn(v, v);
            }
        }
    }
    __syncthreads();
    part[wave][lane] = q;
    __syncthreads();
    if (wave == 0) {
        const float var = __fdiv_rn((float)(part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]), flen);
        const float sd = fmaxf(__fsqrt_rn(var), eps);
        stat[lane] = norm_means ? __fsqrt_rn(sd) : sd;
    }
    __syncthreads();
    const float div = stat[lane];
    if (live) for (int t = wave; t < T; t += 4) {
        const size_t i = (size_t)t * D + c;
        const float v = t < len ? xb[i] : 0.f;
        xb[i] = __fdiv_rn(norm_means ? __fsub_rn(v, mean) : v, div);
    }
}

// global_mvn.py:78-92: x -= mean (norm_means); padded rows = 0; x /= std (norm_vars). Element-wise, bit for bit.
__global__ __launch_bounds__(256) void global_mvn_kernel(float* __restrict__ x, const int* __restrict__ lens, int T, int D,
                                                         const float* __restrict__ mean, const float* __restrict__ sd, int norm_means,
                                                         int norm_vars) {
    const int b = blockIdx.z, t = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= D) return;
    const size_t i = ((size_t)b * T + t) * D + c;
    float v = x[i];
    if (norm_means) v = __fsub_rn(v, mean[c]);
    if (t >= lens[b]) v = 0.f;
    if (norm_vars) v = __fdiv_rn(v, sd[c]);
    x[i] = v;
}

}  // namespace
}  // namespace pf

using namespace pf;

extern "C" {

int pf_utterance_mvn(float* x_dev, const int32_t* lens_dev, int32_t B, int32_t T, int32_t D, int32_t norm_means, int32_t norm_vars,
                     float eps, void* stream) {
    PF_REQUIRE(x_dev && lens_dev && B > 0 && T > 0 && D > 0 && B <= 65535, "utterance_mvn: null/empty");
    hipLaunchKernelGGL(utterance_mvn_kernel, dim3((unsigned)((D + 63) / 64), (unsigned)B), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x_dev, lens_dev, T, D, norm_means, norm_vars, eps);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int pf_global_mvn(float* x_dev, const int32_t* lens_dev, int32_t B, int32_t T, int32_t D, const float* mean_dev, const float* std_dev,
                  int32_t norm_means, int32_t norm_vars, void* stream) {
    PF_REQUIRE(x_dev && lens_dev && mean_dev && std_dev && B > 0 && T > 0 && D > 0 && B <= 65535 && T <= 65535, "global_mvn: null/empty");
    hipLaunchKernelGGL(global_mvn_kernel, dim3((unsigned)((D + 255) / 256), (unsigned)T, (unsigned)B), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x_dev, lens_dev, T, D, mean_dev, std_dev, norm_means, norm_vars);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // extern "C"
