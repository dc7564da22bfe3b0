// LSTM recurrence and the upsampled-weight head of CifPredictorV3 (funasr/models/bicif_paraformer/cif_predictor.py:301-352,
// `blstm` = torch.nn.LSTM(idim, idim, 1, bidirectional) over the 3x upsampled encoder output; the same recurrence serves the
// hotword LSTM of SeACo-Paraformer).
//
// Layouts are chosen so that every access of the per-step kernel is coalesced with lane = utterance:
//   * input projections `pre` are GATES-MAJOR: row n = dir * 4H + gate * H + unit (torch's i, f, g, o order), column
//     t * B + b -- produced by the fp32 MFMA GEMM as W_ih . X_tm^T, X_tm being the time-major [T, B, D] input;
//   * the recurrent state h, c is UNIT-MAJOR [dir][H][Bs] (Bs = B rounded up to 64), so the k-loop reads 256 contiguous
//     bytes per k and the new state is written the same way;
//   * W_hh is re-laid at load time to [dir][unit][k][4 gates]: one 16-byte wave-uniform load per k feeds the four gate
//     accumulators of a lane.
// One launch per time step (both directions in the same launch, grid = H/4 x ndir x ceil(B/64) workgroups of 4 waves; wave =
// hidden unit, lane = utterance): no inter-workgroup synchronisation inside a kernel, the step order is the stream order.
// The step is bound by the fp32 vector rate (B * 4H * H FMAs per direction), one L2 round trip and the launch latency, not
// by HBM: W_hh (4 MB per direction) and the state come from the L2 / Infinity Cache every step.
#include "common.h"
#include "lstm.h"

namespace pf {

namespace {

constexpr int LSTM_MAX_H = 512;   // hidden units whose W_hh slice (4 units x H x 4 gates) fits the 32 KB LDS block
constexpr int LSTM_KC = 64;       // k-rows of the state staged in LDS per pass (64 x 64 floats = 16 KB)
constexpr int LSTM_HREG = LSTM_MAX_H * 16 / 256;   // float4 registers per thread holding the whole state tile
constexpr int LSTM_WREG = LSTM_MAX_H * 4 / 256;    // float4 registers per thread holding the W_hh slice

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Everything a step reads is requested from L2 at once, before any arithmetic: the workgroup's W_hh slice (4 units, 32 KB at
// H = 512) and its [H][64] tile of the previous state (128 KB) are loaded into registers with coalesced 16-byte loads, so
// the step pays ONE memory latency instead of one per k-chunk (the first version streamed W through dependent scalar loads:
// 22 us per step at H = 512). W goes to LDS once and is read back as a wave-uniform broadcast, the state
// tile passes through LDS in 64-row chunks.
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmStepArgs p) {
    __shared__ float4 s_w[4 * LSTM_MAX_H];
    __shared__ float s_h[LSTM_KC * 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int dir = blockIdx.y;
    const int u = blockIdx.x * 4 + wave;
    const int b0 = blockIdx.z * 64;
    const int b = b0 + lane;
    const int H = p.H;
    const int t = dir == 0 ? p.step : p.T - 1 - p.step;
    const bool valid = b < p.B;
    const float* hp = p.h_prev + (size_t)dir * H * p.Bs + b0;
    const float4* wsrc = reinterpret_cast<const float4*>(p.whh) + ((size_t)dir * H + (size_t)blockIdx.x * 4) * H;
    float4 wreg[LSTM_WREG], hreg[LSTM_HREG];
#pragma unroll
    for (int j = 0; j < LSTM_WREG; ++j) {
        const int q = tid + 256 * j;
        wreg[j] = q < 4 * H ? wsrc[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < LSTM_HREG; ++j) {
        const int q = tid + 256 * j;                  // float4 q of the [H][64] tile: row q / 16, columns 4 (q % 16) ..
        hreg[j] = q < 16 * H ? *reinterpret_cast<const float4*>(hp + (size_t)(q >> 4) * p.Bs + (q & 15) * 4)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const size_t n = (size_t)dir * 4 * H + (size_t)g * H + u;
        const float x = valid ? p.pre[n * p.ld_pre + (size_t)t * p.B + b] : 0.f;
        acc[g] = x + (p.b_ih[n] + p.b_hh[n]);
    }
#pragma unroll
    for (int j = 0; j < LSTM_WREG; ++j) s_w[tid + 256 * j] = wreg[j];
    const float4* wl = s_w + wave * H;
    // chunk c holds tile rows [64 c, 64 c + 64) = float4s [1024 c, 1024 c + 1024) = registers 4 c .. 4 c + 3 of every thread
#pragma unroll
    for (int c = 0; c < LSTM_MAX_H / LSTM_KC; ++c) {
        if (c * LSTM_KC < H) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) reinterpret_cast<float4*>(s_h)[tid + 256 * j] = hreg[4 * c + j];
            __syncthreads();
            const int kn = min(LSTM_KC, H - c * LSTM_KC);
#pragma unroll 8
            for (int k = 0; k < kn; ++k) {
                const float hv = s_h[k * 64 + lane];
                const float4 w = wl[c * LSTM_KC + k];
                acc[0] = fmaf(w.x, hv, acc[0]);
                acc[1] = fmaf(w.y, hv, acc[1]);
                acc[2] = fmaf(w.z, hv, acc[2]);
                acc[3] = fmaf(w.w, hv, acc[3]);
            }
        }
    }
    const float ig = sigmoidf_(acc[0]), fg = sigmoidf_(acc[1]), gg = tanhf(acc[2]), og = sigmoidf_(acc[3]);
    const size_t si = ((size_t)dir * H + u) * p.Bs + b;
    const float cn = fg * p.c[si] + ig * gg;
    const float hn = og * tanhf(cn);
    p.c[si] = cn;
    p.h_next[si] = hn;
    const int C = p.ndir * H;
    if (p.out_layout == 1) {
        p.out[((size_t)t * C + (size_t)dir * H + u) * p.Bs + b] = hn;
    } else if (valid) {
        p.out[((size_t)b * p.T + t) * C + (size_t)dir * H + u] = hn;
    }
}

// rows [B, T, D] -> [T, B, D] (float4 granules)
__global__ __launch_bounds__(256) void rows_bt_to_tb_kernel(const float4* __restrict__ in, float4* __restrict__ out, int B,
                                                            int T, int D4) {
    const size_t total = (size_t)B * T * D4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % D4);
        const size_t row = i / D4;
        const int b = (int)(row % B), t = (int)(row / B);       // destination row = t * B + b
        out[i] = in[((size_t)b * T + t) * D4 + c];
    }
}

// second head of CifPredictorV3 on the unit-major LSTM output [T][C][Bs]: one thread per (frame, utterance),
// alpha = relu(sigmoid(dot(out[t, :, b], w) + bias) * smooth - noise), zero past the utterance's U * len frames
__global__ __launch_bounds__(256) void us_alpha_t_kernel(UsAlphaArgs p) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int t = (int)(idx / p.Bs), b = (int)(idx % p.Bs);
    if (t >= p.T || b >= p.B) return;
    const float* x = p.out_t + (size_t)t * p.C * p.Bs + b;
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < p.C; ++c) s = fmaf(x[(size_t)c * p.Bs], p.w[c], s);
    const float z = s + p.bias[0];
    float a = 1.0f / (1.0f + expf(-z));
    a = fmaxf(__fsub_rn(__fmul_rn(a, p.smooth), p.noise), 0.f);
    if (t >= p.U * p.lens[b]) a = 0.f;
    p.alphas[(size_t)b * p.T + t] = a;
}

// get_upsample_timestamp's tail (:341-351), one wave per utterance: alphas *= token_num / sum(alphas), then cif_wo_hidden
// (:89-118) at `threshold`: peaks[t] = running integral, the threshold taken off after a fire; fp32 sequential like the
// reference loop (the row is staged through LDS in chunks so that the serial part never waits on HBM)
constexpr int US_CHUNK = 4096;
__global__ __launch_bounds__(64) void us_scale_scan_kernel(float* __restrict__ alphas, float* __restrict__ peaks,
                                                           const int* __restrict__ token_num, int T, float threshold) {
    __shared__ float s_a[US_CHUNK], s_p[US_CHUNK];
    __shared__ float s_carry;
    const int b = blockIdx.x, lane = threadIdx.x;
    float* al = alphas + (size_t)b * T;
    float* pk = peaks + (size_t)b * T;
    double part = 0.0;
    for (int t = lane; t < T; t += 64) part += (double)al[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    const float total = (float)part;
    const float ratio = (float)token_num[b] / total;
    if (lane == 0) s_carry = 0.f;
    for (int t0 = 0; t0 < T; t0 += US_CHUNK) {
        const int n = min(US_CHUNK, T - t0);
        for (int t = lane; t < n; t += 64) {
            const float a = __fmul_rn(al[t0 + t], ratio);
            s_a[t] = a;
            al[t0 + t] = a;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            float integrate = s_carry;
#pragma unroll 4
            for (int t = 0; t < n; ++t) {
                integrate = __fadd_rn(integrate, s_a[t]);
                s_p[t] = integrate;
                if (integrate >= threshold) integrate = __fsub_rn(integrate, threshold);
            }
            s_carry = integrate;
        }
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < n; t += 64) pk[t0 + t] = s_p[t];
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

int launch_lstm_steps(const LstmStepArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.H > 0 && a.H % 4 == 0 && a.H <= LSTM_MAX_H && a.B > 0 && a.T > 0 && a.Bs % 64 == 0 && a.Bs >= a.B &&
                   (a.ndir == 1 || a.ndir == 2),
               "lstm: need H % 4 == 0, H <= 512, ndir 1 or 2, state stride a multiple of 64");
    PF_REQUIRE(a.pre && a.whh && a.b_ih && a.b_hh && a.h_a && a.h_b && a.c && a.out, "lstm: null argument");
    const size_t state = sizeof(float) * (size_t)a.ndir * a.H * a.Bs;
    PF_HIP_TRY(hipMemsetAsync(a.h_a, 0, state, stream));
    PF_HIP_TRY(hipMemsetAsync(a.c, 0, state, stream));
    const dim3 grid((unsigned)(a.H / 4), (unsigned)a.ndir, (unsigned)(a.Bs / 64)), block(256);
    LstmStepArgs s = a;
    for (int step = 0; step < a.T; ++step) {
        s.step = step;
        s.h_prev = (step & 1) ? a.h_b : a.h_a;
        s.h_next = (step & 1) ? a.h_a : a.h_b;
        hipLaunchKernelGGL(lstm_step_kernel, grid, block, 0, stream, s);
    }
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_rows_bt_to_tb(const float* in, float* out, int B, int T, int D, hipStream_t stream) {
    PF_REQUIRE(in && out && B > 0 && T > 0 && D > 0 && D % 4 == 0, "rows_bt_to_tb: bad argument");
    const size_t total = (size_t)B * T * (D / 4);
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 65535u * 4);
    hipLaunchKernelGGL(rows_bt_to_tb_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const float4*>(in),
                       reinterpret_cast<float4*>(out), B, T, D / 4);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_us_alpha_t(const UsAlphaArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.out_t && a.w && a.bias && a.lens && a.alphas && a.B > 0 && a.T > 0 && a.C > 0 && a.Bs >= a.B && a.U > 0,
               "us_alpha: bad argument");
    const size_t total = (size_t)a.T * a.Bs;
    hipLaunchKernelGGL(us_alpha_t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_us_scale_scan(float* alphas, float* peaks, const int* token_num, int B, int T, float threshold,
                         hipStream_t stream) {
    PF_REQUIRE(alphas && peaks && token_num && B > 0 && T > 0, "us_scale_scan: bad argument");
    hipLaunchKernelGGL(us_scale_scan_kernel, dim3((unsigned)B), dim3(64), 0, stream, alphas, peaks, token_num, T, threshold);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pf
