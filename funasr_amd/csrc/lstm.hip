// LSTM recurrence and the upsampled-weight head of CifPredictorV3 (funasr/models/bicif_paraformer/cif_predictor.py:301-352,
// `blstm` = torch.nn.LSTM(idim, idim, 1, bidirectional) over the 3x upsampled encoder output; the same recurrence serves the
// hotword LSTM of SeACo-Paraformer).
//
// The recurrent product gates[b, 4H] = h[b, :] . W_hh^T is GEMM-shaped and runs on the matrix cores
// (`v_mfma_f32_16x16x4_f32`, exact fp32): one launch per time step, both directions in the same launch, grid = H/4 x ndir x
// ceil(B/64) workgroups of 4 waves. A workgroup owns 4 hidden units = 16 gate rows (MFMA row = unit * 4 + gate, so that the
// four accumulator registers of a lane are the i, f, g, o gates of ONE (unit, utterance) and the cell update is lane-local);
// wave w owns utterances 16 w .. 16 w + 15 of the 64-utterance tile (MFMA columns). Layouts:
//   * input projections `pre` are GATES-MAJOR: row n = dir * 4H + gate * H + unit (torch's i, f, g, o order), column
//     t * B + b -- produced by the fp32 MFMA GEMM as W_ih . X_tm^T, X_tm being the time-major [T, B, D] input;
//   * W_hh stays in torch's [dir][4H][H] layout: the 16 rows of a workgroup are four 4-row blocks, staged once per step into
//     LDS as [16 rows][H + 4] (the pad spreads the rows over the banks); a lane reads its A fragments as float4s;
//   * the k index of MFMA step s in lane group q = lane / 16 is q * H/4 + s (any bijection works as long as A and B agree),
//     which makes a lane's A operands contiguous in k;
//   * the state h is stored in B-FRAGMENT ORDER, [dir][b / 16][s / 4][lane = q * 16 + b % 16][s % 4]: a wave loads its whole
//     B operand (H x 16 utterances) with H/16 fully coalesced 16-byte loads per lane straight into registers, no LDS; the new
//     state is scattered back as single floats (64 per wave). c is unit-major [dir][H][Bs].
// Every load of a step is issued before any arithmetic (one L2 round trip per step). No inter-workgroup synchronisation
// inside a kernel: the step order is the stream order. Work per step: B * 4H * H FMAs per direction (268 MFLOP at B = 64,
// H = 512 = 1.7 us at the fp32 matrix rate).
#include "common.h"
#include "lstm.h"

namespace pf {

namespace {

constexpr int LSTM_RS_PAD = 4;                // W slice in LDS: 16 rows x (H + 4) floats (33 KB at H = 512)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// float index of h[k][b] in fragment order (see above)
__device__ __forceinline__ size_t lstm_state_index(int dir, int H, int Bs, int k, int b) {
    const int Q = H >> 2;
    const int q = k / Q, s = k - q * Q;
    const int lane = q * 16 + (b & 15);
    return ((((size_t)dir * (Bs >> 4) + (b >> 4)) * (H >> 4) + (s >> 2)) * 64 + lane) * 4 + (s & 3);
}

// NM = H / 16: float4 registers per lane holding the wave's B operand = MFMA quads per step
template <int NM>
__global__ __launch_bounds__(256) void lstm_step_kernel(LstmStepArgs p) {
    constexpr int H = NM * 16, RS = H + LSTM_RS_PAD, Q4 = H / 4;   // Q4: float4s per W row = k values per lane group
    constexpr int WREG = (16 * Q4 + 255) / 256;
    __shared__ float s_w[16 * RS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const int dir = blockIdx.y;
    const int u0 = blockIdx.x * 4;
    const int t = dir == 0 ? p.step : p.T - 1 - p.step;
    const int btile = blockIdx.z * 4 + wave;                     // 16-utterance tile of this wave
    const int ul = lane >> 4, j = lane & 15;
    const int u = u0 + ul, b = btile * 16 + j;
    const bool valid = b < p.B;
    // ---- all loads of the step
    float4 wreg[WREG], hreg[NM];
#pragma unroll
    for (int i = 0; i < WREG; ++i) {
        const int f = tid + 256 * i;                              // float4 f of the [16][H] slice, rows in (gate, unit) order
        const int r = f / Q4, c4 = f - r * Q4;
        wreg[i] = r < 16 ? *reinterpret_cast<const float4*>(p.whh + ((size_t)dir * 4 * H + (size_t)(r >> 2) * H + u0 + (r & 3)) * H + 4 * c4)
                         : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4* hsrc = reinterpret_cast<const float4*>(p.h_prev) + (((size_t)dir * (p.Bs >> 4) + btile) * NM) * 64 + lane;
#pragma unroll
    for (int m = 0; m < NM; ++m) hreg[m] = hsrc[(size_t)m * 64];
    f32x4 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const size_t n = (size_t)dir * 4 * H + (size_t)g * H + u;
        const float x = valid ? p.pre[n * p.ld_pre + (size_t)t * p.B + b] : 0.f;
        acc[g] = x + (p.b_ih[n] + p.b_hh[n]);
    }
    const size_t ci = ((size_t)dir * H + u) * p.Bs + b;
    const float c_prev = p.c[ci];
    // ---- W slice to LDS in MFMA row order (row = unit * 4 + gate)
#pragma unroll
    for (int i = 0; i < WREG; ++i) {
        const int f = tid + 256 * i;
        const int r = f / Q4, c4 = f - r * Q4;
        if (r < 16) *reinterpret_cast<float4*>(s_w + ((r & 3) * 4 + (r >> 2)) * RS + 4 * c4) = wreg[i];
    }
    __syncthreads();
    // ---- gates += W . h: lane (row = lane & 15, q = lane >> 4) feeds A[row][q * H/4 + s], B comes from hreg
    const float* wrow = s_w + (lane & 15) * RS + (lane >> 4) * Q4;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const float4 a = *reinterpret_cast<const float4*>(wrow + 4 * m);
        const float4 h4 = hreg[m];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, h4.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, h4.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, h4.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, h4.w, acc, 0, 0, 0);
    }
    // ---- cell update: acc[0..3] = i, f, g, o of (unit u, utterance b)
    const float ig = sigmoidf_(acc[0]), fg = sigmoidf_(acc[1]), gg = tanhf(acc[2]), og = sigmoidf_(acc[3]);
    const float cn = fg * c_prev + ig * gg;
    const float hn = og * tanhf(cn);
    p.c[ci] = cn;
    p.h_next[lstm_state_index(dir, H, p.Bs, u, b)] = hn;
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
    PF_REQUIRE(a.H > 0 && a.H % 16 == 0 && a.B > 0 && a.T > 0 && a.Bs % 64 == 0 && a.Bs >= a.B && (a.ndir == 1 || a.ndir == 2),
               "lstm: need H % 16 == 0, ndir 1 or 2, state stride a multiple of 64");
    void (*kernel)(LstmStepArgs) = nullptr;
    switch (a.H / 16) {                     // the hidden size is a compile-time constant of the step kernel
        case 1: kernel = lstm_step_kernel<1>; break;
        case 2: kernel = lstm_step_kernel<2>; break;
        case 3: kernel = lstm_step_kernel<3>; break;
        case 4: kernel = lstm_step_kernel<4>; break;
        case 6: kernel = lstm_step_kernel<6>; break;
        case 8: kernel = lstm_step_kernel<8>; break;
        case 16: kernel = lstm_step_kernel<16>; break;
        case 20: kernel = lstm_step_kernel<20>; break;
        case 32: kernel = lstm_step_kernel<32>; break;
        default: break;
    }
    PF_REQUIRE(kernel != nullptr, "lstm: hidden size must be one of 16, 32, 48, 64, 96, 128, 256, 320, 512");
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
        hipLaunchKernelGGL(kernel, grid, block, 0, stream, s);
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
