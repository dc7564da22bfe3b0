// CIF predictor tail (continuous integrate-and-fire) and the greedy arg-max of the Paraformer decoder output.
//
// Reference semantics:
//   CifPredictorV2.forward      funasr/models/paraformer/cif_predictor.py:253-314
//   tail_process_fn             funasr/models/paraformer/cif_predictor.py:414-446
//   cif_wo_hidden_v1 / cif_v1   funasr/models/paraformer/cif_predictor.py:818-908
//   greedy arg-max              funasr/models/paraformer/model.py:642
//
// The integer results (which frames fire, how many tokens) must equal the CPU reference, so the scan is done
// exactly as the reference does it: a sequential-order float64 prefix sum of the float32 alphas, rounded to
// float32, then floor-differences (cif_predictor.py:835-846). One lane walks one utterance (T <= ~1000 adds);
// the work is tiny but serial by definition. The weighted frame sums use the same prefix-difference
// formulation as cif_v1 (:877-896) with one thread per (utterance, channel): coalesced 4-B loads across
// channels, a running float64 accumulator rounded to float32 per step (what ATen's CPU cumsum does for
// float32 input), emitted at fire positions.
#include "common.h"
#include "cif.h"

namespace pf {

namespace {

__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ h, float* __restrict__ out, int B,
                                                     int T, int D4, int l_order, int taps) {
    // one thread per float4 of the output row [taps * D]
    const size_t per_row = (size_t)taps * D4;
    const size_t total = (size_t)B * T * per_row;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < total; i += stride) {
        const size_t row = i / per_row;
        const int r = (int)(i % per_row);
        const int tap = r / D4, c4 = r % D4;
        const int b = (int)(row / T), t = (int)(row % T);
        const int ts = t + tap - l_order;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ts >= 0 && ts < T) v = reinterpret_cast<const float4*>(h)[((size_t)b * T + ts) * D4 + c4];
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

// one wave per row: alpha = relu(sigmoid(dot(conv_row, w) + b) * smooth - noise) * mask
__global__ __launch_bounds__(256) void alpha_kernel(AlphaArgs p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.B * p.T) return;
    const int b = row / p.T, t = row % p.T;
    const float* x = p.conv + (size_t)row * p.D;
    float s = 0.f;
    for (int c = lane * 4; c < p.D; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + c);
        const float4 w = *reinterpret_cast<const float4*>(p.w + c);
        s = fmaf(v.x, w.x, s);
        s = fmaf(v.y, w.y, s);
        s = fmaf(v.z, w.z, s);
        s = fmaf(v.w, w.w, s);
    }
    s = wave_sum(s);
    if (lane == 0) {
        const float z = s + p.bias[0];
        float a = 1.0f / (1.0f + expf(-z));
        a = fmaxf(__fsub_rn(__fmul_rn(a, p.smooth), p.noise), 0.f);
        if (t >= p.lens[b]) a = 0.f;
        p.alphas[(size_t)b * p.T_ext + t] = a;
    }
}

// One wave per utterance: the exact reference scan. The recurrence is serial by definition (a sequential-order float64
// prefix sum rounded to float32 per step, cif_predictor.py:835-846), so it runs on lane 0 -- but out of LDS: the wave
// first stages the alphas with coalesced loads and afterwards writes peaks / remainders / fire flags back coalesced,
// instead of one dependent global round trip per frame (0.2 ms -> ~10 us at T = 500).
constexpr int CIF_MAX_T = 4096;     // frames per utterance that fit the LDS staging (4 arrays x 16 KB)
__global__ __launch_bounds__(64) void cif_scan_kernel(CifScanArgs p) {
    __shared__ float s_al[CIF_MAX_T], s_pk[CIF_MAX_T], s_rm[CIF_MAX_T];
    __shared__ int s_ff[CIF_MAX_T];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int Te = p.T + 1;
    float* al = p.alphas + (size_t)b * Te;
    const int len = p.lens[b];
    for (int t = lane; t < Te; t += 64) {
        float a = t < p.T ? al[t] : 0.f;
        if (p.tail_threshold > 0.f) {
            // tail_process_fn: with tail_mask the threshold lands on index len (mask_2 - mask_1), else on index T
            const bool hit = p.tail_mask ? (t == len) : (t == p.T);
            if (hit) a = __fadd_rn(a, p.tail_threshold);
        }
        s_al[t] = a;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        double cs = 0.0;
        float prev_floor = 0.f;
        int n = 0;
#pragma unroll 4
        for (int t = 0; t < Te; ++t) {
            cs += (double)s_al[t];
            const float ps = (float)cs;
            const float fl = floorf(ps);
            const bool fire = (fl - prev_floor) > 0.f;
            const float fires = __fsub_rn(__fadd_rn(fire ? 1.f : 0.f, ps), fl);
            s_pk[t] = fires;
            s_rm[t] = __fsub_rn(fires, floorf(fires));
            s_ff[t] = fire ? 1 : 0;
            n += fire ? 1 : 0;
            prev_floor = fl;
        }
        p.n_fires[b] = n;
    }
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < Te; t += 64) {
        al[t] = s_al[t];
        p.peaks[(size_t)b * Te + t] = s_pk[t];
        p.rems[(size_t)b * Te + t] = s_rm[t];
        p.fire_flag[(size_t)b * Te + t] = s_ff[t];
    }
}

// fallback for very long utterances (T >= CIF_MAX_T): one lane per utterance straight from HBM
__global__ __launch_bounds__(64) void cif_scan_long_kernel(CifScanArgs p) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= p.B) return;
    const int Te = p.T + 1;
    float* al = p.alphas + (size_t)b * Te;
    float* pk = p.peaks + (size_t)b * Te;
    float* rm = p.rems + (size_t)b * Te;
    int* ff = p.fire_flag + (size_t)b * Te;
    const int len = p.lens[b];
    double cs = 0.0;
    float prev_floor = 0.f;
    int n = 0;
    for (int t = 0; t < Te; ++t) {
        float a = t < p.T ? al[t] : 0.f;
        if (p.tail_threshold > 0.f) {
            const bool hit = p.tail_mask ? (t == len) : (t == p.T);
            if (hit) a = __fadd_rn(a, p.tail_threshold);
        }
        al[t] = a;
        cs += (double)a;
        const float ps = (float)cs;
        const float fl = floorf(ps);
        const bool fire = (fl - prev_floor) > 0.f;
        const float fires = __fsub_rn(__fadd_rn(fire ? 1.f : 0.f, ps), fl);
        pk[t] = fires;
        rm[t] = __fsub_rn(fires, floorf(fires));
        ff[t] = fire ? 1 : 0;
        n += fire ? 1 : 0;
        prev_floor = fl;
    }
    p.n_fires[b] = n;
}

// CifPredictorV3: the reference integrates in a Python loop over frames, all in fp32 (bicif_paraformer/cif_predictor.py:39-86)
__global__ __launch_bounds__(64) void cif_scan_loop_kernel(CifScanArgs p, float* __restrict__ curs, int* __restrict__ n_tok) {
    __shared__ float s_al[CIF_MAX_T], s_pk[CIF_MAX_T], s_rm[CIF_MAX_T];
    __shared__ int s_ff[CIF_MAX_T];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int Te = p.T + 1;
    float* al = p.alphas + (size_t)b * Te;
    const int len = p.lens[b];
    for (int t = lane; t < Te; t += 64) {
        float a = t < p.T ? al[t] : 0.f;
        if (p.tail_threshold > 0.f) {
            const bool hit = p.tail_mask ? (t == len) : (t == p.T);
            if (hit) a = __fadd_rn(a, p.tail_threshold);
        }
        s_al[t] = a;
        al[t] = a;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        float integrate = 0.f;
        double total = 0.0;
        int n = 0;
#pragma unroll 4
        for (int t = 0; t < Te; ++t) {
            const float a = s_al[t];
            total += (double)a;
            const float completion = __fsub_rn(1.0f, integrate);
            integrate = __fadd_rn(integrate, a);
            s_pk[t] = integrate;
            const bool fire = integrate >= 1.0f;
            if (fire) integrate = __fsub_rn(integrate, 1.0f);
            const float cur = fire ? completion : a;
            s_al[t] = cur;
            s_rm[t] = __fsub_rn(a, cur);
            s_ff[t] = fire ? 1 : 0;
            n += fire ? 1 : 0;
        }
        p.n_fires[b] = n;
        n_tok[b] = (int)floorf((float)total);
    }
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < Te; t += 64) {
        curs[(size_t)b * Te + t] = s_al[t];
        p.peaks[(size_t)b * Te + t] = s_pk[t];
        p.rems[(size_t)b * Te + t] = s_rm[t];
        p.fire_flag[(size_t)b * Te + t] = s_ff[t];
    }
}

__global__ __launch_bounds__(256) void cif_emit_loop_kernel(CifEmitArgs p) {
    __shared__ float s_cur[CIF_MAX_T], s_rm[CIF_MAX_T];
    __shared__ int s_ff[CIF_MAX_T];
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const int Te = p.T + 1;
    for (int t = threadIdx.x; t < Te; t += 256) {
        s_cur[t] = p.alphas[(size_t)b * Te + t];
        s_rm[t] = p.rems[(size_t)b * Te + t];
        s_ff[t] = p.fire_flag[(size_t)b * Te + t];
    }
    __syncthreads();
    if (c >= p.D) return;
    const float* h = p.hidden + (size_t)b * p.T * p.D + c;
    float* out = p.embeds + (size_t)b * p.N * p.D + c;
    float frame = 0.f;
    int k = 0;
    for (int t0 = 0; t0 < Te; t0 += 8) {
        float hv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = t0 + j;
            hv[j] = t < p.T ? h[(size_t)t * p.D] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = t0 + j;
            if (t >= Te) break;
            frame = __fadd_rn(frame, __fmul_rn(s_cur[t], hv[j]));
            if (s_ff[t]) {
                if (k < p.N) out[(size_t)k * p.D] = frame;
                ++k;
                frame = __fmul_rn(s_rm[t], hv[j]);
            }
        }
    }
    for (; k < p.N; ++k) out[(size_t)k * p.D] = 0.f;
}

// one thread per (utterance, channel). The per-frame scalars (alpha, remainder, fire flag) are staged in LDS once per
// workgroup; the channel loads of 8 consecutive frames are issued together (they do not depend on the running sum), so
// the serial float64 accumulation no longer waits on one HBM round trip per frame.
__global__ __launch_bounds__(256) void cif_emit_kernel(CifEmitArgs p) {
    __shared__ float s_al[CIF_MAX_T], s_rm[CIF_MAX_T];
    __shared__ int s_ff[CIF_MAX_T];
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const int Te = p.T + 1;
    const bool staged = Te <= CIF_MAX_T;
    const float* al = p.alphas + (size_t)b * Te;
    const float* rm = p.rems + (size_t)b * Te;
    const int* ff = p.fire_flag + (size_t)b * Te;
    if (staged) {
        for (int t = threadIdx.x; t < Te; t += 256) {
            s_al[t] = al[t];
            s_rm[t] = rm[t];
            s_ff[t] = ff[t];
        }
        __syncthreads();
    }
    if (c >= p.D) return;
    const float* h = p.hidden + (size_t)b * p.T * p.D + c;
    float* out = p.embeds + (size_t)b * p.N * p.D + c;
    double acc = 0.0;
    float prevP = 0.f, prev_remh = 0.f;
    int k = 0;
    for (int t0 = 0; t0 < Te; t0 += 8) {
        float hv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = t0 + j;
            hv[j] = t < p.T ? h[(size_t)t * p.D] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = t0 + j;
            if (t >= Te) break;
            const float a = staged ? s_al[t] : al[t];
            const float prod = __fmul_rn(a, hv[j]);
            acc += (double)prod;
            const float P = (float)acc;
            if (staged ? s_ff[t] : ff[t]) {
                const float remh = __fmul_rn(staged ? s_rm[t] : rm[t], hv[j]);
                // frames - shift_frames + shift_remain_frames - remain_frames, left to right (cif_predictor.py:896)
                const float v = __fsub_rn(__fadd_rn(__fsub_rn(P, prevP), prev_remh), remh);
                if (k < p.N) out[(size_t)k * p.D] = v;
                prevP = P;
                prev_remh = remh;
                ++k;
            }
        }
    }
    for (; k < p.N; ++k) out[(size_t)k * p.D] = 0.f;
}

__global__ __launch_bounds__(256) void argmax_reduce_kernel(const float* __restrict__ pval,
                                                            const int* __restrict__ pidx, int ld, int nparts,
                                                            int* __restrict__ ids, float* __restrict__ best, int M) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = lane; j < nparts; j += 64) {
        const float v = pval[(size_t)row * ld + j];
        const int i = pidx[(size_t)row * ld + j];
        if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) {
        ids[row] = bi;
        if (best) best[row] = bv;
    }
}

// one workgroup per row: a thread walks columns tid, tid + 256, .. (eight loads in flight: with one wave per row the 131
// dependent trips of an 8404-column row cost 37 us in the streaming step), first maximum wins at every merge
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int ldx, int M, int N,
                                                          int* __restrict__ ids) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x;
    const float* xr = x + (size_t)row * ldx;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    int j = threadIdx.x;
    for (; j + 7 * 256 < N; j += 8 * 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xr[j + 256 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (v[u] > bv) { bv = v[u]; bi = j + 256 * u; }     // ascending columns per thread
    }
    for (; j < N; j += 256) {
        const float v = xr[j];
        if (v > bv) { bv = v; bi = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        ids[row] = bi;
    }
}

}  // namespace

int launch_im2col(const float* hidden, float* out, int B, int T, int D, int l_order, int r_order,
                  hipStream_t stream) {
    PF_REQUIRE(D % 4 == 0, "im2col: D % 4");
    const int taps = l_order + r_order + 1;
    const size_t total = (size_t)B * T * taps * (D / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(im2col_kernel, dim3(blocks), dim3(256), 0, stream, hidden, out, B, T, D / 4, l_order, taps);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_alpha(const AlphaArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.D % 4 == 0, "alpha: D % 4");
    hipLaunchKernelGGL(alpha_kernel, dim3(ceil_div(a.B * a.T, 4)), dim3(256), 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_cif_scan(const CifScanArgs& a, hipStream_t stream) {
    if (a.T + 1 <= CIF_MAX_T) hipLaunchKernelGGL(cif_scan_kernel, dim3(a.B), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL(cif_scan_long_kernel, dim3(ceil_div(a.B, 64)), dim3(64), 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_cif_scan_loop(const CifScanArgs& a, float* curs, int* n_tok, hipStream_t stream) {
    PF_REQUIRE(a.alphas && a.peaks && a.rems && a.fire_flag && a.n_fires && a.lens && curs && n_tok && a.B > 0 && a.T > 0,
               "cif_scan_loop: null/empty argument");
    PF_REQUIRE(a.T + 1 <= CIF_MAX_T, "cif_scan_loop: at most 4095 encoder frames per utterance");
    hipLaunchKernelGGL(cif_scan_loop_kernel, dim3((unsigned)a.B), dim3(64), 0, stream, a, curs, n_tok);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_cif_emit_loop(const CifEmitArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.hidden && a.alphas && a.rems && a.fire_flag && a.embeds && a.B > 0 && a.T > 0 && a.D > 0 && a.N > 0,
               "cif_emit_loop: null/empty argument");
    PF_REQUIRE(a.T + 1 <= CIF_MAX_T, "cif_emit_loop: at most 4095 encoder frames per utterance");
    hipLaunchKernelGGL(cif_emit_loop_kernel, dim3((unsigned)ceil_div(a.D, 256), (unsigned)a.B), dim3(256), 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_cif_emit(const CifEmitArgs& a, hipStream_t stream) {
    if (a.N <= 0) return 0;
    hipLaunchKernelGGL(cif_emit_kernel, dim3(ceil_div(a.D, 256), a.B), dim3(256), 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_argmax_reduce(const float* pval, const int* pidx, int ld, int nparts, int* ids, float* best, int M,
                         hipStream_t stream) {
    hipLaunchKernelGGL(argmax_reduce_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, stream, pval, pidx, ld, nparts,
                       ids, best, M);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_argmax_rows(const float* x, int ldx, int M, int N, int* ids, hipStream_t stream) {
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(M), dim3(256), 0, stream, x, ldx, M, N, ids);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pf
