// Kaldi-compatible log-mel filterbank + LFR stacking + CMVN on gfx950.
//
// Replaces the per-utterance Python loop of WavFrontend.forward (funasr/frontends/wav_frontend.py:149-196):
//   wave * 32768 -> kaldi.fbank(80 mel, 25 ms / 10 ms, hamming, dither 0, snip_edges) -> apply_lfr(7, 6)
//   (wav_frontend.py:63-86) -> apply_cmvn (wav_frontend.py:46-60) -> zero padded batch (:195).
// Kaldi semantics follow the vendored restatement
//   runtime/onnxruntime/third_party/kaldi-native-fbank/kaldi-native-fbank/csrc/feature-window.cc:186-244
//   (DC removal, pre-emphasis, window), feature-fbank.cc:75-106 (power spectrum, mel, log floor).
//
// HBM-bound byte work: 1.92 MB of PCM in, 1.12 MB of features out per 30 s clip. One wave owns one frame: the
// 400 samples are read once (coalesced dwords), the 512-point FFT runs out of LDS (radix-2, 9 stages, 4
// butterflies per lane per stage), the mel projection walks the sparse triangles and the 80 log energies go
// out as coalesced dwords. A second tiny kernel gathers 7 frames per LFR row and applies CMVN with float4
// stores (pure gather, rows past the utterance are written as zeros so the batch is exactly what pad_sequence
// produces).
#include "common.h"
#include "frontend.h"

namespace pf {

namespace {

constexpr int NFFT = 512;
constexpr int NBIN = NFFT / 2 + 1;   // 257
constexpr int FRAMES_PER_BLOCK = 4;
constexpr int MAX_NNZ = 2048;        // non-zero mel weights (80 triangles over 256 bins: ~510)

__device__ __forceinline__ int bitrev9(int i) { return (int)(__brev((unsigned)i) >> 23); }

// Every wave works on its own LDS slice, and the DS operations of one wave are executed in program order: a write
// followed by a read of another lane's element needs no workgroup barrier, only that the compiler keeps the order
// (it must: the addresses may alias) -- 15 s_barriers per frame across 4 unrelated waves were pure stall.
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }

__global__ __launch_bounds__(256) void fbank_kernel(FbankArgs p) {
    // per wave: re[512], im[512]  (pw[257] aliases re after the FFT, raw samples alias im before it)
    __shared__ float lds[FRAMES_PER_BLOCK][2 * NFFT];
    // twiddles and the compacted mel triangles are read by every frame: staged once per workgroup (one barrier),
    // so that the 36 twiddle fetches and the ~40-step triangle walk of a lane hit LDS instead of dependent global loads
    __shared__ float2 tw_s[NFFT / 2];
    __shared__ float cw_s[MAX_NNZ];
    for (int i = threadIdx.x; i < NFFT / 2; i += 256) tw_s[i] = p.twiddle[i];
    for (int i = threadIdx.x; i < p.mel_nnz; i += 256) cw_s[i] = p.mel_compact[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int nfr = p.n_frames[b];
    int f = blockIdx.x * FRAMES_PER_BLOCK + wave;
    const bool valid = f < nfr;
    if (!valid) f = nfr > 0 ? nfr - 1 : 0;
    float* re = lds[wave];
    float* im = lds[wave] + NFFT;
    const bool any = nfr > 0;

    // ---- load + scale, DC removal (feature-window.cc:186-196)
    const float* w = p.wav + (size_t)b * p.wav_stride + (size_t)f * p.frame_shift;
    float x[7];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int i = lane + 64 * j;
        x[j] = (any && i < p.frame_len) ? w[i] * p.in_scale : 0.f;
        s += x[j];
    }
    s = wave_sum(s);
    const float mean = s / (float)p.frame_len;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int i = lane + 64 * j;
        if (i < p.frame_len) im[i] = x[j] - mean;
    }
    wave_sync();
    // ---- pre-emphasis (feature-window.cc:204-215) + window, stored bit-reversed for the DIT FFT
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = lane + 64 * j;
        float v = 0.f;
        if (i < p.frame_len) {
            const float cur = im[i];
            const float prev = im[i > 0 ? i - 1 : 0];
            v = __fsub_rn(cur, __fmul_rn(p.preemph, prev)) * p.window[i];
        }
        re[bitrev9(i)] = v;
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 8; ++j) im[lane + 64 * j] = 0.f;
    wave_sync();

    // ---- 512-point radix-2 decimation-in-time FFT, twiddle table tw[k] = exp(-2 pi i k / 512)
#pragma unroll 1
    for (int st = 1; st <= 9; ++st) {
        const int half = 1 << (st - 1);
        const int tstep = NFFT >> st;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = lane + 64 * j;            // butterfly id 0..255
            const int grp = k >> (st - 1);
            const int pos = k & (half - 1);
            const int i0 = (grp << st) + pos;
            const int i1 = i0 + half;
            const float2 t = tw_s[pos * tstep];
            const float xr = re[i1], xi = im[i1];
            const float tr = xr * t.x - xi * t.y;
            const float ti = xr * t.y + xi * t.x;
            const float ar = re[i0], ai = im[i0];
            re[i0] = ar + tr; im[i0] = ai + ti;
            re[i1] = ar - tr; im[i1] = ai - ti;
        }
        wave_sync();
    }

    // ---- power spectrum: |X|^2 computed as abs() then square like torchaudio's spectrum.abs().pow(2)
    float pw[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int k = lane + 64 * j;
        if (k < NBIN) {
            const float a = re[k], c = im[k];
            const float mag = sqrtf(a * a + c * c);
            pw[j] = mag * mag;
        }
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int k = lane + 64 * j;
        if (k < NBIN) re[k] = pw[j];
    }
    wave_sync();

    // ---- mel projection over the sparse triangles + log floor (feature-fbank.cc:95-106)
    float* out = p.fbank + ((size_t)b * p.max_frames + f) * p.n_mels;
    for (int m = lane; m < p.n_mels; m += 64) {
        const int off = p.mel_offset[m], len = p.mel_len[m];
        const float* mw = cw_s + p.mel_coff[m];
        float e = 0.f;
#pragma unroll 4
        for (int k = 0; k < len; ++k) e = fmaf(mw[k], re[off + k], e);
        e = fmaxf(e, 1.1920928955078125e-07f);
        if (valid) out[m] = logf(e);
    }
}

__global__ __launch_bounds__(256) void lfr_cmvn_kernel(LfrArgs p) {
    // one thread per float4 of the output row: D = n_mels * lfr_m (560 -> 140 float4)
    const int D = p.n_mels * p.lfr_m;
    const int D4 = D >> 2;
    const int b = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)p.T_out * D4) return;
    const int t = (int)(i / D4), c = (int)(i % D4) * 4;
    const int nfr = p.n_frames[b];
    const int T = p.rows_override > 0 ? p.rows_override : (nfr + p.lfr_n - 1) / p.lfr_n;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < T) {
        const int j = c / p.n_mels, m = c % p.n_mels;      // n_mels % 4 == 0: a float4 never straddles frames
        int src = p.lfr_n * t + j - p.left;
        src = src < 0 ? 0 : (src > nfr - 1 ? nfr - 1 : src);
        const float4 v = *reinterpret_cast<const float4*>(p.fbank + ((size_t)b * p.max_frames + src) * p.n_mels + m);
        if (p.cmvn_shift) {
            const float4 sh = *reinterpret_cast<const float4*>(p.cmvn_shift + c);
            const float4 sc = *reinterpret_cast<const float4*>(p.cmvn_scale + c);
            o.x = __fmul_rn(__fadd_rn(v.x, sh.x), sc.x);
            o.y = __fmul_rn(__fadd_rn(v.y, sh.y), sc.y);
            o.z = __fmul_rn(__fadd_rn(v.z, sh.z), sc.z);
            o.w = __fmul_rn(__fadd_rn(v.w, sh.w), sc.w);
        } else {
            o = v;
        }
    }
    *reinterpret_cast<float4*>(p.out + ((size_t)b * p.T_out + t) * D + c) = o;
}

}  // namespace

int launch_fbank(const FbankArgs& a, int B, int max_frames_in_batch, hipStream_t stream) {
    PF_REQUIRE(a.frame_len <= 448 && a.frame_len > 0, "fbank: frame length must be <= 448 samples");
    PF_REQUIRE(a.n_mels <= 128, "fbank: n_mels <= 128");
    PF_REQUIRE(a.mel_nnz <= MAX_NNZ && a.mel_compact && a.mel_coff, "fbank: mel filterbank too dense for the LDS copy");
    if (max_frames_in_batch <= 0) return 0;
    dim3 grid(ceil_div(max_frames_in_batch, FRAMES_PER_BLOCK), B);
    hipLaunchKernelGGL(fbank_kernel, grid, dim3(256), 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

int launch_lfr_cmvn(const LfrArgs& a, int B, hipStream_t stream) {
    PF_REQUIRE(a.n_mels % 4 == 0, "lfr: n_mels must be a multiple of 4");
    if (a.T_out <= 0) return 0;
    const size_t per_seq = (size_t)a.T_out * (a.n_mels * a.lfr_m / 4);
    dim3 grid((unsigned)((per_seq + 255) / 256), B);
    hipLaunchKernelGGL(lfr_cmvn_kernel, grid, dim3(256), 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pf
