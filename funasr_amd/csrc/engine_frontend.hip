// C-ABI layer, frontend family: pf_frontend_* (WavFrontend / WavFrontendOnline, funasr/frontends/wav_frontend.py).
#include <atomic>
#include "engine_internal.h"

namespace pf {

static float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }

int frontend_upload_tables(Frontend* f, const std::vector<float>& window, const std::vector<float>& mel) {
    // the mel triangles (dense [n_mels, 257]) are cut into pieces of <= 8 consecutive fft bins: a lane of the fbank
    // kernel owns <= 2 pieces (weights in registers), a mel bin is the fixed-order sum of its pieces
    const int nm = f->cfg.n_mels, NB = 257;
    std::vector<float> pw;
    std::vector<int> pk0, first(nm), count(nm);
    for (int m = 0; m < nm; ++m) {
        int lo = -1, hi = -1;
        for (int k = 0; k < NB; ++k)
            if (mel[(size_t)m * NB + k] != 0.f) { if (lo < 0) lo = k; hi = k; }
        first[m] = (int)pk0.size();
        if (lo >= 0) {
            for (int k0 = lo; k0 <= hi; k0 += 8) {
                pk0.push_back(k0);
                for (int t = 0; t < 8; ++t) pw.push_back((k0 + t <= hi) ? mel[(size_t)m * NB + k0 + t] : 0.f);
            }
        }
        count[m] = (int)pk0.size() - first[m];
    }
    if (pk0.empty()) { pk0.push_back(0); pw.resize(8, 0.f); }
    if (pk0.size() > 128) { set_error("frontend: mel filterbank needs more than 128 eight-bin pieces"); return -1; }
    f->n_pieces = (int)pk0.size();
    if (f->window.ensure(sizeof(float) * window.size()) || f->piece_w.ensure(sizeof(float) * pw.size()) ||
        f->piece_k0.ensure(sizeof(int) * pk0.size()) || f->mel_first.ensure(sizeof(int) * nm) ||
        f->mel_count.ensure(sizeof(int) * nm))
        return -2;
    PF_HIP_TRY(hipMemcpy(f->window.p, window.data(), sizeof(float) * window.size(), hipMemcpyHostToDevice));
    PF_HIP_TRY(hipMemcpy(f->piece_w.p, pw.data(), sizeof(float) * pw.size(), hipMemcpyHostToDevice));
    PF_HIP_TRY(hipMemcpy(f->piece_k0.p, pk0.data(), sizeof(int) * pk0.size(), hipMemcpyHostToDevice));
    PF_HIP_TRY(hipMemcpy(f->mel_first.p, first.data(), sizeof(int) * nm, hipMemcpyHostToDevice));
    PF_HIP_TRY(hipMemcpy(f->mel_count.p, count.data(), sizeof(int) * nm, hipMemcpyHostToDevice));
    return 0;
}

// Kaldi tables as in kaldi-native-fbank: window coefficients in float64 (feature-window.cc:25-47), mel
// triangles in float32 over fft bins 0..255 (mel-computations.cc:118-210, strict inequalities at :186)
int frontend_default_tables(Frontend* f) {
    const pf_frontend_config& c = f->cfg;
    std::vector<float> window(c.frame_length);
    const double a = 6.283185307179586476925286766559005 / (c.frame_length - 1);
    for (int i = 0; i < c.frame_length; ++i) window[i] = (float)(0.54 - 0.46 * cos(a * (double)i));
    const int NB = 257, nfft = 512;
    std::vector<float> mel((size_t)c.n_mels * NB, 0.f);
    const float nyquist = 0.5f * c.sample_rate;
    const float high = c.high_freq > 0.f ? c.high_freq : nyquist + c.high_freq;
    const float fft_bin_width = (float)c.sample_rate / nfft;
    const float mlow = mel_scale(c.low_freq), mhigh = mel_scale(high);
    const float delta = (mhigh - mlow) / (c.n_mels + 1);
    for (int m = 0; m < c.n_mels; ++m) {
        const float left = mlow + m * delta, center = mlow + (m + 1) * delta, right = mlow + (m + 2) * delta;
        for (int k = 0; k < nfft / 2; ++k) {
            const float mel_k = mel_scale(fft_bin_width * k);
            if (mel_k > left && mel_k < right) {
                mel[(size_t)m * NB + k] =
                    mel_k <= center ? (mel_k - left) / (center - left) : (right - mel_k) / (right - center);
            }
        }
    }
    return frontend_upload_tables(f, window, mel);
}

}  // namespace pf

using namespace pf;

namespace pf {
// Process-wide fence (round 5): fbank_kernel's cross-check switches itself on for every frontend handle once the process has said
// that other work may share a CU with the frontend -- pf_set_concurrency_guard(1) (funasr_amd.dp.guard_shared_gpu: ranks sharing a
// GPU) or an asynchronous streaming step (pf_stream_step_begin: a second stream's kernels can meet the frontend on the chip). The
// two-stream fault of rounds 3 / 4 is a hardware interaction of packed-fp32 VALU instructions with MFMA waves on the same CU
// (tools/micro/pk reproduces it); the library is built without such instructions, and this cross-check stays as the
// second line of defence: the 0.4 % of a step the second evaluation costs is paid wherever kernels of two streams can meet.
static std::atomic<int> g_concurrency_guard{0};
void note_concurrent_streams() { g_concurrency_guard.store(1, std::memory_order_relaxed); }
static int frontend_verify_on(Frontend* f) {
    if (f->verify) return 1;
    if (!g_concurrency_guard.load(std::memory_order_relaxed)) return 0;
    if (!f->faults.p) {
        if (f->faults.ensure(sizeof(unsigned int) * 68)) return 0;
        if (hipMemset(f->faults.p, 0, sizeof(unsigned int) * 68) != hipSuccess) return 0;
    }
    return 1;
}
}  // namespace pf

extern "C" {

// -------------------------------------------------------------------------------------------------- frontend
pf_frontend* pf_frontend_create(const pf_frontend_config* cfg) {
    if (!cfg) { set_error("frontend: null config"); return nullptr; }
    if (check_device()) return nullptr;
    if (cfg->frame_length <= 0 || cfg->frame_length > 512 || cfg->frame_shift <= 0 || cfg->n_mels <= 0 ||
        cfg->n_mels > 128 || cfg->n_mels % 4 || cfg->lfr_m <= 0 || cfg->lfr_n <= 0) {
        set_error("frontend: unsupported config (frame_length <= 512, n_mels % 4 == 0, n_mels <= 128)");
        return nullptr;
    }
    std::unique_ptr<Frontend> f(new Frontend());
    f->cfg = *cfg;
    std::vector<float> tw(1024);                       // exp(-2 pi i k / 512), k = 0 .. 511
    for (int k = 0; k < 512; ++k) {
        const double a = 6.283185307179586476925286766559005 * k / 512.0;
        tw[2 * k] = (float)cos(a);
        tw[2 * k + 1] = (float)(-sin(a));
    }
    if (f->twiddle.ensure(sizeof(float) * 1024)) return nullptr;
    if (hipMemcpy(f->twiddle.p, tw.data(), sizeof(float) * 1024, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("frontend: twiddle upload failed");
        return nullptr;
    }
    if (frontend_default_tables(f.get())) return nullptr;
    return reinterpret_cast<pf_frontend*>(f.release());
}
void pf_frontend_destroy(pf_frontend* f) { delete reinterpret_cast<Frontend*>(f); }

int pf_frontend_set_cmvn(pf_frontend* fh, const float* shift, const float* scale, int32_t n) {
    Frontend* f = reinterpret_cast<Frontend*>(fh);
    PF_REQUIRE(f && shift && scale, "frontend_set_cmvn: null");
    PF_REQUIRE(n == f->feat_dim(), "frontend_set_cmvn: n must equal n_mels * lfr_m");
    if (f->cmvn_shift.ensure(sizeof(float) * n) || f->cmvn_scale.ensure(sizeof(float) * n)) return -2;
    PF_HIP_TRY(hipMemcpy(f->cmvn_shift.p, shift, sizeof(float) * n, hipMemcpyDefault));
    PF_HIP_TRY(hipMemcpy(f->cmvn_scale.p, scale, sizeof(float) * n, hipMemcpyDefault));
    f->has_cmvn = true;
    return 0;
}

int pf_frontend_set_dither(pf_frontend* fh, float dither, uint64_t seed) {
    Frontend* f = reinterpret_cast<Frontend*>(fh);
    PF_REQUIRE(f && dither >= 0.f, "frontend_set_dither: null handle or negative dither");
    f->dither = dither; f->dither_seed = seed; f->dither_calls = 0;
    return 0;
}
int pf_set_concurrency_guard(int32_t on) { g_concurrency_guard.store(on ? 1 : 0, std::memory_order_relaxed); return 0; }
int pf_concurrency_guard(void) { return g_concurrency_guard.load(std::memory_order_relaxed); }

int pf_frontend_set_verify(pf_frontend* fh, int32_t on) {
    Frontend* f = reinterpret_cast<Frontend*>(fh);
    PF_REQUIRE(f, "frontend_set_verify: null handle");
    if (on && !f->faults.p) {
        if (f->faults.ensure(sizeof(unsigned int) * 68)) return -2;          // [0] count, [4 + 4 k ..]: log of the first 16
        PF_HIP_TRY(hipMemset(f->faults.p, 0, sizeof(unsigned int) * 68));
    }
    f->verify = on ? 1 : 0;
    return 0;
}
int pf_frontend_faults(pf_frontend* fh, uint32_t* count_host) {
    Frontend* f = reinterpret_cast<Frontend*>(fh);
    PF_REQUIRE(f && count_host, "frontend_faults: null");
    *count_host = 0;
    if (f->faults.p) PF_HIP_TRY(hipMemcpy(count_host, f->faults.p, sizeof(unsigned int), hipMemcpyDeviceToHost));
    return 0;
}
int pf_frontend_fault_log(pf_frontend* fh, uint32_t* log_host) {
    Frontend* f = reinterpret_cast<Frontend*>(fh);
    PF_REQUIRE(f && log_host, "frontend_fault_log: null");
    for (int i = 0; i < 64; ++i) log_host[i] = 0;
    if (f->faults.p) PF_HIP_TRY(hipMemcpy(log_host, f->faults.as<unsigned int>() + 4, sizeof(unsigned int) * 64, hipMemcpyDeviceToHost));
    return 0;
}
int pf_frontend_set_tables(pf_frontend* fh, const float* window, const float* mel) {
    Frontend* f = reinterpret_cast<Frontend*>(fh);
    PF_REQUIRE(f && window && mel, "frontend_set_tables: null");
    std::vector<float> w(window, window + f->cfg.frame_length);
    std::vector<float> m(mel, mel + (size_t)f->cfg.n_mels * 257);
    return frontend_upload_tables(f, w, m);
}

// The window by name, with kaldi-native-fbank's float64 arithmetic (feature-window.cc:25-55; FrameExtractionOptions.window_type,
// blackman_coeff 0.42 by default). Callers that want torchaudio's float32 tables pass them through pf_frontend_set_tables.
int pf_frontend_set_window(pf_frontend* fh, const char* window_type, float blackman_coeff) {
    Frontend* f = reinterpret_cast<Frontend*>(fh);
    PF_REQUIRE(f && window_type, "frontend_set_window: null");
    const std::string t(window_type);
    const int n = f->cfg.frame_length;
    std::vector<float> w(n);
    const double a = 6.283185307179586476925286766559005 / (n - 1);
    for (int i = 0; i < n; ++i) {
        const double x = (double)i;
        if (t == "hanning") w[i] = (float)(0.5 - 0.5 * cos(a * x));
        else if (t == "sine") w[i] = (float)sin(0.5 * a * x);
        else if (t == "hamming") w[i] = (float)(0.54 - 0.46 * cos(a * x));
        else if (t == "povey") w[i] = (float)pow(0.5 - 0.5 * cos(a * x), 0.85);
        else if (t == "rectangular") w[i] = 1.f;
        else if (t == "blackman") w[i] = (float)(blackman_coeff - 0.5 * cos(a * x) + (0.5 - blackman_coeff) * cos(2 * a * x));
        else { set_error("frontend_set_window: window_type must be hamming | hanning | povey | rectangular | blackman | sine"); return -1; }
    }
    PF_HIP_TRY(hipMemcpy(f->window.p, w.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    return 0;
}
// snip_edges = false: (n + shift / 2) / shift frames centred on multiples of the shift, the waveform mirrored at both ends
// (kaldi FrameExtractionOptions.snip_edges; the reference passes WavFrontend's `snip_edges` through, wav_frontend.py:180)
int pf_frontend_set_snip_edges(pf_frontend* fh, int32_t on) {
    Frontend* f = reinterpret_cast<Frontend*>(fh);
    PF_REQUIRE(f, "frontend_set_snip_edges: null handle");
    f->snip_edges = on ? 1 : 0;
    return 0;
}

int32_t pf_frontend_num_fbank_frames(const pf_frontend* fh, int64_t n) {
    const Frontend* f = reinterpret_cast<const Frontend*>(fh);
    if (!f) return 0;
    if (!f->snip_edges) return (int32_t)((n + f->cfg.frame_shift / 2) / f->cfg.frame_shift);   // feature-window.cc:87-89
    if (n < f->cfg.frame_length) return 0;
    return (int32_t)(1 + (n - f->cfg.frame_length) / f->cfg.frame_shift);   // feature-window.cc:76-86 (snip_edges)
}
int32_t pf_frontend_num_frames(const pf_frontend* fh, int64_t n) {
    const Frontend* f = reinterpret_cast<const Frontend*>(fh);
    if (!f) return 0;
    const int32_t tf = pf_frontend_num_fbank_frames(fh, n);
    return (tf + f->cfg.lfr_n - 1) / f->cfg.lfr_n;                           // wav_frontend.py:73
}

int pf_frontend_forward(pf_frontend* fh, const float* wav, int64_t wav_stride, const int32_t* n_samples, int32_t B,
                        float* feats, int32_t T_out, int32_t* feat_lens, float* fbank_out, void* stream) {
    Frontend* f = reinterpret_cast<Frontend*>(fh);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(f && wav && n_samples && feats && B > 0, "frontend_forward: null argument");
    std::vector<int32_t> nfr(B);
    int max_fr = 0;
    for (int b = 0; b < B; ++b) {
        PF_REQUIRE(n_samples[b] <= wav_stride, "frontend_forward: n_samples exceeds wav_stride");
        nfr[b] = pf_frontend_num_fbank_frames(fh, n_samples[b]);
        PF_REQUIRE(nfr[b] > 0, "frontend_forward: utterance shorter than one analysis window");
        const int t = (nfr[b] + f->cfg.lfr_n - 1) / f->cfg.lfr_n;
        PF_REQUIRE(t <= T_out, "frontend_forward: T_out too small");
        if (feat_lens) feat_lens[b] = t;
        if (nfr[b] > max_fr) max_fr = nfr[b];
    }
    if (f->nfr.ensure(sizeof(int32_t) * B)) return -2;
    if (upload_h2d(f->nfr.p, nfr.data(), sizeof(int32_t) * B, s)) return -2;
    if (!f->snip_edges) {
        if (f->nsamp.ensure(sizeof(int32_t) * B)) return -2;
        if (upload_h2d(f->nsamp.p, n_samples, sizeof(int32_t) * B, s)) return -2;
    }
    float* fb = fbank_out;
    if (!fb) {
        if (f->fbank.ensure(sizeof(float) * (size_t)B * max_fr * f->cfg.n_mels)) return -2;
        fb = f->fbank.as<float>();
    }
    FbankArgs a{};
    a.wav = wav; a.wav_stride = (size_t)wav_stride; a.n_frames = f->nfr.as<int>(); a.fbank = fb; a.max_frames = max_fr;
    a.frame_len = f->cfg.frame_length; a.frame_shift = f->cfg.frame_shift; a.n_mels = f->cfg.n_mels;
    a.in_scale = f->cfg.upscale; a.preemph = f->cfg.preemph; a.window = f->window.as<float>();
    a.twiddle = f->twiddle.as<float2>(); a.piece_w = f->piece_w.as<float>(); a.piece_k0 = f->piece_k0.as<int>();
    a.mel_first = f->mel_first.as<int>(); a.mel_count = f->mel_count.as<int>(); a.n_pieces = f->n_pieces;
    a.dither = f->dither; a.seed = f->dither_seed; a.call = f->dither != 0.f ? f->dither_calls++ : 0;
    a.verify = frontend_verify_on(f); a.faults = a.verify ? f->faults.as<unsigned int>() : nullptr;
    a.n_samples = f->snip_edges ? nullptr : f->nsamp.as<int>();
    a.first_offset = f->cfg.frame_length / 2 - f->cfg.frame_shift / 2;
    int rc;
    {
        double bytes = 0;
        for (int b = 0; b < B; ++b) bytes += 4.0 * n_samples[b] + 4.0 * nfr[b] * f->cfg.n_mels;
        ProfScope ps(PROF_FBANK, bytes, s);
        if ((rc = launch_fbank(a, B, max_fr, s))) return rc;
    }
    LfrArgs l{};
    l.fbank = fb; l.max_frames = max_fr; l.n_frames = f->nfr.as<int>(); l.out = feats; l.T_out = T_out;
    l.n_mels = f->cfg.n_mels; l.lfr_m = f->cfg.lfr_m; l.lfr_n = f->cfg.lfr_n;
    l.left = (f->cfg.lfr_m - 1) / 2; l.rows_override = 0;
    l.cmvn_shift = f->has_cmvn ? f->cmvn_shift.as<float>() : nullptr;
    l.cmvn_scale = f->has_cmvn ? f->cmvn_scale.as<float>() : nullptr;
    return launch_lfr_cmvn(l, B, s);
}

// LFR + CMVN over an explicit frame buffer (WavFrontendOnline.apply_lfr + apply_cmvn, wav_frontend.py:331-380): the
// caller has already put the left context (splice cache) in front, so row i stacks frames [lfr_n*i, lfr_n*i + lfr_m),
// frames past the end repeat the last one (final flush). frames_dev [T, n_mels] -> out_dev [rows, n_mels*lfr_m].
int pf_frontend_lfr_cmvn(pf_frontend* fh, const float* frames_dev, int32_t T, int32_t rows, float* out_dev, void* stream) {
    Frontend* f = reinterpret_cast<Frontend*>(fh);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(f && frames_dev && out_dev && T > 0 && rows >= 0, "frontend_lfr_cmvn: null/empty");
    if (rows == 0) return 0;
    if (f->nfr.ensure(sizeof(int32_t))) return -2;
    int rc;
    if ((rc = launch_fill_int(f->nfr.as<int>(), 1, T, s))) return rc;
    LfrArgs l{};
    l.fbank = frames_dev; l.max_frames = T; l.n_frames = f->nfr.as<int>(); l.out = out_dev; l.T_out = rows;
    l.n_mels = f->cfg.n_mels; l.lfr_m = f->cfg.lfr_m; l.lfr_n = f->cfg.lfr_n; l.left = 0; l.rows_override = rows;
    l.cmvn_shift = f->has_cmvn ? f->cmvn_shift.as<float>() : nullptr;
    l.cmvn_scale = f->has_cmvn ? f->cmvn_scale.as<float>() : nullptr;
    return launch_lfr_cmvn(l, 1, s);
}

// log-mel only: wav_dev [n] -> fbank_dev [T_fb, n_mels] with T_fb = pf_frontend_num_fbank_frames(n)
int pf_frontend_fbank(pf_frontend* fh, const float* wav_dev, int64_t n_samples, float* fbank_dev, void* stream) {
    Frontend* f = reinterpret_cast<Frontend*>(fh);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(f && wav_dev && fbank_dev, "frontend_fbank: null");
    const int nfr = pf_frontend_num_fbank_frames(fh, n_samples);
    if (nfr <= 0) return 0;
    if (f->nfr.ensure(sizeof(int32_t))) return -2;
    int rc;
    if ((rc = launch_fill_int(f->nfr.as<int>(), 1, nfr, s))) return rc;
    FbankArgs a{};
    a.wav = wav_dev; a.wav_stride = (size_t)n_samples; a.n_frames = f->nfr.as<int>(); a.fbank = fbank_dev; a.max_frames = nfr;
    a.frame_len = f->cfg.frame_length; a.frame_shift = f->cfg.frame_shift; a.n_mels = f->cfg.n_mels;
    a.in_scale = f->cfg.upscale; a.preemph = f->cfg.preemph; a.window = f->window.as<float>();
    a.twiddle = f->twiddle.as<float2>(); a.piece_w = f->piece_w.as<float>(); a.piece_k0 = f->piece_k0.as<int>();
    a.mel_first = f->mel_first.as<int>(); a.mel_count = f->mel_count.as<int>(); a.n_pieces = f->n_pieces;
    a.dither = f->dither; a.seed = f->dither_seed; a.call = f->dither != 0.f ? f->dither_calls++ : 0;
    a.verify = frontend_verify_on(f); a.faults = a.verify ? f->faults.as<unsigned int>() : nullptr;
    if (!f->snip_edges) {
        PF_REQUIRE(n_samples < (1ll << 31), "frontend_fbank: waveform too long");
        if (f->nsamp.ensure(sizeof(int32_t))) return -2;
        if ((rc = launch_fill_int(f->nsamp.as<int>(), 1, (int)n_samples, s))) return rc;
        a.n_samples = f->nsamp.as<int>();
        a.first_offset = f->cfg.frame_length / 2 - f->cfg.frame_shift / 2;
    }
    return launch_fbank(a, 1, nfr, s);
}


}  // extern "C"
