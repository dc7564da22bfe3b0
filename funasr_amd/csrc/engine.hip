// C-ABI layer (include/paraformer_hip.h): opaque module handles that own weights + scratch in HBM and schedule
// the gfx950 kernels of this directory on the caller's HIP stream. One handle per reference registry module
// (SANMEncoder, CifPredictorV2, ParaformerSANMDecoder, WavFrontend, CTC); see the header for the mapping.
//
// Memory model: weights are copied once into library-owned HBM (repacked where a kernel wants a different
// layout); activations live in a per-handle workspace that only grows (hipMalloc outside the steady state,
// never torch's caching allocator -- AutoModel calls torch.cuda.empty_cache() after every batch,
// funasr/auto/auto_model.py:846-849). With 288 GB per MI355X nothing is ever recomputed or spilled.
#include <math.h>
#include <string.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/paraformer_hip.h"
#include "cif.h"
#include "common.h"
#include "frontend.h"

namespace pf {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* get_error() { return g_err.c_str(); }

// ------------------------------------------------------------------------------------------------ profiling
// Optional hipEvent instrumentation of the dominant kernels (bench.py roofline line).
struct ProfRec { hipEvent_t a, b; int kind; double work; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_ev_pool;
static hipEvent_t prof_event() {
    if (!g_ev_pool.empty()) { hipEvent_t e = g_ev_pool.back(); g_ev_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
struct ProfScope {
    hipEvent_t a, b; hipStream_t s; int kind; double work; bool on;
    ProfScope(int kind_, double work_, hipStream_t s_) : s(s_), kind(kind_), work(work_), on(g_prof_on) {
        if (on) { a = prof_event(); b = prof_event(); (void)hipEventRecord(a, s); }
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(b, s); g_prof.push_back({a, b, kind, work}); } }
};
enum { PROF_GEMM = 0, PROF_ATTN = 1, PROF_FSMN = 2, PROF_LN = 3, PROF_FBANK = 4, PROF_KINDS = 5 };

// ------------------------------------------------------------------------------------------------ utilities
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { (void)hipDeviceSynchronize(); (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8;
        PF_HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return 0;
    }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

struct Tensor {
    float* d = nullptr;      // device storage (library owned)
    int64_t numel = 0;       // expected element count of the SOURCE tensor
    bool set = false;
    // optional repack description
    int kind = 0;            // 0 plain copy, 1 pad rows [rows, cols] -> [rows, cols_pad], 2 conv [O, I, K] -> [O, K*I]
    int rows = 0, cols = 0, cols_pad = 0, taps = 0;
};

struct TensorTable {
    std::map<std::string, Tensor> t;
    ~TensorTable() { for (auto& kv : t) if (kv.second.d) (void)hipFree(kv.second.d); }
    int add(const std::string& name, int64_t numel) {
        Tensor x; x.numel = numel;
        PF_HIP_TRY(hipMalloc((void**)&x.d, sizeof(float) * (size_t)numel));
        t[name] = x; return 0;
    }
    int add_padded(const std::string& name, int rows, int cols, int cols_pad) {
        Tensor x; x.numel = (int64_t)rows * cols; x.kind = 1; x.rows = rows; x.cols = cols; x.cols_pad = cols_pad;
        PF_HIP_TRY(hipMalloc((void**)&x.d, sizeof(float) * (size_t)rows * cols_pad));
        PF_HIP_TRY(hipMemset(x.d, 0, sizeof(float) * (size_t)rows * cols_pad));
        t[name] = x; return 0;
    }
    int add_conv(const std::string& name, int out_c, int in_c, int taps) {
        Tensor x; x.numel = (int64_t)out_c * in_c * taps; x.kind = 2; x.rows = out_c; x.cols = in_c; x.taps = taps;
        PF_HIP_TRY(hipMalloc((void**)&x.d, sizeof(float) * (size_t)x.numel));
        t[name] = x; return 0;
    }
    int set(const char* name, const float* data, int64_t numel) {
        auto it = t.find(name);
        if (it == t.end()) { set_error(std::string("unknown tensor name: ") + name); return -1; }
        Tensor& x = it->second;
        if (numel != x.numel) {
            set_error(std::string("tensor ") + name + ": expected " + std::to_string(x.numel) + " elements, got " +
                      std::to_string(numel));
            return -1;
        }
        if (x.kind == 0) {
            PF_HIP_TRY(hipMemcpy(x.d, data, sizeof(float) * (size_t)numel, hipMemcpyDefault));
        } else if (x.kind == 1) {
            PF_HIP_TRY(hipMemcpy2D(x.d, sizeof(float) * x.cols_pad, data, sizeof(float) * x.cols,
                                   sizeof(float) * x.cols, x.rows, hipMemcpyDefault));
        } else {
            // [O, I, K] -> [O, K*I]: dst[o][k*I + i] = src[o][i][k]; done on the host (load-time only)
            std::vector<float> src((size_t)numel), dst((size_t)numel);
            PF_HIP_TRY(hipMemcpy(src.data(), data, sizeof(float) * (size_t)numel, hipMemcpyDefault));
            const int O = x.rows, I = x.cols, K = x.taps;
            for (int o = 0; o < O; ++o)
                for (int i = 0; i < I; ++i)
                    for (int k = 0; k < K; ++k)
                        dst[((size_t)o * K + k) * I + i] = src[((size_t)o * I + i) * K + k];
            PF_HIP_TRY(hipMemcpy(x.d, dst.data(), sizeof(float) * (size_t)numel, hipMemcpyHostToDevice));
        }
        x.set = true;
        return 0;
    }
    int missing(std::string* first = nullptr) const {
        int n = 0;
        for (auto& kv : t) if (!kv.second.set) { if (n == 0 && first) *first = kv.first; ++n; }
        return n;
    }
    const float* get(const std::string& name) const { return t.at(name).d; }
};

static int check_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible: the gfx950 kernels are the only implementation (no CPU fallback)");
        return -2;
    }
    return 0;
}

static int round_up(int x, int m) { return (x + m - 1) / m * m; }

static int upload_lens(DevBuf& buf, const int32_t* host, int B, hipStream_t s) {
    if (buf.ensure(sizeof(int32_t) * (size_t)B)) return -2;
    PF_HIP_TRY(hipMemcpyAsync(buf.p, host, sizeof(int32_t) * (size_t)B, hipMemcpyHostToDevice, s));
    return 0;
}

// Instrumented launch helpers ---------------------------------------------------------------------------
static int gemm(const GemmArgs& a, hipStream_t s) {
    ProfScope ps(PROF_GEMM, 2.0 * a.M * (double)a.N * a.K, s);
    return launch_gemm_f32(a, s);
}
static int gemm_simple(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc,
                       int M, int N, int K, int relu, const float* R1, int ldr1, const float* R2, int ldr2,
                       hipStream_t s) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.R1 = R1; g.ldr1 = ldr1; g.R2 = R2; g.ldr2 = ldr2;
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.relu = relu;
    return gemm(g, s);
}
static int layernorm(const float* x, int ldx, const float* g, const float* b, float* y, int ldy, int M, int D,
                     int Dpad, float eps, hipStream_t s) {
    ProfScope ps(PROF_LN, 8.0 * M * (double)D, s);   // bytes: read + write
    return launch_layernorm(x, ldx, g, b, y, ldy, M, D, Dpad, eps, s);
}
static int fsmn(const FsmnArgs& a, hipStream_t s) {
    ProfScope ps(PROF_FSMN, (a.R ? 12.0 : 8.0) * a.B * (double)a.T * a.C, s);
    return launch_fsmn(a, s);
}
static int attention(const AttnArgs& a, double flops, hipStream_t s) {
    ProfScope ps(PROF_ATTN, flops, s);
    return launch_attention_f32(a, s);
}

// ================================================================================================ frontend
struct Frontend {
    pf_frontend_config cfg;
    DevBuf window, twiddle, mel_w, mel_off, mel_len, cmvn_shift, cmvn_scale;
    DevBuf fbank, nfr;
    bool has_cmvn = false;
    int feat_dim() const { return cfg.n_mels * cfg.lfr_m; }
};

static float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }

static int frontend_upload_tables(Frontend* f, const std::vector<float>& window, const std::vector<float>& mel) {
    const int nm = f->cfg.n_mels, NB = 257;
    std::vector<int> off(nm), len(nm);
    for (int m = 0; m < nm; ++m) {
        int first = -1, last = -1;
        for (int k = 0; k < NB; ++k)
            if (mel[(size_t)m * NB + k] != 0.f) { if (first < 0) first = k; last = k; }
        off[m] = first < 0 ? 0 : first;
        len[m] = first < 0 ? 0 : last - first + 1;
    }
    if (f->window.ensure(sizeof(float) * window.size())) return -2;
    if (f->mel_w.ensure(sizeof(float) * mel.size())) return -2;
    if (f->mel_off.ensure(sizeof(int) * nm) || f->mel_len.ensure(sizeof(int) * nm)) return -2;
    PF_HIP_TRY(hipMemcpy(f->window.p, window.data(), sizeof(float) * window.size(), hipMemcpyHostToDevice));
    PF_HIP_TRY(hipMemcpy(f->mel_w.p, mel.data(), sizeof(float) * mel.size(), hipMemcpyHostToDevice));
    PF_HIP_TRY(hipMemcpy(f->mel_off.p, off.data(), sizeof(int) * nm, hipMemcpyHostToDevice));
    PF_HIP_TRY(hipMemcpy(f->mel_len.p, len.data(), sizeof(int) * nm, hipMemcpyHostToDevice));
    return 0;
}

// Kaldi tables as in kaldi-native-fbank: window coefficients in float64 (feature-window.cc:25-47), mel
// triangles in float32 over fft bins 0..255 (mel-computations.cc:118-210, strict inequalities at :186)
static int frontend_default_tables(Frontend* f) {
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

// =============================================================================================== encoder
struct EncLayerW {
    const float *n1g, *n1b, *qkv_w, *qkv_b, *fsmn_w, *out_w, *out_b, *n2g, *n2b, *w1, *b1, *w2, *b2;
    int in_dim, in_pad;
};

struct Encoder {
    pf_encoder_config cfg;
    TensorTable tt;
    std::vector<EncLayerW> layers;   // resolved lazily
    bool resolved = false;
    DevBuf x, xn, qkv, mem, ctx, ffn, lens, pe;
    int pe_T = 0;
};

static void enc_layer_names(std::vector<std::pair<std::string, int>>& out, const pf_encoder_config& c) {
    // (prefix, input_dim) of every SAN-M block in execution order
    out.push_back({"encoders0.0.", c.input_dim});
    for (int i = 0; i < c.n_blocks - 1; ++i) out.push_back({"encoders." + std::to_string(i) + ".", c.d_model});
    for (int i = 0; i < c.tp_blocks; ++i) out.push_back({"tp_encoders." + std::to_string(i) + ".", c.d_model});
}

static int encoder_resolve(Encoder* e) {
    std::string first;
    const int miss = e->tt.missing(&first);
    if (miss) { set_error("encoder: " + std::to_string(miss) + " tensors not set, e.g. " + first); return -3; }
    std::vector<std::pair<std::string, int>> names;
    enc_layer_names(names, e->cfg);
    e->layers.clear();
    for (auto& nm : names) {
        const std::string& p = nm.first;
        EncLayerW w;
        w.in_dim = nm.second; w.in_pad = round_up(nm.second, 32);
        w.n1g = e->tt.get(p + "norm1.weight"); w.n1b = e->tt.get(p + "norm1.bias");
        w.qkv_w = e->tt.get(p + "self_attn.linear_q_k_v.weight"); w.qkv_b = e->tt.get(p + "self_attn.linear_q_k_v.bias");
        w.fsmn_w = e->tt.get(p + "self_attn.fsmn_block.weight");
        w.out_w = e->tt.get(p + "self_attn.linear_out.weight"); w.out_b = e->tt.get(p + "self_attn.linear_out.bias");
        w.n2g = e->tt.get(p + "norm2.weight"); w.n2b = e->tt.get(p + "norm2.bias");
        w.w1 = e->tt.get(p + "feed_forward.w_1.weight"); w.b1 = e->tt.get(p + "feed_forward.w_1.bias");
        w.w2 = e->tt.get(p + "feed_forward.w_2.weight"); w.b2 = e->tt.get(p + "feed_forward.w_2.bias");
        e->layers.push_back(w);
    }
    e->resolved = true;
    return 0;
}

// SinusoidalPositionEncoder.encode (embedding.py:396-420) with libm; used only when the caller passes no table
static int encoder_default_pe(Encoder* e, int T, hipStream_t s) {
    const int D = e->cfg.input_dim;
    if (e->pe_T >= T) return 0;
    const int Tn = T + 64;
    std::vector<float> tab((size_t)Tn * D);
    const int half = D / 2;
    const float inc = logf(10000.0f) / (float)(half - 1);
    for (int t = 0; t < Tn; ++t)
        for (int i = 0; i < half; ++i) {
            const float inv = expf((float)i * (-inc));
            const float st = (float)(t + 1) * inv;
            tab[(size_t)t * D + i] = sinf(st);
            tab[(size_t)t * D + half + i] = cosf(st);
        }
    if (e->pe.ensure(sizeof(float) * tab.size())) return -2;
    PF_HIP_TRY(hipMemcpyAsync(e->pe.p, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice, s));
    PF_HIP_TRY(hipStreamSynchronize(s));
    e->pe_T = Tn;
    return 0;
}

static int encoder_block(Encoder* e, const EncLayerW& w, float* x_in, int ld_in, float* x, int B, int T,
                         hipStream_t s) {
    // EncoderLayerSANM.forward (sanm/encoder.py:72-148), normalize_before, no concat_after
    const pf_encoder_config& c = e->cfg;
    const int M = B * T, D = c.d_model, F = c.ffn_dim;
    float* xn = e->xn.as<float>();
    float* qkv = e->qkv.as<float>();
    float* mem = e->mem.as<float>();
    float* ctx = e->ctx.as<float>();
    float* ffn = e->ffn.as<float>();
    const int* lens = e->lens.as<int>();
    int rc;
    // norm1 -> fused QKV projection
    if ((rc = layernorm(x_in, ld_in, w.n1g, w.n1b, xn, w.in_pad, M, w.in_dim, w.in_pad, c.ln_eps, s))) return rc;
    if ((rc = gemm_simple(xn, w.in_pad, w.qkv_w, w.in_pad, w.qkv_b, qkv, 3 * D, M, 3 * D, w.in_pad, 0, nullptr, 0,
                          nullptr, 0, s))) return rc;
    // FSMN memory on the un-split V projection (attention.py:216-239,322-323)
    FsmnArgs fa{};
    fa.in = qkv + 2 * D; fa.ldin = 3 * D; fa.w = w.fsmn_w; fa.R = nullptr; fa.ldr = 0; fa.out = mem; fa.ldo = D;
    fa.lens = lens; fa.B = B; fa.T = T; fa.C = D; fa.K = c.kernel_size;
    fa.left_pad = (c.kernel_size - 1) / 2 + (c.sanm_shift > 0 ? c.sanm_shift : 0);
    if ((rc = fsmn(fa, s))) return rc;
    // scaled dot-product attention over valid keys (attention.py:284-306,324-326)
    AttnArgs aa{};
    aa.Q = qkv; aa.ldq = 3 * D; aa.K = qkv + D; aa.ldk = 3 * D; aa.V = qkv + 2 * D; aa.ldv = 3 * D;
    aa.O = ctx; aa.ldo = D; aa.klens = lens; aa.B = B; aa.H = c.n_heads; aa.Tq = T; aa.Tk = T;
    aa.scale = powf((float)(D / c.n_heads), -0.5f);
    if ((rc = attention(aa, 4.0 * B * (double)T * T * D, s))) return rc;
    // out projection + fsmn memory (+ residual when in == out, encoder.py:120-137)
    const float* resid = (w.in_dim == D) ? x_in : nullptr;
    if ((rc = gemm_simple(ctx, D, w.out_w, D, w.out_b, x, D, M, D, D, 0, mem, D, resid, ld_in, s))) return rc;
    // norm2 -> FFN -> residual (encoder.py:141-146)
    if ((rc = layernorm(x, D, w.n2g, w.n2b, xn, D, M, D, D, c.ln_eps, s))) return rc;
    if ((rc = gemm_simple(xn, D, w.w1, D, w.b1, ffn, F, M, F, D, 1, nullptr, 0, nullptr, 0, s))) return rc;
    if ((rc = gemm_simple(ffn, F, w.w2, F, w.b2, x, D, M, D, F, 0, nullptr, 0, x, D, s))) return rc;
    return 0;
}

// =============================================================================================== predictor
struct Predictor {
    pf_predictor_config cfg;
    TensorTable tt;
    DevBuf col, conv, lens, alphas, peaks, rems, flags, nfires;
    int last_B = 0, last_T = 0;
};

// ================================================================================================= decoder
struct DecLayerW {
    const float *n1g, *n1b, *w1, *b1, *fng, *fnb, *w2, *n2g, *n2b, *fsmn_w, *n3g, *n3b, *q_w, *q_b, *kv_w, *kv_b,
        *o_w, *o_b;
};
struct Decoder {
    pf_decoder_config cfg;
    TensorTable tt;
    std::vector<DecLayerW> layers;
    DecLayerW last;          // decoders3.0 (FFN only)
    bool resolved = false;
    DevBuf x, t1, t2, ffn, ffn2, q, kv, ctx, mem_lens, tok_lens, pval, pidx, hid;
};

static int decoder_resolve(Decoder* d) {
    std::string first;
    const int miss = d->tt.missing(&first);
    if (miss) { set_error("decoder: " + std::to_string(miss) + " tensors not set, e.g. " + first); return -3; }
    d->layers.clear();
    for (int i = 0; i < d->cfg.n_blocks; ++i) {
        const std::string p = "decoders." + std::to_string(i) + ".";
        DecLayerW w;
        w.n1g = d->tt.get(p + "norm1.weight"); w.n1b = d->tt.get(p + "norm1.bias");
        w.w1 = d->tt.get(p + "feed_forward.w_1.weight"); w.b1 = d->tt.get(p + "feed_forward.w_1.bias");
        w.fng = d->tt.get(p + "feed_forward.norm.weight"); w.fnb = d->tt.get(p + "feed_forward.norm.bias");
        w.w2 = d->tt.get(p + "feed_forward.w_2.weight");
        w.n2g = d->tt.get(p + "norm2.weight"); w.n2b = d->tt.get(p + "norm2.bias");
        w.fsmn_w = d->tt.get(p + "self_attn.fsmn_block.weight");
        w.n3g = d->tt.get(p + "norm3.weight"); w.n3b = d->tt.get(p + "norm3.bias");
        w.q_w = d->tt.get(p + "src_attn.linear_q.weight"); w.q_b = d->tt.get(p + "src_attn.linear_q.bias");
        w.kv_w = d->tt.get(p + "src_attn.linear_k_v.weight"); w.kv_b = d->tt.get(p + "src_attn.linear_k_v.bias");
        w.o_w = d->tt.get(p + "src_attn.linear_out.weight"); w.o_b = d->tt.get(p + "src_attn.linear_out.bias");
        d->layers.push_back(w);
    }
    {
        const std::string p = "decoders3.0.";
        DecLayerW w{};
        w.n1g = d->tt.get(p + "norm1.weight"); w.n1b = d->tt.get(p + "norm1.bias");
        w.w1 = d->tt.get(p + "feed_forward.w_1.weight"); w.b1 = d->tt.get(p + "feed_forward.w_1.bias");
        w.fng = d->tt.get(p + "feed_forward.norm.weight"); w.fnb = d->tt.get(p + "feed_forward.norm.bias");
        w.w2 = d->tt.get(p + "feed_forward.w_2.weight");
        d->last = w;
    }
    d->resolved = true;
    return 0;
}

// shared tail: logits / fused argmax of a [M, D] hidden against a [V, D] vocabulary projection
static int vocab_project(const float* hidden, int M, int D, const float* W, const float* bias, int V, float* logits,
                         int32_t* ids, DevBuf& pval, DevBuf& pidx, hipStream_t s) {
    int rc;
    if (logits) {
        if ((rc = gemm_simple(hidden, D, W, D, bias, logits, V, M, V, D, 0, nullptr, 0, nullptr, 0, s))) return rc;
        if (ids) return launch_argmax_rows(logits, V, M, V, ids, s);
        return 0;
    }
    if (!ids) return 0;
    const int nparts = 2 * ceil_div(V, 128);
    if (pval.ensure(sizeof(float) * (size_t)M * nparts) || pidx.ensure(sizeof(int) * (size_t)M * nparts)) return -2;
    GemmArgs g{};
    g.A = hidden; g.lda = D; g.W = W; g.ldw = D; g.bias = bias; g.C = nullptr; g.ldc = 0; g.M = M; g.N = V; g.K = D;
    g.amax_val = pval.as<float>(); g.amax_idx = pidx.as<int>(); g.amax_ld = nparts;
    if ((rc = gemm(g, s))) return rc;
    return launch_argmax_reduce(pval.as<float>(), pidx.as<int>(), nparts, nparts, ids, nullptr, M, s);
}

struct Ctc {
    int d_model, vocab;
    TensorTable tt;
    DevBuf pval, pidx;
};

}  // namespace pf

using namespace pf;

// ================================================================================================== C ABI
extern "C" {

const char* pf_last_error(void) { return get_error(); }
int pf_abi_version(void) { return PF_ABI_VERSION; }
int pf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- profiling hooks (not part of the reference boundary; used by bench.py)
int pf_prof_enable(int on) {
    g_prof_on = on != 0;
    return 0;
}
int pf_prof_reset(void) {
    for (auto& r : g_prof) { g_ev_pool.push_back(r.a); g_ev_pool.push_back(r.b); }
    g_prof.clear();
    return 0;
}
// totals for one kernel kind: 0 gemm (flops), 1 attention (flops), 2 fsmn (bytes), 3 layernorm (bytes), 4 fbank (bytes)
int pf_prof_read(int kind, double* total_ms, double* total_work, int64_t* launches) {
    double ms = 0, work = 0;
    int64_t n = 0;
    for (auto& r : g_prof) {
        if (r.kind != kind) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) { set_error("prof: event sync failed"); return -2; }
        float t = 0;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) { set_error("prof: elapsed failed"); return -2; }
        ms += t; work += r.work; ++n;
    }
    if (total_ms) *total_ms = ms;
    if (total_work) *total_work = work;
    if (launches) *launches = n;
    return 0;
}

// -------------------------------------------------------------------------------------------------- frontend
pf_frontend* pf_frontend_create(const pf_frontend_config* cfg) {
    if (!cfg) { set_error("frontend: null config"); return nullptr; }
    if (check_device()) return nullptr;
    if (cfg->frame_length <= 0 || cfg->frame_length > 448 || cfg->frame_shift <= 0 || cfg->n_mels <= 0 ||
        cfg->n_mels > 128 || cfg->n_mels % 4 || cfg->lfr_m <= 0 || cfg->lfr_n <= 0) {
        set_error("frontend: unsupported config (frame_length <= 448, n_mels % 4 == 0, n_mels <= 128)");
        return nullptr;
    }
    std::unique_ptr<Frontend> f(new Frontend());
    f->cfg = *cfg;
    std::vector<float> tw(512);
    for (int k = 0; k < 256; ++k) {
        const double a = 6.283185307179586476925286766559005 * k / 512.0;
        tw[2 * k] = (float)cos(a);
        tw[2 * k + 1] = (float)(-sin(a));
    }
    if (f->twiddle.ensure(sizeof(float) * 512)) return nullptr;
    if (hipMemcpy(f->twiddle.p, tw.data(), sizeof(float) * 512, hipMemcpyHostToDevice) != hipSuccess) {
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

int pf_frontend_set_tables(pf_frontend* fh, const float* window, const float* mel) {
    Frontend* f = reinterpret_cast<Frontend*>(fh);
    PF_REQUIRE(f && window && mel, "frontend_set_tables: null");
    std::vector<float> w(window, window + f->cfg.frame_length);
    std::vector<float> m(mel, mel + (size_t)f->cfg.n_mels * 257);
    return frontend_upload_tables(f, w, m);
}

int32_t pf_frontend_num_fbank_frames(const pf_frontend* fh, int64_t n) {
    const Frontend* f = reinterpret_cast<const Frontend*>(fh);
    if (!f || n < f->cfg.frame_length) return 0;
    return (int32_t)(1 + (n - f->cfg.frame_length) / f->cfg.frame_shift);   // feature-window.cc:76-90 (snip_edges)
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
        PF_REQUIRE(nfr[b] > 0, "frontend_forward: utterance shorter than one 25 ms window");
        const int t = (nfr[b] + f->cfg.lfr_n - 1) / f->cfg.lfr_n;
        PF_REQUIRE(t <= T_out, "frontend_forward: T_out too small");
        if (feat_lens) feat_lens[b] = t;
        if (nfr[b] > max_fr) max_fr = nfr[b];
    }
    if (f->nfr.ensure(sizeof(int32_t) * B)) return -2;
    PF_HIP_TRY(hipMemcpyAsync(f->nfr.p, nfr.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice, s));
    float* fb = fbank_out;
    if (!fb) {
        if (f->fbank.ensure(sizeof(float) * (size_t)B * max_fr * f->cfg.n_mels)) return -2;
        fb = f->fbank.as<float>();
    }
    FbankArgs a{};
    a.wav = wav; a.wav_stride = (size_t)wav_stride; a.n_frames = f->nfr.as<int>(); a.fbank = fb; a.max_frames = max_fr;
    a.frame_len = f->cfg.frame_length; a.frame_shift = f->cfg.frame_shift; a.n_mels = f->cfg.n_mels;
    a.in_scale = f->cfg.upscale; a.preemph = f->cfg.preemph; a.window = f->window.as<float>();
    a.twiddle = f->twiddle.as<float2>(); a.mel_weight = f->mel_w.as<float>(); a.mel_offset = f->mel_off.as<int>();
    a.mel_len = f->mel_len.as<int>();
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
    l.cmvn_shift = f->has_cmvn ? f->cmvn_shift.as<float>() : nullptr;
    l.cmvn_scale = f->has_cmvn ? f->cmvn_scale.as<float>() : nullptr;
    return launch_lfr_cmvn(l, B, s);
}

// --------------------------------------------------------------------------------------------------- encoder
pf_encoder* pf_encoder_create(const pf_encoder_config* cfg) {
    if (!cfg) { set_error("encoder: null config"); return nullptr; }
    if (check_device()) return nullptr;
    const pf_encoder_config& c = *cfg;
    if (c.d_model <= 0 || c.n_heads <= 0 || c.d_model / c.n_heads != 128 || c.d_model % c.n_heads ||
        c.input_dim % 4 || c.ffn_dim % 32 || c.d_model % 32 || c.n_blocks < 1 || c.tp_blocks < 0 ||
        c.kernel_size != 11) {
        set_error("encoder: unsupported config (need d_model/n_heads == 128, kernel_size == 11, dims % 32 == 0)");
        return nullptr;
    }
    std::unique_ptr<Encoder> e(new Encoder());
    e->cfg = c;
    std::vector<std::pair<std::string, int>> names;
    enc_layer_names(names, c);
    const int D = c.d_model, F = c.ffn_dim;
    int rc = 0;
    for (auto& nm : names) {
        const std::string& p = nm.first;
        const int in = nm.second, in_pad = round_up(in, 32);
        rc |= e->tt.add(p + "norm1.weight", in);
        rc |= e->tt.add(p + "norm1.bias", in);
        rc |= (in_pad == in) ? e->tt.add(p + "self_attn.linear_q_k_v.weight", (int64_t)3 * D * in)
                             : e->tt.add_padded(p + "self_attn.linear_q_k_v.weight", 3 * D, in, in_pad);
        rc |= e->tt.add(p + "self_attn.linear_q_k_v.bias", 3 * D);
        rc |= e->tt.add(p + "self_attn.fsmn_block.weight", (int64_t)D * c.kernel_size);
        rc |= e->tt.add(p + "self_attn.linear_out.weight", (int64_t)D * D);
        rc |= e->tt.add(p + "self_attn.linear_out.bias", D);
        rc |= e->tt.add(p + "norm2.weight", D);
        rc |= e->tt.add(p + "norm2.bias", D);
        rc |= e->tt.add(p + "feed_forward.w_1.weight", (int64_t)F * D);
        rc |= e->tt.add(p + "feed_forward.w_1.bias", F);
        rc |= e->tt.add(p + "feed_forward.w_2.weight", (int64_t)D * F);
        rc |= e->tt.add(p + "feed_forward.w_2.bias", D);
    }
    rc |= e->tt.add("after_norm.weight", D);
    rc |= e->tt.add("after_norm.bias", D);
    if (c.tp_blocks > 0) {
        rc |= e->tt.add("tp_norm.weight", D);
        rc |= e->tt.add("tp_norm.bias", D);
    }
    if (rc) return nullptr;
    return reinterpret_cast<pf_encoder*>(e.release());
}
void pf_encoder_destroy(pf_encoder* e) { delete reinterpret_cast<Encoder*>(e); }
int pf_encoder_set_tensor(pf_encoder* eh, const char* name, const float* data, int64_t numel) {
    Encoder* e = reinterpret_cast<Encoder*>(eh);
    PF_REQUIRE(e && name && data, "encoder_set_tensor: null");
    e->resolved = false;
    return e->tt.set(name, data, numel);
}
int pf_encoder_missing(const pf_encoder* eh) {
    const Encoder* e = reinterpret_cast<const Encoder*>(eh);
    return e ? e->tt.missing() : -1;
}

int pf_encoder_forward(pf_encoder* eh, const float* xs, const int32_t* lens_host, int32_t B, int32_t T,
                       const float* pe, float* out, int32_t run_blocks, void* stream) {
    Encoder* e = reinterpret_cast<Encoder*>(eh);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(e && xs && lens_host && out && B > 0 && T > 0, "encoder_forward: null/empty argument");
    for (int b = 0; b < B; ++b) PF_REQUIRE(lens_host[b] >= 1 && lens_host[b] <= T, "encoder_forward: lens out of range");
    int rc;
    if (!e->resolved && (rc = encoder_resolve(e))) return rc;
    const pf_encoder_config& c = e->cfg;
    const size_t M = (size_t)B * T;
    const int D = c.d_model, F = c.ffn_dim, Din = c.input_dim, Dpad = round_up(Din, 32);
    const int Fbuf = F > Din ? F : Din;
    if (e->x.ensure(sizeof(float) * M * D) || e->xn.ensure(sizeof(float) * M * (Dpad > D ? Dpad : D)) ||
        e->qkv.ensure(sizeof(float) * M * 3 * D) || e->mem.ensure(sizeof(float) * M * D) ||
        e->ctx.ensure(sizeof(float) * M * D) || e->ffn.ensure(sizeof(float) * M * Fbuf))
        return -2;
    if ((rc = upload_lens(e->lens, lens_host, B, s))) return rc;
    if (!pe) {
        if ((rc = encoder_default_pe(e, T, s))) return rc;
        pe = e->pe.as<float>();
    }
    // xs * sqrt(d_model) + PE (encoder.py:409,428). The scaled input is staged in the FFN scratch: block 0 reads
    // it in norm1 (and as residual when input_dim == d_model) strictly before its own FFN overwrites that buffer.
    float* x0 = e->ffn.as<float>();
    const float scale = (float)sqrt((double)D);
    if ((rc = launch_scale_add_pe(xs, pe, x0, B, T, Din, scale, s))) return rc;
    float* x = e->x.as<float>();
    const int total = (int)e->layers.size();
    const int nrun = run_blocks < 0 ? total : (run_blocks < total ? run_blocks : total);
    if (nrun == 0) {
        PF_HIP_TRY(hipMemcpyAsync(out, x0, sizeof(float) * M * Din, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    for (int l = 0; l < nrun; ++l) {
        if (l == 0) rc = encoder_block(e, e->layers[0], x0, Din, x, B, T, s);
        else rc = encoder_block(e, e->layers[l], x, D, x, B, T, s);
        if (rc) return rc;
        if (c.tp_blocks > 0 && l + 1 == c.n_blocks && (run_blocks < 0 || nrun > c.n_blocks)) {
            // SenseVoice: after_norm sits between `encoders` and `tp_encoders` (sense_voice/model.py:645-652)
            if ((rc = layernorm(x, D, e->tt.get("after_norm.weight"), e->tt.get("after_norm.bias"), x, D, (int)M, D, D,
                                c.ln_eps, s))) return rc;
        }
    }
    if (run_blocks >= 0) {
        PF_HIP_TRY(hipMemcpyAsync(out, x, sizeof(float) * M * D, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    const char* fin_w = c.tp_blocks > 0 ? "tp_norm.weight" : "after_norm.weight";
    const char* fin_b = c.tp_blocks > 0 ? "tp_norm.bias" : "after_norm.bias";
    return layernorm(x, D, e->tt.get(fin_w), e->tt.get(fin_b), out, D, (int)M, D, D, c.ln_eps, s);
}

// ------------------------------------------------------------------------------------------------- predictor
pf_predictor* pf_predictor_create(const pf_predictor_config* cfg) {
    if (!cfg) { set_error("predictor: null config"); return nullptr; }
    if (check_device()) return nullptr;
    const pf_predictor_config& c = *cfg;
    if (c.d_model <= 0 || c.d_model % 32 || c.l_order < 0 || c.r_order < 0 || c.threshold != 1.0f) {
        set_error("predictor: unsupported config (d_model % 32 == 0; threshold must be 1.0: cif_wo_hidden_v1 "
                  "detects fires with floor(), cif_predictor.py:838-846)");
        return nullptr;
    }
    std::unique_ptr<Predictor> p(new Predictor());
    p->cfg = c;
    const int D = c.d_model, taps = c.l_order + c.r_order + 1;
    int rc = 0;
    rc |= p->tt.add_conv("cif_conv1d.weight", D, D, taps);
    rc |= p->tt.add("cif_conv1d.bias", D);
    rc |= p->tt.add("cif_output.weight", D);
    rc |= p->tt.add("cif_output.bias", 1);
    if (rc) return nullptr;
    return reinterpret_cast<pf_predictor*>(p.release());
}
void pf_predictor_destroy(pf_predictor* p) { delete reinterpret_cast<Predictor*>(p); }
int pf_predictor_set_tensor(pf_predictor* ph, const char* name, const float* data, int64_t numel) {
    Predictor* p = reinterpret_cast<Predictor*>(ph);
    PF_REQUIRE(p && name && data, "predictor_set_tensor: null");
    return p->tt.set(name, data, numel);
}
int pf_predictor_missing(const pf_predictor* ph) {
    const Predictor* p = reinterpret_cast<const Predictor*>(ph);
    return p ? p->tt.missing() : -1;
}

int pf_predictor_alphas(pf_predictor* ph, const float* hidden, const int32_t* lens_host, int32_t B, int32_t T,
                        float* alphas, float* peaks, int32_t* token_num, void* stream) {
    Predictor* p = reinterpret_cast<Predictor*>(ph);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(p && hidden && lens_host && token_num && B > 0 && T > 0, "predictor_alphas: null/empty argument");
    for (int b = 0; b < B; ++b) PF_REQUIRE(lens_host[b] >= 1 && lens_host[b] <= T, "predictor_alphas: lens out of range");
    std::string first;
    if (p->tt.missing(&first)) { set_error("predictor: tensor not set: " + first); return -3; }
    const pf_predictor_config& c = p->cfg;
    const int D = c.d_model, taps = c.l_order + c.r_order + 1, Te = T + 1;
    const size_t M = (size_t)B * T;
    if (p->col.ensure(sizeof(float) * M * taps * D) || p->conv.ensure(sizeof(float) * M * D) ||
        p->alphas.ensure(sizeof(float) * (size_t)B * Te) || p->peaks.ensure(sizeof(float) * (size_t)B * Te) ||
        p->rems.ensure(sizeof(float) * (size_t)B * Te) || p->flags.ensure(sizeof(int) * (size_t)B * Te) ||
        p->nfires.ensure(sizeof(int) * (size_t)B))
        return -2;
    int rc;
    if ((rc = upload_lens(p->lens, lens_host, B, s))) return rc;
    // relu(Conv1d(D, D, l+r+1)(pad(hidden))) as an im2col GEMM (cif_predictor.py:275-278)
    if ((rc = launch_im2col(hidden, p->col.as<float>(), B, T, D, c.l_order, c.r_order, s))) return rc;
    if ((rc = gemm_simple(p->col.as<float>(), taps * D, p->tt.get("cif_conv1d.weight"), taps * D,
                          p->tt.get("cif_conv1d.bias"), p->conv.as<float>(), D, (int)M, D, taps * D, 1, nullptr, 0,
                          nullptr, 0, s))) return rc;
    AlphaArgs aa{};
    aa.conv = p->conv.as<float>(); aa.w = p->tt.get("cif_output.weight"); aa.bias = p->tt.get("cif_output.bias");
    aa.lens = p->lens.as<int>(); aa.alphas = p->alphas.as<float>(); aa.B = B; aa.T = T; aa.D = D; aa.T_ext = Te;
    aa.smooth = c.smooth_factor; aa.noise = c.noise_threshold;
    if ((rc = launch_alpha(aa, s))) return rc;
    CifScanArgs sa{};
    sa.alphas = p->alphas.as<float>(); sa.peaks = p->peaks.as<float>(); sa.rems = p->rems.as<float>();
    sa.fire_flag = p->flags.as<int>(); sa.n_fires = p->nfires.as<int>(); sa.lens = p->lens.as<int>(); sa.B = B;
    sa.T = T; sa.tail_threshold = c.tail_threshold; sa.tail_mask = c.tail_mask;
    if ((rc = launch_cif_scan(sa, s))) return rc;
    if (alphas) PF_HIP_TRY(hipMemcpyAsync(alphas, p->alphas.p, sizeof(float) * (size_t)B * Te, hipMemcpyDeviceToDevice, s));
    if (peaks) PF_HIP_TRY(hipMemcpyAsync(peaks, p->peaks.p, sizeof(float) * (size_t)B * Te, hipMemcpyDeviceToDevice, s));
    PF_HIP_TRY(hipMemcpyAsync(token_num, p->nfires.p, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, s));
    PF_HIP_TRY(hipStreamSynchronize(s));
    p->last_B = B; p->last_T = T;
    return 0;
}

int pf_predictor_embeds(pf_predictor* ph, const float* hidden, int32_t B, int32_t T, int32_t N, float* embeds,
                        void* stream) {
    Predictor* p = reinterpret_cast<Predictor*>(ph);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(p && hidden && embeds && N >= 0, "predictor_embeds: null argument");
    PF_REQUIRE(B == p->last_B && T == p->last_T, "predictor_embeds: call pf_predictor_alphas with the same batch first");
    CifEmitArgs ea{};
    ea.hidden = hidden; ea.alphas = p->alphas.as<float>(); ea.rems = p->rems.as<float>();
    ea.fire_flag = p->flags.as<int>(); ea.embeds = embeds; ea.B = B; ea.T = T; ea.D = p->cfg.d_model; ea.N = N;
    return launch_cif_emit(ea, s);
}

// --------------------------------------------------------------------------------------------------- decoder
pf_decoder* pf_decoder_create(const pf_decoder_config* cfg) {
    if (!cfg) { set_error("decoder: null config"); return nullptr; }
    if (check_device()) return nullptr;
    const pf_decoder_config& c = *cfg;
    if (c.d_model <= 0 || c.n_heads <= 0 || c.d_model % c.n_heads || c.d_model / c.n_heads != 128 ||
        c.ffn_dim % 32 || c.d_model % 32 || c.n_blocks < 1 || c.kernel_size != 11 || c.vocab_size <= 0) {
        set_error("decoder: unsupported config (need d_model/n_heads == 128, kernel_size == 11, dims % 32 == 0)");
        return nullptr;
    }
    std::unique_ptr<Decoder> d(new Decoder());
    d->cfg = c;
    const int D = c.d_model, F = c.ffn_dim;
    int rc = 0;
    auto add_ffn = [&](const std::string& p) {
        rc |= d->tt.add(p + "norm1.weight", D);
        rc |= d->tt.add(p + "norm1.bias", D);
        rc |= d->tt.add(p + "feed_forward.w_1.weight", (int64_t)F * D);
        rc |= d->tt.add(p + "feed_forward.w_1.bias", F);
        rc |= d->tt.add(p + "feed_forward.norm.weight", F);
        rc |= d->tt.add(p + "feed_forward.norm.bias", F);
        rc |= d->tt.add(p + "feed_forward.w_2.weight", (int64_t)D * F);
    };
    for (int i = 0; i < c.n_blocks; ++i) {
        const std::string p = "decoders." + std::to_string(i) + ".";
        add_ffn(p);
        rc |= d->tt.add(p + "norm2.weight", D);
        rc |= d->tt.add(p + "norm2.bias", D);
        rc |= d->tt.add(p + "self_attn.fsmn_block.weight", (int64_t)D * c.kernel_size);
        rc |= d->tt.add(p + "norm3.weight", D);
        rc |= d->tt.add(p + "norm3.bias", D);
        rc |= d->tt.add(p + "src_attn.linear_q.weight", (int64_t)D * D);
        rc |= d->tt.add(p + "src_attn.linear_q.bias", D);
        rc |= d->tt.add(p + "src_attn.linear_k_v.weight", (int64_t)2 * D * D);
        rc |= d->tt.add(p + "src_attn.linear_k_v.bias", 2 * D);
        rc |= d->tt.add(p + "src_attn.linear_out.weight", (int64_t)D * D);
        rc |= d->tt.add(p + "src_attn.linear_out.bias", D);
    }
    add_ffn("decoders3.0.");
    rc |= d->tt.add("after_norm.weight", D);
    rc |= d->tt.add("after_norm.bias", D);
    rc |= d->tt.add("output_layer.weight", (int64_t)c.vocab_size * D);
    rc |= d->tt.add("output_layer.bias", c.vocab_size);
    if (rc) return nullptr;
    return reinterpret_cast<pf_decoder*>(d.release());
}
void pf_decoder_destroy(pf_decoder* d) { delete reinterpret_cast<Decoder*>(d); }
int pf_decoder_set_tensor(pf_decoder* dh, const char* name, const float* data, int64_t numel) {
    Decoder* d = reinterpret_cast<Decoder*>(dh);
    PF_REQUIRE(d && name && data, "decoder_set_tensor: null");
    d->resolved = false;
    return d->tt.set(name, data, numel);
}
int pf_decoder_missing(const pf_decoder* dh) {
    const Decoder* d = reinterpret_cast<const Decoder*>(dh);
    return d ? d->tt.missing() : -1;
}

// PositionwiseFeedForwardDecoderSANM (sanm/positionwise_feed_forward.py:12-33): w_2(LN(relu(w_1 x))), w_2 bias-free
static int dec_ffn(Decoder* d, const DecLayerW& w, const float* x, float* out, int M, hipStream_t s) {
    const int D = d->cfg.d_model, F = d->cfg.ffn_dim;
    float* t1 = d->t1.as<float>();
    float* ffn = d->ffn.as<float>();
    float* ffn2 = d->ffn2.as<float>();
    int rc;
    if ((rc = layernorm(x, D, w.n1g, w.n1b, t1, D, M, D, D, d->cfg.ln_eps, s))) return rc;
    if ((rc = gemm_simple(t1, D, w.w1, D, w.b1, ffn, F, M, F, D, 1, nullptr, 0, nullptr, 0, s))) return rc;
    if ((rc = layernorm(ffn, F, w.fng, w.fnb, ffn2, F, M, F, F, d->cfg.ln_eps, s))) return rc;
    return gemm_simple(ffn2, F, w.w2, F, nullptr, out, D, M, D, F, 0, nullptr, 0, nullptr, 0, s);
}

int pf_decoder_forward(pf_decoder* dh, const float* memory, const int32_t* mem_lens, const float* embeds,
                       const int32_t* tok_lens, int32_t B, int32_t T, int32_t N, float* logits, int32_t* ids,
                       float* hidden_out, void* stream) {
    Decoder* d = reinterpret_cast<Decoder*>(dh);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(d && memory && mem_lens && embeds && tok_lens && B > 0 && T > 0 && N > 0, "decoder_forward: null/empty");
    for (int b = 0; b < B; ++b) {
        PF_REQUIRE(mem_lens[b] >= 1 && mem_lens[b] <= T, "decoder_forward: memory lens out of range");
        PF_REQUIRE(tok_lens[b] >= 0 && tok_lens[b] <= N, "decoder_forward: token lens out of range");
    }
    int rc;
    if (!d->resolved && (rc = decoder_resolve(d))) return rc;
    const pf_decoder_config& c = d->cfg;
    const int D = c.d_model, F = c.ffn_dim, V = c.vocab_size;
    const int Mq = B * N, Mk = B * T;
    if (d->x.ensure(sizeof(float) * (size_t)Mq * D) || d->t1.ensure(sizeof(float) * (size_t)Mq * D) ||
        d->t2.ensure(sizeof(float) * (size_t)Mq * D) || d->ffn.ensure(sizeof(float) * (size_t)Mq * F) ||
        d->ffn2.ensure(sizeof(float) * (size_t)Mq * F) || d->q.ensure(sizeof(float) * (size_t)Mq * D) ||
        d->kv.ensure(sizeof(float) * (size_t)Mk * 2 * D) || d->ctx.ensure(sizeof(float) * (size_t)Mq * D) ||
        d->hid.ensure(sizeof(float) * (size_t)Mq * D))
        return -2;
    if ((rc = upload_lens(d->mem_lens, mem_lens, B, s))) return rc;
    if ((rc = upload_lens(d->tok_lens, tok_lens, B, s))) return rc;
    float* x = d->x.as<float>();
    float* t1 = d->t1.as<float>();
    float* t2 = d->t2.as<float>();
    PF_HIP_TRY(hipMemcpyAsync(x, embeds, sizeof(float) * (size_t)Mq * D, hipMemcpyDeviceToDevice, s));
    const int left_pad = (c.kernel_size - 1) / 2 + (c.sanm_shift > 0 ? c.sanm_shift : 0);
    for (int l = 0; l < c.n_blocks; ++l) {
        const DecLayerW& w = d->layers[l];
        // DecoderLayerSANM.forward (paraformer/decoder.py:78-121)
        if ((rc = dec_ffn(d, w, x, t2, Mq, s))) return rc;                                    // tgt = FFN(norm1(tgt))
        if ((rc = layernorm(t2, D, w.n2g, w.n2b, t1, D, Mq, D, D, c.ln_eps, s))) return rc;   // norm2
        FsmnArgs fa{};                                                                        // x = residual + fsmn
        fa.in = t1; fa.ldin = D; fa.w = w.fsmn_w; fa.R = x; fa.ldr = D; fa.out = x; fa.ldo = D;
        fa.lens = d->tok_lens.as<int>(); fa.B = B; fa.T = N; fa.C = D; fa.K = c.kernel_size; fa.left_pad = left_pad;
        if ((rc = fsmn(fa, s))) return rc;
        if ((rc = layernorm(x, D, w.n3g, w.n3b, t1, D, Mq, D, D, c.ln_eps, s))) return rc;    // norm3
        if ((rc = gemm_simple(t1, D, w.q_w, D, w.q_b, d->q.as<float>(), D, Mq, D, D, 0, nullptr, 0, nullptr, 0, s)))
            return rc;
        if ((rc = gemm_simple(memory, D, w.kv_w, D, w.kv_b, d->kv.as<float>(), 2 * D, Mk, 2 * D, D, 0, nullptr, 0,
                              nullptr, 0, s))) return rc;
        AttnArgs aa{};
        aa.Q = d->q.as<float>(); aa.ldq = D; aa.K = d->kv.as<float>(); aa.ldk = 2 * D;
        aa.V = d->kv.as<float>() + D; aa.ldv = 2 * D; aa.O = d->ctx.as<float>(); aa.ldo = D;
        aa.klens = d->mem_lens.as<int>(); aa.B = B; aa.H = c.n_heads; aa.Tq = N; aa.Tk = T;
        aa.scale = powf((float)(D / c.n_heads), -0.5f);
        if ((rc = attention(aa, 4.0 * B * (double)N * T * D, s))) return rc;
        if ((rc = gemm_simple(d->ctx.as<float>(), D, w.o_w, D, w.o_b, x, D, Mq, D, D, 0, nullptr, 0, x, D, s)))
            return rc;                                                                        // x = residual + att
    }
    // decoders3: FFN only, no residual (decoder.py:438, DecoderLayerSANM with self_attn = src_attn = None)
    if ((rc = dec_ffn(d, d->last, x, t2, Mq, s))) return rc;
    float* hid = hidden_out ? hidden_out : d->hid.as<float>();
    if ((rc = layernorm(t2, D, d->tt.get("after_norm.weight"), d->tt.get("after_norm.bias"), hid, D, Mq, D, D,
                        c.ln_eps, s))) return rc;
    return vocab_project(hid, Mq, D, d->tt.get("output_layer.weight"), d->tt.get("output_layer.bias"), V, logits, ids,
                         d->pval, d->pidx, s);
}

// ------------------------------------------------------------------------------------------------------- ctc
pf_ctc* pf_ctc_create(int32_t d_model, int32_t vocab) {
    if (check_device()) return nullptr;
    if (d_model <= 0 || d_model % 32 || vocab <= 0) { set_error("ctc: d_model % 32 == 0 required"); return nullptr; }
    std::unique_ptr<Ctc> c(new Ctc());
    c->d_model = d_model; c->vocab = vocab;
    if (c->tt.add("ctc_lo.weight", (int64_t)vocab * d_model) || c->tt.add("ctc_lo.bias", vocab)) return nullptr;
    return reinterpret_cast<pf_ctc*>(c.release());
}
void pf_ctc_destroy(pf_ctc* c) { delete reinterpret_cast<Ctc*>(c); }
int pf_ctc_set_tensor(pf_ctc* ch, const char* name, const float* data, int64_t numel) {
    Ctc* c = reinterpret_cast<Ctc*>(ch);
    PF_REQUIRE(c && name && data, "ctc_set_tensor: null");
    return c->tt.set(name, data, numel);
}
int pf_ctc_missing(const pf_ctc* ch) {
    const Ctc* c = reinterpret_cast<const Ctc*>(ch);
    return c ? c->tt.missing() : -1;
}
int pf_ctc_greedy(pf_ctc* ch, const float* hidden, int32_t M, int32_t* ids, float* logits, void* stream) {
    Ctc* c = reinterpret_cast<Ctc*>(ch);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(c && hidden && M > 0, "ctc_greedy: null/empty");
    std::string first;
    if (c->tt.missing(&first)) { set_error("ctc: tensor not set: " + first); return -3; }
    return vocab_project(hidden, M, c->d_model, c->tt.get("ctc_lo.weight"), c->tt.get("ctc_lo.bias"), c->vocab, logits,
                         ids, c->pval, c->pidx, s);
}

// -------------------------------------------------------------------------------------------- single kernels
int pf_k_gemm_f32(const float* A, int32_t lda, const float* W, int32_t ldw, const float* bias, const float* R1,
                  int32_t ldr1, const float* R2, int32_t ldr2, float* C, int32_t ldc, int32_t M, int32_t N, int32_t K,
                  int32_t relu, void* stream) {
    return gemm_simple(A, lda, W, ldw, bias, C, ldc, M, N, K, relu, R1, ldr1, R2, ldr2,
                       reinterpret_cast<hipStream_t>(stream));
}
int pf_k_gemm_argmax_f32(const float* A, int32_t lda, const float* W, int32_t ldw, const float* bias, int32_t M,
                         int32_t N, int32_t K, int32_t* ids, float* sval, int32_t* sidx, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(ids && sval && sidx, "gemm_argmax: scratch [M, 2*ceil(N/128)] required");
    const int nparts = 2 * ceil_div(N, 128);
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.M = M; g.N = N; g.K = K;
    g.amax_val = sval; g.amax_idx = sidx; g.amax_ld = nparts;
    int rc;
    if ((rc = gemm(g, s))) return rc;
    return launch_argmax_reduce(sval, sidx, nparts, nparts, ids, nullptr, M, s);
}
int pf_k_layernorm(const float* x, int32_t ldx, const float* gamma, const float* beta, float* y, int32_t ldy,
                   int32_t M, int32_t D, int32_t Dpad, float eps, void* stream) {
    return layernorm(x, ldx, gamma, beta, y, ldy, M, D, Dpad, eps, reinterpret_cast<hipStream_t>(stream));
}
int pf_k_fsmn(const float* in, int32_t ldin, const float* w, const float* R, int32_t ldr, float* out, int32_t ldo,
              const int32_t* lens_dev, int32_t B, int32_t T, int32_t C, int32_t K, int32_t left_pad, void* stream) {
    FsmnArgs fa{};
    fa.in = in; fa.ldin = ldin; fa.w = w; fa.R = R; fa.ldr = ldr; fa.out = out; fa.ldo = ldo; fa.lens = lens_dev;
    fa.B = B; fa.T = T; fa.C = C; fa.K = K; fa.left_pad = left_pad;
    return fsmn(fa, reinterpret_cast<hipStream_t>(stream));
}
int pf_k_attention_f32(const float* Q, int32_t ldq, const float* K, int32_t ldk, const float* V, int32_t ldv,
                       float* O, int32_t ldo, const int32_t* klens_dev, int32_t B, int32_t H, int32_t Tq, int32_t Tk,
                       float scale, void* stream) {
    AttnArgs aa{};
    aa.Q = Q; aa.ldq = ldq; aa.K = K; aa.ldk = ldk; aa.V = V; aa.ldv = ldv; aa.O = O; aa.ldo = ldo;
    aa.klens = klens_dev; aa.B = B; aa.H = H; aa.Tq = Tq; aa.Tk = Tk; aa.scale = scale;
    return attention(aa, 4.0 * B * (double)Tq * Tk * H * 128, reinterpret_cast<hipStream_t>(stream));
}
int pf_k_cif(const float* alphas, const float* hidden, int32_t B, int32_t T, int32_t D, int32_t N, float* peaks,
             int32_t* n_fires, float* embeds, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(alphas && hidden && peaks && n_fires && embeds && B > 0 && T > 0 && D > 0, "k_cif: null/empty");
    static DevBuf al, pk, rm, ff, ln;
    const int Te = T + 1;
    if (al.ensure(sizeof(float) * (size_t)B * Te) || pk.ensure(sizeof(float) * (size_t)B * Te) ||
        rm.ensure(sizeof(float) * (size_t)B * Te) || ff.ensure(sizeof(int) * (size_t)B * Te) ||
        ln.ensure(sizeof(int) * (size_t)B)) return -2;
    std::vector<int> lens(B, T);
    PF_HIP_TRY(hipMemcpyAsync(ln.p, lens.data(), sizeof(int) * B, hipMemcpyHostToDevice, s));
    PF_HIP_TRY(hipMemcpy2DAsync(al.p, sizeof(float) * Te, alphas, sizeof(float) * T, sizeof(float) * T, B,
                                hipMemcpyDeviceToDevice, s));
    CifScanArgs sa{};
    sa.alphas = al.as<float>(); sa.peaks = pk.as<float>(); sa.rems = rm.as<float>(); sa.fire_flag = ff.as<int>();
    sa.n_fires = n_fires; sa.lens = ln.as<int>(); sa.B = B; sa.T = T; sa.tail_threshold = 0.f; sa.tail_mask = 1;
    int rc;
    if ((rc = launch_cif_scan(sa, s))) return rc;
    PF_HIP_TRY(hipMemcpy2DAsync(peaks, sizeof(float) * T, pk.p, sizeof(float) * Te, sizeof(float) * T, B,
                                hipMemcpyDeviceToDevice, s));
    CifEmitArgs ea{};
    ea.hidden = hidden; ea.alphas = al.as<float>(); ea.rems = rm.as<float>(); ea.fire_flag = ff.as<int>();
    ea.embeds = embeds; ea.B = B; ea.T = T; ea.D = D; ea.N = N;
    if ((rc = launch_cif_emit(ea, s))) return rc;
    PF_HIP_TRY(hipStreamSynchronize(s));   // `lens` is a host temporary
    return 0;
}
int pf_k_gemm_f32_time(const float* A, int32_t lda, const float* W, int32_t ldw, const float* bias, float* C,
                       int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t iters, float* ms_out, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(iters > 0 && ms_out, "gemm_time: iters > 0");
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    int rc;
    for (int i = 0; i < 3; ++i) if ((rc = launch_gemm_f32(g, s))) return rc;
    hipEvent_t a, b;
    PF_HIP_TRY(hipEventCreate(&a));
    PF_HIP_TRY(hipEventCreate(&b));
    PF_HIP_TRY(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) if ((rc = launch_gemm_f32(g, s))) return rc;
    PF_HIP_TRY(hipEventRecord(b, s));
    PF_HIP_TRY(hipEventSynchronize(b));
    float ms = 0.f;
    PF_HIP_TRY(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms_out = ms / iters;
    return 0;
}

}  // extern "C"
