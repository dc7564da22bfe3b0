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
#include <mutex>
#include <string>
#include <vector>

#include "../../include/paraformer_hip.h"
#include "cif.h"
#include "lstm.h"
#include "common.h"
#include "frontend.h"
#include "stream.h"
#include "engine_tables.h"

namespace pf {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* get_error() { return g_err.c_str(); }

// ------------------------------------------------------------------------------------------------ profiling
// Optional hipEvent instrumentation of the dominant kernels (bench.py roofline line).
struct ProfRec { hipEvent_t a, b; int kind; double work; const char* tag; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_ev_pool;
static hipEvent_t prof_event() {
    if (!g_ev_pool.empty()) { hipEvent_t e = g_ev_pool.back(); g_ev_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
struct ProfScope {
    hipEvent_t a, b; hipStream_t s; int kind; double work; bool on; const char* tag;
    // `tag_`: a string literal naming the call site ("enc.w1", ...): bench.py reports time / rate per tag (pf_prof_read_tag)
    ProfScope(int kind_, double work_, hipStream_t s_, const char* tag_ = nullptr) : s(s_), kind(kind_), work(work_), on(g_prof_on), tag(tag_) {
        if (on) { a = prof_event(); b = prof_event(); (void)hipEventRecord(a, s); }
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(b, s); g_prof.push_back({a, b, kind, work, tag}); } }
};
enum { PROF_GEMM = 0, PROF_ATTN = 1, PROF_FSMN = 2, PROF_LN = 3, PROF_FBANK = 4, PROF_GEMM3 = 5, PROF_KINDS = 6 };

// ------------------------------------------------------------------------------------------------ utilities
// bumped whenever a workspace moves: captured hipGraphs hold raw workspace pointers and must be re-captured then
static unsigned long long g_ws_epoch = 0;

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
        ++g_ws_epoch;
        return 0;
    }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

// the fused feed-forward runs ONE 128-row workgroup per CU: take it where the workgroups fill whole rounds of the CUs to >= 85 %
static bool ffn_fills_rounds(int M) {
    static const int n_cu = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) n = pr.multiProcessorCount;
        return n;
    }();
    const int blocks = ceil_div(M, 128), rounds = ceil_div(blocks, n_cu);
    return blocks >= (int)(0.85 * rounds * n_cu);
}

struct Tensor {
    float* d = nullptr;      // device storage (library owned)
    int64_t numel = 0;       // expected element count of the SOURCE tensor
    bool set = false;
    // optional repack description
    int kind = 0;            // 0 plain copy, 1 pad rows [rows, cols] -> [rows, cols_pad], 2 conv [O, I, K] -> [O, K*I],
                             // 3 upsampling conv, 4 tiled vector (see add_upsample / add_tiled)
    int rows = 0, cols = 0, cols_pad = 0, taps = 0;
    size_t device_elems() const {       // floats of the device image (repacked layouts differ from the source count)
        if (kind == 1) return (size_t)rows * cols_pad;
        if (kind == 4) return (size_t)rows * cols;
        return (size_t)numel;
    }
};

struct TensorTable {
    std::map<std::string, Tensor> t;
    unsigned long long version = 0;    // bumped by every set(): consumers that cache derived data (streaming f16x2 step) compare it
    ~TensorTable() {
        for (auto& kv : t) if (kv.second.d) (void)hipFree(kv.second.d);
        for (auto& kv : b16) if (kv.second) (void)hipFree(kv.second);
    }
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
    // ConvTranspose1d(I, O, k = stride = U) weight [I][O][U] -> the [U * O, I] operand of one GEMM whose output row
    // (b, t) holds the U upsampled frames of input frame t back to back: dst[(j * O + o) * I + i] = src[(i * O + o) * U + j]
    int add_upsample(const std::string& name, int in_c, int out_c, int U) {
        Tensor x; x.numel = (int64_t)in_c * out_c * U; x.kind = 3; x.rows = out_c; x.cols = in_c; x.taps = U;
        PF_HIP_TRY(hipMalloc((void**)&x.d, sizeof(float) * (size_t)x.numel));
        t[name] = x; return 0;
    }
    // a [n] vector stored `reps` times back to back (the bias of the upsampling GEMM)
    int add_tiled(const std::string& name, int n, int reps) {
        Tensor x; x.numel = n; x.kind = 4; x.rows = reps; x.cols = n;
        PF_HIP_TRY(hipMalloc((void**)&x.d, sizeof(float) * (size_t)n * reps));
        t[name] = x; return 0;
    }
    // LSTM weight_hh [4H][H] (gates i, f, g, o stacked): kept in torch's layout, the step kernel picks its 16 rows (lstm.hip)
    int add_lstm_hh(const std::string& name, int H) { return add(name, (int64_t)4 * H * H); }
    int set(const char* name, const float* data, int64_t numel) {
        auto it = t.find(name);
        if (it == t.end()) { set_error(std::string("unknown tensor name: ") + name); return -1; }
        Tensor& x = it->second;
        ++version;
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
        } else if (x.kind >= 3) {
            std::vector<float> src((size_t)numel);
            PF_HIP_TRY(hipMemcpy(src.data(), data, sizeof(float) * (size_t)numel, hipMemcpyDefault));
            std::vector<float> dst;
            if (x.kind == 3) {
                const int O = x.rows, I = x.cols, U = x.taps;
                dst.resize((size_t)numel);
                for (int i = 0; i < I; ++i)
                    for (int o = 0; o < O; ++o)
                        for (int j = 0; j < U; ++j)
                            dst[((size_t)j * O + o) * I + i] = src[((size_t)i * O + o) * U + j];
            } else {
                dst.resize((size_t)x.cols * x.rows);
                for (int r = 0; r < x.rows; ++r) std::copy(src.begin(), src.end(), dst.begin() + (size_t)r * x.cols);
            }
            PF_HIP_TRY(hipMemcpy(x.d, dst.data(), sizeof(float) * dst.size(), hipMemcpyHostToDevice));
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
    // bf16 copy of a (repacked) tensor for the bf16-operand mode, made on first use and dropped when the fp32
    // master changes
    std::map<std::string, unsigned short*> b16;
    std::map<std::string, int> exp2;    // exponents of the #split2 entries
    void drop_bf16() {
        for (auto& kv : b16) if (kv.second) (void)hipFree(kv.second);
        b16.clear();
        exp2.clear();
    }
    // the three bf16 planes [3][rows, cols] of a [rows, cols] weight (gemm_split3.hip); shares the b16 cache under a
    // suffixed key, so it is dropped with it
    const unsigned short* get_split3(const std::string& name, int rows, int cols, hipStream_t s) {
        const std::string key = name + "#split3";
        auto it = b16.find(key);
        if (it != b16.end()) return it->second;
        const Tensor& x = t.at(name);
        const size_t n = x.kind == 1 ? (size_t)x.rows * x.cols_pad : (size_t)x.numel;
        unsigned short* p = nullptr;
        if (n != (size_t)rows * cols || cols % 8 != 0 || hipMalloc((void**)&p, sizeof(unsigned short) * 3 * n) != hipSuccess) {
            set_error("split3 planes of " + name + " failed");
            return nullptr;
        }
        if (launch_split3(x.d, cols, p, cols, n, rows, cols, s)) { (void)hipFree(p); return nullptr; }
        b16[key] = p;
        return p;
    }
    // the two fp16 planes [2][rows, cols] of weight * 2^e (gemm_f16x2.hip), e from max |w| so that the largest hi lies in
    // [2^14, 2^15); cached like the bf16 copies, the exponent beside it
    const unsigned short* get_split2(const std::string& name, int rows, int cols, int* e_out, hipStream_t s) {
        const std::string key = name + "#split2";
        auto it = b16.find(key);
        if (it != b16.end()) { *e_out = exp2[key]; return it->second; }
        const Tensor& x = t.at(name);
        const size_t n = x.kind == 1 ? (size_t)x.rows * x.cols_pad : (size_t)x.numel;
        float amax = 0.f;
        if (n != (size_t)rows * cols || cols % 8 != 0 || dev_absmax(x.d, n, &amax, s)) {
            set_error("split2 planes of " + name + " failed");
            return nullptr;
        }
        const int e = amax > 0.f ? 14 - (int)floorf(log2f(amax)) : 0;
        unsigned short* p = nullptr;
        if (e < -100 || e > 100 || hipMalloc((void**)&p, sizeof(unsigned short) * 2 * n) != hipSuccess) {
            set_error("split2 planes of " + name + " failed");
            return nullptr;
        }
        if (launch_split2(x.d, cols, p, cols, n, rows, cols, ldexpf(1.f, e), s)) { (void)hipFree(p); return nullptr; }
        b16[key] = p; exp2[key] = e; *e_out = e;
        return p;
    }
    // load-time reductions (one float back to the host)
    static int dev_absmax(const float* x, size_t n, float* out, hipStream_t s) {
        float* d = nullptr;
        PF_HIP_TRY(hipMalloc((void**)&d, sizeof(float)));
        int rc = launch_absmax(x, n, d, s);
        if (!rc && hipMemcpyAsync(out, d, sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) rc = -2;
        if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = -2;
        (void)hipFree(d);
        return rc;
    }
    // max over rows n of (in_bound * sum_k |W[n, k]| + |bias[n]|): an a-priori bound on |W x + b| for |x_k| <= in_bound
    static int dev_linear_bound(const float* W, int rows, int cols, int ld, const float* bias, float in_bound, float* out,
                                hipStream_t s) {
        float* d = nullptr;
        PF_HIP_TRY(hipMalloc((void**)&d, sizeof(float)));
        int rc = launch_rowl1_bound(W, rows, cols, ld, bias, in_bound, d, s);
        if (!rc && hipMemcpyAsync(out, d, sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) rc = -2;
        if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = -2;
        (void)hipFree(d);
        return rc;
    }
    const unsigned short* get_bf16(const std::string& name, hipStream_t s) {
        auto it = b16.find(name);
        if (it != b16.end()) return it->second;
        const Tensor& x = t.at(name);
        const size_t n = x.kind == 1 ? (size_t)x.rows * x.cols_pad : (size_t)x.numel;
        unsigned short* p = nullptr;
        if (n % 4 != 0 || hipMalloc((void**)&p, sizeof(unsigned short) * n) != hipSuccess) {
            set_error("bf16 copy of " + name + " failed");
            return nullptr;
        }
        if (launch_cast_bf16(x.d, p, n, s)) { (void)hipFree(p); return nullptr; }
        b16[name] = p;
        return p;
    }
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

// Host -> device uploads of per-call metadata (lengths, row maps, frame counts). hipMemcpyAsync from PAGEABLE host memory
// may read its source when the copy command EXECUTES, not when the call returns -- with a busy queue (e.g. a second process
// on the GPU) a std::vector local or a caller's ctypes array is gone or rewritten by then, and the kernel behind it sees
// garbage (observed: a wrong frame count in the frontend, i.e. features that differed from run to run). Every such upload
// therefore goes through a ring of PINNED staging slots owned by the library: the bytes are copied into a slot at call time,
// the DMA reads the slot, and a slot is reused only after the event recorded behind its copy has completed.
struct StageRing {
    // One ring PER DEVICE (round-3 review): an event belongs to the device it was created on, so a process-global ring broke
    // as soon as handles lived on two devices. Slots are portable pinned memory; a free slot (its event has completed) is
    // preferred over waiting, and a wait for a busy slot happens OUTSIDE the lock so that one busy stream does not stall the
    // uploads of every other handle and thread.
    struct Slot { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool busy = false; bool claimed = false; };
    static constexpr int N = 16;
    Slot slots[N];
    int next = 0;
    std::mutex mu;
    int upload(void* dst, const void* src, size_t bytes, hipStream_t s) {
        if (bytes == 0) return 0;
        Slot* sl = nullptr;
        {
            std::lock_guard<std::mutex> lock(mu);
            for (int k = 0; k < N && !sl; ++k) {                     // first slot whose last copy has completed
                Slot& c = slots[(next + k) % N];
                if (c.claimed) continue;
                if (c.busy && hipEventQuery(c.ev) != hipSuccess) continue;
                c.busy = false; sl = &c; next = (next + k + 1) % N;
            }
            for (int k = 0; k < N && !sl; ++k) {                     // none free: take the oldest unclaimed one and wait for it below
                Slot& c = slots[(next + k) % N];
                if (!c.claimed) { sl = &c; next = (next + k + 1) % N; }
            }
            if (!sl) { set_error("upload: every staging slot is claimed by another thread"); return -2; }
            sl->claimed = true;
        }
        int rc = 0;
        auto fail = [&](const char* what) { set_error(std::string("upload: ") + what); rc = -2; };
        if (sl->busy) { if (hipEventSynchronize(sl->ev) != hipSuccess) fail("event wait"); sl->busy = false; }
        if (!rc && sl->cap < bytes) {
            if (sl->p) (void)hipHostFree(sl->p);
            sl->p = nullptr; sl->cap = 0;
            const size_t want = bytes + bytes / 4 + 256;
            if (hipHostMalloc(&sl->p, want, hipHostMallocPortable) != hipSuccess) fail("pinned allocation"); else sl->cap = want;
        }
        if (!rc && !sl->ev && hipEventCreateWithFlags(&sl->ev, hipEventDisableTiming) != hipSuccess) fail("event creation");
        if (!rc) {
            memcpy(sl->p, src, bytes);
            if (hipMemcpyAsync(dst, sl->p, bytes, hipMemcpyHostToDevice, s) != hipSuccess) fail("hipMemcpyAsync");
            else if (hipEventRecord(sl->ev, s) != hipSuccess) fail("event record");
            else sl->busy = true;
        }
        std::lock_guard<std::mutex> lock(mu);
        sl->claimed = false;
        return rc;
    }
};
static StageRing* stage_ring_for_current_device() {
    static std::mutex mu;
    static std::map<int, std::unique_ptr<StageRing>> rings;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    auto& r = rings[dev];
    if (!r) r.reset(new StageRing());
    return r.get();
}
static int upload_h2d(void* dst, const void* src, size_t bytes, hipStream_t s) { return stage_ring_for_current_device()->upload(dst, src, bytes, s); }

static int upload_lens(DevBuf& buf, const int32_t* host, int B, hipStream_t s) {
    if (buf.ensure(sizeof(int32_t) * (size_t)B)) return -2;
    return upload_h2d(buf.p, host, sizeof(int32_t) * (size_t)B, s);
}

// Instrumented launch helpers ---------------------------------------------------------------------------
// Kernel choice is by CALLER, never by batch size, so that a clip's (or a stream's) result does not depend on what
// else is in the batch: the offline path always takes the 128 x 128 tile kernel, the streaming step always the small-M
// weight-streaming kernel (whose K slicing depends on K only). g_skinny_max_m is a test hook for pf_k_gemm_f32.
static int g_skinny_max_m = 0;
static thread_local bool g_stream_mode = false;
struct StreamModeScope {
    bool prev;
    StreamModeScope() : prev(g_stream_mode) { g_stream_mode = true; }
    ~StreamModeScope() { g_stream_mode = prev; }
};
static int gemm(const GemmArgs& a, hipStream_t s) {
    if ((g_stream_mode || a.M <= g_skinny_max_m) && gemm_skinny_applicable(a)) return launch_gemm_skinny(a, s);
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
// `appended` (optional): the caller filled the app_* fields; on return it says whether the kernel did the ring append
// itself (few-query kernel, one workgroup per (stream, head)) -- otherwise the caller launches ring_append_kernel
static int attention(const AttnArgs& a, double flops, hipStream_t s, bool x3 = false, int dk = 128, bool* appended = nullptr) {
    ProfScope ps(PROF_ATTN, flops, s);
    if (appended) *appended = false;
    AttnArgs f = a;
    f.few_q = 0;
    if (dk != 128) { f.app_rows = 0; return launch_attention_small(f, dk, s); }      // CT-Transformer sized heads
    if (x3) { f.app_rows = 0; return launch_attention_split3(f, s); }
    f.few_q = (g_stream_mode || g_skinny_max_m > 0) ? 1 : 0;   // by caller (streaming step; g_skinny_max_m: test hook)
    if (!appended || !attention_fuses_append(f)) f.app_rows = 0;
    else *appended = true;
    return launch_attention_f32(f, s);
}

// ================================================================================================ frontend
struct Frontend {
    pf_frontend_config cfg;
    DevBuf window, twiddle, piece_w, piece_k0, mel_first, mel_count, cmvn_shift, cmvn_scale;
    int n_pieces = 0;
    DevBuf fbank, nfr;
    bool has_cmvn = false;
    int feat_dim() const { return cfg.n_mels * cfg.lfr_m; }
    float dither = 0.f; unsigned long long dither_seed = 0; unsigned dither_calls = 0;   // pf_frontend_set_dither
};

static float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }

static int frontend_upload_tables(Frontend* f, const std::vector<float>& window, const std::vector<float>& mel) {
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
    const unsigned short *qkv_w16 = nullptr, *out_w16 = nullptr, *w1_16 = nullptr, *w2_16 = nullptr;   // bf16 mode
    const unsigned short *qkv_w3 = nullptr, *out_w3 = nullptr, *w1_3 = nullptr, *w2_3 = nullptr;       // bf16x3 mode
    // f16x2 mode: weight planes with their exponents, and the exponents of the activation planes (from a-priori bounds)
    const unsigned short *qkv_w2 = nullptr, *out_w2 = nullptr, *w1_2 = nullptr, *w2_2 = nullptr;
    int ew_qkv = 0, ew_out = 0, ew_1 = 0, ew_2 = 0;
    int e_x1 = 0, e_q = 0, e_k = 0, e_v = 0, e_x2 = 0, e_h = 0;
    std::string prefix;
};

struct Encoder {
    pf_encoder_config cfg;
    TensorTable tt;
    std::vector<EncLayerW> layers;   // resolved lazily
    bool resolved = false;
    DevBuf x, xn, qkv, mem, ctx, ffn, lens, pe;
    int pe_T = 0;
    // 0: fp32 MFMA everywhere; 1: bf16 operands for GEMMs + attention (throughput mode, bf16-class error);
    // 2: fp32 results from bf16x3 split operands on the bf16 MFMA (gemm_split3.hip), everything else as in mode 0
    int precision = 0;
    DevBuf xn16, qkv16, ctx16, ffn16;   // mode 1: bf16 activations; mode 2: xn16 / ctx16 / ffn16 hold three planes each
    // mode 3 (f16x2): xn16 / ctx16 / ffn16 hold two fp16 planes each; q2 / k2 / vt2 are the attention operands the QKV
    // projection writes (k2 with 32 rows of slack per plane, vt2 rows of Mp + 64 columns: tiles may run past the last row)
    DevBuf q2, k2, vt2;
    DevBuf splitk;                      // streaming f16x2 step: the split-K partials of w_2 [4][rows][d_model] (gemm_f16x2.hip)
    int Tp = 0;                         // rows per sequence of the current forward (T, or T rounded up to 16 in mode 3)
    // mode 3, packed row layout (pf_encoder_set_row_packing): sequence b occupies the slot [offs[b], offs[b + 1]) =
    // min(len_b + pack_extra, T) rows rounded up to 16, one slot right after the other; the rows behind are not computed
    int pack_extra = -1;                // < 0: off (every row of [B, T] is computed, as the reference does)
    DevBuf offs_dev, map_dev;
    std::vector<int32_t> h_offs, h_map;
    const int* cur_offs = nullptr;      // device offsets of the forward in flight (nullptr: padded layout)
    int cur_M = 0;                      // its row count (a multiple of 16)
    // SANMVadEncoder (pf_encoder_set_vad_mask): every block's attention is causal, the last block's uses the VAD corner
    bool vad_mask = false;
    std::vector<int32_t> h_vad;
    DevBuf vad_dev;
    int cur_mask_mode = 0;              // mask mode of the block being enqueued (AttnArgs.mask_mode)
    // mode 3: the N = 512 projections (linear_out, w_2) run in their full-row form (gemm_f16x2_row.hip) whose epilogue does
    // the residual adds AND the LayerNorm that follows (norm2; the NEXT block's norm1), bitwise equal to the separate kernels.
    // fuse_row = 0 restores the separate launches (A/B measurements, tests)
    int fuse_row = 1;
    // fsmn_fused = 1: the FSMN memory block is computed in linear_out's full-row epilogue (kernel 11, no shift, fuse_row on)
    int fsmn_fused = 1;
    // ffn_fused: w_1 + ReLU + w_2 + residual (+ the next LayerNorm) as ONE launch (gemm_f16x2_ffn.hip), the hidden activations in
    // registers. 1 = where the row count fills whole rounds of 128-row workgroups (the kernel runs one workgroup per CU; a last
    // round that is mostly empty costs a full round -- the two-kernel pair has finer shapes for those batches), 2 = always,
    // 0 = never. The fused launch returns the bits of the pair for the fp32 stream (tested).
    int ffn_fused = 0;                  // (off until the exact-wait schedule of gemm_f16x2_ffn.hip beats the pair)
    int ffn_abl = 0;                    // debugging hook: FfnArgs.abl
    int row_bm = 0;                     // GemmRowArgs.block_rows of the full-row GEMMs (0: by the row count)
    DevBuf fs_grp;                      // int32 [2][M / 16]: valid v rows [lo, hi) of the sequence owning each 16-row group
    std::vector<int32_t> h_fs;
    const int* cur_fs = nullptr; int cur_fs_groups = 0;
    int attn_variant = 3;               // attention_f16x2.hip schedule (3: lazy rescale)
    int row_nt = 1;                     // non-temporal A loads in the full-row GEMMs: 0 none, 1 linear_out (K = 512), 2 linear_out and w_2
    int gemm_tile = 0;                  // Gemm2Args.tile of the block's GEMMs (0: by shape; 5: 128 x 256, two workgroups per CU)
};

// exponent e with bound * 2^e <= 2^15 (a factor 2 under fp16's 65504 for the roundings on the way)
static int exp_for_bound(float bound) {
    if (!(bound > 0.f)) return 15;
    int e = (int)floorf(log2f(32768.f / bound));
    return e > 15 ? 15 : e;
}
static float pow2f(int e) { return ldexpf(1.f, e); }

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
        w.prefix = p;
        w.in_dim = nm.second; w.in_pad = round_up(nm.second, 64);
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

// f16x2 mode: weight planes with their exponents and the exponents of the activation planes of every block (once per
// weight set; load-time reductions with host round trips -- never inside a graph capture)
static int encoder_prepare_x2(Encoder* e, hipStream_t s) {
    const pf_encoder_config& c = e->cfg;
    const int D = c.d_model, F = c.ffn_dim;
    const float dk_scale = powf((float)(D / c.n_heads), -0.5f);
    for (auto& w : e->layers) {
        if (w.qkv_w2) continue;
        const std::string qkv_name = w.prefix + "self_attn.linear_q_k_v.weight";
        w.qkv_w2 = e->tt.get_split2(qkv_name, 3 * D, w.in_pad, &w.ew_qkv, s);
        w.out_w2 = e->tt.get_split2(w.prefix + "self_attn.linear_out.weight", D, D, &w.ew_out, s);
        w.w1_2 = e->tt.get_split2(w.prefix + "feed_forward.w_1.weight", F, D, &w.ew_1, s);
        w.w2_2 = e->tt.get_split2(w.prefix + "feed_forward.w_2.weight", D, F, &w.ew_2, s);
        if (!w.qkv_w2 || !w.out_w2 || !w.w1_2 || !w.w2_2) return -2;
        // a-priori bounds -> plane exponents. LayerNorm: |y| <= sqrt(D) max|gamma| + max|beta|; a Linear over inputs
        // bounded by b: |W x + c| <= b max_n sum_k |W[n, k]| + |c[n]|; attention output <= max |v|; relu only shrinks
        float g1, b1, g2, b2, bq, bk, bv, bh;
        if (TensorTable::dev_absmax(w.n1g, w.in_dim, &g1, s) || TensorTable::dev_absmax(w.n1b, w.in_dim, &b1, s) ||
            TensorTable::dev_absmax(w.n2g, D, &g2, s) || TensorTable::dev_absmax(w.n2b, D, &b2, s)) return -2;
        const float bx1 = sqrtf((float)w.in_dim) * g1 + b1, bx2 = sqrtf((float)D) * g2 + b2;
        if (TensorTable::dev_linear_bound(w.qkv_w, D, w.in_pad, w.in_pad, w.qkv_b, bx1, &bq, s) ||
            TensorTable::dev_linear_bound(w.qkv_w + (size_t)D * w.in_pad, D, w.in_pad, w.in_pad, w.qkv_b + D, bx1, &bk, s) ||
            TensorTable::dev_linear_bound(w.qkv_w + (size_t)2 * D * w.in_pad, D, w.in_pad, w.in_pad, w.qkv_b + 2 * D, bx1, &bv, s) ||
            TensorTable::dev_linear_bound(w.w1, F, D, D, w.b1, bx2, &bh, s)) return -2;
        w.e_x1 = exp_for_bound(bx1); w.e_x2 = exp_for_bound(bx2);
        w.e_q = exp_for_bound(bq * dk_scale); w.e_k = exp_for_bound(bk); w.e_v = exp_for_bound(bv); w.e_h = exp_for_bound(bh);
    }
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

// per-layer streaming context: attention additionally sees the cached K/V ring of this layer and the first
// `append_rows` K/V rows of the window are appended to it afterwards (attention.py:343-361)
struct EncChunkCtx {
    float* ring; int cap; const StreamDev* st; int append_rows;
    const int* lens;     // device [B]: every window row is valid in a chunk
    bool x2 = false;     // the block's four GEMMs on the fp16 matrix cores (two-plane operands, gemm_f16x2.hip), fp32 results
};

// mode 3 only: `xn_ready` = the planes of norm1(x_in) already lie in xn16 (written by the previous block's w_2 epilogue);
// `next` = the block whose norm1 this block's w_2 epilogue should apply (nullptr: none follows directly)
static int encoder_block(Encoder* e, const EncLayerW& w, float* x_in, int ld_in, float* x, int B, int T,
                         hipStream_t s, const EncChunkCtx* cc = nullptr, bool xn_ready = false, const EncLayerW* next = nullptr) {
    // EncoderLayerSANM.forward (sanm/encoder.py:72-148), normalize_before, no concat_after
    const pf_encoder_config& c = e->cfg;
    const int M = (e->cur_offs && !cc) ? e->cur_M : B * T, D = c.d_model, F = c.ffn_dim;
    float* xn = e->xn.as<float>();
    float* qkv = e->qkv.as<float>();
    float* mem = e->mem.as<float>();
    float* ctx = e->ctx.as<float>();
    float* ffn = e->ffn.as<float>();
    const int* lens = cc ? cc->lens : e->lens.as<int>();
    int rc;
    if (e->precision != 0 && D / c.n_heads != 128) { set_error("encoder: the bf16 / bf16x3 / f16x2 modes need d_model / n_heads == 128"); return -1; }
    if (e->precision == 1 && !cc) {
        // ---- bf16-operand mode: LN writes bf16, GEMMs and attention take bf16 operands with fp32 accumulation, the
        //      residual stream x, the FSMN memory and every epilogue stay fp32
        unsigned short* xn16 = e->xn16.as<unsigned short>();
        unsigned short* qkv16 = e->qkv16.as<unsigned short>();
        unsigned short* ctx16 = e->ctx16.as<unsigned short>();
        unsigned short* ffn16 = e->ffn16.as<unsigned short>();
        auto gemm16 = [&](const unsigned short* A, int lda, const unsigned short* W, int ldw, const float* bias, void* C,
                          int ldc, int N, int K, int relu, const float* R1, int ldr1, const float* R2, int ldr2, int c16) {
            GemmArgs g{};
            g.A = reinterpret_cast<const float*>(A); g.lda = lda; g.W = reinterpret_cast<const float*>(W); g.ldw = ldw;
            g.bias = bias; g.R1 = R1; g.ldr1 = ldr1; g.R2 = R2; g.ldr2 = ldr2; g.C = reinterpret_cast<float*>(C); g.ldc = ldc;
            g.M = M; g.N = N; g.K = K; g.relu = relu; g.ab_bf16 = 1; g.c_bf16 = c16;
            ProfScope ps(PROF_GEMM, 2.0 * M * (double)N * K, s);
            return launch_gemm_f32(g, s);
        };
        {
            ProfScope ps(PROF_LN, 6.0 * M * (double)w.in_dim, s);
            if ((rc = launch_layernorm(x_in, ld_in, w.n1g, w.n1b, reinterpret_cast<float*>(xn16), w.in_pad, M, w.in_dim,
                                       w.in_pad, c.ln_eps, s, 1))) return rc;
        }
        if ((rc = gemm16(xn16, w.in_pad, w.qkv_w16, w.in_pad, w.qkv_b, qkv16, 3 * D, 3 * D, w.in_pad, 0, nullptr, 0, nullptr, 0, 1)))
            return rc;
        FsmnArgs fa{};
        fa.in = reinterpret_cast<const float*>(qkv16 + 2 * D); fa.ldin = 3 * D; fa.in_bf16 = 1; fa.w = w.fsmn_w; fa.R = nullptr;
        fa.out = mem; fa.ldo = D; fa.lens = lens; fa.B = B; fa.T = T; fa.C = D; fa.K = c.kernel_size;
        fa.left_pad = (c.kernel_size - 1) / 2 + (c.sanm_shift > 0 ? c.sanm_shift : 0);
        if ((rc = fsmn(fa, s))) return rc;
        AttnArgs aa{};
        aa.Q = reinterpret_cast<const float*>(qkv16); aa.ldq = 3 * D; aa.K = reinterpret_cast<const float*>(qkv16 + D);
        aa.ldk = 3 * D; aa.V = reinterpret_cast<const float*>(qkv16 + 2 * D); aa.ldv = 3 * D;
        aa.O = reinterpret_cast<float*>(ctx16); aa.ldo = D; aa.klens = lens; aa.B = B; aa.H = c.n_heads; aa.Tq = T; aa.Tk = T;
        aa.scale = powf((float)(D / c.n_heads), -0.5f);
        {
            ProfScope ps(PROF_ATTN, 4.0 * B * (double)T * T * D, s);
            if ((rc = launch_attention_bf16(aa, s))) return rc;
        }
        const float* resid16 = (w.in_dim == D) ? x_in : nullptr;
        if ((rc = gemm16(ctx16, D, w.out_w16, D, w.out_b, x, D, D, D, 0, mem, D, resid16, ld_in, 0))) return rc;
        {
            ProfScope ps(PROF_LN, 6.0 * M * (double)D, s);
            if ((rc = launch_layernorm(x, D, w.n2g, w.n2b, reinterpret_cast<float*>(xn16), D, M, D, D, c.ln_eps, s, 1))) return rc;
        }
        if ((rc = gemm16(xn16, D, w.w1_16, D, w.b1, ffn16, F, F, D, 1, nullptr, 0, nullptr, 0, 1))) return rc;
        return gemm16(ffn16, F, w.w2_16, F, w.b2, x, D, D, F, 0, nullptr, 0, x, D, 0);
    }
    if (e->precision == 2 && !cc) {
        // ---- fp32-accurate mode on the bf16 matrix cores: every GEMM operand is three bf16 planes (x = hi + mid + lo
        //      exactly), produced by LayerNorm, by the relu epilogue of w_1 and by one split pass over the attention
        //      output; FSMN, attention, residuals and LayerNorm statistics are the fp32 kernels of mode 0
        unsigned short* xn3 = e->xn16.as<unsigned short>();
        unsigned short* ctx3 = e->ctx16.as<unsigned short>();
        unsigned short* ffn3 = e->ffn16.as<unsigned short>();
        auto gemm3 = [&](const unsigned short* A, int lda, const unsigned short* W, const float* bias, float* C, int ldc,
                         unsigned short* C3, int N, int K, int relu, const float* R1, int ldr1, const float* R2, int ldr2) {
            Gemm3Args g{};
            g.A = A; g.lda = lda; g.a_plane = (size_t)M * lda; g.W = W; g.ldw = K; g.w_plane = (size_t)N * K;
            g.bias = bias; g.R1 = R1; g.ldr1 = ldr1; g.R2 = R2; g.ldr2 = ldr2; g.C = C; g.ldc = ldc;
            g.C3 = C3; g.ldc3 = N; g.c_plane = (size_t)M * N; g.M = M; g.N = N; g.K = K; g.relu = relu;
            ProfScope ps(PROF_GEMM3, 2.0 * M * (double)N * K, s);
            return launch_gemm_split3(g, s);
        };
        {
            ProfScope ps(PROF_LN, 10.0 * M * (double)w.in_dim, s);
            if ((rc = launch_layernorm(x_in, ld_in, w.n1g, w.n1b, reinterpret_cast<float*>(xn3), w.in_pad, M, w.in_dim,
                                       w.in_pad, c.ln_eps, s, 2, 0, (size_t)M * w.in_pad))) return rc;
        }
        if ((rc = gemm3(xn3, w.in_pad, w.qkv_w3, w.qkv_b, qkv, 3 * D, nullptr, 3 * D, w.in_pad, 0, nullptr, 0, nullptr, 0)))
            return rc;
        FsmnArgs fa{};
        fa.in = qkv + 2 * D; fa.ldin = 3 * D; fa.w = w.fsmn_w; fa.R = nullptr; fa.ldr = 0; fa.out = mem; fa.ldo = D;
        fa.lens = lens; fa.B = B; fa.T = T; fa.C = D; fa.K = c.kernel_size;
        fa.left_pad = (c.kernel_size - 1) / 2 + (c.sanm_shift > 0 ? c.sanm_shift : 0);
        if ((rc = fsmn(fa, s))) return rc;
        AttnArgs aa{};
        aa.Q = qkv; aa.ldq = 3 * D; aa.K = qkv + D; aa.ldk = 3 * D; aa.V = qkv + 2 * D; aa.ldv = 3 * D;
        aa.O = nullptr; aa.O3 = ctx3; aa.o_plane = (size_t)M * D; aa.ldo = D;       // straight into the out-projection's planes
        aa.klens = lens; aa.B = B; aa.H = c.n_heads; aa.Tq = T; aa.Tk = T;
        aa.scale = powf((float)(D / c.n_heads), -0.5f);
        if ((rc = attention(aa, 4.0 * B * (double)T * T * D, s, true))) return rc;
        const float* resid3 = (w.in_dim == D) ? x_in : nullptr;
        if ((rc = gemm3(ctx3, D, w.out_w3, w.out_b, x, D, nullptr, D, D, 0, mem, D, resid3, ld_in))) return rc;
        {
            ProfScope ps(PROF_LN, 10.0 * M * (double)D, s);
            if ((rc = launch_layernorm(x, D, w.n2g, w.n2b, reinterpret_cast<float*>(xn3), D, M, D, D, c.ln_eps, s, 2, 0,
                                       (size_t)M * D))) return rc;
        }
        if ((rc = gemm3(xn3, D, w.w1_3, w.b1, nullptr, 0, ffn3, F, D, 1, nullptr, 0, nullptr, 0))) return rc;
        return gemm3(ffn3, F, w.w2_3, w.b2, x, D, nullptr, D, F, 0, nullptr, 0, x, D);
    }
    if (e->precision == 3 && !cc) {
        // ---- fp32-accurate mode on the fp16 matrix cores, three products per result (gemm_f16x2.hip, attention_f16x2.hip):
        //      every GEMM / attention operand is two fp16 planes of the tensor times a power of two; T is padded to Tp
        //      (B here = sequences, T = Tp rows each); FSMN, residuals, LayerNorm statistics, softmax stay fp32
        unsigned short* xn2 = e->xn16.as<unsigned short>();
        unsigned short* ctx2 = e->ctx16.as<unsigned short>();
        unsigned short* ffn2 = e->ffn16.as<unsigned short>();
        unsigned short* q2 = e->q2.as<unsigned short>();
        unsigned short* k2 = e->k2.as<unsigned short>();
        unsigned short* vt2 = e->vt2.as<unsigned short>();
        const int ldvt = M + 64;
        float* vbuf = qkv;                                  // fp32 v projection [M, D] for the FSMN memory block
        auto gemm2 = [&](const unsigned short* A, int lda, int ea, const unsigned short* W, int ew, const float* bias, float* C,
                         int ldc, unsigned short* C2, int ec, int N, int K, int relu, const float* R1, int ldr1,
                         const float* R2, int ldr2) {
            Gemm2Args g{};
            g.A = A; g.lda = lda; g.a_plane = (size_t)M * lda; g.W = W; g.ldw = K; g.w_plane = (size_t)N * K;
            g.oscale = pow2f(-(ea + ew)); g.bias = bias; g.R1 = R1; g.ldr1 = ldr1; g.R2 = R2; g.ldr2 = ldr2;
            g.C = C; g.ldc = ldc; g.C2 = C2; g.ldc2 = N; g.c_plane = (size_t)M * N; g.cscale = pow2f(ec);
            g.M = M; g.N = N; g.K = K; g.relu = relu; g.tile = e->gemm_tile;
            ProfScope ps(PROF_GEMM3, 2.0 * M * (double)N * K, s, C2 ? "enc.w_1 (planes out)" : (K > D ? "enc.w_2" : "enc.linear_out"));
            return launch_gemm_f16x2(g, s);
        };
        const bool fuse = e->fuse_row && gemm_f16x2_row_applicable(D, D) && gemm_f16x2_row_applicable(D, F);
        auto gemm_row = [&](const unsigned short* A, int lda, int ea, const unsigned short* W, int ew, const float* bias, int K,
                            const float* R1, const float* R2, int ldr2, const float* lg, const float* lb, int ey,
                            const float* fs_v = nullptr) {
            GemmRowArgs g{};
            if (fs_v) {
                g.fs_v = fs_v; g.ldfv = D; g.fs_w = w.fsmn_w; g.fs_lo = e->cur_fs; g.fs_hi = e->cur_fs + e->cur_fs_groups;
            }
            g.A = A; g.lda = lda; g.a_plane = (size_t)M * lda; g.W = W; g.ldw = K; g.w_plane = (size_t)D * K;
            g.oscale = pow2f(-(ea + ew)); g.bias = bias; g.R1 = R1; g.ldr1 = D; g.R2 = R2; g.ldr2 = ldr2; g.C = x; g.ldc = D;
            g.ln_g = lg; g.ln_b = lb; g.ln_eps = c.ln_eps;
            if (lg) { g.Y2 = xn2; g.ldy2 = D; g.y_plane = (size_t)M * D; g.yscale = pow2f(ey); }
            g.M = M; g.N = D; g.K = K; g.a_nt = e->row_nt >= (K == D ? 1 : 2); g.block_rows = e->row_bm;
            ProfScope ps(PROF_GEMM3, 2.0 * M * (double)D * K, s, K > D ? "enc.w_2 row (+res +LN)" : "enc.linear_out row (+fsmn +res +LN)");
            return launch_gemm_f16x2_row(g, s);
        };
        if (!(fuse && xn_ready)) {
            ProfScope ps(PROF_LN, 8.0 * M * (double)w.in_dim, s);
            if ((rc = launch_layernorm(x_in, ld_in, w.n1g, w.n1b, reinterpret_cast<float*>(xn2), w.in_pad, M, w.in_dim,
                                       w.in_pad, c.ln_eps, s, 3, 0, (size_t)M * w.in_pad, pow2f(w.e_x1)))) return rc;
        }
        const float dk_scale = powf((float)(D / c.n_heads), -0.5f);
        {
            Gemm2Args g{};
            g.A = xn2; g.lda = w.in_pad; g.a_plane = (size_t)M * w.in_pad; g.W = w.qkv_w2; g.ldw = w.in_pad;
            g.w_plane = (size_t)3 * D * w.in_pad; g.oscale = pow2f(-(w.e_x1 + w.ew_qkv)); g.bias = w.qkv_b;
            g.C = vbuf; g.ldc = D; g.M = M; g.N = 3 * D; g.K = w.in_pad;
            g.qkv_D = D; g.Qp = q2; g.Kp = k2; g.qk_plane = (size_t)(M + 32) * D;
            g.VT = vt2; g.ldvt = ldvt; g.vt_plane = (size_t)D * ldvt;
            g.q_mul = dk_scale * pow2f(w.e_q); g.k_mul = pow2f(w.e_k); g.v_mul = pow2f(w.e_v); g.tile = e->gemm_tile;
            ProfScope ps(PROF_GEMM3, 2.0 * M * 3.0 * D * w.in_pad, s, "enc.qkv (Q,K,V^T planes out)");
            if ((rc = launch_gemm_f16x2(g, s))) return rc;
        }
        // FSMN memory on the fp32 v projection: inside linear_out's epilogue (gemm_f16x2_row.hip) or as its own launch
        const bool fs_fused = fuse && e->fsmn_fused && e->cur_fs && c.kernel_size == 11 && c.sanm_shift <= 0 && M % 16 == 0;
        if (!fs_fused) {
            FsmnArgs fa{};
            fa.in = vbuf; fa.ldin = D; fa.w = w.fsmn_w; fa.R = nullptr; fa.ldr = 0; fa.out = mem; fa.ldo = D;
            fa.lens = lens; fa.B = B; fa.T = T; fa.C = D; fa.K = c.kernel_size; fa.offs = e->cur_offs;
            fa.left_pad = (c.kernel_size - 1) / 2 + (c.sanm_shift > 0 ? c.sanm_shift : 0);
            if ((rc = fsmn(fa, s))) return rc;
        }
        {
            Attn2Args aa{};
            aa.qoffs = aa.koffs = e->cur_offs; aa.Tq = e->cur_offs ? T : 0;
            aa.Q = q2; aa.ldq = D; aa.q_plane = (size_t)(M + 32) * D; aa.K = k2; aa.ldk = D; aa.k_plane = (size_t)(M + 32) * D;
            aa.VT = vt2; aa.ldvt = ldvt; aa.vt_plane = (size_t)D * ldvt;
            aa.O = ctx2; aa.ldo = D; aa.o_plane = (size_t)M * D; aa.klens = lens; aa.B = B; aa.H = c.n_heads; aa.Tp = T;
            aa.sscale = pow2f(-(w.e_q + w.e_k)); aa.oscale = pow2f(-10);      // ctx planes carry v's exponent
            aa.variant = e->attn_variant;
            ProfScope ps(PROF_ATTN, 4.0 * B * (double)T * T * D, s, "enc.self_attention");
            if ((rc = launch_attention_f16x2(aa, s))) return rc;
        }
        const float* resid2 = (w.in_dim == D) ? x_in : nullptr;
        if (fuse) {
            // linear_out + fsmn memory + residual -> x, and norm2(x) -> the planes w_1 reads, in one launch
            if ((rc = gemm_row(ctx2, D, w.e_v, w.out_w2, w.ew_out, w.out_b, D, fs_fused ? nullptr : mem, resid2, ld_in, w.n2g, w.n2b,
                               w.e_x2, fs_fused ? vbuf : nullptr))) return rc;
        } else {
            if ((rc = gemm2(ctx2, D, w.e_v, w.out_w2, w.ew_out, w.out_b, x, D, nullptr, 0, D, D, 0, mem, D, resid2, ld_in))) return rc;
            ProfScope ps(PROF_LN, 8.0 * M * (double)D, s);
            if ((rc = launch_layernorm(x, D, w.n2g, w.n2b, reinterpret_cast<float*>(xn2), D, M, D, D, c.ln_eps, s, 3, 0,
                                       (size_t)M * D, pow2f(w.e_x2)))) return rc;
        }
        if (fuse && e->ffn_fused && ffn_f16x2_applicable(D, F) && (e->ffn_fused == 2 || ffn_fills_rounds(M))) {
            const bool ln = next && next->in_dim == D;
            FfnArgs g{};
            g.X2 = xn2; g.ldx = D; g.x_plane = (size_t)M * D; g.W1 = w.w1_2; g.ldw1 = D; g.w1_plane = (size_t)F * D;
            g.W2 = w.w2_2; g.ldw2 = F; g.w2_plane = (size_t)D * F; g.b1 = w.b1; g.b2 = w.b2;
            g.oscale1 = pow2f(-(w.e_x2 + w.ew_1)); g.hscale = pow2f(w.e_h); g.oscale2 = pow2f(-(w.e_h + w.ew_2));
            g.R = x; g.ldr = D; g.C = x; g.ldc = D;
            if (ln) {
                g.ln_g = next->n1g; g.ln_b = next->n1b; g.ln_eps = c.ln_eps;
                g.Y2 = xn2; g.ldy2 = D; g.y_plane = (size_t)M * D; g.yscale = pow2f(next->e_x1);
            }
            g.M = M; g.D = D; g.F = F; g.abl = e->ffn_abl;
            ProfScope ps(PROF_GEMM3, 2.0 * M * 2.0 * (double)D * F, s, "enc.ffn fused (w_1 +relu +w_2 +res +LN)");
            return launch_ffn_f16x2(g, s);
        }
        if ((rc = gemm2(xn2, D, w.e_x2, w.w1_2, w.ew_1, w.b1, nullptr, 0, ffn2, w.e_h, F, D, 1, nullptr, 0, nullptr, 0))) return rc;
        if (fuse) {
            // w_2 + residual -> x, and (when a block follows directly) its norm1(x) -> the planes its QKV projection reads
            const bool ln = next && next->in_dim == D;
            return gemm_row(ffn2, F, w.e_h, w.w2_2, w.ew_2, w.b2, F, nullptr, x, D, ln ? next->n1g : nullptr, ln ? next->n1b : nullptr,
                            ln ? next->e_x1 : 0);
        }
        return gemm2(ffn2, F, w.e_h, w.w2_2, w.ew_2, w.b2, x, D, nullptr, 0, D, F, 0, nullptr, 0, x, D);
    }
    // Streaming step in its f16x2 form (cc->x2, pf_stream_set_option "gemm_mode" 3): the block's four GEMMs take two-plane
    // fp16 operands with the a-priori exponents of the offline f16x2 mode (encoder_prepare_x2) and write fp32, so the FSMN,
    // the few-query attention over the K/V ring and the ring itself stay the fp32 kernels of the default step. The attention
    // output (a convex combination of v rows, ring rows included: the same projection of earlier frames) is bounded by v's bound.
    const bool x2c = cc && cc->x2;
    unsigned short* xn2c = e->xn16.as<unsigned short>();
    unsigned short* ctx2c = e->ctx16.as<unsigned short>();
    unsigned short* ffn2c = e->ffn16.as<unsigned short>();
    auto gemm2c = [&](const unsigned short* A, int lda, int ea, const unsigned short* W, int ew, const float* bias, float* C, int ldc,
                      unsigned short* C2, int ec, int N, int K, int relu, const float* R1, int ldr1, const float* R2, int ldr2) {
        Gemm2Args g{};
        g.A = A; g.lda = lda; g.a_plane = (size_t)M * lda; g.W = W; g.ldw = K; g.w_plane = (size_t)N * K;
        g.oscale = pow2f(-(ea + ew)); g.bias = bias; g.R1 = R1; g.ldr1 = ldr1; g.R2 = R2; g.ldr2 = ldr2;
        g.C = C; g.ldc = ldc; g.C2 = C2; g.ldc2 = N; g.c_plane = (size_t)M * N; g.cscale = pow2f(ec);
        g.M = M; g.N = N; g.K = K; g.relu = relu;
        // the long-K projection (w_2) of a step is at most a block per CU: always in its split-K form here (by caller, whatever
        // the stream count, so a stream's result does not depend on its neighbours)
        if (C && K >= 4 * D && K % 128 == 0 && e->splitk.p) { g.ksplit = 4; g.part = e->splitk.as<float>(); }
        ProfScope ps(PROF_GEMM3, 2.0 * M * (double)N * K, s);
        return launch_gemm_f16x2(g, s);
    };
    auto ln_planes = [&](const float* src, int ld, const float* g, const float* b, int dim, int dim_pad, int ex) {
        ProfScope ps(PROF_LN, 8.0 * M * (double)dim, s);
        return launch_layernorm(src, ld, g, b, reinterpret_cast<float*>(xn2c), dim_pad, M, dim, dim_pad, c.ln_eps, s, 3, 0,
                                (size_t)M * dim_pad, pow2f(ex));
    };
    if (x2c && (!w.qkv_w2 || !w.out_w2 || !w.w1_2 || !w.w2_2)) { set_error("encoder: streaming f16x2 step without prepared weight planes"); return -1; }
    // norm1 -> fused QKV projection
    if (x2c) {
        if ((rc = ln_planes(x_in, ld_in, w.n1g, w.n1b, w.in_dim, w.in_pad, w.e_x1))) return rc;
        if ((rc = gemm2c(xn2c, w.in_pad, w.e_x1, w.qkv_w2, w.ew_qkv, w.qkv_b, qkv, 3 * D, nullptr, 0, 3 * D, w.in_pad, 0, nullptr, 0,
                         nullptr, 0))) return rc;
    } else {
        if ((rc = layernorm(x_in, ld_in, w.n1g, w.n1b, xn, w.in_pad, M, w.in_dim, w.in_pad, c.ln_eps, s))) return rc;
        if ((rc = gemm_simple(xn, w.in_pad, w.qkv_w, w.in_pad, w.qkv_b, qkv, 3 * D, M, 3 * D, w.in_pad, 0, nullptr, 0,
                              nullptr, 0, s))) return rc;
    }
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
    if (e->cur_mask_mode && !cc) {
        if (D / c.n_heads > 64) { set_error("encoder: the causal / VAD masks are built for heads of d_k <= 64 (attention_small.hip)"); return -1; }
        aa.mask_mode = e->cur_mask_mode; aa.vad_pos = e->vad_dev.as<int>();
    }
    if (cc && cc->cap > 0) {
        // keys = [ring rows 0 .. enc_valid) | this window's K/V], no padding mask (forward_chunk passes mask=None)
        aa.K = cc->ring; aa.ldk = 2 * D; aa.V = cc->ring + D; aa.ldv = 2 * D; aa.Tk = cc->cap;
        aa.K2 = qkv + D; aa.ldk2 = 3 * D; aa.V2 = qkv + 2 * D; aa.ldv2 = 3 * D; aa.T2 = T; aa.n2 = T;
        aa.n1_dev = &cc->st->enc_valid; aa.n1_stride = 0;
    }
    bool appended = false;
    if (cc && cc->cap > 0 && cc->append_rows > 0) {
        aa.app_rows = cc->append_rows; aa.app_r0 = 0; aa.app_wp = &cc->st->enc_wp; aa.app_wp_stride = 0; aa.app_gate = nullptr;
    }
    if ((rc = attention(aa, 4.0 * B * (double)T * T * D, s, false, D / c.n_heads, &appended))) return rc;
    if (cc && cc->cap > 0 && cc->append_rows > 0 && !appended) {
        RingAppendArgs ra{};
        ra.src = qkv + D; ra.ldsrc = 3 * D; ra.src_T = T; ra.r0 = 0; ra.rows = cc->append_rows; ra.cols = 2 * D;
        ra.ring = cc->ring; ra.cap = cc->cap; ra.S = B; ra.st = cc->st;
        if ((rc = launch_ring_append(ra, s))) return rc;
    }
    // out projection + fsmn memory (+ residual when in == out, encoder.py:120-137)
    const float* resid = (w.in_dim == D) ? x_in : nullptr;
    if (x2c) {
        if ((rc = launch_split2(ctx, D, ctx2c, D, (size_t)M * D, M, D, pow2f(w.e_v), s))) return rc;
        if ((rc = gemm2c(ctx2c, D, w.e_v, w.out_w2, w.ew_out, w.out_b, x, D, nullptr, 0, D, D, 0, mem, D, resid, ld_in))) return rc;
        // norm2 -> FFN -> residual: w_1 hands its relu output to w_2 as planes, like the offline mode
        if ((rc = ln_planes(x, D, w.n2g, w.n2b, D, D, w.e_x2))) return rc;
        if ((rc = gemm2c(xn2c, D, w.e_x2, w.w1_2, w.ew_1, w.b1, nullptr, 0, ffn2c, w.e_h, F, D, 1, nullptr, 0, nullptr, 0))) return rc;
        return gemm2c(ffn2c, F, w.e_h, w.w2_2, w.ew_2, w.b2, x, D, nullptr, 0, D, F, 0, nullptr, 0, x, D);
    }
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
    // CifPredictorV3 (bicif_paraformer/cif_predictor.py:121-384): sequential fp32 CIF + the upsampled timestamp head
    bool v3 = false;
    pf_predictor_v3_config c3{};
    DevBuf curs, ntok, up, x_tm, pre, lstm_out, h_a, h_b, cell, tok_dev, ulens, pack;
    bool packed = false;                 // pack = both directions' re-laid weight_hh, then bias_ih, bias_hh back to back
    std::vector<int32_t> ul_host;
};

// ================================================================================================= decoder
struct DecLayerW {
    const float *n1g, *n1b, *w1, *b1, *fng, *fnb, *w2, *n2g, *n2b, *fsmn_w, *n3g, *n3b, *q_w, *q_b, *kv_w, *kv_b,
        *o_w, *o_b;
    // f16x2 mode: weight planes + exponents, and the exponents of the LayerNorm-output planes (from gamma / beta)
    const unsigned short *w1_2 = nullptr, *w2_2 = nullptr, *q_2 = nullptr, *kv_2 = nullptr, *o_2 = nullptr;
    int ew_1 = 0, ew_2 = 0, ew_q = 0, ew_kv = 0, ew_o = 0, e_n1 = 0, e_fn = 0, e_n3 = 0, e_q = 0;
    float kv_l1b[4] = {0.f, 0.f, 0.f, 0.f};      // max row L1 norm and max |bias| of the k half, then of the v half, of linear_k_v
    bool x2_ready = false;
};
// ContextualParaformerDecoder keeps its last attention block under "last_decoder." (contextual_paraformer/decoder.py:241)
static std::string dec_layer_prefix(bool contextual, int n_blocks, int i) {
    return (contextual && i == n_blocks - 1) ? std::string("last_decoder.") : "decoders." + std::to_string(i) + ".";
}

struct Decoder {
    pf_decoder_config cfg;
    bool contextual = false;
    DevBuf xself, xcat, ctx_lens;     // contextual: x after the FSMN residual, [x_src_attn | cx] rows, hotword counts
    TensorTable tt;
    std::vector<DecLayerW> layers;
    DecLayerW last;          // decoders3.0 (FFN only)
    bool resolved = false;
    DevBuf x, t1, t2, ffn, ffn2, q, kv, ctx, mem_lens, tok_lens, pval, pidx, hid;
    int precision = 0;       // 0 fp32, 1 bf16 operands (GEMMs + cross-attention), fp32 residual / LN statistics / FSMN
    DevBuf t16, ffn16, ffn2_16, q16, kv16, ctx16, mem16, hid16;
    DevBuf dsc;              // f16x2 mode: [amax(memory), 2^e, 2^-e] chosen on the device per forward
    DevBuf dscl, dlb;        // per layer {k_mul, v_mul, 1/k_mul, 1/v_mul} (device-chosen) and the constants they come from
    DevBuf k2, vt2;          // cross-attention operands written by the KV form of linear_k_v (attention_f16x2.hip)
    DevBuf splitk;           // streaming f16x2 step: split-K partials of the FFN's w_2
    bool lb_uploaded = false;
    int e_an = INT32_MIN;    // exponent of the after_norm output planes (f16x2 vocabulary projection)
    DevBuf asf_p;            // SeACo score filter: attention probabilities of sequence 0 [H, N, T]
    // token packing (f16x2 greedy route): row offsets per sequence, packed row -> padded row map, packed ids
    DevBuf offs_dev, map_dev, ids_packed;
    std::vector<int32_t> h_offs, h_map;
};

static int decoder_resolve(Decoder* d) {
    std::string first;
    const int miss = d->tt.missing(&first);
    if (miss) { set_error("decoder: " + std::to_string(miss) + " tensors not set, e.g. " + first); return -3; }
    d->layers.clear();
    for (int i = 0; i < d->cfg.n_blocks; ++i) {
        const std::string p = dec_layer_prefix(d->contextual, d->cfg.n_blocks, i);
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
    d->lb_uploaded = false;
    d->e_an = INT32_MIN;
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
    if (g_stream_mode && D % 16 == 0) {
        // small M (streaming): weight-streaming GEMM into a scratch logits block, then a row arg-max
        if (pval.ensure(sizeof(float) * (size_t)M * V)) return -2;
        if ((rc = gemm_simple(hidden, D, W, D, bias, pval.as<float>(), V, M, V, D, 0, nullptr, 0, nullptr, 0, s))) return rc;
        return launch_argmax_rows(pval.as<float>(), V, M, V, ids, s);
    }
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
    int precision = 0;       // 0 fp32 MFMA, 3 f16x2 arg-max route
    DevBuf h2, dsc;          // f16x2: planes of the hidden states, [amax, 2^e, 2^-e]
};


// PositionwiseFeedForwardDecoderSANM (sanm/positionwise_feed_forward.py:12-33): w_2(LN(relu(w_1 x))), w_2 bias-free
static int gemm3_simple(const unsigned short* A3, int lda, int M, const unsigned short* W3, const float* bias, float* C,
                        int ldc, int N, int K, int relu, hipStream_t s) {
    if (!W3) return -2;
    Gemm3Args g{};
    g.A = A3; g.lda = lda; g.a_plane = (size_t)M * lda; g.W = W3; g.ldw = K; g.w_plane = (size_t)N * K;
    g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.relu = relu;
    ProfScope ps(PROF_GEMM3, 2.0 * M * (double)N * K, s);
    return launch_gemm_split3(g, s);
}

static int gemm2_simple(const unsigned short* A2, int lda, int M, int ea, const unsigned short* W2, int ew, const float* bias,
                        float* C, int ldc, int N, int K, int relu, const float* R2, int ldr2, hipStream_t s,
                        const float* oscale_dev = nullptr, float* splitk_part = nullptr) {
    if (!W2) return -2;
    Gemm2Args g{};
    g.A = A2; g.lda = lda; g.a_plane = (size_t)M * lda; g.W = W2; g.ldw = K; g.w_plane = (size_t)N * K;
    g.oscale = pow2f(-(ea + ew)); g.oscale_dev = oscale_dev; g.bias = bias; g.R2 = R2; g.ldr2 = ldr2;
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.relu = relu;
    if (splitk_part && K % 128 == 0) { g.ksplit = 4; g.part = splitk_part; }     // streaming step: w_2 in its split-K form
    ProfScope ps(PROF_GEMM3, 2.0 * M * (double)N * K, s);
    return launch_gemm_f16x2(g, s);
}

// f16x2 mode: exponents and weight planes of one decoder layer (once)
static int dec_layer_x2(Decoder* d, DecLayerW& w, const std::string& p, bool attn, hipStream_t s) {
    if (w.x2_ready) return 0;
    const int D = d->cfg.d_model, F = d->cfg.ffn_dim;
    float g, b;
    if (TensorTable::dev_absmax(w.n1g, D, &g, s) || TensorTable::dev_absmax(w.n1b, D, &b, s)) return -2;
    w.e_n1 = exp_for_bound(sqrtf((float)D) * g + b);
    if (TensorTable::dev_absmax(w.fng, F, &g, s) || TensorTable::dev_absmax(w.fnb, F, &b, s)) return -2;
    w.e_fn = exp_for_bound(sqrtf((float)F) * g + b);
    w.w1_2 = d->tt.get_split2(p + "feed_forward.w_1.weight", F, D, &w.ew_1, s);
    w.w2_2 = d->tt.get_split2(p + "feed_forward.w_2.weight", D, F, &w.ew_2, s);
    if (!w.w1_2 || !w.w2_2) return -2;
    if (attn) {
        if (TensorTable::dev_absmax(w.n3g, D, &g, s) || TensorTable::dev_absmax(w.n3b, D, &b, s)) return -2;
        w.e_n3 = exp_for_bound(sqrtf((float)D) * g + b);
        w.q_2 = d->tt.get_split2(p + "src_attn.linear_q.weight", D, D, &w.ew_q, s);
        w.kv_2 = d->tt.get_split2(p + "src_attn.linear_k_v.weight", 2 * D, D, &w.ew_kv, s);
        w.o_2 = d->tt.get_split2(p + "src_attn.linear_out.weight", D, D, &w.ew_o, s);
        if (!w.q_2 || !w.kv_2 || !w.o_2) return -2;
        float bq;
        if (TensorTable::dev_linear_bound(w.q_w, D, D, D, w.q_b, sqrtf((float)D) * g + b, &bq, s)) return -2;
        w.e_q = exp_for_bound(bq * powf((float)(D / d->cfg.n_heads), -0.5f));
        if (TensorTable::dev_linear_bound(w.kv_w, D, D, D, nullptr, 1.f, &w.kv_l1b[0], s) ||
            TensorTable::dev_absmax(w.kv_b, D, &w.kv_l1b[1], s) ||
            TensorTable::dev_linear_bound(w.kv_w + (size_t)D * D, D, D, D, nullptr, 1.f, &w.kv_l1b[2], s) ||
            TensorTable::dev_absmax(w.kv_b + D, D, &w.kv_l1b[3], s)) return -2;
    }
    w.x2_ready = true;
    return 0;
}

// f16x2 form of dec_ffn: both LayerNorms write two-plane fp16 operands, w_1 and w_2 run on the fp16 matrix cores
static int dec_ffn_x2(Decoder* d, const DecLayerW& w, const float* x, float* out, int M, hipStream_t s, float* splitk_part = nullptr) {
    const int D = d->cfg.d_model, F = d->cfg.ffn_dim;
    float* ffn = d->ffn.as<float>();
    unsigned short* t2p = d->t16.as<unsigned short>();
    unsigned short* f2p = d->ffn16.as<unsigned short>();
    int rc;
    {
        ProfScope ps(PROF_LN, 8.0 * M * (double)D, s);
        if ((rc = launch_layernorm(x, D, w.n1g, w.n1b, reinterpret_cast<float*>(t2p), D, M, D, D, d->cfg.ln_eps, s, 3, 0,
                                   (size_t)M * D, pow2f(w.e_n1)))) return rc;
    }
    if ((rc = gemm2_simple(t2p, D, M, w.e_n1, w.w1_2, w.ew_1, w.b1, ffn, F, F, D, 1, nullptr, 0, s))) return rc;
    {
        ProfScope ps(PROF_LN, 8.0 * M * (double)F, s);
        if ((rc = launch_layernorm(ffn, F, w.fng, w.fnb, reinterpret_cast<float*>(f2p), F, M, F, F, d->cfg.ln_eps, s, 3, 0,
                                   (size_t)M * F, pow2f(w.e_fn)))) return rc;
    }
    return gemm2_simple(f2p, F, M, w.e_fn, w.w2_2, w.ew_2, nullptr, out, D, D, F, 0, nullptr, 0, s, nullptr, splitk_part);
}

// w1_3 != nullptr (bf16x3 mode): norm1 writes the three planes and w_1 runs on the bf16 matrix cores
static int dec_ffn(Decoder* d, const DecLayerW& w, const float* x, float* out, int M, hipStream_t s,
                   const unsigned short* w1_3 = nullptr) {
    const int D = d->cfg.d_model, F = d->cfg.ffn_dim;
    float* t1 = d->t1.as<float>();
    float* ffn = d->ffn.as<float>();
    float* ffn2 = d->ffn2.as<float>();
    int rc;
    if (w1_3) {
        unsigned short* t3 = d->t16.as<unsigned short>();
        {
            ProfScope ps(PROF_LN, 10.0 * M * (double)D, s);
            if ((rc = launch_layernorm(x, D, w.n1g, w.n1b, reinterpret_cast<float*>(t3), D, M, D, D, d->cfg.ln_eps, s, 2, 0,
                                       (size_t)M * D))) return rc;
        }
        if ((rc = gemm3_simple(t3, D, M, w1_3, w.b1, ffn, F, F, D, 1, s))) return rc;
        if ((rc = layernorm(ffn, F, w.fng, w.fnb, ffn2, F, M, F, F, d->cfg.ln_eps, s))) return rc;
        return gemm_simple(ffn2, F, w.w2, F, nullptr, out, D, M, D, F, 0, nullptr, 0, nullptr, 0, s);
    }
    if ((rc = layernorm(x, D, w.n1g, w.n1b, t1, D, M, D, D, d->cfg.ln_eps, s))) return rc;
    if ((rc = gemm_simple(t1, D, w.w1, D, w.b1, ffn, F, M, F, D, 1, nullptr, 0, nullptr, 0, s))) return rc;
    if ((rc = layernorm(ffn, F, w.fng, w.fnb, ffn2, F, M, F, F, d->cfg.ln_eps, s))) return rc;
    return gemm_simple(ffn2, F, w.w2, F, nullptr, out, D, M, D, F, 0, nullptr, 0, nullptr, 0, s);
}


// bf16-operand decoder (throughput mode): every GEMM and the cross-attention take bf16 operands with fp32
// accumulation; the token stream x, the FSMN and the LayerNorm statistics stay fp32. Expects x = embeds and the
// length arrays already staged by pf_decoder_forward.
static int decoder_forward_bf16(Decoder* d, const float* memory, int B, int T, int N, int32_t* ids, float* hidden_out,
                                hipStream_t s) {
    const pf_decoder_config& c = d->cfg;
    const int D = c.d_model, F = c.ffn_dim, V = c.vocab_size, Mq = B * N, Mk = B * T;
    typedef unsigned short u16;
    if (d->t16.ensure(sizeof(u16) * (size_t)Mq * D) || d->ffn16.ensure(sizeof(u16) * (size_t)Mq * F) ||
        d->ffn2_16.ensure(sizeof(u16) * (size_t)Mq * F) || d->q16.ensure(sizeof(u16) * (size_t)Mq * D) ||
        d->kv16.ensure(sizeof(u16) * (size_t)Mk * 2 * D) || d->ctx16.ensure(sizeof(u16) * (size_t)Mq * D) ||
        d->mem16.ensure(sizeof(u16) * (size_t)Mk * D) || d->hid16.ensure(sizeof(u16) * (size_t)Mq * D))
        return -2;
    u16* t16 = d->t16.as<u16>(); u16* ffn16 = d->ffn16.as<u16>(); u16* ffn2_16 = d->ffn2_16.as<u16>();
    u16* q16 = d->q16.as<u16>(); u16* kv16 = d->kv16.as<u16>(); u16* ctx16 = d->ctx16.as<u16>();
    u16* mem16 = d->mem16.as<u16>();
    float* x = d->x.as<float>(); float* t1 = d->t1.as<float>(); float* t2 = d->t2.as<float>();
    int rc;
    if ((rc = launch_cast_bf16(memory, mem16, (size_t)Mk * D, s))) return rc;
    auto w16 = [&](const std::string& name) { return d->tt.get_bf16(name, s); };
    auto gemm16 = [&](const u16* A, int lda, const u16* W, int ldw, const float* bias, void* C, int ldc, int M, int Nn, int K,
                      int relu, const float* R2, int ldr2, int c16) {
        if (!W) return -2;
        GemmArgs g{};
        g.A = reinterpret_cast<const float*>(A); g.lda = lda; g.W = reinterpret_cast<const float*>(W); g.ldw = ldw;
        g.bias = bias; g.R2 = R2; g.ldr2 = ldr2; g.C = reinterpret_cast<float*>(C); g.ldc = ldc;
        g.M = M; g.N = Nn; g.K = K; g.relu = relu; g.ab_bf16 = 1; g.c_bf16 = c16;
        ProfScope ps(PROF_GEMM, 2.0 * M * (double)Nn * K, s);
        return launch_gemm_f32(g, s);
    };
    auto ffn_bf16 = [&](const std::string& p, const DecLayerW& w, const float* xin, float* out) {
        int r;
        if ((r = launch_layernorm(xin, D, w.n1g, w.n1b, reinterpret_cast<float*>(t16), D, Mq, D, D, c.ln_eps, s, 1, 0))) return r;
        if ((r = gemm16(t16, D, w16(p + "feed_forward.w_1.weight"), D, w.b1, ffn16, F, Mq, F, D, 1, nullptr, 0, 1))) return r;
        if ((r = launch_layernorm(reinterpret_cast<const float*>(ffn16), F, w.fng, w.fnb, reinterpret_cast<float*>(ffn2_16), F,
                                  Mq, F, F, c.ln_eps, s, 1, 1))) return r;
        return gemm16(ffn2_16, F, w16(p + "feed_forward.w_2.weight"), F, nullptr, out, D, Mq, D, F, 0, nullptr, 0, 0);
    };
    const int left_pad = (c.kernel_size - 1) / 2 + (c.sanm_shift > 0 ? c.sanm_shift : 0);
    for (int l = 0; l < c.n_blocks; ++l) {
        const DecLayerW& w = d->layers[l];
        const std::string p = "decoders." + std::to_string(l) + ".";
        if ((rc = ffn_bf16(p, w, x, t2))) return rc;
        if ((rc = layernorm(t2, D, w.n2g, w.n2b, t1, D, Mq, D, D, c.ln_eps, s))) return rc;
        FsmnArgs fa{};
        fa.in = t1; fa.ldin = D; fa.w = w.fsmn_w; fa.R = x; fa.ldr = D; fa.out = x; fa.ldo = D;
        fa.lens = d->tok_lens.as<int>(); fa.B = B; fa.T = N; fa.C = D; fa.K = c.kernel_size; fa.left_pad = left_pad;
        if ((rc = fsmn(fa, s))) return rc;
        if ((rc = launch_layernorm(x, D, w.n3g, w.n3b, reinterpret_cast<float*>(t16), D, Mq, D, D, c.ln_eps, s, 1, 0))) return rc;
        if ((rc = gemm16(t16, D, w16(p + "src_attn.linear_q.weight"), D, w.q_b, q16, D, Mq, D, D, 0, nullptr, 0, 1))) return rc;
        if ((rc = gemm16(mem16, D, w16(p + "src_attn.linear_k_v.weight"), D, w.kv_b, kv16, 2 * D, Mk, 2 * D, D, 0, nullptr, 0, 1)))
            return rc;
        AttnArgs aa{};
        aa.Q = reinterpret_cast<const float*>(q16); aa.ldq = D; aa.K = reinterpret_cast<const float*>(kv16); aa.ldk = 2 * D;
        aa.V = reinterpret_cast<const float*>(kv16 + D); aa.ldv = 2 * D; aa.O = reinterpret_cast<float*>(ctx16); aa.ldo = D;
        aa.klens = d->mem_lens.as<int>(); aa.B = B; aa.H = c.n_heads; aa.Tq = N; aa.Tk = T;
        aa.scale = powf((float)(D / c.n_heads), -0.5f);
        {
            ProfScope ps(PROF_ATTN, 4.0 * B * (double)N * T * D, s);
            if ((rc = launch_attention_bf16(aa, s))) return rc;
        }
        if ((rc = gemm16(ctx16, D, w16(p + "src_attn.linear_out.weight"), D, w.o_b, x, D, Mq, D, D, 0, x, D, 0))) return rc;
    }
    if ((rc = ffn_bf16("decoders3.0.", d->last, x, t2))) return rc;
    u16* hid16 = d->hid16.as<u16>();
    if (hidden_out) {
        if ((rc = layernorm(t2, D, d->tt.get("after_norm.weight"), d->tt.get("after_norm.bias"), hidden_out, D, Mq, D, D,
                            c.ln_eps, s))) return rc;
    }
    if (!ids) return 0;
    if ((rc = launch_layernorm(t2, D, d->tt.get("after_norm.weight"), d->tt.get("after_norm.bias"),
                               reinterpret_cast<float*>(hid16), D, Mq, D, D, c.ln_eps, s, 1, 0))) return rc;
    const u16* ow = w16("output_layer.weight");
    if (!ow) return -2;
    const int nparts = 2 * ceil_div(V, 128);
    if (d->pval.ensure(sizeof(float) * (size_t)Mq * nparts) || d->pidx.ensure(sizeof(int) * (size_t)Mq * nparts)) return -2;
    GemmArgs g{};
    g.A = reinterpret_cast<const float*>(hid16); g.lda = D; g.W = reinterpret_cast<const float*>(ow); g.ldw = D;
    g.bias = d->tt.get("output_layer.bias"); g.M = Mq; g.N = V; g.K = D; g.ab_bf16 = 1;
    g.amax_val = d->pval.as<float>(); g.amax_idx = d->pidx.as<int>(); g.amax_ld = nparts;
    {
        ProfScope ps(PROF_GEMM, 2.0 * Mq * (double)V * D, s);
        if ((rc = launch_gemm_f32(g, s))) return rc;
    }
    return launch_argmax_reduce(d->pval.as<float>(), d->pidx.as<int>(), nparts, nparts, ids, nullptr, Mq, s);
}

// ================================================================================================ streaming
// A lock-step batch of S independent streams (the reference handles exactly one: "batch_size must be set 1",
// paraformer_streaming/model.py:705). All per-stream state lives in HBM; the steady-state step is captured in a
// hipGraph keyed by (n_frames, is_final, tail_chunk) and replayed.
struct Stream {
    Encoder* e = nullptr; Predictor* p = nullptr; Decoder* d = nullptr;
    pf_stream_config cfg{};
    int S = 1, keep = 5, Wmax = 0, Nmax = 0, enc_cap = 0, dec_cap = 0, pe_rows = 0;
    DevBuf dev_state, cache_feats, feats_in, win, enc_ring, dec_ring, dec_fsmn, cif_hidden, cif_alpha, dec_valid, dec_wp,
        n_fired, pe, lensW, enc_out, embeds, ids, alphas;
    int32_t* h_ids = nullptr; int32_t* h_n = nullptr;       // pinned
    hipStream_t stream = nullptr;                            // the step runs (and is captured) on its own stream
    hipEvent_t ev = nullptr;
    int start_idx = 0;                                       // host mirror of StreamDev.start_idx
    std::map<int, hipGraphExec_t> graphs;
    std::map<int, int> seen;
    unsigned long long graph_epoch = 0;                      // g_ws_epoch the graphs were captured under
    bool use_graph = true;
    // gemm_mode 3 (pf_stream_set_option): every GEMM of the step on the fp16 matrix cores with two-plane operands
    // (gemm_f16x2.hip; fp32 results, fp32-class accuracy like the offline f16x2 mode); attention, FSMN, CIF, the K/V rings and
    // every LayerNorm statistic stay the fp32 kernels of the default step. Exponents come from a-priori bounds: the decoder's
    // memory is THIS encoder's after_norm output (|y| <= sqrt(D) max|gamma| + max|beta|), so nothing is chosen per step.
    bool x2 = false;
    unsigned long long ver_e = ~0ull, ver_d = ~0ull;         // TensorTable versions the prepared exponents / planes belong to
    int e_mem = 0, e_an = 0;
    std::vector<int> e_ctx;                                  // per decoder layer: exponent of the cross-attention output planes
    DevBuf mem2;                                             // planes of the step's encoder output [2][S * Wmax, D]
    ~Stream() {
        for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
        if (h_ids) (void)hipHostFree(h_ids);
        if (h_n) (void)hipHostFree(h_n);
        if (ev) (void)hipEventDestroy(ev);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

static int stream_reset(Stream* st, hipStream_t s) {
    auto zero = [&](DevBuf& b) -> int {
        if (b.p && b.cap) PF_HIP_TRY(hipMemsetAsync(b.p, 0, b.cap, s));
        return 0;
    };
    int rc = 0;
    rc |= zero(st->dev_state); rc |= zero(st->cache_feats); rc |= zero(st->enc_ring); rc |= zero(st->dec_ring);
    rc |= zero(st->dec_fsmn); rc |= zero(st->cif_hidden); rc |= zero(st->cif_alpha); rc |= zero(st->dec_valid);
    rc |= zero(st->dec_wp); rc |= zero(st->n_fired);
    st->start_idx = 0;
    return rc ? -2 : 0;
}

// gemm_mode 3: weight planes and exponents of both handles (load-time reductions with host round trips: outside any capture)
static int stream_prepare_x2(Stream* st, hipStream_t s) {
    Encoder* e = st->e; Decoder* d = st->d;
    int rc;
    if (!e->resolved && (rc = encoder_resolve(e))) return rc;
    if (!d->resolved && (rc = decoder_resolve(d))) return rc;
    const pf_decoder_config& dc = d->cfg;
    const int D = e->cfg.d_model;
    if (D / e->cfg.n_heads != 128 || D % 256 != 0 || e->cfg.ffn_dim % 256 != 0 || dc.ffn_dim % 256 != 0 || dc.d_model != D ||
        dc.vocab_size <= 0) {
        set_error("stream: gemm_mode 3 (f16x2) needs d_model / n_heads == 128, d_model % 256 == 0, ffn_dim % 256 == 0");
        return -1;
    }
    if ((rc = encoder_prepare_x2(e, s))) return rc;
    for (int l = 0; l < dc.n_blocks; ++l)
        if ((rc = dec_layer_x2(d, d->layers[l], dec_layer_prefix(d->contextual, dc.n_blocks, l), true, s))) return rc;
    if ((rc = dec_layer_x2(d, d->last, "decoders3.0.", false, s))) return rc;
    float g, b;
    if (TensorTable::dev_absmax(e->tt.get("after_norm.weight"), D, &g, s) || TensorTable::dev_absmax(e->tt.get("after_norm.bias"), D, &b, s)) return -2;
    const float bmem = sqrtf((float)D) * g + b;
    st->e_mem = exp_for_bound(bmem);
    st->e_ctx.assign((size_t)dc.n_blocks, 0);
    for (int l = 0; l < dc.n_blocks; ++l)      // |attention output| <= max |v|,  v = Wv m + bv
        st->e_ctx[l] = exp_for_bound(bmem * d->layers[l].kv_l1b[2] + d->layers[l].kv_l1b[3]);
    if (TensorTable::dev_absmax(d->tt.get("after_norm.weight"), D, &g, s) || TensorTable::dev_absmax(d->tt.get("after_norm.bias"), D, &b, s)) return -2;
    st->e_an = exp_for_bound(sqrtf((float)D) * g + b);
    int ew_v = 0;
    if (!d->tt.get_split2("output_layer.weight", dc.vocab_size, D, &ew_v, s)) return -2;
    st->ver_e = e->tt.version; st->ver_d = d->tt.version;
    return 0;
}
static bool stream_x2_ready(const Stream* st) {
    const Encoder* e = st->e; const Decoder* d = st->d;
    if (!e->resolved || !d->resolved || st->ver_e != e->tt.version || st->ver_d != d->tt.version) return false;
    for (auto& w : e->layers) if (!w.qkv_w2 || !w.out_w2 || !w.w1_2 || !w.w2_2) return false;
    for (auto& w : d->layers) if (!w.x2_ready) return false;
    return d->last.x2_ready && d->tt.b16.count("output_layer.weight#split2") != 0 && (int)st->e_ctx.size() == d->cfg.n_blocks;
}

// enqueue one chunk on `s` (no host synchronisation, no allocation after the first call with this shape)
static int stream_enqueue(Stream* st, int n, int is_final, int tail, hipStream_t s) {
    Encoder* e = st->e; Predictor* p = st->p; Decoder* d = st->d;
    const pf_encoder_config& ec = e->cfg;
    const int S = st->S, D = ec.d_model, F = ec.ffn_dim, Din = ec.input_dim, Dpad = round_up(Din, 64);
    const int W = tail ? st->keep : st->keep + n;
    const int M = S * W, Nmax = st->Nmax;
    const StreamDev* dev = st->dev_state.as<StreamDev>();
    StreamModeScope small_m_kernels;
    int rc;
    // ---- workspaces of the three handles (grow-only; the first eager call with a shape allocates)
    {
        const size_t Mz = (size_t)S * st->Wmax;
        const int Fbuf = F > Din ? F : Din;
        if (e->x.ensure(sizeof(float) * Mz * D) || e->xn.ensure(sizeof(float) * Mz * (Dpad > D ? Dpad : D)) ||
            e->qkv.ensure(sizeof(float) * Mz * 3 * D) || e->mem.ensure(sizeof(float) * Mz * D) ||
            e->ctx.ensure(sizeof(float) * Mz * D) || e->ffn.ensure(sizeof(float) * Mz * Fbuf))
            return -2;
        if (st->x2 && (e->xn16.ensure(sizeof(unsigned short) * 2 * Mz * (Dpad > D ? Dpad : D)) ||
                       e->ctx16.ensure(sizeof(unsigned short) * 2 * Mz * D) || e->ffn16.ensure(sizeof(unsigned short) * 2 * Mz * F) ||
                       st->mem2.ensure(sizeof(unsigned short) * 2 * Mz * D) || e->splitk.ensure(sizeof(float) * 4 * Mz * D)))
            return -2;
    }
    if ((rc = launch_fill_int(st->lensW.as<int>(), S, W, s))) return rc;
    // ---- window: [cached rows | x * sqrt(d) + PE]  (scama/encoder.py:496-503)
    StreamEmbedArgs ea{};
    ea.feats = tail ? nullptr : st->feats_in.as<float>(); ea.pe = st->pe.as<float>();
    ea.cache_feats = st->cache_feats.as<float>(); ea.win = st->win.as<float>(); ea.st = dev; ea.S = S; ea.n = n;
    ea.keep = st->keep; ea.Din = Din; ea.pe_rows = st->pe_rows; ea.tail = tail; ea.scale = (float)sqrt((double)D);
    if ((rc = launch_stream_embed(ea, s))) return rc;
    // ---- encoder blocks on the window
    // rows [0, W - chunk_right) of the window's K / V go to the ring. The reference takes them as k_h[:, :, :-(chunk_size[2])]
    // (sanm/attention.py:345-346): with chunk_size[2] == 0 that slice is [:-0] = EMPTY, so such a geometry never caches
    // anything and its look-back has no effect -- reproduced (oracle/fuzz_streaming_vs_reference.py found the difference)
    const int append_rows = (st->cfg.chunk_right > 0 && W - st->cfg.chunk_right > 0) ? W - st->cfg.chunk_right : 0;
    float* x = e->x.as<float>();
    const size_t ring_layer = (size_t)S * st->enc_cap * 2 * D;
    for (size_t l = 0; l < e->layers.size(); ++l) {
        EncChunkCtx cc{st->enc_cap > 0 ? st->enc_ring.as<float>() + l * ring_layer : nullptr, st->enc_cap, dev, append_rows,
                       st->lensW.as<int>(), st->x2};
        if (l == 0) rc = encoder_block(e, e->layers[0], st->win.as<float>(), Din, x, S, W, s, &cc);
        else rc = encoder_block(e, e->layers[l], x, D, x, S, W, s, &cc);
        if (rc) return rc;
    }
    StreamAdvanceArgs adv{};
    adv.st = st->dev_state.as<StreamDev>(); adv.n_frames = tail ? st->keep : n;   // the tail chunk re-feeds `keep` rows (embedding.py:478)
    adv.enc_rows = append_rows; adv.enc_cap = st->enc_cap;
    if ((rc = launch_stream_advance_enc(adv, s))) return rc;
    float* enc_out = st->enc_out.as<float>();
    if ((rc = layernorm(x, D, e->tt.get("after_norm.weight"), e->tt.get("after_norm.bias"), enc_out, D, M, D, D,
                        ec.ln_eps, s))) return rc;
    // ---- predictor chunk (cif_predictor.py:316-392)
    const pf_predictor_config& pc = p->cfg;
    const int taps = pc.l_order + pc.r_order + 1;
    if (p->col.ensure(sizeof(float) * (size_t)S * st->Wmax * taps * D) || p->conv.ensure(sizeof(float) * (size_t)S * st->Wmax * D))
        return -2;
    if ((rc = launch_im2col(enc_out, p->col.as<float>(), S, W, D, pc.l_order, pc.r_order, s))) return rc;
    if ((rc = gemm_simple(p->col.as<float>(), taps * D, p->tt.get("cif_conv1d.weight"), taps * D,
                          p->tt.get("cif_conv1d.bias"), p->conv.as<float>(), D, M, D, taps * D, 1, nullptr, 0, nullptr,
                          0, s))) return rc;
    AlphaArgs aa{};
    aa.conv = p->conv.as<float>(); aa.w = p->tt.get("cif_output.weight"); aa.bias = p->tt.get("cif_output.bias");
    aa.lens = st->lensW.as<int>(); aa.alphas = st->alphas.as<float>(); aa.B = S; aa.T = W; aa.D = D; aa.T_ext = st->Wmax + 1;
    aa.smooth = pc.smooth_factor; aa.noise = pc.noise_threshold;
    if ((rc = launch_alpha(aa, s))) return rc;
    CifChunkArgs ca{};
    ca.hidden = enc_out; ca.alphas = st->alphas.as<float>(); ca.ld_alpha = st->Wmax + 1;
    ca.cif_hidden = st->cif_hidden.as<float>(); ca.cif_alpha = st->cif_alpha.as<float>();
    ca.embeds = st->embeds.as<float>(); ca.n_fired = st->n_fired.as<int>(); ca.S = S; ca.W = W; ca.D = D; ca.Nmax = Nmax;
    ca.lo = st->cfg.chunk_left; ca.hi = is_final ? W : st->cfg.chunk_left + st->cfg.chunk_cur;
    ca.is_final = is_final; ca.tail_threshold = pc.tail_threshold; ca.threshold = pc.threshold;
    if ((rc = launch_cif_chunk(ca, s))) return rc;
    // ---- decoder chunk on Nmax token rows per stream; rows >= n_fired are padding, streams with n_fired == 0 keep
    //      their caches (the reference does not call the decoder then, paraformer_streaming/model.py:589-590)
    const pf_decoder_config& dc = d->cfg;
    const int V = dc.vocab_size, Mq = S * Nmax, Mk = M;
    if (d->x.ensure(sizeof(float) * (size_t)Mq * D) || d->t1.ensure(sizeof(float) * (size_t)Mq * D) ||
        d->t2.ensure(sizeof(float) * (size_t)Mq * D) || d->ffn.ensure(sizeof(float) * (size_t)Mq * dc.ffn_dim) ||
        d->ffn2.ensure(sizeof(float) * (size_t)Mq * dc.ffn_dim) || d->q.ensure(sizeof(float) * (size_t)Mq * D) ||
        d->kv.ensure(sizeof(float) * (size_t)S * st->Wmax * 2 * D) || d->ctx.ensure(sizeof(float) * (size_t)Mq * D) ||
        d->hid.ensure(sizeof(float) * (size_t)Mq * D))
        return -2;
    float* dx = d->x.as<float>();
    float* t1 = d->t1.as<float>();
    float* t2 = d->t2.as<float>();
    const bool x2 = st->x2;
    unsigned short* t2p = nullptr;       // gemm_mode 3: LayerNorm-output planes of the token rows, cross-attention output planes,
    unsigned short* c2p = nullptr;       // planes of the step's encoder output (the cross-attention memory)
    const unsigned short* mem2 = nullptr;
    if (x2) {
        if (d->t16.ensure(sizeof(unsigned short) * 2 * (size_t)Mq * D) || d->ffn16.ensure(sizeof(unsigned short) * 2 * (size_t)Mq * dc.ffn_dim) ||
            d->ctx16.ensure(sizeof(unsigned short) * 2 * (size_t)Mq * D))
            return -2;
        if (d->splitk.ensure(sizeof(float) * 4 * (size_t)Mq * D)) return -2;
        t2p = d->t16.as<unsigned short>(); c2p = d->ctx16.as<unsigned short>();
        if ((rc = launch_split2(enc_out, D, st->mem2.as<unsigned short>(), D, (size_t)Mk * D, Mk, D, pow2f(st->e_mem), s))) return rc;
        mem2 = st->mem2.as<unsigned short>();
    }
    PF_HIP_TRY(hipMemcpyAsync(dx, st->embeds.p, sizeof(float) * (size_t)Mq * D, hipMemcpyDeviceToDevice, s));
    const size_t dring_layer = (size_t)S * st->dec_cap * 2 * D;
    const size_t dfsmn_layer = (size_t)S * (dc.kernel_size - 1) * D;
    for (int l = 0; l < dc.n_blocks; ++l) {
        const DecLayerW& w = d->layers[l];
        if ((rc = x2 ? dec_ffn_x2(d, w, dx, t2, Mq, s, d->splitk.as<float>()) : dec_ffn(d, w, dx, t2, Mq, s))) return rc;
        if ((rc = layernorm(t2, D, w.n2g, w.n2b, t1, D, Mq, D, D, dc.ln_eps, s))) return rc;
        DecFsmnChunkArgs fa{};
        fa.in = t1; fa.resid = dx; fa.out = dx; fa.w = w.fsmn_w; fa.state = st->dec_fsmn.as<float>() + l * dfsmn_layer;
        fa.n_valid = st->n_fired.as<int>(); fa.S = S; fa.N = Nmax; fa.C = D;
        if ((rc = launch_dec_fsmn_chunk(fa, s))) return rc;
        if (x2) {
            {
                ProfScope ps(PROF_LN, 8.0 * Mq * (double)D, s);
                if ((rc = launch_layernorm(dx, D, w.n3g, w.n3b, reinterpret_cast<float*>(t2p), D, Mq, D, D, dc.ln_eps, s, 3, 0,
                                           (size_t)Mq * D, pow2f(w.e_n3)))) return rc;
            }
            if ((rc = gemm2_simple(t2p, D, Mq, w.e_n3, w.q_2, w.ew_q, w.q_b, d->q.as<float>(), D, D, D, 0, nullptr, 0, s))) return rc;
            if ((rc = gemm2_simple(mem2, D, Mk, st->e_mem, w.kv_2, w.ew_kv, w.kv_b, d->kv.as<float>(), 2 * D, 2 * D, D, 0, nullptr, 0, s)))
                return rc;
        } else {
            if ((rc = layernorm(dx, D, w.n3g, w.n3b, t1, D, Mq, D, D, dc.ln_eps, s))) return rc;
            if ((rc = gemm_simple(t1, D, w.q_w, D, w.q_b, d->q.as<float>(), D, Mq, D, D, 0, nullptr, 0, nullptr, 0, s))) return rc;
            if ((rc = gemm_simple(enc_out, D, w.kv_w, D, w.kv_b, d->kv.as<float>(), 2 * D, Mk, 2 * D, D, 0, nullptr, 0,
                                  nullptr, 0, s))) return rc;
        }
        AttnArgs at{};
        at.Q = d->q.as<float>(); at.ldq = D; at.O = d->ctx.as<float>(); at.ldo = D; at.B = S; at.H = dc.n_heads;
        at.Tq = Nmax; at.scale = powf((float)(D / dc.n_heads), -0.5f);
        if (st->dec_cap > 0) {
            float* ring = st->dec_ring.as<float>() + l * dring_layer;
            at.K = ring; at.ldk = 2 * D; at.V = ring + D; at.ldv = 2 * D; at.Tk = st->dec_cap;
            at.K2 = d->kv.as<float>(); at.ldk2 = 2 * D; at.V2 = d->kv.as<float>() + D; at.ldv2 = 2 * D; at.T2 = W; at.n2 = W;
            at.n1_dev = st->dec_valid.as<int>(); at.n1_stride = 1;
        } else {
            at.K = d->kv.as<float>(); at.ldk = 2 * D; at.V = d->kv.as<float>() + D; at.ldv = 2 * D; at.Tk = W;
            at.klens = st->lensW.as<int>();
        }
        bool appended = false;
        if (st->dec_cap > 0) {
            at.app_rows = W; at.app_r0 = 0; at.app_wp = st->dec_wp.as<int>(); at.app_wp_stride = 1; at.app_gate = st->n_fired.as<int>();
        }
        if ((rc = attention(at, 4.0 * S * (double)Nmax * W * D, s, false, 128, &appended))) return rc;
        if (st->dec_cap > 0 && !appended) {
            RingAppendArgs ra{};
            ra.src = d->kv.as<float>(); ra.ldsrc = 2 * D; ra.src_T = W; ra.r0 = 0; ra.rows = W; ra.cols = 2 * D;
            ra.ring = st->dec_ring.as<float>() + l * dring_layer; ra.cap = st->dec_cap; ra.S = S; ra.st = nullptr;
            ra.wp_dev = st->dec_wp.as<int>(); ra.gate_dev = st->n_fired.as<int>();
            if ((rc = launch_ring_append(ra, s))) return rc;
        }
        if (x2) {
            if ((rc = launch_split2(d->ctx.as<float>(), D, c2p, D, (size_t)Mq * D, Mq, D, pow2f(st->e_ctx[l]), s))) return rc;
            if ((rc = gemm2_simple(c2p, D, Mq, st->e_ctx[l], w.o_2, w.ew_o, w.o_b, dx, D, D, D, 0, dx, D, s))) return rc;
        } else if ((rc = gemm_simple(d->ctx.as<float>(), D, w.o_w, D, w.o_b, dx, D, Mq, D, D, 0, nullptr, 0, dx, D, s))) return rc;
    }
    if (st->dec_cap > 0) {
        StreamAdvanceArgs ad{};
        ad.dec_valid = st->dec_valid.as<int>(); ad.dec_wp = st->dec_wp.as<int>(); ad.gate = st->n_fired.as<int>();
        ad.S = S; ad.dec_rows = W; ad.dec_cap = st->dec_cap;
        if ((rc = launch_stream_advance_dec(ad, s))) return rc;
    }
    if ((rc = x2 ? dec_ffn_x2(d, d->last, dx, t2, Mq, s, d->splitk.as<float>()) : dec_ffn(d, d->last, dx, t2, Mq, s))) return rc;
    if (x2) {
        // after_norm writes two-plane operands, the vocabulary projection runs with the row arg-max fused into its epilogue
        // (the offline greedy route of decoder_forward_impl)
        auto wv = d->tt.b16.find("output_layer.weight#split2");
        if (wv == d->tt.b16.end()) { set_error("stream: f16x2 step without prepared vocabulary planes"); return -1; }
        const int ew_v = d->tt.exp2.at("output_layer.weight#split2");
        {
            ProfScope ps(PROF_LN, 8.0 * Mq * (double)D, s);
            if ((rc = launch_layernorm(t2, D, d->tt.get("after_norm.weight"), d->tt.get("after_norm.bias"), reinterpret_cast<float*>(t2p), D,
                                       Mq, D, D, dc.ln_eps, s, 3, 0, (size_t)Mq * D, pow2f(st->e_an)))) return rc;
        }
        const int nparts = gemm_f16x2_argmax_parts(Mq, V);
        if (d->pval.ensure(sizeof(float) * (size_t)Mq * nparts) || d->pidx.ensure(sizeof(int) * (size_t)Mq * nparts)) return -2;
        Gemm2Args g{};
        g.A = t2p; g.lda = D; g.a_plane = (size_t)Mq * D; g.W = wv->second; g.ldw = D; g.w_plane = (size_t)V * D;
        g.oscale = pow2f(-(st->e_an + ew_v)); g.bias = d->tt.get("output_layer.bias"); g.M = Mq; g.N = V; g.K = D;
        g.amax_val = d->pval.as<float>(); g.amax_idx = d->pidx.as<int>(); g.amax_ld = nparts;
        {
            ProfScope ps(PROF_GEMM3, 2.0 * Mq * (double)V * D, s);
            if ((rc = launch_gemm_f16x2(g, s))) return rc;
        }
        if ((rc = launch_argmax_reduce(d->pval.as<float>(), d->pidx.as<int>(), nparts, nparts, st->ids.as<int32_t>(), nullptr, Mq, s))) return rc;
    } else {
        if ((rc = layernorm(t2, D, d->tt.get("after_norm.weight"), d->tt.get("after_norm.bias"), d->hid.as<float>(), D, Mq,
                            D, D, dc.ln_eps, s))) return rc;
        if ((rc = vocab_project(d->hid.as<float>(), Mq, D, d->tt.get("output_layer.weight"), d->tt.get("output_layer.bias"),
                                V, nullptr, st->ids.as<int32_t>(), d->pval, d->pidx, s))) return rc;
    }
    PF_HIP_TRY(hipMemcpyAsync(st->h_ids, st->ids.p, sizeof(int32_t) * (size_t)Mq, hipMemcpyDeviceToHost, s));
    PF_HIP_TRY(hipMemcpyAsync(st->h_n, st->n_fired.p, sizeof(int32_t) * (size_t)S, hipMemcpyDeviceToHost, s));
    return 0;
}


// ---- engine_tables.h: the handles' weight storage for dp_rccl.hip
TensorTable* table_of(int kind, void* h);     // defined at the end of the file (needs every handle type)
int handle_tensor_spans(int kind, void* handle, std::vector<TensorSpan>& out) {
    TensorTable* tt = handle ? table_of(kind, handle) : nullptr;
    if (!tt) { set_error("dp: null handle or unknown handle kind"); return -1; }
    out.clear();
    for (auto& kv : tt->t) out.push_back({kv.first, kv.second.d, kv.second.device_elems(), kv.second.set});
    return 0;
}
}  // namespace pf

using namespace pf;

// ================================================================================================== C ABI
// time `iters` launches of fn on s (after 3 warm-up launches); the pf_k_* measurement hooks
template <class F> static int time_launches(F&& fn, int iters, float* ms_out, hipStream_t s) {
    int rc;
    for (int i = 0; i < 3; ++i) if ((rc = fn())) return rc;
    hipEvent_t a, b;
    PF_HIP_TRY(hipEventCreate(&a));
    PF_HIP_TRY(hipEventCreate(&b));
    PF_HIP_TRY(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) if ((rc = fn())) return rc;
    PF_HIP_TRY(hipEventRecord(b, s));
    PF_HIP_TRY(hipEventSynchronize(b));
    float ms = 0.f;
    PF_HIP_TRY(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms_out = ms / iters;
    return 0;
}

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
// per call site: fills up to `cap` rows (tag pointer, kind, total ms, total work, launches) of the tagged records; returns the row count
int pf_prof_read_tags(int cap, const char** tags, int* kinds, double* total_ms, double* total_work, int64_t* launches) {
    int n = 0;
    for (auto& r : g_prof) {
        if (!r.tag) continue;
        int i = 0;
        while (i < n && !(tags[i] == r.tag && kinds[i] == r.kind)) ++i;
        if (i == n) {
            if (n >= cap) continue;
            tags[n] = r.tag; kinds[n] = r.kind; total_ms[n] = 0; total_work[n] = 0; launches[n] = 0; ++n;
        }
        if (hipEventSynchronize(r.b) != hipSuccess) { set_error("prof: event sync failed"); return -2; }
        float t = 0;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) { set_error("prof: elapsed failed"); return -2; }
        total_ms[i] += t; total_work[i] += r.work; ++launches[i];
    }
    return n;
}
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
    if (upload_h2d(f->nfr.p, nfr.data(), sizeof(int32_t) * B, s)) return -2;
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

// --------------------------------------------------------------------------------------------------- encoder
pf_encoder* pf_encoder_create(const pf_encoder_config* cfg) {
    if (!cfg) { set_error("encoder: null config"); return nullptr; }
    if (check_device()) return nullptr;
    const pf_encoder_config& c = *cfg;
    const int dk = (c.n_heads > 0 && c.d_model > 0) ? c.d_model / c.n_heads : 0;
    if (c.d_model <= 0 || c.n_heads <= 0 || c.d_model % c.n_heads || !(dk == 128 || (dk <= 64 && dk % 4 == 0)) ||
        c.input_dim % 4 || c.ffn_dim % 32 || c.d_model % 32 || c.n_blocks < 1 || c.tp_blocks < 0 ||
        c.kernel_size != 11) {
        set_error("encoder: unsupported config (need d_model/n_heads == 128, or <= 64 for the small-head kernel; "
                  "kernel_size == 11, dims % 32 == 0)");
        return nullptr;
    }
    std::unique_ptr<Encoder> e(new Encoder());
    e->cfg = c;
    // default arithmetic: the f16x2 mode (fp32-class results on the fp16 matrix cores, the measured mode) wherever its
    // kernels exist, the fp32 MFMA otherwise; pf_encoder_set_precision overrides
    e->precision = (c.n_heads > 0 && c.d_model / c.n_heads == 128 && c.d_model % 256 == 0 && c.ffn_dim % 256 == 0) ? 3 : 0;
    std::vector<std::pair<std::string, int>> names;
    enc_layer_names(names, c);
    const int D = c.d_model, F = c.ffn_dim;
    int rc = 0;
    for (auto& nm : names) {
        const std::string& p = nm.first;
        const int in = nm.second, in_pad = round_up(in, 64);
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
    e->tt.drop_bf16();
    return e->tt.set(name, data, numel);
}
/* 0 = fp32 MFMA, 1 = bf16 operands for the GEMMs and the attention (fp32 accumulate, fp32
 * residual stream / LayerNorm statistics / softmax / FSMN): the throughput mode of BASELINE configs[1] */
int pf_encoder_set_precision(pf_encoder* eh, int32_t mode) {
    Encoder* e = reinterpret_cast<Encoder*>(eh);
    PF_REQUIRE(e && mode >= 0 && mode <= 3, "encoder_set_precision: mode must be 0 (fp32 MFMA), 1 (bf16 operands), 2 (fp32 via bf16x3) or 3 (fp32 via f16x2)");
    e->precision = mode;
    return 0;
}
/* f16x2 mode only: extra_rows >= 0 lays the sequences out back to back and computes min(len_b + extra_rows, T) rows of
 * sequence b -- the rest of out_dev reads as zero; extra_rows < 0 (default) computes every row of [B, T] in the padded layout */
/* tuning / A-B options of the f16x2 mode: "fuse_row" (1 = linear_out / w_2 in their full-row form with the residual adds and
 * the following LayerNorm in the epilogue, the default; 0 = separate launches; results are bitwise equal), "attn_variant"
 * (attention_f16x2.hip schedule: 3 lazy rescale, the default; 1 pipelined; 0 plain) */
int pf_encoder_set_option(pf_encoder* eh, const char* key, int32_t value) {
    Encoder* e = reinterpret_cast<Encoder*>(eh);
    PF_REQUIRE(e && key, "encoder_set_option: null");
    const std::string k = key;
    if (k == "fuse_row") { PF_REQUIRE(value == 0 || value == 1, "encoder_set_option: fuse_row is 0 or 1"); e->fuse_row = value; return 0; }
    if (k == "ffn_abl") { e->ffn_abl = value; return 0; }
    if (k == "ffn_fused") { PF_REQUIRE(value >= 0 && value <= 2, "encoder_set_option: ffn_fused is 0, 1 or 2"); e->ffn_fused = value; return 0; }
    if (k == "fsmn_fused") { PF_REQUIRE(value == 0 || value == 1, "encoder_set_option: fsmn_fused is 0 or 1"); e->fsmn_fused = value; return 0; }
    if (k == "row_bm") { PF_REQUIRE(value == 0 || value == 96 || value == 128 || value == 129, "encoder_set_option: row_bm is 0, 96, 128 or 129"); e->row_bm = value; return 0; }
    if (k == "row_nt") { PF_REQUIRE(value >= 0 && value <= 2, "encoder_set_option: row_nt is 0, 1 or 2"); e->row_nt = value; return 0; }
    if (k == "gemm_tile") { PF_REQUIRE(value == 0 || value == 1 || value == 2 || value == 5 || value == 6, "encoder_set_option: gemm_tile is 0, 1, 2, 5 or 6"); e->gemm_tile = value; return 0; }
    if (k == "attn_variant") { PF_REQUIRE(value == 0 || value == 1 || value == 3, "encoder_set_option: attn_variant is 0, 1 or 3"); e->attn_variant = value; return 0; }
    set_error("encoder_set_option: unknown key " + k);
    return -1;
}
/* test hook: fill every activation workspace of the handle with `byte` (0x7B: huge FINITE fp16 / fp32 patterns). A forward
 * must not depend on what earlier batches left in the workspaces -- whatever it reads past its own rows is masked exactly --
 * so results before and after poisoning are bitwise equal (tests/test_stateless_gpu.py). Synchronises. */
static int poison(std::initializer_list<DevBuf*> bufs, int byte) {
    PF_HIP_TRY(hipDeviceSynchronize());
    for (DevBuf* b : bufs) if (b->p && b->cap) PF_HIP_TRY(hipMemset(b->p, byte, b->cap));
    PF_HIP_TRY(hipDeviceSynchronize());
    return 0;
}
int pf_encoder_debug_poison(pf_encoder* eh, int32_t byte) {
    Encoder* e = reinterpret_cast<Encoder*>(eh);
    PF_REQUIRE(e, "encoder_debug_poison: null");
    return poison({&e->x, &e->xn, &e->qkv, &e->mem, &e->ctx, &e->ffn, &e->xn16, &e->qkv16, &e->ctx16, &e->ffn16, &e->q2, &e->k2, &e->vt2}, byte);
}
int pf_decoder_debug_poison(pf_decoder* dh, int32_t byte) {
    Decoder* d = reinterpret_cast<Decoder*>(dh);
    PF_REQUIRE(d, "decoder_debug_poison: null");
    return poison({&d->x, &d->t1, &d->t2, &d->ffn, &d->ffn2, &d->q, &d->kv, &d->ctx, &d->pval, &d->pidx, &d->hid, &d->t16, &d->ffn16,
                   &d->ffn2_16, &d->q16, &d->kv16, &d->ctx16, &d->mem16, &d->hid16, &d->k2, &d->vt2, &d->ids_packed}, byte);
}
int pf_predictor_debug_poison(pf_predictor* ph, int32_t byte) {
    Predictor* p = reinterpret_cast<Predictor*>(ph);
    PF_REQUIRE(p, "predictor_debug_poison: null");
    return poison({&p->col, &p->conv, &p->alphas, &p->peaks, &p->rems, &p->flags, &p->nfires}, byte);
}
int pf_encoder_set_row_packing(pf_encoder* eh, int32_t extra_rows) {
    Encoder* e = reinterpret_cast<Encoder*>(eh);
    PF_REQUIRE(e, "encoder_set_row_packing: null handle");
    e->pack_extra = extra_rows < 0 ? -1 : (extra_rows > (1 << 30) ? (1 << 30) : extra_rows);
    return 0;
}
/* SANMVadEncoder.forward (ct_transformer_streaming/encoder.py:355-430): vad_pos_host != NULL makes every block's attention
 * causal and the last block's use the VAD corner mask of vad_pos_host[b] (B values, consumed by the next forwards with that
 * batch size); NULL switches the masks off. fp32 mode, heads of d_k <= 64. */
int pf_encoder_set_vad_mask(pf_encoder* eh, const int32_t* vad_pos_host, int32_t B) {
    Encoder* e = reinterpret_cast<Encoder*>(eh);
    PF_REQUIRE(e && (vad_pos_host == nullptr || B > 0), "encoder_set_vad_mask: bad argument");
    e->vad_mask = vad_pos_host != nullptr;
    e->h_vad.assign(vad_pos_host ? vad_pos_host : nullptr, vad_pos_host ? vad_pos_host + B : nullptr);
    return 0;
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
    // f16x2 mode: every sequence occupies Tp = T rounded up to 16 rows (attention_f16x2.hip's tile alignment); the
    // extra rows are zero on entry, masked as keys, never returned
    const bool x2 = e->precision == 3 && c.d_model / c.n_heads == 128 && c.d_model % 256 == 0;
    const int Tp = x2 ? round_up(T, 16) : T;
    e->Tp = Tp;
    // packed rows (opt-in, f16x2 mode, full-depth forward): sequence b keeps min(len_b + pack_extra, T) rows in a slot
    // rounded up to 16 rows (attention_f16x2.hip's tile alignment: with every sequence starting on a 16-row boundary its key
    // tiles are the ones of the padded layout, so the kept rows are BITWISE what the padded layout computes and a clip's result
    // stays independent of its batch neighbours); slots lie back to back, nothing is computed for the padding behind them.
    // Taken only when it saves rows.
    e->cur_offs = nullptr;
    int packed_rows = 0, max_rows = 0;
    if (x2 && e->pack_extra >= 0 && run_blocks < 0) {
        e->h_offs.assign((size_t)B + 1, 0);
        for (int b = 0; b < B; ++b) {
            const int rows = lens_host[b] + e->pack_extra < T ? lens_host[b] + e->pack_extra : T;
            const int slot = round_up(rows, 16);
            e->h_offs[b] = packed_rows;
            packed_rows += slot;
            if (slot > max_rows) max_rows = slot;
        }
        e->h_offs[B] = packed_rows;
    }
    const bool pack = packed_rows > 0 && (size_t)packed_rows < (size_t)B * Tp;
    const size_t M = pack ? (size_t)packed_rows : (size_t)B * Tp;
    const int D = c.d_model, F = c.ffn_dim, Din = c.input_dim, Dpad = round_up(Din, 64);
    const int Fbuf = F > Din ? F : Din;
    if (e->precision == 3 && !x2) { set_error("encoder: the f16x2 mode needs d_model / n_heads == 128 and d_model % 256 == 0"); return -1; }
    if (x2) {
        const size_t cap_q = e->q2.cap, cap_k = e->k2.cap, cap_v = e->vt2.cap;
        if (e->xn16.ensure(sizeof(unsigned short) * 2 * M * (Dpad > D ? Dpad : D)) ||
            e->ctx16.ensure(sizeof(unsigned short) * 2 * M * D) || e->ffn16.ensure(sizeof(unsigned short) * 2 * M * F) ||
            e->q2.ensure(sizeof(unsigned short) * 2 * (M + 32) * D) || e->k2.ensure(sizeof(unsigned short) * 2 * (M + 32) * D) ||
            e->vt2.ensure(sizeof(unsigned short) * 2 * D * (M + 64)))
            return -2;
        // slack rows / columns past the last sequence are read by the last key tile (and masked): keep them finite
        if (e->q2.cap != cap_q) PF_HIP_TRY(hipMemsetAsync(e->q2.p, 0, e->q2.cap, s));
        if (e->k2.cap != cap_k) PF_HIP_TRY(hipMemsetAsync(e->k2.p, 0, e->k2.cap, s));
        if (e->vt2.cap != cap_v) PF_HIP_TRY(hipMemsetAsync(e->vt2.p, 0, e->vt2.cap, s));
        if ((rc = encoder_prepare_x2(e, s))) return rc;
    }
    if (e->precision == 1) {
        if (e->xn16.ensure(sizeof(unsigned short) * M * (Dpad > D ? Dpad : D)) || e->qkv16.ensure(sizeof(unsigned short) * M * 3 * D) ||
            e->ctx16.ensure(sizeof(unsigned short) * M * D) || e->ffn16.ensure(sizeof(unsigned short) * M * F))
            return -2;
        for (auto& w : e->layers) {
            if (w.qkv_w16) continue;
            w.qkv_w16 = e->tt.get_bf16(w.prefix + "self_attn.linear_q_k_v.weight", s);
            w.out_w16 = e->tt.get_bf16(w.prefix + "self_attn.linear_out.weight", s);
            w.w1_16 = e->tt.get_bf16(w.prefix + "feed_forward.w_1.weight", s);
            w.w2_16 = e->tt.get_bf16(w.prefix + "feed_forward.w_2.weight", s);
            if (!w.qkv_w16 || !w.out_w16 || !w.w1_16 || !w.w2_16) return -2;
        }
    }
    if (e->precision == 2) {
        if (e->xn16.ensure(sizeof(unsigned short) * 3 * M * (Dpad > D ? Dpad : D)) ||
            e->ctx16.ensure(sizeof(unsigned short) * 3 * M * D) || e->ffn16.ensure(sizeof(unsigned short) * 3 * M * F))
            return -2;
        for (auto& w : e->layers) {
            if (w.qkv_w3) continue;
            w.qkv_w3 = e->tt.get_split3(w.prefix + "self_attn.linear_q_k_v.weight", 3 * D, w.in_pad, s);
            w.out_w3 = e->tt.get_split3(w.prefix + "self_attn.linear_out.weight", D, D, s);
            w.w1_3 = e->tt.get_split3(w.prefix + "feed_forward.w_1.weight", F, D, s);
            w.w2_3 = e->tt.get_split3(w.prefix + "feed_forward.w_2.weight", D, F, s);
            if (!w.qkv_w3 || !w.out_w3 || !w.w1_3 || !w.w2_3) return -2;
        }
    }
    if (e->x.ensure(sizeof(float) * M * D) || e->xn.ensure(sizeof(float) * M * (Dpad > D ? Dpad : D)) ||
        e->qkv.ensure(sizeof(float) * M * 3 * D) || e->mem.ensure(sizeof(float) * M * D) ||
        e->ctx.ensure(sizeof(float) * M * D) || e->ffn.ensure(sizeof(float) * M * Fbuf))
        return -2;
    if ((rc = upload_lens(e->lens, lens_host, B, s))) return rc;
    e->cur_fs = nullptr;
    if (x2 && e->fuse_row && e->fsmn_fused) {
        // every sequence starts on a 16-row boundary in both layouts, so a 16-row group has one owner
        const int G = (int)(M / 16);
        e->h_fs.assign((size_t)2 * G, 0);
        for (int b = 0; b < B; ++b) {
            const int start = pack ? e->h_offs[b] : b * Tp, end = pack ? e->h_offs[b + 1] : (b + 1) * Tp;
            for (int g = start / 16; g < end / 16; ++g) { e->h_fs[g] = start; e->h_fs[(size_t)G + g] = start + lens_host[b]; }
        }
        if (e->fs_grp.ensure(sizeof(int32_t) * 2 * G)) return -2;
        if (upload_h2d(e->fs_grp.p, e->h_fs.data(), sizeof(int32_t) * 2 * G, s)) return -2;
        e->cur_fs = e->fs_grp.as<int>(); e->cur_fs_groups = G;
    }
    e->cur_mask_mode = 0;
    if (e->vad_mask) {
        PF_REQUIRE(e->precision == 0 && (int)e->h_vad.size() == B, "encoder: the VAD-masked encoder runs in the fp32 mode with one vad position per sequence");
        if ((rc = upload_lens(e->vad_dev, e->h_vad.data(), B, s))) return rc;
    }
    if (!pe) {
        if ((rc = encoder_default_pe(e, T, s))) return rc;
        pe = e->pe.as<float>();
    }
    // xs * sqrt(d_model) + PE (encoder.py:409,428). The scaled input is staged in the FFN scratch: block 0 reads
    // it in norm1 (and as residual when input_dim == d_model) strictly before its own FFN overwrites that buffer.
    float* x0 = e->ffn.as<float>();
    const float scale = (float)sqrt((double)D);
    float* x = e->x.as<float>();
    if (pack) {
        // row r of the packed layout <- row h_map[r] of the caller's [B, T]; the up-to-15 rows that fill a slot are further
        // padding rows of that sequence (zero input, masked as keys, FSMN memory 0, never returned)
        e->h_map.assign(M, -1);
        for (int b = 0; b < B; ++b) {
            const int rows = lens_host[b] + e->pack_extra < T ? lens_host[b] + e->pack_extra : T;
            for (int t = 0; t < rows; ++t) e->h_map[(size_t)e->h_offs[b] + t] = b * T + t;
        }
        if (e->offs_dev.ensure(sizeof(int32_t) * ((size_t)B + 1)) || e->map_dev.ensure(sizeof(int32_t) * M)) return -2;
        if (upload_h2d(e->offs_dev.p, e->h_offs.data(), sizeof(int32_t) * ((size_t)B + 1), s) ||
            upload_h2d(e->map_dev.p, e->h_map.data(), sizeof(int32_t) * M, s)) return -2;
        if ((rc = launch_scale_add_pe_rows(xs, pe, x0, e->map_dev.as<int>(), (int)M, T, Din, scale, s))) return rc;
        e->cur_offs = e->offs_dev.as<int>();
        e->cur_M = (int)M;
        const int total = (int)e->layers.size();
        bool xn_ready = false;
        for (int l = 0; l < total && !rc; ++l) {
            // the next block's norm1 rides in this block's w_2 epilogue unless another op sits between them (SenseVoice's after_norm)
            const bool boundary = c.tp_blocks > 0 && l + 1 == c.n_blocks;
            const EncLayerW* next = (l + 1 < total && !boundary) ? &e->layers[l + 1] : nullptr;
            rc = l == 0 ? encoder_block(e, e->layers[0], x0, Din, x, B, max_rows, s, nullptr, false, next)
                        : encoder_block(e, e->layers[l], x, D, x, B, max_rows, s, nullptr, xn_ready, next);
            xn_ready = next != nullptr && next->in_dim == D;
            if (!rc && boundary)
                rc = layernorm(x, D, e->tt.get("after_norm.weight"), e->tt.get("after_norm.bias"), x, D, (int)M, D, D, c.ln_eps, s);
        }
        e->cur_offs = nullptr;
        if (rc) return rc;
        // final LayerNorm scatters the rows back to [B, T]; the rows that were not computed read as zero
        PF_HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * T * D, s));
        ProfScope ps(PROF_LN, 8.0 * (double)packed_rows * D, s);
        return launch_layernorm(x, D, e->tt.get(c.tp_blocks > 0 ? "tp_norm.weight" : "after_norm.weight"),
                                e->tt.get(c.tp_blocks > 0 ? "tp_norm.bias" : "after_norm.bias"), out, D, (int)M, D, D, c.ln_eps, s,
                                0, 0, 0, 1.f, 0, 0, e->map_dev.as<int>());
    }
    if ((rc = launch_scale_add_pe(xs, pe, x0, B, T, Din, scale, s, Tp))) return rc;
    const int total = (int)e->layers.size();
    const int nrun = run_blocks < 0 ? total : (run_blocks < total ? run_blocks : total);
    // [B, Tp, w] workspace rows -> the caller's [B, T, w]
    auto unpad_copy = [&](const float* src, int w) -> int {
        PF_HIP_TRY(hipMemcpy2DAsync(out, sizeof(float) * (size_t)T * w, src, sizeof(float) * (size_t)Tp * w,
                                    sizeof(float) * (size_t)T * w, B, hipMemcpyDeviceToDevice, s));
        return 0;
    };
    if (nrun == 0) return unpad_copy(x0, Din);
    bool xn_ready = false;
    for (int l = 0; l < nrun; ++l) {
        // SANMVadEncoder: `encoders0` and all but the last of `encoders` are causal, the last one takes the VAD corner
        e->cur_mask_mode = e->vad_mask ? ((l >= 1 && l + 1 == total) ? 2 : 1) : 0;
        const bool boundary = c.tp_blocks > 0 && l + 1 == c.n_blocks;
        const EncLayerW* next = (l + 1 < nrun && !boundary) ? &e->layers[l + 1] : nullptr;
        if (l == 0) rc = encoder_block(e, e->layers[0], x0, Din, x, B, Tp, s, nullptr, false, next);
        else rc = encoder_block(e, e->layers[l], x, D, x, B, Tp, s, nullptr, xn_ready, next);
        xn_ready = next != nullptr && next->in_dim == D;
        e->cur_mask_mode = 0;
        if (rc) return rc;
        if (c.tp_blocks > 0 && l + 1 == c.n_blocks && (run_blocks < 0 || nrun > c.n_blocks)) {
            // SenseVoice: after_norm sits between `encoders` and `tp_encoders` (sense_voice/model.py:645-652)
            if ((rc = layernorm(x, D, e->tt.get("after_norm.weight"), e->tt.get("after_norm.bias"), x, D, (int)M, D, D,
                                c.ln_eps, s))) return rc;
        }
    }
    if (run_blocks >= 0) return unpad_copy(x, D);
    const char* fin_w = c.tp_blocks > 0 ? "tp_norm.weight" : "after_norm.weight";
    const char* fin_b = c.tp_blocks > 0 ? "tp_norm.bias" : "after_norm.bias";
    if (Tp == T) return layernorm(x, D, e->tt.get(fin_w), e->tt.get(fin_b), out, D, (int)M, D, D, c.ln_eps, s);
    ProfScope ps(PROF_LN, 8.0 * B * (double)T * D, s);
    return launch_layernorm(x, D, e->tt.get(fin_w), e->tt.get(fin_b), out, D, B * T, D, D, c.ln_eps, s, 0, 0, 0, 1.f, T, Tp);
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
pf_predictor* pf_predictor_create_v3(const pf_predictor_config* cfg, const pf_predictor_v3_config* cfg3) {
    if (!cfg3) { set_error("predictor_v3: null config"); return nullptr; }
    const pf_predictor_v3_config& c3 = *cfg3;
    if (c3.upsample_times < 1 || c3.upsample_times > 8 || (c3.upsample_type != 0 && c3.upsample_type != 1)) {
        set_error("predictor_v3: unsupported config (upsample_times 1..8; upsample_type 0 = cnn, 1 = cnn_blstm)");
        return nullptr;
    }
    if (cfg && !cfg->tail_mask && cfg->tail_threshold > 0.f) {
        set_error("predictor_v3: the reference always applies the tail threshold through the mask (tail_mask = 1)");
        return nullptr;
    }
    pf_predictor* ph = pf_predictor_create(cfg);
    if (!ph) return nullptr;
    Predictor* p = reinterpret_cast<Predictor*>(ph);
    p->v3 = true;
    p->c3 = c3;
    const int D = p->cfg.d_model, U = c3.upsample_times;
    int rc = 0;
    rc |= p->tt.add_upsample("upsample_cnn.weight", D, D, U);
    rc |= p->tt.add_tiled("upsample_cnn.bias", D, U);
    if (c3.upsample_type == 1) {
        for (const char* sfx : {"", "_reverse"}) {
            const std::string s(sfx);
            rc |= p->tt.add("blstm.weight_ih_l0" + s, (int64_t)4 * D * D);
            rc |= p->tt.add_lstm_hh("blstm.weight_hh_l0" + s, D);
            rc |= p->tt.add("blstm.bias_ih_l0" + s, (int64_t)4 * D);
            rc |= p->tt.add("blstm.bias_hh_l0" + s, (int64_t)4 * D);
        }
        rc |= p->tt.add("cif_output2.weight", 2 * D);
    } else {
        rc |= p->tt.add("cif_output2.weight", D);
    }
    rc |= p->tt.add("cif_output2.bias", 1);
    if (rc) { pf_predictor_destroy(ph); return nullptr; }
    return ph;
}
void pf_predictor_destroy(pf_predictor* p) { delete reinterpret_cast<Predictor*>(p); }
int pf_predictor_set_tensor(pf_predictor* ph, const char* name, const float* data, int64_t numel) {
    Predictor* p = reinterpret_cast<Predictor*>(ph);
    PF_REQUIRE(p && name && data, "predictor_set_tensor: null");
    p->packed = false;
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
    if (p->v3) {
        if (p->curs.ensure(sizeof(float) * (size_t)B * Te) || p->ntok.ensure(sizeof(int) * (size_t)B)) return -2;
        if ((rc = launch_cif_scan_loop(sa, p->curs.as<float>(), p->ntok.as<int>(), s))) return rc;
    } else if ((rc = launch_cif_scan(sa, s))) {
        return rc;
    }
    if (alphas) PF_HIP_TRY(hipMemcpyAsync(alphas, p->alphas.p, sizeof(float) * (size_t)B * Te, hipMemcpyDeviceToDevice, s));
    if (peaks) PF_HIP_TRY(hipMemcpyAsync(peaks, p->peaks.p, sizeof(float) * (size_t)B * Te, hipMemcpyDeviceToDevice, s));
    // V3 reports floor(sum alphas) (cif_predictor.py:383), V2's count of fires is the same number by construction
    PF_HIP_TRY(hipMemcpyAsync(token_num, p->v3 ? p->ntok.p : p->nfires.p, sizeof(int32_t) * (size_t)B,
                              hipMemcpyDeviceToHost, s));
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
    if (p->v3) {
        if (N <= 0) return 0;
        ea.alphas = p->curs.as<float>();
        return launch_cif_emit_loop(ea, s);
    }
    return launch_cif_emit(ea, s);
}

// one direction-pair of torch.nn.LSTM on a time-major input: gates-major input projections by the fp32 MFMA GEMM
// (W_ih . X_tm^T, one GEMM per direction), then the per-step recurrence (lstm.hip)
struct LstmW { const float* w_ih[2]; const float* w_hh; const float* b_ih; const float* b_hh; };
static int lstm_forward(const LstmW& w, const float* x_tm, int T, int B, int D, int H, int ndir, float* out, int out_layout,
                        DevBuf& pre, DevBuf& h_a, DevBuf& h_b, DevBuf& cell, hipStream_t s) {
    const size_t cols = (size_t)T * B, ldp = (cols + 3) / 4 * 4;
    const int Bs = (B + 63) / 64 * 64;
    const size_t state = sizeof(float) * (size_t)ndir * H * Bs;
    if (pre.ensure(sizeof(float) * (size_t)ndir * 4 * H * ldp) || h_a.ensure(state) || h_b.ensure(state) || cell.ensure(state))
        return -2;
    int rc;
    for (int d = 0; d < ndir; ++d)
        if ((rc = gemm_simple(w.w_ih[d], D, x_tm, D, nullptr, pre.as<float>() + (size_t)d * 4 * H * ldp, (int)ldp, 4 * H,
                              (int)cols, D, 0, nullptr, 0, nullptr, 0, s))) return rc;
    LstmStepArgs a{};
    a.pre = pre.as<float>(); a.whh = w.w_hh; a.b_ih = w.b_ih; a.b_hh = w.b_hh; a.h_a = h_a.as<float>();
    a.h_b = h_b.as<float>(); a.c = cell.as<float>(); a.out = out; a.ld_pre = ldp; a.T = T; a.B = B; a.Bs = Bs; a.H = H;
    a.ndir = ndir; a.out_layout = out_layout;
    return launch_lstm_steps(a, s);
}

int pf_predictor_timestamp(pf_predictor* ph, const float* hidden, const int32_t* lens_host, const int32_t* token_num_host,
                           int32_t B, int32_t T, float* us_alphas, float* us_peaks, void* stream) {
    Predictor* p = reinterpret_cast<Predictor*>(ph);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(p && hidden && lens_host && token_num_host && us_alphas && us_peaks && B > 0 && T > 0,
               "predictor_timestamp: null/empty argument");
    PF_REQUIRE(p->v3, "predictor_timestamp: the handle was not made by pf_predictor_create_v3");
    for (int b = 0; b < B; ++b) PF_REQUIRE(lens_host[b] >= 1 && lens_host[b] <= T, "predictor_timestamp: lens out of range");
    std::string first;
    if (p->tt.missing(&first)) { set_error("predictor: tensor not set: " + first); return -3; }
    const pf_predictor_config& c = p->cfg;
    const int D = c.d_model, U = p->c3.upsample_times, taps = c.l_order + c.r_order + 1, Tu = T * U;
    const size_t M = (size_t)B * T;
    int rc;
    if ((rc = upload_lens(p->lens, lens_host, B, s))) return rc;
    if (p->tok_dev.ensure(sizeof(int) * (size_t)B)) return -2;
    if (upload_h2d(p->tok_dev.p, token_num_host, sizeof(int32_t) * (size_t)B, s)) return -2;
    const float* src = hidden;
    if (p->c3.use_cif1_cnn) {                                   // the head sees relu(cif_conv1d(hidden)) instead (:317-320)
        if (p->col.ensure(sizeof(float) * M * taps * D) || p->conv.ensure(sizeof(float) * M * D)) return -2;
        if ((rc = launch_im2col(hidden, p->col.as<float>(), B, T, D, c.l_order, c.r_order, s))) return rc;
        if ((rc = gemm_simple(p->col.as<float>(), taps * D, p->tt.get("cif_conv1d.weight"), taps * D,
                              p->tt.get("cif_conv1d.bias"), p->conv.as<float>(), D, (int)M, D, taps * D, 1, nullptr, 0,
                              nullptr, 0, s))) return rc;
        src = p->conv.as<float>();
    }
    // ConvTranspose1d(k = stride = U) == one GEMM: row (b, t) of the output holds frames U t .. U t + U - 1
    if (p->up.ensure(sizeof(float) * M * U * D)) return -2;
    if ((rc = gemm_simple(src, D, p->tt.get("upsample_cnn.weight"), D, p->tt.get("upsample_cnn.bias"), p->up.as<float>(),
                          U * D, (int)M, U * D, D, 0, nullptr, 0, nullptr, 0, s))) return rc;
    if (p->c3.upsample_type == 1) {
        const int Bs = (B + 63) / 64 * 64;
        if (p->x_tm.ensure(sizeof(float) * (size_t)Tu * B * D) || p->lstm_out.ensure(sizeof(float) * (size_t)Tu * 2 * D * Bs))
            return -2;
        if ((rc = launch_rows_bt_to_tb(p->up.as<float>(), p->x_tm.as<float>(), B, Tu, D, s))) return rc;
        // the two directions' recurrent weights / biases live back to back so that one launch serves both
        LstmW w{};
        w.w_ih[0] = p->tt.get("blstm.weight_ih_l0"); w.w_ih[1] = p->tt.get("blstm.weight_ih_l0_reverse");
        const size_t hh = (size_t)4 * D * D, bb = (size_t)4 * D;
        if (p->pack.ensure(sizeof(float) * (2 * hh + 4 * bb))) return -2;
        float* pack = p->pack.as<float>();
        float* bi = pack + 2 * hh;
        float* bh = bi + 2 * bb;
        if (!p->packed) {
            PF_HIP_TRY(hipMemcpyAsync(pack, p->tt.get("blstm.weight_hh_l0"), sizeof(float) * hh, hipMemcpyDeviceToDevice, s));
            PF_HIP_TRY(hipMemcpyAsync(pack + hh, p->tt.get("blstm.weight_hh_l0_reverse"), sizeof(float) * hh, hipMemcpyDeviceToDevice, s));
            PF_HIP_TRY(hipMemcpyAsync(bi, p->tt.get("blstm.bias_ih_l0"), sizeof(float) * bb, hipMemcpyDeviceToDevice, s));
            PF_HIP_TRY(hipMemcpyAsync(bi + bb, p->tt.get("blstm.bias_ih_l0_reverse"), sizeof(float) * bb, hipMemcpyDeviceToDevice, s));
            PF_HIP_TRY(hipMemcpyAsync(bh, p->tt.get("blstm.bias_hh_l0"), sizeof(float) * bb, hipMemcpyDeviceToDevice, s));
            PF_HIP_TRY(hipMemcpyAsync(bh + bb, p->tt.get("blstm.bias_hh_l0_reverse"), sizeof(float) * bb, hipMemcpyDeviceToDevice, s));
            p->packed = true;
        }
        w.w_hh = pack; w.b_ih = bi; w.b_hh = bh;
        if ((rc = lstm_forward(w, p->x_tm.as<float>(), Tu, B, D, D, 2, p->lstm_out.as<float>(), 1, p->pre, p->h_a, p->h_b,
                               p->cell, s))) return rc;
        UsAlphaArgs ua{};
        ua.out_t = p->lstm_out.as<float>(); ua.w = p->tt.get("cif_output2.weight"); ua.bias = p->tt.get("cif_output2.bias");
        ua.lens = p->lens.as<int>(); ua.alphas = us_alphas; ua.B = B; ua.Bs = Bs; ua.T = Tu; ua.C = 2 * D; ua.U = U;
        ua.smooth = p->c3.smooth_factor2; ua.noise = p->c3.noise_threshold2;
        if ((rc = launch_us_alpha_t(ua, s))) return rc;
    } else {
        // plain `cnn` head: the row-major upsampled frames go straight through the one-wave-per-row dot kernel
        p->ul_host.resize(B);
        for (int b = 0; b < B; ++b) p->ul_host[b] = lens_host[b] * U;
        DevBuf& ulens = p->ulens;
        if ((rc = upload_lens(ulens, p->ul_host.data(), B, s))) return rc;
        AlphaArgs aa{};
        aa.conv = p->up.as<float>(); aa.w = p->tt.get("cif_output2.weight"); aa.bias = p->tt.get("cif_output2.bias");
        aa.lens = ulens.as<int>(); aa.alphas = us_alphas; aa.B = B; aa.T = Tu; aa.D = D; aa.T_ext = Tu;
        aa.smooth = p->c3.smooth_factor2; aa.noise = p->c3.noise_threshold2;
        if ((rc = launch_alpha(aa, s))) return rc;
    }
    return launch_us_scale_scan(us_alphas, us_peaks, p->tok_dev.as<int>(), B, Tu, (float)((double)c.threshold - 1e-4), s);
}

// --------------------------------------------------------------------------------------------------- decoder
static pf_decoder* decoder_create_impl(const pf_decoder_config* cfg, bool contextual);
pf_decoder* pf_decoder_create(const pf_decoder_config* cfg) { return decoder_create_impl(cfg, false); }
/* ContextualParaformerDecoder (funasr/models/contextual_paraformer/decoder.py:133-352): n_blocks - 1 standard blocks
 * ("decoders.{i}."), the last block under "last_decoder.", plus the hotword branch "bias_decoder.norm3.*",
 * "bias_decoder.src_attn.linear_{q,k_v,out}.*" and the 1x1 fusion "bias_output.weight" [D, 2D, 1] */
pf_decoder* pf_decoder_create_contextual(const pf_decoder_config* cfg) { return decoder_create_impl(cfg, true); }
static pf_decoder* decoder_create_impl(const pf_decoder_config* cfg, bool contextual) {
    if (!cfg) { set_error("decoder: null config"); return nullptr; }
    if (check_device()) return nullptr;
    const pf_decoder_config& c = *cfg;
    if (c.d_model <= 0 || c.n_heads <= 0 || c.d_model % c.n_heads || c.d_model / c.n_heads != 128 ||
        c.ffn_dim % 32 || c.d_model % 32 || c.n_blocks < 1 || (c.kernel_size != 11 && c.kernel_size != 21) ||
        (c.kernel_size == 21 && c.sanm_shift > 0) || c.vocab_size < 0) {
        set_error("decoder: unsupported config (need d_model/n_heads == 128, kernel_size 11 or 21 (21: sanm_shfit 0), "
                  "dims % 32 == 0; vocab_size 0 = no output layer)");
        return nullptr;
    }
    std::unique_ptr<Decoder> d(new Decoder());
    d->cfg = c;
    d->precision = (c.n_heads > 0 && c.d_model / c.n_heads == 128 && c.d_model % 256 == 0 && c.ffn_dim % 256 == 0) ? 3 : 0;
    d->contextual = contextual;
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
        const std::string p = dec_layer_prefix(contextual, c.n_blocks, i);
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
    if (contextual) {
        rc |= d->tt.add("bias_decoder.norm3.weight", D);
        rc |= d->tt.add("bias_decoder.norm3.bias", D);
        rc |= d->tt.add("bias_decoder.src_attn.linear_q.weight", (int64_t)D * D);
        rc |= d->tt.add("bias_decoder.src_attn.linear_q.bias", D);
        rc |= d->tt.add("bias_decoder.src_attn.linear_k_v.weight", (int64_t)2 * D * D);
        rc |= d->tt.add("bias_decoder.src_attn.linear_k_v.bias", 2 * D);
        rc |= d->tt.add("bias_decoder.src_attn.linear_out.weight", (int64_t)D * D);
        rc |= d->tt.add("bias_decoder.src_attn.linear_out.bias", D);
        rc |= d->tt.add("bias_output.weight", (int64_t)D * 2 * D);
    }
    add_ffn("decoders3.0.");
    rc |= d->tt.add("after_norm.weight", D);
    rc |= d->tt.add("after_norm.bias", D);
    if (c.vocab_size > 0) {          // SeACo's bias decoder has no output layer (use_output_layer: false)
        rc |= d->tt.add("output_layer.weight", (int64_t)c.vocab_size * D);
        rc |= d->tt.add("output_layer.bias", c.vocab_size);
    }
    if (rc) return nullptr;
    return reinterpret_cast<pf_decoder*>(d.release());
}
void pf_decoder_destroy(pf_decoder* d) { delete reinterpret_cast<Decoder*>(d); }
int pf_decoder_set_tensor(pf_decoder* dh, const char* name, const float* data, int64_t numel) {
    Decoder* d = reinterpret_cast<Decoder*>(dh);
    PF_REQUIRE(d && name && data, "decoder_set_tensor: null");
    d->resolved = false;
    d->tt.drop_bf16();
    return d->tt.set(name, data, numel);
}
/* same modes as pf_encoder_set_precision; the bf16 mode serves the fused arg-max route (logits_dev == NULL) */
int pf_decoder_set_precision(pf_decoder* dh, int32_t mode) {
    Decoder* d = reinterpret_cast<Decoder*>(dh);
    PF_REQUIRE(d && mode >= 0 && mode <= 3, "decoder_set_precision: mode must be 0 (fp32 MFMA), 1 (bf16 operands), 2 (fp32 via bf16x3) or 3 (fp32 via f16x2)");
    d->precision = mode;
    return 0;
}
int pf_decoder_missing(const pf_decoder* dh) {
    const Decoder* d = reinterpret_cast<const Decoder*>(dh);
    return d ? d->tt.missing() : -1;
}

// asf_layer >= 0: run blocks 0 .. asf_layer - 1, then block asf_layer up to its cross-attention SCORES and return the
// attention-score filter of sequence 0 in asf_scores [T] (decoder.py:485-513 forward_asf6 / :696-714 get_attn_mat)
struct DecCtxArgs { const float* info; int n_hot; float clas_scale; };    // hotword embeddings [B, n_hot, D] (contextual decoder)
static int decoder_forward_impl(Decoder* d, const float* memory, const int32_t* mem_lens, const float* embeds,
                                const int32_t* tok_lens, int32_t B, int32_t T, int32_t N, float* logits, int32_t* ids,
                                float* hidden_out, hipStream_t s, int asf_layer, float* asf_scores, const DecCtxArgs* cx = nullptr);

int pf_decoder_forward(pf_decoder* dh, const float* memory, const int32_t* mem_lens, const float* embeds,
                       const int32_t* tok_lens, int32_t B, int32_t T, int32_t N, float* logits, int32_t* ids,
                       float* hidden_out, void* stream) {
    return decoder_forward_impl(reinterpret_cast<Decoder*>(dh), memory, mem_lens, embeds, tok_lens, B, T, N, logits, ids,
                                hidden_out, reinterpret_cast<hipStream_t>(stream), -1, nullptr);
}
/* SeACo attention-score filter (seaco_paraformer/model.py:323-335): the bias decoder's blocks 0 .. n_blocks_before - 1 in
 * full, then block n_blocks_before up to its cross-attention probabilities over the T memory rows (= hotword embeddings);
 * scores_dev [T] receives their sum over heads and token positions for sequence 0 (attn[0].sum(0).sum(0)). fp32 kernels. */
int pf_decoder_asf_scores(pf_decoder* dh, const float* memory, const int32_t* mem_lens, const float* embeds,
                          const int32_t* tok_lens, int32_t B, int32_t T, int32_t N, int32_t n_blocks_before,
                          float* scores_dev, void* stream) {
    Decoder* d = reinterpret_cast<Decoder*>(dh);
    PF_REQUIRE(d && scores_dev && n_blocks_before >= 0 && n_blocks_before < d->cfg.n_blocks, "decoder_asf_scores: bad block index");
    return decoder_forward_impl(d, memory, mem_lens, embeds, tok_lens, B, T, N, nullptr, nullptr, nullptr,
                                reinterpret_cast<hipStream_t>(stream), n_blocks_before, scores_dev);
}

/* ContextualParaformerDecoder.forward (contextual_paraformer/decoder.py:293-352): the last attention block's FSMN-side state
 * x_self_attn also queries the hotword embeddings `contextual_dev` [B, n_hot, D] through bias_decoder; its output (times
 * clas_scale) and the block's own cross-attention output are fused by the 1x1 bias_output: x = x_self_attn + W [x_src | cx]. */
int pf_decoder_forward_contextual(pf_decoder* dh, const float* memory, const int32_t* mem_lens, const float* embeds,
                                  const int32_t* tok_lens, const float* contextual_dev, int32_t n_hot, float clas_scale,
                                  int32_t B, int32_t T, int32_t N, float* logits, int32_t* ids, float* hidden_out, void* stream) {
    Decoder* d = reinterpret_cast<Decoder*>(dh);
    PF_REQUIRE(d && d->contextual && contextual_dev && n_hot >= 1, "decoder_forward_contextual: needs a contextual decoder and >= 1 hotword row");
    DecCtxArgs cx{contextual_dev, n_hot, clas_scale};
    return decoder_forward_impl(d, memory, mem_lens, embeds, tok_lens, B, T, N, logits, ids, hidden_out,
                                reinterpret_cast<hipStream_t>(stream), -1, nullptr, &cx);
}

static int decoder_forward_impl(Decoder* d, const float* memory, const int32_t* mem_lens, const float* embeds,
                                const int32_t* tok_lens, int32_t B, int32_t T, int32_t N, float* logits, int32_t* ids,
                                float* hidden_out, hipStream_t s, int asf_layer, float* asf_scores, const DecCtxArgs* cx) {
    PF_REQUIRE(d && memory && mem_lens && embeds && tok_lens && B > 0 && T > 0 && N > 0, "decoder_forward: null/empty");
    for (int b = 0; b < B; ++b) {
        PF_REQUIRE(mem_lens[b] >= 1 && mem_lens[b] <= T, "decoder_forward: memory lens out of range");
        PF_REQUIRE(tok_lens[b] >= 0 && tok_lens[b] <= N, "decoder_forward: token lens out of range");
    }
    int rc;
    if (!d->resolved && (rc = decoder_resolve(d))) return rc;
    const pf_decoder_config& c = d->cfg;
    const int D = c.d_model, F = c.ffn_dim, V = c.vocab_size;
    int Mq = B * N;                                          // rows processed per token-side op (shrinks when packed, below)
    const int Mq_pad = B * N, Mk = B * T;
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
    // the f16x2 greedy route may pack the token rows instead (below); every other route starts from the padded embeddings
    const bool may_pack = d->precision == 3 && asf_layer < 0 && !cx && ids && !logits && !hidden_out && V > 0;
    if (!may_pack) PF_HIP_TRY(hipMemcpyAsync(x, embeds, sizeof(float) * (size_t)Mq * D, hipMemcpyDeviceToDevice, s));
    const int left_pad = (c.kernel_size - 1) / 2 + (c.sanm_shift > 0 ? c.sanm_shift : 0);
    if (V == 0 && asf_layer < 0) PF_REQUIRE(!logits && !ids && hidden_out && d->precision != 1,
                           "decoder_forward: a decoder without output layer returns hidden states only (fp32 / bf16x3)");
    if (d->precision == 1 && !logits && asf_layer < 0 && !cx) return decoder_forward_bf16(d, memory, B, T, N, ids, hidden_out, s);
    // bf16x3 mode: the two GEMMs that are large at every batch size (w_1: N = ffn_dim; linear_k_v: M = B * T) take
    // three-plane operands on the bf16 matrix cores; the D x D projections and w_2 keep the fp32 MFMA tiles
    PF_REQUIRE(d->contextual == (cx != nullptr) || asf_layer >= 0, "decoder_forward: a contextual decoder runs through pf_decoder_forward_contextual");
    const bool x3 = d->precision == 2 && asf_layer < 0 && !cx;
    const bool x2 = d->precision == 3 && asf_layer < 0 && !cx;      // the score filter / hotword branch run on the fp32 kernels
    const int Tp = round_up(T, 16), Mkp = B * Tp;           // f16x2: padded key rows per sequence
    const unsigned short* mem3 = nullptr;
    const unsigned short* mem2 = nullptr;
    float* dsc = nullptr;
    if (x2) {
        // f16x2 mode: w_1, w_2, linear_q, linear_k_v on the fp16 matrix cores (gemm_f16x2.hip). The memory planes' scale
        // is chosen on the device from max |memory| (no host round trip); linear_out keeps the fp32 MFMA tile (its operand,
        // the attention output, has no a-priori bound here)
        const size_t cap_k = d->k2.cap, cap_v = d->vt2.cap;
        if (d->t16.ensure(sizeof(unsigned short) * 2 * (size_t)Mq * D) || d->ffn16.ensure(sizeof(unsigned short) * 2 * (size_t)Mq * F) ||
            d->q16.ensure(sizeof(unsigned short) * 2 * (size_t)Mq * D) || d->ctx16.ensure(sizeof(unsigned short) * 2 * (size_t)Mq * D) ||
            d->mem16.ensure(sizeof(unsigned short) * 2 * (size_t)Mkp * D) || d->dsc.ensure(sizeof(float) * 4) ||
            d->k2.ensure(sizeof(unsigned short) * 2 * ((size_t)Mkp + 32) * D) || d->vt2.ensure(sizeof(unsigned short) * 2 * D * ((size_t)Mkp + 64)) ||
            d->dscl.ensure(sizeof(float) * 4 * c.n_blocks) || d->dlb.ensure(sizeof(float) * 4 * c.n_blocks))
            return -2;
        // rows / columns past the last sequence are read by the last key tile (and masked): keep them finite
        if (d->k2.cap != cap_k) PF_HIP_TRY(hipMemsetAsync(d->k2.p, 0, d->k2.cap, s));
        if (d->vt2.cap != cap_v) PF_HIP_TRY(hipMemsetAsync(d->vt2.p, 0, d->vt2.cap, s));
        for (int l = 0; l < c.n_blocks; ++l)
            if ((rc = dec_layer_x2(d, d->layers[l], dec_layer_prefix(d->contextual, c.n_blocks, l), true, s))) return rc;
        if ((rc = dec_layer_x2(d, d->last, "decoders3.0.", false, s))) return rc;
        if (!d->lb_uploaded) {
            std::vector<float> lb((size_t)4 * c.n_blocks);
            for (int l = 0; l < c.n_blocks; ++l) for (int j = 0; j < 4; ++j) lb[4 * l + j] = d->layers[l].kv_l1b[j];
            PF_HIP_TRY(hipMemcpyAsync(d->dlb.p, lb.data(), sizeof(float) * lb.size(), hipMemcpyHostToDevice, s));
            PF_HIP_TRY(hipStreamSynchronize(s));             // `lb` is a stack object
            d->lb_uploaded = true;
        }
        dsc = d->dsc.as<float>();
        if ((rc = launch_absmax(memory, (size_t)Mk * D, dsc, s))) return rc;
        if ((rc = launch_pow2_scale(dsc, dsc + 1, s))) return rc;
        if ((rc = launch_kv_scales(dsc, d->dlb.as<float>(), c.n_blocks, d->dscl.as<float>(), s))) return rc;
        // memory planes in the padded row layout of attention_f16x2.hip (Tp rows per sequence, padding rows zero)
        if ((rc = launch_split2(memory, D, d->mem16.as<unsigned short>(), D, (size_t)Mkp * D, Mkp, D, 1.f, s, dsc + 1, Tp, T))) return rc;
        mem2 = d->mem16.as<unsigned short>();
    }
    // Token packing (f16x2 greedy route): a batch is padded to its longest hypothesis (N = max token count), but every
    // token-side op is row-wise except the FSMN (per sequence, along tokens) and the attention (per query). So only the
    // VALID token rows are processed, packed back to back: sequence b owns rows [offs[b], offs[b] + tok_lens[b]). The FSMN
    // and the attention kernel take the offsets; ids are scattered back to the caller's [B, N] layout at the end.
    bool pack = false;
    const int* offs_dev = nullptr;
    if (may_pack) {
        int total = 0;
        d->h_offs.assign((size_t)B + 1, 0);
        for (int b = 0; b < B; ++b) { d->h_offs[b] = total; total += tok_lens[b]; }
        d->h_offs[B] = total;
        if (total > 0 && total < Mq_pad) {
            d->h_map.resize((size_t)total);
            for (int b = 0; b < B; ++b) for (int t = 0; t < tok_lens[b]; ++t) d->h_map[(size_t)d->h_offs[b] + t] = b * N + t;
            if (d->offs_dev.ensure(sizeof(int32_t) * ((size_t)B + 1)) || d->map_dev.ensure(sizeof(int32_t) * (size_t)Mq_pad) ||
                d->ids_packed.ensure(sizeof(int32_t) * (size_t)Mq_pad)) return -2;
            if (upload_h2d(d->offs_dev.p, d->h_offs.data(), sizeof(int32_t) * ((size_t)B + 1), s) ||
                upload_h2d(d->map_dev.p, d->h_map.data(), sizeof(int32_t) * (size_t)total, s)) return -2;
            if ((rc = launch_gather_rows(embeds, D, Mq_pad, d->map_dev.as<int>(), x, total, D, s))) return rc;
            pack = true; offs_dev = d->offs_dev.as<int>(); Mq = total;
        }
    }
    if (may_pack && !pack) PF_HIP_TRY(hipMemcpyAsync(x, embeds, sizeof(float) * (size_t)Mq * D, hipMemcpyDeviceToDevice, s));
    if (x3) {
        if (d->t16.ensure(sizeof(unsigned short) * 3 * (size_t)Mq * D) || d->mem16.ensure(sizeof(unsigned short) * 3 * (size_t)Mk * D))
            return -2;
        if ((rc = launch_split3(memory, D, d->mem16.as<unsigned short>(), D, (size_t)Mk * D, Mk, D, s))) return rc;
        mem3 = d->mem16.as<unsigned short>();
    }
    auto w3 = [&](const std::string& name, int rows, int cols) { return x3 ? d->tt.get_split3(name, rows, cols, s) : nullptr; };
    for (int l = 0; l < c.n_blocks; ++l) {
        const DecLayerW& w = d->layers[l];
        const std::string lp = dec_layer_prefix(d->contextual, c.n_blocks, l);
        // DecoderLayerSANM.forward (paraformer/decoder.py:78-121)
        const unsigned short* w1_3 = w3(lp + "feed_forward.w_1.weight", F, D);
        if (x3 && !w1_3) return -2;
        if (x2) rc = dec_ffn_x2(d, w, x, t2, Mq, s);
        else rc = dec_ffn(d, w, x, t2, Mq, s, w1_3);                                          // tgt = FFN(norm1(tgt))
        if (rc) return rc;
        if ((rc = layernorm(t2, D, w.n2g, w.n2b, t1, D, Mq, D, D, c.ln_eps, s))) return rc;   // norm2
        FsmnArgs fa{};                                                                        // x = residual + fsmn
        fa.in = t1; fa.ldin = D; fa.w = w.fsmn_w; fa.R = x; fa.ldr = D; fa.out = x; fa.ldo = D;
        fa.lens = d->tok_lens.as<int>(); fa.B = B; fa.T = N; fa.C = D; fa.K = c.kernel_size; fa.left_pad = left_pad;
        fa.offs = offs_dev;
        if ((rc = fsmn(fa, s))) return rc;
        if (x2) {                                                                             // norm3 -> linear_q
            unsigned short* t2p = d->t16.as<unsigned short>();
            {
                ProfScope ps(PROF_LN, 8.0 * Mq * (double)D, s);
                if ((rc = launch_layernorm(x, D, w.n3g, w.n3b, reinterpret_cast<float*>(t2p), D, Mq, D, D, c.ln_eps, s, 3, 0,
                                           (size_t)Mq * D, pow2f(w.e_n3)))) return rc;
            }
            Gemm2Args g{};                                                                    // q planes, pre-multiplied by d_k^-0.5
            g.A = t2p; g.lda = D; g.a_plane = (size_t)Mq * D; g.W = w.q_2; g.ldw = D; g.w_plane = (size_t)D * D;
            g.oscale = pow2f(-(w.e_n3 + w.ew_q)); g.bias = w.q_b; g.C2 = d->q16.as<unsigned short>(); g.ldc2 = D;
            g.c_plane = (size_t)Mq * D; g.cscale = powf((float)(D / c.n_heads), -0.5f) * pow2f(w.e_q);
            g.M = Mq; g.N = D; g.K = D;
            ProfScope ps(PROF_GEMM3, 2.0 * Mq * (double)D * D, s);
            if ((rc = launch_gemm_f16x2(g, s))) return rc;
        } else {
            if ((rc = layernorm(x, D, w.n3g, w.n3b, t1, D, Mq, D, D, c.ln_eps, s))) return rc;    // norm3
            if ((rc = gemm_simple(t1, D, w.q_w, D, w.q_b, d->q.as<float>(), D, Mq, D, D, 0, nullptr, 0, nullptr, 0, s)))
                return rc;
        }
        if (x2) {
            // linear_k_v in its KV form: K planes and V^T planes straight into the attention kernel's operand layout
            const float* lsc = d->dscl.as<float>() + 4 * l;
            unsigned short* k2 = d->k2.as<unsigned short>();
            unsigned short* vt2 = d->vt2.as<unsigned short>();
            {
                Gemm2Args g{};
                g.A = mem2; g.lda = D; g.a_plane = (size_t)Mkp * D; g.W = w.kv_2; g.ldw = D; g.w_plane = (size_t)2 * D * D;
                g.oscale = pow2f(-w.ew_kv); g.oscale_dev = dsc + 2; g.bias = w.kv_b; g.M = Mkp; g.N = 2 * D; g.K = D;
                g.qkv_D = D; g.kv_form = 1; g.Kp = k2; g.qk_plane = ((size_t)Mkp + 32) * D; g.VT = vt2; g.ldvt = Mkp + 64;
                g.vt_plane = (size_t)D * (Mkp + 64); g.k_mul = 1.f; g.v_mul = 1.f; g.kv_mul_dev = lsc;
                ProfScope ps(PROF_GEMM3, 2.0 * Mkp * 2.0 * D * D, s);
                if ((rc = launch_gemm_f16x2(g, s))) return rc;
            }
            {
                Attn2Args aa{};
                aa.Q = d->q16.as<unsigned short>(); aa.ldq = D; aa.q_plane = (size_t)Mq * D;
                aa.K = k2; aa.ldk = D; aa.k_plane = ((size_t)Mkp + 32) * D; aa.VT = vt2; aa.ldvt = Mkp + 64;
                aa.vt_plane = (size_t)D * (Mkp + 64); aa.O = d->ctx16.as<unsigned short>(); aa.ldo = D; aa.o_plane = (size_t)Mq * D;
                aa.klens = d->mem_lens.as<int>(); aa.B = B; aa.H = c.n_heads; aa.Tp = Tp; aa.Tq = N; aa.qoffs = offs_dev;
                aa.sscale = pow2f(-w.e_q); aa.sscale_dev = lsc + 2; aa.oscale = pow2f(-10);        // ctx planes carry v's scale
                aa.variant = 3;                                                                   // lazy rescale (attention_f16x2.hip)
                ProfScope ps(PROF_ATTN, 4.0 * B * (double)N * T * D, s);
                if ((rc = launch_attention_f16x2(aa, s))) return rc;
            }
            if ((rc = gemm2_simple(d->ctx16.as<unsigned short>(), D, Mq, 0, w.o_2, w.ew_o, w.o_b, x, D, D, D, 0, x, D, s, lsc + 3)))
                return rc;                                                                        // x = residual + att
            continue;
        } else if (x3) {
            if ((rc = gemm3_simple(mem3, D, Mk, w3(lp + "src_attn.linear_k_v.weight", 2 * D, D), w.kv_b, d->kv.as<float>(),
                                   2 * D, 2 * D, D, 0, s))) return rc;
        } else if ((rc = gemm_simple(memory, D, w.kv_w, D, w.kv_b, d->kv.as<float>(), 2 * D, Mk, 2 * D, D, 0, nullptr, 0,
                                     nullptr, 0, s))) return rc;
        const bool ctx_block = cx && l == c.n_blocks - 1;
        if (ctx_block) {
            // x (after the FSMN residual) is x_self_attn: keep it, it is the hotword branch's query and the final residual
            if (d->xself.ensure(sizeof(float) * (size_t)Mq * D) || d->xcat.ensure(sizeof(float) * (size_t)Mq * 2 * D)) return -2;
            PF_HIP_TRY(hipMemcpyAsync(d->xself.p, x, sizeof(float) * (size_t)Mq * D, hipMemcpyDeviceToDevice, s));
        }
        if (l == asf_layer) {
            if (d->asf_p.ensure(sizeof(float) * (size_t)c.n_heads * N * T)) return -2;
            return launch_asf_scores(d->q.as<float>(), D, d->kv.as<float>(), 2 * D, d->asf_p.as<float>(), asf_scores, c.n_heads,
                                     D / c.n_heads, N, T, mem_lens[0], powf((float)(D / c.n_heads), -0.5f), s);
        }
        AttnArgs aa{};
        aa.Q = d->q.as<float>(); aa.ldq = D; aa.K = d->kv.as<float>(); aa.ldk = 2 * D;
        aa.V = d->kv.as<float>() + D; aa.ldv = 2 * D; aa.O = d->ctx.as<float>(); aa.ldo = D;
        aa.klens = d->mem_lens.as<int>(); aa.B = B; aa.H = c.n_heads; aa.Tq = N; aa.Tk = T;
        aa.scale = powf((float)(D / c.n_heads), -0.5f);
        // cross-attention keeps the fp32 MFMA kernel in every fp32-accurate mode: with Tq = tokens (~120) one 128-query
        // block per (utterance, head) is the better shape (102 us vs 111 us for the 256-query split kernel)
        if ((rc = attention(aa, 4.0 * B * (double)N * T * D, s))) return rc;
        if (ctx_block) {
            float* xcat = d->xcat.as<float>();                   // [Mq, 2D]: x_src_attn | cx * clas_scale
            const float* xs = d->xself.as<float>();
            if ((rc = gemm_simple(d->ctx.as<float>(), D, w.o_w, D, w.o_b, xcat, 2 * D, Mq, D, D, 0, nullptr, 0, nullptr, 0, s)))
                return rc;                                                                    // x_src_attn (no residual)
            // bias_decoder: norm3 -> cross-attention over the hotword embeddings (decoder.py:114-130)
            const int Mh = B * cx->n_hot;
            if (d->kv.ensure(sizeof(float) * (size_t)(Mh > Mk ? Mh : Mk) * 2 * D)) return -2;
            std::vector<int32_t> hl((size_t)B, cx->n_hot);
            if ((rc = upload_lens(d->ctx_lens, hl.data(), B, s))) return rc;
            PF_HIP_TRY(hipStreamSynchronize(s));                  // `hl` is a stack object
            if ((rc = layernorm(xs, D, d->tt.get("bias_decoder.norm3.weight"), d->tt.get("bias_decoder.norm3.bias"), t1, D, Mq, D, D,
                                c.ln_eps, s))) return rc;
            if ((rc = gemm_simple(t1, D, d->tt.get("bias_decoder.src_attn.linear_q.weight"), D,
                                  d->tt.get("bias_decoder.src_attn.linear_q.bias"), d->q.as<float>(), D, Mq, D, D, 0, nullptr, 0, nullptr, 0, s))) return rc;
            if ((rc = gemm_simple(cx->info, D, d->tt.get("bias_decoder.src_attn.linear_k_v.weight"), D,
                                  d->tt.get("bias_decoder.src_attn.linear_k_v.bias"), d->kv.as<float>(), 2 * D, Mh, 2 * D, D, 0, nullptr, 0,
                                  nullptr, 0, s))) return rc;
            AttnArgs ab{};
            ab.Q = d->q.as<float>(); ab.ldq = D; ab.K = d->kv.as<float>(); ab.ldk = 2 * D; ab.V = d->kv.as<float>() + D; ab.ldv = 2 * D;
            ab.O = d->ctx.as<float>(); ab.ldo = D; ab.klens = d->ctx_lens.as<int>(); ab.B = B; ab.H = c.n_heads; ab.Tq = N; ab.Tk = cx->n_hot;
            ab.scale = aa.scale;
            if ((rc = attention(ab, 4.0 * B * (double)N * cx->n_hot * D, s))) return rc;
            if ((rc = gemm_simple(d->ctx.as<float>(), D, d->tt.get("bias_decoder.src_attn.linear_out.weight"), D,
                                  d->tt.get("bias_decoder.src_attn.linear_out.bias"), xcat + D, 2 * D, Mq, D, D, 0, nullptr, 0, nullptr, 0, s)))
                return rc;                                                                    // cx
            if (cx->clas_scale != 1.0f && (rc = launch_scale_cols(xcat + D, 2 * D, Mq, D, cx->clas_scale, s))) return rc;
            // bias_output (Conv1d(2D -> D, k = 1, no bias)) and the residual: x = x_self_attn + W [x_src_attn | cx * scale]
            if ((rc = gemm_simple(xcat, 2 * D, d->tt.get("bias_output.weight"), 2 * D, nullptr, x, D, Mq, D, 2 * D, 0, nullptr, 0, xs, D, s)))
                return rc;
            continue;
        }
        if ((rc = gemm_simple(d->ctx.as<float>(), D, w.o_w, D, w.o_b, x, D, Mq, D, D, 0, nullptr, 0, x, D, s)))
            return rc;                                                                        // x = residual + att
    }
    // decoders3: FFN only, no residual (decoder.py:438, DecoderLayerSANM with self_attn = src_attn = None)
    {
        const unsigned short* w1_3 = w3("decoders3.0.feed_forward.w_1.weight", F, D);
        if (x3 && !w1_3) return -2;
        if (x2) rc = dec_ffn_x2(d, d->last, x, t2, Mq, s);
        else rc = dec_ffn(d, d->last, x, t2, Mq, s, w1_3);
        if (rc) return rc;
    }
    float* hid = hidden_out ? hidden_out : d->hid.as<float>();
    if (x2 && V > 0 && ids && !logits && !hidden_out) {
        // greedy route in the f16x2 mode: after_norm writes two-plane operands, the vocabulary projection runs on the fp16
        // matrix cores with the row arg-max fused into its epilogue (no [Mq, V] logits, no fp32 copy of the hidden states)
        int ew_v = 0;
        const unsigned short* wv2 = d->tt.get_split2("output_layer.weight", V, D, &ew_v, s);
        if (!wv2) return -2;
        if (d->e_an == INT32_MIN) {
            float g, b;
            if (TensorTable::dev_absmax(d->tt.get("after_norm.weight"), D, &g, s) || TensorTable::dev_absmax(d->tt.get("after_norm.bias"), D, &b, s)) return -2;
            d->e_an = exp_for_bound(sqrtf((float)D) * g + b);
        }
        unsigned short* h2 = d->t16.as<unsigned short>();
        {
            ProfScope ps(PROF_LN, 8.0 * Mq * (double)D, s);
            if ((rc = launch_layernorm(t2, D, d->tt.get("after_norm.weight"), d->tt.get("after_norm.bias"), reinterpret_cast<float*>(h2), D,
                                       Mq, D, D, c.ln_eps, s, 3, 0, (size_t)Mq * D, pow2f(d->e_an)))) return rc;
        }
        const int nparts = gemm_f16x2_argmax_parts(Mq, V);
        if (d->pval.ensure(sizeof(float) * (size_t)Mq * nparts) || d->pidx.ensure(sizeof(int) * (size_t)Mq * nparts)) return -2;
        Gemm2Args g{};
        g.A = h2; g.lda = D; g.a_plane = (size_t)Mq * D; g.W = wv2; g.ldw = D; g.w_plane = (size_t)V * D;
        g.oscale = pow2f(-(d->e_an + ew_v)); g.bias = d->tt.get("output_layer.bias"); g.M = Mq; g.N = V; g.K = D;
        g.amax_val = d->pval.as<float>(); g.amax_idx = d->pidx.as<int>(); g.amax_ld = nparts;
        {
            ProfScope ps(PROF_GEMM3, 2.0 * Mq * (double)V * D, s);
            if ((rc = launch_gemm_f16x2(g, s))) return rc;
        }
        if (!pack) return launch_argmax_reduce(d->pval.as<float>(), d->pidx.as<int>(), nparts, nparts, ids, nullptr, Mq, s);
        if ((rc = launch_argmax_reduce(d->pval.as<float>(), d->pidx.as<int>(), nparts, nparts, d->ids_packed.as<int>(), nullptr, Mq, s))) return rc;
        PF_HIP_TRY(hipMemsetAsync(ids, 0, sizeof(int32_t) * (size_t)Mq_pad, s));       // padding positions: id 0, like an untouched row
        return launch_scatter_i32(d->ids_packed.as<int>(), d->map_dev.as<int>(), ids, Mq, s);
    }
    if ((rc = layernorm(t2, D, d->tt.get("after_norm.weight"), d->tt.get("after_norm.bias"), hid, D, Mq, D, D,
                        c.ln_eps, s))) return rc;
    if (V == 0) return 0;
    return vocab_project(hid, Mq, D, d->tt.get("output_layer.weight"), d->tt.get("output_layer.bias"), V, logits, ids,
                         d->pval, d->pidx, s);
}

// ------------------------------------------------------------------------------------------------------- ctc
pf_ctc* pf_ctc_create(int32_t d_model, int32_t vocab) {
    if (check_device()) return nullptr;
    if (d_model <= 0 || d_model % 32 || vocab <= 0) { set_error("ctc: d_model % 32 == 0 required"); return nullptr; }
    std::unique_ptr<Ctc> c(new Ctc());
    c->d_model = d_model; c->vocab = vocab;
    c->precision = 3;                        // the fused arg-max route on the fp16 matrix cores (pf_ctc_set_precision(c, 0): fp32 MFMA)
    if (c->tt.add("ctc_lo.weight", (int64_t)vocab * d_model) || c->tt.add("ctc_lo.bias", vocab)) return nullptr;
    return reinterpret_cast<pf_ctc*>(c.release());
}
void pf_ctc_destroy(pf_ctc* c) { delete reinterpret_cast<Ctc*>(c); }
int pf_ctc_set_tensor(pf_ctc* ch, const char* name, const float* data, int64_t numel) {
    Ctc* c = reinterpret_cast<Ctc*>(ch);
    PF_REQUIRE(c && name && data, "ctc_set_tensor: null");
    c->tt.drop_bf16();          // the f16x2 arg-max route caches weight planes: they follow the fp32 master
    return c->tt.set(name, data, numel);
}
int pf_ctc_missing(const pf_ctc* ch) {
    const Ctc* c = reinterpret_cast<const Ctc*>(ch);
    return c ? c->tt.missing() : -1;
}
/* 0 = fp32 MFMA (default); 3 = the arg-max route (logits_dev == NULL) on the fp16 matrix cores from two-plane operands
 * (gemm_f16x2.hip): the hidden states' plane scale is chosen on the device from max |hidden|, fp32-class logits */
int pf_ctc_set_precision(pf_ctc* ch, int32_t mode) {
    Ctc* c = reinterpret_cast<Ctc*>(ch);
    PF_REQUIRE(c && (mode == 0 || mode == 3), "ctc_set_precision: mode must be 0 (fp32 MFMA) or 3 (fp32 via f16x2)");
    c->precision = mode;
    return 0;
}
int pf_ctc_greedy(pf_ctc* ch, const float* hidden, int32_t M, int32_t* ids, float* logits, void* stream) {
    Ctc* c = reinterpret_cast<Ctc*>(ch);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(c && hidden && M > 0, "ctc_greedy: null/empty");
    std::string first;
    if (c->tt.missing(&first)) { set_error("ctc: tensor not set: " + first); return -3; }
    if (c->precision == 3 && ids && !logits && !g_stream_mode && c->d_model % 32 == 0) {
        const int D = c->d_model, V = c->vocab;
        int ew = 0, rc;
        const unsigned short* w2 = c->tt.get_split2("ctc_lo.weight", V, D, &ew, s);
        if (!w2) return -2;
        const int nparts = gemm_f16x2_argmax_parts(M, V);
        if (c->h2.ensure(sizeof(unsigned short) * 2 * (size_t)M * D) || c->dsc.ensure(sizeof(float) * 4) ||
            c->pval.ensure(sizeof(float) * (size_t)M * nparts) || c->pidx.ensure(sizeof(int) * (size_t)M * nparts)) return -2;
        float* dsc = c->dsc.as<float>();
        if ((rc = launch_absmax(hidden, (size_t)M * D, dsc, s))) return rc;
        if ((rc = launch_pow2_scale(dsc, dsc + 1, s))) return rc;
        if ((rc = launch_split2(hidden, D, c->h2.as<unsigned short>(), D, (size_t)M * D, M, D, 1.f, s, dsc + 1))) return rc;
        Gemm2Args g{};
        g.A = c->h2.as<unsigned short>(); g.lda = D; g.a_plane = (size_t)M * D; g.W = w2; g.ldw = D; g.w_plane = (size_t)V * D;
        g.oscale = pow2f(-ew); g.oscale_dev = dsc + 2; g.bias = c->tt.get("ctc_lo.bias"); g.M = M; g.N = V; g.K = D;
        g.amax_val = c->pval.as<float>(); g.amax_idx = c->pidx.as<int>(); g.amax_ld = nparts;
        {
            ProfScope ps(PROF_GEMM3, 2.0 * M * (double)V * D, s);
            if ((rc = launch_gemm_f16x2(g, s))) return rc;
        }
        return launch_argmax_reduce(c->pval.as<float>(), c->pidx.as<int>(), nparts, nparts, ids, nullptr, M, s);
    }
    return vocab_project(hidden, M, c->d_model, c->tt.get("ctc_lo.weight"), c->tt.get("ctc_lo.bias"), c->vocab, logits,
                         ids, c->pval, c->pidx, s);
}

// -------------------------------------------------------------------------------------------------------- vad
// FSMN-VAD network (funasr/models/fsmn_vad_streaming/encoder.py:288-378): in_linear1 -> in_linear2 -> relu ->
// n x [linear (no bias) -> FSMN memory (+ cache) -> affine -> relu] -> out_linear1 -> out_linear2 -> softmax, reduced to
// the summed posterior of the silence pdfs. Dense layers on the fp32 GEMM kernels (K padded to 32 with zero columns),
// the memory / softmax kernels in vad.hip.
struct Vad {
    pf_vad_config cfg;
    TensorTable tt;
    DevBuf a, b, c, cache_tmp, ids;
    size_t zeroed_for = 0;
};
static int vad_pad(int k) { return round_up(k, 32); }

pf_vad* pf_vad_create(const pf_vad_config* cfg) {
    if (!cfg) { set_error("vad: null config"); return nullptr; }
    if (check_device()) return nullptr;
    const pf_vad_config& c = *cfg;
    if (c.input_dim <= 0 || c.input_affine_dim <= 0 || c.linear_dim <= 0 || c.proj_dim <= 0 || c.proj_dim % 4 ||
        c.proj_dim > 128 || c.fsmn_layers < 1 || c.lorder < 1 || c.lstride < 1 || c.rorder != 0 || c.output_affine_dim <= 0 ||
        c.output_dim <= 0 || c.output_dim > 512) {
        set_error("vad: unsupported config (uni-directional FSMN: rorder 0; proj_dim % 4 == 0 and <= 128; output_dim <= 512)");
        return nullptr;
    }
    std::unique_ptr<Vad> v(new Vad());
    v->cfg = c;
    int rc = 0;
    rc |= v->tt.add_padded("in_linear1.linear.weight", c.input_affine_dim, c.input_dim, vad_pad(c.input_dim));
    rc |= v->tt.add("in_linear1.linear.bias", c.input_affine_dim);
    rc |= v->tt.add_padded("in_linear2.linear.weight", c.linear_dim, c.input_affine_dim, vad_pad(c.input_affine_dim));
    rc |= v->tt.add("in_linear2.linear.bias", c.linear_dim);
    for (int i = 0; i < c.fsmn_layers; ++i) {
        const std::string p = "fsmn." + std::to_string(i) + ".";
        rc |= v->tt.add_padded(p + "linear.linear.weight", c.proj_dim, c.linear_dim, vad_pad(c.linear_dim));
        rc |= v->tt.add(p + "fsmn_block.conv_left.weight", (int64_t)c.proj_dim * c.lorder);
        rc |= v->tt.add_padded(p + "affine.linear.weight", c.linear_dim, c.proj_dim, vad_pad(c.proj_dim));
        rc |= v->tt.add(p + "affine.linear.bias", c.linear_dim);
    }
    rc |= v->tt.add_padded("out_linear1.linear.weight", c.output_affine_dim, c.linear_dim, vad_pad(c.linear_dim));
    rc |= v->tt.add("out_linear1.linear.bias", c.output_affine_dim);
    rc |= v->tt.add_padded("out_linear2.linear.weight", c.output_dim, c.output_affine_dim, vad_pad(c.output_affine_dim));
    rc |= v->tt.add("out_linear2.linear.bias", c.output_dim);
    if (rc) return nullptr;
    return reinterpret_cast<pf_vad*>(v.release());
}
void pf_vad_destroy(pf_vad* v) { delete reinterpret_cast<Vad*>(v); }
int pf_vad_set_tensor(pf_vad* vh, const char* name, const float* data, int64_t numel) {
    Vad* v = reinterpret_cast<Vad*>(vh);
    PF_REQUIRE(v && name && data, "vad_set_tensor: null");
    return v->tt.set(name, data, numel);
}
int pf_vad_missing(const pf_vad* vh) {
    const Vad* v = reinterpret_cast<const Vad*>(vh);
    return v ? v->tt.missing() : -1;
}
/* feats_dev [B, T, input_dim]; cache_dev [B, fsmn_layers, (lorder-1)*lstride, proj_dim] is the left context of every
 * memory block, read and updated in place (NULL: zero left context, nothing kept); sil_ids_host: the silence pdfs whose
 * posteriors are summed into p_sil_dev [B, T]; probs_dev [B, T, output_dim] (optional) receives the full softmax.
 * small_m != 0 routes the dense layers through the weight-streaming GEMM (streaming chunks of a few frames). */
int pf_vad_forward(pf_vad* vh, const float* feats, int32_t B, int32_t T, float* cache, const int32_t* sil_ids_host,
                   int32_t n_sil, float* p_sil, float* probs, int32_t small_m, void* stream) {
    Vad* v = reinterpret_cast<Vad*>(vh);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(v && feats && p_sil && sil_ids_host && B > 0 && T > 0 && n_sil >= 1, "vad_forward: null/empty argument");
    std::string first;
    if (v->tt.missing(&first)) { set_error("vad: tensor not set: " + first); return -3; }
    const pf_vad_config& c = v->cfg;
    const int M = B * T;
    const int Kin = vad_pad(c.input_dim), Ka = vad_pad(c.input_affine_dim), Kl = vad_pad(c.linear_dim), Kp = vad_pad(c.proj_dim),
              Ko = vad_pad(c.output_affine_dim);
    int wide = Kin;
    for (int k : {Ka, Kl, Kp, Ko, round_up(c.output_dim, 4)}) wide = k > wide ? k : wide;
    const size_t bytes = sizeof(float) * (size_t)M * wide;
    const int ctx = (c.lorder - 1) * c.lstride;
    if (v->a.ensure(bytes) || v->b.ensure(bytes) || v->c.ensure(bytes) ||
        v->cache_tmp.ensure(sizeof(float) * (size_t)B * (ctx > 0 ? ctx : 1) * c.proj_dim))
        return -2;
    // the GEMMs write N columns of a row whose stride is the NEXT layer's padded K: the pad columns must read as zero
    PF_HIP_TRY(hipMemsetAsync(v->a.p, 0, bytes, s));
    PF_HIP_TRY(hipMemsetAsync(v->b.p, 0, bytes, s));
    PF_HIP_TRY(hipMemsetAsync(v->c.p, 0, bytes, s));
    float* a = v->a.as<float>();
    float* b = v->b.as<float>();
    float* cc = v->c.as<float>();
    int rc;
    std::unique_ptr<StreamModeScope> sm(small_m ? new StreamModeScope() : nullptr);
    // stage the features with a padded row stride
    PF_HIP_TRY(hipMemcpy2DAsync(a, sizeof(float) * Kin, feats, sizeof(float) * c.input_dim, sizeof(float) * c.input_dim, M,
                                hipMemcpyDeviceToDevice, s));
    auto lin = [&](const float* x, int K, const char* wname, const char* bname, float* y, int ldy, int N, int relu) {
        return gemm_simple(x, K, v->tt.get(wname), K, bname ? v->tt.get(bname) : nullptr, y, ldy, M, N, K, relu, nullptr, 0,
                           nullptr, 0, s);
    };
    if ((rc = lin(a, Kin, "in_linear1.linear.weight", "in_linear1.linear.bias", b, Ka, c.input_affine_dim, 0))) return rc;
    if ((rc = lin(b, Ka, "in_linear2.linear.weight", "in_linear2.linear.bias", a, Kl, c.linear_dim, 1))) return rc;
    // a: [M, Kl] holds the block input
    for (int i = 0; i < c.fsmn_layers; ++i) {
        const std::string p = "fsmn." + std::to_string(i) + ".";
        if ((rc = lin(a, Kl, (p + "linear.linear.weight").c_str(), nullptr, b, Kp, c.proj_dim, 0))) return rc;
        float* lc = cache ? cache + (size_t)i * ctx * c.proj_dim : nullptr;       // layer i of stream 0; streams are
        if (cache && B > 1) {                                                      // [B, layers, ctx, proj] apart
            // per-stream caches are not contiguous per layer: run the memory block stream by stream
            for (int bb = 0; bb < B; ++bb) {
                float* sc = cache + ((size_t)bb * c.fsmn_layers + i) * ctx * c.proj_dim;
                if ((rc = launch_vad_fsmn(b + (size_t)bb * T * Kp, Kp, v->tt.get(p + "fsmn_block.conv_left.weight"), sc,
                                          v->cache_tmp.as<float>(), cc + (size_t)bb * T * Kp, Kp, 1, T, c.proj_dim, c.lorder,
                                          c.lstride, s))) return rc;
                if (ctx > 0)
                    PF_HIP_TRY(hipMemcpyAsync(sc, v->cache_tmp.p, sizeof(float) * (size_t)ctx * c.proj_dim,
                                              hipMemcpyDeviceToDevice, s));
            }
        } else {
            if ((rc = launch_vad_fsmn(b, Kp, v->tt.get(p + "fsmn_block.conv_left.weight"), lc, lc ? v->cache_tmp.as<float>() : nullptr,
                                      cc, Kp, B, T, c.proj_dim, c.lorder, c.lstride, s))) return rc;
            if (lc && ctx > 0)
                PF_HIP_TRY(hipMemcpyAsync(lc, v->cache_tmp.p, sizeof(float) * (size_t)ctx * c.proj_dim, hipMemcpyDeviceToDevice, s));
        }
        if ((rc = lin(cc, Kp, (p + "affine.linear.weight").c_str(), (p + "affine.linear.bias").c_str(), a, Kl, c.linear_dim, 1)))
            return rc;
    }
    if ((rc = lin(a, Kl, "out_linear1.linear.weight", "out_linear1.linear.bias", b, Ko, c.output_affine_dim, 0))) return rc;
    const int ldo = round_up(c.output_dim, 4);
    if ((rc = lin(b, Ko, "out_linear2.linear.weight", "out_linear2.linear.bias", cc, ldo, c.output_dim, 0))) return rc;
    return launch_vad_softmax_sil(cc, ldo, M, c.output_dim, sil_ids_host, n_sil, p_sil, probs, c.output_dim, s);
}
/* 10 log10(sum(x^2) + 1e-6) of n_frames frames of frame_len samples, frame_shift apart (ComputeDecibel, model.py:513-530) */
int pf_vad_frame_decibel(const float* wav, int64_t n_samples, int32_t n_frames, int32_t frame_len, int32_t frame_shift,
                         float* out, void* stream) {
    PF_REQUIRE(wav && out, "vad_frame_decibel: null");
    PF_REQUIRE(n_frames > 0 && (int64_t)(n_frames - 1) * frame_shift + frame_len <= n_samples,
               "vad_frame_decibel: the frames reach past the end of the waveform");
    return launch_frame_decibel(wav, n_frames, frame_len, frame_shift, out, reinterpret_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------------------------------- streaming
pf_stream* pf_stream_create(pf_encoder* eh, pf_predictor* ph, pf_decoder* dh, const pf_stream_config* cfg) {
    if (!eh || !ph || !dh || !cfg) { set_error("stream: null argument"); return nullptr; }
    if (check_device()) return nullptr;
    Encoder* e = reinterpret_cast<Encoder*>(eh);
    Predictor* p = reinterpret_cast<Predictor*>(ph);
    Decoder* d = reinterpret_cast<Decoder*>(dh);
    const pf_stream_config& c = *cfg;
    const int K = d->cfg.kernel_size;
    const int dec_left = (K - 1) / 2 + (d->cfg.sanm_shift > 0 ? d->cfg.sanm_shift : 0);
    if (c.n_streams < 1 || c.chunk_left < 0 || c.chunk_cur < 1 || c.chunk_right < 0 || c.enc_look_back < 0 ||
        c.dec_look_back < 0 || c.max_frames < c.chunk_cur || c.max_tokens < 1 || c.max_tokens > 96 ||
        e->cfg.tp_blocks != 0 || e->cfg.d_model != 512 || d->cfg.d_model != 512 || p->cfg.d_model != 512 ||
        dec_left != K - 1) {
        set_error("stream: unsupported config (d_model 512, look_back >= 0 (finite), max_tokens <= 96, causal decoder "
                  "FSMN i.e. sanm_shfit == (kernel_size-1)/2 as in paraformer_streaming/template.yaml:62)");
        return nullptr;
    }
    // CIF fires at most once per frame carrying weight (alpha < 1): a step's window has chunk_left + chunk_right + n frames,
    // the first chunk_left of them are zeroed (cif_predictor.py:343-346), plus the carried remainder and the final tail
    // weight (:347-357). The token capacity must cover that, or tokens would be dropped silently (stream.hip clamps
    // n_fired); the hipGraph cache key packs n_frames into 10 bits
    if (c.max_tokens < c.chunk_right + c.max_frames + 2 || c.max_frames >= 1024) {
        set_error("stream: max_tokens (" + std::to_string(c.max_tokens) + "; the decoder's token rows per step are capped at 96) must cover chunk_right + max_frames + 2 = " +
                  std::to_string(c.chunk_right + c.max_frames + 2) + " possible fires per step; max_frames must stay below 1024");
        return nullptr;
    }
    if (c.chunk_left + c.chunk_right == 0) {
        // the reference keeps x[:, -(chunk_size[0] + chunk_size[2]):] as the overlap window (scama/encoder.py:480-494): with both 0
        // that is x[:, -0:] = the WHOLE window, so its window grows by every chunk and the CIF mask keeps decoding the first
        // chunk_size[1] frames -- a degenerate session this handle does not reproduce; refuse instead of differing silently
        set_error("stream: chunk_size[0] + chunk_size[2] == 0 is not supported (the reference's overlap window x[:, -0:] is the whole history in that geometry)");
        return nullptr;
    }
    int rc;
    if (!e->resolved && (rc = encoder_resolve(e))) return nullptr;
    if (!d->resolved && (rc = decoder_resolve(d))) return nullptr;
    { std::string first; if (p->tt.missing(&first)) { set_error("predictor: tensor not set: " + first); return nullptr; } }
    std::unique_ptr<Stream> st(new Stream());
    st->e = e; st->p = p; st->d = d; st->cfg = c;
    st->S = c.n_streams; st->keep = c.chunk_left + c.chunk_right; st->Wmax = st->keep + c.max_frames;
    st->Nmax = c.max_tokens; st->enc_cap = c.enc_look_back * c.chunk_cur; st->dec_cap = c.dec_look_back * c.chunk_cur;
    st->use_graph = c.use_graph != 0;
    const int S = st->S, D = 512, Din = e->cfg.input_dim;
    const size_t L = e->layers.size(), Ld = (size_t)d->cfg.n_blocks;
    bool bad = false;
    bad |= st->dev_state.ensure(sizeof(StreamDev)) != 0;
    bad |= st->cache_feats.ensure(sizeof(float) * (size_t)S * (st->keep > 0 ? st->keep : 1) * Din) != 0;
    bad |= st->feats_in.ensure(sizeof(float) * (size_t)S * c.max_frames * Din) != 0;
    bad |= st->win.ensure(sizeof(float) * (size_t)S * st->Wmax * Din) != 0;
    bad |= st->enc_ring.ensure(sizeof(float) * (L * S * (st->enc_cap > 0 ? st->enc_cap : 1) * 2 * D)) != 0;
    bad |= st->dec_ring.ensure(sizeof(float) * (Ld * S * (st->dec_cap > 0 ? st->dec_cap : 1) * 2 * D)) != 0;
    bad |= st->dec_fsmn.ensure(sizeof(float) * (Ld * S * (K - 1) * D)) != 0;
    bad |= st->cif_hidden.ensure(sizeof(float) * (size_t)S * D) != 0;
    bad |= st->cif_alpha.ensure(sizeof(float) * (size_t)S) != 0;
    bad |= st->dec_valid.ensure(sizeof(int) * (size_t)S) != 0;
    bad |= st->dec_wp.ensure(sizeof(int) * (size_t)S) != 0;
    bad |= st->n_fired.ensure(sizeof(int) * (size_t)S) != 0;
    bad |= st->lensW.ensure(sizeof(int) * (size_t)S) != 0;
    bad |= st->enc_out.ensure(sizeof(float) * (size_t)S * st->Wmax * D) != 0;
    bad |= st->embeds.ensure(sizeof(float) * (size_t)S * st->Nmax * D) != 0;
    bad |= st->ids.ensure(sizeof(int32_t) * (size_t)S * st->Nmax) != 0;
    bad |= st->alphas.ensure(sizeof(float) * (size_t)S * (st->Wmax + 1)) != 0;
    if (bad) return nullptr;
    if (hipHostMalloc((void**)&st->h_ids, sizeof(int32_t) * (size_t)S * st->Nmax) != hipSuccess ||
        hipHostMalloc((void**)&st->h_n, sizeof(int32_t) * (size_t)S) != hipSuccess ||
        hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&st->ev, hipEventDisableTiming) != hipSuccess) {
        set_error("stream: pinned buffer / stream creation failed");
        return nullptr;
    }
    // default position table (libm); the Python mirror replaces it with the torch-evaluated one for bit-exactness
    {
        const int rows = 4096, half = Din / 2;
        std::vector<float> tab((size_t)rows * Din);
        const float inc = logf(10000.0f) / (float)(half - 1);
        for (int t = 0; t < rows; ++t)
            for (int i = 0; i < half; ++i) {
                const float sc = (float)(t + 1) * expf((float)i * (-inc));
                tab[(size_t)t * Din + i] = sinf(sc);
                tab[(size_t)t * Din + half + i] = cosf(sc);
            }
        if (st->pe.ensure(sizeof(float) * tab.size())) return nullptr;
        if (hipMemcpy(st->pe.p, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        st->pe_rows = rows;
    }
    if (stream_reset(st.get(), st->stream) || hipStreamSynchronize(st->stream) != hipSuccess) return nullptr;
    return reinterpret_cast<pf_stream*>(st.release());
}

void pf_stream_destroy(pf_stream* s) { delete reinterpret_cast<Stream*>(s); }

int pf_stream_set_pe(pf_stream* sh, const float* pe, int32_t rows) {
    Stream* st = reinterpret_cast<Stream*>(sh);
    PF_REQUIRE(st && pe && rows > 0, "stream_set_pe: null/empty");
    PF_HIP_TRY(hipStreamSynchronize(st->stream));
    const size_t bytes = sizeof(float) * (size_t)rows * st->e->cfg.input_dim;
    const void* old = st->pe.p;
    if (st->pe.ensure(bytes)) return -2;
    if (st->pe.p != old) {                     // the table moved: captured graphs hold the old pointer
        for (auto& kv : st->graphs) (void)hipGraphExecDestroy(kv.second);
        st->graphs.clear();
    }
    PF_HIP_TRY(hipMemcpy(st->pe.p, pe, bytes, hipMemcpyDefault));
    st->pe_rows = rows;
    return 0;
}

/* "gemm_mode": 0 = the step's GEMMs on the fp32 weight-streaming / fp32-MFMA kernels (default: the latency path of a few
 * streams), 3 = on the fp16 matrix cores with two-plane operands (gemm_f16x2.hip: fp32-class results, the throughput path of
 * many lock-step streams). Prepares the weight planes (synchronises), drops the captured graphs. */
int pf_stream_set_option(pf_stream* sh, const char* key, int32_t value) {
    Stream* st = reinterpret_cast<Stream*>(sh);
    PF_REQUIRE(st && key, "stream_set_option: null");
    const std::string k = key;
    if (k != "gemm_mode") { set_error("stream_set_option: unknown key " + k); return -1; }
    PF_REQUIRE(value == 0 || value == 3, "stream_set_option: gemm_mode is 0 (fp32 kernels) or 3 (f16x2)");
    PF_HIP_TRY(hipStreamSynchronize(st->stream));
    if (value == 3) {
        int rc = stream_prepare_x2(st, st->stream);
        if (rc) return rc;
    }
    st->x2 = value == 3;
    for (auto& kv : st->graphs) (void)hipGraphExecDestroy(kv.second);
    st->graphs.clear();
    st->seen.clear();
    return 0;
}

int pf_stream_reset(pf_stream* sh, void* stream) {
    Stream* st = reinterpret_cast<Stream*>(sh);
    PF_REQUIRE(st, "stream_reset: null");
    int rc = stream_reset(st, st->stream);
    if (rc) return rc;
    PF_HIP_TRY(hipStreamSynchronize(st->stream));
    return 0;
}

int pf_stream_step(pf_stream* sh, const float* feats, int32_t n_frames, int32_t is_final, int32_t tail_chunk,
                   int32_t* ids_host, int32_t* n_tokens_host, float* enc_out, void* stream) {
    Stream* st = reinterpret_cast<Stream*>(sh);
    hipStream_t us = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(st && ids_host && n_tokens_host, "stream_step: null argument");
    const int n = tail_chunk ? 0 : n_frames;
    PF_REQUIRE(tail_chunk || (feats && n >= 1 && n <= st->cfg.max_frames), "stream_step: n_frames out of range");
    PF_REQUIRE(!tail_chunk || st->keep > 0, "stream_step: a tail chunk needs chunk_left + chunk_right > 0");
    PF_REQUIRE(st->start_idx + n <= st->pe_rows, "stream_step: position table exhausted (pf_stream_set_pe with more rows)");
    const int S = st->S, Din = st->e->cfg.input_dim, D = 512;
    const int W = tail_chunk ? st->keep : st->keep + n;
    hipStream_t s = st->stream;
    // order after whatever produced `feats` on the caller's stream
    PF_HIP_TRY(hipEventRecord(st->ev, us));
    PF_HIP_TRY(hipStreamWaitEvent(s, st->ev, 0));
    if (!tail_chunk)
        PF_HIP_TRY(hipMemcpyAsync(st->feats_in.p, feats, sizeof(float) * (size_t)S * n * Din, hipMemcpyDeviceToDevice, s));
    const int key = n | (is_final ? 1 << 10 : 0) | (tail_chunk ? 1 << 11 : 0);
    int rc;
    if (st->x2 && !stream_x2_ready(st)) {
        // a handle's weights changed since the planes / exponents were prepared (the old planes are freed): prepare again,
        // drop the graphs that point at them
        for (auto& kv : st->graphs) (void)hipGraphExecDestroy(kv.second);
        st->graphs.clear();
        st->seen.clear();
        if ((rc = stream_prepare_x2(st, s))) return rc;
    }
    const bool graphable = st->use_graph && !g_prof_on;
    if (st->graph_epoch != g_ws_epoch) {
        // a workspace of the encoder / predictor / decoder handles moved since the capture (e.g. an offline batch grew
        // it): the graphs hold stale pointers -- drop them, run this step eagerly, capture again next time
        for (auto& kv : st->graphs) (void)hipGraphExecDestroy(kv.second);
        st->graphs.clear();
        st->seen.clear();
        st->graph_epoch = g_ws_epoch;
    }
    if (graphable && st->seen[key] >= 1) {
        auto it = st->graphs.find(key);
        if (it == st->graphs.end()) {
            hipGraph_t graph = nullptr;
            PF_HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            rc = stream_enqueue(st, n, is_final, tail_chunk, s);
            hipError_t ce = hipStreamEndCapture(s, &graph);
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            if (ce != hipSuccess) { set_error(std::string("stream: graph capture failed: ") + hipGetErrorString(ce)); return -2; }
            hipGraphExec_t exec = nullptr;
            PF_HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
            it = st->graphs.emplace(key, exec).first;
        }
        PF_HIP_TRY(hipGraphLaunch(it->second, s));
    } else {
        if ((rc = stream_enqueue(st, n, is_final, tail_chunk, s))) return rc;
        st->seen[key] += 1;
        st->graph_epoch = g_ws_epoch;                        // allocations of this eager pass are accounted for
    }
    if (enc_out)
        PF_HIP_TRY(hipMemcpyAsync(enc_out, st->enc_out.p, sizeof(float) * (size_t)S * W * D, hipMemcpyDeviceToDevice, s));
    PF_HIP_TRY(hipStreamSynchronize(s));
    st->start_idx += tail_chunk ? st->keep : n;
    memcpy(ids_host, st->h_ids, sizeof(int32_t) * (size_t)S * st->Nmax);
    memcpy(n_tokens_host, st->h_n, sizeof(int32_t) * (size_t)S);
    return 0;
}

int pf_stream_peek(pf_stream* sh, float* cif_alpha_host, float* cif_hidden_host, int32_t* start_idx_host) {
    Stream* st = reinterpret_cast<Stream*>(sh);
    PF_REQUIRE(st, "stream_peek: null");
    PF_HIP_TRY(hipStreamSynchronize(st->stream));
    if (cif_alpha_host) PF_HIP_TRY(hipMemcpy(cif_alpha_host, st->cif_alpha.p, sizeof(float) * st->S, hipMemcpyDeviceToHost));
    if (cif_hidden_host) PF_HIP_TRY(hipMemcpy(cif_hidden_host, st->cif_hidden.p, sizeof(float) * (size_t)st->S * 512, hipMemcpyDeviceToHost));
    if (start_idx_host) {
        StreamDev d{};
        PF_HIP_TRY(hipMemcpy(&d, st->dev_state.p, sizeof(StreamDev), hipMemcpyDeviceToHost));
        *start_idx_host = d.start_idx;
    }
    return 0;
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
    return launch_fbank(a, 1, nfr, s);
}

// -------------------------------------------------------------------------------------------- single kernels
/* test hook: pf_k_gemm_f32 takes the small-M kernel for M <= m (default 0 = always the tile kernel) */
int pf_set_skinny_max_m(int32_t m) { g_skinny_max_m = m; return 0; }

int pf_k_gemm_f32(const float* A, int32_t lda, const float* W, int32_t ldw, const float* bias, const float* R1,
                  int32_t ldr1, const float* R2, int32_t ldr2, float* C, int32_t ldc, int32_t M, int32_t N, int32_t K,
                  int32_t relu, void* stream) {
    return gemm_simple(A, lda, W, ldw, bias, C, ldc, M, N, K, relu, R1, ldr1, R2, ldr2,
                       reinterpret_cast<hipStream_t>(stream));
}
/* bf16-operand GEMM (throughput mode): A [M,K] bf16, W [N,K] bf16, fp32 accumulate, fp32 bias/residuals, C fp32 or
 * bf16 (c_bf16); strides in elements */
int pf_k_gemm_bf16(const void* A, int32_t lda, const void* W, int32_t ldw, const float* bias, const float* R1,
                   int32_t ldr1, const float* R2, int32_t ldr2, void* C, int32_t ldc, int32_t M, int32_t N, int32_t K,
                   int32_t relu, int32_t c_bf16, void* stream) {
    GemmArgs g{};
    g.A = reinterpret_cast<const float*>(A); g.lda = lda; g.W = reinterpret_cast<const float*>(W); g.ldw = ldw;
    g.bias = bias; g.R1 = R1; g.ldr1 = ldr1; g.R2 = R2; g.ldr2 = ldr2; g.C = reinterpret_cast<float*>(C); g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.relu = relu; g.ab_bf16 = 1; g.c_bf16 = c_bf16;
    return launch_gemm_f32(g, reinterpret_cast<hipStream_t>(stream));
}
int pf_k_gemm_bf16_time(const void* A, int32_t lda, const void* W, int32_t ldw, const float* bias, void* C, int32_t ldc,
                        int32_t M, int32_t N, int32_t K, int32_t c_bf16, int32_t iters, float* ms_out, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(iters > 0 && ms_out, "gemm_time: iters > 0");
    GemmArgs g{};
    g.A = reinterpret_cast<const float*>(A); g.lda = lda; g.W = reinterpret_cast<const float*>(W); g.ldw = ldw;
    g.bias = bias; g.C = reinterpret_cast<float*>(C); g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.ab_bf16 = 1; g.c_bf16 = c_bf16;
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
int pf_k_split3(const float* x, int32_t ldx, void* y3, int32_t ldy, int64_t plane, int32_t M, int32_t N, void* stream) {
    return launch_split3(x, ldx, reinterpret_cast<unsigned short*>(y3), ldy, (size_t)plane, M, N,
                         reinterpret_cast<hipStream_t>(stream));
}
/* iters > 0 with ms_out: additionally times `iters` back-to-back launches (after 3 warm-up launches) */
int pf_k_gemm_split3(const void* A3, int32_t lda, int64_t a_plane, const void* W3, int32_t ldw, int64_t w_plane,
                     const float* bias, const float* R1, int32_t ldr1, const float* R2, int32_t ldr2, float* C,
                     int32_t ldc, void* C3, int32_t ldc3, int64_t c_plane, int32_t M, int32_t N, int32_t K,
                     int32_t relu, int32_t iters, float* ms_out, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    Gemm3Args g{};
    g.A = reinterpret_cast<const unsigned short*>(A3); g.lda = lda; g.a_plane = (size_t)a_plane;
    g.W = reinterpret_cast<const unsigned short*>(W3); g.ldw = ldw; g.w_plane = (size_t)w_plane;
    g.bias = bias; g.R1 = R1; g.ldr1 = ldr1; g.R2 = R2; g.ldr2 = ldr2; g.C = C; g.ldc = ldc;
    g.C3 = reinterpret_cast<unsigned short*>(C3); g.ldc3 = ldc3; g.c_plane = (size_t)c_plane;
    g.M = M; g.N = N; g.K = K; g.relu = relu;
    int rc;
    if (iters <= 0 || !ms_out) return launch_gemm_split3(g, s);
    for (int i = 0; i < 3; ++i) if ((rc = launch_gemm_split3(g, s))) return rc;
    hipEvent_t a, b;
    PF_HIP_TRY(hipEventCreate(&a));
    PF_HIP_TRY(hipEventCreate(&b));
    PF_HIP_TRY(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) if ((rc = launch_gemm_split3(g, s))) return rc;
    PF_HIP_TRY(hipEventRecord(b, s));
    PF_HIP_TRY(hipEventSynchronize(b));
    float ms = 0.f;
    PF_HIP_TRY(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms_out = ms / iters;
    return 0;
}
/* fp32 [M, N] * scale (a power of two) -> two fp16 planes [2][M, ldy] (gemm_f16x2.hip) */
int pf_k_split2(const float* x, int32_t ldx, void* y2, int32_t ldy, int64_t plane, int32_t M, int32_t N, float scale,
                void* stream) {
    return launch_split2(x, ldx, reinterpret_cast<unsigned short*>(y2), ldy, (size_t)plane, M, N, scale,
                         reinterpret_cast<hipStream_t>(stream));
}
/* fp32-accurate GEMM from two-plane fp16 operands; tile: 0 by shape, 1 = 256 x 128, 2 = 256 x 256;
 * iters > 0 with ms_out: additionally times `iters` back-to-back launches (after 3 warm-up launches) */
int pf_k_gemm_f16x2(const void* A2, int32_t lda, int64_t a_plane, const void* W2, int32_t ldw, int64_t w_plane,
                    float oscale, const float* bias, const float* R1, int32_t ldr1, const float* R2, int32_t ldr2,
                    float* C, int32_t ldc, void* C2, int32_t ldc2, int64_t c_plane, float cscale, int32_t M, int32_t N,
                    int32_t K, int32_t relu, int32_t tile, int32_t iters, float* ms_out, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    Gemm2Args g{};
    g.A = reinterpret_cast<const unsigned short*>(A2); g.lda = lda; g.a_plane = (size_t)a_plane;
    g.W = reinterpret_cast<const unsigned short*>(W2); g.ldw = ldw; g.w_plane = (size_t)w_plane; g.oscale = oscale;
    g.bias = bias; g.R1 = R1; g.ldr1 = ldr1; g.R2 = R2; g.ldr2 = ldr2; g.C = C; g.ldc = ldc;
    g.C2 = reinterpret_cast<unsigned short*>(C2); g.ldc2 = ldc2; g.c_plane = (size_t)c_plane; g.cscale = cscale;
    g.M = M; g.N = N; g.K = K; g.relu = relu; g.tile = tile & 0xfff;
    // measurement hook: operands in the K-blocked layout [K / 32][rows][32]: 0x1000 both (lda = ldw = 32), 0x2000 W only
    // (ldw = 32, lda = K), 0x4000 A only
    if (tile & 0x1000) { g.a_kstep = (long)M * 32; g.w_kstep = (long)N * 32; }
    if (tile & 0x2000) { g.w_kstep = (long)N * 32; g.ldw = 32; g.lda = K; }
    if (tile & 0x4000) { g.a_kstep = (long)M * 32; g.lda = 32; g.ldw = K; }
    // 0x8000: the split-K form (four slices + one reduce launch; fp32 output) with a scratch partial buffer owned by this hook
    static DevBuf splitk_scratch;
    if (tile & 0x8000) {
        if (splitk_scratch.ensure(sizeof(float) * 4 * (size_t)M * N)) return -2;
        g.ksplit = 4; g.part = splitk_scratch.as<float>();
    }
    int rc;
    if (iters <= 0 || !ms_out) return launch_gemm_f16x2(g, s);
    for (int i = 0; i < 3; ++i) if ((rc = launch_gemm_f16x2(g, s))) return rc;
    hipEvent_t a, b;
    PF_HIP_TRY(hipEventCreate(&a));
    PF_HIP_TRY(hipEventCreate(&b));
    PF_HIP_TRY(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) if ((rc = launch_gemm_f16x2(g, s))) return rc;
    PF_HIP_TRY(hipEventRecord(b, s));
    PF_HIP_TRY(hipEventSynchronize(b));
    float ms = 0.f;
    PF_HIP_TRY(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms_out = ms / iters;
    return 0;
}
/* self / cross attention on two-plane fp16 operands (attention_f16x2.hip): Q2 [2][B Tq, H 128] (q * d_k^-0.5 * 2^e_q),
 * K2 [2][>= B Tp + 32, H 128], VT2 [2][H 128, ldvt >= B Tp + 32] (columns = rows with index bits 2 and 3 swapped), O2 out
 * planes [2][B Tq, H 128]. variant: 0 = default schedule, 1 = alternative schedule (measurement hook).
 * iters > 0 with ms_out: additionally times `iters` launches */
int pf_k_attention_f16x2(const void* Q2, int64_t q_plane, const void* K2, int64_t k_plane, const void* VT2, int32_t ldvt,
                         int64_t vt_plane, void* O2, int64_t o_plane, const int32_t* klens_dev, int32_t B, int32_t H,
                         int32_t Tp, int32_t Tq, float sscale, float oscale, int32_t variant, int32_t iters, float* ms_out,
                         void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    Attn2Args aa{};
    const int D = H * 128;
    aa.Q = reinterpret_cast<const unsigned short*>(Q2); aa.ldq = D; aa.q_plane = (size_t)q_plane;
    aa.K = reinterpret_cast<const unsigned short*>(K2); aa.ldk = D; aa.k_plane = (size_t)k_plane;
    aa.VT = reinterpret_cast<const unsigned short*>(VT2); aa.ldvt = ldvt; aa.vt_plane = (size_t)vt_plane;
    aa.O = reinterpret_cast<unsigned short*>(O2); aa.ldo = D; aa.o_plane = (size_t)o_plane;
    aa.klens = klens_dev; aa.B = B; aa.H = H; aa.Tp = Tp; aa.Tq = Tq; aa.sscale = sscale; aa.oscale = oscale;
    aa.variant = variant & 15; aa.xcd_nqb = (variant & 16) ? -1 : 0;          // + 16: plain (not XCD-aware) workgroup order
    int rc;
    if (iters <= 0 || !ms_out) return launch_attention_f16x2(aa, s);
    for (int i = 0; i < 3; ++i) if ((rc = launch_attention_f16x2(aa, s))) return rc;
    hipEvent_t a, b;
    PF_HIP_TRY(hipEventCreate(&a));
    PF_HIP_TRY(hipEventCreate(&b));
    PF_HIP_TRY(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) if ((rc = launch_attention_f16x2(aa, s))) return rc;
    PF_HIP_TRY(hipEventRecord(b, s));
    PF_HIP_TRY(hipEventSynchronize(b));
    float ms = 0.f;
    PF_HIP_TRY(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    *ms_out = ms / iters;
    return 0;
}
/* full-row form (gemm_f16x2_row.hip), N = 512: v = relu?(A W^T oscale + bias) + R1, R2 + v -> C (fp32, optional); with ln_g:
 * LayerNorm(v) -> planes of y * yscale at Y2 (ld 512, planes y_plane apart) or fp32 at Yf (ld 512) */
int pf_k_gemm_f16x2_row(const void* A2, int32_t lda, int64_t a_plane, const void* W2, int32_t ldw, int64_t w_plane, float oscale,
                        const float* bias, const float* R1, int32_t ldr1, const float* R2, int32_t ldr2, float* C, int32_t ldc,
                        const float* ln_g, const float* ln_b, float ln_eps, void* Y2, int64_t y_plane, float yscale, float* Yf,
                        int32_t M, int32_t K, int32_t relu, int32_t a_nt, int32_t iters, float* ms_out, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    GemmRowArgs g{};
    g.A = reinterpret_cast<const unsigned short*>(A2); g.lda = lda; g.a_plane = (size_t)a_plane;
    g.W = reinterpret_cast<const unsigned short*>(W2); g.ldw = ldw; g.w_plane = (size_t)w_plane; g.oscale = oscale;
    g.bias = bias; g.R1 = R1; g.ldr1 = ldr1; g.R2 = R2; g.ldr2 = ldr2; g.C = C; g.ldc = ldc;
    g.ln_g = ln_g; g.ln_b = ln_b; g.ln_eps = ln_eps; g.Y2 = reinterpret_cast<unsigned short*>(Y2); g.ldy2 = 512;
    g.y_plane = (size_t)y_plane; g.yscale = yscale; g.Yf = Yf; g.ldyf = 512; g.M = M; g.N = 512; g.K = K; g.relu = relu;
    g.a_nt = a_nt & 1; g.block_rows = a_nt >> 8;          // bits 8..: GemmRowArgs.block_rows (0 by row count, 96, 128, 129)
    if (iters <= 0 || !ms_out) return launch_gemm_f16x2_row(g, s);
    return time_launches([&] { return launch_gemm_f16x2_row(g, s); }, iters, ms_out, s);
}
/* the encoder block's feed-forward in one launch (gemm_f16x2_ffn.hip): C = R + (relu(X W1^T + b1) W2^T + b2) [+ LayerNorm -> planes Y2
 * or fp32 Yf]; operands are two-plane fp16 tensors ([2][M, 512], [2][F, 512], [2][512, F]), plane strides M 512 / F 512 / 512 F */
int pf_k_ffn_f16x2(const void* X2, const void* W1, const void* W2, const float* b1, const float* b2, float oscale1, float hscale,
                   float oscale2, const float* R, float* Cout, const float* ln_g, const float* ln_b, float ln_eps, void* Y2,
                   float yscale, float* Yf, int32_t M, int32_t F, int32_t iters, float* ms_out, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    FfnArgs g{};
    g.abl = (F >> 24) & 15;                               // measurement-only variants (tools/bench_ffn.py)
    const bool wkb = (F >> 28) & 1;                       // weights in the K-blocked layout [K / 32][rows][32]
    F &= 0xffffff;
    g.X2 = reinterpret_cast<const unsigned short*>(X2); g.ldx = 512; g.x_plane = (size_t)M * 512;
    g.W1 = reinterpret_cast<const unsigned short*>(W1); g.ldw1 = 512; g.w1_plane = (size_t)F * 512;
    g.W2 = reinterpret_cast<const unsigned short*>(W2); g.ldw2 = F; g.w2_plane = (size_t)512 * F;
    g.b1 = b1; g.b2 = b2; g.oscale1 = oscale1; g.hscale = hscale; g.oscale2 = oscale2; g.R = R; g.ldr = 512; g.C = Cout; g.ldc = 512;
    g.ln_g = ln_g; g.ln_b = ln_b; g.ln_eps = ln_eps; g.Y2 = reinterpret_cast<unsigned short*>(Y2); g.ldy2 = 512;
    g.y_plane = (size_t)M * 512; g.yscale = yscale; g.Yf = Yf; g.ldyf = 512; g.M = M; g.D = 512; g.F = F;
    if (wkb) { g.ldw1 = 32; g.w1_kstep = (size_t)F * 32; g.ldw2 = 32; g.w2_kstep = (size_t)512 * 32; }
    if (iters <= 0 || !ms_out) return launch_ffn_f16x2(g, s);
    return time_launches([&] { return launch_ffn_f16x2(g, s); }, iters, ms_out, s);
}
/* the FSMN form of the full-row kernel: the first addend is the FSMN memory block (11 taps, left padding 5) of fs_v [M, 512],
 * valid input rows [fs_lo[g], fs_hi[g]) per 16-row group g; M % 16 == 0, LayerNorm epilogue required */
int pf_k_gemm_f16x2_row_fsmn(const void* A2, int32_t lda, int64_t a_plane, const void* W2, int32_t ldw, int64_t w_plane, float oscale,
                             const float* bias, const float* fs_v, int32_t ldfv, const float* fs_w, const int32_t* fs_lo,
                             const int32_t* fs_hi, const float* R2, int32_t ldr2, float* C, int32_t ldc, const float* ln_g,
                             const float* ln_b, float ln_eps, void* Y2, int64_t y_plane, float yscale, float* Yf, int32_t M, int32_t K,
                             int32_t a_nt, int32_t iters, float* ms_out, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    GemmRowArgs g{};
    g.A = reinterpret_cast<const unsigned short*>(A2); g.lda = lda; g.a_plane = (size_t)a_plane;
    g.W = reinterpret_cast<const unsigned short*>(W2); g.ldw = ldw; g.w_plane = (size_t)w_plane; g.oscale = oscale;
    g.bias = bias; g.fs_v = fs_v; g.ldfv = ldfv; g.fs_w = fs_w; g.fs_lo = fs_lo; g.fs_hi = fs_hi; g.R2 = R2; g.ldr2 = ldr2;
    g.C = C; g.ldc = ldc; g.ln_g = ln_g; g.ln_b = ln_b; g.ln_eps = ln_eps; g.Y2 = reinterpret_cast<unsigned short*>(Y2); g.ldy2 = 512;
    g.y_plane = (size_t)y_plane; g.yscale = yscale; g.Yf = Yf; g.ldyf = 512; g.M = M; g.N = 512; g.K = K;
    g.a_nt = a_nt & 1; g.block_rows = a_nt >> 8;
    if (iters <= 0 || !ms_out) return launch_gemm_f16x2_row(g, s);
    return time_launches([&] { return launch_gemm_f16x2_row(g, s); }, iters, ms_out, s);
}
/* LayerNorm with the two-plane fp16 output the f16x2 GEMMs consume (planes of y * scale, `plane` elements apart, ld ldy) */
int pf_k_layernorm_planes(const float* x, int32_t ldx, const float* gamma, const float* beta, void* y2, int32_t ldy, int64_t plane,
                          float scale, int32_t M, int32_t D, float eps, int32_t iters, float* ms_out, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    auto fn = [&] { return launch_layernorm(x, ldx, gamma, beta, reinterpret_cast<float*>(y2), ldy, M, D, ldy, eps, s, 3, 0, (size_t)plane, scale); };
    if (iters <= 0 || !ms_out) return fn();
    return time_launches(fn, iters, ms_out, s);
}
/* the QKV form (kv_form = 0: N = 3 D -> Q planes, K planes, fp32 V, V^T planes) and the KV form (kv_form = 1: N = 2 D -> K
 * planes, V^T planes) of gemm_f16x2.hip; Qp / Kp planes are qk_plane apart (ld D), VT [2][D, ldvt] vt_plane apart */
int pf_k_gemm_f16x2_qkv(const void* A2, int32_t lda, int64_t a_plane, const void* W2, int32_t ldw, int64_t w_plane, float oscale,
                        const float* bias, int32_t M, int32_t D, int32_t K, int32_t kv_form, void* Qp, void* Kp, int64_t qk_plane,
                        float* Vf, void* VT, int32_t ldvt, int64_t vt_plane, float q_mul, float k_mul, float v_mul, int32_t tile,
                        int32_t iters, float* ms_out, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    Gemm2Args g{};
    g.A = reinterpret_cast<const unsigned short*>(A2); g.lda = lda; g.a_plane = (size_t)a_plane;
    g.W = reinterpret_cast<const unsigned short*>(W2); g.ldw = ldw; g.w_plane = (size_t)w_plane; g.oscale = oscale; g.bias = bias;
    g.C = Vf; g.ldc = D; g.M = M; g.N = (kv_form ? 2 : 3) * D; g.K = K; g.qkv_D = D; g.kv_form = kv_form;
    g.Qp = reinterpret_cast<unsigned short*>(Qp); g.Kp = reinterpret_cast<unsigned short*>(Kp); g.qk_plane = (size_t)qk_plane;
    g.VT = reinterpret_cast<unsigned short*>(VT); g.ldvt = ldvt; g.vt_plane = (size_t)vt_plane;
    g.q_mul = q_mul; g.k_mul = k_mul; g.v_mul = v_mul; g.tile = tile;
    if (iters <= 0 || !ms_out) return launch_gemm_f16x2(g, s);
    return time_launches([&] { return launch_gemm_f16x2(g, s); }, iters, ms_out, s);
}
/* the fused arg-max form (vocabulary / CTC projections): ids[row] = argmax_n (A W^T oscale + bias)[row, n], lowest index on ties;
 * sval / sidx: scratch [M, 2 ceil(N / 256)] */
int pf_k_gemm_f16x2_argmax(const void* A2, int32_t lda, int64_t a_plane, const void* W2, int32_t ldw, int64_t w_plane, float oscale,
                           const float* bias, int32_t M, int32_t N, int32_t K, int32_t* ids, float* sval, int32_t* sidx,
                           void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(ids && sval && sidx, "gemm_f16x2_argmax: scratch required");
    Gemm2Args g{};
    g.A = reinterpret_cast<const unsigned short*>(A2); g.lda = lda; g.a_plane = (size_t)a_plane;
    g.W = reinterpret_cast<const unsigned short*>(W2); g.ldw = ldw; g.w_plane = (size_t)w_plane; g.oscale = oscale; g.bias = bias;
    g.M = M; g.N = N; g.K = K; g.amax_val = sval; g.amax_idx = sidx; g.amax_ld = gemm_f16x2_argmax_parts(M, N);
    int rc;
    if ((rc = launch_gemm_f16x2(g, s))) return rc;
    return launch_argmax_reduce(sval, sidx, g.amax_ld, g.amax_ld, ids, nullptr, M, s);
}
/* fp32 -> bf16 (round to nearest even), n elements */
int pf_k_cast_bf16(const float* x, void* y, int64_t n, void* stream) {
    return launch_cast_bf16(x, reinterpret_cast<unsigned short*>(y), (size_t)n, reinterpret_cast<hipStream_t>(stream));
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
/* y[row] = x[row] - logsumexp(x[row]) over N columns, fp32 (may run in place) */
int pf_k_log_softmax(const float* x, int32_t ldx, float* y, int32_t ldy, int32_t M, int32_t N, void* stream) {
    return launch_log_softmax(x, ldx, y, ldy, M, N, reinterpret_cast<hipStream_t>(stream));
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
/* fp32 Q/K/V -> fp32 O with both products on the bf16 MFMA from three-plane split operands (attention_split3.hip) */
int pf_k_attention_split3(const float* Q, int32_t ldq, const float* K, int32_t ldk, const float* V, int32_t ldv, float* O,
                          int32_t ldo, const int32_t* klens_dev, int32_t B, int32_t H, int32_t Tq, int32_t Tk, float scale,
                          void* stream) {
    AttnArgs aa{};
    aa.Q = Q; aa.ldq = ldq; aa.K = K; aa.ldk = ldk; aa.V = V; aa.ldv = ldv; aa.O = O; aa.ldo = ldo; aa.klens = klens_dev;
    aa.B = B; aa.H = H; aa.Tq = Tq; aa.Tk = Tk; aa.scale = scale;
    return launch_attention_split3(aa, reinterpret_cast<hipStream_t>(stream));
}
/* small heads (d_k <= 64): Q/K/V rows hold H heads of d_k columns (attention_small.hip) */
int pf_k_attention_small(const float* Q, int32_t ldq, const float* K, int32_t ldk, const float* V, int32_t ldv, float* O,
                         int32_t ldo, const int32_t* klens_dev, int32_t B, int32_t H, int32_t d_k, int32_t Tq, int32_t Tk,
                         float scale, void* stream) {
    AttnArgs aa{};
    aa.Q = Q; aa.ldq = ldq; aa.K = K; aa.ldk = ldk; aa.V = V; aa.ldv = ldv; aa.O = O; aa.ldo = ldo; aa.klens = klens_dev;
    aa.B = B; aa.H = H; aa.Tq = Tq; aa.Tk = Tk; aa.scale = scale;
    return launch_attention_small(aa, d_k, reinterpret_cast<hipStream_t>(stream));
}
/* out[i, :] = table[ids[i], :] (embedding lookup), ids int32 on the device, clamped to [0, rows) */
int pf_k_gather_rows(const float* table, int32_t ld, int32_t rows, const int32_t* ids_dev, float* out, int32_t n, int32_t D,
                     void* stream) {
    PF_REQUIRE(table && ids_dev && out, "gather_rows: null");
    return launch_gather_rows(table, ld, rows, ids_dev, out, n, D, reinterpret_cast<hipStream_t>(stream));
}
/* torch.nn.LSTM (one layer, ndir directions, zero initial state) on caller-provided device tensors in torch's layouts:
 * x [B, T, D], w_ih [ndir][4H][D], w_hh [ndir][4H][H], b_ih / b_hh [ndir][4H] -> out [B, T, ndir * H]. Test hook of
 * lstm.hip: the weight re-layout happens on the host here, so the call synchronises. */
int pf_k_lstm(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, int32_t B, int32_t T,
              int32_t D, int32_t H, int32_t ndir, float* out, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(x && w_ih && w_hh && b_ih && b_hh && out && B > 0 && T > 0 && D > 0 && H > 0 && (ndir == 1 || ndir == 2),
               "k_lstm: null/empty argument");
    if (check_device()) return -2;
    DevBuf x_tm, pre, h_a, h_b, cell;
    if (x_tm.ensure(sizeof(float) * (size_t)B * T * D)) return -2;
    int rc;
    if ((rc = launch_rows_bt_to_tb(x, x_tm.as<float>(), B, T, D, s))) return rc;
    LstmW w{};
    w.w_ih[0] = w_ih; w.w_ih[1] = w_ih + (size_t)4 * H * D; w.w_hh = w_hh; w.b_ih = b_ih; w.b_hh = b_hh;
    if ((rc = lstm_forward(w, x_tm.as<float>(), T, B, D, H, ndir, out, 0, pre, h_a, h_b, cell, s))) return rc;
    PF_HIP_TRY(hipStreamSynchronize(s));
    return 0;
}
int pf_k_attention_bf16(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, void* O,
                        int32_t ldo, const int32_t* klens_dev, int32_t B, int32_t H, int32_t Tq, int32_t Tk, float scale,
                        void* stream) {
    AttnArgs aa{};
    aa.Q = reinterpret_cast<const float*>(Q); aa.ldq = ldq; aa.K = reinterpret_cast<const float*>(K); aa.ldk = ldk;
    aa.V = reinterpret_cast<const float*>(V); aa.ldv = ldv; aa.O = reinterpret_cast<float*>(O); aa.ldo = ldo;
    aa.klens = klens_dev; aa.B = B; aa.H = H; aa.Tq = Tq; aa.Tk = Tk; aa.scale = scale;
    return launch_attention_bf16(aa, reinterpret_cast<hipStream_t>(stream));
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
    if (upload_h2d(ln.p, lens.data(), sizeof(int) * B, s)) return -2;
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

// ---- engine_tables.h, second half: needs every handle type
namespace pf {
TensorTable* table_of(int kind, void* h) {
    switch (kind) {
        case HANDLE_ENCODER: return &reinterpret_cast<Encoder*>(h)->tt;
        case HANDLE_PREDICTOR: return &reinterpret_cast<Predictor*>(h)->tt;
        case HANDLE_DECODER: return &reinterpret_cast<Decoder*>(h)->tt;
        case HANDLE_CTC: return &reinterpret_cast<Ctc*>(h)->tt;
        case HANDLE_VAD: return &reinterpret_cast<Vad*>(h)->tt;
        default: return nullptr;
    }
}
int handle_weights_replaced(int kind, void* handle) {
    TensorTable* tt = handle ? table_of(kind, handle) : nullptr;
    if (!tt) { set_error("dp: null handle or unknown handle kind"); return -1; }
    for (auto& kv : tt->t) kv.second.set = true;
    ++tt->version;
    tt->drop_bf16();
    if (kind == HANDLE_ENCODER) reinterpret_cast<Encoder*>(handle)->resolved = false;
    if (kind == HANDLE_DECODER) reinterpret_cast<Decoder*>(handle)->resolved = false;
    if (kind == HANDLE_PREDICTOR) reinterpret_cast<Predictor*>(handle)->packed = false;
    return 0;
}
}  // namespace pf
