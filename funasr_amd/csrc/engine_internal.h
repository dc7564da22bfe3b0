// Internal header of the C-ABI layer (include/paraformer_hip.h): the handle types of every module family and the helpers they
// share. The layer is split by family -- engine.hip (shared state: errors, profiling, staging ring, instrumented launch helpers,
// ABI version), engine_frontend.hip, engine_encoder.hip, engine_decoder.hip (predictor, decoder, CTC), engine_stream.hip,
// engine_vad.hip, engine_kernels.hip (the pf_k_* single-kernel hooks) -- so that a kernel change recompiles one family
// (round-3 review: engine.hip was one 3 400-line translation unit).
//
// Memory model: weights are copied once into library-owned HBM (repacked where a kernel wants a different
// layout); activations live in a per-handle workspace that only grows (hipMalloc outside the steady state,
// never torch's caching allocator -- AutoModel calls torch.cuda.empty_cache() after every batch,
// funasr/auto/auto_model.py:846-849). With 288 GB per MI355X nothing is ever recomputed or spilled.
#pragma once
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

void set_error(const std::string& msg);
const char* get_error();

// ------------------------------------------------------------------------------------------------ profiling
// Optional hipEvent instrumentation of the dominant kernels (bench.py roofline line).
struct ProfRec { hipEvent_t a, b; int kind; double work; const char* tag; };
extern bool g_prof_on;
extern std::vector<ProfRec> g_prof;
extern std::vector<hipEvent_t> g_ev_pool;
hipEvent_t prof_event();
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
extern unsigned long long g_ws_epoch;


struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        const size_t had = cap;
        if (p) { (void)hipDeviceSynchronize(); (void)hipFree(p); p = nullptr; cap = 0; }
        // Growing costs a device-wide synchronisation (the old block may be in use by queued kernels), i.e. it stalls a loop that
        // keeps batches in flight -- so grow rarely: at least 64 KB (the per-clip arrays of a batch: a ragged corpus taken longest
        // first raises the clip count batch after batch) and geometrically (x 1.5) beyond what was there; 288 GB of HBM pay for it
        size_t want = bytes + bytes / 8;
        if (want < had + had / 2) want = had + had / 2;
        if (want < 65536) want = 65536;
        PF_HIP_TRY(hipMalloc(&p, want));
        cap = want;
        ++g_ws_epoch;
        return 0;
    }
    template <class T> T* as() { return reinterpret_cast<T*>(p); }
};

// the fused feed-forward runs ONE 128-row workgroup per CU: take it where the workgroups fill whole rounds of the CUs to >= 85 %
static inline bool ffn_fills_rounds(int M) {
    const int n_cu = device_cu_count();
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

int check_device();
static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// Host -> device uploads of per-call metadata through the library's pinned staging ring (engine.hip)
int upload_h2d(void* dst, const void* src, size_t bytes, hipStream_t s);
int upload_lens(DevBuf& buf, const int32_t* host, int B, hipStream_t s);


// Instrumented launch helpers ---------------------------------------------------------------------------
// Kernel choice is by CALLER, never by batch size, so that a clip's (or a stream's) result does not depend on what
// else is in the batch: the offline path always takes the 128 x 128 tile kernel, the streaming step always the small-M
// weight-streaming kernel (whose K slicing depends on K only). g_skinny_max_m is a test hook for pf_k_gemm_f32.
extern int g_skinny_max_m;
extern thread_local bool g_stream_mode;
// workspace of the calling stream handle for the four-workgroup form of the long-K small-M GEMMs (GemmArgs.ws_part / ws_count)
extern thread_local float* g_ws_part;
extern thread_local int* g_ws_count;
constexpr int WS_TILES = 64;                                  // tiles the workspace covers (N = 512: 32 column tiles x <= 2 row tiles)
constexpr size_t WS_PART_FLOATS = (size_t)WS_TILES * 16 * 512;
struct StreamModeScope {
    bool prev;
    float* prev_part; int* prev_count;
    explicit StreamModeScope(float* part = nullptr, int* count = nullptr) : prev(g_stream_mode), prev_part(g_ws_part), prev_count(g_ws_count) {
        g_stream_mode = true; g_ws_part = part; g_ws_count = count;
    }
    ~StreamModeScope() { g_stream_mode = prev; g_ws_part = prev_part; g_ws_count = prev_count; }
};

// ================================================================================================ frontend
struct Frontend {
    pf_frontend_config cfg;
    DevBuf window, twiddle, piece_w, piece_k0, mel_first, mel_count, cmvn_shift, cmvn_scale;
    int n_pieces = 0;
    DevBuf fbank, nfr;
    bool has_cmvn = false;
    int feat_dim() const { return cfg.n_mels * cfg.lfr_m; }
    float dither = 0.f; unsigned long long dither_seed = 0; unsigned dither_calls = 0;   // pf_frontend_set_dither
    int verify = 0; DevBuf faults;                                                       // pf_frontend_set_verify (FbankArgs.verify)
    int snip_edges = 1; DevBuf nsamp;                                                    // pf_frontend_set_snip_edges (FbankArgs.n_samples)
};
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
    // w_2: 1 = full-row form (residual + the next norm1 in the epilogue), 0 = a 256 x 256 tile GEMM (+ residual) and the next norm1 as
    // its own launch, 2 (default) = the tile form where its 256-row blocks fill whole rounds of the CUs to >= 65 % (the 128 x 512
    // row block re-streams the 4-MB W panel for every 128 rows: 1.28 GB of L2 -> LDS per launch at M = 32768 against 1.0 GB; same-call
    // A/B 55.0 -> 53.8 ms per step, profiles/r05k_ab_w2_row.txt; the mark was 85 % until SenseVoice's M = 22 528 -- 11/16 of a round --
    // measured 47.0 -> 45.8 ms per step in the tile form, profiles/r05_sensevoice_options.json: at the socket's power limit idle CUs
    // cost little). Every choice gives the same bits (tested).
    int row_sched = 0;                  // k-step order of the eight-wave full-row kernel: 0 plain (default since round 5: FSMN form 108 -> 103 us, profiles/r05t), 2 both k-steps' fragments up front
    int w2_row = 2;
    // Gemm2Args.tile of w_2 in its tile form: 7 (default) = the four-wave shape (gemm_f16x2_w4.hip) for this projection only -- its fp32 +
    // residual epilogue is where that shape wins (192.8 -> 180.5 us per launch, 53.6 -> 53.1 ms per step, profiles/r05q_ab_w2_tile.txt;
    // for the plane / QKV forms it ties and its denser matrix code lowers the sustained clock); 0: gemm_tile
    int w2_tile = 7;
    DevBuf fs_grp;                      // int32 [2][M / 16]: valid v rows [lo, hi) of the sequence owning each 16-row group
    std::vector<int32_t> h_fs;
    const int* cur_fs = nullptr; int cur_fs_groups = 0;
    int attn_variant = 3;               // attention_f16x2.hip schedule (3: lazy rescale)
    int row_nt = 1;                     // non-temporal A loads in the full-row GEMMs: 0 none, 1 linear_out (K = 512), 2 linear_out and w_2
    // work of the profiling scopes (bench.py roofline) in ALGORITHMIC rows: sum of the batch's valid frames and of their squares
    // (SURVEY 8d counts per valid frame; the padded layout computes Tp = ceil16(T) rows per sequence). 0: use the computed rows.
    double prof_rows = 0, prof_sq = 0;
    int gemm_tile = 0;                  // Gemm2Args.tile of the block's GEMMs (0: by shape; 5: 128 x 256, two workgroups per CU)
};


// exponent e with bound * 2^e <= 2^15 (a factor 2 under fp16's 65504 for the roundings on the way)
static inline int exp_for_bound(float bound) {
    if (!(bound > 0.f)) return 15;
    int e = (int)floorf(log2f(32768.f / bound));
    return e > 15 ? 15 : e;
}
static inline float pow2f(int e) { return ldexpf(1.f, e); }

// per-layer streaming context: attention additionally sees the cached K/V ring of this layer and the first
// `append_rows` K/V rows of the window are appended to it afterwards (attention.py:343-361)
// the form of the block's w_2 projection (Encoder.w2_row): true = full-row kernel, false = tile GEMM + separate LayerNorm launch
static inline bool encoder_w2_row_form(const Encoder* e, int M) {
    if (e->w2_row != 2) return e->w2_row == 1;
    const int n_cu = device_cu_count();
    const int blocks = ceil_div(M, 256) * (e->cfg.d_model / 256), rounds = ceil_div(blocks, n_cu);
    return !(e->cfg.d_model % 256 == 0 && blocks >= (int)(0.65 * rounds * n_cu));
}

// engine_frontend.hip: from now on every frontend handle of the process cross-checks its fbank frames (another stream may share a CU)
void note_concurrent_streams();

struct EncChunkCtx {
    float* ring; int cap; const StreamDev* st; int append_rows;
    const int* lens;     // device [B]: every window row is valid in a chunk
    int mod = 0;         // > 0: the ring's trim size when its first, untrimmed append needs more rows than that (AttnArgs.app_mod)
    bool x2 = false;     // the block's four GEMMs on the fp16 matrix cores (two-plane operands, gemm_f16x2.hip), fp32 results
    // fp32 step: the LayerNorms ride in the small-M GEMMs on either side (gemm_skinny.hip: the producer's epilogue leaves per-row
    // partial sums, the consumer normalises on the fetch). ln_stats = [rows][d_model / 16][2] scratch; ln_in_ready = the block that
    // ran before left the statistics of this block's input there
    float* ln_stats = nullptr;
    bool ln_in_ready = false;
    const float* ln_c_qkv = nullptr;   // the LayerNorm form's constants of this block's QKV projection: c1 [3 d_model], then c2
    const float* ln_c_w1 = nullptr;    // of w_1: c1 [ffn_dim], then c2
    // f16x2 step: norm1 of the block after this one rides in the second launch of this block's split-K w_2 (launch_splitk_reduce_ln)
    const EncLayerW* x2_out_next = nullptr;
    bool x2_in_ready = false;
    int x2_short_k = 0;          // f16x2 step of few rows: K = d_model projections in the four-slice split-K form too (1: linear_out only, 2: all)
    bool x2_fold = false;        // (with x2_short_k) norm2 from linear_out's second launch
    bool x2_attn_planes = false; // f16x2 step: the attention writes the out-projection's operand planes (AttnArgs.O2)
    bool fsmn_rides = false;     // the FSMN memory block is computed by extra workgroups of the attention launch (AttnArgs.fs_*)
    const EncLayerW* next = nullptr;
};

// =============================================================================================== predictor
struct Predictor {
    pf_predictor_config cfg;
    TensorTable tt;
    DevBuf col, conv, zero_row;
    // the scan state of one batch (what pf_predictor_embeds reads back). Two slots: the split-phase pipeline (engine_pipeline.hip)
    // runs batch i + 1's encoder + scan before batch i's decoder, so batch i's state must survive that; the module-level entry
    // points (pf_predictor_alphas / _embeds / _timestamp) use slot 0
    // (curs / ntok: V3's sequential scan -- per slot too since round 6: pf_predictor_alphas_begin keeps two batches in flight)
    struct CifState { DevBuf lens, alphas, peaks, rems, flags, nfires, curs, ntok; int B = 0, T = 0; } st[2];
    // CifPredictorV3 (bicif_paraformer/cif_predictor.py:121-384): sequential fp32 CIF + the upsampled timestamp head
    bool v3 = false;
    pf_predictor_v3_config c3{};
    DevBuf up, x_tm, pre, lstm_out, h_a, h_b, cell, tok_dev, ulens, pack, ts_lens;
    bool packed = false;                 // pack = both directions' re-laid weight_hh, then bias_ih, bias_hh back to back
    std::vector<int32_t> ul_host;
};

// the predictor's two phases without the host synchronisation between them (engine_decoder.hip; engine_pipeline.hip drives them)
int predictor_alphas_enqueue(Predictor* p, int slot, const float* hidden, const int32_t* lens_host, int B, int T, hipStream_t s);
const int32_t* predictor_counts_dev(Predictor* p, int slot);
const float* predictor_alphas_dev(Predictor* p, int slot);
const float* predictor_peaks_dev(Predictor* p, int slot);
int predictor_embeds_slot(Predictor* p, int slot, const float* hidden, int B, int T, int N, float* embeds, hipStream_t s);

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


struct Decoder {
    pf_decoder_config cfg;
    bool contextual = false;
    DevBuf xself, xcat, ctx_lens;     // contextual: x after the FSMN residual, [x_src_attn | cx] rows, hotword counts
    TensorTable tt;
    std::vector<DecLayerW> layers;
    int n_blocks2 = 0;       // decoders2: num_blocks - att_layer_num blocks of FFN + FSMN without cross-attention (decoder.py:363-380)
    std::vector<DecLayerW> layers2;
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


struct Ctc {
    int d_model, vocab;
    TensorTable tt;
    DevBuf pval, pidx;
    int precision = 0;       // 0 fp32 MFMA, 3 f16x2 arg-max route
    DevBuf h2, dsc;          // f16x2: planes of the hidden states, [amax, 2^e, 2^-e]
};

// ================================================================================================ streaming
// A lock-step batch of S independent streams (the reference handles exactly one: "batch_size must be set 1",
// paraformer_streaming/model.py:705). All per-stream state lives in HBM; the steady-state step is captured in a
// hipGraph keyed by (n_frames, is_final, tail_chunk) and replayed.
struct Stream {
    Encoder* e = nullptr; Predictor* p = nullptr; Decoder* d = nullptr;
    pf_stream_config cfg{};
    int S = 1, keep = 5, Wmax = 0, Nmax = 0, enc_cap = 0, dec_cap = 0, pe_rows = 0;
    int enc_mod = 0;     // > 0: enc_cap rows hold an untrimmed first chunk (chunk_left + max_frames > look_back * chunk_cur); trim size
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
    // fp32 step: LayerNorms carried by the small-M GEMMs: 0 never, 1 steps of <= 32 rows, 2 always (the default: a stream's bits
    // must not depend on how many streams run beside it; measured S = 1 / 2 / 8 / 32: -6 / -4 / -2 / +5 % step time)
    int ln_carry = 2;
    bool fsmn_rides = true;                                  // encoder FSMN inside the attention launch (AttnArgs.fs_*)
    DevBuf ln_stats;
    bool pending = false; int pending_rows = 0;              // a step enqueued by pf_stream_step_begin and not yet collected
    bool wide_k = false;                                     // long-K N = 512 projections of a <= 32-row step over four workgroups per tile
    DevBuf ws_part, ws_count;                                // their slice tiles and tile counters (GemmArgs.ws_part / ws_count)
    // constants c1 = W gamma, c2 = W beta + bias of every LayerNorm -> GEMM pair of the fp32 step (launch_ln_consts), prepared from
    // the handles' weights (ln_ver_*: their TensorTable versions then). Encoder block l: [qkv c1, c2 | w_1 c1, c2]; decoder layer l
    // (and decoders3 as layer n_blocks): [w_1 c1, c2 | w_2 c1, c2 | linear_q c1, c2]; then the vocabulary projection's c1, c2.
    DevBuf ln_consts;
    uint64_t ln_ver_e = ~0ull, ln_ver_d = ~0ull;
    DevBuf dec_ln_a, dec_ln_b, dec_ln_f;                     // decoder: block partials of the token rows (d_model wide twice, ffn wide)
    // f16x2 step, handles of <= 2048 rows (streams x largest window: most CUs idle): K = d_model projections in the four-slice split-K
    // form. 3 (default) = linear_out only -- its second launch computes norm2 and so REPLACES a launch: S = 64 7.91 -> 7.67 ms;
    // 1 = every K = d_model projection (measured break-even: the GEMMs drop from 19 to 15.5 us -- a 128 x 128 block costs ~12 us before
    // its first K stage counts -- and the 100 extra reduce launches of 7.5 us give it back), 2 = that for any handle size, 0 = never.
    // By the handle, never by a step's data.
    int short_k = 3;
    bool ln_folded = true;                                   // f16x2 step: LayerNorms folded into the split-K reductions, attention writes planes
    // f16x2 step: the decoder layers' key/value weights concatenated to ONE [n_blocks * 2 d_model, d_model] matrix (planes with one
    // exponent; prepared with the other planes): one GEMM instead of n_blocks, one ring append instead of n_blocks
    DevBuf kvcat_w, kvcat_b, kvcat_2;
    int kvcat_e = 0;
    bool kv_batched = true;                                  // fp32 step: the decoder's key/value projections of the encoder rows as one launch
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

struct Vad {
    pf_vad_config cfg;
    TensorTable tt;
    DevBuf a, b, c, cache_tmp, ids;
    size_t zeroed_for = 0;
};
static inline int vad_pad(int k) { return round_up(k, 32); }
struct LstmW { const float* w_ih[2]; const float* w_hh; const float* b_ih; const float* b_hh; };
struct DecCtxArgs { const float* info; int n_hot; float clas_scale; };    // hotword embeddings [B, n_hot, D] (contextual decoder)

// ---- defined in one family's file, used by others
int check_device();
int upload_h2d(void* dst, const void* src, size_t bytes, hipStream_t s);
int upload_lens(DevBuf& buf, const int32_t* host, int B, hipStream_t s);
int gemm(const GemmArgs& a, hipStream_t s);
int gemm_simple(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc,
                       int M, int N, int K, int relu, const float* R1, int ldr1, const float* R2, int ldr2,
                       hipStream_t s);
int layernorm(const float* x, int ldx, const float* g, const float* b, float* y, int ldy, int M, int D,
                     int Dpad, float eps, hipStream_t s);
int fsmn(const FsmnArgs& a, hipStream_t s);
int attention(const AttnArgs& a, double flops, hipStream_t s, bool x3 = false, int dk = 128, bool* appended = nullptr);
int frontend_upload_tables(Frontend* f, const std::vector<float>& window, const std::vector<float>& mel);
int frontend_default_tables(Frontend* f);
int encoder_resolve(Encoder* e);
int encoder_prepare_x2(Encoder* e, hipStream_t s);
int encoder_default_pe(Encoder* e, int T, hipStream_t s);
int encoder_block(Encoder* e, const EncLayerW& w, float* x_in, int ld_in, float* x, int B, int T,
                         hipStream_t s, const EncChunkCtx* cc = nullptr, bool xn_ready = false, const EncLayerW* next = nullptr);
std::string dec_layer_prefix(bool contextual, int n_blocks, int i);
int decoder_resolve(Decoder* d);
int vocab_project(const float* hidden, int M, int D, const float* W, const float* bias, int V, float* logits,
                         int32_t* ids, DevBuf& pval, DevBuf& pidx, hipStream_t s);
int gemm3_simple(const unsigned short* A3, int lda, int M, const unsigned short* W3, const float* bias, float* C,
                        int ldc, int N, int K, int relu, hipStream_t s);
int gemm2_simple(const unsigned short* A2, int lda, int M, int ea, const unsigned short* W2, int ew, const float* bias,
                        float* C, int ldc, int N, int K, int relu, const float* R2, int ldr2, hipStream_t s,
                        const float* oscale_dev = nullptr, float* splitk_part = nullptr);
int dec_layer_x2(Decoder* d, DecLayerW& w, const std::string& p, bool attn, hipStream_t s);
struct FoldedLn { const float* g; const float* b; float* y; int out; float oscale; };   // out: 0 fp32 rows, 3 two fp16 planes of y * oscale
int dec_ffn_x2(Decoder* d, const DecLayerW& w, const float* x, float* out, int M, hipStream_t s, float* splitk_part = nullptr,
               const FoldedLn* ln = nullptr, bool fold_fn = false);
int dec_ffn(Decoder* d, const DecLayerW& w, const float* x, float* out, int M, hipStream_t s,
                   const unsigned short* w1_3 = nullptr);
int decoder_forward_bf16(Decoder* d, const float* memory, int B, int T, int N, int32_t* ids, float* hidden_out,
                                hipStream_t s);
int lstm_forward(const LstmW& w, const float* x_tm, int T, int B, int D, int H, int ndir, float* out, int out_layout,
                        DevBuf& pre, DevBuf& h_a, DevBuf& h_b, DevBuf& cell, hipStream_t s);

}  // namespace pf

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
