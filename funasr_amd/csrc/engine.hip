// C-ABI layer (include/paraformer_hip.h), shared state: error string, profiling hooks, the pinned staging ring for per-call
// metadata, the instrumented launch helpers every module family uses, ABI version. The handle types live in engine_internal.h;
// the families in engine_{frontend,encoder,decoder,stream,vad,kernels}.hip.
#include "engine_internal.h"

namespace pf {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* get_error() { return g_err.c_str(); }

bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_ev_pool;
hipEvent_t prof_event() {
    if (!g_ev_pool.empty()) { hipEvent_t e = g_ev_pool.back(); g_ev_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
unsigned long long g_ws_epoch = 0;

int check_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible: the gfx950 kernels are the only implementation (no CPU fallback)");
        return -2;
    }
    return 0;
}


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
int upload_h2d(void* dst, const void* src, size_t bytes, hipStream_t s) { return stage_ring_for_current_device()->upload(dst, src, bytes, s); }

int upload_lens(DevBuf& buf, const int32_t* host, int B, hipStream_t s) {
    if (buf.ensure(sizeof(int32_t) * (size_t)B)) return -2;
    return upload_h2d(buf.p, host, sizeof(int32_t) * (size_t)B, s);
}

int g_skinny_max_m = 0;
thread_local bool g_stream_mode = false;
thread_local float* g_ws_part = nullptr;
thread_local int* g_ws_count = nullptr;
int gemm(const GemmArgs& a, hipStream_t s) {
    if ((g_stream_mode || a.M <= g_skinny_max_m) && gemm_skinny_applicable(a)) {
        // a long K behind few column tiles (w_2 of a step: 4 MB of weights through 32 workgroups): four workgroups per tile,
        // bit for bit the one-workgroup result (gemm_skinny.hip), so the choice may look at the row count
        if (g_ws_part && a.M <= 32 && a.K >= 1024 && ceil_div(a.N, 16) * ceil_div(a.M, a.M <= 16 ? 16 : 32) <= WS_TILES) {
            GemmArgs w = a;
            w.ws_part = g_ws_part; w.ws_count = g_ws_count;
            return launch_gemm_skinny(w, s);
        }
        return launch_gemm_skinny(a, s);
    }
    if (a.ln_stats_in || a.ln_stats_out) { set_error("gemm: the LayerNorm-carrying form exists in the small-M kernel only"); return -1; }
    ProfScope ps(PROF_GEMM, 2.0 * a.M * (double)a.N * a.K, s);
    return launch_gemm_f32(a, s);
}
int gemm_simple(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc,
                       int M, int N, int K, int relu, const float* R1, int ldr1, const float* R2, int ldr2,
                       hipStream_t s) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.R1 = R1; g.ldr1 = ldr1; g.R2 = R2; g.ldr2 = ldr2;
    g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.relu = relu;
    return gemm(g, s);
}
int layernorm(const float* x, int ldx, const float* g, const float* b, float* y, int ldy, int M, int D,
                     int Dpad, float eps, hipStream_t s) {
    ProfScope ps(PROF_LN, 8.0 * M * (double)D, s);   // bytes: read + write
    return launch_layernorm(x, ldx, g, b, y, ldy, M, D, Dpad, eps, s);
}
int fsmn(const FsmnArgs& a, hipStream_t s) {
    ProfScope ps(PROF_FSMN, (a.R ? 12.0 : 8.0) * a.B * (double)a.T * a.C, s);
    return launch_fsmn(a, s);
}
// `appended` (optional): the caller filled the app_* fields; on return it says whether the kernel did the ring append
// itself (few-query kernel, one workgroup per (stream, head)) -- otherwise the caller launches ring_append_kernel
int attention(const AttnArgs& a, double flops, hipStream_t s, bool x3, int dk, bool* appended) {
    ProfScope ps(PROF_ATTN, flops, s);
    if (appended) *appended = false;
    AttnArgs f = a;
    f.few_q = 0;
    if ((dk != 128 || x3) && (f.fs_in || f.O2)) { set_error("attention: the FSMN rider exists in the few-query fp32 kernel only"); return -1; }
    if (dk != 128) { f.app_rows = 0; return launch_attention_small(f, dk, s); }      // CT-Transformer sized heads
    if (x3) { f.app_rows = 0; return launch_attention_split3(f, s); }
    f.few_q = (g_stream_mode || g_skinny_max_m > 0) ? 1 : 0;   // by caller (streaming step; g_skinny_max_m: test hook)
    if (!appended || !attention_fuses_append(f)) f.app_rows = 0;
    else *appended = true;
    return launch_attention_f32(f, s);
}

}  // namespace pf

using namespace pf;

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

