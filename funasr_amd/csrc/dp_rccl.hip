// Utterance-level data parallelism over the GPUs of one node, at the C ABI (include/paraformer_hip.h, pf_dp_*): one process
// per GPU, RCCL over xGMI. The path shards over independent clips with no data-path collective (the reference forks
// one process per GPU over a split wav.scp and concatenates the outputs, examples/aishell/paraformer/run.sh:135-190), so the
// boundary needs exactly two exchanges:
//   * start-up: rank `root`'s weights -> every rank, written STRAIGHT into the module handles' library-owned HBM (one grouped
//     RCCL broadcast per handle over the handle's tensor table; no packed host arena, no nn.Parameter copy, no re-upload --
//     round 3 shipped three copies of the 880 MB per rank);
//   * per batch: a gather of fixed-stride int32 hypotheses on `root` (KBs: latency-bound).
// RCCL is loaded lazily (dlopen) so that the library itself has no link-time dependency on it: single-GPU users, the CPU-side
// ABI tests and hosts without RCCL load libparaformer_hip.so unchanged. A process that already holds an RCCL (PyTorch's) gets
// that copy through the shared SONAME.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>

#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/paraformer_hip.h"
#include "engine_tables.h"

namespace {

// the slice of rccl.h this file needs (names and layouts as in /opt/rocm/include/rccl/rccl.h: ncclUniqueId is 128 opaque bytes,
// ncclInt32 = 2, ncclFloat32 = 7, ncclSuccess = 0)
typedef struct { char internal[128]; } UniqueId;
typedef void* Comm;
struct Rccl {
    void* so = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Gather)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // PF_RCCL_LIB (environment) names the one library to try instead of the default candidates -- a site with its own RCCL
        // build, and the CPU test of the no-RCCL path (tests/test_abi.py)
        const char* forced = getenv("PF_RCCL_LIB");
        std::string why;
        auto attempt = [&](const char* name) {
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!r.so) {
                const char* e = dlerror();                      // ONE call: dlerror() clears the state it returns
                why += std::string(why.empty() ? "" : "; ") + (e ? e : "dlopen failed");
            }
            return r.so != nullptr;
        };
        if (forced && *forced) attempt(forced);
        else
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if (attempt(name)) break;
        if (!r.so) { r.err = "RCCL not found: " + why; return; }
        auto sym = [&](const char* n) { void* p = dlsym(r.so, n); if (!p && r.err.empty()) r.err = std::string("RCCL lacks ") + n; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(sym("ncclBroadcast"));
        r.Gather = reinterpret_cast<decltype(r.Gather)>(sym("ncclGather"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return &r;
}

struct Dp { Comm comm = nullptr; int world = 0, rank = 0, device = 0; };

int fail(const std::string& what) { pf::set_error(what); return -2; }
int nccl_ok(int rc, const char* what) {
    if (rc == 0) return 0;
    Rccl* r = rccl();
    return fail(std::string("dp: ") + what + ": " + (r->GetErrorString ? r->GetErrorString(rc) : "RCCL error"));
}

int broadcast_handle(pf_dp* dh, int kind, void* handle, int root, void* stream) {
    Dp* d = reinterpret_cast<Dp*>(dh);
    if (!d || !handle) return fail("dp: null communicator or handle");
    if (root < 0 || root >= d->world) return fail("dp: root outside the communicator");
    Rccl* r = rccl();
    std::vector<pf::TensorSpan> spans;
    if (pf::handle_tensor_spans(kind, handle, spans)) return -1;
    if (d->rank == root)
        for (auto& sp : spans)
            if (!sp.set) return fail("dp: tensor " + sp.name + " is not set on the root rank");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int rc;
    // one grouped operation: RCCL fuses the members instead of paying a launch and a ring set-up per tensor (~950 of them for
    // Paraformer-large; xGMI is point-to-point: few large transfers, not many small ones)
    if ((rc = nccl_ok(r->GroupStart(), "ncclGroupStart"))) return rc;
    for (auto& sp : spans)
        if ((rc = nccl_ok(r->Broadcast(sp.dev, sp.dev, sp.elems, /*ncclFloat32*/ 7, root, d->comm, s), "ncclBroadcast"))) { (void)r->GroupEnd(); return rc; }
    if ((rc = nccl_ok(r->GroupEnd(), "ncclGroupEnd"))) return rc;
    // planes / resolved tables derived from the old weights are dropped now; what re-derives them reads the new values on
    // `stream` order or later (the derivations run on the forward's stream after this call returns)
    if (hipStreamSynchronize(s) != hipSuccess) return fail("dp: stream synchronisation after the broadcast failed");
    return pf::handle_weights_replaced(kind, handle);
}

}  // namespace

extern "C" {

int pf_dp_unique_id(void* id_out, int32_t cap) {
    Rccl* r = rccl();
    if (!r->err.empty()) return fail("dp: " + r->err);
    if (!id_out || cap < (int32_t)sizeof(UniqueId)) return fail("dp: the unique id needs 128 bytes");
    UniqueId id;
    int rc = nccl_ok(r->GetUniqueId(&id), "ncclGetUniqueId");
    if (rc) return rc;
    memcpy(id_out, &id, sizeof(id));
    return (int)sizeof(id);
}

pf_dp* pf_dp_create(const void* unique_id, int32_t id_bytes, int32_t world, int32_t rank) {
    Rccl* r = rccl();
    if (!r->err.empty()) { fail("dp: " + r->err); return nullptr; }
    if (!unique_id || id_bytes != (int32_t)sizeof(UniqueId) || world < 1 || rank < 0 || rank >= world) { fail("dp: bad communicator arguments"); return nullptr; }
    Dp* d = new Dp();
    d->world = world; d->rank = rank;
    if (hipGetDevice(&d->device) != hipSuccess) { delete d; fail("dp: no current device"); return nullptr; }
    UniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    if (nccl_ok(r->CommInitRank(&d->comm, world, id, rank), "ncclCommInitRank")) { delete d; return nullptr; }
    return reinterpret_cast<pf_dp*>(d);
}

int pf_dp_destroy(pf_dp* dh) {
    Dp* d = reinterpret_cast<Dp*>(dh);
    if (!d) return 0;
    int rc = d->comm ? nccl_ok(rccl()->CommDestroy(d->comm), "ncclCommDestroy") : 0;
    delete d;
    return rc;
}
int pf_dp_world(const pf_dp* dh) { return dh ? reinterpret_cast<const Dp*>(dh)->world : -1; }
int pf_dp_rank(const pf_dp* dh) { return dh ? reinterpret_cast<const Dp*>(dh)->rank : -1; }

int pf_dp_broadcast_encoder(pf_dp* dh, pf_encoder* h, int32_t root, void* stream) { return broadcast_handle(dh, pf::HANDLE_ENCODER, h, root, stream); }
int pf_dp_broadcast_predictor(pf_dp* dh, pf_predictor* h, int32_t root, void* stream) { return broadcast_handle(dh, pf::HANDLE_PREDICTOR, h, root, stream); }
int pf_dp_broadcast_decoder(pf_dp* dh, pf_decoder* h, int32_t root, void* stream) { return broadcast_handle(dh, pf::HANDLE_DECODER, h, root, stream); }
int pf_dp_broadcast_ctc(pf_dp* dh, pf_ctc* h, int32_t root, void* stream) { return broadcast_handle(dh, pf::HANDLE_CTC, h, root, stream); }

int pf_dp_broadcast_raw(pf_dp* dh, void* buf_dev, int64_t bytes, int32_t root, void* stream) {
    Dp* d = reinterpret_cast<Dp*>(dh);
    if (!d || !buf_dev || bytes <= 0) return fail("dp: null communicator / buffer or empty broadcast");
    if (root < 0 || root >= d->world) return fail("dp: root outside the communicator");
    return nccl_ok(rccl()->Broadcast(buf_dev, buf_dev, (size_t)bytes, /*ncclInt8*/ 0, root, d->comm, reinterpret_cast<hipStream_t>(stream)), "ncclBroadcast");
}

int pf_dp_gather_ids(pf_dp* dh, const int32_t* ids_dev, int64_t count, int32_t* out_dev, int32_t root, void* stream) {
    Dp* d = reinterpret_cast<Dp*>(dh);
    if (!d || !ids_dev || count <= 0) return fail("dp: null communicator / ids or empty gather");
    if (root < 0 || root >= d->world) return fail("dp: root outside the communicator");
    if (d->rank == root && !out_dev) return fail("dp: the root rank needs an output buffer of world * count int32");
    return nccl_ok(rccl()->Gather(ids_dev, out_dev, (size_t)count, /*ncclInt32*/ 2, root, d->comm, reinterpret_cast<hipStream_t>(stream)), "ncclGather");
}

}  // extern "C"
