// The offline Paraformer forward as ONE call at the C ABI: features -> token ids.
// What Paraformer.inference does between the frontend and the tokenizer (funasr/models/paraformer/model.py:286-346 encode /
// calc_predictor / cal_decoder_with_predictor, :614-616 the rounded token count and the "nothing fired" early-out, :642 the
// arg-max), at the tensor boundary the reference itself exports for this path (funasr/models/paraformer/export_meta.py:44-68:
// speech [B, T, 560] f32 + speech_lengths [B] i32 -> logits + token_num; here the arg-max is fused, so token ids come back).
// The pipeline object borrows the three module handles and owns only the intermediate buffers; the chain is exactly the one
// funasr_amd/paraformer.py ran module by module until round 5 (pf_encoder_forward -> pf_predictor_alphas [the CIF token count
// sizes the decoder, like the .item() at cif_predictor.py:311] -> pf_predictor_embeds -> pf_decoder_forward with the fused arg-max),
// so the results are bitwise those of the module calls (tested).
#include <vector>

#include "../../include/paraformer_hip.h"
#include "engine_internal.h"

// Round 6: the forward is two phases, and a serving loop may interleave them across batches.
//   pf_paraformer_begin  : encoder + predictor (alphas, scan) ENQUEUED; the token counts start their way to a pinned host buffer
//   pf_paraformer_finish : waits for THAT copy only (an event, not the stream), then enqueues embeds + decoder + fused arg-max
// The CIF token count still sizes the decoder exactly (the .item() of cif_predictor.py:311; no padded N_max, no speculative row
// budget, no re-launch path), but the wait no longer drains the GPU: with `begin(i + 1)` issued before `finish(i)` the stream holds
// batch i + 1's encoder while the host reads batch i's counts and launches its decoder behind it. Two slots (encoder output +
// scan state) make that legal, and `finish` may be given another stream than `begin`: the decoder of batch i then runs BESIDE the
// encoder of batch i + 1 (its few-row kernels fill what the big GEMMs leave idle; an event orders the slot's reuse).
// pf_paraformer_forward = begin + finish back to back: bitwise the module chain, as before.
namespace pf {
namespace {
struct Slot {
    DevBuf enc;
    std::vector<int32_t> lens;
    int32_t* counts_host = nullptr;      // pinned, [cap_B]
    int cap_B = 0;
    hipEvent_t ev = nullptr;
    hipEvent_t done = nullptr;           // recorded behind the decoder of the batch that last used this slot
    bool done_pending = false;
    int B = 0, T = 0;
    bool open = false;                   // begun, not finished
};
struct Pipeline {
    pf_encoder* e = nullptr;
    pf_predictor* p = nullptr;
    pf_decoder* d = nullptr;
    int D = 0;
    Slot slot[2];
    int next = 0;
    DevBuf embeds, ids;
    int last_slot = 0, last_N = 0;
    ~Pipeline() {
        for (Slot& s : slot) {
            if (s.counts_host) (void)hipHostFree(s.counts_host);
            if (s.ev) (void)hipEventDestroy(s.ev);
            if (s.done) (void)hipEventDestroy(s.done);
        }
    }
};
}  // namespace
}  // namespace pf

using namespace pf;

extern "C" {

pf_paraformer* pf_paraformer_create(pf_encoder* e, pf_predictor* p, pf_decoder* d) {
    if (!e || !p || !d) { set_error("paraformer: null module handle"); return nullptr; }
    const Encoder* E = reinterpret_cast<const Encoder*>(e);
    const Predictor* P = reinterpret_cast<const Predictor*>(p);
    const Decoder* Dd = reinterpret_cast<const Decoder*>(d);
    if (E->cfg.d_model != P->cfg.d_model || E->cfg.d_model != Dd->cfg.d_model || Dd->cfg.vocab_size <= 0 || Dd->contextual) {
        set_error("paraformer: encoder / predictor / decoder disagree on d_model, or the decoder has no output layer of its own");
        return nullptr;
    }
    Pipeline* m = new Pipeline();
    m->e = e; m->p = p; m->d = d; m->D = E->cfg.d_model;
    return reinterpret_cast<pf_paraformer*>(m);
}

void pf_paraformer_destroy(pf_paraformer* mh) { delete reinterpret_cast<Pipeline*>(mh); }

int pf_paraformer_begin(pf_paraformer* mh, const float* feats_dev, const int32_t* lens_host, int32_t B, int32_t T,
                        const float* pe_dev, void* stream) {
    Pipeline* m = reinterpret_cast<Pipeline*>(mh);
    PF_REQUIRE(m && feats_dev && lens_host && B > 0 && T > 0, "paraformer_begin: null argument or empty batch");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int si = m->next;
    Slot& S = m->slot[si];
    PF_REQUIRE(!S.open, "paraformer_begin: both slots are in flight (call pf_paraformer_finish for the older batch first)");
    Predictor* P = reinterpret_cast<Predictor*>(m->p);
    if (P->v3) PF_REQUIRE(!m->slot[si ^ 1].open, "paraformer_begin: a V3 predictor keeps one scan state (finish the previous batch first)");
    const size_t D = (size_t)m->D;
    if (S.enc.ensure(sizeof(float) * (size_t)B * T * D)) return -2;
    if (S.cap_B < B) {
        if (S.counts_host) { (void)hipHostFree(S.counts_host); S.counts_host = nullptr; S.cap_B = 0; }
        PF_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&S.counts_host), sizeof(int32_t) * (size_t)B, hipHostMallocDefault));
        S.cap_B = B;
    }
    if (!S.ev) PF_HIP_TRY(hipEventCreateWithFlags(&S.ev, hipEventDisableTiming));
    if (!S.done) PF_HIP_TRY(hipEventCreateWithFlags(&S.done, hipEventDisableTiming));
    // pf_paraformer_finish may run on ANOTHER stream (the decoder of batch i beside the encoder of batch i + 1): what this slot held
    // -- encoder output and scan state of batch i - 1 -- must have been consumed before the encoder writes it again
    if (S.done_pending) { PF_HIP_TRY(hipStreamWaitEvent(s, S.done, 0)); S.done_pending = false; }
    S.lens.assign(lens_host, lens_host + B);
    int rc;
    if ((rc = pf_encoder_forward(m->e, feats_dev, lens_host, B, T, pe_dev, S.enc.as<float>(), -1, stream))) return rc;
    if ((rc = predictor_alphas_enqueue(P, si, S.enc.as<float>(), lens_host, B, T, s))) return rc;
    PF_HIP_TRY(hipMemcpyAsync(S.counts_host, predictor_counts_dev(P, si), sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, s));
    PF_HIP_TRY(hipEventRecord(S.ev, s));
    S.B = B; S.T = T; S.open = true;
    m->next = si ^ 1;
    return si;
}

int pf_paraformer_finish(pf_paraformer* mh, int32_t ticket, int32_t* ids_dev, int32_t ids_ld, int32_t* token_num_host,
                         float* alphas_dev, float* peaks_dev, void* stream) {
    Pipeline* m = reinterpret_cast<Pipeline*>(mh);
    PF_REQUIRE(m && token_num_host && (ticket == 0 || ticket == 1), "paraformer_finish: null argument or a ticket pf_paraformer_begin did not return");
    PF_REQUIRE(!ids_dev || ids_ld > 0, "paraformer_finish: ids_ld must be positive");
    Slot& S = m->slot[ticket];
    PF_REQUIRE(S.open, "paraformer_finish: this ticket is not in flight");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    Predictor* P = reinterpret_cast<Predictor*>(m->p);
    const int B = S.B, T = S.T;
    const size_t D = (size_t)m->D, Te = (size_t)T + 1;
    S.open = false;
    PF_HIP_TRY(hipEventSynchronize(S.ev));                  // the counts of THIS batch are on the host; later work keeps running
    int N = 0;
    for (int b = 0; b < B; ++b) { token_num_host[b] = S.counts_host[b]; N = token_num_host[b] > N ? token_num_host[b] : N; }
    if (alphas_dev) PF_HIP_TRY(hipMemcpyAsync(alphas_dev, predictor_alphas_dev(P, ticket), sizeof(float) * (size_t)B * Te, hipMemcpyDeviceToDevice, s));
    if (peaks_dev) PF_HIP_TRY(hipMemcpyAsync(peaks_dev, predictor_peaks_dev(P, ticket), sizeof(float) * (size_t)B * Te, hipMemcpyDeviceToDevice, s));
    m->last_slot = ticket; m->last_N = N;
    struct DoneMark {                                        // whatever the exit, the slot's consumers end here on `s`
        Slot& S; hipStream_t s;
        ~DoneMark() { if (S.done && hipEventRecord(S.done, s) == hipSuccess) S.done_pending = true; }
    } done_mark{S, s};
    if (N == 0) return 0;                                    // model.py:615-616: nothing fired anywhere in the batch
    PF_REQUIRE(!ids_dev || N <= ids_ld, "paraformer_finish: ids_ld is smaller than the batch's largest token count");
    if (m->embeds.ensure(sizeof(float) * (size_t)B * N * D)) return -2;
    int rc;
    const float* enc = S.enc.as<float>();
    if ((rc = predictor_embeds_slot(P, ticket, enc, B, T, N, m->embeds.as<float>(), s))) return rc;
    int32_t* ids = ids_dev;
    if (!ids_dev || ids_ld != N) {
        if (m->ids.ensure(sizeof(int32_t) * (size_t)B * N)) return -2;
        ids = m->ids.as<int32_t>();
    }
    if ((rc = pf_decoder_forward(m->d, enc, S.lens.data(), m->embeds.as<float>(), token_num_host, B, T, N, nullptr, ids, nullptr, stream))) return rc;
    if (ids_dev && ids != ids_dev)
        PF_HIP_TRY(hipMemcpy2DAsync(ids_dev, sizeof(int32_t) * (size_t)ids_ld, ids, sizeof(int32_t) * (size_t)N, sizeof(int32_t) * (size_t)N, B,
                                    hipMemcpyDeviceToDevice, s));
    return N;
}

int pf_paraformer_forward(pf_paraformer* mh, const float* feats_dev, const int32_t* lens_host, int32_t B, int32_t T,
                          const float* pe_dev, int32_t* ids_dev, int32_t ids_ld, int32_t* token_num_host,
                          float* alphas_dev, float* peaks_dev, void* stream) {
    PF_REQUIRE(token_num_host, "paraformer_forward: null argument or empty batch");
    PF_REQUIRE(!ids_dev || ids_ld > 0, "paraformer_forward: ids_ld must be positive");
    const int t = pf_paraformer_begin(mh, feats_dev, lens_host, B, T, pe_dev, stream);
    if (t < 0) return t;
    return pf_paraformer_finish(mh, t, ids_dev, ids_ld, token_num_host, alphas_dev, peaks_dev, stream);
}

const float* pf_paraformer_encoder_out(const pf_paraformer* mh) {
    const Pipeline* m = reinterpret_cast<const Pipeline*>(mh);
    return m ? reinterpret_cast<const float*>(m->slot[m->last_slot].enc.p) : nullptr;
}
const float* pf_paraformer_embeds(const pf_paraformer* mh) {
    const Pipeline* m = reinterpret_cast<const Pipeline*>(mh);
    return (m && m->last_N > 0) ? reinterpret_cast<const float*>(m->embeds.p) : nullptr;
}

}  // extern "C"
