// The offline Paraformer forward as ONE call at the C ABI: features -> token ids.
// What Paraformer.inference does between the frontend and the tokenizer (funasr/models/paraformer/model.py:286-346 encode /
// calc_predictor / cal_decoder_with_predictor, :614-616 the rounded token count and the "nothing fired" early-out, :642 the
// arg-max), at the tensor boundary the reference itself exports for this path (funasr/models/paraformer/export_meta.py:44-68:
// speech [B, T, 560] f32 + speech_lengths [B] i32 -> logits + token_num; here the arg-max is fused, so token ids come back).
// The pipeline object borrows the three module handles and owns only the intermediate buffers; the chain is exactly the one
// funasr_amd/paraformer.py ran module by module until round 5 (pf_encoder_forward -> pf_predictor_alphas [the one host
// synchronisation: the CIF token count sizes the decoder, like the .item() at cif_predictor.py:311] -> pf_predictor_embeds ->
// pf_decoder_forward with the fused arg-max), so the results are bitwise those of the module calls (tested).
#include <vector>

#include "../../include/paraformer_hip.h"
#include "engine_internal.h"

namespace pf {
namespace {
struct Pipeline {
    pf_encoder* e = nullptr;
    pf_predictor* p = nullptr;
    pf_decoder* d = nullptr;
    int D = 0;
    DevBuf enc, alphas, peaks, embeds, ids;
    std::vector<int32_t> tok;
    int last_B = 0, last_T = 0, last_N = 0;
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

int pf_paraformer_forward(pf_paraformer* mh, const float* feats_dev, const int32_t* lens_host, int32_t B, int32_t T,
                          const float* pe_dev, int32_t* ids_dev, int32_t ids_ld, int32_t* token_num_host,
                          float* alphas_dev, float* peaks_dev, void* stream) {
    Pipeline* m = reinterpret_cast<Pipeline*>(mh);
    PF_REQUIRE(m && feats_dev && lens_host && token_num_host && B > 0 && T > 0, "paraformer_forward: null argument or empty batch");
    PF_REQUIRE(!ids_dev || ids_ld > 0, "paraformer_forward: ids_ld must be positive");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t D = (size_t)m->D;
    if (m->enc.ensure(sizeof(float) * (size_t)B * T * D)) return -2;
    if (!alphas_dev) { if (m->alphas.ensure(sizeof(float) * (size_t)B * (T + 1))) return -2; alphas_dev = m->alphas.as<float>(); }
    if (!peaks_dev) { if (m->peaks.ensure(sizeof(float) * (size_t)B * (T + 1))) return -2; peaks_dev = m->peaks.as<float>(); }
    float* enc = m->enc.as<float>();
    int rc;
    if ((rc = pf_encoder_forward(m->e, feats_dev, lens_host, B, T, pe_dev, enc, -1, stream))) return rc;
    if ((rc = pf_predictor_alphas(m->p, enc, lens_host, B, T, alphas_dev, peaks_dev, token_num_host, stream))) return rc;   // synchronises
    int N = 0;
    for (int b = 0; b < B; ++b) N = token_num_host[b] > N ? token_num_host[b] : N;
    m->last_B = B; m->last_T = T; m->last_N = N;
    if (N == 0) return 0;                                    // model.py:615-616: nothing fired anywhere in the batch
    PF_REQUIRE(!ids_dev || N <= ids_ld, "paraformer_forward: ids_ld is smaller than the batch's largest token count");
    if (m->embeds.ensure(sizeof(float) * (size_t)B * N * D)) return -2;
    if ((rc = pf_predictor_embeds(m->p, enc, B, T, N, m->embeds.as<float>(), stream))) return rc;
    int32_t* ids = ids_dev;
    if (!ids_dev || ids_ld != N) {
        if (m->ids.ensure(sizeof(int32_t) * (size_t)B * N)) return -2;
        ids = m->ids.as<int32_t>();
    }
    if ((rc = pf_decoder_forward(m->d, enc, lens_host, m->embeds.as<float>(), token_num_host, B, T, N, nullptr, ids, nullptr, stream))) return rc;
    if (ids_dev && ids != ids_dev)
        PF_HIP_TRY(hipMemcpy2DAsync(ids_dev, sizeof(int32_t) * (size_t)ids_ld, ids, sizeof(int32_t) * (size_t)N, sizeof(int32_t) * (size_t)N, B,
                                    hipMemcpyDeviceToDevice, s));
    return N;
}

const float* pf_paraformer_encoder_out(const pf_paraformer* mh) {
    const Pipeline* m = reinterpret_cast<const Pipeline*>(mh);
    return m ? reinterpret_cast<const float*>(m->enc.p) : nullptr;
}
const float* pf_paraformer_embeds(const pf_paraformer* mh) {
    const Pipeline* m = reinterpret_cast<const Pipeline*>(mh);
    return (m && m->last_N > 0) ? reinterpret_cast<const float*>(m->embeds.p) : nullptr;
}

}  // extern "C"
