// C-ABI layer, FSMN-VAD family: pf_vad_* (the decision state machine is vad_decision.hip).
#include "engine_internal.h"

using namespace pf;

extern "C" {

// -------------------------------------------------------------------------------------------------------- vad
// FSMN-VAD network (funasr/models/fsmn_vad_streaming/encoder.py:288-378): in_linear1 -> in_linear2 -> relu ->
// n x [linear (no bias) -> FSMN memory (+ cache) -> affine -> relu] -> out_linear1 -> out_linear2 -> softmax, reduced to
// the summed posterior of the silence pdfs. Dense layers on the fp32 GEMM kernels (K padded to 32 with zero columns),
// the memory / softmax kernels in vad.hip.

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


}  // extern "C"
