// C-ABI layer, predictor / decoder / CTC family: pf_predictor_*, pf_decoder_*, pf_ctc_*.
#include "engine_internal.h"

namespace pf {

// ContextualParaformerDecoder keeps its last attention block under "last_decoder." (contextual_paraformer/decoder.py:241)
std::string dec_layer_prefix(bool contextual, int n_blocks, int i) {
    return (contextual && i == n_blocks - 1) ? std::string("last_decoder.") : "decoders." + std::to_string(i) + ".";
}


int decoder_resolve(Decoder* d) {
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
    d->layers2.clear();
    for (int i = 0; i < d->n_blocks2; ++i) {
        const std::string p = "decoders2." + std::to_string(i) + ".";
        DecLayerW w{};
        w.n1g = d->tt.get(p + "norm1.weight"); w.n1b = d->tt.get(p + "norm1.bias");
        w.w1 = d->tt.get(p + "feed_forward.w_1.weight"); w.b1 = d->tt.get(p + "feed_forward.w_1.bias");
        w.fng = d->tt.get(p + "feed_forward.norm.weight"); w.fnb = d->tt.get(p + "feed_forward.norm.bias");
        w.w2 = d->tt.get(p + "feed_forward.w_2.weight");
        w.n2g = d->tt.get(p + "norm2.weight"); w.n2b = d->tt.get(p + "norm2.bias");
        w.fsmn_w = d->tt.get(p + "self_attn.fsmn_block.weight");
        d->layers2.push_back(w);
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
int vocab_project(const float* hidden, int M, int D, const float* W, const float* bias, int V, float* logits,
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



// PositionwiseFeedForwardDecoderSANM (sanm/positionwise_feed_forward.py:12-33): w_2(LN(relu(w_1 x))), w_2 bias-free
int gemm3_simple(const unsigned short* A3, int lda, int M, const unsigned short* W3, const float* bias, float* C,
                        int ldc, int N, int K, int relu, hipStream_t s) {
    if (!W3) return -2;
    Gemm3Args g{};
    g.A = A3; g.lda = lda; g.a_plane = (size_t)M * lda; g.W = W3; g.ldw = K; g.w_plane = (size_t)N * K;
    g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.relu = relu;
    ProfScope ps(PROF_GEMM3, 2.0 * M * (double)N * K, s);
    return launch_gemm_split3(g, s);
}

int gemm2_simple(const unsigned short* A2, int lda, int M, int ea, const unsigned short* W2, int ew, const float* bias,
                        float* C, int ldc, int N, int K, int relu, const float* R2, int ldr2, hipStream_t s,
                        const float* oscale_dev, float* splitk_part) {
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
int dec_layer_x2(Decoder* d, DecLayerW& w, const std::string& p, bool attn, hipStream_t s) {
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
// ln (streaming step, split-K w_2): the LayerNorm that follows the block rides in w_2's second launch (Gemm2Args.ln_*)
// fold_fn (with splitk_part of 4 * M * ffn_dim floats): the FFN's inner norm rides in the second launch of a split-K w_1
int dec_ffn_x2(Decoder* d, const DecLayerW& w, const float* x, float* out, int M, hipStream_t s, float* splitk_part, const FoldedLn* ln, bool fold_fn) {
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
    if (ln && splitk_part && F % 128 == 0) {
        Gemm2Args g{};
        g.A = f2p; g.lda = F; g.a_plane = (size_t)M * F; g.W = w.w2_2; g.ldw = F; g.w_plane = (size_t)D * F;
        g.oscale = pow2f(-(w.e_fn + w.ew_2)); g.C = out; g.ldc = D; g.M = M; g.N = D; g.K = F; g.ksplit = 4; g.part = splitk_part;
        g.ln_g = ln->g; g.ln_b = ln->b; g.ln_eps = d->cfg.ln_eps; g.ln_y = ln->y; g.ln_ldy = D; g.ln_out = ln->out; g.ln_plane = (size_t)M * D;
        g.ln_oscale = ln->oscale;
        if (!w.w2_2) return -2;
        ProfScope ps(PROF_GEMM3, 2.0 * M * (double)D * F, s);
        return launch_gemm_f16x2(g, s);
    }
    if (ln) { set_error("decoder: LayerNorm folding needs the split-K form of w_2"); return -1; }
    return gemm2_simple(f2p, F, M, w.e_fn, w.w2_2, w.ew_2, nullptr, out, D, D, F, 0, nullptr, 0, s, nullptr, splitk_part);
}

// w1_3 != nullptr (bf16x3 mode): norm1 writes the three planes and w_1 runs on the bf16 matrix cores
int dec_ffn(Decoder* d, const DecLayerW& w, const float* x, float* out, int M, hipStream_t s,
                   const unsigned short* w1_3) {
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
int decoder_forward_bf16(Decoder* d, const float* memory, int B, int T, int N, int32_t* ids, float* hidden_out,
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
    for (int l = 0; l < d->n_blocks2; ++l) {                       // decoders2 (decoder.py:363-380): no cross-attention, taps centred
        const DecLayerW& w = d->layers2[l];
        if ((rc = ffn_bf16("decoders2." + std::to_string(l) + ".", w, x, t2))) return rc;
        if ((rc = layernorm(t2, D, w.n2g, w.n2b, t1, D, Mq, D, D, c.ln_eps, s))) return rc;
        FsmnArgs fa{};
        fa.in = t1; fa.ldin = D; fa.w = w.fsmn_w; fa.R = x; fa.ldr = D; fa.out = x; fa.ldo = D;
        fa.lens = d->tok_lens.as<int>(); fa.B = B; fa.T = N; fa.C = D; fa.K = c.kernel_size; fa.left_pad = (c.kernel_size - 1) / 2;
        if ((rc = fsmn(fa, s))) return rc;
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

}  // namespace pf

using namespace pf;

extern "C" {

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

}  // extern "C"

// relu(Conv1d(D, D, l + r + 1)(pad(hidden))) (cif_predictor.py:275-278) as ONE exact-fp32 GEMM whose A operand is gathered from the
// row-shifted views of `hidden` by the DMA sources (gemm_f32.hip, GemmArgs.conv_*): the k order per output element is that of the
// im2col GEMM of rounds 1-5, so alphas are bitwise unchanged, and the [M, taps D] column matrix (197 MB written + 196 MB read per
// headline step) is gone
static int predictor_conv(Predictor* p, const float* hidden, int B, int T, hipStream_t s) {
    const pf_predictor_config& c = p->cfg;
    const int D = c.d_model, taps = c.l_order + c.r_order + 1;
    const size_t M = (size_t)B * T;
    if (p->conv.ensure(sizeof(float) * M * D)) return -2;
    if (!p->zero_row.p) {
        if (p->zero_row.ensure(256)) return -2;
        PF_HIP_TRY(hipMemsetAsync(p->zero_row.p, 0, 256, s));
    }
    GemmArgs g{};
    g.A = hidden; g.lda = D; g.W = p->tt.get("cif_conv1d.weight"); g.ldw = taps * D; g.bias = p->tt.get("cif_conv1d.bias");
    g.C = p->conv.as<float>(); g.ldc = D; g.M = (int)M; g.N = D; g.K = taps * D; g.relu = 1;
    g.conv_taps = taps; g.conv_D = D; g.conv_T = T; g.conv_left = c.l_order; g.conv_zero = p->zero_row.as<float>();
    ProfScope ps(PROF_GEMM, 2.0 * (double)M * D * (double)(taps * D), s);
    return launch_gemm_f32(g, s);
}

namespace pf {
// everything of pf_predictor_alphas up to the scan, ENQUEUED into scan-state slot `slot`; the token counts stay on the device
// (predictor_counts_dev)
int predictor_alphas_enqueue(Predictor* p, int slot, const float* hidden, const int32_t* lens_host, int B, int T, hipStream_t s) {
    PF_REQUIRE(p && hidden && lens_host && B > 0 && T > 0 && (slot == 0 || slot == 1), "predictor_alphas: null/empty argument");
    for (int b = 0; b < B; ++b) PF_REQUIRE(lens_host[b] >= 1 && lens_host[b] <= T, "predictor_alphas: lens out of range");
    std::string first;
    if (p->tt.missing(&first)) { set_error("predictor: tensor not set: " + first); return -3; }
    const pf_predictor_config& c = p->cfg;
    const int D = c.d_model, Te = T + 1;
    Predictor::CifState& S = p->st[slot];
    if (S.alphas.ensure(sizeof(float) * (size_t)B * Te) || S.peaks.ensure(sizeof(float) * (size_t)B * Te) ||
        S.rems.ensure(sizeof(float) * (size_t)B * Te) || S.flags.ensure(sizeof(int) * (size_t)B * Te) ||
        S.nfires.ensure(sizeof(int) * (size_t)B))
        return -2;
    int rc;
    if ((rc = upload_lens(S.lens, lens_host, B, s))) return rc;
    if ((rc = predictor_conv(p, hidden, B, T, s))) return rc;
    AlphaArgs aa{};
    aa.conv = p->conv.as<float>(); aa.w = p->tt.get("cif_output.weight"); aa.bias = p->tt.get("cif_output.bias");
    aa.lens = S.lens.as<int>(); aa.alphas = S.alphas.as<float>(); aa.B = B; aa.T = T; aa.D = D; aa.T_ext = Te;
    aa.smooth = c.smooth_factor; aa.noise = c.noise_threshold;
    if ((rc = launch_alpha(aa, s))) return rc;
    CifScanArgs sa{};
    sa.alphas = S.alphas.as<float>(); sa.peaks = S.peaks.as<float>(); sa.rems = S.rems.as<float>();
    sa.fire_flag = S.flags.as<int>(); sa.n_fires = S.nfires.as<int>(); sa.lens = S.lens.as<int>(); sa.B = B;
    sa.T = T; sa.tail_threshold = c.tail_threshold; sa.tail_mask = c.tail_mask;
    if (p->v3) {
        if (S.curs.ensure(sizeof(float) * (size_t)B * Te) || S.ntok.ensure(sizeof(int) * (size_t)B)) return -2;
        if ((rc = launch_cif_scan_loop(sa, S.curs.as<float>(), S.ntok.as<int>(), s))) return rc;
    } else if ((rc = launch_cif_scan(sa, s))) {
        return rc;
    }
    S.B = B; S.T = T;
    return 0;
}
// V3 reports floor(sum alphas) (cif_predictor.py:383), V2's count of fires is the same number by construction
const int32_t* predictor_counts_dev(Predictor* p, int slot) {
    return reinterpret_cast<const int32_t*>(p->v3 ? p->st[slot].ntok.p : p->st[slot].nfires.p);
}
const float* predictor_alphas_dev(Predictor* p, int slot) { return p->st[slot].alphas.as<float>(); }
const float* predictor_peaks_dev(Predictor* p, int slot) { return p->st[slot].peaks.as<float>(); }
int predictor_embeds_slot(Predictor* p, int slot, const float* hidden, int B, int T, int N, float* embeds, hipStream_t s) {
    PF_REQUIRE(p && hidden && embeds && N >= 0 && (slot == 0 || slot == 1), "predictor_embeds: null argument");
    Predictor::CifState& S = p->st[slot];
    PF_REQUIRE(B == S.B && T == S.T, "predictor_embeds: call pf_predictor_alphas with the same batch first");
    CifEmitArgs ea{};
    ea.hidden = hidden; ea.alphas = S.alphas.as<float>(); ea.rems = S.rems.as<float>();
    ea.fire_flag = S.flags.as<int>(); ea.embeds = embeds; ea.B = B; ea.T = T; ea.D = p->cfg.d_model; ea.N = N;
    if (p->v3) {
        if (N <= 0) return 0;
        ea.alphas = S.curs.as<float>();
        return launch_cif_emit_loop(ea, s);
    }
    return launch_cif_emit(ea, s);
}
}  // namespace pf

extern "C" {

int pf_predictor_alphas(pf_predictor* ph, const float* hidden, const int32_t* lens_host, int32_t B, int32_t T,
                        float* alphas, float* peaks, int32_t* token_num, void* stream) {
    Predictor* p = reinterpret_cast<Predictor*>(ph);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(p && token_num, "predictor_alphas: null/empty argument");
    int rc;
    if ((rc = predictor_alphas_enqueue(p, 0, hidden, lens_host, B, T, s))) return rc;
    const size_t Te = (size_t)T + 1;
    if (alphas) PF_HIP_TRY(hipMemcpyAsync(alphas, p->st[0].alphas.p, sizeof(float) * (size_t)B * Te, hipMemcpyDeviceToDevice, s));
    if (peaks) PF_HIP_TRY(hipMemcpyAsync(peaks, p->st[0].peaks.p, sizeof(float) * (size_t)B * Te, hipMemcpyDeviceToDevice, s));
    PF_HIP_TRY(hipMemcpyAsync(token_num, predictor_counts_dev(p, 0), sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, s));
    PF_HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int pf_predictor_embeds(pf_predictor* ph, const float* hidden, int32_t B, int32_t T, int32_t N, float* embeds,
                        void* stream) {
    return predictor_embeds_slot(reinterpret_cast<Predictor*>(ph), 0, hidden, B, T, N, embeds, reinterpret_cast<hipStream_t>(stream));
}

// The two steps without the synchronisation between them, for callers that keep two batches in flight (include/paraformer_hip.h)
int pf_predictor_alphas_begin(pf_predictor* ph, int32_t slot, const float* hidden, const int32_t* lens_host, int32_t B, int32_t T,
                              float* alphas, float* peaks, int32_t* counts_pinned_host, void* stream) {
    Predictor* p = reinterpret_cast<Predictor*>(ph);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(p && counts_pinned_host && (slot == 0 || slot == 1), "predictor_alphas_begin: null argument or slot not 0 / 1");
    int rc;
    if ((rc = predictor_alphas_enqueue(p, slot, hidden, lens_host, B, T, s))) return rc;
    const size_t Te = (size_t)T + 1;
    if (alphas) PF_HIP_TRY(hipMemcpyAsync(alphas, p->st[slot].alphas.p, sizeof(float) * (size_t)B * Te, hipMemcpyDeviceToDevice, s));
    if (peaks) PF_HIP_TRY(hipMemcpyAsync(peaks, p->st[slot].peaks.p, sizeof(float) * (size_t)B * Te, hipMemcpyDeviceToDevice, s));
    PF_HIP_TRY(hipMemcpyAsync(counts_pinned_host, predictor_counts_dev(p, slot), sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, s));
    return 0;
}

int pf_predictor_embeds_slot(pf_predictor* ph, int32_t slot, const float* hidden, int32_t B, int32_t T, int32_t N, float* embeds,
                             void* stream) {
    return predictor_embeds_slot(reinterpret_cast<Predictor*>(ph), slot, hidden, B, T, N, embeds, reinterpret_cast<hipStream_t>(stream));
}

// one direction-pair of torch.nn.LSTM on a time-major input: gates-major input projections by the fp32 MFMA GEMM
// (W_ih . X_tm^T, one GEMM per direction), then the per-step recurrence (lstm.hip)
}  // extern "C"
namespace pf {
int lstm_forward(const LstmW& w, const float* x_tm, int T, int B, int D, int H, int ndir, float* out, int out_layout,
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
}  // namespace pf
extern "C" {


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
    if ((rc = upload_lens(p->ts_lens, lens_host, B, s))) return rc;        // (its own buffer: both scan-state slots may be in flight)
    if (p->tok_dev.ensure(sizeof(int) * (size_t)B)) return -2;
    if (upload_h2d(p->tok_dev.p, token_num_host, sizeof(int32_t) * (size_t)B, s)) return -2;
    const float* src = hidden;
    if (p->c3.use_cif1_cnn) {                                   // the head sees relu(cif_conv1d(hidden)) instead (:317-320)
        if ((rc = predictor_conv(p, hidden, B, T, s))) return rc;
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
        ua.lens = p->ts_lens.as<int>(); ua.alphas = us_alphas; ua.B = B; ua.Bs = Bs; ua.T = Tu; ua.C = 2 * D; ua.U = U;
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
/* decoders2 (paraformer/decoder.py:363-380, :436-437): n_layers = num_blocks - att_layer_num blocks of FFN + FSMN memory (built with
 * sanm_shfit 0: the taps centred) and NO cross-attention, run between `decoders` and `decoders3`. Tensors "decoders2.<i>.norm1|norm2.*",
 * "decoders2.<i>.feed_forward.*", "decoders2.<i>.self_attn.fsmn_block.weight". Call once, before the first set_tensor. */
int pf_decoder_set_decoders2(pf_decoder* dh, int32_t n_layers) {
    Decoder* d = reinterpret_cast<Decoder*>(dh);
    PF_REQUIRE(d && n_layers >= 0 && n_layers <= 64, "decoder_set_decoders2: null handle or layer count out of range");
    PF_REQUIRE(d->n_blocks2 == 0 || d->n_blocks2 == n_layers, "decoder_set_decoders2: already set");
    if (d->n_blocks2 == n_layers) return 0;
    const int D = d->cfg.d_model, F = d->cfg.ffn_dim;
    int rc = 0;
    for (int i = 0; i < n_layers; ++i) {
        const std::string p = "decoders2." + std::to_string(i) + ".";
        rc |= d->tt.add(p + "norm1.weight", D);
        rc |= d->tt.add(p + "norm1.bias", D);
        rc |= d->tt.add(p + "feed_forward.w_1.weight", (int64_t)F * D);
        rc |= d->tt.add(p + "feed_forward.w_1.bias", F);
        rc |= d->tt.add(p + "feed_forward.norm.weight", F);
        rc |= d->tt.add(p + "feed_forward.norm.bias", F);
        rc |= d->tt.add(p + "feed_forward.w_2.weight", (int64_t)D * F);
        rc |= d->tt.add(p + "norm2.weight", D);
        rc |= d->tt.add(p + "norm2.bias", D);
        rc |= d->tt.add(p + "self_attn.fsmn_block.weight", (int64_t)D * d->cfg.kernel_size);
    }
    if (rc) return -2;
    d->n_blocks2 = n_layers;
    d->resolved = false;
    return 0;
}
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
    // profiling scopes count algorithmic work: valid memory rows, valid (token, frame) pairs
    double mem_rows = 0, tok_mem = 0;
    for (int b = 0; b < B; ++b) { mem_rows += mem_lens[b]; tok_mem += (double)mem_lens[b] * tok_lens[b]; }
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
        for (int l = 0; l < d->n_blocks2; ++l)
            if ((rc = dec_layer_x2(d, d->layers2[l], "decoders2." + std::to_string(l) + ".", false, s))) return rc;
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
                ProfScope ps(PROF_GEMM3, 2.0 * mem_rows * 2.0 * D * D, s);
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
                ProfScope ps(PROF_ATTN, 4.0 * tok_mem * D, s);
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
    // decoders2: FFN -> norm2 -> FSMN memory + residual, no cross-attention; the blocks are built with sanm_shfit 0 whatever the
    // attention blocks use (decoder.py:363-380, DecoderLayerSANM.forward :97-107 with src_attn None)
    for (int l = 0; l < d->n_blocks2; ++l) {
        const DecLayerW& w = d->layers2[l];
        const unsigned short* w1_3 = w3("decoders2." + std::to_string(l) + ".feed_forward.w_1.weight", F, D);
        if (x3 && !w1_3) return -2;
        if (x2) rc = dec_ffn_x2(d, w, x, t2, Mq, s);
        else rc = dec_ffn(d, w, x, t2, Mq, s, w1_3);
        if (rc) return rc;
        if ((rc = layernorm(t2, D, w.n2g, w.n2b, t1, D, Mq, D, D, c.ln_eps, s))) return rc;
        FsmnArgs fa{};
        fa.in = t1; fa.ldin = D; fa.w = w.fsmn_w; fa.R = x; fa.ldr = D; fa.out = x; fa.ldo = D;
        fa.lens = d->tok_lens.as<int>(); fa.B = B; fa.T = N; fa.C = D; fa.K = c.kernel_size; fa.left_pad = (c.kernel_size - 1) / 2;
        fa.offs = offs_dev;
        if ((rc = fsmn(fa, s))) return rc;
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


}  // extern "C"
