// C-ABI layer, streaming family: pf_stream_* (lock-step stream batch, hipGraph-captured step).
#include "engine_internal.h"

namespace pf {


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
    // the layers' linear_k_v weights as one matrix (Stream.kvcat_*): rows [l * 2 D, (l + 1) * 2 D) = layer l
    if (dc.n_blocks > 0) {
        const size_t rows = (size_t)dc.n_blocks * 2 * D, n = rows * D;
        if (st->kvcat_w.ensure(sizeof(float) * n) || st->kvcat_b.ensure(sizeof(float) * rows) || st->kvcat_2.ensure(sizeof(unsigned short) * 2 * n)) return -2;
        for (int l = 0; l < dc.n_blocks; ++l) {
            PF_HIP_TRY(hipMemcpyAsync(st->kvcat_w.as<float>() + (size_t)l * 2 * D * D, d->layers[l].kv_w, sizeof(float) * 2 * D * D, hipMemcpyDeviceToDevice, s));
            PF_HIP_TRY(hipMemcpyAsync(st->kvcat_b.as<float>() + (size_t)l * 2 * D, d->layers[l].kv_b, sizeof(float) * 2 * D, hipMemcpyDeviceToDevice, s));
        }
        float amax = 0.f;
        if (TensorTable::dev_absmax(st->kvcat_w.as<float>(), n, &amax, s)) return -2;
        st->kvcat_e = amax > 0.f ? 14 - (int)floorf(log2f(amax)) : 0;
        if ((rc = launch_split2(st->kvcat_w.as<float>(), D, st->kvcat_2.as<unsigned short>(), D, n, (int)rows, D, ldexpf(1.f, st->kvcat_e), s))) return rc;
    }
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

// ---- constants of the fp32 step's LayerNorm -> GEMM pairs (Stream.ln_consts)
static size_t ln_enc_stride(const Stream* st) { return 2 * ((size_t)3 * st->e->cfg.d_model + st->e->cfg.ffn_dim); }
static size_t ln_dec_stride(const Stream* st) { return 2 * ((size_t)st->d->cfg.ffn_dim + 2 * st->d->cfg.d_model); }
static size_t ln_dec_base(const Stream* st) { return ln_enc_stride(st) * st->e->layers.size(); }
static size_t ln_voc_base(const Stream* st) { return ln_dec_base(st) + ln_dec_stride(st) * (st->d->cfg.n_blocks + 1); }
static bool stream_ln_ready(const Stream* st) {
    return st->ln_consts.p && st->e->resolved && st->d->resolved && st->ln_ver_e == st->e->tt.version && st->ln_ver_d == st->d->tt.version;
}
static int stream_prepare_ln_consts(Stream* st, hipStream_t s) {
    Encoder* e = st->e; Decoder* d = st->d;
    int rc;
    if (!e->resolved && (rc = encoder_resolve(e))) return rc;
    if (!d->resolved && (rc = decoder_resolve(d))) return rc;
    const int D = e->cfg.d_model, F = e->cfg.ffn_dim, Fd = d->cfg.ffn_dim, V = d->cfg.vocab_size;
    if (st->ln_consts.ensure(sizeof(float) * (ln_voc_base(st) + 2 * (size_t)V))) return -2;
    float* base = st->ln_consts.as<float>();
    for (size_t l = 0; l < e->layers.size(); ++l) {
        const EncLayerW& w = e->layers[l];
        float* c = base + l * ln_enc_stride(st);
        if (w.in_dim == D && (rc = launch_ln_consts(w.qkv_w, w.in_pad, 3 * D, D, w.n1g, w.n1b, w.qkv_b, c, c + 3 * D, s))) return rc;
        c += 2 * 3 * D;
        if ((rc = launch_ln_consts(w.w1, D, F, D, w.n2g, w.n2b, w.b1, c, c + F, s))) return rc;
    }
    for (int l = 0; l <= d->cfg.n_blocks; ++l) {
        const DecLayerW& w = l < d->cfg.n_blocks ? d->layers[l] : d->last;
        float* c = base + ln_dec_base(st) + l * ln_dec_stride(st);
        if ((rc = launch_ln_consts(w.w1, D, Fd, D, w.n1g, w.n1b, w.b1, c, c + Fd, s))) return rc;
        c += 2 * Fd;
        if ((rc = launch_ln_consts(w.w2, Fd, D, Fd, w.fng, w.fnb, nullptr, c, c + D, s))) return rc;
        c += 2 * D;
        if (l < d->cfg.n_blocks && (rc = launch_ln_consts(w.q_w, D, D, D, w.n3g, w.n3b, w.q_b, c, c + D, s))) return rc;
    }
    float* cv = base + ln_voc_base(st);
    if ((rc = launch_ln_consts(d->tt.get("output_layer.weight"), D, V, D, d->tt.get("after_norm.weight"), d->tt.get("after_norm.bias"),
                               d->tt.get("output_layer.bias"), cv, cv + V, s))) return rc;
    st->ln_ver_e = e->tt.version; st->ln_ver_d = d->tt.version;
    return 0;
}

// f16x2 step: the decoder FFN's inner LayerNorm (ffn_dim wide) in the second launch of a split-K w_1 -- replaces a launch, for
// handles small enough that the extra blocks find idle CUs (the rule of short_k)
static bool stream_short_k(const Stream* st);
static bool x2_fold_fn(const Stream* st) { return st->ln_folded && stream_short_k(st); }

static int stream_token_rows(const Stream* st, int W, int is_final) {
    const int fires = W + 1 + (is_final ? 1 : 0);
    return fires < st->Nmax ? fires : st->Nmax;
}

// the f16x2 step's K = d_model projections in the split-K form: by the handle's size (streams x largest window), see Stream.short_k
static bool stream_short_k(const Stream* st) {
    // 1: every K = d_model projection when the handle is small, 2: always, 3: linear_out only (its second launch replaces norm2's) when small
    return st->short_k == 2 || ((st->short_k == 1 || st->short_k == 3) && (size_t)st->S * st->Wmax <= 2048);
}

// enqueue one chunk on `s` (no host synchronisation, no allocation after the first call with this shape)
static int stream_enqueue(Stream* st, int n, int is_final, int tail, hipStream_t s) {
    Encoder* e = st->e; Predictor* p = st->p; Decoder* d = st->d;
    const pf_encoder_config& ec = e->cfg;
    const int S = st->S, D = ec.d_model, F = ec.ffn_dim, Din = ec.input_dim, Dpad = round_up(Din, 64);
    const int W = tail ? st->keep : st->keep + n;
    // token rows per stream of THIS step: the integrate-and-fire loop runs over the carried remainder, the W window frames and (final
    // step) the tail weight, and fires at most once per iteration (cif_chunk_kernel) -- the 600 ms geometry's 15-frame window gives
    // 16 rows (one 16-row tile of the small-M kernels) where the handle's capacity is 23
    const int M = S * W, Nmax = stream_token_rows(st, W, is_final);
    const StreamDev* dev = st->dev_state.as<StreamDev>();
    if (st->wide_k && !st->ws_part.p) {
        if (st->ws_part.ensure(sizeof(float) * WS_PART_FLOATS) || st->ws_count.ensure(sizeof(int) * WS_TILES)) return -2;
        PF_HIP_TRY(hipMemset(st->ws_count.p, 0, sizeof(int) * WS_TILES));
    }
    StreamModeScope small_m_kernels(st->wide_k ? st->ws_part.as<float>() : nullptr, st->wide_k ? st->ws_count.as<int>() : nullptr);
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
                       st->mem2.ensure(sizeof(unsigned short) * 2 * Mz * D) ||
                       e->splitk.ensure(sizeof(float) * 4 * Mz * (stream_short_k(st) ? (F > 3 * D ? F : 3 * D) : D))))
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
    // (by the row count of the step: the 16- and 32-row forms of the kernel gain a launch, the 64-row form loses to its registers)
    const bool carry = st->ln_carry && !st->x2 && D % 16 == 0 && (M <= 32 || st->ln_carry > 1) && stream_ln_ready(st);
    if (carry && st->ln_stats.ensure(sizeof(float) * 2 * (size_t)S * st->Wmax * (D / 16))) return -2;
    for (size_t l = 0; l < e->layers.size(); ++l) {
        EncChunkCtx cc{st->enc_cap > 0 ? st->enc_ring.as<float>() + l * ring_layer : nullptr, st->enc_cap, dev, append_rows,
                       st->lensW.as<int>(), st->x2};
        cc.mod = st->enc_mod;
        cc.fsmn_rides = st->fsmn_rides;
        cc.x2_attn_planes = st->x2 && st->ln_folded;
        cc.x2_short_k = (st->x2 && stream_short_k(st)) ? (st->short_k >= 3 ? 1 : 2) : 0;
        cc.x2_fold = st->ln_folded;
        if (st->x2 && st->ln_folded && F >= 4 * D && F % 128 == 0) {
            // (the condition of gemm2c's split-K form; block l + 1 must take the folded planes: same width, no padding columns)
            const bool nxt = l + 1 < e->layers.size() && e->layers[l + 1].in_dim == D && e->layers[l + 1].in_pad == D;
            cc.x2_out_next = nxt ? &e->layers[l + 1] : nullptr;
            cc.x2_in_ready = l > 0 && e->layers[l].in_dim == D && e->layers[l].in_pad == D;
        }
        if (carry) {
            cc.ln_c_qkv = st->ln_consts.as<float>() + l * ln_enc_stride(st);
            cc.ln_c_w1 = cc.ln_c_qkv + 2 * 3 * D;
            cc.ln_stats = st->ln_stats.as<float>();
            cc.ln_in_ready = l > 0 && e->layers[l].in_dim == D;     // block l - 1's w_2 left them
            cc.next = l + 1 < e->layers.size() ? &e->layers[l + 1] : nullptr;
        }
        if (l == 0) rc = encoder_block(e, e->layers[0], st->win.as<float>(), Din, x, S, W, s, &cc);
        else rc = encoder_block(e, e->layers[l], x, D, x, S, W, s, &cc);
        if (rc) return rc;
    }
    if (st->enc_mod > 0 && append_rows > 0) {
        RingFoldArgs rf{st->enc_ring.as<float>(), ring_layer, (int)e->layers.size(), S, st->enc_cap, st->enc_mod, 2 * D, append_rows, dev};
        if ((rc = launch_ring_fold(rf, s))) return rc;
    }
    StreamAdvanceArgs adv{};
    adv.st = st->dev_state.as<StreamDev>(); adv.n_frames = tail ? st->keep : n;   // the tail chunk re-feeds `keep` rows (embedding.py:478)
    adv.enc_rows = append_rows; adv.enc_cap = st->enc_cap; adv.enc_mod = st->enc_mod;
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
        // (the decoder FFN's inner norm folded into a split-K w_1 needs ffn_dim-wide slices: small handles only, like short_k)
        if (d->splitk.ensure(sizeof(float) * 4 * (size_t)Mq * (x2_fold_fn(st) ? dc.ffn_dim : D))) return -2;
        t2p = d->t16.as<unsigned short>(); c2p = d->ctx16.as<unsigned short>();
        if ((rc = launch_split2(enc_out, D, st->mem2.as<unsigned short>(), D, (size_t)Mk * D, Mk, D, pow2f(st->e_mem), s))) return rc;
        mem2 = st->mem2.as<unsigned short>();
    }
    PF_HIP_TRY(hipMemcpyAsync(dx, st->embeds.p, sizeof(float) * (size_t)Mq * D, hipMemcpyDeviceToDevice, s));
    const size_t dring_layer = (size_t)S * st->dec_cap * 2 * D;
    const size_t dfsmn_layer = (size_t)S * (dc.kernel_size - 1) * D;
    // ---- fp32 step: every layer's key/value projection of the step's encoder rows in one launch, ahead of the token chain
    const size_t kv_layer = (size_t)S * st->Wmax * 2 * D;
    const bool kv_batched = !x2 && st->kv_batched && dc.n_blocks <= 16 && D % 16 == 0;
    if (kv_batched) {
        if (d->kv.ensure(sizeof(float) * kv_layer * dc.n_blocks)) return -2;
        GemmArgs g{};
        g.A = enc_out; g.lda = D; g.ldw = D; g.ldc = 2 * D; g.M = Mk; g.N = 2 * D; g.K = D;
        GemmBatch t{};
        t.n = dc.n_blocks;
        for (int l = 0; l < dc.n_blocks; ++l) {
            t.W[l] = d->layers[l].kv_w; t.bias[l] = d->layers[l].kv_b; t.C[l] = d->kv.as<float>() + l * kv_layer;
        }
        ProfScope ps(PROF_GEMM, 2.0 * Mk * (double)(2 * D) * D * dc.n_blocks, s);
        if ((rc = launch_gemm_skinny_batch(g, t, s))) return rc;
    }
    // ---- f16x2 step: the same in one GEMM over the concatenated weight planes (Stream.kvcat_*): layer l's K | V are columns
    // [l * 2 D, (l + 1) * 2 D) of a [rows, n_blocks * 2 D] result
    const bool kv_cat = x2 && st->kv_batched && dc.n_blocks > 0 && st->kvcat_2.p;
    const int kv_ld = kv_cat ? dc.n_blocks * 2 * D : 2 * D;
    if (kv_cat) {
        if (d->kv.ensure(sizeof(float) * (size_t)S * st->Wmax * kv_ld)) return -2;
        if ((rc = gemm2_simple(mem2, D, Mk, st->e_mem, st->kvcat_2.as<unsigned short>(), st->kvcat_e, st->kvcat_b.as<float>(), d->kv.as<float>(), kv_ld,
                               kv_ld, D, 0, nullptr, 0, s))) return rc;
    }
    // ---- fp32 step of a few token rows: the decoder's LayerNorms ride in the launches on either side (gemm_skinny.hip,
    // DecFsmnChunkArgs.ln_*): 11 launches per layer become 6 with the batched projection above
    const bool dcarry = st->ln_carry && !x2 && (Mq <= 32 || st->ln_carry > 1) && Nmax <= 24 && D % 16 == 0 && dc.ffn_dim % 16 == 0 &&
                        dc.ffn_dim <= 2048 && stream_ln_ready(st);
    if (dcarry && (st->dec_ln_a.ensure(sizeof(float) * 2 * (size_t)Mq * (D / 16)) || st->dec_ln_b.ensure(sizeof(float) * 2 * (size_t)Mq * (D / 16)) ||
                   st->dec_ln_f.ensure(sizeof(float) * 2 * (size_t)Mq * (dc.ffn_dim / 16))))
        return -2;
    float* sA = st->dec_ln_a.as<float>();
    float* sB = st->dec_ln_b.as<float>();
    float* sF = st->dec_ln_f.as<float>();
    // st_in != nullptr: the LayerNorm form (gemm_skinny.hip) with the pair's constants c (c1 [N], then c2); g_ = gamma, or nullptr when
    // the producer stored gamma a (out_g of ITS call: the gamma of the LayerNorm that follows)
    auto gemm_ln = [&](const float* A, int lda, const float* Wt, const float* bias, float* C, int N, int K, int relu, const float* R2,
                       float* st_out, const float* st_in, const float* g_, const float* c, const float* out_g = nullptr) {
        GemmArgs g{};
        g.A = A; g.lda = lda; g.W = Wt; g.ldw = K; g.bias = bias; g.R2 = R2; g.ldr2 = N; g.C = C; g.ldc = N; g.M = Mq; g.N = N; g.K = K;
        g.relu = relu; g.ln_stats_out = st_out; g.ln_stats_in = st_in; g.ln_g = g_; g.ln_eps = dc.ln_eps;
        g.ln_c1 = c; g.ln_c2 = c ? c + N : nullptr; g.out_gamma = out_g;
        ProfScope ps(PROF_GEMM, 2.0 * Mq * (double)N * K, s);
        return gemm(g, s);
    };
    // FFN of a layer: norm1 from the partials the layer before left in sA (or its own launch), the FFN's inner norm between
    // w_1 and w_2; `out_stats`: where w_2 leaves the partials of its output
    // (w_1 stores relu(..) * gamma of the FFN's inner norm: w_2's loop then multiplies nothing)
    auto ffn_carry = [&](const DecLayerW& w, int l, bool a_ready, float* out_stats) {
        const int F = dc.ffn_dim;
        const float* c = st->ln_consts.as<float>() + ln_dec_base(st) + (size_t)l * ln_dec_stride(st);
        int r;
        if (a_ready) r = gemm_ln(dx, D, w.w1, w.b1, d->ffn.as<float>(), F, D, 1, nullptr, sF, sA, w.n1g, c, w.fng);
        else {
            if ((r = layernorm(dx, D, w.n1g, w.n1b, t1, D, Mq, D, D, dc.ln_eps, s))) return r;
            r = gemm_ln(t1, D, w.w1, w.b1, d->ffn.as<float>(), F, D, 1, nullptr, sF, nullptr, nullptr, nullptr, w.fng);
        }
        if (r) return r;
        return gemm_ln(d->ffn.as<float>(), F, w.w2, nullptr, t2, D, F, 0, nullptr, out_stats, sF, nullptr, c + 2 * F);
    };
    for (int l = 0; l < dc.n_blocks; ++l) {
        const DecLayerW& w = d->layers[l];
        float* kv_l = d->kv.as<float>() + (kv_batched ? l * kv_layer : kv_cat ? (size_t)l * 2 * D : 0);
        DecFsmnChunkArgs fa{};
        fa.in = t1; fa.resid = dx; fa.out = dx; fa.w = w.fsmn_w; fa.state = st->dec_fsmn.as<float>() + l * dfsmn_layer;
        fa.n_valid = st->n_fired.as<int>(); fa.S = S; fa.N = Nmax; fa.C = D;
        if (dcarry) {
            if ((rc = ffn_carry(w, l, l > 0, sA))) return rc;
            fa.in = t2; fa.ln_stats_in = sA; fa.ln_g = w.n2g; fa.ln_b = w.n2b; fa.ln_eps = dc.ln_eps; fa.ln_stats_out = sB;
            if ((rc = launch_dec_fsmn_chunk(fa, s))) return rc;
            if ((rc = gemm_ln(dx, D, w.q_w, w.q_b, d->q.as<float>(), D, D, 0, nullptr, nullptr, sB, w.n3g,
                              st->ln_consts.as<float>() + ln_dec_base(st) + (size_t)l * ln_dec_stride(st) + 2 * dc.ffn_dim + 2 * D))) return rc;
        } else {
            const bool fold = x2 && st->ln_folded && dc.ffn_dim % 128 == 0;       // norm2 in the second launch of the split-K w_2
            const FoldedLn n2{w.n2g, w.n2b, t1, 0, 1.f};
            if ((rc = x2 ? dec_ffn_x2(d, w, dx, t2, Mq, s, d->splitk.as<float>(), fold ? &n2 : nullptr, x2_fold_fn(st)) : dec_ffn(d, w, dx, t2, Mq, s))) return rc;
            if (!fold && (rc = layernorm(t2, D, w.n2g, w.n2b, t1, D, Mq, D, D, dc.ln_eps, s))) return rc;
            if ((rc = launch_dec_fsmn_chunk(fa, s))) return rc;
        }
        if (dcarry) {                      // (norm3 + the query projection: above)
        } else if (x2) {
            {
                ProfScope ps(PROF_LN, 8.0 * Mq * (double)D, s);
                if ((rc = launch_layernorm(dx, D, w.n3g, w.n3b, reinterpret_cast<float*>(t2p), D, Mq, D, D, dc.ln_eps, s, 3, 0,
                                           (size_t)Mq * D, pow2f(w.e_n3)))) return rc;
            }
            if ((rc = gemm2_simple(t2p, D, Mq, w.e_n3, w.q_2, w.ew_q, w.q_b, d->q.as<float>(), D, D, D, 0, nullptr, 0, s))) return rc;
            if (!kv_cat && (rc = gemm2_simple(mem2, D, Mk, st->e_mem, w.kv_2, w.ew_kv, w.kv_b, d->kv.as<float>(), 2 * D, 2 * D, D, 0, nullptr, 0, s)))
                return rc;
        } else {
            if ((rc = layernorm(dx, D, w.n3g, w.n3b, t1, D, Mq, D, D, dc.ln_eps, s))) return rc;
            if ((rc = gemm_simple(t1, D, w.q_w, D, w.q_b, d->q.as<float>(), D, Mq, D, D, 0, nullptr, 0, nullptr, 0, s))) return rc;
        }
        if (!x2 && !kv_batched &&
            (rc = gemm_simple(enc_out, D, w.kv_w, D, w.kv_b, kv_l, 2 * D, Mk, 2 * D, D, 0, nullptr, 0, nullptr, 0, s))) return rc;
        AttnArgs at{};
        at.Q = d->q.as<float>(); at.ldq = D; at.O = d->ctx.as<float>(); at.ldo = D; at.B = S; at.H = dc.n_heads;
        at.Tq = Nmax; at.scale = powf((float)(D / dc.n_heads), -0.5f);
        if (st->dec_cap > 0) {
            float* ring = st->dec_ring.as<float>() + l * dring_layer;
            at.K = ring; at.ldk = 2 * D; at.V = ring + D; at.ldv = 2 * D; at.Tk = st->dec_cap;
            at.K2 = kv_l; at.ldk2 = kv_ld; at.V2 = kv_l + D; at.ldv2 = kv_ld; at.T2 = W; at.n2 = W;
            at.n1_dev = st->dec_valid.as<int>(); at.n1_stride = 1;
        } else {
            at.K = kv_l; at.ldk = kv_ld; at.V = kv_l + D; at.ldv = kv_ld; at.Tk = W;
            at.klens = st->lensW.as<int>();
        }
        bool appended = false;
        if (st->dec_cap > 0 && !kv_batched && !kv_cat) {
            at.app_rows = W; at.app_r0 = 0; at.app_wp = st->dec_wp.as<int>(); at.app_wp_stride = 1; at.app_gate = st->n_fired.as<int>();
        }
        const bool o2 = x2 && st->ln_folded && Nmax <= 32;
        if (o2) { at.O2 = c2p; at.ldo2 = D; at.o2_plane = (size_t)Mq * D; at.o2_scale = pow2f(st->e_ctx[l]); }
        if ((rc = attention(at, 4.0 * S * (double)Nmax * W * D, s, false, 128, &appended))) return rc;
        if (st->dec_cap > 0 && !appended && !kv_batched && !kv_cat) {
            RingAppendArgs ra{};
            ra.src = kv_l; ra.ldsrc = 2 * D; ra.src_T = W; ra.r0 = 0; ra.rows = W; ra.cols = 2 * D;
            ra.ring = st->dec_ring.as<float>() + l * dring_layer; ra.cap = st->dec_cap; ra.S = S; ra.st = nullptr;
            ra.wp_dev = st->dec_wp.as<int>(); ra.gate_dev = st->n_fired.as<int>();
            if ((rc = launch_ring_append(ra, s))) return rc;
        }
        if (x2) {
            if (!o2 && (rc = launch_split2(d->ctx.as<float>(), D, c2p, D, (size_t)Mq * D, Mq, D, pow2f(st->e_ctx[l]), s))) return rc;
            if ((rc = gemm2_simple(c2p, D, Mq, st->e_ctx[l], w.o_2, w.ew_o, w.o_b, dx, D, D, D, 0, dx, D, s))) return rc;
        } else if (dcarry) {
            if ((rc = gemm_ln(d->ctx.as<float>(), D, w.o_w, w.o_b, dx, D, D, 0, dx, sA, nullptr, nullptr, nullptr))) return rc;
        } else if ((rc = gemm_simple(d->ctx.as<float>(), D, w.o_w, D, w.o_b, dx, D, Mq, D, D, 0, nullptr, 0, dx, D, s))) return rc;
    }
    if (st->dec_cap > 0 && (kv_batched || kv_cat) && dc.n_blocks > 0) {
        // every layer's append in one launch (the rings are read by the attentions above, written here)
        RingAppendArgs ra{};
        ra.src = d->kv.as<float>(); ra.ldsrc = kv_ld; ra.src_T = W; ra.r0 = 0; ra.rows = W; ra.cols = 2 * D;
        ra.ring = st->dec_ring.as<float>(); ra.cap = st->dec_cap; ra.S = S; ra.st = nullptr;
        ra.wp_dev = st->dec_wp.as<int>(); ra.gate_dev = st->n_fired.as<int>();
        ra.n_layers = dc.n_blocks; ra.src_layer = kv_cat ? (size_t)2 * D : kv_layer; ra.ring_layer = dring_layer;
        if ((rc = launch_ring_append(ra, s))) return rc;
    }
    if (st->dec_cap > 0) {
        StreamAdvanceArgs ad{};
        ad.dec_valid = st->dec_valid.as<int>(); ad.dec_wp = st->dec_wp.as<int>(); ad.gate = st->n_fired.as<int>();
        ad.S = S; ad.dec_rows = W; ad.dec_cap = st->dec_cap;
        if ((rc = launch_stream_advance_dec(ad, s))) return rc;
    }
    bool an_folded = false;
    if (dcarry) {
        // decoders3's FFN, then after_norm on the fetch of the vocabulary projection (the hidden rows are not an output of a step)
        if ((rc = ffn_carry(d->last, dc.n_blocks, dc.n_blocks > 0, sA))) return rc;
        if (d->pval.ensure(sizeof(float) * (size_t)Mq * V)) return -2;
        if ((rc = gemm_ln(t2, D, d->tt.get("output_layer.weight"), d->tt.get("output_layer.bias"), d->pval.as<float>(), V, D, 0, nullptr,
                          nullptr, sA, d->tt.get("after_norm.weight"), st->ln_consts.as<float>() + ln_voc_base(st)))) return rc;
        if ((rc = launch_argmax_rows(d->pval.as<float>(), V, Mq, V, st->ids.as<int32_t>(), s))) return rc;
    } else if (x2) {
        // decoders3's FFN; after_norm's planes (the vocabulary projection's operand) from the second launch of its split-K w_2
        const bool fold = st->ln_folded && dc.ffn_dim % 128 == 0;
        const FoldedLn an{d->tt.get("after_norm.weight"), d->tt.get("after_norm.bias"), reinterpret_cast<float*>(t2p), 3, pow2f(st->e_an)};
        if ((rc = dec_ffn_x2(d, d->last, dx, t2, Mq, s, d->splitk.as<float>(), fold ? &an : nullptr, x2_fold_fn(st)))) return rc;
        an_folded = fold;
    } else if ((rc = dec_ffn(d, d->last, dx, t2, Mq, s))) return rc;
    if (dcarry) {                          // (after_norm + the vocabulary projection: above)
    } else if (x2) {
        // after_norm writes two-plane operands, the vocabulary projection runs with the row arg-max fused into its epilogue
        // (the offline greedy route of decoder_forward_impl)
        auto wv = d->tt.b16.find("output_layer.weight#split2");
        if (wv == d->tt.b16.end()) { set_error("stream: f16x2 step without prepared vocabulary planes"); return -1; }
        const int ew_v = d->tt.exp2.at("output_layer.weight#split2");
        if (!an_folded) {
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

extern "C" {

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
        dec_left != K - 1 || d->n_blocks2 != 0) {
        set_error("stream: unsupported config (d_model 512, look_back >= 0 (finite), max_tokens <= 96, no decoders2, causal decoder "
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
    // The reference trims the encoder's K / V cache to look_back * chunk_size[1] rows from the SECOND chunk on and leaves the first
    // chunk's cache -- chunk_left + n rows -- as it is (sanm/attention.py:353-361). Where that is more than the trim size (chunk_left >
    // (look_back - 1) * chunk_cur: e.g. [5, 10, 5] with look-back 1) the ring holds the larger count and trims from its second append
    // on (RingAppendArgs.mod); found by tools/fuzz_gpu_streaming_vs_oracle.py, profiles/r06ak_streaming_fuzz.json
    if (st->enc_cap > 0 && c.chunk_right > 0 && c.chunk_left + c.max_frames > st->enc_cap) {
        st->enc_mod = st->enc_cap;
        st->enc_cap = c.chunk_left + c.max_frames;
    }
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
    if (k != "gemm_mode" && k != "ln_carry" && k != "fsmn_rides" && k != "kv_batched" && k != "wide_k" && k != "ln_folded" && k != "short_k") { set_error("stream_set_option: unknown key " + k); return -1; }
    PF_REQUIRE(k != "gemm_mode" || value == 0 || value == 3, "stream_set_option: gemm_mode is 0 (fp32 kernels) or 3 (f16x2)");
    PF_HIP_TRY(hipStreamSynchronize(st->stream));
    if (k == "ln_carry") st->ln_carry = value < 0 ? 0 : value > 2 ? 2 : value;
    else if (k == "fsmn_rides") st->fsmn_rides = value != 0;
    else if (k == "kv_batched") st->kv_batched = value != 0;
    else if (k == "wide_k") st->wide_k = value != 0;
    else if (k == "ln_folded") st->ln_folded = value != 0;
    else if (k == "short_k") st->short_k = value < 0 ? 0 : value > 3 ? 3 : value;
    else {
        if (value == 3) {
            int rc = stream_prepare_x2(st, st->stream);
            if (rc) return rc;
        }
        st->x2 = value == 3;
    }
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
    st->pending = false;                                     // (a step in flight is dropped with the session it belonged to)
    return 0;
}

int pf_stream_step(pf_stream* sh, const float* feats, int32_t n_frames, int32_t is_final, int32_t tail_chunk,
                   int32_t* ids_host, int32_t* n_tokens_host, float* enc_out, void* stream) {
    PF_REQUIRE(sh && ids_host && n_tokens_host, "stream_step: null argument");
    int rc = pf_stream_step_begin(sh, feats, n_frames, is_final, tail_chunk, enc_out, stream);
    if (rc) return rc;
    return pf_stream_step_end(sh, ids_host, n_tokens_host);
}

/* the two halves of pf_stream_step: _begin enqueues the step on the handle's own HIP stream and returns without waiting, _end waits
 * for it and hands the ids / counts over. Several handles (each over its OWN encoder / predictor / decoder handles: the workspaces
 * belong to those) may have a step in flight at a time: a step of a few dozen streams leaves most of the chip idle, and steps of
 * independent handles overlap there. */
int pf_stream_step_begin(pf_stream* sh, const float* feats, int32_t n_frames, int32_t is_final, int32_t tail_chunk, float* enc_out,
                         void* stream) {
    Stream* st = reinterpret_cast<Stream*>(sh);
    hipStream_t us = reinterpret_cast<hipStream_t>(stream);
    PF_REQUIRE(st, "stream_step: null argument");
    PF_REQUIRE(!st->pending, "stream_step_begin: the previous step of this handle was not collected (pf_stream_step_end)");
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
    if (!st->x2 && st->ln_carry && !stream_ln_ready(st)) {
        // first step, or a handle's weights changed: the LayerNorm -> GEMM constants follow them (graphs hold no stale values, only
        // the buffer's address -- which may have moved)
        for (auto& kv : st->graphs) (void)hipGraphExecDestroy(kv.second);
        st->graphs.clear();
        st->seen.clear();
        if ((rc = stream_prepare_ln_consts(st, s))) return rc;
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
    st->start_idx += tail_chunk ? st->keep : n;
    st->pending = true;
    note_concurrent_streams();                               // an asynchronous step: other handles' kernels may now meet it on the chip
    st->pending_rows = stream_token_rows(st, W, is_final);
    return 0;
}

int pf_stream_step_end(pf_stream* sh, int32_t* ids_host, int32_t* n_tokens_host) {
    Stream* st = reinterpret_cast<Stream*>(sh);
    PF_REQUIRE(st && ids_host && n_tokens_host, "stream_step_end: null argument");
    PF_REQUIRE(st->pending, "stream_step_end: no step in flight");
    PF_HIP_TRY(hipStreamSynchronize(st->stream));            // a failed synchronisation leaves the step pending (the caller may retry / reset)
    st->pending = false;
    const int S = st->S;
    // the step ran stream_token_rows() rows per stream; the caller's layout is [n_streams, max_tokens]
    const int rows = st->pending_rows;
    for (int i = 0; i < S; ++i) {
        memcpy(ids_host + (size_t)i * st->Nmax, st->h_ids + (size_t)i * rows, sizeof(int32_t) * (size_t)rows);
        for (int k = rows; k < st->Nmax; ++k) ids_host[(size_t)i * st->Nmax + k] = 0;
    }
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


}  // extern "C"
