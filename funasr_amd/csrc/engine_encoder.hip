// C-ABI layer, encoder family: pf_encoder_* (SANMEncoder / SenseVoiceEncoderSmall / SANMEncoderChunkOpt blocks).
#include "engine_internal.h"

namespace pf {

static void enc_layer_names(std::vector<std::pair<std::string, int>>& out, const pf_encoder_config& c) {
    // (prefix, input_dim) of every SAN-M block in execution order
    out.push_back({"encoders0.0.", c.input_dim});
    for (int i = 0; i < c.n_blocks - 1; ++i) out.push_back({"encoders." + std::to_string(i) + ".", c.d_model});
    for (int i = 0; i < c.tp_blocks; ++i) out.push_back({"tp_encoders." + std::to_string(i) + ".", c.d_model});
}

int encoder_resolve(Encoder* e) {
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
int encoder_prepare_x2(Encoder* e, hipStream_t s) {
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
int encoder_default_pe(Encoder* e, int T, hipStream_t s) {
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


// mode 3 only: `xn_ready` = the planes of norm1(x_in) already lie in xn16 (written by the previous block's w_2 epilogue);
// `next` = the block whose norm1 this block's w_2 epilogue should apply (nullptr: none follows directly)
int encoder_block(Encoder* e, const EncLayerW& w, float* x_in, int ld_in, float* x, int B, int T,
                         hipStream_t s, const EncChunkCtx* cc, bool xn_ready, const EncLayerW* next) {
    // EncoderLayerSANM.forward (sanm/encoder.py:72-148), normalize_before, no concat_after
    const pf_encoder_config& c = e->cfg;
    const int M = (e->cur_offs && !cc) ? e->cur_M : B * T, D = c.d_model, F = c.ffn_dim;
    // profiling scopes count algorithmic work: valid frames, not the rows the padded / packed layouts compute
    const double Mw = (!cc && e->prof_rows > 0) ? e->prof_rows : (double)M;
    const double Aw = (!cc && e->prof_sq > 0) ? 4.0 * e->prof_sq * c.d_model : 4.0 * B * (double)T * T * c.d_model;
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
            g.M = M; g.N = N; g.K = K; g.relu = relu; g.tile = (!C2 && K > D && e->w2_tile) ? e->w2_tile : e->gemm_tile;
            if ((g.tile & 15) == 7 && !gemm_f16x2_w4_ok(g)) g.tile = 0;     // (shapes the four-wave kernel does not take: by shape)
            ProfScope ps(PROF_GEMM3, 2.0 * Mw * (double)N * K, s, C2 ? "enc.w_1 (planes out)" : (K > D ? "enc.w_2" : "enc.linear_out"));
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
            g.M = M; g.N = D; g.K = K; g.a_nt = (e->row_nt >= (K == D ? 1 : 2) ? 1 : 0) | (e->row_sched ? 0 : 2); g.block_rows = e->row_bm;
            ProfScope ps(PROF_GEMM3, 2.0 * Mw * (double)D * K, s, K > D ? "enc.w_2 row (+res +LN)" : "enc.linear_out row (+fsmn +res +LN)");
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
            ProfScope ps(PROF_GEMM3, 2.0 * Mw * 3.0 * D * w.in_dim, s, "enc.qkv (Q,K,V^T planes out)");
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
            ProfScope ps(PROF_ATTN, Aw, s, "enc.self_attention");
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
#if defined(PF_MEASUREMENT_KERNELS)      // the one-launch feed-forward (gemm_f16x2_ffn.hip): a tie with the pair at best, measurement library only
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
            ProfScope ps(PROF_GEMM3, 2.0 * Mw * 2.0 * (double)D * F, s, "enc.ffn fused (w_1 +relu +w_2 +res +LN)");
            return launch_ffn_f16x2(g, s);
        }
#endif
        if ((rc = gemm2(xn2, D, w.e_x2, w.w1_2, w.ew_1, w.b1, nullptr, 0, ffn2, w.e_h, F, D, 1, nullptr, 0, nullptr, 0))) return rc;
        if (fuse && encoder_w2_row_form(e, M)) {
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
    // ln: the LayerNorm that follows the projection, folded into the second launch of the split-K form (planes -> xn2c)
    struct LnFold { const float* g; const float* b; int e; };
    auto gemm2c = [&](const unsigned short* A, int lda, int ea, const unsigned short* W, int ew, const float* bias, float* C, int ldc,
                      unsigned short* C2, int ec, int N, int K, int relu, const float* R1, int ldr1, const float* R2, int ldr2,
                      const LnFold* ln = nullptr) {
        Gemm2Args g{};
        g.A = A; g.lda = lda; g.a_plane = (size_t)M * lda; g.W = W; g.ldw = K; g.w_plane = (size_t)N * K;
        g.oscale = pow2f(-(ea + ew)); g.bias = bias; g.R1 = R1; g.ldr1 = ldr1; g.R2 = R2; g.ldr2 = ldr2;
        g.C = C; g.ldc = ldc; g.C2 = C2; g.ldc2 = N; g.c_plane = (size_t)M * N; g.cscale = pow2f(ec);
        g.M = M; g.N = N; g.K = K; g.relu = relu;
        // the long-K projection (w_2) of a step is at most a block per CU: always in its split-K form here (by caller, whatever
        // the stream count, so a stream's result does not depend on its neighbours)
        if (C && K >= 4 * D && K % 128 == 0 && e->splitk.p) { g.ksplit = 4; g.part = e->splitk.as<float>(); }
        // a step of few rows leaves most CUs idle and a block's K loop runs at ~0.85 us per 32-deep stage whatever the block count:
        // the K = d_model projections in four slices too (x2_short_k: by the handle's stream count, never by the step's data)
        // (x2_short_k 1: only where the split's second launch REPLACES a LayerNorm launch -- linear_out + norm2; 2: every projection)
        if ((cc->x2_short_k == 2 || (cc->x2_short_k == 1 && ln)) && K % 128 == 0 && e->splitk.p) { g.ksplit = 4; g.part = e->splitk.as<float>(); }
        if (ln) {
            if (g.ksplit <= 1 || N != D) { set_error("encoder: LayerNorm folding needs the split-K form of an N = d_model projection"); return -1; }
            g.ln_g = ln->g; g.ln_b = ln->b; g.ln_eps = c.ln_eps; g.ln_y = reinterpret_cast<float*>(xn2c); g.ln_ldy = D;
            g.ln_out = 3; g.ln_plane = (size_t)M * D; g.ln_oscale = pow2f(ln->e);
        }
        ProfScope ps(PROF_GEMM3, 2.0 * M * (double)N * K, s);
        return launch_gemm_f16x2(g, s);
    };
    auto ln_planes = [&](const float* src, int ld, const float* g, const float* b, int dim, int dim_pad, int ex) {
        ProfScope ps(PROF_LN, 8.0 * M * (double)dim, s);
        return launch_layernorm(src, ld, g, b, reinterpret_cast<float*>(xn2c), dim_pad, M, dim, dim_pad, c.ln_eps, s, 3, 0,
                                (size_t)M * dim_pad, pow2f(ex));
    };
    // fp32 step with the LayerNorms carried by the small-M GEMMs (EncChunkCtx.ln_stats): gemm() routes to gemm_skinny.hip in a step
    const bool carry = cc && !x2c && cc->ln_stats && g_stream_mode && D % 16 == 0 && F % 16 == 0;
    // (st_in != nullptr: the LayerNorm form with gamma g_ and the pair's constants cst = c1 [N], then c2)
    auto gemm_ln = [&](const float* A, int lda, const float* Wt, int ldw, const float* bias, float* C, int ldc, int N, int K, int relu,
                       const float* R1, int ldr1, const float* R2, int ldr2, float* st_out, const float* st_in, const float* g_,
                       const float* cst) {
        GemmArgs g{};
        g.A = A; g.lda = lda; g.W = Wt; g.ldw = ldw; g.bias = bias; g.R1 = R1; g.ldr1 = ldr1; g.R2 = R2; g.ldr2 = ldr2;
        g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.relu = relu;
        g.ln_stats_out = st_out; g.ln_stats_in = st_in; g.ln_g = g_; g.ln_eps = c.ln_eps; g.ln_c1 = cst; g.ln_c2 = cst ? cst + N : nullptr;
        ProfScope ps(PROF_GEMM, 2.0 * M * (double)N * K, s);
        return gemm(g, s);
    };
    if (x2c && (!w.qkv_w2 || !w.out_w2 || !w.w1_2 || !w.w2_2)) { set_error("encoder: streaming f16x2 step without prepared weight planes"); return -1; }
    // norm1 -> fused QKV projection
    if (x2c) {
        // (norm1's planes already written by the block before: the second launch of its w_2, see gemm2c)
        if (!cc->x2_in_ready && (rc = ln_planes(x_in, ld_in, w.n1g, w.n1b, w.in_dim, w.in_pad, w.e_x1))) return rc;
        if ((rc = gemm2c(xn2c, w.in_pad, w.e_x1, w.qkv_w2, w.ew_qkv, w.qkv_b, qkv, 3 * D, nullptr, 0, 3 * D, w.in_pad, 0, nullptr, 0,
                         nullptr, 0))) return rc;
    } else if (carry && cc->ln_in_ready) {
        // norm1 on the fetch: the block before left the row partials of x in its w_2 epilogue
        if ((rc = gemm_ln(x_in, ld_in, w.qkv_w, w.in_pad, w.qkv_b, qkv, 3 * D, 3 * D, w.in_pad, 0, nullptr, 0, nullptr, 0, nullptr,
                          cc->ln_stats, w.n1g, cc->ln_c_qkv))) return rc;
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
    // streaming step: the memory block rides in the attention launch (AttnArgs.fs_*: same bits, one launch less in the chain)
    const bool fsmn_rides = cc && cc->fsmn_rides && g_stream_mode && T <= 32 && D / c.n_heads == 128 && c.kernel_size == 11 &&
                            fa.left_pad == 5 && D <= 1024;
    if (!fsmn_rides && (rc = fsmn(fa, s))) return rc;
    // scaled dot-product attention over valid keys (attention.py:284-306,324-326)
    AttnArgs aa{};
    if (fsmn_rides) { aa.fs_in = fa.in; aa.fs_ldin = fa.ldin; aa.fs_w = fa.w; aa.fs_out = fa.out; aa.fs_ldo = fa.ldo; aa.fs_T = T; }
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
        aa.app_mod = cc->mod;
    }
    // f16x2 step: the attention writes the out-projection's operand planes itself (no split2 launch)
    const bool o2 = x2c && cc->x2_attn_planes && g_stream_mode && T <= 32 && D / c.n_heads == 128;
    if (o2) { aa.O2 = ctx2c; aa.ldo2 = D; aa.o2_plane = (size_t)M * D; aa.o2_scale = pow2f(w.e_v); }
    if ((rc = attention(aa, 4.0 * B * (double)T * T * D, s, false, D / c.n_heads, &appended))) return rc;
    if (cc && cc->cap > 0 && cc->append_rows > 0 && !appended) {
        RingAppendArgs ra{};
        ra.src = qkv + D; ra.ldsrc = 3 * D; ra.src_T = T; ra.r0 = 0; ra.rows = cc->append_rows; ra.cols = 2 * D;
        ra.ring = cc->ring; ra.cap = cc->cap; ra.S = B; ra.st = cc->st; ra.mod = cc->mod;
        if ((rc = launch_ring_append(ra, s))) return rc;
    }
    // out projection + fsmn memory (+ residual when in == out, encoder.py:120-137)
    const float* resid = (w.in_dim == D) ? x_in : nullptr;
    if (x2c) {
        if (!o2 && (rc = launch_split2(ctx, D, ctx2c, D, (size_t)M * D, M, D, pow2f(w.e_v), s))) return rc;
        // (short-K split: norm2's planes come from linear_out's second launch)
        const LnFold n2{w.n2g, w.n2b, w.e_x2};
        const bool fold2 = cc->x2_short_k != 0 && cc->x2_fold && D % 128 == 0 && e->splitk.p;
        if ((rc = gemm2c(ctx2c, D, w.e_v, w.out_w2, w.ew_out, w.out_b, x, D, nullptr, 0, D, D, 0, mem, D, resid, ld_in, fold2 ? &n2 : nullptr))) return rc;
        // norm2 -> FFN -> residual: w_1 hands its relu output to w_2 as planes, like the offline mode
        if (!fold2 && (rc = ln_planes(x, D, w.n2g, w.n2b, D, D, w.e_x2))) return rc;
        if ((rc = gemm2c(xn2c, D, w.e_x2, w.w1_2, w.ew_1, w.b1, nullptr, 0, ffn2c, w.e_h, F, D, 1, nullptr, 0, nullptr, 0))) return rc;
        const LnFold n1n{cc->x2_out_next ? cc->x2_out_next->n1g : nullptr, cc->x2_out_next ? cc->x2_out_next->n1b : nullptr,
                         cc->x2_out_next ? cc->x2_out_next->e_x1 : 0};
        return gemm2c(ffn2c, F, w.e_h, w.w2_2, w.ew_2, w.b2, x, D, nullptr, 0, D, F, 0, nullptr, 0, x, D, cc->x2_out_next ? &n1n : nullptr);
    }
    if (carry) {
        // norm2 rides between linear_out and w_1; the next block's norm1 between w_2 and its QKV projection
        const bool emit = cc->next && cc->next->in_dim == D;
        if ((rc = gemm_ln(ctx, D, w.out_w, D, w.out_b, x, D, D, D, 0, mem, D, resid, ld_in, cc->ln_stats, nullptr, nullptr, nullptr))) return rc;
        if ((rc = gemm_ln(x, D, w.w1, D, w.b1, ffn, F, F, D, 1, nullptr, 0, nullptr, 0, nullptr, cc->ln_stats, w.n2g, cc->ln_c_w1))) return rc;
        return gemm_ln(ffn, F, w.w2, F, w.b2, x, D, D, F, 0, nullptr, 0, x, D, emit ? cc->ln_stats : nullptr, nullptr, nullptr, nullptr);
    }
    if ((rc = gemm_simple(ctx, D, w.out_w, D, w.out_b, x, D, M, D, D, 0, mem, D, resid, ld_in, s))) return rc;
    // norm2 -> FFN -> residual (encoder.py:141-146)
    if ((rc = layernorm(x, D, w.n2g, w.n2b, xn, D, M, D, D, c.ln_eps, s))) return rc;
    if ((rc = gemm_simple(xn, D, w.w1, D, w.b1, ffn, F, M, F, D, 1, nullptr, 0, nullptr, 0, s))) return rc;
    if ((rc = gemm_simple(ffn, F, w.w2, F, w.b2, x, D, M, D, F, 0, nullptr, 0, x, D, s))) return rc;
    return 0;
}

}  // namespace pf

using namespace pf;

extern "C" {

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
    // the product's keys: fuse_row, fsmn_fused (the unfused chains are the bitwise references of the fused kernels), w2_row (which form
    // w_2 takes), gemm_tile (0 by shape; 2 / 7 / 10 pin a shape for A/B runs). Everything else selects measured-and-off shapes and exists
    // in the measurement library only (`make measure`): DESIGN 3 lists each with the file that holds its numbers.
    if (k == "fuse_row") { PF_REQUIRE(value == 0 || value == 1, "encoder_set_option: fuse_row is 0 or 1"); e->fuse_row = value; return 0; }
    if (k == "fsmn_fused") { PF_REQUIRE(value == 0 || value == 1, "encoder_set_option: fsmn_fused is 0 or 1"); e->fsmn_fused = value; return 0; }
    if (k == "w2_row") { PF_REQUIRE(value >= 0 && value <= 2, "encoder_set_option: w2_row is 0, 1 or 2"); e->w2_row = value; return 0; }
#if defined(PF_MEASUREMENT_KERNELS)
    if (k == "gemm_tile") { PF_REQUIRE(value == 0 || value == 1 || value == 2 || value == 5 || value == 6 || value == 7 || value == 8 || value == 9 || value == 10 || value == 12, "encoder_set_option: gemm_tile is 0, 1, 2, 5, 6, 7, 8, 9, 10 or 12"); e->gemm_tile = value; return 0; }
    if (k == "ffn_abl") { e->ffn_abl = value; return 0; }
    if (k == "ffn_fused") { PF_REQUIRE(value >= 0 && value <= 2, "encoder_set_option: ffn_fused is 0, 1 or 2"); e->ffn_fused = value; return 0; }
    if (k == "row_bm") { PF_REQUIRE(value == 0 || value == 96 || value == 128 || value == 129 || value == 130, "encoder_set_option: row_bm is 0, 96, 128, 129 or 130"); e->row_bm = value; return 0; }
    if (k == "row_sched") { PF_REQUIRE(value == 0 || value == 2, "encoder_set_option: row_sched is 0 or 2"); e->row_sched = value; return 0; }
    if (k == "w2_tile") { PF_REQUIRE(value == 0 || value == 2 || value == 7 || value == 8 || value == 10, "encoder_set_option: w2_tile is 0, 2, 7, 8 or 10"); e->w2_tile = value; return 0; }
    if (k == "row_nt") { PF_REQUIRE(value >= 0 && value <= 2, "encoder_set_option: row_nt is 0, 1 or 2"); e->row_nt = value; return 0; }
    if (k == "attn_variant") { PF_REQUIRE(value == 0 || value == 1 || value == 3, "encoder_set_option: attn_variant is 0, 1 or 3"); e->attn_variant = value; return 0; }
#else
    if (k == "gemm_tile") { PF_REQUIRE(value == 0 || value == 2 || value == 7 || value == 10, "encoder_set_option: gemm_tile is 0, 2, 7 or 10"); e->gemm_tile = value; return 0; }
    if (k == "ffn_abl" || k == "ffn_fused" || k == "row_bm" || k == "row_sched" || k == "w2_tile" || k == "row_nt" || k == "attn_variant") {
        set_error("encoder_set_option: " + k + " selects a measured-and-off shape: build and load the measurement library (make -C funasr_amd/csrc measure; PF_LIB_PATH)");
        return -1;
    }
#endif
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
    return poison({&p->conv, &p->st[0].alphas, &p->st[0].peaks, &p->st[0].rems, &p->st[0].flags, &p->st[0].nfires, &p->st[1].alphas, &p->st[1].peaks,
                   &p->st[1].rems, &p->st[1].flags, &p->st[1].nfires}, byte);
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
    e->prof_rows = e->prof_sq = 0;
    for (int b = 0; b < B; ++b) { e->prof_rows += lens_host[b]; e->prof_sq += (double)lens_host[b] * lens_host[b]; }
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
        const int w2_rows = (int)M;
        bool xn_ready = false;
        for (int l = 0; l < total && !rc; ++l) {
            // the next block's norm1 rides in this block's w_2 epilogue unless another op sits between them (SenseVoice's after_norm)
            const bool boundary = c.tp_blocks > 0 && l + 1 == c.n_blocks;
            const EncLayerW* next = (l + 1 < total && !boundary) ? &e->layers[l + 1] : nullptr;
            rc = l == 0 ? encoder_block(e, e->layers[0], x0, Din, x, B, max_rows, s, nullptr, false, next)
                        : encoder_block(e, e->layers[l], x, D, x, B, max_rows, s, nullptr, xn_ready, next);
            xn_ready = next != nullptr && next->in_dim == D && encoder_w2_row_form(e, w2_rows);
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
    const int w2_rows = B * Tp;
    bool xn_ready = false;
    for (int l = 0; l < nrun; ++l) {
        // SANMVadEncoder: `encoders0` and all but the last of `encoders` are causal, the last one takes the VAD corner
        e->cur_mask_mode = e->vad_mask ? ((l >= 1 && l + 1 == total) ? 2 : 1) : 0;
        const bool boundary = c.tp_blocks > 0 && l + 1 == c.n_blocks;
        const EncLayerW* next = (l + 1 < nrun && !boundary) ? &e->layers[l + 1] : nullptr;
        if (l == 0) rc = encoder_block(e, e->layers[0], x0, Din, x, B, Tp, s, nullptr, false, next);
        else rc = encoder_block(e, e->layers[l], x, D, x, B, Tp, s, nullptr, xn_ready, next);
        xn_ready = next != nullptr && next->in_dim == D && encoder_w2_row_form(e, w2_rows);
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


}  // extern "C"
