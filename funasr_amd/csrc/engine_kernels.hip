// C-ABI layer, single-kernel hooks pf_k_* (kernel tests and micro-benchmarks bind these; not part of the reference boundary).
#include "engine_internal.h"

using namespace pf;

extern "C" {

// -------------------------------------------------------------------------------------------- single kernels
/* test hook: pf_k_gemm_f32 takes the small-M kernel for M <= m (default 0 = always the tile kernel) */
/* the small-M kernel with a LayerNorm carried between two GEMMs (gemm_skinny.hip): stats_out [M][N / 16][2] receives the block
 * partials of the finished outputs; stats_in [M][K / 16][2] (+ gamma, beta, eps) makes the A operand LayerNorm(A) on the fetch */
int pf_k_gemm_skinny_ln(const float* A, int32_t lda, const float* W, int32_t ldw, const float* bias, const float* R2, int32_t ldr2,
                        float* Cout, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t relu, float* stats_out,
                        const float* stats_in, const float* ln_g, const float* ln_c, float ln_eps, const float* out_gamma, float* ws_part,
                        int32_t* ws_count, void* stream) {
    GemmArgs g{};
    g.ws_part = ws_part; g.ws_count = ws_count; g.ln_c1 = ln_c; g.ln_c2 = ln_c ? ln_c + N : nullptr; g.out_gamma = out_gamma;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.R2 = R2; g.ldr2 = ldr2; g.C = Cout; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.relu = relu; g.ln_stats_out = stats_out; g.ln_stats_in = stats_in; g.ln_g = ln_g; g.ln_eps = ln_eps;
    PF_REQUIRE(gemm_skinny_applicable(g), "gemm_skinny_ln: K % 16");
    return launch_gemm_skinny(g, reinterpret_cast<hipStream_t>(stream));
}
/* c[0 .. N) = W gamma, c[N .. 2 N) = W beta + bias: the constants of pf_k_gemm_skinny_ln's LayerNorm form (rowwise.hip) */
int pf_k_ln_consts(const float* W, int32_t ldw, int32_t N, int32_t K, const float* gamma, const float* beta, const float* bias, float* c,
                   void* stream) {
    return launch_ln_consts(W, ldw, N, K, gamma, beta, bias, c, c + N, reinterpret_cast<hipStream_t>(stream));
}
int pf_set_skinny_max_m(int32_t m) { g_skinny_max_m = m; return 0; }

int pf_k_gemm_f32(const float* A, int32_t lda, const float* W, int32_t ldw, const float* bias, const float* R1,
                  int32_t ldr1, const float* R2, int32_t ldr2, float* C, int32_t ldc, int32_t M, int32_t N, int32_t K,
                  int32_t relu, void* stream) {
    return gemm_simple(A, lda, W, ldw, bias, C, ldc, M, N, K, relu, R1, ldr1, R2, ldr2,
                       reinterpret_cast<hipStream_t>(stream));
}
/* Conv1d over time as ONE exact-fp32 GEMM with the im2col gathered by the operand loads (gemm_f32.hip, GemmArgs.conv_*; the
 * predictor's cif_conv1d, cif_predictor.py:275-278): C [B T, N] = relu?(col(hidden) W^T + bias), hidden [B, T, D], W [N, taps D]
 * with column tap * D + c = torch's weight[n, c, tap]; zero_dev: >= 128 B of zeros. Bitwise pf_k_gemm_f32 on the materialised
 * column matrix. */
int pf_k_conv1d_gemm_f32(const float* hidden, const float* W, const float* bias, float* C, int32_t B, int32_t T, int32_t D,
                         int32_t N, int32_t taps, int32_t left, int32_t relu, const float* zero_dev, void* stream) {
    GemmArgs g{};
    g.A = hidden; g.lda = D; g.W = W; g.ldw = taps * D; g.bias = bias; g.C = C; g.ldc = N; g.M = B * T; g.N = N; g.K = taps * D;
    g.relu = relu; g.conv_taps = taps; g.conv_D = D; g.conv_T = T; g.conv_left = left; g.conv_zero = zero_dev;
    return launch_gemm_f32(g, reinterpret_cast<hipStream_t>(stream));
}
/* 1: this library was built with the measured-and-off kernel shapes and ablation builds (make measure); 0: the product */
int pf_measurement_build(void) {
#if defined(PF_MEASUREMENT_KERNELS)
    return 1;
#else
    return 0;
#endif
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
    g.a_nt = a_nt & 3; g.block_rows = a_nt >> 8;          // bit 1: k-step order (A/B); bits 8..: GemmRowArgs.block_rows (0 by row count, 96, 128, 129, 130)
    if (iters <= 0 || !ms_out) return launch_gemm_f16x2_row(g, s);
    return time_launches([&] { return launch_gemm_f16x2_row(g, s); }, iters, ms_out, s);
}
/* the encoder block's feed-forward in one launch (gemm_f16x2_ffn.hip): C = R + (relu(X W1^T + b1) W2^T + b2) [+ LayerNorm -> planes Y2
 * or fp32 Yf]; operands are two-plane fp16 tensors ([2][M, 512], [2][F, 512], [2][512, F]), plane strides M 512 / F 512 / 512 F */
#if defined(PF_MEASUREMENT_KERNELS)
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
#else
int pf_k_ffn_f16x2(const void* X2, const void* W1, const void* W2, const float* b1, const float* b2, float oscale1, float hscale,
                   float oscale2, const float* R, float* Cout, const float* ln_g, const float* ln_b, float ln_eps, void* Y2,
                   float yscale, float* Yf, int32_t M, int32_t F, int32_t iters, float* ms_out, void* stream) {
    set_error("pf_k_ffn_f16x2: the one-launch feed-forward is in the measurement library only (make -C funasr_amd/csrc measure)");
    return -1;
}
#endif
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
    g.a_nt = a_nt & 3; g.block_rows = a_nt >> 8;
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
