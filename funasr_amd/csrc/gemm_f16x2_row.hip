// Full-row form of gemm_f16x2.hip for the N = 512 projections of a SAN-M block (self_attn.linear_out, feed_forward.w_2):
// one workgroup owns 128 COMPLETE output rows (128 x 512 tile), so what follows the projection in the reference --
//     x = residual + (dropout(linear_out(ctx)) + fsmn_memory)       funasr/models/sanm/encoder.py:120-137
//     x = residual + feed_forward(norm2(x));  next block: norm1(x)   funasr/models/sanm/encoder.py:141-146, :96-98
//     LayerNorm                                                       funasr/models/transformer/layer_norm.py:13-38
// -- runs in the epilogue: bias, the two addends, the fp32 residual stream written ONCE, then LayerNorm over the row and the
// two fp16 planes the next GEMM consumes. That deletes the stand-alone LayerNorm launch and its re-read of the residual
// stream (67 MB per launch at B T = 32768), and every A panel is fetched by exactly one workgroup.
//
// FSMN form (linear_out only; MODE bit 2): the first addend is not read from memory but COMPUTED here -- the FSMN memory block
//     fsmn_memory = mask (conv_k11(pad(mask v)) + mask v)             funasr/models/sanm/attention.py:216-239
// over the fp32 v projection (11 taps along time, per channel), with fsmn_kernel's fma chain (rowwise.hip): the separate
// launch, its output tensor and its re-read here (187 + 67 MB per block at B T = 32768) are gone; a workgroup reads the v rows
// of its 128 output rows plus a 5-row halo on each side.
//
// Arithmetic: the products, their k order and the epilogue's operation order are those of gemm_f16x2_kernel; the row
// statistics are summed in layernorm_kernel's order (common.h ln_*: 4-column chunks, chunk l + chunk l + 64, xor butterfly
// 32, 16 .. 1 over l). The fused result is therefore BITWISE the result of gemm_f16x2 followed by layernorm_kernel (tested).
//
// Design (gfx950): 8 waves as 2 (M) x 4 (N), a wave owns 64 x 128 = 2 x 4 MFMA tiles of 32 x 32 (the per-wave shape of the
// 256 x 256 kernel: 12 ds_read_b128 + 24 v_mfma_f32_32x32x16_f16 per 16-deep step); 32-deep K stages of 64-B LDS rows --
// 2 planes x (128 + 512) rows = 80 KB per stage, double buffered = all 160 KB of LDS, one workgroup per CU -- moved
// HBM / L2 -> LDS by asm-issued global_load_lds_dwordx4 pieces (10 per wave and stage), chunk swizzle c ^ ((row >> 2) & 3) on
// the DMA source address and on the read. Epilogue: accumulators -> wave-private LDS slab -> row-major float4 pieces kept in
// registers (64 x 128 per wave = 128 VGPRs), statistics exchanged through LDS in three rounds, 16-B plane stores.
#include "common.h"
#include "gemm_f16x2_row_epilogue.h"

namespace pf {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));


// MODE bit 0: R1 addend, bit 1: R2 addend (v = v + R1, then v = R2 + v, like gemm_f16x2_kernel), bit 2: the first addend is
// the FSMN memory block of p.fs_v computed in place (excludes bit 0); LN: LayerNorm epilogue
template <int MODE, bool LN, bool A_NT, int SCHED = 2>
__global__ __launch_bounds__(512, 2) void gemm_f16x2_row_kernel(GemmRowArgs p) {
    constexpr int WM = 2, WN = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int m0 = blockIdx.x * RW_BM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int hh = lane >> 5, idx = lane & 31;

    // ---- DMA sources. A stage is 80 pieces of 1 KB (16 rows of one plane), linear in LDS: A plane 0 (8 pieces), A plane 1,
    //      W plane 0 (32 pieces), W plane 1; wave w issues pieces w, w + 8, ..: its pieces 0 / 1 are the A planes (rows
    //      16 w ..), 2..5 / 6..9 the W planes (columns 16 (w + 8 j) ..). Lane l lands at row l / 4, physical chunk l % 4 and
    //      fetches the logical chunk the read-side swizzle expects there.
    const unsigned short* src[RW_PPW];
    {
        const int prow = lane >> 2;
        const int chunk = (lane & 3) ^ ((prow >> 2) & 3);
        int row = m0 + wave * 16 + prow;
        row = row < p.M ? row : p.M - 1;
        src[0] = p.A + (size_t)row * p.lda + chunk * 8;
        src[1] = src[0] + p.a_plane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = (wave + 8 * j) * 16 + prow;
            src[2 + j] = p.W + (size_t)col * p.ldw + chunk * 8;
            src[6 + j] = src[2 + j] + p.w_plane;
        }
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (unsigned)wave * 1024);
    auto piece = [&](int i, int buf, int kt) {
        const unsigned dst = lds0 + (unsigned)buf * RW_STAGE_B + (unsigned)i * 8192;
        if (A_NT && i < 2) glds16_nt(src[i] + kt * RW_KS, dst);
        else glds16(src[i] + kt * RW_KS, dst);
    };

    floatx16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int jj = 0; jj < WN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    const int f = (idx >> 2) & 3;
    const int aoff = (wr * 64 + idx) * RW_ROWB;
    const int boff = 2 * RW_A_PLANE_B + (wc * 128 + idx) * RW_ROWB;
    int coff[2];
#pragma unroll
    for (int st = 0; st < 2; ++st) coff[st] = ((2 * st + hh) ^ f) * 16;

#if defined(PF_MEASUREMENT_KERNELS)
    const int nk = (p.a_nt & 128) ? 0 : p.K / RW_KS;                      // bit 7: no K loop (the epilogue alone)
#else
    const int nk = p.K / RW_KS;
#endif
    // MEASURED AND OFF (round 6, review item 3; profiles/r06aa_row_prefetch_ab.txt: 113 -> 121 (either half) / 128 us (both)):
    // epilogue operands pulled towards the chip DURING the K loop (a_nt bits 2 / 3; FSMN form): the loop reads 1 MB of A / W per stage
    // from L2 and leaves HBM idle, the epilogue then asks for 544 KB of fp32 rows per workgroup -- the residual rows and the v rows
    // with their halo -- all at once, on every CU at the same moment. One dword load per cache line (128 B), spread over the first
    // nk - 1 stages, result unused: the line lands in L2 / the 256-MB memory-side cache and the epilogue's float4 loads find it
    // there. Pure data movement: the bits of the result cannot change. (`pf_sink` stays live until the wait behind the load.)
    constexpr int PF_R2_LINES = RW_BM * 16, PF_V_LINES = (RW_BM + RW_FS_KS - 1) * 16;      // 2-KB rows = 16 lines
#if defined(PF_MEASUREMENT_KERNELS)
    const bool pf_on = (MODE & 4) && (p.a_nt & 12) && nk > 1;            // bit 2: the residual rows, bit 3: the v rows
#else
    constexpr bool pf_on = false;                                        // measured and off (profiles/r06aa): the product has no such loads
#endif
    const int pf_first = (p.a_nt & 4) ? 0 : PF_R2_LINES, pf_last = (p.a_nt & 8) ? PF_R2_LINES + PF_V_LINES : PF_R2_LINES;
    const int pf_per_stage = pf_on ? (pf_last - pf_first + nk - 2) / (nk - 1) : 0;
    unsigned pf_sink = 0;
    if (nk > 0) {
#pragma unroll
        for (int i = 0; i < RW_PPW; ++i) piece(i, 0, 0);
    }
    for (int kt = 0; kt < nk; ++kt) {
        glds_wait_all();
        asm volatile("" : : "v"(pf_sink));
        __syncthreads();
        if (pf_on && kt + 1 < nk) {
            for (int l = tid; l < pf_per_stage; l += 512) {
                const int idx = pf_first + kt * pf_per_stage + l;
                const float* src_line = nullptr;
                if (idx >= pf_last) {
                } else if (idx < PF_R2_LINES) {
                    int row = m0 + (idx >> 4);
                    row = row < p.M ? row : p.M - 1;
                    src_line = p.R2 + (size_t)row * p.ldr2 + (idx & 15) * 32;
                } else if (idx < PF_R2_LINES + PF_V_LINES) {
                    const int j = idx - PF_R2_LINES;
                    int row = m0 - RW_FS_LP + (j >> 4);
                    row = row < 0 ? 0 : (row < p.M ? row : p.M - 1);
                    src_line = p.fs_v + (size_t)row * p.ldfv + (j & 15) * 32;
                }
                if (src_line) asm volatile("global_load_dword %0, %1, off" : "=v"(pf_sink) : "v"(src_line) : "memory");
            }
        }
        const bool nxt = kt + 1 < nk;
        const int nb = (kt + 1) & 1;
        const unsigned char* sb = smem + (kt & 1) * RW_STAGE_B;
#if defined(PF_MEASUREMENT_KERNELS)
        const bool early = (p.a_nt & 512) != 0;            // bit 9: the next stage's pieces all at the top of the stage (A/B, r06ac)
        if (early && nxt) {
#pragma unroll
            for (int i = 0; i < RW_PPW; ++i) piece(i, nb, kt + 1);
        }
#else
        constexpr bool early = false;
#endif
#define RW_PIECE(I) do { if (nxt && !early) piece(I, nb, kt + 1); } while (0)
#define RW_PROD(AF, BF, PA, PB)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < WM; ++i) _Pragma("unroll") for (int jj = 0; jj < WN; ++jj)                  \
        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AF[i][PA], BF[jj][PB], acc[i][jj], 0, 0, 0)
#define RW_LOAD(AF, BF, S)                                                                                            \
    _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                                                                \
        _Pragma("unroll") for (int i = 0; i < WM; ++i)                                                                \
            AF[i][pl] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(sb + pl * RW_A_PLANE_B + aoff + i * 32 * RW_ROWB + coff[S]));   \
        _Pragma("unroll") for (int jj = 0; jj < WN; ++jj)                                                             \
            BF[jj][pl] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(sb + pl * RW_B_PLANE_B + boff + jj * 32 * RW_ROWB + coff[S])); \
    }
        // both k-steps' fragments are requested before the first MFMA (one LDS-latency bubble per stage), the next stage's
        // DMA pieces early; per k-step the two small products first, hi * hi last: the order of gemm_f16x2_kernel
        f16x8 a0[WM][2], b0[WN][2], a1[WM][2], b1[WN][2];
        if constexpr (SCHED == 2) {
            RW_LOAD(a0, b0, 0)
            RW_LOAD(a1, b1, 1)
            RW_PIECE(0); RW_PIECE(1); RW_PIECE(2); RW_PIECE(3);
            RW_PROD(a0, b0, 1, 0); RW_PIECE(4); RW_PIECE(5);
            RW_PROD(a0, b0, 0, 1); RW_PIECE(6); RW_PIECE(7);
            RW_PROD(a0, b0, 0, 0); RW_PIECE(8); RW_PIECE(9);
            RW_PROD(a1, b1, 1, 0);
            RW_PROD(a1, b1, 0, 1);
            RW_PROD(a1, b1, 0, 0);
        } else {
            // the plain order (round 5: what the wide tile GEMM now runs, profiles/r05s): a k-step's fragments, its products with
            // the next stage's pieces between them, then the next k-step's fragments
            RW_LOAD(a0, b0, 0)
            RW_PROD(a0, b0, 1, 0); RW_PIECE(0); RW_PIECE(1); RW_PIECE(2);
            RW_PROD(a0, b0, 0, 1); RW_PIECE(3); RW_PIECE(4); RW_PIECE(5);
            RW_PROD(a0, b0, 0, 0); RW_PIECE(6); RW_PIECE(7);
            RW_LOAD(a1, b1, 1)
            RW_PROD(a1, b1, 1, 0); RW_PIECE(8); RW_PIECE(9);
            RW_PROD(a1, b1, 0, 1);
            RW_PROD(a1, b1, 0, 0);
        }
#undef RW_PROD
#undef RW_LOAD
#undef RW_PIECE
    }
#if defined(PF_MEASUREMENT_KERNELS)
    if (p.a_nt & 256) {                                    // bit 8: the K loop alone (one store so that the accumulators stay live)
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int jj = 0; jj < WN; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][jj][r];
        if (sacc == 12345.678f && p.C) p.C[tid] = sacc;
        return;
    }
#endif
    if (pf_on) {                                           // (nothing of the asm-issued loads is in flight when compiler-counted loads start)
        glds_wait_all();
        asm volatile("" : : "v"(pf_sink));
    }

    gemm2_row_epilogue<MODE, LN, 512, WM, 0>(p, acc, smem, m0, tid, wave, wr, wc, lane, true);
}

template <int MODE, bool LN, bool A_NT, int SCHED = 2>
int launch_row_t(const GemmRowArgs& a, hipStream_t stream) {
    static PerDeviceOnce configured;
    if (!configured.done()) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_row_kernel<MODE, LN, A_NT, SCHED>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, RW_LDS_B));
        configured.mark();
    }
    hipLaunchKernelGGL((gemm_f16x2_row_kernel<MODE, LN, A_NT, SCHED>), dim3((unsigned)ceil_div(a.M, RW_BM)), dim3(512), RW_LDS_B, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}
template <int MODE, bool LN>
int launch_row_m(const GemmRowArgs& a, hipStream_t stream) {
    // a_nt bit 0: non-temporal A loads; bit 1: the plain k-step order (the product's since round 5, profiles/r05t); without it both
    // k-steps' fragments up front -- the measurement library only
#if defined(PF_MEASUREMENT_KERNELS)
    if (!(a.a_nt & 2)) return (a.a_nt & 1) ? launch_row_t<MODE, LN, true>(a, stream) : launch_row_t<MODE, LN, false>(a, stream);
#endif
    return (a.a_nt & 1) ? launch_row_t<MODE, LN, true, 0>(a, stream) : launch_row_t<MODE, LN, false, 0>(a, stream);
}

}  // namespace

bool gemm_f16x2_row_applicable(int N, int K) { return N == RW_BN && K > 0 && K % RW_KS == 0; }

int launch_gemm_f16x2_row(const GemmRowArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.M > 0 && a.N == RW_BN && a.K > 0 && a.K % RW_KS == 0, "gemm_f16x2_row: N must be 512 and K a multiple of 32");
    PF_REQUIRE(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.a_plane % 8 == 0 && a.w_plane % 8 == 0, "gemm_f16x2_row: operand strides % 8");
    PF_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm_f16x2_row: operands must be 16-B aligned");
    if (a.C) PF_REQUIRE(a.ldc % 4 == 0 && ((uintptr_t)a.C & 15) == 0, "gemm_f16x2_row: C alignment");
    if (a.bias) PF_REQUIRE(((uintptr_t)a.bias & 15) == 0, "gemm_f16x2_row: bias alignment");
    if (a.R1) PF_REQUIRE(a.ldr1 % 4 == 0 && ((uintptr_t)a.R1 & 15) == 0, "gemm_f16x2_row: R1 alignment");
    if (a.R2) PF_REQUIRE(a.ldr2 % 4 == 0 && ((uintptr_t)a.R2 & 15) == 0, "gemm_f16x2_row: R2 alignment");
    const bool ln = a.ln_g != nullptr;
    if (ln) {
        PF_REQUIRE(a.ln_b && ((uintptr_t)a.ln_g & 15) == 0 && ((uintptr_t)a.ln_b & 15) == 0, "gemm_f16x2_row: LayerNorm parameters");
        PF_REQUIRE((a.Y2 != nullptr) != (a.Yf != nullptr), "gemm_f16x2_row: the LayerNorm form writes planes (Y2) or fp32 (Yf)");
        if (a.Y2) PF_REQUIRE(a.ldy2 % 8 == 0 && a.y_plane % 8 == 0 && ((uintptr_t)a.Y2 & 15) == 0, "gemm_f16x2_row: plane output alignment");
        else PF_REQUIRE(a.ldyf % 4 == 0 && ((uintptr_t)a.Yf & 15) == 0, "gemm_f16x2_row: fp32 LayerNorm output alignment");
    } else {
        PF_REQUIRE(a.C, "gemm_f16x2_row: nothing to write");
    }
    if (a.fs_v)
        PF_REQUIRE(!a.R1 && a.fs_w && a.fs_lo && a.fs_hi && a.ldfv % 4 == 0 && ((uintptr_t)a.fs_v & 15) == 0 && a.M % 16 == 0 && ln,
                   "gemm_f16x2_row: the FSMN form needs taps, the 16-row group bounds, M % 16 == 0, the LayerNorm epilogue and no R1");
    // block height by the time the block count costs in WHOLE rounds over the CUs (one block per CU): 22 528 rows are 176
    // blocks of 128 rows = one round with 80 CUs idle, or 235 blocks of 96 rows = one round of 0.78 the time
    // (tools/bench_r03.py `row8`). Both kernels give the same bits, so the choice may depend on the batch's row count.
#if defined(PF_MEASUREMENT_KERNELS)
    static const int env_prefetch = [] { const char* e = getenv("PF_ROW_PREFETCH"); return e ? atoi(e) : -1; }();
    // A/B switches: PF_ROW_PREFETCH 1 both / 2 residual / 3 v rows; + 16 / 32 / 64 / 128: the epilogue's ablation bits as they are
    const int env_bits = (env_prefetch & ~3) | ((env_prefetch & 3) == 1 ? 12 : (env_prefetch & 3) == 2 ? 4 : (env_prefetch & 3) == 3 ? 8 : 0);
    if (env_prefetch >= 0 && a.fs_v && a.R2 && (a.a_nt & ~3) != env_bits) {
        GemmRowArgs b = a;
        b.a_nt = (a.a_nt & 3) | env_bits;
        return launch_gemm_f16x2_row(b, stream);
    }
#endif
    int bm = a.block_rows;
    if (bm == 0) {
        const int n_cu = device_cu_count();
        const double cost128 = (double)ceil_div(ceil_div(a.M, 128), n_cu), cost96 = 0.80 * (double)ceil_div(ceil_div(a.M, 96), n_cu);
        bm = cost96 < cost128 - 0.02 ? 96 : 128;
    }
#if defined(PF_MEASUREMENT_KERNELS)
    if (bm == 130) return launch_gemm_f16x2_w4_row(a, stream);       // the four-wave 1 x 4 grid (gemm_f16x2_w4.hip): measured and off
    if (bm != 128) {
        PF_REQUIRE(bm == 96 || bm == 129, "gemm_f16x2_row: block_rows is 0, 96, 128, 129 or 130");
#else
    if (bm != 128) {
        PF_REQUIRE(bm == 96, "gemm_f16x2_row: block_rows is 0, 96 or 128 (129 / 130: the measurement library, make measure)");
#endif
        const int rc = launch_gemm_f16x2_row8(a, bm == 96 ? 96 : 128, stream);
        if (rc != -3) return rc;                      // -3: form not built in that height
    }
    if (a.fs_v) return a.R2 ? launch_row_m<6, true>(a, stream) : launch_row_m<4, true>(a, stream);
    const int mode = (a.R1 ? 1 : 0) | (a.R2 ? 2 : 0);
    switch (mode * 2 + (ln ? 1 : 0)) {
        case 0: return launch_row_m<0, false>(a, stream);
        case 1: return launch_row_m<0, true>(a, stream);
        case 2: return launch_row_m<1, false>(a, stream);
        case 3: return launch_row_m<1, true>(a, stream);
        case 4: return launch_row_m<2, false>(a, stream);
        case 5: return launch_row_m<2, true>(a, stream);
        case 6: return launch_row_m<3, false>(a, stream);
        default: return launch_row_m<3, true>(a, stream);
    }
}

}  // namespace pf
