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

namespace pf {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int RW_BM = 128, RW_BN = 512, RW_KS = 32, RW_ROWB = 64;
constexpr int RW_A_PLANE_B = RW_BM * RW_ROWB;                       // 8 KB
constexpr int RW_B_PLANE_B = RW_BN * RW_ROWB;                       // 32 KB
constexpr int RW_STAGE_B = 2 * (RW_A_PLANE_B + RW_B_PLANE_B);       // 80 KB
constexpr int RW_PPW = RW_STAGE_B / 1024 / 8;                       // 10 pieces per wave and stage
constexpr int RW_ELD = 132;                                         // slab row (floats): 128 columns + 4
constexpr int RW_SLAB_B = 8 * 32 * RW_ELD * 4;                      // 135168
constexpr int RW_LDS_B = 2 * RW_STAGE_B;                            // 163840 = the CU's whole LDS
constexpr int RW_P_FLOATS = RW_BM * 4 * 32;                         // statistics exchange [row][wave column][lane]: 64 KB
constexpr int RW_FS_KS = 11, RW_FS_LP = 5;                          // FSMN taps / left padding (the offline encoder's kernel 11)
constexpr int RW_FSW_OFF_B = RW_SLAB_B;                             // FSMN taps [11][512] floats behind the slabs: 22 KB
static_assert(RW_SLAB_B <= RW_LDS_B && (RW_P_FLOATS + 2 * RW_BM) * 4 <= RW_LDS_B, "epilogue LDS");
static_assert(RW_FSW_OFF_B + RW_FS_KS * RW_BN * 4 <= RW_LDS_B, "FSMN taps do not fit behind the slabs");

// MODE bit 0: R1 addend, bit 1: R2 addend (v = v + R1, then v = R2 + v, like gemm_f16x2_kernel), bit 2: the first addend is
// the FSMN memory block of p.fs_v computed in place (excludes bit 0); LN: LayerNorm epilogue
template <int MODE, bool LN, bool A_NT>
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

    const int nk = p.K / RW_KS;
#pragma unroll
    for (int i = 0; i < RW_PPW; ++i) piece(i, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        glds_wait_all();
        __syncthreads();
        const bool nxt = kt + 1 < nk;
        const int nb = (kt + 1) & 1;
        const unsigned char* sb = smem + (kt & 1) * RW_STAGE_B;
#define RW_PIECE(I) do { if (nxt) piece(I, nb, kt + 1); } while (0)
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
        RW_LOAD(a0, b0, 0)
        RW_LOAD(a1, b1, 1)
        RW_PIECE(0); RW_PIECE(1); RW_PIECE(2); RW_PIECE(3);
        RW_PROD(a0, b0, 1, 0); RW_PIECE(4); RW_PIECE(5);
        RW_PROD(a0, b0, 0, 1); RW_PIECE(6); RW_PIECE(7);
        RW_PROD(a0, b0, 0, 0); RW_PIECE(8); RW_PIECE(9);
        RW_PROD(a1, b1, 1, 0);
        RW_PROD(a1, b1, 0, 1);
        RW_PROD(a1, b1, 0, 0);
#undef RW_PROD
#undef RW_LOAD
#undef RW_PIECE
    }

    // ---- epilogue, part 1: accumulators (C/D layout of the 32 x 32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) +
    //      4 (lane >> 5)) -> wave-private slab -> float4 pieces of rows: half-wave h takes rows 16 h .. 16 h + 15 of the
    //      32-row tile, lane c4 its columns 4 c4 .. 4 c4 + 3 of the wave's 128; the finished values stay in registers
    constexpr bool HAS_R1 = (MODE & 1) != 0, HAS_R2 = (MODE & 2) != 0, FSMN = (MODE & 4) != 0;
    static_assert(!(HAS_R1 && FSMN), "the FSMN form computes the first addend");
    __syncthreads();
    float* smf = reinterpret_cast<float*>(smem);
    float* slab = smf + wave * (32 * RW_ELD);
    const float* fsw = reinterpret_cast<const float*>(smem + RW_FSW_OFF_B);      // [tap][512]
    if constexpr (FSMN) {
        // taps [512][11] -> LDS [11][512]: thread = channel (consecutive threads, consecutive LDS words)
        float* dst = reinterpret_cast<float*>(smem + RW_FSW_OFF_B);
#pragma unroll
        for (int j = 0; j < RW_FS_KS; ++j) dst[j * RW_BN + tid] = p.fs_w[(size_t)tid * RW_FS_KS + j];
        __syncthreads();
    }
    const int c4 = idx, rsub = hh;
    const int col = wc * 128 + c4 * 4;
    const float oscale = p.oscale_dev ? p.oscale * *p.oscale_dev : p.oscale;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + col);
    float4 ov[WM][16];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
#pragma unroll
        for (int jj = 0; jj < WN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[((r & 3) + 8 * (r >> 2) + 4 * hh) * RW_ELD + jj * 32 + idx] = acc[i][jj][r];
        const int row0 = m0 + wr * 64 + i * 32 + rsub * 16;
        if constexpr (FSMN) {
            // this half-wave's 16 rows are one 16-row group of one sequence: valid v rows [lo, hi) (sequence start .. start +
            // len), everything else -- other sequences, padding rows, rows outside the batch -- counts as zero input, and
            // output rows >= hi get no memory. Four output rows at a time over a sliding window of 14 v rows (26 per tile).
            const int grp = row0 >> 4;
            const bool gok = row0 < p.M;
            const int lo = gok ? p.fs_lo[grp] : 0, hi = gok ? p.fs_hi[grp] : 0;
            float4 win[16 + RW_FS_KS - 1];
            auto load_row = [&](int k) {
                const int vr = row0 - RW_FS_LP + k;
                const bool ok = vr >= lo && vr < hi;
                const float4 t = *reinterpret_cast<const float4*>(p.fs_v + (size_t)(ok ? vr : (lo < hi ? lo : 0)) * p.ldfv + col);
                win[k] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            };
#pragma unroll
            for (int k = 0; k < RW_FS_KS - 1; ++k) load_row(k);
#pragma unroll
            for (int h4 = 0; h4 < 4; ++h4) {
                float4 r2[4], fa[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    load_row(RW_FS_KS - 1 + h4 * 4 + t);
                    fa[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (HAS_R2) {
                        const int row = row0 + h4 * 4 + t;
                        r2[t] = *reinterpret_cast<const float4*>(p.R2 + (size_t)(row < p.M ? row : p.M - 1) * p.ldr2 + col);
                    }
                }
#pragma unroll
                for (int j = 0; j < RW_FS_KS; ++j) {
                    const float4 wj = *reinterpret_cast<const float4*>(fsw + j * RW_BN + col);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float4 x = win[h4 * 4 + t + j];
                        fa[t].x = fmaf(wj.x, x.x, fa[t].x); fa[t].y = fmaf(wj.y, x.y, fa[t].y);
                        fa[t].z = fmaf(wj.z, x.z, fa[t].z); fa[t].w = fmaf(wj.w, x.w, fa[t].w);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = row0 + h4 * 4 + t;
                    const float4 c = win[h4 * 4 + t + RW_FS_LP];                  // the (masked) input row itself
                    float4 mem = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (row < hi) mem = make_float4(fa[t].x + c.x, fa[t].y + c.y, fa[t].z + c.z, fa[t].w + c.w);
                    const float4 vt = *reinterpret_cast<const float4*>(slab + (rsub * 16 + h4 * 4 + t) * RW_ELD + c4 * 4);
                    float o[4] = {vt.x * oscale + bias4.x, vt.y * oscale + bias4.y, vt.z * oscale + bias4.z,
                                  vt.w * oscale + bias4.w};
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
                    }
                    o[0] = o[0] + mem.x; o[1] = o[1] + mem.y; o[2] = o[2] + mem.z; o[3] = o[3] + mem.w;
                    if constexpr (HAS_R2) { o[0] = r2[t].x + o[0]; o[1] = r2[t].y + o[1]; o[2] = r2[t].z + o[2]; o[3] = r2[t].w + o[3]; }
                    const float4 o4 = make_float4(o[0], o[1], o[2], o[3]);
                    ov[i][h4 * 4 + t] = o4;
                    if (p.C && row < p.M) *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + col) = o4;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            continue;
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            float4 v[8], r1[8], r2[8];
#pragma unroll
            for (int t = 0; t < 8; ++t)
                v[t] = *reinterpret_cast<const float4*>(slab + (rsub * 16 + h2 * 8 + t) * RW_ELD + c4 * 4);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int row = row0 + h2 * 8 + t;
                const int rr = row < p.M ? row : p.M - 1;
                if constexpr (HAS_R1) r1[t] = *reinterpret_cast<const float4*>(p.R1 + (size_t)rr * p.ldr1 + col);
                if constexpr (HAS_R2) r2[t] = *reinterpret_cast<const float4*>(p.R2 + (size_t)rr * p.ldr2 + col);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int row = row0 + h2 * 8 + t;
                float o[4] = {v[t].x * oscale + bias4.x, v[t].y * oscale + bias4.y, v[t].z * oscale + bias4.z,
                              v[t].w * oscale + bias4.w};
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
                }
                if constexpr (HAS_R1) { o[0] = o[0] + r1[t].x; o[1] = o[1] + r1[t].y; o[2] = o[2] + r1[t].z; o[3] = o[3] + r1[t].w; }
                if constexpr (HAS_R2) { o[0] = r2[t].x + o[0]; o[1] = r2[t].y + o[1]; o[2] = r2[t].z + o[2]; o[3] = r2[t].w + o[3]; }
                const float4 o4 = make_float4(o[0], o[1], o[2], o[3]);
                ov[i][h2 * 8 + t] = o4;
                if (p.C && row < p.M) *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + col) = o4;
            }
            // keep the eight-row groups apart: without the fence the scheduler hoists every slab read of the tile above the
            // arithmetic and spills (209 VGPRs in the form without addends)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (!LN) return;

    // ---- part 2: LayerNorm over the 512 columns of every row. The stand-alone kernel gives lane l the chunks l and l + 64
    //      of a row, adds the two, then runs the 64-lane xor butterfly: chunk index = 32 wc + c4 here, so the first two
    //      levels are (wc 0 + wc 2) + (wc 1 + wc 3) per c4 and the rest a butterfly over c4. Per-chunk partials go through
    //      LDS ([row][wc][c4]); wave (wr, wc) reduces rows 16 wc .. 16 wc + 15 of its row half.
    float* P = smf;                                   // aliases the slabs: every wave is past its slab reads
    float* ST = smf + RW_P_FLOATS;                    // mean[128], rstd[128]
    const int lrow0 = wr * 64 + rsub * 16;            // + 32 i + it: this lane's rows inside the tile
    auto reduce_rows = [&](bool second) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int rl = wr * 64 + wc * 16 + rsub * 8 + j;
            const float* pr = P + (size_t)rl * 128 + c4;
            float v = (pr[0] + pr[64]) + (pr[32] + pr[96]);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            const float res = second ? ln_rstd(v, RW_BN, p.ln_eps) : ln_mean(v, RW_BN);
            if (c4 == 0) ST[(second ? RW_BM : 0) + rl] = res;
        }
    };
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int it = 0; it < 16; ++it)
            P[(size_t)(lrow0 + 32 * i + it) * 128 + wc * 32 + c4] = ln_sum4(ov[i][it]);
    __syncthreads();
    reduce_rows(false);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const float mean = ST[lrow0 + 32 * i + it];
            P[(size_t)(lrow0 + 32 * i + it) * 128 + wc * 32 + c4] = ln_sqdev4(ov[i][it], mean);
        }
    __syncthreads();
    reduce_rows(true);
    __syncthreads();
    const float4 g4 = *reinterpret_cast<const float4*>(p.ln_g + col);
    const float4 b4 = *reinterpret_cast<const float4*>(p.ln_b + col);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int rl = lrow0 + 32 * i + it;
            const int row = m0 + rl;
            if (row >= p.M) continue;
            const float4 y = ln_apply4(ov[i][it], ST[rl], ST[RW_BM + rl], g4, b4);
            if (p.Y2) {
                const float yv[4] = {y.x, y.y, y.z, y.w};
                store_split2x4_pair(p.Y2 + (size_t)row * p.ldy2 + col, p.y_plane, yv, p.yscale, lane);
            } else {
                *reinterpret_cast<float4*>(p.Yf + (size_t)row * p.ldyf + col) = y;
            }
        }
}

template <int MODE, bool LN, bool A_NT>
int launch_row_t(const GemmRowArgs& a, hipStream_t stream) {
    static bool configured = false;
    if (!configured) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x2_row_kernel<MODE, LN, A_NT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, RW_LDS_B));
        configured = true;
    }
    hipLaunchKernelGGL((gemm_f16x2_row_kernel<MODE, LN, A_NT>), dim3((unsigned)ceil_div(a.M, RW_BM)), dim3(512), RW_LDS_B, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}
template <int MODE, bool LN>
int launch_row_m(const GemmRowArgs& a, hipStream_t stream) {
    return a.a_nt ? launch_row_t<MODE, LN, true>(a, stream) : launch_row_t<MODE, LN, false>(a, stream);
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
    int bm = a.block_rows;
    if (bm == 0) {
        static const int n_cu = [] {
            int dev = 0, n = 256;
            hipDeviceProp_t pr;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
                n = pr.multiProcessorCount;
            return n;
        }();
        const double cost128 = (double)ceil_div(ceil_div(a.M, 128), n_cu), cost96 = 0.80 * (double)ceil_div(ceil_div(a.M, 96), n_cu);
        bm = cost96 < cost128 - 0.02 ? 96 : 128;
    }
    if (bm != 128) {
        PF_REQUIRE(bm == 96 || bm == 129, "gemm_f16x2_row: block_rows is 0, 96, 128 or 129");
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
