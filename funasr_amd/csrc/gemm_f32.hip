// Exact-fp32 GEMM on the CDNA4 matrix cores: C = epilogue(A[M,K] * W[N,K]^T).
//
// Replaces every dense nn.Linear / Conv1d-as-GEMM on the Paraformer path (reference call sites:
// funasr/models/sanm/attention.py:256,306 linear_q_k_v / linear_out,
// funasr/models/transformer/positionwise_feed_forward.py:32 w_1 / w_2,
// funasr/models/paraformer/cif_predictor.py:277 cif_conv1d (as a 3-tap im2col GEMM),
// funasr/models/paraformer/decoder.py:444 output_layer, funasr/models/ctc/ctc.py:192 ctc_lo).
//
// Design (gfx950). v_mfma_f32_32x32x2_f32 is an exact f32 fma chain at the f32 vector rate (157 TFLOP/s peak),
// which is what the parity bar (encoder activations <= 1e-3, CIF fire indices equal to the fp32 CPU path) needs.
//   * 128 x 128 x 32 block tile, 4 waves in a 2 x 2 grid, each wave owns 2 x 2 MFMA tiles of 32 x 32
//     (64 accumulator registers); 64 MFMAs = 4096 matrix-pipe cycles per wave per K tile against
//     16 ds_read_b128 and 8 LDS-DMA pieces, i.e. the kernel is matrix-pipe bound by construction.
//   * Both operands are K-contiguous in HBM (torch Linear layout) and go HBM -> LDS directly with
//     global_load_lds_dwordx4 (no staging registers), double buffered: the DMA of K tile t+1 is in flight while
//     tile t is multiplied, one workgroup barrier per K tile. 2 x 32 KB of LDS per workgroup -> two workgroups
//     (2 waves per SIMD) per CU, so one workgroup's barrier/epilogue is covered by the other's MFMAs.
//   * LDS image: rows of 32 floats (128 B), 16-B chunk c of row r stored at chunk c ^ ((r >> 1) & 7). The DMA
//     writes lane-linear, so the permutation is applied to the per-lane SOURCE address (each row is still one
//     full 128-B line) and again on the read; every 16-lane ds_read_b128 group then hits 16 distinct slots.
//   * The two half-waves of an MFMA operand take K offsets {0..15} and {16..31} of the tile (the MFMA k index is
//     only a pairing between A and B, so any bijection applied to both is legal): operand fetch = ds_read_b128.
//   * XCD-aware block order: consecutive workgroup ids round-robin over the 8 XCDs, so XCD x gets the row
//     panels x, x+8, ... and walks all column blocks of a panel back to back: the A panel is fetched from HBM
//     once and re-read from that XCD's L2.
//
// The same kernel is instantiated for bf16 OPERANDS (throughput mode, v_mfma_f32_32x32x16_bf16, fp32 accumulate,
// fp32 bias / residual epilogue, fp32 or bf16 output): an LDS row is 128 B either way (32 floats or 64 bf16), so
// staging, swizzle and read addressing are byte-identical; only the MFMA and the K extent of a tile differ.
#include "common.h"

namespace pf {

namespace {

constexpr int BN = 128;
constexpr int ROW_BYTES = 128;                        // one LDS row = one 128-B line of an operand row
constexpr int TILE_FLOATS = 128 * ROW_BYTES / 4;      // a 128-row operand tile (4096 floats = 16 KB)
constexpr int BUF_FLOATS = 2 * TILE_FLOATS;           // A tile + B tile (the A tile of the 64-row variant is half used)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f32_to_bf16_rn(float f) {
    unsigned int u = __builtin_bit_cast(unsigned int, f);
    u += 0x7fffu + ((u >> 16) & 1u);                  // round to nearest even (inputs are finite)
    return (unsigned short)(u >> 16);
}
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int MODE_ARGMAX = 4;   // MODE bit 0: + R1, bit 1: + R2; 4: fused arg-max instead of a C store

// TM = 32-row MFMA tiles per wave in M: 2 -> the 128 x 128 block tile; 1 -> a 64 x 128 block tile, chosen by the
// launcher when the 128-row grid would leave most CUs idle (decoder-sized problems: twice the workgroups, same
// per-element fma chain, so the choice never changes a bit of the result)
// CONV: the A operand is the im2col of a Conv1d over time, gathered by the DMA sources (GemmArgs.conv_*)
template <int MODE, bool BF16, int TM, bool CONV = false>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma_kernel(GemmArgs p, int nM, int nN) {
    constexpr int BM = 64 * TM;
    constexpr int ES = BF16 ? 2 : 4;                  // operand element size
    constexpr int BK = ROW_BYTES / ES;                // k extent of one tile: 32 floats / 64 bf16
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF_FLOATS];   // 64 KB, the only LDS object

    // ---- XCD-aware tile order
    const int L = blockIdx.x;
    const int xcd = L & 7, j = L >> 3;
    const int mblk = (j / nN) * 8 + xcd, nblk = j % nN;
    if (mblk >= nM) return;
    const int m0 = mblk * BM, n0 = nblk * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int hh = lane >> 5, idx = lane & 31;

    // ---- LDS-DMA source addresses: wave w stages rows [32w, 32w+32) of both tiles, 4 pieces of 8 rows each;
    //      lane l of a piece lands at row (l >> 3), physical chunk (l & 7)
    const char* asrc[2 * TM];
    const char* wsrc[4];
    const char* zsrc[2 * TM];   // CONV: this lane's chunk of the zero row
    int at[2 * TM];             // CONV: the row's time index inside its sequence
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave * 32 + i * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int col = n0 + r;
        col = col < p.N ? col : p.N - 1;
        wsrc[i] = reinterpret_cast<const char*>(p.W) + ((size_t)col * p.ldw) * ES + c * 16;
    }
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i) {
        const int r = wave * (16 * TM) + i * 8 + (lane >> 3);           // wave w stages A rows [16 TM w, 16 TM (w+1))
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int row = m0 + r;
        row = row < p.M ? row : p.M - 1;
        asrc[i] = reinterpret_cast<const char*>(p.A) + ((size_t)row * p.lda) * ES + c * 16;
        if constexpr (CONV) {
            at[i] = row % p.conv_T;
            zsrc[i] = reinterpret_cast<const char*>(p.conv_zero) + c * 16;
        }
    }
    const int nk = p.K / BK;

    const unsigned lds_a = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + (unsigned)wave * (16 * TM) * ROW_BYTES);
    const unsigned lds_b = __builtin_amdgcn_readfirstlane(lds_addr_of(smem) + TILE_FLOATS * 4 + (unsigned)wave * 32 * ROW_BYTES);
    auto stage = [&](int buf, int kt) {
        // issued from inline asm (common.h: glds16) so that the prefetch really stays in flight under the MFMAs of
        // the current tile
        const unsigned off = (unsigned)buf * (BUF_FLOATS * 4);
        if constexpr (CONV) {
            // K tile kt lies inside tap `tap`: the rows `shift` frames away, columns kin .. kin + 31 (zeros outside the sequence)
            const int k0 = kt * BK, tap = k0 / p.conv_D, kin = k0 - tap * p.conv_D, shift = tap - p.conv_left;
            const ptrdiff_t delta = ((ptrdiff_t)shift * p.lda + kin) * ES;
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i) {
                const int ts = at[i] + shift;
                glds16((ts >= 0 && ts < p.conv_T) ? asrc[i] + delta : zsrc[i], lds_a + off + i * 8 * ROW_BYTES);
            }
        } else {
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i) glds16(asrc[i] + kt * ROW_BYTES, lds_a + off + i * 8 * ROW_BYTES);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(wsrc[i] + kt * ROW_BYTES, lds_b + off + i * 8 * ROW_BYTES);
    };

    floatx16 acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    // per-lane read offsets (floats): row * 32 + physical chunk * 4
    const int f = (idx >> 1) & 7;
    int coff[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) coff[s4] = ((hh * 4 + s4) ^ f) * 4;
    const int arow = (wr * (32 * TM) + idx) * 32;
    const int brow = TILE_FLOATS + (wc * 64 + idx) * 32;

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        glds_wait_all();                       // this wave's pieces of tile kt have landed ...
        __syncthreads();                       // ... and so have everybody else's; buffer (kt+1)&1 is free again
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const float* sb = smem + (kt & 1) * BUF_FLOATS;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            float4 av[TM], bv[2];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const float4*>(sb + arow + i * 32 * 32 + coff[s4]);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) bv[jj] = *reinterpret_cast<const float4*>(sb + brow + jj * 32 * 32 + coff[s4]);
            if constexpr (BF16) {
                // one 16-B chunk = 8 bf16 = the whole K=16 operand of a 32x32x16 MFMA for this half-wave
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[i]),
                                                                            __builtin_bit_cast(bf16x8, bv[jj]), acc[i][jj], 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const float ae = e == 0 ? av[i].x : (e == 1 ? av[i].y : (e == 2 ? av[i].z : av[i].w));
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const float be = e == 0 ? bv[jj].x : (e == 1 ? bv[jj].y : (e == 2 ? bv[jj].z : bv[jj].w));
                            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, be, acc[i][jj], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    // ---- epilogue. C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if constexpr (MODE == MODE_ARGMAX) {
        // fused row arg-max over this wave's 64 columns: one partial per (row, 2 * nblk + wc)
        float bv[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int col = n0 + wc * 64 + jj * 32 + idx;
            bv[jj] = (p.bias && col < p.N) ? p.bias[col] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * (32 * TM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                float best = -INFINITY;
                int besti = 0x7fffffff;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int col = n0 + wc * 64 + jj * 32 + idx;
                    if (col < p.N) {
                        const float v = acc[i][jj][r] + bv[jj];
                        if (v > best) { best = v; besti = col; }
                    }
                }
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) {   // stays inside the 32-lane half (same row)
                    const float ov = __shfl_xor(best, o, 64);
                    const int oi = __shfl_xor(besti, o, 64);
                    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
                }
                if (idx == 0 && row < p.M) {
                    const size_t o = (size_t)row * p.amax_ld + 2 * nblk + wc;
                    p.amax_val[o] = best;
                    p.amax_idx[o] = besti;
                }
            }
        }
    } else {
        constexpr bool HAS_R1 = (MODE & 1) != 0, HAS_R2 = (MODE & 2) != 0;
        // The accumulator tile goes through a wave-private LDS slab (32 rows x 64 columns per pass, row stride 68
        // floats) so that every global access of the epilogue is a float4 of a 256-B row segment: one
        // instruction touches 4 rows x 256 B instead of 2 rows x 128 B, and bias / residual loads are issued
        // in bulk instead of one dependent load per element.
        constexpr int ELD = 68;
        __syncthreads();                                       // every wave is done reading the operand tiles
        float* slab = smem + wave * (32 * ELD);
        const int c4 = lane & 15, rsub = lane >> 4;
        const int col = n0 + wc * 64 + c4 * 4;
        const bool vec_ok = p.vec_epilogue != 0;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) {
            if (vec_ok && col + 3 < p.N) bias4 = *reinterpret_cast<const float4*>(p.bias + col);
            else {
                if (col + 0 < p.N) bias4.x = p.bias[col + 0];
                if (col + 1 < p.N) bias4.y = p.bias[col + 1];
                if (col + 2 < p.N) bias4.z = p.bias[col + 2];
                if (col + 3 < p.N) bias4.w = p.bias[col + 3];
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    slab[((r & 3) + 8 * (r >> 2) + 4 * hh) * ELD + jj * 32 + idx] = acc[i][jj][r];
            // (same wave wrote and reads the slab: DS operations of one wave execute in order)
            float4 v[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) v[it] = *reinterpret_cast<const float4*>(slab + (it * 4 + rsub) * ELD + c4 * 4);
            const int row0 = m0 + wr * (32 * TM) + i * 32 + rsub;
            if (vec_ok && col + 3 < p.N) {
                float4 r1[8], r2[8];
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = row0 + it * 4;
                    const int rr = row < p.M ? row : p.M - 1;
                    if constexpr (HAS_R1) r1[it] = *reinterpret_cast<const float4*>(p.R1 + (size_t)rr * p.ldr1 + col);
                    if constexpr (HAS_R2) r2[it] = *reinterpret_cast<const float4*>(p.R2 + (size_t)rr * p.ldr2 + col);
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = row0 + it * 4;
                    float4 o;
                    o.x = v[it].x + bias4.x; o.y = v[it].y + bias4.y; o.z = v[it].z + bias4.z; o.w = v[it].w + bias4.w;
                    if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                    if constexpr (HAS_R1) { o.x = o.x + r1[it].x; o.y = o.y + r1[it].y; o.z = o.z + r1[it].z; o.w = o.w + r1[it].w; }
                    if constexpr (HAS_R2) { o.x = r2[it].x + o.x; o.y = r2[it].y + o.y; o.z = r2[it].z + o.z; o.w = r2[it].w + o.w; }
                    if (row < p.M) {
                        if (p.c_bf16) {
                            uint2 pk;
                            pk.x = (unsigned)f32_to_bf16_rn(o.x) | ((unsigned)f32_to_bf16_rn(o.y) << 16);
                            pk.y = (unsigned)f32_to_bf16_rn(o.z) | ((unsigned)f32_to_bf16_rn(o.w) << 16);
                            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.C) + (size_t)row * p.ldc + col) = pk;
                        } else {
                            *reinterpret_cast<float4*>(p.C + (size_t)row * p.ldc + col) = o;
                        }
                    }
                }
            } else {
                // ragged right edge / unaligned leading dimensions: element-wise
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = row0 + it * 4;
                    if (row >= p.M) continue;
                    const float vv[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
                    const float bb[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (col + e >= p.N) continue;
                        float o = vv[e] + bb[e];
                        if (p.relu) o = fmaxf(o, 0.f);
                        if constexpr (HAS_R1) o = o + p.R1[(size_t)row * p.ldr1 + col + e];
                        if constexpr (HAS_R2) o = p.R2[(size_t)row * p.ldr2 + col + e] + o;
                        if (p.c_bf16) reinterpret_cast<unsigned short*>(p.C)[(size_t)row * p.ldc + col + e] = f32_to_bf16_rn(o);
                        else p.C[(size_t)row * p.ldc + col + e] = o;
                    }
                }
            }
        }
    }
}

}  // namespace

int launch_gemm_f32(const GemmArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem");
    const int bk = a.ab_bf16 ? 64 : 32, align = a.ab_bf16 ? 8 : 4;
    PF_REQUIRE(a.K % bk == 0, "gemm: K must be a multiple of 32 (fp32) / 64 (bf16): pad the operand");
    PF_REQUIRE(a.lda % align == 0 && a.ldw % align == 0, "gemm: operand row strides must be multiples of 16 bytes");
    PF_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm: operands must be 16-B aligned");
    // 64-row block tiles when the 128-row grid cannot even give every CU its two workgroups (2 x 256 CUs)
    const int nN = ceil_div(a.N, BN);
    const bool half_tile = ceil_div(a.M, 128) * nN < 512;
    const int BMr = half_tile ? 64 : 128;
    const int nM = ceil_div(a.M, BMr);
    if (a.amax_val) PF_REQUIRE(a.amax_ld >= 2 * nN, "gemm: amax_ld too small");
    const int nMpad = (nM + 7) / 8 * 8;
    GemmArgs g = a;
    g.vec_epilogue = 0;
    if (!a.amax_val) {
        PF_REQUIRE(a.C != nullptr, "gemm: null output");
        bool ok = a.N % 4 == 0 && a.ldc % 4 == 0 && ((uintptr_t)a.C & 15) == 0;
        if (a.bias) ok = ok && ((uintptr_t)a.bias & 15) == 0;
        if (a.R1) ok = ok && a.ldr1 % 4 == 0 && ((uintptr_t)a.R1 & 15) == 0;
        if (a.R2) ok = ok && a.ldr2 % 4 == 0 && ((uintptr_t)a.R2 & 15) == 0;
        g.vec_epilogue = ok ? 1 : 0;
    }
    const dim3 grid((unsigned)nMpad * nN), block(256);
    const int mode = a.amax_val ? MODE_ARGMAX : ((a.R1 ? 1 : 0) | (a.R2 ? 2 : 0));
    if (a.conv_taps > 0) {
        PF_REQUIRE(!a.ab_bf16 && mode == 0 && a.conv_zero && a.conv_D % 32 == 0 && a.K == a.conv_taps * a.conv_D && a.conv_T > 0 &&
                   a.M % a.conv_T == 0 && a.conv_left >= 0 && a.conv_left < a.conv_taps && ((uintptr_t)a.conv_zero & 15) == 0,
                   "gemm: the im2col form needs fp32 operands, no addends, conv_D % 32 == 0, K == taps * conv_D, M == B * conv_T");
        if (half_tile) hipLaunchKernelGGL((gemm_f32_mfma_kernel<0, false, 1, true>), grid, block, 0, stream, g, nM, nN);
        else hipLaunchKernelGGL((gemm_f32_mfma_kernel<0, false, 2, true>), grid, block, 0, stream, g, nM, nN);
        PF_HIP_TRY(hipGetLastError());
        return 0;
    }
#define PF_LAUNCH_GEMM(MODE_)                                                                                          \
    do {                                                                                                               \
        if (a.ab_bf16) {                                                                                               \
            if (half_tile) hipLaunchKernelGGL((gemm_f32_mfma_kernel<MODE_, true, 1>), grid, block, 0, stream, g, nM, nN);  \
            else hipLaunchKernelGGL((gemm_f32_mfma_kernel<MODE_, true, 2>), grid, block, 0, stream, g, nM, nN);             \
        } else {                                                                                                       \
            if (half_tile) hipLaunchKernelGGL((gemm_f32_mfma_kernel<MODE_, false, 1>), grid, block, 0, stream, g, nM, nN); \
            else hipLaunchKernelGGL((gemm_f32_mfma_kernel<MODE_, false, 2>), grid, block, 0, stream, g, nM, nN);            \
        }                                                                                                              \
    } while (0)
    switch (mode) {
        case 0: PF_LAUNCH_GEMM(0); break;
        case 1: PF_LAUNCH_GEMM(1); break;
        case 2: PF_LAUNCH_GEMM(2); break;
        case 3: PF_LAUNCH_GEMM(3); break;
        default: PF_LAUNCH_GEMM(MODE_ARGMAX); break;
    }
#undef PF_LAUNCH_GEMM
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pf
