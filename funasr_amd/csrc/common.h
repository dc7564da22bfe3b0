// Shared helpers for the gfx950 kernels of the Paraformer / SenseVoice hot path.
// CDNA4 only: 64-wide wavefronts, MFMA, LDS. No CUDA/compat branches on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <atomic>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace pf {

// A process may drive more than one device (threads, replicas on different GPUs): function attributes are per device, and so is the
// CU count a shape choice looks at -- neither may be cached in a plain function-local static (ADVICE, round 5).
inline int current_device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
    return d & 63;
}
struct PerDeviceOnce {                       // `static PerDeviceOnce once; if (!once.done()) { ...configure...; once.mark(); }`
    std::atomic<bool> flag[64];
    PerDeviceOnce() { for (auto& f : flag) f.store(false, std::memory_order_relaxed); }
    bool done() const { return flag[current_device_slot()].load(std::memory_order_acquire); }
    void mark() { flag[current_device_slot()].store(true, std::memory_order_release); }
};
inline int device_cu_count() {               // multiProcessorCount of the CURRENT device (256 on an MI355X)
    static std::atomic<int> n[64];
    const int slot = current_device_slot();
    int v = n[slot].load(std::memory_order_relaxed);
    if (v > 0) return v;
    int dev = 0;
    hipDeviceProp_t pr;
    v = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) v = pr.multiProcessorCount;
    n[slot].store(v, std::memory_order_relaxed);
    return v;
}


// thread-local last error text, surfaced through pf_last_error()
void set_error(const std::string& msg);
const char* get_error();

#define PF_HIP_TRY(expr)                                                         \
    do {                                                                         \
        hipError_t _e = (expr);                                                  \
        if (_e != hipSuccess) {                                                  \
            pf::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));    \
            return -2;                                                           \
        }                                                                        \
    } while (0)

#define PF_REQUIRE(cond, msg)                                                    \
    do {                                                                         \
        if (!(cond)) {                                                           \
            pf::set_error(std::string("invalid argument: ") + (msg));            \
            return -1;                                                           \
        }                                                                        \
    } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// fp32 -> bf16, round to nearest even (finite inputs)
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    unsigned int u = __builtin_bit_cast(unsigned int, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
    return __builtin_bit_cast(float, (unsigned int)h << 16);
}

// fp32 -> three bf16 with x = hi + mid + lo exactly (each difference below is exact in fp32; 8 + 8 + 8 significand bits)
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    h = f32_to_bf16(x);
    const float r1 = x - bf16_to_f32(h);
    m = f32_to_bf16(r1);
    const float r2 = r1 - bf16_to_f32(m);
    l = f32_to_bf16(r2);
}
// two values -> three packed bf16 pairs (low half = first value) with v_cvt_pk_bf16_f32 (round to nearest even)
typedef __bf16 pf_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pf_floatx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    const pf_floatx2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pf_bf16x2));
}
__device__ __forceinline__ void split3_pk(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16(a, b);
    float ra = a - __builtin_bit_cast(float, h << 16), rb = b - __builtin_bit_cast(float, h & 0xffff0000u);
    m = cvt_pk_bf16(ra, rb);
    ra -= __builtin_bit_cast(float, m << 16); rb -= __builtin_bit_cast(float, m & 0xffff0000u);
    l = cvt_pk_bf16(ra, rb);
}
// four consecutive values -> the three planes at p, p + plane, p + 2 plane (8-B stores)
__device__ __forceinline__ void store_split3x4(unsigned short* p, size_t plane, const float (&o)[4]) {
    unsigned h0, m0, l0, h1, m1, l1;
    split3_pk(o[0], o[1], h0, m0, l0);
    split3_pk(o[2], o[3], h1, m1, l1);
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + plane) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(p + 2 * plane) = make_uint2(l0, l1);
}

// ---------------------------------------------------------------- two-plane fp16 split (gemm_f16x2.hip)
// x (already multiplied by the tensor's power-of-two scale) = hi + lo to 2^-23 |x| in the worst case (x at the bottom of hi's
// binade and lo at the top of its own; 2^-25 |x| on average): hi = f16(x) (11 significand bits, round to nearest even),
// lo = f16(x - hi) (the difference is exact in fp32; lo keeps its leading 11 bits, or every bit down to 2^-24 once it is
// subnormal). |x| must stay below 65504: every producer's scale comes from an a-priori bound.
typedef _Float16 pf_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {
    const pf_floatx2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pf_f16x2));
}
__device__ __forceinline__ void split2_pk(float a, float b, unsigned& h, unsigned& l) {
    const pf_floatx2 v = {a, b};
    const pf_f16x2 hv = __builtin_convertvector(v, pf_f16x2);
    const pf_floatx2 back = __builtin_convertvector(hv, pf_floatx2);
    h = __builtin_bit_cast(unsigned, hv);
    l = cvt_pk_f16(a - back.x, b - back.y);
}
// four consecutive values (times `scale`, a power of two) -> the two planes at p and p + plane (8-B stores)
__device__ __forceinline__ void store_split2x4(unsigned short* p, size_t plane, const float (&o)[4], float scale) {
    unsigned h0, l0, h1, l1;
    split2_pk(o[0] * scale, o[1] * scale, h0, l0);
    split2_pk(o[2] * scale, o[3] * scale, h1, l1);
    *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(p + plane) = make_uint2(l0, l1);
}

// the same, but lanes 2k / 2k + 1 -- which hold columns c .. c + 3 / c + 4 .. c + 7 of ONE row, both active -- trade halves
// (one DPP quad permute per dword) so that each issues ONE 16-B store (even lane: the hi plane's 8 columns, odd lane: the lo
// plane's) instead of two 8-B stores: half the store instructions, each at the full store width. `p` = this lane's four
// columns in the hi plane (16-B aligned for the even lane).
__device__ __forceinline__ unsigned dpp_swap1(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ void store_split2x4_pair(unsigned short* p, size_t plane, const float (&o)[4], float scale, int lane) {
    unsigned h0, l0, h1, l1;
    split2_pk(o[0] * scale, o[1] * scale, h0, l0);
    split2_pk(o[2] * scale, o[3] * scale, h1, l1);
    const bool odd = (lane & 1) != 0;
    const unsigned r0 = dpp_swap1(odd ? h0 : l0), r1 = dpp_swap1(odd ? h1 : l1);
    if (!odd) *reinterpret_cast<uint4*>(p) = make_uint4(h0, h1, r0, r1);
    else *reinterpret_cast<uint4*>(p - 4 + plane) = make_uint4(r0, r1, l0, l1);
}

// ---------------------------------------------------------------- LayerNorm arithmetic shared by layernorm_kernel (rowwise.hip)
// and the fused residual + LayerNorm epilogue (gemm_f16x2_row.hip): both evaluate exactly these expression trees in the same
// order (per 4-column chunk, then chunk l + chunk l + 64, then the 64-lane xor butterfly 32, 16, .. 1), so the fused epilogue
// returns the bits of the stand-alone kernel.
// Contraction is OFF inside these helpers: under the default -ffp-contract=fast the compiler decides per call site whether
// a * a + b * b becomes mul + fma or two muls and an add (it depends on how the surrounding code was vectorised), so two
// kernels calling the same helper on the same values could round differently. With every product and sum rounded on its own
// the helpers are ONE arithmetic, whatever kernel they are inlined into (the fused kernels are tested bitwise against the
// stand-alone LayerNorm, and the choice between them may depend on the batch's row count).
__device__ __forceinline__ float ln_sum4(const float4 v) {
#pragma clang fp contract(off)
    return (v.x + v.y) + (v.z + v.w);
}
__device__ __forceinline__ float ln_sqdev4(const float4 v, const float mean) {
#pragma clang fp contract(off)
    const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
    return (a * a + b * b) + (cc * cc + d * d);
}
__device__ __forceinline__ float ln_mean(const float s, const int D) {
#pragma clang fp contract(off)
    return s / (float)D;
}
__device__ __forceinline__ float ln_rstd(const float q, const int D, const float eps) {
#pragma clang fp contract(off)
    return 1.0f / sqrtf(q / (float)D + eps);
}
__device__ __forceinline__ float4 ln_apply4(const float4 v, const float mean, const float rstd, const float4 g, const float4 b) {
#pragma clang fp contract(off)
    float4 o;
    o.x = (v.x - mean) * rstd * g.x + b.x;
    o.y = (v.y - mean) * rstd * g.y + b.y;
    o.z = (v.z - mean) * rstd * g.z + b.z;
    o.w = (v.w - mean) * rstd * g.w + b.w;
    return o;
}

// ---------------------------------------------------------------- LDS-DMA issued behind the compiler's back
// hipcc cannot prove that a ds_read does not alias an LDS-DMA in flight (SIInsertWaitcnts only separates them with
// alias-scope metadata HIP does not attach), so with __builtin_amdgcn_global_load_lds it puts `s_waitcnt vmcnt(0)` in
// front of the first ds_read of every k-step: the prefetch of tile t+1 is drained before tile t is multiplied and the
// double buffer overlaps nothing. Issued from inline asm the DMA is invisible to that pass; the kernel then owns the
// ordering: glds_wait_all() by every wave, then the workgroup barrier, then the ds_reads (cdna guide 5.7, item 1).
// lds_byte_addr must be wave-uniform (readfirstlane it); each lane lands 16 bytes at lds_byte_addr + 16 * lane.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}
// the same with the non-temporal hint: for operand panels exactly one workgroup reads (they should not displace the weight
// panels every workgroup of the XCD re-reads from its L2)
__device__ __forceinline__ void glds16_nt(const void* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte_addr)
                 : "memory");
}
// the same with a uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset (the saddr form): the per-piece part of the
// address costs scalar ALU only and no address VGPRs
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
                 : "memory");
}
__device__ __forceinline__ void glds_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// ---------------------------------------------------------------- kernel launchers
// (all take device pointers; `stream` is the HIP stream the caller owns)

struct GemmArgs {
    const float* A;   int lda;   // [M, K] activations, row stride lda
    const float* W;   int ldw;   // [N, K] weights (torch Linear layout), row stride ldw
    const float* bias;           // [N] or nullptr
    const float* R1;  int ldr1;  // optional addend #1 (added first:  v = v + R1)
    const float* R2;  int ldr2;  // optional addend #2 (added second: v = R2 + v)
    float* C;         int ldc;   // [M, N]
    int M, N, K;                 // K % 32 == 0
    int relu;                    // apply relu after bias, before addends
    // optional fused arg-max over N (used for vocabulary projections): when amax_val != nullptr the
    // kernel does not write C but per-(row, column-block) partial maxima
    float* amax_val; int* amax_idx; int amax_ld;
    int vec_epilogue;            // set by the launcher: all epilogue operands allow aligned float4 access
    int ab_bf16;                 // A and W hold bf16 (lda / ldw in ELEMENTS); fp32 accumulate and epilogue
    int c_bf16;                  // C is written as bf16 (ldc in elements)
    // ---- small-M kernel only (gemm_skinny.hip): LayerNorm carried between GEMMs instead of a launch of its own (the streaming
    // step is a chain of ~600 dependent launches; a stand-alone LayerNorm of 15 rows costs a whole launch slot).
    // Producer side: ln_stats_out [M][N / 16][2] receives, per output row and 16-column block, (sum, sum of squares) of the
    // FINISHED outputs (after bias / ReLU / addends); N % 16 == 0.
    // Consumer side: ln_stats_in != nullptr computes C = W LayerNorm(A) + bias as rstd (W (gamma A) - mean c1) + c2 with the row's
    // (mean, rstd) from the K / 16 block partials (summed in an order that depends on K only; K <= 2048 = the normalised width,
    // no padding columns) and the per-weight constants ln_c1 = W gamma, ln_c2 = W beta + bias ([N] each, launch_ln_consts; `bias`
    // is then ignored). ln_g = gamma multiplies the A fragments in the loop; ln_g == nullptr: the producer stored gamma A
    // (out_gamma). Producer side also: out_gamma [N] != nullptr stores C * out_gamma (the partials stay those of C).
    const float* ln_c1; const float* ln_c2; const float* out_gamma;
    // the four-workgroup form of a <= 32-row problem (see gemm_skinny.hip): slice tiles [tiles][16][rows of a tile][16] and one
    // zeroed int per tile (the kernel leaves them zero); tiles = ceil(N / 16) * ceil(M / 16 or 32). Same bits as without.
    float* ws_part; int* ws_count;
    float* ln_stats_out;
    const float* ln_stats_in; const float* ln_g; float ln_eps;
    // ---- tile kernel only (gemm_f32.hip), fp32 operands: A is read as the im2col of a Conv1d over time WITHOUT materialising it
    // (CifPredictorV2's cif_conv1d, funasr/models/paraformer/cif_predictor.py:275-278). conv_taps > 0: K = conv_taps * conv_D; column
    // k = tap * conv_D + c of row (b, t) is A[(b, t + tap - conv_left), c], zero where t + tap - conv_left falls outside [0, conv_T);
    // M = B * conv_T rows, conv_D % 32 == 0 (a K tile never straddles two taps). conv_zero: >= 128 B of zeros on the device. The k order
    // per output element is that of the materialised GEMM: bitwise the same result.
    int conv_taps, conv_D, conv_T, conv_left; const float* conv_zero;
};
int launch_gemm_f32(const GemmArgs& a, hipStream_t stream);
// small-M weight-streaming variant (gemm_skinny.hip); same contract, no fused arg-max
bool gemm_skinny_applicable(const GemmArgs& a);
// the small-M GEMM for up to 16 weight matrices sharing A / M / N / K / strides (gemm_skinny.hip): one launch, a row's bits
// those of the single launches
struct GemmBatch { int n; const float* W[16]; const float* bias[16]; float* C[16]; };
// c1[n] = sum_k gamma[k] W[n][k], c2[n] = sum_k beta[k] W[n][k] + bias[n] (bias may be null): the constants of the LayerNorm form
// of the small-M GEMM (rowwise.hip; one wave per n, fixed summation order)
int launch_ln_consts(const float* W, int ldw, int N, int K, const float* gamma, const float* beta, const float* bias, float* c1, float* c2,
                     hipStream_t stream);
int launch_gemm_skinny_batch(const GemmArgs& a, const GemmBatch& t, hipStream_t stream);
int launch_gemm_skinny(const GemmArgs& a, hipStream_t stream);

int launch_cast_bf16(const float* x, unsigned short* y, size_t n, hipStream_t stream);

// fp32-accurate GEMM with both operands as three bf16 planes (gemm_split3.hip): x = hi + mid + lo, the planes of a
// matrix are `*_plane` ELEMENTS apart, row strides in elements
struct Gemm3Args {
    const unsigned short* A; int lda; size_t a_plane;   // [3][M, K]
    const unsigned short* W; int ldw; size_t w_plane;   // [3][N, K]
    const float* bias;
    const float* R1; int ldr1;                          // v = v + R1, then v = R2 + v (as GemmArgs)
    const float* R2; int ldr2;
    float* C; int ldc;                                  // fp32 output ...
    unsigned short* C3; int ldc3; size_t c_plane;       // ... or (C3 != nullptr) three bf16 planes of the result
    int M, N, K;                                        // K % 32 == 0, N % 4 == 0
    int relu;
};
int launch_gemm_split3(const Gemm3Args& a, hipStream_t stream);
// fp32 [M, N] -> three bf16 planes [M, ldy]; columns N..ldy-1 are written as zero
int launch_split3(const float* x, int ldx, unsigned short* y, int ldy, size_t plane, int M, int N, hipStream_t stream);

// fp32-accurate GEMM with both operands as TWO fp16 planes (gemm_f16x2.hip): x * 2^e = hi + lo; three MFMA products
// (hi*hi + hi*lo + lo*hi) into one fp32 accumulator, the epilogue multiplies by `oscale` = 2^-(e_A + e_W) first
struct Gemm2Args {
    const unsigned short* A; int lda; size_t a_plane;   // [2][M, K] fp16
    const unsigned short* W; int ldw; size_t w_plane;   // [2][N, K] fp16
    float oscale;                                       // exact power of two
    const float* oscale_dev;                            // optional device float multiplied into oscale (a scale chosen on the device)
    const float* bias;
    const float* R1; int ldr1;                          // v = v + R1, then v = R2 + v (as GemmArgs)
    const float* R2; int ldr2;
    float* C; int ldc;                                  // fp32 output ...
    unsigned short* C2; int ldc2; size_t c_plane;       // ... or (C2 != nullptr) two fp16 planes of result * cscale
    float cscale;
    int M, N, K;                                        // K % 32 == 0, N % 4 == 0
    int relu;
    int tile;                                           // 0 = pick by shape, 1 = 256 x 128, 2 = 256 x 256 (measurement hook)
    long a_kstep, w_kstep;                              // > 0: K-blocked operand layout [K / 32][rows][32] (lda / ldw = 32, *_kstep = rows * 32)
    // QKV form (qkv_D > 0, N == 3 qkv_D, qkv_D % 256 == 0, M % 16 == 0): the fused q|k|v projection feeding
    // attention_f16x2.hip. Columns [0, D) -> planes of (result * q_mul) at Qp (ld D); [D, 2D) -> planes of
    // (result * k_mul) at Kp; [2D, 3D) -> fp32 at C (ld ldc; the FSMN memory block reads it) and the TRANSPOSED
    // planes of (result * v_mul) at VT[d][col(row)], col = row with bits 2 and 3 swapped
    int qkv_D;
    unsigned short* Qp; unsigned short* Kp; size_t qk_plane;
    unsigned short* VT; int ldvt; size_t vt_plane;
    float q_mul, k_mul, v_mul;
    // KV form (kv_form != 0, N == 2 qkv_D): the decoder's fused k|v projection of the memory: columns [0, D) -> K planes,
    // [D, 2D) -> V^T planes only (C may be null). kv_mul_dev (optional): device floats {k, v} multiplied into k_mul / v_mul
    // (scales chosen on the device from max |memory|)
    int kv_form;
    const float* kv_mul_dev;
    // fused row arg-max (vocabulary projections): when amax_val != nullptr nothing is written to C / C2 but one partial
    // (max, lowest index of the max) per (row, wave column range): entry 2 * column_block + wave_column of row `row`
    float* amax_val; int* amax_idx; int amax_ld;
    // split-K form (ksplit > 1; fp32 output only, K % (32 ksplit) == 0): the K range is cut into ksplit slices computed by
    // ksplit times as many 128 x 128 blocks (grid.y = slice) into `part` [ksplit][M][N] (fp32, scaled by oscale), then ONE
    // reduce launch sums the slices in slice order and applies bias / relu / R1 / R2 -> C. For GEMMs of at most a block per CU
    // with a long K (the streaming step's w_2: M = 15 S rows, K = 2048), whose lone blocks run at a fraction of the matrix rate.
    // Deterministic, but NOT the bits of the unsplit kernel (another summation order): a caller takes it always or never.
    int ksplit; float* part;
    // split-K form only: ln_g != nullptr folds the LayerNorm that follows into the second launch (launch_splitk_reduce_ln,
    // rowwise.hip: the bits of the two launches): LayerNorm(C) goes to ln_y as fp32 (ln_out 0, row stride ln_ldy) or as two fp16
    // planes of result * ln_oscale (ln_out 3, ln_plane elements apart). N <= 512, no relu.
    // Without ln_g the split-K form may write the planes of the result * cscale (C2 / ldc2 / c_plane: w_1 -> w_2) instead of C.
    const float* ln_g; const float* ln_b; float ln_eps; float* ln_y; int ln_ldy; int ln_out; size_t ln_plane; float ln_oscale;
    int kslices;                                        // set by the launcher: what the kernel sees (slice = blockIdx.y)
};
// number of arg-max partials per row launch_gemm_f16x2 writes for an N-column problem
int gemm_f16x2_argmax_parts(int M, int N);
int launch_gemm_f16x2(const Gemm2Args& a, hipStream_t stream);
// the four-wave 256 x 256 shape (gemm_f16x2_w4.hip; Gemm2Args.tile 7): bitwise the results of the other shapes.
// abl (measurement only): 0 product, 1 no global stores, 2 no epilogue, 3 no epilogue and no operand DMA
bool gemm_f16x2_w4_ok(const Gemm2Args& a);
int launch_gemm_f16x2_w4(const Gemm2Args& a, int abl, hipStream_t stream);
// the persistent, wave-specialised 256 x 128 shape (gemm_f16x2_ps.hip; Gemm2Args.tile 10): bitwise the results of the other shapes
bool gemm_f16x2_ps_ok(const Gemm2Args& a);
int launch_gemm_f16x2_ps(const Gemm2Args& a, hipStream_t stream);
struct GemmRowArgs;
bool gemm_f16x2_w4_row_ok(const GemmRowArgs& a);
int launch_gemm_f16x2_w4_row(const GemmRowArgs& a, hipStream_t stream);
// Full-row form for N == 512 (gemm_f16x2_row.hip): one workgroup per 128 complete rows, epilogue
//   v = relu?(A W^T * oscale + bias);  v = v + R1;  v = R2 + v;  C = v (fp32, optional);
//   ln_g != nullptr: y = LayerNorm(v; ln_g, ln_b, ln_eps) -> two fp16 planes of y * yscale at Y2, or fp32 at Yf
// bitwise equal to launch_gemm_f16x2 followed by launch_layernorm (same products, same summation orders)
struct GemmRowArgs {
    const unsigned short* A; int lda; size_t a_plane;   // [2][M, K] fp16
    const unsigned short* W; int ldw; size_t w_plane;   // [2][512, K] fp16
    float oscale; const float* oscale_dev;
    const float* bias;
    const float* R1; int ldr1;
    const float* R2; int ldr2;
    float* C; int ldc;                                  // may alias R1 / R2 (every element is read, then written, by one lane)
    const float* ln_g; const float* ln_b; float ln_eps;
    unsigned short* Y2; int ldy2; size_t y_plane; float yscale;
    float* Yf; int ldyf;
    int M, N, K;                                        // N == 512, K % 32 == 0
    int relu;
    int a_nt;                                           // non-temporal hint on the A panel's loads (measurement hook)
    // FSMN form (fs_v != nullptr, R1 == nullptr, ln_g != nullptr, M % 16 == 0): the first addend is the FSMN memory block of
    // the fp32 rows fs_v [M, 512] (row stride ldfv) with taps fs_w [512, 11] (left padding 5), computed in the epilogue;
    // fs_lo / fs_hi (device int32 [M / 16]): the valid input rows [lo, hi) of the sequence that owns each 16-row group
    const float* fs_v; int ldfv; const float* fs_w; const int* fs_lo; const int* fs_hi;
    // rows per block: 0 = by the row count (whole rounds over the CUs: 128, or 96 where that needs less time), 128 = the
    // 2 x 4-wave kernel (gemm_f16x2_row.hip), 96 / 129 = the 1 x 8-wave kernel (gemm_f16x2_row8.hip) with 96 / 128 rows.
    // Every choice gives the same bits.
    int block_rows;
};
// The encoder block's feed-forward in one launch (gemm_f16x2_ffn.hip): C = R + (relu(X W1^T + b1) W2^T + b2), optionally followed
// by LayerNorm(C) as planes (Y2) or fp32 (Yf). X2 / W1 / W2 are two-plane fp16 operands; the hidden activations get the plane
// scale `hscale` (2^e_h) and never leave the registers. D == 512, F % 128 == 0.
struct FfnArgs {
    const unsigned short* X2; int ldx; size_t x_plane;    // [2][M, 512] planes of norm2(x) * 2^e_x
    const unsigned short* W1; int ldw1; size_t w1_plane;  // [2][F, 512]
    const unsigned short* W2; int ldw2; size_t w2_plane;  // [2][512, F]
    const float* b1; const float* b2;                     // [F], [512] (b2 may be null)
    float oscale1, hscale, oscale2;                       // 2^-(e_x + e_w1), 2^e_h, 2^-(e_h + e_w2)
    const float* R; int ldr;                              // residual stream [M, 512]
    float* C; int ldc;                                    // may alias R
    const float* ln_g; const float* ln_b; float ln_eps;
    unsigned short* Y2; int ldy2; size_t y_plane; float yscale;
    float* Yf; int ldyf;
    int M, D, F;
    int abl;                                              // measurement only (tools/bench_ffn.py): see ffn_f16x2_kernel
    // elements between the 32-deep K stages of an operand row; 0 = row-major planes (32). K-blocked operands [K / 32][rows][32]
    // (ld* = 32, *_kstep = rows * 32): a 16-row DMA piece is one contiguous KB
    size_t x_kstep, w1_kstep, w2_kstep;
};
bool ffn_f16x2_applicable(int D, int F);
int launch_ffn_f16x2(const FfnArgs& a, hipStream_t stream);
int launch_gemm_f16x2_row8(const GemmRowArgs& a, int bm, hipStream_t stream);
bool gemm_f16x2_row_applicable(int N, int K);
int launch_gemm_f16x2_row(const GemmRowArgs& a, hipStream_t stream);
// fp32 [M, N] * scale -> two fp16 planes [M, ldy]; columns N..ldy-1 are written as zero
// seq_out > 0: output row b * seq_out + t holds input row b * seq_in + t for t < seq_in and zeros for the padding rows
// (M counts OUTPUT rows)
int launch_split2(const float* x, int ldx, unsigned short* y, int ldy, size_t plane, int M, int N, float scale,
                  hipStream_t stream, const float* scale_dev = nullptr, int seq_out = 0, int seq_in = 0);
// per decoder layer l: out[4 l .. 4 l + 3] = {k_mul, v_mul, 1 / k_mul, 1 / v_mul}, the power-of-two plane scales of
// k = Wk m + bk and v from the bounds (*amax_dev) * l1[2 l + {0, 1}] + bmax[2 l + {0, 1}]
int launch_kv_scales(const float* amax_dev, const float* l1_bmax_dev, int n_layers, float* out, hipStream_t stream);
// sc[0] = 2^e, sc[1] = 2^-e with e = floor(log2(32768 / *amax_dev)): the plane scale of a tensor whose max |x| is only
// known on the device (and its inverse for the consuming GEMM's oscale_dev)
int launch_pow2_scale(const float* amax_dev, float* sc, hipStream_t stream);
// max |x| over n floats -> *out_dev (one float, device); load-time helper for the per-tensor weight scale
int launch_absmax(const float* x, size_t n, float* out_dev, hipStream_t stream);
// max over rows n of (in_bound * sum_k |W[n, k]| + |bias[n]|) -> *out_dev: a-priori bound on a Linear's outputs
int launch_rowl1_bound(const float* W, int rows, int cols, int ld, const float* bias, float in_bound, float* out_dev,
                       hipStream_t stream);

// out_mode 3: y receives the two fp16 planes of result * oscale (gemm_f16x2.hip); seq_out > 0: output row
// b * seq_out + t reads input row b * seq_in + t
// out_mode 1 / in_bf16: y / x is a bf16 buffer (ldy / ldx in elements); out_mode 2: y receives the three bf16 planes
// of the result (split3), `plane` elements apart; statistics are always fp32
int launch_splitk_reduce_ln(const float* part, int slices, size_t slice_stride, int M, int D, const float* bias, int relu, const float* R1, int ldr1,
                            const float* R2, int ldr2, float* C, int ldc, const float* gamma, const float* beta, float eps, float* y, int ldy, int out_mode,
                            size_t plane, float oscale, hipStream_t stream);
int launch_layernorm(const float* x, int ldx, const float* gamma, const float* beta, float* y, int ldy,
                     int M, int D, int Dpad, float eps, hipStream_t stream, int out_mode = 0, int in_bf16 = 0,
                     size_t plane = 0, float oscale = 1.f, int seq_out = 0, int seq_in = 0, const int* out_map = nullptr);
// out_map (device int32 [M]): row r is written to output row out_map[r] (negative: dropped) -- the packed row layout back
// to the caller's [B, T] rows

int launch_scale_cols(float* x, int ld, int M, int N, float sc, hipStream_t stream);
// y[row] = x[row] - logsumexp(x[row]) over N columns (may run in place)
int launch_log_softmax(const float* x, int ldx, float* y, int ldy, int M, int N, hipStream_t stream);
// Tp > T: y is the padded layout [B, Tp, D] (rows t >= T zero)
int launch_scale_add_pe(const float* x, const float* pe, float* y, int B, int T, int D, float scale,
                        hipStream_t stream, int Tp = 0);
// packed row layout: y[r] = x[map[r]] * scale + pe[map[r] % T] for the M rows of `map` (device int32; negative: zero row)
int launch_scale_add_pe_rows(const float* x, const float* pe, float* y, const int* map, int M, int T, int D, float scale,
                             hipStream_t stream);

struct FsmnArgs {
    const float* in;  int ldin;   // [B*T, C] view (row stride ldin)
    const float* w;               // [C, K] depthwise taps
    const float* R;   int ldr;    // optional residual (added after masking), or nullptr
    float* out;       int ldo;
    const int* lens;              // device int32 [B]: valid rows per sequence
    int B, T, C, K, left_pad;
    int in_bf16;                  // `in` holds bf16 (ldin in elements)
    const int* offs;              // optional packed layout (device int32 [B + 1]): sequence b occupies rows [offs[b], offs[b + 1]) of in / R / out, the first lens[b] of them valid
                                  // (T then only sizes the grid: >= max lens)
};
int launch_fsmn(const FsmnArgs& a, hipStream_t stream);

struct AttnArgs {
    const float* Q; int ldq;      // row (b*Tq + t), head h at column h*128
    const float* K; int ldk;      // row (b*Tk + t)
    const float* V; int ldv;
    float* O;       int ldo;
    const int* klens;             // device int32 [B] valid keys per sequence (>= 1); unused with a second source
    int B, H, Tq, Tk;
    float scale;
    // optional second key/value source (streaming): keys [0, n1[b]) come from (K, V), the next n2 from (K2, V2)
    const float* K2; int ldk2;    // row (b*T2 + t)
    const float* V2; int ldv2;
    int T2, n2;
    const int* n1_dev; int n1_stride;   // device int32, element b * n1_stride (stride 0 = one value for all)
    // fp32 kernels only: write the three bf16 planes of the result (split3; same ldo, `o_plane` elements apart) instead of O
    unsigned short* O3; size_t o_plane;
    // launch_attention_f32 only: the caller is the streaming step (Tq <= 32 rows per stream, a few dozen keys): take the
    // few-query kernel. Set by the CALLER, never derived from the batch (a result must not depend on its neighbours)
    int few_q;
    // few-query kernel with ONE workgroup per (stream, head) (Tq <= 16) and a second source only: after its keys are read,
    // the workgroup appends rows [app_r0, app_r0 + app_rows) of (K2, V2) -- its head's columns -- to the ring (K, V; Tk =
    // capacity) at write pointer app_wp[b * app_wp_stride], like ring_append_kernel (stream.hip); app_gate[b] < 1 skips
    int app_rows, app_r0; const int* app_wp; int app_wp_stride; const int* app_gate;
    // app_mod > 0 (the ENCODER's ring when chunk_left + max_frames > look_back * chunk_cur): the reference leaves the FIRST chunk's cache
    // untrimmed (sanm/attention.py:356-361) and trims to look_back * chunk_size[1] rows from the second chunk on (:353-355). The ring
    // then holds Tk >= app_mod rows: the first append (n1 == 0) is written linearly, later ones modulo app_mod; rows [0, n1) are valid
    int app_mod;
    // launch_attention_small only (SANMVadEncoder, ct_transformer_streaming/encoder.py:372-418): 0 = key padding only,
    // 1 = also causal (key <= query), 2 = also the VAD corner (transformer/utils/mask.py:38-52): queries before
    // vad_pos[b] - 1 do not see keys from vad_pos[b] on (when 0 < vad_pos[b] < Tk)
    int mask_mode; const int* vad_pos;
    // few-query kernel only (the streaming step is a chain of dependent launches; the FSMN memory block reads the same V
    // projection and is independent of the attention): fs_in != nullptr adds workgroups that compute the block's FSMN
    // memory, the arithmetic of fsmn_kernel<11, 5> (rowwise.hip) with every one of the fs_T rows per stream valid:
    // fs_out[b * fs_T + t] = fs_in[..] + sum_j fs_w[c][j] * fs_in[b * fs_T + t - 5 + j] over H * 128 channels
    const float* fs_in; int fs_ldin; const float* fs_w; float* fs_out; int fs_ldo; int fs_T;
    // few-query kernel only: O2 != nullptr writes the result as the two fp16 planes of result * o2_scale (split2, o2_plane
    // elements apart, row stride ldo2) -- the out-projection's operand in the f16x2 step -- instead of O (the bits of O + split2)
    unsigned short* O2; int ldo2; size_t o2_plane; float o2_scale;
};
inline bool attention_takes_fewq(const AttnArgs& a) { return a.few_q && a.Tq <= 32 && !a.O3; }
// true when launch_attention_f32 will take the few-query kernel AND perform the append itself
inline bool attention_fuses_append(const AttnArgs& a) {
    // (the kernel holds the rows to append in four float4 per thread: min(app_rows, Tk) <= 16)
    return a.few_q && a.Tq <= 16 && !a.O3 && a.K2 && a.app_rows > 0 && (a.app_rows < a.Tk ? a.app_rows : a.Tk) <= 16;
}
int launch_attention_f32(const AttnArgs& a, hipStream_t stream);
// bf16 Q/K/V in, bf16 O out (strides in elements), fp32 softmax statistics and accumulators
int launch_attention_bf16(const AttnArgs& a, hipStream_t stream);
// fp32 in / fp32 (or plane) out like launch_attention_f32, both products on the bf16 MFMA from three-plane split operands
int launch_attention_split3(const AttnArgs& a, hipStream_t stream);
// self-attention on two-plane fp16 operands written by the QKV form of gemm_f16x2.hip (attention_f16x2.hip)
struct Attn2Args {
    const unsigned short* Q; int ldq; size_t q_plane;     // [2][B Tp, H 128] fp16 planes of q * d_k^-0.5 * 2^e_q
    const unsigned short* K; int ldk; size_t k_plane;     // [2][B Tp (+32 rows slack), H 128] planes of k * 2^e_k
    const unsigned short* VT; int ldvt; size_t vt_plane;  // [2][H 128, ldvt >= B Tp + 32] transposed planes of v * 2^e_v
    unsigned short* O; int ldo; size_t o_plane;           // [2][B Tp, H 128] planes of the result * 2^e_ctx
    const int* klens;                                     // device int32 [B] valid keys per sequence (1 .. Tp)
    int B, H, Tp;                                         // Tp % 16 == 0: rows per sequence of K / V^T
    int Tq;                                               // rows per sequence of Q / O (0 = Tp: self-attention)
    float sscale;                                         // 2^-(e_q + e_k)
    const float* sscale_dev;                              // optional device float multiplied into sscale
    float oscale;                                         // 2^(e_ctx - e_v - 10)
    int variant;                                          // kernel schedule (measurement hook), 0 = default
    const int* qoffs;                                     // optional packed queries: sequence b owns rows [qoffs[b], qoffs[b+1]) of
                                                          // Q / O (device int32 [B + 1]); Tq then only sizes the grid
    int xcd_nqb;                                          // set by the launcher (XCD-aware workgroup order); callers: 0, or -1 = plain order
    const int* koffs;                                     // optional packed keys: sequence b's keys are rows [koffs[b], koffs[b] +
                                                          // klens[b]) of K / V^T, ANY start row (tiles start at the 16-row group
                                                          // below it, the keys of the neighbour in front are masked); Tp unused
};
int launch_attention_f16x2(const Attn2Args& a, hipStream_t stream);

// small heads (d_k <= 64) on short sequences: plain fp32 FMAs, one wave per query (attention_small.hip)
// SeACo's attention-score filter: out[k] = sum over heads and queries of softmax_k(q . k / sqrt(d_k)) for ONE sequence
// (Q [N, H dk], K [T, >= H dk] row strides ldq / ldk; keys >= klen get probability 0); P_scratch holds H * N * T floats
int launch_asf_scores(const float* Q, int ldq, const float* K, int ldk, float* P_scratch, float* out, int H, int dk, int N, int T,
                      int klen, float scale, hipStream_t stream);
bool attention_small_applicable(const AttnArgs& a, int dk);
int launch_attention_small(const AttnArgs& a, int dk, hipStream_t stream);
// out[i] = table[ids[i]] (embedding lookup; ids clamped to [0, rows))
// out[map[i], 0..n_per) = in[i, 0..n_per): int32 rows scattered through a row map (token ids back to the padded [B, N] layout)
int launch_scatter_i32(const int* in, const int* map, int* out, int n, hipStream_t stream);
int launch_gather_rows(const float* table, int ld, int rows, const int* ids, float* out, int n, int D, hipStream_t stream);

// FSMN-VAD row kernels (vad.hip)
int launch_vad_fsmn(const float* x, int ldx, const float* w, const float* cache_in, float* cache_out, float* y, int ldy,
                    int B, int T, int C, int L, int S, hipStream_t stream);
int launch_vad_softmax_sil(const float* x, int ldx, int M, int N, const int* ids, int n_ids, float* p_sil, float* probs,
                           int ldp, hipStream_t stream);
int launch_frame_decibel(const float* wav, int n_frames, int flen, int shift, float* out, hipStream_t stream);

}  // namespace pf
