// Flash-style scaled-dot-product attention with bf16 operands on the CDNA4 matrix cores (d_k = 128): the
// throughput-mode twin of attention_f32.hip (same mapping, same masking semantics, fp32 softmax statistics and
// accumulators; Q/K/V/P enter v_mfma_f32_32x32x16_bf16 as bf16, the output is written as bf16 for the out-projection).
//
// Reference semantics: funasr/models/sanm/attention.py:270-306,322-327 (scores, key mask -inf, softmax, mask 0, .V).
//
// One workgroup = 4 waves = 128 queries of one (sequence, head); each wave owns 32 queries x d_k. Both products are
// issued "swapped" so a lane owns ONE query (q = lane & 31):
//   S^T[key][q] = sum_d K[key][d] Q[q][d]     A = K tile rows (LDS, 16-B chunks XOR-swizzled by key & 15), B = Q (regs)
//   O^T[d][q]   = sum_key V[key][d] P[q][key] A = V^T (LDS, transposed while staging), B = P
// The MFMA k index is only a pairing between A and B: for the second product the k slots of half-wave h are exactly
// the keys whose scores that lane's accumulator registers already hold (key = (r&3) + 8(r>>2) + 4h), so P goes from
// the S^T accumulator to the B operand with a register-local f32 -> bf16 pack, no cross-lane traffic, and V^T is read
// as two 8-B LDS reads per MFMA.
#include "common.h"

namespace pf {

namespace {

constexpr int DK = 128;
constexpr int KT = 32;             // keys per tile
constexpr int VLD = 36;            // V^T row stride in elements (72 B: conflict-free ds_read_b64 across d)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned pack2(float a, float b) {
    return (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16);
}

__global__ __launch_bounds__(256, 2) void attention_bf16_kernel(AttnArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned short smem[KT * DK + DK * VLD];
    unsigned short* Ks = smem;                 // [32 keys][128 d], chunk c of key r at c ^ (r & 15)
    unsigned short* Vt = smem + KT * DK;       // [128 d][VLD]

    const unsigned short* Qg = reinterpret_cast<const unsigned short*>(p.Q);
    const unsigned short* Kg = reinterpret_cast<const unsigned short*>(p.K);
    const unsigned short* Vg = reinterpret_cast<const unsigned short*>(p.V);
    unsigned short* Og = reinterpret_cast<unsigned short*>(p.O);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, idx = lane & 31;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q = blockIdx.x * 128 + wave * 32 + idx;
    const int qc = q < p.Tq ? q : p.Tq - 1;
    const int klen = p.klens[b];

    // Q fragments: step s covers d in [16s, 16s+16); half-wave h takes the 8 d's of chunk 2s + h
    bf16x8 qf[8];
    {
        const unsigned short* qp = Qg + ((size_t)b * p.Tq + qc) * p.ldq + head * DK + hh * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) qf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qp + 16 * s));
    }

    floatx16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // tile loaders. K: chunk-fastest (coalesced 256-B rows); V: key-fastest (conflict-free transposed LDS writes)
    const int kc = tid & 15, kr0 = tid >> 4;          // K: chunk, key (+16 second pass)
    const int vk = tid & 31, vc0 = tid >> 5;          // V: key, chunk (+8 second pass)
    const unsigned short* kbase = Kg + (size_t)b * p.Tk * p.ldk + head * DK + kc * 8;
    const unsigned short* vbase = Vg + (size_t)b * p.Tk * p.ldv + head * DK;

    const int ntiles = (klen + KT - 1) / KT;
    for (int kt = 0; kt < ntiles; ++kt) {
        const int k0 = kt * KT;
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = kr0 + 16 * i;
            int kr = k0 + r;
            kr = kr < klen ? kr : klen - 1;
            const uint4 kv = *reinterpret_cast<const uint4*>(kbase + (size_t)kr * p.ldk);
            *reinterpret_cast<uint4*>(&Ks[r * DK + ((kc ^ (r & 15)) * 8)]) = kv;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = vc0 + 8 * i;
            int kr = k0 + vk;
            kr = kr < klen ? kr : klen - 1;
            const uint4 vv = *reinterpret_cast<const uint4*>(vbase + (size_t)kr * p.ldv + c * 8);
            const unsigned w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Vt[(c * 8 + 2 * e) * VLD + vk] = (unsigned short)(w[e] & 0xffffu);
                Vt[(c * 8 + 2 * e + 1) * VLD + vk] = (unsigned short)(w[e] >> 16);
            }
        }
        __syncthreads();

        // ---- S^T tile (32 keys x 32 queries)
        floatx16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const unsigned short* kp = &Ks[idx * DK];
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const uint4 kv = *reinterpret_cast<const uint4*>(kp + (((2 * st + hh) ^ (idx & 15)) * 8));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kv), qf[st], s, 0, 0, 0);
        }

        // ---- online softmax for query (lane & 31); this lane holds keys k0 + (r&3) + 8(r>>2) + 4h
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            s[r] = key < klen ? s[r] * p.scale : -INFINITY;     // (q * d_k^-0.5) . k == (q . k) * d_k^-0.5
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = expf(s[r] - m_new);
            psum += s[r];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

        // ---- O^T += V^T P^T. step st uses this lane's registers r in [8st, 8st+8): keys 16st + 4h + {0..3, 8..11}
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            uint4 pk;
            pk.x = pack2(s[8 * st + 0], s[8 * st + 1]);
            pk.y = pack2(s[8 * st + 2], s[8 * st + 3]);
            pk.z = pack2(s[8 * st + 4], s[8 * st + 5]);
            pk.w = pack2(s[8 * st + 6], s[8 * st + 7]);
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pk);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned short* vp = &Vt[(d * 32 + idx) * VLD + 16 * st + 4 * hh];
                const uint2 lo = *reinterpret_cast<const uint2*>(vp);
                const uint2 hi = *reinterpret_cast<const uint2*>(vp + 8);
                uint4 vv;
                vv.x = lo.x; vv.y = lo.y; vv.z = hi.x; vv.w = hi.y;
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf, o[d], 0, 0, 0);
            }
        }
    }

    if (q < p.Tq) {
        const float inv = 1.0f / l_run;
        unsigned short* op = Og + ((size_t)b * p.Tq + q) * p.ldo + head * DK;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 t;
                t.x = pack2(o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv);
                t.y = pack2(o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv);
                *reinterpret_cast<uint2*>(op + d * 32 + 8 * g + 4 * hh) = t;
            }
    }
}

}  // namespace

int launch_attention_bf16(const AttnArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.B > 0 && a.H > 0 && a.Tq > 0 && a.Tk > 0, "attention: empty problem");
    PF_REQUIRE(a.ldq % 8 == 0 && a.ldk % 8 == 0 && a.ldv % 8 == 0 && a.ldo % 4 == 0, "attention_bf16: strides % 8");
    PF_REQUIRE(a.K2 == nullptr, "attention_bf16: the two-source (streaming) form is fp32 only");
    dim3 grid(ceil_div(a.Tq, 128), a.H, a.B);
    hipLaunchKernelGGL(attention_bf16_kernel, grid, dim3(256), 0, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pf
