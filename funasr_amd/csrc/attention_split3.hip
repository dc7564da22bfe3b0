// Flash-style scaled-dot-product attention with fp32 inputs and fp32-class results on the bf16 matrix cores (d_k = 128):
// both products run as six v_mfma_f32_32x32x16_bf16 per operand pair on operands split into three bf16 planes
// (x = hi + mid + lo exactly, see gemm_split3.hip), fp32 accumulators and softmax statistics. The mode-"bf16x3" twin of
// attention_f32.hip: same inputs (the fp32 Q/K/V projections), same masking semantics, same swapped-product mapping.
//
// Reference semantics: funasr/models/sanm/attention.py:270-306,322-327 (scores, key mask -inf, softmax, mask 0, .V).
//
// One workgroup = 8 waves = 256 queries of one (sequence, head); each wave owns 32 queries x d_k, a lane owns one
// query (q = lane & 31). Per 32-key tile:
//   1. the fp32 K and V tiles arrive in LDS by asm-issued global_load_lds_dwordx4 (issued one tile ahead, in flight
//      under the previous tile's MFMAs);
//   2. split pass (all 512 threads, each element once): K -> three bf16 planes [32 keys][128 d], 16-B chunk c of key r
//      at c ^ (r & 15); V -> three TRANSPOSED planes V^T [128 d][36] (the transposition a bf16 MFMA operand needs is
//      done by this pass's 2-byte writes; the V tile's DMA image is chunk-swizzled by key so that the pass reads it
//      without bank conflicts);
//   3. S^T[key][q] = sum_d K[key][d] Q[q][d]: A = K planes (LDS), B = Q planes (registers, split once per workgroup, Q
//      pre-multiplied by d_k^-0.5 like the reference); online softmax per lane; P is split in registers;
//      O^T[d][q] += V^T[d][key] P[q][key]: the k slots of half-wave h are exactly the keys whose scores that lane's
//      accumulator registers hold, so P goes from the S^T accumulator to the B operand without leaving the lane.
// 96 MFMAs (3072 matrix-pipe cycles) per wave per tile against 8192 for the fp32 MFMA form.
#include "common.h"

namespace pf {

namespace {

constexpr int DK = 128, KT = 32, VLD = 36;
constexpr int OFF_V32 = KT * DK * 4;                     // fp32 stage: K tile [0, 16K), V tile [16K, 32K)
constexpr int OFF_KP = 2 * KT * DK * 4;                  // K planes: 3 x 8 KB
constexpr int KP_PLANE_B = KT * DK * 2;
constexpr int OFF_VT = OFF_KP + 3 * KP_PLANE_B;          // V^T planes: 3 x 128 x 36 x 2 B
constexpr int VT_PLANE_B = DK * VLD * 2;
constexpr int LDS_BYTES = OFF_VT + 3 * VT_PLANE_B;       // 84992
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512, 2) void attention_split3_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, idx = lane & 31;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q = blockIdx.x * 256 + wave * 32 + idx;
    const int qc = q < p.Tq ? q : p.Tq - 1;
    const int klen = p.klens[b];

    // ---- Q planes: step s covers d in [16 s, 16 s + 16); half-wave h holds the 8 d's of chunk 2 s + h
    bf16x8 qf[3][8];
    {
        const float* qp = p.Q + ((size_t)b * p.Tq + qc) * p.ldq + head * DK + hh * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(qp + 16 * s);
            const float4 c = *reinterpret_cast<const float4*>(qp + 16 * s + 4);
            const float v[8] = {a.x * p.scale, a.y * p.scale, a.z * p.scale, a.w * p.scale,
                                c.x * p.scale, c.y * p.scale, c.z * p.scale, c.w * p.scale};
            uint4 h, m, l;
            split3_pk(v[0], v[1], h.x, m.x, l.x);
            split3_pk(v[2], v[3], h.y, m.y, l.y);
            split3_pk(v[4], v[5], h.z, m.z, l.z);
            split3_pk(v[6], v[7], h.w, m.w, l.w);
            qf[0][s] = __builtin_bit_cast(bf16x8, h);
            qf[1][s] = __builtin_bit_cast(bf16x8, m);
            qf[2][s] = __builtin_bit_cast(bf16x8, l);
        }
    }

    floatx16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // ---- DMA of one fp32 K/V tile: 32 pieces of 1 KB (2 keys x 512 B), K pieces 0..15 then V pieces 0..15; wave w issues
    //      pieces 2w, 2w+1 of each. The V image is chunk-swizzled by key (chunk c of key r at c ^ (r & 31)).
    const float* kbase = p.K + (size_t)b * p.Tk * p.ldk + head * DK;
    const float* vbase = p.V + (size_t)b * p.Tk * p.ldv + head * DK;
    const unsigned lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    auto stage = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int piece = wave * 2 + i;
            const int r = piece * 2 + (lane >> 5);
            int kr = k0 + r;
            kr = kr < klen ? kr : klen - 1;                    // rows past the last valid key feed masked scores only
            const int cp = lane & 31;
            glds16(kbase + (size_t)kr * p.ldk + cp * 4, lds_base + piece * 1024);
            glds16(vbase + (size_t)kr * p.ldv + ((cp ^ (r & 31)) * 4), lds_base + OFF_V32 + piece * 1024);
        }
    };

    // split-pass roles
    const int kf4 = tid;                                       // K: float4 index f = tid + 512 i -> key f / 32, d = 4 (f % 32)
    const int vkey = tid & 31, vch = tid >> 5;                 // V: key, 8-wide d chunk (0..15)

    const int ntiles = (klen + KT - 1) / KT;
    stage(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int k0 = kt * KT;
        glds_wait_all();
        __syncthreads();                    // fp32 tile kt landed; every wave is done with the planes of tile kt-1

        // ---- split pass
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = kf4 + 512 * i;
            const int r = f >> 5, c4 = f & 31;
            const float4 v = *reinterpret_cast<const float4*>(smem + f * 16);
            unsigned h0, m0, l0, h1, m1, l1;
            split3_pk(v.x, v.y, h0, m0, l0);
            split3_pk(v.z, v.w, h1, m1, l1);
            unsigned char* dst = smem + OFF_KP + r * 256 + (((c4 >> 1) ^ (r & 15)) * 16) + (c4 & 1) * 8;
            *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(dst + KP_PLANE_B) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(dst + 2 * KP_PLANE_B) = make_uint2(l0, l1);
        }
        {
            const unsigned char* vrow = smem + OFF_V32 + vkey * 512;
            const float4 a = *reinterpret_cast<const float4*>(vrow + (((2 * vch) ^ vkey) * 16));
            const float4 c = *reinterpret_cast<const float4*>(vrow + (((2 * vch + 1) ^ vkey) * 16));
            const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
            unsigned short* vt = reinterpret_cast<unsigned short*>(smem + OFF_VT) + (vch * 8) * VLD + vkey;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                unsigned h, m, l;
                split3_pk(v[e], v[e + 1], h, m, l);
                vt[e * VLD] = (unsigned short)(h & 0xffffu);
                vt[(e + 1) * VLD] = (unsigned short)(h >> 16);
                vt[VT_PLANE_B / 2 + e * VLD] = (unsigned short)(m & 0xffffu);
                vt[VT_PLANE_B / 2 + (e + 1) * VLD] = (unsigned short)(m >> 16);
                vt[VT_PLANE_B + e * VLD] = (unsigned short)(l & 0xffffu);
                vt[VT_PLANE_B + (e + 1) * VLD] = (unsigned short)(l >> 16);
            }
        }
        __syncthreads();                    // planes of tile kt complete; the fp32 stage is free again
        if (kt + 1 < ntiles) stage(k0 + KT);

        // ---- S^T tile (32 keys x 32 queries), two accumulators so consecutive MFMAs do not chain
        floatx16 sa, sb;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sa[r] = 0.f; sb[r] = 0.f; }
        {
            const unsigned char* kp = smem + OFF_KP + idx * 256;
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int co = ((2 * st + hh) ^ (idx & 15)) * 16;
                const bf16x8 kh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(kp + co));
                const bf16x8 km = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(kp + KP_PLANE_B + co));
                const bf16x8 kl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(kp + 2 * KP_PLANE_B + co));
                sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qf[1][st], sa, 0, 0, 0);
                sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[2][st], sb, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qf[0][st], sa, 0, 0, 0);
                sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[1][st], sb, 0, 0, 0);
                sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qf[0][st], sa, 0, 0, 0);
                sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qf[0][st], sb, 0, 0, 0);
            }
        }

        // ---- online softmax for query (lane & 31); this lane holds keys k0 + (r&3) + 8(r>>2) + 4h
        float s[16];
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            s[r] = key < klen ? sa[r] + sb[r] : -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __expf(s[r] - m_new);
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
            uint4 ph, pm, pl;
            split3_pk(s[8 * st + 0], s[8 * st + 1], ph.x, pm.x, pl.x);
            split3_pk(s[8 * st + 2], s[8 * st + 3], ph.y, pm.y, pl.y);
            split3_pk(s[8 * st + 4], s[8 * st + 5], ph.z, pm.z, pl.z);
            split3_pk(s[8 * st + 6], s[8 * st + 7], ph.w, pm.w, pl.w);
            const bf16x8 Ph = __builtin_bit_cast(bf16x8, ph), Pm = __builtin_bit_cast(bf16x8, pm), Pl = __builtin_bit_cast(bf16x8, pl);
            bf16x8 vf[4][3];
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int pn = 0; pn < 3; ++pn) {
                    const unsigned short* vp = reinterpret_cast<const unsigned short*>(smem + OFF_VT + pn * VT_PLANE_B) +
                                               (d * 32 + idx) * VLD + 16 * st + 4 * hh;
                    const uint2 lo = *reinterpret_cast<const uint2*>(vp);
                    const uint2 hi = *reinterpret_cast<const uint2*>(vp + 8);
                    uint4 vv;
                    vv.x = lo.x; vv.y = lo.y; vv.z = hi.x; vv.w = hi.y;
                    vf[d][pn] = __builtin_bit_cast(bf16x8, vv);
                }
            // product-major: consecutive MFMAs write the four different d accumulators
#define PF_PV(PV_, PP_) _Pragma("unroll") for (int d = 0; d < 4; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[d][PV_], PP_, o[d], 0, 0, 0)
            PF_PV(1, Pm);
            PF_PV(0, Pl);
            PF_PV(2, Ph);
            PF_PV(0, Pm);
            PF_PV(1, Ph);
            PF_PV(0, Ph);
#undef PF_PV
        }
    }

    if (q < p.Tq) {
        const float inv = 1.0f / l_run;
        const size_t orow = ((size_t)b * p.Tq + q) * p.ldo + head * DK;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float t[4] = {o[d][4 * g + 0] * inv, o[d][4 * g + 1] * inv, o[d][4 * g + 2] * inv, o[d][4 * g + 3] * inv};
                const size_t off = orow + d * 32 + 8 * g + 4 * hh;
                if (p.O3) store_split3x4(p.O3 + off, p.o_plane, t);
                else *reinterpret_cast<float4*>(p.O + off) = make_float4(t[0], t[1], t[2], t[3]);
            }
    }
}

}  // namespace

int launch_attention_split3(const AttnArgs& a, hipStream_t stream) {
    PF_REQUIRE(a.B > 0 && a.H > 0 && a.Tq > 0 && a.Tk > 0, "attention: empty problem");
    PF_REQUIRE(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.ldo % 4 == 0, "attention_split3: strides % 4");
    PF_REQUIRE(((uintptr_t)a.Q & 15) == 0 && ((uintptr_t)a.K & 15) == 0 && ((uintptr_t)a.V & 15) == 0, "attention_split3: 16-B alignment");
    PF_REQUIRE(a.K2 == nullptr, "attention_split3: the two-source (streaming) form uses the fp32 MFMA kernel");
    PF_REQUIRE(a.O || a.O3, "attention: null output");
    if (a.O3) PF_REQUIRE(a.o_plane % 4 == 0 && ((uintptr_t)a.O3 & 7) == 0, "attention: plane output alignment");
    static PerDeviceOnce configured;
    if (!configured.done()) {
        PF_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_split3_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        configured.mark();
    }
    dim3 grid(ceil_div(a.Tq, 256), a.H, a.B);
    hipLaunchKernelGGL(attention_split3_kernel, grid, dim3(512), LDS_BYTES, stream, a);
    PF_HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace pf
