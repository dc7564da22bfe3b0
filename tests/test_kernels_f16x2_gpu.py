"""Kernel-level parity of the kernels that carry the headline mode (`f16x2`), one kernel per test, through the C ABI:
gemm_f16x2_kernel in every epilogue form and block shape, its full-row form with the fused LayerNorm (gemm_f16x2_row.hip),
attention_f16x2_kernel in every schedule that is kept, split2 and the plane-range machinery.

References are float64 evaluations of the reference's arithmetic -- nn.Linear (+ relu, + residuals)
funasr/models/sanm/attention.py:241-306, funasr/models/transformer/positionwise_feed_forward.py:14-34, LayerNorm
funasr/models/transformer/layer_norm.py:13-38, scaled dot-product attention funasr/models/sanm/attention.py:270-306 -- on the
values the planes actually hold (hi + lo, exact in float64), so the bars test the kernels' arithmetic and not the split.

Error model of one f16x2 product sum: an operand is hi + lo to 2^-23 of its magnitude in the worst case (hi's
11 bits + lo's 11 bits + lo's sign: 2^-24 typically), the dropped lo*lo term is <= 2^-22 |a||w|, the fp32 accumulation adds
~sqrt(K) 2^-24 of sum |a||w|; worst case 2^-21 sum |a||w|. The bar on random operands is |err| <= 4e-7 sum_k |a_k||w_k|
(+ the epilogue's fp32 roundings), the bar on adversarial operands the worst case.
"""
import math

import pytest
import torch

from funasr_amd import synth

pytestmark = pytest.mark.gpu


def _measure():
    """the measured-and-off kernel shapes (tiles 5 / 6 / 8 / 9 / 12, row heights 129 / 130, attention variants 0 / 1, the one-launch
    feed-forward, ...) exist in the measurement library only (make -C funasr_amd/csrc measure; PF_LIB_PATH=.../libparaformer_hip_measure.so):
    with it loaded these tests also check that every such shape returns the product's bits"""
    from funasr_amd import _lib
    return bool(_lib.load().pf_measurement_build())


need_measure = pytest.mark.skipif("not __import__('torch').cuda.is_available() or not __import__('funasr_amd._lib', fromlist=['x']).load().pf_measurement_build()",
                                  reason="kernel shape of the measurement library (make measure) -- not in the product")


def _planes_value(p2: torch.Tensor) -> torch.Tensor:
    """float64 value of a two-plane tensor [2, ...] (still times its power-of-two scale)"""
    return p2[0].double() + p2[1].double()


def _gemm_ref(a2, w2, scale_exp, bias=None, relu=False, add1=None, add2=None):
    a, w = _planes_value(a2), _planes_value(w2)
    out = (a @ w.T) * 2.0 ** -scale_exp
    mag = (a.abs() @ w.abs().T) * 2.0 ** -scale_exp          # sum |a||w|: the scale of the rounding errors
    if bias is not None:
        out = out + bias.double()
    if relu:
        out = torch.relu(out)
    if add1 is not None:
        out = out + add1.double()
    if add2 is not None:
        out = out + add2.double()
    return out, mag


def _check(out, ref, mag, extra=0.0, what=""):
    err = (out.double() - ref).abs()
    bound = 4e-7 * mag + 2.5e-7 * ref.abs() + extra
    worst = (err / bound).max().item()
    assert worst <= 1.0, f"{what}: error {err.max().item():.3e} is {worst:.2f}x the bound"


def _operands(M, N, K, cuda, ea=8, ew=12, seed=0):
    from funasr_amd import ops
    g = torch.Generator().manual_seed(seed + M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g).to(cuda)
    # asymmetric weights catch a transposed C write
    w = (torch.randn(N, K, generator=g) * K ** -0.5 * torch.linspace(0.5, 1.5, N)[:, None]).to(cuda)
    return ops.split2(a, ea), ops.split2(w, ew), ea + ew, g


SHAPES = [(1536, 512), (2048, 512), (512, 2048), (512, 512), (1028, 576)]       # (N, K); 1028: ragged N, narrow tile only


@pytest.mark.parametrize("M", [70, 500, 4000, 32768])
@pytest.mark.parametrize("N,K", SHAPES)
def test_gemm_f16x2_fp32_forms_vs_float64(cuda, M, N, K):
    """fp32 output with bias / relu / one or two addends, both block shapes (bitwise equal)"""
    from funasr_amd import ops
    if M == 32768 and N == 1028:
        pytest.skip("one ragged shape at the big M is enough")
    a2, w2, se, g = _operands(M, N, K, cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    r1 = torch.randn(M, N, generator=g).to(cuda)
    r2 = torch.randn(M, N, generator=g).to(cuda)
    forms = [dict(), dict(relu=True), dict(add2=r2), dict(relu=True, add1=r1, add2=r2), dict(add1=r1)]
    for kw in forms:
        ref, mag = _gemm_ref(a2, w2, se, bias, kw.get("relu", False), kw.get("add1"), kw.get("add2"))
        narrow = ops.gemm_f16x2(a2, w2, bias, scale_exp=se, tile=1, **kw)
        _check(narrow, ref, mag, what=f"256x128 {sorted(kw)}")
        if N % 256 == 0:
            wide = ops.gemm_f16x2(a2, w2, bias, scale_exp=se, tile=2, **kw)
            assert torch.equal(wide, narrow), f"256x256 and 256x128 tiles differ {sorted(kw)}"
            auto = ops.gemm_f16x2(a2, w2, bias, scale_exp=se, tile=0, **kw)
            assert torch.equal(auto, narrow)
            if _measure():
                pair = ops.gemm_f16x2(a2, w2, bias, scale_exp=se, tile=5, **kw)
                assert torch.equal(pair, narrow), f"128x256 two-workgroups-per-CU shape differs {sorted(kw)}"
                ring = ops.gemm_f16x2(a2, w2, bias, scale_exp=se, tile=6, **kw)
                assert torch.equal(ring, narrow), f"256x256 deep-ring shape differs {sorted(kw)}"
            if K >= 64:
                four = ops.gemm_f16x2(a2, w2, bias, scale_exp=se, tile=7, **kw)
                assert torch.equal(four, narrow), f"four-wave 256x256 shape (gemm_f16x2_w4.hip) differs {sorted(kw)}"
        # the persistent wave-specialised shape (gemm_f16x2_ps.hip; transposed product, stores straight from the accumulators);
        # shapes it does not take (M % 16, N % 128, both residuals) fall back by themselves: equal either way
        ps = ops.gemm_f16x2(a2, w2, bias, scale_exp=se, tile=10, **kw)
        assert torch.equal(ps, narrow), f"persistent 256x128 shape (gemm_f16x2_ps.hip) differs {sorted(kw)}"
        if kw.get("add2") is not None and kw.get("add1") is None:
            x = kw["add2"].clone()                  # in place, C == R2: the encoder's x = x + w_2(...)
            ops.gemm_f16x2(a2, w2, bias, scale_exp=se, tile=10, relu=kw.get("relu", False), add2=x, out=x)
            assert torch.equal(x, narrow), "persistent shape, in place"


@pytest.mark.parametrize("M", [15, 960, 3840])
def test_gemm_f16x2_split_k_form(cuda, M):
    """the streaming step's w_2 (K = 2048 over at most a block per CU): four K slices + one reduce launch -- fp32-class against
    float64 like the unsplit kernel, deterministic, every epilogue form"""
    from funasr_amd import ops
    N, K = 512, 2048
    a2, w2, se, g = _operands(M, N, K, cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    r1 = torch.randn(M, N, generator=g).to(cuda)
    r2 = torch.randn(M, N, generator=g).to(cuda)
    for kw in (dict(), dict(relu=True), dict(add2=r2), dict(relu=True, add1=r1, add2=r2)):
        ref, mag = _gemm_ref(a2, w2, se, bias, kw.get("relu", False), kw.get("add1"), kw.get("add2"))
        out = ops.gemm_f16x2(a2, w2, bias, scale_exp=se, split_k=True, **kw)
        _check(out, ref, mag, what=f"split-K {sorted(kw)}")
        assert torch.equal(out, ops.gemm_f16x2(a2, w2, bias, scale_exp=se, split_k=True, **kw)), "split-K form is not deterministic"
    # in place on the residual (C aliases R2), as the step uses it
    x = r2.clone()
    ops.gemm_f16x2(a2, w2, bias, scale_exp=se, split_k=True, add2=x, out=x)
    assert torch.equal(x, ops.gemm_f16x2(a2, w2, bias, scale_exp=se, split_k=True, add2=r2))


@pytest.mark.parametrize("M", [70, 4000, 32768])
def test_gemm_f16x2_plane_output(cuda, M):
    """w_1's form: relu, result written as two fp16 planes of result * 2^e (the next GEMM's operand)"""
    from funasr_amd import ops
    N, K = 2048, 512
    a2, w2, se, g = _operands(M, N, K, cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    ref, mag = _gemm_ref(a2, w2, se, bias, relu=True)
    eo = 9
    p_n = ops.gemm_f16x2(a2, w2, bias, relu=True, scale_exp=se, out_planes=True, out_scale_exp=eo, tile=1)
    p_w = ops.gemm_f16x2(a2, w2, bias, relu=True, scale_exp=se, out_planes=True, out_scale_exp=eo, tile=2)
    p_p = ops.gemm_f16x2(a2, w2, bias, relu=True, scale_exp=se, out_planes=True, out_scale_exp=eo, tile=5 if _measure() else 2)
    assert torch.equal(p_n, p_w) and torch.equal(p_p, p_w)
    # 128 x 128 four-wave shape (tile 0's pick where the 256-row shapes would leave most CUs idle, e.g. M = 70) and tile 0 itself
    assert torch.equal(ops.gemm_f16x2(a2, w2, bias, relu=True, scale_exp=se, out_planes=True, out_scale_exp=eo, tile=3), p_w)
    assert torch.equal(ops.gemm_f16x2(a2, w2, bias, relu=True, scale_exp=se, out_planes=True, out_scale_exp=eo, tile=0), p_w)
    if _measure():
        assert torch.equal(ops.gemm_f16x2(a2, w2, bias, relu=True, scale_exp=se, out_planes=True, out_scale_exp=eo, tile=6), p_w)
        assert torch.equal(ops.gemm_f16x2(a2, w2, bias, relu=True, scale_exp=se, out_planes=True, out_scale_exp=eo, tile=12), p_w), "finisher form"
    assert torch.equal(ops.gemm_f16x2(a2, w2, bias, relu=True, scale_exp=se, out_planes=True, out_scale_exp=eo, tile=7), p_w), "four-wave shape"
    assert torch.equal(ops.gemm_f16x2(a2, w2, bias, relu=True, scale_exp=se, out_planes=True, out_scale_exp=eo, tile=10), p_w), "persistent shape"
    assert torch.equal(ops.gemm_f16x2(a2, w2, bias, relu=False, scale_exp=se, out_planes=True, out_scale_exp=eo, tile=10),
                       ops.gemm_f16x2(a2, w2, bias, relu=False, scale_exp=se, out_planes=True, out_scale_exp=eo, tile=2)), "persistent shape, no ReLU"
    assert torch.isfinite(p_w.float()).all()
    val = _planes_value(p_w) * 2.0 ** -eo
    # the split adds <= 2^-22 of the element (+ the subnormal floor 2^-25 in the scaled domain)
    _check(val, ref, mag, extra=2.0 ** -22 * ref.abs() + 2.0 ** -25 * 2.0 ** -eo, what="plane output")
    # and the fp32 form of the same product equals hi + lo of the planes to the split's precision
    f32 = ops.gemm_f16x2(a2, w2, bias, relu=True, scale_exp=se, tile=2).double()
    assert ((val - f32).abs() <= 2.0 ** -22 * f32.abs() + 2.0 ** -25 * 2.0 ** -eo).all()


@pytest.mark.parametrize("M,K", [(80, 512), (512, 576), (4096, 512), (32768, 512), (22528, 512), (33536, 512)])
@pytest.mark.parametrize("kv_form", [False, True])
def test_gemm_f16x2_qkv_and_kv_forms(cuda, M, K, kv_form):
    """the fused q|k|v projection writing the attention kernel's operands: Q / K planes, fp32 V, transposed V planes whose
    columns are rows with index bits 2 and 3 swapped; the KV form (decoder memory): K planes + V^T planes"""
    from funasr_amd import ops
    D = 512
    nseg = 2 if kv_form else 3
    a2, w2, se, g = _operands(M, nseg * D, K, cuda)
    bias = torch.randn(nseg * D, generator=g).to(cuda)
    q_mul, k_mul, v_mul = 128 ** -0.5 * 2.0 ** 7, 2.0 ** 6, 2.0 ** 5
    out = ops.gemm_f16x2_qkv(a2, w2, bias, D, se, q_mul, k_mul, v_mul, kv_form=kv_form)
    pair = ops.gemm_f16x2_qkv(a2, w2, bias, D, se, q_mul, k_mul, v_mul, kv_form=kv_form, tile=5 if _measure() else 2)
    # tile 0 (above) picks by the row count: 128 x 128 blocks at M = 22 528, whole rounds of 256 x 256 blocks + a 128 x 128 tail at M = 33 536; 2: 256 x 256 only; 3: 128 x 128 only
    for tile in (2, 3):
        one = ops.gemm_f16x2_qkv(a2, w2, bias, D, se, q_mul, k_mul, v_mul, kv_form=kv_form, tile=tile)
        for key in ("q2", "k2", "v", "vt"):
            assert (out[key] is None and one[key] is None) or torch.equal(out[key], one[key]), f"tile {tile}: {key} differs"
    ring = ops.gemm_f16x2_qkv(a2, w2, bias, D, se, q_mul, k_mul, v_mul, kv_form=kv_form, tile=6 if _measure() else 3)
    four = ops.gemm_f16x2_qkv(a2, w2, bias, D, se, q_mul, k_mul, v_mul, kv_form=kv_form, tile=7)
    pers = ops.gemm_f16x2_qkv(a2, w2, bias, D, se, q_mul, k_mul, v_mul, kv_form=kv_form, tile=10)
    for key in ("q2", "k2", "v", "vt"):
        assert (out[key] is None and four[key] is None) or torch.equal(out[key], four[key]), f"four-wave shape: {key} differs"
        assert (out[key] is None and pers[key] is None) or torch.equal(out[key], pers[key]), f"persistent shape: {key} differs"
        assert (out[key] is None and pair[key] is None) or torch.equal(out[key], pair[key]), f"128x256 shape: {key} differs"
        assert (out[key] is None and ring[key] is None) or torch.equal(out[key], ring[key]), f"deep-ring shape: {key} differs"
    ref, mag = _gemm_ref(a2, w2, se, bias)
    segs = dict(k=0, v=1) if kv_form else dict(q=0, k=1, v=2)

    def seg(x, name):
        return x[:, segs[name] * D:(segs[name] + 1) * D]

    def check_planes(p2, name, mul):
        assert torch.isfinite(p2.float()).all()
        val = _planes_value(p2) / mul
        r, m = seg(ref, name), seg(mag, name)
        _check(val, r, m, extra=2.0 ** -22 * r.abs() + 2.0 ** -25 / mul, what=f"{name} planes")

    if not kv_form:
        check_planes(out["q2"][:, :M], "q", q_mul)
        _check(out["v"], seg(ref, "v"), seg(mag, "v"), what="fp32 v")
        assert not out["q2"][:, M:].any(), "rows past M of the Q planes were written"
    check_planes(out["k2"][:, :M], "k", k_mul)
    assert not out["k2"][:, M:].any()
    cols = ops.vt_columns(torch.arange(M, device=cuda))
    vt = out["vt"][:, :, cols].transpose(1, 2)                  # back to [2, M, D]
    check_planes(vt, "v", v_mul)
    touched = torch.zeros(M + 64, dtype=torch.bool, device=cuda)
    touched[cols] = True
    assert not out["vt"][:, :, ~touched].any(), "V^T columns outside the row set were written"


@pytest.mark.parametrize("M,N", [(70, 8404), (2000, 8404), (500, 25055), (11000, 8404)])
def test_gemm_f16x2_fused_argmax(cuda, M, N):
    """vocabulary / CTC projection with the arg-max in the epilogue: equals the arg-max of the float64 logits wherever their
    top-2 gap exceeds the kernel's error bound; ties go to the lowest index"""
    from funasr_amd import ops
    K = 512
    a2, w2, se, g = _operands(M, N, K, cuda, seed=5)
    bias = torch.randn(N, generator=g).to(cuda)
    ids = ops.gemm_f16x2_argmax(a2, w2, bias, scale_exp=se).long()
    ref, mag = _gemm_ref(a2, w2, se, bias)
    top2 = ref.topk(2, dim=1)
    rid = top2.indices[:, 0]
    gap = top2.values[:, 0] - top2.values[:, 1]
    tol = 2 * (4e-7 * mag.max(dim=1).values + 2.5e-7 * top2.values[:, 0].abs())
    clear = gap > tol
    assert clear.float().mean() > 0.99
    assert torch.equal(ids[clear], rid[clear])
    # where the gap is inside the bound, the kernel's pick must still be within the bound of the maximum
    picked = ref.gather(1, ids[:, None])[:, 0]
    assert ((top2.values[:, 0] - picked) <= tol).all()
    # exact ties -> lowest index (torch.argmax semantics): duplicate weight rows
    w2t = w2.clone()
    w2t[:, 7] = w2t[:, 3]
    bt = bias.clone(); bt[7] = bt[3] = 50.0
    ids_t = ops.gemm_f16x2_argmax(a2, w2t, bt, scale_exp=se)
    assert (ids_t == 3).all()


@pytest.mark.parametrize("M", [70, 500, 4000, 32768])
@pytest.mark.parametrize("K,r1,r2", [(512, True, True), (512, True, False), (2048, False, True), (512, False, False)])
def test_row_form_is_bitwise_gemm_then_layernorm(cuda, M, K, r1, r2):
    """gemm_f16x2_row.hip (128 x 512 full-row tile; residual adds + LayerNorm + plane split in the epilogue) returns the bits of
    gemm_f16x2_kernel followed by layernorm_kernel: linear_out / w_2 and the LayerNorm behind them
    (funasr/models/sanm/encoder.py:120-146, transformer/layer_norm.py:13-38)"""
    from funasr_amd import ops
    a2, w2, se, g = _operands(M, 512, K, cuda, seed=9)
    bias = torch.randn(512, generator=g).to(cuda)
    add1 = (torch.randn(M, 512, generator=g) * 3).to(cuda) if r1 else None
    add2 = (torch.randn(M, 512, generator=g) * 10 + 2).to(cuda) if r2 else None
    gamma = (torch.rand(512, generator=g) * 2 + 0.1).to(cuda)
    beta = torch.randn(512, generator=g).to(cuda)
    eps, ey = 1e-12, 7
    c_ref = ops.gemm_f16x2(a2, w2, bias, add1=add1, add2=add2, scale_exp=se, tile=2)
    y_ref = ops.layernorm_planes(c_ref, gamma, beta, eps, scale_exp=ey)
    yf_ref = ops.layernorm(c_ref, gamma, beta, eps)
    for nt in (False, True):
        c, y = ops.gemm_f16x2_row(a2, w2, bias, add1=add1, add2=add2, scale_exp=se, ln=(gamma, beta, eps), out_scale_exp=ey, a_nt=nt)
        assert torch.equal(c, c_ref), "fp32 result of the row form differs from gemm_f16x2"
        assert torch.equal(y, y_ref), "LayerNorm planes of the row form differ from layernorm_kernel"
    c, yf = ops.gemm_f16x2_row(a2, w2, bias, add1=add1, add2=add2, scale_exp=se, ln=(gamma, beta, eps), ln_planes=False)
    assert torch.equal(c, c_ref) and torch.equal(yf, yf_ref)
    c_only, none = ops.gemm_f16x2_row(a2, w2, bias, add1=add1, add2=add2, scale_exp=se)
    assert none is None and torch.equal(c_only, c_ref)
    # every block height (0: chosen by the row count; 128: the 2 x 4-wave kernel; 96 / 129: the 1 x 8-wave kernel) gives the same bits
    for br in ((0, 128, 96, 129, 130) if _measure() else (0, 128, 96)):             # 129 / 130 (measurement library): the 1 x 8 grid at 128 rows, the four-wave 1 x 4 grid
        c, y = ops.gemm_f16x2_row(a2, w2, bias, add1=add1, add2=add2, scale_exp=se, ln=(gamma, beta, eps), out_scale_exp=ey,
                                  block_rows=br, a_nt=(br == 96))
        assert torch.equal(c, c_ref) and torch.equal(y, y_ref), f"block_rows {br}"
    if r2 and not r1:
        for br in ((96, 129, 130) if _measure() else (96,)):
            c, none = ops.gemm_f16x2_row(a2, w2, bias, add2=add2, scale_exp=se, block_rows=br)
            assert none is None and torch.equal(c, c_ref), f"block_rows {br} without LayerNorm"
            nc, yf = ops.gemm_f16x2_row(a2, w2, bias, add2=add2, scale_exp=se, ln=(gamma, beta, eps), ln_planes=False, want_c=False,
                                        block_rows=br)
            assert nc is None and torch.equal(yf, yf_ref)
    nc, y_only = ops.gemm_f16x2_row(a2, w2, bias, add1=add1, add2=add2, scale_exp=se, ln=(gamma, beta, eps), out_scale_exp=ey, want_c=False)
    assert nc is None and torch.equal(y_only, y_ref)
    # in place: C aliases the second addend (the encoder's residual stream)
    if r2:
        x = add2.clone()
        lib_c, y_ip = _row_in_place(ops, a2, w2, bias, add1, x, se, gamma, beta, eps, ey)
        assert torch.equal(lib_c, c_ref) and torch.equal(y_ip, y_ref)
    # and against float64 LayerNorm of the float64 product (the reference's arithmetic, not our kernels')
    ref, mag = _gemm_ref(a2, w2, se, bias, add1=add1, add2=add2)
    ln64 = torch.nn.functional.layer_norm(ref, (512,), gamma.double(), beta.double(), eps)
    got = _planes_value(y) * 2.0 ** -ey
    assert (got - ln64).abs().max().item() < 2e-5 * max(1.0, ln64.abs().max().item())


@pytest.mark.parametrize("slots,lens", [
    ([64, 64, 64], [64, 50, 1]),                       # padded layout: equal 16-aligned slots
    ([16, 112, 48, 704, 16], [3, 100, 48, 690, 16]),   # packed layout: ragged slots, a full one, one shorter than the halo
    ([2048] * 16, [2048, 2047, 1500, 17, 16, 15, 6, 5, 4, 1, 2000, 1999, 1024, 1023, 640, 128]),
])
@pytest.mark.parametrize("r2", [True, False])
def test_row_form_with_fsmn_in_the_epilogue_is_bitwise_fsmn_then_row(cuda, slots, lens, r2):
    """FSMN form of gemm_f16x2_row.hip: the memory block (funasr/models/sanm/attention.py:216-239: 11-tap depthwise conv over
    the masked v rows + the masked rows themselves, masked again) computed in linear_out's epilogue returns the bits of
    fsmn_kernel followed by the row kernel with the memory as first addend. Rows behind a sequence's length hold garbage."""
    from funasr_amd import ops
    M = sum(slots)
    a2, w2, se, g = _operands(M, 512, 512, cuda, seed=21)
    bias = torch.randn(512, generator=g).to(cuda)
    add2 = (torch.randn(M, 512, generator=g) * 10 + 2).to(cuda) if r2 else None
    gamma = (torch.rand(512, generator=g) * 2 + 0.1).to(cuda)
    beta = torch.randn(512, generator=g).to(cuda)
    taps = (torch.randn(512, 11, generator=g) * 0.3).to(cuda)
    v = (torch.randn(M, 512, generator=g) * 2).to(cuda)
    eps, ey = 1e-12, 7
    mem = torch.empty(M, 512, device=cuda)
    lo = torch.empty(M // 16, dtype=torch.int32)
    hi = torch.empty(M // 16, dtype=torch.int32)
    start = 0
    for rows, n in zip(slots, lens):
        v[start + n:start + rows] = 3.0e30                                    # must be masked, as input and as output
        one = ops.fsmn(v[start:start + rows].view(1, rows, 512), taps, torch.tensor([n], dtype=torch.int32, device=cuda), 5)
        mem[start:start + rows] = one[0]
        lo[start // 16:(start + rows) // 16] = start
        hi[start // 16:(start + rows) // 16] = start + n
        start += rows
    assert mem.abs().max().item() < 1e6 and all(mem[sum(slots[:i]) + n:sum(slots[:i + 1])].abs().max().item() == 0
                                                for i, n in enumerate(lens) if n < slots[i])
    c_ref, y_ref = ops.gemm_f16x2_row(a2, w2, bias, add1=mem, add2=add2, scale_exp=se, ln=(gamma, beta, eps), out_scale_exp=ey)
    lo, hi = lo.to(cuda), hi.to(cuda)
    for nt in (False, True):
        c, y = ops.gemm_f16x2_row_fsmn(a2, w2, bias, v, taps, lo, hi, add2=add2, scale_exp=se, ln=(gamma, beta, eps),
                                       out_scale_exp=ey, a_nt=nt)
        assert torch.equal(c, c_ref), "fp32 result of the FSMN form differs from fsmn_kernel + row kernel"
        assert torch.equal(y, y_ref), "LayerNorm planes of the FSMN form differ"
    nc, yf = ops.gemm_f16x2_row_fsmn(a2, w2, bias, v, taps, lo, hi, add2=add2, scale_exp=se, ln=(gamma, beta, eps), ln_planes=False,
                                     want_c=False)
    assert nc is None and torch.equal(yf, ops.layernorm(c_ref, gamma, beta, eps))
    for br in ((0, 128, 96, 129, 130) if _measure() else (0, 128, 96)):
        c, y = ops.gemm_f16x2_row_fsmn(a2, w2, bias, v, taps, lo, hi, add2=add2, scale_exp=se, ln=(gamma, beta, eps),
                                       out_scale_exp=ey, block_rows=br)
        assert torch.equal(c, c_ref) and torch.equal(y, y_ref), f"FSMN form, block_rows {br}"


@pytest.mark.parametrize("frames", [[103], [36, 103, 500], [255, 256, 257, 16]])
@pytest.mark.parametrize("packing", ["padded", "all_rows", 1])
def test_encoder_schedule_options_are_bitwise_equal(cuda, frames, packing):
    """fuse_row / fsmn_fused only move work between launches: the encoder output has the same bits with either setting, in the
    padded and the packed row layouts (SANMEncoder.forward, funasr/models/sanm/encoder.py:374-451)"""
    from funasr_amd import synth
    from funasr_amd.sanm_encoder import SANMEncoder
    ec = dict(synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=3)["encoder"])
    enc = SANMEncoder(**ec, input_layer="pe")
    enc.load_state_dict(synth.encoder_state_dict(ec, seed=11), strict=False)
    enc = enc.to(cuda).set_precision("f16x2")
    enc.set_row_packing({"padded": None, "all_rows": SANMEncoder.ALL_ROWS}.get(packing, packing))
    g = torch.Generator().manual_seed(sum(frames))
    feats = (torch.randn(len(frames), max(frames), 560, generator=g) * 0.8).to(cuda)
    lens = torch.tensor(frames, dtype=torch.int32)
    outs = {}
    # ffn_fused: 0 the w_1 -> w_2 pair, 2 the one-launch feed-forward (gemm_f16x2_ffn.hip) whatever the row count, 1 by the row count
    # (the test model's few rows always take w_2's full-row form: w2_row = 2 is the default choice by row count)
    measure = _measure()                  # row_bm / ffn_fused / w2_tile select shapes of the measurement library (make measure)
    combos = ((0, 0, 0, 0), (1, 0, 128, 0), (1, 1, 128, 0), (1, 1, 96, 0), (1, 0, 96, 0), (1, 1, 129, 0), (1, 1, 0, 0), (1, 1, 0, 2), (1, 0, 128, 2),
              (1, 1, 0, 1), (1, 1, 130, 0), (1, 0, 130, 0)) if measure else ((0, 0, 0, 0), (1, 0, 0, 0), (1, 1, 0, 0), (0, 1, 0, 0))
    for fuse_row, fsmn_fused, row_bm, ffn_fused in combos:
        enc.set_option("fuse_row", fuse_row).set_option("fsmn_fused", fsmn_fused)
        if measure:
            enc.set_option("row_bm", row_bm).set_option("ffn_fused", ffn_fused)
        outs[(fuse_row, fsmn_fused, row_bm, ffn_fused)] = enc(feats, lens)[0].clone()
    # w_2 as a tile GEMM + its own LayerNorm launch (w2_row 0) against the full-row form (1), the four-wave (gemm_tile 7) and the
    # persistent (10) GEMM shapes
    enc.set_option("fuse_row", 1).set_option("fsmn_fused", 1)
    if measure:
        enc.set_option("row_bm", 0).set_option("ffn_fused", 0)
    for w2_row, gemm_tile, w2_tile in ((0, 0, 0), (1, 0, 0), (0, 7, 7), (1, 7, 7), (2, 7, 0), (0, 0, 7), (0, 0, 2), (0, 10, 10), (2, 10, 7), (0, 0, 10), (0, 2, 7), (2, 2, 7)):
        if not measure and w2_tile != 7:
            continue                               # (w2_tile: measurement library; the product runs w_2's tile form on the four-wave shape)
        enc.set_option("w2_row", w2_row).set_option("gemm_tile", gemm_tile)
        if measure:
            enc.set_option("w2_tile", w2_tile)
        outs[("w2_row", w2_row, "gemm_tile", gemm_tile, "w2_tile", w2_tile)] = enc(feats, lens)[0].clone()
    enc.set_option("w2_row", 2).set_option("gemm_tile", 0)
    if measure:
        enc.set_option("w2_tile", 7)
    base = outs[(0, 0, 0, 0)]
    assert torch.isfinite(base).all() and base.abs().max().item() > 0.1
    for key, out in outs.items():
        assert torch.equal(out, base), f"(fuse_row, fsmn_fused, row_bm, ffn_fused) = {key} changes the encoder's bits"


def _row_in_place(ops, a2, w2, bias, add1, x, se, gamma, beta, eps, ey):
    import ctypes as C
    from funasr_amd import _lib
    lib = _lib.load()
    _, M, K = a2.shape
    y2 = torch.empty(2, M, 512, device=a2.device, dtype=torch.float16)
    ms = C.c_float(0)
    p = lambda t: None if t is None else t.data_ptr()
    _lib.check(lib.pf_k_gemm_f16x2_row(p(a2), K, M * K, p(w2), K, 512 * K, float(2.0 ** -se), p(bias), p(add1), 512 if add1 is not None else 0,
                                       p(x), 512, p(x), 512, p(gamma), p(beta), float(eps), p(y2), M * 512, float(2.0 ** ey), None,
                                       M, K, 0, 0, 0, C.byref(ms), torch.cuda.current_stream().cuda_stream), "pf_k_gemm_f16x2_row")
    return x, y2


def _attn_ref(q, k, v, klens, H, scale):
    B, Tq, D = q.shape
    Tk = k.shape[1]
    dk = D // H
    qh = q.double().view(B, Tq, H, dk).transpose(1, 2) * scale
    kh = k.double().view(B, Tk, H, dk).transpose(1, 2)
    vh = v.double().view(B, Tk, H, dk).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    mask = torch.arange(Tk, device=q.device)[None, :] >= klens.to(q.device)[:, None]
    s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    p = torch.softmax(s, dim=-1).masked_fill(mask[:, None, None, :], 0.0)
    return (p @ vh).transpose(1, 2).reshape(B, Tq, D)


@pytest.mark.parametrize("B,Tq,Tk,H", [(3, 48, 48, 4), (2, 512, 512, 4), (3, 704, 704, 4), (4, 120, 512, 4), (2, 300, 208, 2)])
def test_attention_f16x2_variants_vs_float64(cuda, B, Tq, Tk, H):
    """every schedule kept in attention_f16x2.hip (0 plain, 1 pipelined, 3 lazy rescale), XCD-aware and plain workgroup order,
    self- and cross-attention shapes, ragged key lengths: <= 2e-6 of the value range against float64 attention; 0 == 1 bitwise;
    the workgroup order moves no bit"""
    from funasr_amd import ops
    D = 128 * H
    g = torch.Generator().manual_seed(B * 1000 + Tq + Tk)
    q = torch.randn(B, Tq, D, generator=g).to(cuda)
    k = torch.randn(B, Tk, D, generator=g).to(cuda)
    v = torch.randn(B, Tk, D, generator=g).to(cuda)
    # a few large scores: running-maximum moves late in the key range (the lazy variant's stale-maximum path)
    k[:, Tk // 2:, :] *= 1.5
    q[0, :, :128] *= 3.0
    klens = torch.tensor([Tk] + [max(1, Tk - 17 * (i + 1) - (5 if i % 2 else 0)) for i in range(B - 1)], dtype=torch.int32, device=cuda)
    scale = 128 ** -0.5
    ref = _attn_ref(q, k, v, klens, H, scale)
    # operand ranges: |q scale| * 2^eq < 2^15 etc.
    eq, ek, ev = 9, 10, 11
    outs = {}
    for variant in ((0, 1, 3) if _measure() else (3,)):      # 3 (lazy rescale) is the product's schedule; 0 / 1: measurement library
        for plain in (0, 16):
            o = ops.attention_f16x2(q, k, v, klens, H, scale, eq=eq, ek=ek, ev=ev, variant=variant + plain)
            assert torch.isfinite(o).all()
            err = (o.double() - ref).abs().max().item()
            assert err < 2e-6 * max(1.0, ref.abs().max().item()), f"variant {variant} order {plain}: {err:.3e}"
            outs[(variant, plain)] = o
        assert torch.equal(outs[(variant, 0)], outs[(variant, 16)]), f"variant {variant}: the workgroup order changed the result"
    if (0, 0) in outs:
        assert torch.equal(outs[(0, 0)], outs[(1, 0)]), "pipelined schedule is not bitwise the plain one"


def test_split2_planes_and_subnormal_floor(cuda):
    """x 2^e = hi + lo to 2^-23 relative in the worst case (x at the bottom of hi's binade, lo at the top of its own), or to
    the fp16 subnormal step once lo underflows; no plane overflows at the bound"""
    from funasr_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(257, 520, generator=g)
    x[0, 0], x[0, 1], x[1, 0] = 1.9999, -1.9999, 2.0 ** -30
    x = x.clamp(-1.9999, 1.9999).to(cuda)
    e = 14                                                       # bound 2: 2 * 2^14 = 2^15
    p = ops.split2(x, e, kpad=32)
    assert p.shape == (2, 257, 544) and torch.isfinite(p.float()).all()
    assert not p[:, :, 520:].any()
    val = _planes_value(p)[:, :520] * 2.0 ** -e
    err = (val - x.double()).abs()
    assert (err <= 2.0 ** -23 * x.double().abs() + 2.0 ** -25 * 2.0 ** -e).all()
    assert (err / x.double().abs().clamp_min(1e-3)).mean().item() < 2.0 ** -25       # typical: far below the worst case


@pytest.mark.parametrize("K,N", [(512, 2048), (2048, 512)])
def test_f16x2_range_adversarial_scales(cuda, K, N):
    """The a-priori plane exponents (engine_encoder.hip: LayerNorm output <= sqrt(D) max|gamma| + max|beta|; Linear <= b max_n sum_k |W| +
    |c|) on trained-scale parameters: gamma = 30, weights x 100. No plane overflows when every input sits AT the bound, and
    inputs at 2^-20 of the bound keep the error inside the split's floor: |err| <= 2^-21 sum |a||w| (both operands' 2^-23 plus
    the dropped lo*lo term) + the subnormal step of each operand's scaled domain."""
    from funasr_amd import ops
    M = 300
    g = torch.Generator().manual_seed(K)
    gamma, beta_max = 30.0, 4.0
    bound_a = math.sqrt(K) * gamma + beta_max                   # LayerNorm bound over K columns
    ea = math.floor(math.log2(32768.0 / bound_a))
    w = (torch.randn(N, K, generator=g) * K ** -0.5 * 100.0).to(cuda)
    ew = 14 - math.floor(math.log2(w.abs().max().item()))
    bias = (torch.randn(N, generator=g) * 100).to(cuda)
    w2 = ops.split2(w, ew)
    assert torch.isfinite(w2.float()).all() and w2[0].float().abs().max().item() < 32768.0
    bound_out = bound_a * w.abs().sum(dim=1).max().item() + bias.abs().max().item()
    eo = math.floor(math.log2(32768.0 / bound_out))
    for frac, name in ((1.0, "at the bound"), (2.0 ** -20, "2^-20 of the bound"), (0.37, "mixed")):
        a = torch.full((M, K), bound_a * frac)
        a *= torch.where(torch.rand(M, K, generator=g) < 0.5, -1.0, 1.0)
        if name == "mixed":
            a *= torch.rand(M, K, generator=g) * 2.0 ** -torch.randint(0, 24, (M, K), generator=g).float()
        a = a.to(cuda)
        a2 = ops.split2(a, ea)
        assert torch.isfinite(a2.float()).all(), f"A planes overflow {name}"
        out = ops.gemm_f16x2(a2, w2, bias, scale_exp=ea + ew, tile=1)
        planes = ops.gemm_f16x2(a2, w2, bias, scale_exp=ea + ew, out_planes=True, out_scale_exp=eo, tile=1)
        assert torch.isfinite(out).all() and torch.isfinite(planes.float()).all(), f"overflow {name}"
        ref = a.double() @ w.double().T + bias.double()
        mag = a.double().abs() @ w.double().abs().T
        floor = 2.0 ** -25 * (2.0 ** -ea * w.double().abs().sum(dim=1)[None, :] + 2.0 ** -ew * a.double().abs().sum(dim=1)[:, None])
        err = (out.double() - ref).abs()
        bound = 2.0 ** -21 * mag + 1.5 * floor + 2.5e-7 * ref.abs() + 1e-7 * bias.abs().max().item()
        worst = (err / bound).max().item()
        assert worst <= 1.0, f"{name}: error is {worst:.2f}x the bound (max {err.max().item():.3e})"
        val = _planes_value(planes) * 2.0 ** -eo
        assert ((val - out.double()).abs() <= 2.0 ** -22 * out.double().abs() + 2.0 ** -25 * 2.0 ** -eo).all()


def test_layernorm_planes_gamma30_no_overflow(cuda):
    """LayerNorm writing planes at the exponent the engine derives from gamma = 30 / beta = 4: finite, <= 2^15, split-accurate"""
    from funasr_amd import ops
    D, M = 512, 999
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(M, D, generator=g) * 50 + 7).to(cuda)
    x[0] = 0.0; x[0, 5] = 1e4                                    # one-hot row: a normalised value near sqrt(D)
    gamma = torch.full((D,), 30.0); gamma[::3] = -30.0
    beta = torch.full((D,), 4.0)
    e = math.floor(math.log2(32768.0 / (math.sqrt(D) * 30.0 + 4.0)))
    p = ops.layernorm_planes(x, gamma.to(cuda), beta.to(cuda), 1e-12, scale_exp=e)
    assert torch.isfinite(p.float()).all() and p[0].float().abs().max().item() <= 32768.0
    ref = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double().to(cuda), beta.double().to(cuda), 1e-12)
    val = _planes_value(p) * 2.0 ** -e
    assert (val - ref).abs().max().item() < 3e-6 * ref.abs().max().item()


def test_ctc_planes_follow_a_weight_reload(cuda):
    """pf_ctc_set_tensor drops the cached weight planes of the f16x2 arg-max route: after a second load_state_dict the ids are
    those of the NEW weights (compared with the fp32 route)"""
    from funasr_amd.ctc import CTC
    g = torch.Generator().manual_seed(0)
    ctc = CTC(odim=700, encoder_output_size=512).to(cuda)
    h = torch.randn(2, 40, 512, generator=g).to(cuda)
    lens = torch.tensor([40, 33])
    ids = {}
    for rnd in range(2):
        sd = {"ctc_lo.weight": torch.randn(700, 512, generator=g) * 0.05, "ctc_lo.bias": torch.randn(700, generator=g)}
        ctc.load_state_dict(sd)
        ctc.to(cuda)
        for mode in ("f16x2", "fp32"):
            ctc.set_precision(mode)
            ids[(rnd, mode)] = ctc.argmax(h).cpu()
        agree = (ids[(rnd, "f16x2")] == ids[(rnd, "fp32")]).float().mean().item()
        assert agree > 0.98, f"load {rnd}: f16x2 ids disagree with fp32 ids ({agree:.3f})"
    assert not torch.equal(ids[(0, "fp32")], ids[(1, "fp32")])


def _ffn_case(M, F, cuda, seed=0):
    from funasr_amd import ops
    g = torch.Generator().manual_seed(900 + seed + M + F)
    e_x, e_w1, e_h, e_w2 = 8, 12, 6, 12
    xn = torch.randn(M, 512, generator=g)                                              # a LayerNorm output
    w1 = torch.randn(F, 512, generator=g) * 512 ** -0.5 * torch.linspace(0.5, 1.5, F)[:, None]
    w2 = torch.randn(512, F, generator=g) * F ** -0.5 * torch.linspace(1.5, 0.5, 512)[:, None]
    b1 = torch.randn(F, generator=g) * 0.3
    b2 = torch.randn(512, generator=g)
    resid = torch.randn(M, 512, generator=g) * 4 + 1
    gamma = torch.rand(512, generator=g) * 2 + 0.1
    beta = torch.randn(512, generator=g)
    t = dict(x2=ops.split2(xn.to(cuda), e_x), w1=ops.split2(w1.to(cuda), e_w1), w2=ops.split2(w2.to(cuda), e_w2), b1=b1.to(cuda),
             b2=b2.to(cuda), resid=resid.to(cuda), gamma=gamma.to(cuda), beta=beta.to(cuda), e=(e_x, e_w1, e_h, e_w2))
    return t


@need_measure
@pytest.mark.parametrize("M", [70, 128, 500, 4000, 32768])
@pytest.mark.parametrize("F", [2048, 1024])
def test_fused_ffn_vs_the_two_kernel_pair_and_float64(cuda, M, F):
    """gemm_f16x2_ffn.hip: x + w_2(relu(w_1 xn)) and the next LayerNorm in one launch, the hidden activations in registers
    (funasr/models/transformer/positionwise_feed_forward.py:14-34, sanm/encoder.py:141-146) against (a) the pair
    gemm_f16x2 (plane output) -> gemm_f16x2_row it replaces: bitwise where the matrix core sums a swapped-operand product
    identically, else within 2^-22 of the result's scale; (b) float64 with the hidden planes' own rounding."""
    from funasr_amd import ops
    t = _ffn_case(M, F, cuda)
    e_x, e_w1, e_h, e_w2 = t["e"]
    eps, ey = 1e-12, 7
    # (a) the pair
    h2 = ops.gemm_f16x2(t["x2"], t["w1"], t["b1"], relu=True, scale_exp=e_x + e_w1, out_planes=True, out_scale_exp=e_h)
    c_ref, y_ref = ops.gemm_f16x2_row(h2, t["w2"], t["b2"], add2=t["resid"], scale_exp=e_h + e_w2, ln=(t["gamma"], t["beta"], eps),
                                      out_scale_exp=ey)
    c, y = ops.ffn_f16x2(t["x2"], t["w1"], t["w2"], t["b1"], t["b2"], t["resid"], e_x, e_w1, e_h, e_w2, ln=(t["gamma"], t["beta"], eps),
                         out_scale_exp=ey)
    scale = c_ref.abs().max().item()
    d = (c - c_ref).abs().max().item()
    bitwise = torch.equal(c, c_ref) and torch.equal(y, y_ref)
    yv, yr = _planes_value(y) * 2.0 ** -ey, _planes_value(y_ref) * 2.0 ** -ey
    print(f"[M={M} F={F}] fused vs pair: fp32 stream max |d| {d:.3e} (scale {scale:.2f}), LayerNorm planes max |d| "
          f"{(yv - yr).abs().max().item():.3e}, bitwise {bitwise}")
    # the block-level choice between the pair and the fused launch depends on the batch's row count (engine_internal.h ffn_fills_rounds),
    # so a clip's bits must not depend on it: BITWISE, stream and planes
    assert bitwise
    # (b) float64 of what the kernel is specified to compute: hidden planes rounded like w_1's plane epilogue
    x64, w164, w264 = _planes_value(t["x2"]), _planes_value(t["w1"]), _planes_value(t["w2"])
    h64 = torch.relu((x64 @ w164.T) * 2.0 ** -(e_x + e_w1) + t["b1"].double())
    mag1 = (x64.abs() @ w164.abs().T) * 2.0 ** -(e_x + e_w1)
    ref = (h64 @ w264.T) * 2.0 ** -e_w2 + t["b2"].double() + t["resid"].double()
    mag2 = (h64.abs() @ w264.abs().T) * 2.0 ** -e_w2 + (mag1 @ w264.abs().T) * 2.0 ** -e_w2
    tol = 4e-7 * mag2 + 2.5e-7 * ref.abs() + 1e-6
    assert ((c.double() - ref).abs() <= tol).all(), float(((c.double() - ref).abs() / tol).max())
    ln64 = torch.nn.functional.layer_norm(ref, (512,), t["gamma"].double(), t["beta"].double(), eps)
    assert (yv - ln64).abs().max().item() < 2e-5 * max(1.0, ln64.abs().max().item())
    # forms: fp32 LayerNorm output, no LayerNorm, in place, no b2
    c2, yf = ops.ffn_f16x2(t["x2"], t["w1"], t["w2"], t["b1"], t["b2"], t["resid"], e_x, e_w1, e_h, e_w2, ln=(t["gamma"], t["beta"], eps),
                           ln_planes=False)
    assert torch.equal(c2, c) and (yf.double() - ln64).abs().max().item() < 2e-5 * max(1.0, ln64.abs().max().item())
    c3, none = ops.ffn_f16x2(t["x2"], t["w1"], t["w2"], t["b1"], t["b2"], t["resid"], e_x, e_w1, e_h, e_w2)
    assert none is None and torch.equal(c3, c)
    x_ip = t["resid"].clone()
    c4, y4 = ops.ffn_f16x2(t["x2"], t["w1"], t["w2"], t["b1"], t["b2"], x_ip, e_x, e_w1, e_h, e_w2, ln=(t["gamma"], t["beta"], eps),
                           out_scale_exp=ey, in_place=True)
    assert c4 is x_ip and torch.equal(x_ip, c) and torch.equal(y4, y)
    c5, _ = ops.ffn_f16x2(t["x2"], t["w1"], t["w2"], t["b1"], None, t["resid"], e_x, e_w1, e_h, e_w2)
    assert ((c5.double() + t["b2"].double() - ref).abs() <= tol + 1e-6).all()


@need_measure
def test_fused_ffn_rows_are_independent_and_deterministic(cuda):
    """a row's result does not depend on the batch around it (the utterance-DP premise) and repeats bit for bit"""
    from funasr_amd import ops
    t = _ffn_case(1000, 2048, cuda, seed=3)
    e = t["e"]
    c, _ = ops.ffn_f16x2(t["x2"], t["w1"], t["w2"], t["b1"], t["b2"], t["resid"], *e)
    c_again, _ = ops.ffn_f16x2(t["x2"], t["w1"], t["w2"], t["b1"], t["b2"], t["resid"], *e)
    assert torch.equal(c, c_again)
    sub = slice(130, 517)
    cs, _ = ops.ffn_f16x2(t["x2"][:, sub].contiguous(), t["w1"], t["w2"], t["b1"], t["b2"], t["resid"][sub].contiguous(), *e)
    assert torch.equal(cs, c[sub])


@pytest.mark.timeout(600)
def test_optional_kernel_schedules_are_bitwise_inside_the_full_depth_encoder(cuda):
    """The one-launch feed-forward (`ffn_fused` 2) and the deep-ring GEMM shape (`gemm_tile` 6) against the default schedule over
    the WHOLE headline encoder -- 50 blocks, 64 x 500 frames, 12 800 workgroup executions of each kernel per pass, real
    HBM / L2 / Infinity-Cache traffic. Round 4's first version of the fused kernel passed every kernel-level test above bit for
    bit and still corrupted a few workgroups per thousand HERE: its counted vmcnt waits assumed that a wave's LDS-DMA pieces
    retire in issue order, which they do not when their sources differ. Both optional schedules now wait exactly; this is the
    test that would have caught it."""
    from funasr_amd.paraformer import Paraformer
    cfg = synth.PARAFORMER_LARGE
    model = Paraformer.from_config(cfg)
    model.load_state_dict(synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS), strict=False)
    model = model.to(cuda).set_precision("f16x2")
    g = torch.Generator().manual_seed(4)
    feats = (torch.randn(64, 500, 560, generator=g) * 0.8).to(cuda)
    lens = torch.full((64,), 500, dtype=torch.int32)
    measure = _measure()
    defaults = dict(ffn_fused=0, gemm_tile=0, w2_row=2, row_bm=0, w2_tile=7) if measure else dict(gemm_tile=0, w2_row=2)
    def run(**opts):
        for k, v in {**defaults, **opts}.items():
            model.encoder.set_option(k, v)
        return model.encode(feats, lens, all_rows=True)[0].clone()
    base = run()
    assert torch.isfinite(base).all()
    # the product's alternatives: w_2's full-row form, the four-wave and the persistent GEMM shapes for every tile GEMM; with the
    # measurement library also the one-launch feed-forward, the deep ring, the four-wave row form and the persistent shape's finisher form
    cases = [dict(w2_row=1), dict(gemm_tile=7), dict(gemm_tile=10), dict(gemm_tile=10), dict(gemm_tile=2, w2_row=0)]
    if measure:
        cases += [dict(ffn_fused=2), dict(gemm_tile=6), dict(ffn_fused=2), dict(gemm_tile=6), dict(gemm_tile=7, row_bm=130),
                  dict(gemm_tile=7, w2_row=1, row_bm=130), dict(w2_tile=0), dict(gemm_tile=12), dict(gemm_tile=12, w2_tile=10)]
    for opts in cases:
        assert torch.equal(run(**opts), base), f"{opts} changes the encoder's bits at full depth"
    run()
