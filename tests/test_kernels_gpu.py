"""Kernel-level numerics of the gfx950 kernels, one kernel per test, driven through the C ABI (pf_k_*).

Each kernel is compared with a plain PyTorch fp32/fp64 CPU evaluation of the same op on seeded inputs. These are
the unit tests under the module-level parity tests (tests/test_parity_gpu.py), which use the oracle.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


@pytest.fixture(params=["tile", "small_m"])
def gemm_path(request):
    """the 128x128 tile kernel (gemm_f32.hip) and the small-M weight-streaming kernel (gemm_skinny.hip) share one
    contract; run every GEMM test through both"""
    from funasr_amd import _lib
    lib = _lib.load()
    lib.pf_set_skinny_max_m(0 if request.param == "tile" else 1 << 30)
    yield request.param
    lib.pf_set_skinny_max_m(0)


@pytest.mark.parametrize("B,T,D,taps,left", [(3, 37, 512, 3, 1), (64, 500, 512, 3, 1), (2, 5, 64, 5, 2), (1, 1, 32, 3, 1), (5, 130, 96, 2, 0)])
def test_conv1d_gathered_by_the_gemm_loads_is_bitwise_the_im2col_gemm(cuda, B, T, D, taps, left):
    """cif_conv1d (funasr/models/paraformer/cif_predictor.py:275-278) without the materialised column matrix: the same k order per
    output element, so the same bits as the im2col GEMM of rounds 1-5 -- and fp32-close to torch's conv1d."""
    from funasr_amd import ops
    g = torch.Generator().manual_seed(B * 31 + T)
    h = torch.randn(B, T, D, generator=g)
    weight = torch.randn(D, D, taps, generator=g) / math.sqrt(D * taps)                  # torch Conv1d layout [out, in, tap]
    bias = torch.randn(D, generator=g)
    w = weight.permute(0, 2, 1).reshape(D, taps * D).contiguous()                        # column tap * D + c
    pad = torch.nn.functional.pad(h, (0, 0, left, taps - 1 - left))
    col = torch.cat([pad[:, k:k + T] for k in range(taps)], dim=-1).reshape(B * T, taps * D).contiguous()
    got = ops.conv1d_gemm(h.cuda(), w.cuda(), bias.cuda(), left=left, relu=True)
    want = ops.gemm(col.cuda(), w.cuda(), bias.cuda(), relu=True)
    assert torch.equal(got, want)
    ref = torch.relu(torch.nn.functional.conv1d(pad.transpose(1, 2).double(), weight.double(), bias.double())).transpose(1, 2).reshape(B * T, D)
    assert _rel(got.cpu(), ref) < 2e-6


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 512, 512), (500, 1536, 576), (333, 8404, 512),
                                   (1000, 512, 2048), (64, 1, 64), (7, 130, 96), (15, 1536, 576), (20, 8404, 512),
                                   (960, 2048, 512)])
def test_gemm_f32_matches_fp64(cuda, gemm_path, M, N, K):
    from funasr_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    # asymmetric weights catch a transposed C write (cdna guide rule 16)
    w = torch.randn(N, K, generator=g) * torch.linspace(0.5, 1.5, N)[:, None]
    bias = torch.randn(N, generator=g)
    r1 = torch.randn(M, N, generator=g)
    r2 = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().T + bias.double()
    out = ops.gemm(a.to(cuda), w.to(cuda), bias.to(cuda)).cpu()
    assert _rel(out, ref) < 2e-6
    ref2 = torch.relu(ref) + r1.double() + r2.double()
    out2 = ops.gemm(a.to(cuda), w.to(cuda), bias.to(cuda), relu=True, add1=r1.to(cuda), add2=r2.to(cuda)).cpu()
    assert _rel(out2, ref2) < 2e-6


def test_small_m_gemm_rows_do_not_depend_on_m(cuda):
    """gemm_skinny.hip: every variant (1 / 2 / 4 row tiles, 16 waves x 1 slice or 4 waves x 4 slices) sums the 16 K slices in
    one fixed order, so a row's result is bitwise the same whatever M is -- a stream's bits do not depend on how many streams
    run beside it."""
    from funasr_amd import _lib, ops
    lib = _lib.load()
    lib.pf_set_skinny_max_m(1 << 30)
    try:
        g = torch.Generator().manual_seed(11)
        for N, K in ((1536, 576), (512, 2048), (130, 96)):
            a = torch.randn(300, K, generator=g)
            w = torch.randn(N, K, generator=g)
            bias = torch.randn(N, generator=g)
            full = ops.gemm(a.to(cuda), w.to(cuda), bias.to(cuda)).cpu()
            for M in (1, 15, 16, 20, 33, 64, 65, 200):
                part = ops.gemm(a[:M].contiguous().to(cuda), w.to(cuda), bias.to(cuda)).cpu()
                assert torch.equal(part, full[:M]), (N, K, M)
    finally:
        lib.pf_set_skinny_max_m(0)


def test_small_m_gemm_carries_a_layernorm_between_two_gemms(cuda):
    """gemm_skinny.hip, the streaming step's fused LayerNorms: the producer's epilogue leaves (sum, sum of squared deviations from the
    block mean) per row and 16-column block, the consumer merges them by Chan's formula and normalises its A operand on the fetch.
    Against float64 LayerNorm + GEMM (fp32-class: tolerance 2e-6 of the output range); the pair's rows are bitwise independent of M
    like the plain kernel's; against the stand-alone LayerNorm kernel + GEMM < 2e-6; and rows whose mean is a hundred times their
    spread (the former Q / K - mean^2 variance lost three digits there) stay fp32-class in the statistics."""
    from funasr_amd import ops
    g = torch.Generator().manual_seed(5)
    D, F, eps = 512, 2048, 1e-12
    x0 = torch.randn(200, D, generator=g) * 3 + 0.7          # a row mean that is not small against the spread
    ctx = torch.randn(200, D, generator=g)
    wo = torch.randn(D, D, generator=g) / math.sqrt(D)
    bo = torch.randn(D, generator=g)
    w1 = torch.randn(F, D, generator=g) / math.sqrt(D)
    b1 = torch.randn(F, generator=g)
    gam = torch.rand(D, generator=g) + 0.5
    bet = torch.randn(D, generator=g)
    dev = lambda t: t.to(cuda)
    full = None
    for M in (200, 1, 15, 16, 20, 33, 64, 65):
        x, st = ops.gemm_small_m_ln(dev(ctx[:M].contiguous()), dev(wo), dev(bo), add2=dev(x0[:M].contiguous()), want_stats=True)
        h, none = ops.gemm_small_m_ln(x, dev(w1), dev(b1), relu=True, stats_in=st, ln=(dev(gam), dev(bet), eps))
        assert none is None
        if full is None:
            full = (x.cpu(), st.cpu(), h.cpu())
            xr = x0.double() + (ctx.double() @ wo.double().T + bo.double())
            assert _rel(full[0], xr) < 2e-6
            blocks = full[0].double().view(200, D // 16, 16)
            assert (full[1][..., 0].double() - blocks.sum(-1)).abs().max() < 1e-4
            assert (full[1][..., 1].double() - ((blocks - blocks.mean(-1, keepdim=True)) ** 2).sum(-1)).abs().max() < 2e-4
            xn = torch.nn.functional.layer_norm(full[0].double(), (D,), gam.double(), bet.double(), eps)
            hr = torch.relu(xn @ w1.double().T + b1.double())
            assert _rel(full[2], hr) < 2e-6
            xn_k = ops.layernorm(x, dev(gam), dev(bet), eps)
            h_k, _ = ops.gemm_small_m_ln(xn_k, dev(w1), dev(b1), relu=True)
            assert _rel(full[2], h_k.cpu()) < 2e-6
            # the producer may store gamma-scaled outputs (partials unchanged); the consumer then multiplies nothing in its loop
            xg, stg = ops.gemm_small_m_ln(dev(ctx), dev(wo), dev(bo), add2=dev(x0), want_stats=True, out_gamma=dev(gam))
            assert torch.equal(stg, st) and torch.equal(xg.cpu(), (full[0] * gam))
            hg, _ = ops.gemm_small_m_ln(xg, dev(w1), dev(b1), relu=True, stats_in=stg, ln=(dev(gam), dev(bet), eps), a_has_gamma=True)
            assert _rel(hg.cpu(), hr) < 2e-6
        else:
            assert torch.equal(x.cpu(), full[0][:M]) and torch.equal(st.cpu(), full[1][:M]) and torch.equal(h.cpu(), full[2][:M]), M
        if M <= 32:
            # the four-workgroups-per-tile form (long K behind few column tiles): the same bits, with and without the LayerNorm forms
            w2 = torch.randn(D, F, generator=torch.Generator().manual_seed(M)) / math.sqrt(F)
            for kw in (dict(), dict(want_stats=True), dict(add2=x, relu=True)):
                y0, s0 = ops.gemm_small_m_ln(h, dev(w2), dev(bo), **kw)
                y1, s1 = ops.gemm_small_m_ln(h, dev(w2), dev(bo), four_workgroups=True, **kw)
                assert torch.equal(y0, y1) and (s0 is None or torch.equal(s0, s1)), (M, kw)
            x4, st4 = ops.gemm_small_m_ln(dev(ctx[:M].contiguous()), dev(wo), dev(bo), add2=dev(x0[:M].contiguous()), want_stats=True,
                                          four_workgroups=True)
            h4, _ = ops.gemm_small_m_ln(x4, dev(w1), dev(b1), relu=True, stats_in=st4, ln=(dev(gam), dev(bet), eps), four_workgroups=True)
            assert torch.equal(x4, x) and torch.equal(st4, st) and torch.equal(h4, h), M


    # rows with |mean| >> std: residual stream offset by 300 with spread ~1. The merged variance must be the two-pass one to fp32
    # rounding (the one-pass form's error here: eps * mean^2 / var ~ 5e-3 relative); the consumer's output keeps the cancellation of
    # rstd (W (gamma a) - mean c1) + c2 (inherent to carrying the LayerNorm: ~ eps * |mean| / std of the output range)
    xb = (torch.randn(64, D, generator=g) + 300.0)
    zero_w, zero_b = torch.zeros(D, D), torch.zeros(D)
    x, st = ops.gemm_small_m_ln(dev(ctx[:64].contiguous()), dev(zero_w), dev(zero_b), add2=dev(xb), want_stats=True)
    assert torch.equal(x.cpu(), xb)
    blocks = xb.double().view(64, D // 16, 16)
    m2 = st.cpu()[..., 1].double().sum(-1) + (16.0 * (blocks.mean(-1) - xb.double().mean(-1, keepdim=True)) ** 2).sum(-1)
    var_ref = xb.double().var(-1, unbiased=False)
    assert ((m2 / D - var_ref).abs() / var_ref).max().item() < 1e-5, "merged block partials are not the two-pass variance"
    h, _ = ops.gemm_small_m_ln(x, dev(w1), dev(b1), relu=True, stats_in=st, ln=(dev(gam), dev(bet), eps))
    xn = torch.nn.functional.layer_norm(xb.double(), (D,), gam.double(), bet.double(), eps)
    hr = torch.relu(xn @ w1.double().T + b1.double())
    assert _rel(h.cpu(), hr) < 3e-4


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (500, 1536, 576), (333, 2048, 512), (1000, 512, 2048), (70, 130, 192)])
def test_gemm_bf16_operands_fp32_accumulate(cuda, M, N, K):
    """bf16-operand mode: against fp64 on the SAME bf16-rounded operands only the fp32 accumulation differs."""
    from funasr_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * torch.linspace(0.5, 1.5, N)[:, None]
    bias = torch.randn(N, generator=g)
    r2 = torch.randn(M, N, generator=g)
    ab = ops.cast_bf16(a.to(cuda))
    wb = ops.cast_bf16(w.to(cuda))
    assert torch.equal(ab.cpu(), a.to(torch.bfloat16)) and torch.equal(wb.cpu(), w.to(torch.bfloat16))   # RNE like torch
    ref = ab.cpu().double() @ wb.cpu().double().T + bias.double()
    out = ops.gemm_bf16(ab, wb, bias.to(cuda)).cpu()
    assert _rel(out, ref) < 2e-6
    out2 = ops.gemm_bf16(ab, wb, bias.to(cuda), relu=True, add2=r2.to(cuda), out_bf16=True).cpu()
    ref2 = (torch.relu(ref) + r2.double())
    assert torch.equal(out2, ref2.float().to(torch.bfloat16)) or _rel(out2.float(), ref2) < 4e-3   # one bf16 rounding


def test_split3_planes_are_exact(cuda):
    """x = hi + mid + lo exactly (three bf16 planes carry fp32's 24 significand bits), padding columns are zero."""
    from funasr_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(257, 560, generator=g) * torch.pow(10.0, torch.rand(257, 560, generator=g) * 12 - 6)
    x[0, :8] = torch.tensor([0.0, 1.0, -1.0, 3.0e-5, 65504.0, 1.0e-20, -7.25, 0.1])
    p = ops.split3(x.to(cuda)).cpu()
    assert p.shape == (3, 257, 576) and torch.count_nonzero(p[:, :, 560:]) == 0
    s = (p[0, :, :560].float() + p[1, :, :560].float()) + p[2, :, :560].float()
    assert torch.equal(s, x)
    assert torch.equal(p[0, :, :560], x.to(torch.bfloat16))        # hi = round-to-nearest-even bf16


@pytest.mark.parametrize("M,N,K", [(256, 128, 32), (500, 1536, 576), (333, 2048, 512), (1000, 512, 2048), (70, 132, 192)])
def test_gemm_split3_fp32_class_accuracy(cuda, M, N, K):
    """bf16x3 operands, six MFMA products: error against fp64 is in the class of the fp32 MFMA kernel's."""
    from funasr_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * torch.linspace(0.5, 1.5, N)[:, None]
    bias = torch.randn(N, generator=g)
    r1 = torch.randn(M, N, generator=g)
    r2 = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().T + bias.double()
    a3, w3 = ops.split3(a.to(cuda)), ops.split3(w.to(cuda))
    out = ops.gemm_split3(a3, w3, bias.to(cuda)).cpu()
    f32 = ops.gemm(a.to(cuda), w.to(cuda), bias.to(cuda)).cpu()
    e3, e32 = _rel(out, ref), _rel(f32, ref)
    assert e3 < 2e-6 and e3 < 3 * e32 + 1e-7, (e3, e32)
    # epilogue forms: relu, residuals, and the plane output (its planes sum to the fp32 result bit for bit)
    out2 = ops.gemm_split3(a3, w3, bias.to(cuda), relu=True, add1=r1.to(cuda), add2=r2.to(cuda)).cpu()
    ref2 = torch.relu(ref) + r1.double() + r2.double()
    assert _rel(out2, ref2) < 2e-6
    out1 = ops.gemm_split3(a3, w3, None, add1=r1.to(cuda)).cpu()
    assert _rel(out1, (a.double() @ w.double().T) + r1.double()) < 2e-6
    o3 = ops.gemm_split3(a3, w3, bias.to(cuda), relu=True, out_planes=True).cpu()
    o = ops.gemm_split3(a3, w3, bias.to(cuda), relu=True).cpu()
    assert torch.equal((o3[0].float() + o3[1].float()) + o3[2].float(), o)


def test_gemm_split3_rows_do_not_depend_on_the_batch(cuda):
    from funasr_amd import ops
    g = torch.Generator().manual_seed(11)
    a = torch.randn(1000, 512, generator=g).to(cuda)
    w3 = ops.split3(torch.randn(1536, 512, generator=g).to(cuda))
    full = ops.gemm_split3(ops.split3(a), w3)
    part = ops.gemm_split3(ops.split3(a[300:437].contiguous()), w3)
    assert torch.equal(full[300:437], part)


def test_gemm_f32_strided_and_inplace_residual(cuda, gemm_path):
    from funasr_amd import ops
    g = torch.Generator().manual_seed(5)
    big = torch.randn(300, 1536, generator=g).to(cuda)
    a = big[:, 512:1024]            # row stride 1536
    w = torch.randn(512, 512, generator=g).to(cuda)
    x = torch.randn(300, 512, generator=g).to(cuda)
    ref = x.double().cpu() + a.double().cpu() @ w.double().cpu().T
    ops.gemm(a, w, None, add2=x, out=x)   # in-place residual update
    assert _rel(x.cpu(), ref) < 2e-6


@pytest.mark.parametrize("M,N", [(77, 8404), (300, 25055), (5, 100)])
def test_gemm_fused_argmax(cuda, M, N):
    from funasr_amd import ops
    g = torch.Generator().manual_seed(N)
    a = torch.randn(M, 512, generator=g)
    w = torch.randn(N, 512, generator=g)
    bias = torch.randn(N, generator=g)
    ids = ops.gemm_argmax(a.to(cuda), w.to(cuda), bias.to(cuda)).cpu().long()
    logits = a.double() @ w.double().T + bias.double()
    ref = logits.argmax(-1)
    # ties are measure-zero; allow an fp32-level near-tie
    bad = (ids != ref).nonzero().flatten()
    for i in bad.tolist():
        assert abs(logits[i, ids[i]] - logits[i, ref[i]]) < 1e-4 * abs(logits[i, ref[i]])
    assert len(bad) <= 1


@pytest.mark.parametrize("D,pad,eps", [(512, 512, 1e-12), (560, 576, 1e-12), (2048, 2048, 1e-12), (512, 512, 1e-5)])
def test_layernorm(cuda, D, pad, eps):
    from funasr_amd import ops
    g = torch.Generator().manual_seed(D)
    x = torch.randn(333, D, generator=g) * 3 + 0.5
    gamma = torch.randn(D, generator=g)
    beta = torch.randn(D, generator=g)
    ref = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), beta.double(), eps)
    y = ops.layernorm(x.to(cuda), gamma.to(cuda), beta.to(cuda), eps, pad_to=pad).cpu()
    assert (y[:, :D].double() - ref).abs().max().item() < 5e-6
    assert (y[:, D:] == 0).all()


@pytest.mark.parametrize("left_pad", [5, 10])
def test_fsmn(cuda, left_pad):
    from funasr_amd import ops
    g = torch.Generator().manual_seed(11 + left_pad)
    B, T, Cc, K = 3, 53, 512, 11
    qkv = torch.randn(B, T, 3 * Cc, generator=g)
    v = qkv[:, :, 2 * Cc:]
    w = torch.randn(Cc, 1, K, generator=g) * 0.3
    lens = torch.tensor([53, 20, 1], dtype=torch.int32)
    res = torch.randn(B, T, Cc, generator=g)
    mask = (torch.arange(T)[None, :] < lens[:, None]).double()[:, :, None]
    inp = v.double() * mask
    xp = torch.nn.functional.pad(inp.transpose(1, 2), (left_pad, K - 1 - left_pad))
    conv = torch.nn.functional.conv1d(xp, w.double(), groups=Cc).transpose(1, 2)
    ref = (conv + inp) * mask
    out = ops.fsmn(qkv.to(cuda)[:, :, 2 * Cc:], w.to(cuda), lens.to(cuda), left_pad).cpu()
    assert (out.double() - ref).abs().max().item() < 1e-5
    out2 = ops.fsmn(qkv.to(cuda)[:, :, 2 * Cc:], w.to(cuda), lens.to(cuda), left_pad, residual=res.to(cuda)).cpu()
    assert (out2.double() - (ref + res.double())).abs().max().item() < 1e-5


@pytest.mark.parametrize("B,Tq,Tk,lens", [(2, 83, 83, [83, 40]), (1, 500, 500, [500]), (3, 37, 150, [150, 1, 33]),
                                          (2, 130, 64, [64, 31])])
def test_attention_f32(cuda, B, Tq, Tk, lens):
    from funasr_amd import ops
    H, dk = 4, 128
    g = torch.Generator().manual_seed(Tq * 13 + Tk)
    q = torch.randn(B, Tq, H * dk, generator=g)
    kv = torch.randn(B, Tk, 2 * H * dk, generator=g)
    k, v = kv[:, :, :H * dk], kv[:, :, H * dk:]
    klens = torch.tensor(lens, dtype=torch.int32)
    scale = dk ** -0.5
    qh = q.double().view(B, Tq, H, dk).transpose(1, 2) * scale
    kh = k.double().reshape(B, Tk, H, dk).transpose(1, 2)
    vh = v.double().reshape(B, Tk, H, dk).transpose(1, 2)
    s = qh @ kh.transpose(-1, -2)
    m = (torch.arange(Tk)[None, :] >= klens[:, None])[:, None, None, :]
    p = torch.softmax(s.masked_fill(m, float("-inf")), -1).masked_fill(m, 0.0)
    ref = (p @ vh).transpose(1, 2).reshape(B, Tq, H * dk)
    kvd = kv.to(cuda)
    out = ops.attention(q.to(cuda), kvd[:, :, :H * dk], kvd[:, :, H * dk:], klens.to(cuda), H, scale).cpu()
    assert (out.double() - ref).abs().max().item() < 2e-5
    # the bf16x3 form (three-plane split operands on the bf16 MFMA) meets the same bar
    out3 = ops.attention_split3(q.to(cuda), kvd[:, :, :H * dk], kvd[:, :, H * dk:], klens.to(cuda), H, scale).cpu()
    e3, e32 = (out3.double() - ref).abs().max().item(), (out.double() - ref).abs().max().item()
    assert e3 < 2e-5 and e3 < 3 * e32 + 1e-6, (e3, e32)


@pytest.mark.parametrize("B,Tq,Tk,lens", [(1, 15, 55, [55]), (3, 15, 64, [64, 1, 17]), (2, 20, 130, [130, 65]), (2, 1, 16, [16, 5]),
                                          (4, 32, 33, [33, 32, 16, 2])])
def test_attention_f32_few_query_kernel(cuda, B, Tq, Tk, lens):
    """attention_f32_fewq_kernel (the streaming step's window / token attention: keys split over the waves, flash-decoding
    merge) against fp64 softmax attention, and against the 128-query kernel."""
    from funasr_amd import _lib, ops
    H, dk = 4, 128
    g = torch.Generator().manual_seed(Tq * 131 + Tk)
    q = torch.randn(B, Tq, H * dk, generator=g)
    kv = torch.randn(B, Tk, 2 * H * dk, generator=g)
    klens = torch.tensor(lens, dtype=torch.int32)
    scale = dk ** -0.5
    qh = q.double().view(B, Tq, H, dk).transpose(1, 2) * scale
    kh = kv[:, :, :H * dk].double().reshape(B, Tk, H, dk).transpose(1, 2)
    vh = kv[:, :, H * dk:].double().reshape(B, Tk, H, dk).transpose(1, 2)
    m = (torch.arange(Tk)[None, :] >= klens[:, None])[:, None, None, :]
    p = torch.softmax((qh @ kh.transpose(-1, -2)).masked_fill(m, float("-inf")), -1).masked_fill(m, 0.0)
    ref = (p @ vh).transpose(1, 2).reshape(B, Tq, H * dk)
    kvd = kv.to(cuda)
    tile = ops.attention(q.to(cuda), kvd[:, :, :H * dk], kvd[:, :, H * dk:], klens.to(cuda), H, scale).cpu()
    lib = _lib.load()
    lib.pf_set_skinny_max_m(1 << 30)                        # the test hook that selects the streaming step's kernels
    try:
        few = ops.attention(q.to(cuda), kvd[:, :, :H * dk], kvd[:, :, H * dk:], klens.to(cuda), H, scale).cpu()
    finally:
        lib.pf_set_skinny_max_m(0)
    assert (few.double() - ref).abs().max().item() < 2e-5
    assert (few - tile).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,Tq,Tk,lens", [(2, 100, 100, [100, 37]), (1, 500, 500, [500]), (3, 40, 300, [300, 1, 129])])
def test_attention_bf16(cuda, B, Tq, Tk, lens):
    """bf16-operand attention against fp64 softmax attention on the same bf16-rounded Q/K/V: the differences are the
    bf16 rounding of P (2^-9 relative) and of the output."""
    from funasr_amd import ops
    H, dk = 4, 128
    g = torch.Generator().manual_seed(B * 1000 + Tq)
    q = (torch.randn(B, Tq, H * dk, generator=g)).to(torch.bfloat16)
    k = (torch.randn(B, Tk, H * dk, generator=g)).to(torch.bfloat16)
    v = (torch.randn(B, Tk, H * dk, generator=g)).to(torch.bfloat16)
    kl = torch.tensor(lens, dtype=torch.int32)
    scale = dk ** -0.5
    out = ops.attention_bf16(q.to(cuda), k.to(cuda), v.to(cuda), kl.to(cuda), H, scale).cpu().double()
    qh = q.double().view(B, Tq, H, dk).transpose(1, 2) * scale
    kh = k.double().view(B, Tk, H, dk).transpose(1, 2)
    vh = v.double().view(B, Tk, H, dk).transpose(1, 2)
    sc = qh @ kh.transpose(-1, -2)
    mask = torch.arange(Tk)[None, :] >= kl[:, None]
    sc = sc.masked_fill(mask[:, None, None, :], float("-inf"))
    ref = (torch.softmax(sc, -1) @ vh).transpose(1, 2).reshape(B, Tq, H * dk)
    err = (out - ref).abs()
    assert err.max().item() < 3e-2 and err.mean().item() < 3e-3, (err.max().item(), err.mean().item())
