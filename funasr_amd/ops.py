"""Tensor-level wrappers over the single-kernel entry points (pf_k_*) of the C ABI.

torch is used only as the owner of device memory and of the current HIP stream; every function launches exactly
one gfx950 kernel from libparaformer_hip.so and raises if the library or a GPU is missing.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor in GPU memory (there is no CPU path)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    return t


def gemm(a: torch.Tensor, w: torch.Tensor, bias=None, relu=False, add1=None, add2=None, out=None):
    """out = (relu?)(a @ w.T + bias) (+ add1) (+ add2);  a [M, K] (row stride a.stride(0)), w [N, K]."""
    lib = _lib.load()
    _f32c(a, "a"), _f32c(w, "w")
    M, K = a.shape
    N = w.shape[0]
    assert a.stride(1) == 1 and w.stride(1) == 1
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    _lib.check(lib.pf_k_gemm_f32(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(bias),
                                 _ptr(add1), add1.stride(0) if add1 is not None else 0,
                                 _ptr(add2), add2.stride(0) if add2 is not None else 0,
                                 _ptr(out), out.stride(0), M, N, K, int(relu), _stream()), "pf_k_gemm_f32")
    return out


def conv1d_gemm(hidden: torch.Tensor, w: torch.Tensor, bias=None, left: int = 1, relu=False):
    """Conv1d over time as one exact-fp32 GEMM, im2col gathered by the operand loads (pf_k_conv1d_gemm_f32): hidden [B, T, D],
    w [N, taps * D] (column tap * D + c) -> [B * T, N]."""
    lib = _lib.load()
    _f32c(hidden, "hidden"), _f32c(w, "w")
    B, T, D = hidden.shape
    N, taps = w.shape[0], w.shape[1] // D
    out = torch.empty(B * T, N, device=hidden.device, dtype=torch.float32)
    zero = torch.zeros(64, device=hidden.device, dtype=torch.float32)
    _lib.check(lib.pf_k_conv1d_gemm_f32(_ptr(hidden), _ptr(w), _ptr(bias), _ptr(out), B, T, D, N, taps, left, int(relu), _ptr(zero),
                                        _stream()), "pf_k_conv1d_gemm_f32")
    return out


def gemm_small_m_ln(a: torch.Tensor, w: torch.Tensor, bias=None, relu=False, add2=None, stats_in=None, ln=None, want_stats=False,
                    four_workgroups=False, out_gamma=None, a_has_gamma=False):
    """The small-M GEMM with a LayerNorm carried between GEMMs (gemm_skinny.hip). want_stats returns the [M, N / 16, 2] block
    partials (sum, sum of squared deviations from the block mean: merged by Chan's formula in the consumer) of the finished outputs; out_gamma [N]: the stored outputs are out * out_gamma (the partials stay
    those of out). With stats_in [M, K / 16, 2] and ln = (gamma, beta, eps): out = w LayerNorm(a) + bias, evaluated as
    rstd (w (gamma a) - mean c1) + c2 with c1 = w gamma, c2 = w beta + bias (pf_k_ln_consts); a_has_gamma: `a` holds gamma a already.
    four_workgroups (M <= 32): four workgroups share a tile's sixteen K slices, the same bits. Returns (out, stats or None)."""
    lib = _lib.load()
    _f32c(a, "a"), _f32c(w, "w")
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    stats = torch.empty(M, N // 16, 2, device=a.device, dtype=torch.float32) if want_stats else None
    g, eps, cst = None, 0.0, None
    if ln is not None:
        g, b, eps = ln
        cst = torch.empty(2 * N, device=a.device, dtype=torch.float32)
        _lib.check(lib.pf_k_ln_consts(_ptr(w), w.stride(0), N, K, _ptr(g), _ptr(b), _ptr(bias), _ptr(cst), _stream()), "pf_k_ln_consts")
        if a_has_gamma:
            g = None
    ws = cnt = None
    if four_workgroups:
        tiles = ((N + 15) // 16) * ((M + (15 if M <= 16 else 31)) // (16 if M <= 16 else 32))
        ws = torch.empty(tiles * 16 * 512, device=a.device, dtype=torch.float32)
        cnt = torch.zeros(tiles, device=a.device, dtype=torch.int32)
    _lib.check(lib.pf_k_gemm_skinny_ln(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(bias),
                                       _ptr(add2), add2.stride(0) if add2 is not None else 0, _ptr(out), N, M, N, K, int(relu),
                                       _ptr(stats), _ptr(stats_in), _ptr(g), _ptr(cst), float(eps), _ptr(out_gamma), _ptr(ws), _ptr(cnt),
                                       _stream()), "pf_k_gemm_skinny_ln")
    if cnt is not None:
        assert int(cnt.abs().sum()) == 0, "the tile counters must be left at zero"
    return out, stats


def gemm_argmax(a: torch.Tensor, w: torch.Tensor, bias=None):
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    nparts = 2 * ((N + 127) // 128)
    ids = torch.empty(M, device=a.device, dtype=torch.int32)
    sval = torch.empty(M, nparts, device=a.device, dtype=torch.float32)
    sidx = torch.empty(M, nparts, device=a.device, dtype=torch.int32)
    _lib.check(lib.pf_k_gemm_argmax_f32(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(bias), M, N, K,
                                        _ptr(ids), _ptr(sval), _ptr(sidx), _stream()), "pf_k_gemm_argmax_f32")
    return ids


def log_softmax(x: torch.Tensor, inplace: bool = False) -> torch.Tensor:
    """row-wise log-softmax over the last dimension of a contiguous fp32 tensor (one wave per row)"""
    lib = _lib.load()
    _f32c(x, "x")
    assert x.is_contiguous()
    N = x.shape[-1]
    M = x.numel() // N
    y = x if inplace else torch.empty_like(x)
    _lib.check(lib.pf_k_log_softmax(_ptr(x), N, _ptr(y), N, M, N, _stream()), "pf_k_log_softmax")
    return y


def layernorm(x: torch.Tensor, gamma, beta, eps: float, pad_to: int | None = None):
    lib = _lib.load()
    _f32c(x, "x")
    M, D = x.shape
    Dpad = pad_to or D
    y = torch.empty(M, Dpad, device=x.device, dtype=torch.float32)
    _lib.check(lib.pf_k_layernorm(_ptr(x), x.stride(0), _ptr(gamma), _ptr(beta), _ptr(y), Dpad, M, D, Dpad,
                                  float(eps), _stream()), "pf_k_layernorm")
    return y


def fsmn(x: torch.Tensor, w: torch.Tensor, lens: torch.Tensor, left_pad: int, residual=None):
    """x [B, T, C] (last-dim contiguous view allowed), w [C, 1, K] or [C, K], lens int32 [B] on device."""
    lib = _lib.load()
    B, T, Cc = x.shape
    assert x.stride(2) == 1 and x.stride(0) == T * x.stride(1)
    w2 = w.reshape(Cc, -1).contiguous()
    out = torch.empty(B, T, Cc, device=x.device, dtype=torch.float32)
    _lib.check(lib.pf_k_fsmn(_ptr(x), x.stride(1), _ptr(w2), _ptr(residual), Cc if residual is not None else 0,
                             _ptr(out), Cc, _ptr(lens), B, T, Cc, w2.shape[1], left_pad, _stream()), "pf_k_fsmn")
    return out


def attention(q, k, v, klens, n_heads: int, scale: float):
    """q [B, Tq, H*128], k/v [B, Tk, H*128] (row-strided views allowed), klens int32 [B] on device."""
    lib = _lib.load()
    B, Tq, D = q.shape
    Tk = k.shape[1]
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    out = torch.empty(B, Tq, D, device=q.device, dtype=torch.float32)
    _lib.check(lib.pf_k_attention_f32(_ptr(q), q.stride(1), _ptr(k), k.stride(1), _ptr(v), v.stride(1), _ptr(out), D,
                                      _ptr(klens), B, n_heads, Tq, Tk, float(scale), _stream()), "pf_k_attention_f32")
    return out


def attention_split3(q, k, v, klens, n_heads: int, scale: float, time_iters: int = 0):
    """Same contract as attention(); both products on the bf16 MFMA from three-plane split operands (fp32-class)."""
    lib = _lib.load()
    B, Tq, D = q.shape
    Tk = k.shape[1]
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    out = torch.empty(B, Tq, D, device=q.device, dtype=torch.float32)
    def run():
        _lib.check(lib.pf_k_attention_split3(_ptr(q), q.stride(1), _ptr(k), k.stride(1), _ptr(v), v.stride(1), _ptr(out), D,
                                             _ptr(klens), B, n_heads, Tq, Tk, float(scale), _stream()), "pf_k_attention_split3")
    run()
    if time_iters <= 0:
        return out
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(time_iters):
        run()
    b.record()
    torch.cuda.synchronize()
    return out, a.elapsed_time(b) / time_iters


def attention_small(q, k, v, klens, n_heads: int, scale: float):
    """q [B, Tq, H*dk], k/v [B, Tk, H*dk] with dk <= 64 (row-strided views allowed) -> [B, Tq, H*dk]; fp32 FMAs."""
    lib = _lib.load()
    B, Tq, D = q.shape
    Tk = k.shape[1]
    assert q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1 and D % n_heads == 0
    out = torch.empty(B, Tq, D, device=q.device, dtype=torch.float32)
    _lib.check(lib.pf_k_attention_small(_ptr(q), q.stride(1), _ptr(k), k.stride(1), _ptr(v), v.stride(1), _ptr(out), D,
                                        _ptr(klens), B, n_heads, D // n_heads, Tq, Tk, float(scale), _stream()),
               "pf_k_attention_small")
    return out


def gather_rows(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """table [rows, D] fp32, ids int32 [n] (device) -> [n, D]: the embedding lookup of the punctuation model."""
    lib = _lib.load()
    t = _f32c(table, "table").contiguous()
    i = ids.to(device=t.device, dtype=torch.int32).contiguous()
    out = torch.empty(i.numel(), t.shape[1], device=t.device, dtype=torch.float32)
    _lib.check(lib.pf_k_gather_rows(_ptr(t), t.stride(0), t.shape[0], _ptr(i), _ptr(out), i.numel(), t.shape[1], _stream()),
               "pf_k_gather_rows")
    return out


def lstm(x: torch.Tensor, w_ih: torch.Tensor, w_hh: torch.Tensor, b_ih: torch.Tensor, b_hh: torch.Tensor) -> torch.Tensor:
    """One layer of torch.nn.LSTM (batch_first, zero initial state): x [B, T, D], w_ih [ndir, 4H, D], w_hh [ndir, 4H, H],
    b_ih / b_hh [ndir, 4H] (the layer's state_dict entries stacked over directions) -> [B, T, ndir * H]."""
    lib = _lib.load()
    xc = _f32c(x, "x").contiguous()
    B, T, D = xc.shape
    ndir, H = w_hh.shape[0], w_hh.shape[2]
    ws = [_f32c(t, "w").contiguous() for t in (w_ih, w_hh, b_ih, b_hh)]
    out = torch.empty(B, T, ndir * H, device=xc.device, dtype=torch.float32)
    _lib.check(lib.pf_k_lstm(_ptr(xc), _ptr(ws[0]), _ptr(ws[1]), _ptr(ws[2]), _ptr(ws[3]), B, T, D, H, ndir, _ptr(out),
                             _stream()), "pf_k_lstm")
    return out


def cif(alphas: torch.Tensor, hidden: torch.Tensor, n_max: int):
    """alphas [B, T], hidden [B, T, D] -> (peaks [B, T], n_fires int32 [B], embeds [B, n_max, D])."""
    lib = _lib.load()
    B, T = alphas.shape
    D = hidden.shape[2]
    peaks = torch.empty(B, T, device=alphas.device, dtype=torch.float32)
    nf = torch.empty(B, device=alphas.device, dtype=torch.int32)
    emb = torch.empty(B, n_max, D, device=alphas.device, dtype=torch.float32)
    _lib.check(lib.pf_k_cif(_ptr(alphas.contiguous()), _ptr(hidden.contiguous()), B, T, D, n_max, _ptr(peaks),
                            _ptr(nf), _ptr(emb), _stream()), "pf_k_cif")
    return peaks, nf, emb


def gemm_time_ms(a, w, bias, out, iters: int = 20) -> float:
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    ms = C.c_float(0)
    _lib.check(lib.pf_k_gemm_f32_time(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(bias), _ptr(out),
                                      out.stride(0), M, N, K, iters, C.byref(ms), _stream()), "pf_k_gemm_f32_time")
    return float(ms.value)


def cast_bf16(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> bf16 (round to nearest even) on the device; returns a torch.bfloat16 tensor of the same shape."""
    lib = _lib.load()
    _f32c(x, "x")
    xc = x.contiguous()
    y = torch.empty(xc.shape, device=x.device, dtype=torch.bfloat16)
    _lib.check(lib.pf_k_cast_bf16(_ptr(xc), _ptr(y), xc.numel(), _stream()), "pf_k_cast_bf16")
    return y


def gemm_bf16(a: torch.Tensor, w: torch.Tensor, bias=None, relu=False, add1=None, add2=None, out_bf16=False):
    """a [M, K] bf16, w [N, K] bf16 -> fp32 (or bf16) [M, N]; fp32 accumulate / bias / residuals."""
    lib = _lib.load()
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty(M, N, device=a.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    _lib.check(lib.pf_k_gemm_bf16(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(bias),
                                  _ptr(add1), add1.stride(0) if add1 is not None else 0,
                                  _ptr(add2), add2.stride(0) if add2 is not None else 0,
                                  _ptr(out), out.stride(0), M, N, K, int(relu), int(out_bf16), _stream()), "pf_k_gemm_bf16")
    return out


def gemm_bf16_time_ms(a, w, bias, out, iters: int = 20) -> float:
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    ms = C.c_float(0)
    _lib.check(lib.pf_k_gemm_bf16_time(_ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(bias), _ptr(out), out.stride(0),
                                       M, N, K, int(out.dtype == torch.bfloat16), iters, C.byref(ms), _stream()),
               "pf_k_gemm_bf16_time")
    return float(ms.value)


def split3(x: torch.Tensor, kpad: int = 32) -> torch.Tensor:
    """fp32 [M, N] -> bf16 planes [3, M, Npad] with x = hi + mid + lo exactly (columns N..Npad-1 zero)."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    M, N = x.shape
    ld = (N + kpad - 1) // kpad * kpad
    y = torch.empty(3, M, ld, device=x.device, dtype=torch.bfloat16)
    _lib.check(lib.pf_k_split3(_ptr(x), x.stride(0), _ptr(y), ld, M * ld, M, N, _stream()), "pf_k_split3")
    return y


def gemm_split3(a3: torch.Tensor, w3: torch.Tensor, bias=None, relu=False, add1=None, add2=None, out_planes=False,
                time_iters: int = 0):
    """a3 [3, M, K], w3 [3, N, K] bf16 planes (ops.split3) -> fp32 [M, N] (or its planes [3, M, N]); fp32-class accuracy
    from six bf16 MFMA products per operand pair. With time_iters > 0 returns (out, ms per launch)."""
    lib = _lib.load()
    assert a3.dtype == torch.bfloat16 and w3.dtype == torch.bfloat16 and a3.is_contiguous() and w3.is_contiguous()
    _, M, K = a3.shape
    N = w3.shape[1]
    assert w3.shape[2] == K
    ms = C.c_float(0)
    if out_planes:
        out = torch.empty(3, M, N, device=a3.device, dtype=torch.bfloat16)
        c, ldc, c3, ldc3, cpl = None, 0, out, N, M * N
    else:
        out = torch.empty(M, N, device=a3.device, dtype=torch.float32)
        c, ldc, c3, ldc3, cpl = out, N, None, 0, 0
    _lib.check(lib.pf_k_gemm_split3(_ptr(a3), K, M * K, _ptr(w3), K, N * K, _ptr(bias),
                                    _ptr(add1), add1.stride(0) if add1 is not None else 0,
                                    _ptr(add2), add2.stride(0) if add2 is not None else 0,
                                    _ptr(c), ldc, _ptr(c3), ldc3, cpl, M, N, K, int(relu), int(time_iters),
                                    C.byref(ms), _stream()), "pf_k_gemm_split3")
    return (out, float(ms.value)) if time_iters > 0 else out


def split2(x: torch.Tensor, scale_exp: int = 0, kpad: int = 32) -> torch.Tensor:
    """fp32 [M, N] -> fp16 planes [2, M, Npad] with x * 2**scale_exp = hi + lo to 2^-24 relative (columns N..Npad-1 zero)."""
    lib = _lib.load()
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    M, N = x.shape
    ld = (N + kpad - 1) // kpad * kpad
    y = torch.empty(2, M, ld, device=x.device, dtype=torch.float16)
    _lib.check(lib.pf_k_split2(_ptr(x), x.stride(0), _ptr(y), ld, M * ld, M, N, float(2.0 ** scale_exp), _stream()), "pf_k_split2")
    return y


def kblocked(p2: torch.Tensor) -> torch.Tensor:
    """planes [2, rows, K] -> the K-blocked layout [2, K / 32, rows, 32] (a 16-row DMA piece is one contiguous KB)"""
    _, R, K = p2.shape
    return p2.view(2, R, K // 32, 32).permute(0, 2, 1, 3).contiguous()


def gemm_f16x2(a2: torch.Tensor, w2: torch.Tensor, bias=None, relu=False, add1=None, add2=None, scale_exp: int = 0,
               out_planes=False, out_scale_exp: int = 0, tile: int = 0, time_iters: int = 0, kblock=False, split_k: bool = False,
               out: torch.Tensor = None):
    """a2 [2, M, K], w2 [2, N, K] fp16 planes (ops.split2; scale_exp = the SUM of their scale exponents) -> fp32 [M, N]
    (or the planes [2, M, N] of the result * 2**out_scale_exp); fp32-class accuracy from three fp16 MFMA products per
    operand pair. With time_iters > 0 returns (out, ms per launch). split_k: the four-slice split-K form (fp32 output; the
    streaming step's long-K projections) -- deterministic, fp32-class, not the bits of the unsplit kernel."""
    lib = _lib.load()
    assert a2.dtype == torch.float16 and w2.dtype == torch.float16 and a2.is_contiguous() and w2.is_contiguous()
    if split_k:
        assert not out_planes and not kblock
        tile |= 0x8000
    if kblock == "w":                            # only W made by kblocked()
        _, M, K = a2.shape
        N, ld = w2.shape[2], K
        tile |= 0x2000
    elif kblock == "a":
        _, kb, M, _ = a2.shape
        K, N, ld = kb * 32, w2.shape[1], 32
        tile |= 0x4000
    elif kblock:                                 # operands made by kblocked(): [2, K / 32, rows, 32]
        _, kb, M, _ = a2.shape
        K, N, ld = kb * 32, w2.shape[2], 32
        assert w2.shape[1] == kb
        tile |= 0x1000
    else:
        _, M, K = a2.shape
        N, ld = w2.shape[1], K
        assert w2.shape[2] == K
    ms = C.c_float(0)
    if out_planes:
        out = torch.empty(2, M, N, device=a2.device, dtype=torch.float16)
        c, ldc, c2, ldc2, cpl = None, 0, out, N, M * N
    else:
        if out is None:
            out = torch.empty(M, N, device=a2.device, dtype=torch.float32)
        assert out.dtype == torch.float32 and tuple(out.shape) == (M, N) and out.is_contiguous()     # may alias add1 / add2
        c, ldc, c2, ldc2, cpl = out, N, None, 0, 0
    _lib.check(lib.pf_k_gemm_f16x2(_ptr(a2), ld, M * K, _ptr(w2), ld, N * K, float(2.0 ** -scale_exp), _ptr(bias),
                                   _ptr(add1), add1.stride(0) if add1 is not None else 0,
                                   _ptr(add2), add2.stride(0) if add2 is not None else 0,
                                   _ptr(c), ldc, _ptr(c2), ldc2, cpl, float(2.0 ** out_scale_exp), M, N, K, int(relu),
                                   int(tile), int(time_iters), C.byref(ms), _stream()), "pf_k_gemm_f16x2")
    return (out, float(ms.value)) if time_iters > 0 else out


def gemm_f16x2_row(a2: torch.Tensor, w2: torch.Tensor, bias=None, add1=None, add2=None, scale_exp: int = 0, relu=False,
                   ln=None, out_scale_exp: int = 0, ln_planes: bool = True, want_c: bool = True, a_nt: bool = False,
                   time_iters: int = 0, block_rows: int = 0):
    """Full-row form (gemm_f16x2_row.hip, N = 512): c = relu?(a w^T + bias) + add1, add2 + c; with ln = (gamma, beta, eps)
    also y = LayerNorm(c) as fp16 planes [2, M, 512] of y * 2**out_scale_exp (ln_planes) or fp32 [M, 512].
    block_rows: 0 by the row count, 128 the 2 x 4-wave kernel, 96 / 129 the 1 x 8-wave kernel (gemm_f16x2_row8.hip) with 96 /
    128 rows per block. Returns (c or None, y or None[, ms])."""
    lib = _lib.load()
    assert a2.dtype == torch.float16 and w2.dtype == torch.float16 and a2.is_contiguous() and w2.is_contiguous()
    _, M, K = a2.shape
    assert w2.shape[1] == 512 and w2.shape[2] == K
    dev = a2.device
    c = torch.empty(M, 512, device=dev, dtype=torch.float32) if (want_c or ln is None) else None
    y2 = yf = None
    g = b = None
    eps = 0.0
    if ln is not None:
        g, b, eps = ln
        if ln_planes:
            y2 = torch.empty(2, M, 512, device=dev, dtype=torch.float16)
        else:
            yf = torch.empty(M, 512, device=dev, dtype=torch.float32)
    ms = C.c_float(0)
    _lib.check(lib.pf_k_gemm_f16x2_row(_ptr(a2), K, M * K, _ptr(w2), K, 512 * K, float(2.0 ** -scale_exp), _ptr(bias),
                                       _ptr(add1), add1.stride(0) if add1 is not None else 0,
                                       _ptr(add2), add2.stride(0) if add2 is not None else 0,
                                       _ptr(c), 512, _ptr(g), _ptr(b), float(eps), _ptr(y2), M * 512, float(2.0 ** out_scale_exp),
                                       _ptr(yf), M, K, int(relu), (int(a_nt) & 3) | (int(block_rows) << 8), int(time_iters),
                                       C.byref(ms), _stream()),
               "pf_k_gemm_f16x2_row")
    y = y2 if y2 is not None else yf
    return (c, y, float(ms.value)) if time_iters > 0 else (c, y)


def ffn_f16x2(x2: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, b1: torch.Tensor, b2, resid: torch.Tensor, e_x: int, e_w1: int,
              e_h: int, e_w2: int, ln=None, out_scale_exp: int = 0, ln_planes: bool = True, want_c: bool = True,
              in_place: bool = False, time_iters: int = 0, abl: int = 0, w_kblocked: bool = False):
    """The encoder block's feed-forward in one launch (gemm_f16x2_ffn.hip): c = resid + (relu(x w1^T + b1) w2^T + b2) with
    x2 [2, M, 512], w1 [2, F, 512], w2 [2, 512, F] fp16 planes of x 2^e_x, w1 2^e_w1, w2 2^e_w2; the hidden activations are
    split into planes of h 2^e_h in registers. With ln = (gamma, beta, eps) also y = LayerNorm(c) as planes [2, M, 512] of
    y 2^out_scale_exp or fp32. in_place: c is written into `resid`. Returns (c or None, y or None[, ms])."""
    lib = _lib.load()
    for t_ in (x2, w1, w2):
        assert t_.dtype == torch.float16 and t_.is_contiguous()
    _, M, D = x2.shape
    if w_kblocked:                               # weights made by kblocked(): [2, K / 32, rows, 32]
        F = w1.shape[2]
        assert D == 512 and tuple(w1.shape) == (2, 16, F, 32) and tuple(w2.shape) == (2, F // 32, 512, 32)
    else:
        F = w1.shape[1]
        assert D == 512 and tuple(w1.shape) == (2, F, 512) and tuple(w2.shape) == (2, 512, F)
    assert resid.dtype == torch.float32 and tuple(resid.shape) == (M, 512) and resid.is_contiguous()
    dev = x2.device
    c = resid if in_place else (torch.empty(M, 512, device=dev, dtype=torch.float32) if (want_c or ln is None) else None)
    y2 = yf = g = b = None
    eps = 0.0
    if ln is not None:
        g, b, eps = ln
        if ln_planes:
            y2 = torch.empty(2, M, 512, device=dev, dtype=torch.float16)
        else:
            yf = torch.empty(M, 512, device=dev, dtype=torch.float32)
    ms = C.c_float(0)
    _lib.check(lib.pf_k_ffn_f16x2(_ptr(x2), _ptr(w1), _ptr(w2), _ptr(b1), _ptr(b2), float(2.0 ** -(e_x + e_w1)), float(2.0 ** e_h),
                                  float(2.0 ** -(e_h + e_w2)), _ptr(resid), _ptr(c), _ptr(g), _ptr(b), float(eps), _ptr(y2),
                                  float(2.0 ** out_scale_exp), _ptr(yf), M, F | (int(abl) << 24) | (int(bool(w_kblocked)) << 28), int(time_iters), C.byref(ms), _stream()),
               "pf_k_ffn_f16x2")
    y = y2 if y2 is not None else yf
    return (c, y, float(ms.value)) if time_iters > 0 else (c, y)


def gemm_f16x2_row_fsmn(a2: torch.Tensor, w2: torch.Tensor, bias, v: torch.Tensor, taps: torch.Tensor, lo: torch.Tensor,
                        hi: torch.Tensor, add2=None, scale_exp: int = 0, ln=None, out_scale_exp: int = 0, ln_planes: bool = True,
                        want_c: bool = True, a_nt: bool = False, time_iters: int = 0, block_rows: int = 0):
    """FSMN form of the full-row kernel: c = (a w^T + bias) + fsmn_memory(v), add2 + c, y = LayerNorm(c). v fp32 [M, 512], taps
    [512, 11]; lo / hi int32 [M / 16]: valid v rows [lo, hi) of the sequence owning each 16-row group. Returns (c or None, y[, ms])."""
    lib = _lib.load()
    assert a2.dtype == torch.float16 and w2.dtype == torch.float16 and a2.is_contiguous() and w2.is_contiguous()
    _, M, K = a2.shape
    assert w2.shape[1] == 512 and w2.shape[2] == K and M % 16 == 0 and ln is not None
    assert v.dtype == torch.float32 and v.shape == (M, 512) and v.stride(1) == 1 and taps.shape == (512, 11) and taps.is_contiguous()
    assert lo.dtype == torch.int32 and hi.dtype == torch.int32 and lo.numel() == M // 16 and hi.numel() == M // 16
    dev = a2.device
    c = torch.empty(M, 512, device=dev, dtype=torch.float32) if want_c else None
    g, b, eps = ln
    y2 = torch.empty(2, M, 512, device=dev, dtype=torch.float16) if ln_planes else None
    yf = None if ln_planes else torch.empty(M, 512, device=dev, dtype=torch.float32)
    ms = C.c_float(0)
    _lib.check(lib.pf_k_gemm_f16x2_row_fsmn(_ptr(a2), K, M * K, _ptr(w2), K, 512 * K, float(2.0 ** -scale_exp), _ptr(bias),
                                            _ptr(v), v.stride(0), _ptr(taps), _ptr(lo), _ptr(hi),
                                            _ptr(add2), add2.stride(0) if add2 is not None else 0, _ptr(c), 512, _ptr(g), _ptr(b),
                                            float(eps), _ptr(y2), M * 512, float(2.0 ** out_scale_exp), _ptr(yf), M, K,
                                            (int(a_nt) & 3) | (int(block_rows) << 8), int(time_iters), C.byref(ms), _stream()),
               "pf_k_gemm_f16x2_row_fsmn")
    y = y2 if y2 is not None else yf
    return (c, y, float(ms.value)) if time_iters > 0 else (c, y)


def layernorm_planes(x: torch.Tensor, gamma, beta, eps: float, scale_exp: int = 0, time_iters: int = 0):
    """LayerNorm over the last dim of fp32 [M, D] -> fp16 planes [2, M, D] of y * 2**scale_exp (layernorm_kernel, plane output)."""
    lib = _lib.load()
    M, D = x.shape
    assert x.stride(1) == 1
    y = torch.empty(2, M, D, device=x.device, dtype=torch.float16)
    ms = C.c_float(0)
    _lib.check(lib.pf_k_layernorm_planes(_ptr(x), x.stride(0), _ptr(gamma), _ptr(beta), _ptr(y), D, M * D, float(2.0 ** scale_exp),
                                         M, D, float(eps), int(time_iters), C.byref(ms), _stream()), "pf_k_layernorm_planes")
    return (y, float(ms.value)) if time_iters > 0 else y


def vt_columns(m: torch.Tensor) -> torch.Tensor:
    """column of V^T that holds row m: bits 2 and 3 of the row index swapped (attention_f16x2.hip)"""
    return (m & ~12) | ((m & 4) << 1) | ((m & 8) >> 1)


def gemm_f16x2_qkv(a2: torch.Tensor, w2: torch.Tensor, bias, D: int, scale_exp: int, q_mul: float, k_mul: float, v_mul: float,
                   kv_form: bool = False, tile: int = 0, time_iters: int = 0):
    """QKV form (w2 [2, 3 D, K]) / KV form (w2 [2, 2 D, K]) of gemm_f16x2: returns dict(q2, k2 planes [2, M + 32, D] (q2 None in
    the KV form), v fp32 [M, D] (None in the KV form), vt planes [2, D, M + 64])."""
    lib = _lib.load()
    _, M, K = a2.shape
    dev = a2.device
    nseg = 2 if kv_form else 3
    assert w2.shape[1] == nseg * D and M % 16 == 0
    q2 = None if kv_form else torch.zeros(2, M + 32, D, device=dev, dtype=torch.float16)
    k2 = torch.zeros(2, M + 32, D, device=dev, dtype=torch.float16)
    v = None if kv_form else torch.empty(M, D, device=dev, dtype=torch.float32)
    ldvt = M + 64
    vt = torch.zeros(2, D, ldvt, device=dev, dtype=torch.float16)
    ms = C.c_float(0)
    _lib.check(lib.pf_k_gemm_f16x2_qkv(_ptr(a2), K, M * K, _ptr(w2), K, nseg * D * K, float(2.0 ** -scale_exp), _ptr(bias), M, D, K,
                                       int(kv_form), _ptr(q2), _ptr(k2), (M + 32) * D, _ptr(v), _ptr(vt), ldvt, D * ldvt,
                                       float(q_mul), float(k_mul), float(v_mul), int(tile), int(time_iters), C.byref(ms), _stream()),
               "pf_k_gemm_f16x2_qkv")
    out = dict(q2=q2, k2=k2, v=v, vt=vt)
    if time_iters > 0:
        out["ms"] = float(ms.value)
    return out


def gemm_f16x2_argmax(a2: torch.Tensor, w2: torch.Tensor, bias, scale_exp: int = 0) -> torch.Tensor:
    """ids[row] = argmax_n (a w^T * 2**-scale_exp + bias)[row, n] through the fused arg-max epilogue of gemm_f16x2."""
    lib = _lib.load()
    _, M, K = a2.shape
    N = w2.shape[1]
    nparts = 2 * ((N + 255) // 256)
    dev = a2.device
    ids = torch.empty(M, device=dev, dtype=torch.int32)
    sval = torch.empty(M, nparts, device=dev, dtype=torch.float32)
    sidx = torch.empty(M, nparts, device=dev, dtype=torch.int32)
    _lib.check(lib.pf_k_gemm_f16x2_argmax(_ptr(a2), K, M * K, _ptr(w2), K, N * K, float(2.0 ** -scale_exp), _ptr(bias), M, N, K,
                                          _ptr(ids), _ptr(sval), _ptr(sidx), _stream()), "pf_k_gemm_f16x2_argmax")
    return ids


def attention_f16x2(q, k, v, klens, n_heads: int, scale: float, eq: int = 6, ek: int = 6, ev: int = 6, variant: int = 0,
                    time_iters: int = 0):
    """fp32 q [B, Tq, D], k / v [B, Tp, D] (Tp % 16 == 0) -> fp32 [B, Tq, D] through attention_f16x2.hip: builds the kernel's
    operand layouts here (planes via ops.split2, V transposed with index bits 2 and 3 of the row swapped) -- the layouts the
    QKV form of gemm_f16x2 writes in the engine. Test / measurement helper."""
    lib = _lib.load()
    B, Tq, D = q.shape
    Tp = k.shape[1]
    assert Tp % 16 == 0 and D == n_heads * 128
    dev = q.device
    q2 = split2((q * scale).reshape(B * Tq, D).contiguous(), eq)
    k2 = torch.zeros(2, B * Tp + 32, D, device=dev, dtype=torch.float16)
    k2[:, : B * Tp] = split2(k.reshape(B * Tp, D).contiguous(), ek)
    v2 = split2(v.reshape(B * Tp, D).contiguous(), ev)
    col = vt_columns(torch.arange(B * Tp, device=dev))
    ldvt = B * Tp + 64
    vt = torch.zeros(2, D, ldvt, device=dev, dtype=torch.float16)
    vt[:, :, col] = v2.transpose(1, 2)
    o2 = torch.empty(2, B * Tq, D, device=dev, dtype=torch.float16)
    ms = C.c_float(0)
    _lib.check(lib.pf_k_attention_f16x2(_ptr(q2), B * Tq * D, _ptr(k2), (B * Tp + 32) * D, _ptr(vt), ldvt, D * ldvt, _ptr(o2), B * Tq * D,
                                        _ptr(klens), B, n_heads, Tp, Tq, float(2.0 ** -(eq + ek)), float(2.0 ** -10), int(variant),
                                        int(time_iters), C.byref(ms), _stream()), "pf_k_attention_f16x2")
    out = ((o2[0].float() + o2[1].float()) * 2.0 ** -ev).view(B, Tq, D)
    return (out, float(ms.value)) if time_iters > 0 else out


def attention_bf16(q, k, v, klens, n_heads: int, scale: float):
    """bf16 q [B, Tq, H*128], k/v [B, Tk, H*128] (row-strided views allowed) -> bf16 [B, Tq, H*128]."""
    lib = _lib.load()
    B, Tq, D = q.shape
    Tk = k.shape[1]
    assert q.dtype == torch.bfloat16 and q.stride(2) == 1 and k.stride(2) == 1 and v.stride(2) == 1
    out = torch.empty(B, Tq, D, device=q.device, dtype=torch.bfloat16)
    _lib.check(lib.pf_k_attention_bf16(_ptr(q), q.stride(1), _ptr(k), k.stride(1), _ptr(v), v.stride(1), _ptr(out), D,
                                       _ptr(klens), B, n_heads, Tq, Tk, float(scale), _stream()), "pf_k_attention_bf16")
    return out
