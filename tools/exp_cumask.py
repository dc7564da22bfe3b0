#!/usr/bin/env python3
"""Round 5 experiment: two half-batches on two CU-masked streams (half of the chip each) against one full batch on the whole chip.
A tile GEMM's epilogue is a chip-wide store burst at the fabric's write rate with every matrix pipe idle; two independent halves
run out of phase, so one half's bursts lie under the other half's MFMA loops. Measurement infrastructure, not product code."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from funasr_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
lib = C.CDLL(os.path.join(ROOT, "tools", "micro", "cumask_probe.so"))
lib.cumask_stream_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_int]
lib.cumask_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]


def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits(32 * w + b)) for w in range(8)])
    s = C.c_void_p()
    rc = lib.cumask_stream_create(C.byref(s), words, 8)
    assert rc == 0, rc
    return s.value


def census(stream_ptr, label):
    out = torch.zeros(2 * 1024, dtype=torch.int32, device=dev)
    lib.cumask_probe(stream_ptr, out.data_ptr(), 1024, 200)
    torch.cuda.synchronize()
    o = out.cpu().view(-1, 2).numpy()
    xcc = o[:, 0] & 0xf
    cu = (o[:, 1] >> 8) & 0xf
    se = (o[:, 1] >> 13) & 0x7
    sh = (o[:, 1] >> 12) & 0x1
    places = sorted(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))
    per_xcc = {int(x): len({p for p in places if p[0] == x}) for x in sorted(set(xcc.tolist()))}
    print(json.dumps({"census": label, "distinct_cus": len(places), "per_xcc": per_xcc,
                      "block_mod8_is_xcc": bool(((o[:, 0] & 0xf) == (torch.arange(1024).numpy() % 8)).all())}), flush=True)


PARTS = {"low_high": (lambda b: b < 128, lambda b: b >= 128), "even_odd": (lambda b: b % 2 == 0, lambda b: b % 2 == 1),
         "mod16": (lambda b: (b // 8) % 2 == 0, lambda b: (b // 8) % 2 == 1)}
g = torch.Generator(device=dev).manual_seed(1)


def ops_for(M):
    a = torch.randn(M, 512, device=dev, generator=g)
    w1 = torch.randn(2048, 512, device=dev, generator=g) * 512 ** -0.5
    w2 = torch.randn(512, 2048, device=dev, generator=g) * 2048 ** -0.5
    b1 = torch.randn(2048, device=dev, generator=g)
    b2 = torch.randn(512, device=dev, generator=g)
    res = torch.randn(M, 512, device=dev, generator=g)
    gam, bet = torch.ones(512, device=dev), torch.zeros(512, device=dev)
    return dict(a2=ops.split2(a, 8), w1=ops.split2(w1, 12), w2=ops.split2(w2, 12), b1=b1, b2=b2, res=res, ln=(gam, bet, 1e-12))


def ffn_pair(o):
    h2 = ops.gemm_f16x2(o["a2"], o["w1"], o["b1"], scale_exp=20, relu=True, out_planes=True, out_scale_exp=9)
    ops.gemm_f16x2_row(h2, o["w2"], o["b2"], add2=o["res"], scale_exp=21, ln=o["ln"], out_scale_exp=8)


census(None, "default stream")
N = 40
full = ops_for(32768)
for _ in range(3):
    ffn_pair(full)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(N):
    ffn_pair(full)
torch.cuda.synchronize()
base = (time.time() - t0) / N * 1e6
print(json.dumps({"one_stream_full_chip_M32768_us_per_pair": round(base, 1)}), flush=True)
halves = [ops_for(16384), ops_for(16384)]
for name, (fa, fb) in PARTS.items():
    sa, sb = masked_stream(fa), masked_stream(fb)
    census(sa, name + " A")
    census(sb, name + " B")
    ta, tb = torch.cuda.ExternalStream(sa), torch.cuda.ExternalStream(sb)
    for offset in (0, 1):
        for s_, h in ((ta, halves[0]), (tb, halves[1])):
            with torch.cuda.stream(s_):
                for _ in range(3):
                    ffn_pair(h)
        torch.cuda.synchronize()
        t0 = time.time()
        if offset:                       # B starts one GEMM late: the two halves are out of phase from the first launch on
            with torch.cuda.stream(ta):
                ops.gemm_f16x2(halves[0]["a2"], halves[0]["w1"], halves[0]["b1"], scale_exp=20, relu=True, out_planes=True, out_scale_exp=9)
        for _ in range(N):
            with torch.cuda.stream(ta):
                ffn_pair(halves[0])
            with torch.cuda.stream(tb):
                ffn_pair(halves[1])
        torch.cuda.synchronize()
        t = (time.time() - t0) / N * 1e6
        print(json.dumps({"partition": name, "offset_start": offset, "two_masked_streams_M16384_each_us_per_pair_of_pairs": round(t, 1),
                          "vs_one_stream": round(t / base, 3)}), flush=True)
# control: the same two streams' work without masks
ua, ub = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(N):
    with torch.cuda.stream(ua):
        ffn_pair(halves[0])
    with torch.cuda.stream(ub):
        ffn_pair(halves[1])
torch.cuda.synchronize()
t = (time.time() - t0) / N * 1e6
print(json.dumps({"partition": "none (two plain streams)", "us": round(t, 1), "vs_one_stream": round(t / base, 3)}), flush=True)
