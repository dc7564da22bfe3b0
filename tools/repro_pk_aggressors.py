#!/usr/bin/env python3
"""Round 5 hunt for the packed-fp32 fault's trigger. The victim is fbank_kernel built WITH packed-fp32 VALU instructions
(funasr_amd/libparaformer_hip_pk.so: the product library with frontend.hip compiled without the -packed-fp32-ops switch-off), its
cross-check on; the aggressor runs on a second stream of the same process. Which aggressors disturb it?
  build: make -C funasr_amd/csrc pk        usage: repro_pk_aggressors.py SECONDS [pk|nopk]        record: profiles/r05_pk_reproducer.txt"""
import ctypes as C
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from funasr_amd import _lib
which = sys.argv[2] if len(sys.argv) > 2 else "pk"
if which == "pk":
    _lib.LIB_PATH = os.path.join(ROOT, "funasr_amd", "libparaformer_hip_pk.so")
import torch
from funasr_amd import ops, synth
from funasr_amd.wav_frontend import WavFrontend

dev = torch.device("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
g = torch.Generator().manual_seed(0)
sh, sc = synth.synthetic_cmvn(560)
wav = synth.speech_like(235000, seed=7).to(dev)[None]
side = torch.cuda.Stream(device=dev)
with torch.cuda.stream(side):
    a = ops.split2(torch.randn(1024, 512, generator=g).to(dev), 8)
    w = ops.split2((torch.randn(2048, 512, generator=g) * 512 ** -0.5).to(dev), 12)
    b = torch.zeros(2048, device=dev)
    big_a = ops.split2(torch.randn(8192, 512, generator=g).to(dev), 8)
    af = torch.randn(1024, 512, generator=g).to(dev)
    wf = (torch.randn(2048, 512, generator=g) * 512 ** -0.5).to(dev)
    x = torch.randn(16 * 64 * 8, device=dev) * 256.0
    hi = x.to(torch.float16); lo = (x - hi.float()).to(torch.float16); q = 4 * 64 * 8
    planes = torch.stack([hi[:q], lo[:q], hi[q:2 * q], lo[q:2 * q]]).contiguous()
    scratch = torch.zeros(16, device=dev)
torch.cuda.synchronize()
AGG = {
    "none": lambda: None,
    "gemm_f16x2 128x128 (round 4's aggressor)": lambda: ops.gemm_f16x2(a, w, b, scale_exp=20, tile=3),
    "gemm_f16x2 128x128, no epilogue": lambda: ops.gemm_f16x2(a, w, b, scale_exp=20, tile=0x82),
    "gemm_f16x2 128x128, no operand DMA": lambda: ops.gemm_f16x2(a, w, b, scale_exp=20, tile=0x83),
    "gemm_f16x2 256x256 eight waves": lambda: ops.gemm_f16x2(big_a, w, b, scale_exp=20, tile=2),
    "gemm_f16x2_w4 four waves": lambda: ops.gemm_f16x2(big_a, w, b, scale_exp=20, tile=7),
    "exact-fp32 MFMA GEMM": lambda: ops.gemm(af, wf, b),
}
# synthetic aggressors (tools/micro/pk_aggr.hip): 256-thread workgroups, two per CU, launched on the side stream
pa = C.CDLL(os.path.join(ROOT, "tools", "micro", "pk_aggr.so"))
pa.pk_aggr_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
srcbuf = (torch.rand(65536 // 2, device=dev) * 1.875 + 0.125).to(torch.float16)
def synth_aggr(kind, lds):
    def f():
        rc = pa.pk_aggr_launch(C.c_void_p(side.cuda_stream), kind, lds, srcbuf.data_ptr(), 512, 400, scratch.data_ptr())
        assert rc == 0, rc
    return f
for label, kind, lds in (("synthetic: f16 MFMA + ds_read_b128 + barriers, 64 KB LDS", 11, 64), ("synthetic: f16 MFMA + barriers, register operands, 64 KB LDS", 9, 64),
                         ("synthetic: ds_read_b128 + barriers only, 64 KB LDS", 10, 64), ("synthetic: f16 MFMA + ds_read_b128, no barriers", 3, 64),
                         ("synthetic: f16 MFMA only, no barriers", 1, 64), ("synthetic: bf16 MFMA + ds_read_b128 + barriers", 15, 64),
                         ("synthetic: f16 MFMA + ds_read_b128 + barriers, 16 KB LDS", 11, 16), ("synthetic: f16 MFMA + barriers, register operands, 16 KB LDS", 9, 16),
                         ("synthetic: packed-fp32 VALU + ds_read_b128 + barriers (no MFMA)", 26, 64), ("synthetic: packed-fp32 VALU + barriers (no MFMA), 16 KB LDS", 24, 16)):
    AGG[label] = synth_aggr(kind, lds)
out = {"victim_library": which}
for name, run in AGG.items():
    fe = WavFrontend(cmvn=torch.stack([sh, sc]), lfr_m=7, lfr_n=6, dither=0.0, device=dev, verify=True)
    ref = fe(wav, [235000])[0].clone()
    torch.cuda.synchronize()
    t0 = time.time(); it = bad = 0
    while time.time() - t0 < secs:
        with torch.cuda.stream(side):
            if name != "none":
                for _ in range(40):
                    run()
        for _ in range(4):
            f = fe(wav, [235000])[0]
            it += 1
            bad += 0 if torch.equal(f, ref) else 1
    torch.cuda.synchronize()
    out[name] = {"frontend_calls": it, "mismatching_outputs": bad, "cross_check_disagreements": fe.faults()}
    print(json.dumps({name: out[name]}), flush=True)
print(json.dumps(out))
