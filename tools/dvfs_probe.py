#!/usr/bin/env python3
"""Clock / power under a sustained f16x2 GEMM loop: runs one shape for a few seconds per variant while sampling rocm-smi, and
reports the launch time next to the sampled sclk and socket power. Variants: block shape (tile 2 = two 32-deep stages,
6 = deep ring of 16-deep stages) x operand data (random / zeros: zeros toggle no data lines, the MFMA count is the same)."""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import ops

dev = torch.device("cuda:0")
M = 32768
shapes = {"w2": (512, 2048, dict(resid=True)), "w1": (2048, 512, dict(relu=True, out_planes=True, out_scale_exp=9))}


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        card = next(iter(d.values()))
        keep = {k: v for k, v in card.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower()}
        return keep
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[:80]}


for name in sys.argv[1:] or ["w2"]:
    N, K, kw = shapes[name]
    kw = dict(kw)
    for data in ("random", "zeros"):
        a = torch.randn(M, K, device=dev) if data == "random" else torch.zeros(M, K, device=dev)
        w = torch.randn(N, K, device=dev) * K ** -0.5 if data == "random" else torch.zeros(N, K, device=dev)
        b = torch.randn(N, device=dev)
        a2, w2 = ops.split2(a, 8), ops.split2(w, 12)
        k2 = dict(kw)
        if k2.pop("resid", False):
            k2["add2"] = torch.randn(M, N, device=dev)
        for tile in (2, 6):
            samples = []
            stop = False

            def sampler():
                time.sleep(1.0)
                while not stop:
                    samples.append(smi())
                    time.sleep(0.7)
            th = threading.Thread(target=sampler)
            th.start()
            t0 = time.time()
            times = []
            while time.time() - t0 < 4.0:
                times.append(ops.gemm_f16x2(a2, w2, b, scale_exp=20, tile=tile, time_iters=200, **k2)[1])
            stop = True
            th.join()
            print(json.dumps({"shape": name, "data": data, "tile": tile, "us": [round(t * 1e3, 1) for t in times[:2] + times[-2:]],
                              "smi": samples[:1] + samples[-2:]}), flush=True)
