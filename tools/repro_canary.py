#!/usr/bin/env python3
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from funasr_amd import _lib
lib = _lib.load()
torch.zeros(1, device="cuda:0")
t0 = time.time(); bad = 0; n = 0
while time.time() - t0 < float(os.environ.get("REPRO_SECONDS", "20")):
    r = lib.pf_debug_lds_canary(1024, 200, torch.cuda.current_stream().cuda_stream)
    assert r >= 0, _lib.last_error()
    bad += r; n += 1
print(json.dumps({"canary_launches": n, "lds_words_changed": bad}))
