import sys, os, time, json, tempfile, cProfile, pstats
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo/tests")
import torch
from funasr_amd import synth
from funasr_amd.auto_model import AutoModel
from bench_generate import model_dir
from sweep import durations
from _model_dir import write_wav
work = tempfile.mkdtemp()
cfg = synth.PARAFORMER_LARGE
model_dir(os.path.join(work, "m"), cfg)
durs = durations(6000)
pool = [synth.speech_like(int(14.7 * 16000) + 1, seed=1000 + i) for i in range(16)]
paths = []
for i, d in enumerate(durs):
    p = os.path.join(work, f"c{i}.wav"); write_wav(p, pool[i % 16].roll(31 * i)[: int(d * 16000)]); paths.append(p)
am = AutoModel(model=os.path.join(work, "m"), device="cuda:0", batch_size=256, disable_pbar=True)
am.model.load_state_dict(synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS), strict=False); am.model.to("cuda:0")
am.generate(input=paths[:4])       # library loaded, kernels configured
torch.cuda.synchronize()
pr = cProfile.Profile(); t = time.perf_counter(); pr.enable()
am.generate(input=paths, batch_size_rows=32768)
pr.disable(); torch.cuda.synchronize(); t1 = time.perf_counter() - t
t = time.perf_counter(); am.generate(input=paths, batch_size_rows=32768); torch.cuda.synchronize(); t2 = time.perf_counter() - t
print(json.dumps({"first_call_s": round(t1, 3), "second_call_s": round(t2, 3)}))
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
