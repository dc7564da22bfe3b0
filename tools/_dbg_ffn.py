import sys, torch
sys.path.insert(0, '/root/repo')
from funasr_amd import synth
from funasr_amd.paraformer import Paraformer
from funasr_amd.wav_frontend import WavFrontend
dev = torch.device('cuda:0')
cfg = synth.PARAFORMER_LARGE
model = Paraformer.from_config(cfg)
model.load_state_dict(synth.paraformer_state_dict(cfg, seed=0, cif_bias=synth.BENCH_CIF_BIAS), strict=False)
model = model.to(dev).set_precision("f16x2")
shift, scale = synth.synthetic_cmvn(560)
fe = WavFrontend(cmvn=torch.stack([shift, scale]), lfr_m=7, lfr_n=6, dither=0.0, device=dev)
B = 64
clips = [synth.speech_like(480000, seed=i) for i in range(B)]
wav = torch.stack(clips).to(dev)
feats, flens = fe(wav, [480000] * B)
res = {}
import os
model.encoder.set_option("ffn_abl", int(os.environ.get("FFN_ABL", "0")))
for ff in (0, 2, 1, 0):
    model.encoder.set_option("ffn_fused", ff)
    for allrows in (True, False):
        enc, olens = model.encode(feats, flens, all_rows=allrows)
        res.setdefault((ff, allrows), []).append(enc.clone())
for allrows in (True, False):
    a = res[(0, allrows)][0]
    print("all_rows", allrows, "repeat ffn0 equal:", torch.equal(a, res[(0, allrows)][1]))
    for ff in (2, 1):
        b = res[(ff, allrows)][0]
        d = (a - b).abs()
        print(f"  ffn_fused {ff}: max|d| {d.max().item():.3e}, differing {int((d > 0).sum())} of {d.numel()}, nan {int(torch.isnan(b).sum())}")
        if d.max() > 0:
            bad = (d > 0).nonzero()
            print("    clips:", bad[:, 0].unique().tolist()[:20], "frames range:", int(bad[:, 1].min()), int(bad[:, 1].max()))
