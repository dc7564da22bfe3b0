"""Generate tests/golden/*.npz by running the REFERENCE's own code on seeded inputs.  TEST INFRASTRUCTURE ONLY.

Run in the build container (needs /root/reference):   python -m oracle.make_golden
Sources of truth:
  * the reference's nn.Modules imported from /root/reference (SANMEncoder, CifPredictorV2, cif_v1,
    ParaformerSANMDecoder, SenseVoiceEncoderSmall, CTC, SinusoidalPositionEncoder, apply_lfr/apply_cmvn/load_cmvn),
    loaded with funasr_amd.synth state_dicts through load_state_dict(strict=True) (which also pins key names/shapes);
  * the reference-vendored kaldi-native-fbank built by oracle/Makefile (oracle/_ref/libknf_ref.so) for fbank.
The fixtures hold inputs, reference outputs, the config and the weight seed; tests rebuild the weights from the
seed (CPU torch.Generator is deterministic) and compare the oracle (CPU suite) and the HIP path (-m gpu) to them.
"""
from __future__ import annotations

import ctypes
import json
import os
import shutil
import sys
import wave

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from funasr_amd import synth  # noqa: E402
from oracle import ref_import  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF = ref_import.REF_ROOT
AM_MVN = os.path.join(REF, "runtime/triton_gpu/model_repo_paraformer_large_online/lfr_cmvn_pe/am.mvn")
WAV = os.path.join(REF, "runtime/funasr_api/asr_example.wav")


def knf_fbank(wave_scaled: np.ndarray, n_mels=80) -> np.ndarray:
    lib = ctypes.CDLL(os.path.join(HERE, "_ref", "libknf_ref.so"))
    lib.knf_fbank.restype = ctypes.c_int
    lib.knf_fbank.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                              ctypes.c_float, ctypes.c_void_p, ctypes.c_int64]
    w = np.ascontiguousarray(wave_scaled, dtype=np.float32)
    max_frames = max(1, 1 + (len(w) - 400) // 160)
    out = np.zeros((max_frames, n_mels), dtype=np.float32)
    n = lib.knf_fbank(w.ctypes.data, len(w), n_mels, 25, 10, 16000.0, out.ctypes.data, max_frames)
    return out[:n]


def save(name, **arrs):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KB)")


def sd_checksum(sd) -> float:
    return float(sum(v.double().abs().sum() for v in sd.values()))


def main():
    os.makedirs(GOLD, exist_ok=True)
    R = ref_import.modules()
    wf = R["wav_frontend"]
    torch.manual_seed(0)
    torch.set_num_threads(4)

    # ------------------------------------------------------------------ frontend (knf fbank + reference LFR/CMVN)
    shutil.copyfile(AM_MVN, os.path.join(GOLD, "am.mvn"))
    cmvn = wf.load_cmvn(AM_MVN)
    with wave.open(WAV, "rb") as f:
        assert f.getframerate() == 16000 and f.getnchannels() == 1 and f.getsampwidth() == 2
        pcm = np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16)
    clip_a = pcm[16000:16000 + 20777].copy()                       # 1.3 s of real speech, ragged length
    clip_b = (synth.speech_like(9000, seed=3).numpy() * 32768).astype(np.int16)
    fb, feats = {}, {}
    for key, clip in (("a", clip_a), ("b", clip_b)):
        wave_f = clip.astype(np.float32) / 32768.0                 # what load_audio hands to the frontend
        fbk = knf_fbank(wave_f * 32768.0)
        mat = wf.apply_lfr(torch.from_numpy(fbk.copy()), 7, 6)
        mat = wf.apply_cmvn(mat, cmvn)
        fb[key], feats[key] = fbk, mat.numpy()
    save("frontend", pcm_a=clip_a, pcm_b=clip_b, fbank_knf_a=fb["a"], fbank_knf_b=fb["b"], feats_a=feats["a"],
         feats_b=feats["b"], cmvn=cmvn.numpy())
    lfr = {}
    g = torch.Generator().manual_seed(11)
    for T in (1, 2, 3, 5, 6, 7, 8, 13, 100):
        x = torch.randn(T, 4, generator=g)
        lfr[f"in_{T}"] = x.numpy()
        lfr[f"out_{T}"] = wf.apply_lfr(x.clone(), 7, 6).numpy()
    save("lfr", **lfr)

    # ------------------------------------------------------------------ positional encoding
    pe = R["SinusoidalPositionEncoder"]()(torch.zeros(1, 600, 560))[0]
    save("pe", head=pe[:40].numpy(), tail=pe[560:600].numpy())

    # ------------------------------------------------------------------ encoder (3 blocks)
    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=3, dec_blocks=2, vocab=97)
    ec = cfg["encoder"]
    enc = R["SANMEncoder"](input_size=ec["input_size"], output_size=ec["output_size"],
                           attention_heads=ec["attention_heads"], linear_units=ec["linear_units"],
                           num_blocks=ec["num_blocks"], dropout_rate=0.1, input_layer="pe",
                           pos_enc_class=R["SinusoidalPositionEncoder"], normalize_before=True,
                           kernel_size=ec["kernel_size"], sanm_shfit=ec["sanm_shfit"],
                           selfattention_layer_type="sanm").eval()
    esd = synth.encoder_state_dict(ec, seed=100)
    enc.load_state_dict(esd, strict=True)
    g = torch.Generator().manual_seed(21)
    xs = torch.randn(2, 23, 560, generator=g) * 0.7
    lens = torch.tensor([23, 17], dtype=torch.int32)
    xs[1, 17:] = 0.0
    inter = []
    hooks = [m.register_forward_hook(lambda mod, i, o: inter.append(o[0].detach().clone()))
             for m in list(enc.encoders0) + list(enc.encoders)]
    with torch.no_grad():
        out, olens, _ = enc(xs.clone(), lens)
    for h in hooks:
        h.remove()
    save("encoder", xs=xs.numpy(), lens=lens.numpy(), out=out.numpy(), olens=olens.numpy(),
         block1=inter[0].numpy(), block3=inter[2].numpy(), seed=np.int64(100), checksum=np.float64(sd_checksum(esd)),
         cfg=json.dumps(ec))

    # ------------------------------------------------------------------ predictor + CIF
    pc = cfg["predictor"]
    pred = R["CifPredictorV2"](idim=pc["idim"], l_order=pc["l_order"], r_order=pc["r_order"],
                               threshold=pc["threshold"], tail_threshold=pc["tail_threshold"]).eval()
    psd = synth.predictor_state_dict(pc, seed=101)
    pred.load_state_dict(psd, strict=True)
    g = torch.Generator().manual_seed(22)
    hid = torch.randn(3, 40, 512, generator=g)
    plens = torch.tensor([40, 25, 16], dtype=torch.int32)   # every utterance must fire: the reference
    # raises IndexError at cif_predictor.py:887 when the LAST utterance of a batch fires no token
    mask = (torch.arange(40)[None, :] < plens[:, None]).float()[:, None, :]
    with torch.no_grad():
        emb, tok, alphas, peaks = pred(hid, None, mask)
    save("predictor", hidden=hid.numpy(), lens=plens.numpy(), embeds=emb.numpy(), token_num=tok.numpy(),
         alphas=alphas.numpy(), peaks=peaks.numpy(), seed=np.int64(101), cfg=json.dumps(pc))
    g = torch.Generator().manual_seed(23)
    al = torch.rand(24, 501, generator=g) * 0.6
    hh = torch.randn(24, 501, 4, generator=g)
    fr, fires = R["cif_v1"](hh, al, 1.0)
    fires2, fidx = R["cif_wo_hidden_v1"](al, 1.0, return_fire_idxs=True)
    save("cif", alphas=al.numpy(), hidden=hh.numpy(), frames=fr.numpy(), fires=fires.numpy(),
         fire_idx=fidx.numpy())

    # ------------------------------------------------------------------ decoder (2 blocks, vocab 97)
    dc = cfg["decoder"]
    dec = R["ParaformerSANMDecoder"](vocab_size=dc["vocab_size"], encoder_output_size=dc["encoder_output_size"],
                                     attention_heads=dc["attention_heads"], linear_units=dc["linear_units"],
                                     num_blocks=dc["num_blocks"], att_layer_num=dc["att_layer_num"],
                                     kernel_size=dc["kernel_size"], sanm_shfit=dc["sanm_shfit"]).eval()
    dsd = synth.decoder_state_dict(dc, seed=102, with_embed=True)
    dec.load_state_dict(dsd, strict=True)
    g = torch.Generator().manual_seed(24)
    memory = torch.randn(2, 30, 512, generator=g)
    mlens = torch.tensor([30, 21], dtype=torch.int32)
    embeds = torch.randn(2, 9, 512, generator=g)
    tlens = torch.tensor([9, 4], dtype=torch.int64)
    with torch.no_grad():
        logits, _ = dec(memory, mlens, embeds, tlens)
    save("decoder", memory=memory.numpy(), mem_lens=mlens.numpy(), embeds=embeds.numpy(), tok_lens=tlens.numpy(),
         logits=logits.numpy(), seed=np.int64(102), cfg=json.dumps(dc))

    # ------------------------------------------------------------------ end-to-end greedy ids (Paraformer.inference glue)
    full = synth.paraformer_state_dict(cfg, seed=7)
    enc.load_state_dict({k[len("encoder."):]: v for k, v in full.items() if k.startswith("encoder.")}, strict=True)
    pred.load_state_dict({k[len("predictor."):]: v for k, v in full.items() if k.startswith("predictor.")}, strict=True)
    dd = {k[len("decoder."):]: v for k, v in full.items() if k.startswith("decoder.")}
    dd["embed.0.weight"] = dsd["embed.0.weight"]
    dec.load_state_dict(dd, strict=True)
    g = torch.Generator().manual_seed(25)
    feats_in = torch.randn(3, 61, 560, generator=g) * 0.7
    flens = torch.tensor([61, 44, 30], dtype=torch.int32)
    for b in range(3):
        feats_in[b, flens[b]:] = 0
    with torch.no_grad():
        eo, el, _ = enc(feats_in.clone(), flens)
        m = (torch.arange(61)[None, :] < el[:, None]).float()[:, None, :]
        emb, tok, alphas, peaks = pred(eo, None, m)
        tok = tok.round().long()
        lg, _ = dec(eo, el, emb, tok)
        lp = torch.log_softmax(lg, dim=-1)
    ids = -np.ones((3, int(tok.max())), dtype=np.int64)
    for b in range(3):
        ids[b, : int(tok[b])] = lp[b, : int(tok[b])].argmax(-1).numpy()
    save("pipeline", feats=feats_in.numpy(), lens=flens.numpy(), enc=eo.numpy(), alphas=alphas.numpy(),
         peaks=peaks.numpy(), token_num=tok.numpy(), raw_ids=ids, seed=np.int64(7), cfg=json.dumps(cfg))

    # ------------------------------------------------------------------ SenseVoice encoder (2 + 1 tp blocks) + CTC
    sv = synth.tiny(synth.SENSEVOICE_SMALL, enc_blocks=2, vocab=211, tp_blocks=1)
    sc = sv["encoder"]
    senc = R["SenseVoiceEncoderSmall"](input_size=sc["input_size"], output_size=sc["output_size"],
                                       attention_heads=sc["attention_heads"], linear_units=sc["linear_units"],
                                       num_blocks=sc["num_blocks"], tp_blocks=sc["tp_blocks"], input_layer="pe",
                                       kernel_size=sc["kernel_size"], sanm_shfit=sc["sanm_shfit"],
                                       selfattention_layer_type="sanm").eval()
    ssd = synth.sensevoice_state_dict(sv, seed=9)
    senc.load_state_dict({k[len("encoder."):]: v for k, v in ssd.items() if k.startswith("encoder.")}, strict=True)
    ctc = R["CTC"](odim=sv["vocab_size"], encoder_output_size=sc["output_size"]).eval()
    ctc.load_state_dict({"ctc_lo.weight": ssd["ctc.ctc_lo.weight"], "ctc_lo.bias": ssd["ctc.ctc_lo.bias"]}, strict=False)
    g = torch.Generator().manual_seed(26)
    sx = torch.randn(2, 19, 560, generator=g) * 0.7
    slens = torch.tensor([19, 11], dtype=torch.int32)
    sx[1, 11:] = 0
    with torch.no_grad():
        so, sol = senc(sx.clone(), slens)
        slp = ctc.log_softmax(so)
    save("sensevoice", xs=sx.numpy(), lens=slens.numpy(), out=so.numpy(), olens=sol.numpy(),
         frame_ids=slp.argmax(-1).numpy(), seed=np.int64(9), cfg=json.dumps(sv))


if __name__ == "__main__":
    main()
