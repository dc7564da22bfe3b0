"""Generate tests/golden/streaming.npz by running the REFERENCE's own streaming classes.  TEST INFRASTRUCTURE ONLY.

Run in the build container (needs /root/reference):   python -m oracle.make_golden_streaming
Drives `ParaformerStreaming.inference` (funasr/models/paraformer_streaming/model.py:650-763) -- i.e. the reference's
WavFrontendOnline, SANMEncoderChunkOpt.forward_chunk, CifPredictorV2.forward_chunk, ParaformerSANMDecoder.forward_chunk
and the chunk loop itself -- over a seeded clip fed in two calls (mid-stream + final, with a tail shorter than
960 samples so the tail-chunk path runs), with seeded funasr_amd.synth weights loaded by load_state_dict(strict=True).
The only substitution: `torchaudio.compliance.kaldi.fbank` (third-party, absent) is the oracle's `kaldi_fbank`,
itself pinned to the reference-vendored kaldi-native-fbank in tests/test_oracle.py.
Per chunk it records the online features, the encoder output, the CIF state and the token ids.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from funasr_amd import synth  # noqa: E402
from oracle import paraformer_oracle as O  # noqa: E402
from oracle import ref_import  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SEED = 31
CHUNK = [0, 10, 5]


def stream_cfg():
    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=3, dec_blocks=2, vocab=97)
    cfg["decoder"]["sanm_shfit"] = 5                       # online decoder (paraformer_streaming/template.yaml:62)
    return cfg


def build():
    """the reference's ParaformerStreaming + WavFrontendOnline on the seeded weights -> (model, frontend, cfg, enc_conf)"""
    ref_import.install()
    import torchaudio.compliance.kaldi as kaldi           # the stub module

    def fbank(waveform, num_mel_bins=80, frame_length=25, frame_shift=10, dither=0.0, energy_floor=0.0,
              window_type="hamming", sample_frequency=16000, **kw):
        assert dither == 0.0 and window_type == "hamming" and energy_floor == 0.0
        return O.kaldi_fbank(waveform[0], num_mel_bins, float(frame_length), float(frame_shift), float(sample_frequency))

    kaldi.fbank = fbank
    import funasr.frontends.wav_frontend as wf
    wf.kaldi.fbank = fbank
    from funasr.models.paraformer_streaming.model import ParaformerStreaming
    import funasr.models.scama.encoder  # noqa: F401  (registers SANMEncoderChunkOpt)
    import funasr.models.paraformer.cif_predictor  # noqa: F401
    import funasr.models.paraformer.decoder  # noqa: F401

    torch.manual_seed(0)
    torch.set_num_threads(4)
    cfg = stream_cfg()
    ec, dc, pc = cfg["encoder"], cfg["decoder"], cfg["predictor"]
    enc_conf = dict(output_size=ec["output_size"], attention_heads=ec["attention_heads"], linear_units=ec["linear_units"],
                    num_blocks=ec["num_blocks"], dropout_rate=0.1, positional_dropout_rate=0.1, attention_dropout_rate=0.1,
                    input_layer="pe_online", normalize_before=True, kernel_size=11, sanm_shfit=0,
                    selfattention_layer_type="sanm", chunk_size=[12, 15], stride=[8, 10], pad_left=[0, 0],
                    encoder_att_look_back_factor=[4, 4], decoder_att_look_back_factor=[1, 1])
    dec_conf = dict(attention_heads=dc["attention_heads"], linear_units=dc["linear_units"], num_blocks=dc["num_blocks"],
                    dropout_rate=0.1, positional_dropout_rate=0.1, self_attention_dropout_rate=0.1,
                    src_attention_dropout_rate=0.1, att_layer_num=dc["att_layer_num"], kernel_size=11, sanm_shfit=5)
    pred_conf = dict(idim=512, threshold=1.0, l_order=1, r_order=1, tail_threshold=0.45)
    model = ParaformerStreaming(encoder="SANMEncoderChunkOpt", encoder_conf=enc_conf, decoder="ParaformerSANMDecoder",
                                decoder_conf=dec_conf, predictor="CifPredictorV2", predictor_conf=pred_conf,
                                ctc_weight=0.0, input_size=560, vocab_size=dc["vocab_size"], predictor_bias=1,
                                sampling_ratio=0.75)
    sd = synth.paraformer_state_dict(cfg, seed=SEED, cif_bias=0.6)
    sd["decoder.embed.0.weight"] = torch.zeros(dc["vocab_size"], 512)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("criterion") or "sampler" in k for k in missing), missing
    model.eval()

    am_mvn = os.path.join(GOLD, "am.mvn")
    frontend = wf.WavFrontendOnline(cmvn_file=am_mvn, fs=16000, window="hamming", n_mels=80, frame_length=25,
                                    frame_shift=10, lfr_m=7, lfr_n=6, dither=0.0)
    return model, frontend, cfg, enc_conf


class Tok:                                                 # ids2tokens -> the ids themselves (as strings)
    def ids2tokens(self, ids):
        return [str(int(i)) for i in ids]


def clip():
    # 5 full chunks + 700 leftover samples in call 1; call 2 (final) brings 3 more chunks and a 500-sample tail
    n1, n2 = 5 * 9600 + 700, 3 * 9600 - 200
    wav = synth.speech_like(n1 + n2, seed=77)
    pcm = (wav * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
    return pcm, pcm.to(torch.float32) / 32768.0, n1


def run(model, frontend, enc_conf, wav, n1, chunk, enc_lb, dec_lb):
    """two calls of the reference's inference (mid-stream + final) -> one record per chunk it decoded"""
    records = []
    orig_generate = model.generate_chunk

    def spy(speech, speech_lengths=None, **kw):
        cache = kw["cache"]
        rec = dict(feats=speech.detach().clone().numpy(), is_final=bool(kw.get("is_final", False)),
                   tail=bool(cache["encoder"]["tail_chunk"]))
        enc_holder = {}
        orig_encode = model.encode_chunk

        def enc_spy(s, sl, cache=None, **k2):
            out = orig_encode(s, sl, cache=cache, **k2)
            enc_holder["enc"] = out[0].detach().clone()
            return out
        model.encode_chunk = enc_spy
        toks = orig_generate(speech, speech_lengths, **kw)
        model.encode_chunk = orig_encode
        rec["enc"] = enc_holder["enc"].numpy()
        rec["tokens"] = [int(t) for t in toks]
        rec["cif_alphas"] = cache["encoder"]["cif_alphas"].detach().clone().reshape(-1).numpy()
        rec["cif_hidden"] = cache["encoder"]["cif_hidden"].detach().clone().reshape(-1).numpy()
        rec["start_idx"] = int(cache["encoder"]["start_idx"])
        records.append(rec)
        return toks

    model.generate_chunk = spy
    cache = {}
    kw = dict(chunk_size=chunk, encoder_chunk_look_back=enc_lb, decoder_chunk_look_back=dec_lb, device="cpu",
              encoder_conf=enc_conf, frontend_conf=dict(n_mels=80, lfr_m=7))
    try:
        with torch.no_grad():
            r1, _ = model.inference([wav[:n1]], key=["utt"], tokenizer=Tok(), frontend=frontend, cache=cache, is_final=False, **kw)
            r2, _ = model.inference([wav[n1:]], key=["utt"], tokenizer=Tok(), frontend=frontend, cache=cache, is_final=True, **kw)
    finally:
        model.generate_chunk = orig_generate
    print("chunk", chunk, "look back", enc_lb, dec_lb, "| call 1 text:", r1[0]["text"], "| call 2 text:", r2[0]["text"])
    print("chunks:", len(records), "tokens/chunk:", [len(r["tokens"]) for r in records], "tail:", [r["tail"] for r in records])
    return records


def main():
    model, frontend, cfg, enc_conf = build()
    pcm, wav, n1 = clip()
    records = run(model, frontend, enc_conf, wav, n1, CHUNK, 4, 1)
    arrs = dict(pcm=pcm.numpy(), n1=np.int64(n1), seed=np.int64(SEED), cif_bias=np.float32(0.6),
                config=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8), n_chunks=np.int64(len(records)))
    for i, r in enumerate(records):
        arrs[f"feats_{i}"] = r["feats"].astype(np.float32)
        arrs[f"enc_{i}"] = r["enc"].astype(np.float32)
        arrs[f"tokens_{i}"] = np.asarray(r["tokens"], dtype=np.int64)
        arrs[f"cif_alphas_{i}"] = r["cif_alphas"].astype(np.float32)
        arrs[f"cif_hidden_{i}"] = r["cif_hidden"].astype(np.float32)
        arrs[f"flags_{i}"] = np.asarray([r["is_final"], r["tail"], r["start_idx"]], dtype=np.int64)
    path = os.path.join(GOLD, "streaming.npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KB)")


def main_geometries():
    """other chunk sizes / look-back settings of the SAME clip and weights -> tests/golden/streaming_geometries.npz (per
    session: the online features, token ids and flags of every chunk)"""
    model, frontend, cfg, enc_conf = build()
    _, wav, n1 = clip()
    arrs, sessions = {}, []
    # the last two bring more than 24 possible fires per step (the streaming decoder's former row cap)
    for si, (chunk, enc_lb, dec_lb) in enumerate((([5, 10, 5], 2, 2), ([0, 8, 4], 1, 0), ([0, 10, 5], 0, 1),
                                                  ([0, 20, 10], 2, 1), ([0, 16, 8], 1, 1))):
        frontend.cache_reset() if hasattr(frontend, "cache_reset") else None
        records = run(model, frontend, enc_conf, wav, n1, chunk, enc_lb, dec_lb)
        sessions.append(dict(chunk=chunk, enc_lb=enc_lb, dec_lb=dec_lb, n_chunks=len(records)))
        for i, r in enumerate(records):
            arrs[f"s{si}_feats_{i}"] = r["feats"].astype(np.float32)
            arrs[f"s{si}_tokens_{i}"] = np.asarray(r["tokens"], dtype=np.int64)
            arrs[f"s{si}_flags_{i}"] = np.asarray([r["is_final"], r["tail"], r["start_idx"]], dtype=np.int64)
    path = os.path.join(GOLD, "streaming_geometries.npz")
    np.savez_compressed(path, sessions=json.dumps(sessions), **arrs)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KB)")


def main_right0():
    """chunk_size[2] == 0 with an encoder look-back: the reference's `[: -chunk_size[2]]` slice is empty there, i.e. nothing is
    ever cached (sanm/attention.py:345-346) -> tests/golden/streaming_right0.npz (features, encoder window, tokens, flags)"""
    model, frontend, cfg, enc_conf = build()
    _, wav, n1 = clip()
    arrs, sessions = {}, []
    for si, (chunk, enc_lb, dec_lb) in enumerate((([5, 11, 0], 3, 0), ([0, 12, 0], 2, 1))):
        records = run(model, frontend, enc_conf, wav, n1, chunk, enc_lb, dec_lb)
        sessions.append(dict(chunk=chunk, enc_lb=enc_lb, dec_lb=dec_lb, n_chunks=len(records)))
        for i, r in enumerate(records):
            arrs[f"s{si}_feats_{i}"] = r["feats"].astype(np.float32)
            arrs[f"s{si}_enc_{i}"] = r["enc"].astype(np.float32)
            arrs[f"s{si}_tokens_{i}"] = np.asarray(r["tokens"], dtype=np.int64)
            arrs[f"s{si}_flags_{i}"] = np.asarray([r["is_final"], r["tail"], r["start_idx"]], dtype=np.int64)
    path = os.path.join(GOLD, "streaming_right0.npz")
    np.savez_compressed(path, sessions=json.dumps(sessions), **arrs)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KB)")


if __name__ == "__main__":
    if "--geometries" in sys.argv:
        main_geometries()
    elif "--right0" in sys.argv:
        main_right0()
    else:
        main()
