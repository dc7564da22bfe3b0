#!/usr/bin/env python3
"""Golden vectors for the REALTIME punctuation path, produced by the REFERENCE's own classes
(funasr/models/ct_transformer_streaming/model.py `CTTransformerStreaming` over encoder.py `SANMVadEncoder`): TEST
INFRASTRUCTURE, build container only; writes tests/golden/punc_streaming.npz.
  * network: punc_forward logits of the reference class on seeded weights (d_model 256, 8 heads of 32, 3 blocks, causal
    FSMN: sanm_shfit 5) for ragged batches with VAD positions inside, at the edges of and beyond the text -- also pins
    oracle/punc_oracle.py `punc_forward_vad`;
  * sessions: the reference's inference() called chunk after chunk with ONE cache dict, driven with an INJECTED network
    (oracle.punc_oracle.injected_marks): texts per call, returned text / punc_array and the carried `pre_text` after every
    call -- Chinese, English and mixed chunks from 1 to ~70 words, long runs without any sentence end;
  * end to end: sessions with the real seeded network.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import punc_oracle, ref_import  # noqa: E402
from oracle.make_golden_punc import VOCAB, random_text  # noqa: E402

ENC = dict(input_size=256, output_size=256, attention_heads=8, linear_units=1024, num_blocks=3, kernel_size=11, sanm_shfit=5)


def main():
    ref_import.install()
    import funasr.models.ct_transformer_streaming.encoder  # noqa: F401  (registers SANMVadEncoder)
    from funasr.models.ct_transformer_streaming.model import CTTransformerStreaming
    from funasr.tokenizer.char_tokenizer import CharTokenizer
    torch.set_num_threads(4)
    enc_conf = dict(input_size=256, output_size=256, attention_heads=8, linear_units=1024, num_blocks=3, dropout_rate=0.1,
                    positional_dropout_rate=0.1, attention_dropout_rate=0.0, input_layer="pe",
                    pos_enc_class="SinusoidalPositionEncoder", normalize_before=True, kernel_size=11, sanm_shfit=5,
                    selfattention_layer_type="sanm", padding_idx=0)
    model = CTTransformerStreaming(encoder="SANMVadEncoder", encoder_conf=enc_conf, vocab_size=len(VOCAB),
                                   punc_list=punc_oracle.PUNC_LIST, embed_unit=256, att_unit=256, ignore_id=0,
                                   sentence_end_id=3).eval()
    sd = punc_oracle.synthetic_state_dict(len(VOCAB), ENC, seed=17)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and not [m for m in missing if "embed.1" not in m], (missing, unexpected)
    tok = CharTokenizer(token_list=VOCAB, unk_symbol="<unk>")
    rng = np.random.default_rng(21)
    # ---- network
    nets, worst = [], 0.0
    for lens, vads in (([41, 17, 2], [7, 0, 1]), ([30, 30, 9], [45, 29, 5]), ([1], [0]), ([64, 12], [2, 11])):
        T = max(lens)
        ids = torch.from_numpy(rng.integers(1, len(VOCAB) - 1, size=(len(lens), T)).astype(np.int64))
        tl = torch.tensor(lens, dtype=torch.int32)
        vi = torch.tensor(vads, dtype=torch.int32)
        with torch.no_grad():
            y, _ = model.punc_forward(ids, tl, vi)
        mine = punc_oracle.punc_forward_vad(ids, tl, vads, sd, ENC)
        worst = max(worst, max((mine[b, : lens[b]] - y[b, : lens[b]]).abs().max().item() for b in range(len(lens))))
        nets.append(dict(ids=ids.numpy().tolist(), lens=lens, vad=vads, logits=y.numpy().tolist()))
    assert worst < 2e-5, f"oracle differs from the reference network by {worst}"
    # ---- sessions with an injected network
    sessions = []
    real_forward = model.punc_forward
    for si in range(30):
        kind = ("zh", "en", "mix")[si % 3]
        never_end = si % 6 == 4
        split = 20 if si % 4 else int(rng.choice([5, 8, 33]))

        def fake(text, text_lengths, vad_indexes, _ne=never_end, **kw):
            m = punc_oracle.injected_marks(text[0].cpu().numpy(), _ne)
            return torch.nn.functional.one_hot(torch.from_numpy(m), 6).float()[None], None
        model.punc_forward = fake
        cache, calls = {}, []
        for ci in range(int(rng.integers(2, 7))):
            n = int(rng.choice([1, 2, 3, 7, 19, 20, 21, 45, 70])) if never_end or ci % 2 else int(rng.integers(1, 30))
            text = random_text(rng, n, kind)
            res, _ = model.inference([text], key=["k"], tokenizer=tok, device="cpu", split_size=split, cache=cache)
            pa = res[0]["punc_array"]
            calls.append(dict(text=text, out=res[0]["text"], punc_array=[int(x) for x in pa.reshape(-1).tolist()],
                              punc_shape=list(pa.shape), pre_text=list(cache["pre_text"])))
        sessions.append(dict(never_end=never_end, split_size=split, calls=calls))
    model.punc_forward = real_forward
    # ---- end to end with the real network
    e2e = []
    for chunks in (["今天天气真不错", "我们一起去公园散步吧", "欢迎大家来体验"], ["hello world i am", "fine thanks ok", "the quick brown fox"],
                   [random_text(rng, 12, "mix"), random_text(rng, 33, "mix"), random_text(rng, 5, "zh"), random_text(rng, 64, "zh")]):
        cache, calls = {}, []
        for text in chunks:
            with torch.no_grad():
                res, _ = model.inference([text], key=["k"], tokenizer=tok, device="cpu", cache=cache)
            pa = res[0]["punc_array"]
            calls.append(dict(text=text, out=res[0]["text"], punc_array=[int(x) for x in pa.reshape(-1).tolist()],
                              punc_shape=list(pa.shape), pre_text=list(cache["pre_text"])))
        e2e.append(calls)
    out = os.path.join(os.path.dirname(HERE), "tests", "golden", "punc_streaming.npz")
    np.savez_compressed(out, seed=17, vocab=json.dumps(VOCAB, ensure_ascii=False), enc_cfg=json.dumps(ENC),
                        nets=json.dumps(nets), sessions=json.dumps(sessions, ensure_ascii=False), e2e=json.dumps(e2e, ensure_ascii=False))
    print(f"wrote {out}: oracle vs reference network {worst:.2e}; {len(sessions)} sessions "
          f"({sum(len(s['calls']) for s in sessions)} calls), {len(e2e)} end-to-end sessions")
    print("e.g.", [c["out"] for c in e2e[0]], "|", [c["out"][:30] for c in sessions[2]["calls"]])


if __name__ == "__main__":
    main()
