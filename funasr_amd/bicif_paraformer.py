"""BiCifParaformer on gfx950: Paraformer with the timestamp predictor.

Host-side mirror of `BiCifParaformer` (funasr/models/bicif_paraformer/model.py:45-441, `model_classes["BiCifParaformer"]`;
the base of SeACo-Paraformer, the model behind the `paraformer-zh` alias): same constructor, same state_dict layout
(predictor.upsample_cnn.*, predictor.blstm.*, predictor.cif_output2.* on top of Paraformer's keys) and the same
`inference()` results `{"key", "text", "timestamp"}`: after the greedy decode the predictor's second head gives frame
weights on a 3x finer time axis (`calc_predictor_timestamp`, :166-178), `ts_prediction_lfr6_standard` turns their fires
into token spans and `sentence_postprocess` merges them with the text (:374-388). All device work (encoder, CIF, decoder,
upsampling GEMM, BLSTM, second head, scan) is enqueued on the current HIP stream before anything is read back.
"""
from __future__ import annotations

import torch

from .paraformer import Paraformer
from .register import tables
from .timestamps import cif_token_spans
from .tokenizer import sentence_postprocess


@tables.register("model_classes", "BiCifParaformer")
class BiCifParaformer(Paraformer):
    _always_timestamps = True
    _unwrap_key_lists = False       # bicif_paraformer/model.py:347-414 and seaco_paraformer/model.py use `key[i]` as it comes

    @classmethod
    def from_config(cls, cfg: dict) -> "BiCifParaformer":
        ec = dict(cfg["encoder"])
        input_size = ec.pop("input_size")
        dc = dict(cfg["decoder"])
        vocab = dc.pop("vocab_size")
        dc.pop("encoder_output_size", None)
        return cls(encoder="SANMEncoder", encoder_conf=dict(ec, input_layer="pe"), decoder="ParaformerSANMDecoder",
                   decoder_conf=dc, predictor="CifPredictorV3", predictor_conf=dict(cfg["predictor"]), ctc_weight=0.0,
                   input_size=input_size, vocab_size=vocab)

    def calc_predictor(self, encoder_out, encoder_out_lens):
        outs = self.predictor(encoder_out, None, None, ignore_id=self.ignore_id, lengths=encoder_out_lens)
        return outs[0], outs[1], outs[2], outs[3]

    def calc_predictor_timestamp(self, encoder_out, encoder_out_lens, token_num):
        return self.predictor.get_upsample_timestamp(encoder_out, None, token_num, lengths=encoder_out_lens)

    def enqueue_features(self, speech: torch.Tensor, speech_lengths, return_intermediate: bool = False):
        enc, olens = self.encode(speech, speech_lengths)
        embeds, token_num, alphas, peaks = self.calc_predictor(enc, olens)
        tok = [int(round(v)) for v in token_num.tolist()]
        ids, us_alphas, us_peaks = None, None, None
        if max(tok) >= 1:                                            # model.py:343-344
            ids, _ = self.decoder.greedy(enc, olens, embeds, tok)
            _, _, us_alphas, us_peaks = self.calc_predictor_timestamp(enc, olens, tok)
        pending = dict(tok=tok, ids=ids, B=enc.shape[0], extra=dict(us_alphas=us_alphas, us_peaks=us_peaks, olens=olens))
        if return_intermediate:
            pending["extra"].update(enc=enc, embeds=embeds, alphas=alphas, peaks=peaks)
        return pending

    def collect(self, pending: dict) -> dict:
        out = super().collect(pending)
        if out.get("us_alphas") is not None:                         # one more small D2H pair per batch
            out["us_alphas_host"], out["us_peaks_host"] = out["us_alphas"].cpu(), out["us_peaks"].cpu()
            out["olens_host"] = [int(v) for v in (out["olens"].tolist() if isinstance(out["olens"], torch.Tensor) else out["olens"])]
        return out

    def _token_timestamps(self, res: dict, i: int, token, kwargs):
        # the reference slices the upsampled weights with a literal 3 (bicif_paraformer/model.py:403-404, seaco_paraformer/
        # model.py:554-555) -- the upsample factor of the published models -- whatever `upsample_times` says; kept
        n = res["olens_host"][i] * 3
        return cif_token_spans(res["us_alphas_host"][i][:n], res["us_peaks_host"][i][:n], list(token),
                               vad_offset=kwargs.get("begin_time", 0))

    def _postprocess(self, tokenizer, token, text, stamps):
        text, stamps, _ = sentence_postprocess(token, stamps)        # unconditional in the reference (:382-384)
        return text, stamps
