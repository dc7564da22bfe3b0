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

from .hip_module import HostCopyRing, host_i32
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

    # The chain in two halves around its one host wait (the CIF token counts size the decoder and the timestamp scan, like the
    # .item() at cif_predictor.py:311): `enqueue_begin` puts encoder + predictor and the counts' D2H copy on the stream and returns
    # at once, `enqueue_finish` waits for the counts and enqueues the rest. A loop over batches calls begin(i + 1) before finish(i)
    # (AutoModel.inference through inference_begin / inference_launch): the host then never sits in that wait with nothing queued.
    def enqueue_begin(self, speech: torch.Tensor, speech_lengths, return_intermediate: bool = False) -> dict:
        enc, olens = self.encode(speech, speech_lengths)
        return dict(enc=enc, olens=olens, want=return_intermediate, pred=self.predictor.forward_begin(enc, olens))

    def enqueue_finish(self, half: dict) -> dict:
        enc, olens = half["enc"], half["olens"]
        embeds, token_num, alphas, peaks = self.predictor.forward_finish(half["pred"])
        half.update(alphas=alphas, peaks=peaks)
        tok = [int(round(v)) for v in token_num.tolist()]
        ids, us_alphas, us_peaks = None, None, None
        if max(tok) >= 1:                                            # model.py:343-344
            ids, _ = self.decoder.greedy(enc, olens, embeds, tok)
            _, _, us_alphas, us_peaks = self.calc_predictor_timestamp(enc, olens, tok)
        pending = dict(tok=tok, ids=ids, B=enc.shape[0], extra=dict(us_alphas=us_alphas, us_peaks=us_peaks, olens=olens))
        if ids is not None:
            # the batch's D2H copies go on the stream NOW, into pinned memory: collect() then waits for THIS batch only (a .cpu()
            # at collect time is ordered behind everything enqueued since -- the next batches of an overlapped loop)
            ring = self.__dict__.setdefault("_host_ring", HostCopyRing())
            pending["ids_host"] = (ids, ring.start(ids))
            pending["us_host"] = (ring.start(us_alphas), ring.start(us_peaks))
        if half["want"]:
            pending["extra"].update(enc=enc, embeds=embeds, alphas=half["alphas"], peaks=half["peaks"])
        return pending

    def enqueue_features(self, speech: torch.Tensor, speech_lengths, return_intermediate: bool = False):
        return self.enqueue_finish(self.enqueue_begin(speech, speech_lengths, return_intermediate))

    _split_enqueue_features = enqueue_features      # inference_begin uses the halves only while no subclass overrides the chain

    def collect(self, pending: dict) -> dict:
        out = super().collect(pending)
        if out.get("us_alphas") is not None:                         # one more small D2H pair per batch
            early = pending.get("us_host")
            if early is not None:
                out["us_alphas_host"], out["us_peaks_host"] = HostCopyRing.wait(early[0]).clone(), HostCopyRing.wait(early[1]).clone()
            else:
                out["us_alphas_host"], out["us_peaks_host"] = out["us_alphas"].cpu(), out["us_peaks"].cpu()
            out["olens_host"] = host_i32(out["olens"])[1]
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
