"""state_dict layouts (parameter names and shapes) of the REFERENCE model classes on this path, built from small configs
(build container only; TEST INFRASTRUCTURE) -> tests/golden/ref_state_dicts.json. tests/test_state_dicts.py builds the
host-side mirrors from the same configs and requires the same names and shapes: a published model.pt loads unchanged.

    python oracle/make_golden_state_dicts.py
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_import  # noqa: E402

ENC = dict(output_size=512, attention_heads=4, linear_units=2048, num_blocks=2, dropout_rate=0.1, input_layer="pe",
           pos_enc_class="SinusoidalPositionEncoder", normalize_before=True, kernel_size=11, sanm_shfit=0,
           selfattention_layer_type="sanm")
DEC = dict(attention_heads=4, linear_units=2048, num_blocks=2, att_layer_num=2, kernel_size=11, sanm_shfit=0)
V2 = dict(idim=512, threshold=1.0, l_order=1, r_order=1, tail_threshold=0.45)
V3 = dict(V2, smooth_factor2=0.25, noise_threshold2=0.01, upsample_times=3, use_cif1_cnn=False, upsample_type="cnn_blstm")
SEACO_DEC = dict(attention_heads=4, linear_units=1024, num_blocks=4, kernel_size=21, sanm_shfit=0, use_output_layer=False,
                 wo_input_layer=True)
VAD_ENC = dict(input_dim=400, input_affine_dim=140, fsmn_layers=4, linear_dim=250, proj_dim=128, lorder=20, rorder=0, lstride=1,
               rstride=0, output_affine_dim=140, output_dim=248)
PUNC_ENC = dict(input_size=256, output_size=256, attention_heads=8, linear_units=1024, num_blocks=2, dropout_rate=0.1,
                input_layer="pe", pos_enc_class="SinusoidalPositionEncoder", normalize_before=True, kernel_size=11, sanm_shfit=0,
                selfattention_layer_type="sanm", padding_idx=0)
CONFIGS = {
    "Paraformer": dict(encoder="SANMEncoder", encoder_conf=ENC, decoder="ParaformerSANMDecoder", decoder_conf=DEC,
                       predictor="CifPredictorV2", predictor_conf=V2, input_size=560, vocab_size=97, ctc_weight=0.0),
    "BiCifParaformer": dict(encoder="SANMEncoder", encoder_conf=ENC, decoder="ParaformerSANMDecoder", decoder_conf=DEC,
                            predictor="CifPredictorV3", predictor_conf=V3, input_size=560, vocab_size=97, ctc_weight=0.0),
    "SeacoParaformer": dict(encoder="SANMEncoder", encoder_conf=ENC, decoder="ParaformerSANMDecoder", decoder_conf=DEC,
                            predictor="CifPredictorV3", predictor_conf=V3, seaco_decoder="ParaformerSANMDecoder",
                            seaco_decoder_conf=SEACO_DEC, input_size=560, vocab_size=97, ctc_weight=0.0, inner_dim=512,
                            bias_encoder_type="lstm"),
    "ParaformerStreaming": dict(encoder="SANMEncoderChunkOpt",
                                encoder_conf=dict(ENC, chunk_size=[12, 15], stride=[8, 10], pad_left=[0, 0],
                                                  encoder_att_look_back_factor=[4, 4], decoder_att_look_back_factor=[1, 1]),
                                decoder="ParaformerSANMDecoder", decoder_conf=dict(DEC, sanm_shfit=5),
                                predictor="CifPredictorV2", predictor_conf=V2, input_size=560, vocab_size=97, ctc_weight=0.0),
    "SenseVoiceSmall": dict(encoder="SenseVoiceEncoderSmall",
                            encoder_conf=dict(output_size=512, attention_heads=4, linear_units=2048, num_blocks=2, tp_blocks=1,
                                              dropout_rate=0.1, input_layer="pe", pos_enc_class="SinusoidalPositionEncoder",
                                              normalize_before=True, kernel_size=11, sanm_shfit=0, selfattention_layer_type="sanm"),
                            input_size=560, vocab_size=97),
    "FsmnVADStreaming": dict(encoder="FSMN", encoder_conf=VAD_ENC),
    "CTTransformer": dict(encoder="SANMEncoder", encoder_conf=PUNC_ENC, vocab_size=50, punc_list=["<unk>", "_", "，", "。", "？", "、"],
                          embed_unit=256, att_unit=256, ignore_id=0, sentence_end_id=3),
    "CTTransformerStreaming": dict(encoder="SANMVadEncoder", encoder_conf=dict(PUNC_ENC, num_blocks=3, sanm_shfit=5), vocab_size=50,
                                   punc_list=["<unk>", "_", "，", "。", "？", "、"], embed_unit=256, att_unit=256, ignore_id=0,
                                   sentence_end_id=3),
    "ContextualParaformer": dict(encoder="SANMEncoder", encoder_conf=ENC, decoder="ContextualParaformerDecoder", decoder_conf=DEC,
                                 predictor="CifPredictorV2", predictor_conf=V2, input_size=560, vocab_size=97, ctc_weight=0.0,
                                 inner_dim=512, bias_encoder_type="lstm"),
    "Paraformer+CTC": dict(encoder="SANMEncoder", encoder_conf=ENC, decoder="ParaformerSANMDecoder", decoder_conf=DEC,
                           predictor="CifPredictorV2", predictor_conf=V2, input_size=560, vocab_size=97, ctc_weight=0.3),
}


def main():
    ref_import.install()
    import funasr.models.sanm.encoder  # noqa: F401
    import funasr.models.scama.encoder  # noqa: F401
    import funasr.models.paraformer.decoder  # noqa: F401
    import funasr.models.paraformer.cif_predictor  # noqa: F401
    import funasr.models.bicif_paraformer.cif_predictor  # noqa: F401
    import funasr.models.fsmn_vad_streaming.encoder  # noqa: F401
    import funasr.models.sense_voice.model  # noqa: F401
    from funasr.models.bicif_paraformer.model import BiCifParaformer
    from funasr.models.ct_transformer.model import CTTransformer
    from funasr.models.ct_transformer_streaming.model import CTTransformerStreaming
    from funasr.models.contextual_paraformer.model import ContextualParaformer
    import funasr.models.contextual_paraformer.decoder  # noqa: F401
    import funasr.models.ct_transformer_streaming.encoder  # noqa: F401
    from funasr.models.fsmn_vad_streaming.model import FsmnVADStreaming
    from funasr.models.paraformer.model import Paraformer
    from funasr.models.paraformer_streaming.model import ParaformerStreaming
    from funasr.models.seaco_paraformer.model import SeacoParaformer
    from funasr.models.sense_voice.model import SenseVoiceSmall
    classes = dict(Paraformer=Paraformer, BiCifParaformer=BiCifParaformer, SeacoParaformer=SeacoParaformer,
                   ParaformerStreaming=ParaformerStreaming, SenseVoiceSmall=SenseVoiceSmall, FsmnVADStreaming=FsmnVADStreaming,
                   CTTransformer=CTTransformer, CTTransformerStreaming=CTTransformerStreaming,
                   ContextualParaformer=ContextualParaformer)
    classes["Paraformer+CTC"] = Paraformer                  # the beam-search configuration: a CTC head next to the decoder
    out = {}
    for name, conf in CONFIGS.items():
        if conf is None or name not in classes:
            continue
        try:
            model = classes[name](**conf)
        except Exception as e:  # noqa: BLE001
            print(name, "could not be built from the reference:", type(e).__name__, e)
            continue
        out[name] = {"config": conf, "state_dict": {k: list(v.shape) for k, v in model.state_dict().items()}}
        print(name, len(out[name]["state_dict"]), "tensors")
    path = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_state_dicts.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
