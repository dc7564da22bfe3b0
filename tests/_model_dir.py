"""Builds a FunASR-format model directory (config.yaml, model.pt, tokens.json, am.mvn) around a tiny synthetic
Paraformer so that the AutoModel plumbing can be exercised without the hub."""
import json
import os
import shutil
import wave

import numpy as np
import torch
import yaml

from funasr_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
VOCAB = ["<blank>", "<s>", "</s>"] + list("的一是了我不人在他有这个上们来到时大地为子中你说生国年着就那和要她出也得里后自以会") + \
        ["hello", "wor@@", "ld", "a", "b", "<unk>"]


def tiny_cfg():
    return synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=2, dec_blocks=2, vocab=len(VOCAB))


def make_model_dir(path: str, seed: int = 21) -> dict:
    os.makedirs(path, exist_ok=True)
    cfg = tiny_cfg()
    ec, dc, pc = cfg["encoder"], cfg["decoder"], cfg["predictor"]
    conf = {
        "model": "Paraformer",
        "model_conf": {"ctc_weight": 0.0, "lsm_weight": 0.1, "length_normalized_loss": True, "predictor_weight": 1.0,
                       "predictor_bias": 1, "sampling_ratio": 0.75},
        "encoder": "SANMEncoder",
        "encoder_conf": {"output_size": ec["output_size"], "attention_heads": ec["attention_heads"],
                         "linear_units": ec["linear_units"], "num_blocks": ec["num_blocks"], "dropout_rate": 0.1,
                         "positional_dropout_rate": 0.1, "attention_dropout_rate": 0.1, "input_layer": "pe",
                         "pos_enc_class": "SinusoidalPositionEncoder", "normalize_before": True,
                         "kernel_size": ec["kernel_size"], "sanm_shfit": ec["sanm_shfit"],
                         "selfattention_layer_type": "sanm"},
        "decoder": "ParaformerSANMDecoder",
        "decoder_conf": {"attention_heads": dc["attention_heads"], "linear_units": dc["linear_units"],
                         "num_blocks": dc["num_blocks"], "dropout_rate": 0.1, "positional_dropout_rate": 0.1,
                         "self_attention_dropout_rate": 0.1, "src_attention_dropout_rate": 0.1,
                         "att_layer_num": dc["att_layer_num"], "kernel_size": dc["kernel_size"],
                         "sanm_shfit": dc["sanm_shfit"]},
        "predictor": "CifPredictorV2",
        "predictor_conf": {"idim": pc["idim"], "threshold": 1.0, "l_order": 1, "r_order": 1, "tail_threshold": 0.45},
        "frontend": "WavFrontend",
        "frontend_conf": {"fs": 16000, "window": "hamming", "n_mels": 80, "frame_length": 25, "frame_shift": 10,
                          "lfr_m": 7, "lfr_n": 6},
        "tokenizer": "CharTokenizer",
        "tokenizer_conf": {"unk_symbol": "<unk>", "split_with_space": True},
    }
    with open(os.path.join(path, "config.yaml"), "w", encoding="utf-8") as f:
        yaml.safe_dump(conf, f, allow_unicode=True)
    sd = synth.paraformer_state_dict(cfg, seed=seed, cif_bias=0.3)
    sd["decoder.embed.0.weight"] = torch.zeros(len(VOCAB), dc["encoder_output_size"])
    torch.save({"state_dict": sd}, os.path.join(path, "model.pt"))          # wrapped like a training checkpoint
    with open(os.path.join(path, "tokens.json"), "w", encoding="utf-8") as f:
        json.dump(VOCAB, f, ensure_ascii=False)
    shutil.copy(os.path.join(HERE, "golden", "am.mvn"), os.path.join(path, "am.mvn"))
    return dict(cfg=cfg, sd=sd)


def write_wav(path: str, x: torch.Tensor, fs: int = 16000) -> np.ndarray:
    pcm = (x.clamp(-1, 1) * 32767.0).round().to(torch.int16).numpy()
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(fs)
        f.writeframes(pcm.tobytes())
    return pcm


def write_mvn(path: str, shift: torch.Tensor, scale: torch.Tensor) -> None:
    """Kaldi-nnet am.mvn (<AddShift> / <Rescale> rows, the format funasr/frontends/wav_frontend.py:15-43 parses)"""
    d = shift.numel()
    row = lambda v: " ".join(f"{float(x):.6g}" for x in v.tolist())                  # noqa: E731
    with open(path, "w", encoding="utf-8") as f:
        f.write(f"<Nnet> \n<Splice> {d} {d}\n[ 0 ]\n<AddShift> {d} {d} \n<LearnRateCoef> 0 [ {row(shift)} ]\n"
                f"<Rescale> {d} {d} \n<LearnRateCoef> 0 [ {row(scale)} ]\n</Nnet> \n")


VAD_ENCODER_CONF = dict(input_dim=400, input_affine_dim=140, fsmn_layers=4, linear_dim=250, proj_dim=128, lorder=20, rorder=0,
                        lstride=1, rstride=0, output_affine_dim=140, output_dim=248)


def make_vad_model_dir(path: str, encoder_sd: dict) -> None:
    """FSMN-VAD model directory in the hub layout: config.yaml (FsmnVADStreaming / FSMN / WavFrontendOnline), model.pt with
    `encoder.`-prefixed keys, am.mvn for the 400-dim LFR(5, 1) features"""
    os.makedirs(path, exist_ok=True)
    conf = {
        "model": "FsmnVADStreaming",
        "model_conf": {"sample_rate": 16000, "detect_mode": 1, "max_end_silence_time": 800, "max_start_silence_time": 3000,
                       "window_size_ms": 200, "sil_to_speech_time_thres": 150, "speech_to_sil_time_thres": 150,
                       "speech_2_noise_ratio": 1.0, "do_extend": 1, "lookback_time_start_point": 200,
                       "lookahead_time_end_point": 100, "max_single_segment_time": 60000, "snr_thres": -100.0,
                       "noise_frame_num_used_for_snr": 100, "decibel_thres": -100.0, "speech_noise_thres": 0.6,
                       "fe_prior_thres": 0.0001, "silence_pdf_num": 1, "sil_pdf_ids": [0], "frame_in_ms": 10,
                       "frame_length_ms": 25},
        "encoder": "FSMN",
        "encoder_conf": dict(VAD_ENCODER_CONF),
        "frontend": "WavFrontendOnline",
        "frontend_conf": {"fs": 16000, "window": "hamming", "n_mels": 80, "frame_length": 25, "frame_shift": 10,
                          "dither": 0.0, "lfr_m": 5, "lfr_n": 1},
    }
    with open(os.path.join(path, "config.yaml"), "w", encoding="utf-8") as f:
        yaml.safe_dump(conf, f)
    torch.save({"encoder." + k: v for k, v in encoder_sd.items()}, os.path.join(path, "model.pt"))
    write_mvn(os.path.join(path, "am.mvn"), torch.full((400,), -8.0), torch.full((400,), 0.25))


def make_punc_model_dir(path: str, vocab: list, enc_cfg: dict, sd: dict, punc_list: list, model: str = "CTTransformer",
                        encoder: str = "SANMEncoder") -> None:
    """CT-Transformer model directory in the hub layout (config.yaml naming CTTransformer / SANMEncoder / CharTokenizer --
    or the realtime pair CTTransformerStreaming / SANMVadEncoder --, model.pt, tokens.json)"""
    import json
    os.makedirs(path, exist_ok=True)
    conf = {
        "model": model,
        "model_conf": {"ignore_id": 0, "embed_unit": enc_cfg["input_size"], "att_unit": enc_cfg["output_size"],
                       "dropout_rate": 0.1, "punc_list": list(punc_list), "punc_weight": [1.0] * len(punc_list),
                       "sentence_end_id": 3},
        "encoder": encoder,
        "encoder_conf": dict(enc_cfg, input_layer="pe"),
        "tokenizer": "CharTokenizer",
        "tokenizer_conf": {"unk_symbol": "<unk>"},
    }
    with open(os.path.join(path, "config.yaml"), "w", encoding="utf-8") as f:
        yaml.safe_dump(conf, f, allow_unicode=True)
    with open(os.path.join(path, "tokens.json"), "w", encoding="utf-8") as f:
        json.dump(list(vocab), f, ensure_ascii=False)
    torch.save(dict(sd), os.path.join(path, "model.pt"))


def make_seaco_model_dir(path: str, seed: int = 31, no_bias: int = 5) -> dict:
    """SeACo-Paraformer model directory in the hub layout of `paraformer-zh` (config.yaml naming SeacoParaformer /
    CifPredictorV3 / the kernel-21 bias decoder, model.pt, tokens.json, am.mvn, seg_dict)"""
    from oracle import bicif_oracle as BO
    from oracle import seaco_oracle as SO
    base = make_model_dir(path, seed=seed)
    cfg = base["cfg"]
    cfg["predictor"] = dict(idim=512, threshold=1.0, l_order=1, r_order=1, tail_threshold=0.45, smooth_factor=1.0,
                            noise_threshold=0.0, smooth_factor2=0.25, noise_threshold2=0.01, upsample_times=3,
                            use_cif1_cnn=False, upsample_type="cnn_blstm")
    cfg["seaco_decoder"] = dict(SO.SEACO_DECODER)
    with open(os.path.join(path, "config.yaml"), encoding="utf-8") as f:
        conf = yaml.safe_load(f)
    conf["model"] = "SeacoParaformer"
    conf["model_conf"].update(inner_dim=512, bias_encoder_type="lstm", bias_encoder_bid=False, NO_BIAS=no_bias)
    conf["predictor"] = "CifPredictorV3"
    conf["predictor_conf"] = {k: v for k, v in cfg["predictor"].items() if k not in ("smooth_factor", "noise_threshold")}
    conf["seaco_decoder"] = "ParaformerSANMDecoder"
    conf["seaco_decoder_conf"] = {"attention_heads": 4, "linear_units": 1024, "num_blocks": 4, "dropout_rate": 0.1,
                                  "kernel_size": 21, "sanm_shfit": 0, "use_output_layer": False, "wo_input_layer": True}
    with open(os.path.join(path, "config.yaml"), "w", encoding="utf-8") as f:
        yaml.safe_dump(conf, f, allow_unicode=True)
    sd = SO.seaco_state_dict(cfg, seed, no_bias)
    torch.save(sd, os.path.join(path, "model.pt"))
    with open(os.path.join(path, "seg_dict"), "w", encoding="utf-8") as f:
        for ch in VOCAB[3:-6]:
            f.write(f"{ch} {ch}\n")
        f.write("world wor@@ ld\nhello hello\n")
    return dict(cfg=cfg, sd=sd, no_bias=no_bias)
