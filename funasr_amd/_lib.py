"""ctypes binding of the C ABI declared in include/paraformer_hip.h.

The shared library is the product: there is no Python/CPU fallback. If it is missing, ``load()`` raises with the
build command; if no GPU is visible, every ``pf_*_create`` fails and the wrappers raise ``HipRuntimeError``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# PF_LIB_PATH (environment): another build of the same library (A/B runs of compiler flags; tools/repro_pk_aggressors.py)
LIB_PATH = os.environ.get("PF_LIB_PATH") or os.path.join(_HERE, "libparaformer_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

ABI_VERSION = 5            # PF_ABI_VERSION of include/paraformer_hip.h

_lock = threading.Lock()
_lib = None


class HipRuntimeError(RuntimeError):
    """A pf_* entry point returned a non-zero status."""


class pf_frontend_config(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_int32), ("frame_length", C.c_int32), ("frame_shift", C.c_int32),
        ("n_mels", C.c_int32), ("lfr_m", C.c_int32), ("lfr_n", C.c_int32),
        ("low_freq", C.c_float), ("high_freq", C.c_float), ("preemph", C.c_float), ("upscale", C.c_float),
    ]


class pf_encoder_config(C.Structure):
    _fields_ = [
        ("input_dim", C.c_int32), ("d_model", C.c_int32), ("n_heads", C.c_int32), ("ffn_dim", C.c_int32),
        ("n_blocks", C.c_int32), ("tp_blocks", C.c_int32), ("kernel_size", C.c_int32), ("sanm_shift", C.c_int32),
        ("ln_eps", C.c_float),
    ]


class pf_predictor_v3_config(C.Structure):
    _fields_ = [
        ("upsample_times", C.c_int32), ("upsample_type", C.c_int32), ("use_cif1_cnn", C.c_int32),
        ("smooth_factor2", C.c_float), ("noise_threshold2", C.c_float),
    ]


class pf_predictor_config(C.Structure):
    _fields_ = [
        ("d_model", C.c_int32), ("l_order", C.c_int32), ("r_order", C.c_int32),
        ("threshold", C.c_float), ("smooth_factor", C.c_float), ("noise_threshold", C.c_float),
        ("tail_threshold", C.c_float), ("tail_mask", C.c_int32),
    ]


class pf_decoder_config(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32), ("d_model", C.c_int32), ("n_heads", C.c_int32), ("ffn_dim", C.c_int32),
        ("n_blocks", C.c_int32), ("kernel_size", C.c_int32), ("sanm_shift", C.c_int32), ("ln_eps", C.c_float),
    ]


class pf_vad_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("input_dim", "input_affine_dim", "fsmn_layers", "linear_dim", "proj_dim", "lorder",
                                          "rorder", "lstride", "rstride", "output_affine_dim", "output_dim")]


class pf_vad_options(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("sample_rate", "detect_mode", "max_end_silence_time", "max_start_silence_time",
                                          "window_size_ms", "sil_to_speech_time_thres", "speech_to_sil_time_thres", "do_extend",
                                          "lookback_time_start_point", "lookahead_time_end_point", "max_single_segment_time",
                                          "noise_frame_num_used_for_snr", "frame_in_ms", "frame_length_ms")] + \
               [(n, C.c_double) for n in ("speech_2_noise_ratio", "snr_thres", "decibel_thres", "speech_noise_thres",
                                           "fe_prior_thres")]


class pf_stream_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_streams", "chunk_left", "chunk_cur", "chunk_right", "enc_look_back",
                                          "dec_look_back", "max_frames", "max_tokens", "use_graph")]


_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_pi32 = C.POINTER(C.c_int32)

# name -> (restype, argtypes); mirrors include/paraformer_hip.h declaration by declaration
SIGNATURES = {
    "pf_last_error": (C.c_char_p, []),
    "pf_abi_version": (C.c_int, []),
    "pf_device_count": (C.c_int, []),
    "pf_set_concurrency_guard": (C.c_int, [_i32]),
    "pf_concurrency_guard": (C.c_int, []),
    "pf_frontend_create": (_vp, [C.POINTER(pf_frontend_config)]),
    "pf_frontend_destroy": (None, [_vp]),
    "pf_frontend_set_cmvn": (C.c_int, [_vp, _vp, _vp, _i32]),
    "pf_frontend_set_dither": (C.c_int, [_vp, _f32, C.c_uint64]),
    "pf_frontend_set_verify": (C.c_int, [_vp, _i32]),
    "pf_frontend_faults": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "pf_frontend_fault_log": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "pf_frontend_set_tables": (C.c_int, [_vp, _vp, _vp]),
    "pf_utterance_mvn": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "pf_global_mvn": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp]),
    "pf_frontend_set_window": (C.c_int, [_vp, C.c_char_p, _f32]),
    "pf_frontend_set_snip_edges": (C.c_int, [_vp, _i32]),
    "pf_frontend_num_fbank_frames": (_i32, [_vp, _i64]),
    "pf_frontend_num_frames": (_i32, [_vp, _i64]),
    "pf_frontend_forward": (C.c_int, [_vp, _vp, _i64, _pi32, _i32, _vp, _i32, _pi32, _vp, _vp]),
    "pf_encoder_create": (_vp, [C.POINTER(pf_encoder_config)]),
    "pf_encoder_destroy": (None, [_vp]),
    "pf_encoder_set_tensor": (C.c_int, [_vp, C.c_char_p, _vp, _i64]),
    "pf_encoder_missing": (C.c_int, [_vp]),
    "pf_encoder_set_precision": (C.c_int, [_vp, _i32]),
    "pf_encoder_set_row_packing": (C.c_int, [_vp, _i32]),
    "pf_encoder_set_option": (C.c_int, [_vp, C.c_char_p, _i32]),
    "pf_encoder_debug_poison": (C.c_int, [_vp, _i32]),
    "pf_decoder_debug_poison": (C.c_int, [_vp, _i32]),
    "pf_predictor_debug_poison": (C.c_int, [_vp, _i32]),
    "pf_encoder_set_vad_mask": (C.c_int, [_vp, _vp, _i32]),
    "pf_encoder_forward": (C.c_int, [_vp, _vp, _pi32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "pf_predictor_create": (_vp, [C.POINTER(pf_predictor_config)]),
    "pf_predictor_destroy": (None, [_vp]),
    "pf_predictor_set_tensor": (C.c_int, [_vp, C.c_char_p, _vp, _i64]),
    "pf_predictor_missing": (C.c_int, [_vp]),
    "pf_predictor_alphas": (C.c_int, [_vp, _vp, _pi32, _i32, _i32, _vp, _vp, _pi32, _vp]),
    "pf_predictor_embeds": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "pf_predictor_alphas_begin": (C.c_int, [_vp, _i32, _vp, _pi32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "pf_predictor_embeds_slot": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp]),
    "pf_predictor_create_v3": (_vp, [C.POINTER(pf_predictor_config), C.POINTER(pf_predictor_v3_config)]),
    "pf_predictor_timestamp": (C.c_int, [_vp, _vp, _pi32, _pi32, _i32, _i32, _vp, _vp, _vp]),
    "pf_decoder_create": (_vp, [C.POINTER(pf_decoder_config)]),
    "pf_decoder_destroy": (None, [_vp]),
    "pf_decoder_set_decoders2": (C.c_int, [_vp, _i32]),
    "pf_decoder_set_tensor": (C.c_int, [_vp, C.c_char_p, _vp, _i64]),
    "pf_decoder_missing": (C.c_int, [_vp]),
    "pf_decoder_set_precision": (C.c_int, [_vp, _i32]),
    "pf_decoder_forward": (C.c_int, [_vp, _vp, _pi32, _vp, _pi32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "pf_decoder_create_contextual": (_vp, [C.POINTER(pf_decoder_config)]),
    "pf_decoder_forward_contextual": (C.c_int, [_vp, _vp, _pi32, _vp, _pi32, _vp, _i32, _f32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "pf_decoder_asf_scores": (C.c_int, [_vp, _vp, _pi32, _vp, _pi32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "pf_vad_create": (_vp, [C.POINTER(pf_vad_config)]),
    "pf_vad_destroy": (None, [_vp]),
    "pf_vad_set_tensor": (C.c_int, [_vp, C.c_char_p, _vp, _i64]),
    "pf_vad_missing": (C.c_int, [_vp]),
    "pf_vad_forward": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _pi32, _i32, _vp, _vp, _i32, _vp]),
    "pf_vad_frame_decibel": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    "pf_vad_decision_create": (_vp, [C.POINTER(pf_vad_options)]),
    "pf_vad_decision_destroy": (None, [_vp]),
    "pf_vad_decision_set_thresholds": (None, [_vp, C.c_double, C.c_double]),
    "pf_vad_decision_state": (C.c_int, [_vp]),
    "pf_vad_decision_push": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32]),
    "pf_ctc_create": (_vp, [_i32, _i32]),
    "pf_ctc_destroy": (None, [_vp]),
    "pf_ctc_set_tensor": (C.c_int, [_vp, C.c_char_p, _vp, _i64]),
    "pf_ctc_missing": (C.c_int, [_vp]),
    "pf_ctc_greedy": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "pf_ctc_set_precision": (C.c_int, [_vp, _i32]),
    "pf_stream_create": (_vp, [_vp, _vp, _vp, C.POINTER(pf_stream_config)]),
    "pf_stream_destroy": (None, [_vp]),
    "pf_stream_set_pe": (C.c_int, [_vp, _vp, _i32]),
    "pf_stream_reset": (C.c_int, [_vp, _vp]),
    "pf_stream_set_option": (C.c_int, [_vp, C.c_char_p, _i32]),
    "pf_stream_step": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _pi32, _pi32, _vp, _vp]),
    "pf_stream_step_begin": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "pf_stream_step_end": (C.c_int, [_vp, _pi32, _pi32]),
    "pf_stream_peek": (C.c_int, [_vp, _vp, _vp, _pi32]),
    "pf_frontend_fbank": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "pf_frontend_lfr_cmvn": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "pf_set_skinny_max_m": (C.c_int, [_i32]),
    "pf_measurement_build": (C.c_int, []),
    "pf_k_conv1d_gemm_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "pf_k_gemm_f32": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pf_k_gemm_bf16": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pf_k_gemm_bf16_time": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_float), _vp]),
    "pf_k_cast_bf16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "pf_k_split3": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _i32, _i32, _vp]),
    "pf_k_gemm_split3": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i64,
                                   _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_float), _vp]),
    "pf_k_split2": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _i32, _i32, _f32, _vp]),
    "pf_k_gemm_f16x2": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _f32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i64,
                                  _f32, _i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_float), _vp]),
    "pf_k_attention_f16x2": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _f32, _f32,
                                       _i32, _i32, C.POINTER(C.c_float), _vp]),
    "pf_k_gemm_f16x2_row": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _f32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _f32,
                                      _vp, _i64, _f32, _vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_float), _vp]),
    "pf_k_gemm_skinny_ln": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp]),
    "pf_k_ln_consts": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "pf_k_ffn_f16x2": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _f32, _vp, _f32, _vp, _i32, _i32, _i32,
                                 C.POINTER(C.c_float), _vp]),
    "pf_k_gemm_f16x2_row_fsmn": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _f32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _i32,
                                           _vp, _vp, _f32, _vp, _i64, _f32, _vp, _i32, _i32, _i32, _i32, C.POINTER(C.c_float), _vp]),
    "pf_k_layernorm_planes": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _i64, _f32, _i32, _i32, _f32, _i32, C.POINTER(C.c_float), _vp]),
    "pf_k_gemm_f16x2_qkv": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _f32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp,
                                      _i32, _i64, _f32, _f32, _f32, _i32, _i32, C.POINTER(C.c_float), _vp]),
    "pf_k_gemm_f16x2_argmax": (C.c_int, [_vp, _i32, _i64, _vp, _i32, _i64, _f32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "pf_k_gemm_argmax_f32": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "pf_k_log_softmax": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    "pf_k_layernorm": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "pf_k_fsmn": (C.c_int, [_vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pf_k_attention_f32": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "pf_k_attention_small": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "pf_k_gather_rows": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp]),
    "pf_k_attention_split3": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "pf_k_attention_bf16": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _f32, _vp]),
    "pf_k_lstm": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "pf_k_cif": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "pf_k_gemm_f32_time": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(C.c_float), _vp]),
    # utterance data parallelism at the C ABI (dp_rccl.hip)
    "pf_dp_unique_id": (C.c_int, [_vp, _i32]),
    "pf_dp_create": (_vp, [_vp, _i32, _i32, _i32]),
    "pf_dp_destroy": (C.c_int, [_vp]),
    "pf_dp_world": (C.c_int, [_vp]),
    "pf_dp_rank": (C.c_int, [_vp]),
    "pf_dp_broadcast_encoder": (C.c_int, [_vp, _vp, _i32, _vp]),
    "pf_dp_broadcast_predictor": (C.c_int, [_vp, _vp, _i32, _vp]),
    "pf_dp_broadcast_decoder": (C.c_int, [_vp, _vp, _i32, _vp]),
    "pf_dp_broadcast_ctc": (C.c_int, [_vp, _vp, _i32, _vp]),
    "pf_paraformer_create": (_vp, [_vp, _vp, _vp]),
    "pf_paraformer_destroy": (None, [_vp]),
    "pf_paraformer_forward": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "pf_paraformer_begin": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "pf_paraformer_finish": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "pf_paraformer_encoder_out": (_vp, [_vp]),
    "pf_paraformer_embeds": (_vp, [_vp]),
    "pf_dp_gather_ids": (C.c_int, [_vp, _vp, _i64, _vp, _i32, _vp]),
    "pf_dp_broadcast_raw": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    # profiling hooks used by bench.py (not part of the reference boundary)
    "pf_prof_enable": (C.c_int, [C.c_int]),
    "pf_prof_reset": (C.c_int, []),
    "pf_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pf_prof_read_tags": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_int64)]),
}


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libparaformer_hip.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j8"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libparaformer_hip.so failed:\n" + res.stdout)
    return LIB_PATH


def load():
    """Load the shared library (once) and attach the prototypes. Raises if it has not been built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is the only implementation of this package. "
                f"Build it with `make -C {CSRC_DIR}` (or `python -c 'import __graft_entry__ as g; g.build()'`)."
            )
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError = ABI drift between header and library
            fn.restype = res
            fn.argtypes = args
        if lib.pf_abi_version() != ABI_VERSION:
            raise ImportError(f"{LIB_PATH} reports ABI version {lib.pf_abi_version()}, this package binds version {ABI_VERSION}: "
                              f"rebuild it (`make -C {CSRC_DIR}`)")
        _lib = lib
        return lib


def last_error() -> str:
    msg = load().pf_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(status: int, what: str) -> None:
    if status != 0:
        raise HipRuntimeError(f"{what} failed with status {status}: {last_error()}")


def check_handle(handle, what: str):
    if not handle:
        raise HipRuntimeError(f"{what} failed: {last_error()}")
    return handle
