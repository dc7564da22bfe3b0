"""Drop the HIP classes into the reference's own registry.

`funasr.register.tables.register(table, key)` overwrites an existing key (funasr/register.py:172-177); calling
`funasr_amd.install()` after `import funasr` therefore re-points "WavFrontend", "SANMEncoder", "CifPredictorV2",
"ParaformerSANMDecoder", "Paraformer", "SenseVoiceEncoderSmall" and "SenseVoiceSmall" at the gfx950 implementations,
and `funasr.AutoModel(model=<dir>, device="cuda")` builds them by name (funasr/auto/auto_model.py:591-646) with no
other change. Without the `funasr` package pass any object with a compatible `register(table, key)` method.
"""
from __future__ import annotations


def hip_classes():
    """(table, key, class) triples of everything this package provides."""
    from . import cif_predictor, paraformer, paraformer_decoder, sanm_encoder, sense_voice, tokenizer, wav_frontend

    out = [
        ("frontend_classes", "WavFrontend", wav_frontend.WavFrontend),
        ("frontend_classes", "wav_frontend", wav_frontend.WavFrontend),
        ("encoder_classes", "SANMEncoder", sanm_encoder.SANMEncoder),
        ("encoder_classes", "SenseVoiceEncoderSmall", sanm_encoder.SenseVoiceEncoderSmall),
        ("predictor_classes", "CifPredictorV2", cif_predictor.CifPredictorV2),
        ("decoder_classes", "ParaformerSANMDecoder", paraformer_decoder.ParaformerSANMDecoder),
        ("model_classes", "Paraformer", paraformer.Paraformer),
        ("model_classes", "SenseVoiceSmall", sense_voice.SenseVoiceSmall),
    ]
    if hasattr(tokenizer, "CharTokenizer"):
        out.append(("tokenizer_classes", "CharTokenizer", tokenizer.CharTokenizer))
    return out


def install(tables=None, include_tokenizer: bool = False):
    """Register the HIP classes into `tables` (default: funasr.register.tables). Returns the list of (table, key)."""
    if tables is None:
        from funasr.register import tables as _tables      # the reference package

        tables = _tables
    done = []
    for table, key, cls in hip_classes():
        if table == "tokenizer_classes" and not include_tokenizer:
            continue                                        # the reference tokenizers are host-only and reused as-is
        tables.register(table, key)(cls)
        done.append((table, key))
    return done
