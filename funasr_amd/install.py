"""Drop the HIP classes into the reference's own registry.

`funasr.register.tables.register(table, key)` overwrites an existing key (funasr/register.py:172-177); calling
`funasr_amd.install()` after `import funasr` therefore re-points "WavFrontend", "SANMEncoder", "CifPredictorV2" / "V3",
"ParaformerSANMDecoder", "Paraformer", "BiCifParaformer", "SeacoParaformer", "ContextualParaformer" (+ its decoder),
"ParaformerStreaming", "SenseVoiceSmall",
"FsmnVADStreaming", "CTTransformer", "CTTransformerStreaming" (and their encoders / frontends) at the gfx950 implementations,
and `funasr.AutoModel(model=<dir>, device="cuda")` builds them by name (funasr/auto/auto_model.py:591-646) with no
other change. Without the `funasr` package pass any object with a compatible `register(table, key)` method.
"""
from __future__ import annotations


def hip_classes():
    """(table, key, class) triples of everything this package provides: every class the package's own registry
    (funasr_amd.register.tables) holds after its modules are imported -- frontends, encoders, predictors (V2, V3), decoder,
    the model classes (Paraformer, BiCifParaformer, SeacoParaformer, ParaformerStreaming, SenseVoiceSmall,
    FsmnVADStreaming, CTTransformer, CTTransformerStreaming) and the tokenizers."""
    from . import (bicif_paraformer, cif_predictor, contextual_paraformer, ct_transformer, fsmn_vad, normalize, paraformer,  # noqa: F401
                   paraformer_decoder, paraformer_streaming, sanm_encoder, seaco_paraformer, sense_voice, tokenizer, wav_frontend)
    from .register import TABLE_NAMES, tables as own

    out = []
    for table in TABLE_NAMES:
        for key, cls in sorted(getattr(own, table, {}).items()):
            out.append((table, key, cls))
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
