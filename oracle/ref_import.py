"""Import the reference's own modules from /root/reference without running funasr/__init__.py (which walks all
60 model families, ~60 s) and without its missing third-party imports.  TEST INFRASTRUCTURE ONLY, and only usable
in the build container (the GPU box has no /root/reference).

`funasr` is registered as a namespace-style stub whose __path__ points at the reference tree, so
`import funasr.models.sanm.encoder` executes the reference source unchanged. Absent third-party packages that
the hot path never calls at inference (torchaudio, librosa, kaldiio, rapidfuzz, omegaconf, soundfile, ...) are
replaced by attribute-tolerant stubs so that module-level `import x` / `from x import Y` statements succeed.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("FUNASR_REFERENCE", "/root/reference")
_STUBS = ["torchaudio", "torchaudio.compliance", "torchaudio.compliance.kaldi", "librosa", "kaldiio", "rapidfuzz",
          "rapidfuzz.distance", "omegaconf", "soundfile", "jieba", "editdistance", "hydra", "modelscope"]


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "funasr", "models", "sanm"))


def install() -> None:
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    sys.dont_write_bytecode = True          # never write __pycache__ into the read-only reference
    if "funasr" not in sys.modules:
        pkg = types.ModuleType("funasr")
        pkg.__path__ = [os.path.join(REF_ROOT, "funasr")]
        sys.modules["funasr"] = pkg
    for name in _STUBS:
        try:
            if name not in sys.modules:
                importlib.import_module(name)
        except Exception:
            m = _Stub(name)
            m.__path__ = []
            sys.modules[name] = m
            if "." in name:
                parent, child = name.rsplit(".", 1)
                setattr(sys.modules[parent], child, m)


def modules():
    """Return the reference classes/functions on the hot path."""
    install()
    from funasr.models.sanm.encoder import SANMEncoder
    from funasr.models.paraformer.cif_predictor import CifPredictorV2, cif_v1, cif_wo_hidden_v1
    from funasr.models.paraformer.decoder import ParaformerSANMDecoder
    from funasr.models.sense_voice.model import SenseVoiceEncoderSmall
    from funasr.models.ctc.ctc import CTC
    import funasr.frontends.wav_frontend as wav_frontend
    from funasr.models.transformer.embedding import SinusoidalPositionEncoder

    return dict(SANMEncoder=SANMEncoder, CifPredictorV2=CifPredictorV2, cif_v1=cif_v1,
                cif_wo_hidden_v1=cif_wo_hidden_v1, ParaformerSANMDecoder=ParaformerSANMDecoder,
                SenseVoiceEncoderSmall=SenseVoiceEncoderSmall, CTC=CTC, wav_frontend=wav_frontend,
                SinusoidalPositionEncoder=SinusoidalPositionEncoder)
