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


def _install_omegaconf() -> None:
    """A small WORKING stand-in for the parts of omegaconf the reference's AutoModel construction path uses
    (funasr/download/download_model_from_hub.py:72-103, funasr/auto/auto_model.py:18,565-571): OmegaConf.load / merge /
    to_container / create and the DictConfig / ListConfig types, on plain dicts read by yaml.safe_load."""
    if "omegaconf" in sys.modules and getattr(sys.modules["omegaconf"], "_pf_functional", False):
        return
    try:
        import omegaconf  # noqa: F401  (the real package, if the image ever gains it)
        if not isinstance(sys.modules["omegaconf"], _Stub):
            return
    except Exception:
        pass
    import yaml

    class DictConfig(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    class ListConfig(list):
        pass

    def wrap(x):
        if isinstance(x, dict):
            return DictConfig({k: wrap(v) for k, v in x.items()})
        if isinstance(x, (list, tuple)) and not isinstance(x, str):
            return ListConfig([wrap(v) for v in x])
        return x

    def unwrap(x):
        if isinstance(x, dict):
            return {k: unwrap(v) for k, v in x.items()}
        if isinstance(x, list):
            return [unwrap(v) for v in x]
        return x

    def merge2(a, b):
        out = DictConfig(a)
        for k, v in b.items():
            out[k] = merge2(out[k], v) if isinstance(out.get(k), dict) and isinstance(v, dict) else wrap(v)
        return out

    class OmegaConf:
        @staticmethod
        def load(path):
            with open(path, "r", encoding="utf-8") as f:
                return wrap(yaml.safe_load(f) or {})

        @staticmethod
        def create(obj=None):
            return wrap(obj if obj is not None else {})

        @staticmethod
        def merge(*cfgs):
            out = DictConfig()
            for c in cfgs:
                out = merge2(out, wrap(c))
            return out

        @staticmethod
        def to_container(cfg, resolve=True):
            return unwrap(cfg)

    m = types.ModuleType("omegaconf")
    m.OmegaConf, m.DictConfig, m.ListConfig, m._pf_functional = OmegaConf, DictConfig, ListConfig, True
    sys.modules["omegaconf"] = m


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
    _install_omegaconf()
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


def reference_automodel():
    """The reference's own `funasr.auto.auto_model.AutoModel` class (funasr/auto/auto_model.py), imported with the stand-ins
    above, plus its registry and the host-side classes its construction path resolves by name (tokenizers). The model /
    frontend / encoder / predictor / decoder keys are left EMPTY here: `funasr_amd.install()` fills them."""
    install()
    import funasr.tokenizer.char_tokenizer  # noqa: F401  registers CharTokenizer
    from funasr.auto.auto_model import AutoModel
    from funasr.register import tables
    return AutoModel, tables
