"""funasr_amd.install(): the HIP classes land in a registry with the reference's `register(table, key)` protocol under
the reference's own key names -- checked against a stand-in object and, when /root/reference is present, against the
reference's real funasr/register.py (loaded as a single file: importing the whole package takes a minute)."""
import importlib.util
import os

import pytest

from funasr_amd import install as inst

REF_REGISTER = "/root/reference/funasr/register.py"


class FakeTables:
    def __init__(self):
        self.calls = {}

    def register(self, table, key=None):
        def deco(cls):
            self.calls[(table, key)] = cls
            return cls
        return deco


def test_install_into_protocol_object():
    t = FakeTables()
    done = inst.install(t)
    assert ("model_classes", "Paraformer") in done and ("encoder_classes", "SANMEncoder") in done
    assert t.calls[("predictor_classes", "CifPredictorV2")].__name__ == "CifPredictorV2"
    assert not any(tb == "tokenizer_classes" for tb, _ in done)          # reference tokenizers are reused as-is
    for pair in (("model_classes", "SeacoParaformer"), ("model_classes", "BiCifParaformer"), ("predictor_classes", "CifPredictorV3"),
                 ("model_classes", "ParaformerStreaming"), ("model_classes", "FsmnVADStreaming"), ("encoder_classes", "FSMN"),
                 ("model_classes", "CTTransformer"), ("frontend_classes", "WavFrontendOnline"), ("model_classes", "SenseVoiceSmall"),
                 ("model_classes", "CTTransformerStreaming"), ("encoder_classes", "SANMVadEncoder"), ("model_classes", "ContextualParaformer")):
        assert pair in done, pair


@pytest.mark.skipif(not os.path.exists(REF_REGISTER), reason="reference checkout not present (GPU box)")
def test_install_overrides_keys_in_the_reference_registry():
    spec = importlib.util.spec_from_file_location("_ref_register", REF_REGISTER)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tables = mod.tables

    @tables.register("encoder_classes", "SANMEncoder")
    class Placeholder:                                                   # what the reference would have registered
        pass

    inst.install(tables)
    from funasr_amd.sanm_encoder import SANMEncoder
    assert tables.encoder_classes.get("SANMEncoder") is SANMEncoder       # re-registration overrides (register.py:172-177)
    assert tables.model_classes.get("Paraformer").__name__ == "Paraformer"
    assert tables.frontend_classes.get("WavFrontend").__name__ == "WavFrontend"
