"""The C-ABI shared library loads without a GPU and exports every symbol include/paraformer_hip.h declares; the
ctypes prototypes cover the header declaration by declaration; nothing computes here."""
import ctypes
import os
import re

import pytest

from funasr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "paraformer_hip.h"), encoding="utf-8").read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = header_symbols()
    for must in ("pf_frontend_forward", "pf_encoder_forward", "pf_predictor_alphas", "pf_predictor_embeds",
                 "pf_decoder_forward", "pf_ctc_greedy", "pf_last_error", "pf_k_gemm_f32", "pf_k_attention_f32"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build the extension first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in header_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_ctypes_prototypes_cover_the_header():
    declared = set(header_symbols())
    bound = set(_lib.SIGNATURES)
    assert declared <= bound, f"no ctypes prototype for: {sorted(declared - bound)}"
    extra = bound - declared
    assert all(s.startswith("pf_prof_") for s in extra), f"bound but not declared: {sorted(extra)}"


def test_load_and_version_and_no_cpu_fallback():
    lib = _lib.load()
    assert lib.pf_abi_version() == 5
    if lib.pf_device_count() == 0:
        cfg = _lib.pf_encoder_config(560, 512, 4, 2048, 2, 0, 11, 0, 1e-12)
        h = lib.pf_encoder_create(ctypes.byref(cfg))
        assert not h, "create must fail without a GPU (no CPU fallback)"
        assert "no HIP device" in _lib.last_error()


def test_modules_refuse_cpu_tensors():
    import torch
    from funasr_amd import synth
    from funasr_amd.paraformer import Paraformer

    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=1, dec_blocks=1, vocab=31)
    model = Paraformer.from_config(cfg)                      # parameters on cpu
    with pytest.raises(RuntimeError, match="AMD GPU"):
        model.encoder(torch.zeros(1, 8, 560), [8])


def test_dp_entry_points_exist_and_fail_cleanly_without_a_gpu():
    """include/paraformer_hip.h pf_dp_*: the C-ABI form of the multi-GPU split (weights broadcast into the handles, hypotheses
    gathered). No GPU here: a communicator cannot be made, and the calls must say so instead of crashing."""
    import ctypes as C
    from funasr_amd import _lib
    lib = _lib.load()
    for name in ("pf_dp_unique_id", "pf_dp_create", "pf_dp_destroy", "pf_dp_world", "pf_dp_rank", "pf_dp_broadcast_encoder",
                 "pf_dp_broadcast_predictor", "pf_dp_broadcast_decoder", "pf_dp_broadcast_ctc", "pf_dp_gather_ids", "pf_dp_broadcast_raw"):
        assert hasattr(lib, name), name
    assert lib.pf_dp_world(None) == -1 and lib.pf_dp_rank(None) == -1 and lib.pf_dp_destroy(None) == 0
    assert lib.pf_dp_create(None, 128, 1, 0) is None                       # bad arguments -> NULL + message
    assert lib.pf_last_error()
    buf = (C.c_char * 128)()
    assert lib.pf_dp_unique_id(buf, 16) < 0                                # short buffer refused
    assert lib.pf_dp_gather_ids(None, None, 4, None, 0, None) != 0
    assert lib.pf_dp_broadcast_raw(None, None, 4, 0, None) != 0


def test_dp_reports_a_missing_rccl_instead_of_crashing():
    """with no loadable RCCL the first pf_dp_* call must return the 'RCCL not found' error (dlerror() was called twice,
    the second call returns NULL -> std::string + nullptr). PF_RCCL_LIB names the one library to try; a fresh process because the
    loader runs once."""
    import subprocess
    import sys
    code = ("import ctypes as C\n"
            "from funasr_amd import _lib\n"
            "lib = _lib.load()\n"
            "buf = (C.c_char * 128)()\n"
            "rc = lib.pf_dp_unique_id(buf, 128)\n"
            "assert rc < 0, rc\n"
            "msg = _lib.last_error()\n"
            "assert 'RCCL not found' in msg and 'no_such_rccl' in msg, msg\n"
            "assert lib.pf_dp_create(buf, 128, 1, 0) is None and 'RCCL not found' in _lib.last_error()\n"
            "print('ok')\n")
    env = dict(os.environ, PF_RCCL_LIB="/nonexistent/libno_such_rccl.so", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
