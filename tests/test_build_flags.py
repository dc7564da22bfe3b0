"""Build-level guard of DESIGN 4's fix: the kernels without matrix instructions of their own must contain no packed-fp32 VALU
instructions (fbank_kernel built WITH them returned wrong frames next to the 128 x 128 f16x2 GEMM on a shared CU). hipcc cross-compiles
gfx950 without a GPU, so this runs in the CPU suite."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "funasr_amd", "csrc")
NOPK_SOURCES = ["frontend", "rowwise", "cif", "stream", "vad", "vad_decision", "lstm", "gemm_skinny", "attention_f32", "attention_small", "gemm_f32"]


def _make_vars():
    text = open(os.path.join(CSRC, "Makefile")).read()
    nopk = re.search(r"^NOPK := (.*)$", text, re.M).group(1).strip()
    flagged = set(re.findall(r"^FLAGS_(\w+) := \$\(NOPK\)$", text, re.M))
    return text, nopk, flagged


def test_makefile_builds_the_non_matrix_kernels_without_packed_fp32():
    text, nopk, flagged = _make_vars()
    assert "packed-fp32-ops" in nopk and "$(FLAGS_$*)" in text
    assert flagged == set(NOPK_SOURCES), flagged ^ set(NOPK_SOURCES)


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_frontend_compiles_to_no_packed_fp32_instruction(tmp_path):
    _, nopk, _ = _make_vars()
    out = str(tmp_path / "frontend.s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", CSRC] + nopk.split() + \
          ["-o", out, os.path.join(CSRC, "frontend.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = open(out).read()
    assert "fbank_kernel" in asm
    assert len(re.findall(r"v_pk_(?:add|mul|fma)_f32", asm)) == 0
    # and the flag is what does it: the default code generation of the same file uses them by the hundred
    out2 = str(tmp_path / "frontend_default.s")
    r = subprocess.run([c for c in cmd if c not in nopk.split()][:-3] + ["-o", out2, os.path.join(CSRC, "frontend.hip")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(re.findall(r"v_pk_(?:add|mul|fma)_f32", open(out2).read())) > 100
