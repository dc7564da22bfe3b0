"""Build-level guard of DESIGN 4's fence: no kernel of the library may contain a packed-fp32 VALU instruction (v_pk_add / v_pk_mul /
v_pk_fma_f32 return wrong lanes next to f16 / bf16 MFMA waves with barriers on a shared CU: tools/micro/pk,
profiles/r05_pk_reproducer.txt). hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU suite."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "funasr_amd", "csrc")
PK = r"v_pk_(?:add|mul|fma)_f32"


def _nopk():
    text = open(os.path.join(CSRC, "Makefile")).read()
    return text, re.search(r"^NOPK := (.*)$", text, re.M).group(1).strip()


def test_makefile_builds_every_kernel_without_packed_fp32():
    text, nopk = _nopk()
    assert "packed-fp32-ops" in nopk
    assert re.search(r"^CXXFLAGS \+= \$\(NOPK\)$", text, re.M), "the flag must apply to every source"
    assert re.search(r"^\t\$\(HIPCC\) \$\(CXXFLAGS\) -c \$< -o \$@$", text, re.M), "one compile rule, with CXXFLAGS, for every source"


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_frontend_compiles_to_no_packed_fp32_instruction(tmp_path):
    _, nopk = _nopk()
    out = str(tmp_path / "frontend.s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", CSRC] + nopk.split() + \
          ["-o", out, os.path.join(CSRC, "frontend.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = open(out).read()
    assert "fbank_kernel" in asm
    assert len(re.findall(PK, asm)) == 0
    # and the flag is what does it: the default code generation of the same file uses them by the hundred
    out2 = str(tmp_path / "frontend_default.s")
    r = subprocess.run([c for c in cmd if c not in nopk.split()][:-3] + ["-o", out2, os.path.join(CSRC, "frontend.hip")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(re.findall(PK, open(out2).read())) > 100


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")
def test_built_objects_hold_no_packed_fp32_instruction(tmp_path):
    """Disassembles the gfx950 code object of every object file the Makefile built (build() ran before the suite)."""
    objs = sorted(glob.glob(os.path.join(CSRC, "build", "*.o")))
    if not objs:
        pytest.skip("library not built in this tree")
    llvm = "/opt/rocm/lib/llvm/bin"
    checked = 0
    for o in objs:
        fat, co = str(tmp_path / (os.path.basename(o) + ".fat")), str(tmp_path / (os.path.basename(o) + ".co"))
        r = subprocess.run([f"{llvm}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", o], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fat):
            continue                                   # host-only object (no device code)
        r = subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--output={co}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        d = subprocess.run([f"{llvm}/llvm-objdump", "-d", co], capture_output=True, text=True, timeout=600)
        assert d.returncode == 0, d.stderr[-500:]
        n = len(re.findall(PK, d.stdout))
        assert n == 0, f"{os.path.basename(o)}: {n} packed-fp32 instructions"
        checked += "v_mfma" in d.stdout or "s_endpgm" in d.stdout
    assert checked >= 15, checked
