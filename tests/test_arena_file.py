"""One-file weight arena (funasr_amd/arena_file.py): round trip of a Paraformer state_dict, strictness, bf16 variant."""
import os

import pytest
import torch

from funasr_amd import synth
from funasr_amd.arena_file import load_arena, read_arena, save_arena
from funasr_amd.paraformer import Paraformer


def test_arena_round_trip_and_strictness(tmp_path):
    cfg = synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=2, dec_blocks=1, vocab=50)
    sd = synth.paraformer_state_dict(cfg, seed=3)
    src = Paraformer.from_config(cfg)
    src.load_state_dict(sd, strict=False)
    path = str(tmp_path / "model.arena")
    nbytes = save_arena(src, path)
    header, flat = read_arena(path)
    assert header["dtype"] == "float32" and flat.numel() == header["numel"] and nbytes >= 4 * flat.numel()
    dst = Paraformer.from_config(cfg)
    assert load_arena(dst, path) == header["numel"]
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k
    # bf16 file: half the bytes, weights rounded like .to(bfloat16)
    p16 = str(tmp_path / "model16.arena")
    assert save_arena(src, p16, dtype="bfloat16") < 0.55 * nbytes
    d16 = Paraformer.from_config(cfg)
    load_arena(d16, p16)
    k = "encoder.encoders0.0.self_attn.linear_q_k_v.weight"
    assert torch.equal(d16.state_dict()[k], src.state_dict()[k].to(torch.bfloat16).float())
    # strict: a model of another depth is refused, a corrupted file too
    other = Paraformer.from_config(synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=3, dec_blocks=1, vocab=50))
    with pytest.raises(RuntimeError, match="mismatch"):
        load_arena(other, path)
    bad = str(tmp_path / "bad.arena")
    with open(bad, "wb") as f:
        f.write(b"not an arena")
    with pytest.raises(ValueError):
        read_arena(bad)
    assert not os.path.exists(path + ".tmp")


def test_model_directory_prefers_the_arena_file(tmp_path):
    """AutoModel.build_model on a directory that holds model.arena next to model.pt loads the arena (same weights)."""
    from funasr_amd.auto_model import AutoModel, load_model_dir
    from tests._model_dir import make_model_dir
    d = str(tmp_path / "m")
    info = make_model_dir(d)
    ref, _ = AutoModel.build_model(model=d, device="cpu")
    pt = os.path.join(d, "model.pt")
    from funasr_amd.arena_file import read_arena_header
    save_arena(ref.state_dict(), str(tmp_path / "elsewhere.arena"))            # no checkpoint beside it: no stamp
    assert "source" not in read_arena_header(str(tmp_path / "elsewhere.arena"))
    save_arena(ref, os.path.join(d, "model.arena"))              # written beside model.pt the documented way: stamped as its twin
    assert load_model_dir(d)["init_param"].endswith("model.arena")
    save_arena(ref, os.path.join(d, "model.arena"), source=pt)   # explicitly stamped twin of model.pt: preferred
    assert load_model_dir(d)["init_param"].endswith("model.arena")
    sd = torch.load(pt, map_location="cpu", weights_only=True)   # the checkpoint is replaced (fine-tune / new download) ...
    torch.save({k: v + 1 for k, v in sd.items()} if all(torch.is_tensor(v) for v in sd.values()) else sd, pt)
    os.utime(pt, ns=(1, 1))
    assert load_model_dir(d)["init_param"].endswith("model.pt")  # ... and a stale arena no longer shadows it
    os.remove(pt)                                                # the arena alone is enough
    got, _ = AutoModel.build_model(model=d, device="cpu")
    for (k, a), (_, b) in zip(ref.state_dict().items(), got.state_dict().items()):
        assert torch.equal(a, b), k


def test_arena_header_is_bounds_checked(tmp_path):
    """offsets / element counts in the header are untrusted: a tensor outside the declared blob, or a blob beyond the file,
    is rejected before anything is mapped"""
    import json, struct
    from funasr_amd.arena_file import MAGIC
    sd = {"w": torch.arange(12, dtype=torch.float32).reshape(3, 4)}
    path = str(tmp_path / "a.arena")
    save_arena(sd, path)
    raw = open(path, "rb").read()
    (hlen,) = struct.unpack("<Q", raw[8:16])
    head = json.loads(raw[16:16 + hlen])

    def rewrite(h, name):
        hb = json.dumps(h).encode()
        pad = (-(len(MAGIC) + 8 + len(hb))) % 64
        p = str(tmp_path / name)
        with open(p, "wb") as f:
            f.write(MAGIC + struct.pack("<Q", len(hb)) + hb + b"\0" * pad + raw[-48:])
        return p

    bad = dict(head, index=[dict(head["index"][0], offset=5)])
    with pytest.raises(ValueError):
        read_arena(rewrite(bad, "off.arena"))
    with pytest.raises(ValueError):
        read_arena(rewrite(dict(head, numel=10 ** 9), "numel.arena"))
    read_arena(rewrite(head, "ok.arena"))
