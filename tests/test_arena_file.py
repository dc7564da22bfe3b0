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
    save_arena(ref, os.path.join(d, "model.arena"))
    assert load_model_dir(d)["init_param"].endswith("model.arena")
    os.remove(os.path.join(d, "model.pt"))                       # the arena alone is enough
    got, _ = AutoModel.build_model(model=d, device="cpu")
    for (k, a), (_, b) in zip(ref.state_dict().items(), got.state_dict().items()):
        assert torch.equal(a, b), k
