"""One-file weight arena: every parameter of a model in ONE contiguous blob plus a small JSON index.

`model.pt` (funasr/download/download_model_from_hub.py:80-97, loaded by load_pretrained_model.py:39-104) is a pickled dict
of ~950 tensors: reading it means un-pickling and 950 small host->device copies. The arena file holds the same
`state_dict` as `{magic, index: [{name, shape, offset}], dtype}` + raw little-endian data, so a process maps it, checks
names and shapes against its model (strict, like load_state_dict) and hands the library one large copy. It is the on-disk
twin of the in-memory arena `dp.broadcast_model` ships over RCCL. fp32 keeps the checkpoint exactly (what the fp32 /
bf16x3 modes need); bf16 halves the file for the bf16-operand mode (weights are then bf16-rounded, like `.to(bfloat16)`).

    save_arena(model, "model.arena")            # from a loaded model or a state_dict
    load_arena(model, "model.arena")            # strict: same names, same shapes
"""
from __future__ import annotations

import json
import os
import struct
import warnings
from typing import Dict, Union

import numpy as np
import torch

MAGIC = b"PFARENA1"


def source_stamp(path: str):
    """(size, mtime_ns) of the checkpoint an arena was made from: load_model_dir ignores an arena whose stamp no longer
    matches the model.pt beside it (a replaced / fine-tuned checkpoint must not be shadowed by a stale arena)."""
    st = os.stat(path)
    return [int(st.st_size), int(st.st_mtime_ns)]


def save_arena(model_or_state: Union[torch.nn.Module, Dict[str, torch.Tensor]], path: str, dtype: str = "float32",
               source: str = None) -> int:
    """-> bytes written. Layout: MAGIC | u64 header length | header JSON | padding to 64 | data. `source`: the checkpoint
    file this arena mirrors (its size / mtime go into the header, see source_stamp); default: a `model.pt` beside `path`."""
    if dtype not in ("float32", "bfloat16"):
        raise ValueError("arena dtype must be 'float32' or 'bfloat16'")
    sd = model_or_state.state_dict() if isinstance(model_or_state, torch.nn.Module) else model_or_state
    if source is None:
        # an arena written beside a checkpoint mirrors that checkpoint: stamp it, or load_model_dir would take it for a stale
        # leftover and fall back to the slow torch.load path
        beside = os.path.join(os.path.dirname(os.path.abspath(path)), "model.pt")
        if os.path.exists(beside):
            source = beside
    index, off = [], 0
    for name, t in sd.items():
        if not torch.is_floating_point(t):
            raise TypeError(f"{name}: the arena holds floating-point parameters only")
        index.append({"name": name, "shape": list(t.shape), "offset": off})
        off += t.numel()
    head = {"dtype": dtype, "numel": off, "index": index}
    if source is not None:
        head["source"] = {"name": os.path.basename(source), "stamp": source_stamp(source)}
    header = json.dumps(head).encode("utf-8")
    pad = (-(len(MAGIC) + 8 + len(header))) % 64
    tdt = torch.float32 if dtype == "float32" else torch.bfloat16
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", len(header)))
        f.write(header)
        f.write(b"\0" * pad)
        for _, t in sd.items():
            flat = t.detach().to("cpu", tdt).contiguous().reshape(-1)
            f.write(flat.view(torch.int16 if tdt == torch.bfloat16 else torch.float32).numpy().tobytes())
    os.replace(tmp, path)
    return os.path.getsize(path)


def read_arena_header(path: str) -> dict:
    with open(path, "rb") as f:
        if f.read(len(MAGIC)) != MAGIC:
            raise ValueError(f"{path}: not a weight arena file")
        (hlen,) = struct.unpack("<Q", f.read(8))
        if hlen > os.path.getsize(path):
            raise ValueError(f"{path}: arena header length beyond the file")
        return json.loads(f.read(hlen).decode("utf-8"))


def read_arena(path: str):
    """-> (header dict, flat torch tensor viewing the memory-mapped data; float32 or bfloat16)"""
    with open(path, "rb") as f:
        if f.read(len(MAGIC)) != MAGIC:
            raise ValueError(f"{path}: not a weight arena file")
        (hlen,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(hlen).decode("utf-8"))
    data_off = len(MAGIC) + 8 + hlen
    data_off += (-data_off) % 64
    # the header is untrusted input: every tensor must lie inside the declared element count, and that inside the file
    itemsize = 4 if header.get("dtype") == "float32" else 2
    numel = int(header["numel"])
    if header.get("dtype") not in ("float32", "bfloat16") or numel < 0 or data_off + numel * itemsize > os.path.getsize(path):
        raise ValueError(f"{path}: arena header declares {numel} elements of {header.get('dtype')}, beyond the file's length")
    for e in header["index"]:
        n = 1
        for d in e["shape"]:
            n *= int(d)
        if int(e["offset"]) < 0 or int(e["offset"]) + n > numel:
            raise ValueError(f"{path}: tensor {e['name']} ({n} elements at {e['offset']}) lies outside the arena")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", UserWarning)             # read-only mapping: the tensor is only ever read
        if header["dtype"] == "float32":
            arr = np.memmap(path, dtype=np.float32, mode="r", offset=data_off, shape=(header["numel"],))
            flat = torch.from_numpy(np.asarray(arr))
        else:
            arr = np.memmap(path, dtype=np.int16, mode="r", offset=data_off, shape=(header["numel"],))
            flat = torch.from_numpy(np.asarray(arr)).view(torch.bfloat16)
    return header, flat


def load_arena(model: torch.nn.Module, path: str, strict: bool = True) -> int:
    """Fill `model`'s parameters / buffers from the arena: ONE device copy of the whole blob, then views. -> elements."""
    header, flat = read_arena(path)
    own = model.state_dict()
    names = {e["name"] for e in header["index"]}
    if strict:
        missing, extra = [k for k in own if k not in names], [n for n in names if n not in own]
        if missing or extra:
            raise RuntimeError(f"arena / model mismatch: missing in arena {missing[:5]}, unexpected {extra[:5]}")
    dev = next((p.device for p in model.parameters()), torch.device("cpu"))
    blob = flat.to(dev)                                         # the single host -> device copy
    with torch.no_grad():
        for e in header["index"]:
            dst = own.get(e["name"])
            if dst is None:
                continue
            if list(dst.shape) != e["shape"]:
                raise RuntimeError(f"{e['name']}: arena shape {e['shape']} vs model {list(dst.shape)}")
            n = dst.numel()
            dst.copy_(blob[e["offset"]: e["offset"] + n].view(e["shape"]).to(dst.dtype))
    for m in model.modules():                                   # the HIP mirrors push their weights lazily
        if hasattr(m, "mark_dirty"):
            m.mark_dirty()
    return int(header["numel"])
