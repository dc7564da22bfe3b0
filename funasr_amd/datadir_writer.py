"""`output_dir` results in the reference's on-disk layout (funasr/utils/datadir_writer.py:7-105, used at
funasr/models/paraformer/model.py:571-575,688-692): `<output_dir>/1best_recog/{token,text}` with one `key value` line per
utterance -- what the AISHELL recipe's scoring step reads (examples/aishell/paraformer/run.sh:185-198)."""
from __future__ import annotations

import os
import warnings


class DatadirWriter:
    def __init__(self, p):
        self.path = str(p)
        self.children = {}
        self.fd = None
        self.keys = set()

    def __getitem__(self, key: str) -> "DatadirWriter":
        if self.fd is not None:
            raise RuntimeError("This writer points out a file")
        if key not in self.children:
            self.children[key] = DatadirWriter(os.path.join(self.path, key))
        return self.children[key]

    def __setitem__(self, key: str, value: str):
        if self.children:
            raise RuntimeError("This writer points out a directory")
        if key in self.keys:
            warnings.warn(f"Duplicated: {key}")
        if self.fd is None:
            os.makedirs(os.path.dirname(self.path) or ".", exist_ok=True)
            self.fd = open(self.path, "w", encoding="utf-8")
        self.keys.add(key)
        self.fd.write(f"{key} {value}\n")
        self.fd.flush()

    def close(self):
        for c in self.children.values():
            c.close()
        if self.fd is not None:
            self.fd.close()
            self.fd = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
