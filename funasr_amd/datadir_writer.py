"""`output_dir` results in the reference's on-disk layout: `<output_dir>/1best_recog/{token,text}` with one `key value` line per
utterance -- what the AISHELL recipe's scoring step reads (examples/aishell/paraformer/run.sh:185-198). Same behaviour as the
reference's writer (funasr/utils/datadir_writer.py:7-105, used at funasr/models/paraformer/model.py:571-575,688-692): indexing
descends into a sub-directory, assignment appends a line to the file the node stands for, a node is one or the other for life.
"""
from __future__ import annotations

import os
import warnings
from typing import Dict, Optional, TextIO

_DIR, _FILE = "directory", "file"


class DatadirWriter:
    """A node of the Kaldi-style data directory: `writer["1best_recog"]["text"]["utt1"] = "..."`.
    The first use decides what the node is -- `node[name]` makes it a directory, `node[key] = value` a file opened lazily."""

    def __init__(self, p):
        self.path = os.fspath(p)
        self._kind: Optional[str] = None
        self._entries: Dict[str, "DatadirWriter"] = {}
        self._out: Optional[TextIO] = None
        self.keys = set()

    def _become(self, kind: str) -> None:
        if self._kind is None:
            self._kind = kind
        elif self._kind != kind:
            raise RuntimeError(f"This writer points out a {self._kind}")       # the reference's message for either misuse

    def __getitem__(self, name: str) -> "DatadirWriter":
        self._become(_DIR)
        node = self._entries.get(name)
        if node is None:
            node = self._entries[name] = DatadirWriter(os.path.join(self.path, name))
        return node

    def __setitem__(self, key: str, value: str) -> None:
        self._become(_FILE)
        if key in self.keys:
            warnings.warn(f"Duplicated: {key}")
        if self._out is None:
            parent = os.path.dirname(self.path)
            if parent:
                os.makedirs(parent, exist_ok=True)
            self._out = open(self.path, "w", encoding="utf-8")
        self.keys.add(key)
        print(key, value, file=self._out, flush=True)                          # "<key> <value>\n", visible to a reader at once

    def close(self) -> None:
        if self._kind == _DIR:
            nodes = list(self._entries.values())
            for node in nodes:
                node.close()
            # sibling files of one directory (text / token / score ...) are expected to list the same utterances
            for a, b in zip(nodes, nodes[1:]):
                if a._kind == _FILE and b._kind == _FILE and a.keys != b.keys:
                    warnings.warn(f"Ids are mismatching between {a.path} and {b.path}")
        elif self._out is not None:
            self._out.close()
            self._out = None

    def __enter__(self) -> "DatadirWriter":
        return self

    def __exit__(self, *exc) -> None:
        self.close()
