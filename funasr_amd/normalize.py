"""`normalize_classes` on gfx950: UtteranceMVN and GlobalMVN (csrc/normalize.hip).

Host-side mirrors of funasr/models/normalize/utterance_mvn.py:9-49 and funasr/models/normalize/global_mvn.py:12-92, registered under
the reference's table / key names; same constructor keywords and `forward(x [B, T, D], ilens [B]) -> (x, ilens)`. The models apply
them between the frontend and the encoder (funasr/models/paraformer/model.py:305-306, sense_voice/model.py:836-837). Like the
reference at inference (`x.requires_grad` false) they work IN PLACE on the feature tensor. No CPU fallback.
"""
from __future__ import annotations

from pathlib import Path
from typing import Union

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .hip_module import device_lens, host_i32, stream_ptr
from .register import tables


def _device_features(x: torch.Tensor, what: str) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError(f"{what} runs only on an AMD GPU through libparaformer_hip.so (no CPU fallback); got a '{x.device}' tensor")
    if x.dtype != torch.float32 or not x.is_contiguous() or x.dim() != 3:
        raise ValueError(f"{what}: expected a contiguous float32 [B, T, D] tensor")
    return x


def _lens(ilens, x: torch.Tensor):
    if ilens is None:
        ilens = [x.size(1)] * x.size(0)                             # utterance_mvn.py:69-70
    _, host = host_i32(ilens, x.size(0))
    return device_lens(host, x.device)


@tables.register("normalize_classes", "UtteranceMVN")
class UtteranceMVN(nn.Module):
    def __init__(self, norm_means: bool = True, norm_vars: bool = False, eps: float = 1.0e-20):
        super().__init__()
        self.norm_means, self.norm_vars, self.eps = norm_means, norm_vars, eps

    def extra_repr(self):
        return f"norm_means={self.norm_means}, norm_vars={self.norm_vars}"

    def forward(self, x: torch.Tensor, ilens=None):
        x = _device_features(x, "UtteranceMVN")
        B, T, D = x.shape
        ld = _lens(ilens, x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().pf_utterance_mvn(x.data_ptr(), ld.data_ptr(), B, T, D, int(bool(self.norm_means)),
                                                    int(bool(self.norm_vars)), float(self.eps), stream_ptr()), "pf_utterance_mvn")
        x._pf_keep = ld                                             # the kernel reads the lengths asynchronously
        return x, ilens


@tables.register("normalize_classes", "GlobalMVN")
class GlobalMVN(nn.Module):
    def __init__(self, stats_file: Union[Path, str], norm_means: bool = True, norm_vars: bool = True, eps: float = 1.0e-20):
        super().__init__()
        self.norm_means, self.norm_vars, self.eps = norm_means, norm_vars, eps
        self.stats_file = Path(stats_file)
        stats = np.load(self.stats_file)                            # global_mvn.py:41-53: both layouts of the stats file
        if isinstance(stats, np.ndarray):
            count = stats[0].flatten()[-1]
            mean = stats[0, :-1] / count
            var = stats[1, :-1] / count - mean * mean
        else:
            count, sum_v, sum_square_v = stats["count"], stats["sum"], stats["sum_square"]
            mean = sum_v / count
            var = sum_square_v / count - mean * mean
        std = np.sqrt(np.maximum(var, eps))
        self.register_buffer("mean", torch.from_numpy(np.asarray(mean)))
        self.register_buffer("std", torch.from_numpy(np.asarray(std)))

    def extra_repr(self):
        return f"stats_file={self.stats_file}, norm_means={self.norm_means}, norm_vars={self.norm_vars}"

    def forward(self, x: torch.Tensor, ilens=None):
        x = _device_features(x, "GlobalMVN")
        B, T, D = x.shape
        self.mean = self.mean.to(x.device, x.dtype).contiguous()    # global_mvn.py:72-73
        self.std = self.std.to(x.device, x.dtype).contiguous()
        if self.mean.numel() != D:
            raise ValueError(f"GlobalMVN: the stats file holds {self.mean.numel()} dimensions, the features {D}")
        ld = _lens(ilens, x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().pf_global_mvn(x.data_ptr(), ld.data_ptr(), B, T, D, self.mean.data_ptr(), self.std.data_ptr(),
                                                 int(bool(self.norm_means)), int(bool(self.norm_vars)), stream_ptr()), "pf_global_mvn")
        x._pf_keep = ld
        return x, ilens
