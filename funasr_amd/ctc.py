"""CTC head on gfx950: vocabulary projection with the arg-max fused into the GEMM epilogue.

Host-side mirror of `CTC` (funasr/models/ctc/ctc.py:9-216) for inference: state_dict key ctc_lo.{weight,bias},
`log_softmax(hs_pad)`, `argmax(hs_pad)`.
"""
from __future__ import annotations

import torch

from . import _lib
from .hip_module import HipModule, linear, stream_ptr


class CTC(HipModule):
    _prefix = "pf_ctc"

    def __init__(self, odim: int, encoder_output_size: int, dropout_rate: float = 0.0, ctc_type: str = "builtin",
                 reduce: bool = True, ignore_nan_grad: bool = True, **kwargs):
        super().__init__()
        self.odim, self.eprojs = odim, encoder_output_size
        self.ctc_lo = linear(odim, encoder_output_size)

    def _mode(self) -> str:
        return getattr(self, "_precision", None) or ("f16x2" if self.eprojs % 32 == 0 else "fp32")

    def set_precision(self, mode=None):
        """"f16x2" (default): the arg-max route on the fp16 matrix cores from two-plane operands (fp32-class logits, three
        products; gemm_f16x2.hip). "fp32": the fp32 MFMA kernel; other mode names keep it too. None restores the default."""
        self._precision = mode
        return self

    def _make_config(self):
        return None

    def _create_args(self):
        return (self.eprojs, self.odim)

    def _run(self, hs_pad, want_logits):
        lib, h = self._ensure_handle()
        _lib.check(lib.pf_ctc_set_precision(h, 3 if self._mode() == "f16x2" else 0), "pf_ctc_set_precision")
        dev = self._handle_device
        x = hs_pad.to(device=dev, dtype=torch.float32).contiguous()
        lead = x.shape[:-1]
        M = x.numel() // x.shape[-1]
        ids = torch.empty(M, device=dev, dtype=torch.int32)
        logits = torch.empty(M, self.odim, device=dev, dtype=torch.float32) if want_logits else None
        with torch.cuda.device(dev):
            _lib.check(lib.pf_ctc_greedy(h, x.data_ptr(), M, ids.data_ptr(),
                                         logits.data_ptr() if want_logits else None, stream_ptr()), "pf_ctc_greedy")
        return ids.view(*lead), (logits.view(*lead, self.odim) if want_logits else None)

    def log_softmax(self, hs_pad):
        from . import ops
        _, logits = self._run(hs_pad, True)
        return ops.log_softmax(logits.contiguous(), inplace=True)          # row kernel, in place over the GEMM's output

    def argmax(self, hs_pad, ban_ids=None):
        """frame-wise arg-max over the vocabulary, fused into the projection GEMM. `ban_ids`: classes that must not win (the
        reference writes -inf into their log-probabilities, sense_voice/model.py:1004-1005) -- here a sibling head with the
        same weights and a -inf bias for them, so the fused route stays a single launch."""
        if ban_ids:
            return self._banned_head(tuple(sorted(int(i) for i in ban_ids)))._run(hs_pad, False)[0]
        ids, _ = self._run(hs_pad, False)
        return ids

    def _banned_head(self, ban: tuple) -> "CTC":
        cache = self.__dict__.setdefault("_ban_heads", {})
        key = (ban, self.ctc_lo.weight.data_ptr(), self.ctc_lo.weight._version, self.ctc_lo.bias._version)
        if cache.get("key") != key:                              # weights moved or changed: rebuild
            head = CTC(self.odim, self.eprojs)
            with torch.no_grad():
                bias = self.ctc_lo.bias.detach().clone()
                bias[list(ban)] = float("-inf")
            head.load_state_dict({"ctc_lo.weight": self.ctc_lo.weight.detach(), "ctc_lo.bias": bias})
            head = head.to(self.ctc_lo.weight.device)
            cache.clear()
            cache.update(key=key, head=head)
        cache["head"].set_precision(getattr(self, "_precision", None))
        return cache["head"]
