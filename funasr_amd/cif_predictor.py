"""CIF predictor on gfx950 (csrc/cif.hip + the f32 GEMM for the 3-tap conv).

Host-side mirror of `CifPredictorV2` (funasr/models/paraformer/cif_predictor.py:208-314,
`predictor_classes["CifPredictorV2"]`): same constructor keywords, state_dict keys (cif_conv1d.*, cif_output.*) and
`forward(hidden, target_label=None, mask [B,1,T]) -> (acoustic_embeds [B,N,D], token_num [B], alphas [B,T+1],
cif_peak [B,T+1])` for inference (no target labels). One divergence, documented in DESIGN.md: the reference raises
IndexError (cif_predictor.py:887) when the last utterance of a batch fires no token; here such an utterance simply
gets token_num 0 and zero rows.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .hip_module import Holder, HipModule, ParamHolder, host_i32, stream_ptr
from .register import tables


@tables.register("predictor_classes", "CifPredictorV2")
class CifPredictorV2(HipModule):
    _prefix = "pf_predictor"

    def __init__(self, idim, l_order, r_order, threshold=1.0, dropout=0.1, smooth_factor=1.0, noise_threshold=0,
                 tail_threshold=0.0, tf2torch_tensor_name_prefix_torch="predictor",
                 tf2torch_tensor_name_prefix_tf="seq2seq/cif", tail_mask=True, **kwargs):
        super().__init__()
        self.idim, self.l_order, self.r_order = idim, l_order, r_order
        self.threshold, self.smooth_factor, self.noise_threshold = threshold, smooth_factor, noise_threshold
        self.tail_threshold, self.tail_mask = tail_threshold, tail_mask
        self.cif_conv1d = ParamHolder((idim, idim, l_order + r_order + 1), (idim,))
        self.cif_output = ParamHolder((1, idim), (1,))

    def _make_config(self):
        return _lib.pf_predictor_config(self.idim, self.l_order, self.r_order, float(self.threshold),
                                        float(self.smooth_factor), float(self.noise_threshold),
                                        float(self.tail_threshold), int(bool(self.tail_mask)))

    def forward(self, hidden, target_label=None, mask=None, ignore_id=-1, mask_chunk_predictor=None,
                target_label_length=None, lengths=None):
        if target_label is not None or target_label_length is not None or mask_chunk_predictor is not None:
            raise NotImplementedError("CifPredictorV2(HIP) implements the inference path (no target labels)")
        lib, h = self._ensure_handle()
        dev = self._handle_device
        hid = hidden.to(device=dev, dtype=torch.float32).contiguous()
        B, T, D = hid.shape
        if lengths is None:
            if mask is None:
                lengths = [T] * B
            else:   # [B, 1, T] validity mask as built by Paraformer.calc_predictor (paraformer/model.py:323-325)
                lengths = mask.reshape(B, -1).to(torch.float32).sum(-1).round().to(torch.int32)
        lens_c, lens = host_i32(lengths, B)
        alphas = torch.empty(B, T + 1, device=dev, dtype=torch.float32)
        peaks = torch.empty(B, T + 1, device=dev, dtype=torch.float32)
        tok = (C.c_int32 * B)()
        with torch.cuda.device(dev):
            _lib.check(lib.pf_predictor_alphas(h, hid.data_ptr(), lens_c, B, T, alphas.data_ptr(), peaks.data_ptr(),
                                               tok, stream_ptr()), "pf_predictor_alphas")
            token_num = list(tok)
            N = max(token_num)
            embeds = torch.empty(B, N, D, device=dev, dtype=torch.float32)
            if N > 0:
                _lib.check(lib.pf_predictor_embeds(h, hid.data_ptr(), B, T, N, embeds.data_ptr(), stream_ptr()),
                           "pf_predictor_embeds")
        # the reference returns the floored float count (cif_predictor.py:443-446)
        token_num_t = torch.tensor(token_num, dtype=torch.float32, device=dev)
        if self.tail_threshold <= 0.0:
            alphas, peaks = alphas[:, :T], peaks[:, :T]
        return embeds, token_num_t, alphas, peaks

    # ---- `forward` in two halves around its host wait (include/paraformer_hip.h pf_predictor_alphas_begin / _embeds_slot): a loop
    #      over batches calls forward_begin(batch i + 1) before forward_finish(batch i); two batches may be in flight (slots 0 / 1)
    def forward_begin(self, hidden, lengths) -> dict:
        lib, h = self._ensure_handle()
        dev = self._handle_device
        hid = hidden.to(device=dev, dtype=torch.float32).contiguous()
        B, T, D = hid.shape
        lens_c, _ = host_i32(lengths, B)
        slot = self.__dict__["_slot"] = 1 - self.__dict__.get("_slot", 1)
        alphas = torch.empty(B, T + 1, device=dev, dtype=torch.float32)
        peaks = torch.empty(B, T + 1, device=dev, dtype=torch.float32)
        counts = torch.empty(B, dtype=torch.int32, pin_memory=True)
        with torch.cuda.device(dev):
            _lib.check(lib.pf_predictor_alphas_begin(h, slot, hid.data_ptr(), lens_c, B, T, alphas.data_ptr(), peaks.data_ptr(),
                                                     counts.data_ptr(), stream_ptr()), "pf_predictor_alphas_begin")
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
        return dict(hid=hid, lens_c=lens_c, slot=slot, alphas=alphas, peaks=peaks, counts=counts, ev=ev)

    def forward_finish(self, st: dict):
        """-> what `forward` returns (embeds, token_num -- a HOST tensor here --, alphas, peaks) for the batch of `forward_begin`"""
        lib, h = self._ensure_handle()
        hid, dev = st["hid"], self._handle_device
        B, T, D = hid.shape
        st["ev"].synchronize()
        token_num = st["counts"].tolist()
        N = max(token_num)
        with torch.cuda.device(dev):
            embeds = torch.empty(B, N, D, device=dev, dtype=torch.float32)
            if N > 0:
                _lib.check(lib.pf_predictor_embeds_slot(h, st["slot"], hid.data_ptr(), B, T, N, embeds.data_ptr(), stream_ptr()),
                           "pf_predictor_embeds_slot")
        # the counts stay on the HOST here (a device tensor would cost a pageable H2D copy ordered behind everything already queued
        # on the stream -- the next batch's encoder -- which is the wait these halves exist to avoid)
        token_num_t = torch.tensor(token_num, dtype=torch.float32)
        alphas, peaks = st["alphas"], st["peaks"]
        if self.tail_threshold <= 0.0:
            alphas, peaks = alphas[:, :T], peaks[:, :T]
        return embeds, token_num_t, alphas, peaks


@tables.register("predictor_classes", "CifPredictorV3")
class CifPredictorV3(CifPredictorV2):
    """Host-side mirror of `CifPredictorV3` (funasr/models/bicif_paraformer/cif_predictor.py:121-384), the predictor of
    BiCifParaformer / SeACo-Paraformer: same state_dict keys (cif_conv1d.*, cif_output.*, upsample_cnn.*, blstm.*,
    cif_output2.*). `forward` is V2's contract computed with V3's sequential fp32 integrate-and-fire (`cif`, :39-86; the
    fifth return value, the training-only token_num2, is None); `get_upsample_timestamp(hidden, mask, token_num)` returns
    (None, None, us_alphas, us_cif_peak): the upsampled pair of :301-352 (the re-downsampled pair is unused at inference).
    `cnn_attn` upsampling is not built."""
    _create_name = "pf_predictor_create_v3"

    def __init__(self, idim, l_order, r_order, threshold=1.0, dropout=0.1, smooth_factor=1.0, noise_threshold=0,
                 tail_threshold=0.0, tf2torch_tensor_name_prefix_torch="predictor",
                 tf2torch_tensor_name_prefix_tf="seq2seq/cif", smooth_factor2=1.0, noise_threshold2=0, upsample_times=5,
                 upsample_type="cnn", use_cif1_cnn=True, tail_mask=True, **kwargs):
        super().__init__(idim, l_order, r_order, threshold=threshold, dropout=dropout, smooth_factor=smooth_factor,
                         noise_threshold=noise_threshold, tail_threshold=tail_threshold, tail_mask=True)
        if upsample_type not in ("cnn", "cnn_blstm"):
            # `cnn_attn` cannot run in the reference either: it hands the T-frame mask to an attention over the upsampled frames
            # (bicif_paraformer/cif_predictor.py:251,327 -> size mismatch; tests/test_reference_unreachable_options.py)
            raise NotImplementedError(f"CifPredictorV3(HIP): upsample_type {upsample_type!r} is not built (cnn, cnn_blstm; the "
                                      "reference's cnn_attn fails on its own mask)")
        self.upsample_times, self.upsample_type, self.use_cif1_cnn = int(upsample_times), upsample_type, bool(use_cif1_cnn)
        self.smooth_factor2, self.noise_threshold2 = smooth_factor2, noise_threshold2
        self.upsample_cnn = ParamHolder((idim, idim, self.upsample_times), (idim,))
        if upsample_type == "cnn_blstm":
            self.blstm = Holder()
            for sfx in ("", "_reverse"):
                for name, shape in (("weight_ih_l0", (4 * idim, idim)), ("weight_hh_l0", (4 * idim, idim)),
                                    ("bias_ih_l0", (4 * idim,)), ("bias_hh_l0", (4 * idim,))):
                    self.blstm.register_parameter(name + sfx, torch.nn.Parameter(torch.zeros(*shape), requires_grad=False))
            self.cif_output2 = ParamHolder((1, 2 * idim), (1,))
        else:
            self.cif_output2 = ParamHolder((1, idim), (1,))

    def _make_config(self):
        return None

    def _create_args(self):
        self._cfg_pair = (super()._make_config(),
                          _lib.pf_predictor_v3_config(self.upsample_times, 1 if self.upsample_type == "cnn_blstm" else 0,
                                                      int(self.use_cif1_cnn), float(self.smooth_factor2),
                                                      float(self.noise_threshold2)))
        return C.byref(self._cfg_pair[0]), C.byref(self._cfg_pair[1])

    def forward(self, hidden, target_label=None, mask=None, ignore_id=-1, mask_chunk_predictor=None,
                target_label_length=None, lengths=None):
        embeds, token_num, alphas, peaks = super().forward(hidden, target_label, mask, ignore_id, mask_chunk_predictor,
                                                           target_label_length, lengths)
        return embeds, token_num, alphas, peaks, None

    def get_upsample_timestamp(self, hidden, mask=None, token_num=None, lengths=None):
        lib, h = self._ensure_handle()
        dev = self._handle_device
        hid = hidden.to(device=dev, dtype=torch.float32).contiguous()
        B, T, D = hid.shape
        if lengths is None:
            lengths = [T] * B if mask is None else mask.reshape(B, -1).to(torch.float32).sum(-1).round().to(torch.int32)
        lens_c, _ = host_i32(lengths, B)
        if token_num is None:
            raise NotImplementedError("get_upsample_timestamp(HIP) needs the token counts (the reference's inference call)")
        tok_c, _ = host_i32(token_num, B)
        U = self.upsample_times
        us_alphas = torch.empty(B, U * T, device=dev, dtype=torch.float32)
        us_peaks = torch.empty(B, U * T, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(lib.pf_predictor_timestamp(h, hid.data_ptr(), lens_c, tok_c, B, T, us_alphas.data_ptr(),
                                                  us_peaks.data_ptr(), stream_ptr()), "pf_predictor_timestamp")
        self._keep = (lens_c, tok_c)                      # read by an async copy: alive until the next call
        # the re-downsampled pair (ds_alphas, ds_cif_peak) of the reference is not used at inference
        return None, None, us_alphas, us_peaks
