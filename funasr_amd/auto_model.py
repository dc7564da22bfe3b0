"""`AutoModel` for environments where the `funasr` package itself cannot be imported (it needs omegaconf, hydra,
torchaudio, kaldiio, ... -- none of them present in the build image).

Mirrors the dispatch half of funasr/auto/auto_model.py that sits on the hot path:
  * `build_model` (:522-675): local model directory -> `config.yaml` (merged with the constructor kwargs, kwargs win)
    -> tokenizer / frontend / model looked up BY NAME in the registry (:591-646) -> `model.pt` loaded with the
    reference's state-dict conventions (funasr/train_utils/load_pretrained_model.py:39-104: `state_dict` /
    `model_state_dict` / `model` wrappers stripped, `module.` prefixes dropped, strict load) -> `.to(device)`, `.eval()`;
  * `generate` / `inference` (:689-850): `prepare_data_iterator` (:347-415; wav path, wav.scp / jsonl file lists,
    lists of paths / arrays / tensors, raw PCM bytes), the batch loop calling
    `model.inference(data_in=..., key=..., tokenizer=..., frontend=..., **kwargs)` under `torch.no_grad()` (:812-814),
    RTF bookkeeping from `meta_data["batch_data_time"]` (:823-833).
Model directory format (funasr/download/download_model_from_hub.py:80-97): `config.yaml`, `model.pt`, `tokens.json`,
`am.mvn`. No hub download (no network): `model` must be a local directory.
`inference_with_vad` (:852-1254), ASR half: a VAD model -- any object that follows the FunASR model contract and returns
`[{"key", "value": [[beg_ms, end_ms], ...]}]`, e.g. the reference's own FsmnVADStreaming -- cuts each recording into
segments; the segments are sorted by length, packed into batches under the reference's `batch_size_s` /
`batch_size_threshold_s` policy, decoded by the HIP path, put back in order and merged (texts joined, per-token
timestamps shifted by the segment start); with a `punc_model` (CT-Transformer directory or object) the joined text is
punctuated once per recording (results that carry `words` keep their spelling, funasr_amd/punc_align.py) and
`sentence_timestamp` cuts it into sentence records. The speaker branch raises. `vad_model` may also be a local FSMN-VAD
model directory: the network runs on the GPU (funasr_amd/fsmn_vad.py), its decision logic on the host
(funasr_amd/vad_decision.py). `generate` hands a requested text-level hotword correction to the reference's own module (`_text_hotword_step`).

When the real package is importable, use `funasr.AutoModel` itself after `funasr_amd.install()`.
"""
from __future__ import annotations

import collections
import copy
import json
import logging
import os
import random
import string
import time
from typing import Any, Dict, List, Tuple

import torch

from . import bicif_paraformer as _bicif_paraformer  # noqa: F401  (registers BiCifParaformer / CifPredictorV3)
from . import contextual_paraformer as _contextual_paraformer  # noqa: F401  (registers ContextualParaformer + its decoder)
from . import ct_transformer as _ct_transformer  # noqa: F401  (registers CTTransformer)
from . import fsmn_vad as _fsmn_vad  # noqa: F401  (registers FSMN / FsmnVADStreaming)
from . import paraformer as _paraformer  # noqa: F401  (registers the model classes)
from . import paraformer_streaming as _paraformer_streaming  # noqa: F401  (WavFrontendOnline)
from . import seaco_paraformer as _seaco_paraformer  # noqa: F401  (registers SeacoParaformer)
from . import sense_voice as _sense_voice  # noqa: F401
from .register import tables


def deep_update(dst: dict, src: dict) -> dict:
    """funasr/utils/misc.py deep_update: recursive dict merge, `src` wins."""
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            deep_update(dst[k], v)
        else:
            dst[k] = v
    return dst


def _rand_key() -> str:
    chars = string.ascii_letters + string.digits
    return "rand_key_" + "".join(random.choice(chars) for _ in range(13))


def prepare_data_iterator(data_in, input_len=None, data_type=None, key=None) -> Tuple[List[str], List[Any]]:
    """funasr/auto/auto_model.py:347-415 for sound inputs."""
    filelist = (".scp", ".txt", ".json", ".jsonl", ".text")
    key_list: List[str] = []
    data_list: List[Any] = []
    if isinstance(data_in, str) and (data_in.startswith("http://") or data_in.startswith("https://")):
        raise NotImplementedError("URL inputs need a network; pass a local path")
    if isinstance(data_in, str) and os.path.exists(data_in):
        ext = os.path.splitext(data_in)[1].lower()
        if ext in filelist:
            with open(data_in, encoding="utf-8") as fin:
                for line in fin:
                    if not line.strip():
                        continue
                    k = _rand_key()
                    if data_in.endswith(".jsonl"):
                        obj = json.loads(line.strip())
                        data, k = obj["source"], obj.get("key", k)
                    else:
                        parts = line.strip().split(maxsplit=1)
                        data = parts[1] if len(parts) > 1 else parts[0]
                        k = parts[0] if len(parts) > 1 else k
                    data_list.append(data)
                    key_list.append(k)
        else:
            k = key if key is not None else os.path.splitext(os.path.basename(data_in))[0]
            data_list, key_list = [data_in], [k]
    elif isinstance(data_in, (list, tuple)):
        # the reference's loop (:397-405) keeps ONE `key` variable: a given key (whatever object it is) labels every item, a file
        # path sets it to its name, and the first random key is reused for the items after it -- reproduced as it is, the records'
        # `key` fields are part of the drop-in surface (tests/test_reference_vad_pipeline_differential.py compares them)
        data_list = list(data_in)
        for d in data_in:
            if isinstance(d, str) and os.path.exists(d):
                key = os.path.splitext(os.path.basename(d))[0]
            elif key is None:
                key = _rand_key()
            key_list.append(key)
    else:       # raw text (punctuation models), samples, features; a missing wav path is reported by audio.load_audio
        data_list = [data_in]
        key_list = [key if key is not None else _rand_key()]
    return key_list, data_list


def _load_yaml(path: str) -> dict:
    import yaml

    with open(path, "r", encoding="utf-8") as f:
        return yaml.safe_load(f) or {}


def load_model_dir(model_dir: str) -> dict:
    """config.yaml + the file conventions of download_model_from_hub.py:80-97."""
    cfg_path = os.path.join(model_dir, "config.yaml")
    if not os.path.exists(cfg_path):
        raise FileNotFoundError(f"{model_dir}: no config.yaml (expected a FunASR model directory)")
    kwargs = _load_yaml(cfg_path)
    conf_json = os.path.join(model_dir, "configuration.json")
    metas = {}
    if os.path.exists(conf_json):
        with open(conf_json, "r", encoding="utf-8") as f:
            metas = (json.load(f) or {}).get("file_path_metas", {}) or {}

    def resolve(meta_key: str, default_name: str):
        name = metas.get(meta_key, default_name)
        if isinstance(name, dict):
            return None
        p = os.path.join(model_dir, name)
        return p if os.path.exists(p) else None

    # a `model.arena` beside (or instead of) model.pt is preferred: one mapped blob instead of ~950 pickled tensors
    init_param = resolve("init_param", "model.arena") if "init_param" not in metas else None
    pt = resolve("init_param", "model.pt")
    if init_param and pt:
        # the arena must be the twin of the checkpoint beside it: stale (model.pt replaced since) or unstamped -> ignore it
        from .arena_file import read_arena_header, source_stamp
        src = read_arena_header(init_param).get("source")
        if not src or src.get("stamp") != source_stamp(pt):
            logging.warning("%s does not match %s (size / mtime): loading the checkpoint instead", init_param, pt)
            init_param = None
    init_param = init_param or pt
    if init_param:
        kwargs["init_param"] = init_param
    tokens = resolve("tokenizer_conf.token_list", "tokens.json") if "tokenizer_conf" not in metas else None
    tokens = tokens or (os.path.join(model_dir, "tokens.json") if os.path.exists(os.path.join(model_dir, "tokens.json")) else None)
    if tokens:
        kwargs.setdefault("tokenizer_conf", {})
        kwargs["tokenizer_conf"] = dict(kwargs["tokenizer_conf"] or {}, token_list=tokens)
    mvn = os.path.join(model_dir, "am.mvn")
    if os.path.exists(mvn):
        kwargs.setdefault("frontend_conf", {})
        kwargs["frontend_conf"] = dict(kwargs["frontend_conf"] or {}, cmvn_file=mvn)
    bpe = os.path.join(model_dir, "chn_jpn_yue_eng_ko_spectok.bpe.model")
    if os.path.exists(bpe):
        kwargs.setdefault("tokenizer_conf", {})
        kwargs["tokenizer_conf"] = dict(kwargs["tokenizer_conf"] or {}, bpemodel=bpe)
    kwargs["model_path"] = model_dir
    return kwargs


def load_pretrained_model(path: str, model: torch.nn.Module, ignore_init_mismatch: bool = True, **kwargs) -> None:
    """funasr/train_utils/load_pretrained_model.py:14-114 without scope maps: wrapper keys stripped (:45-47), `module.`
    prefix dropped, shape-mismatched tensors skipped with a log line (:94-97), then a strict load (:104). A path ending in
    `.arena` is a one-file weight arena (funasr_amd/arena_file.py): one mapped blob, one copy, strict names and shapes."""
    if str(path).endswith(".arena"):
        from .arena_file import load_arena
        load_arena(model, path, strict=True)
        return
    # tensors only, like the reference's default torch.load on torch >= 2.6 (a model.pt may come from a third party); the
    # full unpickler is an explicit opt-in for legacy checkpoints that wrap non-tensor objects
    unsafe = bool(kwargs.get("trust_pickle", False)) or os.environ.get("FUNASR_AMD_TRUST_PICKLE") == "1"
    src = torch.load(path, map_location="cpu", weights_only=not unsafe)
    for k in ("state_dict", "model_state_dict", "model"):
        if isinstance(src, dict) and k in src and isinstance(src[k], dict):
            src = src[k]
    dst = model.state_dict()
    out = {}
    for k, v in dst.items():
        cand = k
        if cand not in src and ("module." + cand) in src:
            cand = "module." + cand
        if cand in src:
            if ignore_init_mismatch and tuple(src[cand].shape) != tuple(v.shape):
                logging.info("ignore_init_mismatch: %s %s vs %s", k, tuple(src[cand].shape), tuple(v.shape))
                out[k] = v
            else:
                out[k] = src[cand]
        else:
            logging.warning("Miss key in ckpt: model: %s", k)
            out[k] = v
    model.load_state_dict(out, strict=True)


_TEXT_HOTWORD_KEYS = ("postprocess_hotwords", "postprocess_hotword_file")


def _text_hotword_step(cfg):
    """The text-level hotword correction that ends the reference's `generate` (funasr/auto/auto_model.py:742-748) is host-side
    string replacement on the final text -- outside the hot path and not restated here. When a call asks for it,
    the reference's OWN module does it (funasr.utils.postprocess_hotwords, importable wherever this package is installed into
    a FunASR checkout via install()); without FunASR the request is refused instead of being ignored."""
    if not any(cfg.get(k) for k in _TEXT_HOTWORD_KEYS):
        return lambda results, cfg: results
    try:
        from funasr.utils.postprocess_hotwords import apply_postprocess_hotwords_to_results
    except ImportError as e:
        raise NotImplementedError("postprocess_hotwords= is FunASR's text-level post-processing (funasr/utils/postprocess_hotwords.py); "
                                  "it needs the funasr package on the path (use funasr.AutoModel with funasr_amd.install())") from e
    return apply_postprocess_hotwords_to_results


class AutoModel:
    def __init__(self, **kwargs):
        if kwargs.get("spk_model") is not None:
            raise NotImplementedError("spk_model: the speaker pipeline is outside the HIP hot path")
        punc_model = kwargs.pop("punc_model", None)
        self.punc_kwargs = dict(kwargs.pop("punc_kwargs", None) or {})
        if isinstance(punc_model, str):                                  # CT-Transformer model directory (:478-490)
            pk = dict(self.punc_kwargs, model=punc_model)
            pk.setdefault("device", kwargs.get("device", "cuda"))
            punc_model, self.punc_kwargs = self.build_model(**pk)
        self.punc_model = punc_model
        vad_model = kwargs.pop("vad_model", None)
        self.vad_kwargs = dict(kwargs.pop("vad_kwargs", None) or {})
        if isinstance(vad_model, str):
            # a local FSMN-VAD model directory (config.yaml: FsmnVADStreaming / FSMN / WavFrontendOnline, model.pt,
            # am.mvn) or a registered class name with its configuration in vad_kwargs (:463-476)
            vk = dict(self.vad_kwargs, model=vad_model)
            vk.setdefault("device", kwargs.get("device", "cuda"))
            vad_model, self.vad_kwargs = self.build_model(**vk)
        self.vad_model = vad_model
        log_level = getattr(logging, str(kwargs.get("log_level", "WARNING")).upper(), logging.WARNING)
        logging.getLogger().setLevel(log_level)
        model, kwargs = self.build_model(**kwargs)
        self.kwargs = kwargs
        self.model = model
        self.model_path = kwargs.get("model_path")
        self._base_kwargs = copy.deepcopy({k: v for k, v in kwargs.items() if k not in ("tokenizer", "frontend")})

    # --------------------------------------------------------------------------------------------- build_model
    @staticmethod
    def build_model(**kwargs) -> Tuple[torch.nn.Module, Dict[str, Any]]:
        model_arg = kwargs.get("model")
        if isinstance(model_arg, str) and os.path.isdir(model_arg):
            file_kwargs = load_model_dir(model_arg)
            user = {k: v for k, v in kwargs.items() if k != "model"}
            kwargs = deep_update(file_kwargs, user)                       # constructor kwargs win (:728,782)
        elif isinstance(model_arg, str) and model_arg not in tables.model_classes:
            raise FileNotFoundError(f"model={model_arg!r}: not a local model directory and not a registered model "
                                    f"class; hub download is not available (no network)")
        torch.manual_seed(kwargs.get("seed", 0))
        device = kwargs.get("device", "cuda")
        if str(device).startswith("cuda") and not torch.cuda.is_available():
            # the reference silently falls back to cpu + batch_size 1 (:551-561); there is no CPU implementation of
            # this path, so keep the plumbing working and let inference() raise a clear error instead
            logging.warning("no GPU visible: the model is built on cpu, inference needs an AMD GPU")
            device = "cpu"
            kwargs["batch_size"] = 1
        if kwargs.get("ngpu", 1) == 0:
            device = "cpu"
            kwargs["batch_size"] = 1
        kwargs["device"] = device
        try:                                                  # _resolve_ncpu + torch.set_num_threads (:46-53,563-566)
            ncpu = max(int(kwargs.get("ncpu", 4)), 1)
        except (TypeError, ValueError):
            ncpu = 4
        kwargs["ncpu"] = ncpu
        if torch.get_num_threads() != ncpu:
            torch.set_num_threads(ncpu)
        # tokenizer (:591-601)
        tokenizer = kwargs.get("tokenizer")
        vocab_size = -1
        if isinstance(tokenizer, str):
            tok_cls = tables.tokenizer_classes.get(tokenizer)
            if tok_cls is None:
                raise KeyError(f"tokenizer {tokenizer!r} is not registered: {sorted(tables.tokenizer_classes)}")
            tokenizer = tok_cls(**(kwargs.get("tokenizer_conf") or {}))
            kwargs["token_list"] = getattr(tokenizer, "token_list", None)
            vocab_size = len(kwargs["token_list"]) if kwargs["token_list"] is not None else -1
            if vocab_size == -1 and hasattr(tokenizer, "get_vocab_size"):
                vocab_size = tokenizer.get_vocab_size()
        kwargs["tokenizer"] = tokenizer
        kwargs["vocab_size"] = vocab_size                      # the models receive it with every inference call (:571,600)
        # frontend (:626-634)
        frontend = kwargs.get("frontend")
        kwargs["input_size"] = kwargs.get("input_size")
        if isinstance(frontend, str):
            fe_cls = tables.frontend_classes.get(frontend)
            if fe_cls is None:
                raise KeyError(f"frontend {frontend!r} is not registered: {sorted(tables.frontend_classes)}")
            fconf = dict(kwargs.get("frontend_conf") or {})
            # frontend_conf.dither is honoured (kaldi.fbank's per-sample Gaussian noise, drawn on the device from a seeded
            # counter-based generator: funasr_amd/wav_frontend.py); configs that do not set it get this class's default 0
            frontend = fe_cls(device=None if device == "cpu" else device, **fconf)
            kwargs["input_size"] = frontend.output_size()
        kwargs["frontend"] = frontend
        # model (:636-646)
        name = kwargs.get("model")
        model_class = tables.model_classes.get(name)
        if model_class is None:
            raise KeyError(f"model class {name!r} is not registered in model_classes: {sorted(tables.model_classes)}")
        model_conf = dict(kwargs.get("model_conf") or {})
        build_kwargs = {k: v for k, v in kwargs.items() if k not in ("model", "model_conf", "tokenizer", "frontend")}
        deep_update(build_kwargs, model_conf)
        if vocab_size > 0 or "vocab_size" not in build_kwargs:
            build_kwargs["vocab_size"] = vocab_size
        model = model_class(**build_kwargs)
        init_param = kwargs.get("init_param")
        if init_param is not None and os.path.exists(init_param):
            load_pretrained_model(init_param, model, ignore_init_mismatch=kwargs.get("ignore_init_mismatch", True))
        elif init_param is not None:
            logging.warning("init_param %s does not exist: model keeps its initial weights", init_param)
        # the reference casts the whole module (auto_model.py:664-668); here the kwargs choose the OPERAND mode of the GEMMs and
        # the attention -- parameters, residual stream, LayerNorm statistics, softmax and the CIF predictor stay fp32:
        # bf16=True -> bf16 operands, fp32 accumulate (bf16-class error, like the reference's cast);
        # fp16=True -> the fp16 matrix cores with two-plane operands ("f16x2": fp32-class results, also the default)
        if kwargs.get("bf16", False) or kwargs.get("fp16", False):
            mode = "bf16" if kwargs.get("bf16", False) else "f16x2"
            if hasattr(model, "set_precision"):
                model.set_precision(mode)
            else:
                logging.warning("%s has no arithmetic modes: fp16 / bf16 ignored", type(model).__name__)
        model.to(device)
        model.eval()
        return model, kwargs

    # ------------------------------------------------------------------------------------------------- generate
    def generate(self, input, input_len=None, progress_callback=None, **cfg):
        apply_postprocess_hotwords_to_results = _text_hotword_step(cfg)
        if getattr(self, "vad_model", None) is None:                        # :729-742
            results = self.inference(input, input_len=input_len, progress_callback=progress_callback, **cfg)
            punc_model = getattr(self, "punc_model", None)
            if punc_model is not None:                                       # no VAD: every result is punctuated on its own
                for result in results:
                    pk = dict(copy.deepcopy({k: v for k, v in self.punc_kwargs.items() if k not in ("tokenizer", "frontend")}),
                              **{k: self.punc_kwargs[k] for k in ("tokenizer", "frontend") if k in self.punc_kwargs})
                    pk.setdefault("device", self.kwargs.get("device", "cuda"))
                    punc_res = self.inference(result["text"], model=punc_model, kwargs=pk, **cfg)
                    if cfg.get("return_raw_text", self.kwargs.get("return_raw_text", False)):
                        result["raw_text"] = copy.copy(result["text"])
                    result["text"] = punc_res[0]["text"]
            return apply_postprocess_hotwords_to_results(results, cfg)      # text-level hotword correction (:742,:748)
        return apply_postprocess_hotwords_to_results(self.inference_with_vad(input, input_len=input_len, **cfg), cfg)

    def inference(self, input, input_len=None, model=None, kwargs=None, key=None, progress_callback=None, batch_bounds=None, **cfg):
        """auto_model.py:750-850. `batch_bounds` (this package's own): [begin, end) index ranges into the input list that replace the
        count-based batches of `batch_size` -- inference_with_vad hands a recording's planned segment batches over in ONE call, so
        that they overlap like any other list of batches."""
        if kwargs is None:                                                   # _reset_runtime_configs (:1318-1359)
            keep = {k: self.kwargs[k] for k in ("tokenizer", "frontend") if k in self.kwargs}
            self.kwargs = dict(copy.deepcopy(self._base_kwargs), **keep)
            kwargs = self.kwargs
        kwargs.pop("cache", None)
        deep_update(kwargs, cfg)
        model = self.model if model is None else model
        batch_size = int(kwargs.get("batch_size", 1))
        key_list, data_list = prepare_data_iterator(input, input_len=input_len, data_type=kwargs.get("data_type"), key=key)
        results_all: List[dict] = []
        speed_stats: Dict[str, Any] = {}
        time_speech_total, time_escape_total = 0.0, 0.0
        call_kwargs = {k: v for k, v in kwargs.items() if k not in ("model", "key", "data_in")}
        n = len(data_list)

        def account(res, dt, end):
            nonlocal time_speech_total, time_escape_total
            results, meta = (res[0] if len(res) > 0 else [{"text": ""}]), (res[1] if len(res) > 1 else {})
            results_all.extend(results)
            batch_data_time = meta.get("batch_data_time", -1)
            speed_stats.update(load_data=meta.get("load_data", 0.0), extract_feat=meta.get("extract_feat", 0.0),
                               forward=f"{dt:0.3f}", batch_size=f"{len(results)}",
                               rtf=f"{dt / batch_data_time:0.3f}" if batch_data_time else "nan")
            if progress_callback:
                try:
                    progress_callback(end, n)
                except Exception as e:  # noqa: BLE001 - same tolerance as the reference (:835-839)
                    logging.error("progress_callback error: %s", e)
            time_speech_total += batch_data_time
            time_escape_total += dt

        def make_batch(beg, end):
            batch = {"data_in": data_list[beg:end], "key": key_list[beg:end]}
            if end - beg == 1 and kwargs.get("data_type") == "fbank":
                batch["data_in"] = data_list[beg]
                batch["data_lengths"] = input_len
            return batch

        # MI355X-native batching of a plain list (this package's own option, like inference_with_vad's): `batch_size_rows` = a budget
        # of encoder rows per batch. The inputs are taken longest first (lengths from the WAV headers / array sizes, nothing is
        # decoded for the plan), cut into batches by funasr_amd.dp.plan_batches_by_rows, and the records returned in INPUT order.
        restore = None
        if (batch_bounds is None and kwargs.get("batch_size_rows") and n > 1 and model is getattr(self, "model", None)
                and hasattr(model, "inference_begin") and kwargs.get("data_type", "sound") != "fbank" and kwargs.get("frontend") is not None):
            from .audio import peek_num_samples
            fe = kwargs["frontend"]
            lens = [peek_num_samples(d, getattr(fe, "fs", 16000)) for d in data_list]
            if all(v is not None and v > 0 for v in lens) and hasattr(fe, "num_frames"):
                restore = sorted(range(n), key=lambda i: -lens[i])
                data_list, key_list = [data_list[i] for i in restore], [key_list[i] for i in restore]
                from . import dp
                batch_bounds = dp.plan_batches_by_rows([fe.num_frames(lens[i]) for i in restore], int(kwargs["batch_size_rows"]), extra_rows=1,
                                                       packed=getattr(getattr(model, "encoder", None), "_mode", lambda: "fp32")() == "f16x2")
        bounds = [(beg, min(n, beg + batch_size)) for beg in range(0, n, batch_size)] if batch_bounds is None else [(int(b), int(e)) for b, e in batch_bounds]
        # More than one batch and a model that offers its `inference` in three parts (paraformer.py inference_begin / _launch /
        # _end): the loop of auto_model.py:790-840 with the batches overlapped -- host work of batch i + 1 and of batch i - 1
        # beside the GPU work of batch i. Same records in the same order; `pipeline=False` keeps the plain loop.
        done = 0
        if len(bounds) > 1 and kwargs.get("pipeline", True) and hasattr(model, "inference_begin"):
            inflight = collections.deque()                                    # [pending, end, host seconds so far, launched]

            def launch(e):
                t1 = time.perf_counter()
                with torch.no_grad():
                    model.inference_launch(e[0])
                e[2] += time.perf_counter() - t1
                e[3] = True

            def finish(e):
                nonlocal done
                if not e[3]:
                    launch(e)
                t1 = time.perf_counter()
                with torch.no_grad():
                    res = model.inference_end(e[0])
                account(res, e[2] + time.perf_counter() - t1, e[1])
                done += 1

            try:
                for beg, end in bounds:
                    t1 = time.perf_counter()
                    with torch.no_grad():
                        pending = model.inference_begin(**make_batch(beg, end), **call_kwargs)
                    if pending is None:                                       # this configuration has no split form
                        break
                    if inflight and not inflight[-1][3]:
                        launch(inflight[-1])                                  # batch i's decoder, behind ITS encoder, beside batch i + 1's
                    inflight.append([pending, end, time.perf_counter() - t1, False])
                    while len(inflight) > 2:
                        finish(inflight.popleft())
                if inflight and not inflight[-1][3]:
                    launch(inflight[-1])                                      # the last batch's decoder under the text of the one before
                while inflight:
                    finish(inflight.popleft())
            except BaseException:
                for e in inflight:                                            # no ticket stays open in the library
                    try:
                        model.inference_end(e[0])
                    except Exception:  # noqa: BLE001
                        pass
                raise
        for beg, end in bounds[done:]:
            t1 = time.perf_counter()
            with torch.no_grad():
                res = model.inference(**make_batch(beg, end), **call_kwargs)
            account(res, time.perf_counter() - t1, end)
        self.speed_stats = dict(speed_stats, rtf_avg=(time_escape_total / time_speech_total) if time_speech_total else None)
        try:
            device = next(model.parameters()).device
            if device.type == "cuda":
                with torch.cuda.device(device):
                    torch.cuda.empty_cache()                                   # :846-849
        except StopIteration:
            pass
        if restore is not None and len(results_all) == n:          # (a batch that decoded nothing leaves the count off: as decoded)
            ordered: List[Any] = [None] * n
            for pos, i in enumerate(restore):
                ordered[i] = results_all[pos]
            results_all = ordered
        return results_all

    # --------------------------------------------------------------------------------------- inference_with_vad
    @staticmethod
    def plan_vad_batches(durations_ms: List[int], batch_size_ms: int, threshold_ms: int) -> List[Tuple[int, int]]:
        """The reference's packing of length-sorted VAD segments (:946-966) as [begin, end) index ranges: a segment joins
        the open batch while it is not the last one, is shorter than `threshold_ms` and (longest so far) x (count) stays
        under `batch_size_ms`; otherwise the batch -- INCLUDING this segment -- is decoded and a new one is opened."""
        plan, beg, end, longest = [], 0, 1, 0
        n = len(durations_ms)
        for j, d in enumerate(durations_ms):
            if j < n - 1 and d < threshold_ms and max(longest, d) * (j + 1 - beg) < batch_size_ms:
                longest = max(longest, d)
                end += 1
                continue
            plan.append((beg, end))
            beg, end, longest = end, end + 1, d
        return plan

    def _rows_plan(self, durations_ms, kwargs):
        """MI355X-native alternative to batch_size_s: a budget of encoder ROWS per batch (funasr_amd/dp.py; 32 768 = one round of
        GEMM blocks over the chip). frames = LFR frames of the segment, 60 ms each; the list is length-sorted."""
        from . import dp
        fe = kwargs.get("frontend")
        frames = [fe.num_frames(int(d * 16)) if hasattr(fe, "num_frames") else max(1, int(d) // 60) for d in durations_ms]
        return dp.plan_batches_by_rows(frames, int(kwargs["batch_size_rows"]), extra_rows=1,
                                       packed=getattr(getattr(self.model, "encoder", None), "_mode", lambda: "fp32")() == "f16x2")

    @staticmethod
    def _recording_on_device(speech, kwargs, fs):
        """one upload of the recording: the segment slices are device views, a batch is padded on the device (load_utils.py:413's
        pad_sequence, same samples) and nothing is padded or copied per batch on the host"""
        if (str(kwargs.get("device", "")).startswith("cuda") and speech.device.type == "cpu" and kwargs.get("fs", 16000) == fs
                and torch.cuda.is_available()):
            return speech.to(kwargs["device"])
        return speech

    def _decode_one_recording(self, speech, segments, kwargs, cfg, budget, threshold_ms) -> List[dict]:
        """auto_model.py:905-985: the recording's segments, shortest first, in dynamic batches -> their records in THAT order"""
        order = sorted(range(len(segments)), key=lambda j: segments[j][1] - segments[j][0])
        durs = [segments[j][1] - segments[j][0] for j in order]
        # the reference updates the budget IN PLACE (auto_model.py:924-928): once a recording's shortest segment exceeds
        # it, the raised budget -- or the 0 of device="cpu" -- also applies to the recordings after it in the same call
        # (found by tests/test_reference_vad_pipeline_differential.py against the reference's own loop)
        budget[0] = max(budget[0], durs[0])
        if kwargs["device"] == "cpu":
            budget[0] = 0
        plan = self.plan_vad_batches(durs, budget[0], threshold_ms)
        if kwargs.get("batch_size_rows") and kwargs["device"] != "cpu":
            plan = self._rows_plan(durs, kwargs)
        # slice_padding_audio_samples (funasr/utils/vad_utils.py:28-51): 16 samples per millisecond. The reference decodes the
        # planned batches with one self.inference call each (:967-985); where batches can overlap the whole plan goes over in one
        # call (a batch that decodes nothing still yields ONE record, i.e. the caller's count check fails exactly as there)
        clips = [speech[int(segments[j][0] * 16): min(int(segments[j][1] * 16), len(speech))] for j in order]
        if len(plan) > 1 and kwargs.get("pipeline", True) and hasattr(self.model, "inference_begin"):
            return self.inference(clips, input_len=None, model=self.model, kwargs=kwargs, batch_bounds=plan, **cfg)
        decoded: List[dict] = []
        for beg, end in plan:
            results = self.inference(clips[beg:end], input_len=None, model=self.model, kwargs=kwargs, **cfg)
            if len(results) < 1:
                continue
            decoded.extend(results)
        return decoded

    def _decode_across_recordings(self, res, data_list, kwargs, cfg, fs, group_samples: int = 1 << 28) -> Dict[int, List[dict]]:
        """-> {recording index: its segments' records in SEGMENT order} for the recordings decoded here; {} when the mode does not
        apply (no `batch_size_rows`, `batch_across_recordings=False`, one recording, cpu, a model without the split inference).
        Recordings are taken in groups of at most `group_samples` samples (4.7 h at 16 kHz: 1 GiB on the device); a group's segments,
        longest first, are cut into batches by the rows budget and decoded in ONE overlapped `inference` call. What a segment's
        record can owe to its batch is what it owes to it in the reference: the LONGEST clip of a batch has no frame behind its last
        one, every other clip has the encoder's row for a padding frame there, and CifPredictorV2's conv reads that row
        (cif_predictor.py:196-205,275-277) -- the last token of a batch's longest clip may differ from the one it gets as a shorter
        member of another batch, exactly as when `batch_size_s` is changed there (200 one-minute calls: 200 of 87 108 characters
        against per-recording batches, profiles/r06u_*); CifPredictorV3's timestamp head runs over the padded batch."""
        if not (kwargs.get("batch_size_rows") and kwargs.get("batch_across_recordings", True)
                and str(kwargs.get("device", "")).startswith("cuda") and torch.cuda.is_available()
                and hasattr(self.model, "inference_begin") and sum(1 for r in res if len(r["value"]) > 0) > 1):
            return {}
        from .audio import load_audio_list
        done: Dict[int, List[dict]] = {}
        group: List[tuple] = []

        def flush():
            flat = [(seg[1] - seg[0], i, j) for i, _, segs in group for j, seg in enumerate(segs)]
            flat.sort(key=lambda t: -t[0])                                            # stable: longest first
            speech_of = {i: sp for i, sp, _ in group}
            segs_of = {i: segs for i, _, segs in group}
            clips = [speech_of[i][int(segs_of[i][j][0] * 16): min(int(segs_of[i][j][1] * 16), len(speech_of[i]))] for _, i, j in flat]
            decoded = self.inference(clips, input_len=None, model=self.model, kwargs=kwargs,
                                     batch_bounds=self._rows_plan([d for d, _, _ in flat], kwargs), **cfg)
            if len(decoded) == len(flat):                                             # else: the per-recording path decides
                for (_, i, j), rec in zip(flat, decoded):
                    done.setdefault(i, [None] * len(segs_of[i]))[j] = rec
            group.clear()

        total = 0
        for i, vad_res in enumerate(res):
            if len(vad_res["value"]) == 0:
                continue
            speech = self._recording_on_device(load_audio_list([data_list[i]], fs=fs, audio_fs=kwargs.get("fs", 16000))[0], kwargs, fs)
            if total > 0 and total + speech.numel() > group_samples:
                flush()
                total = 0
            group.append((i, speech, vad_res["value"]))
            total += speech.numel()
        if group:
            flush()
        return done

    def inference_with_vad(self, input, input_len=None, **cfg):
        """VAD -> length-sorted dynamic batches -> ASR -> merge -> punctuation -> sentence records
        (funasr/auto/auto_model.py:852-1254 without the speaker branch). Returns one dict per recording: key, text,
        [timestamp], [raw_text], [sentence_info]."""
        if self.vad_model is None:
            raise RuntimeError("inference_with_vad needs AutoModel(vad_model=<VAD model object>)")
        from .audio import load_audio_list
        vad_kwargs = dict(copy.deepcopy({k: v for k, v in self.vad_kwargs.items() if k not in ("tokenizer", "frontend")}),
                          **{k: self.vad_kwargs[k] for k in ("tokenizer", "frontend") if k in self.vad_kwargs})
        vad_kwargs.setdefault("device", self.kwargs.get("device", "cuda"))
        res = self.inference(input, input_len=input_len, model=self.vad_model, kwargs=vad_kwargs, **cfg)      # step 1
        if cfg.get("merge_vad", False):
            from .vad_utils import merge_vad
            for r in res:
                r["value"] = merge_vad(r["value"], self.kwargs.get("merge_length_s", 15) * 1000)
        keep = {k: self.kwargs[k] for k in ("tokenizer", "frontend") if k in self.kwargs}                       # step 2
        kwargs = dict(copy.deepcopy(self._base_kwargs), **keep)
        deep_update(kwargs, cfg)
        batch_size = max(int(kwargs.get("batch_size_s", 300)) * 1000, 1)
        threshold_ms = int(kwargs.get("batch_size_threshold_s", 60)) * 1000
        kwargs["batch_size"] = batch_size
        key_list, data_list = prepare_data_iterator(input, input_len=input_len, data_type=kwargs.get("data_type"))
        fs = getattr(kwargs.get("frontend"), "fs", 16000)
        out: List[dict] = []
        budget = [batch_size]                                       # updated in place across recordings, like the reference's
        # MI355X-native: with a rows budget the segments of SEVERAL recordings share batches (the reference batches inside one
        # recording only, :905-985 -- a one-minute call is a sixth of a 300-s batch and a thirtieth of a round of GEMM blocks)
        across = self._decode_across_recordings(res, data_list, kwargs, cfg, fs)
        for i, vad_res in enumerate(res):
            key, segments = vad_res["key"], vad_res["value"]
            n = len(segments)
            if n == 0:
                out.append({"key": key, "text": "", "timestamp": []})
                continue
            if i in across:
                order, decoded = list(range(n)), across[i]
            else:
                speech = load_audio_list([data_list[i]], fs=fs, audio_fs=kwargs.get("fs", 16000))[0]
                decoded = self._decode_one_recording(self._recording_on_device(speech, kwargs, fs), segments, kwargs, cfg, budget, threshold_ms)
                order = sorted(range(n), key=lambda j: segments[j][1] - segments[j][0])   # stable, ascending duration
            if len(decoded) != n:
                out.append({"key": key, "text": "", "timestamp": []})
                continue
            restored = [None] * n
            for pos, j in enumerate(order):
                restored[j] = decoded[pos]
            merged: Dict[str, Any] = {}
            for j in range(n):                                                            # :1005-1038
                for k, v in restored[j].items():
                    if k.startswith("timestamp"):
                        merged.setdefault(k, [])
                        for t in v:
                            if isinstance(t, dict):                              # dict stamps in seconds (Fun-ASR-Nano style)
                                t["start_time"] = (float(t["start_time"]) * 1000 + int(segments[j][0])) / 1000
                                t["end_time"] = (float(t["end_time"]) * 1000 + int(segments[j][0])) / 1000
                            else:
                                t[0] = int(t[0]) + int(segments[j][0])
                                t[1] = int(t[1]) + int(segments[j][0])
                        merged[k].extend(v)
                    elif "text" in k:
                        merged[k] = v if k not in merged else merged[k] + " " + v
                    elif k != "key":
                        merged[k] = v if k not in merged else merged[k] + v
            if "timestamps" in merged and "timestamp" not in merged:                 # :1039-1044
                merged["timestamp"] = [[int(t["start_time"] * 1000), int(t["end_time"] * 1000)] for t in merged["timestamps"]]
            if not len(merged.get("text", "").strip()):
                continue
            return_raw_text = kwargs.get("return_raw_text", False)
            # ASR units with their own spelling (`words`, one per timestamp): punctuation must not re-spell them (:1048-1058)
            words, word_stamps = merged.get("words"), merged.get("timestamp")
            word_text = None
            if (isinstance(words, list) and words and all(isinstance(w, str) and w.strip() for w in words)
                    and isinstance(word_stamps, list) and len(words) == len(word_stamps)):
                word_text = " ".join(words)
            # punctuation (:1060-1086): the recording's text, joined without blanks between CJK chunks, goes through the
            # punc model once; its text (or the original spelling with the marks inserted) replaces the joined ASR text
            punc_res, punc_array, punc_text = None, None, None
            punc_model = getattr(self, "punc_model", None)
            if punc_model is not None and "timestamps" not in merged:
                from . import punc_align
                from .vad_utils import join_vad_texts
                pk = dict(copy.deepcopy({k: v for k, v in self.punc_kwargs.items() if k not in ("tokenizer", "frontend")}),
                          **{k: self.punc_kwargs[k] for k in ("tokenizer", "frontend") if k in self.punc_kwargs})
                pk.setdefault("device", self.kwargs.get("device", "cuda"))
                raw_text = copy.copy(merged["text"])
                punc_text = join_vad_texts(item.get("text", "") for item in restored)
                punc_res = self.inference(punc_text, model=punc_model, kwargs=pk, **cfg)
                if return_raw_text:
                    merged["raw_text"] = raw_text
                punc_array = punc_res[0].get("punc_array")
                surface = punc_align.punctuate_surface_text(punc_text, punc_array, punc_model) if word_text is not None else None
                merged["text"] = surface or punc_res[0]["text"]
            # which text / timestamps the sentence cutter sees (:1088-1121)
            stamp_text = punc_text
            stamps = merged.get("timestamp", [])
            misaligned = False
            surface_sentences = None
            if punc_res is not None:
                from . import punc_align
                try:
                    n_punc = len(punc_array)
                except TypeError:
                    n_punc, punc_array = -1, None
                if word_text is not None and n_punc == len(words):
                    stamp_text = word_text
                elif word_text is not None and n_punc > 0:
                    units = punc_align.merge_timestamp_units(punc_text, words, word_stamps, punc_array, punc_model)
                    if units is not None:
                        stamp_text, stamps = units
                if punc_array is not None and n_punc != len(stamps):
                    punc_array, misaligned = None, True
                if word_text is not None and punc_array is not None:
                    surface_sentences = punc_align.timestamp_sentences_from_surface(punc_text, stamps, punc_array, punc_model,
                                                                                    return_raw_text=return_raw_text)
            if kwargs.get("sentence_timestamp", False):                                   # :1198-1234
                from .timestamps import timestamp_sentence
                from .vad_utils import vad_segment_sentences
                if not len(merged["text"].strip()):
                    merged["sentence_info"] = []
                elif punc_model is None and punc_res is None and not stamps:
                    merged["sentence_info"] = vad_segment_sentences(restored, segments)
                elif punc_res is None:
                    logging.warning("punc_model is required for sentence_timestamp, skipping sentence segmentation.")
                    merged["sentence_info"] = []
                elif misaligned:
                    logging.warning("punctuation timestamps could not be aligned, falling back to VAD segments.")
                    merged["sentence_info"] = vad_segment_sentences(restored, segments)
                elif surface_sentences is not None:
                    merged["sentence_info"] = surface_sentences
                else:
                    merged["sentence_info"] = timestamp_sentence(punc_array, stamps, stamp_text, return_raw_text=return_raw_text,
                                                                 english=kwargs.get("en_post_proc", False))
            merged["key"] = key
            out.append(merged)
        return out
