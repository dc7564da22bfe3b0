"""Cases of tests/test_streaming_f16x2_gpu.py, each run in its OWN process (`python tests/_stream_f16x2_cases.py <case>`): a
GPU fault in a kernel path that has not yet seen hardware must not take the rest of the suite with it. Exit code 0 = passed;
an assertion prints its message and exits 1.

The streaming step in its f16x2 form (StreamBatch(precision="f16x2"), pf_stream_set_option("gemm_mode", 3)): every GEMM of
the step on the fp16 matrix cores with two-plane operands (fp32-class results), everything else as in the default step.
Bars are the default step's: token ids / counts / position counter equal to the REFERENCE's ParaformerStreaming sessions
(tests/golden/streaming.npz, streaming_geometries.npz), encoder window and carried CIF state within 1e-3."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from funasr_amd import synth  # noqa: E402


def load():
    g = np.load(os.path.join(GOLD, "streaming.npz"), allow_pickle=False)
    cfg = json.loads(bytes(g["config"]).decode())
    sd = synth.paraformer_state_dict(cfg, seed=int(g["seed"]), cif_bias=float(g["cif_bias"]))
    return g, cfg, sd


def build(cfg, sd, dev):
    from funasr_amd.paraformer_streaming import ParaformerStreaming
    model = ParaformerStreaming.from_config(cfg)
    model.load_state_dict(sd, strict=False)
    return model.to(dev)


def golden_session(dev, use_graph):
    from funasr_amd.paraformer_streaming import StreamBatch
    g, cfg, sd = load()
    model = build(cfg, sd, dev)
    sb = StreamBatch(model, 1, [0, 10, 5], 4, 1, use_graph=use_graph, precision="f16x2")
    for i in range(int(g["n_chunks"])):
        fin, tail, start_idx = (int(v) for v in g[f"flags_{i}"])
        feats = None if tail else torch.from_numpy(g[f"feats_{i}"]).to(dev)
        ids, enc = sb.step(feats, is_final=bool(fin), tail_chunk=bool(tail), return_enc=True)
        ref = torch.from_numpy(g[f"enc_{i}"])
        assert tuple(enc.shape) == tuple(ref.shape), i
        err = (enc.cpu() - ref).abs().max().item()
        assert err < 1e-3, (i, err)
        got = [t for t in ids[0] if t not in (0, 1, 2)]
        assert got == g[f"tokens_{i}"].tolist(), (i, got, g[f"tokens_{i}"].tolist())
        st = sb.peek()
        assert st["start_idx"] == start_idx
        assert abs(st["cif_alphas"][0] - float(g[f"cif_alphas_{i}"][0])) < 1e-4, i
        assert (st["cif_hidden"][0] - torch.from_numpy(g[f"cif_hidden_{i}"])).abs().max().item() < 1e-3, i
    sb.close()


def case_golden_eager(dev):
    golden_session(dev, False)


def case_golden_graph(dev):
    golden_session(dev, True)


def case_geometries(dev):
    """the reference's own sessions at other chunk sizes / look-back settings (tests/golden/streaming_geometries.npz)"""
    from funasr_amd.paraformer_streaming import StreamBatch
    g, cfg, sd = load()
    model = build(cfg, sd, dev)
    gg = np.load(os.path.join(GOLD, "streaming_geometries.npz"), allow_pickle=False)
    for si, s in enumerate(json.loads(str(gg["sessions"]))):
        sb = StreamBatch(model, 1, s["chunk"], s["enc_lb"], s["dec_lb"], use_graph=True, precision="f16x2")
        for i in range(s["n_chunks"]):
            fin, tail, start_idx = (int(v) for v in gg[f"s{si}_flags_{i}"])
            feats = None if tail else torch.from_numpy(gg[f"s{si}_feats_{i}"]).to(dev)
            ids = sb.step(feats, is_final=bool(fin), tail_chunk=bool(tail))
            got = [t for t in ids[0] if t not in (0, 1, 2)]
            assert got == gg[f"s{si}_tokens_{i}"].tolist(), (s, i, got)
            assert sb.peek()["start_idx"] == start_idx
        sb.close()


def _session(model, dev, S, precision, graph, g, seed=5, reps=1):
    from funasr_amd.paraformer_streaming import StreamBatch
    gen = torch.Generator()
    sb = StreamBatch(model, S, [0, 10, 5], 4, 1, use_graph=graph, precision=precision)
    out = []
    for _ in range(reps):
        gen.manual_seed(seed)
        for i in range(int(g["n_chunks"])):
            fin, tail, _ = (int(v) for v in g[f"flags_{i}"])
            feats = None
            if not tail:
                f0 = torch.from_numpy(g[f"feats_{i}"])
                others = [f0 * 0.5 + 0.1 * torch.randn(f0.shape, generator=gen) for _ in range(S - 1)]
                feats = torch.cat([f0] + others, 0).to(dev)
            ids, enc = sb.step(feats, is_final=bool(fin), tail_chunk=bool(tail), return_enc=True)
            out.append((ids, enc.cpu()))
        sb.reset()
    sb.close()
    return out


def case_batch_independence_and_graph(dev):
    """f16x2 step: graph replay == eager bitwise; stream 0 of a 9-stream batch == the 1-stream session bitwise (every GEMM
    block shape gives the same bits, all other ops are per row / per stream); reset() reproduces the session"""
    g, cfg, sd = load()
    model = build(cfg, sd, dev)
    n = int(g["n_chunks"])
    e1 = _session(model, dev, 1, "f16x2", False, g, reps=2)
    g1 = _session(model, dev, 1, "f16x2", True, g)
    g9 = _session(model, dev, 9, "f16x2", True, g)
    for a, b in zip(e1[:n], g1):
        assert a[0] == b[0] and torch.equal(a[1], b[1]), "graph replay differs from eager"
    for a, b in zip(e1[:n], g9):
        assert a[0][0] == b[0][0] and torch.equal(a[1][0], b[1][0]), "stream 0 depends on its batch neighbours"
    for a, b in zip(e1[:n], e1[n:]):
        assert a[0] == b[0] and torch.equal(a[1], b[1]), "reset() does not reproduce the session"


def case_many_streams_vs_fp32_step(dev):
    """9 lock-step streams: the f16x2 step against the default fp32 step on the same features -- token ids and counts equal on
    every stream and chunk, encoder windows within 2e-4 (both are fp32-class; the golden bar vs the reference is 1e-3)"""
    g, cfg, sd = load()
    model = build(cfg, sd, dev)
    a = _session(model, dev, 9, "fp32", True, g)
    b = _session(model, dev, 9, "f16x2", True, g)
    worst = 0.0
    for i, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0], (i, x[0], y[0])
        worst = max(worst, (x[1] - y[1]).abs().max().item())
    assert worst < 2e-4, worst
    print(f"max |enc f16x2 - enc fp32| = {worst:.3e}")


def case_weight_reload(dev, precision="f16x2"):
    """the prepared planes / exponents (f16x2 step) and the LayerNorm -> GEMM constants (fp32 step, Stream.ln_consts) follow a weight
    reload on the live handles (TensorTable version): a StreamBatch created before load_state_dict gives the results of a model built
    from the new weights"""
    from funasr_amd.paraformer_streaming import StreamBatch
    g, cfg, sd = load()
    sd2 = synth.paraformer_state_dict(cfg, seed=int(g["seed"]) + 1, cif_bias=float(g["cif_bias"]))
    model = build(cfg, sd, dev)
    sb = StreamBatch(model, 2, [0, 10, 5], 4, 1, use_graph=True, precision=precision)
    feats = [torch.from_numpy(g[f"feats_{i}"]).repeat(2, 1, 1).to(dev) for i in range(4)]
    for f in feats:
        sb.step(f)
    model.load_state_dict(sd2, strict=False)
    sb.reset()
    got = [sb.step(f, return_enc=True) for f in feats]
    sb.close()
    fresh = build(cfg, sd2, dev)
    sr = StreamBatch(fresh, 2, [0, 10, 5], 4, 1, use_graph=False, precision=precision)
    ref = [sr.step(f, return_enc=True) for f in feats]
    sr.close()
    for i, (a, b) in enumerate(zip(got, ref)):
        assert a[0] == b[0] and torch.equal(a[1], b[1]), i


def case_weight_reload_fp32_step(dev):
    case_weight_reload(dev, "fp32")


def case_oracle_geometry_many_tokens(dev):
    """chunk_size [0, 20, 10] (up to 42 fires per step) against the reference-pinned streaming oracle"""
    from funasr_amd.paraformer_streaming import StreamBatch
    from oracle import streaming_oracle as S
    g, cfg, sd = load()
    model = build(cfg, sd, dev)
    chunk, enc_lb, dec_lb = [0, 20, 10], 2, 1
    sb = StreamBatch(model, 1, chunk, enc_lb, dec_lb, precision="f16x2")
    st = S.model_init(cfg, tuple(chunk), enc_lb, dec_lb)
    gen = torch.Generator().manual_seed(chunk[1] * 10 + enc_lb)
    for i in range(6):
        fin = i == 5
        n = chunk[1] if not fin else chunk[1] + 2
        feats = torch.randn(1, n, 560, generator=gen) * 0.7
        trace = []
        with torch.no_grad():
            oids = S.generate_chunk(feats.clone(), st, sd, cfg, fin, trace)
        ids, enc = sb.step(feats.to(dev), is_final=fin, return_enc=True)
        assert (enc.cpu() - trace[0]["enc"]).abs().max().item() < 1e-3, i
        assert [t for t in ids[0] if t not in (0, 1, 2)] == oids, (i, ids[0], oids)
        assert len(ids[0]) == trace[0]["n"], i
    sb.close()


CASES = {k[5:]: v for k, v in globals().items() if k.startswith("case_")}

if __name__ == "__main__":
    name = sys.argv[1]
    assert torch.cuda.is_available(), "needs the MI355X"
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    CASES[name](torch.device("cuda:0"))
    print(f"case {name}: ok")
