"""First contact of the N > 1 code paths on ONE GPU: two ranks (torch.distributed.run, gloo, both on cuda:0) through the very
commands the driver uses on an 8-GPU node -- `bench.py --gpus 2` and the corpus sweep -- so that RCCL day is boring:
rendezvous, the packed-arena weight broadcast, sharding, the device-packed hypothesis gather, max-over-ranks timing and the
rank-0 JSON line are all exercised; the sweep's hypotheses must come back in CORPUS order and equal the 1-rank result."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(cmd, timeout, **extra_env):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


def _torchrun(n, script_args):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(_port())] + script_args


@pytest.mark.timeout(900)
def test_sweep_two_ranks_equals_one_rank_in_corpus_order(cuda, tmp_path):
    """every clip decoded alone (--batch-seconds 1: a clip's tail token depends on what it is padded with, which is
    reference behaviour, so equality across shardings is defined per clip), 1 rank vs 2 ranks"""
    one, two = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    common = ["--clips", "48", "--batch-seconds", "1", "--no-overlap"]
    _run([sys.executable, "tools/sweep.py"] + common + ["--dump", one], 400)
    # --serialize-gpu: the two ranks of this dry run share ONE GPU; they take turns on it (see tools/sweep.py and DESIGN 6)
    out = _run(_torchrun(2, ["tools/sweep.py"] + common + ["--dist-backend", "gloo", "--serialize-gpu", "--dump", two]), 500)
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["clips"] == 48 and line["value"] > 0
    a, b = json.load(open(one)), json.load(open(two))
    assert len(a) == len(b) == 48 and all(len(x) > 0 for x in a)
    assert a == b


@pytest.mark.timeout(900)
def test_sweep_two_ranks_sharing_one_gpu_without_serialising(cuda, tmp_path):
    """the same comparison WITHOUT --serialize-gpu: both ranks interleave their kernels on the one GPU. Rounds 3 and 4 (until its
    last day) had this as an expected failure: about one frame in 10^4 came back wrong from fbank_kernel. Cause: its
    packed-fp32 VALU instructions next to waves of the 128 x 128 f16x2 GEMM on the same CU; the non-matrix kernels are now built
    without such instructions (csrc/Makefile). No cross-check, no serialisation: the dumps must be equal."""
    one, two = str(tmp_path / "one.json"), str(tmp_path / "two.json")
    common = ["--clips", "48", "--batch-seconds", "1", "--no-overlap"]
    _run([sys.executable, "tools/sweep.py"] + common + ["--dump", one], 400)
    _run(_torchrun(2, ["tools/sweep.py"] + common + ["--dist-backend", "gloo", "--dump", two]), 500, PF_FRONTEND_VERIFY="0")
    assert json.load(open(one)) == json.load(open(two))


@pytest.mark.timeout(900)
def test_bench_two_ranks_prints_one_whole_job_line(cuda):
    out = _run(_torchrun(2, ["bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo"]), 600)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]                       # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["rccl_ranks"] == 2 and d["config"]["weight_arena_bytes_broadcast"] > 8e8
    assert len(d["config"]["per_rank_ms_per_step"]) == 2 and max(d["config"]["per_rank_ms_per_step"]) == pytest.approx(d["ms_per_step"], rel=1e-2)
    assert len(d["config"]["weight_broadcast_seconds_per_rank"]) == 2 and all(t > 0 for t in d["config"]["weight_broadcast_seconds_per_rank"])
    assert d["config"]["hypothesis_gather_bytes_per_rank_per_step"] == 64 * 513 * 4
    # whole-job aggregate: both ranks' clips over the max-over-ranks time
    assert abs(d["value"] - 2 * 64 * 30.0 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 0.01
    assert "roofline" in d and d["roofline"]["frac"] > 0


@pytest.mark.timeout(1500)
def test_bench_eight_ranks_dry_run_on_one_gpu(cuda):
    """The driver's 8-GPU command with every rank on the one visible GPU (gloo): what an 8 x MI355X node will run first
    (BASELINE configs[3]; the reference's recipe is one process per GPU, examples/aishell/paraformer/run.sh:135-190). One
    whole-job line from rank 0, eight per-rank times, the 880-MB weight arena broadcast once, every rank pinned to its own
    host cores, no CPU-baseline leg at N > 1. Smaller batches than the headline (eight replicas share one GPU's memory and time)."""
    # (kept cheap: eight Python ranks import torch, build the 220 M-parameter mirror and take the arena through gloo on the box's
    # few host cores -- small batches, short clips, no calibration of the output layer)
    out = _run(_torchrun(8, ["bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "16", "--seconds", "10",
                             "--random-output-layer", "--dist-backend", "gloo"]), 1400)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 8 and c["rccl_ranks"] == 8 and c["parallelism"] == "utterance-dp8" and d["scaling"] == "weak"
    assert len(c["per_rank_ms_per_step"]) == 8 and all(t > 0 for t in c["per_rank_ms_per_step"])
    assert max(c["per_rank_ms_per_step"]) == pytest.approx(d["ms_per_step"], rel=1e-2)
    assert 8.0e8 < c["weight_arena_bytes_broadcast"] < 9.6e8 and c["weights_route"] == "arena"
    assert len(c["weight_broadcast_seconds_per_rank"]) == 8
    assert c["hypothesis_gather_bytes_per_rank_per_step"] == 16 * 513 * 4
    assert c["host_cores_rank0"] is not None and c["host_cores_rank0"]["threads"] >= 1
    assert abs(d["value"] - 8 * 16 * 10.0 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 0.01
    assert "cpu_baseline" not in d                      # N = 1 only (and only rank 0 would run it)


@pytest.mark.timeout(900)
def test_bench_bare_command_launches_its_own_ranks(cuda):
    """the driver's literal command, NO torchrun wrapper: `python bench.py --gpus 2 ...` must itself become a 2-rank job
    (funasr_amd.dp.ensure_ranks) and print a 2-rank line -- round 3 printed n_gpus 1 here"""
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo"],
                       cwd=ROOT, env=dict(env_clean, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["config"]["parallelism"] == "utterance-dp2"
    assert len(d["config"]["per_rank_ms_per_step"]) == 2


def test_bench_refuses_a_world_size_other_than_gpus(cuda):
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                       env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_c_abi_communicator_broadcasts_into_the_handles_and_gathers(cuda):
    """pf_dp_* (dp_rccl.hip) through RCCL itself, with the one rank a single-GPU box allows: unique id -> communicator ->
    grouped in-place broadcast over an encoder handle's tensors (the handle must come out unchanged and usable, with every
    derived plane re-made) -> gather of packed hypotheses. Multi-rank behaviour is RCCL's; what is ours is the handle plumbing."""
    import torch
    from funasr_amd import dp, synth
    from funasr_amd.sanm_encoder import SANMEncoder
    ec = dict(synth.tiny(synth.PARAFORMER_LARGE, enc_blocks=2)["encoder"])
    enc = SANMEncoder(**ec, input_layer="pe")
    enc.load_state_dict(synth.encoder_state_dict(ec, seed=5), strict=False)
    enc = enc.to(cuda).set_precision("f16x2")
    g = torch.Generator().manual_seed(3)
    feats = (torch.randn(2, 40, 560, generator=g) * 0.8).to(cuda)
    lens = torch.tensor([40, 23], dtype=torch.int32)
    before = enc(feats, lens)[0].clone()
    comm = dp.DeviceComm(1, 0, cuda, lambda ident: ident)
    assert comm.world == 1 and comm.rank == 0
    comm.broadcast_module(enc, src=0)
    assert torch.equal(enc(feats, lens)[0], before)
    packed = dp.pack_hypotheses([[5, 6, 7], [], [9]], 8, device=cuda).to(torch.int32).contiguous()
    out = comm.gather_ids(packed, dst=0)
    torch.cuda.synchronize()
    assert out.shape == (1, 3, 9) and torch.equal(out[0], packed)
    assert dp.unpack_hypotheses(out[0]) == [[5, 6, 7], [], [9]]
    comm.close()
