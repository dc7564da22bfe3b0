"""Utterance-level data parallelism (funasr_amd/dp.py) at world_size 2 on the gloo backend: sharding covers the corpus
exactly once, the packed-arena weight broadcast reproduces rank 0's parameters bit for bit, and the fixed-stride
hypothesis gather returns every clip's ids in corpus order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from funasr_amd import dp


def test_shard_indices_partition_and_balance():
    lengths = [480000, 16000, 32000, 8000, 90000, 90000, 1234, 77777, 5, 31000, 64000]
    for world in (1, 2, 3, 8):
        shards = [dp.shard_indices(lengths, world, r) for r in range(world)]
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(len(lengths)))
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
        # descending-length deal: the r-th longest clip goes to rank r
        order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
        for r in range(min(world, len(lengths))):
            assert shards[r][0] == order[r]


def test_pack_unpack_hypotheses_roundtrip():
    ids = [[5, 6, 7], [], [1] * 40, [8403]]
    t = dp.pack_hypotheses(ids, n_pad=32)
    assert t.shape == (4, 33) and t.dtype == torch.int32
    back = dp.unpack_hypotheses(t)
    assert back == [[5, 6, 7], [], [1] * 32, [8403]]          # truncated to n_pad like bench.py


def test_device_side_pack_equals_host_pack():
    """pack_hypotheses_device (tensor ops on the decoder's [B, N] arg-max tensor) == pack_hypotheses (per-clip host loop)"""
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 8404, (5, 40), generator=g, dtype=torch.int32)
    counts = [40, 0, 17, 33, 1]
    host = dp.pack_hypotheses([ids[b, :c].tolist() for b, c in enumerate(counts)], n_pad=32)
    assert torch.equal(dp.pack_hypotheses_device(ids, counts, n_pad=32), host)
    wide = dp.pack_hypotheses_device(ids, counts, n_pad=64)                 # n_pad beyond N: still -1 padded
    assert wide.shape == (5, 65) and dp.unpack_hypotheses(wide) == [ids[b, :c].tolist() for b, c in enumerate(counts)]


def _fake_decode_factory(lengths):
    def decode(indices):
        return [[(i * 7 + k) % 8404 for k in range(lengths[i] % 13)] for i in indices]
    return decode


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                          # different weights per rank before the broadcast
        model = torch.nn.Sequential(torch.nn.Linear(37, 19), torch.nn.LayerNorm(19), torch.nn.Linear(19, 5, bias=False))
        ref = None
        if rank == 0:
            ref = [p.detach().clone() for p in model.parameters()]
        nbytes = dp.broadcast_model(model, src=0)
        assert nbytes == 4 * sum(p.numel() for p in model.parameters())
        flat = dp.pack_arena(list(model.parameters()))
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert all(torch.equal(both[0], b) for b in both), "weights differ across ranks after the arena broadcast"
        if rank == 0:
            assert all(torch.equal(a, b) for a, b in zip(ref, model.parameters()))
        lengths = [480000, 16000, 32000, 8000, 90000, 90000, 1234, 77777, 401, 31000, 64000]
        hyps = dp.recognize_sharded(_fake_decode_factory(lengths), lengths, n_pad=16, dst=0)
        if rank == 0:
            expect = _fake_decode_factory(lengths)(list(range(len(lengths))))
            assert hyps == expect
        else:
            assert hyps is None
        dist.barrier()
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"fail: {e!r}"))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_world_size_2_gloo_broadcast_shard_gather():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
    res = sorted(q.get(timeout=5) for _ in range(2))
    assert res == [(0, "ok"), (1, "ok")], res
    assert all(p.exitcode == 0 for p in procs)


def test_row_budget_batch_planner():
    """dp.plan_batches_by_rows: consecutive batches over a length-sorted list, every batch within the row budget (or a single
    clip), packed slots (frames + extra in 16-row slots, capped by the batch's longest clip) or the padded product."""
    import random
    from funasr_amd import dp
    rng = random.Random(3)
    frames = sorted((rng.randint(20, 250) for _ in range(400)), reverse=True)
    for order in (frames, frames[::-1]):
        for packed in (True, False):
            for budget in (64, 1000, 32768):
                plan = dp.plan_batches_by_rows(order, budget, extra_rows=1, packed=packed)
                assert plan[0][0] == 0 and plan[-1][1] == len(order)
                assert all(a[1] == b[0] and a[0] < a[1] for a, b in zip(plan, plan[1:]))
                for b, e in plan:
                    longest = max(order[b:e])
                    rows = sum(dp.encoder_rows(f, longest, 1, packed) for f in order[b:e])
                    assert rows <= budget or e - b == 1
                    if e < len(order):             # greedy: the next clip would not have fitted
                        longest2 = max(longest, order[e])
                        rows2 = sum(dp.encoder_rows(f, longest2, 1, packed) for f in order[b:e + 1])
                        assert rows2 > budget
    assert dp.encoder_rows(83, 245, 1, True) == 96 and dp.encoder_rows(83, 245, 1, False) == 256
    assert dp.encoder_rows(245, 245, 1, True) == 256            # the longest clip has no padding row behind it
    assert dp.plan_batches_by_rows([], 100) == []


def test_ensure_ranks_relaunches_a_bare_command_and_refuses_mismatches(tmp_path):
    """`python script.py --gpus 2` without a launcher becomes a 2-rank torch.distributed.run job on 127.0.0.1 (what bench.py and
    tools/sweep.py do with --gpus N); a launcher that started another world size is an error, not a silent 1-rank run"""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "job.py"
    script.write_text(textwrap.dedent(f"""
        import os, sys
        sys.path.insert(0, {root!r})
        from funasr_amd.dp import ensure_ranks
        n = int(sys.argv[sys.argv.index("--gpus") + 1])
        world = ensure_ranks(n)
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
        assert dist.get_world_size() == world == n
        if dist.get_rank() == 0:
            print("WORLD", world, flush=True)
        dist.destroy_process_group()
    """))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(script), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert [l for l in r.stdout.splitlines() if l.startswith("WORLD")] == ["WORLD 2"]
    r = subprocess.run([sys.executable, str(script), "--gpus", "4"], env=dict(env, WORLD_SIZE="2", RANK="0"), capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_rank_thread_count_follows_the_cpu_quota(monkeypatch, tmp_path):
    """dp.pin_rank_to_cores: a rank takes its slice of the usable cores, and never more threads than its share of the container's CPU
    quota (a 2-rank dry run with 128 threads per rank on a 16-CPU quota took 7 minutes for 12 s of work)"""
    import builtins
    import os
    import torch
    from funasr_amd import dp
    if not hasattr(os, "sched_setaffinity"):
        pytest.skip("no sched_setaffinity on this platform")
    before_aff, before_thr, before_env = os.sched_getaffinity(0), torch.get_num_threads(), os.environ.get("OMP_NUM_THREADS")
    pinned = {}
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)))
    monkeypatch.setattr(os, "sched_setaffinity", lambda pid, cores: pinned.update(cores=list(cores)))
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            f = tmp_path / "cpu.max"
            f.write_text("1600000 100000\n")
            return real_open(f, *a, **k)
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fake_open)
    try:
        out2 = dp.pin_rank_to_cores(1, 2)
        assert pinned["cores"] == list(range(128, 256)) and out2 == {"cores": [128, 255], "threads": 8}
        out8 = dp.pin_rank_to_cores(3, 8)
        assert pinned["cores"] == list(range(96, 128)) and out8["threads"] == 2
        assert dp.pin_rank_to_cores(0, 64)["threads"] == 1
    finally:
        monkeypatch.undo()
        os.sched_setaffinity(0, before_aff)
        torch.set_num_threads(before_thr)
        if before_env is None:
            os.environ.pop("OMP_NUM_THREADS", None)
        else:
            os.environ["OMP_NUM_THREADS"] = before_env
