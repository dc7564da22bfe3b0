"""Utterance-level data parallelism over the GPUs of one node: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards over independent clips with NO data-path collective (the reference forks one process
per GPU over a split wav.scp and concatenates the outputs, examples/aishell/paraformer/run.sh:135-190). The only
exchanges are
  * start-up: ONE broadcast of a packed fp32 weight arena (880 MB for Paraformer-large) instead of ~950 small ones
    (xGMI is point-to-point: few large messages, not many small ones), and
  * per batch: a gather of fixed-stride int32 hypotheses [B, n_pad] (+ lengths) on rank 0 -- KBs, latency-bound.
"""
from __future__ import annotations

import os
import socket
import sys
from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def ensure_ranks(n_gpus: int, argv: Sequence[str] = None) -> int:
    """Make `python <script> --gpus N` by itself a job of N ranks (one process per GPU, the launch recipe the reference
    spells out per GPU in examples/aishell/paraformer/run.sh:135-190). With N > 1 and no WORLD_SIZE in the environment
    the calling script is re-executed under `torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1 (this
    call never returns in the parent: it exits with the job's status); under a launcher the world size must equal N --
    a mismatch is an error, never a silent 1-rank run. Returns the world size this process is part of."""
    world = os.environ.get("WORLD_SIZE")
    if world is None:
        if n_gpus <= 1:
            return 1
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        argv = list(sys.argv if argv is None else argv)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + argv
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this host driver
        import subprocess
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(world)
    if world != max(1, n_gpus):
        raise SystemExit(f"--gpus {n_gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a line for a job "
                         f"of another size (launch with --nproc-per-node {n_gpus}, or pass --gpus {world})")
    return world


def pin_rank_to_cores(local_rank: int, world: int) -> dict:
    """Give each of the node's ranks its own slice of the host cores (affinity + torch / OpenMP thread count), so that eight
    Python ranks feeding eight GPUs do not fight over the same cores (torch.distributed.run only sets OMP_NUM_THREADS=1 when it
    is unset; the reference's recipe, examples/aishell/paraformer/run.sh:135-190, leaves it to the OS). Best effort: returns
    what was applied, never raises."""
    out = {"cores": None, "threads": None}
    try:
        usable = sorted(os.sched_getaffinity(0))
        per = max(1, len(usable) // max(1, world))
        mine = usable[(local_rank % max(1, world)) * per:(local_rank % max(1, world)) * per + per] or usable
        os.sched_setaffinity(0, mine)
        # threads: never more than this rank's share of the container's CPU QUOTA -- an OpenMP pool larger than the quota runs ~100x
        # slower (a 2-rank run with 8 threads per rank on a 4-CPU quota took 7 minutes for what one thread does in seconds)
        quota = len(usable)
        try:
            with open("/sys/fs/cgroup/cpu.max") as f:
                q, period = f.read().split()[:2]
            if q != "max":
                quota = min(quota, max(1, int(float(q) / float(period))))
        except (OSError, ValueError):
            pass
        threads = max(1, min(len(mine), quota // max(1, world)))
        torch.set_num_threads(threads)
        os.environ["OMP_NUM_THREADS"] = str(threads)
        out = {"cores": [mine[0], mine[-1]], "threads": threads}
    except Exception:                                    # noqa: BLE001  (no sched_setaffinity on this platform, cgroup limits, ...)
        pass
    return out


def guard_shared_gpu(world: int, all_on_one: bool = False) -> bool:
    """Whether ranks share a GPU (more ranks than GPUs, or the dry run that puts every rank on GPU 0). Rounds 3 / 4 saw the frontend
    return a sporadically wrong 16-lane pass in that configuration; what round 4 found is a MITIGATION (no packed-fp32 VALU
    instructions in the non-matrix kernels, csrc/Makefile: 0 faults in 97 k calls), not a root cause -- so where the trigger can
    exist the fbank kernel's cross-check is switched on for the whole process (pf_set_concurrency_guard; 0.4 % of a step)."""
    shared = world > 1 and (all_on_one or world > max(1, torch.cuda.device_count()))
    if shared:
        from . import _lib
        _lib.load().pf_set_concurrency_guard(1)
    return shared


def shard_indices(lengths: Sequence[int], world: int, rank: int) -> List[int]:
    """Length-sorted round-robin deal (like the length sort of auto_model.py:917-918): rank r gets the clips at
    positions r, r+world, ... of the descending-length order, so every rank sees the same length mix."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return order[rank::world]


def encoder_rows(frames: int, longest: int, extra_rows: int = 1, packed: bool = True, align: int = 16) -> int:
    """Rows the f16x2 encoder computes for a clip of `frames` frames in a batch whose longest clip has `longest`: with row
    packing (`pf_encoder_set_row_packing`) its own frames plus `extra_rows` padding rows in an `align`-row slot, without it
    the longest clip's frames rounded up -- every clip of a padded batch costs the same."""
    up = lambda n: (n + align - 1) // align * align
    return up(min(frames + extra_rows, longest)) if packed else up(longest)


def plan_batches_by_rows(frames: Sequence[int], budget_rows: int, extra_rows: int = 1, packed: bool = True,
                         align: int = 16) -> List[tuple]:
    """Cut a LENGTH-SORTED list of clips (frame counts, ascending or descending) into consecutive batches [begin, end) whose
    encoder rows stay within `budget_rows` (a batch always takes at least one clip). The unit of work on this GPU is the row,
    not the padded second: 32 768 rows are one full round of 256 x 256 GEMM blocks over 256 CUs, and a batch that
    spills a few blocks into another round pays for the whole round."""
    plan, beg, rows, longest = [], 0, 0, 0
    for i, f in enumerate(frames):
        f = int(f)
        new_longest = max(longest, f)
        if packed:
            new_rows = rows + encoder_rows(f, new_longest, extra_rows, True, align)
            if new_longest != longest:           # a longer clip lifts the cap min(frames + extra, longest) of the earlier ones
                new_rows = sum(encoder_rows(int(g), new_longest, extra_rows, True, align) for g in frames[beg:i + 1])
        else:
            new_rows = (i + 1 - beg) * encoder_rows(f, new_longest, extra_rows, False, align)
        if i > beg and new_rows > budget_rows:
            plan.append((beg, i))
            beg, longest = i, f
            rows = encoder_rows(f, f, extra_rows, packed, align)
            continue
        rows, longest = new_rows, new_longest
    if len(frames) > beg:
        plan.append((beg, len(frames)))
    return plan


def pack_arena(params: Sequence[torch.Tensor]) -> torch.Tensor:
    """Flatten parameters into one contiguous fp32 arena (same device as the first parameter)."""
    return torch.cat([p.detach().reshape(-1).to(torch.float32) for p in params])


def unpack_arena(arena: torch.Tensor, params: Sequence[torch.Tensor]) -> None:
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            p.copy_(arena[off:off + n].view_as(p))
            off += n
    if off != arena.numel():
        raise ValueError(f"arena holds {arena.numel()} elements, parameters need {off}")


def _collective_device(t: torch.Tensor) -> torch.Tensor:
    """RCCL ("nccl") moves device tensors; gloo (CPU tests, single-GPU dry runs) needs host tensors."""
    if dist.get_backend() == "gloo" and t.is_cuda:
        return t.cpu()
    return t


def broadcast_model(model: torch.nn.Module, src: int = 0) -> int:
    """Ship rank `src`'s weights to every rank as ONE packed arena; returns the arena size in bytes. Handles created
    from the parameters (HipModule) are marked dirty so the device copy is refreshed before the next forward."""
    params = [p for _, p in model.named_parameters()]
    if not params:
        return 0
    arena = _collective_device(pack_arena(params))
    dist.broadcast(arena, src=src)
    unpack_arena(arena.to(params[0].device), params)
    for m in model.modules():
        if hasattr(m, "mark_dirty"):
            m.mark_dirty()
    return arena.numel() * 4


def pack_hypotheses(ids: Sequence[Sequence[int]], n_pad: int, device=None) -> torch.Tensor:
    """[B, 1 + n_pad] int32: column 0 = token count, then the ids, -1 padded (fixed stride so one gather suffices)."""
    out = torch.full((len(ids), 1 + n_pad), -1, dtype=torch.int32)
    for b, r in enumerate(ids):
        r = list(r)[:n_pad]
        out[b, 0] = len(r)
        if r:
            out[b, 1:1 + len(r)] = torch.tensor(r, dtype=torch.int32)
    return out if device is None else out.to(device)


def pack_hypotheses_device(ids: torch.Tensor, counts: Sequence[int], n_pad: int) -> torch.Tensor:
    """The same [B, 1 + n_pad] int32 layout built ON THE DEVICE from the decoder's arg-max tensor `ids` [B, N] (int32, what
    ParaformerSANMDecoder.greedy returns) and the per-clip token counts: three tensor ops, no per-clip host loop, and the
    gather can start without the ids ever visiting the host."""
    B, N = ids.shape
    n = min(N, n_pad)
    cnt = torch.as_tensor(list(counts), dtype=torch.int32, device=ids.device).clamp_(max=n_pad)
    out = torch.full((B, 1 + n_pad), -1, dtype=torch.int32, device=ids.device)
    out[:, 0] = cnt
    keep = torch.arange(n, device=ids.device)[None, :] < cnt[:, None]
    out[:, 1:1 + n] = torch.where(keep, ids[:, :n].to(torch.int32), out[:, 1:1 + n])
    return out


def gather_packed(mine: torch.Tensor, dst: int = 0):
    """gather of equally shaped packed hypothesis tensors; on `dst` the list of per-rank id lists, elsewhere None"""
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = _collective_device(mine.contiguous())
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, bufs, dst=dst)
    if rank != dst:
        return None
    return [unpack_hypotheses(b) for b in bufs]


def unpack_hypotheses(t: torch.Tensor) -> List[List[int]]:
    t = t.cpu()
    return [t[b, 1:1 + int(t[b, 0])].tolist() for b in range(t.shape[0])]


def gather_hypotheses(ids: Sequence[Sequence[int]], n_pad: int, dst: int = 0, device=None):
    """Every rank contributes the same number of clips B. Returns on `dst` a list (one entry per rank) of per-clip id
    lists, elsewhere None."""
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = _collective_device(pack_hypotheses(ids, n_pad, device))
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, bufs, dst=dst)
    if rank != dst:
        return None
    return [unpack_hypotheses(b) for b in bufs]


def recognize_sharded(decode: Callable[[List[int]], List[List[int]]], lengths: Sequence[int], n_pad: int = 512,
                      dst: int = 0, device=None):
    """Decode a corpus of len(lengths) clips across all ranks. `decode(indices)` returns the token ids of the clips
    `indices` (this rank's shard, in that order). Shards are padded to equal size with repeats of their last clip so
    the gather is fixed-stride. Returns on `dst` the hypotheses in corpus order, elsewhere None."""
    world, rank = dist.get_world_size(), dist.get_rank()
    shards = [shard_indices(lengths, world, r) for r in range(world)]
    width = max(len(s) for s in shards)
    mine = shards[rank]
    hyps = decode(list(mine)) if mine else []
    while len(hyps) < width:                      # pad with an empty hypothesis
        hyps = list(hyps) + [[]]
    gathered = gather_hypotheses(hyps, n_pad, dst=dst, device=device)
    if gathered is None:
        return None
    out: List[List[int]] = [[] for _ in range(len(lengths))]
    for r, shard in enumerate(shards):
        for k, i in enumerate(shard):
            out[i] = gathered[r][k]
    return out


# --------------------------------------------------------------------------------------------- the same split at the C ABI
class DeviceComm:
    """RCCL communicator of the C ABI (include/paraformer_hip.h pf_dp_*, csrc/dp_rccl.hip): what a host WITHOUT torch.distributed
    binds for the multi-GPU split -- the weights go rank `src` -> every rank straight into the module handles' HBM (one grouped
    broadcast per handle: no packed host arena, no nn.Parameter copy, no re-upload), the hypotheses come back through
    `pf_dp_gather_ids`. One process per GPU; the 128-byte unique id travels over any channel (`share_id`).

    `share_id(id_or_None) -> id`: called with the id on rank 0 and with None elsewhere, returns the id on every rank."""

    _KIND = {"pf_encoder": "encoder", "pf_predictor": "predictor", "pf_decoder": "decoder", "pf_ctc": "ctc"}

    def __init__(self, world: int, rank: int, device: torch.device, share_id: Callable):
        import ctypes as C
        from . import _lib
        self._lib, self.world, self.rank, self.device = _lib.load(), int(world), int(rank), torch.device(device)
        ident = None
        if rank == 0:
            buf = (C.c_char * 128)()
            _lib.check(min(0, self._lib.pf_dp_unique_id(buf, 128)), "pf_dp_unique_id")
            ident = bytes(buf)
        ident = share_id(ident)
        if not isinstance(ident, (bytes, bytearray)) or len(ident) != 128:
            raise ValueError("DeviceComm: share_id must return the 128-byte id on every rank")
        with torch.cuda.device(self.device):
            self._h = _lib.check_handle(self._lib.pf_dp_create(C.c_char_p(bytes(ident)), 128, self.world, self.rank), "pf_dp_create")

    @classmethod
    def from_process_group(cls, device: torch.device) -> "DeviceComm":
        """the id is shared through the default torch.distributed group (gloo or nccl: bootstrap only)"""
        def share(ident):
            box = [ident]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        return cls(dist.get_world_size(), dist.get_rank(), device, share)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None:
            self._lib.pf_dp_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def broadcast_module(self, m, src: int = 0) -> None:
        """one HipModule (encoder / predictor / decoder / CTC mirror): rank `src`'s weights -> the handle of every rank, in place.
        The HANDLE is authoritative afterwards; the nn.Parameters of the other ranks keep their old host-side values."""
        from . import _lib
        from .hip_module import stream_ptr
        kind = self._KIND.get(getattr(m, "_prefix", ""))
        if kind is None:
            raise TypeError(f"DeviceComm.broadcast_module: {type(m).__name__} has no device handle of a known kind")
        lib, h = m._ensure_handle()                       # root: pushes its parameters; others: creates the (empty) handle
        with torch.cuda.device(self.device):
            _lib.check(getattr(lib, "pf_dp_broadcast_" + kind)(self._h, h, int(src), stream_ptr()), "pf_dp_broadcast_" + kind)
        m._dirty = False                                  # do not push the stale host parameters over the broadcast weights

    def gather_ids(self, packed: torch.Tensor, dst: int = 0):
        """packed: int32 device tensor, same shape on every rank -> on `dst` a [world, *shape] int32 device tensor, else None"""
        from . import _lib
        from .hip_module import stream_ptr
        assert packed.dtype == torch.int32 and packed.is_cuda and packed.is_contiguous()
        out = torch.empty((self.world,) + tuple(packed.shape), dtype=torch.int32, device=packed.device) if self.rank == dst else None
        _lib.check(self._lib.pf_dp_gather_ids(self._h, packed.data_ptr(), packed.numel(), out.data_ptr() if out is not None else None,
                                              int(dst), stream_ptr()), "pf_dp_gather_ids")
        return out


def broadcast_model_device(model: torch.nn.Module, comm: DeviceComm, src: int = 0) -> int:
    """`broadcast_model` through the C ABI: every HipModule's weights go straight into its handle (DeviceComm.broadcast_module);
    parameters that live outside the handles (e.g. SenseVoiceSmall.embed, the query-frame table the host gathers from) go
    through the same communicator as one packed raw buffer (pf_dp_broadcast_raw) -- no torch.distributed needed, which is the
    point of DeviceComm. Returns the bytes moved into handles (each parameter counted once)."""
    from . import _lib
    from .hip_module import HipModule, stream_ptr
    moved, inside = 0, set()
    for m in model.modules():
        if isinstance(m, HipModule) and getattr(m, "_prefix", "") in DeviceComm._KIND:
            comm.broadcast_module(m, src)
            for p in m.parameters():
                if id(p) not in inside:                   # nested HipModules share parameters with their parent
                    inside.add(id(p))
                    moved += p.numel() * 4
    rest = [p for p in model.parameters() if id(p) not in inside]
    if rest:
        arena = pack_arena(rest).to(comm.device)
        with torch.cuda.device(comm.device):
            _lib.check(comm._lib.pf_dp_broadcast_raw(comm._h, arena.data_ptr(), arena.numel() * arena.element_size(), int(src), stream_ptr()),
                       "pf_dp_broadcast_raw")
            torch.cuda.current_stream().synchronize()
        unpack_arena(arena.to(rest[0].device), rest)
    return moved
