"""Batch-sharded multi-GPU inference: one process per GPU, contiguous batch shards, persistent weight
replicas, ONE all-gather of the per-shard (B/G, J, 3) outputs per batch.

This replaces the reference's only multi-GPU mechanism, ``torch.nn.DataParallel`` (valid_mpl.py:177-178:
scatter inputs, re-broadcast ALL parameters every forward, gather outputs to device 0), with the MI355X
shape of the same thing: every pose is independent, so there is no data-path
collective besides the output exchange -- 209 kB per rank at B=8192 over 8 GPUs, a latency-bound RCCL
all-gather over xGMI.  Backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [lo, hi) of a batch: the first ``batch % world`` ranks get one extra pose."""
    if batch < 0 or world <= 0 or not (0 <= rank < world):
        raise ValueError("bad shard_range arguments")
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_inputs(poses: Sequence[torch.Tensor], rays, centers, world: int, rank: int):
    """Slice this rank's contiguous shard out of full-batch view lists (views are lists of (B,J,3)/(B,1,3))."""
    lo, hi = shard_range(poses[0].shape[0], world, rank)
    cut = lambda lst: None if lst is None else [t[lo:hi].contiguous() for t in lst]
    return cut(poses), cut(rays), cut(centers), (lo, hi)


def gather_outputs(local: torch.Tensor, batch: int, group=None) -> torch.Tensor:
    """All-gather per-shard outputs (b_r, J, 3) into the full (batch, J, 3) tensor on every rank.

    Shards produced by ``shard_range`` differ by at most one pose; they are padded to the largest shard so
    that a single ``all_gather_into_tensor`` (one RCCL call) moves everything, then compacted locally.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(batch, world, rank)
    if local.shape[0] != hi - lo:
        raise RuntimeError("rank %d holds %d poses, its shard of %d is %d" % (rank, local.shape[0], batch, hi - lo))
    # a process group of ONE rank still runs the collective (RCCL accepts a 1-rank communicator): the same code path,
    # streams and events whatever the world size -- there is no separate single-GPU branch to go untested
    mx = -(-batch // world)
    if batch % world == 0:
        out = torch.empty((batch,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: hi - lo] = local
    buf = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, pad, group=group)
    parts = []
    for r in range(world):
        a, b = shard_range(batch, world, r)
        parts.append(buf[r * mx: r * mx + (b - a)])
    return torch.cat(parts, 0)


class GatherHandle:
    """Result of an asynchronous output exchange: ``wait()`` returns the full (batch, J, 3) tensor.

    With the RCCL backend the all-gather runs on the process group's own stream (it waits for the forward that produced
    ``local`` through an event, the compute stream is NOT blocked), so the forward of the next batch overlaps it;
    ``wait()`` makes the current stream wait for the collective -- call it when the gathered poses are needed."""

    def __init__(self, work, buf, batch, world, keep):
        self._work, self._buf, self._batch, self._world, self._keep = work, buf, batch, world, keep

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
        self._keep = None
        if self._buf.is_cuda:
            # a forward of this device that lost a hand-off produced NaN poses: raise here, with the batch it poisoned, not
            # on whatever call happens to come next (openmpl_amd.cabi.raise_if_device_error)
            from . import cabi
            cabi.raise_if_device_error(self._buf.device.index)
        buf, batch, world = self._buf, self._batch, self._world
        if batch % world == 0:
            return buf
        mx = -(-batch // world)
        parts = []
        for r in range(world):
            a, b = shard_range(batch, world, r)
            parts.append(buf[r * mx: r * mx + (b - a)])
        return torch.cat(parts, 0)


GATHER_MODES = ("stream", "overlap")


def gather_outputs_async(local: torch.Tensor, batch: int, group=None, mode: str = "overlap") -> GatherHandle:
    """Start the ONE all-gather of a batch (see gather_outputs).

    mode "overlap": the collective runs on the process group's own stream behind an event and the stream that produced ``local``
    is NOT blocked -- the next forward overlaps it, and ``GatherHandle.wait()`` is where the current stream is made to wait.
    mode "stream": the collective is ordered INTO the current stream -- it starts behind the forward that produced ``local`` and
    whatever the caller enqueues next starts behind it (a stream-level wait, the host does not block): see ShardedLifter."""
    if mode not in GATHER_MODES:
        raise ValueError("gather mode must be one of %r" % (GATHER_MODES,))
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(batch, world, rank)
    if local.shape[0] != hi - lo:
        raise RuntimeError("rank %d holds %d poses, its shard of %d is %d" % (rank, local.shape[0], batch, hi - lo))
    mx = -(-batch // world)             # world == 1 included: the collective always runs (see gather_outputs)
    src = local.contiguous()
    if batch % world:
        src = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        src[: hi - lo] = local
    buf = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if mode == "stream":
        # a SYNCHRONOUS collective: ordered with respect to the current stream (torch >= 2.7 enqueues it on that very stream, older
        # versions on the process group's stream between two events); the host does not block with RCCL (gloo: it does)
        dist.all_gather_into_tensor(buf, src, group=group, async_op=False)
        work = None
    else:
        work = dist.all_gather_into_tensor(buf, src, group=group, async_op=True)
    return GatherHandle(work, buf, batch, world, src)


class ShardedLifter:
    """Batch-sharded lifting with one resident model replica per rank.

    ``lifter(poses, rays=..., centers=...)`` takes FULL-batch inputs on every rank (each rank slices its contiguous
    shard) and returns the full (B,J,3) result on every rank -- the call shape of the reference's DataParallel wrapper
    (valid_mpl.py:177-178).  ``lifter.lift_shard(poses, ..., batch=B)`` takes inputs that are ALREADY sharded (rank r
    holds poses shard_range(B, world, r) -- what a per-rank data loader produces: no rank ever touches another rank's
    frames) and returns a GatherHandle.

    ``gather`` -- where the output exchange of batch i runs relative to the forward of batch i+1:

    * ``"stream"`` (default): ordered into the compute stream, between the tail kernel of batch i and the first kernel of batch
      i+1.  The block stack of a forward is ONE persistent launch that takes every compute unit (one 160-KiB workgroup per CU,
      teams spin on their partners: csrc/h2_phase.hpp h2_launch_stack); a collective kernel on another stream would have to
      wait for a compute unit of that launch, or -- when it got there first -- keep one workgroup of it out while it waits for
      the SAME collective of a slower rank, whose kernel in turn waits behind that rank's persistent launch.  In the stream
      the collective of a rank meets nothing of its own device: rank skew costs what the slowest rank's forward costs, once.
      Price: the 209-kB exchange (tens of microseconds) sits on the critical path of a ~1.3-ms step.
    * ``"overlap"``: on the process group's stream behind an event; the next forward is enqueued before the wait (rounds 1-5).
      Kept for launches that leave compute units free (small batches) and for the A/B in bench.py (extra.force_dist_*)."""

    def __init__(self, model: Callable, group=None, gather: str = "stream"):
        if gather not in GATHER_MODES:
            raise ValueError("gather mode must be one of %r" % (GATHER_MODES,))
        self.model = model
        self.group = group
        self.gather = gather
        # a shard must equal the rows of the single-process result bit for bit, whatever the world size leaves of the batch: the
        # small-batch engine (<= 80 token rows, another fp32 arithmetic: ~1e-7 apart) is therefore off DURING a sharded call
        # (MultiView_MPL.set_small_batch_engine; models wrapped in MultiView_MPL_G expose it through .features).  It is a
        # per-call override: the caller's own setting is restored afterwards, direct model(...) calls keep the engine they chose
        self._engine_owner = next((m for m in (getattr(model, "features", None), model)
                                   if hasattr(m, "set_small_batch_engine") and hasattr(m, "_small_batch_engine")), None)

    def _forward_batch_invariant(self, poses, rays, centers):
        own = self._engine_owner
        if own is None:
            return self.model(poses, rays=rays, centers=centers)
        keep = own._small_batch_engine
        own.set_small_batch_engine(False)
        try:
            return self.model(poses, rays=rays, centers=centers)
        finally:
            own.set_small_batch_engine(keep)

    def lift_shard(self, poses, rays=None, centers=None, batch: Optional[int] = None) -> GatherHandle:
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        nloc = poses[0].shape[0]
        if batch is None:
            batch = nloc * world                       # equal shards
        lo, hi = shard_range(batch, world, rank)
        if nloc != hi - lo:
            raise RuntimeError("rank %d was handed %d poses, its shard of %d is %d" % (rank, nloc, batch, hi - lo))
        if nloc:
            out = self._forward_batch_invariant(poses, rays, centers)
            if isinstance(out, tuple):                  # head_kadkhod returns (x3, [x1, x2]): exchange the final estimate
                out = out[0]
        else:
            out = poses[0].new_zeros((0, poses[0].shape[1], 3))
        return gather_outputs_async(out, batch, self.group, self.gather)

    def __call__(self, poses, rays=None, centers=None):
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        B = poses[0].shape[0]
        p, r, c, _ = shard_inputs(poses, rays, centers, world, rank)
        return self.lift_shard(p, r, c, batch=B).wait()
