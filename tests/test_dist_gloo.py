"""World-size-2 gloo tests (CPU) of the batch-sharded multi-GPU path: partitioning + the single all-gather.

The model callable here is the ORACLE (test infrastructure) -- the HIP forward cannot run without a GPU; what
is under test is openmpl_amd/dist.py, which is device agnostic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from openmpl_amd import detrng
from openmpl_amd.dist import ShardedLifter, gather_outputs, shard_inputs, shard_range
from oracle import mpl_oracle


def test_shard_range_is_a_contiguous_balanced_partition():
    for batch in (0, 1, 2, 7, 8, 1023, 1024, 8192):
        for world in (1, 2, 3, 4, 8):
            r = [shard_range(batch, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == batch
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


FLAGS = dict(num_joints=17, embed_dim_ratio=32, num_heads=8, depth=1, num_views=2, pose_3d_emb_learnable=True)


def _worker(rank, world, port, batch, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    shapes = mpl_oracle.param_shapes(FLAGS)
    sd = {k: torch.from_numpy(v) for k, v in detrng.make_state_dict(shapes, seed=3).items()}
    p, r, c = detrng.make_inputs(batch, 2, seed=9)
    P, R, C = ([torch.from_numpy(x) for x in l] for l in (p, r, c))
    model = lambda poses, rays=None, centers=None: mpl_oracle.forward(sd, FLAGS, poses, rays, centers)
    lifter = ShardedLifter(model)
    out = lifter(P, rays=R, centers=C)
    # pre-sharded inputs (what a per-rank loader hands over), two exchanges in flight before the first wait
    ps, rs, cs, _ = shard_inputs(P, R, C, world, rank)
    h1 = lifter.lift_shard(ps, rs, cs, batch=batch)
    h2 = lifter.lift_shard([x * 0.5 for x in ps], rs, cs, batch=batch)
    o1, o2 = h1.wait(), h2.wait()
    assert torch.equal(o1, out) and o2.shape == out.shape and not torch.equal(o2, out)
    try:
        lifter.lift_shard([x[:-1] for x in ps], rs, cs, batch=batch)
        raise AssertionError("a wrong shard size must raise")
    except RuntimeError:
        pass
    lo, hi = shard_range(batch, world, rank)
    # explicit gather of a rank-tagged tensor checks ordering independently of the model
    tag = torch.full((hi - lo, 17, 3), float(rank)) + torch.arange(lo, hi).reshape(-1, 1, 1)
    g = gather_outputs(tag, batch)
    if rank == 0:
        full = mpl_oracle.forward(sd, FLAGS, P, R, C)
        # CPU GEMM blocking depends on the batch size, so compare to rounding, not bitwise
        q.put((max(mpl_oracle.rel_errors(out, full)) < 1e-6 and out.shape == full.shape, g[:, 0, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [8, 7])
def test_sharded_lifter_world2_gloo(batch):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    same, tags = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same, "sharded result differs from the single-process result"
    want = []
    for r in range(world):
        lo, hi = shard_range(batch, world, r)
        want += [float(r + i) for i in range(lo, hi)]
    assert tags == want


def _worker_w8(rank, world, port, batch, q):
    """configs[3]'s partition at its real world size: 8 ranks, uneven batch; the 'model' is a cheap per-pose map (what
    is under test is the partition and the ONE collective of openmpl_amd/dist.py, not the forward)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    g = torch.Generator().manual_seed(1)
    P = [torch.randn(batch, 17, 3, generator=g) for _ in range(4)]
    calls = []

    def model(poses, rays=None, centers=None):
        calls.append(poses[0].shape[0])
        return torch.stack([p * (v + 1) for v, p in enumerate(poses)], 0).sum(0) + 0.25

    lifter = ShardedLifter(model)
    out = lifter(P)                                              # DataParallel call shape: full batch on every rank
    lo, hi = shard_range(batch, world, rank)
    ps, _, _, (a, b) = shard_inputs(P, None, None, world, rank)
    assert (a, b) == (lo, hi) and calls == [hi - lo]
    h = lifter.lift_shard(ps, batch=batch)                       # pre-sharded call shape
    out2 = h.wait()
    want = model(P)
    ok = torch.equal(out, want) and torch.equal(out2, want) and out.shape == (batch, 17, 3)
    oks = [None] * world
    dist.all_gather_object(oks, bool(ok))
    if rank == 0:
        q.put((all(oks), [shard_range(batch, world, r) for r in range(world)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [8190, 8192])
def test_sharded_lifter_world8_gloo_uneven_batch(batch):
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_w8, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok, ranges = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok, "8-rank sharded result differs from the single-process result on some rank"
    sizes = [b - a for a, b in ranges]
    assert sum(sizes) == batch and max(sizes) - min(sizes) <= 1 and ranges[0][0] == 0 and ranges[-1][1] == batch
    if batch == 8192:
        assert sizes == [1024] * 8                               # BASELINE.json configs[3]: 1024 poses per GPU


def _worker_skew(rank, world, port, gather, q):
    """Rank skew: the ranks take turns being late (one of them sleeps inside its forward, a different one every step) while
    the caller pipelines exactly as bench.py does -- the exchange of step i is waited for after step i + 1 has been issued.
    Whatever the skew and wherever the collective runs, step i's result must be step i's poses from EVERY rank."""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    step = [0]

    def model(poses, rays=None, centers=None):
        if step[0] % world == rank:
            time.sleep(0.15)                                   # this rank is the slow one of this step
        return poses[0] * (step[0] + 1.0) + rank

    lifter = ShardedLifter(model, gather=gather)
    nloc, steps = 3, 6
    x = torch.arange(nloc * 17 * 3, dtype=torch.float32).reshape(nloc, 17, 3)
    got, pending = [], None
    t0 = time.perf_counter()
    for i in range(steps):
        step[0] = i
        h = lifter.lift_shard([x], batch=nloc * world)
        if gather == "stream":
            assert h._work is None                             # ordered into the stream: nothing left to wait for
        if pending is not None:
            got.append(pending.wait())
        pending = h
    got.append(pending.wait())
    dt = time.perf_counter() - t0
    ok = len(got) == steps
    for i, g in enumerate(got):
        want = torch.cat([x * (i + 1.0) + r for r in range(world)], 0)
        ok = ok and torch.equal(g, want)
    oks = [None] * world
    dist.all_gather_object(oks, (bool(ok), dt))
    if rank == 0:
        q.put(oks)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("gather", ["stream", "overlap"])
def test_rank_skew_does_not_mix_steps(gather):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_skew, args=(r, world, port, gather, q)) for r in range(world)]
    for p in procs:
        p.start()
    oks = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(o[0] for o in oks), "a step's exchange returned another step's (or rank's) poses under rank skew"
    # every step has exactly one slow rank (0.15 s).  "stream": the ranks meet once per step, six steps cost ~0.9 s on every rank
    # (not 0.9 s x the world size: skew does not accumulate).  "overlap": a rank may run one step ahead of the exchange, so the
    # alternating delays partly hide behind each other (~0.45 s)
    lo = 0.85 if gather == "stream" else 0.4
    assert all(lo < o[1] < 2.5 for o in oks), oks


def test_gather_mode_is_validated():
    with pytest.raises(ValueError):
        ShardedLifter(lambda *a, **k: None, gather="sometimes")


def test_single_rank_group_still_runs_the_collective():
    """World size 1: the all-gather is issued all the same (no single-GPU shortcut to go untested)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        import unittest.mock as mock
        x = torch.arange(5 * 17 * 3, dtype=torch.float32).reshape(5, 17, 3)
        with mock.patch.object(dist, "all_gather_into_tensor", wraps=dist.all_gather_into_tensor) as spy:
            lifter = ShardedLifter(lambda poses, rays=None, centers=None: poses[0] * 2)
            out = lifter([x])
            h = lifter.lift_shard([x], batch=5)
            out2 = h.wait()
            assert spy.call_count == 2
        assert torch.equal(out, x * 2) and torch.equal(out2, x * 2) and out.data_ptr() != x.data_ptr()
        assert torch.equal(gather_outputs(x, 5), x)
    finally:
        dist.destroy_process_group()


def test_sharded_lifter_asks_the_model_for_batch_invariant_bits():
    """A shard must equal the rows of the single-process result bit for bit whatever the world size leaves of the batch: the lifter
    switches the model's small-batch engine (another fp32 arithmetic for <= 80 token rows) off -- on the model itself or, for the
    cfg wrapper MultiView_MPL_G, on the model inside (.features) -- for the duration of ITS OWN calls only: the
    caller's setting is back afterwards, also when the forward raises, and merely wrapping a model changes nothing."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        class Inner:
            def __init__(self):
                self._small_batch_engine = True
                self.seen = []
                self.fail = False

            def set_small_batch_engine(self, mode):
                self._small_batch_engine = mode

            def __call__(self, poses, rays=None, centers=None):
                self.seen.append(self._small_batch_engine)
                if self.fail:
                    raise RuntimeError("forward failed")
                return poses[0] * 2

        class Wrapper:
            def __init__(self):
                self.features = Inner()

            def __call__(self, poses, rays=None, centers=None):
                return self.features(poses, rays=rays, centers=centers)

        x = torch.ones(3, 17, 3)
        inner = Inner()
        lifter = ShardedLifter(inner)
        assert inner._small_batch_engine is True and inner.seen == []           # wrapping alone has no side effect
        assert torch.equal(lifter([x]), x * 2)
        assert inner.seen == [False] and inner._small_batch_engine is True      # off during the call, restored after it
        inner.fail = True
        with pytest.raises(RuntimeError, match="forward failed"):
            lifter([x])
        assert inner._small_batch_engine is True
        w = Wrapper()
        w.features._small_batch_engine = "auto"
        assert torch.equal(ShardedLifter(w)([x]), x * 2)
        assert w.features.seen == [False] and w.features._small_batch_engine == "auto"
        assert torch.equal(ShardedLifter(lambda poses, rays=None, centers=None: poses[0])([x]), x)   # a plain callable: nothing to switch
    finally:
        dist.destroy_process_group()
