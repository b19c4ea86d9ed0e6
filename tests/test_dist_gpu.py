"""GPU test of the multi-GPU path (BASELINE.json configs[3]): the HIP model through
openmpl_amd.dist.ShardedLifter over RCCL, in fresh child processes (one per GPU), against the single-process result.

Batch-split invariance of the team kernels is bitwise (tests/test_gpu_parity.py) and ShardedLifter keeps every shard on them
(set_small_batch_engine(False)), so the sharded result must equal the single-GPU result BITWISE, whatever the world size.
World size 1 always runs; world size 2 is SKIPPED -- visibly -- when the box has one GPU.
The children are started by this process, which itself never initialises the GPU (tests/conftest.py orders this module
first and only counts devices)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(backend, world, out, batch, gather="stream", one_gpu=False):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if one_gpu else str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, WORKER, backend, out, str(batch), gather], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for p, log in zip(procs, logs):
        assert p.returncode == 0, "rank failed:\n" + log[-4000:]
    return dict(np.load(out))


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
@pytest.mark.parametrize("batch,gather", [(64, "stream"), (37, "stream"), (12, "stream"), (5, "stream"), (64, "overlap"), (12, "overlap")])
def test_sharded_lifter_rccl_matches_single_process_bitwise(tmp_path, batch, gather, world):
    """batch 12 / 5 at V = 4: shards (and at world 1 the whole batch of 5) of at most 80 token rows -- the sizes at which the
    small-batch engine would otherwise change the bits (ShardedLifter switches it off for its own calls).  Both places the
    collective can run: ordered into the compute stream (default) and on the process group's stream beside the next forward."""
    if torch.cuda.device_count() < world:
        pytest.skip("world size %d needs %d GPUs, this box has %d (the N > 1 RCCL leg stays unmeasured here)"
                    % (world, world, torch.cuda.device_count()))
    single = _run("single", 1, str(tmp_path / "single.npz"), batch)
    got = _run("nccl", world, str(tmp_path / ("w%d.npz" % world)), batch, gather)
    for i in range(2):
        assert got["full%d" % i].shape == (batch, 17, 3)
        assert np.array_equal(got["full%d" % i], single["full%d" % i]), "world %d full-batch call differs" % world
        assert np.array_equal(got["shard%d" % i], single["full%d" % i]), "world %d pre-sharded call differs" % world
    assert not np.array_equal(single["full0"], single["full1"])


@pytest.mark.gpu
@pytest.mark.parametrize("batch,gather", [(64, "stream"), (37, "overlap"), (5, "stream")])
def test_two_ranks_on_one_gpu_match_single_process_bitwise(tmp_path, batch, gather):
    """No box of this build has two GPUs, so the world-2 RCCL leg above is skipped everywhere.  This leg runs the SAME rank code
    (tests/dist_worker.py: ShardedLifter, shard_inputs, lift_shard with both exchanges issued before the first wait, uneven shards,
    shards of at most 80 token rows) with two ranks that SHARE the one GPU and exchange through gloo (device tensors staged through
    the host) -- everything but the RCCL transport: the HIP forward of each rank's shard, the shard arithmetic, the batch-invariant
    engine selection, the order of the gathered poses.  Bitwise the single-process result.  (Small launches: two processes on one
    GPU must not both want every compute unit for a persistent launch -- the single-tenant rule of INTEGRATION.md.)"""
    single = _run("single", 1, str(tmp_path / "single.npz"), batch)
    got = _run("gloo", 2, str(tmp_path / "w2.npz"), batch, gather, one_gpu=True)
    for i in range(2):
        assert got["full%d" % i].shape == (batch, 17, 3)
        assert np.array_equal(got["full%d" % i], single["full%d" % i]), "two ranks on one GPU: full-batch call differs"
        assert np.array_equal(got["shard%d" % i], single["full%d" % i]), "two ranks on one GPU: pre-sharded call differs"


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` invoked plainly (no torchrun) starts N fresh child ranks itself; with one GPU on the
    box N = 1 still goes through the same rank code path (process group of size 1 when --force-dist is given)."""
    n = min(2, torch.cuda.device_count())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1",
                        "--no-extra", "--no-cpu-baseline", "--force-dist"], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-4000:]
    import json
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == n and j["rccl_ranks"] == n and j["value"] > 0
    assert j["parity"]["max_scaled"] < 1e-4
