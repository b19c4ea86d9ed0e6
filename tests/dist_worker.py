"""Child process of tests/test_dist_gpu.py and tests/test_dist_gloo.py: one rank of the batch-sharded lifter.

    python tests/dist_worker.py <backend> <out.npz> <batch> [stream|overlap]     (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from env)

backend "single": no process group, plain model call (the reference result the sharded runs must equal bitwise);
backend "nccl" (= RCCL): the HIP model on cuda:LOCAL_RANK through ShardedLifter; rank 0 writes the npz;
backend "gloo": the same with the collective staged through the host -- lets TWO ranks share ONE GPU (LOCAL_RANK 0 for both)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

FLAGS = dict(num_joints=17, embed_dim_ratio=32, num_heads=8, depth=2, num_views=4, pose_3d_emb_learnable=True)


def main():
    backend, out_path, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    gather = sys.argv[4] if len(sys.argv) > 4 else "stream"
    from openmpl_amd import detrng
    from openmpl_amd.multiview_mpl import MultiView_MPL
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    m = MultiView_MPL(**FLAGS)
    detrng.fill_module_(m, seed=5)
    m = m.to(dev).eval()
    # the single-process reference of a sharded run is the batch-invariant arithmetic: ShardedLifter switches the small-batch
    # engine (another fp32 arithmetic for <= 80 token rows) off, and so does the reference run
    m.set_small_batch_engine(False)
    batches = []
    for step in range(2):
        p, r, c = detrng.make_inputs(batch, FLAGS["num_views"], seed=77, step=step)
        batches.append(tuple([torch.from_numpy(x).to(dev) for x in l] for l in (p, r, c)))
    res = {}
    with torch.no_grad():
        if backend == "single":
            for i, (P, R, C) in enumerate(batches):
                res["full%d" % i] = m(P, rays=R, centers=C).cpu().numpy()
        else:
            import torch.distributed as dist
            from openmpl_amd.dist import ShardedLifter, shard_inputs
            if backend == "nccl":
                dist.init_process_group(backend, device_id=dev)
            else:                       # gloo with device tensors (staged through the host): the one-GPU world-2 leg of test_dist_gpu.py
                dist.init_process_group(backend)
            lifter = ShardedLifter(m, gather=gather)
            # (1) the DataParallel call shape: full batch on every rank
            for i, (P, R, C) in enumerate(batches):
                res["full%d" % i] = lifter(P, rays=R, centers=C).cpu().numpy()
            # (2) pre-sharded inputs, both exchanges issued before the first wait ("overlap": in flight beside the next forward)
            hs = []
            for P, R, C in batches:
                p, r, c, _ = shard_inputs(P, R, C, world, rank)
                hs.append(lifter.lift_shard(p, r, c, batch=batch))
            for i, h in enumerate(hs):
                res["shard%d" % i] = h.wait().cpu().numpy()
            torch.cuda.synchronize()
            dist.barrier()
            dist.destroy_process_group()
    if rank == 0:
        np.savez(out_path, **res)


if __name__ == "__main__":
    main()
