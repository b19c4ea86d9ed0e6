"""Probe host CPU resources and the oracle's throughput vs torch thread count (bounded)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from openmpl_amd import detrng
from oracle import mpl_oracle
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
os.system("grep -m1 'model name' /proc/cpuinfo; nproc")
flags = dict(num_joints=17, embed_dim_ratio=32, num_heads=8, depth=12, num_views=4, pose_3d_emb_learnable=True)
shapes = mpl_oracle.param_shapes(flags)
sd = {k: torch.from_numpy(v) for k, v in detrng.make_state_dict(shapes, seed=11).items()}
for B in (256,):
    p, r, c = detrng.make_inputs(B, 4, seed=1)
    P = [torch.from_numpy(x) for x in p]; R = [torch.from_numpy(x) for x in r]; C = [torch.from_numpy(x) for x in c]
    for th in (8, 16, 32, 64, 128):
        torch.set_num_threads(th)
        mpl_oracle.forward(sd, flags, P, R, C)
        t0 = time.perf_counter(); n = 0
        while n < 2 or time.perf_counter() - t0 < 2.0:
            mpl_oracle.forward(sd, flags, P, R, C); n += 1
            if time.perf_counter() - t0 > 20: break
        dt = time.perf_counter() - t0
        print("B=%d threads=%d: %.1f poses/s (%d fwd in %.1fs)" % (B, th, B * n / dt, n, dt), flush=True)
