#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X multi-view pose-lifting forward pass.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): poses/s at V=4, J=17, batch 1024 (per GPU), fp32, the paper's CHOSEN flag set
(configs/h36m: NETWORK.DIM 32, depth 12, heads 8).  A "step" = one forward of one batch of 1024 synthetic
poses that are already resident in HBM when the timed region starts; with N ranks every rank lifts its own
1024 poses (weak scaling) and the per-shard (B,17,3) outputs are exchanged with ONE RCCL all-gather per
step, the MI355X equivalent of the reference's DataParallel gather (valid_mpl.py:178).

Rank 0 prints ONE JSON line: value = whole-job poses/s, plus
  roofline     -- dominant kernel (the GEMMs of the FPT blocks): algorithmic fp32 FLOPs per launch / mean launch
                  duration measured live with HIP events on the launch stream (mpl_profile_start/stop) against the
                  157.3 TFLOP/s fp32 matrix peak.  The default fp32 path computes those GEMMs on the bf16 matrix
                  cores from exactly split operands (csrc/x3_gemm.hip: fp32 in, fp32 out, at least fp32-accurate
                  products, fp32 accumulation), so `matrix_pipe` also prices the executed bf16 MFMA work against
                  the 2.5 PFLOP/s bf16 peak; --precision fp32_mfma runs the native fp32 MFMA kernels instead.
  cpu_baseline -- the oracle (a port of the reference's CPU PyTorch path) timed on this box's host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 matrix peak (same guide)
PEAK_HBM_GBS = 8000.0

CHOSEN = dict(pose_3d_emb_learnable=True)
FULL = dict(pose_3d_emb_learnable=True, confidence_input_as_third=True, input_rays_as_token=True,
            multiple_spatial_blocks=True, add_3D_pos_encoding_to_rays=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024, help="poses per GPU per step")
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--flagset", choices=("chosen", "full"), default="chosen")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary FULL-flag-set measurement")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp32_mfma"],
                    help="fp32: split-operand GEMMs on the bf16 matrix cores (default); fp32_mfma: native fp32 MFMA")
    return ap.parse_args()


def effective_cpus():
    """Host cores this process may actually use: min(affinity, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pr = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // pr))
        except Exception:
            pass
    return max(1, n)


def build_model(flagset, views, depth, dev):
    import torch
    from openmpl_amd import detrng
    from openmpl_amd.multiview_mpl import MultiView_MPL
    flags = dict(num_joints=17, embed_dim_ratio=32, num_heads=8, depth=depth, num_views=views)
    flags.update(CHOSEN if flagset == "chosen" else FULL)
    m = MultiView_MPL(**flags)
    detrng.fill_module_(m, seed=11)          # random-init weights of the named architecture (no checkpoints offline)
    return m.to(dev).eval(), flags


def make_batch(batch, views, dev, seed, step=0):
    import torch
    from openmpl_amd import detrng
    p, r, c = detrng.make_inputs(batch, views, seed=seed, step=step)
    mk = lambda lst: [torch.from_numpy(x).to(dev) for x in lst]
    return mk(p), mk(r), mk(c)


def gemm_flops_per_forward(flags, batch):
    """Algorithmic FLOPs of the ln_gemm launches of one forward: per FPT block 16*N*D^2 (= the four Linear
    layers, 2*MAC) x (depth+1) applications (the 4*N^2*D attention term runs in another kernel)."""
    D = 17 * 32 * (2 if flags.get("input_rays_as_token") else 1)
    V = flags["num_views"]
    apps = flags["depth"] + 1
    return apps * 16.0 * V * D * D * batch, apps * 4


def timed_steps(model, batches, steps, warmup, dist, gather_buf):
    import torch
    if dist is not None:
        from openmpl_amd.dist import gather_outputs as _go
        gather_outputs = lambda out, total: _go(out, total)
    def step(i):
        P, R, C = batches[i % len(batches)]
        out = model(P, rays=R, centers=C)
        if dist is not None:
            out = gather_outputs(out, gather_buf)      # ONE RCCL all-gather of the (B/G,17,3) shards
        return out
    with torch.no_grad():
        for i in range(warmup):
            step(i)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0


def main():
    a = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist_mod.init_process_group("nccl", device_id=dev)   # backend "nccl" IS RCCL on ROCm
        dist = dist_mod

    model, flags = build_model(a.flagset, a.views, a.depth, dev)
    model.set_matmul_precision(a.precision)
    split = a.precision == "fp32" and model._x3_supported()
    # a few distinct resident batches so that no step can reuse a cached result
    batches = [make_batch(a.batch, a.views, dev, seed=1000 + rank, step=s) for s in range(4)]
    gather_buf = world * a.batch if dist is not None else None   # global batch size of the gathered result

    dt = timed_steps(model, batches, a.steps, a.warmup, dist, gather_buf)
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    value = world * a.batch * a.steps / dt

    result = None
    if rank == 0:
        from openmpl_amd import cabi
        from oracle import mpl_oracle
        # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream
        P, R, C = batches[0]
        n_prof = 5
        with torch.no_grad():
            cabi.profile_start()
            for _ in range(n_prof):
                model(P, rays=R, centers=C)
            torch.cuda.synchronize()
            prof = cabi.profile_stop()
        gemm_ms, gemm_n = prof["gemm"]
        fl, launches = gemm_flops_per_forward(flags, a.batch)
        assert gemm_n == launches * n_prof, (gemm_n, launches)
        avg_launch_ms = gemm_ms / gemm_n
        achieved = (fl / launches) / (avg_launch_ms * 1e-3) / 1e12
        kernel_ms = {k: round(t / n_prof, 4) for k, (t, n) in prof.items()}
        total_flop = mpl_oracle.flop_count({k: v for k, v in flags.items()})
        io_bytes = (a.views * (17 * 2 if a.flagset == "chosen" else 17 * 6 + 3) * 4 + 51 * 4)   # 748 / 1884 B
        weight_bytes = sum(p.numel() for p in model.parameters()) * 4
        # HBM/fabric bytes per launch cannot be counted from inside this process: they come from the committed
        # rocprofv3 PMC passes of the same kernels (profiles/), and only for the profiled workload shape.
        traffic, traffic_src = None, None
        try:
            if a.flagset == "chosen" and a.batch == 1024 and a.views == 4:
                tj = json.load(open(os.path.join(ROOT, "profiles", "r01_gemm_traffic.json")))
                if tj.get("kernel") != ("x3_gemm_kernel" if split else "ln_gemm_ng_kernel"):
                    raise KeyError("profile is for the other GEMM kernel")
                traffic, traffic_src = round(tj["traffic_bytes_per_launch"]), tj["source"]
        except Exception:
            pass
        # executed matrix-pipe work of the split path: 6 bf16 products per fp32 product on 144-column (9 x 16) tiles
        pipe = None
        if split:
            ex = achieved * 6.0 * 144.0 / 136.0
            pipe = dict(instruction="v_mfma_f32_16x16x32_bf16", executed=round(ex, 1), peak=PEAK_BF16_MFMA_TFLOPS,
                        unit="TFLOP/s", frac=round(ex / PEAK_BF16_MFMA_TFLOPS, 4),
                        note="6 bf16 partial products per fp32 product (3-way exact operand split), 9 MFMA column "
                             "tiles per 136 output columns")
        roof = dict(bound="mfma", kernel="x3_gemm_kernel" if split else "ln_gemm_ng_kernel",
                    arithmetic=("fp32 operands split exactly into 3 bf16 terms, 6 significant partial products per "
                                "product on the bf16 matrix cores, fp32 accumulation (error vs fp64 <= native fp32)")
                    if split else "native fp32 MFMA (v_mfma_f32_16x16x4_f32)",
                    achieved=round(achieved, 2), peak=PEAK_FP32_MFMA_TFLOPS, matrix_pipe=pipe,
                    unit="TFLOP/s", frac=round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), traffic=traffic,
                    traffic_source=traffic_src,
                    avg_launch_us=round(avg_launch_ms * 1e3, 2), launches_per_step=launches,
                    flops_per_launch=fl / launches,
                    whole_forward_tflops=round(value / world * total_flop / 1e12, 2),
                    whole_forward_frac=round(value / world * total_flop / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                    hbm_frac=round(value / world * (io_bytes + weight_bytes / a.batch) / 1e9 / PEAK_HBM_GBS, 5),
                    kernel_ms_per_step=kernel_ms)

        # ---- parity of this very run against the oracle (bounded: 64 poses)
        nb = min(64, a.batch)
        cp = [x[:nb].cpu() for x in P]; cr = [x[:nb].cpu() for x in R]; cc = [x[:nb].cpu() for x in C]
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        with torch.no_grad():
            got = model([x[:nb].contiguous() for x in P], rays=[x[:nb].contiguous() for x in R],
                        centers=[x[:nb].contiguous() for x in C]).cpu()
        ref = mpl_oracle.forward(sd, flags, cp, cr, cc)
        mx, nw = mpl_oracle.rel_errors(got, ref)
        parity = dict(max_scaled=float("%.3e" % mx), norm_wise=float("%.3e" % nw),
                      mpjpe_vs_ref=float("%.3e" % mpl_oracle.mpjpe(got, ref)), poses=nb, tol=1e-4)

        # ---- CPU baseline: the oracle on the host cores, bounded sample of the same workload
        cpu = None
        if not a.no_cpu_baseline:
            cores = effective_cpus()
            torch.set_num_threads(cores)
            cb = min(a.batch, 256)                     # bounded sample: batches of 256 poses of the same workload
            cp = [x[:cb].cpu() for x in P]; cr = [x[:cb].cpu() for x in R]; cc = [x[:cb].cpu() for x in C]
            mpl_oracle.forward(sd, flags, cp, cr, cc)      # warm-up
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 12.0 and reps < 400:
                mpl_oracle.forward(sd, flags, cp, cr, cc)
                reps += 1
            cdt = time.perf_counter() - t0
            cpu = dict(value=round(cb * reps / cdt, 1), unit="poses/s", cores=cores, kind="port",
                       sample="%d forwards of %d poses in %.1f s (V=%d, depth %d, fp32, torch threads=%d = cgroup CPU "
                              "quota of the box), oracle/mpl_oracle.py" % (reps, cb, cdt, a.views, a.depth, cores))

        extra = {}
        if not a.no_extra and world == 1 and a.flagset == "chosen":
            # the other fp32 engine on the same workload, and all of them against an fp64 evaluation of the reference
            # semantics: the split-operand path must not be less accurate than fp32 arithmetic
            other = "fp32_mfma" if a.precision == "fp32" else "fp32"
            model.set_matmul_precision(other)
            n_o = max(10, a.steps // 2)
            v_o = a.batch * n_o / timed_steps(model, batches, n_o, 3, None, None)
            with torch.no_grad():
                got_o = model([x[:nb].contiguous() for x in P], rays=[x[:nb].contiguous() for x in R],
                              centers=[x[:nb].contiguous() for x in C]).cpu()
            model.set_matmul_precision(a.precision)
            ref64 = mpl_oracle.forward(sd, flags, [x[:nb].cpu() for x in P], [x[:nb].cpu() for x in R],
                                       [x[:nb].cpu() for x in C], dtype=torch.float64)
            e = lambda y: float("%.3e" % mpl_oracle.rel_errors(y.double(), ref64)[0])
            extra[other + "_poses_per_s"] = round(v_o, 1)
            extra["max_scaled_err_vs_fp64"] = {"hip_" + a.precision: e(got), "hip_" + other: e(got_o),
                                               "reference_fp32_cpu": e(ref)}
            # secondary: the FULL flag set of hm_0_...yaml (per-view SPT, conf channel, ray tokens, FPT width 1088)
            m2, f2 = build_model("full", a.views, a.depth, dev)
            dt2 = timed_steps(m2, batches, max(5, a.steps // 4), 3, None, None)
            v2 = a.batch * max(5, a.steps // 4) / dt2
            extra["full_flagset_poses_per_s"] = round(v2, 1)
            extra["full_flagset_tflops"] = round(v2 * mpl_oracle.flop_count(f2) / 1e12, 2)
            del m2
            # BASELINE.json configs[2]: CMU Panoptic shape V=8, batch 1024, bf16 matrix cores (yaml depth 2), with the
            # fp32 run of the same shape and the bf16 deviation from the fp32 reference semantics (reported, not gated)
            m3, f3 = build_model("chosen", 8, 2, dev)
            b3 = [make_batch(a.batch, 8, dev, seed=2000, step=s) for s in range(2)]
            n3 = max(10, a.steps // 2)
            v32 = a.batch * n3 / timed_steps(m3, b3, n3, 3, None, None)
            m3.set_matmul_precision("bf16")
            v16 = a.batch * n3 / timed_steps(m3, b3, n3, 3, None, None)
            P3, R3, C3 = b3[0]
            with torch.no_grad():
                o16 = m3([x[:64].contiguous() for x in P3], rays=[x[:64].contiguous() for x in R3],
                         centers=[x[:64].contiguous() for x in C3]).cpu()
            sd3 = {k: v.detach().cpu() for k, v in m3.state_dict().items()}
            r3 = mpl_oracle.forward(sd3, f3, [x[:64].cpu() for x in P3], [x[:64].cpu() for x in R3], [x[:64].cpu() for x in C3])
            mx3, nw3 = mpl_oracle.rel_errors(o16, r3)
            # PCIe-inclusive rate of the headline workload: the same forwards fed from pinned host
            # tensors, H2D of the V x (B,17,3) poses (+ rays, centers: the API's 1680 B/pose) inside the timed region
            host = [tuple([t.cpu().pin_memory() for t in lst] for lst in b) for b in batches[:2]]
            def h2d_step(i):
                Ph, Rh, Ch = host[i % 2]
                up = lambda lst: [t.to(dev, non_blocking=True) for t in lst]
                return model(up(Ph), rays=up(Rh), centers=up(Ch))
            with torch.no_grad():
                for i in range(3):
                    h2d_step(i)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n_o):
                    h2d_step(i)
                torch.cuda.synchronize()
            extra["h2d_inclusive_poses_per_s"] = round(a.batch * n_o / (time.perf_counter() - t0), 1)
            # BASELINE.json configs[4]: large-view stress V=31, batch 256 (31-token FPT; separate attention kernel), and
            # the 17V = 527-token joints x views grid (KPTOK, LDS-resident K/V of one head)
            for tag, fl in (("v31_b256_chosen", {}), ("v31_b256_kptok", dict(FPT_blocks_view_keypoint_tokens=True))):
                from openmpl_amd import detrng as _dr
                from openmpl_amd.multiview_mpl import MultiView_MPL as _M
                f5 = dict(num_joints=17, embed_dim_ratio=32, num_heads=8, depth=a.depth, num_views=31, **CHOSEN, **fl)
                m5 = _M(**f5)
                _dr.fill_module_(m5, seed=11)
                m5 = m5.to(dev).eval()
                b5 = [make_batch(256, 31, dev, seed=3000, step=s) for s in range(2)]
                n5 = max(5, a.steps // 5)
                extra[tag + "_poses_per_s"] = round(256 * n5 / timed_steps(m5, b5, n5, 2, None, None), 1)
                del m5
            extra["cmu_v8_depth2"] = {"fp32_poses_per_s": round(v32, 1), "bf16_poses_per_s": round(v16, 1),
                                      "bf16_max_scaled_vs_ref": float("%.3e" % mx3), "bf16_norm_wise_vs_ref": float("%.3e" % nw3),
                                      "bf16_mpjpe_vs_ref": float("%.3e" % mpl_oracle.mpjpe(o16, r3))}
            del m3

        result = {
            "metric": "poses/sec (V=%d, J=17, batch=%d per GPU) fp32" % (a.views, a.batch),
            "value": round(value, 1), "unit": "poses/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Human3.6M config: V=%d J=17 batch=%d fp32, %s flag set, depth %d, DIM 32, heads 8"
                                   % (a.views, a.batch, a.flagset.upper(), a.depth),
                       "global_batch": world * a.batch, "parallelism": "dp%d (batch shards + 1 all_gather/step)" % world
                       if world > 1 else "single GPU"},
            "roofline": roof, "cpu_baseline": cpu, "parity": parity, "extra": extra,
        }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
