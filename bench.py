#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X multi-view pose-lifting forward pass.

    python bench.py --gpus N --steps K --warmup W          (plain: for N > 1 it starts its own N child ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             (also fine: RANK / WORLD_SIZE / LOCAL_RANK from the env)

Metric (BASELINE.json): poses/s at V=4, J=17, batch 1024 (per GPU), fp32, the paper's CHOSEN flag set
(configs/h36m: NETWORK.DIM 32, depth 12, heads 8).  A "step" = one forward of one batch of 1024 synthetic poses
that are already resident in HBM when the timed region starts; with N ranks every rank lifts its own 1024 poses
(weak scaling, inputs pre-sharded: no rank touches another rank's frames) and the per-shard (B,17,3) outputs are
exchanged with ONE RCCL all-gather per step -- the MI355X equivalent of the reference's DataParallel gather
(valid_mpl.py:177-178) -- ordered into the compute stream between two forwards (--gather stream, default: the block stack of a
forward is one persistent launch on every compute unit, a collective kernel beside it would fight it for one) or issued
asynchronously beside the next step's forward (--gather overlap: rounds 1-5).

When invoked plainly with --gpus N > 1 the parent process imports nothing that touches the GPU: it starts N fresh
children (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and relays rank 0's line.

Rank 0 prints ONE JSON line: value = whole-job poses/s, plus
  roofline     -- dominant kernel (h2_stack_kernel: every GEMM of the FPT block stack in one persistent launch): FLOPs per
                  launch / mean launch duration, measured live with HIP events on the launch stream (mpl_profile_start/stop)
                  over a whole region of `steps` forwards.  The default fp32 path computes those GEMMs on the fp16 matrix
                  cores from operands split in two fp16 parts (fp32 in, fp32 out, fp32 accumulation, three partial products
                  per product: csrc/h2_gemm.hip), so the binding ceiling is the 16-bit matrix pipe: `frac` = `frac_useful` =
                  algorithmic FLOPs x 3 products / 2.5 PFLOP/s; `frac_executed` adds the 144/136 column-tile padding the
                  pipe also executes; `fp32_equivalent` keeps the algorithmic fp32 FLOP/s against the 157.3 TFLOP/s fp32
                  matrix peak.  `kernels` lists every kernel kind of the forward (SPT included) the same way; `rooflines`
                  repeats the entry for the other engines (bf16 at V = 8, the small-batch engine at one frame).
  cpu_baseline -- the oracle (a port of the reference's CPU PyTorch path) timed on this box's host cores, headline
                  workload first, BASELINE.md section 4's other configurations under `others`.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import subprocess
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
    ap.add_argument("--flagset", choices=("chosen", "full", "kptok"), default="chosen",
                    help="chosen / full: the two shipped YAML families; kptok: CHOSEN + FPT_blocks_view_keypoint_tokens (the joints x "
                         "views token grid, 17 V tokens of width 32: BASELINE configs[4] with --views 31 --batch 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements (other configs / engines)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the rank code path (process group + all-gather) even at one GPU")
    ap.add_argument("--gather", choices=("stream", "overlap"), default="stream",
                    help="where the per-step RCCL all-gather runs: ordered into the compute stream between the tail kernel of step i "
                         "and the first kernel of step i+1 (default: nothing competes with the persistent block-stack launch for "
                         "compute units), or on the process group's stream beside the next forward (rounds 1-5)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp32_mfma", "bf16"],
                    help="fp32: fp16x2 split-operand GEMMs on the fp16 matrix cores (default); fp32_mfma: native fp32 MFMA; "
                         "bf16: bf16 operands (BASELINE.json configs[2]: use with --views 8 --depth 2 / 12)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------ parent: start the ranks
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n: int) -> int:
    """Start n fresh child processes of this script, one per GPU.  Nothing GPU-related has been imported here."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in procs:                            # one rank died: the others would hang in a collective
                        q.terminate()
            time.sleep(0.05)
    except KeyboardInterrupt:
        for q in procs:
            q.terminate()
        rc = 130
    return rc


# ------------------------------------------------------------------------------------------ helpers
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def model_flags(flagset, views, depth, **more):
    flags = dict(num_joints=17, embed_dim_ratio=32, num_heads=8, depth=depth, num_views=views)
    flags.update(FULL if flagset == "full" else CHOSEN)
    if flagset == "kptok":
        flags["FPT_blocks_view_keypoint_tokens"] = True
    flags.update(more)
    return flags


def build_model(flags, dev):
    from openmpl_amd import detrng
    from openmpl_amd.multiview_mpl import MultiView_MPL
    m = MultiView_MPL(**flags)
    detrng.fill_module_(m, seed=11)          # random-init weights of the named architecture (no checkpoints offline)
    return m.to(dev).eval()


def make_batch(batch, views, dev, seed, step=0):
    import torch
    from openmpl_amd import detrng
    p, r, c = detrng.make_inputs(batch, views, seed=seed, step=step)
    mk = lambda lst: [torch.from_numpy(x).to(dev) for x in lst]
    return mk(p), mk(r), mk(c)


def fpt_width(flags):
    if flags.get("FPT_blocks_view_keypoint_tokens"):
        return 32
    return 17 * 32 * (2 if flags.get("input_rays_as_token") else 1)


def gemm_flops_per_forward(flags, batch):
    """Algorithmic FLOPs of the FPT GEMM launches of one forward: per block 16*N*D^2 (= the four Linear layers,
    2*MAC) x (depth+1) applications -- the 4*N^2*D attention term that the qkv instance also
    executes in its epilogue is NOT counted."""
    D = fpt_width(flags)
    apps = flags["depth"] + 1
    return apps * 16.0 * flags["num_views"] * D * D * batch, apps * 4


def spt_flops_per_forward(flags, batch):
    """SPT stage (one launch): per view 17 tokens x 32 channels, depth+1 block applications."""
    N, D = 17, 32
    per_view = (flags["depth"] + 1) * (16.0 * N * D * D + 4.0 * N * N * D)
    return per_view * flags["num_views"] * batch


def timed_steps(model, batches, steps, warmup, lifter=None, global_batch=None):
    """EXACTLY `steps` forwards between two (barrier +) device synchronisations.  With a lifter every step also starts
    its all-gather; the wait for step i's exchange is issued after step i+1's forward has been enqueued."""
    import torch
    import torch.distributed as dist

    def run(n):
        pending = None
        for i in range(n):
            P, R, C = batches[i % len(batches)]
            if lifter is None:
                model(P, rays=R, centers=C)
            else:
                h = lifter.lift_shard(P, R, C, batch=global_batch)
                if pending is not None:
                    pending.wait()
                pending = h
        if pending is not None:
            pending.wait()

    with torch.no_grad():
        run(warmup)
        if lifter is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        if lifter is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0


MIN_LEG_STEPS, LEG_REGIONS = 20, 5


def timed_regions(model, batches, batch, steps=MIN_LEG_STEPS, warmup=5, regions=LEG_REGIONS, lifter=None, global_batch=None):
    """A secondary measurement done like the headline: `warmup` untimed forwards, then `regions` back-to-back timed regions of
    `steps` (>= 20) forwards each, every region bracketed by device synchronisations; the MEDIAN region is the number (the first
    region after a model was built carries clock ramp and cold caches: it is kept, as `first_ms`)."""
    steps = max(MIN_LEG_STEPS, steps)
    timed_steps(model, batches, 0, warmup, lifter, global_batch)
    ms = [timed_steps(model, batches, steps, 0, lifter, global_batch) / steps * 1e3 for _ in range(regions)]
    first = ms[0]
    ms.sort()
    med = ms[len(ms) // 2]
    return dict(poses_per_s=round(batch / med * 1e3, 1), median_ms=round(med, 4), min_ms=round(ms[0], 4), max_ms=round(ms[-1], 4),
                first_ms=round(first, 4), regions=regions, steps_per_region=steps, warmup=warmup)


def stack_kernel_name(precision, batch, views, D, dev, launches=1, heads=8, n_apps=13):
    """Name of the kernel that runs the FPT block stack: the library reports the form it takes for this shape on this device
    (`mpl_block_stack_form`: ONE rule, csrc/h2_phase.hpp h2_stack_form + csrc/api.hip block_stack_impl; pinned by
    tests/test_h2_gpu.py::test_the_library_reports_the_form_of_a_stack_launch)."""
    import torch
    from openmpl_amd import cabi
    parts = {"fp32": 2, "bf16": 1}.get(precision, 0)
    if launches != 1 and parts:
        return cabi.FORM_KERNELS[cabi.FORM_PER_GEMM]
    with torch.cuda.device(dev):
        form = cabi.load().mpl_block_stack_form_ex(batch, views, D, heads, n_apps, max(1, n_apps - 1), 1, parts, 0)
    cabi.check(form if form < 0 else 0, "mpl_block_stack_form_ex")
    name = cabi.FORM_KERNELS[form]
    return name % parts if "%d" in name else name


def stack_roofline(model, flags, batch, batches, precision, dev, n_prof=20):
    """Roofline-shaped entry of the dominant kernel (the persistent block-stack launch) of any configuration: mean launch
    duration from HIP events on the launch stream over n_prof forwards; `frac` (= `frac_useful`) = algorithmic FLOPs x partial
    products per product / duration against the dense 16-bit matrix peak, `frac_executed` adds the padding the pipe also
    multiplies (144 / 136 column tiles; bf16: the zero k-tile that pads 17 k-tiles to 9 pairs)."""
    import torch
    from openmpl_amd import cabi
    with torch.no_grad():
        for i in range(3):
            P, R, C = batches[i % len(batches)]
            model(P, rays=R, centers=C)
        cabi.profile_start()
        for i in range(n_prof):
            P, R, C = batches[i % len(batches)]
            model(P, rays=R, centers=C)
        torch.cuda.synchronize()
        prof = cabi.profile_stop()
    gemm_ms, gemm_n = prof["gemm"]
    fl, gemms = gemm_flops_per_forward(flags, batch)
    D = fpt_width(flags)
    np_ = 1 if precision == "bf16" else 2
    kpad = (2 * (-(-(D // 32) // 2)) / (D // 32)) if np_ == 1 else 1.0
    products = {2: 3.0, 1: 1.0}[np_]
    launches = max(1, gemm_n // n_prof)
    us = gemm_ms / max(1, gemm_n) * 1e3
    useful = (fl / launches) * products / (us * 1e-6) / 1e12
    total = sum(t for t, _ in prof.values())
    return dict(kernel=stack_kernel_name(precision, batch, flags["num_views"], D, dev, launches), bound="mfma",
                launches_per_step=launches, launches_timed=gemm_n, avg_launch_us=round(us, 1), achieved=round(useful, 1),
                peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s", frac=round(useful / PEAK_BF16_MFMA_TFLOPS, 4),
                frac_useful=round(useful / PEAK_BF16_MFMA_TFLOPS, 4),
                frac_executed=round(useful * 144.0 / 136.0 * kpad / PEAK_BF16_MFMA_TFLOPS, 4),
                kernel_ms_per_step={k: round(t / n_prof, 4) for k, (t, n) in prof.items()},
                share_of_kernel_time=round(gemm_ms / total, 3) if total else None)


def time_cpu(flags, sd, P, R, C, budget_s, label):
    """Oracle (port of the reference CPU path) on the host cores: forwards of the given batch for ~budget_s seconds."""
    from oracle import mpl_oracle
    nb = P[0].shape[0]
    mpl_oracle.forward(sd, flags, P, R, C)         # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        mpl_oracle.forward(sd, flags, P, R, C)
        reps += 1
        if time.perf_counter() - t0 >= budget_s or reps >= 2000:
            break
    dt = time.perf_counter() - t0
    return dict(config=label, value=round(nb * reps / dt, 1), unit="poses/s",
                sample="%d forwards of %d poses in %.1f s" % (reps, nb, dt))


def latest_profile(name):
    """profiles/rNN_<name> of the newest round that has one."""
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + name)))
    return c[-1] if c else None


# ------------------------------------------------------------------------------------------ one rank
def run_rank(a):
    # rank 0 prints exactly ONE line on stdout, the JSON result: whatever native libraries write to file descriptor 1 meanwhile
    # (RCCL prints its version banner there when a process group is created) goes to stderr instead
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, a.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    if local >= torch.cuda.device_count():
        raise SystemExit("rank %d: --gpus %d but this node exposes %d GPU(s)" % (rank, a.gpus, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    lifter = None
    flags = model_flags(a.flagset, a.views, a.depth)
    model = build_model(flags, dev)
    model.set_matmul_precision(a.precision)
    if world > 1 or a.force_dist:
        import torch.distributed as dist
        from openmpl_amd.dist import ShardedLifter
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # backend "nccl" IS RCCL on ROCm
        lifter = ShardedLifter(model, gather=a.gather)
    split = a.precision in ("fp32", "bf16") and model._x3_supported()
    # a few distinct resident batches (per rank: pre-sharded inputs) so that no step can reuse a cached result
    batches = [make_batch(a.batch, a.views, dev, seed=1000 + rank, step=s) for s in range(4)]

    dt = timed_steps(model, batches, a.steps, a.warmup, lifter, world * a.batch)
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    value = world * a.batch * a.steps / dt

    result = None
    if rank == 0:
        result = report(a, model, flags, batches, dev, world, value, ms_per_step, split, dist is not None)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), file=json_out, flush=True)


def kptok_roofline(model, flags, batch, views, batches, n_prof=20):
    """Roofline entry of the joints x views token grid (17 V tokens of width 32): the dominant kernel is the long-sequence attention
    (token_attention_long_p4_kernel: K / V of one head LDS resident, keys in pairs), a VALU kernel.  Its roofline is the VALU issue
    floor in slots of 2 cycles (one plain wave64 instruction on a SIMD-32): per (query, key) and head the score is 4 FMAs = 2 packed
    instructions of 2 slots each = 4 slots, v_exp_f32 runs at quarter rate = 4 slots, the exponent argument, the running maximum
    and the row sum ~2.5, P.V 4 FMAs = 4 slots: ~15 slots per (query, key) -- a packed fp32 instruction does two lanes' worth of
    work and occupies the pipe twice as long, so pairing keys saves instructions, not pipe time -- on 1024 SIMDs x 2.4 GHz."""
    import torch
    from openmpl_amd import cabi
    with torch.no_grad():
        for i in range(3):
            P, R, C = batches[i % len(batches)]
            model(P, rays=R, centers=C)
        cabi.profile_start()
        for i in range(n_prof):
            P, R, C = batches[i % len(batches)]
            model(P, rays=R, centers=C)
        torch.cuda.synchronize()
        prof = cabi.profile_stop()
    N, H = 17 * views, 8
    att_ms, att_n = prof["attention"]
    us = att_ms / max(1, att_n) * 1e3
    pairs = float(batch) * H * N * N                                     # (query, key) pairs per launch
    slots = 15.0
    floor_us = pairs * slots / 64.0 * 2.0 / (1024 * 2.4e9) * 1e6         # wave64 VALU instruction = 2 cycles on a SIMD-32
    flop = pairs * 16.0                                                  # 2 x (4 score + 4 P.V) multiply-adds per pair
    return dict(bound="valu", kernel="token_attention_long_p4_kernel", launches_per_step=att_n // n_prof, launches_timed=att_n,
                avg_launch_us=round(us, 1), valu_floor_us=round(floor_us, 1), achieved=round(flop / (us * 1e-6) / 1e12, 2),
                peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(floor_us / us, 4),
                note="frac = VALU issue floor / measured launch time (15 two-cycle slots per query-key pair and head: 4 score FMAs, "
                     "v_exp_f32 at quarter rate = 4, ~2.5 for exponent argument / maximum / row sum, 4 P.V FMAs + the sum; packed "
                     "instructions count as two slots); `achieved` = attention FLOP/s against the fp32 vector peak (157.3 TFLOP/s)",
                kernel_ms_per_step={k: round(t / n_prof, 4) for k, (t, n) in prof.items()},
                share_of_kernel_time=round(att_ms / sum(t for t, _ in prof.values()), 3), traffic=None)


def kptok_report(a, model, flags, batches, dev, world, value, ms_per_step, used_dist):
    """`--flagset kptok`: the bench line of the joints x views token grid (roofline: kptok_roofline)."""
    import torch
    from oracle import mpl_oracle
    N = 17 * a.views
    roof = kptok_roofline(model, flags, a.batch, a.views, batches, max(a.steps, 20))
    P, R, C = batches[0]
    nb = min(16, a.batch)
    sub = lambda lst, n: [x[:n].contiguous() for x in lst]
    cpu = lambda lst, n: [x[:n].cpu() for x in lst]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        got = model(sub(P, nb), rays=sub(R, nb), centers=sub(C, nb)).cpu()
    ref = mpl_oracle.forward(sd, flags, cpu(P, nb), cpu(R, nb), cpu(C, nb))
    mx, nw = mpl_oracle.rel_errors(got, ref)
    return {
        "metric": "poses/sec (V=%d, J=17, batch=%d per GPU) fp32, joints x views token grid" % (a.views, a.batch),
        "value": round(value, 1), "unit": "poses/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "rccl_ranks": world if used_dist else 0,
        "config": {"workload": "large-view stress: V=%d J=17 batch=%d fp32, CHOSEN + FPT_blocks_view_keypoint_tokens (%d-token grid), depth %d"
                               % (a.views, a.batch, N, a.depth), "global_batch": world * a.batch,
                   "parallelism": "single GPU" if not used_dist else "dp%d" % world},
        "roofline": roof, "cpu_baseline": None,
        "parity": dict(max_scaled=float("%.3e" % mx), norm_wise=float("%.3e" % nw), poses=nb, tol=1e-4), "extra": {},
    }


def report(a, model, flags, batches, dev, world, value, ms_per_step, split, used_dist):
    import torch
    from openmpl_amd import cabi
    from oracle import mpl_oracle
    if a.flagset == "kptok":
        return kptok_report(a, model, flags, batches, dev, world, value, ms_per_step, used_dist)
    # ---- per-kernel time, measured live with HIP events on the launch stream: a WHOLE region of `steps` forwards run like the
    # timed one (same batches, back to back, warm), with every launch bracketed -- not a handful of forwards behind an idle gap
    P, R, C = batches[0]
    n_prof = max(a.steps, 50)
    with torch.no_grad():
        for i in range(5):
            Pw, Rw, Cw = batches[i % len(batches)]
            model(Pw, rays=Rw, centers=Cw)
        cabi.profile_start()
        for i in range(n_prof):
            Pw, Rw, Cw = batches[i % len(batches)]
            model(Pw, rays=Rw, centers=Cw)
        torch.cuda.synchronize()
        prof = cabi.profile_stop()
    gemm_ms, gemm_n = prof["gemm"]
    fl, gemms = gemm_flops_per_forward(flags, a.batch)
    # the packed-operand engines run ALL GEMMs of the block stack in ONE persistent launch (csrc/h2_phase.hpp,
    # h2_stack_kernel); the fp32-MFMA engine (and MPL_X3_LAUNCHES=1) launch one kernel per GEMM
    assert gemm_n % n_prof == 0 and gemm_n // n_prof in (1, gemms), (gemm_n, gemms)
    launches = gemm_n // n_prof
    avg_launch_ms = gemm_ms / gemm_n
    alg = (fl / launches) / (avg_launch_ms * 1e-3) / 1e12          # fp32-algorithmic TFLOP/s of the mean launch
    kernel_ms = {k: round(t / n_prof, 4) for k, (t, n) in prof.items()}
    total_flop = mpl_oracle.flop_count({k: v for k, v in flags.items()})
    io_bytes = (a.views * (17 * 2 if a.flagset == "chosen" else 17 * 6 + 3) * 4 + 51 * 4)   # 748 / 1884 B
    weight_bytes = sum(p.numel() for p in model.parameters()) * 4
    D = fpt_width(flags)
    # the kernel that ran, by the library's own rules (row tiles of (64 // V) V rows, teams of D / 136 workgroups, the
    # device's CU count): teams that own two or more row tiles run the two-tile stage (fp16x2 operands)
    gemm_kernel = stack_kernel_name(a.precision if split else "fp32_mfma", a.batch, a.views, D, dev, launches)
    pairs = gemm_kernel.startswith("h2_stack2")
    small = gemm_kernel == "sm_stack_kernel"
    if small:       # at most 32 token rows: the small-batch engine (sm_stack.hip), exact fp32 MFMA on the whole chip
        split = False
    # HBM/fabric bytes per launch cannot be counted from inside this process: they come from the committed rocprofv3
    # PMC passes of this same command (profiles/rNN_gemm_traffic.json) -- for the profiled workload shape, and ONLY while the
    # library is built from the sources the passes ran on (source hash recorded by tools/make_profiles.py): a number taken
    # with another kernel is not this kernel's traffic
    traffic, traffic_src = None, None
    try:
        from openmpl_amd import build as mpl_build
        tp = latest_profile("gemm_traffic.json")
        if tp and a.flagset == "chosen" and a.batch == 1024 and a.views == 4 and a.depth == 12:
            tj = json.load(open(tp))
            if tj.get("kernel", "").split("<")[0] == gemm_kernel.split("<")[0] and a.precision == "fp32":
                if tj.get("srchash") == mpl_build.source_hash():
                    traffic, traffic_src = round(tj["traffic_bytes_per_launch"]), os.path.relpath(tp, ROOT) + ": " + tj["source"]
                else:
                    traffic_src = "%s was taken with other kernel sources (srchash %s...): not reported" % (
                        os.path.relpath(tp, ROOT), str(tj.get("srchash"))[:12])
    except Exception:
        pass
    # algorithmic bytes of the mean GEMM launch: A + W operand + C (+ residual) once each
    M = a.batch * a.views
    np_ = 1 if a.precision == "bf16" else 2            # 16-bit parts per operand element of the packed-operand engine
    kpad = 1.0
    if split and np_ == 1:                    # bf16 engine: an odd k-tile count is padded with one zero k-tile (pairs of k-tiles)
        kpad = 2 * (-(-(D // 32) // 2)) / (D // 32)
    w_bytes = 2.0 * np_ * 144.0 / 136.0 * kpad if split else 4.0   # packed operand: np bf16 parts in 144/136-padded fragment order
    a_bytes = 2.0 * np_ * kpad if split else 4.0                   # activations travel between the GEMMs as np bf16 parts
    # fp16x2: the residual stream x (fp32) IS the operand of the LayerNorm GEMMs; bf16 hands a packed copy of x on as well
    x_copy = M * D * a_bytes if (split and np_ != 2) else 0
    alg_bytes = ((M * D * a_bytes + 3 * D * D * w_bytes + M * D * a_bytes)                 # LN1 + qkv + attention: x in, att out
                 + (M * D * a_bytes + D * D * w_bytes + 2 * M * D * 4 + x_copy)            # proj: att in, x in/out (+ packed x out)
                 + (M * D * a_bytes + 2 * D * D * w_bytes + M * 2 * D * a_bytes)           # fc1: x in, hid out
                 + (M * 2 * D * a_bytes + 2 * D * D * w_bytes + 2 * M * D * 4 + x_copy)) / 4.0   # fc2: hid in, x in/out
    if split:
        # matrix-pipe work: 3 (fp32: two fp16 parts per operand) or 1 (bf16) partial products per product = USEFUL work; the pipe
        # EXECUTES 144-column (9 x 16) tiles for 136 columns (and, bf16, the zero k-tile that pads an odd k-tile count) on top
        products = {2: 3.0, 1: 1.0}[np_]
        useful = alg * products
        ex = useful * 144.0 / 136.0 * kpad
        roof = dict(bound="mfma", kernel=gemm_kernel,
                    instruction="v_mfma_f32_16x16x32_f16" if np_ == 2 else "v_mfma_f32_16x16x32_bf16",
                    achieved=round(useful, 1), peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s",
                    frac=round(useful / PEAK_BF16_MFMA_TFLOPS, 4), frac_useful=round(useful / PEAK_BF16_MFMA_TFLOPS, 4),
                    frac_executed=round(ex / PEAK_BF16_MFMA_TFLOPS, 4), executed_tflops=round(ex, 1),
                    arithmetic=("fp32 GEMM: operands split into 2 fp16 terms under exact power-of-two scales, 3 partial products "
                                "per fp32 product on the fp16 matrix cores (same rate as bf16; 9 MFMA column tiles per 136 output "
                                "columns), fp32 accumulation -- as accurate as an fp32 GEMM (extra.max_scaled_err_vs_fp64); "
                                "`achieved` / `frac` = algorithmic fp32 FLOPs x 3 (useful fp16 MFMA FLOP/s); `frac_executed` x 144/136; "
                                if np_ == 2 else
                                "bf16 GEMM: one bf16 per operand element, fp32 accumulation; `achieved` / `frac` = algorithmic FLOPs "
                                "(useful bf16 MFMA FLOP/s); `frac_executed` x 144/136 (9 MFMA column tiles per 136 columns) x %.4f "
                                "(17 k-tiles padded to 9 pairs); " % kpad) +
                               "one launch = every GEMM of the FPT block stack (persistent row-tile chains), its duration also "
                               "contains the fused softmax attention, the operand packing and the LayerNorm statistics",
                    fp32_equivalent=dict(achieved=round(alg, 2), peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                                         frac=round(alg / PEAK_FP32_MFMA_TFLOPS, 4),
                                         note="algorithmic fp32 FLOP/s against the fp32 matrix pipe this kernel does not use"),
                    clock_note="`peak` is the nominal 2.4 GHz figure; under this launch (256 workgroups, one per CU) the part "
                               "sustains 1.95-2.22 GHz by box (GRBM_GUI_ACTIVE / duration: 2.22 in profiles/r04_pmc_summary.txt, 1.96 in r03; 2.4 GHz with up to 128 "
                               "workgroups of the same kernel), i.e. a 2.0-2.3 PFLOP/s ceiling: frac x 1.08-1.23 against it")
    else:
        roof = dict(bound="mfma", kernel=gemm_kernel, instruction="v_mfma_f32_16x16x4_f32", achieved=round(alg, 2),
                    peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(alg / PEAK_FP32_MFMA_TFLOPS, 4),
                    arithmetic="native fp32 MFMA")
    if small:
        # up to 80 token rows: 2-4 FLOP per weight byte-quarter -- the launch streams every fp32 weight of the stack once (L2 / HBM)
        # and is bounded by that and by its 65 all-to-all hand-offs, not by the matrix pipe
        gbs = alg_bytes * gemms / launches / (avg_launch_ms * 1e-3) / 1e9
        roof = dict(bound="hbm", kernel=gemm_kernel, instruction="v_mfma_f32_16x16x4_f32", achieved=round(gbs, 1), peak=PEAK_HBM_GBS,
                    unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4),
                    arithmetic="native fp32 MFMA; whole chip per GEMM (one 16-column tile per workgroup), 5 all-to-all hand-offs of {value, tag} pairs "
                               "per block application; `achieved` = algorithmic bytes (fp32 weights once + activations) / launch duration",
                    fp32_equivalent=dict(achieved=round(alg, 2), peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                                         frac=round(alg / PEAK_FP32_MFMA_TFLOPS, 4)))
    spt_ms, spt_n = prof["spt"]
    spt_fl = spt_flops_per_forward(flags, a.batch)
    spt_us = spt_ms / max(1, spt_n) * 1e3
    spt_t = spt_fl / (spt_us * 1e-6) / 1e12
    # the SPT Linear layers run from two-part fp16 operands (spt3_kernel, the arithmetic of the h2 engine) unless fp32_mfma was
    # asked for: the executed matrix work is then 3 fp16 products per product of the four Linear layers (272 = 17 x 16 rows
    # per workgroup, no padding); the 17 x 17 x hd 4 attention is VALU work and not part of `achieved`
    spt_packed = a.precision != "fp32_mfma"
    spt_lin = (flags["depth"] + 1) * 16.0 * 17 * 32 * 32 * flags["num_views"] * a.batch
    if spt_packed:
        spt_ex = spt_lin * 3.0 / (spt_us * 1e-6) / 1e12
        spt_roof = dict(kernel="spt3_kernel", instruction="v_mfma_f32_16x16x32_f16 (two-part operands, 3 products) + VALU attention",
                        achieved=round(spt_ex, 2), peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s",
                        frac=round(spt_ex / PEAK_BF16_MFMA_TFLOPS, 4), algorithmic_tflops=round(spt_t, 2))
    else:
        spt_roof = dict(kernel="spt_kernel", instruction="v_mfma_f32_16x16x4_f32 + VALU attention", achieved=round(spt_t, 2),
                        peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(spt_t / PEAK_FP32_MFMA_TFLOPS, 4))
    # second view of the same launch: the operand bytes it moves from L2 into LDS by LDS-DMA (26 KiB per stage = one k-tile of
    # one pass: A 8 KiB + W 18 KiB, fp16x2 engine) against what that path delivers on this chip (tools/h2_probe.hip,
    # profiles/r03_h2_probe.txt: 33 TB/s from an L2-resident shared region, 15 TB/s from per-workgroup regions beyond L2, 21 TB/s
    # from the shared region beside the engine's MFMA + ds_read pattern): this, not the matrix pipe, is what bounds the kernel
    lds_dma = None
    if split and np_ == 2 and launches == 1 and not pairs:
        stages = (flags["depth"] + 1) * 8 * (D // 32)            # per workgroup: (3 + 1 + 2) D/32 + 2 D/32 (fc2: K = 2 D) stages per application
        tiles = -(-M // 64)
        dma_bytes = float(stages) * 26624 * tiles * (D // 136)
        lds_dma = dict(bytes_per_launch=dma_bytes, achieved=round(dma_bytes / (avg_launch_ms * 1e-3) / 1e12, 2), unit="TB/s",
                       peak_beside_matrix_work=21.4, peak_l2_resident=33.4, peak_beyond_l2=15.0,
                       frac_of_contended_peak=round(dma_bytes / (avg_launch_ms * 1e-3) / 1e12 / 21.4, 3),
                       source="profiles/r03_h2_probe.txt")
    alg_bytes_per_gemm = alg_bytes
    alg_bytes *= gemms / launches             # per LAUNCH from here on (one launch = every GEMM of the stack by default)
    kernels = [dict(kernel=gemm_kernel, launches_per_step=launches, gemms_per_launch=gemms // launches, avg_launch_us=round(avg_launch_ms * 1e3, 2),
                    flops_per_launch=fl / launches, algorithmic_bytes_per_launch=round(alg_bytes),
                    algorithmic_bytes_per_gemm=round(alg_bytes_per_gemm),
                    share_of_kernel_time=round(gemm_ms / sum(t for t, _ in prof.values()), 3)),
               dict(spt_roof, launches_per_step=1, avg_launch_us=round(spt_us, 2), flops_per_launch=spt_fl, bound="mfma",
                    share_of_kernel_time=round(spt_ms / sum(t for t, _ in prof.values()), 3))]
    roof.update(launches_timed=gemm_n, traffic=traffic, traffic_source=traffic_src,
                traffic_ratio=(round(traffic / alg_bytes, 3) if traffic else None), lds_dma=lds_dma,
                avg_launch_us=round(avg_launch_ms * 1e3, 2),
                launches_per_step=launches, gemms_per_launch=gemms // launches, flops_per_launch=fl / launches,
                algorithmic_bytes_per_launch=round(alg_bytes), kernels=kernels,
                whole_forward_tflops=round(value / world * total_flop / 1e12, 2),
                whole_forward_frac_of_fp32_peak=round(value / world * total_flop / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                hbm_frac=round(value / world * (io_bytes + weight_bytes / a.batch) / 1e9 / PEAK_HBM_GBS, 5),
                kernel_ms_per_step=kernel_ms,
                kernel_ms_note="per-launch HIP-event brackets of a whole extra region of %d forwards run like the timed one "
                               "(every bracket adds ~1 us per launch to that region)" % n_prof)

    # ---- parity of this very run against the oracle (bounded: 64 poses)
    nb = min(64, a.batch)
    sub = lambda lst, n: [x[:n].contiguous() for x in lst]
    cpu = lambda lst, n: [x[:n].cpu() for x in lst]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        got = model(sub(P, nb), rays=sub(R, nb), centers=sub(C, nb)).cpu()
    ref = mpl_oracle.forward(sd, flags, cpu(P, nb), cpu(R, nb), cpu(C, nb))
    mx, nw = mpl_oracle.rel_errors(got, ref)
    parity = dict(max_scaled=float("%.3e" % mx), norm_wise=float("%.3e" % nw),
                  mpjpe_vs_ref=float("%.3e" % mpl_oracle.mpjpe(got, ref)), poses=nb,
                  tol=None if a.precision == "bf16" else 1e-4,     # bf16: the deviation is reported, not gated
                  note="a bounded in-run check (%d poses against the fp32 oracle); the full-size checks (B = 1024 / 8192 against the "
                       "fp64 oracle, every golden) are tests/test_gpu_parity.py, `pytest -m gpu`" % nb)

    # ---- CPU baseline: the oracle on the host cores (rank 0, N = 1 only)
    cpu_base = None
    if not a.no_cpu_baseline and world == 1:
        cores = effective_cpus()
        torch.set_num_threads(cores)
        label = "configs[1] V=%d B=%d depth %d %s fp32" % (a.views, a.batch, a.depth, a.flagset.upper())
        head = time_cpu(flags, sd, cpu(P, a.batch), cpu(R, a.batch), cpu(C, a.batch), 10.0, label)
        others = []
        if not a.no_extra:
            # configs[0] (V=2, B=1, depth 12: the CPU plumbing case), configs[1] with the other
            # flag set, configs[2]'s shape (V=8, depth 2; the CPU runs it in fp32)
            from openmpl_amd import detrng
            def case(fl_, batch, budget, lab):
                shapes = mpl_oracle.param_shapes(fl_)
                sdc = {k: torch.from_numpy(v) for k, v in detrng.make_state_dict(shapes, seed=11).items()}
                p, r, c = detrng.make_inputs(batch, fl_["num_views"], seed=1000)
                t = lambda l: [torch.from_numpy(x) for x in l]
                return time_cpu(fl_, sdc, t(p), t(r), t(c), budget, lab)
            others.append(case(model_flags("chosen", 2, 12), 1, 3.0, "configs[0] V=2 B=1 depth 12 CHOSEN fp32 (latency case)"))
            other_set = "full" if a.flagset == "chosen" else "chosen"
            others.append(case(model_flags(other_set, a.views, a.depth), a.batch, 8.0,
                               "configs[1] V=%d B=%d depth %d %s fp32" % (a.views, a.batch, a.depth, other_set.upper())))
            others.append(case(model_flags("chosen", 8, 2), 1024, 6.0, "configs[2] shape V=8 B=1024 depth 2 CHOSEN (fp32 on the CPU)"))
        cpu_base = dict(value=head["value"], unit="poses/s", cores=cores, cpu=cpu_model(), kind="port",
                        sample=head["sample"] + " (%s; torch threads = %d = cgroup CPU quota of the box), "
                                                "oracle/mpl_oracle.py" % (label, cores),
                        others=others)

    extra = {}
    if world == 1 and not used_dist:
        # the headline is ONE region of `steps` forwards (26 ms of GPU time at the defaults); the same region repeated gives the
        # number a distribution, and >= 1 s of GPU-busy time behind it
        reps = 25
        ms = sorted(timed_steps(model, batches, a.steps, 0) / a.steps * 1e3 for _ in range(reps))
        extra["repeat_ms_per_step"] = dict(regions=reps, steps_per_region=a.steps, median=round(ms[reps // 2], 4), min=round(ms[0], 4),
                                           max=round(ms[-1], 4), p10=round(ms[reps // 10], 4), p90=round(ms[reps - 1 - reps // 10], 4),
                                           median_poses_per_s=round(a.batch / ms[reps // 2] * 1e3, 1))
    if not a.no_extra and world == 1 and a.flagset == "chosen":
        extra.update(extras(a, model, flags, batches, dev, sd, got, ref, nb))
        if not used_dist:
            extra.update(dist_and_dp_extras(a, model, batches, dev))

    return {
        "metric": "poses/sec (V=%d, J=17, batch=%d per GPU) %s" % (a.views, a.batch, "bf16" if a.precision == "bf16" else "fp32"),
        "value": round(value, 1), "unit": "poses/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if a.precision == "bf16" else "f32", "data": "synthetic",
        "rccl_ranks": world if used_dist else 0,
        "config": {"workload": "%s config: V=%d J=17 batch=%d %s, %s flag set, depth %d, DIM 32, heads 8"
                               % ("CMU Panoptic" if a.views == 8 else "Human3.6M", a.views, a.batch,
                                  "bf16" if a.precision == "bf16" else "fp32", a.flagset.upper(), a.depth),
                   "global_batch": world * a.batch,
                   "parallelism": ("dp%d: pre-sharded batch, 1 RCCL all_gather of (B/G,17,3) per step (%s)" % (world, {
                       "stream": "ordered into the compute stream between two forwards", "overlap": "async, beside the next forward"}[a.gather]))
                   if used_dist else "single GPU"},
        "roofline": roof, "rooflines": rooflines(roof, extra, a), "cpu_baseline": cpu_base, "parity": parity, "extra": extra,
    }


def rooflines(roof, extra, a):
    """The roofline entries of this run side by side: the headline kernel and, from launches timed in this process, one entry
    per secondary leg that stands for a BASELINE config -- FULL, the bf16 engine at configs[2]'s shape (depth 2 and 12), V = 31 with
    view tokens and with the joints x views grid (configs[4]), the shipped call shape V = 2 B = 256, one frame (configs[0])."""
    keys = ("kernel", "bound", "avg_launch_us", "launches_timed", "achieved", "peak", "unit", "frac", "frac_useful", "frac_executed",
            "valu_floor_us")
    pick = lambda d, wl: dict({k: d[k] for k in keys if k in d}, workload=wl)
    out = [pick(roof, "V=%d B=%d depth %d %s %s" % (a.views, a.batch, a.depth, a.flagset.upper(), a.precision))]
    for key, wl in (("cmu_v8_depth2", "V=8 B=%d depth 2 CHOSEN bf16" % a.batch), ("cmu_v8_depth12", "V=8 B=%d depth 12 CHOSEN bf16" % a.batch)):
        if key in extra and "bf16_roofline" in extra[key]:
            out.append(pick(extra[key]["bf16_roofline"], wl))
        if key in extra and "fp32_roofline" in extra[key]:
            out.append(pick(extra[key]["fp32_roofline"], wl.replace("bf16", "fp32")))
    for tag, leg in extra.get("legs", {}).items():
        if "roofline" in leg:
            out.append(dict(pick(leg["roofline"], leg["workload"]), poses_per_s=leg["timing"]["poses_per_s"]))
    if "v2_b1_roofline" in extra:
        out.append(pick(extra["v2_b1_roofline"], "V=2 B=1 depth %d CHOSEN fp32 (one frame)" % a.depth))
    return out


def extras(a, model, flags, batches, dev, sd, got, ref, nb):
    """Secondary measurements (single GPU): the other fp32 engine, accuracy of all engines against fp64, the FULL flag
    set, BASELINE.json configs[2] (V=8, bf16) and configs[4] (V=31), small-batch latency, the PCIe-inclusive rate.
    Every throughput leg is timed like the headline (timed_regions: >= 20 steps per region, median of >= 5 regions) and the
    legs that stand for a BASELINE config carry the roofline entry of their dominant kernel (extra.legs, `rooflines`)."""
    import torch
    from openmpl_amd import detrng
    from openmpl_amd.multiview_mpl import MultiView_MPL
    from oracle import mpl_oracle
    extra = {}
    legs = {}
    P, R, C = batches[0]
    sub = lambda lst, n: [x[:n].contiguous() for x in lst]
    cpu = lambda lst, n: [x[:n].cpu() for x in lst]
    ref64 = mpl_oracle.forward(sd, flags, cpu(P, nb), cpu(R, nb), cpu(C, nb), dtype=torch.float64)
    e = lambda y: float("%.3e" % mpl_oracle.rel_errors(y.double(), ref64)[0])
    errs = {"hip_" + a.precision: e(got), "reference_fp32_cpu": e(ref)}
    for other in ("fp32", "fp32_mfma"):                    # the other fp32 engine: rate and distance from fp64
        if other == a.precision:
            continue
        model.set_matmul_precision(other)
        t_o = timed_regions(model, batches, a.batch)
        with torch.no_grad():
            got_o = model(sub(P, nb), rays=sub(R, nb), centers=sub(C, nb)).cpu()
        extra[other + "_poses_per_s"] = t_o["poses_per_s"]
        legs[other] = dict(workload="V=%d B=%d depth %d %s %s" % (a.views, a.batch, a.depth, a.flagset.upper(), other), timing=t_o)
        errs["hip_" + other] = e(got_o)
    model.set_matmul_precision(a.precision)
    extra["max_scaled_err_vs_fp64"] = errs
    # secondary: the FULL flag set of hm_0_...yaml (per-view SPT, conf channel, ray tokens, FPT width 1088)
    f2 = model_flags("full", a.views, a.depth)
    m2 = build_model(f2, dev)
    t2 = timed_regions(m2, batches, a.batch)
    extra["full_flagset_poses_per_s"] = t2["poses_per_s"]
    extra["full_flagset_tflops"] = round(t2["poses_per_s"] * mpl_oracle.flop_count(f2) / 1e12, 2)
    legs["full_v%d" % a.views] = dict(workload="V=%d B=%d depth %d FULL fp32" % (a.views, a.batch, a.depth), timing=t2,
                                      roofline=stack_roofline(m2, f2, a.batch, batches, "fp32", dev))
    del m2
    # BASELINE.json configs[2]: CMU Panoptic shape V=8, batch 1024, bf16 matrix cores (yaml depth 2), with the fp32 run
    # of the same shape and the bf16 deviation from the fp32 reference semantics (reported, not gated)
    f3 = model_flags("chosen", 8, 2)
    m3 = build_model(f3, dev)
    b3 = [make_batch(a.batch, 8, dev, seed=2000, step=s) for s in range(2)]
    t32 = timed_regions(m3, b3, a.batch)
    m3.set_matmul_precision("bf16")
    t16 = timed_regions(m3, b3, a.batch)
    v32, v16 = t32["poses_per_s"], t16["poses_per_s"]
    P3, R3, C3 = b3[0]
    with torch.no_grad():
        o16 = m3(sub(P3, 64), rays=sub(R3, 64), centers=sub(C3, 64)).cpu()
    sd3 = {k: v.detach().cpu() for k, v in m3.state_dict().items()}
    r3 = mpl_oracle.forward(sd3, f3, cpu(P3, 64), cpu(R3, 64), cpu(C3, 64))
    mx3, nw3 = mpl_oracle.rel_errors(o16, r3)
    roof16 = stack_roofline(m3, f3, a.batch, b3, "bf16", dev)
    extra["cmu_v8_depth2"] = {"fp32_poses_per_s": v32, "bf16_poses_per_s": v16, "bf16_ms_per_step": t16["median_ms"],
                              "bf16_timing": t16, "fp32_timing": t32, "bf16_roofline": roof16,
                              "bf16_tflops": round(v16 * mpl_oracle.flop_count(f3) / 1e12, 1),
                              "bf16_frac_of_bf16_peak": round(v16 * mpl_oracle.flop_count(f3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
                              "bf16_max_scaled_vs_ref": float("%.3e" % mx3), "bf16_norm_wise_vs_ref": float("%.3e" % nw3),
                              "bf16_mpjpe_vs_ref": float("%.3e" % mpl_oracle.mpjpe(o16, r3))}
    del m3
    # the same shape at depth 12 (config 3 at L = 2 and L = 12)
    f3d = model_flags("chosen", 8, 12)
    m3d = build_model(f3d, dev)
    t32d = timed_regions(m3d, b3, a.batch)
    roof32d = stack_roofline(m3d, f3d, a.batch, b3, "fp32", dev)
    m3d.set_matmul_precision("bf16")
    t16d = timed_regions(m3d, b3, a.batch)
    extra["cmu_v8_depth12"] = {"fp32_poses_per_s": t32d["poses_per_s"], "bf16_poses_per_s": t16d["poses_per_s"],
                               "bf16_ms_per_step": t16d["median_ms"], "bf16_timing": t16d, "fp32_timing": t32d,
                               "fp32_roofline": roof32d,
                               "bf16_roofline": stack_roofline(m3d, f3d, a.batch, b3, "bf16", dev)}
    del m3d
    # PCIe-inclusive rate of the headline workload: the same forwards fed from pinned host tensors, H2D
    # of the V x (B,17,3) poses (+ rays, centers: the API's 1680 B/pose) inside the timed region
    host = [tuple([t.cpu().pin_memory() for t in lst] for lst in b) for b in batches[:2]]
    n_o = MIN_LEG_STEPS

    def host_fed(step_fn):
        with torch.no_grad():
            for i in range(3):
                step_fn(i)
            ms = []
            for _ in range(LEG_REGIONS):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n_o):
                    step_fn(i)
                torch.cuda.synchronize()
                ms.append((time.perf_counter() - t0) / n_o * 1e3)
        ms.sort()
        return round(a.batch / ms[len(ms) // 2] * 1e3, 1)

    def h2d_step(i):
        Ph, Rh, Ch = host[i % 2]
        up = lambda lst: [t.to(dev, non_blocking=True) for t in lst]
        return model(up(Ph), rays=up(Rh), centers=up(Ch))
    extra["h2d_inclusive_poses_per_s"] = host_fed(h2d_step)
    # the same through ONE pinned staging buffer and one copy per batch (openmpl_amd.inputs.HostStager)
    from openmpl_amd.inputs import HostStager
    stagers = [HostStager(a.batch, a.views, 17, dev) for _ in range(2)]       # double-buffered: batch i+1 is packed while i runs
    plain = [tuple([t.cpu() for t in lst] for lst in b) for b in batches[:2]]

    def staged_step(i):
        Pv, Rv, Cv = stagers[i % 2].stage(*plain[i % 2])
        return model(Pv, rays=Rv, centers=Cv)
    extra["h2d_inclusive_staged_poses_per_s"] = host_fed(staged_step)
    # BASELINE.json configs[0] shape on the GPU: single-frame latency (V=2, B=1, depth 12), synchronised per call
    f0 = model_flags("chosen", 2, a.depth)
    m0 = build_model(f0, dev)
    b0 = [make_batch(1, 2, dev, seed=4000, step=s) for s in range(2)]

    def per_call_us(n=50, regions=LEG_REGIONS):
        us = []
        for _ in range(regions):
            t0 = time.perf_counter()
            for i in range(n):
                m0(b0[i % 2][0], rays=b0[i % 2][1], centers=b0[i % 2][2])
                torch.cuda.synchronize()
            us.append((time.perf_counter() - t0) / n * 1e6)
        us.sort()
        return round(us[len(us) // 2], 1)
    with torch.no_grad():
        for i in range(5):
            m0(b0[i % 2][0], rays=b0[i % 2][1], centers=b0[i % 2][2])
        torch.cuda.synchronize()
        extra["v2_b1_latency_us"] = per_call_us()
        m0.use_torch_op(True)
        extra["v2_b1_latency_torch_op_us"] = per_call_us()
        m0.use_torch_op("auto")
        extra["v2_b1_back_to_back"] = timed_regions(m0, b0, 1)
        # roofline of the small-batch engine (sm_stack_kernel: every GEMM on the whole chip, the fp32 nn.Linear weights streamed
        # once per forward): algorithmic bytes = the FPT weights of the depth + 1 block applications, against the HBM peak
        from openmpl_amd import cabi
        cabi.profile_start()
        for i in range(20):
            m0(b0[i % 2][0], rays=b0[i % 2][1], centers=b0[i % 2][2])
        torch.cuda.synchronize()
        pr0 = cabi.profile_stop()
        us0 = pr0["gemm"][0] / max(1, pr0["gemm"][1]) * 1e3
        w0 = (a.depth + 1) * 8.0 * 544 * 544 * 4
        extra["v2_b1_roofline"] = dict(kernel=stack_kernel_name("fp32", 1, 2, 544, dev), bound="hbm", launches_timed=pr0["gemm"][1],
                                       avg_launch_us=round(us0, 1), algorithmic_bytes_per_launch=round(w0),
                                       achieved=round(w0 / (us0 * 1e-6) / 1e9, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                                       frac=round(w0 / (us0 * 1e-6) / 1e9 / PEAK_HBM_GBS, 4),
                                       kernel_ms_per_step={k: round(t / 20, 4) for k, (t, n) in pr0.items()},
                                       note="65 all-to-all hand-offs (latency), not bandwidth, bound this launch (DESIGN.md section 4)")
        # the same frame through the team kernels (the small-batch engine of sm_stack.hip switched off: mpl_x3_stack_mode bit 3)
        try:
            cabi.check(cabi.load().mpl_x3_stack_mode(8), "stack mode")
            for i in range(5):
                m0(b0[i % 2][0], rays=b0[i % 2][1], centers=b0[i % 2][2])
            torch.cuda.synchronize()
            extra["v2_b1_latency_team_kernels_us"] = per_call_us()
        finally:
            cabi.check(cabi.load().mpl_x3_stack_mode(0), "stack mode")
    del m0
    # batch curve (the reference's shipped call shape is TEST.BATCH_SIZE 256 with two views, configs/h36m/mpl_amass/h36m.yaml:107,
    # :37-39): whole forward, CHOSEN flag set, depth 12.  The block stack is a chain of 52 dependent GEMMs per 64-row tile that
    # one team of D / 136 workgroups walks serially, so its time (~1 ms) does not shrink with the batch: small batches leave
    # CUs idle ("small batches")
    curve, curve_t = {}, {}
    for Vc in (2, 4):
        fc = model_flags("chosen", Vc, a.depth)
        mc = build_model(fc, dev)
        for Bc in (1, 32, 256, 1024):
            bc = [make_batch(Bc, Vc, dev, seed=5000 + Bc, step=s) for s in range(2)]
            tc = timed_regions(mc, bc, Bc)
            curve["v%d_b%d" % (Vc, Bc)] = tc["poses_per_s"]
            curve_t["v%d_b%d" % (Vc, Bc)] = tc
            if Vc == 2 and Bc == 256:          # the reference's shipped call shape: its own roofline entry (row-narrow teams)
                legs["v2_b256"] = dict(workload="V=2 B=256 depth %d CHOSEN fp32 (the reference's shipped TEST.BATCH_SIZE)" % a.depth,
                                       timing=tc, roofline=stack_roofline(mc, fc, Bc, bc, "fp32", dev))
        del mc
    extra["batch_curve_poses_per_s"] = curve
    extra["batch_curve_timing"] = curve_t
    # BASELINE.json configs[4]: large-view stress V=31, batch 256 (31-token FPT), and the 17V = 527-token joints x views
    # grid (KPTOK, LDS-resident K/V of one head)
    for tag, fl in (("v31_b256_chosen", {}), ("v31_b256_kptok", dict(FPT_blocks_view_keypoint_tokens=True))):
        f5 = model_flags("chosen", 31, a.depth, **fl)
        m5 = build_model(f5, dev)
        b5 = [make_batch(256, 31, dev, seed=3000, step=s) for s in range(2)]
        t5 = timed_regions(m5, b5, 256)
        extra[tag + "_poses_per_s"] = t5["poses_per_s"]
        extra[tag + "_tflops"] = round(t5["poses_per_s"] * mpl_oracle.flop_count(f5) / 1e12, 1)
        roof5 = kptok_roofline(m5, f5, 256, 31, b5) if fl else stack_roofline(m5, f5, 256, b5, "fp32", dev)
        legs[tag] = dict(workload="V=31 B=256 depth %d CHOSEN%s fp32" % (a.depth, " + joints x views token grid" if fl else ""),
                         timing=t5, roofline=roof5)
        del m5
    # the non-default tails at the reference's default width (TRANSFORMER_OUTPUT_HEAD_HIDDEN_DIM = 1024, config.py:98):
    # step time relative to the default head on the same batches
    heads = {}
    base = timed_regions(model, batches, a.batch)["median_ms"]
    for tag, fl in (("deep_head", dict(deep_head=True, hidden_dim=1024)), ("head_kadkhod", dict(head_kadkhod=True, hidden_dim=1024)),
                    ("linear_weighted_mean", dict(linear_weighted_mean=True))):
        mh = build_model(model_flags("chosen", a.views, a.depth, **fl), dev)
        t = timed_regions(mh, batches, a.batch)["median_ms"]
        heads[tag] = dict(ms_per_step=round(t, 4), vs_default_head=round(t / base, 3))
        del mh
    heads["default_head_ms_per_step"] = round(base, 4)
    extra["heads_hidden_dim_1024"] = heads
    extra["legs"] = legs
    extra["legs_note"] = ("every leg: %d warm-up forwards, then the MEDIAN of %d regions of >= %d forwards (timed_regions); `first_ms` is "
                          "the first region after the model was built" % (5, LEG_REGIONS, MIN_LEG_STEPS))
    return extra


def dist_and_dp_extras(a, model, batches, dev):
    """The two multi-GPU mechanisms on ONE GPU: the rank path (process group of one rank, the RCCL all-gather issued every
    step: what --force-dist measures) and the reference's own torch.nn.DataParallel wrapper (valid_mpl.py:177-178) fed CPU
    tensors, over device 0 alone and over every visible GPU."""
    import torch
    import torch.distributed as dist
    from openmpl_amd.dist import ShardedLifter
    out = {}
    n = max(MIN_LEG_STEPS, a.steps)
    try:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        # direct | collective ordered into the compute stream (default) | collective on the process group's stream beside the
        # next forward (rounds 1-5), alternating in ONE process: the price of the rank path with one rank
        modes = {"direct": None, "stream": ShardedLifter(model, gather="stream"), "overlap": ShardedLifter(model, gather="overlap")}
        runs = {k: [] for k in modes}
        for rep in range(3):
            for k, lf in modes.items():
                runs[k].append(timed_regions(model, batches, a.batch, n, 3, LEG_REGIONS, lf, a.batch if lf is not None else None))
        best = {k: sorted(r, key=lambda t: t["median_ms"])[len(r) // 2] for k, r in runs.items()}
        out["force_dist_poses_per_s"] = best[a.gather]["poses_per_s"]
        out["force_dist"] = dict(gather_default=a.gather, direct=best["direct"], stream=best["stream"], overlap=best["overlap"],
                                 stream_vs_direct=round(best["stream"]["median_ms"] / best["direct"]["median_ms"], 4),
                                 overlap_vs_direct=round(best["overlap"]["median_ms"] / best["direct"]["median_ms"], 4),
                                 note="three alternations of {direct, stream, overlap}, each the median of %d regions of %d steps; "
                                      "a process group of ONE rank (RCCL all_gather_into_tensor of (B,17,3) every step)" % (LEG_REGIONS, n))
        dist.destroy_process_group()
    except Exception as e:            # a box without a working RCCL: say so, the headline does not depend on it
        out["force_dist_poses_per_s"] = "failed: %r" % (e,)
    host = [tuple([t.cpu() for t in lst] for lst in b) for b in batches[:2]]
    ids = {"device0": [0], "all_visible_gpus": list(range(torch.cuda.device_count()))}
    dp_out = {}
    for tag, dev_ids in ids.items():
        dp = torch.nn.DataParallel(model, device_ids=dev_ids).eval()
        with torch.no_grad():
            for i in range(3):
                dp(host[i % 2][0], rays=host[i % 2][1], centers=host[i % 2][2])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                dp(host[i % 2][0], rays=host[i % 2][1], centers=host[i % 2][2])
            torch.cuda.synchronize()
        dp_out[tag] = dict(gpus=len(dev_ids), steps=n, poses_per_s=round(a.batch * n / (time.perf_counter() - t0), 1))
    dp_out["note"] = ("torch.nn.DataParallel(model) called with CPU tensors as validate() does (function_mpl.py:350): scatter copies "
                      "the inputs, replicas share the per-device packed operands (packed once, not per forward)")
    out["data_parallel"] = dp_out
    return out


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus))
    run_rank(a)


if __name__ == "__main__":
    main()
