"""Golden vectors for the metric epilogue from the REFERENCE's own functions (build container only):
lib/core/evaluate.py (calc_mpjpe, calc_distance_per_dim) and lib/core/loss.py (MPJPE, Weighted_MPJPE), loaded in place
with a stub for `core.inference` (which drags in cv2).  python tests/golden/make_golden_metrics.py"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from openmpl_amd import detrng  # noqa: E402

REF = "/root/reference/MPL/lib/core"


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


core = types.ModuleType("core")
inf = types.ModuleType("core.inference")
inf.get_max_preds = lambda *a, **k: None
sys.modules["core"], sys.modules["core.inference"] = core, inf
ev = load("_ref_evaluate", os.path.join(REF, "evaluate.py"))
ls = load("_ref_loss", os.path.join(REF, "loss.py"))

for tag, B, with_nan in (("a", 37, False), ("b", 300, True)):
    out = detrng.normal(5, "m.out." + tag, (B, 17, 3), 0.0, 0.7)
    tgt = out + detrng.normal(5, "m.err." + tag, (B, 17, 3), 0.0, 0.05)
    w = detrng.uniform(5, "m.w." + tag, (B, 17, 1), 0.0, 1.0)
    scale = np.array([1.7, 1.7, 1.7] if tag == "a" else [2.0, 3.5, 1.0], dtype=np.float32)
    offset = np.array([0.3, -0.2, 1.1] if tag == "a" else [0.0, 0.0, 0.0], dtype=np.float32)
    if with_nan:   # missing ground-truth joints: the reference's nansum / nanmean paths
        mask = detrng.uniform(5, "m.nan", (B, 17, 1), 0, 1) < 0.03
        tgt = np.where(mask, np.float32(np.nan), tgt).astype(np.float32)
    # reference pipeline: loss on the raw tensors (function_mpl.py:394), metrics on the de-normalised copies (:476-488)
    rec = dict(out=out, tgt=tgt, w=w, scale=scale, offset=offset)
    if not with_nan:
        l, ax = ls.MPJPE()(torch.from_numpy(out), torch.from_numpy(tgt))
        lw, _ = ls.Weighted_MPJPE()(torch.from_numpy(out), torch.from_numpy(tgt), torch.from_numpy(w))
        rec.update(loss=np.float64(l), loss_axis=np.array([float(a) for a in ax]), loss_weighted=np.float64(lw))
    o = out * scale + offset
    t = tgt * scale + offset
    pa, ma = ev.calc_mpjpe(o, t, mode="absolute")
    pr, mr = ev.calc_mpjpe(o, t, mode="relative")
    d, dm = ev.calc_distance_per_dim(o, t)
    rec.update(pjpe_abs=pa, mpjpe_abs=ma, pjpe_rel=pr, mpjpe_rel=mr, dist=d, dist_mean=dm)
    # evaluate.py:101-104 / :110-113: joints left out of the MEAN (config.NOT_CONSIDER_SOME_KP_IN_EVAL), per-joint errors unchanged
    nck = [0, 9, 10] if tag == "a" else [16, 3, 3, -2]
    pa2, ma2 = ev.calc_mpjpe(o, t, mode="absolute", not_consider_kp=nck)
    pr2, mr2 = ev.calc_mpjpe(o, t, mode="relative", not_consider_kp=nck)
    assert np.array_equal(pa2, pa) and np.array_equal(pr2, pr)
    rec.update(nck=np.array(nck), mpjpe_abs_nck=ma2, mpjpe_rel_nck=mr2)
    np.savez_compressed(os.path.join(HERE, "metrics_%s.npz" % tag), **rec)
    print(tag, float(ma), float(mr), dm)
