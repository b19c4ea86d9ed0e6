"""Generate tests/golden/*.npz by running the REFERENCE model (build container only).

    python tests/golden/make_golden.py [case ...]

For each case in ``cases.py``: build ``MultiView_MPL(**flags)`` from
/root/reference (via oracle/ref_import.py -- loaded in place, never copied), fill
every parameter from ``openmpl_amd.detrng`` (seeded, regenerable), run it on
``detrng.make_inputs`` and store inputs, output and three intermediate taps:

  spt_view0 : output of Spatial_forward_features for view 0   (multiview_mpl.py:412)
  fpt_in    : the (B,V,D_f) token tensor entering forward_features (:495-499)
  fused     : the (B,1,544) output of forward_features          (:446)

Fixtures are data only (inputs + expected outputs + flags); no reference source.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from openmpl_amd import detrng            # noqa: E402
from oracle import ref_import             # noqa: E402
from tests.golden.cases import CASES, MICRO  # noqa: E402


def run_case(case, store_weights=False):
    flags = case["flags"]
    m = ref_import.build_reference(dict(flags, drop_path_rate=0.1))
    detrng.fill_module_(m, seed=case["wseed"])
    V, B = flags["num_views"], case["batch"]
    poses, rays, centers = detrng.make_inputs(B, V, flags["num_joints"], seed=case["iseed"])
    taps = {}
    first = []

    def spt_hook(_mod, _inp, out):
        if not first:
            first.append(1)
            taps["spt_view0"] = out.detach().clone()

    h = m.Spatial_norm.register_forward_hook(spt_hook)
    orig_ff = m.forward_features

    def ff(xs):
        taps["fpt_in"] = xs.detach().clone()
        y = orig_ff(xs)
        taps["fused"] = y.detach().clone()
        return y

    m.forward_features = ff
    with torch.no_grad():
        out = m([torch.from_numpy(p.copy()) for p in poses],
                rays=[torch.from_numpy(r) for r in rays],
                centers=[torch.from_numpy(c) for c in centers])
    h.remove()
    rec = dict(
        flags=np.frombuffer(json.dumps(flags, sort_keys=True).encode(), dtype=np.uint8),
        meta=np.frombuffer(json.dumps(dict(batch=B, wseed=case["wseed"], iseed=case["iseed"],
                                           torch=torch.__version__)).encode(), dtype=np.uint8),
        poses=np.stack(poses, 0), rays=np.stack(rays, 0), centers=np.stack(centers, 0),
    )
    if isinstance(out, tuple):
        rec["out"] = out[0].numpy()
        rec["out_x1"] = out[1][0].numpy()
        rec["out_x2"] = out[1][1].numpy()
    else:
        rec["out"] = out.numpy()
    for k, v in taps.items():
        rec["tap_" + k] = v.numpy()
    if store_weights:
        for k, v in m.state_dict().items():
            rec["w:" + k] = v.numpy()
    path = os.path.join(HERE, case["name"] + ".npz")
    np.savez_compressed(path, **rec)
    print("%-32s out|max|=%.4f  %d kB" % (case["name"], float(np.abs(rec["out"]).max()),
                                          os.path.getsize(path) // 1024))


def main(argv):
    want = set(argv)
    for case in CASES:
        if not want or case["name"] in want:
            run_case(case)
    if not want or MICRO["name"] in want:
        run_case(MICRO, store_weights=True)


if __name__ == "__main__":
    main(sys.argv[1:])
