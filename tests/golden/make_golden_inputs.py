"""Golden vectors for on-device input preparation from the REFERENCE's own methods
(build container only): JointsDataset.normalize_screen_coordinates (:817-820) and create_3d_ray_coords (:872-904) of
lib/dataset/joints_dataset_mpl.py, plus the camera normalisation of __getitem__ (:615-623), loaded in place with a
stub `cv2`.  python tests/golden/make_golden_inputs.py"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from openmpl_amd import detrng  # noqa: E402

LIB = "/root/reference/MPL/lib"
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, LIB)
spec = importlib.util.spec_from_file_location("_ref_joints_dataset", os.path.join(LIB, "dataset", "joints_dataset_mpl.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
DS = mod.JointsDataset_MPL


def rot(seed):
    a = detrng.normal(seed, "rot", (3, 3), 0, 1).astype(np.float64)
    q, r = np.linalg.qr(a)
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


for tag, (w, h), norm_cam in (("h36m", (1000, 1000), True), ("cmu", (1920, 1080), True), ("raw", (1000, 1000), False)):
    B, V, J = 5, 3, 17
    fake = types.SimpleNamespace(downsample=1, use_grid=False, use_t=True, bug_test=False, image_size=[w, h])
    px = np.stack([detrng.uniform(9, "px.%s.%d" % (tag, v), (B, J, 2), 0.0, 1.0) * np.array([w, h]) for v in range(V)], 1)
    conf = np.stack([detrng.uniform(9, "cf.%s.%d" % (tag, v), (B, J), 0.0, 1.0) for v in range(V)], 1)
    cams, poses, rays, cens = [], np.zeros((V, B, J, 3), np.float32), np.zeros((V, B, J, 3), np.float32), np.zeros((V, B, 1, 3), np.float32)
    for v in range(V):
        cam = dict(fx=1100.0 + 37 * v, fy=1120.0 - 11 * v, cx=w / 2 + 13.0 * v, cy=h / 2 - 7.0 * v, R=rot(100 + v),
                   t=detrng.normal(9, "t.%d" % v, (3, 1), 0, 3).astype(np.float64))
        cams.append(np.concatenate([[cam["fx"], cam["fy"], cam["cx"], cam["cy"]], cam["R"].reshape(-1), cam["t"].reshape(-1)]))
        camera = dict(cam)
        if norm_cam:                       # __getitem__ :615-623 (INPUTS_NORMALIZED and NORMALIZE_CAMERAS)
            cc = DS.normalize_screen_coordinates(fake, np.array([camera["cx"], camera["cy"]]), w, h)
            camera["cx"], camera["cy"] = cc[0], cc[1]
            fl = np.array([camera["fx"], camera["fy"]]) / w * 2
            camera["fx"], camera["fy"] = fl[0], fl[1]
        for b in range(B):
            joints = px[b, v].astype(np.float64).copy()
            if tag != "raw":
                joints = DS.normalize_screen_coordinates(fake, joints, w, h)        # :764
            joints_ds = joints / fake.downsample                                    # :768
            ray = DS.create_3d_ray_coords(fake, camera, None, joints_ds=joints_ds)  # :769 (torch float tensor)
            poses[v, b, :, :2] = joints.astype(np.float32)
            poses[v, b, :, 2] = conf[b, v]
            rays[v, b] = ray.numpy()
            cens[v, b, 0] = cam["t"].reshape(-1).astype(np.float32)                  # :646 cam_center = T.T
    np.savez_compressed(os.path.join(HERE, "inputs_%s.npz" % tag), px=px.astype(np.float32), conf=conf.astype(np.float32),
                        cams=np.stack(cams).astype(np.float64), wh=np.array([w, h], np.float32),
                        normalize=np.array([tag != "raw", norm_cam]), poses=poses, rays=rays, centers=cens)
    print(tag, poses[0, 0, 0], rays[0, 0, 0], cens[0, 0, 0])
