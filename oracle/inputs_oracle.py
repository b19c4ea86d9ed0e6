"""CPU oracle for on-device input preparation.  TEST INFRASTRUCTURE ONLY.

numpy (float64, cast to float32 at the end like the reference's `.float()`) restatement of what
lib/dataset/joints_dataset_mpl.py does per sample and view before the model sees anything:
screen normalisation :817-820, camera normalisation :615-623, ray construction :872-904, cam_center :646,
input concat :772.  Pinned by tests/golden/inputs_*.npz (made by running those reference methods).
"""
import numpy as np


def prepare_inputs(px, conf, cams, w, h, normalize_inputs=True, normalize_cameras=True):
    """px (B,V,J,2) pixel joints, conf (B,V,J), cams (V,16) = [fx,fy,cx,cy,R(9 row-major),t(3)].
    Returns poses, rays (V,B,J,3) and centers (V,B,1,3) float32."""
    px = px.astype(np.float64)
    B, V, J, _ = px.shape
    poses = np.zeros((V, B, J, 3), np.float32)
    rays = np.zeros((V, B, J, 3), np.float32)
    cens = np.zeros((V, B, 1, 3), np.float32)
    for v in range(V):
        fx, fy, cx, cy = cams[v, :4]
        R = cams[v, 4:13].reshape(3, 3)
        t = cams[v, 13:16]
        j = px[:, v]
        if normalize_inputs:                                   # :817-820 (X/w)*2 - [1, h/w]
            j = (j / w) * 2 - np.array([1.0, h / w])
        if normalize_inputs and normalize_cameras:             # :615-623
            cx, cy = (cx / w) * 2 - 1.0, (cy / w) * 2 - h / w
            fx, fy = fx / w * 2, fy / w * 2
        cam = np.stack([(j[..., 0] - cx) / fx, (j[..., 1] - cy) / fy, np.ones(j.shape[:-1])], -1)   # :883-898
        world = cam @ R + t                                    # (R.T @ c.T + t).T == c @ R + t
        poses[v, :, :, :2] = j
        poses[v, :, :, 2] = conf[:, v]                         # :772
        rays[v] = world
        cens[v, :, 0] = t                                      # :646
    return poses, rays, cens
