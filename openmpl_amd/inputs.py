"""Device-side replacement of the per-sample input preparation of OpenMPL's datasets.

Reference: lib/dataset/joints_dataset_mpl.py:615-623, :646, :762-774, :817-820, :872-904.  From raw detections
(B,V,J,2) [+ confidences (B,V,J)] and one calibration per view it produces exactly what the model call
`model(input, centers=centers, rays=rays)` (function_mpl.py:350) takes.  One HIP kernel through the C ABI; no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import cabi


def pack_cameras(cameras, device) -> torch.Tensor:
    """cameras: sequence of dicts with fx, fy, cx, cy, R (3x3 world->camera), t (camera centre, 3) -> device (V,16) f64."""
    rows = [np.concatenate([[float(c["fx"]), float(c["fy"]), float(c["cx"]), float(c["cy"])],
                            np.asarray(c["R"], np.float64).reshape(-1), np.asarray(c["t"], np.float64).reshape(-1)])
            for c in cameras]
    return torch.from_numpy(np.stack(rows)).to(device)


def prepare_inputs(joints_px: torch.Tensor, conf: Optional[torch.Tensor], cams: torch.Tensor, image_size: Tuple[float, float],
                   normalize_inputs: bool = True, normalize_cameras: bool = True
                   ) -> Tuple[List[torch.Tensor], List[torch.Tensor], List[torch.Tensor]]:
    """joints_px (B,V,J,2) float32 GPU, conf (B,V,J) or None, cams (V,16) float64 GPU (pack_cameras).
    Returns (poses, rays, centers): lists of V tensors (B,J,3), (B,J,3), (B,1,3)."""
    if joints_px.device.type != "cuda":
        raise RuntimeError("prepare_inputs has no CPU path: tensors must live on a GPU")
    if joints_px.ndim != 4 or joints_px.shape[-1] != 2 or joints_px.dtype != torch.float32:
        raise RuntimeError("joints_px must be float32 (B,V,J,2)")
    B, V, J, _ = joints_px.shape
    if tuple(cams.shape) != (V, 16) or cams.dtype != torch.float64 or cams.device != joints_px.device:
        raise RuntimeError("cams must be float64 (V,16) on the same device (see pack_cameras)")
    dev = joints_px.device
    lib = cabi.load()
    joints_px = joints_px.contiguous()
    if conf is not None:
        conf = conf.reshape(B, V, J).to(torch.float32).contiguous()
    mk = lambda *s: [torch.empty(s, dtype=torch.float32, device=dev) for _ in range(V)]
    poses, rays, centers = mk(B, J, 3), mk(B, J, 3), mk(B, 1, 3)
    arr = lambda lst: (cabi._fp * V)(*[t.data_ptr() for t in lst])
    with torch.cuda.device(dev):
        rc = lib.mpl_prepare_inputs(joints_px.data_ptr(), None if conf is None else conf.data_ptr(), cams.data_ptr(), B, V, J,
                                    float(image_size[0]), float(image_size[1]), int(normalize_inputs), int(normalize_cameras),
                                    arr(poses), arr(rays), arr(centers), torch.cuda.current_stream().cuda_stream)
    cabi.check(rc, "mpl_prepare_inputs")
    return poses, rays, centers


class HostStager:
    """One pinned staging buffer + ONE host-to-device copy per batch for callers that hold the inputs on the host (the
    loop of validate(), function_mpl.py:334-351, hands `model(input, centers=, rays=)` CPU tensors): the V x {(b,J,3) poses,
    (b,J,3) rays, (b,1,3) centers} are packed back to back into pinned memory and cross PCIe as one transfer instead of
    3V small ones; the returned lists are views into one device buffer (layouts exactly those the model expects).

    `batch` is the LARGEST batch: a shorter one (the ragged last batch of a loader without drop_last) is staged into a
    prefix.  The returned views ALIAS the stager's device buffer, which the next stage() overwrites: consume them on the
    stream that was current when stage() ran (the copy is enqueued there, so a forward launched on it afterwards is ordered
    behind the copy and ahead of the next one), or use two stagers alternately as bench.py does."""

    def __init__(self, batch: int, views: int, joints: int, device):
        self.shape = (batch, views, joints)
        n = views * batch * (2 * joints * 3 + 3)
        self.host = torch.empty(n, dtype=torch.float32).pin_memory()
        self.dev = torch.empty(n, dtype=torch.float32, device=device)
        self._copied = None          # event behind the last host-to-device copy out of self.host

    def stage(self, poses, rays, centers):
        B, V, J = self.shape
        if self._copied is not None:
            self._copied.synchronize()      # the previous transfer has read the pinned buffer before it is overwritten
        b = poses[0].shape[0] if len(poses) else 0
        if len(poses) != V or len(rays) != V or len(centers) != V or not 0 < b <= B:
            raise RuntimeError("HostStager was built for %d views of at most %d poses" % (V, B))
        off, views = 0, ([], [], [])
        for k, (lst, shp) in enumerate(((poses, (b, J, 3)), (rays, (b, J, 3)), (centers, (b, 1, 3)))):
            n = shp[0] * shp[1] * shp[2]
            for t in lst:
                if tuple(t.shape) != shp:
                    raise RuntimeError("HostStager: expected a tensor of shape %s, got %s" % (shp, tuple(t.shape)))
                self.host[off:off + n].view(shp).copy_(t)
                views[k].append(self.dev[off:off + n].view(shp))
                off += n
        self.dev[:off].copy_(self.host[:off], non_blocking=True)
        self._copied = torch.cuda.Event()
        self._copied.record(torch.cuda.current_stream(self.dev.device))
        return views
