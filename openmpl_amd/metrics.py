"""Device-side replacement of the host epilogue of OpenMPL's validate(): room de-normalisation + MPJPE reductions.

Reference: lib/core/function_mpl.py:474-494 (`output.clone().cpu().numpy()` then numpy), lib/core/evaluate.py:91-125,
lib/core/loss.py:39-57, :110-124.  One tiny HIP kernel through the C ABI (mpl_pose_metrics); no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import torch

from . import cabi


def pose_metrics(output: torch.Tensor, target: torch.Tensor, weight: Optional[torch.Tensor] = None,
                 scale: Optional[Sequence[float]] = None, offset: Optional[Sequence[float]] = None,
                 not_consider_kp: Optional[Sequence[int]] = None) -> Dict[str, torch.Tensor]:
    """output/target (B,J,3) float32 on the GPU; weight (B,J) or (B,J,1) optional (Weighted_MPJPE);
    scale/offset: per-axis room de-normalisation (function_mpl.py:476-488); not_consider_kp: joints deleted from the means
    mpjpe_abs / mpjpe_rel (config.NOT_CONSIDER_SOME_KP_IN_EVAL, evaluate.py:101-104, np.delete semantics).  Returns device tensors:
    loss, loss_axis(3), pjpe_abs(J), mpjpe_abs, pjpe_rel(J), mpjpe_rel, dist(J,3), dist_mean(3)."""
    if output.device.type != "cuda" or target.device != output.device:
        raise RuntimeError("pose_metrics has no CPU path: tensors must live on the same GPU")
    if output.shape != target.shape or output.ndim != 3 or output.shape[2] != 3:
        raise RuntimeError("expected (B,J,3) tensors of equal shape")
    if output.dtype != torch.float32 or target.dtype != torch.float32:
        raise RuntimeError("float32 tensors required")
    B, J, _ = output.shape
    lib = cabi.load()
    output, target = output.contiguous(), target.contiguous()
    if weight is not None:
        weight = weight.reshape(B, J).to(torch.float32).contiguous()
    n = lib.mpl_pose_metrics_size(J)
    res = torch.empty(n, dtype=torch.float32, device=output.device)
    sc = (C.c_float * 3)(*[float(v) for v in scale]) if scale is not None else None
    of = (C.c_float * 3)(*[float(v) for v in offset]) if offset is not None else None
    mask = 0
    for k in (not_consider_kp if not_consider_kp is not None else ()):
        if not -J <= int(k) < J:
            raise IndexError("index %d is out of bounds for axis 0 with size %d" % (int(k), J))      # what np.delete raises
        mask |= 1 << (int(k) % J)
    with torch.cuda.device(output.device):
        rc = lib.mpl_pose_metrics_ex(output.data_ptr(), target.data_ptr(), None if weight is None else weight.data_ptr(), B, J,
                                     sc, of, mask, res.data_ptr(), torch.cuda.current_stream().cuda_stream)
    cabi.check(rc, "mpl_pose_metrics")
    # NaN poses of a forward that lost a hand-off would be SKIPPED by the nansum / nanmean semantics of the reference's metrics
    # and score as zero error: the C entry refuses while the device's error word is set, the kernel writes NaN results when
    # the failing forward is still in flight on this stream, and a failure that has already been reported raises here
    cabi.raise_if_device_error(output.device.index)
    o = 4
    return dict(loss=res[0], loss_axis=res[1:4], pjpe_abs=res[o:o + J], mpjpe_abs=res[o + J],
                pjpe_rel=res[o + J + 1:o + 2 * J + 1], mpjpe_rel=res[o + 2 * J + 1],
                dist=res[o + 2 * J + 2:o + 2 * J + 2 + 3 * J].reshape(J, 3), dist_mean=res[o + 2 * J + 2 + 3 * J:])
