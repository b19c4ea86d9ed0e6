"""CPU oracle for the output-side epilogue: room de-normalisation and MPJPE-family
reductions that the reference runs on the host after every forward.

TEST INFRASTRUCTURE ONLY (same rules as mpl_oracle.py).  numpy restatement, each function citing the reference
lines it follows; pinned by tests/golden/metrics_*.npz, produced by running the reference's own functions
(tests/golden/make_golden_metrics.py).
"""
import numpy as np


def denormalise(x, scale, offset):
    """function_mpl.py:476-488 -- `preds * room_scale + room_center` (equal scaling, :480-481) or per-axis
    `preds[:, :, a] *= room_a_scale` (:485-488, offset 0); applied to predictions and ground truth alike."""
    return x * np.asarray(scale, dtype=x.dtype).reshape(1, 1, 3) + np.asarray(offset, dtype=x.dtype).reshape(1, 1, 3)


def calc_mpjpe(output, target, mode="absolute", not_consider_kp=None):
    """evaluate.py:91-114.  NaN terms are skipped inside the squared sum (np.nansum); joints listed in not_consider_kp
    (config.NOT_CONSIDER_SOME_KP_IN_EVAL) are left out of the mean over joints, np.delete semantics (:101-104, :110-113)."""
    if mode == "relative":                                   # :105-107 root-relative
        output = output - output[:, 0:1, :]
        target = target - target[:, 0:1, :]
    pjpe = np.sqrt(np.nansum((output - target) ** 2, axis=2)).mean(axis=0)     # :100 / :108
    if not_consider_kp is not None:
        J = pjpe.shape[0]
        keep = np.ones(J, dtype=bool)
        keep[[int(k) % J for k in not_consider_kp]] = False  # np.delete: duplicates count once, negative indices wrap
        return pjpe, pjpe[keep].mean()
    return pjpe, pjpe.mean()


def calc_distance_per_dim(output, target):
    """evaluate.py:117-125.  Per joint and axis mean |error| over the non-NaN samples (np.nanmean)."""
    d = np.nanmean(np.abs(output - target), axis=0)
    return d, d.mean(axis=0)


def mpjpe_loss(output, target, w=None):
    """loss.py:39-57 (MPJPE) and :110-124 (Weighted_MPJPE, w of shape (B,J) after squeeze)."""
    err = np.linalg.norm(output - target, axis=2)
    axis = [np.mean(np.abs(output[:, :, a] - target[:, :, a])) for a in range(3)]   # :50-52 / :120-122
    if w is not None:
        err = w.reshape(err.shape) * err
    return err.mean(), axis


def all_metrics(output, target, w=None, scale=(1, 1, 1), offset=(0, 0, 0), not_consider_kp=None):
    """Everything mpl_pose_metrics returns, in its result order."""
    loss, axis = mpjpe_loss(output, target, w)
    o, t = denormalise(output, scale, offset), denormalise(target, scale, offset)
    pa, ma = calc_mpjpe(o, t, "absolute", not_consider_kp)
    pr, mr = calc_mpjpe(o, t, "relative", not_consider_kp)
    d, dm = calc_distance_per_dim(o, t)
    return dict(loss=np.float64(loss), loss_axis=np.array(axis), pjpe_abs=pa, mpjpe_abs=ma, pjpe_rel=pr, mpjpe_rel=mr,
                dist=d, dist_mean=dm)
