"""Output-side epilogue: oracle pinned on reference-generated goldens; HIP kernel vs both."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as mo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ("pjpe_abs", "mpjpe_abs", "pjpe_rel", "mpjpe_rel", "dist", "dist_mean")


@pytest.mark.parametrize("tag", ["a", "b"])
def test_metrics_oracle_matches_reference_golden(tag):
    g = np.load(os.path.join(GOLD, "metrics_%s.npz" % tag))
    m = mo.all_metrics(g["out"], g["tgt"], None, g["scale"], g["offset"])
    for k in KEYS:
        np.testing.assert_allclose(m[k], g[k], rtol=1e-6, atol=1e-7)
    if "loss" in g.files:
        np.testing.assert_allclose(m["loss"], g["loss"], rtol=1e-6)
        np.testing.assert_allclose(m["loss_axis"], g["loss_axis"], rtol=1e-6)
        np.testing.assert_allclose(mo.mpjpe_loss(g["out"], g["tgt"], g["w"])[0], g["loss_weighted"], rtol=1e-6)
    # evaluate.py:101-104, :110-113: not_consider_kp (duplicates and negative indices included in fixture b)
    m2 = mo.all_metrics(g["out"], g["tgt"], None, g["scale"], g["offset"], not_consider_kp=g["nck"].tolist())
    np.testing.assert_allclose(m2["mpjpe_abs"], g["mpjpe_abs_nck"], rtol=1e-6)
    np.testing.assert_allclose(m2["mpjpe_rel"], g["mpjpe_rel_nck"], rtol=1e-6)
    np.testing.assert_allclose(m2["pjpe_abs"], g["pjpe_abs"], rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_pose_metrics_kernel_matches_reference_golden(tag):
    from openmpl_amd.metrics import pose_metrics
    g = np.load(os.path.join(GOLD, "metrics_%s.npz" % tag))
    out, tgt = torch.from_numpy(g["out"]).cuda(), torch.from_numpy(g["tgt"]).cuda()
    m = pose_metrics(out, tgt, scale=g["scale"], offset=g["offset"])
    for k in KEYS:
        np.testing.assert_allclose(m[k].cpu().numpy(), g[k], rtol=2e-5, atol=1e-6, err_msg=k)
    if "loss" in g.files:
        np.testing.assert_allclose(float(m["loss"]), g["loss"], rtol=2e-5)
        np.testing.assert_allclose(m["loss_axis"].cpu().numpy(), g["loss_axis"], rtol=2e-5)
        mw = pose_metrics(out, tgt, weight=torch.from_numpy(g["w"]).cuda())
        np.testing.assert_allclose(float(mw["loss"]), g["loss_weighted"], rtol=2e-5)
    m2 = pose_metrics(out, tgt, scale=g["scale"], offset=g["offset"], not_consider_kp=g["nck"].tolist())
    np.testing.assert_allclose(float(m2["mpjpe_abs"]), g["mpjpe_abs_nck"], rtol=2e-5)
    np.testing.assert_allclose(float(m2["mpjpe_rel"]), g["mpjpe_rel_nck"], rtol=2e-5)
    np.testing.assert_allclose(m2["pjpe_abs"].cpu().numpy(), g["pjpe_abs"], rtol=2e-5, atol=1e-6)
    with pytest.raises(IndexError):
        pose_metrics(out, tgt, not_consider_kp=[17])


@pytest.mark.gpu
def test_pose_metrics_large_batch_against_oracle_and_loud_errors():
    from openmpl_amd.metrics import pose_metrics
    rs = np.random.RandomState(0)
    out = rs.randn(8192, 17, 3).astype(np.float32)
    tgt = (out + 0.1 * rs.randn(8192, 17, 3)).astype(np.float32)
    m = pose_metrics(torch.from_numpy(out).cuda(), torch.from_numpy(tgt).cuda(), scale=(2.0, 3.0, 0.5))
    ref = mo.all_metrics(out.astype(np.float64), tgt.astype(np.float64), None, (2.0, 3.0, 0.5))
    for k in KEYS + ("loss", "loss_axis"):
        np.testing.assert_allclose(m[k].cpu().numpy(), ref[k], rtol=1e-5, err_msg=k)
    with pytest.raises(RuntimeError, match="no CPU path"):
        pose_metrics(torch.from_numpy(out), torch.from_numpy(tgt))
