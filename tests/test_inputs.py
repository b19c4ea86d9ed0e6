"""On-device input preparation: oracle pinned on reference-generated goldens; HIP kernel vs both."""
import os

import numpy as np
import pytest
import torch

from oracle import inputs_oracle as io

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TAGS = ["h36m", "cmu", "raw"]


def _load(tag):
    g = np.load(os.path.join(GOLD, "inputs_%s.npz" % tag))
    return g, float(g["wh"][0]), float(g["wh"][1]), bool(g["normalize"][0]), bool(g["normalize"][1])


@pytest.mark.parametrize("tag", TAGS)
def test_inputs_oracle_matches_reference_golden(tag):
    g, w, h, ni, nc = _load(tag)
    p, r, c = io.prepare_inputs(g["px"], g["conf"], g["cams"], w, h, ni, nc)
    np.testing.assert_allclose(p, g["poses"], rtol=2e-7, atol=2e-7)
    np.testing.assert_allclose(r, g["rays"], rtol=3e-7, atol=5e-7)
    np.testing.assert_array_equal(c, g["centers"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_prepare_inputs_kernel_matches_reference_golden(tag):
    from openmpl_amd.inputs import prepare_inputs
    g, w, h, ni, nc = _load(tag)
    cams = torch.from_numpy(g["cams"]).cuda()
    p, r, c = prepare_inputs(torch.from_numpy(g["px"]).cuda(), torch.from_numpy(g["conf"]).cuda(), cams, (w, h), ni, nc)
    for v in range(g["px"].shape[1]):
        np.testing.assert_allclose(p[v].cpu().numpy(), g["poses"][v], rtol=2e-7, atol=2e-7)
        np.testing.assert_allclose(r[v].cpu().numpy(), g["rays"][v], rtol=3e-7, atol=5e-7)
        np.testing.assert_array_equal(c[v].cpu().numpy(), g["centers"][v])


@pytest.mark.gpu
def test_prepared_inputs_feed_the_model_and_loud_errors():
    """raw detections -> prepare_inputs -> MultiView_MPL (FULL flags use rays and centers) == oracle end to end."""
    from openmpl_amd import detrng
    from openmpl_amd.inputs import pack_cameras, prepare_inputs
    from openmpl_amd.multiview_mpl import MultiView_MPL
    from oracle import mpl_oracle
    g, w, h, ni, nc = _load("h36m")
    flags = dict(num_joints=17, embed_dim_ratio=32, num_heads=8, depth=2, num_views=3, pose_3d_emb_learnable=True,
                 confidence_input_as_third=True, input_rays_as_token=True, multiple_spatial_blocks=True,
                 add_3D_pos_encoding_to_rays=True)
    m = MultiView_MPL(**flags)
    detrng.fill_module_(m, seed=5)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    cams = pack_cameras([dict(fx=c[0], fy=c[1], cx=c[2], cy=c[3], R=c[4:13].reshape(3, 3), t=c[13:16]) for c in g["cams"]], "cuda")
    p, r, c = prepare_inputs(torch.from_numpy(g["px"]).cuda(), torch.from_numpy(g["conf"]).cuda(), cams, (w, h), ni, nc)
    with torch.no_grad():
        out = m(p, rays=r, centers=c)
    op, orr, oc = io.prepare_inputs(g["px"], g["conf"], g["cams"], w, h, ni, nc)
    ref = mpl_oracle.forward(sd, flags, [torch.from_numpy(x) for x in op], [torch.from_numpy(x) for x in orr],
                             [torch.from_numpy(x) for x in oc])
    mx, nw = mpl_oracle.rel_errors(out.cpu(), ref)
    assert mx < 1e-4 and nw < 1e-4
    with pytest.raises(RuntimeError, match="no CPU path"):
        prepare_inputs(torch.from_numpy(g["px"]), None, cams, (w, h))


@pytest.mark.gpu
def test_host_stager_reuse_and_short_last_batch():
    """HostStager (validate()'s host-tensor call shape, function_mpl.py:334-351): one pinned buffer + one copy per batch, reused
    across batches, a ragged last batch staged into a prefix; the forward on the staged views equals the forward on plain
    device copies bit for bit."""
    from openmpl_amd import detrng
    from openmpl_amd.inputs import HostStager
    from openmpl_amd.multiview_mpl import MultiView_MPL
    dev = torch.device("cuda:0")
    flags = dict(num_joints=17, embed_dim_ratio=32, num_heads=8, depth=2, num_views=4, pose_3d_emb_learnable=True)
    m = MultiView_MPL(**flags)
    detrng.fill_module_(m, seed=4)
    m = m.to(dev).eval()
    st = HostStager(48, 4, 17, dev)
    for B, seed in ((48, 1), (48, 2), (7, 3), (48, 4)):
        p, r, c = detrng.make_inputs(B, 4, seed=seed)
        P, R, Cn = ([torch.from_numpy(x) for x in lst] for lst in (p, r, c))
        with torch.no_grad():
            Pv, Rv, Cv = st.stage(P, R, Cn)
            assert Pv[0].shape == (B, 17, 3) and Cv[3].shape == (B, 1, 3) and Pv[0].device == dev
            got = m(Pv, rays=Rv, centers=Cv)
            want = m([x.to(dev) for x in P], rays=[x.to(dev) for x in R], centers=[x.to(dev) for x in Cn])
        assert torch.equal(got, want)
    with pytest.raises(RuntimeError):
        big = detrng.make_inputs(49, 4, seed=5)
        st.stage(*([torch.from_numpy(x) for x in lst] for lst in big))
