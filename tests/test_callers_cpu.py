"""CPU-only pins of the caller-side rows of SURVEY 8f against values the REFERENCE's own modules produced
(tools/gen_golden.py: gen_render_utils, gen_loss, gen_contrast):
  f2  n3dt.render_utils.RenderUtils ray grid / intrinsics scaling / orbit + base cameras
      == Utils/RenderUtils.py:31-107 (imported and run by the generator)
  f3  n3dt.train.data_losses == HeadNeRFLossUtils.calc_total_loss(use_vgg_loss=False) (Utils/HeadNeRFLossUtils.py:125-236)
  and the oracle on the high-contrast fixture (saturating alpha, O(10) features).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, synthetic_case
from oracle import oracle as orc


@pytest.mark.parametrize("fs,views", [(32, 45), (64, 7), (8, 5)])
def test_render_utils_values_match_the_reference(fs, views):
    from n3dt import BaseOptions
    from n3dt.render_utils import RenderUtils
    g, _ = load_golden("render_utils")
    opt = BaseOptions({"featmap_size": fs, "featmap_nc": 256, "pred_img_size": fs * 4, "num_sample_coarse": 8})
    ru = RenderUtils(views, torch.device("cpu"), opt, inv_inmat=g["inv_inmat_32"])
    k = "fs%d_v%d." % (fs, views)
    assert np.array_equal(ru.ray_xy.numpy(), g[k + "ray_xy"])
    assert np.array_equal(ru.ray_uv.numpy(), g[k + "ray_uv"])
    np.testing.assert_allclose(ru.inv_inmat.numpy(), g[k + "inv_inmat"], rtol=1e-7, atol=0)
    # cameras: the reference computes in float64 numpy and stores float32; so do we (torch float64)
    np.testing.assert_allclose(ru.Rmats.numpy(), g[k + "Rmats"], atol=1e-7)
    np.testing.assert_allclose(ru.Tvecs.numpy(), g[k + "Tvecs"], atol=1e-6)
    assert np.array_equal(ru.base_cam_info["batch_Rmats"].numpy(), g[k + "base_R"])
    assert np.array_equal(ru.base_cam_info["batch_Tvecs"].numpy(), g[k + "base_T"])
    assert len(ru.cam_info_list) == views
    for i in (0, views - 1):
        assert torch.equal(ru.cam_info_list[i]["batch_Rmats"][0], ru.Rmats[i])
    # without a file the synthetic intrinsics are the same matrix (inv_intrinsics(fs) == inv_intrinsics(32) rescaled)
    ru2 = RenderUtils(views, torch.device("cpu"), opt)
    np.testing.assert_allclose(ru2.inv_inmat.numpy(), g[k + "inv_inmat"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_python_loss_terms_match_the_reference(case):
    from n3dt.train import data_losses
    g, m = load_golden("loss")
    info = [c for c in m["cases"] if c["name"] == case][0]
    k = case + "."
    merge = torch.from_numpy(g[k + "merge_img"]).requires_grad_(True)
    bg = torch.from_numpy(g[k + "bg_img"]).requires_grad_(True)
    t = data_losses({"merge_img": merge, "bg_img": bg}, torch.from_numpy(g[k + "gt"]), torch.from_numpy(g[k + "mask"]),
                    bg_value=1.0 if info["bg_type"] == "white" else 0.0)
    total = t["bg_loss"] + t["head_loss"] + t["nonhead_loss"]
    got = np.array([t["bg_loss"].item(), t["head_loss"].item(), t["nonhead_loss"].item(), total.item()])
    np.testing.assert_allclose(got, g[k + "terms"], rtol=1e-6)
    total.backward()
    np.testing.assert_allclose(merge.grad.numpy(), g[k + "d_merge"], atol=1e-9, rtol=1e-5)
    np.testing.assert_allclose(bg.grad.numpy(), g[k + "d_bg"], atol=1e-9, rtol=1e-5)


def test_oracle_on_the_contrast_fixture():
    """alpha saturates on a third of the rays, features reach 12: the oracle against the reference's outputs."""
    g, m = load_golden("contrast")
    opt, sd, inp = synthetic_case(m)  # checks the weight checksum
    assert m["stats"]["frac_rays_bg_alpha_lt_0.01"] > 0.3 and m["stats"]["feat_abs_max"] > 10
    out = orc.forward(sd, opt, inp)
    step = int(g["ray_index_step"])
    np.testing.assert_allclose(out["fg_feat"][:, :, ::step], g["fg_feat"], atol=2e-3)
    # the x400 density head amplifies fp32 rounding differences between two fp32 evaluations (torch's GEMM order
    # vs the oracle's): 1e-6 on the raw head becomes 4e-4 on sigma; measured 2e-4 on bg_alpha
    np.testing.assert_allclose(out["bg_alpha"], g["bg_alpha"], atol=1e-3)
    assert np.abs(out["merge_img"] - g["merge_img_q16"].astype(np.float32) / 65535.0).max() <= 1e-3


def test_fitting_camera_parametrisation():
    """n3dt.fitting: Euler -> rotation (Rz Ry Rx, FittingSingleImage_new.py:736-766) against scipy, its autograd against finite
    differences, and the camera composition R = dR R0, T = dR T0 + dT (:797-803)."""
    from scipy.spatial.transform import Rotation
    from n3dt import fitting
    g = torch.Generator().manual_seed(3)
    ang = (torch.rand(5, 3, generator=g, dtype=torch.float64) - 0.5) * 2.0
    R = fitting.eulurangle2Rmat(ang)
    ref = Rotation.from_euler("xyz", ang.numpy()).as_matrix()  # extrinsic x, y, z = Rz Ry Rx
    np.testing.assert_allclose(R.numpy(), ref, atol=1e-12)
    assert torch.autograd.gradcheck(fitting.eulurangle2Rmat, (ang.clone().requires_grad_(True),), eps=1e-6, atol=1e-6)
    base_shape, base_appea = torch.zeros(1, 179), torch.zeros(1, 127)
    cam = {"batch_Rmats": torch.diag(torch.tensor([1.0, -1.0, -1.0])).view(1, 3, 3), "batch_Tvecs": torch.tensor([[[0.0], [0.0], [12.0]]]),
           "batch_inv_inmats": torch.eye(3).view(1, 3, 3)}
    st = fitting.FittingState(base_shape, base_appea, cam)
    code, c0 = st.build_code_and_cam()
    assert torch.equal(c0["batch_Rmats"], cam["batch_Rmats"]) and torch.equal(c0["batch_Tvecs"], cam["batch_Tvecs"])
    assert code["shape_code"].shape == (1, 179) and code["bg_code"] is None
    with torch.no_grad():
        st.delta_EulurAngles.copy_(torch.tensor([[0.0, 0.3, 0.0]]))
        st.delta_Tvecs.copy_(torch.tensor([[[0.1], [0.0], [0.0]]]))
    _, c1 = st.build_code_and_cam()
    dR = fitting.eulurangle2Rmat(st.delta_EulurAngles)
    np.testing.assert_allclose(c1["batch_Tvecs"].detach().numpy(), (dR.detach() @ cam["batch_Tvecs"] + st.delta_Tvecs.detach()).numpy(), atol=1e-7)
    opt, sched = st.make_optimizer()
    assert [g_["lr"] for g_ in opt.param_groups] == [0.015, 0.015, 0.01, 0.001, 0.001]
