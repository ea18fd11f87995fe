"""Round-2 parity additions (through the C ABI, `-m gpu`): the high-contrast fixture in every precision, the alpha -> 1
edge through the stand-alone compositing operator AND the fused epilogues, the caller-side rows (orbit sweep, loss tail)
against the reference's own outputs, the kernel variants selectable at run time, packed-weight cache invalidation, and
BASELINE config 3 at its full size."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, synthetic_case, options_from_manifest
from test_gpu_parity import dev, to_dev, build_net, fwd, feats, RGB_TOL

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# RGB L-inf on the high-contrast fixture, measured (tools/contrast_probe.py; CPU emulation of the roundings in DESIGN 4):
#   fp32 7.8e-5 | fp16 3.7e-3 | bf16 2.4e-2.  Only the fp32 mode holds the 1e-3 north-star gate on a network this sharp:
#   a x400 density head turns the 2^-9 (bf16) / 2^-12 (fp16) relative rounding of weights AND activations into alpha
#   errors of up to 0.07 / 0.007 at the samples where a ray saturates.  The bounds below are what each mode is held to.
#   The split-operand mode (bf16x3: every product as three bf16 MFMAs on hi / lo halves, nerf_fwd_x16s.hip) holds the gate: 8e-5.
CONTRAST_RGB = {"fp32": 2e-4, "fp16": 6e-3, "bf16": 4e-2, "bf16x3": 1e-3}   # bf16x3: the north-star gate itself (measured 8e-5)
CONTRAST_WEIGHT = {"fp32": 2e-3, "fp16": 1.5e-2, "bf16": 0.12, "bf16x3": 2e-3}
CONTRAST_FEAT = {"fp32": 5e-3, "fp16": 0.15, "bf16": 0.7, "bf16x3": 5e-3}   # features reach 12.7


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16", "bf16x3"])
def test_contrast_fixture(precision):
    """Sharp densities (alpha saturates on 36 % of the rays, stays under 0.5 on 48 %), features up to 12.7 -- the regime
    round 1's seed-0 fixtures did not reach -- against the reference's outputs."""
    g, m = load_golden("contrast")
    opt, sd, inp = synthetic_case(m)
    net = build_net(opt, sd, precision)
    d = to_dev(inp)
    f = feats(net, d, want_weight=True)
    step = int(g["ray_index_step"])
    out = fwd(net, d)
    e_w = np.abs(f["weight"].cpu().numpy()[:, None][:, :, ::step] - g["weight"]).max()
    e_a = np.abs(f["bg_alpha"].cpu().numpy()[:, None] - g["bg_alpha"]).max()
    e_f = np.abs(f["fg_feat"].permute(0, 2, 1).cpu().numpy()[:, :, ::step] - g["fg_feat"]).max()
    e_rgb = np.abs(out["merge_img"].cpu().numpy() - g["merge_img_q16"].astype(np.float32) / 65535.0)
    print("contrast %s: weight %.2e bg_alpha %.2e fg_feat %.2e RGB max %.2e mean %.2e" % (precision, e_w, e_a, e_f, e_rgb.max(), e_rgb.mean()))
    assert e_w <= CONTRAST_WEIGHT[precision] and e_a <= CONTRAST_WEIGHT[precision] and e_f <= CONTRAST_FEAT[precision]
    assert e_rgb.max() <= CONTRAST_RGB[precision]
    assert e_rgb.mean() <= 0.1 * CONTRAST_RGB[precision]  # the large errors sit on the few saturating rays
    assert np.abs(out["bg_img"].cpu().numpy() - g["bg_img_q16"].astype(np.float32) / 65535.0).max() <= RGB_TOL.get(precision, 5e-4)


@pytest.mark.parametrize("name", ["tiny_test", "tiny_train", "cfg1", "cfg2r", "hr", "cfg4"])
def test_split_precision_mode_on_the_reference_fixtures(name):
    """precision="bf16x3" (N3DT_BF16X3: bf16 MFMA, every operand split hi + lo, three products; the 2-D renderer on its fp16
    path) against every reference fixture: the volumetric stage at the exact-fp32 kernel's tolerances (features 5e-5), RGB at
    2e-4 (the fp16 renderer's share; the gate is 1e-3)."""
    from n3dt import synthetic as syn
    g, m = load_golden(name)
    opt, sd, inp = synthetic_case(m)
    net = build_net(opt, sd, "bf16x3")
    d = to_dev(inp)
    t_rand = None
    if m.get("mode") == "train":
        t_rand = syn.stratified_noise(m["batch"], opt.featmap_size ** 2, opt.num_sample_coarse, m["t_rand_seed"]).to(dev())
    f = feats(net, d, t_rand, want_weight=True)
    step = int(g["ray_index_step"]) if "ray_index_step" in g else 1
    np.testing.assert_allclose(f["fg_feat"].permute(0, 2, 1).cpu().numpy()[:, :, ::step], g["fg_feat"], atol=5e-5)
    np.testing.assert_allclose(f["bg_alpha"].cpu().numpy()[:, None], g["bg_alpha"], atol=5e-5)
    if "weight" in g:
        np.testing.assert_allclose(f["weight"].cpu().numpy()[:, None], g["weight"], atol=2e-5)
    out = fwd(net, d, m.get("mode", "test"), t_rand)
    img = out["merge_img"].cpu().numpy()
    if "merge_img" in g:
        ref = g["merge_img"]
    elif "merge_img_q16" in g:
        ref = g["merge_img_q16"].astype(np.float32) / 65535.0
    else:
        c0, cs = int(g["crop_origin"]), g["merge_img_crop_q16"].shape[-1]
        img, ref = img[:, :, c0:c0 + cs, c0:c0 + cs], g["merge_img_crop_q16"].astype(np.float32) / 65535.0
    err = np.abs(img - ref).max()
    print("bf16x3 %s: RGB max|err| %.2e" % (name, err))
    assert err <= 2e-4


def test_saturated_alpha_through_the_compositing_operator():
    """SURVEY Q6 on the GPU: the reference's own alpha -> 1 vectors (edges.npz sat.*: densities 1e4 and 3e38, an all-zero
    ray) through n3dt_composite -- the transmittance after a saturated sample is 1e-10-ish, not 0."""
    from n3dt import ops
    g, _ = load_golden("edges")
    t = lambda k: torch.from_numpy(g[k]).to(dev())  # noqa: E731
    feat, ba, dp, w = ops.composite(t("sat.rgb"), t("sat.density"), t("sat.dists"), t("sat.zvals"))
    torch.cuda.synchronize()
    np.testing.assert_allclose(w.cpu().numpy(), g["sat.weight"], atol=1e-7, rtol=1e-5)
    np.testing.assert_allclose(feat.cpu().numpy(), g["sat.fg_feat"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(ba.cpu().numpy(), g["sat.bg_alpha"], atol=1e-6)
    np.testing.assert_allclose(dp.cpu().numpy(), g["sat.depth"], atol=1e-4, rtol=1e-5)
    wn = w.cpu().numpy()
    assert np.all(np.isfinite(wn))
    # ray 0 saturates at sample 2: the samples behind it keep weight alpha * 1e-10 * T, not 0
    assert 0.0 < wn[0, 0, 0, 3] < 1e-9 and wn[0, 0, 0, 2] > 0.1


def _oracle_seams(sd, opt, inp):
    """fg_feat [B,C,Nr], bg_alpha [B,1,Nr], weight [B,1,Nr,Ns] of the CPU oracle, seam by seam."""
    from oracle import oracle as orc
    n = lambda t: t.detach().cpu().numpy()  # noqa: E731
    s = orc.sample(n(inp["batch_xy"]), n(inp["batch_Rmats"]), n(inp["batch_Tvecs"]), n(inp["batch_inv_inmats"]),
                   opt.num_sample_coarse, opt.world_z1, opt.world_z2, None)
    rgb, dens = orc.mlp(sd, orc.embed(s["pts"]), n(inp["shape_code"]), n(inp["appea_code"]), n(inp["audiostyle"]))
    fg, ba, _, w = orc.composite(rgb, dens, s["z_dists"], s["zvals"])
    return {"fg_feat": fg, "bg_alpha": ba, "weight": w}


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16"])
def test_saturated_alpha_through_the_fused_epilogue(precision):
    """The same edge through the FUSED kernels' epilogue (density -> alpha -> exclusive transmittance scan -> per-block
    partials -> ray head): a density-head bias of 1e4 makes alpha == 1 exactly at every sample, so the weights must be
    1, 1e-10, 1e-20, ... (the reference's `1 - alpha + 1e-10`, NetWorks/utils.py:284-285) -- against the CPU oracle, which
    tests/test_oracle_golden.py pins to the reference's sat.* vectors."""
    from n3dt import BaseOptions, synthetic as syn
    opt = BaseOptions({"featmap_size": 8, "featmap_nc": 256, "pred_img_size": 32, "num_sample_coarse": 40})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    sd["fg_CD_predictor.density_module.bias"] = torch.full((1,), 1.0e4)
    inp = syn.frame_inputs(opt, 2)
    ref = _oracle_seams(sd, opt, inp)
    f = feats(build_net(opt, sd, precision), to_dev(inp), want_weight=True)
    w = f["weight"].cpu().numpy()
    assert np.all(w[:, :, 0] == 1.0)
    np.testing.assert_allclose(w[:, :, 1], 1e-10, rtol=1e-5)
    np.testing.assert_allclose(w[:, :, 2], 1e-20, rtol=1e-5)
    np.testing.assert_allclose(w, ref["weight"][:, 0], rtol=1e-4, atol=1e-37)
    assert np.all(f["bg_alpha"].cpu().numpy() == ref["bg_alpha"][:, 0])  # 1 - (1 + 1e-10 + ...) == 0 in fp32
    tol = {"fp32": 2e-5, "bf16": 5e-3, "fp16": 1e-3}[precision]
    np.testing.assert_allclose(f["fg_feat"].permute(0, 2, 1).cpu().numpy(), ref["fg_feat"], atol=tol)
    # and a partially saturated case: bias 0 on frame 0's scale, huge gain -> alpha in {0, ~1} per sample
    sd2 = syn.contrast_state_dict(opt, density_gain=4000.0, density_bias=-600.0, feat_gain=1.0)
    ref2 = _oracle_seams(sd2, opt, inp)
    f2 = feats(build_net(opt, sd2, "fp32"), to_dev(inp), want_weight=True)
    np.testing.assert_allclose(f2["weight"].cpu().numpy(), ref2["weight"][:, 0], atol=5e-3)
    assert float(ref2["weight"].max()) > 0.99


def test_novel_view_and_morph_sweeps_against_the_reference():
    """the reference's own RenderUtils.render_novel_views (5 serial forwards) and render_morphing_res (4) with
    the audio-less net, as uint8 frames (tools/gen_golden.py: gen_render_utils), against ONE batched launch each here.
    `(img * 255).astype(uint8)` truncates, so an fp32-rounding difference flips a value by one step at most."""
    from n3dt import BaseOptions, synthetic as syn
    from n3dt.render_utils import RenderUtils
    g, m = load_golden("render_utils")
    opt = options_from_manifest(m)
    sdn = syn.make_state_dict(opt, seed=4, audio_dim=0, bg_noise=0.1)
    assert np.allclose(syn.state_dict_checksum(sdn), m["weights_checksum_noaudio"], rtol=1e-9, atol=1e-6)
    net = build_net(opt, sdn, "fp32", audio_dim=0)
    ru = RenderUtils(5, dev(), opt, inv_inmat=g["inv_inmat_32"], audio_dim=0)
    sh, ap, _ = syn.latents(2, 179, 127, 0)
    c1 = {"bg_code": None, "shape_code": sh[0:1].to(dev()), "appea_code": ap[0:1].to(dev())}
    c2 = {"bg_code": None, "shape_code": sh[1:2].to(dev()), "appea_code": ap[1:2].to(dev())}
    views = np.stack(ru.render_novel_views(net, c1))
    morph = np.stack(ru.render_morphing_res(net, c1, c2, 4))
    for got, ref in ((views, g["sweep.novel_views_u8"]), (morph, g["sweep.morph_u8"])):
        assert got.shape == ref.shape and got.dtype == np.uint8
        diff = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        assert diff.max() <= 1 and (diff != 0).mean() < 0.02, (diff.max(), (diff != 0).mean())


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_fused_loss_tail_against_the_reference(case):
    """n3dt_loss_fwd / n3dt_loss_bwd against HeadNeRFLossUtils.calc_total_loss(use_vgg_loss=False) run by the
    generator: NaNs in merge_img, mask values on both sides of (and exactly at) 0.5, white and black backgrounds."""
    from n3dt.train import fused_data_losses
    g, m = load_golden("loss")
    info = [c for c in m["cases"] if c["name"] == case][0]
    k = case + "."
    merge = torch.from_numpy(g[k + "merge_img"]).to(dev()).requires_grad_(True)
    bg = torch.from_numpy(g[k + "bg_img"]).to(dev()).requires_grad_(True)
    t = fused_data_losses({"merge_img": merge, "bg_img": bg}, torch.from_numpy(g[k + "gt"]).to(dev()),
                          torch.from_numpy(g[k + "mask"]).to(dev()), bg_value=1.0 if info["bg_type"] == "white" else 0.0)
    total = t["bg_loss"] + t["head_loss"] + t["nonhead_loss"]
    got = np.array([t["bg_loss"].item(), t["head_loss"].item(), t["nonhead_loss"].item(), total.item()])
    np.testing.assert_allclose(got, g[k + "terms"], rtol=2e-6)
    total.backward()
    np.testing.assert_allclose(merge.grad.cpu().numpy(), g[k + "d_merge"], atol=1e-9, rtol=1e-5)
    np.testing.assert_allclose(bg.grad.cpu().numpy(), g[k + "d_bg"], atol=1e-9, rtol=1e-5)
    # the same through the reference-shaped object (same call, same keys incl. the reference's `nonhaed_loss`)
    from n3dt.train import HeadNeRFLossUtils
    lu = HeadNeRFLossUtils(bg_type=info["bg_type"], use_vgg_loss=False)
    res = lu.calc_total_loss(None, None, {"coarse_dict": {"merge_img": merge.detach(), "bg_img": bg.detach()}},
                             torch.from_numpy(g[k + "gt"]).to(dev()), torch.from_numpy(g[k + "mask"]).to(dev()), None)
    assert list(res.keys()) == ["bg_loss", "head_loss", "nonhaed_loss", "total_loss"]
    np.testing.assert_allclose([float(res[n]) for n in res], g[k + "terms"], rtol=2e-6)
    with pytest.raises(NotImplementedError):
        HeadNeRFLossUtils(bg_type="white", use_vgg_loss=True)
    with pytest.raises(ValueError):  # a [B,3,P,P] background is not what the kernel is built for: refuse, do not misread
        fused_data_losses({"merge_img": merge, "bg_img": bg.expand(2, -1, -1, -1)}, torch.from_numpy(g[k + "gt"]).to(dev()),
                          torch.from_numpy(g[k + "mask"]).to(dev()))


@pytest.mark.parametrize("tiling", ["3"])
def test_run_time_kernel_variants_match_the_default(tiling, tmp_path):
    """N3DT_X16_TILING = 3 (16x16x32 MFMA, nerf_fwd_x16b.hip) ships in the library as a run-time switch: against the default
    tiling and against the reference fixtures (cfg1: 32 samples = one block per ray; tiny_train: ragged 8-sample blocks,
    jitter).  The switch is read once per process, hence child processes.  (Tiling 2, 4 waves x 64 samples, was withdrawn in
    round 4: a diagnostic build exposed a latent ordering hazard in that instantiation -- nerf_fwd_x16.hip, tools/tiling_probe.py.)"""
    tool = os.path.join(REPO, "tools", "variant_check.py")
    for name in ("cfg1", "tiny_train"):
        outs = {}
        for t in ("1", tiling):
            out = str(tmp_path / ("%s_t%s.npz" % (name, t)))
            subprocess.run([sys.executable, tool, name, "bf16", out], check=True, env=dict(os.environ, N3DT_X16_TILING=t), timeout=600)
            outs[t] = np.load(out)
        a, b = outs["1"], outs[tiling]
        g, m = load_golden(name)
        # same products in a different association order: bf16 rounding of the activations differs in the last bit
        assert np.abs(a["fg_feat"] - b["fg_feat"]).max() <= 5e-3 and np.abs(a["merge_img"] - b["merge_img"]).max() <= 1e-3
        ref = g["merge_img"] if "merge_img" in g else g["merge_img_q16"].astype(np.float32) / 65535.0
        assert np.abs(b["merge_img"] - ref).max() <= RGB_TOL["bf16"], (name, tiling)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_packed_weight_cache_follows_every_kind_of_weight_write(precision):
    """the packed (MFMA-ordered) weight copies are cached per version counter.  optimizer steps, `copy_` under
    no_grad and load_state_dict move the counters; `.data` writes (the reference trainer's load_ckpt) do not, and need
    invalidate_packed() -- n3dt.checkpoint.load_ckpt does it."""
    from n3dt import BaseOptions, synthetic as syn, checkpoint
    opt = BaseOptions({"featmap_size": 8, "featmap_nc": 256, "pred_img_size": 32, "num_sample_coarse": 32})
    sd_a = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    sd_b = syn.make_state_dict(opt, seed=5, bg_noise=0.1)
    d = to_dev(syn.frame_inputs(opt, 2))
    want_b = fwd(build_net(opt, sd_b, precision), d)["merge_img"]
    want_a = fwd(build_net(opt, sd_a, precision), d)["merge_img"]
    assert float((want_a - want_b).abs().max()) > 1e-2
    net = build_net(opt, sd_a, precision)
    assert torch.equal(fwd(net, d)["merge_img"], want_a)           # cache populated with A
    net.load_state_dict(sd_b)                                       # post-hook drops the cache
    assert torch.equal(fwd(net, d)["merge_img"], want_b)
    for k, v in net.state_dict().items():                           # the reference's load_ckpt: `.data.copy_`
        v.data.copy_(sd_a[k])
    net.invalidate_packed()
    assert torch.equal(fwd(net, d)["merge_img"], want_a)
    assert checkpoint.load_ckpt(net, sd_b) == []                    # n3dt's own: no explicit call needed
    assert torch.equal(fwd(net, d)["merge_img"], want_b)
    with torch.no_grad():                                           # optimizer-style in-place update
        for k, p in net.named_parameters():
            p.copy_(sd_a[k])
    assert torch.equal(fwd(net, d)["merge_img"], want_a)


def test_config3_full_size_training_step():
    """BASELINE config 3 at its real size (fs 64 -> 512^2, 64 samples, B = 2: 524 288 sample points, 7.8 GB of saved fp32
    activations): the fused bf16 training path against the exact fp32 path (same bounds as the small-geometry test in
    test_gpu_train.py), determinism of the training forward, and a loss that falls over three Adam steps."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    from n3dt.train import fused_data_losses, disk_mask
    from test_gpu_train import _grads
    fs, ns, B = 64, 64, 2
    opt = BaseOptions({"featmap_size": fs, "featmap_nc": 256, "pred_img_size": 512, "num_sample_coarse": ns})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    t_rand = syn.stratified_noise(B, fs * fs, ns, 7).to(dev())
    img32, g32 = _grads(opt, sd, B, "fp32", t_rand)
    torch.cuda.empty_cache()
    img16, g16 = _grads(opt, sd, B, "bf16", t_rand)
    img16b, _ = _grads(opt, sd, B, "bf16", t_rand)
    assert torch.equal(img16, img16b), "the training forward must be deterministic"
    assert img32.shape == (B, 3, 512, 512) and float((img32 - img16).abs().max()) <= 2e-3
    for k in g32:
        a, b = g32[k].double().flatten(), g16[k].double().flatten()
        assert float((a - b).abs().max()) <= 5e-2 * float(a.abs().max()) + 1e-12, k
        assert float((a * b).sum() / (a.norm() * b.norm() + 1e-30)) >= 0.995, k
    del g32, g16
    torch.cuda.empty_cache()
    net = HeadNeRFNet(opt, False, False, train_precision="bf16").to(dev())
    net.load_state_dict(sd)
    optim = torch.optim.Adam(net.parameters(), lr=1e-3)
    d = to_dev(syn.frame_inputs(opt, B))
    gt = torch.full((B, 3, 512, 512), 0.5, device=dev())
    mask = disk_mask(B, 512).to(dev())
    losses = []
    for _ in range(3):
        out = net("train", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                  d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand)["coarse_dict"]
        t = fused_data_losses(out, gt, mask)
        loss = t["bg_loss"] + t["head_loss"] + t["nonhead_loss"]
        optim.zero_grad()
        loss.backward()
        optim.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[2] < losses[0], losses


@pytest.mark.parametrize("precision,hier", [("bf16", False), ("fp32", False), ("bf16", True)])
def test_graph_replay_is_bit_identical_to_the_kernel_by_kernel_path(precision, hier):
    """mode="test" forwards replay a hipGraph recorded once per call shape (n3dt_graph_*, n3dt_stage_inputs).  The replay
    must equal the un-captured launch sequence bit for bit, follow new inputs (staged per call), new weights (re-packed in
    place) and a different batch size (a second graph), and accept an expand()ed ray grid."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    opt = BaseOptions({"featmap_size": 16, "featmap_nc": 256, "pred_img_size": 64, "num_sample_coarse": 32, "num_sample_fine": 32})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1, hier_sampling=hier)

    def make(use_graph):
        net = HeadNeRFNet(opt, False, hier, precision=precision, use_graph=use_graph).to(dev())
        net.load_state_dict(sd, strict=True)
        return net

    plain, graphed = make(False), make(True)
    keys = ["coarse_dict"] + (["fine_dict"] if hier else [])

    def both(d):
        with torch.no_grad():
            a = plain("test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                      d["batch_Tvecs"], d["batch_inv_inmats"])
            b = graphed("test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                        d["batch_Tvecs"], d["batch_inv_inmats"])
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(a[k]["merge_img"], b[k]["merge_img"]) and torch.equal(a[k]["bg_img"], b[k]["bg_img"]), k
        return b

    d3 = to_dev(syn.frame_inputs(opt, 3))
    first = both(d3)["coarse_dict"]["merge_img"].clone()
    assert len(graphed._graphs) == 1
    d3b = to_dev(syn.frame_inputs(opt, 3, first_frame=50, yaw_range=0.5))       # new inputs, same shape: same graph
    second = both(d3b)["coarse_dict"]["merge_img"]
    assert len(graphed._graphs) == 1 and not torch.equal(first, second)
    assert torch.equal(first, both(d3)["coarse_dict"]["merge_img"])               # outputs are copies: `first` was not overwritten
    both(to_dev(syn.frame_inputs(opt, 1)))                                       # another batch size: a second graph
    assert len(graphed._graphs) == 2
    with torch.no_grad():                                                        # optimizer-style update of both nets
        for net in (plain, graphed):
            for p in net.parameters():
                p.mul_(1.01)
    third = both(d3)["coarse_dict"]["merge_img"]
    assert not torch.equal(first, third) and len(graphed._graphs) == 2
    graphed.release_graphs()
    assert torch.equal(third, both(d3)["coarse_dict"]["merge_img"])


@pytest.mark.parametrize("train_precision", ["fp32", "bf16"])
def test_gradients_through_the_hierarchical_pass_including_the_cameras(train_precision):
    """SURVEY 8f-1 x 8f-4 (VERDICT r1 missing #4): d loss / d (batch_Rmats, batch_Tvecs, latent codes) THROUGH coarse + fine
    pass against the reference's autograd (tools/gen_golden.py: gen_hier -- the reference's own modules driven in
    _forward's order with gradients enabled).  The fine planes move with the camera's z exactly like the coarse ones."""
    from n3dt import HeadNeRFNet, synthetic as syn
    from n3dt.train import data_losses, disk_mask
    g, m = load_golden("hier_train")
    opt, sd, inp = synthetic_case(m)
    B, n_r = m["batch"], opt.featmap_size ** 2
    net = HeadNeRFNet(opt, False, True, train_precision=train_precision).to(dev())
    net.load_state_dict(sd, strict=True)
    d = to_dev(inp)
    names = ("batch_Rmats", "batch_Tvecs", "shape_code", "appea_code", "audiostyle")
    for k in names:
        d[k] = d[k].clone().requires_grad_(True)
    t_rand = syn.stratified_noise(B, n_r, opt.num_sample_coarse, m["t_rand_seed"]).to(dev())
    fine_u = torch.from_numpy(g["fine_u"]).to(dev())
    out = net("train", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
              d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand, fine_u=fine_u)
    gt = torch.full_like(out["coarse_dict"]["merge_img"], 0.5)
    mask = disk_mask(B, opt.pred_img_size).to(dev())
    total = sum(data_losses(out["coarse_dict"], gt, mask).values()) + sum(data_losses(out["fine_dict"], gt, mask).values())
    tol = {"fp32": (1e-3, 2e-3), "bf16": (3e-2, 0.35)}[train_precision]   # (loss rtol, gradient error / tensor scale)
    np.testing.assert_allclose(float(total.detach()), float(g["loss_total"]), rtol=tol[0])
    total.backward()
    for k in names:
        ref = g["grad_in." + k]
        got = d[k].grad.cpu().numpy()
        err = np.abs(got - ref).max() / np.abs(ref).max()
        cos = float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30))
        print("hier grads %s %s: max err / scale %.2e, cosine %.5f" % (train_precision, k, err, cos))
        # the bf16 path's camera gradients carry the 2^k-amplified rounding of d PE (test_gpu_train.py: direction holds)
        assert err <= (tol[1] if k.startswith("batch_") else min(tol[1], 0.1)), (k, err)
        assert cos >= (0.9999 if train_precision == "fp32" else 0.98), (k, cos)


def test_fine_samp_func_seam_has_the_reference_call():
    """VERDICT r1 missing #5: net.fine_samp_func(batch_weight, coarse_sample_dict, disturb) (NetWorks/utils.py:211) on the
    reference's own coarse weights -> the reference's fine zvals / z_dists / first-ray points (hier fixtures)."""
    from n3dt import HeadNeRFNet, synthetic as syn
    for name in ("hier_test", "hier_train"):
        g, m = load_golden(name)
        opt, sd, inp = synthetic_case(m)
        B, n_r = m["batch"], opt.featmap_size ** 2
        net = HeadNeRFNet(opt, False, True).to(dev())
        net.load_state_dict(sd, strict=True)
        d = to_dev(inp)
        train = m["mode"] == "train"
        t_rand = syn.stratified_noise(B, n_r, opt.num_sample_coarse, m["t_rand_seed"]).to(dev()) if train else None
        coarse = net.sample_func(d["batch_xy"], d["batch_Rmats"], d["batch_Tvecs"], d["batch_inv_inmats"], train, t_rand=t_rand)
        np.testing.assert_allclose(coarse["zvals"].cpu().numpy(), g["coarse_zvals"], atol=2e-6)
        fine_u = torch.from_numpy(g["fine_u"]).to(dev()) if train else None
        fine = net.fine_samp_func(torch.from_numpy(g["coarse_weight"]).to(dev()), coarse, train, fine_u=fine_u)
        assert set(fine.keys()) == {"pts", "dirs", "zvals", "z_dists"}
        N = opt.num_sample_coarse + opt.num_sample_fine
        assert fine["pts"].shape == (B, 3, n_r, N) and fine["dirs"].shape == (B, 3, n_r, N)
        np.testing.assert_allclose(fine["zvals"].cpu().numpy(), g["fine_zvals"], atol=2e-5)
        np.testing.assert_allclose(fine["z_dists"].cpu().numpy(), g["fine_z_dists"], atol=2e-5)
        np.testing.assert_allclose(fine["pts"][:, :, :1].cpu().numpy(), g["fine_pts_ray0"], atol=5e-5)


@pytest.mark.parametrize("train_precision", ["fp32", "bf16"])
def test_single_image_fitting_loop_recovers_a_perturbed_camera_and_codes(train_precision):
    """The reference's fitting use-case (FittingSingleImage_new.py:825-916) end to end on the HIP path: a target is rendered
    from perturbed codes and a perturbed camera; starting from the unperturbed ones, n3dt.fitting's loop (the reference's
    parametrisation, learning rates and schedule) must bring the loss down and move the camera towards the target."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn, fitting
    from n3dt.train import data_losses
    opt = BaseOptions({"featmap_size": 16, "featmap_nc": 256, "pred_img_size": 64, "num_sample_coarse": 32})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    net = HeadNeRFNet(opt, False, False, train_precision=train_precision).to(dev())
    net.load_state_dict(sd)
    for p in net.parameters():
        p.requires_grad_(False)  # fitting optimises codes and camera only (the reference leaves the net untouched)
    d = to_dev(syn.frame_inputs(opt, 1, yaw_range=0.0))
    cam0 = {k: d[k] for k in ("batch_Rmats", "batch_Tvecs", "batch_inv_inmats")}
    # the target: the same head seen from a camera rotated by 0.12 rad about y and shifted by 0.2, codes offset
    true_ang = torch.tensor([[0.0, 0.12, 0.0]], device=dev())
    dR = fitting.eulurangle2Rmat(true_ang)
    g = torch.Generator().manual_seed(5)
    tgt_shape = d["shape_code"] + 0.3 * torch.randn(1, 179, generator=g).to(dev())
    with torch.no_grad():
        target = net("test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, tgt_shape, d["appea_code"], dR.bmm(cam0["batch_Rmats"]),
                     dR.bmm(cam0["batch_Tvecs"]) + torch.tensor([[[0.2], [0.0], [0.0]]], device=dev()), cam0["batch_inv_inmats"])
    gt = target["coarse_dict"]["merge_img"].detach().clone()
    from n3dt.train import disk_mask
    mask = disk_mask(1, 64, radius=0.45).to(dev())
    st = fitting.FittingState(d["shape_code"], d["appea_code"], cam0)
    optim, sched = st.make_optimizer()
    losses = []
    for _ in range(40):
        _, terms, total = fitting.fit_step(net, st, optim, sched, d["batch_xy"], d["batch_uv"], d["audiostyle"], gt, mask, data_losses)
        losses.append(float(total))
    print("fitting %s: total loss %.3e -> %.3e" % (train_precision, losses[0], losses[-1]))
    # (the background term depends on the frozen parameters only and dominates the total of a random-init network: what the
    # five fitted tensors can remove is a few percent of it, and they must remove it steadily)
    assert losses[-1] < 0.95 * losses[0] and all(b < a for a, b in zip(losses[::8], losses[8::8])), losses[::8]
    assert all(v.grad is not None and torch.isfinite(v.grad).all() for v in st.variables())
    # (the camera / code gradients themselves are pinned against the reference's autograd in test_gpu_train.py and in
    # test_gradients_through_the_hierarchical_pass_including_the_cameras; this test is about the loop using them)
    assert float(st.delta_EulurAngles.detach().abs().max()) > 1e-3 and float(st.iden_offset.detach().abs().max()) > 1e-2


def test_split_precision_mode_properties_at_full_size():
    """precision="bf16x3" at the BASELINE size (fs 64, 64 samples, B = 8): size-independent properties -- bit-determinism,
    weights a sub-probability distribution, frame independence -- and agreement with the exact-fp32 kernel on the same inputs
    (feature maps 1e-4, RGB 2e-4: the two parity-grade modes against each other, no stored answer needed)."""
    from n3dt import BaseOptions, synthetic as syn
    opt = BaseOptions({"featmap_size": 64, "featmap_nc": 256, "pred_img_size": 512, "num_sample_coarse": 64})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    d = to_dev(syn.frame_inputs(opt, 8))
    net = build_net(opt, sd, "bf16x3")
    a = feats(net, d, want_weight=True)
    b = feats(net, d, want_weight=True)
    assert torch.equal(a["fg_feat"], b["fg_feat"]) and torch.equal(a["bg_alpha"], b["bg_alpha"])
    w = a["weight"]
    assert float(w.min()) >= 0.0 and float((w.sum(-1) + a["bg_alpha"] - 1.0).abs().max()) <= 1e-5
    one = {k: (v[5:6] if torch.is_tensor(v) else v) for k, v in d.items()}
    assert torch.equal(feats(net, one)["fg_feat"][0], a["fg_feat"][5])
    ref = build_net(opt, sd, "fp32")
    r = feats(ref, d)
    assert float((a["fg_feat"] - r["fg_feat"]).abs().max()) <= 1e-4
    img, img_ref = fwd(net, d)["merge_img"], fwd(ref, d)["merge_img"]
    err = float((img - img_ref).abs().max())
    print("bf16x3 vs fp32 at 8 x 512^2: RGB max|diff| %.2e" % err)
    assert err <= 2e-4


def test_randomised_parity_sweep_against_the_oracle():
    """tools/fuzz_parity.py, 16 seeded cases: random feature-map sizes (incl. non powers of two), 1 - 3 renderer stages, 1 - 100
    samples per ray (ragged 16- / 32-sample blocks), batch 1 - 5, gaze / audio-less variants, test and train mode, in all four
    render precisions against the CPU oracle (400 cases over three seeds were run during round 2: fp32 <= 3.6e-7, bf16x3
    <= 9.8e-5, fp16 <= 3.2e-4, bf16 <= 3e-3 on RGB)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(REPO, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    sys.argv = ["fuzz_parity.py", "16", "3"]
    try:
        assert mod.main() == 0
    finally:
        sys.argv = argv


def test_oversized_renderer_batches_go_through_in_slices():
    """NeuralRenderer.render_hwc slices a batch that the 32-bit level indexing of the kernels cannot take in one call (255 maps at
    512^2); forced here with a cap of 2 maps per call: same images as the single call, bit for bit."""
    from n3dt import BaseOptions, HeadNeRFNet
    opt = BaseOptions({"featmap_size": 8, "featmap_nc": 256, "pred_img_size": 32, "num_sample_coarse": 8})
    net = HeadNeRFNet(opt, False, False).to(dev())
    nr = net.neural_render
    maps = torch.randn(5, 8, 8, 256, generator=torch.Generator().manual_seed(3)).to(dev())
    for prec in ("fp32", "bf16"):
        whole = nr.render_hwc(maps, prec).clone()
        nr._max_maps_per_call = 2
        try:
            sliced = nr.render_hwc(maps, prec)
        finally:
            nr._max_maps_per_call = None
        assert torch.equal(whole, sliced), prec


def test_cached_renderer_batch_follows_the_background_parameter():
    """forward() keeps its renderer input batch (merged maps + background map) per batch size and stream, with the background slot
    filled once per parameter version: an in-place update of `neural_render.bg_featmap` (what an optimizer does) must show in the
    next forward, and a write through `.data` after invalidate_packed()."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    opt = BaseOptions({"featmap_size": 8, "featmap_nc": 256, "pred_img_size": 32, "num_sample_coarse": 8})
    net = HeadNeRFNet(opt, False, False, precision="bf16").to(dev())
    net.load_state_dict(syn.make_state_dict(opt, seed=0, bg_noise=0.1))
    d = {k: (v.to(dev()) if torch.is_tensor(v) else v) for k, v in syn.frame_inputs(opt, 2).items()}

    def run(n):
        with torch.no_grad():
            o = n("test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                  d["batch_Tvecs"], d["batch_inv_inmats"])["coarse_dict"]
        return o["merge_img"].clone(), o["bg_img"].clone()

    m0, b0 = run(net)
    m0b, b0b = run(net)
    assert torch.equal(m0, m0b) and torch.equal(b0, b0b)
    with torch.no_grad():
        net.neural_render.bg_featmap.mul_(0.5)  # version counter moves
    m1, b1 = run(net)
    assert float((b1 - b0).abs().max()) > 1e-3
    net.neural_render.bg_featmap.data.mul_(2.0)  # back to the original values; `.data` writes move no counter
    net.invalidate_packed()
    m2, b2 = run(net)
    assert torch.equal(b2, b0) and torch.equal(m2, m0)
