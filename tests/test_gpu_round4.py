"""Round-4 GPU tests: RCCL on the device path (one-rank group), config 4's training shape (B = 4), the whole training step as
one hipGraph replay, a network trained by the build's own trainer across the inference precisions, include_vd."""
import contextlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev():
    return torch.device("cuda:0")


def to_dev(inp):
    return {k: (v.to(dev()) if torch.is_tensor(v) else v) for k, v in inp.items()}


def test_rccl_one_rank_group_reduces_both_buckets_in_order(tmp_path):
    """VERDICT r3 #3 (the reference pins one GPU, talker_trainer.py:704-714): config-4 training step, B = 4, fused bf16
    path, with GradReducer forced to register its hooks on a world-size-1 `nccl` group: bucket 0 (HeadNeRFNet's arena) is
    all-reduced async from inside backward on RCCL's stream right behind the ctypes-launched weight-gradient kernels, bucket 1
    at wait().  Gradients must equal the no-reducer run's up to the fp32-atomic ordering noise the arena test allows."""
    out = str(tmp_path / "rccl.json")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    import socket
    with socket.socket() as sock:   # a free rendezvous port (the worker's default is a fixed one)
        sock.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(sock.getsockname()[1])
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "_rccl_worker.py"), out], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, "worker failed:\n%s\n%s" % (r.stdout[-2000:], r.stderr[-4000:])
    rec = json.load(open(out))
    assert rec["world"] == 1 and rec["backend"] == "nccl"
    assert rec["hook_launches"] >= 1, "no collective was launched from inside backward: %s" % rec
    assert rec["last_launch_order"][0] == 0, rec
    assert rec["late_rounds"] == 0, rec
    assert rec["grads_in_arena"] >= rec["n_params"] - 1, rec
    # same band as test_gradient_arena_holds_the_same_gradients_as_fresh_buffers (bf16 path: fp32 atomics in varying order)
    assert rec["worst_rel"] <= 2e-2, rec


def _train_setup(fs, ns, pred, B, graph):
    from n3dt import BaseOptions, HeadNeRFNet, parallel, synthetic as syn
    from n3dt.train import fused_data_losses, disk_mask
    opt = BaseOptions({"featmap_size": fs, "featmap_nc": 256, "pred_img_size": pred, "num_sample_coarse": ns})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    d = to_dev(syn.frame_inputs(opt, B))
    net = HeadNeRFNet(opt, False, False, train_precision="bf16").to(dev())
    net.load_state_dict(sd, strict=True)
    capt = dict(capturable=True) if graph else {}
    optim = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True, **capt)
    bucket = parallel.FlatBucket(numel=4096).to(dev())
    optim2 = torch.optim.Adam(bucket.parameters(), lr=1e-7, betas=(0.5, 0.999), fused=True, **capt)
    gt = torch.full((B, 3, pred, pred), 0.5, device=dev())
    mask = disk_mask(B, pred).to(dev())
    t_rand = syn.stratified_noise(B, fs * fs, ns, seed=3).to(dev())
    losses = []

    def step():
        out = net("train", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                  d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand)
        t = fused_data_losses(out["coarse_dict"], gt, mask)
        optim.zero_grad()
        t["total_loss"].backward()
        bucket.fill_grad(1e-3)
        optim.step()
        optim2.step()
        return t["total_loss"].detach()
    return net, step, bucket


def test_whole_training_step_replays_as_one_graph():
    """The reference's step -- forward("train"), three MSE terms, backward, two Adam steps (talker_trainer.py:1002-1067) -- recorded
    once into a hipGraph (n3dt.train.GraphedTrainStep) and replayed: after the same number of steps from the same weights, data
    and jitter, the graphed run's parameters and loss equal the eager run's up to the fp32-atomic ordering noise of the weight
    gradients (Adam normalises the update, so a parameter moves by ~lr per step whatever the gradient's size: 12 steps x 1e-4)."""
    from n3dt.train import GraphedTrainStep
    n_steps, warm = 12, 3
    net_e, step_e, _ = _train_setup(16, 32, 64, 2, graph=False)
    for _ in range(n_steps):
        loss_e = step_e()
    net_g, step_g, bucket_g = _train_setup(16, 32, 64, 2, graph=True)
    g = GraphedTrainStep(step_g, warmup=warm)          # `warm` eager steps + the captured one do not run at capture time...
    for _ in range(n_steps - warm):                     # ... so replay until both runs made n_steps optimizer steps
        loss_g = g()
    torch.cuda.synchronize()
    assert abs(float(loss_g) - float(loss_e)) <= 2e-3 * abs(float(loss_e)) + 1e-7, (float(loss_g), float(loss_e))
    moved = 0.0
    for (n, a), (_, b) in zip(net_e.named_parameters(), net_g.named_parameters()):
        d0 = float((a - b).abs().max())
        # Ten repetitions (tools/graph_step_spread_probe.py, profiles/r04_q_graph_step_spread_10runs.log): the largest difference
        # between TWO EAGER runs is 1.0e-4 .. 3.0e-4 and between an eager and a graphed run 1.0e-4 .. 3.0e-4 -- one distribution
        # (Adam moves a weight whose gradient is summation-order noise by up to lr per step either way).  2 x its worst draw,
        # half of the 1.2e-3 a parameter travelled (round 4's first bound, 2.5e-4, sat inside the distribution: 2 draws of 20 above)
        assert d0 <= 6e-4, (n, d0)
        moved = max(moved, d0)
    # the second optimizer ran inside the graph too: 9 replays x lr 1e-7 on a constant gradient
    assert float(bucket_g.flat.detach().abs().max()) > 5e-7


def test_a_network_trained_by_the_builds_own_trainer_across_the_inference_precisions():
    """VERDICT r3 #5.  The released checkpoints are absent (/root/reference/.MISSING_LARGE_BLOBS), so "does a 16-bit mode hold
    1e-3 on a trained head" is answered on a head the build's OWN exact-fp32 trainer makes sharp: seed-0 weights, config 4's
    geometry, a disk of colour on white, Adam until alpha saturates on >= 30 % of the rays (opaque AND carried by one sample).
    Frame 0 is then rendered in four precisions against the CPU oracle ON THOSE WEIGHTS: fp32 and bf16x3 must hold the
    north-star's 1e-3; bf16 / fp16 report their own error (asserted only against garbage)."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    from oracle import oracle as orc
    opt = BaseOptions({"featmap_size": 32, "featmap_nc": 256, "pred_img_size": 256, "num_sample_coarse": 64})
    net, info = syn.train_sharp_head(opt, dev(), steps=600, lr=1e-3, batch=2, want_share=0.3)
    assert info["alpha_saturated_ray_share"] >= 0.3 and info["one_sample_rays_share"] >= 0.3, info
    assert info["loss_last"] < 0.05 * info["loss_first"], info
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    one = syn.frame_inputs(opt, 1)
    d = to_dev(one)
    ref = orc.forward(sd, opt, one)
    errs = {}
    for prec in ("bf16", "fp16", "bf16x3", "fp32"):
        n2 = HeadNeRFNet(opt, False, False, precision=prec).to(dev())
        n2.load_state_dict(sd, strict=True)
        with torch.no_grad():
            r = n2("test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                   d["batch_Tvecs"], d["batch_inv_inmats"])["coarse_dict"]
        errs[prec] = max(float(np.abs(r["merge_img"].cpu().numpy() - ref["merge_img"]).max()),
                         float(np.abs(r["bg_img"].cpu().numpy() - ref["bg_img"]).max()))
    print("trained-network RGB L-inf vs the oracle:", errs, info)
    assert errs["fp32"] <= 1e-3 and errs["bf16x3"] <= 1e-3, errs
    assert errs["bf16"] <= 0.2 and errs["fp16"] <= 0.2, errs


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_a_non_finite_feature_stays_local_in_the_renderer(precision):
    """One NaN in the renderer's input map.  In the reference it spreads only as far as the 3x3 blurs and the bilinear
    upsampling carry it (NetWorks/neural_renderer.py:72-91): a ~25-pixel patch of the 256^2 image.  The 16-bit path evaluates the
    blur / activation / RGB stage as a matrix product over 32- or 64-pixel tiles (csrc/nr_blur_mfma.inc): every halo slot is
    multiplied by its stencil weight, 0 included, so a NaN anywhere in a tile's input patch reaches ALL of that tile's output pixels
    (0 * NaN) -- a documented deviation: the poisoned set is a superset of the reference's, larger by at most a
    tile per level, and everything outside it is bit-identical to the clean render.  The exact fp32 path poisons exactly the
    reference's pixels."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    from oracle import oracle as orc
    opt = BaseOptions({"featmap_size": 32, "featmap_nc": 256, "pred_img_size": 256, "num_sample_coarse": 8})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    net = HeadNeRFNet(opt, False, False, precision=precision).to(dev())
    net.load_state_dict(sd, strict=True)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(1, 256, 32, 32, generator=gen)
    xn = x.clone()
    xn[0, 17, 13, 21] = float("nan")
    ref = orc.neural_render(sd, xn.numpy(), 3)
    bad_ref = ~np.isfinite(ref).all(axis=1)[0]          # [256,256] pixels the reference poisons
    assert 100 < bad_ref.sum() < 2000

    def render(t):
        hwc = t.permute(0, 2, 3, 1).contiguous().to(dev())
        with torch.no_grad():
            return net.neural_render.render_hwc(hwc, precision).cpu().numpy()
    clean, dirty = render(x), render(xn)
    bad = ~np.isfinite(dirty).all(axis=1)[0]
    assert (bad | ~bad_ref).all(), "a pixel the reference poisons came out finite"
    if precision == "fp32":
        assert (bad == bad_ref).all()
    ys, xs = np.nonzero(bad_ref)
    y0, y1, x0, x1 = ys.min(), ys.max(), xs.min(), xs.max()
    grow = 56  # a tile edge (8 pixels) per level, scaled by the upsamplings behind that level: 8 * (4 + 2 + 1)
    allowed = np.zeros_like(bad)
    allowed[max(0, y0 - grow):y1 + grow + 1, max(0, x0 - grow):x1 + grow + 1] = True
    assert not (bad & ~allowed).any(), "NaN reached pixels far from the reference's patch: %d" % int((bad & ~allowed).sum())
    assert np.array_equal(dirty[0][:, ~bad], clean[0][:, ~bad]), "finite pixels changed"


# ---- include_vd=True (NetWorks/HeadNeRFNet.py:56-63,86,141-142) ------------------------------------------------------------
VD_RGB_TOL = {"fp32": 1e-4, "bf16": 1e-3, "fp16": 5e-4, "bf16x3": 1e-4}
VD_FEAT_TOL = {"fp32": 2e-5, "bf16": 5e-3, "fp16": 1e-3, "bf16x3": 1e-4}


def _vd_setup(name, precision="fp32", train_precision="fp32"):
    from conftest import load_golden, synthetic_case
    from n3dt import HeadNeRFNet, synthetic as syn
    g, m = load_golden(name)
    opt, sd, inp = synthetic_case(m)
    net = HeadNeRFNet(opt, include_vd=True, hier_sampling=False, precision=precision, train_precision=train_precision).to(dev())
    net.load_state_dict(sd, strict=True)
    t_rand = None
    if m["mode"] == "train":
        t_rand = syn.stratified_noise(m["batch"], opt.featmap_size ** 2, opt.num_sample_coarse, m["t_rand_seed"]).to(dev())
    return g, m, opt, sd, net, to_dev(inp), t_rand


@pytest.mark.parametrize("precision", ["fp32", "bf16", "fp16", "bf16x3"])
@pytest.mark.parametrize("name", ["vd_test", "vd_train"])
def test_include_vd_render_against_the_reference(name, precision):
    """The module built with include_vd=True against the REFERENCE module built the same way (fixtures vd_*): the 27 view-direction
    channels of RGB_layer_1 enter as a per-ray bias (one 27-wide product per ray, csrc/nerf_aux.hip: ray_vd_bias_kernel), in the
    exact-fp32 kernel and the three MFMA kernels."""
    g, m, opt, sd, net, d, t_rand = _vd_setup(name, precision)
    with torch.no_grad():
        f = net.render_features(d["batch_xy"], d["audiostyle"], d["shape_code"], d["appea_code"], d["batch_Rmats"], d["batch_Tvecs"],
                                d["batch_inv_inmats"], t_rand=t_rand)
        out = net(m["mode"], d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                  d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand)["coarse_dict"]
    np.testing.assert_allclose(f["fg_feat"].permute(0, 2, 1).cpu().numpy(), g["fg_feat"], atol=VD_FEAT_TOL[precision])
    np.testing.assert_allclose(f["bg_alpha"].cpu().numpy()[:, None], g["bg_alpha"], atol=VD_FEAT_TOL[precision])
    assert np.abs(out["merge_img"].cpu().numpy() - g["merge_img"]).max() <= VD_RGB_TOL[precision]
    # the stand-alone seams: vd_encoder(dirs) and fg_CD_predictor(audio, embed_vps, embed_vds) with the 154-channel embed_vds
    if precision == "fp32":
        smp = net.sample_func(d["batch_xy"], d["batch_Rmats"], d["batch_Tvecs"], d["batch_inv_inmats"], m["mode"] == "train", t_rand=t_rand)
        vd = net.vd_encoder(smp["dirs"])
        np.testing.assert_allclose(vd[:, :, :, 0].cpu().numpy(), g["vd_embed_ray"], atol=1e-6)
        ns = opt.num_sample_coarse
        n_r = smp["pts"].shape[2]
        vp = torch.cat([net.vp_encoder(smp["pts"]), d["shape_code"][:, :, None, None].expand(-1, -1, n_r, ns)], dim=1)
        vds = torch.cat([vd, d["appea_code"][:, :, None, None].expand(-1, -1, n_r, ns)], dim=1)
        rgb, dens = net.fg_CD_predictor(d["audiostyle"][:, :, None, None].expand(-1, -1, n_r, ns), vp, vds)
        np.testing.assert_allclose(rgb.cpu().numpy(), g["feat"], atol=2e-4)


def test_include_vd_mid_size_against_the_oracle():
    """fs 16, 48 samples (a ragged second block), B = 3, the hierarchical pass off: every precision against the CPU oracle."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    from oracle import oracle as orc
    opt = BaseOptions({"featmap_size": 16, "featmap_nc": 256, "pred_img_size": 64, "num_sample_coarse": 48})
    sd = syn.make_state_dict(opt, seed=4, bg_noise=0.1, include_vd=True)
    inp = syn.frame_inputs(opt, 3)
    ref = orc.forward(sd, opt, inp, include_vd=True)
    d = to_dev(inp)
    for precision in ("fp32", "bf16", "fp16", "bf16x3"):
        net = HeadNeRFNet(opt, include_vd=True, hier_sampling=False, precision=precision).to(dev())
        net.load_state_dict(sd, strict=True)
        with torch.no_grad():
            out = net("test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                      d["batch_Tvecs"], d["batch_inv_inmats"])["coarse_dict"]
        assert np.abs(out["merge_img"].cpu().numpy() - ref["merge_img"]).max() <= VD_RGB_TOL[precision], precision


@pytest.mark.parametrize("name", ["vd_test", "vd_train"])
def test_include_vd_gradients_match_the_reference_autograd(name):
    """Exact-fp32 training path with include_vd=True against the reference's autograd: every parameter (the 27 view-direction
    columns of RGB_layer_1 in full), the latent codes and the cameras -- the rotation's gradient now has a second route, through
    the ray direction into the encoder -- and one Adam step."""
    from n3dt.train import data_losses, disk_mask
    g, m, opt, sd, net, d, t_rand = _vd_setup(name)
    for k in ("audiostyle", "shape_code", "appea_code", "batch_Rmats", "batch_Tvecs"):
        d[k] = d[k].clone().requires_grad_(True)
    out = net(m["mode"], d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
              d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand)["coarse_dict"]
    terms = data_losses(out, torch.full_like(out["merge_img"], 0.5), disk_mask(m["batch"], opt.pred_img_size).to(dev()))
    np.testing.assert_allclose([float(terms[k].detach()) for k in ("bg_loss", "head_loss", "nonhead_loss")], g["loss_terms"], atol=1e-6)
    (terms["bg_loss"] + terms["head_loss"] + terms["nonhead_loss"]).backward()
    for k in ("audiostyle", "shape_code", "appea_code", "batch_Rmats", "batch_Tvecs"):
        ref = g["grad_in." + k]
        tol = 5e-2 if k.startswith("batch_") else 2e-2
        assert np.abs(d[k].grad.cpu().numpy() - ref).max() <= tol * np.abs(ref).max(), k
    for pname, p in net.named_parameters():
        assert p.grad is not None, pname
        idx, val = g["grad_p.%s.idx" % pname], g["grad_p.%s.val" % pname]
        got = p.grad.reshape(-1)[torch.from_numpy(idx).to(dev())].cpu().numpy()
        assert np.abs(got - val).max() <= 2e-2 * (np.abs(val).max() + 1e-12), pname
    gv = net.fg_CD_predictor.RGB_layer_1.weight.grad[:, 384:411, 0, 0].cpu().numpy()
    assert np.abs(gv - g["grad_vd_columns"]).max() <= 2e-2 * np.abs(g["grad_vd_columns"]).max()
    assert np.abs(g["grad_vd_columns"]).max() > 0


def test_include_vd_bf16_training_path_against_the_fp32_path():
    """The fused bf16 training path with include_vd=True: its per-ray bias rides in the merged RGB stage, its gradient is the
    per-ray sum of that stage's dZ tiles (dz_ray_rowsum_kernel).  Against the exact path on the same inputs, the usual bf16 band."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    from n3dt.train import data_losses, disk_mask
    opt = BaseOptions({"featmap_size": 16, "featmap_nc": 256, "pred_img_size": 64, "num_sample_coarse": 40})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1, include_vd=True)
    B = 2
    t_rand = syn.stratified_noise(B, 256, 40, 7).to(dev())

    def grads(tp):
        net = HeadNeRFNet(opt, include_vd=True, hier_sampling=False, train_precision=tp).to(dev())
        net.load_state_dict(sd, strict=True)
        net.neural_render.train_precision = "fp32"
        d = to_dev(syn.frame_inputs(opt, B))
        d["batch_Rmats"] = d["batch_Rmats"].clone().requires_grad_(True)
        out = net("train", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                  d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand)["coarse_dict"]
        t = data_losses(out, torch.full_like(out["merge_img"], 0.5), disk_mask(B, opt.pred_img_size).to(dev()))
        (t["bg_loss"] + t["head_loss"] + t["nonhead_loss"]).backward()
        gd = {n: p.grad.detach().clone() for n, p in net.named_parameters() if n.startswith("fg_CD_predictor")}
        return out["merge_img"].detach(), gd
    img32, g32 = grads("fp32")
    img16, g16 = grads("bf16")
    assert float((img32 - img16).abs().max()) <= 2e-3
    for k in g32:
        a, b = g32[k].double().flatten(), g16[k].double().flatten()
        assert float((a - b).abs().max()) <= 5e-2 * float(a.abs().max()) + 1e-12, k
        assert float((a * b).sum() / (a.norm() * b.norm() + 1e-30)) >= 0.995, k
    vd32 = g32["fg_CD_predictor.RGB_layer_1.weight"][:, 384:411]
    assert float(vd32.abs().max()) > 0


def test_include_vd_with_the_hierarchical_pass_and_gaze():
    """include_vd together with the other constructor arguments: hier_sampling=True (the fine network has its own 27 view-direction
    columns; its per-ray bias is formed from the same directions) and include_gaze=True -- inference against the oracle's coarse +
    fine passes, and the differentiable path: gradients reach both networks' view-direction columns."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    from n3dt.train import data_losses, disk_mask
    from oracle import oracle as orc
    opt = BaseOptions({"featmap_size": 8, "featmap_nc": 256, "pred_img_size": 32, "num_sample_coarse": 16, "num_sample_fine": 24})
    kw = {"include_gaze": True, "eye_gaze_dim": 64}
    sd = syn.make_state_dict(opt, seed=2, bg_noise=0.1, hier_sampling=True, include_vd=True, **kw)
    inp = syn.frame_inputs(opt, 2, **kw)
    ref = orc.forward_hier(sd, opt, inp, include_vd=True)
    d = to_dev(inp)
    net = HeadNeRFNet(opt, include_vd=True, hier_sampling=True, **kw).to(dev())
    net.load_state_dict(sd, strict=True)
    with torch.no_grad():
        out = net("test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                  d["batch_Tvecs"], d["batch_inv_inmats"])
    assert np.abs(out["coarse_dict"]["merge_img"].cpu().numpy() - ref["coarse_merge_img"]).max() <= 1e-4
    assert np.abs(out["fine_dict"]["merge_img"].cpu().numpy() - ref["fine_merge_img"]).max() <= 1e-3
    # differentiable path: same images, and both networks' view-direction columns receive gradients
    o2 = net("test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
             d["batch_Tvecs"], d["batch_inv_inmats"])
    assert float((o2["fine_dict"]["merge_img"].detach() - out["fine_dict"]["merge_img"]).abs().max()) <= 2e-5
    t = data_losses(o2["fine_dict"], torch.full_like(o2["fine_dict"]["merge_img"], 0.5), disk_mask(2, 32).to(dev()))
    t2 = data_losses(o2["coarse_dict"], torch.full_like(o2["coarse_dict"]["merge_img"], 0.5), disk_mask(2, 32).to(dev()))
    (t["head_loss"] + t["nonhead_loss"] + t2["head_loss"] + t2["nonhead_loss"]).backward()
    for name in ("fg_CD_predictor", "fine_fg_CD_predictor"):
        g = getattr(net, name).RGB_layer_1.weight.grad
        assert g is not None and float(g[:, 384:411].abs().max()) > 0 and float(g[:, :384].abs().max()) > 0 and float(g[:, 411:].abs().max()) > 0, name


@pytest.mark.parametrize("tp", ["fp32", "bf16"])
def test_include_vd_frozen_network_gives_the_same_input_gradients(tp):
    """Single-image fitting (FittingSingleImage_new.py:826-859) freezes the network and optimises codes and cameras.  With
    include_vd the rotation's gradient has a route through the ray direction: the frozen backward (no weight-gradient stage) must
    return the same d codes / d cameras as the full one."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    from n3dt.train import data_losses, disk_mask
    opt = BaseOptions({"featmap_size": 16, "featmap_nc": 256, "pred_img_size": 64, "num_sample_coarse": 32})
    sd = syn.make_state_dict(opt, seed=1, bg_noise=0.1, include_vd=True)
    names = ("audiostyle", "shape_code", "appea_code", "batch_Rmats", "batch_Tvecs")

    def grads(frozen):
        net = HeadNeRFNet(opt, include_vd=True, hier_sampling=False, train_precision=tp).to(dev())
        net.load_state_dict(sd, strict=True)
        if frozen:
            for p in net.parameters():
                p.requires_grad_(False)
        d = to_dev(syn.frame_inputs(opt, 1))
        for k in names:
            d[k] = d[k].clone().requires_grad_(True)
        out = net("test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                  d["batch_Tvecs"], d["batch_inv_inmats"])["coarse_dict"]
        t = data_losses(out, torch.full_like(out["merge_img"], 0.5), disk_mask(1, 64).to(dev()))
        (t["bg_loss"] + t["head_loss"] + t["nonhead_loss"]).backward()
        return {k: d[k].grad.detach().clone() for k in names if d[k].grad is not None}
    full, froz = grads(False), grads(True)
    assert set(full) == set(froz) == set(names)
    for k in names:
        scale = float(full[k].abs().max())
        assert float((full[k] - froz[k]).abs().max()) <= (2e-4 if tp == "fp32" else 2e-3) * scale + 1e-12, k


def test_maximum_size_batches_equal_their_frames_rendered_alone():
    """Size-independent property at sizes past every 32-bit limit (tools/big_batch_probe.py): frames are independent, so a batch
    rendered in one call equals the same frames rendered one at a time, bit for bit -- 20 reading-N frames of config 2 in one call
    (5.2 M rays, 336 M sample points, fg_feat 1.25 G floats = 5.4 GB: element offsets past 2^30, byte offsets past 2^32), 64 heads
    through the whole forward, and 16 heads in one fused-bf16 training step (latent-code gradients of a head, times the batch size,
    equal the one-head step's up to the order of fp32 atomic sums)."""
    import importlib.util
    from n3dt import BaseOptions
    spec = importlib.util.spec_from_file_location("big_batch_probe", os.path.join(os.path.dirname(__file__), "..", "tools", "big_batch_probe.py"))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    opt = BaseOptions({"featmap_size": 64, "featmap_nc": 256, "pred_img_size": 512, "num_sample_coarse": 64})
    assert probe.render_sections(dev(), opt, 20, "bf16") == 0
    torch.cuda.empty_cache()
    assert probe.train_section(dev(), opt, 16) == 0
    torch.cuda.empty_cache()


def test_all_work_is_enqueued_on_the_callers_stream():
    """Boundary contract (include/n3dt.h: every entry point takes the stream; SURVEY 8b "all work on the current stream"): with the
    default stream kept busy for ~0.5 s, an inference forward, a hierarchical forward and three training steps issued on a side
    stream -- and joined by synchronising THAT stream only -- equal the default-stream run.  Anything the library or the host code
    put on the null stream (a memset, a copy, a pack) would still be queued behind the busy work when the side stream finishes."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn

    def run(side):
        torch.manual_seed(0)
        net, step, _ = _train_setup(16, 32, 64, 2, graph=False)
        opt = BaseOptions({"featmap_size": 16, "featmap_nc": 256, "pred_img_size": 64, "num_sample_coarse": 32, "num_sample_fine": 16})
        hier = HeadNeRFNet(opt, False, True).to(dev())
        hier.load_state_dict(syn.make_state_dict(opt, seed=2, bg_noise=0.1, hier_sampling=True), strict=True)
        d = to_dev(syn.frame_inputs(opt, 2))
        call = lambda n: n("test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"],  # noqa: E731
                           d["batch_Rmats"], d["batch_Tvecs"], d["batch_inv_inmats"])
        torch.cuda.synchronize()
        busy = None
        if side is not None:
            x = torch.randn(8192, 8192, device=dev())
            for _ in range(40):          # ~0.5 s of fp32 GEMMs on the default stream, not waited for below
                busy = x @ x
        ctx = torch.cuda.stream(side) if side is not None else contextlib.nullcontext()
        with ctx:
            with torch.no_grad():
                img = call(net)["coarse_dict"]["merge_img"].clone()
                o = call(hier)
                fine = o["fine_dict"]["merge_img"].clone()
            losses = [step() for _ in range(3)]
            w = net.fg_CD_predictor.FeaExt_module_3.weight.detach().clone()
        (side.synchronize() if side is not None else torch.cuda.synchronize())
        res = (img.cpu(), fine.cpu(), torch.stack(losses).cpu(), w.cpu())
        torch.cuda.synchronize()
        del busy
        return res
    ref = run(None)
    got = run(torch.cuda.Stream())
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
    assert float((ref[2] - got[2]).abs().max()) <= 2e-4 * float(ref[2].abs().max())     # fp32 atomics of the bf16 training path
    assert float((ref[3] - got[3]).abs().max()) <= 1e-4


def test_two_threads_two_streams_two_networks():
    """The C ABI is re-entrant (include/n3dt.h: no state but the thread-local error string): two host threads, each with its own
    network (different geometry and precision), its own stream and 30 forwards + 2 training steps, running at the same time
    (ctypes releases the GIL inside every entry point), give what each gives alone."""
    import threading
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    from n3dt.train import fused_data_losses, disk_mask
    specs = [dict(fs=16, ns=32, pred=64, prec="bf16", seed=0, vd=False), dict(fs=8, ns=24, pred=32, prec="fp16", seed=5, vd=True)]

    def work(sp, stream, out):
        try:
            opt = BaseOptions({"featmap_size": sp["fs"], "featmap_nc": 256, "pred_img_size": sp["pred"], "num_sample_coarse": sp["ns"]})
            with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
                net = HeadNeRFNet(opt, sp["vd"], False, precision=sp["prec"], train_precision="bf16").to(dev())
                net.load_state_dict(syn.make_state_dict(opt, seed=sp["seed"], bg_noise=0.1, include_vd=sp["vd"]), strict=True)
                d = to_dev(syn.frame_inputs(opt, 2))
                a = (d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                     d["batch_Tvecs"], d["batch_inv_inmats"])
                with torch.no_grad():
                    for _ in range(30):
                        img = net("test", *a)["coarse_dict"]["merge_img"]
                    img = img.clone()
                t_rand = syn.stratified_noise(2, sp["fs"] ** 2, sp["ns"], seed=3).to(dev())
                gt, mask = torch.full((2, 3, sp["pred"], sp["pred"]), 0.5, device=dev()), disk_mask(2, sp["pred"]).to(dev())
                for _ in range(2):
                    net.zero_grad()
                    loss = fused_data_losses(net("train", *a, t_rand=t_rand)["coarse_dict"], gt, mask)["total_loss"]
                    loss.backward()
                g = net.fg_CD_predictor.FeaExt_module_2.weight.grad.detach().clone()
                (stream.synchronize() if stream is not None else torch.cuda.synchronize())
                out.update(img=img.cpu(), loss=float(loss.detach()), g=g.cpu())
        except BaseException as e:  # noqa: BLE001  (reported by the main thread)
            out["error"] = repr(e)

    alone = [dict(), dict()]
    for sp, o in zip(specs, alone):
        work(sp, None, o)
        assert "error" not in o, o
    both = [dict(), dict()]
    threads = [threading.Thread(target=work, args=(sp, torch.cuda.Stream(), o)) for sp, o in zip(specs, both)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    for a, b in zip(alone, both):
        assert "error" not in b, b
        assert torch.equal(a["img"], b["img"])
        assert abs(a["loss"] - b["loss"]) <= 1e-4 * abs(a["loss"])
        assert float((a["g"] - b["g"]).abs().max()) <= 2e-2 * float(a["g"].abs().max())   # fp32 atomics of the bf16 path


def _dirty_the_register_files():
    """Leave NaNs in the vector and accumulator registers (and LDS) of every CU: large fp32 / bf16 / fp16 GEMMs of NaN matrices
    (the BLAS kernels keep their accumulators in AGPRs and stage tiles through LDS) and an elementwise pass."""
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        x = torch.full((4096, 4096), float("nan"), device=dev(), dtype=dt)
        y = x @ x
        y = torch.sin(y) + x
    torch.cuda.synchronize()
    del x, y


@pytest.mark.parametrize("ns", [8, 40, 65])
def test_results_do_not_depend_on_what_the_registers_held(ns):
    """Round 4 found a build (the withdrawn 64-sample tiling under diagnostic flags) whose lanes past N_s read stale accumulator
    registers: right in a fresh process, wrong once other kernels had run on the CU (docs/tuning_log.md: hipcc parked two values in
    AGPRs under a reduced lane mask).  The static gate looks for that code shape; this is the dynamic side: every precision of the
    inference path and both training paths, at sample counts that leave 24 / 24 / 31 dead lanes in the last 32-sample block, give
    the SAME result after the register files of the whole chip were filled with NaNs (bit-identical for inference, fp32-atomic
    noise for the gradients)."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    from n3dt.train import fused_data_losses, disk_mask
    opt = BaseOptions({"featmap_size": 12, "featmap_nc": 256, "pred_img_size": 48, "num_sample_coarse": ns})
    sd = syn.make_state_dict(opt, seed=3, bg_noise=0.1)
    d = to_dev(syn.frame_inputs(opt, 3))
    a = (d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"], d["batch_Tvecs"],
         d["batch_inv_inmats"])
    t_rand = syn.stratified_noise(3, 144, ns, seed=5).to(dev())
    gt, mask = torch.full((3, 3, 48, 48), 0.5, device=dev()), disk_mask(3, 48).to(dev())

    def everything():
        out = {}
        for prec in ("bf16", "fp16", "bf16x3", "fp32"):
            net = HeadNeRFNet(opt, False, False, precision=prec).to(dev())
            net.load_state_dict(sd, strict=True)
            with torch.no_grad():
                out["img_" + prec] = net("test", *a)["coarse_dict"]["merge_img"].clone()
                f = net.render_features(d["batch_xy"], d["audiostyle"], d["shape_code"], d["appea_code"], d["batch_Rmats"], d["batch_Tvecs"],
                                        d["batch_inv_inmats"], t_rand=t_rand, want_depth=True, want_weight=True)
                for k in ("fg_feat", "bg_alpha", "depth", "weight"):
                    out["%s_%s" % (k, prec)] = f[k].clone()
        for tp in ("fp32", "bf16"):
            net = HeadNeRFNet(opt, False, False, train_precision=tp).to(dev())
            net.load_state_dict(sd, strict=True)
            loss = fused_data_losses(net("train", *a, t_rand=t_rand)["coarse_dict"], gt, mask)["total_loss"]
            loss.backward()
            out["loss_" + tp] = loss.detach().clone()
            out["g7_" + tp] = net.fg_CD_predictor.FeaExt_module_7.weight.grad.detach().clone()
            out["gd_" + tp] = net.fg_CD_predictor.density_module.weight.grad.detach().clone()
        torch.cuda.synchronize()
        return out
    clean = everything()
    _dirty_the_register_files()
    dirty = everything()
    for k in clean:
        assert torch.isfinite(dirty[k]).all(), k
        if k.startswith(("g7_", "gd_", "loss_")):
            scale = float(clean[k].abs().max()) + 1e-30
            assert float((clean[k] - dirty[k]).abs().max()) <= 2e-2 * scale, k
        else:
            assert torch.equal(clean[k], dirty[k]), (k, float((clean[k].float() - dirty[k].float()).abs().max()))


def test_hierarchical_and_view_direction_paths_do_not_depend_on_register_contents():
    """The same dirty-register check for the kernels the first test does not reach: the fine-sampling kernel and the fine pass
    (hier_sampling=True), the per-ray view-direction bias (include_vd=True), the gaze-extended shape code, and the frozen-network
    backward of single-image fitting -- at a sample count with dead lanes (N_c = 20, N_f = 13: 33 planes)."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    from n3dt.train import data_losses, disk_mask
    opt = BaseOptions({"featmap_size": 10, "featmap_nc": 256, "pred_img_size": 40, "num_sample_coarse": 20, "num_sample_fine": 13})

    def everything():
        out = {}
        for name, kw, sdkw in (("hier", dict(include_vd=False, hier_sampling=True), dict(hier_sampling=True)),
                               ("vd", dict(include_vd=True, hier_sampling=False), dict(include_vd=True)),
                               ("gaze", dict(include_vd=False, hier_sampling=False, include_gaze=True, eye_gaze_dim=64),
                                dict(include_gaze=True, eye_gaze_dim=64))):
            d = to_dev(syn.frame_inputs(opt, 2, **{k: v for k, v in sdkw.items() if k in ("include_gaze", "eye_gaze_dim")}))
            a = (d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"], d["batch_Tvecs"],
                 d["batch_inv_inmats"])
            for prec in ("bf16", "fp32"):
                net = HeadNeRFNet(opt, precision=prec, **kw).to(dev())
                net.load_state_dict(syn.make_state_dict(opt, seed=4, bg_noise=0.1, **sdkw), strict=True)
                with torch.no_grad():
                    o = net("test", *a)
                for part in o:
                    out["%s_%s_%s" % (name, prec, part)] = o[part]["merge_img"].clone()
            # fitting: frozen network, gradients to codes and cameras (exact fp32 path)
            net = HeadNeRFNet(opt, train_precision="fp32", **kw).to(dev())
            net.load_state_dict(syn.make_state_dict(opt, seed=4, bg_noise=0.1, **sdkw), strict=True)
            for p in net.parameters():
                p.requires_grad_(False)
            leaf = {k: d[k].clone().requires_grad_(True) for k in ("shape_code", "appea_code", "batch_Rmats", "batch_Tvecs")}
            o = net("test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, leaf["shape_code"], leaf["appea_code"], leaf["batch_Rmats"],
                    leaf["batch_Tvecs"], d["batch_inv_inmats"])
            last = o["fine_dict"] if "fine_dict" in o else o["coarse_dict"]
            t = data_losses(last, torch.full_like(last["merge_img"], 0.5), disk_mask(2, 40).to(dev()))
            (t["head_loss"] + t["nonhead_loss"]).backward()
            for k, v in leaf.items():
                out["%s_d_%s" % (name, k)] = v.grad.detach().clone()
        torch.cuda.synchronize()
        return out
    clean = everything()
    _dirty_the_register_files()
    dirty = everything()
    for k in clean:
        assert torch.isfinite(dirty[k]).all(), k
        if "_d_" in k:
            assert float((clean[k] - dirty[k]).abs().max()) <= 1e-3 * float(clean[k].abs().max()) + 1e-12, k   # fp32 atomics
        else:
            assert torch.equal(clean[k], dirty[k]), (k, float((clean[k] - dirty[k]).abs().max()))
