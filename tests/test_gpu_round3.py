"""Round-3 parity additions (through the C ABI, `-m gpu`): free ray sets (reading N of SURVEY 8d: any `batch_xy`, as
NetWorks/utils.py:147-161 takes), the driver line's own parity check, hipGraph replay across weight reloads, the flat
gradient arena of the multi-GPU path on real parameters."""
import os
import sys

import numpy as np
import pytest
import torch

from test_gpu_parity import dev, to_dev, build_net, feats, FEAT_TOL

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FREE_FEAT_TOL = dict(FEAT_TOL, bf16x3=5e-5)


def _free_ray_case(n_x, n_y, batch, n_samples, seed=5):
    """`n_x` x `n_y` rays at SUB-PIXEL positions over an 8 x 8-pixel image plane (so N_r != featmap_size^2 and xy is not an
    integer grid), per-frame yawed cameras, per-frame ray sets (xy differs between the frames)."""
    from n3dt import BaseOptions, synthetic as syn
    opt = BaseOptions({"featmap_size": 8, "featmap_nc": 256, "pred_img_size": 32, "num_sample_coarse": n_samples})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    inp = syn.frame_inputs(opt, batch)
    gen = torch.Generator().manual_seed(seed)
    ix = torch.arange(n_x * n_y) % n_x
    iy = torch.div(torch.arange(n_x * n_y), n_x, rounding_mode="floor")
    xy = torch.stack([(ix.float() + 0.5) * (8.0 / n_x), (iy.float() + 0.5) * (8.0 / n_y)], 0)          # [2, N_r]
    xy = xy.unsqueeze(0) + 0.3 * (torch.rand(batch, 2, n_x * n_y, generator=gen) - 0.5)                # jittered per frame
    inp["batch_xy"] = xy.contiguous()
    inp["batch_uv"] = None
    return opt, sd, inp


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "fp16", "bf16"])
@pytest.mark.parametrize("shape", [(40, 24, 2, 64), (31, 31, 3, 40), (7, 1, 1, 33)])
def test_render_features_on_a_free_ray_set(shape, precision):
    """render_features() with N_r != featmap_size^2 (960, 961 and 7 rays; 64 / 40 / 33 samples, i.e. whole, ragged and
    dead-wave sample blocks) against the CPU oracle's feature stage (`skip_neural_render`), test and train mode."""
    from n3dt import synthetic as syn
    from oracle import oracle as orc
    n_x, n_y, batch, ns = shape
    opt, sd, inp = _free_ray_case(n_x, n_y, batch, ns)
    n_r = n_x * n_y
    assert n_r != opt.featmap_size ** 2
    net = build_net(opt, sd, precision)
    d = to_dev(inp)
    for t_rand in (None, syn.stratified_noise(batch, n_r, ns, seed=11)):
        ref = orc.forward(sd, opt, inp, t_rand=t_rand, skip_neural_render=True)
        f = feats(net, d, None if t_rand is None else t_rand.to(dev()), want_merge=False, want_weight=True)
        assert f["fg_feat"].shape == (batch, n_r, 256) and f["bg_alpha"].shape == (batch, n_r)
        e_f = np.abs(f["fg_feat"].permute(0, 2, 1).cpu().numpy() - ref["fg_feat"]).max()
        e_a = np.abs(f["bg_alpha"].cpu().numpy()[:, None] - ref["bg_alpha"]).max()
        print("free rays %s %s train=%s: fg_feat %.2e bg_alpha %.2e" % (shape, precision, t_rand is not None, e_f, e_a))
        assert e_f <= FREE_FEAT_TOL[precision] and e_a <= FREE_FEAT_TOL[precision]
        w = f["weight"].cpu().numpy()
        np.testing.assert_allclose(w.sum(-1) + f["bg_alpha"].cpu().numpy(), 1.0, atol=2e-5)


def test_forward_refuses_a_free_ray_set():
    """forward() renders an image, so it keeps the reference's implicit contract N_r = featmap_size^2 (the `view` at
    NetWorks/HeadNeRFNet.py:103 would raise there); the error names render_features()."""
    opt, sd, inp = _free_ray_case(5, 3, 1, 16)
    net = build_net(opt, sd, "fp32")
    d = to_dev(inp)
    with pytest.raises(ValueError, match="render_features"):
        net("test", d["batch_xy"], None, d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
            d["batch_Tvecs"], d["batch_inv_inmats"])


def test_bench_parity_check_holds_the_gate():
    """bench.py's own `parity_check` leg (the driver line proves its output): fp32 and bf16x3 within 1e-3 of the oracle on the
    seed-0 AND the sharp-density weights -- run here at config 1's size so the oracle takes seconds."""
    sys.path.insert(0, REPO)
    import bench
    from n3dt import BaseOptions, synthetic as syn

    class Ctx:
        dev = torch.device("cuda:0")

    opt = BaseOptions({"featmap_size": 32, "featmap_nc": 256, "pred_img_size": 256, "num_sample_coarse": 32})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    rec = bench.parity_check(Ctx(), opt, sd, None)
    print(rec)
    assert rec["ok"] and rec["fp32"] <= 1e-4 and rec["bf16x3"] <= 2e-4 and rec["bf16"] <= 1e-3
    assert rec["contrast"]["fp32"] <= 1e-3 and rec["contrast"]["bf16x3"] <= 1e-3
    assert rec["contrast"]["bf16"] > rec["contrast"]["bf16x3"]   # the headline mode is the loose one there, and the line says so


@pytest.mark.parametrize("hier", [False, True])
def test_graph_replay_follows_load_state_dict_and_invalidate_packed(hier):
    """a recorded hipGraph holds the addresses of the packed MLP weights and of the ray-major background
    map.  load_state_dict (post hook), checkpoint loads and broadcast_parameters all call invalidate_packed(): the buffers must
    stay where the graph reads them (re-packed in place), so graphed net -> forward -> load other weights -> forward equals the
    plain net bit for bit, also after a `.data` write + explicit invalidate_packed(), with a varying batch in between."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    opt = BaseOptions({"featmap_size": 16, "featmap_nc": 256, "pred_img_size": 64, "num_sample_coarse": 32, "num_sample_fine": 32})
    sd_a = syn.make_state_dict(opt, seed=0, bg_noise=0.1, hier_sampling=hier)
    sd_b = syn.make_state_dict(opt, seed=5, bg_noise=0.3, hier_sampling=hier)

    def make(use_graph):
        net = HeadNeRFNet(opt, False, hier, precision="bf16", use_graph=use_graph).to(dev())
        net.load_state_dict(sd_a, strict=True)
        return net

    plain, graphed = make(False), make(True)
    keys = ["coarse_dict"] + (["fine_dict"] if hier else [])

    def both(d):
        with torch.no_grad():
            args = (d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                    d["batch_Tvecs"], d["batch_inv_inmats"])
            a, b = plain("test", *args), graphed("test", *args)
        torch.cuda.synchronize()
        for k in keys:
            assert torch.equal(a[k]["merge_img"], b[k]["merge_img"]) and torch.equal(a[k]["bg_img"], b[k]["bg_img"]), k
        return b["coarse_dict"]["merge_img"].clone()

    d2, d1 = to_dev(syn.frame_inputs(opt, 2)), to_dev(syn.frame_inputs(opt, 1))
    first = both(d2)
    packed_before = {k: v[1].data_ptr() for k, v in graphed._pack_cache.items()}
    bg_before = graphed._bg_cache[1].data_ptr()
    for net in (plain, graphed):
        net.load_state_dict(sd_b, strict=True)        # post hook -> invalidate_packed()
    assert {k: v[1].data_ptr() for k, v in graphed._pack_cache.items()} == packed_before, "packed buffers must survive invalidate_packed()"
    assert graphed._bg_cache[1].data_ptr() == bg_before
    second = both(d2)
    assert not torch.equal(first, second) and len(graphed._graphs) == 1
    both(d1)                                           # another batch size in between (the grow-only batch buffers are shared)
    assert torch.equal(second, both(d2))
    with torch.no_grad():                              # the reference trainer's own load_ckpt writes through .data (no version bump)
        for net in (plain, graphed):
            for k, v in net.state_dict().items():
                v.data.copy_(sd_a[k].to(v.device))
            net.invalidate_packed()
    assert torch.equal(first, both(d2)) and len(graphed._graphs) == 2


@pytest.mark.parametrize("train_precision", ["fp32", "bf16"])
def test_gradient_arena_holds_the_same_gradients_as_fresh_buffers(train_precision):
    """The backward kernels accumulate into slices of HeadNeRFNet.grad_arena() (one fill per step; the buffer a data-parallel
    step all-reduces in place).  Same values as with per-call buffers; every .grad is a slice of the arena; a second backward
    without zero_grad accumulates (autograd adds) instead of overwriting; a module applied twice in one graph stays correct."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    from n3dt.train import fused_data_losses, disk_mask
    opt = BaseOptions({"featmap_size": 8, "featmap_nc": 256, "pred_img_size": 32, "num_sample_coarse": 32})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    d = to_dev(syn.frame_inputs(opt, 2))
    gt = torch.full((2, 3, 32, 32), 0.5, device=dev())
    mask = disk_mask(2, 32).to(dev())
    t_rand = syn.stratified_noise(2, 64, 32, seed=3).to(dev())

    def grads(use_arena, passes=1, twice=False):
        net = HeadNeRFNet(opt, False, False, train_precision=train_precision).to(dev())
        net.load_state_dict(sd, strict=True)
        net.use_grad_arena = use_arena
        for _ in range(passes):
            loss = 0.0
            for _ in range(2 if twice else 1):
                out = net("train", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                          d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand)
                t = fused_data_losses(out["coarse_dict"], gt, mask)
                loss = loss + t["bg_loss"] + t["head_loss"] + t["nonhead_loss"]
            loss.backward()
        torch.cuda.synchronize()
        return net, {n: p.grad.clone() for n, p in net.named_parameters()}

    net_a, ga = grads(True)
    _, gf = grads(False)
    arena = net_a.grad_arena()
    n_views = sum(arena.is_view(arena.index[id(p)], p.grad) for p in net_a.parameters())
    assert n_views >= len(list(net_a.parameters())) - 1, "gradients should live in the arena (bg_featmap may be a sum of two)"
    tol = dict(rtol=1e-4, atol=1e-7) if train_precision == "fp32" else dict(rtol=2e-2, atol=1e-5)   # (fp32 atomics: order varies)
    for n in ga:
        torch.testing.assert_close(ga[n], gf[n], **tol, msg=lambda m, n=n: "%s: %s" % (n, m))
    # accumulation over two passes without zero_grad: twice the gradient, not the second pass alone
    _, g2 = grads(True, passes=2)
    _, g2t = grads(True, twice=True)
    for n in ga:
        torch.testing.assert_close(g2[n], 2.0 * gf[n], rtol=max(tol["rtol"], 1e-3), atol=1e-5, msg=lambda m, n=n: "2 passes %s: %s" % (n, m))
        torch.testing.assert_close(g2t[n], 2.0 * gf[n], rtol=max(tol["rtol"], 1e-3), atol=1e-5, msg=lambda m, n=n: "applied twice %s: %s" % (n, m))
    # the next step after zero_grad(set_to_none=True) reuses the arena: gradients equal the first step's again
    for p in net_a.parameters():
        p.grad = None
    out = net_a("train", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand)
    t = fused_data_losses(out["coarse_dict"], gt, mask)
    (t["bg_loss"] + t["head_loss"] + t["nonhead_loss"]).backward()
    for n, p in net_a.named_parameters():
        torch.testing.assert_close(p.grad, gf[n], **tol, msg=lambda m, n=n: "step 2 %s: %s" % (n, m))


def test_bf16_and_fp32_training_converge_to_the_same_loss():
    """convergence equivalence of the fused mixed-precision training path.  200 Adam steps (lr 1e-4, the
    reference's optimizer, talker_trainer.py:722-723) at config 4's geometry (32 x 32 rays x 64 samples -> 256^2, B = 2) from the
    same weights, data and stratified jitter, once with train_precision="fp32" (the exact path pinned to the reference's
    autograd) and once with "bf16": the loss curves stay within 2 % of each other at steps 50 / 100 (measured: 0.01 %) and within
    10 % at step 200 (measured: bf16 4 - 8 % lower, after a 160-fold fall of the loss; two fp32 runs differ by 0.3 - 4 % there) and the images the two
    trained networks render (exact fp32 inference) agree within 1e-2 on average (99 % of the pixels within 0.15: see the end)."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    from n3dt.train import fused_data_losses, disk_mask
    opt = BaseOptions({"featmap_size": 32, "featmap_nc": 256, "pred_img_size": 256, "num_sample_coarse": 64})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    B = 2
    d = to_dev(syn.frame_inputs(opt, B))
    mask = disk_mask(B, 256).to(dev())
    # a target with structure (not a constant): a smooth colour ramp inside the head mask
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 256), torch.linspace(0, 1, 256), indexing="ij")
    gt = torch.stack([0.3 + 0.4 * xx, 0.6 - 0.3 * yy, 0.5 + 0.2 * xx * yy]).unsqueeze(0).repeat(B, 1, 1, 1).to(dev())
    args = lambda: (d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],  # noqa: E731
                    d["batch_Tvecs"], d["batch_inv_inmats"])

    def run(train_precision):
        net = HeadNeRFNet(opt, False, False, precision="fp32", train_precision=train_precision).to(dev())
        net.load_state_dict(sd, strict=True)
        optim = torch.optim.Adam(net.parameters(), lr=1e-4)
        gen = torch.Generator(device=dev()).manual_seed(11)
        curve = []
        for step in range(200):
            t_rand = torch.rand(B, 1024, 65, generator=gen, device=dev())
            out = net("train", *args(), t_rand=t_rand)
            loss = fused_data_losses(out["coarse_dict"], gt, mask)["total_loss"]
            optim.zero_grad()
            loss.backward()
            optim.step()
            curve.append(loss.detach())
        curve = torch.stack(curve).cpu().numpy()
        with torch.no_grad():
            img = net("test", *args())["coarse_dict"]["merge_img"].clone()
        return curve, img

    c32, i32 = run("fp32")
    c32b, i32b = run("fp32")   # the exact path again: its fp32 atomics sum in a different order every run -- the noise floor
    c16, i16 = run("bf16")
    print("loss fp32:", c32[[0, 49, 99, 199]], " fp32 again:", c32b[[0, 49, 99, 199]], " bf16:", c16[[0, 49, 99, 199]])
    assert c32[199] < 0.05 * c32[0], "the fp32 run did not train"
    for k in (49, 99, 199):
        # (mean of five steps around k: the stratified jitter makes single steps noisy.)  After a 150-fold fall of the loss two
        # runs of the SAME fp32 path differ by a few per cent at step 200 (summation order alone); the bf16 run must sit within
        # 2 % of the fp32 runs plus that spread
        a, a2, b = c32[k - 4:k + 1].mean(), c32b[k - 4:k + 1].mean(), c16[k - 4:k + 1].mean()
        ref, spread = 0.5 * (a + a2), abs(a - a2)
        print("step %d: fp32 %.6f / %.6f, bf16 %.6f (%.2f %% off their mean; fp32 spread %.2f %%)" % (k + 1, a, a2, b, 100 * abs(b - ref) / ref, 100 * spread / ref))
        # measured (three boxes): steps 50 / 100 agree to 0.01 %; at step 200, where the loss has fallen 160-fold, the bf16 run
        # is 4 - 8 % BELOW the fp32 runs (0.00448 - 0.00501 against 0.00475 - 0.00505)
        tol = 0.02 if k < 150 else 0.10
        assert abs(b - ref) <= tol * ref + 2.0 * spread, "step %d: fp32 %.6f / %.6f, bf16 %.6f" % (k + 1, a, a2, b)
    def stats(a, b):
        diff = (a - b).abs().flatten()
        q = torch.quantile(diff[::7].float(), torch.tensor([0.5, 0.99], device=diff.device)).tolist()
        return float(diff.mean()), q[0], q[1], float(diff.max())

    s16, s32 = stats(i32, i16), stats(i32, i32b)
    print("final images, bf16 vs fp32 : mean |diff| %.2e, median %.2e, p99 %.2e, max %.2e" % s16)
    print("final images, fp32 vs fp32 : mean |diff| %.2e, median %.2e, p99 %.2e, max %.2e" % s32)
    # Two 200-step trajectories are not pixel-identical even in the SAME arithmetic (second line: fp32 atomics sum in a different
    # order each run): part of the image is still moving fast at step 200 (mask edge).  Eight repetitions of this test in one
    # process (tools/converge_spread_probe.py, profiles/r04_q_converge_spread_8runs.log): bf16 vs fp32 mean 1.8e-3 .. 6.7e-3,
    # p99 3.2e-2 .. 1.0e-1; two fp32 runs against each other mean 3.5e-4 .. 3.7e-3, p99 6.0e-3 .. 7.5e-2 -- the same order, and a
    # "floor" that itself moves tenfold from draw to draw.  (Round 3 bounded the bf16 figure by 5e-3 + 3 x THIS run's fp32 floor;
    # with a low draw of the floor that failed in 1 run of 8.)  Fixed bounds at 1.5 x the worst of those eight runs:
    assert s16[0] <= 1.0e-2 and s16[2] <= 0.15


@pytest.mark.parametrize("train_precision", ["fp32", "bf16"])
def test_frozen_network_backward_gives_the_same_input_gradients(train_precision):
    """Single-image fitting optimises codes and cameras through a FROZEN network (FittingSingleImage_new.py:826-859).  With no
    parameter requiring grad the backward takes the `grads == NULL` route of the C ABI -- no weight-gradient stage, the per-frame
    bias sums of dZ_0 / dZ_5 / the RGB rows from a row-sum pass over the saved tiles -- and must return the same d codes / d cameras
    as the full backward, through the coarse pass and the 2-D renderer."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    from n3dt.train import fused_data_losses, disk_mask
    opt = BaseOptions({"featmap_size": 16, "featmap_nc": 256, "pred_img_size": 64, "num_sample_coarse": 40})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    B = 2
    t_rand = syn.stratified_noise(B, 256, 40, seed=3).to(dev())
    mask = disk_mask(B, 64).to(dev())
    gt = torch.full((B, 3, 64, 64), 0.5, device=dev())
    names = ("audiostyle", "shape_code", "appea_code", "batch_Rmats", "batch_Tvecs")

    def grads(frozen):
        net = HeadNeRFNet(opt, False, False, train_precision=train_precision).to(dev())
        net.load_state_dict(sd, strict=True)
        if frozen:
            for p in net.parameters():
                p.requires_grad_(False)
        d = to_dev(syn.frame_inputs(opt, B))
        for k in names:
            d[k] = d[k].clone().requires_grad_(True)
        out = net("train", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                  d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand)
        fused_data_losses(out["coarse_dict"], gt, mask)["total_loss"].backward()
        torch.cuda.synchronize()
        assert all(p.grad is None for p in net.parameters()) == frozen
        return {k: d[k].grad.clone() for k in names}

    full, froz = grads(False), grads(True)
    for k in names:
        scale = float(full[k].abs().max())
        err = float((full[k] - froz[k]).abs().max()) / scale
        # the row sums are formed in a different order (fp32 atomics either way); everything else is the same arithmetic
        assert err <= 2e-4, "%s: frozen vs full backward differ by %.2e of scale" % (k, err)


def test_standalone_renderer_module_is_differentiable():
    """`net.neural_render(x)` called on its own under autograd (the reference's module signature: x [nb, C, fs, fs]) -- the
    non-split call shape of _NeuralRenderFn: gradient with respect to x against a central finite difference along a random
    direction (exact fp32 path), and the fused bf16 path against the fp32 one for x and every renderer parameter."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    opt = BaseOptions({"featmap_size": 16, "featmap_nc": 256, "pred_img_size": 64, "num_sample_coarse": 8})
    sd = syn.make_state_dict(opt, seed=2, bg_noise=0.1)
    gen = torch.Generator().manual_seed(4)
    x0 = (0.5 * torch.randn(2, 256, 16, 16, generator=gen)).to(dev())
    wgt = torch.randn(2, 3, 64, 64, generator=gen).to(dev())

    def run(train_precision):
        net = HeadNeRFNet(opt, False, False, train_precision=train_precision).to(dev())
        net.load_state_dict(sd, strict=True)
        x = x0.clone().requires_grad_(True)
        img = net.neural_render(x)
        assert img.shape == (2, 3, 64, 64) and img.requires_grad
        (img * wgt).sum().backward()
        torch.cuda.synchronize()
        return net, x.grad.clone(), {n: p.grad.clone() for n, p in net.neural_render.named_parameters() if p.grad is not None}

    net32, gx32, gp32 = run("fp32")
    assert len(gp32) == len(list(net32.neural_render.parameters())) - 1  # every parameter but bg_featmap (not part of this call)
    u = torch.randn(x0.shape, generator=gen).to(dev())
    h = 1e-2
    with torch.no_grad():
        f = lambda xx: float((net32.neural_render(xx) * wgt).double().sum())  # noqa: E731
        numeric = (f(x0 + h * u) - f(x0 - h * u)) / (2 * h)
    analytic = float((gx32 * u).double().sum())
    assert abs(numeric - analytic) <= 2e-2 * abs(analytic) + 1e-4, (numeric, analytic)
    _, gx16, gp16 = run("bf16")
    for name, a, b in [("x", gx32, gx16)] + [(n, gp32[n], gp16[n]) for n in gp32]:
        a, b = a.double().flatten(), b.double().flatten()
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        err = float((a - b).abs().max() / (a.abs().max() + 1e-30))
        # (a random per-pixel loss weight on a random map: the weight gradients cancel heavily, bf16 rounding shows as up to
        # ~14 % of scale on single entries of the first block while the direction holds; realistic losses: tools/fuzz_train_nr.py)
        assert cos >= 0.995 and err <= 0.2, (name, cos, err)
    # the two weight-gradient kernels of the fused path (LDS-transposed row-major front end vs the 2-byte-gather kernel) compute
    # the same bf16 products with fp32 accumulation: same gradients up to summation order
    os.environ["N3DT_NR_DW_LDS"] = "0"
    try:
        _, gx16g, gp16g = run("bf16")
    finally:
        del os.environ["N3DT_NR_DW_LDS"]
    assert torch.equal(gx16, gx16g)
    for n in gp16:
        scale = float(gp16[n].abs().max()) + 1e-30
        assert float((gp16[n] - gp16g[n]).abs().max()) / scale <= 1e-3, n


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(64, 256, 512, 3), (16, 256, 64, 2), (4, 256, 32, 5), (12, 256, 96, 1), (6, 256, 48, 2)])
@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_matrix_pipe_blur_agrees_with_the_per_output_blur(geom, precision):
    """The Blur + LeakyReLU + feat_2_rgb stage of the fused 16-bit renderer has two forms: the 3 x 3 stencil as an MFMA over a
    4 x 8 tile's halo (csrc/nr_blur_mfma.inc, the default where the map sizes allow it) and one thread per pixel x 8 channels
    (`N3DT_NR_BLUR_MFMA=0`, and every shape the first does not take: the 6 x 6 map here).  Same products, fp32 accumulation in a
    different order: the images agree to fp32 rounding (a 16-bit rounding of an activation flips too rarely to show); both sit in
    the 16-bit band around the exact fp32 renderer.  Covers image borders on every side of a tile (4 x 4 maps are ALL border), a map size that is not a power of
    two, several maps per call."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    fs, nc, out, nb = geom
    opt = BaseOptions({"featmap_size": fs, "featmap_nc": nc, "pred_img_size": out, "num_sample_coarse": 8})
    sd = syn.make_state_dict(opt, seed=5, bg_noise=0.1)
    gen = torch.Generator().manual_seed(fs + nb)
    x = (0.7 * torch.randn(nb, fs, fs, nc, generator=gen)).to(dev())  # ray-major maps, as the volumetric stage hands them over

    def run(prec, env):
        net = HeadNeRFNet(opt, False, False, precision=prec).to(dev())
        net.load_state_dict(sd, strict=True)
        if env is not None:
            os.environ["N3DT_NR_BLUR_MFMA"] = env
        try:
            with torch.no_grad():
                img = net.neural_render.render_hwc(x, prec)
            torch.cuda.synchronize()
        finally:
            os.environ.pop("N3DT_NR_BLUR_MFMA", None)
        return img

    ref = run("fp32", None)
    a, b = run(precision, "1"), run(precision, "0")
    assert a.shape == b.shape == ref.shape == (nb, 3, out, out)
    assert bool(torch.isfinite(a).all())
    band = 2e-2 if precision == "bf16" else 3e-3
    ea, eb, d = float((a - ref).abs().max()), float((b - ref).abs().max()), float((a - b).abs().max())
    print("%s %s: mfma vs fp32 %.2e, per-output vs fp32 %.2e, mfma vs per-output %.2e (mean %.2e)" % (geom, precision, ea, eb, d, float((a - b).abs().mean())))
    assert ea <= band and eb <= band
    assert d <= 5e-6 and float((a - b).abs().mean()) <= 5e-7  # (measured 2.4e-7 / 4e-8: exact products, nine-term fp32 sums)


def _random_free_shapes(n, seed):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        n_x, n_y = int(rng.randint(1, 48)), int(rng.randint(1, 40))
        out.append((n_x, n_y, int(rng.choice([1, 2, 3, 5])), int(rng.choice([1, 2, 7, 16, 31, 32, 33, 48, 63, 64, 65, 96]))))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("shape", _random_free_shapes(10, 2026))
def test_render_features_on_random_free_ray_sets(shape):
    """Ten seeded random ray-set shapes (1 .. 1 833 rays per frame, 1 .. 96 samples: single-sample rays, ragged and whole sample
    blocks, ray counts on both sides of every tile and workgroup multiple) in the two parity-grade modes against the oracle's
    feature stage, train-mode jitter included."""
    from n3dt import synthetic as syn
    from oracle import oracle as orc
    n_x, n_y, batch, ns = shape
    opt, sd, inp = _free_ray_case(n_x, n_y, batch, ns, seed=n_x * 97 + n_y)
    n_r = n_x * n_y
    d = to_dev(inp)
    t_rand = syn.stratified_noise(batch, n_r, ns, seed=3)
    ref = orc.forward(sd, opt, inp, t_rand=t_rand, skip_neural_render=True)
    for precision in ("fp32", "bf16x3"):
        net = build_net(opt, sd, precision)
        f = feats(net, d, t_rand.to(dev()), want_merge=False, want_weight=True)
        assert f["fg_feat"].shape == (batch, n_r, 256)
        e_f = np.abs(f["fg_feat"].permute(0, 2, 1).cpu().numpy() - ref["fg_feat"]).max()
        e_a = np.abs(f["bg_alpha"].cpu().numpy()[:, None] - ref["bg_alpha"]).max()
        print("random free rays %s %s: fg_feat %.2e bg_alpha %.2e" % (shape, precision, e_f, e_a))
        # (one or two samples per ray: a sample spans the whole 6-unit slab and alpha = 1 - exp(-sigma * dist) multiplies sigma's
        #  rounding by it -- tools/fuzz_parity.py's alpha band for the split-operand mode)
        tol = FREE_FEAT_TOL[precision] * (2.0 if ns <= 2 and precision == "bf16x3" else 1.0)
        assert e_f <= tol and e_a <= tol
        w = f["weight"].cpu().numpy()
        np.testing.assert_allclose(w.sum(-1) + f["bg_alpha"].cpu().numpy(), 1.0, atol=2e-5)


def _random_hier_cases(n, seed):
    rng = np.random.RandomState(seed)
    return [(int(rng.choice([4, 6, 8, 12, 16])), int(rng.choice([8, 16, 24, 32, 40])), int(rng.choice([8, 16, 33, 48])),
             int(rng.choice([1, 2, 3])), bool(rng.randint(2)), int(rng.randint(1000))) for _ in range(n)]


@pytest.mark.gpu
@pytest.mark.parametrize("case", _random_hier_cases(6, 77))
def test_hierarchical_pass_on_random_geometries(case):
    """SURVEY 8f row 4 away from the two reference fixtures: six seeded random geometries (map 4 .. 16, 8 .. 40 coarse and 8 .. 48
    fine samples incl. counts that leave ragged sample blocks, batch 1 - 3, test and train mode with explicit jitter / inverse-CDF
    draws) of the hierarchical pass in the exact fp32 mode against the oracle's `forward_hier`: the fine sample planes from the
    GPU's own coarse weights, the fine features and both images."""
    from n3dt import BaseOptions, HeadNeRFNet, synthetic as syn
    from oracle import oracle as orc
    fs, nc, nf, B, train, wseed = case
    opt = BaseOptions({"featmap_size": fs, "featmap_nc": 256, "pred_img_size": fs * 4, "num_sample_coarse": nc, "num_sample_fine": nf})
    sd = syn.make_state_dict(opt, seed=wseed, bg_noise=0.1, hier_sampling=True)
    inp = syn.frame_inputs(opt, B)
    n_r = fs * fs
    t_rand = fine_u = None
    if train:
        t_rand = syn.stratified_noise(B, n_r, nc, wseed + 1)
        fine_u = torch.rand(B * n_r, nf + 1, generator=torch.Generator().manual_seed(wseed + 2))
    ref = orc.forward_hier(sd, opt, inp, t_rand=t_rand, fine_u=fine_u)
    net = HeadNeRFNet(opt, include_vd=False, hier_sampling=True, precision="fp32").to(dev())
    net.load_state_dict(sd, strict=True)
    d = to_dev(inp)
    with torch.no_grad():
        out = net("train" if train else "test", d["batch_xy"], d["batch_uv"], d["audiostyle"], bg_code=None, shape_code=d["shape_code"],
                  appea_code=d["appea_code"], batch_Rmats=d["batch_Rmats"], batch_Tvecs=d["batch_Tvecs"],
                  batch_inv_inmats=d["batch_inv_inmats"], t_rand=None if t_rand is None else t_rand.to(dev()),
                  fine_u=None if fine_u is None else fine_u.to(dev()))
    e_c = np.abs(out["coarse_dict"]["merge_img"].cpu().numpy() - ref["coarse_merge_img"]).max()
    e_f = np.abs(out["fine_dict"]["merge_img"].cpu().numpy() - ref["fine_merge_img"]).max()
    print("hier %s: coarse image %.2e, fine image %.2e" % (case, e_c, e_f))
    # (the fine planes are an inverse CDF of the coarse weights: a weight error moves a plane by error / pdf, so the fine image
    #  inherits the coarse pass's rounding amplified -- the band of test_hierarchical_pass)
    assert e_c <= 1e-4 and e_f <= 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(1, 5), (2, 12), (3, 17), (1, 64), (2, 100), (5, 33)])
def test_fused_loss_tail_on_odd_image_sizes(size):
    """n3dt_loss_fwd / n3dt_loss_bwd at image sizes whose pixel count is and is not a multiple of four (the kernels read 16 bytes
    per lane where they can) against the plain PyTorch form of the same three terms (`n3dt.train.data_losses`, itself pinned to
    the reference by the `loss` fixture): values, the kernel's own total, and both image gradients; NaNs in the rendered image,
    mask values at exactly 0.5, a black background."""
    from n3dt.train import data_losses, fused_data_losses
    B, P = size
    gen = torch.Generator().manual_seed(100 * B + P)
    merge = torch.rand(B, 3, P, P, generator=gen)
    merge[0, 1, P // 2, P // 3] = float("nan")
    bg = torch.rand(1, 3, P, P, generator=gen)
    gt = torch.rand(B, 3, P, P, generator=gen)
    mask = torch.rand(B, 1, P, P, generator=gen)
    mask[0, 0, 0, 0] = 0.5
    for bg_value in (1.0, 0.0):
        m1, b1 = merge.clone().to(dev()).requires_grad_(True), bg.clone().to(dev()).requires_grad_(True)
        m2, b2 = merge.clone().to(dev()).requires_grad_(True), bg.clone().to(dev()).requires_grad_(True)
        t = fused_data_losses({"merge_img": m1, "bg_img": b1}, gt.to(dev()), mask.to(dev()), bg_value=bg_value)
        r = data_losses({"merge_img": m2, "bg_img": b2}, gt.to(dev()), mask.to(dev()), bg_value=bg_value)
        for k in ("bg_loss", "head_loss", "nonhead_loss"):
            np.testing.assert_allclose(float(t[k].detach()), float(r[k].detach()), rtol=5e-6)
        ref_total = r["bg_loss"] + r["head_loss"] + r["nonhead_loss"]
        np.testing.assert_allclose(float(t["total_loss"].detach()), float(ref_total.detach()), rtol=5e-6)
        t["total_loss"].backward()
        ref_total.backward()
        np.testing.assert_allclose(m1.grad.cpu().numpy(), torch.nan_to_num(m2.grad, nan=0.0).cpu().numpy(), atol=1e-9, rtol=1e-5)
        np.testing.assert_allclose(b1.grad.cpu().numpy(), b2.grad.cpu().numpy(), atol=1e-9, rtol=1e-5)
