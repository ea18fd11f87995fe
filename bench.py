#!/usr/bin/env python3
"""Headline benchmark: rendered frames/s of the head-render hot path on N MI355X GPUs.

Workload (BASELINE.json configs[1], reference-faithful reading):
  one step = forward("test") of `--batch` synthetic heads per GPU: 64x64 rays x 64 samples/ray
  -> latent-conditioned MLP (fused HIP kernel) -> alpha compositing -> 2-D neural renderer ->
  512x512 RGB, plus the background image (the reference renders both every forward).
Inputs and weights are synthetic (seeded) and resident in HBM before the timed region.
Frames shard across ranks with no data-path collective (weak scaling: fixed frames per GPU).

Contract: python bench.py --gpus N --steps K --warmup W  -> rank 0 prints ONE JSON line.
Extra objects in that line:
  roofline      the fused MLP kernel: algorithmic FLOP (2 702 592 per sample point)
                / its average launch time (hipEvents recorded around that launch on its stream,
                inside the timed region) against the dense bf16 MFMA peak.
  cpu_baseline  the CPU restatement (oracle/, an OpenMP port -- the reference's Python cannot
                travel to the GPU box) timed on this host's cores on one frame of the workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
sys.path.insert(0, REPO)

FLOP_PER_POINT = 2702592          # latent-folded MLP query, SURVEY 8(d)
FLOP_PER_POINT_EXECUTED = 2375 * 32768 // 32  # MFMA work actually issued per point: 2280 weight pieces + 95 bias MFMAs per 32 samples (RGB_layer_0 merged into RGB_layer_1, RGB_layer_2 per ray)
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}  # dense, MI355X_MICROARCH.md


def bench_train(args, net, opt, d, dev, world, rank):
    """BASELINE config 3: one reference-shaped train step = forward("train") -> 3 MSE terms -> backward -> Adam
    (+ one flat-buffer RCCL all-reduce of the gradients when world > 1).  Exact-fp32 kernels."""
    import torch
    import torch.distributed as dist
    from n3dt import parallel
    from n3dt.train import fused_data_losses as data_losses, disk_mask
    B = d["batch_xy"].shape[0]
    net.precision = "fp32"
    tp = "bf16" if args.precision == "bf16" else "fp32"
    net.train_precision = tp
    net.neural_render.train_precision = tp
    optim = torch.optim.Adam(net.parameters(), lr=1e-4)
    gt = torch.full((B, 3, opt.pred_img_size, opt.pred_img_size), 0.5, device=dev)
    mask = disk_mask(B, opt.pred_img_size).to(dev)
    if world > 1:
        parallel.broadcast_parameters(net)

    def step():
        pred = net("train", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"],
                   d["batch_Rmats"], d["batch_Tvecs"], d["batch_inv_inmats"])
        t = data_losses(pred["coarse_dict"], gt, mask)
        loss = t["bg_loss"] + t["head_loss"] + t["nonhead_loss"]
        optim.zero_grad()
        loss.backward()
        parallel.allreduce_gradients(net.parameters(), world)
        optim.step()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu") if world > 1 else \
        torch.tensor([elapsed], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.destroy_process_group()
    elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "trained frames/sec @%d^2 x %d samples/ray (fwd+bwd+Adam)" % (opt.pred_img_size, opt.num_sample_coarse),
            "value": world * B * args.steps / elapsed,
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": tp, "data": "synthetic",
            "config": {"workload": "%s: %d heads/GPU/step, %dx%d rays x %d samples -> %dx%d, 3 MSE terms, Adam" % (
                "cfg3" if args.config == "cfg2" else args.config + "-train", B, opt.featmap_size, opt.featmap_size,
                opt.num_sample_coarse, opt.pred_img_size, opt.pred_img_size),
                       "parallelism": "frames sharded over %d rank(s), one flat gradient all-reduce per step" % world},
        }), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="frames per GPU per step")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--rays", default="R", choices=["R", "N"],
                    help="R: rays = featmap_size^2 (reference-faithful); N: 512^2 rays, feature stage only")
    ap.add_argument("--mode", default="render", choices=["render", "train"],
                    help="render: forward-only (BASELINE config 2, the headline); train: fwd+loss+bwd+Adam (config 3, fp32)")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg4", "cfg5"],
                    help="BASELINE.json geometry: cfg2 (headline: 64x64 rays x 64 samples -> 512^2; cfg3 with --mode train), "
                         "cfg4 (32x32 rays x 64 samples -> 256^2, 4 heads per GPU), cfg5 (HR: 32x32 rays x 96 samples -> 1024^2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.config != "cfg2":
        args.no_cpu_baseline = True  # the reported CPU baseline is the headline workload's
        if args.batch == 16:
            args.batch = 4

    import numpy as np
    import torch
    import torch.distributed as dist
    from n3dt import HeadNeRFNet, BaseOptions, synthetic as syn, _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # one process per GPU.  N3DT_DIST_BACKEND=gloo is a rehearsal knob for boxes with fewer GPUs than ranks
    # (ranks then share devices round-robin); the real runs use nccl = RCCL over xGMI.
    backend = os.environ.get("N3DT_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    fs, ns, pred = {"cfg2": (64, 64, 512), "cfg4": (32, 64, 256), "cfg5": (32, 96, 1024)}[args.config]
    opt = BaseOptions({"featmap_size": fs, "featmap_nc": 256, "pred_img_size": pred, "num_sample_coarse": ns})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    net = HeadNeRFNet(opt, include_vd=False, hier_sampling=False, precision=args.precision).to(dev)
    net.load_state_dict(sd, strict=True)
    B = args.batch if args.mode == "render" else min(args.batch, 2)
    n_side = 512 if args.rays == "N" else None
    # every rank renders its own frames: frame indices rank*B .. rank*B+B-1
    inp = syn.frame_inputs(opt, B, n_side=n_side, first_frame=rank * B)
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    n_rays = d["batch_xy"].shape[-1]
    points_per_step = B * n_rays * ns

    if args.mode == "train":
        return bench_train(args, net, opt, d, dev, world, rank)

    def step():
        if args.rays == "R":
            return net("test", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"],
                       d["batch_Rmats"], d["batch_Tvecs"], d["batch_inv_inmats"])
        return net.render_features(d["batch_xy"], d["audiostyle"], d["shape_code"], d["appea_code"], d["batch_Rmats"],
                                   d["batch_Tvecs"], d["batch_inv_inmats"], want_merge=False)

    L = _lib.lib()
    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        L.n3dt_prof_enable(args.steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
    ms = (ctypes.c_float * args.steps)()
    n_rec = ctypes.c_int(0)
    L.n3dt_prof_collect(ms, args.steps, ctypes.byref(n_rec))
    L.n3dt_prof_enable(0)
    kern_ms = float(np.mean([ms[i] for i in range(n_rec.value)])) if n_rec.value else float("nan")

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        frames = world * B * args.steps
        achieved = points_per_step * FLOP_PER_POINT / (kern_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.precision]
        traffic = None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get("%s_%s_b%d" % (args.rays, args.precision, B))
        res = {
            "metric": "rendered frames/sec @%d^2 x %d samples/ray" % (pred, ns),
            "value": frames / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {
                "workload": ("%s-R: %d heads/GPU/step, %dx%d rays x %d samples -> fused MLP+composite -> neural renderer "
                             "-> %dx%d RGB (+ background image)" % (args.config, B, fs, fs, ns, pred, pred)) if args.rays == "R" else
                            ("%s-N: %d heads/GPU/step, 512x512 rays x %d samples, feature stage only" % (args.config, B, ns)),
                "frames_per_gpu_per_step": B, "rays_per_frame": n_rays, "samples_per_ray": ns,
                "parallelism": "frames sharded over %d rank(s), no data-path collective" % world,
            },
            "roofline": {
                "kernel": "nerf_fwd_x16_kernel" if args.precision != "fp32" else "nerf_fwd_f32_kernel",
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": traffic,
                "avg_launch_ms": kern_ms, "points_per_launch": points_per_step,
                "flop_per_point_algorithmic": FLOP_PER_POINT,
                "executed_tflops": points_per_step * FLOP_PER_POINT_EXECUTED / (kern_ms * 1e-3) / 1e12,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as orc
            # the GPU box exposes every host core but grants a 16-CPU share per GPU: more threads only thrash
            avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            cores = max(1, min(avail, int(os.environ.get("N3DT_CPU_THREADS", "16"))))
            orc.set_num_threads(cores)
            one = syn.frame_inputs(opt, 1)
            orc.forward(sd, opt, one, skip_neural_render=(args.rays == "N"))  # warm (page-in, thread pool)
            t1 = time.perf_counter()
            reps = 2
            for _ in range(reps):
                orc.forward(sd, opt, one, skip_neural_render=(args.rays == "N"))
            cpu_s = (time.perf_counter() - t1) / reps
            res["cpu_baseline"] = {
                "value": 1.0 / cpu_s, "unit": "frames/s", "cores": cores, "kind": "port",
                "sample": "%d x 1 frame of the cfg2-R workload (64x64 rays x 64 samples -> 512^2), fp32 OpenMP C restatement" % reps,
            }
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
