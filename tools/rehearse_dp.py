"""Two-rank rehearsal of the data-parallel training step on ONE GPU (gloo backend; the real runs use nccl = RCCL over xGMI):
the config-3 step of bench.py with the two-bucket GradReducer, under torch.profiler.  Rank 0 prints a JSON line with the
all-reduce bytes per step, how many collectives were launched from inside backward, and the number of `aten::cat` /
`aten::copy_` calls per step that touch more than a million elements (the round-2 path concatenated 25 M gradients into a fresh
buffer and copied them back every step: both must be gone).

usage: python tools/rehearse_dp.py            (spawns its two ranks itself)"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
sys.path.insert(0, REPO)


def rank_main():
    import torch
    import torch.distributed as dist
    from torch.profiler import profile, ProfilerActivity
    from n3dt import HeadNeRFNet, BaseOptions, parallel, synthetic as syn
    from n3dt.train import fused_data_losses, disk_mask
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    opt = BaseOptions({"featmap_size": 64, "featmap_nc": 256, "pred_img_size": 512, "num_sample_coarse": 64})
    net = HeadNeRFNet(opt, False, False, train_precision="bf16").to(dev)
    net.load_state_dict(syn.make_state_dict(opt, seed=rank, bg_noise=0.1))
    parallel.broadcast_parameters(net)
    bucket = parallel.FlatBucket().to(dev)
    optim = torch.optim.Adam(net.parameters(), lr=1e-4, fused=True)
    optim2 = torch.optim.Adam(bucket.parameters(), lr=1e-7, betas=(0.5, 0.999))
    reducer = parallel.GradReducer([net.grad_arena(), bucket.parameters()], world)
    B = 2
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.frame_inputs(opt, B, first_frame=rank * B).items()}
    gt = torch.full((B, 3, 512, 512), 0.5, device=dev)
    mask = disk_mask(B, 512).to(dev)

    def step():
        out = net("train", d["batch_xy"], d["batch_uv"], d["audiostyle"], None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                  d["batch_Tvecs"], d["batch_inv_inmats"])
        loss = fused_data_losses(out["coarse_dict"], gt, mask)["total_loss"]
        optim.zero_grad()
        loss.backward()
        bucket.fill_grad(1e-3)
        reducer.wait()
        optim.step()
        optim2.step()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    steps = 3
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
        for _ in range(steps):
            loss = step()
        torch.cuda.synchronize()
    big = {"aten::cat": 0, "aten::copy_": 0}
    for ev in prof.events():
        if ev.name in big and ev.input_shapes:
            first = ev.input_shapes[0]
            shapes = first if (first and isinstance(first[0], (list, tuple))) else [first]   # cat takes a tensor list
            n = 0
            for shp in shapes:
                m = 1
                for s_ in shp or []:
                    m *= int(s_)
                n += m if shp else 0
            if n > 1_000_000:
                big[ev.name] += 1
    # every rank must hold the same averaged gradients afterwards
    csum = torch.tensor([float(net.grad_arena().flat.double().sum())], dtype=torch.float64)
    both = [torch.zeros_like(csum) for _ in range(world)]
    dist.all_gather(both, csum)
    if rank == 0:
        print(json.dumps({
            "world": world, "backend": "gloo (rehearsal on one GPU)", "loss": float(loss),
            "allreduce_bytes_per_step": reducer.bytes_per_step(), "buckets": [a.numel * 4 for a in reducer.arenas],
            "launched_inside_backward_per_step": reducer.hook_launches / (steps + 3 - 1),
            "last_launch_order": reducer.last_launch_order,
            "big_cat_calls_per_step": big["aten::cat"] / steps, "big_copy_calls_per_step": big["aten::copy_"] / steps,
            "gradient_checksums_equal": bool(abs(float(both[0]) - float(both[1])) <= 1e-6 * abs(float(both[0])) + 1e-12),
        }), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if "WORLD_SIZE" in os.environ:
        rank_main()
    else:
        import importlib.util
        spec = importlib.util.spec_from_file_location("n3dt_launch", os.path.join(REPO, "nerf-3dtalker-code_amd", "n3dt", "launch.py"))
        launch = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(launch)
        sys.exit(launch.spawn_ranks([os.path.abspath(__file__)], 2, timeout=600))
