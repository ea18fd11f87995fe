"""Child process of tests/test_gpu_round4.py::test_rccl_one_rank_group_*: a WORLD-SIZE-1 `nccl` (= RCCL) process group on the
one GPU of the box.  The group is initialised before anything else touches the device.  A one-rank all-reduce is the identity,
so what this exercises is ORDERING: the hand-written backward kernels are launched through ctypes on torch's current stream, the
bucket's collective is launched `async_op=True` from inside backward (an autograd hook) and runs on RCCL's own stream, the
second bucket goes out at wait().  gloo stages through the host and cannot show a stream-ordering bug there."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))


def main():
    out_path = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from n3dt import BaseOptions, HeadNeRFNet, parallel, synthetic as syn
    from n3dt.train import fused_data_losses, disk_mask

    # config 4's training shape: 4 heads, 32 x 32 rays x 64 samples -> 256^2, fused bf16 path
    B = 4
    opt = BaseOptions({"featmap_size": 32, "featmap_nc": 256, "pred_img_size": 256, "num_sample_coarse": 64})
    sd = syn.make_state_dict(opt, seed=0, bg_noise=0.1)
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in syn.frame_inputs(opt, B).items()}
    gt = torch.full((B, 3, 256, 256), 0.5, device=dev)
    mask = disk_mask(B, 256).to(dev)
    t_rand = syn.stratified_noise(B, 1024, 64, seed=3).to(dev)
    net = HeadNeRFNet(opt, False, False, train_precision="bf16").to(dev)
    net.load_state_dict(sd, strict=True)
    bucket = parallel.FlatBucket(numel=1 << 20).to(dev)

    def backward_once():
        for p in list(net.parameters()) + list(bucket.parameters()):
            p.grad = None
        # the co-trained module takes part in the graph the way the reference's LSTM does: its output IS the renderer's
        # audiostyle input, so its gradient is complete AFTER HeadNeRFNet's (talker_trainer.py:1008-1063)
        audio = d["audiostyle"] + 1e-3 * bucket.flat[:B * 64].view(B, 64)
        out = net("train", d["batch_xy"], d["batch_uv"], audio, None, d["shape_code"], d["appea_code"], d["batch_Rmats"],
                  d["batch_Tvecs"], d["batch_inv_inmats"], t_rand=t_rand)
        t = fused_data_losses(out["coarse_dict"], gt, mask)
        t["total_loss"].backward()

    # reference gradients: no reducer
    backward_once()
    torch.cuda.synchronize()
    ref = {n: p.grad.clone() for n, p in net.named_parameters()}
    ref_b = bucket.flat.grad.clone()

    reducer = parallel.GradReducer([net.grad_arena(), bucket.parameters()], world=1, force=True)
    worst, worst_rel = 0.0, 0.0
    steps = 4
    for step in range(steps):
        backward_once()
        reducer.wait()
        # NO synchronize here: wait() must have made the current stream wait for RCCL's, and the comparison kernels below are
        # ordered behind it on that stream
        for n, p in net.named_parameters():
            diff = float((p.grad - ref[n]).abs().max())
            scale = float(ref[n].abs().max())
            worst = max(worst, diff)
            worst_rel = max(worst_rel, diff / (scale + 1e-30))
        worst = max(worst, float((bucket.flat.grad - ref_b).abs().max()))
    arena = net.grad_arena()
    in_arena = sum(arena.is_view(arena.index[id(p)], p.grad) for p in net.parameters())
    rec = {"world": dist.get_world_size(), "backend": dist.get_backend(), "steps": steps, "hook_launches": reducer.hook_launches,
           "last_launch_order": reducer.last_launch_order, "late_rounds": reducer.late_rounds, "worst_abs": worst, "worst_rel": worst_rel,
           "grads_in_arena": in_arena, "n_params": len(list(net.parameters())), "bytes_per_step": reducer.bytes_per_step()}
    reducer.close()
    torch.cuda.synchronize()
    with open(out_path, "w") as f:
        json.dump(rec, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
