"""Train-step harness around HeadNeRFNet (the caller side of SURVEY 8a row a12).

The reference's loop is `pred = model("train", ...)` -> data losses -> `backward()` -> two Adam steps
(talker_trainer.py:1008-1067).  The renderer's forward/backward run in libn3dt; the loss terms and the
optimizer are plain PyTorch, exactly as in the reference (they are outside the accelerated path).
"""
import torch
import torch.nn.functional as F


def disk_mask(batch, size, radius=0.35):
    """Synthetic head mask used by the benchmark / fixtures."""
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    r2 = (xx - size / 2.0) ** 2 + (yy - size / 2.0) ** 2
    m = (r2 <= (radius * size) ** 2).float()
    return m.view(1, 1, size, size).repeat(batch, 1, 1, 1)


def data_losses(coarse, gt_rgb, mask, bg_value=1.0):
    """The three MSE data terms of the reference loss (Utils/HeadNeRFLossUtils.py:125-146,196-236); the VGG
    perceptual term needs pretrained weights that cannot be fetched offline and is left to the caller."""
    bg_img = coarse["bg_img"]
    bg_loss = torch.mean((bg_img - bg_value) * (bg_img - bg_value))
    res = torch.nan_to_num(coarse["merge_img"], nan=0.0)
    head = (mask >= 0.5).expand(-1, 3, -1, -1)
    nonhead = (mask < 0.5).expand(-1, 3, -1, -1)
    head_loss = F.mse_loss(res[head], gt_rgb[head])
    tv = res[nonhead] - bg_value
    nonhead_loss = torch.mean(tv * tv)
    return {"bg_loss": bg_loss, "head_loss": head_loss, "nonhead_loss": nonhead_loss}


def make_optimizer(net, lr=1e-4):
    """Adam + StepLR as the reference builds them (talker_trainer.py:722-727)."""
    opt = torch.optim.Adam(net.parameters(), lr=lr)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=10, gamma=0.1)
    return opt, sched


def train_step(net, optimizer, inputs, gt_rgb, mask, t_rand=None, extra_optimizers=()):
    """One reference-shaped step: forward("train") -> losses -> zero_grad -> backward -> step."""
    pred = net("train", inputs["batch_xy"], inputs["batch_uv"], inputs["audiostyle"], bg_code=None,
               shape_code=inputs["shape_code"], appea_code=inputs["appea_code"], batch_Rmats=inputs["batch_Rmats"],
               batch_Tvecs=inputs["batch_Tvecs"], batch_inv_inmats=inputs["batch_inv_inmats"], t_rand=t_rand)
    terms = data_losses(pred["coarse_dict"], gt_rgb, mask)
    total = terms["bg_loss"] + terms["head_loss"] + terms["nonhead_loss"]
    for o in extra_optimizers:
        o.zero_grad()
    optimizer.zero_grad()
    total.backward()
    optimizer.step()
    for o in extra_optimizers:
        o.step()
    terms["total_loss"] = total
    return pred, terms
