"""Single-image fitting harness (the caller side of SURVEY 8f-1; reference: FittingSingleImage_new.py:736-916).

The reference optimises five small tensors THROUGH the renderer for 300 Adam iterations: identity / expression /
appearance offsets added to the base codes, three Euler angles and a translation composed with the base camera
(`R = dR R0`, `T = dR T0 + dT`, :797-803).  Everything here is plain PyTorch on a handful of scalars -- the work is in
`HeadNeRFNet.forward()` with gradients enabled, whose backward (libn3dt) differentiates the cameras and the codes.
"""
import torch


def eulurangle2Rmat(angles):
    """[N,3] Euler angles (x, y, z) -> [N,3,3] rotation Rz Ry Rx, differentiable (FittingSingleImage_new.py:736-766)."""
    n = angles.size(0)
    sx, sy, sz = torch.sin(angles[:, 0]), torch.sin(angles[:, 1]), torch.sin(angles[:, 2])
    cx, cy, cz = torch.cos(angles[:, 0]), torch.cos(angles[:, 1]), torch.cos(angles[:, 2])
    one, zero = torch.ones_like(sx), torch.zeros_like(sx)
    rx = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], dim=-1).view(n, 3, 3)
    ry = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], dim=-1).view(n, 3, 3)
    rz = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], dim=-1).view(n, 3, 3)
    return rz.bmm(ry.bmm(rx))


class FittingState:
    """The optimisation variables of perform_fitting (:826-840) around fixed base codes and a base camera."""

    def __init__(self, base_shape, base_appea, cam_info, opt_cam=True, iden_dims=100):
        dev = base_shape.device
        self.base_shape, self.base_appea, self.cam_info, self.opt_cam, self.iden_dims = base_shape, base_appea, cam_info, opt_cam, iden_dims
        n = base_shape.shape[0]
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev, requires_grad=True)  # noqa: E731
        self.iden_offset = z(n, iden_dims)
        self.expr_offset = z(n, base_shape.shape[1] - iden_dims)
        self.appea_offset = z(n, base_appea.shape[1])
        self.delta_EulurAngles = z(n, 3)
        self.delta_Tvecs = z(n, 3, 1)

    def variables(self):
        v = [self.iden_offset, self.expr_offset, self.appea_offset]
        return v + [self.delta_EulurAngles, self.delta_Tvecs] if self.opt_cam else v

    def build_code_and_cam(self):
        """(code_info, cam_info) for forward(): :772-809."""
        shape_code = self.base_shape + torch.cat([self.iden_offset, self.expr_offset], dim=-1)
        appea_code = self.base_appea + self.appea_offset
        code_info = {"bg_code": None, "shape_code": shape_code, "appea_code": appea_code}
        if not self.opt_cam:
            return code_info, self.cam_info
        dR = eulurangle2Rmat(self.delta_EulurAngles)
        cam = {"batch_Rmats": dR.bmm(self.cam_info["batch_Rmats"]),
               "batch_Tvecs": dR.bmm(self.cam_info["batch_Tvecs"]) + self.delta_Tvecs,
               "batch_inv_inmats": self.cam_info["batch_inv_inmats"]}
        return code_info, cam

    def make_optimizer(self, init_learn_rate=0.01, step_decay=300):
        """Adam with the reference's per-group learning rates and its exponential decay (:843-861)."""
        groups = [{"params": [self.iden_offset], "lr": init_learn_rate * 1.5}, {"params": [self.expr_offset], "lr": init_learn_rate * 1.5},
                  {"params": [self.appea_offset], "lr": init_learn_rate * 1.0}]
        if self.opt_cam:
            groups += [{"params": [self.delta_EulurAngles], "lr": init_learn_rate * 0.1},
                       {"params": [self.delta_Tvecs], "lr": init_learn_rate * 0.1}]
        # (PyTorch's single-kernel implementation of the same update where the variables live on a GPU: five tiny tensors are
        # otherwise ~30 launches per iteration of a loop that is host-bound)
        optimizer = torch.optim.Adam(groups, betas=(0.9, 0.999), fused=self.iden_offset.is_cuda)
        scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda epoch: 0.1 ** (epoch / step_decay))
        return optimizer, scheduler


def fit_step(net, state, optimizer, scheduler, batch_xy, batch_uv, audiostyle, gt_rgb, mask, loss_fn):
    """One iteration of the fitting loop (:865-901): forward("test") WITH gradients -> data terms -> backward -> Adam."""
    code_info, cam_info = state.build_code_and_cam()
    with torch.set_grad_enabled(True):
        pred = net("test", batch_xy, batch_uv, audiostyle, **code_info, **cam_info)
        terms = loss_fn(pred["coarse_dict"], gt_rgb, mask)
        total = terms["total_loss"] if "total_loss" in terms else terms["bg_loss"] + terms["head_loss"] + terms["nonhead_loss"]
    optimizer.zero_grad()
    total.backward()
    optimizer.step()
    scheduler.step()
    return pred, terms, total.detach()
