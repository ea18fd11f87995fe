"""Ray grid, orbit cameras and the novel-view / morphing sweeps.

Mirrors the caller-side helpers of the reference (Utils/RenderUtils.py:31-50 ray grid, :53-107 orbit
cameras, :110-128 novel views, :130-157 morphing), with one difference that matters on this hardware: the
reference renders its 45 views (or N morph steps) as 45 serial forward() calls of batch 1; here the whole
sweep is ONE batched forward, so a single launch of the fused kernel sees all frames at once.

The reference reads its intrinsics from ConfigFiles/cam_inmat_info_32x32.json, which is not shipped
(SURVEY header); pass `inv_inmat` (3x3) or the synthetic intrinsics of the benchmark are used.
"""
import math

import torch

from . import ops

from . import synthetic


class RenderUtils(object):
    def __init__(self, view_num, device, opt, inv_inmat=None, audio_dim=64):
        self.view_num = view_num
        self.device = device
        self.opt = opt
        self.audio_dim = audio_dim
        self.build_base_info(inv_inmat)
        self.build_cam_info()

    def build_base_info(self, inv_inmat=None):
        fs = self.opt.featmap_size
        xy, uv = synthetic.ray_grid(fs)                      # x = i % w, y = i // w (RenderUtils.py:35-39)
        self.ray_xy = xy.to(self.device)
        self.ray_uv = uv.to(self.device)
        if inv_inmat is None:
            inv_inmat = synthetic.inv_intrinsics(fs, 1)[0]
        else:
            inv_inmat = torch.as_tensor(inv_inmat, dtype=torch.float32).clone()
            inv_inmat[:2, :2] /= (fs / 32.0)                  # the file holds 32x32 intrinsics (RenderUtils.py:48)
        self.inv_inmat = inv_inmat.view(1, 3, 3).to(self.device)

    def build_cam_info(self):
        """view_num cameras on a circle of radius 5.3 at height z = 12, all looking at the origin."""
        tv_z, tv_x = 0.5 + 11.5, 5.3
        radius = math.sqrt((tv_x ** 2 + tv_z ** 2) - tv_z ** 2)
        up = torch.tensor([0.0, -1.0, 0.0], dtype=torch.float64)
        Rs, Ts = [], []
        for i in range(self.view_num):
            angle = 360.0 * i / (self.view_num - 1) if self.view_num > 1 else 0.0   # np.linspace(0, 360, view_num)
            theta = angle / 180.0 * 3.1415926535
            vp = torch.tensor([math.cos(theta) * radius, math.sin(theta) * radius, tv_z], dtype=torch.float64)
            d1 = -vp                                          # towards the origin
            d2 = torch.linalg.cross(up, d1)
            d3 = torch.linalg.cross(d1, d2)
            cols = [d / torch.linalg.norm(d) for d in (d2, d3, d1)]
            Rs.append(torch.stack(cols, dim=1).float())       # columns: right, down, forward
            Ts.append(vp.float().view(3, 1))
        self.Rmats = torch.stack(Rs).to(self.device)          # [V,3,3]
        self.Tvecs = torch.stack(Ts).to(self.device)          # [V,3,1]
        self.cam_info_list = [{"batch_Rmats": self.Rmats[i:i + 1], "batch_Tvecs": self.Tvecs[i:i + 1],
                               "batch_inv_inmats": self.inv_inmat} for i in range(self.view_num)]
        base_r = torch.eye(3)
        base_r[1:, :] *= -1
        base_t = torch.zeros(3, 1)
        base_t[2, 0] = tv_z
        self.base_cam_info = {"batch_Rmats": base_r.view(1, 3, 3).to(self.device),
                              "batch_Tvecs": base_t.view(1, 3, 1).to(self.device), "batch_inv_inmats": self.inv_inmat}

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _to_uint8_list(img):
        """[V,3,P,P] in (0,1) -> list of HxWx3 uint8 arrays, as the reference returns them (RenderUtils.py:123-125)."""
        arr = ops.img_to_uint8(img.detach()).cpu().numpy()
        return [arr[i] for i in range(arr.shape[0])]

    def _audio(self, code_info, n):
        a = code_info.get("audiostyle")
        if a is None:
            a = torch.zeros(1, self.audio_dim, device=self.device)
        return a.expand(n, -1)

    def render_batch(self, net, shape_code, appea_code, audiostyle, Rmats, Tvecs):
        """One batched forward over n = len(Rmats) frames."""
        n = Rmats.shape[0]
        with torch.no_grad():
            pred = net("test", self.ray_xy.expand(n, -1, -1), self.ray_uv.expand(n, -1, -1), audiostyle, bg_code=None,
                       shape_code=shape_code, appea_code=appea_code, batch_Rmats=Rmats, batch_Tvecs=Tvecs,
                       batch_inv_inmats=self.inv_inmat.expand(n, -1, -1))
        return pred["coarse_dict"]["merge_img"]

    def render_novel_views(self, net, code_info):
        """All view_num orbit views of one head in a single launch."""
        n = self.view_num
        img = self.render_batch(net, code_info["shape_code"].expand(n, -1), code_info["appea_code"].expand(n, -1),
                                self._audio(code_info, n), self.Rmats, self.Tvecs)
        return self._to_uint8_list(img)

    def render_morphing_res(self, net, code_info_1, code_info_2, nums):
        """Linear morph between two heads from the base camera, `nums` frames in a single launch."""
        tv = 1.0 - torch.arange(nums, device=self.device, dtype=torch.float32) / max(nums - 1, 1)
        tv = tv.view(nums, 1)
        shape = code_info_1["shape_code"] * tv + code_info_2["shape_code"] * (1 - tv)
        appea = code_info_1["appea_code"] * tv + code_info_2["appea_code"] * (1 - tv)
        a1, a2 = self._audio(code_info_1, 1), self._audio(code_info_2, 1)
        audio = a1 * tv + a2 * (1 - tv)
        img = self.render_batch(net, shape, appea, audio, self.base_cam_info["batch_Rmats"].expand(nums, -1, -1),
                                self.base_cam_info["batch_Tvecs"].expand(nums, -1, -1))
        return self._to_uint8_list(img)
