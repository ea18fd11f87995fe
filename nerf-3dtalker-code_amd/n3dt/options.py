"""Hyper-parameters of the head-render path.

Mirrors the attribute set the reference's HeadNeRFNet reads from its `opt`
argument (reference: HeadNeRFOptions.py:5-34).  Any object carrying these
attributes is accepted by `n3dt.HeadNeRFNet` (duck-typed), so the reference's
own `BaseOptions` instance can be passed unchanged.
"""


class BaseOptions(object):
    def __init__(self, para_dict=None):
        self.bg_type = "white"

        self.iden_code_dims = 100
        self.expr_code_dims = 79
        self.text_code_dims = 100
        self.illu_code_dims = 27

        self.auxi_shape_code_dims = 179
        self.auxi_appea_code_dims = 127

        self.num_sample_coarse = 64
        self.num_sample_fine = 128

        self.world_z1 = 2.5
        self.world_z2 = -3.5
        self.mlp_hidden_nchannels = 384

        para_dict = para_dict or {}
        self.featmap_size = para_dict.get("featmap_size", 32)
        self.featmap_nc = para_dict.get("featmap_nc", 256)
        self.pred_img_size = para_dict.get("pred_img_size", 256)
        # not in the reference's checkpoint `para`; accepted here so that the
        # synthetic configs can vary the per-ray sample count
        if "num_sample_coarse" in para_dict:
            self.num_sample_coarse = para_dict["num_sample_coarse"]
        if "num_sample_fine" in para_dict:
            self.num_sample_fine = para_dict["num_sample_fine"]

    def para(self):
        """The checkpoint `para` dict (reference: talker_trainer.py:915-936)."""
        return {
            "featmap_size": self.featmap_size,
            "featmap_nc": self.featmap_nc,
            "pred_img_size": self.pred_img_size,
        }
