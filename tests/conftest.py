import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The built libraries are git-ignored: a fresh checkout compiles them once (hipcc cross-compiles gfx950 without a GPU).
    Nothing is built when they are already there (the GPU box receives them with the snapshot)."""
    import shutil
    import subprocess
    lib = os.path.join(REPO, "nerf-3dtalker-code_amd", "lib", "libn3dt.so")
    if not os.path.exists(lib) and shutil.which("make") and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.check_call(["make", "-s", "-j", str(min(8, os.cpu_count() or 1)), "-C", os.path.join(REPO, "nerf-3dtalker-code_amd")])


def load_golden(name):
    data = np.load(os.path.join(GOLDEN, name + ".npz"))
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        manifest = json.load(f)
    return data, manifest


@pytest.fixture(scope="session")
def golden():
    return load_golden


def options_from_manifest(m):
    from n3dt.options import BaseOptions
    d = {"featmap_size": m["featmap_size"], "featmap_nc": m["featmap_nc"],
         "pred_img_size": m["pred_img_size"], "num_sample_coarse": m["num_sample_coarse"]}
    if "num_sample_fine" in m:
        d["num_sample_fine"] = m["num_sample_fine"]
    return BaseOptions(d)


def synthetic_case(m, **kw):
    """Rebuild (opt, state_dict, inputs) of a fixture from its manifest and verify the weight checksum."""
    from n3dt import synthetic as syn
    opt = options_from_manifest(m)
    if m.get("weights_kind") == "contrast":
        sd = syn.contrast_state_dict(opt, seed=m.get("weights_seed", 0))
    else:
        sd = syn.make_state_dict(opt, seed=m.get("weights_seed", 0), bg_noise=m.get("bg_noise", 0.0),
                                 hier_sampling=bool(m.get("hier_sampling", False)), include_vd=bool(m.get("include_vd", False)), **kw)
    cs = syn.state_dict_checksum(sd)
    ref = m.get("weights_checksum")
    if ref is not None and not kw:
        assert np.allclose(cs, ref, rtol=1e-9, atol=1e-6), "synthetic weight generator drifted from the fixture"
    inp = syn.frame_inputs(opt, m.get("batch", 1), yaw_range=m.get("yaw_range", 0.3), **kw)
    return opt, sd, inp
