"""Diagnostic: the tiny_train fixture through the fused render kernel's tilings 1 and 2, per library variant (N3DT_LIB).
Prints the worst feature error per frame against the reference fixture and which rays are off."""
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "nerf-3dtalker-code_amd"))
from conftest import load_golden  # noqa: E402

g, m = load_golden("tiny_train")
ref = g["fg_feat"]
for t in ("1", "2"):
    out = "/tmp/tt_%s.npz" % t
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "variant_check.py"), "tiny_train", "bf16", out], check=True,
                   env=dict(os.environ, N3DT_X16_TILING=t, N3DT_X16_TILING2_DIAG="1"), stdout=subprocess.DEVNULL)
    f = np.load(out)["fg_feat"].transpose(0, 2, 1)
    e = np.abs(f - ref).max(axis=1)
    print("lib %s tiling %s: max err per frame %s; rays off (frame 0) %s" % (os.path.basename(os.environ.get("N3DT_LIB", "default")), t, e.max(axis=1),
                                                                         np.nonzero(e[0] > 1e-2)[0][:16]))
