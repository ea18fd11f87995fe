"""Scan hipcc's gfx950 assembly of the stream kernels for scalar / flat memory accesses issued after the fragment stream
has started.

The weight stream awaits its LDS fragment reads with COUNTED waits (s_waitcnt lgkmcnt(DEPTH-1), x16_core.h).  Scalar and
flat accesses share that counter and return out of order: they cannot let a wait pass early (the awaited fragment is the
oldest LDS operation in flight), but they lengthen the waits and take issue slots inside the MFMA stream.  hipcc fetches
kernel arguments lazily; kernels pin what they need up front (x16_pin) and this script checks the result.

usage: python tools/check_smem_hazard.py        (compiles the four sources to assembly, ~5 min; exit code 1 on a finding)
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "nerf-3dtalker-code_amd")
SOURCES = ["nerf_fwd_x16", "nerf_fwd_x16b", "train_mlp", "neural_render"]


def scan(path):
    bad = []
    name, first = None, None
    lines = open(path).read().split("\n")
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, first = m.group(1), None
            continue
        if name is None:
            continue
        if "s_endpgm" in l:
            name = None
            continue
        if first is None and "ds_read_b128" in l and "ASMSTART" in lines[i - 1]:
            first = i
        # flat_ accesses count on both wait counters and return out of order as well
        if first is not None and re.search(r"\b(s_(buffer_)?load_|flat_(load|store|atomic))", l):
            bad.append((name, i - first, l.strip()))
    return bad


def main():
    rc = 0
    with tempfile.TemporaryDirectory() as tmp:
        for src in SOURCES:
            out = os.path.join(tmp, src + ".s")
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(REPO, "include"),
                   "-I" + os.path.join(PKG, "csrc"), "-S", "--cuda-device-only", os.path.join(PKG, "csrc", src + ".hip"), "-o", out]
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            for name, off, ins in scan(out):
                print("%s: %s: scalar/flat access %d lines after the first stream read: %s" % (src, name, off, ins))
                rc = 1
            print("%s: scanned" % src)
    return rc


if __name__ == "__main__":
    sys.exit(main())
