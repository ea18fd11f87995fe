"""Scan hipcc's gfx950 assembly of the stream kernels for scalar / flat memory accesses issued after the fragment stream
has started.

The weight stream awaits its LDS fragment reads with COUNTED waits (s_waitcnt lgkmcnt(DEPTH-1), x16_core.h).  Scalar and
flat accesses share that counter and return out of order: they cannot let a wait pass early (the awaited fragment is the
oldest LDS operation in flight), but they lengthen the waits and take issue slots inside the MFMA stream.  hipcc fetches
kernel arguments lazily; kernels pin what they need up front (x16_pin) and this script checks the result.

A second check (scan_inflight) looks for anything that reads or writes a fragment's registers while its LDS read is still
in flight.

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


def _vregs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def scan_inflight(path):
    """Second check: between an inline-asm fragment read (ds_read_b128) and the counted wait that retires it, nothing may
    read or write its destination registers.  hipcc does not know the read is asynchronous: a copy, a spill, or -- when the
    read's result is never consumed -- a re-use of the registers for something else all compile without a word (the last
    one produced wild stores in the fused renderer block).  Straight-line model: branches are ignored, which is exact for
    these kernels' unrolled streams and conservative at loop back edges."""
    bad = []
    name, pending = None, []
    lines = open(path).read().split("\n")
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, pending = m.group(1), []
            continue
        if name is None:
            continue
        t = l.strip()
        if "s_endpgm" in t:
            name = None
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if t.startswith("ds_read_b128") and "ASMSTART" in lines[i - 1]:
            pending.append((_vregs(t.split()[1].rstrip(",")), i))
            continue
        if t.startswith("s_waitcnt"):
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:  # LDS returns in order: all but the newest n reads have landed
                n = int(m.group(1))
                pending = pending[-n:] if n > 0 else []
            continue
        if pending:
            touched = set()
            for tok in re.split(r"[\s,]+", t)[1:]:
                touched |= _vregs(tok)
            for dst, li in pending:
                if touched & dst:
                    bad.append((name, i - li, t))
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
            for name, off, ins in scan_inflight(out):
                print("%s: %s: touches a fragment still in flight (read issued %d lines earlier): %s" % (src, name, off, ins))
                rc = 1
            print("%s: scanned" % src)
    return rc


if __name__ == "__main__":
    sys.exit(main())
