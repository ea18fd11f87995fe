"""Scan hipcc's gfx950 assembly of the stream kernels for scalar / flat memory accesses issued after the fragment stream
has started.

The weight stream awaits its LDS fragment reads with COUNTED waits (s_waitcnt lgkmcnt(DEPTH-1), x16_core.h).  Scalar and
flat accesses share that counter and return out of order: they cannot let a wait pass early (the awaited fragment is the
oldest LDS operation in flight), but they lengthen the waits and take issue slots inside the MFMA stream.  hipcc fetches
kernel arguments lazily; kernels pin what they need up front (x16_pin) and this script checks the result.

A second check (scan_inflight) looks for anything that reads or writes a fragment's registers while its LDS read is still
in flight.

A third check (scan_join_copies) is for a code-generation fault of the compiler itself, found in round 4 in a diagnostic build of
the 4-wave x 64-sample tiling (docs/tuning_log.md): a VGPR -> AGPR copy of a value that is live ACROSS a divergent region placed
at the region's join label BEFORE the `s_or_b64 exec, exec, ...` that restores the lane mask.  The copy then runs for the
region's lanes only and the other lanes of the AGPR keep whatever it held: `dist = 0` of the lanes past N_s was lost that way
(they carried compositing weight, differently from run to run).  No shipped kernel may contain the pattern; it runs over EVERY
kernel of every file, not only the stream kernels.

The scan runs on build/<name>.s, the device assembly the Makefile keeps from the compile that produced the shipped objects
(same FLAGS, -save-temps=obj), for every source file: all tilings and the training kernels included.

usage: python tools/check_smem_hazard.py   or   make -C nerf-3dtalker-code_amd check     (exit code 1 on a finding)
"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "nerf-3dtalker-code_amd")


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


def _in_asm_block(lines, i):
    """line i sits between ;;#ASMSTART and ;;#ASMEND"""
    j = i - 1
    while j >= 0 and i - j < 64:
        if "ASMEND" in lines[j]:
            return False
        if "ASMSTART" in lines[j]:
            return True
        j -= 1
    return False


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
        # every LDS read issued from inline asm: the stream's ds_read_b128 fragments and (round 4) the transposed
        # ds_read_b64_tr_b16 reads of the weight-gradient and matrix-pipe blur kernels (x16_tr_issue; several per asm block)
        if t.startswith("ds_read") and _in_asm_block(lines, i):
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


def scan_join_copies(path):
    """AGPR writes between the join label of a divergent region (a target of s_cbranch_execz) and the EXEC restore that follows
    it: they execute under the region's reduced lane mask.  Returns [(kernel, label, line offset in the kernel, instruction)]."""
    text = open(path).read()
    found = []
    for m in re.finditer(r"^(_Z\w+):", text, re.M):
        end = text.find("s_endpgm", m.end())
        if end < 0:
            continue
        lines = [l.split(";")[0].rstrip() for l in text[m.end():end].split("\n")]
        targets = set(re.findall(r"s_cbranch_execz\s+(\.LBB\d+_\d+)", "\n".join(lines)))
        for i, l in enumerate(lines):
            lab = l.strip().rstrip(":")
            if not l.strip().endswith(":") or lab not in targets:
                continue
            for j in range(i + 1, len(lines)):
                t = lines[j].strip()
                if not t:
                    continue
                if re.match(r"s_(or|mov)_b64 exec\b", t) or re.match(r"s_\w+_saveexec_b64", t) or t.endswith(":") or \
                        t.startswith(("s_cbranch", "s_branch", "s_setpc")):
                    break
                if t.startswith(("v_accvgpr_write", "v_accvgpr_mov")):
                    found.append((m.group(1), lab, j, t))
    return found


def stream_kernels(path):
    """Names of the kernels in an assembly file that contain an inline-asm fragment read (the stream kernels)."""
    names, name = [], None
    lines = open(path).read().split("\n")
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name = m.group(1)
        elif name and "ds_read_b128" in l and "ASMSTART" in lines[i - 1] and name not in names:
            names.append(name)
    return names


def shipped_assembly():
    """build/<name>.s: the device assembly the Makefile keeps from the very compile that produced the shipped objects
    (-save-temps=obj, same FLAGS).  (Re)builds first, so a stale or missing file cannot be scanned by mistake."""
    subprocess.run(["make", "-s", "-j", str(min(8, os.cpu_count() or 1)), "-C", PKG], check=True)
    build = os.path.join(PKG, "build")
    files = sorted(os.path.join(build, f) for f in os.listdir(build) if f.endswith(".s"))
    lib = os.path.join(PKG, "lib", "libn3dt.so")
    for f in files:
        if os.path.getmtime(f) > os.path.getmtime(lib) + 1.0:
            raise RuntimeError("%s is newer than libn3dt.so: the library was not linked from it" % f)
    return files


def check_shipped(verbose=True):
    """Scan every kernel of the shipped build.  Returns (findings, {file: [stream kernel names]})."""
    findings, kernels = [], {}
    for path in shipped_assembly():
        src = os.path.basename(path)[:-2]
        kernels[src] = stream_kernels(path)
        for name, off, ins in scan(path):
            findings.append("%s: %s: scalar/flat access %d lines after the first stream read: %s" % (src, name, off, ins))
        for name, off, ins in scan_inflight(path):
            findings.append("%s: %s: touches a fragment still in flight (read issued %d lines earlier): %s" % (src, name, off, ins))
        for name, lab, off, ins in scan_join_copies(path):
            findings.append("%s: %s: `%s` (%s + %d lines) runs before the EXEC restore of the region it joins" % (src, name, ins, lab, off))
        if verbose:
            print("%s: scanned (%d stream kernels)" % (src, len(kernels[src])))
    return findings, kernels


def main():
    findings, kernels = check_shipped()
    for f in findings:
        print(f)
    if not any(kernels.values()):
        print("no stream kernel found in build/*.s -- the scan did not see the shipped code")
        return 1
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
