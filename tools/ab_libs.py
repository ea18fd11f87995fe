"""Interleaved A/B of library builds on one box: runs `bench.py <args>` alternately with each N3DT_LIB and prints ms/step and the
fused kernel's launch time.  usage: ab_libs.py <rounds> <lib1,lib2,...> -- <bench args>"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1])
libs = sys.argv[2].split(",")
args = sys.argv[4:]
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ)
        if l != "default":
            env["N3DT_LIB"] = os.path.join(REPO, l)
        out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--no-extras", "--no-cpu-baseline"] + args, env=env,
                             capture_output=True, text=True)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(l, "FAILED", out.stderr[-400:])
            continue
        d = json.loads(line[-1])
        res[l].append((d["ms_per_step"], (d.get("roofline") or {}).get("avg_launch_ms", float("nan"))))
for l in libs:
    print("%-60s ms/step %s | kernel ms %s" % (l, " ".join("%.3f" % a for a, _ in res[l]), " ".join("%.3f" % b for _, b in res[l])))
