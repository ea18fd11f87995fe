"""A/B of an environment switch on one box: ab_env.py <rounds> VAR=a,b -- <bench args>"""
import json, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1]); var, vals = sys.argv[2].split("="); vals = vals.split(","); args = sys.argv[4:]
res = {v: [] for v in vals}
for r in range(rounds):
    for v in vals:
        env = dict(os.environ); env[var] = v
        out = subprocess.run([sys.executable, os.path.join(R, "bench.py"), "--no-extras", "--no-cpu-baseline"] + args, env=env, capture_output=True, text=True)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        res[v].append(json.loads(line[-1])["ms_per_step"] if line else float("nan"))
for v in vals: print("%s=%s  ms/step %s" % (var, v, " ".join("%.4f" % a for a in res[v])))
