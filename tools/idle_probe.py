"""Is the training step bound by energy rather than by time?  Insert an idle spin of N device cycles into every step
(torch.cuda._sleep: one thread spinning, the chip otherwise idle) and see by how much the step grows.  Run on the GPU box; it
writes a patched copy of bench.py into gpurun_out/ and runs that.  Measured: 98 us of idle cost 5 - 10 us per step,
490 us cost 430 us, 980 us cost 970 us."""
import os, sys, json, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(REPO, "bench.py")).read()
src = src.replace("        optim.step()\n        if optim_a2s is not None:", "        _c = int(os.environ.get('IDLE_CYCLES', '0'))\n        if _c:\n            torch.cuda._sleep(_c)\n        optim.step()\n        if optim_a2s is not None:", 1)
tmp = os.path.join(REPO, "gpurun_out", "bench_idle.py")
os.makedirs(os.path.dirname(tmp), exist_ok=True)
open(tmp, "w").write(src.replace('REPO = os.path.dirname(os.path.abspath(__file__))', 'REPO = %r' % REPO))
for rep in range(2):
    for cyc in (0, 200000, 1000000, 2000000):
        out = subprocess.run([sys.executable, tmp, "--mode", "train", "--steps", "40", "--warmup", "8", "--no-extras", "--no-cpu-baseline"],
                             env=dict(os.environ, IDLE_CYCLES=str(cyc)), capture_output=True, text=True).stdout.strip().splitlines()[-1]
        print("idle %8d cycles: %.3f ms/step" % (cyc, json.loads(out)["ms_per_step"]), flush=True)
