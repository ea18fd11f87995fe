#!/usr/bin/env python3
"""How far apart do 200-step training trajectories land?  Runs tests/test_gpu_round3.py's convergence test N times in one process
and prints its own lines (loss at steps 50 / 100 / 200 for two fp32 runs and one bf16 run, image differences bf16-vs-fp32 and
fp32-vs-fp32) plus whether the test's bounds held -- the evidence the bounds of that test are set from.
usage: tools/converge_spread_probe.py [N = 6]"""
import os
import sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, os.path.join(R, "nerf-3dtalker-code_amd"))
os.chdir(R)
import test_gpu_round3 as t  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
fails = 0
for i in range(n):
    print("---- run %d" % i, flush=True)
    try:
        t.test_bf16_and_fp32_training_converge_to_the_same_loss()
        print("bounds held", flush=True)
    except AssertionError as e:
        fails += 1
        print("BOUNDS EXCEEDED: %s" % (str(e).splitlines()[0] if str(e) else "image statistics"), flush=True)
print("%d of %d runs exceeded the bounds" % (fails, n))
