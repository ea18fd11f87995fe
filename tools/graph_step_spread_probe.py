#!/usr/bin/env python3
"""Spread of test_whole_training_step_replays_as_one_graph's statistic: after 12 Adam steps from the same weights, data and jitter,
the largest parameter difference between (a) two EAGER runs and (b) an eager and a graphed run.  Adam normalises the update, so a
weight whose gradient is summation-order noise moves by up to lr per step in either direction: the eager-vs-eager figure is the floor
the graphed run is held to.   usage: tools/graph_step_spread_probe.py [N = 6]"""
import os
import sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, os.path.join(R, "nerf-3dtalker-code_amd"))
os.chdir(R)
import torch  # noqa: E402
import test_gpu_round4 as t  # noqa: E402
from n3dt.train import GraphedTrainStep  # noqa: E402


def eager(n_steps):
    net, step, _ = t._train_setup(16, 32, 64, 2, graph=False)
    for _ in range(n_steps):
        loss = step()
    return net, float(loss)


def graphed(n_steps, warm=3):
    net, step, _ = t._train_setup(16, 32, 64, 2, graph=True)
    g = GraphedTrainStep(step, warmup=warm)
    for _ in range(n_steps - warm):
        loss = g()
    torch.cuda.synchronize()
    return net, float(loss)


def worst(a, b):
    w, name = 0.0, ""
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        d = float((p - q).abs().max())
        if d > w:
            w, name = d, n
    return w, name


n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for i in range(n):
    e1, l1 = eager(12)
    e2, l2 = eager(12)
    g1, lg = graphed(12)
    print("run %d: eager vs eager %.3e (%s), loss %.3e apart | eager vs graph %.3e (%s), loss %.3e apart" % (
        (i,) + worst(e1, e2) + (abs(l1 - l2) / abs(l1),) + worst(e1, g1) + (abs(l1 - lg) / abs(l1),)), flush=True)
