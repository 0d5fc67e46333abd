#!/usr/bin/env python
"""Per-launch timing of one forward step (HIP events on the launch stream): one line per C-ABI call with its
GEMM shape, duration, achieved TFLOP/s and algorithmic GB/s. Usage: python tools/profile_layers.py [bf16|fp32]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bench import build_model  # noqa: E402
from recipe import synthetic_adj  # noqa: E402
from torchreid import _hip, hip_ops as ops  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else _hip.LP_NAME
B, S = 32, 8
dev = torch.device("cuda:0")
model, _ = build_model(dev, prec)
x = torch.randn((B, S, 3, 256, 128), device=dev)
adj = synthetic_adj(B, S).to(dev)
for _ in range(3):
    model(x, adj)
torch.cuda.synchronize()

# wrap conv to remember shapes
shapes = []
orig = ops.conv_bn_act


def conv_logged(x_, w_, b_, stride, pad, relu, residual=None, **kw):
    shapes.append(("conv %dx%d s%d %4d->%4d @%dx%d%s" % (w_.shape[1], w_.shape[2], stride, w_.shape[3], w_.shape[0],
                                                        x_.shape[1], x_.shape[2], " +res" if residual is not None else "")))
    return orig(x_, w_, b_, stride, pad, relu, residual, **kw)


import torchreid.models._vmgn_hip as eng  # noqa: E402
eng.ops.conv_bn_act = conv_logged
reps = 5
tot = {}
for r in range(reps):
    shapes.clear()
    _hip.PROFILE = []
    model(x, adj)
    torch.cuda.synchronize()
    prof, _hip.PROFILE = _hip.PROFILE, None
    ci = 0
    for i, (name, s, e, tag) in enumerate(prof):
        label = name
        if name in ("agrl_conv2d_bn_act", "agrl_conv2d_bn_act_split16"):
            label = shapes[ci]
            ci += 1
        key = (i, label)
        t = tot.setdefault(key, {"ms": 0.0, "tag": tag})
        t["ms"] += s.elapsed_time(e)
print("%-3s %-46s %9s %9s %9s" % ("#", "call", "us", "TFLOP/s", "GB/s"))
total = 0.0
for (i, label), t in sorted(tot.items()):
    us = 1e3 * t["ms"] / reps
    total += us
    tag = t["tag"]
    tf = "%9.1f" % (tag["flops"] / (us * 1e-6) / 1e12) if tag else "%9s" % "-"
    gb = "%9.0f" % (tag["bytes"] / (us * 1e-6) / 1e9) if tag else "%9s" % "-"
    print("%-3d %-46s %9.1f %s %s" % (i, label, us, tf, gb))
print("total %.1f us" % total)
