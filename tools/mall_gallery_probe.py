#!/usr/bin/env python
"""Round-5 review, item 4c: is the step's distance matrix faster when the resident gallery (12 180 x 4096 fp16 = 99.8 MB) sits in the
256 MB Infinity Cache? Times agrl_distmat (32 queries, cosine) with HIP events in three states of the memory-side cache:
  cold      a 1 GiB buffer has been streamed through the chip since the gallery was last touched
  warm      the gallery was read once (agrl_diag_read_stream, what a side-stream prefetch under the launch-bound GCN kernels would do)
  in-step   behind one whole forward (the ~4 GB of activation traffic a step moves between two distance matrices)
and the cost of the prefetch pass itself. Usage: python tools/mall_gallery_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from bench import build_model, synthetic_pose_adjacency  # noqa: E402
from torchreid import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(3)
gal = ops.row_l2_normalize(torch.randn((12180, 4096), device=dev, generator=g), True, ops.LP_DTYPE)
q = ops.row_l2_normalize(torch.randn((32, 4096), device=dev, generator=g), True, ops.LP_DTYPE)
out = torch.empty((32, 12180), dtype=torch.float32, device=dev)
flush = torch.empty((256 << 20,), dtype=torch.float32, device=dev)   # 1 GiB
model, _ = build_model(dev, ops.LP_NAME)
clips = torch.randn((32, 8, 3, 256, 128), device=dev, generator=g)
adj = synthetic_pose_adjacency(32, 8, dev, g)


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b)


def dist():
    ops.distmat(q, gal, "cosine", out=out)


def prefetch(wgs=1024):
    ops.read_stream(gal, None, wgs)


res = {"cold": [], "warm": [], "in_step": [], "in_step_prefetched": [], "prefetch_pass": []}
for rep in range(12):
    flush.add_(1.0)
    torch.cuda.synchronize()
    res["cold"].append(timed(dist))
    flush.add_(1.0)
    torch.cuda.synchronize()
    res["prefetch_pass"].append(timed(prefetch))
    res["warm"].append(timed(dist))
    model(clips, adj)
    res["in_step"].append(timed(dist))
    model(clips, adj)
    prefetch()
    res["in_step_prefetched"].append(timed(dist))
for k, v in res.items():
    v = sorted(v[2:])
    print("%-20s median %.1f us  (min %.1f, max %.1f)" % (k, v[len(v) // 2], v[0], v[-1]))
