"""Timing of agrl_rank_topk and agrl_distmat_topk at the MARS full-eval size (1980 x 12180, k = 50).
AGRL_TOPK_RADIX=1 in the environment selects the five-pass radix kernel for the A/B."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid.metrics.distance import hip_distmat_device, hip_distmat_topk_device
dev = "cuda:0"
m, n, D, k = 1980, 12180, 4096, 50
g = torch.Generator(device=dev).manual_seed(1)
q, gal = torch.randn((m, D), device=dev, generator=g), torch.randn((n, D), device=dev, generator=g)


def timed(fn, reps=20):
    ts = []
    for _ in range(reps + 3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return statistics.median(ts[3:])


for prec in (ops.LP_NAME, "fp32"):
    d = hip_distmat_device(q, gal, "cosine", prec)
    t_topk = timed(lambda: ops.rank_topk(d, k))
    t_sep = timed(lambda: ops.rank_topk(hip_distmat_device(q, gal, "cosine", prec), k))
    t_fused = timed(lambda: hip_distmat_topk_device(q, gal, "cosine", k, prec))
    print("%s: rank_topk %.1f us (%.2f TB/s of the 96.5 MB matrix) | normalise + distmat + topk %.1f us | normalise + distmat_topk %.1f us%s" % (
        prec, t_topk, 4.0 * m * n / t_topk / 1e6, t_sep, t_fused, "  [radix kernel forced]" if os.environ.get("AGRL_TOPK_RADIX") else ""))
for rows, cols in ((256, 1523), (32, 12180), (1980, 400)):
    d = torch.randn((rows, cols), device=dev, generator=g)
    print("rank_topk %d x %d: %.1f us" % (rows, cols, timed(lambda: ops.rank_topk(d, k))))
