"""Times the --re-rank post-process at the MARS evaluation sizes (m = 1980 queries, n = 12 180 gallery)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid.metrics.distance import hip_distmat_device
dev = "cuda:0"
m, n, D = 1980, 12180, 4096
g = torch.Generator(device=dev).manual_seed(0)
cent = torch.randn((625, D), device=dev, generator=g)
qf = cent[torch.randint(0, 625, (m,), device=dev, generator=g)] + 0.7 * torch.randn((m, D), device=dev, generator=g)
gf = cent[torch.randint(0, 625, (n,), device=dev, generator=g)] + 0.7 * torch.randn((n, D), device=dev, generator=g)
torch.cuda.synchronize(); t0 = time.perf_counter()
qg = hip_distmat_device(qf, gf, "cosine", "fp32"); qq = hip_distmat_device(qf, qf, "cosine", "fp32"); gg = hip_distmat_device(gf, gf, "cosine", "fp32")
torch.cuda.synchronize(); t1 = time.perf_counter()
out = ops.re_ranking(qg, qq, gg)
torch.cuda.synchronize(); t2 = time.perf_counter()
out = ops.re_ranking(qg, qq, gg)
torch.cuda.synchronize(); t3 = time.perf_counter()
print("three distance matrices (fp32): %.1f ms; re_ranking: %.1f ms (first call %.1f ms); peak memory %.2f GB; finite %s"
      % (1e3 * (t1 - t0), 1e3 * (t3 - t2), 1e3 * (t2 - t1), torch.cuda.max_memory_allocated() / 1e9, bool(torch.isfinite(out).all())))
