"""PCIe-inclusive rate of the forward: fp32 frames handed over as PINNED host tensors (what the reference's loaders
produce with pin_memory), uploaded by evaluation.device_prefetch on a copy stream under the previous batch's forward,
against the HBM-resident rate. usage: python tools/h2d_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from torchreid._hip import LP_NAME
from bench import build_model
from recipe import synthetic_adj
from torchreid import evaluation
dev = torch.device("cuda:0")
model, _ = build_model(dev, LP_NAME)
B, S, NB = 32, 8, 12
host = [torch.randn((B, S, 3, 256, 128)).pin_memory() for _ in range(3)]
adj = synthetic_adj(B, S).pin_memory()
def batches(n):
    for i in range(n):
        yield host[i % 3], list(range(B)), [0] * B, adj
def run(prefetch):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    evaluation.extract_features(model, batches(NB), prefetch=prefetch)
    torch.cuda.synchronize(); return time.perf_counter() - t0
x_dev = host[0].to(dev); adj_dev = adj.to(dev)
for _ in range(3): model(x_dev, adj_dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(NB): model(x_dev, adj_dev)
torch.cuda.synchronize(); t_res = time.perf_counter() - t0
run(True); run(False)
t_pre, t_sync = run(True), run(False)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); d = host[0].to(dev, non_blocking=True); e.record(); torch.cuda.synchronize()
print("H2D of one 256-frame batch (%.1f MB pinned): %.2f ms = %.1f GB/s" % (host[0].numel() * 4 / 1e6, s.elapsed_time(e), host[0].numel() * 4 / s.elapsed_time(e) / 1e6))
fr = B * S * NB
print("forward only, frames resident in HBM : %7.0f frames/s" % (fr / t_res))
print("with H2D, prefetched on a copy stream: %7.0f frames/s" % (fr / t_pre))
print("with H2D, uploaded in the loop       : %7.0f frames/s" % (fr / t_sync))
