"""Does the whole eval forward capture into a HIP graph (torch.cuda.CUDAGraph drives hipStreamBeginCapture; every C-ABI
entry point launches on the capturing stream), and what does replay cost against eager launches?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from torchreid._hip import LP_NAME
from bench import build_model
from recipe import synthetic_adj
dev = torch.device("cuda:0")
model, _ = build_model(dev, LP_NAME)
B, S = 32, 8
x = torch.randn((B, S, 3, 256, 128), device=dev)
adj = synthetic_adj(B, S).to(dev)
for _ in range(3):
    ref = model(x, adj)
torch.cuda.synchronize()
def timeit(fn, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
t_eager = timeit(lambda: model(x, adj))
# host-side cost of issuing one eager forward (no sync)
t0 = time.perf_counter(); model(x, adj); t_issue = (time.perf_counter() - t0) * 1e3; torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    model(x, adj)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = model(x, adj)
g.replay(); torch.cuda.synchronize()
print("graph output equals eager:", bool(torch.equal(out, ref)))
t_graph = timeit(g.replay)
t0 = time.perf_counter(); g.replay(); t_rissue = (time.perf_counter() - t0) * 1e3; torch.cuda.synchronize()
print("eager %.3f ms/forward (host issue %.3f ms) | graph replay %.3f ms/forward (host issue %.3f ms)" % (t_eager, t_issue, t_graph, t_rissue))
