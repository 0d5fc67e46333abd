"""Interleaved A/B timing of the GraphLayer Linear (agrl_linear_nobias, 1792 x 2048 x 2048) and the graph-matrix launches.
usage: linear_ab.py "NAME:K=V,..." ..."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops, _hip
from torchreid._hip import LP_DTYPE
dev = "cuda:0"
variants = []
for a in sys.argv[1:]:
    name, _, kv = a.partition(":")
    variants.append((name, dict(x.split("=") for x in kv.split(",") if x)))
if not variants:
    variants = [("default", {})]
keys = sorted({k for _, d in variants for k in d})
B, V, C = 32, 56, 2048
f = torch.randn((B, V, C), device=dev)
flp = f.to(LP_DTYPE)
w = (torch.randn((C, C), device=dev) * 0.01).to(LP_DTYPE)
w32 = w.float()
adj = (torch.rand((B, V, V), device=dev) > 0.5).float()
sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
CASES = [("graph_linear_mix bf16 1792x2048x2048", lambda: ops.graph_linear_mix(flp, w, f, sc, sh, 0.1, 0.1), 2.0 * B * V * C * C),
         ("linear bf16 1792x2048x2048", lambda: ops.linear_nobias(flp.view(B * V, C), w), 2.0 * B * V * C * C),
         ("linear fp32 1792x2048x2048", lambda: ops.linear_nobias(f.view(B * V, C), w32), 2.0 * B * V * C * C),
         ("graph_matrix (gram + finalize)", lambda: ops.graph_matrix(f, adj, True, True), 2.0 * B * V * V * C)]
for label, fn, flops in CASES:
    times = {nm: [] for nm, _ in variants}
    for rnd in range(14):
        for nm, env in variants:
            for k in keys:
                os.environ.pop(k, None)
            os.environ.update(env)
            _hip.reload_options()  # the library reads its switches once; re-read after flipping them
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                fn()
            e.record()
            torch.cuda.synchronize()
            if rnd >= 2:
                times[nm].append(s.elapsed_time(e) * 100)
    print("%-34s " % label + "  ".join("%s %6.1fus %6.1fTF" % (nm, statistics.median(times[nm]), flops / statistics.median(times[nm]) / 1e6) for nm, _ in variants))
