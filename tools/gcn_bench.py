"""The GraphLayer message-pass unit at the bench shape (B = 32, V = 56, C = 2048): one-launch form (agrl_graph_message_pass)
against the three-kernel form (gram + finalize + propagate), interleaved, back-to-back launches."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
dev = "cuda:0"
B, V, C = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 56, 2048)
f = torch.rand((B, 1, C), device=dev) + 0.02 * torch.randn((B, V, C), device=dev)
h = torch.randn((B, V, C), device=dev)
adj = (torch.rand((B, V, V), device=dev) > 0.5).float()
sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
def fused(): ops.graph_message_pass(f, h, adj, sc, sh, 0.1, 0.1, True, True, want_lp=True)
def pose_only(): ops.graph_message_pass(f, h, adj, sc, sh, 0.1, 0.1, True, False, want_lp=True)
def three():
    G = ops.graph_matrix(f, adj, True, True)
    ops.graph_propagate(f, h, G, sc, sh, 0.1, 0.1, want_lp=True)
times = {"one launch": [], "three kernels": [], "one launch, pose graph only (no Gram / hand-off)": []}
for rnd in range(10):
    for name, fn in (("one launch", fused), ("three kernels", three), ("one launch, pose graph only (no Gram / hand-off)", pose_only)):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        if rnd >= 2: times[name].append(s.elapsed_time(e) * 50)
nbytes = 4.0 * (3 * B * V * C + B * V * V)
for k, v in times.items():
    t = statistics.median(v)
    print("%-50s %6.1f us  %5.2f TB/s of the %.1f MB unit" % (k, t, nbytes / t / 1e6, nbytes / 1e6))
