"""One whole GraphLayer (vmgn.py:142-172) at the bench shape, old order  Linear -> gram -> finalize -> propagate  against the
commuted order  gram -> finalize -> P = G f -> one GEMM with the BN / LeakyReLU / residual epilogue.  Interleaved, 20 layers
back to back per sample. Run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations.
usage: gcn_layer_bench.py [B V C] [precision]   (AGRL_GRAPH_LINEAR_MMAJOR=1: conv-style XCD map for the A/B)"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid._hip import LP_DTYPE, LP_NAME
dev = "cuda:0"
B, V, C = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 56, 2048)
prec = sys.argv[4] if len(sys.argv) > 4 else LP_NAME
dt = LP_DTYPE if prec == LP_NAME else torch.float32
f = torch.rand((B, 1, C), device=dev) + 0.02 * torch.randn((B, V, C), device=dev)
f_lp = f.to(dt)
w = (torch.randn((C, C), device=dev) * 0.02).to(dt)
adj = (torch.rand((B, V, V), device=dev) > 0.5).float()
sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)


def old():
    h = ops.linear_nobias(f_lp.view(B * V, C), w).view(B, V, C)
    G = ops.graph_matrix(f, adj, True, True)
    return ops.graph_propagate(f, h, G, sc, sh, 0.1, 0.1, want_lp=prec == LP_NAME)[0]


def new():
    G = ops.graph_matrix(f, adj, True, True)
    P = ops.graph_apply_operand(G, f, dt)
    return ops.graph_linear_mix(P, w, f, sc, sh, 0.1, 0.1)


def tracklet():
    P, _ = ops.graph_tracklet_operand(f, adj, True, True, dt)
    return ops.graph_linear_mix(P, w, f, sc, sh, 0.1, 0.1)


with ops.f32_split(prec == "bf16x3"):
    a, b = old(), new()
    print("tracklet form vs three-launch form: %.2e" % ((tracklet() - b).abs().max() / b.abs().max()).item())
    torch.cuda.synchronize()
    print("max rel difference between the two orders: %.2e" % ((a - b).abs().max() / a.abs().max()).item())
    times = {"Linear -> message pass": [], "(G f) W^T, fused epilogue": [], "(G f) W^T, graph + G f per tracklet": []}
    for rnd in range(10):
        for name, fn in (("Linear -> message pass", old), ("(G f) W^T, fused epilogue", new), ("(G f) W^T, graph + G f per tracklet", tracklet)):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                fn()
            e.record()
            torch.cuda.synchronize()
            if rnd >= 2:
                times[name].append(s.elapsed_time(e) * 50)
for k, v in times.items():
    t = statistics.median(v)
    print("B=%d V=%d C=%d %s  %-38s %7.1f us per layer  %6.1f TFLOP/s of the Linear's %.1f GFLOP" % (B, V, C, prec, k, t, 2.0 * B * V * C * C / t / 1e6, 2.0 * B * V * C * C / 1e9))
