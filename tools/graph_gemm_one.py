"""Runs the GraphLayer GEMM (agrl_graph_linear_mix, B x 56 x 2048 -> 2048) a few times (for rocprofv3 --pmc). usage: graph_gemm_one.py [B] [K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid._hip import LP_DTYPE
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
V, C, dev = 56, 2048, "cuda:0"
K = int(sys.argv[2]) if len(sys.argv) > 2 else C
f = torch.randn((B, V, C), device=dev)
P = torch.randn((B, V, K), device=dev).to(LP_DTYPE)
w = (torch.randn((C, K), device=dev) * 0.02).to(LP_DTYPE)
sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
for _ in range(5):
    ops.graph_linear_mix(P, w, f, sc, sh, 0.1, 0.1)
torch.cuda.synchronize()
