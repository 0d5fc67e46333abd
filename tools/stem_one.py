"""Runs the bf16 stem a few times (for rocprofv3 --pmc)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
dev = "cuda:0"
x = torch.randn((256, 3, 256, 128), device=dev)
w = torch.randn((64, 7, 7, 3), device=dev) * 0.05
b = torch.randn(64, device=dev)
wp = ops.pack_stem_weights_bf16(w)
for _ in range(5):
    ops.stem_bf16(x, wp, b)
torch.cuda.synchronize()
