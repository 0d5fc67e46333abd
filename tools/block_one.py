"""Runs the fused layer-1 block kernel / layer-2 tail a few times (for rocprofv3 --pmc). usage: block_one.py [block|l2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid._hip import LP_DTYPE
dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "block"
N = 256
if which == "block":
    H, W = 64, 32
    z = torch.randn((N, H, W, 64), device=dev).to(LP_DTYPE)
    res = torch.randn((N, H, W, 256), device=dev).to(LP_DTYPE)
    w2 = (torch.randn((64, 3, 3, 64), device=dev) / 24).to(LP_DTYPE)
    w3 = (torch.randn((256, 1, 1, 64), device=dev) / 8).to(LP_DTYPE)
    w1 = (torch.randn((64, 1, 1, 256), device=dev) / 16).to(LP_DTYPE)
    b2, b3, b1 = torch.randn(64, device=dev), torch.randn(256, device=dev), torch.randn(64, device=dev)
    for _ in range(5):
        ops.bottleneck_block(z, w2, b2, w3, b3, res, w1, b1)
else:
    H, W = 32, 16
    y2 = torch.randn((N, H, W, 128), device=dev).to(LP_DTYPE)
    res = torch.randn((N, H, W, 512), device=dev).to(LP_DTYPE)
    w3 = (torch.randn((512, 1, 1, 128), device=dev) / 11).to(LP_DTYPE)
    w1 = (torch.randn((128, 1, 1, 512), device=dev) / 22).to(LP_DTYPE)
    b3, b1 = torch.randn(512, device=dev), torch.randn(128, device=dev)
    for _ in range(5):
        ops.bottleneck_tail(y2, w3, b3, res, w1, b1)
torch.cuda.synchronize()
