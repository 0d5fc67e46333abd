"""Runs ONE conv shape a few times (for rocprofv3 --pmc). usage: conv_one.py H W Cin Cout R res"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid._hip import LP_DTYPE
H, W, Cin, Cout, R, res = [int(v) for v in sys.argv[1:7]]
N, dev = 256, "cuda:0"
x = torch.randn((N, H, W, Cin), device=dev).to(LP_DTYPE)
w = (torch.randn((Cout, R, R, Cin), device=dev) / (Cin * R * R) ** 0.5).to(LP_DTYPE)
b = torch.randn((Cout,), device=dev)
r = torch.randn((N, H, W, Cout), device=dev).to(LP_DTYPE) if res else None
for _ in range(5):
    ops.conv_bn_act(x, w, b, 1, R // 2, True, r)
torch.cuda.synchronize()
