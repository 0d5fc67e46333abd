"""Frame-resident layer-3 Bottleneck (agrl_bottleneck_frame) against the three separate launches: interleaved timing."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
dev = "cuda:0"
Fr, Cin = 256, 1024
x = torch.randn((Fr, 16, 8, Cin), device=dev).relu().bfloat16()
w1 = (torch.randn((256, 1, 1, Cin), device=dev) / Cin ** 0.5).bfloat16()
w2 = (torch.randn((256, 3, 3, 256), device=dev) / (9 * 256) ** 0.5).bfloat16()
w3 = (0.5 * torch.randn((Cin, 1, 1, 256), device=dev) / 16).bfloat16()
b1, b2, b3 = torch.randn(256, device=dev), torch.randn(256, device=dev), torch.randn(Cin, device=dev)
def fused(): return ops.bottleneck_frame(x, w1, b1, w2, b2, w3, b3)
def separate():
    a = ops.conv_bn_act(x, w1, b1, 1, 0, True)
    a = ops.conv_bn_act(a, w2, b2, 1, 1, True)
    return ops.conv_bn_act(a, w3, b3, 1, 0, True, residual=x)
times = {"fused": [], "separate": []}
for rnd in range(8):
    for name, fn in (("fused", fused), ("separate", separate)):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        if rnd: times[name].append(s.elapsed_time(e) * 100)
fl = 2.0 * Fr * 128 * (Cin * 256 * 2 + 9 * 256 * 256)
for k, v in times.items():
    print("%-9s %7.1f us  %5.0f TFLOP/s" % (k, statistics.median(v), fl / statistics.median(v) / 1e6))
