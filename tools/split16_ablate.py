"""Timing of the in-loop split-fp16 3x3 convs (pre-split operands) under the ablation library's switches:
usage (GPU box): make -C agrl.pytorch_amd/csrc ABLATE=1; AGRL_HIP_LIB=agrl.pytorch_amd/lib/libagrl_hip_ablate.so AGRL_IGEMM_DBG=<bits> python tools/split16_ablate.py
bits: 8 no steady-state DMA, 32 no DMA waits, 1 no global stores, 2 no residual loads (results wrong by design)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
dev = "cuda:0"
for (H, W, Cin, Cout, R, stride) in ((64, 32, 64, 64, 3, 1), (32, 16, 128, 128, 3, 1), (64, 32, 256, 64, 1, 1), (64, 32, 64, 256, 1, 1), (32, 16, 128, 512, 1, 1), (32, 16, 512, 256, 1, 1)):
    x = torch.randn((256, H, W, Cin), device=dev).clamp(min=0)
    w = ops.split16_inloop_weights((torch.randn((Cout, R, R, Cin), device=dev) / (Cin * R * R) ** 0.5))
    b = torch.randn((Cout,), device=dev)
    pre = R == 3
    if pre:
        w1 = ops.split16_inloop_weights(torch.randn((Cin, 1, 1, Cin), device=dev) / Cin ** 0.5)
        x = ops.conv_bn_act(x, w1, torch.zeros(Cin, device=dev), 1, 0, True, out_presplit=True)
    res = torch.randn((256, H, W, Cout), device=dev) if (R == 1 and Cout > Cin) else None     # conv3 of a Bottleneck: + residual
    for _ in range(3):
        ops.conv_bn_act(x, w, b, stride, R // 2, True, residual=res, x_presplit=pre, out_presplit=pre)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.conv_bn_act(x, w, b, stride, R // 2, True, residual=res, x_presplit=pre, out_presplit=pre)
    e.record()
    torch.cuda.synchronize()
    print("DBG=%s  %dx%d %d->%d R%d: %.1f us" % (os.environ.get("AGRL_IGEMM_DBG", "0"), H, W, Cin, Cout, R, s.elapsed_time(e) * 100))
