"""Times the fused layer-1 bottleneck tail against the two separate convs it replaces."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid._hip import LP_DTYPE
dev = "cuda:0"
N, H, W = 256, 64, 32
y2 = torch.randn((N, H, W, 64), device=dev).to(LP_DTYPE)
res = torch.randn((N, H, W, 256), device=dev).to(LP_DTYPE)
w3 = (torch.randn((256, 1, 1, 64), device=dev) / 8).to(LP_DTYPE)
w1 = (torch.randn((64, 1, 1, 256), device=dev) / 16).to(LP_DTYPE)
b3, b1 = torch.randn(256, device=dev), torch.randn(64, device=dev)
def fused(): return ops.bottleneck_tail(y2, w3, b3, res, w1, b1)
def split():
    o = ops.conv_bn_act(y2, w3, b3, 1, 0, True, residual=res)
    return o, ops.conv_bn_act(o, w1, b1, 1, 0, True)
for name, fn in (("fused", fused), ("split", split)):
    ts = []
    for r in range(8):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(s.elapsed_time(e) * 100)
    us = statistics.median(ts)
    gb = 2.0 * (y2.numel() + 2 * res.numel() + N * H * W * 64) / 1e9
    print("%s %7.1f us  (fused traffic %.0f MB -> %.2f TB/s)" % (name, us, gb * 1e3, gb / us * 1e-3 * 1e3 if False else gb / (us * 1e-6) / 1e3))

# layer-2 form: conv3 128 -> 512 + residual, next conv1 512 -> 128 at 32 x 16
H, W = 32, 16
y2 = torch.randn((N, H, W, 128), device=dev).to(LP_DTYPE)
res = torch.randn((N, H, W, 512), device=dev).to(LP_DTYPE)
w3 = (torch.randn((512, 1, 1, 128), device=dev) / 11).to(LP_DTYPE)
w1 = (torch.randn((128, 1, 1, 512), device=dev) / 22).to(LP_DTYPE)
b3, b1 = torch.randn(512, device=dev), torch.randn(128, device=dev)
for name, fn in (("layer2 fused", fused), ("layer2 split", split)):
    ts = []
    for r in range(8):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        if r >= 2: ts.append(s.elapsed_time(e) * 100)
    us = statistics.median(ts)
    gb = 2.0 * (y2.numel() + 2 * res.numel() + N * H * W * 128) / 1e9
    print("%s %7.1f us  (fused traffic %.0f MB -> %.2f TB/s)" % (name, us, gb * 1e3, gb / (us * 1e-6) / 1e3))

# whole layer-1 block: 3x3 64 -> 64, conv3 64 -> 256 + residual, next conv1 256 -> 64 at 64 x 32
H, W = 64, 32
zin = torch.randn((N, H, W, 64), device=dev).to(LP_DTYPE)
res = torch.randn((N, H, W, 256), device=dev).to(LP_DTYPE)
w2 = (torch.randn((64, 3, 3, 64), device=dev) / 24).to(LP_DTYPE)
w3 = (torch.randn((256, 1, 1, 64), device=dev) / 8).to(LP_DTYPE)
b2, b3 = torch.randn(64, device=dev), torch.randn(256, device=dev)
for cn in (64, 128):
    w1 = (torch.randn((cn, 1, 1, 256), device=dev) / 16).to(LP_DTYPE)
    b1 = torch.randn(cn, device=dev)
    def fusedb(): return ops.bottleneck_block(zin, w2, b2, w3, b3, res, w1, b1)
    def splitb():
        y = ops.conv_bn_act(zin, w2, b2, 1, 1, True)
        return ops.bottleneck_tail(y, w3, b3, res, w1, b1)
    for name, fn in (("block fused cn=%d" % cn, fusedb), ("block 3x3 + tail cn=%d" % cn, splitb)):
        ts = []
        for r in range(8):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): fn()
            e.record(); torch.cuda.synchronize()
            if r >= 2: ts.append(s.elapsed_time(e) * 100)
        us = statistics.median(ts)
        gb = 2.0 * (zin.numel() + 2 * res.numel() + N * H * W * cn) / 1e9
        print("%s %7.1f us  (fused traffic %.0f MB -> %.2f TB/s)" % (name, us, gb * 1e3, gb / (us * 1e-6) / 1e3))
