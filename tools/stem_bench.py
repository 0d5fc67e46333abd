"""The 16-bit MFMA stem (csrc/stem_mfma.hip) on the bench shape: HIP-event times over rotating inputs (3 x 100 MB of frames: no
input stays in the memory-side cache), algorithmic GB/s. usage: stem_bench.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda:0"
xs = [torch.randn((256, 3, 256, 128), device=dev) for _ in range(3)]
w = torch.randn((64, 7, 7, 3), device=dev) * 0.05
b = torch.randn(64, device=dev)
wp = ops.pack_stem_weights_lp16(w)
for i in range(3):
    out = ops.stem_lp16(xs[i], wp, b)
torch.cuda.synchronize()
ts = []
for r in range(rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = ops.stem_lp16(xs[r % 3], wp, b)
    e1.record()
    e1.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
nbytes = xs[0].numel() * 4 + out.numel() * 2
print("stem 256 x 3 x 256 x 128 -> 256 x 64 x 32 x 64: median %.1f us (min %.1f) = %.2f TB/s algorithmic (%.0f MB)" % (
    ts[len(ts) // 2], ts[0], nbytes / ts[len(ts) // 2] / 1e6, nbytes / 1e6))
