"""conv3 + residual -> next conv1: the back-to-back kernel (csrc/bottleneck_seam.hip) against the two launches it replaces,
interleaved in one process on the bench shapes (256 frames of 16 x 8). usage: seam_bench.py [rounds] [frames]
Prints per shape: median / min of each arm in us (HIP events on the launch stream), TFLOP/s of the fused arm."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid._hip import LP_DTYPE

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3


for cmid, cout, cnext in ((256, 1024, 256), (512, 2048, 512), (256, 1024, 512)):
    y2 = torch.randn((frames, 16, 8, cmid), device=dev).to(LP_DTYPE)
    res = torch.randn((frames, 16, 8, cout), device=dev).to(LP_DTYPE)
    w3 = (torch.randn((cout, 1, 1, cmid), device=dev) / cmid ** 0.5).to(LP_DTYPE)
    w1 = (torch.randn((cnext, 1, 1, cout), device=dev) / cout ** 0.5).to(LP_DTYPE)
    b3, b1 = torch.randn((cout,), device=dev), torch.randn((cnext,), device=dev)

    packed = ops.bottleneck_seam_pack(w3, w1)

    def fused():
        return ops.bottleneck_seam(y2, packed, b3, res, b1, (cmid, cout, cnext))

    def split():
        o = ops.conv_bn_act(y2, w3, b3, 1, 0, True, residual=res)
        return o, ops.conv_bn_act(o, w1, b1, 1, 0, True)

    for _ in range(3):
        fused(), split()
    torch.cuda.synchronize()
    tf, ts = [], []
    for _ in range(rounds):
        tf.append(timed(fused))
        ts.append(timed(split))
    tf.sort(), ts.sort()
    flops = 2.0 * frames * 128 * (cmid * cout + cout * cnext)
    print("seam %4d/%4d/%3d  fused %.1f us (min %.1f)  two launches %.1f us (min %.1f)  fused %.0f TFLOP/s" % (
        cmid, cout, cnext, tf[len(tf) // 2], tf[0], ts[len(ts) // 2], ts[0], flops / tf[len(tf) // 2] * 1e-6))
