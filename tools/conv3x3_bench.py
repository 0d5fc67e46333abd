"""3x3 conv of layers 3 / 4: the four-wave packed-weight kernel (csrc/conv3x3_fat.hip) against conv3x3_wide_kernel, interleaved in
one process on the bench shapes (256 frames of 16 x 8). usage: conv3x3_bench.py [rounds] [frames]
Prints per shape: max |diff| between the two, median / min of each arm in us (HIP events), TFLOP/s of the packed arm. For
per-kernel times without the event overhead run it under rocprofv3 --kernel-trace (tools/seam_trace.sh pattern)."""
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


for cin, cout in ((512, 512), (256, 256)):
    x = torch.randn((frames, 16, 8, cin), device=dev).to(LP_DTYPE)
    w = (torch.randn((cout, 3, 3, cin), device=dev) / (9 * cin) ** 0.5).to(LP_DTYPE)
    b = torch.randn((cout,), device=dev)
    packed = ops.conv3x3_pack(w)

    def fat():
        return ops.conv3x3_packed(x, packed, b, cout, True)

    def wide():
        return ops.conv_bn_act(x, w, b, 1, 1, True)

    d = (fat().float() - wide().float()).abs().max().item()
    for _ in range(3):
        fat(), wide()
    torch.cuda.synchronize()
    tf, ts = [], []
    for _ in range(rounds):
        tf.append(timed(fat))
        ts.append(timed(wide))
    tf.sort(), ts.sort()
    flops = 2.0 * frames * 128 * 9 * cin * cout
    print("conv3x3 %3d->%3d  max|packed - wide| %.3g  packed %.1f us (min %.1f)  wide %.1f us (min %.1f)  packed %.0f TFLOP/s" % (
        cin, cout, d, tf[len(tf) // 2], tf[0], ts[len(ts) // 2], ts[0], flops / tf[len(tf) // 2] * 1e-6))
