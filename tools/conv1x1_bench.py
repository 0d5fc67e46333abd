"""1x1 convs of layers 3 / 4: the four-wave packed-weight kernel (csrc/conv1x1_fat.hip) against igemm_wide_kernel (conv_bn_act /
conv1x1_dual), interleaved in one process on the bench shapes (256 frames of 16 x 8). usage: conv1x1_bench.py [rounds] [frames]
Prints per shape: bit equality, median / min of each arm in us (HIP events), TFLOP/s of the packed arm. For per-kernel times without
the event overhead run it under rocprofv3 --kernel-trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid._hip import LP_DTYPE

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
only = int(sys.argv[3]) if len(sys.argv) > 3 else None   # run one shape of the list (PMC passes: one kernel name = one shape)
dev = "cuda:0"


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3


for si, (k1, k2, cout) in enumerate(((2048, 0, 512), (1024, 0, 512), (1024, 512, 2048), (1024, 0, 256), (512, 0, 2048))):
    if only is not None and si != only:
        continue
    x = torch.relu(torch.randn((frames, 16, 8, k1), device=dev)).to(LP_DTYPE)
    x2 = torch.relu(torch.randn((frames, 16, 8, k2), device=dev)).to(LP_DTYPE) if k2 else None
    w = (torch.randn((cout, k1 + k2), device=dev) / (k1 + k2) ** 0.5).to(LP_DTYPE)
    b = torch.randn((cout,), device=dev)
    packed = ops.conv1x1_pack(w)

    def fat():
        return ops.conv1x1_packed(x, packed, b, cout, True, x2=x2)

    def wide():
        if k2:
            return ops.conv1x1_dual(x, x2, w, b, True)
        return ops.conv_bn_act(x, w.view(cout, 1, 1, k1), b, 1, 0, True)

    same = torch.equal(fat(), wide())
    for _ in range(3):
        fat(), wide()
    torch.cuda.synchronize()
    tf, ts = [], []
    for _ in range(rounds):
        tf.append(timed(fat))
        ts.append(timed(wide))
    tf.sort(), ts.sort()
    flops = 2.0 * frames * 128 * (k1 + k2) * cout
    print("conv1x1 %4d+%3d->%4d  equal %s  packed %.1f us (min %.1f)  wide %.1f us (min %.1f)  packed %.0f TFLOP/s" % (
        k1, k2, cout, same, tf[len(tf) // 2], tf[0], ts[len(ts) // 2], ts[0], flops / tf[len(tf) // 2] * 1e-6))
