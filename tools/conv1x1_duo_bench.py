"""Layer 4's conv3 + residual (512 -> 2048 on 256 frames of 16 x 8) and the pool-fused last conv: the two-workgroups-per-CU kernel
(csrc/conv1x1_duo.hip) against igemm_wide_kernel (conv_bn_act(residual=...) / conv1x1_bn_act_pool), interleaved in one process.
usage: conv1x1_duo_bench.py [rounds] [frames] [-] [res,pool4,pool1]
Prints per form: bit equality, median / min of each arm in us (HIP events; ~8 us of launch overhead inside -- run under
tools/kernel_trace.sh for the kernels' own durations), . Before every timed call the block's conv1 and 3x3 conv run (the cache state of the model)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid import _hip
from torchreid._hip import LP_DTYPE

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"
K, Cout = 512, 2048


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3


x = torch.relu(torch.randn((frames, 16, 8, K), device=dev)).to(LP_DTYPE)
res = torch.relu(torch.randn((frames, 16, 8, Cout), device=dev)).to(LP_DTYPE)
w = (torch.randn((Cout, 1, 1, K), device=dev) / K ** 0.5).to(LP_DTYPE)
b = torch.randn((Cout,), device=dev)
packed = ops.conv1x1_pack(w)
# cache state as in the model: before every timed call the block's conv1 (2048 -> 512, reads the residual map) and 3x3 conv run
w1 = (torch.randn((512, 1, 1, Cout), device=dev) / Cout ** 0.5).to(LP_DTYPE)
w2 = (torch.randn((512, 3, 3, 512), device=dev) / (9 * 512) ** 0.5).to(LP_DTYPE)
b1 = torch.randn((512,), device=dev)
p1, p2 = ops.conv1x1_pack(w1), ops.conv3x3_pack(w2)


def before():
    y1 = ops.conv1x1_packed(res, p1, b1, 512, True)
    return ops.conv3x3_packed(y1, p2, b1, 512, True)


flops = 2.0 * frames * 128 * K * Cout
forms = sys.argv[4].split(",") if len(sys.argv) > 4 else ["res", "pool4", "pool1"]
for form in forms:
    if form == "res":
        duo = lambda: ops.conv1x1_packed_res(x, packed, b, Cout, res)
        wide = lambda: ops.conv_bn_act(x, w, b, 1, 0, True, residual=res)
        same = torch.equal(duo(), wide())
    else:
        splits, mean = ([4, 2, 1], True) if form == "pool4" else ([1], False)
        duo = lambda: ops.conv1x1_packed_res_pool(x, packed, b, Cout, res, splits, mean, False)
        wide = lambda: ops.conv1x1_bn_act_pool(x, w, b, res, splits, mean, False)
        same = torch.equal(duo()[0], wide()[0])
    for st in ["-"]:
        for _ in range(3):
            duo(), wide()
        torch.cuda.synchronize()
        td, tw = [], []
        for _ in range(rounds):
            before()
            td.append(timed(duo))
            before()
            tw.append(timed(wide))
        td.sort(), tw.sort()
        print("%-5s %s equal %s  duo %.1f us (min %.1f)  wide %.1f us (min %.1f)  duo %.0f TFLOP/s" % (
            form, st, same, td[len(td) // 2], td[0], tw[len(tw) // 2], tw[0], flops / td[len(td) // 2] * 1e-6))
