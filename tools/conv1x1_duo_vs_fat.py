"""Plain 1x1 convs of layer 4 (no residual) through conv1x1_duo_kernel (128 x 256 tiles, two workgroups per CU) against conv1x1_fat_kernel
(256 x 256, one per CU) and igemm_wide_kernel: HIP events, interleaved. usage: conv1x1_duo_vs_fat.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid._hip import LP_DTYPE
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda:0"


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3


for k, cout, hh, ww in ((2048, 512, 16, 8), (1024, 512, 16, 8), (512, 256, 32, 16), (1024, 256, 16, 8)):
    x = torch.relu(torch.randn((256, hh, ww, k), device=dev)).to(LP_DTYPE)
    w = (torch.randn((cout, 1, 1, k), device=dev) / k ** 0.5).to(LP_DTYPE)
    b = torch.randn((cout,), device=dev)
    packed = ops.conv1x1_pack(w)
    arms = {"duo": lambda: ops.conv1x1_packed_res(x, packed, b, cout, None), "fat": lambda: ops.conv1x1_packed(x, packed, b, cout, True),
            "wide": lambda: ops.conv_bn_act(x, w, b, 1, 0, True)}
    same = torch.equal(arms["duo"](), arms["fat"]())
    for _ in range(3):
        for f in arms.values():
            f()
    torch.cuda.synchronize()
    t = {n: [] for n in arms}
    for _ in range(rounds):
        for n, f in arms.items():
            t[n].append(timed(f))
    print("conv1x1 %4d->%4d @%dx%d equal %s  " % (k, cout, hh, ww, same) + "  ".join("%s %.1f us (min %.1f)" % (n, sorted(v)[len(v) // 2], min(v)) for n, v in t.items()))

k1, k2, cout = 1024, 512, 2048
x = torch.relu(torch.randn((256, 16, 8, k1), device=dev)).to(LP_DTYPE)
x2 = torch.relu(torch.randn((256, 16, 8, k2), device=dev)).to(LP_DTYPE)
w = (torch.randn((cout, k1 + k2), device=dev) / (k1 + k2) ** 0.5).to(LP_DTYPE)
b = torch.randn((cout,), device=dev)
packed = ops.conv1x1_pack(w)
arms = {"duo": lambda: ops.conv1x1_packed(x, packed, b, cout, True, x2=x2, duo=True), "fat": lambda: ops.conv1x1_packed(x, packed, b, cout, True, x2=x2)}
same = torch.equal(arms["duo"](), arms["fat"]())
for _ in range(3):
    for f in arms.values():
        f()
torch.cuda.synchronize()
t = {n: [] for n in arms}
for _ in range(rounds):
    for n, f in arms.items():
        t[n].append(timed(f))
print("conv1x1 [1024|512]->2048 equal %s  " % same + "  ".join("%s %.1f us (min %.1f)" % (n, sorted(v)[len(v) // 2], min(v)) for n, v in t.items()))
