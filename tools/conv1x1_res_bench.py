import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/agrl.pytorch_amd") else os.getcwd()
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid._hip import LP_DTYPE
dev = "cuda:0"
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3
for k, cout in ((256, 1024), (512, 2048)):
    x = torch.relu(torch.randn((256, 16, 8, k), device=dev)).to(LP_DTYPE)
    res = torch.relu(torch.randn((256, 16, 8, cout), device=dev)).to(LP_DTYPE)
    w = (torch.randn((cout, 1, 1, k), device=dev) / k ** 0.5).to(LP_DTYPE)
    b = torch.randn((cout,), device=dev)
    packed = ops.conv1x1_pack(w)
    arms = {"duo": lambda: ops.conv1x1_packed_res(x, packed, b, cout, res), "wide": lambda: ops.conv_bn_act(x, w, b, 1, 0, True, residual=res)}
    same = torch.equal(arms["duo"](), arms["wide"]())
    for _ in range(3):
        for f in arms.values(): f()
    torch.cuda.synchronize()
    t = {n: [] for n in arms}
    for _ in range(30):
        for n, f in arms.items(): t[n].append(timed(f))
    print("conv1x1 %d->%d + res equal %s  " % (k, cout, same) + "  ".join("%s %.1f us (min %.1f)" % (n, sorted(v)[len(v) // 2], min(v)) for n, v in t.items()))
