"""One shape of conv1x1_duo_kernel per process, for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/duo_traffic.sh):
duo_traffic.py {res|pool|dual4|strided2|strided3|wide_res} -- 6 launches over rotating inputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid._hip import LP_DTYPE
dev = "cuda:0"
case = sys.argv[1]
def t(*shape):
    return [torch.relu(torch.randn(shape, device=dev)).to(LP_DTYPE) for _ in range(3)]
def wgt(cout, k):
    return (torch.randn((cout, 1, 1, k), device=dev) / k ** 0.5).to(LP_DTYPE)
F = 256
if case in ("res", "pool", "wide_res"):
    x, r, w, b = t(F, 16, 8, 512), t(F, 16, 8, 2048), wgt(2048, 512), torch.randn(2048, device=dev)
    pk = ops.conv1x1_pack(w)
    alg = 2 * (x[0].numel() + r[0].numel() + w.numel()), (0 if case == "pool" else 2 * r[0].numel())
    if case == "res": run = lambda i: ops.conv1x1_packed_res(x[i], pk, b, 2048, r[i])
    elif case == "pool": run = lambda i: ops.conv1x1_packed_res_pool(x[i], pk, b, 2048, r[i], [1, 2, 4], True, True)
    else: run = lambda i: ops.conv_bn_act(x[i], w, b, 1, 0, True, residual=r[i])
elif case == "dual4":
    a, y, w, b = t(F, 16, 8, 1024), t(F, 16, 8, 512), wgt(2048, 1536), torch.randn(2048, device=dev)
    pk = ops.conv1x1_pack(w)
    alg = 2 * (a[0].numel() + y[0].numel() + w.numel()), 2 * F * 128 * 2048
    run = lambda i: ops.conv1x1_packed(a[i], pk, b, 2048, True, x2=y[i], duo=True)
elif case in ("strided2", "strided3"):
    hi, wi, k1, k2, co = (64, 32, 256, 128, 512) if case == "strided2" else (32, 16, 512, 256, 1024)
    a, y, w, b = t(F, hi, wi, k1), t(F, hi // 2, wi // 2, k2), wgt(co, k1 + k2), torch.randn(co, device=dev)
    pk = ops.conv1x1_pack(w)
    alg = 2 * (a[0].numel() // 4 + y[0].numel() + w.numel()), 2 * y[0].numel() // k2 * co
    run = lambda i: ops.conv1x1_packed_dual_strided(a[i], y[i], pk, b, co, 2, True)
for i in range(6):
    run(i % 3)
torch.cuda.synchronize()
print("%s algorithmic read %.1f MB write %.1f MB" % (case, alg[0] / 1e6, alg[1] / 1e6))
