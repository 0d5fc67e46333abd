"""Weight-gradient kernel (agrl_conv_wgrad) per conv shape of the config-4 step (256 frames) against the transposes + NT GEMM it
replaced: us and TFLOP/s (fp32 MFMA peak 157). usage: python tools/wgrad_bench.py [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "agrl.pytorch_amd")]
import torch
from torchreid import hip_ops as ops, _hip
F_ = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
# (H, W, Cin, Cout, R, stride, pad) of the input map
shapes = [(64, 32, 64, 64, 1, 1, 0), (64, 32, 64, 64, 3, 1, 1), (64, 32, 64, 256, 1, 1, 0), (64, 32, 256, 64, 1, 1, 0),
          (64, 32, 256, 128, 1, 1, 0), (64, 32, 128, 128, 3, 2, 1), (32, 16, 128, 512, 1, 1, 0), (64, 32, 256, 512, 1, 2, 0),
          (32, 16, 512, 128, 1, 1, 0), (32, 16, 128, 128, 3, 1, 1), (32, 16, 512, 256, 1, 1, 0), (32, 16, 256, 256, 3, 2, 1),
          (16, 8, 256, 1024, 1, 1, 0), (32, 16, 512, 1024, 1, 2, 0), (16, 8, 1024, 256, 1, 1, 0), (16, 8, 256, 256, 3, 1, 1),
          (16, 8, 1024, 512, 1, 1, 0), (16, 8, 512, 512, 3, 1, 1), (16, 8, 512, 2048, 1, 1, 0), (16, 8, 1024, 2048, 1, 1, 0),
          (16, 8, 2048, 512, 1, 1, 0)]
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
tot_new = tot_old = 0.0
for (H, W, Cin, Cout, R, st, pad) in shapes:
    OH, OW = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
    x = torch.randn((F_, H, W, Cin), device=dev)
    dy = torch.randn((F_, OH, OW, Cout), device=dev)
    fl = 2.0 * F_ * OH * OW * Cout * Cin * R * R
    t_new = timeit(lambda: ops.conv_wgrad(x, dy, (Cout, Cin, R, R), st, pad))
    def old():
        xt = ops.im2col_t(x, R, R, st, pad); dyt = ops.im2col_t(dy, 1, 1, 1, 0)
        return ops.gemm_nt_splitk(dyt, xt)
    t_old = timeit(old)
    tot_new += t_new; tot_old += t_old
    print("%3dx%-3d %4d -> %4d %dx%d/%d  wgrad %8.1f us %6.1f TF/s | transposes + NT GEMM %8.1f us" % (H, W, Cin, Cout, R, R, st, t_new, fl / t_new / 1e6, t_old))
print("sum over the distinct shapes: %.2f ms vs %.2f ms" % (tot_new / 1e3, tot_old / 1e3))
