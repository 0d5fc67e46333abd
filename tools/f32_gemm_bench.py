"""Exact-fp32 GEMM rate of the implicit-GEMM kernel per shape (the train step's forward / data-gradient GEMMs): TFLOP/s against
the 155 TFLOP/s the fp32 MFMA sustains (tools/mfma_peak). usage: python tools/f32_gemm_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "agrl.pytorch_amd")]
import torch
from torchreid import hip_ops as ops
dev = torch.device("cuda:0")
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for split in (False, True):
    print("split-bf16" if split else "exact fp32")
    with ops.f32_split(split):
        for (M, K, N) in [(8192, 8192, 8192), (32768, 2048, 512), (32768, 512, 2048), (32768, 1024, 2048), (131072, 256, 1024), (131072, 1024, 256),
                          (524288, 64, 256), (524288, 256, 64), (32768, 4608, 512)]:
            x = torch.randn((M, K), device=dev); w = torch.randn((N, K), device=dev)
            t = timeit(lambda: ops.linear_nobias(x, w))
            print("  linear %7d x %5d x %5d  %9.1f us %6.1f TF/s" % (M, K, N, t, 2.0 * M * K * N / t / 1e6))
        for (F_, H, W, Cin, Cout) in [(256, 16, 8, 512, 512), (256, 16, 8, 256, 256), (256, 32, 16, 128, 128), (256, 64, 32, 64, 64)]:
            x = torch.randn((F_, H, W, Cin), device=dev); w = torch.randn((Cout, 3, 3, Cin), device=dev)
            t = timeit(lambda: ops.conv_bn_act(x, w, None, 1, 1, False))
            print("  conv3x3 %3dx%-3d %4d -> %4d     %9.1f us %6.1f TF/s" % (H, W, Cin, Cout, t, 2.0 * F_ * H * W * Cin * Cout * 9 / t / 1e6))
