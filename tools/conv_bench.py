"""Times single conv shapes of the hot path (bf16), optionally with ablation bits (AGRL_IGEMM_DBG)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid._hip import LP_DTYPE
dev = "cuda:0"
SHAPES = [  # N, H, W, Cin, Cout, R, stride, res
    (256, 64, 32, 64, 128, 1, 1, False),
    (256, 64, 32, 64, 256, 1, 1, False),
    (256, 64, 32, 64, 512, 1, 1, False),
    (256, 64, 32, 128, 128, 1, 1, False),
    (256, 64, 32, 128, 128, 1, 1, True),
    (256, 64, 32, 256, 64, 1, 1, False),
    (256, 16, 8, 512, 2048, 1, 1, True),
    (256, 16, 8, 2048, 512, 1, 1, False),
    (256, 16, 8, 512, 512, 3, 1, False),
]
if len(sys.argv) > 1:
    SHAPES = [SHAPES[int(i)] for i in sys.argv[1].split(",")]
for (N, H, W, Cin, Cout, R, stride, res) in SHAPES:
    x = torch.randn((N, H, W, Cin), device=dev).to(LP_DTYPE)
    w = (torch.randn((Cout, R, R, Cin), device=dev) / (Cin * R * R) ** 0.5).to(LP_DTYPE)
    b = torch.randn((Cout,), device=dev)
    OH = (H + 2 * (R // 2) - R) // stride + 1
    OW = (W + 2 * (R // 2) - R) // stride + 1
    r = torch.randn((N, OH, OW, Cout), device=dev).to(LP_DTYPE) if res else None
    for _ in range(3):
        ops.conv_bn_act(x, w, b, stride, R // 2, True, r)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.conv_bn_act(x, w, b, stride, R // 2, True, r)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 50
    fl = 2.0 * N * OH * OW * Cout * R * R * Cin
    by = 2.0 * (x.numel() + w.numel() + N * OH * OW * Cout * (2 if res else 1))
    print("%-40s %8.1f us %7.1f TF %6.0f GB/s" % (str((H, W, Cin, Cout, R, stride, res)), us, fl / us / 1e6, by / us / 1e3))
