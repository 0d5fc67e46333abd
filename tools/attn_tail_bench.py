"""The per-tracklet tail: agrl_attn_tail (one launch, one workgroup per tracklet) against agrl_row_sqnorm + agrl_attn_pool_bnneck +
agrl_row_l2_normalize (three launches, 8 workgroups per tracklet in the middle one), per tracklet count. usage: attn_tail_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid._hip import LP_DTYPE
dev = "cuda:0"
S, P, C, hw = 8, 7, 2048, 128


def timed(fn, rounds=30):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for B in (32, 64, 128, 256, 512):
    nodes = torch.rand((B, S, P, C), device=dev)
    gsum = torch.rand((B * S, C), device=dev)
    v = [torch.rand(C, device=dev) for _ in range(4)]

    def split():
        sqn = ops.row_sqnorm(nodes.view(B * S * P, C))
        out = ops.attn_pool_bnneck(nodes, sqn, gsum, v[0], v[1], v[2], v[3], B, S, P, hw)
        return ops.row_l2_normalize(out, True, LP_DTYPE)

    def fused():
        return ops.attn_tail(nodes, gsum, v[0], v[1], v[2], v[3], B, S, P, hw, query_dtype=LP_DTYPE)

    for _ in range(3):
        split(), fused()
    print("%4d tracklets: three launches %.1f us, one launch %.1f us" % (B, timed(split), timed(fused)))
