"""BASELINE configs[4] on one GPU: the 1980 x 12 180 x 4096 distance matrix and agrl_distmat_topk, per column width of the fp32-output
tile (AGRL_DISTMAT_TILE_N = 256: 384 tiles = 1.5 rounds of 256 CUs; 192: 512 tiles = 2 exact rounds at 3/4 of the cost each).
usage: config5_bench.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops
from torchreid import _hip
from torchreid._hip import LP_DTYPE
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda:0"
m, n, D = 1980, 12180, 4096
g = torch.Generator(device=dev).manual_seed(3)
q = ops.row_l2_normalize(torch.randn((m, D), device=dev, generator=g), True, LP_DTYPE)
gal = ops.row_l2_normalize(torch.randn((n, D), device=dev, generator=g), True, LP_DTYPE)


def timed(fn):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


res = {}
for tile in ("256", "192", "auto"):
    if tile == "auto":
        os.environ.pop("AGRL_DISTMAT_TILE_N", None)
    else:
        os.environ["AGRL_DISTMAT_TILE_N"] = tile
    _hip.reload_options()
    d = ops.distmat(q, gal, "cosine")
    idx, val = ops.distmat_topk(q, gal, "cosine", 50)
    torch.cuda.synchronize()
    res[tile] = (d, idx, val)
    t_d = timed(lambda: ops.distmat(q, gal, "cosine"))
    t_k = timed(lambda: ops.distmat_topk(q, gal, "cosine", 50))
    print("tile %-4s distmat %.3f ms = %.0f TFLOP/s = %.3f of 2.5 PF | distmat_topk50 %.3f ms" % (
        tile, t_d, 2.0 * m * n * D / t_d / 1e9, 2.0 * m * n * D / t_d / 1e9 / 2500, t_k))
print("192 == 256: matrix", torch.equal(res["192"][0], res["256"][0]), "top-50 idx", torch.equal(res["192"][1], res["256"][1]),
      "val", torch.equal(res["192"][2], res["256"][2]))
