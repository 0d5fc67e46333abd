"""Interleaved A/B timing of conv-kernel variants selected through environment variables (read by the launcher at
every call). usage: conv_ab.py "NAME:K=V,K=V" "NAME2:K=V" ... [--shapes 0,1,2]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops, _hip
from torchreid._hip import LP_DTYPE
dev = "cuda:0"
SHAPES = [  # N, H, W, Cin, Cout, R, stride, res
    (256, 64, 32, 64, 256, 1, 1, False),
    (256, 64, 32, 64, 256, 1, 1, True),
    (256, 64, 32, 256, 64, 1, 1, False),
    (256, 32, 16, 128, 512, 1, 1, True),
    (256, 16, 8, 256, 1024, 1, 1, True),
    (256, 16, 8, 1024, 256, 1, 1, False),
    (256, 16, 8, 256, 256, 3, 1, False),
    (256, 16, 8, 512, 2048, 1, 1, True),
    (256, 16, 8, 2048, 512, 1, 1, False),
    (256, 16, 8, 1024, 2048, 1, 1, False),
    (256, 16, 8, 512, 512, 3, 1, False),
    (256, 64, 32, 64, 64, 3, 1, False),
    (256, 32, 16, 128, 128, 3, 1, False),
    (256, 16, 8, 512, 2048, 1, 1, False),
    (256, 16, 8, 1024, 512, 1, 1, False),
    (256, 32, 16, 512, 256, 1, 1, False),
    (256, 32, 16, 512, 128, 1, 1, False),
    (256, 32, 16, 256, 256, 3, 2, False),   # 17: the stride-2 3x3 of layer 3's first block
    (256, 64, 32, 128, 128, 3, 2, False),   # 18: the stride-2 3x3 of layer 2's first block
]
args = sys.argv[1:]
if "--shapes" in args:
    i = args.index("--shapes")
    SHAPES = [SHAPES[int(v)] for v in args[i + 1].split(",")]
    args = args[:i] + args[i + 2:]
variants = []
for a in args:
    name, _, kv = a.partition(":")
    variants.append((name, dict(x.split("=") for x in kv.split(",") if x)))
keys = sorted({k for _, d in variants for k in d})
ROUNDS, REPS = 7, 10
print("%-38s " % "shape" + " ".join("%14s" % n for n, _ in variants))
for (N, H, W, Cin, Cout, R, stride, res) in SHAPES:
    x = torch.randn((N, H, W, Cin), device=dev).to(LP_DTYPE)
    w = (torch.randn((Cout, R, R, Cin), device=dev) / (Cin * R * R) ** 0.5).to(LP_DTYPE)
    b = torch.randn((Cout,), device=dev)
    r = torch.randn((N, H, W, Cout), device=dev).to(LP_DTYPE) if res else None
    times = {n: [] for n, _ in variants}
    for rnd in range(ROUNDS + 1):
        for name, env in variants:
            for k in keys:
                os.environ.pop(k, None)
            os.environ.update(env)
            _hip.reload_options()  # the library reads its switches once; re-read after flipping them
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(REPS):
                ops.conv_bn_act(x, w, b, stride, R // 2, True, r)
            e.record()
            torch.cuda.synchronize()
            if rnd:
                times[name].append(s.elapsed_time(e) * 1000 / REPS)
    fl = 2.0 * N * (H // stride) * (W // stride) * Cout * R * R * Cin
    print("%-38s " % str((H, W, Cin, Cout, R, res)) + " ".join("%7.1fus %4.0fTF" % (statistics.median(times[n]), fl / statistics.median(times[n]) / 1e6) for n, _ in variants))
