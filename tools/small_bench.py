"""Interleaved A/B timing of the GCN message pass and the streaming distance matrix (variants via environment
variables read by the launchers at every call). usage: small_bench.py "NAME:K=V,..." ..."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops, _hip
from torchreid._hip import LP_DTYPE
dev = "cuda:0"
variants = []
for a in sys.argv[1:]:
    name, _, kv = a.partition(":")
    variants.append((name, dict(x.split("=") for x in kv.split(",") if x)))
if not variants:
    variants = [("default", {})]
keys = sorted({k for _, d in variants for k in d})
B, V, C = 32, 56, 2048
f = torch.randn((B, V, C), device=dev)
h = torch.randn((B, V, C), device=dev)
G = torch.rand((B, V, V), device=dev)
sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
m, n, D = 32, 12180, 4096
q = torch.randn((m, D), device=dev).to(LP_DTYPE)
g = torch.randn((n, D), device=dev).to(LP_DTYPE)
q32, g32 = q.float(), g.float()
trash = torch.empty((512 << 20,), dtype=torch.uint8, device=dev)  # flushes L2 + MALL between timed calls
def prop(): ops.graph_propagate(f, h, G, sc, sh, 0.1, 0.1, want_lp=True)
def dm(): ops.distmat(q, g, "cosine")
def dm32(): ops.distmat(q32, g32, "cosine")
o_ = torch.empty_like(f)
def tadd(): torch.add(f, h, alpha=0.1, out=o_)
def tcopy(): o_.copy_(f)
CASES = [("torch.add 3 x 14.7 MB (reference stream)", tadd, 4.0 * 3 * B * V * C),
         ("torch copy 2 x 14.7 MB (reference stream)", tcopy, 4.0 * 2 * B * V * C),
         ("propagate B32 V56 (44.4 MB + 7.3 MB bf16 copy)", prop, 4.0 * (3 * B * V * C + B * V * V) + 2.0 * B * V * C),
         ("distmat bf16 32x12180x4096", dm, 2.0 * (m + n) * D + 4.0 * m * n),
         ("distmat fp32 32x12180x4096", dm32, 4.0 * (m + n) * D + 4.0 * m * n)]
for label, fn, nbytes in CASES:
    for cold in (False, True):
        times = {nm: [] for nm, _ in variants}
        for rnd in range(12):
            for nm, env in variants:
                for k in keys:
                    os.environ.pop(k, None)
                os.environ.update(env)
                _hip.reload_options()  # the library reads its switches once; re-read after flipping them
                if cold:
                    trash.fill_(rnd & 0xff)
                reps = 1 if cold else 20  # warm: back-to-back launches (launch latency amortised, operands in L2/MALL)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(reps):
                    fn()
                e.record()
                torch.cuda.synchronize()
                if rnd >= 2:
                    times[nm].append(s.elapsed_time(e) * 1000 / reps)
        print("%-52s %-5s " % (label, "cold" if cold else "warm") + "  ".join(
            "%s %6.1fus %5.2fTB/s" % (nm, statistics.median(times[nm]), nbytes / statistics.median(times[nm]) / 1e6) for nm, _ in variants))

# the N = 8 match stage of bench.py: 256 gathered queries against a 1523-row gallery shard
q8 = torch.randn((256, D), device=dev).to(LP_DTYPE); g8 = torch.randn((1523, D), device=dev).to(LP_DTYPE)
for _ in range(3): ops.distmat(q8, g8, "cosine")
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): ops.distmat(q8, g8, "cosine")
e.record(); torch.cuda.synchronize()
print("distmat bf16 256 x 1523 x 4096 (8-GPU shard): %.1f us" % (s.elapsed_time(e) * 50))
