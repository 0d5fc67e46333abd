"""Race screen for the hand-synchronised kernels: repeat the full-size layer shapes many times on random data and demand
bitwise-identical outputs against the first run and against the simpler kernel of the same contraction where one exists
(the LDS rings, counted waits and in-place epilogues either work every time or show up here as rare differing tiles)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops, _hip
from torchreid._hip import LP_DTYPE
dev = "cuda:0"
torch.manual_seed(0)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30

def conv_case(N, H, W, Cin, Cout, R, stride, res, env_alt):
    x = torch.randn((N, H, W, Cin), device=dev).to(LP_DTYPE)
    w = (torch.randn((Cout, R, R, Cin), device=dev) / (Cin * R * R) ** 0.5).to(LP_DTYPE)
    b = torch.randn((Cout,), device=dev)
    OH, OW = (H + 2 * (R // 2) - R) // stride + 1, (W + 2 * (R // 2) - R) // stride + 1
    r = torch.randn((N, OH, OW, Cout), device=dev).to(LP_DTYPE) if res else None
    for k in env_alt:
        os.environ[k] = env_alt[k]
    _hip.reload_options()
    ref = ops.conv_bn_act(x, w, b, stride, R // 2, True, r).clone()
    for k in env_alt:
        os.environ.pop(k)
    _hip.reload_options()
    bad = 0
    for _ in range(REPS):
        out = ops.conv_bn_act(x, w, b, stride, R // 2, True, r)
        bad += int(not torch.equal(out, ref))
    return bad

cases = [("wide 1x1 2048->512", (256, 16, 8, 2048, 512, 1, 1, False, {"AGRL_IGEMM_WIDE": "0"})),
         ("wide 1x1 512->2048 +res", (256, 16, 8, 512, 2048, 1, 1, True, {"AGRL_IGEMM_WIDE": "0"})),
         ("wide persistent 1024->2048", (256, 16, 8, 1024, 2048, 1, 1, False, {"AGRL_IGEMM_WIDE": "0"})),
         ("wide128 1x1 1024->256", (256, 16, 8, 1024, 256, 1, 1, False, {"AGRL_IGEMM_WIDE": "0"})),
         ("wide strided 256->512 s2", (256, 64, 32, 256, 512, 1, 2, False, {"AGRL_IGEMM_WIDE": "0"})),
         ("3x3 two-block 512->512", (256, 16, 8, 512, 512, 3, 1, False, {"AGRL_CONV3X3_WIDE": "0"})),
         ("3x3 c64 64->64", (256, 64, 32, 64, 64, 3, 1, False, {"AGRL_CONV3X3_C64": "0"}))]
total = 0
for name, c in cases:
    bad = conv_case(*c)
    total += bad
    print("%-28s %d / %d runs differ from the baseline kernel" % (name, bad, REPS))

# fused layer-1 block (3x3 + conv3 + residual + next conv1) vs the split launches
Nb, Hb, Wb = 256, 64, 32
zin = torch.randn((Nb, Hb, Wb, 64), device=dev).to(LP_DTYPE)
resb = torch.randn((Nb, Hb, Wb, 256), device=dev).to(LP_DTYPE)
w2b = (torch.randn((64, 3, 3, 64), device=dev) / 24).to(LP_DTYPE)
w3b = (torch.randn((256, 1, 1, 64), device=dev) / 8).to(LP_DTYPE)
b2b, b3b = torch.randn(64, device=dev), torch.randn(256, device=dev)
for cn in (64, 128):
    w1b = (torch.randn((cn, 1, 1, 256), device=dev) / 16).to(LP_DTYPE)
    b1b = torch.randn(cn, device=dev)
    yb = ops.conv_bn_act(zin, w2b, b2b, 1, 1, True)
    ob = ops.conv_bn_act(yb, w3b, b3b, 1, 0, True, residual=resb)
    zb = ops.conv_bn_act(ob, w1b, b1b, 1, 0, True)
    bad = 0
    for _ in range(REPS):
        o, zz = ops.bottleneck_block(zin, w2b, b2b, w3b, b3b, resb, w1b, b1b)
        bad += int(not (torch.equal(o, ob) and torch.equal(zz, zb)))
    total += bad
    print("bottleneck block cnext=%-3d     %d / %d runs differ from the split convs" % (cn, bad, REPS))
xsb = torch.randn((Nb, Hb, Wb, 64), device=dev).to(LP_DTYPE)
wsb = (torch.randn((256, 1, 1, 64), device=dev) / 8).to(LP_DTYPE)
bsb = torch.randn(256, device=dev)
w1b = (torch.randn((64, 1, 1, 256), device=dev) / 16).to(LP_DTYPE)
b1b = torch.randn(64, device=dev)
o0, z0 = ops.bottleneck_block(zin, w2b, b2b, w3b, b3b, None, w1b, b1b, shortcut=(xsb, wsb, bsb))
o0, z0 = o0.clone(), z0.clone()
bad = 0
for _ in range(REPS):
    o, zz = ops.bottleneck_block(zin, w2b, b2b, w3b, b3b, None, w1b, b1b, shortcut=(xsb, wsb, bsb))
    bad += int(not (torch.equal(o, o0) and torch.equal(zz, z0)))
total += bad
print("bottleneck block + downsample %d / %d runs differ from the first run" % (bad, REPS))
del zin, resb, xsb

# fused tails vs the split convs
N, H, W = 256, 64, 32
y2 = torch.randn((N, H, W, 64), device=dev).to(LP_DTYPE)
res = torch.randn((N, H, W, 256), device=dev).to(LP_DTYPE)
w3 = (torch.randn((256, 1, 1, 64), device=dev) / 8).to(LP_DTYPE)
b3 = torch.randn(256, device=dev)
for cn in (64, 128):
    w1 = (torch.randn((cn, 1, 1, 256), device=dev) / 16).to(LP_DTYPE)
    b1 = torch.randn(cn, device=dev)
    o_ref = ops.conv_bn_act(y2, w3, b3, 1, 0, True, residual=res)
    z_ref = ops.conv_bn_act(o_ref, w1, b1, 1, 0, True)
    bad = 0
    for _ in range(REPS):
        o, z = ops.bottleneck_tail(y2, w3, b3, res, w1, b1)
        bad += int(not (torch.equal(o, o_ref) and torch.equal(z, z_ref)))
    total += bad
    print("%-28s %d / %d runs differ from the split convs" % ("bottleneck tail cnext=%d" % cn, bad, REPS))
x0 = torch.randn((N, H, W, 64), device=dev).to(LP_DTYPE)
ws = (torch.randn((256, 1, 1, 64), device=dev) / 8).to(LP_DTYPE)
bs = torch.randn(256, device=dev)
w1 = (torch.randn((64, 1, 1, 256), device=dev) / 16).to(LP_DTYPE)
b1 = torch.randn(64, device=dev)
o0, z0 = ops.bottleneck_tail(y2, w3, b3, None, w1, b1, shortcut=(x0, ws, bs))
o0, z0 = o0.clone(), z0.clone()
bad = 0
for _ in range(REPS):
    o, z = ops.bottleneck_tail(y2, w3, b3, None, w1, b1, shortcut=(x0, ws, bs))
    bad += int(not (torch.equal(o, o0) and torch.equal(z, z0)))
total += bad
print("%-28s %d / %d runs differ from the first run" % ("bottleneck tail + downsample", bad, REPS))
# streaming kernels: run-to-run determinism
f = torch.randn((32, 56, 2048), device=dev); h = torch.randn((32, 56, 2048), device=dev); G = torch.rand((32, 56, 56), device=dev)
sc, sh = torch.rand(2048, device=dev) + 0.5, torch.randn(2048, device=dev)
p0 = ops.graph_propagate(f, h, G, sc, sh, 0.1, 0.1, want_lp=True)
q = torch.randn((32, 4096), device=dev).to(LP_DTYPE); g = torch.randn((12180, 4096), device=dev).to(LP_DTYPE)
d0 = ops.distmat(q, g, "cosine").clone()
bad = 0
for _ in range(REPS):
    p1 = ops.graph_propagate(f, h, G, sc, sh, 0.1, 0.1, want_lp=True)
    bad += int(not (torch.equal(p1[0], p0[0]) and torch.equal(p1[1], p0[1])))
    bad += int(not torch.equal(ops.distmat(q, g, "cosine"), d0))
total += bad
print("%-28s %d / %d runs differ from the first run" % ("propagate + distmat", bad, 2 * REPS))
# round-3 kernels: GraphLayer in the commuted form (graph, G f, GEMM with the fused epilogue), its one-workgroup-per-tracklet
# form at 256 tracklets, the single-pass top-k (LDS atomics + in-launch radix fallback) and the full-row argsort
adj = (torch.rand((32, 56, 56), device=dev) > 0.5).float()
wl = (torch.randn((2048, 2048), device=dev) * 0.02).to(LP_DTYPE)
def commuted(ff, aa):
    Gm = ops.graph_matrix(ff, aa, True, True)
    P = ops.graph_apply_operand(Gm, ff, LP_DTYPE)
    return ops.graph_linear_mix(P, wl, ff, sc, sh, 0.1, 0.1)
c0 = commuted(f, adj).clone()
f256 = torch.rand((256, 1, 2048), device=dev) + 0.02 * torch.randn((256, 56, 2048), device=dev)
adj256 = (torch.rand((256, 56, 56), device=dev) > 0.5).float()
t0 = [t.clone() for t in ops.graph_tracklet_operand(f256, adj256, True, True, LP_DTYPE, want_graph=True)]
dk = torch.randn((1980, 12180), device=dev)
dk[5] = 0.25   # a row of equal values: the radix path inside the single-pass launch
k0 = [t.clone() for t in ops.rank_topk(dk, 50)]
a0 = ops.rank_argsort(dk[:64]).clone()
bad = 0
for _ in range(REPS):
    bad += int(not torch.equal(commuted(f, adj), c0))
    t1 = ops.graph_tracklet_operand(f256, adj256, True, True, LP_DTYPE, want_graph=True)
    bad += int(not (torch.equal(t1[0], t0[0]) and torch.equal(t1[1], t0[1])))
    k1 = ops.rank_topk(dk, 50)
    bad += int(not (torch.equal(k1[0], k0[0]) and torch.equal(k1[1], k0[1])))
    bad += int(not torch.equal(ops.rank_argsort(dk[:64]), a0))
total += bad
print("%-28s %d / %d runs differ from the first run" % ("gcn commuted / tracklet / top-k / argsort", bad, 4 * REPS))
torch.cuda.synchronize()
print("RACE SCREEN", "CLEAN" if total == 0 else "FAILED (%d)" % total)
sys.exit(1 if total else 0)
