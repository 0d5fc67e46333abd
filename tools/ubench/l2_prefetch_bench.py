"""conv3 + residual of layer 4 (conv1x1_duo_kernel) with its HBM operands prefetched into the L2 by dedicated CUs on a second stream
(csrc/l2_prefetch.hip) against the launch alone: HIP events on the main stream around [fork, GEMM, join], rotating operand sets.
(Needs a library built with tools/ubench/l2_prefetch.hip: see its header. The result is in profiles/r05_l2_prefetch_experiment.txt.)
usage: l2_prefetch_bench.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import torch
from torchreid import hip_ops as ops, _hip
from torchreid._hip import LP_DTYPE
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F, NS = 256, 4
def t(*shape):
    return [torch.relu(torch.randn(shape, device=dev)).to(LP_DTYPE) for _ in range(NS)]
x, r = t(F, 16, 8, 512), t(F, 16, 8, 2048)
a, y = t(F, 16, 8, 1024), t(F, 16, 8, 512)
w = (torch.randn((2048, 1, 1, 512), device=dev) / 512 ** 0.5).to(LP_DTYPE)
wd = (torch.randn((2048, 1, 1, 1536), device=dev) / 1536 ** 0.5).to(LP_DTYPE)
b = torch.randn(2048, device=dev)
pk, pkd = ops.conv1x1_pack(w), ops.conv1x1_pack(wd)
sink = torch.zeros(4, dtype=torch.int32, device=dev)
pf = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)

def prefetch(t1, t2, P):
    with torch.cuda.stream(pf):
        ops.call("agrl_l2_prefetch", ops.ptr(t1), t1.numel() * 2, ops.ptr(t2) if t2 is not None else None, 0 if t2 is None else t2.numel() * 2,
                 ops.ptr(sink), P, _hip.stream_ptr(dev))

gemms = {
    "conv3 + residual, stored": (lambda i: ops.conv1x1_packed_res(x[i], pk, b, 2048, r[i]), lambda i: (r[i], x[i])),
    "conv3 + residual, pooled": (lambda i: ops.conv1x1_packed_res_pool(x[i], pk, b, 2048, r[i], [1, 2, 4], True, True), lambda i: (r[i], x[i])),
    "conv3 + downsample [1024 | 512] -> 2048": (lambda i: ops.conv1x1_packed(a[i], pkd, b, 2048, True, x2=y[i], duo=True), lambda i: (a[i], y[i])),
}
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(main); fn(e0); e1.record(main); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3
for name, (gemm, operands) in gemms.items():
    arms = {"alone": None, "P=4 first": (4, 1), "P=8 first": (8, 1), "P=4 both": (4, 2), "P=8 both": (8, 2), "P=6 both": (6, 2)}
    res = {k: [] for k in arms}
    only = {4: [], 8: []}
    for it in range(rounds + 2):
        i = it % NS
        for k, cfg in arms.items():
            def run(e0, cfg=cfg, i=i):
                if cfg is not None:
                    pf.wait_event(e0)
                    o = operands(i)
                    prefetch(o[0], o[1] if cfg[1] == 2 else None, cfg[0])
                gemm(i)
                if cfg is not None:
                    ej = torch.cuda.Event(); ej.record(pf); main.wait_event(ej)
            v = timed(run)
            if it >= 2: res[k].append(v)
        for P in only:
            def run(e0, P=P, i=i):
                pf.wait_event(e0); o = operands(i); prefetch(o[0], o[1], P)
                ej = torch.cuda.Event(); ej.record(pf); main.wait_event(ej)
            v = timed(run)
            if it >= 2: only[P].append(v)
    med = lambda v: sorted(v)[len(v) // 2]
    print(name)
    print("   " + "   ".join("%s %.1f us" % (k, med(v)) for k, v in res.items()))
    print("   the prefetch of both operands alone: " + "   ".join("P=%d %.1f us" % (P, med(v)) for P, v in only.items()))
