"""Conditioning probe of the per-parameter gradient parity of the train step: errors of (a) the CPU fp32 step, (b) the stock-torch
GPU step, (c) the native-trunk GPU step, all against the SAME step in float64 on the CPU."""
import sys, os, copy, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'agrl.pytorch_amd'), os.path.join(ROOT, 'tests')]
import torch
from test_gpu_train import _problem, _step, rel, DEV
S, H, W = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (6, 64, 32))]
ref, dev, x, adj, pids, _ = _problem(S, H, W)
ref64 = copy.deepcopy(ref).double()
t = time.time(); l64 = _step(ref64, x.double(), adj.double(), pids, False); print("fp64 cpu step %.1f s" % (time.time() - t))
t = time.time(); l32 = _step(ref, x, adj, pids, False); print("fp32 cpu step %.1f s" % (time.time() - t))
l_dev = _step(dev, x.to(DEV), adj.to(DEV), pids.to(DEV), True)
g64 = {k: p.grad for k, p in ref64.named_parameters() if p.grad is not None}
def errs(model):
    rows = sorted(((rel(p.grad, g64[k]), k) for k, p in model.named_parameters() if k in g64), reverse=True)
    return rows
import statistics
for tag, m in (("cpu fp32", ref), ("native trunk gpu", dev)):
    r = errs(m); print("%-18s vs fp64: worst %.2e (%s) median %.2e  p90 %.2e" % (tag, r[0][0], r[0][1], r[len(r)//2][0], r[len(r)//10][0]))
native = {k: p.grad.clone() for k, p in dev.named_parameters() if p.grad is not None}
dev.hip_train = False
_step(dev, x.to(DEV), adj.to(DEV), pids.to(DEV), True)
r = errs(dev); print("%-18s vs fp64: worst %.2e (%s) median %.2e  p90 %.2e" % ("stock torch gpu", r[0][0], r[0][1], r[len(r)//2][0], r[len(r)//10][0]))
print("loss fp64 %.8f fp32cpu %.8f native %.8f" % (l64.item(), l32.item(), l_dev.item()))
