import sys, os, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'agrl.pytorch_amd'), os.path.join(ROOT, 'tests')]
import torch
from test_gpu_train import _problem, rel, DEV
from torchreid.models._train_hip import featuremaps_train
S, H, W = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (6, 64, 32))]
ref, dev, x, adj, pids, _ = _problem(S, H, W)
frames = x.view(-1, 3, H, W)
g = torch.Generator().manual_seed(17)
ref64 = copy.deepcopy(ref).double()
stock = copy.deepcopy(dev)
for m in (ref, dev, ref64, stock): m.train()
a1, a2 = ref.featuremaps(frames)
w1, w2 = torch.randn(a1.shape, generator=g), torch.randn(a2.shape, generator=g)
def loss(p, q, dt=None):
    return ((p * w1.to(p.device, p.dtype)).sum() + (q * q * w2.to(p.device, p.dtype)).sum()) / p.numel()
loss(a1, a2).backward()
c1, c2 = ref64.featuremaps(frames.double()); loss(c1, c2).backward()
b1, b2 = featuremaps_train(dev, frames.to(DEV)); loss(b1, b2).backward()
s1, s2 = stock.featuremaps(frames.to(DEV)); loss(s1, s2).backward()
g64 = {k: p.grad for k, p in ref64.named_parameters() if p.grad is not None}
for tag, m, o in (("cpu32", ref, (a1, a2)), ("native", dev, (b1, b2)), ("stockgpu", stock, (s1, s2))):
    r = sorted(((rel(p.grad, g64[k]), k) for k, p in m.named_parameters() if k in g64), reverse=True)
    print("%-9s maps %.2e %.2e | grads worst %.2e (%s) 2nd %.2e (%s) median %.2e" % (tag, rel(o[0], c1), rel(o[1], c2), r[0][0], r[0][1], r[1][0], r[1][1], r[len(r)//2][0]))
    if tag == "native":
        for e, k in r[:12]: print("    %.2e %s" % (e, k))
