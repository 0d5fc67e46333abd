"""Per-entry-point time of one native train step at BASELINE config 4's per-GPU shape (16 tracklets x 16 frames, fp32): HIP
events around every C-ABI call (_hip.PROFILE) + the wall time of the step; usage: python tools/train_profile.py [B S]"""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "agrl.pytorch_amd"), os.path.join(ROOT, "tests")]
import torch
from recipe import recipe_state_dict
from bench import synthetic_pose_adjacency
from torchreid import _hip, losses, models
B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 16)
dev = torch.device("cuda:0")
m = models.init_model("vmgn", num_classes=702, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                      pyramid_part=True, use_pose=True, learn_graph=True, consistent_loss=True)
m.load_state_dict(recipe_state_dict(m.state_dict(), seed=4))
m = m.to(dev).train()
gen = torch.Generator(device=dev); gen.manual_seed(4)
x = torch.randn((B, S, 3, 256, 128), device=dev, generator=gen)
adj = synthetic_pose_adjacency(B, S, dev, gen)
pids = torch.arange(B // 4, device=dev).repeat_interleave(4)
ce = losses.CrossEntropyLabelSmooth(num_classes=702, use_gpu=True)
htri = losses.TripletLoss(margin=0.3, soft=True)
def step():
    outs, feats = m(x, adj)
    loss = losses.DeepSupervision(ce, outs, pids) + losses.DeepSupervision(htri, feats, pids)
    m.zero_grad()
    loss.backward()
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
_hip.PROFILE = []
step(); torch.cuda.synchronize()
prof, _hip.PROFILE = _hip.PROFILE, None
agg = collections.defaultdict(lambda: [0.0, 0])
for name, s, e, tag in prof:
    agg[name][0] += s.elapsed_time(e); agg[name][1] += 1
tot = sum(v[0] for v in agg.values())
print("native step wall %.1f ms; C-ABI kernels %.1f ms in %d launches" % (wall * 1e3, tot, sum(v[1] for v in agg.values())))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("  %-32s %8.2f ms %5d launches" % (k, v[0], v[1]))
