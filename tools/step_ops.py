"""Which torch ops run inside one bench step (they should be allocations only): torch.profiler over 3 eager steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bench
from torchreid import hip_ops as ops
from torchreid._hip import LP_DTYPE, LP_NAME
dev = torch.device("cuda:0")
model, sd = bench.build_model(dev, LP_NAME)
gen = torch.Generator(device=dev); gen.manual_seed(1)
clips = torch.randn((32, 8, 3, 256, 128), device=dev, generator=gen)
adj = bench.synthetic_pose_adjacency(32, 8, dev, gen)
gal = ops.row_l2_normalize(torch.randn((12180, 4096), device=dev), True, LP_DTYPE)
out = torch.empty((32, 12180), device=dev)
def step():
    emb = model(clips, adj)
    q = ops.row_l2_normalize(emb, True, LP_DTYPE)
    return ops.distmat(q, gal, "cosine", out=out)
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=40, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="count", row_limit=25, max_name_column_width=50, max_src_column_width=120))
