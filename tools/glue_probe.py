"""Which torch-side launches (copies, elementwise kernels) sit between the library's kernels in one bench step?
Run on the GPU box: python tools/glue_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "agrl.pytorch_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from torchreid._hip import LP_DTYPE, LP_NAME

dev = torch.device("cuda:0")
model, sd = bench.build_model(dev, LP_NAME)
gen = torch.Generator(device=dev); gen.manual_seed(1)
clips = torch.randn((32, 8, 3, 256, 128), device=dev, generator=gen)
adj = bench.synthetic_pose_adjacency(32, 8, dev, gen)
from torchreid import hip_ops as ops
g = ops.row_l2_normalize(torch.randn((12180, 4096), device=dev), True, LP_DTYPE)
out = torch.empty((32, 12180), device=dev)
def step():
    emb = model(clips, adj)
    q = ops.row_l2_normalize(emb, True, LP_DTYPE)
    return ops.distmat(q, g, "cosine", out=out)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
seen = {}
for ev in prof.events():
    n = ev.name
    if n.startswith("aten::") and any(k in n for k in ("copy_", "fill_", "zero_", "cat", "to", "contiguous", "clone", "mul", "add", "sub", "div", "ones", "zeros", "empty_like", "_to_copy")):
        st = [s for s in (ev.stack or []) if "torchreid" in s or "bench" in s or "glue_probe" in s]
        key = (n, st[0] if st else "?")
        seen[key] = seen.get(key, 0) + 1
for (n, st), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print("%3d  %-28s %s" % (c, n, st))
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
