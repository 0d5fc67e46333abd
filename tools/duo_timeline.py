"""Phase timeline of conv1x1_duo_kernel from the profiling build (tools/duo_ablate.sh 64 -> lib/libagrl_hip_duoabl64.so):
s_memtime stamps of matrix wave 0 and memory wave 4 of every workgroup, per channel tile -> median phase durations.
usage: AGRL_HIP_LIB=.../libagrl_hip_duoabl64.so python tools/duo_timeline.py"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from torchreid import hip_ops as ops
from torchreid import _hip
from torchreid._hip import LP_DTYPE

dev = "cuda:0"
frames, K, Cout = 256, 512, 2048
x = torch.relu(torch.randn((frames, 16, 8, K), device=dev)).to(LP_DTYPE)
res = torch.relu(torch.randn((frames, 16, 8, Cout), device=dev)).to(LP_DTYPE)
w = (torch.randn((Cout, 1, 1, K), device=dev) / K ** 0.5).to(LP_DTYPE)
b = torch.randn((Cout,), device=dev)
packed = ops.conv1x1_pack(w)
w1 = (torch.randn((512, 1, 1, Cout), device=dev) / Cout ** 0.5).to(LP_DTYPE)
w2 = (torch.randn((512, 3, 3, 512), device=dev) / (9 * 512) ** 0.5).to(LP_DTYPE)
b1 = torch.randn((512,), device=dev)
p1, p2 = ops.conv1x1_pack(w1), ops.conv3x3_pack(w2)
nwg = frames
buf = torch.zeros((nwg, 8, 2, 8), dtype=torch.int64, device=dev)
lib = _hip.lib()
lib.agrl_duo_trace_buffer.argtypes = [ctypes.c_void_p]
assert lib.agrl_duo_trace_buffer(buf.data_ptr()) == 0
for it in range(3):
    y1 = ops.conv1x1_packed(res, p1, b1, 512, True)
    y2 = ops.conv3x3_packed(y1, p2, b1, 512, True)
    buf.zero_()
    ops.conv1x1_packed_res(x, packed, b, Cout, res)
    torch.cuda.synchronize()
assert lib.agrl_duo_trace_buffer(None) == 0
t = buf.cpu().numpy().astype(np.int64)
mx, mem = t[:, :, 0, :], t[:, :, 1, :]
t0 = mx[:, 0, 0].min()
print("span %d ticks (first k-loop start -> last E2)" % (mx[:, 7, 3].max() - t0))
def stat(name, d):
    print("  %-58s median %7.0f  p10 %7.0f  p90 %7.0f" % (name, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
stat("matrix: k-loop (4 slabs)", mx[:, :, 1] - mx[:, :, 0])
stat("matrix: wait at E1 (memory wave not there yet)", mx[:, :, 2] - mx[:, :, 1])
stat("matrix: combine", mx[:, :, 3] - mx[:, :, 2])
stat("matrix: E2 -> next tile's loop start", mx[:, 1:, 0] - mx[:, :-1, 3])
stat("matrix: whole tile period", mx[:, 1:, 0] - mx[:, :-1, 0])
stat("memory: parked from E1 to E2", mem[:, :, 1] - mem[:, :, 0])
stat("memory: image reads + store issue", mem[:, :, 2] - mem[:, :, 1])
stat("memory: residual DMA issue", mem[:, :-1, 3] - mem[:, :-1, 2])
stat("memory: after issue -> residual landed (next E1)", mem[:, 1:, 0] - mem[:, :-1, 3])
print("  one workgroup, matrix wave (loop start, loop end, E1 passed, combined) per tile, ticks from start:")
for k in range(8):
    print("    ", [int(v - t0) for v in mx[0, k, :4]], " memory wave:", [int(v - t0) for v in mem[0, k, :4]])
