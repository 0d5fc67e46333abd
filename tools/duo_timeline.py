"""Phase timeline of conv1x1_duo_kernel from the profiling build (tools/duo_ablate.sh 64 -> lib/libagrl_hip_duoabl64.so):
per-workgroup s_memtime stamps -> median phase durations, per-CU overlap of the two resident workgroups.
usage: AGRL_HIP_LIB=.../libagrl_hip_duoabl64.so python tools/duo_timeline.py [strided2 | strided3]   (default: conv3 + residual of layer 4;
strided2 / strided3: the two-source first blocks of layers 2 / 3)"""
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
case = sys.argv[1] if len(sys.argv) > 1 else "res"
if case in ("strided2", "strided3"):
    hi, wi, k1, k2, co = (64, 32, 256, 128, 512) if case == "strided2" else (32, 16, 512, 256, 1024)
    sa = torch.relu(torch.randn((frames, hi, wi, k1), device=dev)).to(LP_DTYPE)
    sy = torch.relu(torch.randn((frames, hi // 2, wi // 2, k2), device=dev)).to(LP_DTYPE)
    spk = ops.conv1x1_pack((torch.randn((co, 1, 1, k1 + k2), device=dev) / (k1 + k2) ** 0.5).to(LP_DTYPE))
    sb = torch.randn((co,), device=dev)
    nwg = frames * (hi // 2) * (wi // 2) // 128 * (co // 256)
else:
    nwg = frames * (Cout // 256)
buf = torch.zeros((nwg, 12), dtype=torch.int64, device=dev)
lib = _hip.lib()
lib.agrl_duo_trace_buffer.argtypes = [ctypes.c_void_p]
assert lib.agrl_duo_trace_buffer(buf.data_ptr()) == 0
for it in range(3):
    y1 = ops.conv1x1_packed(res, p1, b1, 512, True)
    y2 = ops.conv3x3_packed(y1, p2, b1, 512, True)
    buf.zero_()
    if case in ("strided2", "strided3"):
        ops.conv1x1_packed_dual_strided(sa, sy, spk, sb, co, 2, True)
    else:
        ops.conv1x1_packed_res(x, packed, b, Cout, res)
    torch.cuda.synchronize()
t = buf.cpu().numpy().astype(np.int64)
hw, xcc = t[:, 10], t[:, 11]
cu = ((xcc & 0xf) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
names = ["start", "setup", "prologue (first loads + barrier)", "k-loop", "drain + barrier", "residual DMA + wait", "combine", "issue stores", "stores acked"]
# s_memtime: 100 MHz constant on gfx9? print the raw span to calibrate against the kernel's duration
span = (t[:, 8].max() - t[:, 0].min())
print("workgroups %d, distinct CUs %d, span of stamps %d ticks" % (nwg, len(np.unique(cu)), span))
d = np.diff(t[:, :9], axis=1)
for k in range(8):
    print("  %-34s median %7.0f  p10 %7.0f  p90 %7.0f ticks" % (names[k + 1], np.median(d[:, k]), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90)))
life = t[:, 8] - t[:, 0]
print("  workgroup lifetime median %.0f ticks; sum of lifetimes / span / CUs = %.2f resident workgroups per CU" % (np.median(life), life.sum() / span / len(np.unique(cu))))
# per CU: fraction of time with two workgroups in the SAME kind of phase
same = tot = 0
for c in np.unique(cu)[:64]:
    rows = t[cu == c]
    ev = []
    for r in rows:
        ev.append((r[2], r[3], 0))   # k-loop
        ev.append((r[4], r[8], 1))   # epilogue
    T0, T1 = rows[:, 0].min(), rows[:, 8].max()
    grid = np.linspace(T0, T1, 2000)
    kl = np.zeros_like(grid); ep = np.zeros_like(grid)
    for a, b_, kind in ev:
        m = (grid >= a) & (grid < b_)
        if kind == 0: kl += m
        else: ep += m
    tot += len(grid)
    same += ((kl >= 2) | (ep >= 2)).sum()
print("  fraction of CU time with both resident workgroups in the same phase kind: %.2f" % (same / tot))
order = np.argsort(t[:, 0])
print("  first 6 workgroups on one CU (start, k-loop begin, k-loop end, epi begin, stores acked), ticks from kernel start:")
c0 = cu[order[0]]
rows = t[cu == c0]; rows = rows[np.argsort(rows[:, 0])][:8]
for r in rows:
    print("   ", [int(v - t[:, 0].min()) for v in (r[0], r[2], r[3], r[4], r[5], r[6], r[7], r[8])])
