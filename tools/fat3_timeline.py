"""Phase timeline of conv3x3_fat_kernel from a profiling build (conv3x3_fat.hip with -DFAT_ABL=16 linked as lib/libagrl_hip_fat3trace.so,
see tools/fat3_trace_build.sh): s_memtime stamps of thread 0 of every workgroup -> median phase durations per layer shape.
usage: AGRL_HIP_LIB=.../libagrl_hip_fat3trace.so python tools/fat3_timeline.py"""
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
frames = 256
lib = _hip.lib()
lib.agrl_fat3_trace_buffer.argtypes = [ctypes.c_void_p]
for cin, cout in ((256, 256), (512, 512)):
    x = torch.relu(torch.randn((frames, 16, 8, cin), device=dev)).to(LP_DTYPE)
    w = (torch.randn((cout, 3, 3, cin), device=dev) / (9 * cin) ** 0.5).to(LP_DTYPE)
    b = torch.randn((cout,), device=dev)
    packed = ops.conv3x3_pack(w)
    nslab = cin // 64
    buf = torch.zeros((2048, 16), dtype=torch.int64, device=dev)
    assert lib.agrl_fat3_trace_buffer(buf.data_ptr()) == 0
    for _ in range(3):
        buf.zero_()
        ops.conv3x3_packed(x, packed, b, cout, True)
        torch.cuda.synchronize()
    assert lib.agrl_fat3_trace_buffer(None) == 0
    t = buf.cpu().numpy().astype(np.int64)
    t = t[t[:, 0] > 0]
    span = t[:, 14].max() - t[:, 0].min()
    print("conv3x3 %d -> %d: %d workgroups, kernel span %d ticks" % (cin, cout, len(t), span))

    def stat(name, d):
        print("  %-52s median %7.0f  p10 %7.0f  p90 %7.0f ticks" % (name, np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    stat("start (after the first workgroup of the launch)", t[:, 0] - t[:, 0].min())
    stat("prologue: setup + first patches + first weights", t[:, 1] - t[:, 0])
    stat("first barrier (slab 0's patches of every wave)", t[:, 2] - t[:, 1])
    for s_ in range(1, min(nslab, 8)):
        stat("slab %d" % (s_ - 1), t[:, 2 + s_] - t[:, 1 + s_])
    stat("last slab", t[:, 11] - t[:, 1 + min(nslab, 8)])
    stat("drain (vmcnt(0), nops)", t[:, 12] - t[:, 11])
    stat("epilogue: bias / ReLU / pack / store issue", t[:, 13] - t[:, 12])
    stat("stores acknowledged", t[:, 14] - t[:, 13])
    stat("workgroup lifetime", t[:, 14] - t[:, 0])
    per_slab = 9 * 2 * 4 * (8 if len(t) > 200 and cin == 256 else 16) * 16
    print("  (a slab's MFMAs at one per 16 cycles: %d cycles)" % per_slab)
