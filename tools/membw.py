"""Calibrates achievable HBM bandwidth on the box with plain torch ops (copy / fill / read-reduce)."""
import torch, time
dev = "cuda:0"
for mb in (268, 1024):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=LP_DTYPE, device=dev).normal_()
    b = torch.empty_like(a)
    for name, fn, bytes_ in (("copy", lambda: b.copy_(a), 2 * n * 2), ("fill", lambda: b.fill_(1.0), n * 2), ("sum", lambda: a.sum(), n * 2)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 100
        print("%5d MB %5s: %7.1f us  %.2f TB/s" % (mb, name, us, bytes_ / us / 1e6))
