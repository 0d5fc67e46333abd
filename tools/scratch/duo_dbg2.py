import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
        sys.path.insert(0, p)
    import torch
    from torchreid import hip_ops as ops
    from torchreid._hip import LP_DTYPE
    mode = sys.argv[1]
    N, K, Cout = 256, 512, 2048
    dev = "cuda:0"
    torch.manual_seed(0)
    x = torch.relu(torch.randn((N, 16, 8, K), device=dev)).to(LP_DTYPE)
    res = torch.relu(torch.randn((N, 16, 8, Cout), device=dev)).to(LP_DTYPE)
    w = (torch.randn((Cout, 1, 1, K), device=dev) / K ** 0.5).to(LP_DTYPE)
    b = torch.randn((Cout,), device=dev)
    packed = ops.conv1x1_pack(w)
    torch.cuda.synchronize()
    if mode == "twice":
        a = ops.conv1x1_packed_res(x, packed, b, Cout, res)
        c = ops.conv1x1_packed_res(x, packed, b, Cout, res)
        torch.cuda.synchronize()
        print("twice ok", torch.equal(a, c))
    elif mode == "duo_then_wide":
        a = ops.conv1x1_packed_res(x, packed, b, Cout, res)
        c = ops.conv_bn_act(x, w, b, 1, 0, True, residual=res)
        torch.cuda.synchronize()
        print("duo_then_wide ok", torch.equal(a, c))
    elif mode == "loop20":
        for i in range(20):
            a = ops.conv1x1_packed_res(x, packed, b, Cout, res)
        torch.cuda.synchronize()
        print("loop20 ok")
    elif mode == "loop20sync":
        for i in range(20):
            a = ops.conv1x1_packed_res(x, packed, b, Cout, res)
            torch.cuda.synchronize()
        print("loop20sync ok")
    elif mode == "pool":
        a = ops.conv1x1_packed_res_pool(x, packed, b, Cout, res, [4, 2, 1], True, True)
        torch.cuda.synchronize()
        c = ops.conv1x1_bn_act_pool(x, w, b, res, [4, 2, 1], True, True)
        torch.cuda.synchronize()
        print("pool ok", torch.equal(a[0], c[0]), torch.equal(a[1], c[1]))
    elif mode == "small_twice":
        x = x[:3].contiguous(); res = res[:3].contiguous()
        a = ops.conv1x1_packed_res(x, packed, b, Cout, res)
        c = ops.conv1x1_packed_res(x, packed, b, Cout, res)
        torch.cuda.synchronize()
        print("small twice ok", torch.equal(a, c))
    sys.exit(0)
for mode in ("twice", "duo_then_wide", "loop20sync", "loop20", "pool", "small_twice"):
    for extra in ({}, {"AMD_SERIALIZE_KERNEL": "3"}):
        env = dict(os.environ, **extra)
        pr = subprocess.run([sys.executable, __file__, mode], env=env, capture_output=True, text=True, timeout=120)
        tail = [l for l in (pr.stdout + pr.stderr).splitlines() if "amdgpu.ids" not in l]
        print("%s %s rc=%d | %s" % (mode, extra, pr.returncode, " | ".join(tail[-2:])[:300]), flush=True)
