import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd")):
        sys.path.insert(0, p)
    import torch
    from torchreid import hip_ops as ops
    from torchreid._hip import LP_DTYPE
    N, K, Cout, use_res = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dev = "cuda:0"
    torch.manual_seed(0)
    x = torch.relu(torch.randn((N, 16, 8, K), device=dev)).to(LP_DTYPE)
    res = torch.relu(torch.randn((N, 16, 8, Cout), device=dev)).to(LP_DTYPE) if use_res else None
    w = (torch.randn((Cout, 1, 1, K), device=dev) / K ** 0.5).to(LP_DTYPE)
    b = torch.randn((Cout,), device=dev)
    packed = ops.conv1x1_pack(w)
    ref = ops.conv_bn_act(x, w, b, 1, 0, True, residual=res)
    torch.cuda.synchronize()
    out = ops.conv1x1_packed_res(x, packed, b, Cout, res)
    torch.cuda.synchronize()
    d = (out.float() - ref.float()).abs()
    print("N=%d K=%d Cout=%d res=%d nsplit=%s: max diff %.4g, mismatching %d of %d" % (N, K, Cout, use_res, os.environ.get("AGRL_DUO_NSPLIT"), d.max().item(), int((d > 0).sum()), d.numel()))
    sys.exit(0)
for ns in ("8", "4", "1"):
    for (N, K, Cout, r) in ((1, 512, 2048, 1), (1, 512, 2048, 0), (1, 128, 2048, 1), (8, 512, 2048, 1), (256, 512, 2048, 1)):
        env = dict(os.environ, AGRL_DUO_NSPLIT=ns)
        pr = subprocess.run([sys.executable, __file__, str(N), str(K), str(Cout), str(r)], env=env, capture_output=True, text=True, timeout=120)
        tail = [l for l in (pr.stdout + pr.stderr).splitlines() if "amdgpu.ids" not in l]
        print("ns=%s case=%s rc=%d | %s" % (ns, (N, K, Cout, r), pr.returncode, " | ".join(tail[-2:])[:300]), flush=True)
