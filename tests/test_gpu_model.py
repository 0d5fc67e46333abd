"""End-to-end GPU parity of the vmgn eval forward (the drop-in boundary) against the CPU oracle."""
import time

import numpy as np
import pytest
import torch

from lp16 import LP16, LP_DTYPE, LP_EMBED_TOL, LP_STAGE_TOL

from oracle import vmgn_oracle as O
from recipe import recipe_state_dict, synthetic_adj, synthetic_clips

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(num_classes=5, **kw):
    from torchreid import models
    cfg = dict(num_classes=num_classes, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
               pyramid_part=True, use_pose=True, learn_graph=True, consistent_loss=False, num_parts=3, bnneck=True)
    cfg.update(kw)
    m = models.init_model("vmgn", **cfg)
    sd = recipe_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd)
    return m.eval(), sd


def rel(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return ((got - ref).abs().max() / ref.abs().max()).item()


@pytest.mark.parametrize("cfg", [(2, 4, True, True), (1, 8, True, True), (2, 4, True, False), (2, 4, False, True)])
def test_vmgn_eval_fp32_matches_oracle(cfg):
    B, S, use_pose, learn_graph = cfg
    m, sd = build(use_pose=use_pose, learn_graph=learn_graph)
    x, adj = synthetic_clips(B, S, seed=S), synthetic_adj(B, S, seed=S)
    with torch.no_grad():
        ref = O.vmgn_eval(x, adj, sd, use_pose=use_pose, learn_graph=learn_graph)
    m = m.to(DEV)
    m.hip_precision = "fp32"
    got = m(x.to(DEV), adj.to(DEV))
    torch.cuda.synchronize()
    assert got.shape == (B, 4096) and got.dtype == torch.float32 and got.is_cuda
    e = rel(got, ref)
    print("vmgn fp32", cfg, "max rel err %.3e  (|ref|max %.3f, tracklet spread %.3e)" % (
        e, ref.abs().max().item(), (ref[0] - ref[-1]).abs().max().item()))
    assert e < 1e-3  # north-star bar; measured ~1e-5
    # split-bf16 mode (fp32 tensors, every conv / Linear product as three bf16 MFMAs): same bar, 2-3 x the rate
    m.hip_precision = "bf16x3"
    got3 = m(x.to(DEV), adj.to(DEV))
    torch.cuda.synchronize()
    e3 = rel(got3, ref)
    print("vmgn bf16x3", cfg, "max rel err %.3e" % e3)
    assert got3.dtype == torch.float32 and e3 < 1e-3 and not torch.equal(got3, got)
    # split-fp16 mode (round 6: fp32 tensors, conv products as three fp16 MFMAs on fp16 high / low halves, weights pre-scaled by a
    # power of two): 22 significand bits per operand -- held an order of magnitude tighter than the north-star bar, and closer to
    # the exact mode than split-bf16 is
    m.hip_precision = "fp16x3"
    goth = m(x.to(DEV), adj.to(DEV))
    torch.cuda.synchronize()
    eh, dh, d3 = rel(goth, ref), rel(goth, got.cpu()), rel(got3, got.cpu())
    print("vmgn fp16x3", cfg, "max rel err vs oracle %.3e; vs the exact-fp32 HIP forward: fp16x3 %.3e, bf16x3 %.3e" % (eh, dh, d3))
    assert goth.dtype == torch.float32 and eh < 1e-4 and dh < d3 and dh < 2e-6


@pytest.mark.parametrize("cfg", [(3, 5, 128, 64, 4, True), (2, 16, 256, 128, 4, True), (1, 1, 256, 128, 4, True),
                                 (2, 4, 256, 128, 2, True), (2, 4, 256, 128, 4, False), (5, 3, 192, 96, 4, True)])
@pytest.mark.parametrize("precision,tol", [(LP16, LP_EMBED_TOL), ("fp32", 1e-3), ("fp16x3", 1e-4)])
def test_vmgn_eval_shape_variants(cfg, precision, tol):
    """Frame sizes / clip lengths / split counts off the bench configuration: every dispatch (fused pooling or not, wide
    or narrow tiles, streaming or LDS message pass, fused layer-1 tails) must agree with the oracle."""
    B, S, H, W, num_split, pyramid = cfg
    m, sd = build(num_split=num_split, pyramid_part=pyramid)
    x = synthetic_clips(B, S, H=H, W=W, seed=B + S)
    adj = synthetic_adj(B, S, seed=B + S, num_split=num_split, pyramid_part=pyramid)
    with torch.no_grad():
        ref = O.vmgn_eval(x, adj, sd, num_split=num_split, pyramid_part=pyramid)
    m = m.to(DEV)
    m.hip_precision = precision
    got = m(x.to(DEV), adj.to(DEV))
    torch.cuda.synchronize()
    e = rel(got, ref)
    print("vmgn", precision, cfg, "max rel err %.3e" % e)
    assert got.shape == (B, 4096) and e < tol


def test_vmgn_eval_16_bit_mode_close_and_ranking_preserved():
    B, S = 4, 4
    m, sd = build()
    x, adj = synthetic_clips(B, S, seed=9, identities=[0, 0, 1, 2]), synthetic_adj(B, S, seed=9)
    with torch.no_grad():
        ref = O.vmgn_eval(x, adj, sd)
    m = m.to(DEV)
    m.hip_precision = LP16
    got = m(x.to(DEV), adj.to(DEV))
    torch.cuda.synchronize()
    e = rel(got, ref)
    print("vmgn %s max rel err %.3e" % (LP16, e))
    assert e < LP_EMBED_TOL
    # the nearest neighbour of tracklet 0 is its same-identity twin under both precisions
    d_ref = O.cosine(ref, ref) + 10 * torch.eye(B)
    d_got = O.cosine(got.float().cpu(), got.float().cpu()) + 10 * torch.eye(B)
    assert d_ref[0].argmin().item() == 1 and d_got[0].argmin().item() == 1


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), (LP16, LP_EMBED_TOL), ("fp16x3", 1e-4)])
def test_gsta_sibling_eval_matches_oracle(precision, tol):
    """``gsta`` (single layer4 branch, one BNNeck) through the same HIP kernels vs oracle.gsta_eval (pinned on the
    reference's gsta.py by tests/golden/gsta_b2s4.npz)."""
    from torchreid import models
    m = models.init_model("gsta", num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                          pyramid_part=True, use_pose=True, learn_graph=True)
    sd = recipe_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd)
    m.eval()
    m.hip_precision = precision
    for B, S in ((2, 4), (3, 8)):
        x, adj = synthetic_clips(B, S, seed=B + S), synthetic_adj(B, S, seed=B + S)
        with torch.no_grad():
            ref = O.gsta_eval(x, adj, sd)
        got = m.to(DEV)(x.to(DEV), adj.to(DEV))
        torch.cuda.synchronize()
        assert got.shape == (B, 2048)
        err = ((got.cpu().double() - ref.double()).abs().max() / ref.double().abs().max()).item()
        print("gsta", precision, (B, S), "rel err %.3e" % err)
        assert err < tol


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16x3", 1e-3), ("fp16x3", 1e-4), (LP16, LP_EMBED_TOL)])
@pytest.mark.parametrize("variant", ["default_gammas", "pam_on", "pam_and_graph_on"])
def test_ganet_sibling_eval_matches_oracle(variant, precision, tol):
    """``ganet`` (position-attention part nodes, diagonal-masked graph layers, concatenated outputs) through the HIP kernels
    vs oracle.ganet_eval (pinned on the reference's ganet.py by tests/golden/ganet_b2s4.npz): as constructed (both gammas 0:
    nodes = 2 x slice means, graph layers pass their input through), with the attention module's gamma set (the value conv
    folded into one Linear on the attention-weighted slice mean), and with the graph layers' gamma set as well (masked
    graphs, input + gamma h')."""
    from torchreid import models
    m = models.init_model("ganet", num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1, knn=4,
                          pyramid_part=True, use_pose=True, learn_graph=True, pretrained=False)
    sd = recipe_state_dict(m.state_dict(), seed=0)
    if variant == "default_gammas":
        sd["pam_layer.gamma"] = torch.zeros(1)
    m.load_state_dict(sd)
    graph_gamma = 0.1 if variant == "pam_and_graph_on" else 0.0
    for layer in m.graph_layers:
        layer.gamma = graph_gamma
    m.eval()
    m.hip_precision = precision
    for B, S in ((2, 4), (3, 8)):
        x, adj = synthetic_clips(B, S, seed=B + S), synthetic_adj(B, S, seed=B + S)
        with torch.no_grad():
            ref = O.ganet_eval(x, adj, sd, graph_gamma=graph_gamma)
        got = m.to(DEV)(x.to(DEV), adj.to(DEV))
        torch.cuda.synchronize()
        assert got.shape == (B, 3 * 2048)
        err = rel(got, ref)
        print("ganet", variant, precision, (B, S), "rel err %.3e" % err)
        assert err < tol


@pytest.fixture(scope="module")
def bench_size_oracle():
    """The CPU oracle at the BENCHMARKED size (BASELINE configs[1]: 32 tracklets x 8 frames of 256 x 128), every stage."""
    B, S = 32, 8
    m, sd = build()
    x, adj = synthetic_clips(B, S, seed=32), synthetic_adj(B, S, seed=32)
    t0 = time.time()
    with torch.no_grad():
        x4_1, x4_2 = O.featuremaps(x.view(B * S, 3, 256, 128), sd)
        out, parts = O.tail(x4_1, x4_2, adj, sd, B, S, [4, 2, 1], 2, return_parts=True)
        parts["G0"] = O.graph_matrix(parts["nodes"], adj)
        parts["out"] = out
    print("oracle at B=32 S=8: %.1f s on %d threads" % (time.time() - t0, torch.get_num_threads()))
    return m, x, adj, parts


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16x3", 1e-3), ("fp16x3", 1e-3), (LP16, LP_STAGE_TOL)])
def test_vmgn_eval_at_benchmarked_size_stage_by_stage(bench_size_oracle, precision, tol):
    """B = 32, S = 8: the dispatch bench.py times (256 x 256 / 256 x 128 wide tiles, persistent forms, the two-block 3x3
    kernel, pool-fused last convs -- chosen by tile counts that B <= 5 never reaches) against the oracle, stage by stage:
    global feature (x4_1 mean), part nodes (x4_2), the first layer's graph, graph output, attention feature, embedding.
    fp32, the split-bf16 mode and the fp16 throughput mode meet the north-star bar 1e-3; the bf16 build's throughput mode is held to 1e-2."""
    from torchreid.models._vmgn_hip import hip_forward
    m, x, adj, ref = bench_size_oracle
    B, S = x.shape[:2]
    m = m.to(DEV)
    m.hip_precision = precision
    m.invalidate_hip_cache()
    stages = {}
    got = hip_forward(m, x.to(DEV), adj.to(DEV), stages=stages)
    torch.cuda.synchronize()
    g_f = stages["gsum"].view(B, S, -1).sum(1) / (S * stages["hw"])
    errs = {"g_f(x4_1)": rel(g_f, ref["g_f"]), "nodes(x4_2)": rel(stages["nodes"].view(B, S * 7, -1), ref["nodes"]),
            "nodes_out": rel(stages["nodes_out"], ref["nodes_out"]), "att_f": rel(stages["att_f"], ref["att_f"]),
            "embedding": rel(got, ref["out"])}
    assert torch.equal(stages["g_f"], g_f) or rel(stages["g_f"], g_f) < 1e-6

    def offdiag_profile(G):   # see test_graph_layer: independent of the (documented) diagonal deviation
        G = G.detach().cpu().float()
        learned = 2 * G - torch.nn.functional.normalize(adj, p=1, dim=2)
        learned = learned - torch.diag_embed(learned.diagonal(dim1=1, dim2=2))
        return learned / learned.sum(dim=2, keepdim=True).clamp(min=1e-30)
    if precision != LP16:   # in bf16 mode the graph is computed from nodes that already carry the trunk's bf16 error
        errs["G0 offdiag profile"] = rel(offdiag_profile(stages["G0"]), offdiag_profile(ref["G0"]))
    rows = torch.nn.functional.cosine_similarity(got.cpu().double(), ref["out"].double(), dim=1)
    print("vmgn %s B=32 S=8 vs oracle: %s | min cosine(embedding row, oracle row) %.8f" % (
        precision, ", ".join("%s %.2e" % kv for kv in errs.items()), rows.min().item()))
    for name, e in errs.items():
        assert e < tol, (name, e)
    m.hip_precision = "fp32"


def test_under_data_parallel_wrapper():
    """The driver wraps the model in nn.DataParallel (train_vidreid_xent_htri.py:318) and calls it from there in test() and in
    train(): eval forward through the wrapper equals the oracle, and a train forward + backward through the wrapper (replica
    thread, scatter / gather around the native nodes) leaves gradients on the wrapped module's parameters."""
    m, sd = build(num_classes=6, consistent_loss=True)
    x, adj = synthetic_clips(4, 6, 64, 32, seed=9), synthetic_adj(4, 6, seed=9)
    with torch.no_grad():
        ref = O.vmgn_eval(x, adj, sd)
    m = m.to(DEV)
    m.hip_precision = "fp32"
    dp = torch.nn.DataParallel(m, device_ids=[0])
    got = dp(x.to(DEV), adj.to(DEV))
    assert tuple(got.shape) == (4, 4096) and rel(got, ref) < 1e-3
    dp.train()
    from torchreid import losses
    outs, feats = dp(x.to(DEV), adj.to(DEV))
    pids = torch.tensor([0, 0, 1, 1], device=DEV)
    loss = (losses.DeepSupervision(losses.CrossEntropyLabelSmooth(6, use_gpu=True), outs, pids) +
            losses.DeepSupervision(losses.TripletLoss(0.3, True), feats, pids))
    loss.backward()
    grads = [p.grad for p in m.parameters() if p.requires_grad]
    assert all(g is not None and torch.isfinite(g).all() for g in grads) and float(loss.detach()) > 0
    m.eval()


def test_weight_cache_tracks_parameter_updates():
    m, sd = build()
    m = m.to(DEV)
    x, adj = synthetic_clips(1, 4).to(DEV), synthetic_adj(1, 4).to(DEV)
    a = m(x, adj).clone()
    with torch.no_grad():
        m.att_bottleneck.weight.mul_(2.0)
    b = m(x, adj)
    torch.cuda.synchronize()
    assert not torch.allclose(a[:, 2048:], b[:, 2048:])
    assert torch.allclose(a[:, :2048], b[:, :2048])


def test_bad_adj_shape_raises():
    m, _ = build()
    m = m.to(DEV)
    with pytest.raises(ValueError):
        m(synthetic_clips(1, 4).to(DEV), torch.ones(1, 20, 20, device=DEV))


def test_model_takes_the_bit_packed_adjacency():
    """model(x, adj) with adj as the bit-packed int32 (B, V, ceil(V/32)) tensor (packed on the host as a loader would, 448 B per
    tracklet over PCIe instead of 12.5 KB) == the same call with the reference's fp32 (B, V, V) adjacency, bitwise; a packed
    tensor of the wrong shape raises."""
    from torchreid import hip_ops as ops
    m, _ = build()
    m = m.to(DEV)
    m.hip_precision = "fp32"
    x, adj = synthetic_clips(3, 8, seed=21), synthetic_adj(3, 8, seed=21)
    ref = m(x.to(DEV), adj.to(DEV))
    bits = ops.adjacency_pack_host(adj)
    assert bits.dtype == torch.int32 and tuple(bits.shape) == (3, 56, 2) and bits.numel() * 4 * 28 == adj.numel() * 4
    got = m(x.to(DEV), bits.to(DEV))
    assert torch.equal(ref, got)
    with pytest.raises(ValueError):
        m(x.to(DEV), bits[:, :, :1].contiguous().to(DEV))


def test_throughput_probe():
    """Not an assertion on speed: prints a first timing of the full forward at the BASELINE config-2 shape."""
    m, _ = build()
    m = m.to(DEV)
    m.hip_static_weights = True
    B, S = 32, 8
    x = torch.randn(B, S, 3, 256, 128, device=DEV)
    adj = synthetic_adj(B, S).to(DEV)
    for prec in ("fp32", LP16):
        m.hip_precision = prec
        for _ in range(2):
            m(x, adj)
        torch.cuda.synchronize()
        t = time.time()
        n = 5
        for _ in range(n):
            m(x, adj)
        torch.cuda.synchronize()
        dt = (time.time() - t) / n
        print("forward B=32 S=8 %s: %.2f ms  -> %.0f frames/s" % (prec, dt * 1e3, B * S / dt))


def test_16_bit_build_with_undamped_batchnorm_statistics():
    """The recipe damps every bn3 / downsample gamma to 0.15-0.35; a trained ResNet50 has them around 1. With gamma in 0.8 .. 1.2 on
    EVERY block (tests/recipe.imagenet_like_state_dict) the residual stream grows block by block: this test puts numbers on the fp16
    build's headroom. Asserted: the oracle's largest activation of the whole trunk stays far below fp16's 65504 (so the range is not
    what the recipe's damping was hiding: with RANDOM conv weights every undamped block doubles the stream -- 15 after the first
    block, 2.8 k after layer 3, 13.7 k at the end of layer 4, a growth no trained network has -- and that still fits), the 16-bit
    forward is finite, and it is as close to the oracle as with the damped recipe (fp16: the north star's 1e-3; bf16 build: its
    usual bar). If a checkpoint ever did leave the range, the guards of
    tests/test_gpu_eval.py::test_out_of_range_activations_fail_loudly_in_the_16_bit_mode fire with the layer named."""
    from recipe import imagenet_like_state_dict
    from torchreid import models
    B, S = 2, 4
    m = models.init_model("vmgn", num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                          pyramid_part=True, use_pose=True, learn_graph=True)
    sd = imagenet_like_state_dict(m.state_dict(), seed=0)
    m.load_state_dict(sd)
    m.eval()
    x, adj = synthetic_clips(B, S, seed=12), synthetic_adj(B, S, seed=12)
    peaks = {}

    def hook(name):
        def f(mod, inp, out):
            peaks[name] = max(peaks.get(name, 0.0), float(out.detach().abs().max()))
        return f
    hs = [mod.register_forward_hook(hook(n)) for n, mod in m.named_modules() if n and n.count(".") <= 1 and n.startswith("layer")]
    with torch.no_grad():
        ref_module = m(x, adj)
        ref = O.vmgn_eval(x, adj, sd)
    for h in hs:
        h.remove()
    # the trunk half of the embedding (global branch) is what the range question is about. The graph half is ill-conditioned in this
    # regime: with part features of magnitude 1e4 every pairwise distance overflows exp(), the learned graph is 0 / 0-like noise, and
    # two fp32 restatements of the reference (this build's module tree and the oracle) already differ by 1e-2 there.
    C = ref.shape[1] // 2
    assert rel(ref_module[:, :C], ref[:, :C]) < 1e-5
    peak = max(peaks.values())
    per_layer = {k: round(v, 1) for k, v in peaks.items() if k.count(".") == 0}
    print("undamped BatchNorm statistics: largest activation per stage", per_layer, "-> %.1f of 65504 (graph half: module vs oracle %.1e)" % (
        peak, rel(ref_module[:, C:], ref[:, C:])))
    assert peak < 65504 / 2, peaks
    m = m.to(DEV)
    m.hip_precision = LP16
    got = m(x.to(DEV), adj.to(DEV))
    m.hip_precision = "fp32"
    got32 = m(x.to(DEV), adj.to(DEV))
    torch.cuda.synchronize()
    e, e32 = rel(got[:, :C], ref[:, :C]), rel(got32[:, :C], ref[:, :C])
    eg, eg32 = rel(got[:, C:], ref[:, C:]), rel(got32[:, C:], ref[:, C:])
    print("undamped BatchNorm statistics: trunk half %s max rel err %.3e, fp32 %.3e; graph half %.3e / %.3e" % (LP16, e, e32, eg, eg32))
    assert torch.isfinite(got).all() and e32 < 1e-3 and e < LP_EMBED_TOL
    assert eg32 < 5e-2 and eg < 1e-1
    # the conforming mode in the same regime: activations up to 13.7 k sit 5 x below fp16's range (the high half), their low halves
    # times 2^11 below 2^15 -- the split planes and the in-loop split hold the exact mode's accuracy there too
    m.hip_precision = "fp16x3"
    got3 = m(x.to(DEV), adj.to(DEV))
    torch.cuda.synchronize()
    e3 = rel(got3[:, :C], ref[:, :C])
    print("undamped BatchNorm statistics: trunk half fp16x3 max rel err %.3e (exact fp32 %.3e)" % (e3, e32))
    assert torch.isfinite(got3).all() and e3 < 4 * e32 + 1e-6
