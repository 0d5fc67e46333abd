"""Pins the CPU oracle (oracle/vmgn_oracle.py) to golden vectors captured from the reference implementation
itself (tests/golden/make_golden.py, run in the build container where /root/reference is mounted).
No GPU, no reference needed at test time."""
import os

import numpy as np
import pytest
import torch

from oracle import vmgn_oracle as O
from recipe import recipe_state_dict, recipe_tensor, synthetic_adj, synthetic_clips

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def close(a, b, tol):
    a, b = torch.as_tensor(np.asarray(a)).double(), torch.as_tensor(np.asarray(b)).double()
    err = ((a - b).abs().max() / b.abs().max().clamp(min=1e-30)).item()
    assert err < tol, err
    return err


def vmgn_state_dict():
    from torchreid import models
    m = models.init_model("vmgn", num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2,
                          num_scale=1, pyramid_part=True, use_pose=True, learn_graph=True)
    return m, recipe_state_dict(m.state_dict(), seed=0)


@pytest.mark.parametrize("tag", ["v28", "v56", "v56_pose", "v56_learn", "v112"])
def test_graph_layer(tag):
    z = gold("graph_layer_" + tag)
    B, V, C, use_pose, learn, fseed, wseed = [int(v) for v in z["meta"]]
    sd = recipe_state_dict({"gl.bn.weight": (C,), "gl.bn.bias": (C,), "gl.bn.running_mean": (C,), "gl.bn.running_var": (C,)}, seed=wseed)
    # the generator drew BN tensors under the keys 'bn.*' and the weight under 'graph_layers.0.linear.weight'
    for leaf in ("weight", "bias", "running_mean", "running_var"):
        sd["gl.bn." + leaf] = recipe_tensor("bn." + leaf, (C,), wseed)
    sd["gl.linear.weight"] = recipe_tensor("graph_layers.0.linear.weight", (C, C), wseed)
    g = torch.Generator().manual_seed(fseed)
    f = torch.rand((B, 1, C), generator=g) + 0.02 * torch.randn((B, V, C), generator=g)
    if z["f"].size:
        assert np.array_equal(z["f"], f.numpy())
    adj = torch.from_numpy(z["adj"])
    assert torch.equal(adj, synthetic_adj(B, V // 7, seed=V))
    close(O.sim_matrix(f), z["sim"], 2e-2)  # the fp32 diagonal is cancellation noise (see test_gpu_kernels)
    offdiag = ~np.eye(V, dtype=bool)[None].repeat(B, 0)
    close(O.sim_matrix(f).numpy()[offdiag], z["sim"][offdiag], 1e-3)
    close(O.graph_layer(f, adj, sd, "gl", bool(use_pose), bool(learn)), z["out"], 1e-5)


def test_tail():
    z = gold("tail")
    B, S, c, h, w, seed = [int(v) for v in z["meta"]]
    _, sd = vmgn_state_dict()
    g = torch.Generator().manual_seed(seed)
    x41 = torch.rand((B * S, c, h, w), generator=g)
    x42 = torch.rand((B * S, c, h, w), generator=g)
    adj = synthetic_adj(B, S, seed=seed)
    out, parts = O.tail(x41, x42, adj, sd, B, S, [4, 2, 1], 2, return_parts=True)
    close(parts["g_f"], z["g_f"], 1e-6)
    close(parts["nodes"][:, :, :64], z["nodes"], 1e-6)
    # uniform-random maps give near-identical part nodes: the learned graph is then sensitive to the fp32
    # cancellation noise of d2 (restatement and reference reduce in different orders) -> 1e-4 instead of 1e-5
    close(parts["nodes_out"][:, :, :64], z["nodes_out"], 1e-4)
    close(parts["att_f"], z["att_f"], 1e-4)
    close(out, z["out"], 1e-4)


@pytest.mark.parametrize("tag", ["b2s4", "b1s8"])
def test_vmgn_eval(tag):
    z = gold("vmgn_eval_" + tag)
    B, S, seed, wseed = [int(v) for v in z["meta"]]
    m, sd = vmgn_state_dict()
    x, adj = synthetic_clips(B, S, seed=seed), synthetic_adj(B, S, seed=seed)
    with torch.no_grad():
        x4_1, x4_2 = O.featuremaps(x.view(B * S, 3, 256, 128), sd)
        close(x4_1.mean(dim=(2, 3)), z["x4_1_mean"], 1e-5)
        close(x4_2.mean(dim=(2, 3)), z["x4_2_mean"], 1e-5)
        close(O.tail(x4_1, x4_2, adj, sd, B, S, [4, 2, 1], 2), z["out"], 1e-5)
        # the drop-in module's stock-torch path (what runs for CPU tensors) gives the same answer
        m.load_state_dict(sd)
        m.eval()
        close(m(x, adj), z["out"], 1e-5)


def test_vmgn_train_outputs_match_reference():
    z = gold("vmgn_train_b2s8")
    B, S, seed, rng_seed = [int(v) for v in z["meta"]]
    from torchreid import models
    m = models.init_model("vmgn", num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2,
                          num_scale=1, pyramid_part=True, use_pose=True, learn_graph=True, consistent_loss=True)
    m.load_state_dict(recipe_state_dict(m.state_dict(), seed=0))
    m.train()
    x, adj = synthetic_clips(B, S, seed=seed), synthetic_adj(B, S, seed=seed)
    torch.manual_seed(rng_seed)
    outs, feats = m(x, adj)
    assert len(outs) == 5 and len(feats) == 5
    for i in range(5):
        close(outs[i].detach(), z["logit%d" % i], 1e-4)
        close(feats[i].detach(), z["feat%d" % i], 1e-4)


def test_gsta_sibling_model_matches_reference():
    """The single-branch sibling ``gsta``: oracle.gsta_eval and this build's module tree against the
    reference's gsta.py -- eval output, state-dict keys/shapes, and train-mode outputs with the consistent loss (whose
    dropped frames come from numpy's global RNG)."""
    from torchreid import models
    z = gold("gsta_b2s4")
    m = models.init_model("gsta", num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                          pyramid_part=True, use_pose=True, learn_graph=True, consistent_loss=True)
    own = m.state_dict()
    assert sorted(own.keys()) == list(z["keys"])
    assert [str(tuple(own[k].shape)) for k in sorted(own.keys())] == list(z["shapes"])
    assert not m.bottleneck.bias.requires_grad
    sd = recipe_state_dict(own, seed=0)
    m.load_state_dict(sd)
    x, adj = synthetic_clips(2, 4, seed=4), synthetic_adj(2, 4, seed=4)
    with torch.no_grad():
        close(O.gsta_eval(x, adj, sd), z["out"], 1e-5)
        m.eval()
        close(m(x, adj), z["out"], 1e-5)
    m.train()
    np.random.seed(123)
    outs, feats = m(synthetic_clips(2, 8, seed=8), synthetic_adj(2, 8, seed=8))
    close(torch.stack([o.detach() for o in outs]), z["train_logits"], 1e-4)
    close(torch.stack([f.detach() for f in feats]), z["train_feats"], 1e-4)


def test_ganet_sibling_model_matches_reference():
    """The sibling ``ganet``: oracle.ganet_eval / oracle.pam_module and this build's module tree against
    the reference's ganet.py -- eval output with the graph layers' default gamma = 0 AND with gamma = 0.1 (the diagonal-masked
    graph arithmetic made visible), the position-attention module on one pyramid slice, state-dict keys / shapes, and
    train-mode outputs with the consistent loss. The factory is constructible through init_model with ``knn`` passed."""
    from torchreid import models
    z = gold("ganet_b2s4")
    assert "ganet" in models.get_names()
    with pytest.raises(TypeError):   # reference ganet.py:458: knn is a required positional the driver never passes
        models.init_model("ganet", num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                          pyramid_part=True, use_pose=True, learn_graph=True)
    m = models.init_model("ganet", num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                          knn=4, pyramid_part=True, use_pose=True, learn_graph=True, pretrained=False, consistent_loss=True)
    own = m.state_dict()
    assert sorted(own.keys()) == list(z["keys"])
    assert [str(tuple(own[k].shape)) for k in sorted(own.keys())] == list(z["shapes"])
    assert not m.bottleneck.bias.requires_grad and float(m.pam_layer.gamma.detach()) == 0.0 and m.graph_layers[0].gamma == 0
    sd = recipe_state_dict(own, seed=0)
    m.load_state_dict(sd)
    x, adj = synthetic_clips(2, 4, seed=6), synthetic_adj(2, 4, seed=6)
    with torch.no_grad():
        close(O.ganet_eval(x, adj, sd), z["out_gamma0"], 1e-5)
        close(O.ganet_eval(x, adj, sd, graph_gamma=0.1), z["out_gamma01"], 1e-5)
        assert np.abs(z["out_gamma0"] - z["out_gamma01"]).max() > 1e-3     # the second fixture does exercise the graph
        fm = O.stage(O.stem(x.view(8, 3, 256, 128), sd), sd, "layer1", 3, 1)
        for name, blocks, stride in O.RESNET50_STAGES[1:]:
            fm = O.stage(fm, sd, name, blocks, stride)
        fm = O.stage(fm, sd, "layer4", 3, 1)
        sl = fm[:, :, 4:8]
        close(O.pam_module(sl, sd, "pam_layer").mean(dim=(2, 3)), z["pam_slice_mean"], 1e-5)   # gamma * attended + slice
        m.eval()
        close(m(x, adj), z["out_gamma0"], 1e-5)
        for layer in m.graph_layers:
            layer.gamma = 0.1
        close(m(x, adj), z["out_gamma01"], 1e-5)
        for layer in m.graph_layers:
            layer.gamma = 0
    m.train()
    np.random.seed(321)
    outs, feats = m(synthetic_clips(2, 6, seed=9), synthetic_adj(2, 6, seed=9))
    close(torch.stack([o.detach() for o in outs]), z["train_logits"], 1e-4)
    close(torch.stack([f.detach() for f in feats]), z["train_feats"], 1e-4)


def test_distmat():
    z = gold("distmat")
    m, n, D, seed = [int(v) for v in z["meta"]]
    g = torch.Generator().manual_seed(seed)
    q, gal = torch.randn((m, D), generator=g), torch.randn((n, D), generator=g)
    close(O.euclidean_squared(q, gal), z["euclidean"], 1e-6)
    close(O.cosine(q, gal), z["cosine"], 1e-6)
    from torchreid.metrics import distance
    if not torch.cuda.is_available():  # the host path of the drop-in (no GPU present)
        close(distance.compute_distance_matrix(q, gal, "euclidean"), z["euclidean"], 1e-6)
        close(distance.compute_distance_matrix(q, gal, "cosine"), z["cosine"], 1e-6)


def test_rank_mars_bit_exact():
    z = gold("rank_mars")
    cmc, mAP, ap, _, _ = O.evaluate_mars(z["dist"], z["q_pids"], z["g_pids"], z["q_camids"], z["g_camids"], 50, return_all=True)
    assert mAP == float(z["mAP"])
    assert np.array_equal(cmc, z["cmc"])
    assert np.array_equal(ap, z["ap"])
    if not torch.cuda.is_available():
        from torchreid import metrics
        cmc2, mAP2 = metrics.evaluate_rank(z["dist"], z["q_pids"], z["g_pids"], z["q_camids"], z["g_camids"], use_metric_mars=True)
        assert mAP2 == float(z["mAP"]) and np.array_equal(cmc2, z["cmc"])


def test_rank_market1501_matches_reference_python_and_cython():
    """oracle.eval_market1501 against the reference's python evaluator (golden fixture, rank.py:95-150) and, when it has
    been built from /root/reference (oracle/build_ref.py -> oracle/_ref/), against the reference's Cython evaluator."""
    z = gold("rank_market1501")
    args = (z["dist"], z["q_pids"], z["g_pids"], z["q_camids"], z["g_camids"])
    cmc, mAP, ap, first = O.eval_market1501(*args, 50, return_all=True)
    assert np.array_equal(cmc, z["cmc"])                      # CMC: integer counts / num_valid -> exact
    assert abs(mAP - float(z["mAP"])) < 1e-14                 # fp64 sums in a different association order
    assert np.isnan(ap).sum() == int(np.isin(z["q_pids"], [12, 13]).sum()) > 0   # identities absent from the gallery
    if not torch.cuda.is_available():
        from torchreid import metrics
        cmc2, mAP2 = metrics.evaluate_rank(*args, use_metric_market1501=True)
        assert np.array_equal(cmc2, z["cmc"]) and abs(mAP2 - float(z["mAP"])) < 1e-14
    # the reference's NATIVE evaluator (rank_cylib/rank_cy.pyx:154-241): its outputs on seeded inputs were captured by
    # make_golden.py (F14) -- the compiled extension is not needed here and never travels to the GPU box
    zc = gold("rank_market1501_cy")
    for m, n in ((40, 500), (64, 12180)):
        d, q_pids, g_pids, q_cam, g_cam = market1501_case(m, n)
        assert abs(d.astype(np.float64).sum() - float(zc["dist_checksum_%dx%d" % (m, n)])) < 1e-9
        cmc, mAP = O.eval_market1501(d, q_pids, g_pids, q_cam, g_cam, 50)
        # these inputs contain exact distance ties (columns 3 and 5): the reference orders them by numpy's default UNSTABLE
        # argsort (rank.py:105), the oracle and the product by the stable order -- the tie-free fixture above pins 1e-14
        assert np.allclose(cmc, zc["cmc_py_%dx%d" % (m, n)], atol=1e-6) and abs(mAP - float(zc["mAP_py_%dx%d" % (m, n)])) < 1e-6
        assert np.allclose(zc["cmc_cy_%dx%d" % (m, n)], cmc, atol=1e-6)          # the Cython twin accumulates in fp32
        assert abs(float(zc["mAP_cy_%dx%d" % (m, n)]) - mAP) < 1e-6
    from oracle import build_ref
    cy = build_ref.load()
    if cy is not None:   # build container only: the live extension agrees with its captured outputs
        i64 = [np.ascontiguousarray(a, dtype=np.int64) for a in args[1:]]
        cmc_cy, mAP_cy = cy.eval_market1501_cy(np.ascontiguousarray(z["dist"], dtype=np.float32), i64[0], i64[1], i64[2], i64[3], 50)
        assert np.allclose(cmc_cy, z["cmc"], atol=1e-6) and abs(mAP_cy - float(z["mAP"])) < 1e-6


def market1501_case(m, n):
    """The seeded inputs of make_golden.py F14 / tests/test_gpu_kernels.py::test_rank_market1501_device."""
    rng = np.random.RandomState(m + n)
    d = rng.rand(m, n).astype(np.float32)
    d[:, 5] = d[:, 3]
    npid = max(4, n // 40)
    q_pids, g_pids = rng.randint(0, npid + 2, m), rng.randint(0, npid, n)
    q_cam, g_cam = rng.randint(0, 6, m), rng.randint(0, 6, n)
    return d, q_pids, g_pids, q_cam, g_cam


def xent_case():
    z = gold("xent")
    n_out, n, K, seed = [int(v) for v in z["meta"]]
    g = torch.Generator().manual_seed(seed)
    logits = [3.0 * torch.randn((n, K), generator=g) for _ in range(n_out)]
    pids = torch.randint(0, K, (n,), generator=g)
    assert torch.equal(logits[0], torch.from_numpy(z["logits0"])) and torch.equal(pids, torch.from_numpy(z["pids"]))
    return z, logits, pids, K


def test_xent_label_smooth_matches_reference():
    """oracle.xent_label_smooth / deep_supervision and this build's CPU CrossEntropyLabelSmooth + DeepSupervision against
    the reference's (losses/cross_entropy_loss.py:26-37, losses/__init__.py:9-20): value over a five-logit list and the
    gradient w.r.t. the first logits, for eps in {0.1, 0, 0.3}."""
    from torchreid import losses
    z, logits, pids, K = xent_case()
    for eps in (0.1, 0.0, 0.3):
        tag = "eps%02d" % int(round(eps * 100))
        xs = [x.clone().requires_grad_(True) for x in logits]
        loss = O.deep_supervision(lambda a, b: O.xent_label_smooth(a, b, eps), xs, pids)
        loss.backward()
        close(loss.detach(), z["loss_" + tag], 1e-6)
        close(O.xent_label_smooth(logits[0], pids, eps), z["single_" + tag], 1e-6)
        close(xs[0].grad, z["grad0_" + tag], 1e-5)
        xs = [x.clone().requires_grad_(True) for x in logits]
        crit = losses.CrossEntropyLabelSmooth(num_classes=K, epsilon=eps, use_gpu=False)
        loss = losses.DeepSupervision(crit, xs, pids)
        loss.backward()
        close(loss.detach(), z["loss_" + tag], 1e-6)
        close(xs[0].grad, z["grad0_" + tag], 1e-5)


BACKWARD_KEYS = ("conv1.weight", "bn1.weight", "layer1.0.conv1.weight", "layer1.0.bn3.bias", "layer2.0.downsample.0.weight",
                 "layer3.5.conv3.weight", "layer3.5.bn2.weight", "layer4_1.2.conv3.weight", "layer4_2.0.conv2.weight",
                 "layer4_2.2.bn3.weight", "graph_layers.0.linear.weight", "graph_layers.1.linear.weight", "graph_layers.1.bn.weight",
                 "global_bottleneck.weight", "att_bottleneck.weight", "global_classifier.weight", "att_classifier.weight")


def backward_case(device="cpu"):
    """The train step of make_golden.py F15 on this build's model: -> (fixture, model after loss.backward(), loss, xent, htri)."""
    from torchreid import losses, models
    z = gold("vmgn_backward_b4s8")
    B, S, seed_x, seed_t, seed_w = [int(v) for v in z["meta"]]
    m = models.init_model("vmgn", num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                          pyramid_part=True, use_pose=True, learn_graph=True, consistent_loss=True)
    m.load_state_dict(recipe_state_dict(m.state_dict(), seed=seed_w))
    m = m.to(device)
    m.train()
    pids = torch.tensor([0, 0, 1, 1])
    x, adj = synthetic_clips(B, S, seed=seed_x, identities=pids.tolist()), synthetic_adj(B, S, seed=seed_x)
    torch.manual_seed(seed_t)
    outs, feats = m(x.to(device), adj.to(device))
    pd = pids.to(device)
    lx = losses.DeepSupervision(losses.CrossEntropyLabelSmooth(num_classes=5, use_gpu=device != "cpu"), outs, pd)
    lt = losses.DeepSupervision(losses.TripletLoss(margin=0.3, soft=True), feats, pd)
    loss = lx + lt
    loss.backward()
    return z, m, loss, lx, lt


def gradient_errors(z, model):
    """Per fixture key: relative error of the stored slice in the max norm, of the whole tensor's L2 norm, and of the stored
    slice in the L2 norm."""
    named = dict(model.named_parameters())
    out = {}
    for key in BACKWARD_KEYS:
        ref = torch.from_numpy(z["g:" + key]).double()
        got = named[key].grad.detach().double().cpu()
        got = got[:ref.shape[0]] if got.shape != ref.shape else got
        out[key] = (((got - ref).abs().max() / ref.abs().max().clamp(min=1e-30)).item(),
                    abs(named[key].grad.detach().double().norm().item() - float(z["n:" + key])) / max(float(z["n:" + key]), 1e-30),
                    ((got - ref).norm() / ref.norm().clamp(min=1e-30)).item())
    return out


@pytest.mark.timeout(900)
def test_train_step_backward_matches_reference_gradients():
    """loss.backward() of one xent + htri step with the consistent loss (B = 4, S = 8): this build's CPU module tree against
    the gradients the REFERENCE's own model produced for the same weights / clips / frame subsets
    (train_vidreid_xent_htri.py:397-411; make_golden.py F15)."""
    z, m, loss, lx, lt = backward_case("cpu")
    close(loss.detach(), z["loss"], 1e-5)
    close(lx.detach(), z["xent"], 1e-5)
    close(lt.detach(), z["htri"], 1e-5)
    errs = gradient_errors(z, m)
    for key, (e_slice, e_norm, e_l2) in errs.items():
        print("%-34s slice max %.2e L2 %.2e | tensor norm %.2e" % (key, e_slice, e_l2, e_norm))
    worst = max(errs.items(), key=lambda kv: kv[1][0])
    print("worst slice error %.2e (%s)" % (worst[1][0], worst[0]))
    for key, (e_slice, e_norm, e_l2) in errs.items():
        assert e_slice < 1e-4 and e_norm < 1e-5 and e_l2 < 1e-4, (key, e_slice, e_norm, e_l2)   # measured: <= 1.3e-5 / 9e-7


def test_rank_cuhk03_matches_reference_and_rng_stream():
    """oracle.eval_cuhk03 (and, on a host without GPU, this build's evaluate_rank(use_metric_cuhk03=True)) against the
    reference's python evaluator (rank.py:22-92) under the same np.random seed: CMC bit-exact, mAP to fp64 rounding, and
    numpy's global RNG left in the same state (the next draw agrees)."""
    z = gold("rank_cuhk03")
    args = (z["dist"], z["q_pids"], z["g_pids"], z["q_camids"], z["g_camids"])
    assert int(np.isin(z["q_pids"], [60, 61, 62, 63]).sum()) > 0          # invalid queries are part of the case
    for max_rank in (50, 20):
        np.random.seed(int(z["seed"]))
        cmc, mAP = O.eval_cuhk03(*args, max_rank=max_rank)
        assert np.random.randint(0, 1 << 30) == int(z["next_draw_%d" % max_rank])
        assert np.array_equal(cmc, z["cmc_%d" % max_rank]) and abs(mAP - float(z["mAP_%d" % max_rank])) < 1e-14
        if not torch.cuda.is_available():
            from torchreid import metrics
            np.random.seed(int(z["seed"]))
            cmc2, mAP2 = metrics.evaluate_rank(*args, max_rank=max_rank, use_metric_cuhk03=True)
            assert np.random.randint(0, 1 << 30) == int(z["next_draw_%d" % max_rank])
            assert np.array_equal(cmc2, z["cmc_%d" % max_rank]) and abs(mAP2 - float(z["mAP_%d" % max_rank])) < 1e-14


def test_re_ranking_matches_reference():
    """oracle.re_ranking (and, on a host without GPU, this build's torchreid.utils.re_ranking) against the reference's
    utils/re_ranking.py on both metrics and three (k1, k2, lambda) settings."""
    z = gold("re_ranking")
    qf, gf = torch.from_numpy(z["qf"]), torch.from_numpy(z["gf"])
    for metric, fn in (("euclidean", O.euclidean_squared), ("cosine", O.cosine)):
        qg, qq, gg = fn(qf, gf).numpy(), fn(qf, qf).numpy(), fn(gf, gf).numpy()
        for tag, kw in (("default", {}), ("k8_3", dict(k1=8, k2=3, lambda_value=0.2)), ("k6_1", dict(k1=6, k2=1, lambda_value=0.5))):
            ref = z[metric + "_" + tag]
            assert np.abs(O.re_ranking(qg, qq, gg, **kw) - ref).max() < 1e-6
            if not torch.cuda.is_available():
                from torchreid.utils.re_ranking import re_ranking
                assert np.abs(re_ranking(qg, qq, gg, **kw) - ref).max() < 1e-6


def test_triplet():
    z = gold("triplet")
    n, d, seed = [int(v) for v in z["meta"]]
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn((n, d), generator=g)
    pids = torch.from_numpy(z["pids"])
    for soft, key in ((True, "soft"), (False, "margin")):
        x = feats.clone().requires_grad_(True)
        loss = O.triplet_hard(x, pids, 0.3, soft)[0]
        loss.backward()
        assert abs(loss.item() - float(z["loss_" + key])) < 1e-6
        close(x.grad, z["grad_" + key], 1e-4)
        from torchreid import losses
        x2 = feats.clone().requires_grad_(True)
        loss2 = losses.TripletLoss(0.3, soft)(x2, pids)  # CPU tensors: host path of the drop-in
        loss2.backward()
        assert abs(loss2.item() - float(z["loss_" + key])) < 1e-6
        close(x2.grad, z["grad_" + key], 1e-4)


def test_pose_adjacency():
    z = gold("pose_adjacency")
    S, width, height, num_split = [int(v) for v in z["meta"]]
    sets = [O.pose_part_sets(z["poses"][t] if z["detected"][t] else None, height, num_split) for t in range(S)]
    adj = O.pose_adjacency(sets, num_split, True)
    assert np.array_equal(adj, z["adj"])
    assert np.array_equal(adj, adj.T) and adj.diagonal().sum() == 0
    assert adj[3 * 7:(3 + 1) * 7].sum() == 0  # the undetected frame is an all-zero block


def harness_model(z):
    """this build's vmgn with the recipe weights + the BNNeck calibration the reference model was given (test_harness.npz)"""
    import harness_split as HS
    from torchreid import models
    m = models.init_model("vmgn", num_classes=HS.N_ID, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2,
                          num_scale=1, pyramid_part=True, use_pose=True, learn_graph=True)
    sd = recipe_state_dict(m.state_dict(), seed=0)
    for name, mean, var in (("global_bottleneck", "g_mean", "g_var"), ("att_bottleneck", "a_mean", "a_var")):
        sd[name + ".running_mean"] = torch.from_numpy(z[mean])
        sd[name + ".running_var"] = torch.from_numpy(z[var])
        sd[name + ".weight"] = torch.ones_like(sd[name + ".weight"])
        sd[name + ".bias"] = torch.zeros_like(sd[name + ".bias"])
    m.load_state_dict(sd)
    return m.eval(), sd


def test_reference_test_function_end_to_end_on_the_cpu():
    """SURVEY 8(a) row 13 on the reference's OWN test() (train_vidreid_xent_htri.py:450-542; fixture test_harness.npz, written by
    tests/golden/make_test_harness.py calling that function on the reference model over tests/harness_split.py's loaders): this
    build's CPU path through the same API calls test() makes -- model(imgs, adj) per batch, torch.cat, compute_distance_matrix,
    evaluate_rank(use_metric_mars=True) -- reproduces the distance matrix to fp32 rounding and Rank-1 / mAP / CMC exactly; the
    oracle's restatement of the tail likewise."""
    import harness_split as HS
    from torchreid import metrics
    z = gold("test_harness")
    q_pids, q_cams, g_pids, g_cams = HS.make_split()
    assert all(np.array_equal(a, z[k]) for a, k in ((q_pids, "q_pids"), (q_cams, "q_cams"), (g_pids, "g_pids"), (g_cams, "g_cams")))
    m, sd = harness_model(z)
    with torch.no_grad():
        qf = torch.cat([m(x, adj) for x, _, _, adj in HS.loader(q_pids, q_cams, HS.Q_SEED)])
        gf = torch.cat([m(x, adj) for x, _, _, adj in HS.loader(g_pids, g_cams, HS.G_SEED)])
        # the oracle on the first query batch
        x, _, _, adj = next(iter(HS.loader(q_pids, q_cams, HS.Q_SEED)))
        close(O.vmgn_eval(x, adj, sd), qf[:x.shape[0]], 1e-4)   # (the calibrated BNNeck divides by near-zero variances: 2e-7 before it)
    for metric in ("cosine", "euclidean"):
        d = metrics.compute_distance_matrix(qf, gf, metric).numpy()
        ref = z[metric + "_evenly_distmat"]
        err = np.abs(d - ref).max() / np.abs(ref).max()
        cmc, mAP = metrics.evaluate_rank(d, q_pids, g_pids, q_cams, g_cams, use_metric_mars=True)
        print("test() harness, %s: distmat rel err %.2e, Rank-1 %.4f mAP %.6f (reference %.4f %.6f)" % (
            metric, err, cmc[0], mAP, z[metric + "_evenly_rank1"], z[metric + "_evenly_mAP"]))
        assert err < 1e-5
        # the ranking is exact wherever the reference's own distances are further apart than the fp32 disagreement
        srt = np.sort(ref, axis=1)
        if np.min(np.diff(srt[:, :51], axis=1)) > 4 * np.abs(d - ref).max():
            assert np.array_equal(cmc, z[metric + "_evenly_cmc"]) and mAP == float(z[metric + "_evenly_mAP"])
        assert abs(cmc[0] - float(z[metric + "_evenly_rank1"])) < 1e-12 and abs(mAP - float(z[metric + "_evenly_mAP"])) < 1e-6
