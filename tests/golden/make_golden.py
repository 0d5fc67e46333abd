"""Generates tests/golden/*.npz by running the REFERENCE implementation (weleen/AGRL.pytorch, mounted read-only at
/root/reference) in the build container. Run:  python tests/golden/make_golden.py

The reference is imported by file path with three harness-side shims (nothing in the reference is modified and no
reference source is copied into this repository):
  1. ``init_pretrained_weights`` is replaced by a no-op (it would download ImageNet weights);
  2. a stub ``sklearn.metrics.base`` module is installed (removed from scikit-learn >= 0.24; the reference's
     rank.py imports it at module level for a dead-code branch);
  3. stub ``torchvision`` / ``torchreid.transforms`` modules are installed so dataset_loader.py (home of
     generate_graph / adj_graph) imports; the stubs are never called.
Only small data (inputs that cannot be re-derived from seeds, and expected outputs) is written. Weights and most
inputs come from tests/recipe.py seeds so the same tensors can be rebuilt anywhere.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("AGRL_REFERENCE_ROOT", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)  # the reference's (empty-__init__) torchreid package; this build's package is NOT on the path

from recipe import recipe_state_dict, recipe_tensor, synthetic_adj, synthetic_clips  # noqa: E402


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def install_shims():
    base = types.ModuleType("sklearn.metrics.base")
    from sklearn.metrics import _base
    base._average_binary_score = _base._average_binary_score
    sys.modules["sklearn.metrics.base"] = base
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tv.transforms = tvt
    tvt.functional = tvf
    tvt.__dict__.update({n: object for n in ("Compose", "ToTensor", "Normalize", "Resize", "RandomCrop")})
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf})
    tr = types.ModuleType("torchreid.transforms")
    tr.ImageData = object
    sys.modules["torchreid.transforms"] = tr


def save(name, **arrays):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:   # python make_golden.py NAME...: rewrite only those fixtures
        return
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def main():
    torch.set_num_threads(8)
    install_shims()
    ref_vmgn = load("ref_vmgn", "torchreid/models/vmgn.py")
    ref_vmgn.init_pretrained_weights = lambda *a, **k: None
    ref_dist = load("ref_distance", "torchreid/metrics/distance.py")
    ref_rank = load("ref_rank", "torchreid/metrics/rank.py")
    ref_trip = load("ref_triplet", "torchreid/losses/hard_mine_triplet_loss.py")
    ref_dl = load("ref_dataset_loader", "torchreid/dataset_loader.py")

    # ---- F1: GraphLayer (vmgn.py:68-172) ----------------------------------------------------------------
    for tag, V, C, use_pose, learn in [("v28", 28, 256, True, True), ("v56", 56, 2048, True, True),
                                        ("v56_pose", 56, 256, True, False), ("v56_learn", 56, 256, False, True),
                                        ("v112", 112, 256, True, True)]:
        B = 1 if C == 2048 else 2
        layer = ref_vmgn.GraphLayer(C, C, learn_graph=learn, use_pose=use_pose)
        sd = recipe_state_dict({"linear.weight": (C, C), "bn.weight": (C,), "bn.bias": (C,), "bn.running_mean": (C,),
                                "bn.running_var": (C,)}, seed=11)
        sd["linear.weight"] = recipe_tensor("graph_layers.0.linear.weight", (C, C), 11)
        layer.load_state_dict(sd, strict=False)
        layer.eval()
        g = torch.Generator().manual_seed(V * 7 + C)
        f = torch.rand((B, 1, C), generator=g) + 0.02 * torch.randn((B, V, C), generator=g)
        adj = synthetic_adj(B, V // 7, seed=V)
        with torch.no_grad():
            out = layer(f, adj)
            sim = layer.get_sim_matrix(f)
        save("graph_layer_" + tag, f=f if C <= 256 else np.zeros(0), adj=adj, out=out, sim=sim,
             meta=np.array([B, V, C, int(use_pose), int(learn), V * 7 + C, 11]))

    # ---- F2: pooling + attention tail (vmgn.py:270-278, :298-321) on small feature maps ----------------------
    model = ref_vmgn.vmgn(num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                          pyramid_part=True, use_pose=True, learn_graph=True, consistent_loss=False)
    sd_full = recipe_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd_full)
    model.eval()
    B, S, c, h, w = 2, 4, 2048, 16, 8
    g = torch.Generator().manual_seed(21)
    x41 = torch.rand((B * S, c, h, w), generator=g)
    x42 = torch.rand((B * S, c, h, w), generator=g)
    adj = synthetic_adj(B, S, seed=21)
    with torch.no_grad():
        g_f = model.global_avg_pool(x41.view(B, S, c, h, w).transpose(1, 2).contiguous()).view(B, -1)
        v_f = [model.parts_avgpool[i](x42).view(B, S, c, n) for i, n in enumerate(model.total_split_list)]
        nodes = torch.cat(v_f, dim=3).transpose(2, 3).contiguous().view(B, S * model.total_split, c)
        f = nodes
        for i in range(model.num_gb):
            f = model.graph_layers[i](f, adj)
        att_f = model._attention_op(f.view(B, S, model.total_split, c)).mean(dim=1).view(B, -1)
        out = torch.cat([model.global_bottleneck(g_f), model.att_bottleneck(att_f)], dim=1)
    save("tail", g_f=g_f, nodes=nodes[:, :, :64], nodes_out=f[:, :, :64], att_f=att_f, out=out,
         meta=np.array([B, S, c, h, w, 21]))

    # ---- F3: full vmgn eval forward (vmgn.py:292-321), config-1 shape --------------------------------------
    for tag, B, S, seed in [("b2s4", 2, 4, 4), ("b1s8", 1, 8, 8)]:
        x, adj = synthetic_clips(B, S, seed=seed), synthetic_adj(B, S, seed=seed)
        with torch.no_grad():
            y = model(x, adj)
            x4_1, x4_2 = model.featuremaps(x.view(B * S, 3, 256, 128))
        save("vmgn_eval_" + tag, out=y, x4_1_mean=x4_1.mean(dim=(2, 3)), x4_2_mean=x4_2.mean(dim=(2, 3)),
             meta=np.array([B, S, seed, 0]))

    # ---- F10: sibling model gsta (gsta.py:173-336): eval forward + state-dict keys + train outputs --------------
    ref_gsta = load("ref_gsta", "torchreid/models/gsta.py")
    ref_gsta.init_pretrained_weights = lambda *a, **k: None
    gm = ref_gsta.gsta(num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                       pyramid_part=True, use_pose=True, learn_graph=True, consistent_loss=True)
    gsd = recipe_state_dict(gm.state_dict(), seed=0)
    gm.load_state_dict(gsd)
    gm.eval()
    x, adj = synthetic_clips(2, 4, seed=4), synthetic_adj(2, 4, seed=4)
    with torch.no_grad():
        gy = gm(x, adj)
    gm.train()
    xt, adjt = synthetic_clips(2, 8, seed=8), synthetic_adj(2, 8, seed=8)
    np.random.seed(123)
    outs, feats = gm(xt, adjt)
    save("gsta_b2s4", out=gy, keys=np.array(sorted(gsd.keys())), shapes=np.array([str(tuple(gsd[k].shape)) for k in sorted(gsd.keys())]),
         train_logits=torch.stack([o.detach() for o in outs]), train_feats=torch.stack([f.detach() for f in feats]),
         meta=np.array([2, 4, 4, 0]))

    # ---- F11: sibling model ganet (ganet.py:98-477): PAM part nodes, diagonal-masked graph layers, concatenated outputs --
    ref_ganet = load("ref_ganet", "torchreid/models/ganet.py")
    am = ref_ganet.ganet(num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1, knn=4,
                         pyramid_part=True, use_pose=True, learn_graph=True, pretrained=False, consistent_loss=True)
    asd = recipe_state_dict(am.state_dict(), seed=0)
    am.load_state_dict(asd)
    am.eval()
    x, adj = synthetic_clips(2, 4, seed=6), synthetic_adj(2, 4, seed=6)
    with torch.no_grad():
        ay0 = am(x, adj)                       # graph layers with the constructor's gamma = 0 (ganet.py:175)
        for layer in am.graph_layers:
            layer.gamma = 0.1
        ay1 = am(x, adj)                       # the masked-graph arithmetic made visible
        fm = am.featuremaps(x.view(8, 3, 256, 128))
        pam_out, _ = am.pam_layer(fm[:, :, 4:8])
        for layer in am.graph_layers:
            layer.gamma = 0
    am.train()
    xt, adjt = synthetic_clips(2, 6, seed=9), synthetic_adj(2, 6, seed=9)
    np.random.seed(321)
    aouts, afeats = am(xt, adjt)
    save("ganet_b2s4", out_gamma0=ay0, out_gamma01=ay1, pam_slice_mean=pam_out.mean(dim=(2, 3)),
         keys=np.array(sorted(asd.keys())), shapes=np.array([str(tuple(asd[k].shape)) for k in sorted(asd.keys())]),
         train_logits=torch.stack([o.detach() for o in aouts]), train_feats=torch.stack([f_.detach() for f_ in afeats]),
         meta=np.array([2, 4, 6, 0]))

    # ---- F8: train-mode outputs with the consistent loss (vmgn.py:323-357) ---------------------------------
    model_t = ref_vmgn.vmgn(num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                            pyramid_part=True, use_pose=True, learn_graph=True, consistent_loss=True)
    model_t.load_state_dict(sd_full)
    model_t.train()
    x, adj = synthetic_clips(2, 8, seed=31), synthetic_adj(2, 8, seed=31)
    torch.manual_seed(1234)
    outs, feats = model_t(x, adj)
    save("vmgn_train_b2s8", **{"logit%d" % i: o for i, o in enumerate(outs)}, **{"feat%d" % i: f_ for i, f_ in enumerate(feats)},
         meta=np.array([2, 8, 31, 1234]))

    # ---- F4: distance matrices (distance.py:59-89) ------------------------------------------------------
    g = torch.Generator().manual_seed(41)
    q, gal = torch.randn((37, 4096), generator=g), torch.randn((101, 4096), generator=g)
    save("distmat", euclidean=ref_dist.compute_distance_matrix(q, gal, "euclidean"),
         cosine=ref_dist.compute_distance_matrix(q, gal, "cosine"), meta=np.array([37, 101, 4096, 41]))

    # ---- F5: MARS ranking (rank.py:160-212) --------------------------------------------------------------
    rng = np.random.RandomState(51)
    m, n = 30, 300
    dist = rng.rand(m, n).astype(np.float32)
    q_pids, g_pids = rng.randint(0, 8, m), rng.randint(0, 8, n)
    g_pids[rng.rand(n) < 0.05] = -1
    q_cam, g_cam = rng.randint(0, 6, m), rng.randint(0, 6, n)
    cmc, mAP = ref_rank.evaluate_rank(dist, q_pids, g_pids, q_cam, g_cam, max_rank=50, use_metric_mars=True)
    aps = []
    for k in range(m):  # per-query AP through the reference's Compute_AP
        good = np.where((q_pids[k] == g_pids) & (q_cam[k] != g_cam))[0]
        junk = np.where((g_pids == -1) | ((q_pids[k] == g_pids) & (q_cam[k] == g_cam)))[0]
        aps.append(ref_rank.Compute_AP(good, junk, np.argsort(dist[k])[:50])[0])
    save("rank_mars", dist=dist, q_pids=q_pids, g_pids=g_pids, q_camids=q_cam, g_camids=g_cam, cmc=cmc, mAP=np.float64(mAP),
         ap=np.array(aps, dtype=np.float64))

    # ---- F9: market1501 protocol, python evaluator (rank.py:95-150; reached through evaluate_rank, :232-236) ----
    rng = np.random.RandomState(91)
    m, n = 40, 500
    dist = rng.rand(m, n).astype(np.float32)
    q_pids, g_pids = rng.randint(0, 14, m), rng.randint(0, 12, n)   # identities 12, 13 never appear in the gallery
    q_cam, g_cam = rng.randint(0, 6, m), rng.randint(0, 6, n)
    cmc, mAP = ref_rank.evaluate_rank(dist, q_pids, g_pids, q_cam, g_cam, max_rank=50, use_metric_market1501=True,
                                      use_cython=False)
    save("rank_market1501", dist=dist, q_pids=q_pids, g_pids=g_pids, q_camids=q_cam, g_camids=g_cam, cmc=cmc,
         mAP=np.float64(mAP))

    # ---- F12: cuhk03 protocol (rank.py:22-92), numpy's global RNG seeded right before the call -------------------
    rng = np.random.RandomState(121)
    m, n = 36, 420
    dist = rng.rand(m, n).astype(np.float32)
    q_pids, g_pids = rng.randint(0, 64, m), rng.randint(0, 60, n)   # identities 60..63 never appear in the gallery
    q_cam, g_cam = rng.randint(0, 2, m), rng.randint(0, 2, n)
    out = {}
    for max_rank in (50, 20):
        np.random.seed(1203)
        cmc, mAP = ref_rank.evaluate_rank(dist, q_pids, g_pids, q_cam, g_cam, max_rank=max_rank, use_metric_cuhk03=True,
                                          use_cython=False)
        out["cmc_%d" % max_rank], out["mAP_%d" % max_rank] = cmc, np.float64(mAP)
        out["next_draw_%d" % max_rank] = np.random.randint(0, 1 << 30)  # the global stream must end in the same place
    save("rank_cuhk03", dist=dist, q_pids=q_pids, g_pids=g_pids, q_camids=q_cam, g_camids=g_cam, seed=np.array(1203), **out)

    # ---- F11: k-reciprocal re-ranking (utils/re_ranking.py:30-95) on the three distance matrices ----------------
    ref_rr = load("ref_re_ranking", "torchreid/utils/re_ranking.py")
    g = torch.Generator().manual_seed(111)
    cent = torch.randn((9, 48), generator=g)
    qf = cent[torch.randint(0, 9, (14,), generator=g)] + 0.6 * torch.randn((14, 48), generator=g)
    gf = cent[torch.randint(0, 9, (60,), generator=g)] + 0.6 * torch.randn((60, 48), generator=g)
    mats = {}
    for metric in ("euclidean", "cosine"):
        qg = ref_dist.compute_distance_matrix(qf, gf, metric).numpy()
        qq = ref_dist.compute_distance_matrix(qf, qf, metric).numpy()
        gg = ref_dist.compute_distance_matrix(gf, gf, metric).numpy()
        mats[metric + "_default"] = ref_rr.re_ranking(qg, qq, gg)
        mats[metric + "_k8_3"] = ref_rr.re_ranking(qg, qq, gg, k1=8, k2=3, lambda_value=0.2)
        mats[metric + "_k6_1"] = ref_rr.re_ranking(qg, qq, gg, k1=6, k2=1, lambda_value=0.5)
    save("re_ranking", qf=qf, gf=gf, **mats)

    # ---- F6: batch-hard triplet loss (hard_mine_triplet_loss.py:24-50) ----------------------------------
    g = torch.Generator().manual_seed(61)
    feats = torch.randn((16, 2048), generator=g)
    pids = torch.arange(4).repeat_interleave(4)
    res = {}
    for soft in (True, False):
        fx = feats.clone().requires_grad_(True)
        loss = ref_trip.TripletLoss(margin=0.3, soft=soft)(fx, pids)
        loss.backward()
        res["loss_soft" if soft else "loss_margin"] = loss.detach()
        res["grad_soft" if soft else "grad_margin"] = fx.grad
    save("triplet", pids=pids, meta=np.array([16, 2048, 61]), **res)

    # ---- F13: label-smoothed cross entropy + DeepSupervision (losses/cross_entropy_loss.py:26-37, losses/__init__.py:9-20):
    # value and logit gradients of the five-output list the consistent loss produces --------------------------------------
    ref_xent = load("ref_xent", "torchreid/losses/cross_entropy_loss.py")
    g = torch.Generator().manual_seed(131)
    n_out, n, K = 5, 16, 702
    logits = [(3.0 * torch.randn((n, K), generator=g)).requires_grad_(True) for _ in range(n_out)]
    pids = torch.randint(0, K, (n,), generator=g)
    res = {}
    for eps in (0.1, 0.0, 0.3):
        crit = ref_xent.CrossEntropyLabelSmooth(num_classes=K, epsilon=eps, use_gpu=False)
        loss = 0.
        for x in logits:           # DeepSupervision, losses/__init__.py:9-20 (restated here: the package __init__ is not importable)
            loss += crit(x, pids)
        loss /= len(logits)
        for x in logits:
            x.grad = None
        loss.backward()
        tag = "eps%02d" % int(round(eps * 100))
        res["loss_" + tag] = loss.detach()
        res["single_" + tag] = crit(logits[0], pids).detach()
        res["grad0_" + tag] = logits[0].grad.clone()
    save("xent", logits0=logits[0].detach(), pids=pids, meta=np.array([n_out, n, K, 131]), **res)

    # ---- F14: the reference's own native evaluator (rank_cylib/rank_cy.pyx:154-241, built by oracle/build_ref.py) on the
    # seeded inputs tests/test_gpu_kernels.py::test_rank_market1501_device regenerates: its outputs as data, so the built
    # extension itself never has to travel to the GPU box --------------------------------------------------------------
    from oracle import build_ref
    build_ref.build(verbose=False)
    cy = build_ref.load()
    res = {}
    for m, n in ((40, 500), (64, 12180)):
        rng = np.random.RandomState(m + n)
        d = rng.rand(m, n).astype(np.float32)
        d[:, 5] = d[:, 3]
        npid = max(4, n // 40)
        q_pids, g_pids = rng.randint(0, npid + 2, m), rng.randint(0, npid, n)
        q_cam, g_cam = rng.randint(0, 6, m), rng.randint(0, 6, n)
        i64 = [np.ascontiguousarray(a, dtype=np.int64) for a in (q_pids, g_pids, q_cam, g_cam)]
        cmc_cy, mAP_cy = cy.eval_market1501_cy(d, i64[0], i64[1], i64[2], i64[3], 50)
        cmc_py, mAP_py = ref_rank.evaluate_rank(d, q_pids, g_pids, q_cam, g_cam, max_rank=50, use_metric_market1501=True, use_cython=False)
        res["cmc_cy_%dx%d" % (m, n)], res["mAP_cy_%dx%d" % (m, n)] = np.asarray(cmc_cy), np.float64(mAP_cy)
        res["cmc_py_%dx%d" % (m, n)], res["mAP_py_%dx%d" % (m, n)] = np.asarray(cmc_py), np.float64(mAP_py)
        res["dist_checksum_%dx%d" % (m, n)] = np.float64(d.astype(np.float64).sum())   # guards the seeded regeneration
    save("rank_market1501_cy", **res)

    # ---- F15: loss.backward() of the reference's train step (train_vidreid_xent_htri.py:397-411): B = 4 (2 ids x 2), S = 8,
    # xent + htri with the consistent loss; loss value and the gradients of a few named parameters (slices of the large ones) --
    ref_trip2 = ref_trip
    model_b = ref_vmgn.vmgn(num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                            pyramid_part=True, use_pose=True, learn_graph=True, consistent_loss=True)
    sd_b = recipe_state_dict(model_b.state_dict(), seed=2)
    model_b.load_state_dict(sd_b)
    model_b.train()
    pids = torch.tensor([0, 0, 1, 1])
    x, adj = synthetic_clips(4, 8, seed=151, identities=pids.tolist()), synthetic_adj(4, 8, seed=151)
    torch.manual_seed(1234)
    outs, feats = model_b(x, adj)
    crit_x = ref_xent.CrossEntropyLabelSmooth(num_classes=5, use_gpu=False)
    crit_t = ref_trip2.TripletLoss(margin=0.3, soft=True)
    lx = sum(crit_x(o, pids) for o in outs) / len(outs)
    lt = sum(crit_t(f_, pids) for f_ in feats) / len(feats)
    loss = lx + lt
    loss.backward()
    named = dict(model_b.named_parameters())
    res = {"loss": loss.detach(), "xent": lx.detach(), "htri": lt.detach(), "meta": np.array([4, 8, 151, 1234, 2])}
    for key, rows in (("conv1.weight", None), ("bn1.weight", None), ("layer1.0.conv1.weight", None), ("layer1.0.bn3.bias", None),
                      ("layer2.0.downsample.0.weight", 8), ("layer3.5.conv3.weight", 8), ("layer3.5.bn2.weight", None),
                      ("layer4_1.2.conv3.weight", 4), ("layer4_2.0.conv2.weight", 1), ("layer4_2.2.bn3.weight", None),
                      ("graph_layers.0.linear.weight", 4), ("graph_layers.1.linear.weight", 4), ("graph_layers.1.bn.weight", None),
                      ("global_bottleneck.weight", None), ("att_bottleneck.weight", None), ("global_classifier.weight", None),
                      ("att_classifier.weight", None)):
        gfull = named[key].grad
        res["g:" + key] = gfull if rows is None else gfull[:rows]
        res["n:" + key] = gfull.double().norm()          # norm of the WHOLE gradient tensor
    save("vmgn_backward_b4s8", **res)

    # ---- F7: pose adjacency (dataset_loader.py:218-388) -------------------------------------------------
    rng = np.random.RandomState(71)
    S, width, height = 8, 128, 256
    paths, poses, pose_arr = [], {}, np.zeros((S, 18, 3), dtype=np.float32)
    detected = np.ones(S, dtype=bool)
    for t in range(S):
        path = "data/mars/bbox_test/0001/0001C1T0001F%03d.jpg" % (t + 1)
        paths.append(path)
        kp = np.stack([rng.rand(18) * width, rng.rand(18) * height, rng.rand(18)], axis=1).astype(np.float32)
        pose_arr[t] = kp
        if t == 3:
            detected[t] = False  # no pose entry for this frame -> all-zero block
        else:
            poses[path.split("/")[-1]] = kp
    ims = [torch.zeros(3, 256, 128) for _ in range(S)]
    adj = ref_dl.generate_graph(ims, im_paths=paths, im_sizes=[(width, height)] * S, poses=poses, num_split=4,
                                num_parts=3, num_scale=1, pyramid_part=True)
    save("pose_adjacency", poses=pose_arr, detected=detected, adj=adj, meta=np.array([S, width, height, 4]))


if __name__ == "__main__":
    main()
