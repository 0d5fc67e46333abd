"""The test()-equivalent pipeline on the GPU vs the CPU oracle on identical synthetic weights and synthetic
MARS-shaped tracklets: embeddings -> distance matrix -> MARS ranking -> Rank-1 / mAP ("matched Rank-1/mAP")."""
import numpy as np
import pytest
import torch

from lp16 import LP16, LP_DTYPE

from oracle import vmgn_oracle as O
from recipe import calibrate_bnneck, recipe_state_dict, synthetic_adj, synthetic_clips

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

N_ID, S = 12, 4
G_PER_ID, Q_PER_ID = 5, 1


def make_split(rng):
    """query: 1 tracklet per identity (camera 0); gallery: 5 per identity on cameras 0..4 + a few junk (pid -1)."""
    q_pids = np.arange(N_ID)
    q_cams = np.zeros(N_ID, dtype=np.int64)
    g_pids = np.repeat(np.arange(N_ID), G_PER_ID)
    g_cams = np.tile(np.arange(G_PER_ID), N_ID)
    g_pids = np.concatenate([g_pids, -np.ones(4, dtype=np.int64)])
    g_cams = np.concatenate([g_cams, rng.randint(0, 5, 4)])
    return q_pids, q_cams, g_pids, g_cams


def batches(pids, cams, seed, bs=8):
    idents = [p if p >= 0 else 1000 + i for i, p in enumerate(pids)]  # junk tracklets get their own pattern
    for i in range(0, len(pids), bs):
        sl = slice(i, i + bs)
        b = len(pids[sl])
        yield (synthetic_clips(b, S, seed=seed + i, identities=idents[sl]), pids[sl], cams[sl], synthetic_adj(b, S, seed=seed + i))


@pytest.fixture(scope="module")
def world():
    from torchreid import models
    rng = np.random.RandomState(5)
    q_pids, q_cams, g_pids, g_cams = make_split(rng)
    m = models.init_model("vmgn", num_classes=N_ID, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2,
                          num_scale=1, pyramid_part=True, use_pose=True, learn_graph=True)
    sd = recipe_state_dict(m.state_dict(), seed=0)

    def oracle_parts(pids, cams, seed):
        g, a = [], []
        for x, _, _, adj in batches(pids, cams, seed):
            B = x.shape[0]
            x4_1, x4_2 = O.featuremaps(x.view(B * S, 3, 256, 128), sd)
            _, parts = O.tail(x4_1, x4_2, adj, sd, B, S, [4, 2, 1], 2, return_parts=True)
            g.append(parts["g_f"])
            a.append(parts["att_f"])
        return torch.cat(g), torch.cat(a)

    with torch.no_grad():
        qg, qa = oracle_parts(q_pids, q_cams, 100)
        gg, ga = oracle_parts(g_pids, g_cams, 500)
        sd = calibrate_bnneck(sd, torch.cat([qg, gg]), torch.cat([qa, ga]))
        qf = torch.cat([O._bn(qg, sd, "global_bottleneck"), O._bn(qa, sd, "att_bottleneck")], 1)
        gf = torch.cat([O._bn(gg, sd, "global_bottleneck"), O._bn(ga, sd, "att_bottleneck")], 1)
    m.load_state_dict(sd)
    m.eval()
    ref = {}
    for metric, fn in (("cosine", O.cosine), ("euclidean", O.euclidean_squared)):
        d = fn(qf, gf).numpy()
        cmc, mAP, ap, _, order = O.evaluate_mars(d, q_pids, g_pids, q_cams, g_cams, 50, return_all=True)
        srt = np.sort(d, axis=1)
        ref[metric] = dict(d=d, cmc=cmc, mAP=mAP, order=order, min_gap=float(np.min(np.diff(srt[:, :51], axis=1))),
                           scale=float(np.abs(d).max()))
    return dict(model=m.to(DEV), qf=qf, gf=gf, ref=ref, split=(q_pids, q_cams, g_pids, g_cams))


@pytest.mark.parametrize("metric", ["cosine", "euclidean"])
def test_fp32_pipeline_matches_oracle_rank1_map_and_indices(world, metric):
    from torchreid import evaluation
    m = world["model"]
    m.hip_precision = "fp32"
    q_pids, q_cams, g_pids, g_cams = world["split"]
    qf, _, _ = evaluation.extract_features(m, batches(q_pids, q_cams, 100))
    gf, _, _ = evaluation.extract_features(m, batches(g_pids, g_cams, 500))
    ref = world["ref"][metric]
    e_q = ((qf.cpu() - world["qf"]).abs().max() / world["qf"].abs().max()).item()
    cmc, mAP, idx, val = evaluation.match_and_rank(qf, q_pids, q_cams, gf, g_pids, g_cams, metric, 50, "fp32", return_topk=True)
    # ranking indices are only well defined where neighbouring distances differ by more than the fp32 error
    err = np.abs(val - np.take_along_axis(ref["d"], ref["order"], 1)).max()
    print("%s: embedding rel err %.2e, top-50 distance abs err %.2e, oracle min gap %.2e, Rank-1 %.3f mAP %.4f (oracle %.3f %.4f)" % (
        metric, e_q, err, ref["min_gap"], cmc[0], mAP, ref["cmc"][0], ref["mAP"]))
    assert e_q < 1e-3
    assert err < 1e-3 * ref["scale"]
    near_ties = 0
    for r in range(idx.shape[0]):
        if not np.array_equal(idx[r], ref["order"][r]):
            # any disagreement must be a swap between distances closer than the combined rounding error
            bad = np.where(idx[r] != ref["order"][r])[0]
            assert np.all(np.abs(ref["d"][r, idx[r][bad]] - ref["d"][r, ref["order"][r][bad]]) < 2 * err + 1e-6)
            near_ties += 1
    print("rows with near-tie swaps: %d / %d" % (near_ties, idx.shape[0]))
    if ref["min_gap"] > 4 * err:
        assert near_ties == 0
        assert mAP == ref["mAP"] and np.array_equal(cmc, ref["cmc"])
    assert abs(cmc[0] - ref["cmc"][0]) < 1e-9 and abs(mAP - ref["mAP"]) < 1e-6


def test_16_bit_pipeline_keeps_rank1_and_map(world):
    from torchreid import evaluation
    m = world["model"]
    m.hip_precision = LP16
    q_pids, q_cams, g_pids, g_cams = world["split"]
    ref = world["ref"]["cosine"]

    def loader(p, c, s):
        return batches(p, c, s)

    r1, mAP = evaluation.evaluate(m, loader(q_pids, q_cams, 100), loader(g_pids, g_cams, 500), "cosine")
    print("%s: Rank-1 %.3f mAP %.4f (oracle %.3f %.4f)" % (LP16, r1, mAP, ref["cmc"][0], ref["mAP"]))
    m.hip_precision = "fp32"
    assert abs(r1 - ref["cmc"][0]) <= 1.0 / N_ID + 1e-9
    assert abs(mAP - ref["mAP"]) < (0.005 if LP16 == "fp16" else 0.03)


def test_dense_clip_pooling_path(world):
    from torchreid import evaluation
    m = world["model"]
    m.hip_precision = "fp32"
    x = synthetic_clips(6, S, seed=3)
    adj = synthetic_adj(6, S, seed=3)
    flat, _, _ = evaluation.extract_features(m, [(x, np.arange(6), np.zeros(6), adj)])
    dense, pids, _ = evaluation.extract_features(m, [(x.view(2, 3, S, 3, 256, 128), np.arange(2), np.zeros(2), adj.view(2, 3, 28, 28))], pool="avg")
    assert dense.shape == (2, 4096) and len(pids) == 2
    assert torch.allclose(dense, flat.view(2, 3, -1).mean(1), atol=1e-6)
    dmax, _, _ = evaluation.extract_features(m, [(x.view(2, 3, S, 3, 256, 128), np.arange(2), np.zeros(2), adj.view(2, 3, 28, 28))], pool="max")
    assert torch.allclose(dmax, flat.view(2, 3, -1).max(1)[0], atol=1e-6)


def test_re_rank_branch_matches_oracle_pipeline(world):
    """test() with --re-rank: distance matrices -> k-reciprocal re-ranking -> MARS ranking, device vs oracle."""
    from torchreid import evaluation
    q_pids, q_cams, g_pids, g_cams = world["split"]
    qf, gf = world["qf"], world["gf"]
    d = O.re_ranking(O.cosine(qf, gf).numpy(), O.cosine(qf, qf).numpy(), O.cosine(gf, gf).numpy())
    cmc_ref, map_ref = O.evaluate_mars(d, q_pids, g_pids, q_cams, g_cams, 50)
    cmc, mAP = evaluation.match_and_rank(qf.to(DEV), q_pids, q_cams, gf.to(DEV), g_pids, g_cams, "cosine", 50, "fp32", re_rank=True)
    assert abs(mAP - map_ref) < 1e-3 and np.abs(cmc - cmc_ref).max() < 0.1  # near-ties of the Jaccard term may swap
    print("re-rank branch: mAP %.6f (oracle %.6f) rank-1 %.4f (oracle %.4f)" % (mAP, map_ref, cmc[0], cmc_ref[0]))


@pytest.mark.parametrize("factor, message", [(3e4, "fp16's range"), (1e6, "overflow fp16")])
def test_out_of_range_activations_fail_loudly_in_the_16_bit_mode(factor, message):
    """fp16 build: a checkpoint whose activations leave fp16's range (the stem's folded BatchNorm scaled by 3e4) gives inf / nan
    embeddings and extract_features refuses them instead of ranking garbage; one whose FOLDED WEIGHTS already overflow (scaled
    by 1e6) is refused at pack time with the layer named. The bf16 build has fp32's range and passes both."""
    from torchreid import evaluation, models
    m = models.init_model("vmgn", num_classes=4, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                          pyramid_part=True, use_pose=True, learn_graph=True)
    sd = recipe_state_dict(m.state_dict(), seed=0)
    sd["bn1.weight"] = sd["bn1.weight"] * factor
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m.hip_precision = LP16
    x, adj = synthetic_clips(2, 4, seed=3), synthetic_adj(2, 4, seed=3)
    batch = [(x, np.zeros(2, dtype=np.int64), np.zeros(2, dtype=np.int64), adj)]
    if LP16 == "fp16":
        with pytest.raises(FloatingPointError, match=message):
            evaluation.extract_features(m, batch, prefetch=False)
    else:
        f, _, _ = evaluation.extract_features(m, batch, prefetch=False)
        assert torch.isfinite(f).all()


def test_out_of_range_activations_fail_loudly_in_the_conforming_mode():
    """'fp16x3' carries activations as fp16 high / low halves (in the planes and inside the in-loop split): beyond 65504 the high half is
    inf, the embedding non-finite, and extract_features refuses it -- in both builds of the library (the split is fp16 in either)."""
    from torchreid import evaluation, models
    m = models.init_model("vmgn", num_classes=4, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                          pyramid_part=True, use_pose=True, learn_graph=True)
    sd = recipe_state_dict(m.state_dict(), seed=0)
    sd["bn1.weight"] = sd["bn1.weight"] * 3e4
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m.hip_precision = "fp16x3"
    x, adj = synthetic_clips(2, 4, seed=3), synthetic_adj(2, 4, seed=3)
    batch = [(x, np.zeros(2, dtype=np.int64), np.zeros(2, dtype=np.int64), adj)]
    with pytest.raises(FloatingPointError, match="non-finite"):
        evaluation.extract_features(m, batch, prefetch=False)
    m.hip_precision = "fp32"      # the exact mode has fp32's range: the same checkpoint passes
    f, _, _ = evaluation.extract_features(m, batch, prefetch=False)
    assert torch.isfinite(f).all()


@pytest.mark.parametrize("sample", ["evenly", "dense"])
@pytest.mark.parametrize("metric", ["cosine", "euclidean"])
def test_device_pipeline_equals_the_reference_test_function(metric, sample):
    """SURVEY 8(a) row 13, end to end on the reference's OWN test() (train_vidreid_xent_htri.py:450-542): tests/golden/
    test_harness.npz holds what that function returned -- (Rank-1, mAP), CMC, the distance matrix -- for the reference model on the
    loaders of tests/harness_split.py (evenly-sampled batches, and the dense samplers' one-tracklet-of-n-clips batches with mean
    pooling), both metrics. The device-resident harness (evaluation.evaluate: HIP forward in exact fp32, clip pooling, agrl_distmat_topk,
    agrl_rank_mars) on the same inputs must give the same Rank-1 / mAP / CMC, and compute_distance_matrix on its features the same
    matrix (1e-5; index-exact ranking up to near-ties of the reference's own distances)."""
    import os
    import harness_split as HS
    from torchreid import evaluation, metrics, models
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "test_harness.npz"))
    q_pids, q_cams, g_pids, g_cams = HS.make_split()
    m = models.init_model("vmgn", num_classes=HS.N_ID, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2,
                          num_scale=1, pyramid_part=True, use_pose=True, learn_graph=True)
    sd = recipe_state_dict(m.state_dict(), seed=0)
    for name, mean, var in (("global_bottleneck", "g_mean", "g_var"), ("att_bottleneck", "a_mean", "a_var")):
        sd[name + ".running_mean"] = torch.from_numpy(z[mean])
        sd[name + ".running_var"] = torch.from_numpy(z[var])
        sd[name + ".weight"] = torch.ones_like(sd[name + ".weight"])
        sd[name + ".bias"] = torch.zeros_like(sd[name + ".bias"])
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    m.hip_precision = "fp32"
    mk = HS.loader if sample == "evenly" else HS.dense_loader
    tag = "%s_%s" % (metric, sample)
    r1, mAP = evaluation.evaluate(m, mk(q_pids, q_cams, HS.Q_SEED), mk(g_pids, g_cams, HS.G_SEED), metric, pool="avg")
    qf, _, _ = evaluation.extract_features(m, mk(q_pids, q_cams, HS.Q_SEED), pool="avg")
    gf, _, _ = evaluation.extract_features(m, mk(g_pids, g_cams, HS.G_SEED), pool="avg")
    d = metrics.compute_distance_matrix(qf, gf, metric).cpu().numpy()
    ref = z[tag + "_distmat"]
    err = np.abs(d - ref).max() / np.abs(ref).max()
    cmc, mAP2 = metrics.evaluate_rank(d, q_pids, g_pids, q_cams, g_cams, use_metric_mars=True)
    print("reference test() vs device harness, %s: distmat rel err %.2e, Rank-1 %.4f mAP %.6f (reference %.4f %.6f)" % (
        tag, err, r1, mAP, z[tag + "_rank1"], z[tag + "_mAP"]))
    assert err < 1e-5
    assert abs(r1 - float(z[tag + "_rank1"])) < 1e-12 and abs(mAP - float(z[tag + "_mAP"])) < 1e-6
    assert abs(mAP2 - mAP) < 1e-9 and abs(cmc[0] - r1) < 1e-12      # the API path and the fused device path agree
    srt = np.sort(ref, axis=1)
    if np.min(np.diff(srt[:, :51], axis=1)) > 4 * np.abs(d - ref).max():   # no near-tie in any top-50 list: index-exact
        assert np.array_equal(cmc, z[tag + "_cmc"]) and mAP2 == float(z[tag + "_mAP"])


def test_query_operand_from_the_tail_kernel_equals_the_separate_launches(world, monkeypatch):
    """hip_ops.query_operands: for the tensor the forward just returned the cosine operand / euclidean norms come from agrl_attn_tail
    (no extra launch) and are bit-identical to agrl_row_l2_normalize / agrl_row_sqnorm on it; any other tensor (a clone, a gathered
    batch) takes the separate launches; the distance matrix is the same either way."""
    from torchreid import hip_ops as ops
    from torchreid import _hip
    monkeypatch.setenv("AGRL_HIP_FUSE_ATTN_TAIL", "1")   # (the model takes the one-launch tail from 224 tracklets per GPU on its own)
    _hip.reload_options()                                # (the host side caches its switches like the library does)
    m = world["model"]
    x, adj = synthetic_clips(4, S, seed=21).to(DEV), synthetic_adj(4, S, seed=21).to(DEV)
    gal = torch.randn((300, 4096), device=DEV)
    for prec, dt in ((LP16, LP_DTYPE), ("fp32", torch.float32)):
        m.hip_precision = prec
        emb = m(x, adj)
        assert m._hip_query is not None and m._hip_query.lookup(emb, dt) is not None
        assert m._hip_query.lookup(emb.clone(), dt) is None
        for metric in ("cosine", "euclidean"):
            q_op, qn = ops.query_operands(m, emb, metric, dt)
            q_ref, qn_ref = ops.query_operands(None, emb, metric, dt)
            assert torch.equal(q_op, q_ref) and (qn is None) == (qn_ref is None) and (qn is None or torch.equal(qn, qn_ref))
            if metric == "cosine":
                g_op = ops.row_l2_normalize(gal, True, dt)
                assert torch.equal(ops.distmat(q_op, g_op, "cosine"), ops.distmat(q_ref, g_op, "cosine"))
        monkeypatch.setenv("AGRL_HIP_FUSE_ATTN_TAIL", "0")
        _hip.reload_options()
        emb3 = m(x, adj)                                   # the three-launch tail: the same embedding bit for bit, no operand cache
        monkeypatch.setenv("AGRL_HIP_FUSE_ATTN_TAIL", "1")
        _hip.reload_options()
        assert torch.equal(emb3, emb) and m._hip_query is None
    m.hip_precision = "fp32"
