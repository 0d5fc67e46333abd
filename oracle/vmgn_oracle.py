"""ORACLE -- TEST INFRASTRUCTURE ONLY. Never imported by the product (agrl.pytorch_amd/), only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.

A from-scratch CPU restatement of AGRL's per-tracklet forward-and-match path, written as plain functions over
a state-dict of tensors (no nn.Module, no reference code). Arithmetic is delegated to torch CPU functional ops
in fp32 (or fp64 with ``dtype=torch.float64`` to bound the oracle's own rounding error) because the reference's
arithmetic IS torch's CPU kernels; every function cites the reference lines it restates
(paths relative to weleen/AGRL.pytorch).

Parity pin: tests/golden/*.npz were produced by importing the reference itself in the build container
(tests/golden/make_golden.py) and tests/test_oracle_golden.py checks every function here against them.
The reference has no tests or golden vectors of its own for this path, so those captured
fixtures are the only pin.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

RESNET50_STAGES = (("layer1", 3, 1), ("layer2", 4, 2), ("layer3", 6, 2))
BN_EPS = 1e-5


def calc_splits(num_split):
    """torchreid/utils/reidtools.py:13-15."""
    assert num_split > 0 and num_split & (num_split - 1) == 0
    return [i for i in range(num_split, 0, -1) if num_split % i == 0]


def _bn(x, sd, prefix):
    """Eval-mode BatchNorm (running statistics), nn.BatchNorm{1,2}d as used at vmgn.py:49-63, :169, :301, :318."""
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], training=False, eps=BN_EPS)


def bottleneck(x, sd, prefix, stride):
    """Bottleneck.forward, vmgn.py:45-65 (stride on the 3x3; downsample = 1x1 conv + BN when present)."""
    out = F.relu(_bn(F.conv2d(x, sd[prefix + ".conv1.weight"]), sd, prefix + ".bn1"))
    out = F.relu(_bn(F.conv2d(out, sd[prefix + ".conv2.weight"], stride=stride, padding=1), sd, prefix + ".bn2"))
    out = _bn(F.conv2d(out, sd[prefix + ".conv3.weight"]), sd, prefix + ".bn3")
    if prefix + ".downsample.0.weight" in sd:
        x = _bn(F.conv2d(x, sd[prefix + ".downsample.0.weight"], stride=stride), sd, prefix + ".downsample.1")
    return F.relu(out + x)


def stem(x, sd):
    """conv1 7x7/2 + bn1 + relu + maxpool 3x3/2, vmgn.py:281-284."""
    x = F.relu(_bn(F.conv2d(x, sd["conv1.weight"], stride=2, padding=3), sd, "bn1"))
    return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


def stage(x, sd, name, blocks, stride):
    for i in range(blocks):
        x = bottleneck(x, sd, "%s.%d" % (name, i), stride if i == 0 else 1)
    return x


def featuremaps(frames, sd):
    """GSTA.featuremaps, vmgn.py:280-290: shared trunk, then the two layer4 copies (last stride 1, vmgn.py:224)."""
    x = stem(frames, sd)
    for name, blocks, stride in RESNET50_STAGES:
        x = stage(x, sd, name, blocks, stride)
    return stage(x, sd, "layer4_1", 3, 1), stage(x, sd, "layer4_2", 3, 1)


def global_feature(x4_1, B, S):
    """AdaptiveAvgPool3d(1) over (S,h,w), vmgn.py:298-300."""
    _, c, h, w = x4_1.shape
    return x4_1.view(B, S, c, h * w).permute(0, 2, 1, 3).reshape(B, c, S * h * w).mean(dim=2)


def part_nodes(x4_2, B, S, splits):
    """Part pooling to graph nodes, vmgn.py:304-308. Node index = frame*P + part, parts ordered as ``splits``."""
    _, c, h, w = x4_2.shape
    cols = []
    for n in splits:
        for j in range(n):
            lo, hi = (j * h) // n, -((-(j + 1) * h) // n)  # AdaptiveAvgPool bin [floor(j*h/n), ceil((j+1)*h/n))
            cols.append(x4_2[:, :, lo:hi, :].mean(dim=(2, 3)))
    nodes = torch.stack(cols, dim=1)  # (B*S, P, c)
    return nodes.reshape(B, S * len(cols), c)


def sim_matrix(f):
    """GraphLayer.get_sim_matrix (dist_method='l2'), vmgn.py:114-120. Diagonal NOT masked."""
    sq = (f * f).sum(dim=2)
    d2 = sq.unsqueeze(1) + sq.unsqueeze(2) - 2 * torch.bmm(f, f.transpose(1, 2))
    return 2 / (d2.clamp(min=1e-12).sqrt().exp() + 1)


def l1_rows(m):
    """F.normalize(m, p=1, dim=2), vmgn.py:157, :162."""
    return m / m.abs().sum(dim=2, keepdim=True).clamp(min=1e-12)


def graph_matrix(f, adj, use_pose=True, learn_graph=True):
    """vmgn.py:155-166."""
    assert use_pose or learn_graph
    if not learn_graph:
        return l1_rows(adj)
    g = l1_rows(sim_matrix(f))
    return (l1_rows(adj) + g) / 2 if use_pose else g


def graph_layer(f, adj, sd, prefix, use_pose=True, learn_graph=True, gamma=0.1, slope=0.1):
    """GraphLayer.forward, vmgn.py:142-172 (eval BatchNorm1d over the N*V rows)."""
    h = f @ sd[prefix + ".linear.weight"].t()
    msg = torch.bmm(graph_matrix(f, adj, use_pose, learn_graph), h)
    n, v, c = msg.shape
    msg = F.leaky_relu(_bn(msg.reshape(n * v, c), sd, prefix + ".bn"), slope).reshape(n, v, c)
    return (1 - gamma) * f + gamma * msg


def attention_pool(f):
    """GSTA._attention_op + mean over parts, vmgn.py:270-278, :313-317. f: (B,S,P,c) -> (B,c)."""
    norm = f.pow(2).sum(dim=3, keepdim=True).sqrt()
    att = norm / norm.abs().sum(dim=1, keepdim=True).clamp(min=1e-12)
    return (f * att).sum(dim=1).mean(dim=1)


def tail(x4_1, x4_2, adj, sd, B, S, splits, num_gb, use_pose=True, learn_graph=True, return_parts=False):
    """Everything after the conv stages, vmgn.py:296-321 (eval): -> (B, 4096)."""
    g_f = global_feature(x4_1, B, S)
    f0 = part_nodes(x4_2, B, S, splits)
    f = f0
    for i in range(num_gb):
        f = graph_layer(f, adj, sd, "graph_layers.%d" % i, use_pose, learn_graph)
    att_f = attention_pool(f.reshape(B, S, sum(splits), f.shape[-1]))
    out = torch.cat([_bn(g_f, sd, "global_bottleneck"), _bn(att_f, sd, "att_bottleneck")], dim=1)
    if return_parts:
        return out, {"g_f": g_f, "nodes": f0, "nodes_out": f, "att_f": att_f}
    return out


def vmgn_eval(x, adj, sd, num_split=4, pyramid_part=True, num_gb=2, use_pose=True, learn_graph=True):
    """GSTA.forward in eval mode, vmgn.py:292-321. x (B,S,3,H,W), adj (B,V,V) -> (B,4096)."""
    B, S = x.shape[:2]
    splits = calc_splits(num_split) if pyramid_part else [num_split]
    x4_1, x4_2 = featuremaps(x.reshape((B * S,) + tuple(x.shape[2:])), sd)
    return tail(x4_1, x4_2, adj, sd, B, S, splits, num_gb, use_pose, learn_graph)


def gsta_eval(x, adj, sd, num_split=4, pyramid_part=True, num_gb=2, use_pose=True, learn_graph=True):
    """Single-branch sibling (torchreid/models/gsta.py:273-298, eval): ResNet50 with one layer4 (last stride 1,
    gsta.py:194) -> part pooling -> the same GraphLayer x num_gb -> attention pooling -> BN ``bottleneck`` -> (B,2048)."""
    B, S = x.shape[:2]
    splits = calc_splits(num_split) if pyramid_part else [num_split]
    f = stem(x.reshape((B * S,) + tuple(x.shape[2:])), sd)
    for name, blocks, stride in RESNET50_STAGES:
        f = stage(f, sd, name, blocks, stride)
    f = stage(f, sd, "layer4", 3, 1)
    nodes = part_nodes(f, B, S, splits)
    for i in range(num_gb):
        nodes = graph_layer(nodes, adj, sd, "graph_layers.%d" % i, use_pose, learn_graph)
    att_f = attention_pool(nodes.reshape(B, S, sum(splits), nodes.shape[-1]))
    return _bn(att_f, sd, "bottleneck")


def pam_module(x, sd, prefix, gamma=None):
    """PAM_Module.forward (position attention), torchreid/models/ganet.py:98-136: 1x1 query / key / value convs WITH bias,
    energy = q^T k over the slice's h*w positions, softmax over the key axis, out = value . attention^T,
    returns gamma * out + x (gamma: the module's scalar parameter, zero at construction)."""
    n, c, h, w = x.shape
    q = F.conv2d(x, sd[prefix + ".query_conv.weight"], sd[prefix + ".query_conv.bias"]).reshape(n, -1, h * w).permute(0, 2, 1)
    k = F.conv2d(x, sd[prefix + ".key_conv.weight"], sd[prefix + ".key_conv.bias"]).reshape(n, -1, h * w)
    att = torch.softmax(torch.bmm(q, k), dim=-1)
    v = F.conv2d(x, sd[prefix + ".value_conv.weight"], sd[prefix + ".value_conv.bias"]).reshape(n, -1, h * w)
    out = torch.bmm(v, att.permute(0, 2, 1)).reshape(n, c, h, w)
    g = sd[prefix + ".gamma"] if gamma is None else gamma
    return g * out + x


def ganet_graph_layer(f, adj, sd, prefix, use_pose=True, learn_graph=True, gamma=0.0, slope=0.1):
    """ganet's GraphLayer.forward, torchreid/models/ganet.py:253-283: like vmgn's, but the DIAGONAL of both the pose graph
    and the learned similarity is masked to zero before the row-L1 normalisation, and the residual form is
    ``input + gamma * h'`` with the constructor default gamma = 0 (ganet.py:175)."""
    h = f @ sd[prefix + ".linear.weight"].t()
    n, v, c = h.shape
    mask = 1.0 - torch.eye(v, dtype=f.dtype).unsqueeze(0)
    graph = None
    if use_pose:
        graph = l1_rows(mask * adj)
    if learn_graph:
        learned = l1_rows(mask * sim_matrix(f))
        graph = learned if graph is None else (graph + learned) / 2
    msg = torch.bmm(graph, h)
    msg = F.leaky_relu(_bn(msg.reshape(n * v, c), sd, prefix + ".bn"), slope).reshape(n, v, c)
    return f + gamma * msg


def ganet_nodes(fmap, sd, B, S, splits):
    """Part nodes of ganet, ganet.py:384-400: the map is cut into h // n row slices per pyramid level (NOT adaptive bins:
    remainder rows are dropped), every slice goes through the position attention module, ``pam_f + slice`` is average
    pooled. -> (B, S*P, c)."""
    _, c, h, w = fmap.shape
    cols = []
    for n in splits:
        step = h // n
        for i in range(n):
            sl = fmap[:, :, step * i: step * (i + 1)]
            cols.append((pam_module(sl, sd, "pam_layer") + sl).mean(dim=(2, 3)))
    return torch.stack(cols, dim=1).reshape(B, S * len(cols), c)


def ganet_eval(x, adj, sd, num_split=4, pyramid_part=True, num_gb=2, use_pose=True, learn_graph=True, graph_gamma=0.0):
    """``ganet`` in eval mode, torchreid/models/ganet.py:378-424: single-branch ResNet50 (layer4 stride 1, :307) -> PAM part
    nodes -> num_gb diagonal-masked graph layers whose outputs are CONCATENATED with their input along the channels (:402-405)
    -> attention pooling over the (num_gb + 1) * 2048 channels -> BN ``bottleneck`` -> (B, (num_gb + 1) * 2048)."""
    B, S = x.shape[:2]
    splits = calc_splits(num_split) if pyramid_part else [num_split]
    f = stem(x.reshape((B * S,) + tuple(x.shape[2:])), sd)
    for name, blocks, stride in RESNET50_STAGES:
        f = stage(f, sd, name, blocks, stride)
    f = stage(f, sd, "layer4", 3, 1)
    outs = [ganet_nodes(f, sd, B, S, splits)]
    for i in range(num_gb):
        outs.append(ganet_graph_layer(outs[-1], adj, sd, "graph_layers.%d" % i, use_pose, learn_graph, graph_gamma))
    cat = torch.cat(outs, dim=2)
    att_f = attention_pool(cat.reshape(B, S, sum(splits), cat.shape[-1]))
    return _bn(att_f, sd, "bottleneck")


# ---- match side ------------------------------------------------------------------------------------------

def euclidean_squared(q, g):
    """torchreid/metrics/distance.py:59-73: ||q||^2 + ||g||^2 - 2 q g^T, no clamp, no sqrt."""
    return (q * q).sum(dim=1, keepdim=True) + (g * g).sum(dim=1, keepdim=True).t() - 2 * (q @ g.t())


def cosine(q, g):
    """torchreid/metrics/distance.py:76-89."""
    qn = q / q.pow(2).sum(dim=1, keepdim=True).sqrt().clamp(min=1e-12)
    gn = g / g.pow(2).sum(dim=1, keepdim=True).sqrt().clamp(min=1e-12)
    return 1 - qn @ gn.t()


def stable_topk(row, k):
    """np.argsort(row)[:k] (rank.py:170-172) made deterministic: ties -> lower index, NaN last."""
    return np.argsort(np.asarray(row), kind="stable")[:k]


def compute_ap(good, junk, order):
    """Compute_AP, rank.py:180-212, on index sets. Returns (ap, cmc[len(order)])."""
    good, junk = set(int(i) for i in good), set(int(i) for i in junk)
    ngood = len(good)
    cmc = np.zeros(len(order))
    old_recall, old_precision, ap = 0, 1.0, 0
    inter = j = good_now = njunk = 0
    for n, gi in enumerate(int(i) for i in order):
        flag = 0
        if gi in good:
            cmc[n - njunk:] = 1
            flag = 1
            good_now += 1
        if gi in junk:
            njunk += 1
            continue
        if flag:
            inter += 1
        recall = inter / ngood
        precision = inter / (j + 1)
        ap += (recall - old_recall) * (old_precision + precision) / 2
        old_recall, old_precision = recall, precision
        j += 1
        if good_now == ngood:
            break
    return ap, cmc


def evaluate_mars(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=50, return_all=False):
    """evaluate_mars, rank.py:160-177: plain means over ALL queries; top-``max_rank`` truncation before AP."""
    distmat = np.asarray(distmat)
    q_pids, g_pids, q_camids, g_camids = map(np.asarray, (q_pids, g_pids, q_camids, g_camids))
    m = distmat.shape[0]
    cmc = np.zeros((m, max_rank))
    ap = np.zeros(m)
    orders = np.zeros((m, max_rank), dtype=np.int64)
    for k in range(m):
        good = np.where((q_pids[k] == g_pids) & (q_camids[k] != g_camids))[0]
        junk = np.where((g_pids == -1) | ((q_pids[k] == g_pids) & (q_camids[k] == g_camids)))[0]
        orders[k] = stable_topk(distmat[k], max_rank)
        ap[k], cmc[k] = compute_ap(good, junk, orders[k])
    if return_all:
        return cmc.mean(axis=0), ap.mean(), ap, cmc, orders
    return cmc.mean(axis=0), ap.mean()


def eval_market1501(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=50, return_all=False):
    """eval_market1501, rank.py:95-150 (and its Cython twin rank_cylib/rank_cy.pyx:154-241): per query the gallery
    samples of the same identity AND camera are discarded; CMC = first correct match; AP over the FULL ranking
    (sum of precision at every correct match / number of correct matches); queries whose identity is absent from the
    (kept) gallery are skipped; both averages run over the valid queries only. The ranking is made deterministic
    (stable sort: ties -> lower gallery index). -> (cmc float32 (max_rank,), mAP) [, ap (m,) with NaN for invalid
    queries, first_hit (m,) 0-based rank of the first match or -1]."""
    distmat = np.asarray(distmat)
    q_pids, g_pids, q_camids, g_camids = map(np.asarray, (q_pids, g_pids, q_camids, g_camids))
    m, n = distmat.shape
    max_rank = min(max_rank, n)
    all_cmc, ap = [], np.full(m, np.nan)
    first_hit = np.full(m, -1, dtype=np.int64)
    for k in range(m):
        order = np.argsort(distmat[k], kind="stable")
        keep = ~((g_pids[order] == q_pids[k]) & (g_camids[order] == q_camids[k]))
        raw = (g_pids[order] == q_pids[k])[keep].astype(np.int64)
        if not raw.any():
            continue
        cmc = np.minimum(raw.cumsum(), 1)
        all_cmc.append(cmc[:max_rank])
        first_hit[k] = int(np.argmax(raw))
        prec = raw.cumsum() / (np.arange(raw.size) + 1.0)
        ap[k] = (prec * raw).sum() / raw.sum()
    assert all_cmc, "Error: all query identities do not appear in gallery"
    cmc = np.asarray(all_cmc).astype(np.float32).sum(0) / float(len(all_cmc))
    mAP = float(np.nanmean(ap))
    if return_all:
        return cmc, mAP, ap, first_hit
    return cmc, mAP


def eval_cuhk03(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=50, num_repeats=10, rng=None):
    """eval_cuhk03, rank.py:22-92 (single-gallery-shot protocol): per valid query the gallery samples of the same identity
    AND camera are discarded; ``num_repeats`` trials each keep ONE randomly chosen sample per gallery identity
    (``np.random.choice`` over the identity's positions in the kept ranking, identities visited in order of first
    appearance, :60-66) and the CMC of the trials is averaged (:67-72); AP as in market1501 over the full kept ranking
    (:74-80). Draws come from numpy's GLOBAL legacy RNG (or ``rng``), one per identity per trial per valid query, in
    exactly that order -- seeding ``np.random`` reproduces the reference's result. Stable ranking."""
    rng = np.random if rng is None else rng
    distmat = np.asarray(distmat)
    q_pids, g_pids, q_camids, g_camids = map(np.asarray, (q_pids, g_pids, q_camids, g_camids))
    m, n = distmat.shape
    max_rank = min(max_rank, n)
    all_cmc, all_ap = [], []
    for k in range(m):
        order = np.argsort(distmat[k], kind="stable")
        keep = ~((g_pids[order] == q_pids[k]) & (g_camids[order] == q_camids[k]))
        raw = (g_pids[order] == q_pids[k])[keep].astype(np.int64)
        if not raw.any():
            continue
        groups = {}
        for pos, pid in enumerate(g_pids[order][keep]):
            groups.setdefault(int(pid), []).append(pos)
        cmc = 0.0
        for _ in range(num_repeats):
            mask = np.zeros(raw.size, dtype=bool)
            for idxs in groups.values():
                mask[rng.choice(idxs)] = True
            trial = np.minimum(raw[mask].cumsum(), 1)
            cmc = cmc + trial[:max_rank].astype(np.float32)
        all_cmc.append(cmc / num_repeats)
        prec = raw.cumsum() / (np.arange(raw.size) + 1.0)
        all_ap.append((prec * raw).sum() / raw.sum())
    assert all_cmc, "Error: all query identities do not appear in gallery"
    cmc = np.asarray(all_cmc).astype(np.float32).sum(0) / float(len(all_cmc))
    return cmc, float(np.mean(all_ap))


def re_ranking(q_g_dist, q_q_dist, g_g_dist, k1=20, k2=6, lambda_value=0.3):
    """k-reciprocal re-ranking, torchreid/utils/re_ranking.py:30-95 (Zhong et al., CVPR 2017), restated step by step:
    joint squared distance matrix of all N = m + n samples, each row scaled by its maximum (:37-38); k-reciprocal
    neighbour sets with the 2/3-overlap expansion (:46-61); Gaussian-kernel weights normalised per row (:63-64); local
    query expansion = mean of the k2 nearest rows (:66-70); Jaccard distance from the element-wise minima (:78-87);
    final = (1 - lambda) jaccard + lambda original, query rows x gallery columns (:89-94). fp32 throughout, rankings
    made deterministic by a stable sort."""
    q_g, q_q, g_g = (np.asarray(a, dtype=np.float32) for a in (q_g_dist, q_q_dist, g_g_dist))
    m, n = q_g.shape
    N = m + n
    X = np.square(np.block([[q_q, q_g], [q_g.T, g_g]]).astype(np.float32)).astype(np.float32)
    D = np.ascontiguousarray((X / X.max(axis=0)).T.astype(np.float32))
    rank = np.argsort(D, axis=1, kind="stable").astype(np.int64)
    half = int(np.around(k1 / 2.0)) + 1

    def recip(i, k):
        fwd = rank[i, :k]
        return fwd[(rank[fwd, :k] == i).any(axis=1)]

    V = np.zeros((N, N), dtype=np.float32)
    for i in range(N):
        base = recip(i, k1 + 1)
        ext = [base]
        for c in base:
            rc = recip(int(c), half)
            if len(np.intersect1d(rc, base)) > 2.0 / 3 * len(rc):
                ext.append(rc)
        idx = np.unique(np.concatenate(ext))
        w = np.exp(-D[i, idx])
        V[i, idx] = w / np.sum(w)
    if k2 != 1:
        V = np.stack([V[rank[i, :k2]].mean(axis=0) for i in range(N)]).astype(np.float32)
    jac = np.zeros((m, N), dtype=np.float32)
    for i in range(m):
        t = np.zeros(N, dtype=np.float32)
        for c in np.nonzero(V[i])[0]:
            rows = np.nonzero(V[:, c])[0]
            t[rows] = t[rows] + np.minimum(V[i, c], V[rows, c])
        jac[i] = 1 - t / (2.0 - t)
    final = jac * (1 - lambda_value) + D[:m] * lambda_value
    return final[:, m:]


def triplet_hard(x, pids, margin=0.3, soft=True):
    """TripletLoss.forward, hard_mine_triplet_loss.py:24-50. Returns (loss, dist_ap, dist_an, idx_ap, idx_an)."""
    n = x.shape[0]
    sq = (x * x).sum(dim=1, keepdim=True)
    dist = (sq + sq.t() - 2 * (x @ x.t())).clamp(min=1e-12).sqrt()
    same = pids.view(n, 1) == pids.view(1, n)
    idx_ap = dist.masked_fill(~same, -math.inf).argmax(dim=1)
    idx_an = dist.masked_fill(same, math.inf).argmin(dim=1)
    dist_ap, dist_an = dist.gather(1, idx_ap.view(n, 1)).view(n), dist.gather(1, idx_an.view(n, 1)).view(n)
    if soft:
        loss = torch.log(1 + torch.exp(dist_ap - dist_an)).mean()
    else:
        loss = F.relu(dist_ap - dist_an + margin).mean()
    return loss, dist_ap, dist_an, idx_ap, idx_an


def xent_label_smooth(logits, targets, epsilon=0.1):
    """CrossEntropyLabelSmooth.forward, losses/cross_entropy_loss.py:26-37: q = (1 - eps) onehot + eps / K,
    loss = sum_k mean_i(-q_ik log_softmax(z)_ik)."""
    n, K = logits.shape
    log_probs = F.log_softmax(logits, dim=1)
    q = torch.zeros_like(log_probs).scatter_(1, targets.view(n, 1), 1)
    q = (1 - epsilon) * q + epsilon / K
    return (-q * log_probs).mean(0).sum()


def deep_supervision(criterion, xs, y):
    """DeepSupervision, losses/__init__.py:9-20: the mean of the criterion over the output list."""
    loss = 0.
    for x in xs:
        loss = loss + criterion(x, y)
    return loss / len(xs)


# ---- pose adjacency (input contract) -------------------------------------------------------------------------

def pose_adjacency(part_sets, num_split=4, pyramid_part=True):
    """adj_graph(method='same'), dataset_loader.py:345-388.

    ``part_sets``: per frame, dict part-name -> set of 1-based stripe ids (already made contiguous, as
    generate_graph does at dataset_loader.py:327-331). Returns a (V,V) 0/1 float32 array, V = frames*P."""
    splits = calc_splits(num_split) if pyramid_part else [num_split]
    P = sum(splits)
    k = int(round(math.log2(num_split)))
    frames = []
    for ps in part_sets:
        ext = {}
        for name, ids in ps.items():
            full = set(ids)
            if pyramid_part:
                for sid in ids:  # stripe sid also lives in its coarser ancestors, dataset_loader.py:354-368
                    for i in range(1, k + 1):
                        full.add(int(math.ceil(sid / 2 ** i)) + (2 ** (k + 1) - 2 ** (k + 1 - i)))
            ext[name] = full
        frames.append(ext)
    adj = np.zeros((P * len(frames), P * len(frames)), dtype=np.float32)
    for name in ("head", "body", "leg"):
        nodes = sorted(sid + t * P - 1 for t, ext in enumerate(frames) for sid in ext.get(name, ()))
        for a in nodes:
            for b in nodes:
                if a != b:
                    adj[a, b] = 1
    return adj


def pose_part_sets(pose, height, num_split=4, threshold=0.1):
    """Keypoints -> body-part stripe sets of ONE frame, generate_graph, dataset_loader.py:308-331.

    ``pose``: (18,3) array of (x, y, confidence) AlphaPose keypoints, or None when no person was detected.
    Each confident keypoint's y is bucketed into one of ``num_split`` horizontal stripes (1-based,
    bisect_right on the stripe borders, clamped to [1, num_split]); a part spanning several stripes is made
    contiguous."""
    import bisect

    if pose is None:
        return {}
    borders = list(np.arange(0, height + 1, height / num_split))
    groups = {"head": [0, 1, 14, 15, 16, 17], "body": [2, 3, 4, 5, 6, 7], "leg": [8, 9, 10, 11, 12, 13]}
    out = {}
    for name, ids in groups.items():
        for kp in ids:
            if pose[kp, 2] > threshold:
                sid = min(num_split, max(1, bisect.bisect_right(borders, pose[kp, 1])))
                out.setdefault(name, set()).add(sid)
    for name, ids in out.items():
        if len(ids) > 1:
            ids.update(range(min(ids), max(ids) + 1))
    return out
