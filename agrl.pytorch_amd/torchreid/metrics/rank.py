"""CMC / mAP evaluation (reference: torchreid/metrics/rank.py:160-238).

Only the branch the reference's driver uses is on the hot path: ``use_metric_mars=True`` ->
``evaluate_mars`` + ``Compute_AP`` (reference rank.py:160-212). On a GPU host it runs on the device:
``agrl_rank_topk`` (exact top-``max_rank`` per query, ties towards the lower gallery index) followed by
``agrl_rank_mars`` (the AP / CMC walk in fp64, operation order of the reference). Without any GPU the
same semantics are evaluated by the numpy host code below.

The market1501 protocol (reference rank.py:95-150 and its Cython twin rank_cylib/rank_cy.pyx:154-241, SURVEY.md
section 8(f) row 1) runs on the device too: ``agrl_rank_market1501`` counts the rank of every correct match directly
from the distance row (no full argsort). The cuhk03 protocol (rank.py:22-92: random single-gallery-shot trials drawn
from numpy's GLOBAL RNG) is defined by that host RNG stream, so its trials are evaluated on the host -- with every
draw of the evaluation made by ONE vectorised ``np.random.randint`` call that consumes the stream exactly like the
reference's per-identity ``np.random.choice`` calls (same values, same final state) -- while the ranking (stable sort)
and the AP (``agrl_rank_market1501``) come from the device.
"""
from __future__ import absolute_import
from __future__ import print_function
from __future__ import division

import numpy as np
import torch


def _stable_topk(scores, k):
    """Indices of the k smallest entries, ascending, ties -> lower index, NaN last."""
    return np.argsort(scores, kind='stable')[:k]


def compute_ap_cmc(good_mask, junk_mask, order, ngood):
    """AP and CMC of one query from its ranked (truncated) gallery list.

    ``good_mask`` / ``junk_mask``: bool arrays over the ranked list; ``ngood``: number of good gallery
    entries overall (not only those inside the truncated list). Follows reference rank.py:180-212
    step for step, including its trapezoidal AP and the junk-shifted CMC index."""
    k = len(order)
    cmc = np.zeros((k,))
    old_recall, old_precision, ap = 0, 1., 0
    hits = seen = good_now = njunk = 0
    for pos in range(k):
        is_good = bool(good_mask[pos])
        if is_good:
            cmc[pos - njunk:] = 1
            good_now += 1
        if junk_mask[pos]:
            njunk += 1
            continue
        if is_good:
            hits += 1
        recall = hits / ngood
        precision = hits / (seen + 1)
        ap += (recall - old_recall) * (old_precision + precision) / 2
        old_recall, old_precision = recall, precision
        seen += 1
        if good_now == ngood:
            break
    return ap, cmc


def _evaluate_mars_host(distmat, q_pids, g_pids, q_camids, g_camids, max_rank):
    num_q, num_g = distmat.shape
    if max_rank > num_g:
        raise ValueError('max_rank={} exceeds the gallery size {}'.format(max_rank, num_g))
    cmc = np.zeros((num_q, max_rank))
    ap = np.zeros(num_q)
    for k in range(num_q):
        same_pid = g_pids == q_pids[k]
        same_cam = g_camids == q_camids[k]
        good = same_pid & ~same_cam
        junk = (g_pids == -1) | (same_pid & same_cam)
        order = _stable_topk(distmat[k], max_rank)
        ap[k], cmc[k] = compute_ap_cmc(good[order], junk[order], order, int(good.sum()))
    return np.mean(cmc, axis=0), np.mean(ap)


def hip_evaluate_mars_device(distmat, q_pids, g_pids, q_camids, g_camids, max_rank):
    """Device-resident evaluation: ``distmat`` fp32 CUDA (m,n); pid/camid int32 CUDA tensors.
    Returns (ap fp64 (m), cmc fp32 (m,max_rank), topk_idx int32 (m,max_rank)) on the device."""
    from torchreid import hip_ops as ops
    idx, _ = ops.rank_topk(distmat, max_rank)
    ap, cmc = ops.rank_mars(idx, q_pids, q_camids, g_pids, g_camids)
    return ap, cmc, idx


def evaluate_mars(distmat, q_pids, g_pids, q_camids, g_camids, max_rank):
    """(CMC (max_rank,), mAP) averaged over ALL queries (reference rank.py:160-177)."""
    q_pids, g_pids = np.asarray(q_pids), np.asarray(g_pids)
    q_camids, g_camids = np.asarray(q_camids), np.asarray(g_camids)
    on_device = isinstance(distmat, torch.Tensor) and distmat.is_cuda
    if not (on_device or torch.cuda.is_available()):
        d = distmat.numpy() if isinstance(distmat, torch.Tensor) else np.asarray(distmat)
        return _evaluate_mars_host(d, q_pids, g_pids, q_camids, g_camids, max_rank)

    from torchreid import _hip
    _hip.lib()
    dev = distmat.device if on_device else torch.device('cuda', torch.cuda.current_device())
    d = distmat if on_device else torch.as_tensor(np.ascontiguousarray(distmat, dtype=np.float32))
    d = d.to(device=dev, dtype=torch.float32)
    if d.stride(-1) != 1:
        d = d.contiguous()
    if max_rank > d.size(1):
        raise ValueError('max_rank={} exceeds the gallery size {}'.format(max_rank, d.size(1)))

    def i32(a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(dev)

    ap, cmc, _ = hip_evaluate_mars_device(d, i32(q_pids), i32(g_pids), i32(q_camids), i32(g_camids), max_rank)
    ap = ap.cpu().numpy()
    if np.isnan(ap).any():
        # the reference divides by ngood == 0 here (rank.py:203)
        raise ZeroDivisionError('query {} has no cross-camera match in the gallery'.format(int(np.isnan(ap).argmax())))
    return np.mean(cmc.cpu().numpy().astype(np.float64), axis=0), np.mean(ap)


def _evaluate_market1501_host(distmat, q_pids, g_pids, q_camids, g_camids, max_rank):
    """numpy evaluation with the reference's semantics (rank.py:95-150), stable ranking."""
    num_q, num_g = distmat.shape
    all_cmc, all_ap = [], []
    for k in range(num_q):
        order = np.argsort(distmat[k], kind='stable')
        keep = ~((g_pids[order] == q_pids[k]) & (g_camids[order] == q_camids[k]))
        raw = (g_pids[order] == q_pids[k])[keep].astype(np.int64)
        if not raw.any():
            continue
        all_cmc.append(np.minimum(raw.cumsum(), 1)[:max_rank])
        all_ap.append(((raw.cumsum() / (np.arange(raw.size) + 1.0)) * raw).sum() / raw.sum())
    assert len(all_ap) > 0, 'Error: all query identities do not appear in gallery'
    return np.asarray(all_cmc).astype(np.float32).sum(0) / float(len(all_ap)), np.mean(all_ap)


def evaluate_market1501(distmat, q_pids, g_pids, q_camids, g_camids, max_rank):
    """(CMC float32 (max_rank,), mAP) over the VALID queries (reference rank.py:95-150)."""
    q_pids, g_pids = np.asarray(q_pids), np.asarray(g_pids)
    q_camids, g_camids = np.asarray(q_camids), np.asarray(g_camids)
    on_device = isinstance(distmat, torch.Tensor) and distmat.is_cuda
    num_g = distmat.shape[1]
    if num_g < max_rank:
        max_rank = num_g
        print('Note: number of gallery samples is quite small, got {}'.format(num_g))
    if not (on_device or torch.cuda.is_available()):
        d = distmat.numpy() if isinstance(distmat, torch.Tensor) else np.asarray(distmat)
        return _evaluate_market1501_host(d, q_pids, g_pids, q_camids, g_camids, max_rank)
    from torchreid import _hip, hip_ops as ops
    _hip.lib()
    dev = distmat.device if on_device else torch.device('cuda', torch.cuda.current_device())
    d = distmat if on_device else torch.as_tensor(np.ascontiguousarray(distmat, dtype=np.float32))
    d = d.to(device=dev, dtype=torch.float32)
    if d.stride(-1) != 1:
        d = d.contiguous()

    def i32(a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(dev)

    ap, cmc, valid = ops.rank_market1501(d, i32(q_pids), i32(q_camids), i32(g_pids), i32(g_camids), max_rank)
    valid = valid.cpu().numpy()
    if (valid < 0).any():
        raise RuntimeError('query {} has more correct matches than agrl_rank_market1501 holds'.format(int((valid < 0).argmax())))
    ok = valid == 1
    assert ok.any(), 'Error: all query identities do not appear in gallery'
    num_valid = float(ok.sum())
    all_cmc = cmc.cpu().numpy()[ok].astype(np.float32).sum(0) / num_valid
    return all_cmc, np.mean(ap.cpu().numpy()[ok])


def _cuhk03_trials(order, q_pids, g_pids, q_camids, g_camids, max_rank, num_repeats):
    """The trial part of the cuhk03 protocol (reference rank.py:39-72) given every query's ranking ``order`` (m, n):
    -> (per-query mean trial CMC (n_valid, L) float32, valid mask (m,)). All draws of the evaluation are made by one
    ``np.random.randint(0, sizes)`` call: per valid query, per trial, one draw per gallery identity in order of first
    appearance in the kept ranking -- value for value what ``np.random.choice(idxs)`` (rank.py:65) draws."""
    m, n = order.shape
    plans, sizes = [], []
    valid = np.zeros(m, dtype=bool)
    for k in range(m):
        pid_sorted = g_pids[order[k]]
        keep = ~((pid_sorted == q_pids[k]) & (g_camids[order[k]] == q_camids[k]))
        kept = pid_sorted[keep]
        if not (kept == q_pids[k]).any():
            continue
        valid[k] = True
        uniq, first, inv, counts = np.unique(kept, return_index=True, return_inverse=True, return_counts=True)
        by_group = np.argsort(inv, kind='stable')           # kept positions, grouped by identity, ascending inside
        start = np.cumsum(counts) - counts
        visit = np.argsort(first, kind='stable')            # identities in order of first appearance (dict order)
        plans.append((by_group, start[visit], int(np.nonzero(uniq[visit] == q_pids[k])[0][0])))
        sizes.append(np.tile(counts[visit], num_repeats))
    if not plans:
        return None, valid
    draws = np.random.randint(0, np.concatenate(sizes))
    cmcs, at = [], 0
    for (by_group, start, gq), sz in zip(plans, sizes):
        G = start.size
        pick = by_group[start[None, :] + draws[at:at + sz.size].reshape(num_repeats, G)]   # chosen position per identity
        at += sz.size
        hit = (pick < pick[:, gq:gq + 1]).sum(axis=1)       # trial rank of the one correct match
        L = min(G, max_rank)
        trial = (np.arange(L)[None, :] >= hit[:, None]).astype(np.float32)
        cmc = np.float32(0.0)
        for r in range(num_repeats):                        # the reference accumulates trial by trial in fp32
            cmc = cmc + trial[r]
        cmcs.append(cmc / num_repeats)
    return cmcs, valid


def evaluate_cuhk03(distmat, q_pids, g_pids, q_camids, g_camids, max_rank, num_repeats=10):
    """(CMC float32, mAP) with the single-gallery-shot protocol, reference rank.py:22-92. Ranking and AP on the device
    when there is one; the random trials follow numpy's global RNG like the reference (seed ``np.random`` to reproduce)."""
    q_pids, g_pids = np.asarray(q_pids), np.asarray(g_pids)
    q_camids, g_camids = np.asarray(q_camids), np.asarray(g_camids)
    on_device = isinstance(distmat, torch.Tensor) and distmat.is_cuda
    num_g = distmat.shape[1]
    if num_g < max_rank:
        max_rank = num_g
        print('Note: number of gallery samples is quite small, got {}'.format(num_g))
    if not (on_device or torch.cuda.is_available()):
        d = distmat.numpy() if isinstance(distmat, torch.Tensor) else np.asarray(distmat)
        order = np.argsort(d, axis=1, kind='stable')
        cmcs, valid = _cuhk03_trials(order, q_pids, g_pids, q_camids, g_camids, max_rank, num_repeats)
        assert cmcs is not None, 'Error: all query identities do not appear in gallery'
        aps = []
        for k in np.nonzero(valid)[0]:
            keep = ~((g_pids[order[k]] == q_pids[k]) & (g_camids[order[k]] == q_camids[k]))
            raw = (g_pids[order[k]] == q_pids[k])[keep].astype(np.int64)
            aps.append(((raw.cumsum() / (np.arange(raw.size) + 1.0)) * raw).sum() / raw.sum())
    else:
        from torchreid import _hip, hip_ops as ops
        _hip.lib()
        dev = distmat.device if on_device else torch.device('cuda', torch.cuda.current_device())
        d = distmat if on_device else torch.as_tensor(np.ascontiguousarray(distmat, dtype=np.float32))
        d = d.to(device=dev, dtype=torch.float32)
        if d.stride(-1) != 1:
            d = d.contiguous()

        def i32(a):
            return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(dev)

        ap, _, dvalid = ops.rank_market1501(d, i32(q_pids), i32(q_camids), i32(g_pids), i32(g_camids), min(max_rank, num_g))
        if num_g <= ops.RANK_ARGSORT_MAX_N:
            order = ops.rank_argsort(d).cpu().numpy().astype(np.int64)      # native full-row stable order (agrl_rank_argsort)
        else:                                                                # a row of composites no longer fits the LDS
            order = torch.sort(d, dim=1, stable=True)[1].cpu().numpy()
        cmcs, valid = _cuhk03_trials(order, q_pids, g_pids, q_camids, g_camids, max_rank, num_repeats)
        assert cmcs is not None, 'Error: all query identities do not appear in gallery'
        dvalid = dvalid.cpu().numpy()
        if (dvalid < 0).any():
            raise RuntimeError('query {} has more correct matches than agrl_rank_market1501 holds'.format(int((dvalid < 0).argmax())))
        assert np.array_equal(dvalid == 1, valid)
        aps = ap.cpu().numpy()[valid]
    num_valid = float(valid.sum())
    all_cmc = np.asarray(cmcs).astype(np.float32).sum(0) / num_valid
    return all_cmc, np.mean(aps)


def evaluate_rank(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=50, use_metric_cuhk03=False,
                  use_metric_market1501=False, use_metric_mars=False, use_cython=True):
    """Evaluate CMC and mAP; same signature and dispatch order as reference rank.py:215-238
    (returns None when no metric flag is set)."""
    if use_metric_cuhk03:
        return evaluate_cuhk03(distmat, q_pids, g_pids, q_camids, g_camids, max_rank)
    if use_metric_market1501:  # ``use_cython`` selects between two implementations of the same protocol upstream
        return evaluate_market1501(distmat, q_pids, g_pids, q_camids, g_camids, max_rank)
    elif use_metric_mars:
        return evaluate_mars(distmat, q_pids, g_pids, q_camids, g_camids, max_rank)
