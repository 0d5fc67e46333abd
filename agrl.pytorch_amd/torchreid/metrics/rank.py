"""CMC / mAP evaluation (reference: torchreid/metrics/rank.py:160-238).

Only the branch the reference's driver uses is on the hot path: ``use_metric_mars=True`` ->
``evaluate_mars`` + ``Compute_AP`` (reference rank.py:160-212). On a GPU host it runs on the device:
``agrl_rank_topk`` (exact top-``max_rank`` per query, ties towards the lower gallery index) followed by
``agrl_rank_mars`` (the AP / CMC walk in fp64, operation order of the reference). Without any GPU the
same semantics are evaluated by the numpy host code below.

The market1501 / cuhk03 protocols (reference rank.py:22-150 and its Cython twin) are SURVEY.md section 8(f)
"next" rows and are not built yet: asking for them raises NotImplementedError rather than returning
something else.
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


def evaluate_rank(distmat, q_pids, g_pids, q_camids, g_camids, max_rank=50, use_metric_cuhk03=False,
                  use_metric_market1501=False, use_metric_mars=False, use_cython=True):
    """Evaluate CMC and mAP; same signature and dispatch order as reference rank.py:215-238
    (returns None when no metric flag is set)."""
    if use_metric_market1501 or use_metric_cuhk03:
        raise NotImplementedError(
            'market1501 / cuhk03 protocols are outside the vmgn hot path of this build (the reference driver '
            'only uses use_metric_mars=True, train_vidreid_xent_htri.py:531)')
    elif use_metric_mars:
        return evaluate_mars(distmat, q_pids, g_pids, q_camids, g_camids, max_rank)
