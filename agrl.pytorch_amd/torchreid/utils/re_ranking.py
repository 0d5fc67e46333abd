"""k-reciprocal re-ranking (Zhong et al., CVPR 2017): drop-in for ``torchreid/utils/re_ranking.py`` of
weleen/AGRL.pytorch (:30-95), the ``--re-rank`` post-process of ``test()`` (train_vidreid_xent_htri.py:523-527).

Same signature and return type (numpy in, numpy ``(num_query, num_gallery)`` float32 out). On a GPU host the whole
procedure runs on the device (``agrl_re_ranking``: joint distance matrix, exact top-(k1+1) neighbour lists,
k-reciprocal expansion, local query expansion, Jaccard distance -- four dense (m+n)^2 fp32 matrices in HBM instead of
the reference's Python loops); CUDA tensors are accepted as well and then a CUDA tensor is returned. Without any GPU the
numpy host code below evaluates the same procedure (the reference's own CPU-runnable configuration).
"""
from __future__ import absolute_import
from __future__ import division

import numpy as np
import torch


def _re_ranking_host(q_g, q_q, g_g, k1, k2, lambda_value):
    m, n = q_g.shape
    total = m + n
    joint = np.square(np.block([[q_q, q_g], [q_g.T, g_g]]).astype(np.float32)).astype(np.float32)
    dist = np.ascontiguousarray((joint / joint.max(axis=0)).T.astype(np.float32))
    rank = np.argsort(dist, axis=1, kind='stable')
    half = int(np.around(k1 / 2.)) + 1

    def reciprocal(i, k):
        fwd = rank[i, :k]
        return fwd[(rank[fwd, :k] == i).any(axis=1)]

    weights = np.zeros((total, total), dtype=np.float32)
    for i in range(total):
        base = reciprocal(i, k1 + 1)
        members = [base]
        for cand in base:
            cset = reciprocal(int(cand), half)
            if len(np.intersect1d(cset, base)) > 2. / 3 * len(cset):
                members.append(cset)
        idx = np.unique(np.concatenate(members))
        w = np.exp(-dist[i, idx])
        weights[i, idx] = w / np.sum(w)
    if k2 != 1:
        weights = np.stack([weights[rank[i, :k2]].mean(axis=0) for i in range(total)]).astype(np.float32)
    jaccard = np.zeros((m, total), dtype=np.float32)
    for i in range(m):
        acc = np.zeros(total, dtype=np.float32)
        for c in np.nonzero(weights[i])[0]:
            rows = np.nonzero(weights[:, c])[0]
            acc[rows] = acc[rows] + np.minimum(weights[i, c], weights[rows, c])
        jaccard[i] = 1 - acc / (2. - acc)
    final = jaccard * (1 - lambda_value) + dist[:m] * lambda_value
    return final[:, m:]


def re_ranking(q_g_dist, q_q_dist, g_g_dist, k1=20, k2=6, lambda_value=0.3):
    tensors = [isinstance(a, torch.Tensor) for a in (q_g_dist, q_q_dist, g_g_dist)]
    on_device = all(tensors) and q_g_dist.is_cuda
    if not (on_device or torch.cuda.is_available()):
        arrs = [(a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)).astype(np.float32) for a in (q_g_dist, q_q_dist, g_g_dist)]
        return _re_ranking_host(arrs[0], arrs[1], arrs[2], k1, k2, lambda_value)
    from torchreid import _hip, hip_ops as ops
    _hip.lib()
    dev = q_g_dist.device if on_device else torch.device('cuda', torch.cuda.current_device())

    def up(a):
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32))
        return t.to(device=dev, dtype=torch.float32).contiguous()

    out = ops.re_ranking(up(q_g_dist), up(q_q_dist), up(g_g_dist), k1, k2, lambda_value)
    return out if on_device else out.cpu().numpy()
