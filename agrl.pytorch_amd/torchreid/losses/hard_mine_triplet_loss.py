"""Batch-hard triplet loss (reference: torchreid/losses/hard_mine_triplet_loss.py:8-50).

CUDA fp32 features: ONE native call (``agrl_triplet_loss``) does the O(n^2) mining -- pairwise distances
``sqrt(clamp(|a|^2 + |b|^2 - 2ab, 1e-12))``, hardest positive (max over same id, self included) and hardest negative (min over
other ids) per anchor, replacing the reference's Python loop of 2n masked reductions -- the loss value and the gradient w.r.t.
the features through the 2n selected pairs; nothing returns to the host (an anchor without a negative makes the loss NaN instead
of raising). ``mine`` (``agrl_triplet_hard_mine``) stays available on its own. CPU tensors use the stock-torch formulation.
"""
from __future__ import absolute_import
from __future__ import division

import torch
import torch.nn as nn


def _pair_dist(x, idx):
    sq = x.pow(2).sum(dim=1)
    other = x.index_select(0, idx)
    d2 = sq + sq.index_select(0, idx) - 2 * (x * other).sum(dim=1)
    return d2.clamp(min=1e-12).sqrt()


class _NativeTriplet(torch.autograd.Function):
    """Loss value and d loss / d features from ONE C-ABI call (``agrl_triplet_loss``: mining, loss, gradient through the 2n
    selected pairs); backward only scales the stored gradient by the incoming scalar."""

    @staticmethod
    def forward(ctx, inputs, targets, margin, soft):
        from torchreid import hip_ops as ops
        loss, grad = ops.triplet_loss(inputs.detach().float().contiguous(), targets.detach().to(torch.int32).contiguous(), margin, soft)
        ctx.save_for_backward(grad)
        return loss.view(())

    @staticmethod
    def backward(ctx, go):
        (grad,) = ctx.saved_tensors
        return grad * go, None, None, None


class TripletLoss(nn.Module):
    """``margin``: used when ``soft`` is False (MarginRankingLoss); ``soft``: log(1+exp(ap-an))."""

    def __init__(self, margin=0.3, soft=True):
        super(TripletLoss, self).__init__()
        self.margin = margin
        self.soft = soft
        self.ranking_loss = nn.MarginRankingLoss(margin=margin)
        self.hip_native = True   # False: the stock-torch formulation below on CUDA tensors too (bench.py's baseline step)

    def mine(self, inputs, targets):
        """Indices (idx_ap, idx_an) of the hardest positive / negative of every anchor."""
        n = inputs.size(0)
        if inputs.is_cuda and self.hip_native:
            from torchreid import hip_ops as ops
            _, _, idx_ap, idx_an = ops.triplet_hard_mine(inputs.detach().float().contiguous(),
                                                         targets.detach().to(torch.int32).contiguous())
            # no host round trip here (the loss is called 2-5 x per step through DeepSupervision): an anchor without any
            # negative keeps idx_an = -1 and forward() turns its distance -- and with it the loss -- into NaN instead of
            # raising (the reference's boolean-mask .min() on an empty selection raises, hard_mine_triplet_loss.py:43)
            return idx_ap.long(), idx_an.long()
        with torch.no_grad():
            sq = inputs.pow(2).sum(dim=1, keepdim=True)
            dist = torch.addmm(sq + sq.t(), inputs, inputs.t(), beta=1, alpha=-2).clamp(min=1e-12).sqrt()
            same = targets.view(n, 1).eq(targets.view(1, n))
            idx_ap = dist.masked_fill(~same, float('-inf')).argmax(dim=1)
            if bool(same.all(dim=1).any()):
                raise RuntimeError('an anchor has no negative in the batch')
            idx_an = dist.masked_fill(same, float('inf')).argmin(dim=1)
        return idx_ap, idx_an

    def forward(self, inputs, targets):
        if self.hip_native and inputs.is_cuda and inputs.dtype == torch.float32:
            return _NativeTriplet.apply(inputs, targets, self.margin, self.soft)
        idx_ap, idx_an = self.mine(inputs, targets)
        dist_ap = _pair_dist(inputs, idx_ap)
        missing = idx_an < 0
        dist_an = _pair_dist(inputs, idx_an.clamp(min=0))
        dist_an = torch.where(missing, torch.full_like(dist_an, float('nan')), dist_an)
        if self.soft:
            return torch.log(1 + torch.exp(dist_ap - dist_an)).mean()
        return self.ranking_loss(dist_an, dist_ap, torch.ones_like(dist_an))
