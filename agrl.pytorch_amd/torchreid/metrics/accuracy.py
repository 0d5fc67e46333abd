"""Top-k accuracy over a list of classifier outputs (reference: torchreid/metrics/accuracy.py:9-33).
Logging only -- stock PyTorch, not part of the hot path."""
from __future__ import absolute_import
from __future__ import division

import numpy as np
import torch


def accuracy(output, target, topk=(1,)):
    """``output``: tensor (batch, classes) or list/tuple of them; returns ndarray (n_outputs, len(topk))
    of fractions in [0, 1] (a 1-D array of length len(topk) per output)."""
    outputs = output if isinstance(output, (tuple, list)) else [output]
    kmax = max(topk)
    n = target.size(0)
    rows = []
    with torch.no_grad():
        for logits in outputs:
            pred = logits.topk(kmax, dim=1, largest=True, sorted=True)[1]
            hit = pred.eq(target.view(-1, 1).expand_as(pred))
            rows.append([hit[:, :k].float().sum().item() / n for k in topk])
    return np.array(rows)
