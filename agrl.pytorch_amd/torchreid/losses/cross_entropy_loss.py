"""Label-smoothed cross entropy (reference: torchreid/losses/cross_entropy_loss.py:8-37).

CUDA fp32 logits: loss value + logit gradient from ONE native call (``agrl_xent_label_smooth``, row a14 of SURVEY 8a: the
train step). A label outside [0, num_classes) makes the loss NaN (the reference's ``scatter_`` raises a device assert there; the
kernel never reads out of bounds). ``hip_native = False`` on the instance (or CPU tensors) selects the stock-torch formulation.
"""
from __future__ import absolute_import
from __future__ import division

import torch
import torch.nn as nn


class CrossEntropyLabelSmooth(nn.Module):
    """``-(sum_c q_c log p_c)`` averaged over the batch, ``q = (1-eps) onehot + eps/K``."""

    def __init__(self, num_classes, epsilon=0.1, use_gpu=True):
        super(CrossEntropyLabelSmooth, self).__init__()
        self.num_classes = num_classes
        self.epsilon = epsilon
        self.use_gpu = use_gpu
        self.logsoftmax = nn.LogSoftmax(dim=1)
        self.hip_native = True

    def forward(self, inputs, targets):
        if self.hip_native and inputs.is_cuda and inputs.dtype == torch.float32 and inputs.dim() == 2 and inputs.size(1) == self.num_classes:
            # value + gradient from one native call (agrl_xent_label_smooth)
            from torchreid.models._train_hip import HipXent
            return HipXent.apply(inputs, targets.to(inputs.device), self.epsilon)
        log_probs = self.logsoftmax(inputs)
        onehot = torch.zeros_like(log_probs).scatter_(1, targets.view(-1, 1).to(log_probs.device), 1)
        smooth = (1 - self.epsilon) * onehot + self.epsilon / self.num_classes
        return (-smooth * log_probs).mean(0).sum()
