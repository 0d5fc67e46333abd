"""Weight initialisers the model constructor applies (reference: torchreid/utils/torchtools.py:51-88).

They are matched by class-name substring like the reference so that any module type it would have
touched is touched here too.
"""
from __future__ import absolute_import

from torch import nn


def _kind(m):
    name = type(m).__name__
    for key in ('Linear', 'Conv', 'BatchNorm'):
        if key in name:
            return key
    return None


def weights_init_kaiming(m):
    kind = _kind(m)
    if kind == 'Linear':
        nn.init.kaiming_normal_(m.weight, a=0, mode='fan_out')
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif kind == 'Conv':
        nn.init.kaiming_normal_(m.weight, a=0, mode='fan_in')
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif kind == 'BatchNorm' and m.affine:
        nn.init.normal_(m.weight, 1.0, 0.001)
        nn.init.constant_(m.bias, 0.0)


def weights_init_classifier(m):
    if _kind(m) == 'Linear':
        nn.init.normal_(m.weight.data, std=0.001)
        if m.bias is not None:
            nn.init.constant_(m.bias.data, 0.0)
