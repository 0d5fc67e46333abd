from __future__ import absolute_import
from __future__ import division
from __future__ import print_function

from .cross_entropy_loss import CrossEntropyLabelSmooth
from .hard_mine_triplet_loss import TripletLoss


def DeepSupervision(criterion, xs, y):
    """Mean of ``criterion(x, y)`` over the outputs ``xs`` (reference losses/__init__.py:9-20)."""
    total = 0.
    for x in xs:
        total += criterion(x, y)
    return total / len(xs)
