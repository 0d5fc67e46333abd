"""Part-split helper used by the model (mirrors torchreid/utils/reidtools.py:13-15 of the reference)."""
from __future__ import absolute_import


def calc_splits(num_split):
    """Pyramid of horizontal-stripe counts: every divisor of ``num_split`` from large to small.

    ``calc_splits(4) == [4, 2, 1]``. ``num_split`` must be a power of two (reference: reidtools.py:14).
    """
    if num_split <= 0 or (num_split & (num_split - 1)) != 0:
        raise AssertionError('num_split must be the power of 2, {} is not supported'.format(num_split))
    out = []
    n = num_split
    while n >= 1:
        out.append(n)
        n //= 2
    return out
