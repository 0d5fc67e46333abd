from __future__ import absolute_import

from .accuracy import accuracy
from .rank import evaluate_rank
from .distance import compute_distance_matrix
