"""Model registry (reference: torchreid/models/__init__.py:17-41).

``vmgn`` -- the model BASELINE.json's north_star names -- and its two siblings that share the kernels (SURVEY.md section 8f
row 4): the single-branch predecessor ``gsta`` (same GraphLayer) and ``ganet`` (position-attention part nodes, diagonal-masked
graph layers) are provided by this build; the reference's other sibling architectures are
out of the hot path.
"""
from __future__ import absolute_import

import inspect
import os
import shutil

from .vmgn import *
from .gsta import *
from .ganet import *

__model_factory = {
    'vmgn': vmgn,
    'gsta': gsta,
    'ganet': ganet,
}


def get_names():
    return list(__model_factory.keys())


def init_model(name, *args, **kwargs):
    if name not in __model_factory:
        raise KeyError("Unknown model: {}".format(name))
    factory = __model_factory[name]
    if 'save_dir' in kwargs:
        # the driver keeps a copy of the model definition next to its logs (reference :37-40)
        src = inspect.getfile(factory)
        shutil.copyfile(src, os.path.join(os.path.abspath(kwargs['save_dir']), os.path.basename(src)))
    return factory(*args, **kwargs)
