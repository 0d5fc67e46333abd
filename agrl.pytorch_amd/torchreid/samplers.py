"""Identity-balanced batch samplers (reference: torchreid/samplers.py:18-111).

The reference's driver obtains ``np``, ``torch``, ``random`` and ``copy`` through
``from torchreid.samplers import *`` (train_vidreid_xent_htri.py:28), so this module binds those names
at top level and deliberately defines no ``__all__``.
"""
from __future__ import absolute_import
from __future__ import division

from collections import defaultdict
import copy
import random

import numpy as np
import torch
from torch.utils.data.sampler import *  # noqa: F401,F403 -- the driver evals e.g. 'SequentialSampler' by name (:227)


class RandomSampler(RandomSampler):  # noqa: F405
    """``--train-sampler RandomSampler``: the stock sampler behind the constructor every train sampler is called with,
    ``(data_source, batch_size=..., num_instances=...)`` (train_vidreid_xent_htri.py:227; reference samplers.py:12-15)."""

    def __init__(self, data_source, batch_size, num_instances):
        super(RandomSampler, self).__init__(data_source)


class RandomIdentitySampler(Sampler):
    """P identities x K instances per batch: yields indices so that every consecutive group of
    ``num_instances`` samples shares an identity and every ``batch_size`` block holds
    ``batch_size // num_instances`` identities (reference samplers.py:18-76)."""

    def __init__(self, data_source, batch_size, num_instances):
        self.data_source = data_source
        self.batch_size = batch_size
        self.num_instances = num_instances
        self.num_pids_per_batch = batch_size // num_instances
        self.index_dic = defaultdict(list)
        for index, (_, pid, _) in enumerate(data_source):
            self.index_dic[pid].append(index)
        self.pids = list(self.index_dic.keys())
        self.length = 0
        for pid in self.pids:
            n = max(len(self.index_dic[pid]), num_instances)
            self.length += n - n % num_instances

    def __iter__(self):
        chunks = defaultdict(list)
        for pid in self.pids:
            idxs = copy.deepcopy(self.index_dic[pid])
            if len(idxs) < self.num_instances:
                idxs = list(np.random.choice(idxs, size=self.num_instances, replace=True))
            random.shuffle(idxs)
            for start in range(0, len(idxs) - self.num_instances + 1, self.num_instances):
                chunks[pid].append(idxs[start:start + self.num_instances])
        alive = [pid for pid in self.pids if chunks[pid]]
        order = []
        while len(alive) >= self.num_pids_per_batch:
            for pid in random.sample(alive, self.num_pids_per_batch):
                order.extend(chunks[pid].pop(0))
                if not chunks[pid]:
                    alive.remove(pid)
        return iter(order)

    def __len__(self):
        return self.length  # the constructor-time estimate, as in the reference (an epoch may yield fewer)


class RandomIdentitySamplerV1(Sampler):
    """For every identity (random order) draw ``num_instances`` samples (reference samplers.py:79-111)."""

    def __init__(self, data_source, num_instances=4, **kwargs):
        self.data_source = data_source
        self.num_instances = num_instances
        self.index_dic = defaultdict(list)
        for index, (_, pid, _) in enumerate(data_source):
            self.index_dic[pid].append(index)
        self.pids = list(self.index_dic.keys())
        self.num_identities = len(self.pids)

    def __iter__(self):
        order = []
        for i in torch.randperm(self.num_identities):
            pool = self.index_dic[self.pids[int(i)]]
            replace = len(pool) < self.num_instances
            order.extend(np.random.choice(pool, size=self.num_instances, replace=replace))
        return iter(order)

    def __len__(self):
        return self.num_identities * self.num_instances
