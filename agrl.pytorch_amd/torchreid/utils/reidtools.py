"""``torchreid.utils.reidtools``: the part-split helper the model uses (reference torchreid/utils/reidtools.py:13-15)
and the ranked-results dump the driver imports from the same module (``visualize_ranked_results``,
train_vidreid_xent_htri.py:25, :535-540; reference reidtools.py:18-80). This file shadows the reference's module under
the PYTHONPATH overlay, so it carries both names."""
from __future__ import absolute_import
from __future__ import print_function

import os
import os.path as osp
import shutil

import numpy as np


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


def _copy_ranked(src, dst_dir, rank, prefix):
    """One query / gallery entry -> ``dst_dir``: a tracklet (tuple / list of frame paths) becomes a sub-directory
    ``<prefix>_topNNN/``, a single image a file ``<prefix>_topNNN_name_<basename>``."""
    tag = prefix + '_top' + str(rank).zfill(3)
    if isinstance(src, (tuple, list)):
        sub = osp.join(dst_dir, tag)
        os.makedirs(sub, exist_ok=True)
        for path in src:
            shutil.copy(path, sub)
    else:
        shutil.copy(src, osp.join(dst_dir, tag + '_name_' + osp.basename(src)))


def visualize_ranked_results(distmat, dataset, save_dir='log/ranked_results', topk=20):
    """For every query copy its ``topk`` nearest gallery entries (same identity seen by the same camera skipped) next
    to the query itself under ``save_dir/id<query>_cam<camid>/``. ``distmat`` (num_query, num_gallery) array-like on
    the host; ``dataset.query`` / ``dataset.gallery`` lists of (path | tuple of paths, pid, camid)."""
    distmat = np.asarray(distmat)
    num_q, num_g = distmat.shape
    print('Visualizing top-{} ranks'.format(topk))
    print('# query: {}\n# gallery {}'.format(num_q, num_g))
    print("Saving images to '{}'".format(save_dir))
    assert num_q == len(dataset.query)
    assert num_g == len(dataset.gallery)
    order = np.argsort(distmat, axis=1)
    os.makedirs(save_dir, exist_ok=True)
    for q in range(num_q):
        q_path, q_pid, q_cam = dataset.query[q]
        name = q_path[0].split('/')[-2] if isinstance(q_path, (tuple, list)) else osp.basename(q_path)
        q_dir = osp.join(save_dir, 'id' + name + '_cam' + str(q_cam))
        os.makedirs(q_dir, exist_ok=True)
        _copy_ranked(q_path, q_dir, 0, 'query')
        rank = 1
        for g in order[q]:
            g_path, g_pid, g_cam = dataset.gallery[g]
            if q_pid == g_pid and q_cam == g_cam:
                continue
            _copy_ranked(g_path, q_dir, rank, 'gallery')
            rank += 1
            if rank > topk:
                break
    print('Done')
