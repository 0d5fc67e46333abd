"""The small synthetic query / gallery split on which the reference driver's own ``test()`` was run (tests/golden/
make_test_harness.py -> tests/golden/test_harness.npz): labels and seeded loaders shared by the generator (build container) and
the tests, so that both sides see bit-identical inputs without shipping them.

10 identities, seq_len 4. Query: one tracklet per identity on camera 0. Gallery: five per identity on cameras 1..5, one more per
identity on the query's own camera (junk for that query: same pid, same camera) and four distractors with pid -1 -- every query
has cross-camera matches and the gallery holds 64 >= 50 rows, the two preconditions of evaluate_mars."""
import numpy as np
import torch

from recipe import synthetic_adj, synthetic_clips

N_ID, S = 10, 4
Q_SEED, G_SEED = 300, 700
DENSE_CLIPS = 2


def make_split():
    q_pids = np.arange(N_ID, dtype=np.int64)
    q_cams = np.zeros(N_ID, dtype=np.int64)
    g_pids = np.concatenate([np.repeat(np.arange(N_ID), 5), np.arange(N_ID), -np.ones(4, dtype=np.int64)]).astype(np.int64)
    g_cams = np.concatenate([np.tile(np.arange(1, 6), N_ID), np.zeros(N_ID, dtype=np.int64), np.array([1, 2, 3, 0])]).astype(np.int64)
    return q_pids, q_cams, g_pids, g_cams


def _idents(pids):
    return [int(p) if p >= 0 else 1000 + i for i, p in enumerate(pids)]   # distractors get their own pattern


def loader(pids, cams, seed, bs=8):
    """the 'evenly' test sampler's batches: (imgs (b,S,3,256,128), pids (b,), camids (b,), adj (b,V,V))"""
    idents = _idents(pids)
    for i in range(0, len(pids), bs):
        sl = slice(i, i + bs)
        b = len(pids[sl])
        yield (synthetic_clips(b, S, seed=seed + i, identities=idents[sl]), torch.from_numpy(pids[sl]), torch.from_numpy(cams[sl]),
               synthetic_adj(b, S, seed=seed + i))


def dense_loader(pids, cams, seed):
    """the 'dense' test sampler's batches: ONE tracklet of n clips per batch: (imgs (1,n,S,3,256,128), pids (1,), camids (1,),
    adj (1,n,V,V)) -- the reference pools the n clip embeddings with view(n, 1, -1).mean(0) (train_vidreid_xent_htri.py:471-476)"""
    idents = _idents(pids)
    for i in range(len(pids)):
        x = synthetic_clips(DENSE_CLIPS, S, seed=seed + 31 * i, identities=[idents[i]] * DENSE_CLIPS)
        adj = synthetic_adj(DENSE_CLIPS, S, seed=seed + 31 * i)
        yield x.unsqueeze(0), torch.from_numpy(pids[i:i + 1]), torch.from_numpy(cams[i:i + 1]), adj.unsqueeze(0)
