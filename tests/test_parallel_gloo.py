"""world_size-2 gloo test of the data-parallel match logic (no GPU): tracklet shards, all-gather of embeddings,
gallery shards, candidate merge -> identical to the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vmgn_oracle as O


def _cpu_topk(d, k):
    order = np.stack([O.stable_topk(row, k) for row in d.numpy()])
    idx = torch.from_numpy(order)
    return idx, torch.gather(d, 1, idx)


def _worker(rank, world, port, tmp):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "agrl.pytorch_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    from torchreid import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    emb = torch.randn((6 * world, 64), generator=g)            # what the forward would produce, all tracklets
    gallery = torch.randn((101, 64), generator=g)
    gallery[40] = gallery[7]                                    # exact ties across shards
    gallery[90] = gallery[7]
    lo, hi = parallel.shard_bounds(emb.size(0), rank, world)
    q_all = parallel.all_gather_rows(emb[lo:hi].clone())
    assert torch.equal(q_all, emb)
    glo, ghi = parallel.shard_bounds(gallery.size(0), rank, world)
    idx, val = parallel.sharded_topk(q_all, gallery[glo:ghi], glo, 20, O.cosine, _cpu_topk)
    counts = [parallel.shard_bounds(gallery.size(0), r_, world)[1] - parallel.shard_bounds(gallery.size(0), r_, world)[0] for r_ in range(world)]
    g_back = parallel.all_gather_ragged_rows(gallery[glo:ghi].clone(), counts)
    assert torch.equal(g_back, gallery)
    torch.save((idx, val), os.path.join(tmp, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_shard_bounds():
    from torchreid.parallel import shard_bounds
    for n in (0, 1, 7, 8, 12180):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_sharded_match_equals_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(0)
    emb = torch.randn((6 * world, 64), generator=g)
    gallery = torch.randn((101, 64), generator=g)
    gallery[40] = gallery[7]
    gallery[90] = gallery[7]
    ref_idx, ref_val = _cpu_topk(O.cosine(emb, gallery), 20)
    for r in range(world):
        idx, val = torch.load(os.path.join(str(tmp_path), "r%d.pt" % r))
        assert torch.equal(idx, ref_idx)
        assert torch.allclose(val, ref_val, atol=1e-6)
