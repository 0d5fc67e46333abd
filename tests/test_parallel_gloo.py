"""world_size-2 / 3 / 8 gloo tests of the data-parallel match logic (no GPU): tracklet shards, all-gather of embeddings,
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
    assert (r, w) == (rank, world) and parallel.collectives_active()
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


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_match_equals_single_process(tmp_path, world):
    """world 3: ragged shards (101 = 34 + 34 + 33); world 8: every gallery shard (12-13 rows) is SMALLER than k = 20, so every
    rank pads its candidate list -- the merged list must still be the single-process top-20, ties (rows 7 / 40 / 90, three
    different shards) in gallery-index order."""
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


def _forced_worker(rank, world, port, tmp):
    os.environ["AGRL_DIST_FORCE_GROUP"] = "1"
    _worker(rank, world, port, tmp)


@pytest.mark.timeout(300)
def test_forced_group_at_world_size_one_runs_the_collective_path(tmp_path):
    """AGRL_DIST_FORCE_GROUP=1 (what tests/test_gpu_configs.py::test_rccl_path_on_one_gpu uses with the nccl backend): a process group
    of ONE rank is built and all_gather_rows / sharded_topk take their collective + candidate-merge branches instead of the
    single-process shortcut -- same result."""
    from torchreid import parallel
    assert not parallel.collectives_active()
    mp.spawn(_forced_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    g = torch.Generator().manual_seed(0)
    emb = torch.randn((6, 64), generator=g)
    gallery = torch.randn((101, 64), generator=g)
    gallery[40] = gallery[7]
    gallery[90] = gallery[7]
    ref_idx, ref_val = _cpu_topk(O.cosine(emb, gallery), 20)
    idx, val = torch.load(os.path.join(str(tmp_path), "r0.pt"))
    assert torch.equal(idx, ref_idx) and torch.allclose(val, ref_val, atol=1e-6)


def _train_worker(rank, world, port, tmp):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "agrl.pytorch_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    from recipe import recipe_state_dict, synthetic_adj, synthetic_clips
    from torchreid import losses, models, parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_from_env("gloo")
    torch.set_num_threads(2)
    m, x, adj, pids = _train_problem(models, recipe_state_dict, synthetic_clips, synthetic_adj)
    lo, hi = parallel.shard_bounds(x.size(0), rank, world)
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    loss = parallel.train_step(m, x[lo:hi], adj[lo:hi], pids[lo:hi], losses.CrossEntropyLabelSmooth(5, use_gpu=False),
                               losses.TripletLoss(margin=0.3, soft=True), opt)
    grads = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    # the overlapped form: .grad are views into flat buckets, every bucket's all-reduce is issued from a post-accumulate
    # hook while backward is still running (small buckets here so that several are in flight) -> the same gradients
    m2, _, _, _ = _train_problem(models, recipe_state_dict, synthetic_clips, synthetic_adj)
    buckets = parallel.GradientBuckets(m2.parameters(), bucket_bytes=8 << 20)
    assert len(buckets.buckets) > 3
    opt2 = torch.optim.SGD(m2.parameters(), lr=0.0)
    for _ in range(2):   # twice: zero_grad() must keep the views and re-arm the counters
        loss2 = parallel.train_step(m2, x[lo:hi], adj[lo:hi], pids[lo:hi], losses.CrossEntropyLabelSmooth(5, use_gpu=False),
                                    losses.TripletLoss(margin=0.3, soft=True), opt2, buckets=buckets)
    assert abs(loss2[0] - loss[0]) < 1e-6 * abs(loss[0])
    for k, p in m2.named_parameters():
        if k in grads:
            assert p.grad.data_ptr() >= buckets.buckets[buckets._bucket_of[p]][0].data_ptr()
            assert torch.allclose(p.grad, grads[k], rtol=1e-5, atol=1e-7 * grads[k].abs().max().item() + 1e-12), k
    buckets.remove()
    torch.save((loss, grads), os.path.join(tmp, "t%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _train_problem(models, recipe_state_dict, synthetic_clips, synthetic_adj):
    torch.manual_seed(0)
    m = models.init_model("vmgn", num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=1, num_scale=1,
                          pyramid_part=True, use_pose=True, learn_graph=True, consistent_loss=False)
    m.load_state_dict(recipe_state_dict(m.state_dict(), seed=1))
    pids = torch.tensor([0, 0, 1, 1])
    x = synthetic_clips(4, 2, H=64, W=32, seed=3, identities=pids.tolist())
    adj = synthetic_adj(4, 2, seed=3)
    return m, x, adj, pids


@pytest.mark.timeout(600)
def test_sharded_train_step_equals_replicated_global_step(tmp_path):
    """BASELINE config 4's data-parallel train step: 2 ranks x 2 tracklets (per-replica BatchNorm statistics, global
    xent + batch-hard triplet, gradient all-reduce) == one process that runs the two replicas' forwards itself, gathers
    and backpropagates -- nn.DataParallel's arithmetic (train_vidreid_xent_htri.py:318, :399-411)."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "agrl.pytorch_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from recipe import recipe_state_dict, synthetic_adj, synthetic_clips
    from torchreid import losses, models
    world = 2
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    m, x, adj, pids = _train_problem(models, recipe_state_dict, synthetic_clips, synthetic_adj)
    m.train()
    outs, feats = [], []
    for r in range(world):  # the replicas' forwards: BatchNorm batch statistics are per replica
        o, f = m(x[2 * r:2 * r + 2], adj[2 * r:2 * r + 2])
        outs.append(o)
        feats.append(f)
    outs = [torch.cat([outs[0][i], outs[1][i]]) for i in range(len(outs[0]))]
    feats = [torch.cat([feats[0][i], feats[1][i]]) for i in range(len(feats[0]))]
    loss = losses.DeepSupervision(losses.CrossEntropyLabelSmooth(5, use_gpu=False), outs, pids) + \
        losses.DeepSupervision(losses.TripletLoss(margin=0.3, soft=True), feats, pids)
    loss.backward()
    ref = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    for r in range(world):
        (l, lx, lh), grads = torch.load(os.path.join(str(tmp_path), "t%d.pt" % r))
        assert abs(l - loss.item()) < 1e-5 * abs(loss.item())
        assert set(grads) == set(ref)
        worst = max(((grads[k] - ref[k]).abs().max() / ref[k].abs().max().clamp(min=1e-12)).item() for k in ref)
        assert worst < 1e-3, worst


@pytest.mark.timeout(300)
def test_bucketed_step_skips_parameters_without_gradient_like_the_plain_step():
    """An htri-only step leaves the classifiers without a gradient: the plain path's optimizer.zero_grad() makes their
    .grad None and Adam skips them; the bucketed path keeps zero-filled .grad views and must park them around
    optimizer.step() (GradientBuckets.only_touched) -- otherwise weight decay and the moment updates move those weights."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "agrl.pytorch_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from recipe import recipe_state_dict, synthetic_adj, synthetic_clips
    from torchreid import losses, models, parallel
    results = []
    for use_buckets in (False, True):
        m, x, adj, pids = _train_problem(models, recipe_state_dict, synthetic_clips, synthetic_adj)
        before = {k: v.clone() for k, v in m.state_dict().items()}
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=5e-2)
        buckets = parallel.GradientBuckets(m.parameters(), bucket_bytes=8 << 20) if use_buckets else None
        for _ in range(2):
            parallel.train_step(m, x, adj, pids, losses.CrossEntropyLabelSmooth(5, use_gpu=False),
                                losses.TripletLoss(margin=0.3, soft=True), opt, htri_only=True, buckets=buckets)
        if buckets is not None:
            for p in m.parameters():  # the views are back after the step
                assert p.grad is not None or not p.requires_grad
            buckets.remove()
        results.append((before, {k: v.clone() for k, v in m.state_dict().items()}))
    (b0, plain), (b1, bucketed) = results
    for k in ("global_classifier.weight", "att_classifier.weight"):
        assert torch.equal(plain[k], b0[k]), k            # untouched by the plain step
        assert torch.equal(bucketed[k], b1[k]), k         # and by the bucketed one
    for k in plain:
        assert torch.allclose(plain[k].float(), bucketed[k].float(), rtol=1e-5, atol=1e-7), k


class _StubModel(torch.nn.Module):
    """(b, S, 3, H, W), adj -> (b, 8): stands in for the eval forward in the collective-discipline test below."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(1))

    def forward(self, x, adj):
        return x.flatten(1)[:, :8] * self.w


def _local_only_worker(rank, world, port, tmp):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "agrl.pytorch_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    from torchreid import evaluation, parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    parallel.init_from_env("gloo")
    g = torch.Generator().manual_seed(0)
    emb = torch.randn((6, 64), generator=g)
    gallery = torch.randn((101, 64), generator=g)
    batches = [(torch.randn((4, 2, 3, 4, 4), generator=g), np.arange(4), np.zeros(4), torch.zeros((4, 14, 14)))]
    model = _StubModel()
    # the default under a group is COLLECTIVE: both ranks call, both return (the MAX all-reduce of the non-finite flag pairs up)
    f, _, _ = evaluation.extract_features(model, batches, prefetch=False)
    assert f.shape == (4, 8)
    if rank == 0:
        # ... and local_only is the opt-out: rank 0 alone, whole gallery, no collective, no rank offset
        f0, _, _ = evaluation.extract_features(model, batches, prefetch=False, local_only=True)
        assert torch.equal(f0, f)
        idx, val = parallel.sharded_topk(emb, gallery, 0, 20, O.cosine, _cpu_topk, local_only=True)
        torch.save((idx, val), os.path.join(tmp, "local.pt"))
    dist.barrier()   # rank 1 has been waiting here: a collective hidden in the local_only calls would have paired with it
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_local_only_calls_issue_no_collective_under_a_group(tmp_path):
    """Round-5 advice: under an initialised group ``extract_features`` all-reduces its non-finite flag by default (every rank
    raises together), and ``local_only=True`` -- what ``evaluation.evaluate`` passes to all three of its stages -- makes the
    extraction and ``sharded_topk`` safe to call from ONE rank: no collective, global indices. (The device form of the same
    check, with the real model and ``evaluate`` itself: tests/test_gpu_dist.py.)"""
    mp.spawn(_local_only_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    g = torch.Generator().manual_seed(0)
    emb = torch.randn((6, 64), generator=g)
    gallery = torch.randn((101, 64), generator=g)
    ref_idx, ref_val = _cpu_topk(O.cosine(emb, gallery), 20)
    idx, val = torch.load(os.path.join(str(tmp_path), "local.pt"))
    assert torch.equal(idx, ref_idx) and torch.allclose(val, ref_val, atol=1e-6)


def test_query_operand_cache_is_keyed_on_identity_not_address():
    """Round-5 review: QueryOperandCache used to key on (data_ptr, shape, _version, device). Free the embedding, let the
    allocator hand the SAME address to a fresh tensor of the same shape (version 0 as well, filled by raw kernels that never
    bump it) and it would have served the previous batch's normalised rows. Now: a weak reference, identity only. The recycled
    address is produced deterministically here: two tensor objects over one arena, the contents rewritten behind torch's back."""
    import ctypes
    import gc
    from torchreid import hip_ops as ops
    arena = torch.zeros(32 * 64)
    emb = arena.view(32, 64)
    emb.copy_(torch.randn((32, 64)))
    arena_version = arena._version
    query = {"sqn": (emb * emb).sum(1), "normalized": emb / emb.norm(dim=1, keepdim=True)}
    cache = ops.QueryOperandCache(emb, query)
    old_key = (emb.data_ptr(), tuple(emb.shape), emb._version, emb.device)
    assert cache.lookup(emb, torch.float32) is query
    assert cache.lookup(emb, torch.float16) is None                 # other operand type
    assert cache.lookup(emb.clone(), torch.float32) is None         # equal contents, other tensor
    del emb
    gc.collect()
    assert cache.ref() is None
    other = torch.randn((32, 64))
    ctypes.memmove(arena.data_ptr(), other.data_ptr(), 32 * 64 * 4)  # what a raw kernel does: new rows, no version bump
    fresh = arena.view(32, 64)                                       # 'the allocator handed out the same block again'
    assert arena._version == arena_version and torch.equal(fresh, other)
    assert (fresh.data_ptr(), tuple(fresh.shape), fresh._version, fresh.device) == old_key   # the round-5 key would have HIT
    assert cache.lookup(fresh, torch.float32) is None, "stale rows served for a recycled address"
    live = torch.randn((4, 8))
    c2 = ops.QueryOperandCache(live, {"sqn": None, "normalized": live})
    assert c2.lookup(live, torch.float32) is not None
    live.add_(1.0)                                                   # in-place edit of the live tensor: version moves on
    assert c2.lookup(live, torch.float32) is None
