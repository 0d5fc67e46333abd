"""Device-resident evaluation harness: the dataflow of the reference's ``test()``
(train_vidreid_xent_htri.py:450-542) with everything after the data loader kept on the GPU.

    reference test()                                   this harness
    ------------------------------------------------   ---------------------------------------------------------
    model(imgs, adj) per batch            (:469,:499)   same call (the HIP eval forward)
    dense/skipdense clip pooling          (:471-476)    ``pool_clips`` (mean / max over a tracklet's clips)
    features.data.cpu(); torch.cat        (:477-511)    embeddings stay in HBM; RCCL all-gather when sharded
    compute_distance_matrix on CPU        (:520)        ``agrl_distmat_topk`` against the rank's gallery shard: distance rows of
    evaluate_rank(use_metric_mars=True)   (:531)        one query block at a time + top-k; candidate merge + ``agrl_rank_mars``
    returns cmc[0], mAP                   (:542)        returns (cmc, mAP) (+ the top-k lists on request)

With ``torch.distributed`` initialised (one process per GPU) the tracklet batches and the gallery rows are sharded
over the ranks (torchreid/parallel.py); every rank returns the same (cmc, mAP).
"""
from __future__ import annotations

import numpy as np
import torch

from torchreid import hip_ops as ops
from torchreid import parallel
from torchreid.metrics.distance import hip_distmat_device, hip_distmat_topk_device


def pool_clips(features, num_clips, pool="avg"):
    """(tracklets*num_clips, D) -> (tracklets, D): mean or max over each tracklet's clips
    (reference: features.view(n, 1, -1) then mean/max over dim 0, train_vidreid_xent_htri.py:471-476)."""
    if num_clips == 1:
        return features
    if features.is_cuda:
        return ops.clip_pool(features, num_clips, "avg" if pool == "avg" else "max")
    f = features.view(-1, num_clips, features.size(-1))
    return f.mean(dim=1) if pool == "avg" else f.max(dim=1)[0]


def device_prefetch(batches, device):
    """Upload batch i+1 on a copy stream while batch i is being processed (the reference's test() uploads inside the
    loop, train_vidreid_xent_htri.py:460, and its loaders hand over pinned tensors, :222-247 ``pin_memory``): with pinned
    sources the 100 MB of fp32 frames per 256-frame batch cross PCIe under the previous batch's forward. Yields the same
    tuples with imgs / adj on ``device``; tensors already there pass through."""
    if device.type != "cuda":
        for item in batches:
            yield item
        return
    copy_stream = torch.cuda.Stream(device=device)
    main = torch.cuda.current_stream(device)

    def upload(item):
        imgs, pid, camid, adj = item
        with torch.cuda.stream(copy_stream):
            imgs_d, adj_d = imgs.to(device, non_blocking=True), adj.to(device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(copy_stream)
        return imgs_d, pid, camid, adj_d, done

    pending = None
    for item in batches:
        nxt = upload(item)
        if pending is not None:
            yield _claim(pending, main)
        pending = nxt
    if pending is not None:
        yield _claim(pending, main)


def _claim(pending, main):
    imgs_d, pid, camid, adj_d, done = pending
    main.wait_event(done)
    imgs_d.record_stream(main)
    adj_d.record_stream(main)
    return imgs_d, pid, camid, adj_d


@torch.no_grad()
def extract_features(model, batches, pool="avg", prefetch=True, local_only=False, sync_ranks=None):
    """``batches`` yields (imgs, pids, camids, adj) like the reference's loaders; imgs is (b,S,3,H,W) or, for the
    dense samplers, (b,n,S,3,H,W) with adj (b,n,V,V). Returns (features (N,D) on the model's device, pids, camids).

    Under an active process group (world > 1) the call is COLLECTIVE by default: every rank extracts its slice and then meets
    the others in ``match_and_rank``, so the non-finite flag is all-reduced (MAX) -- by every rank, whatever device its model
    is on -- and all of them raise together instead of one leaving the rest blocked in the next collective.
    ``local_only=True`` is the explicit opt-out for a call that only THIS rank makes (a rank-0-only ``evaluate``): no
    collective is issued and a non-finite embedding raises on this rank alone. (``sync_ranks``: the round-4 spelling,
    ``sync_ranks=False`` == ``local_only=True``.)"""
    device = next(model.parameters()).device
    model.eval()
    feats, pids, camids = [], [], []
    nonfinite = None
    if prefetch:
        batches = device_prefetch(batches, device)
    for imgs, pid, camid, adj in batches:
        imgs, adj = imgs.to(device, non_blocking=True), adj.to(device, non_blocking=True)
        clips = 1
        if imgs.dim() == 6:
            b, clips = imgs.shape[:2]
            imgs = imgs.reshape((b * clips,) + tuple(imgs.shape[2:]))
            adj = adj.reshape((b * clips,) + tuple(adj.shape[2:]))
        raw = model(imgs, adj)
        if raw.is_cuda:   # accumulated on the device, read once after the last batch: no per-batch synchronisation
            bad = ~torch.isfinite(raw).all()
            nonfinite = bad if nonfinite is None else (nonfinite | bad)
        feats.append(pool_clips(raw, clips, pool))
        pids.extend(np.asarray(pid).tolist())
        camids.extend(np.asarray(camid).tolist())
    out = torch.cat(feats, 0)
    # One check per extraction, on the RAW model outputs (before the dense samplers' clip pooling): the fp16 build stores
    # activations with a range of 65504 -- a checkpoint whose activations leave it yields inf / nan embeddings, and ranking those
    # would be silent garbage. (Nothing on this path comes near the limit with the recipe or with trained ResNet50 statistics.)
    if sync_ranks is not None:
        local_only = not sync_ranks
    sync = not local_only and parallel.world_size() > 1
    if nonfinite is not None or sync:
        flag = (nonfinite.to(torch.int32) if nonfinite is not None else torch.zeros((), dtype=torch.int32, device=out.device)).reshape(1)
        if sync:   # every rank takes part, also one whose features came from the CPU path (flag 0)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        if bool(flag.item()):
            from torchreid import _hip
            raise FloatingPointError(
                "non-finite embeddings from the %s forward (hip_precision=%r)%s" % (
                    _hip.LP_NAME if getattr(model, "hip_precision", "fp32") == _hip.LP_NAME else "HIP", getattr(model, "hip_precision", None),
                    ": activations (or BatchNorm-folded weights) left fp16's range -- set AGRL_HIP_LP16=bf16 (libagrl_hip_bf16.so) or "
                    "hip_precision='fp32'" if _hip.LP_NAME == "fp16" and getattr(model, "hip_precision", "fp32") == "fp16" else ""))
    return out, np.asarray(pids), np.asarray(camids)


def _i32(a, device):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(device)


@torch.no_grad()
def match_and_rank(qf, q_pids, q_camids, gf, g_pids, g_camids, dist_metric="cosine", max_rank=50,
                   precision="fp32", return_topk=False, re_rank=False, local_only=False):
    """Distance matrix + MARS ranking on the device. ``qf``: ALL query embeddings (m,D); ``gf``: this rank's gallery
    rows when a process group is active (rows ``shard_bounds(n, rank, world)`` of the gallery), else the whole
    gallery. ``g_pids``/``g_camids`` always describe the WHOLE gallery. Returns (cmc ndarray (max_rank,), mAP float).
    ``local_only=True``: ``gf`` is the WHOLE gallery and only this rank calls -- no rank offset, no candidate all-gather,
    whatever process group is initialised."""
    device = qf.device
    world = 1 if local_only else parallel.world_size()
    if local_only and gf.size(0) != len(g_pids):
        raise ValueError("local_only=True needs the whole gallery: {} rows for {} labels".format(gf.size(0), len(g_pids)))
    n = len(g_pids)
    if max_rank > n:
        raise ValueError("max_rank={} exceeds the gallery size {}".format(max_rank, n))
    lo = parallel.shard_bounds(n, torch.distributed.get_rank(), world)[0] if world > 1 else 0

    def dist_fn(q, g):
        return hip_distmat_device(q, g, dist_metric, precision)

    def topk_fn(d, k):
        return ops.rank_topk(d, k)

    if re_rank:
        # the reference's --re-rank branch (train_vidreid_xent_htri.py:523-527): three distance matrices, k-reciprocal
        # re-ranking, then the ranking of the re-ranked matrix. Needs the whole gallery on one device.
        if world > 1:
            raise NotImplementedError("re-ranking works on the full (m+n)^2 matrix: gather the gallery embeddings first")
        q32, g32 = qf.float().contiguous(), gf.float().contiguous()
        dist = ops.re_ranking(dist_fn(q32, g32), dist_fn(q32, q32), dist_fn(g32, g32))
        idx, val = topk_fn(dist, max_rank)
        idx = idx.to(torch.int64)
    else:
        # distance + top-k fused (agrl_distmat_topk): the (m, n) matrix of the reference's test() is never materialised
        def match_fn(q, g, k):
            return hip_distmat_topk_device(q, g, dist_metric, k, precision)

        idx, val = parallel.sharded_topk(qf.float().contiguous(), gf.float().contiguous(), lo, max_rank, dist_fn, topk_fn,
                                         match_fn=match_fn if qf.is_cuda else None, local_only=local_only)
    ap, cmc = ops.rank_mars(idx.to(torch.int32).contiguous(), _i32(q_pids, device), _i32(q_camids, device),
                            _i32(g_pids, device), _i32(g_camids, device))
    ap = ap.cpu().numpy()
    if np.isnan(ap).any():
        raise ZeroDivisionError("query {} has no cross-camera match in the gallery".format(int(np.isnan(ap).argmax())))
    out = (np.mean(cmc.cpu().numpy().astype(np.float64), axis=0), float(np.mean(ap)))
    if return_topk:
        return out + (idx.cpu().numpy(), val.cpu().numpy())
    return out


@torch.no_grad()
def evaluate(model, query_batches, gallery_batches, dist_metric="cosine", pool="avg", max_rank=50,
             ranks=(1, 5, 10, 20), verbose=False, re_rank=False):
    """Single-process form of the reference's ``test()``: returns (rank1, mAP) and, like the reference, can print the
    CMC table. This process extracts EVERY batch itself, so the gallery it ranks against is whole: all three stages run with
    ``local_only=True`` -- no collective anywhere, no rank offset -- and the call is safe on one rank of an initialised process
    group (``tests/test_parallel_gloo.py::test_rank0_only_evaluate``). For the sharded form every rank runs ``extract_features``
    on its slice and then ``match_and_rank`` (both collective by default)."""
    qf, q_pids, q_camids = extract_features(model, query_batches, pool, local_only=True)
    gf, g_pids, g_camids = extract_features(model, gallery_batches, pool, local_only=True)
    cmc, mAP = match_and_rank(qf, q_pids, q_camids, gf, g_pids, g_camids, dist_metric, max_rank,
                              getattr(model, "hip_precision", "fp32"), re_rank=re_rank, local_only=True)
    if verbose:
        print("Results ----------")
        print("mAP: {:.2%}".format(mAP))
        print("CMC curve")
        for r in ranks:
            print("Rank-{:<3}: {:.2%}".format(r, cmc[r - 1]))
        print("------------------")
    return cmc[0], mAP
