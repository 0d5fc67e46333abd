"""One-process-per-GPU data parallelism for the forward-and-match path.

The reference's only parallelism is single-process ``nn.DataParallel`` (train_vidreid_xent_htri.py:318), whose
scatter/gather funnels everything through GPU 0. On an 8-GPU xGMI node this build shards instead:

  * tracklets : each rank runs the eval forward on its slice of the batch (weights replicated, no traffic)
  * exchange  : ONE collective per batch -- an RCCL all-gather of the (b_local, 4096) embeddings
  * gallery   : each rank keeps a contiguous shard of gallery rows resident and computes the distance columns
                of that shard for ALL queries; a per-shard top-k plus a second small all-gather of the
                (index, distance) candidates gives every rank the global top-k for the ranking step

``torch.distributed`` (backend "nccl" == RCCL on ROCm, "gloo" in the CPU tests) only moves bytes; the arithmetic
is the HIP kernels. The per-rank compute is passed in as callables so the CPU tests can exercise the
sharding/merge logic under gloo without a GPU.
"""
from __future__ import annotations

import contextlib
import os

import torch
import torch.distributed as dist


def forced_group():
    """AGRL_DIST_FORCE_GROUP=1: build the process group and run every collective of this module even at world size 1 -- the way
    to put the RCCL branch (device-bound communicator, all_gather_into_tensor on device tensors, the candidate merge) under a
    real communicator on a 1-GPU box (tests/test_gpu_configs.py); never set in production."""
    return os.environ.get("AGRL_DIST_FORCE_GROUP", "0") == "1"


def collectives_active():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced_group())


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment; returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or forced_group()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # AGRL_DIST_BACKEND=gloo lets several ranks share ONE GPU (control-flow tests of the N > 1 path on a 1-GPU box)
            backend = os.environ.get("AGRL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            # bind this process to its GPU BEFORE the communicator exists: RCCL then builds its rings on the right
            # device and barrier()/collectives never have to guess one
            device = torch.device("cuda", local_rank)
            torch.cuda.set_device(device)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_bounds(n, rank, world):
    """Contiguous, balanced split of ``n`` items: the first ``n % world`` ranks get one extra. -> (lo, hi)"""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local):
    """All-gather of equally-sized row blocks: (b, D) on every rank -> (b*world, D), rank-major."""
    world = world_size()
    if not collectives_active():
        return local
    out = torch.empty((local.size(0) * world,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def all_gather_ragged_rows(local, counts):
    """All-gather of row blocks with per-rank row counts ``counts`` (list, same on every rank)."""
    world = world_size()
    if not collectives_active():
        return local
    cap = max(counts)
    padded = local.new_zeros((cap,) + tuple(local.shape[1:]))
    padded[: local.size(0)] = local
    gathered = all_gather_rows(padded).view((world, cap) + tuple(local.shape[1:]))
    return torch.cat([gathered[r, : counts[r]] for r in range(world)], dim=0)


def sharded_topk(q_all, gallery_shard, shard_lo, k, distmat_fn, topk_fn, match_fn=None, local_only=False):
    """Global top-k of every query against a gallery sharded by rows across ranks.

    ``distmat_fn(q, g) -> (m, n_local)`` and ``topk_fn(d, k) -> (idx (m,k') int, val (m,k'))`` (ascending
    (distance, index), ties towards the lower index) are the per-rank kernels; ``match_fn(q, g, k) -> (idx, val)``, when
    given, replaces the pair for the local step (the fused distance + top-k that never writes the (m, n_local) matrix).
    Returns (idx (m,k) global gallery indices, val (m,k)), identical on every rank and identical to a single-GPU top-k of
    the full matrix: shards are contiguous and rank-ordered, so 'position in the concatenated candidate list' orders ties
    exactly like 'global gallery index'. ``local_only=True``: ``gallery_shard`` is the whole gallery and only this rank calls --
    the local step's result is returned as it is, no collective."""
    k_local = min(k, gallery_shard.size(0))
    if match_fn is not None:
        idx, val = match_fn(q_all, gallery_shard, k_local)
    else:
        idx, val = topk_fn(distmat_fn(q_all, gallery_shard), k_local)
    idx = idx.to(torch.int64) + shard_lo
    world = world_size()
    if local_only or not collectives_active():
        return idx, val
    if k_local < k:  # a shard smaller than k: pad with +inf candidates
        pad = k - k_local
        idx = torch.cat([idx, idx.new_full((idx.size(0), pad), -1)], dim=1)
        val = torch.cat([val, val.new_full((val.size(0), pad), float("inf"))], dim=1)
    m = idx.size(0)
    # (world, m, k) -> (m, world*k), rank-major inside each query row
    cand_idx = all_gather_rows(idx.view(1, m, k)).permute(1, 0, 2).reshape(m, world * k).contiguous()
    cand_val = all_gather_rows(val.view(1, m, k)).permute(1, 0, 2).reshape(m, world * k).contiguous()
    pos, best = topk_fn(cand_val, k)
    return torch.gather(cand_idx, 1, pos.to(torch.int64)), best


# ---- train step (BASELINE config 4): one process per GPU instead of the reference's nn.DataParallel -----------------
class _GatherRows(torch.autograd.Function):
    """All-gather of equally sized row blocks that autograd can see. Every rank goes on to compute the SAME global loss
    from the gathered tensor, so the gradient of that loss w.r.t. this rank's rows is simply its slice of the incoming
    gradient (nn.DataParallel's gather-to-GPU-0 / scatter-back, train_vidreid_xent_htri.py:318, :399-411, without GPU 0)."""

    @staticmethod
    def forward(ctx, local):
        ctx.rows = local.size(0)
        return all_gather_rows(local)

    @staticmethod
    def backward(ctx, grad):
        r = dist.get_rank() if world_size() > 1 else 0
        return grad[r * ctx.rows:(r + 1) * ctx.rows].contiguous()


def gather_rows_with_grad(local):
    return _GatherRows.apply(local) if world_size() > 1 else local


class GradientBuckets(object):
    """Gradient all-reduce overlapped with backward (the reduce-add of nn.DataParallel's replica gradients,
    train_vidreid_xent_htri.py:318, :411, as RCCL all-reduces over xGMI).

    The parameters' ``.grad`` tensors are VIEWS into a few flat buffers (filled in reverse registration order, the order
    backward produces them in), so a bucket needs no gather copy before and no scatter copy after its collective. A
    post-accumulate hook per parameter counts a bucket down; the moment its last gradient has been accumulated the
    bucket's all-reduce is issued asynchronously -- it runs on RCCL's stream under the rest of backward. ``finish()``
    waits for the handles. Bucket size: xGMI is point-to-point (7 links x ~153 GB/s per GPU), an all-reduce of
    S bytes moves 2 (N-1)/N S per GPU spread over the links, so 64 MB buckets cost ~0.15 ms each at N = 8 -- large enough
    to amortise the ~20 us launch latency of a collective, small enough that the first one starts while layer3's
    backward is still running (vmgn: 188 MB of fp32 gradients -> 3 buckets)."""

    def __init__(self, parameters, bucket_bytes=64 << 20):
        self.params = [p for p in parameters if p.requires_grad]
        self.buckets = []        # (flat buffer, [params])
        self._bucket_of = {}
        order = list(reversed(self.params))
        cur, size = [], 0
        groups = []
        for p in order:
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                groups.append(cur)
                cur, size = [], 0
        if cur:
            groups.append(cur)
        for group in groups:
            flat = torch.zeros(sum(p.numel() for p in group), dtype=group[0].dtype, device=group[0].device)
            off = 0
            for p in group:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
                self._bucket_of[p] = len(self.buckets)
            self.buckets.append((flat, group))
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._hit = set()
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.zero_grad()

    def zero_grad(self):
        """Instead of optimizer.zero_grad(): keeps the .grad views alive (set_to_none would detach them from the buffers)."""
        for i, (flat, group) in enumerate(self.buckets):
            flat.zero_()
            self._pending[i] = len(group)
        self._handles = []
        self._hit = set()

    def _on_grad(self, p):
        i = self._bucket_of[p]
        self._hit.add(p)
        self._pending[i] -= 1
        if self._pending[i] == 0 and world_size() > 1:
            flat = self.buckets[i][0]
            self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """Wait for the collectives issued during backward; buckets whose parameters received no gradient this step (a
        frozen head, an unused branch) are reduced here so every rank issues the same sequence."""
        if world_size() > 1:
            for i, (flat, _) in enumerate(self.buckets):
                if self._pending[i] > 0:
                    self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
                    self._pending[i] = 0
        for h in self._handles:
            h.wait()
        self._handles = []

    @contextlib.contextmanager
    def only_touched(self):
        """Around optimizer.step(): parameters that received NO gradient in this backward (the classifiers of an
        htri-only step, an unused branch) show ``.grad is None`` -- as they do after optimizer.zero_grad() on the plain
        path -- so the optimiser skips them instead of applying weight decay / moment updates on a zero gradient. The
        views into the flat buffers come back afterwards. Every rank runs the same graph, so every rank skips the same."""
        parked = [(p, p.grad) for p in self.params if p not in self._hit]
        for p, _ in parked:
            p.grad = None
        try:
            yield
        finally:
            for p, g in parked:
                p.grad = g

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def allreduce_gradients(parameters, bucket_bytes=64 << 20):
    """SUM the parameter gradients over the ranks (DataParallel's reduce-add of replica gradients) in flat buckets: a few
    large RCCL all-reduces over xGMI instead of one per tensor (the 188 MB of fp32 gradients of vmgn go in 3 buckets)."""
    if world_size() == 1:
        return
    grads = [p.grad for p in parameters if p.grad is not None]
    bucket, size = [], 0

    def flush():
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        off = 0
        for g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    for g in grads:
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
            bucket, size = [], 0
    flush()


def train_step(model, imgs, adj, pids, criterion_xent, criterion_htri, optimizer, htri_only=False, buckets=None):
    """One xent + htri step of the reference's train() (train_vidreid_xent_htri.py:397-413) on THIS rank's shard of the
    batch: local forward (BatchNorm statistics per replica, as under DataParallel), logits / features / labels gathered
    over the ranks, losses and the batch-hard mining (native kernel) on the GLOBAL batch, backward, gradient all-reduce,
    optimizer step. Returns the (global) loss values. With one rank it is exactly the reference's step.
    ``buckets``: a GradientBuckets over model.parameters() -> the all-reduces overlap backward; without it the gradients
    are reduced in flat buckets after backward."""
    from torchreid.losses import DeepSupervision
    model.train()
    outputs, features = model(imgs, adj)
    outputs = [gather_rows_with_grad(o) for o in (outputs if isinstance(outputs, (list, tuple)) else [outputs])]
    features = [gather_rows_with_grad(f) for f in (features if isinstance(features, (list, tuple)) else [features])]
    pids_all = all_gather_rows(pids.view(-1, 1)).view(-1)
    xent = DeepSupervision(criterion_xent, outputs, pids_all)
    htri = DeepSupervision(criterion_htri, features, pids_all)
    loss = htri if htri_only else xent + htri
    if buckets is not None:
        buckets.zero_grad()
        loss.backward()
        buckets.finish()
        with buckets.only_touched():
            optimizer.step()
    else:
        optimizer.zero_grad()
        loss.backward()
        allreduce_gradients(list(model.parameters()))
        optimizer.step()
    return float(loss.detach()), float(xent.detach()), float(htri.detach())
