"""One rank of the two-ranks-on-one-GPU tests of tests/test_gpu_dist.py (gloo transports the collectives: RCCL needs a GPU per
rank; everything else -- the native HIP train step under ``GradientBuckets``, the device evaluation harness -- is what an N-GPU
run executes). Started as ``python tests/dist_gpu_worker.py <case> <out_dir>`` with RANK / WORLD_SIZE / MASTER_* in the
environment; a fresh interpreter per rank, so no process that has touched the GPU is ever forked or re-exec'd."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

DEV = "cuda:0"


def train_problem():
    """BASELINE config 4 in small: P x K = 2 x 2 tracklets of 6 frames, 64 x 32 pixels, consistent loss on (the reference's
    train(), train_vidreid_xent_htri.py:397-413)."""
    from recipe import recipe_state_dict, synthetic_adj, synthetic_clips
    from torchreid import models
    kw = dict(num_classes=5, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1, pyramid_part=True,
              use_pose=True, learn_graph=True, consistent_loss=True)
    m = models.init_model("vmgn", **kw)
    m.load_state_dict(recipe_state_dict(m.state_dict(), seed=3))
    pids = torch.tensor([0, 1, 0, 1])        # every rank's shard holds both identities (mining is global either way)
    x = synthetic_clips(4, 6, H=64, W=32, seed=9, identities=pids.tolist())
    adj = synthetic_adj(4, 6, seed=9)
    return m.to(DEV), x.to(DEV), adj.to(DEV), pids.to(DEV)


def case_train(rank, world, out_dir):
    """parallel.train_step on this rank's half of the batch: native forward (per-replica BatchNorm statistics), gathered logits /
    features, global mining, native backward with the bucketed all-reduce issued from the post-accumulate hooks."""
    from torchreid import losses, parallel
    m, x, adj, pids = train_problem()
    assert m.hip_train
    lo, hi = parallel.shard_bounds(x.size(0), rank, world)
    buckets = parallel.GradientBuckets(m.parameters(), bucket_bytes=8 << 20)
    assert len(buckets.buckets) > 3
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    from torchreid import _hip
    _hip.PROFILE = []
    for _ in range(2):       # twice: the views and the hook counters must survive a step
        torch.manual_seed(1234)   # the consistent loss draws its frame subsets on the host: every rank the same
        loss = parallel.train_step(m, x[lo:hi], adj[lo:hi], pids[lo:hi], losses.CrossEntropyLabelSmooth(5, use_gpu=True),
                                   losses.TripletLoss(margin=0.3, soft=True), opt, buckets=buckets)
    names = sorted({r[0] for r in _hip.PROFILE})
    _hip.PROFILE = None
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None and p in buckets._hit}
    for k, p in m.named_parameters():
        if p.grad is not None:
            assert p.grad.is_cuda and p.grad.data_ptr() >= buckets.buckets[buckets._bucket_of[p]][0].data_ptr()
    buckets.remove()
    torch.save({"loss": loss, "grads": grads, "entry_points": names}, os.path.join(out_dir, "train_r%d.pt" % rank))


def case_evaluate(rank, world, out_dir):
    """evaluation.evaluate called by rank 0 ALONE under an initialised 2-rank group (round-5 advice): no stage of it may issue
    a collective or offset the gallery by the rank. Rank 1 meanwhile waits in a barrier that rank 0 joins only afterwards --
    a collective inside evaluate() would pair with that barrier (a mismatch error) or block until the test's timeout."""
    from recipe import recipe_state_dict
    from torchreid import evaluation, models
    if rank == 0:
        m = models.init_model("vmgn", num_classes=10, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1,
                              pyramid_part=True, use_pose=True, learn_graph=True)
        m.load_state_dict(recipe_state_dict(m.state_dict(), seed=5))
        m = m.to(DEV).eval()
        m.hip_precision = "fp32"
        S = 4

        def loader(n, seed):
            from recipe import synthetic_adj, synthetic_clips
            pids = np.arange(n) % 5
            cams = (np.arange(n) // 5 + seed) % 3
            x = synthetic_clips(n, S, H=64, W=32, seed=seed, identities=pids.tolist())
            adj = synthetic_adj(n, S, seed=seed)
            return [(x[i:i + 8], pids[i:i + 8], cams[i:i + 8], adj[i:i + 8]) for i in range(0, n, 8)], pids, cams

        qb, q_pids, q_cams = loader(10, 0)
        gb, g_pids, g_cams = loader(60, 1)
        r1, mAP = evaluation.evaluate(m, qb, gb, "cosine")
        # the same through the single-process building blocks, explicitly local
        qf, _, _ = evaluation.extract_features(m, qb, local_only=True)
        gf, _, _ = evaluation.extract_features(m, gb, local_only=True)
        cmc, mAP2, idx, _ = evaluation.match_and_rank(qf, q_pids, q_cams, gf, g_pids, g_cams, "cosine", 50, "fp32",
                                                      return_topk=True, local_only=True)
        assert idx.min() >= 0 and idx.max() < 60 and idx.max() >= 30, "gallery indices must be global, not offset by a rank"
        torch.save({"rank1": float(r1), "mAP": float(mAP), "rank1_b": float(cmc[0]), "mAP_b": float(mAP2)},
                   os.path.join(out_dir, "eval_r0.pt"))
    dist.barrier()


def main():
    case, out_dir = sys.argv[1], sys.argv[2]
    from torchreid import parallel
    rank, world, _ = parallel.init_from_env("gloo")
    assert world == 2 and parallel.collectives_active()
    {"train": case_train, "evaluate": case_evaluate}[case](rank, world, out_dir)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
