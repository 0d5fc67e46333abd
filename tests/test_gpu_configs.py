"""BASELINE.json parity-test configurations at their full sizes (the ones that are not bench lines):
config 5 -- MARS full eval, 1980 x 12180 x 4096 distance matrix + top-50 ranking -- through size-independent properties
plus sampled rows against the CPU oracle; config 4 -- a Duke-shaped (seq_len 16, V = 112) xent + htri train step with
on-GPU batch-hard mining -- against the same step computed on the CPU by the same module tree."""
import os

import numpy as np
import pytest
import torch

from lp16 import LP16, LP_DTYPE

from oracle import vmgn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("precision", ["fp32", LP16])
def test_config5_full_mars_distmat_and_rank(precision):
    from torchreid import hip_ops as ops
    from torchreid.metrics.distance import hip_distmat_device
    m, n, D, k = 1980, 12180, 4096, 50
    g = torch.Generator().manual_seed(55)
    centers = torch.randn((625, D), generator=g)
    g_pid = torch.randint(0, 625, (n,), generator=g)
    q_pid = torch.randint(0, 625, (m,), generator=g)
    gal = centers[g_pid] + 0.7 * torch.randn((n, D), generator=g)
    qry = centers[q_pid] + 0.7 * torch.randn((m, D), generator=g)
    qd, gd = qry.to(DEV), gal.to(DEV)
    for metric, fn in (("cosine", O.cosine), ("euclidean", O.euclidean_squared)):
        dist = hip_distmat_device(qd, gd, metric, precision)
        assert dist.shape == (m, n) and torch.isfinite(dist).all()
        rows = torch.randperm(m, generator=g)[:48]
        ref = fn(qry[rows].double(), gal.double())
        err = ((dist[rows.to(DEV)].double().cpu() - ref).abs().max() / ref.abs().max()).item()
        assert err < (1e-5 if precision == "fp32" else 1e-2), (metric, err)
        # ranking of the device matrix: exact top-k, ascending, ties towards the lower index, idempotent
        idx, val = ops.rank_topk(dist, k)
        tv, ti = torch.topk(dist, k, dim=1, largest=False, sorted=True)
        assert torch.equal(val, tv)
        assert bool((val[:, 1:] >= val[:, :-1]).all())
        assert torch.equal(torch.gather(dist, 1, idx.long()), val)
        distinct = (tv[:, 1:] != tv[:, :-1]).all(dim=1)            # rows without exact ties: indices must be identical
        assert torch.equal(idx.long()[distinct], ti[distinct])
        from torchreid.metrics.distance import hip_distmat_topk_device
        idx_f, val_f = hip_distmat_topk_device(qd, gd, metric, k, precision)    # never writes the 96.5 MB matrix
        assert torch.equal(idx_f, idx) and torch.equal(val_f, val)
        idx2, val2 = ops.rank_topk(val.contiguous(), k)
        assert torch.equal(val2, val) and torch.equal(idx2.long(), torch.arange(k, device=DEV).expand(m, k))
        # a query's own identity dominates its top ranks (sanity of the whole match)
        hit = (g_pid.to(DEV)[idx[:, 0].long()] == q_pid.to(DEV)).float().mean().item()
        assert hit > 0.9, hit
        # column-sharded computation (what each rank of an 8-GPU run holds) equals the full matrix
        lo, hi = 3 * (n // 8), 4 * (n // 8)
        shard = hip_distmat_device(qd, gd[lo:hi].contiguous(), metric, precision)
        tol = 0 if precision == "fp32" else 2e-2
        assert (shard - dist[:, lo:hi]).abs().max().item() <= tol * dist.abs().max().item() + (1e-5 if precision == "fp32" else 0)


def test_config4_train_step_seq16_matches_cpu_step():
    """One xent + htri step at seq_len 16 (V = 112, consistent loss on): GPU (native mining) vs the CPU module tree."""
    from recipe import recipe_state_dict, synthetic_adj, synthetic_clips
    from torchreid import losses, models
    S, P, K, ncls = 16, 2, 2, 5
    kw = dict(num_classes=ncls, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2, num_scale=1, pyramid_part=True,
              use_pose=True, learn_graph=True, consistent_loss=True)
    ref = models.init_model("vmgn", **kw)
    sd = recipe_state_dict(ref.state_dict(), seed=3)
    ref.load_state_dict(sd)
    dev = models.init_model("vmgn", **kw)
    dev.load_state_dict(sd)
    dev = dev.to(DEV)
    pids = torch.arange(P).repeat_interleave(K)
    x = synthetic_clips(P * K, S, H=128, W=64, seed=9, identities=pids.tolist())  # half-size frames: CPU time
    adj = synthetic_adj(P * K, S, seed=9)
    xent = losses.CrossEntropyLabelSmooth(num_classes=ncls, use_gpu=False)
    xent_d = losses.CrossEntropyLabelSmooth(num_classes=ncls, use_gpu=True)
    htri = losses.TripletLoss(margin=0.3, soft=True)

    def step(model, x_, adj_, y_, ce):
        model.train()
        torch.manual_seed(1234)  # the consistent loss draws its frame subsets with torch.randperm on the host
        outs, feats = model(x_, adj_)
        loss = losses.DeepSupervision(ce, outs, y_) + losses.DeepSupervision(htri, feats, y_)
        loss.backward()
        gn = torch.sqrt(sum((p.grad.detach().double() ** 2).sum() for p in model.parameters() if p.grad is not None))
        return loss.item(), gn.item(), len(outs), len(feats)

    l_ref, g_ref, no, nf = step(ref, x, adj, pids, xent)
    l_dev, g_dev, no2, nf2 = step(dev, x.to(DEV), adj.to(DEV), pids.to(DEV), xent_d)
    assert (no, nf) == (no2, nf2) == (5, 5)
    print("train step S=16: loss cpu %.6f gpu %.6f | grad norm cpu %.4e gpu %.4e" % (l_ref, l_dev, g_ref, g_dev))
    assert abs(l_ref - l_dev) < 2e-3 * abs(l_ref)
    assert abs(g_ref - g_dev) < 2e-2 * g_ref


def test_bench_self_launches_two_ranks():
    """BASELINE configs[2]'s launch path on a 1-GPU box: ``python bench.py --gpus 2`` itself starts two fresh rank
    processes (the parent never touches the GPU), which share the one GPU over gloo here (RCCL needs one GPU per rank), run
    the all-gather + sharded-gallery step, and rank 0 prints the JSON line with n_gpus = 2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--batch", "8", "--sustain-seconds", "0", "--profile-steps", "1"], env=env, cwd=root,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["config"]["ranks"] == 2 and r["config"]["global_batch"] == 16
    assert r["config"]["gallery_rows_per_gpu"] == 6090 and len(r["config"]["per_rank_ms_per_step"]) == 2
    assert r["value"] > 0 and "allgather_us" in r["config"] and r["scaling"] == "weak"
    print("bench --gpus 2 (2 ranks on one GPU, gloo): %.0f frames/s, all-gather %.0f us" % (r["value"], r["config"]["allgather_us"]))


def test_rccl_path_on_one_gpu():
    """The RCCL branch under a REAL communicator before the first multi-GPU run: a fresh child (never a re-exec of a process that
    touched the GPU) runs ``bench.py --gpus 1`` with AGRL_DIST_BACKEND=nccl and AGRL_DIST_FORCE_GROUP=1 -- world size 1, but
    parallel.init_from_env binds the device and builds the communicator (device_id), every step's embeddings go through
    all_gather_into_tensor on device tensors on the match stream (side-stream ordering against the next batch's forward), the barrier
    / max-over-ranks timing collectives run, and parallel.sharded_topk takes its candidate all-gather + merge path. The line must say
    rccl and the sharded top-50 must equal the single-process one."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(AGRL_DIST_BACKEND="nccl", AGRL_DIST_FORCE_GROUP="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29631", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "8",
                          "--sustain-seconds", "0", "--profile-steps", "1", "--no-modes", "--no-accuracy", "--no-config5", "--no-config4",
                          "--no-cpu-baseline"], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    r = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1])
    cfg = r["config"]
    assert r["n_gpus"] == 1 and cfg["collective_backend"] == "rccl" and cfg["ranks"] == 1
    assert "allgather_us" in cfg and cfg["gallery_rows_all_ranks"] == [12180]
    assert cfg["sharded_top50_equal_up_to_near_ties"] and cfg["sharded_top50_swaps_not_explained_by_a_near_tie"] == 0
    assert isinstance(cfg["sharded_top50_equals_single_process"], bool)    # the strict reading (index lists equal) is reported beside it
    print("bench --gpus 1 over a real RCCL communicator: %.0f frames/s, all-gather %.0f us" % (r["value"], cfg["allgather_us"]))


def test_bench_config3_eight_ranks_on_one_gpu():
    """BASELINE configs[2] at its real shape without the hardware: ``python bench.py --gpus 8 --batch 32`` = 8 ranks x 32
    tracklets (256 global), the 12 180-row gallery in 1 523 / 1 522-row shards, all-gather of embeddings, per-shard distance --
    the 8 processes share the one GPU over gloo (RCCL needs a GPU per rank; the transport is all that differs). Rank 0's line
    must report 8 ranks, the shard sizes, and a sharded top-50 (per-shard top-k + candidate merge) equal to a single-process
    top-50 over the whole gallery."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                          "--batch", "32", "--sustain-seconds", "0", "--profile-steps", "1", "--dist-timeout", "1200"], env=env, cwd=root,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    line = [l for l in out.stdout.decode().splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    cfg = r["config"]
    assert r["n_gpus"] == 8 and cfg["ranks"] == 8 and cfg["global_batch"] == 256 and cfg["frames_per_step"] == 2048
    assert cfg["gallery_rows_all_ranks"] == [1523] * 4 + [1522] * 4 and cfg["gallery_rows_per_gpu"] == 1523
    assert len(cfg["per_rank_ms_per_step"]) == 8 and r["scaling"] == "weak" and r["value"] > 0
    assert cfg["sharded_top50_equal_up_to_near_ties"] is True, cfg.get("sharded_top50_max_abs_diff")
    print("sharded top-50: strictly equal %s, swapped positions %d, max |d distance| %.3g (near-tie window %.3g)" % (
        cfg["sharded_top50_equals_single_process"], cfg["sharded_top50_swapped_positions"], cfg["sharded_top50_max_abs_diff"],
        cfg["sharded_top50_near_tie_tolerance"]))
    assert cfg["allgather_bytes_per_rank"] == 32 * 4096 * 4
    print("bench --gpus 8 --batch 32 (8 ranks on one GPU, gloo): %.0f frames/s, all-gather %.0f us, top-50 merge equal" % (
        r["value"], cfg["allgather_us"]))


def test_bench_launcher_notices_a_rank_that_dies_mid_run():
    """``python bench.py --gpus 2`` with rank 1 exiting (code 3) right after the warm-up: rank 0 is then blocked in the
    all-gather; the launcher must terminate it and return non-zero well inside --dist-timeout."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(AGRL_BENCH_FAULT_RANK="1", AGRL_BENCH_FAULT_CODE="3")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8",
                          "--sustain-seconds", "0", "--profile-steps", "1", "--dist-timeout", "600"], env=env, cwd=root,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    dt = time.time() - t0
    err = out.stderr.decode()
    print("launcher returned %d after %.0f s" % (out.returncode, dt))
    # (rank 0 may notice the closed connection itself and exit non-zero before the next poll: either rank can be the one named)
    assert out.returncode != 0, err[-2000:]
    assert "exited with code" in err and "remaining ranks terminated" in err and dt < 500
