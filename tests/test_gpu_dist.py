"""Two ranks sharing the one GPU of a box (gloo carries the collectives; RCCL needs a GPU per rank): what an N-GPU run executes
besides the transport -- the NATIVE train step of BASELINE config 4 under ``GradientBuckets`` (until round 6 only CPU tensors had
run through it, tests/test_parallel_gloo.py) and a rank-0-only ``evaluation.evaluate`` under an initialised group."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))


def _run_two_ranks(case, out_dir, timeout=900):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   AGRL_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_gpu_worker.py"), case, str(out_dir)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0].decode())
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d exited with %s:\n%s" % (rank, p.returncode, o[-3000:])


@pytest.mark.timeout(1200)
def test_native_train_step_under_gradient_buckets_two_ranks(tmp_path):
    """2 ranks x 2 tracklets, native HIP forward / backward per rank, logits / features gathered, global xent + batch-hard triplet,
    bucketed all-reduce overlapped with backward == ONE process that runs the two replicas' native forwards itself, concatenates
    and backpropagates (nn.DataParallel's arithmetic, train_vidreid_xent_htri.py:318, :399-411). Both sides run the same kernels on
    the same shards, so the bar is tight: loss 1e-6, every gradient tensor 1e-4 of its largest entry."""
    from dist_gpu_worker import train_problem
    from torchreid import losses
    _run_two_ranks("train", tmp_path)
    m, x, adj, pids = train_problem()
    m.train()
    outs, feats = [], []
    for r in range(2):
        torch.manual_seed(1234)
        o, f = m(x[2 * r:2 * r + 2], adj[2 * r:2 * r + 2])
        outs.append(o)
        feats.append(f)
    outs = [torch.cat([outs[0][i], outs[1][i]]) for i in range(len(outs[0]))]
    feats = [torch.cat([feats[0][i], feats[1][i]]) for i in range(len(feats[0]))]
    loss = losses.DeepSupervision(losses.CrossEntropyLabelSmooth(5, use_gpu=True), outs, pids) + \
        losses.DeepSupervision(losses.TripletLoss(margin=0.3, soft=True), feats, pids)
    m.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    ref = {k: p.grad.detach().cpu() for k, p in m.named_parameters() if p.grad is not None}
    got = [torch.load(os.path.join(str(tmp_path), "train_r%d.pt" % r)) for r in range(2)]
    assert {"agrl_conv2d_bn_act", "agrl_bn_backward", "agrl_conv_wgrad", "agrl_triplet_loss"} <= set(got[0]["entry_points"])
    worst = 0.0
    for r in range(2):
        l = got[r]["loss"][0]
        assert abs(l - loss.item()) < 1e-6 * abs(loss.item()), (l, loss.item())
        assert set(got[r]["grads"]) == set(ref)
        for k in ref:
            e = ((got[r]["grads"][k] - ref[k]).abs().max() / ref[k].abs().max().clamp(min=1e-20)).item()
            worst = max(worst, e)
    for k in ref:   # every rank holds the SAME reduced gradient
        assert torch.equal(got[0]["grads"][k], got[1]["grads"][k]), k
    print("native train step, 2 ranks on one GPU under GradientBuckets: loss %.7f (single process %.7f), worst gradient tensor error %.2e over %d tensors"
          % (got[0]["loss"][0], loss.item(), worst, len(ref)))
    assert worst < 1e-4, worst


@pytest.mark.timeout(900)
def test_rank0_only_evaluate_under_a_two_rank_group(tmp_path):
    """evaluation.evaluate from rank 0 alone while rank 1 sits in a barrier: completes (no hidden collective), ranks against the
    WHOLE gallery with global indices, and equals the explicit local_only building blocks."""
    _run_two_ranks("evaluate", tmp_path, timeout=600)
    r = torch.load(os.path.join(str(tmp_path), "eval_r0.pt"))
    assert r["rank1"] == r["rank1_b"] and abs(r["mAP"] - r["mAP_b"]) < 1e-12 and 0.0 <= r["mAP"] <= 1.0
    print("rank-0-only evaluate under a 2-rank group: Rank-1 %.3f mAP %.4f" % (r["rank1"], r["mAP"]))
