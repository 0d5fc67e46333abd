#!/usr/bin/env python
"""Build-container script: run the CPU oracle over the WHOLE synthetic MARS split of tests/fullsplit.py (1 980 + 12 180
tracklets x 8 frames of 256 x 128 = 113 280 frames; ~1.5 h on 6 cores) and commit what the GPU box compares against:

    tests/golden/fullsplit_oracle.npz
        cal_{g,a}_{mean,var}   BNNeck calibration (statistics of the oracle's pre-BN features of gallery rows 0..2047)
        <metric>_{cmc,mAP}     MARS CMC (50,) / mAP of the oracle pipeline           (metric in cosine, euclidean)
        <metric>_idx           (1980, 51) int16   stable ascending order, first 51 gallery indices per query
        <metric>_val           (1980, 51) float32 the oracle's distances at those positions
        q_emb_head             (16, 4096) float32 the first 16 query embeddings (embedding-level check)

    python tests/golden/make_fullsplit.py [--threads 6] [--parts /tmp/fullsplit_parts] [--limit N]

Pre-BN features are checkpointed every chunk under --parts, so an interrupted run resumes. Reference lines restated by
the oracle functions it calls: models/vmgn.py:280-321, metrics/distance.py:59-89, metrics/rank.py:160-212.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "agrl.pytorch_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import fullsplit as FS  # noqa: E402
from oracle import vmgn_oracle as O  # noqa: E402
from recipe import recipe_state_dict  # noqa: E402

CHUNK = 32


def oracle_adj(poses, detected):
    poses, detected = poses.numpy(), detected.numpy()
    out = []
    for b in range(poses.shape[0]):
        sets = [O.pose_part_sets(poses[b, s] if detected[b, s] else None, float(FS.HEIGHT), 4, 0.1) for s in range(poses.shape[1])]
        out.append(O.pose_adjacency(sets, 4, True))
    return torch.from_numpy(np.stack(out))


def features(sd, pids, cams, first, parts_dir, tag, limit):
    """Pre-BN (g_f, att_f) of every tracklet of one set, chunk files under parts_dir."""
    n = len(pids) if limit is None else min(limit, len(pids))
    g_all, a_all = [], []
    t0 = time.time()
    for i, (x, _, _, adj) in enumerate(FS.batches(pids[:n], cams[:n], first, "cpu", CHUNK, oracle_adj)):
        path = os.path.join(parts_dir, "%s_%05d.npz" % (tag, i))
        if os.path.exists(path):
            z = np.load(path)
            g_all.append(z["g"]), a_all.append(z["a"])
            continue
        B = x.shape[0]
        with torch.no_grad():
            x4_1, x4_2 = O.featuremaps(x.view(B * FS.SEQ_LEN, 3, FS.HEIGHT, FS.WIDTH), sd)
            _, parts = O.tail(x4_1, x4_2, adj, sd, B, FS.SEQ_LEN, [4, 2, 1], 2, return_parts=True)
        g, a = parts["g_f"].numpy(), parts["att_f"].numpy()
        np.savez(path + ".tmp.npz", g=g, a=a)
        os.replace(path + ".tmp.npz", path)
        g_all.append(g), a_all.append(a)
        done = (i + 1) * CHUNK
        print("%s %d / %d tracklets, %.0f s" % (tag, min(done, n), n, time.time() - t0), flush=True)
    return torch.from_numpy(np.concatenate(g_all)), torch.from_numpy(np.concatenate(a_all))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--parts", default="/tmp/fullsplit_parts")
    ap.add_argument("--limit", type=int, default=None, help="debug: only the first N tracklets of each set")
    ap.add_argument("--out", default=os.path.join(HERE, "fullsplit_oracle.npz"))
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    try:  # glibc hands every large activation back to the kernel and page-faults it in again: 3x slower than keeping it
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-1, 1 << 30)   # M_TRIM_THRESHOLD
        libc.mallopt(-3, 1 << 30)   # M_MMAP_THRESHOLD
    except OSError:
        pass
    os.makedirs(args.parts, exist_ok=True)
    from torchreid import models
    m = models.init_model("vmgn", num_classes=FS.N_IDS, loss={"xent", "htri"}, last_stride=1, num_split=4, num_gb=2,
                          num_scale=1, pyramid_part=True, use_pose=True, learn_graph=True, consistent_loss=False)
    sd = recipe_state_dict(m.state_dict(), seed=0)
    del m
    q_pids, q_cams, g_pids, g_cams = FS.labels()
    qg, qa = features(sd, q_pids, q_cams, 0, args.parts, "q", args.limit)
    gg, ga = features(sd, g_pids, g_cams, FS.QUERY_ROWS, args.parts, "g", args.limit)
    ncal = min(2048, gg.shape[0])
    cal = {}
    for name, key, f in (("global_bottleneck", "g", gg[:ncal]), ("att_bottleneck", "a", ga[:ncal])):
        mean, var = f.mean(0), f.var(0, unbiased=False).clamp(min=1e-8)
        sd[name + ".running_mean"], sd[name + ".running_var"] = mean.clone(), var.clone()
        sd[name + ".weight"], sd[name + ".bias"] = torch.ones_like(mean), torch.zeros_like(mean)
        cal["cal_%s_mean" % key], cal["cal_%s_var" % key] = mean.numpy(), var.numpy()
    qf = torch.cat([O._bn(qg, sd, "global_bottleneck"), O._bn(qa, sd, "att_bottleneck")], 1)
    gf = torch.cat([O._bn(gg, sd, "global_bottleneck"), O._bn(ga, sd, "att_bottleneck")], 1)
    nq, ng = qf.shape[0], gf.shape[0]
    out = dict(cal, q_emb_head=qf[:16].numpy(), n_query=nq, n_gallery=ng)
    for metric, fn in (("cosine", O.cosine), ("euclidean", O.euclidean_squared)):
        d = fn(qf, gf).numpy()
        cmc, mAP = O.evaluate_mars(d, q_pids[:nq], g_pids[:ng], q_cams[:nq], g_cams[:ng], 50)
        order = np.argsort(d, axis=1, kind="stable")[:, :51]
        out[metric + "_cmc"], out[metric + "_mAP"] = np.asarray(cmc, dtype=np.float64), np.float64(mAP)
        out[metric + "_idx"] = order.astype(np.int16)
        out[metric + "_val"] = np.take_along_axis(d, order, 1).astype(np.float32)
        print("%s: Rank-1 %.6f mAP %.6f" % (metric, cmc[0], mAP), flush=True)
    np.savez_compressed(args.out, **out)
    print("wrote", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
