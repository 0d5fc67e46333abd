"""The full-size synthetic MARS split of SURVEY.md 8(d): 625 identities, 1 980 query and 12 180 gallery tracklets
(5 % junk, pid -1) over 6 cameras, seq_len 8, 256 x 128 frames -- defined so that the CPU (oracle, build container)
and the GPU (product, GPU box) see BIT-IDENTICAL inputs.

Everything random is an integer hash of (tracklet, element) evaluated with torch integer ops -- no transcendental, no
library RNG stream -- followed by exact fp32 steps (a 24-bit integer -> float conversion, one multiply by a power of
two, one subtract), one IEEE multiply and one IEEE add. The same code runs on ``cpu`` and on ``cuda``.

Used by  tests/golden/make_fullsplit.py  (build container: runs the oracle over all 113 280 frames once and commits
Rank-1 / mAP / top-51 lists),  tests/test_gpu_fullsplit.py  and  bench.py's accuracy block  (GPU box: product vs the
committed lists). Imports neither the oracle nor the product.
"""
import numpy as np
import torch

N_IDS, QUERY_ROWS, GALLERY_ROWS, N_CAMS = 625, 1980, 12180, 6
SEQ_LEN, HEIGHT, WIDTH = 8, 256, 128
N_JUNK_PATTERNS = 1024
_M32 = 0xFFFFFFFF
_GOLD = 0x9E3779B1


def labels():
    """-> q_pids, q_cams, g_pids, g_cams (numpy int64). Every query has >= 1 cross-camera match in the gallery (the MARS
    property reference rank.py:203 relies on)."""
    rng = np.random.RandomState(0xFF)
    g_pids = rng.randint(0, N_IDS, GALLERY_ROWS)
    junk = rng.rand(GALLERY_ROWS) < 0.05
    g_pids[junk] = -1
    g_cams = rng.randint(0, N_CAMS, GALLERY_ROWS)
    q_pids = rng.randint(0, N_IDS, QUERY_ROWS)
    q_cams = rng.randint(0, N_CAMS, QUERY_ROWS)
    for i in range(QUERY_ROWS):
        if not np.any((g_pids == q_pids[i]) & (g_cams != q_cams[i])):
            j = rng.randint(0, GALLERY_ROWS)
            g_pids[j], g_cams[j] = q_pids[i], (q_cams[i] + 1) % N_CAMS
    return q_pids, q_cams, g_pids, g_cams


def _mix(x):
    """32-bit avalanche hash on int64 tensors holding values < 2**32 (every product stays below 2**63: no wrap-around is
    relied upon)."""
    x = ((x >> 16) ^ x) * 0x45D9F3B & _M32
    x = ((x >> 16) ^ x) * 0x45D9F3B & _M32
    return (x >> 16) ^ x


def _uniform(x):
    """hash words -> fp32 uniform in (-0.5, 0.5), exact arithmetic."""
    return ((x >> 8).to(torch.float32) + 0.5) * (1.0 / 16777216.0) - 0.5


def _stream_seed(ids, salt):
    """One 32-bit stream seed per (id, salt)."""
    return _mix((ids.to(torch.int64) * _GOLD + salt) & _M32)


def patterns(pattern_ids, device):
    """(b,) pattern ids -> (b, 3, H, W) fp32: an 8 x 4 grid of uniform values (unit variance) held over 32 x 32 blocks."""
    ids = torch.as_tensor(np.asarray(pattern_ids), device=device)
    seed = _stream_seed(ids, 0x51ED27).view(-1, 1)
    e = torch.arange(3 * 8 * 4, device=device, dtype=torch.int64).view(1, -1)
    low = _uniform(_mix((seed + e * _GOLD) & _M32)) * 3.4641016  # sqrt(12): unit variance
    low = low.view(-1, 3, 8, 4)
    return low.repeat_interleave(HEIGHT // 8, dim=2).repeat_interleave(WIDTH // 4, dim=3)


def clips(tracklet_ids, pattern_ids, device, seq_len=SEQ_LEN):
    """-> (b, S, 3, H, W) fp32: the tracklet's identity pattern + per-element noise of standard deviation 0.5."""
    t = torch.as_tensor(np.asarray(tracklet_ids), device=device)
    seed = _stream_seed(t, 0xC11B5).view(-1, 1)
    n = seq_len * 3 * HEIGHT * WIDTH
    e = torch.arange(n, device=device, dtype=torch.int64).view(1, -1)
    x = _mix((seed + e * _GOLD) & _M32)
    del e
    noise = _uniform(x) * 1.7320508  # 0.5 * sqrt(12)
    del x
    noise = noise.view(-1, seq_len, 3, HEIGHT, WIDTH)
    return noise + patterns(pattern_ids, device).view(-1, 1, 3, HEIGHT, WIDTH)


def poses(tracklet_ids, device, seq_len=SEQ_LEN):
    """Synthetic AlphaPose keypoints -> poses (b, S, 18, 3) fp32 (x in [0,128), y in [0,256), confidence in [0,1)) and
    detected (b, S) bool (one frame in ten has no detection)."""
    t = torch.as_tensor(np.asarray(tracklet_ids), device=device)
    seed = _stream_seed(t, 0xB05E).view(-1, 1)
    e = torch.arange(seq_len * 18 * 3, device=device, dtype=torch.int64).view(1, -1)
    u = _uniform(_mix((seed + e * _GOLD) & _M32)) + 0.5
    p = u.view(-1, seq_len, 18, 3) * torch.tensor([float(WIDTH), float(HEIGHT), 1.0], device=device)
    seed_d = _stream_seed(t, 0xDE7EC7).view(-1, 1)
    f = torch.arange(seq_len, device=device, dtype=torch.int64).view(1, -1)
    detected = (_uniform(_mix((seed_d + f * _GOLD) & _M32)) + 0.5) >= 0.1
    return p.contiguous(), detected


def pattern_ids(pids, row_offset):
    """Identity pattern of each tracklet; junk tracklets (pid -1) get one of N_JUNK_PATTERNS distractor patterns by row."""
    pids = np.asarray(pids)
    rows = row_offset + np.arange(len(pids))
    return np.where(pids >= 0, pids, N_IDS + rows % N_JUNK_PATTERNS)


def batches(pids, cams, first_tracklet, device, bs, make_adj):
    """Yields (clips, pids, cams, adj) like the reference's loaders. ``first_tracklet``: global id of row 0 (queries are
    tracklets 0..1979, gallery rows follow); ``make_adj(poses, detected)`` builds the (b, V, V) adjacency -- the product's
    agrl_pose_adjacency on the GPU box, the oracle's restatement of generate_graph in the build container."""
    pids, cams = np.asarray(pids), np.asarray(cams)
    for i in range(0, len(pids), bs):
        p = pids[i:i + bs]
        tid = first_tracklet + i + np.arange(len(p))
        x = clips(tid, pattern_ids(p, i), device)
        ps, det = poses(tid, device)
        yield x, p, cams[i:i + bs], make_adj(ps, det)


def compare_topk(idx, val, ref_idx, ref_val, k=50):
    """Device top-k lists (m,k) against the oracle's first k+1 positions (m,k+1). Ranking indices are only defined where
    neighbouring distances differ by more than the two implementations' rounding error, so every position where the indices
    differ must be EXPLAINED: the device's choice sits in the oracle's list at a distance within ``tol`` of the oracle's
    entry at that position (a near-tie swap), tol = 2 x (largest distance disagreement at matching positions) + 1e-6.
    -> dict(agreement, rows_equal, top1_agreement, max_abs_val_err, tol, unexplained, oracle_min_gap)."""
    idx, val = np.asarray(idx)[:, :k].astype(np.int64), np.asarray(val)[:, :k].astype(np.float64)
    ref_idx, ref_val = np.asarray(ref_idx).astype(np.int64), np.asarray(ref_val).astype(np.float64)
    same = idx == ref_idx[:, :k]
    err = float(np.abs(val - ref_val[:, :k])[same].max()) if same.any() else float("inf")
    tol = 2.0 * err + 1e-6
    unexplained = 0
    rows, cols = np.nonzero(~same)
    for r, c in zip(rows, cols):
        pos = np.nonzero(ref_idx[r] == idx[r, c])[0]
        if len(pos) == 0 or abs(ref_val[r, pos[0]] - ref_val[r, c]) > tol:
            unexplained += 1
    return {"agreement": float(same.mean()), "rows_equal": float(same.all(axis=1).mean()), "top1_agreement": float(same[:, 0].mean()),
            "max_abs_val_err": err, "tol": tol, "swapped_positions": int((~same).sum()), "unexplained": int(unexplained),
            "oracle_min_gap": float(np.diff(ref_val, axis=1).min())}


def load_oracle_fixture():
    """tests/golden/fullsplit_oracle.npz (made by tests/golden/make_fullsplit.py), or None when it has not been generated."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsplit_oracle.npz")
    return np.load(path) if os.path.exists(path) else None


def apply_calibration(model, z):
    """Load the fixture's BNNeck statistics into the model's two BatchNorm1d layers (weight 1, bias 0)."""
    with torch.no_grad():
        for bn, key in ((model.global_bottleneck, "g"), (model.att_bottleneck, "a")):
            bn.running_mean.copy_(torch.from_numpy(z["cal_%s_mean" % key]).to(bn.running_mean.device))
            bn.running_var.copy_(torch.from_numpy(z["cal_%s_var" % key]).to(bn.running_var.device))
            bn.weight.fill_(1.0)
            bn.bias.zero_()
    if hasattr(model, "invalidate_hip_cache"):
        model.invalidate_hip_cache()
