"""Seeded weight / input recipes shared by the golden-vector generator and the tests.

Weights are generated per state-dict key from a key-derived seed, so the SAME numbers can be produced for the
reference model (in tests/golden/make_golden.py, build container only) and for this build's model (everywhere)
without shipping 188 MB of weights. BatchNorm running statistics are randomised so eval-mode BN is non-trivial.
"""
import math
import zlib

import numpy as np
import torch


def _gen(key, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    return g


def recipe_tensor(key, shape, seed):
    g = _gen(key, seed)
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=g)
    if leaf == "running_var":
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == "gamma":  # ganet's attention-module scalars (zero at construction; a trained model has them non-zero)
        return torch.full(shape, 0.5)
    if len(shape) == 4:  # conv: kaiming fan_in
        fan_in = shape[1] * shape[2] * shape[3]
        w = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        if "query_conv" in key or "key_conv" in key:
            w = w * 0.15  # keeps ganet's position-attention energies O(1): a softmax that is neither uniform nor one-hot
        return w
    if len(shape) == 2:  # linear
        std = 0.02 if "graph_layers" in key else 0.001
        return torch.randn(shape, generator=g) * std
    if leaf == "weight":  # BN gamma; keep the residual trunk from blowing up
        if ".bn3." in key or "downsample.1" in key:
            return 0.15 + 0.2 * torch.rand(shape, generator=g)
        return 0.8 + 0.4 * torch.rand(shape, generator=g)
    if leaf == "bias":
        return 0.1 * torch.randn(shape, generator=g)
    raise KeyError(key)


def imagenet_like_state_dict(template, seed=0):
    """The recipe with the BatchNorm statistics of a TRAINED ResNet50 instead of the damped ones: every bn3 / downsample gamma in
    U(0.8, 1.2) (the recipe keeps them in 0.15 .. 0.35 so that the random residual trunk does not grow), every other gamma as
    before -- the residual stream's magnitude then grows block by block the way an undamped trunk's does. Used to put numbers on
    the fp16 build's headroom (activations against 65504) instead of assuming it."""
    out = recipe_state_dict(template, seed)
    for key in out:
        if key.endswith(".weight") and (".bn3." in key or "downsample.1" in key) and out[key].dim() == 1:
            g = _gen(key + "#imagenet", seed)
            out[key] = 0.8 + 0.4 * torch.rand(out[key].shape, generator=g)
    return out


def recipe_state_dict(template, seed=0):
    """``template``: a state_dict (or dict name -> tensor/shape) giving keys and shapes."""
    out = {}
    for key, value in template.items():
        shape = tuple(value.shape) if hasattr(value, "shape") else tuple(value)
        out[key] = recipe_tensor(key, shape, seed)
    return out


def synthetic_clips(B, S, H=256, W=128, seed=0, identities=None):
    """MARS-shaped clips: a smooth identity-specific pattern (so embeddings differ between identities)
    plus per-frame noise, roughly zero-mean / unit-range like ImageNet-normalised frames."""
    g = torch.Generator()
    g.manual_seed(1000 + seed)
    noise = 0.5 * torch.randn((B, S, 3, H, W), generator=g)
    if identities is None:
        identities = list(range(B))
    pats = []
    for pid in identities:
        gp = torch.Generator()
        gp.manual_seed(77777 + int(pid))
        low = torch.randn((1, 3, 8, 4), generator=gp)
        pats.append(torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=False))
    pattern = torch.cat(pats, dim=0).view(B, 1, 3, H, W)
    return noise + pattern


def synthetic_part_sets(S, rng, num_split=4, p_missing=0.1):
    """Per frame: dict part -> contiguous 1-based stripe interval (or nothing when the pose is 'undetected')."""
    frames = []
    for _ in range(S):
        ps = {}
        if rng.rand() >= p_missing:
            for name in ("head", "body", "leg"):
                lo = rng.randint(1, num_split + 1)
                hi = rng.randint(lo, num_split + 1)
                ps[name] = set(range(lo, hi + 1))
        frames.append(ps)
    return frames


def synthetic_adj(B, S, seed=0, num_split=4, pyramid_part=True):
    """Binary symmetric zero-diagonal pose adjacency (B,V,V) following the reference's adj_graph rule."""
    from oracle.vmgn_oracle import pose_adjacency

    rng = np.random.RandomState(2000 + seed)
    adjs = [pose_adjacency(synthetic_part_sets(S, rng, num_split), num_split, pyramid_part) for _ in range(B)]
    return torch.from_numpy(np.stack(adjs))


def calibrate_bnneck(sd, g_f, att_f):
    """Give the two BNNeck layers running statistics that match the synthetic data (as training would): the
    embeddings become centred / unit-variance per dimension, so cosine distances between identities are O(1)
    instead of ~1e-4 (random-init features share a dominant common component)."""
    sd = dict(sd)
    for name, f in (("global_bottleneck", g_f), ("att_bottleneck", att_f)):
        sd[name + ".running_mean"] = f.mean(dim=0).clone()
        sd[name + ".running_var"] = f.var(dim=0, unbiased=False).clamp(min=1e-8).clone()
        sd[name + ".weight"] = torch.ones_like(sd[name + ".weight"])
        sd[name + ".bias"] = torch.zeros_like(sd[name + ".bias"])
    return sd
