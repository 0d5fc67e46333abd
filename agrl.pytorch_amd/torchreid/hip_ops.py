"""Tensor-level wrappers of the C-ABI entry points (include/agrl_hip.h).

Each function checks shapes/dtypes, allocates the output through torch's caching allocator on the
input's device, and launches on the calling thread's current HIP stream. No arithmetic happens here.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _hip
from ._hip import F32, LP16, LP_DTYPE, LP_NAME, METRIC_COSINE, METRIC_EUCLIDEAN, call, dtype_code, ptr


import contextlib
import threading
import weakref

_MODE = threading.local()


@contextlib.contextmanager
def f32_split(on=True):
    """Inside this block fp32 convs / Linears use the split-bf16 recipe (dtype code AGRL_F32X3: every product as three
    bf16 MFMAs on the high / low halves of the operands, ~1e-5 relative, 2-3 x the exact-fp32 rate). Per thread."""
    prev = getattr(_MODE, 'split', False)
    _MODE.split = bool(on)
    try:
        yield
    finally:
        _MODE.split = prev


_SWITCHES = {}


def switch(name, default='1'):
    """A Python-side tuning switch (AGRL_HIP_*), read from the environment ONCE and cached like the library's own options:
    ``_hip.reload_options()`` (what the A/B tools and the tests call after changing the environment) drops the cache. Round-5 review:
    the dispatch predicates used to call os.environ.get 17 times per Bottleneck of every forward."""
    v = _SWITCHES.get(name)
    if v is None:
        v = _SWITCHES[name] = os.environ.get(name, default)
    return v


def switch_on(name, default='1'):
    return switch(name, default) != '0'


_hip.RELOAD_HOOKS.append(_SWITCHES.clear)


def _gemm_code(dt):
    if dt == torch.float32 and getattr(_MODE, 'split', False):
        return _hip.F32X3
    return dtype_code(dt)


def _stream(t):
    return _hip.stream_ptr(t.device)


def _dev(t):
    return torch.cuda.device(t.device)


def stem(x_nchw, w_ohwi, bias, out_dtype):
    """(N,3,H,W) fp32 NCHW -> (N,PH,PW,64) NHWC. vmgn.py:281-284."""
    assert x_nchw.dtype == torch.float32 and x_nchw.dim() == 4 and x_nchw.size(1) == 3
    x_nchw = x_nchw.contiguous()
    N, _, H, W = x_nchw.shape
    CH, CW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    PH, PW = (CH + 2 - 3) // 2 + 1, (CW + 2 - 3) // 2 + 1
    out = torch.empty((N, PH, PW, 64), dtype=out_dtype, device=x_nchw.device)
    with _dev(x_nchw):
        call("agrl_stem_conv_bn_relu_maxpool", ptr(x_nchw), ptr(w_ohwi), ptr(bias), ptr(out), N, H, W,
             dtype_code(out_dtype), _stream(x_nchw))
    return out


PRECISIONS = ('fp32', LP_NAME, 'bf16x3', 'fp16x3')


def check_precision(precision, allowed=PRECISIONS, what="hip_precision"):
    """'fp32' exact-fp32 MFMA | LP_NAME ('fp16' or 'bf16': the loaded library's 16-bit storage type, fp32 accumulation) |
    'bf16x3' fp32 tensors, split-bf16 products | 'fp16x3' (round 6) fp32 tensors, conv products as three fp16 MFMAs on fp16 high /
    low halves with pre-scaled weights (22 significand bits per operand). Asking for the OTHER 16-bit type is an error, not a silent substitution."""
    if precision in allowed:
        return precision
    if precision in ('fp16', 'bf16'):
        raise ValueError("%s %r: the loaded library stores 16-bit data as %s (set AGRL_HIP_LP16=%s before importing torchreid to load "
                         "the other build)" % (what, precision, LP_NAME, precision))
    raise ValueError("%s must be one of %s, got %r" % (what, ", ".join(repr(a) for a in allowed), precision))


def is_lp16(precision):
    """True for the 16-bit mode (validates the name)."""
    return check_precision(precision, PRECISIONS, "precision") == LP_NAME


def pack_stem_weights_lp16(w_ohwi):
    """(64,7,7,3) fp32 OHWI (BN folded) -> (64,240) bf16: per filter row 8 taps x 4 channels, zero padded; 480-byte rows
    (30 sixteen-byte slots: the stride that makes the kernel's weight-fragment reads bank-conflict free)."""
    assert tuple(w_ohwi.shape) == (64, 7, 7, 3)
    w = torch.zeros((64, 7, 8, 4), dtype=torch.float32, device=w_ohwi.device)
    w[:, :, :7, :3] = w_ohwi.float()
    packed = torch.zeros((64, 240), dtype=LP_DTYPE, device=w_ohwi.device)
    packed[:, :224] = w.view(64, 224).to(LP_DTYPE)
    return packed.contiguous()


def stem_lp16(x_nchw, w_packed, bias):
    """16-bit-MFMA stem: (N,3,H,W) fp32 NCHW -> (N,PH,PW,64) NHWC in the library's 16-bit type (LP_DTYPE). vmgn.py:281-284."""
    assert x_nchw.dtype == torch.float32 and x_nchw.dim() == 4 and x_nchw.size(1) == 3
    assert w_packed.dtype == LP_DTYPE and tuple(w_packed.shape) == (64, 240)
    x_nchw = x_nchw.contiguous()
    N, _, H, W = x_nchw.shape
    CH, CW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    PH, PW = (CH + 2 - 3) // 2 + 1, (CW + 2 - 3) // 2 + 1
    out = torch.empty((N, PH, PW, 64), dtype=LP_DTYPE, device=x_nchw.device)
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * N * CH * CW * 64 * 147, "bytes": 4.0 * x_nchw.numel() + 2.0 * out.numel() + 2.0 * w_packed.numel(),
                            "conv": (7, 2, 3, 64, PH, PW)}
    with _dev(x_nchw):
        call("agrl_stem_conv_bn_relu_maxpool_lp16", ptr(x_nchw), ptr(w_packed), ptr(bias), ptr(out), N, H, W,
             _stream(x_nchw))
    return out


def conv_bn_act(x, w_ohwi, bias, stride, pad, relu, residual=None, x_presplit=False, out_presplit=False):
    """NHWC conv (BN folded) [+ residual] [+ ReLU]. vmgn.py:45-65. ``x_presplit`` / ``out_presplit`` ('fp16x3' weights only): the
    input was / the output is to be stored with its fp16 halves already formed (agrl_conv2d_bn_act_split16: a tensor only a GEMM reads;
    the same shape and bytes, not readable as fp32)."""
    N, H, W, Cin = x.shape
    Cout, R, S, Cin2 = w_ohwi.shape
    assert Cin == Cin2 and x.dtype == w_ohwi.dtype
    OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    out = torch.empty((N, OH, OW, Cout), dtype=x.dtype, device=x.device)
    if residual is not None:
        assert residual.shape == out.shape and residual.dtype == out.dtype
    if _hip.PROFILE is not None:
        e = x.element_size()
        _hip.PROFILE_TAG = {"flops": 2.0 * N * OH * OW * Cout * R * S * Cin,
                            "bytes": e * (x.numel() + w_ohwi.numel() + out.numel() * (2 if residual is not None else 1)),
                            "conv": (R, stride, Cin, Cout, OH, OW)}
    unscale = getattr(w_ohwi, 'agrl_unscale', None)
    with _dev(x):
        if unscale is not None:
            # 'fp16x3': weights pre-scaled by a power of two and pre-split at pack time (split16_inloop_weights); every product as three fp16 MFMAs
            assert x.dtype == torch.float32 and getattr(w_ohwi, 'agrl_presplit', False), "fp16x3 weights come from split16_inloop_weights"
            call("agrl_conv2d_bn_act_split16", ptr(x), ptr(w_ohwi), ptr(bias), ptr(residual), ptr(out), N, H, W, Cin, Cout, R, S,
                 stride, pad, 1 if relu else 0, float(unscale), 1 if x_presplit else 0, 1 if out_presplit else 0, _stream(x))
        else:
            assert not (x_presplit or out_presplit), "pre-split activations exist in the 'fp16x3' arithmetic only"
            call("agrl_conv2d_bn_act", ptr(x), ptr(w_ohwi), ptr(bias), ptr(residual), ptr(out), N, H, W, Cin, Cout, R, S,
                 stride, pad, 1 if relu else 0, _gemm_code(x.dtype), _stream(x))
    return out


def conv1x1_dual_split16(x, x2, w_cat, bias, stride, relu=True, x2_presplit=False, out_planes=0):
    """relu([x sampled at the stride | x2] @ w_cat^T + bias) in the split-fp16 arithmetic on fp32 tensors: a first Bottleneck's conv3 + its
    1x1 stride-s downsample conv in one GEMM (vmgn.py:56-64). x (N,H,W,K1) block input, x2 (N,OH,OW,K2) conv2's output, w_cat (Cout,
    K1+K2) from split16_inloop_weights of the concatenation (ONE power of two for both halves) -> (N,OH,OW,Cout) fp32. ``x2_presplit``: x2
    was written by conv_bn_act(..., out_presplit=True). ``out_planes`` 2 / 3: the result leaves as split-fp16 planes (N,OH,OW,out_planes
    Cout) fp16 -- to_split16_planes of the fp32 map, without the map."""
    N, H, W, K1 = x.shape
    K2 = x2.shape[3]
    Cout = w_cat.shape[0]
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    assert x.dtype == torch.float32 and x2.dtype == torch.float32 and x.is_contiguous() and x2.is_contiguous()
    assert tuple(x2.shape[:3]) == (N, OH, OW) and tuple(w_cat.shape) == (Cout, K1 + K2) and getattr(w_cat, 'agrl_presplit', False)
    assert out_planes in (0, 2, 3)
    out = (torch.empty((N, OH, OW, out_planes * Cout), dtype=torch.float16, device=x.device) if out_planes
           else torch.empty((N, OH, OW, Cout), dtype=torch.float32, device=x.device))
    if _hip.PROFILE is not None:
        M = N * OH * OW
        _hip.PROFILE_TAG = {"flops": 2.0 * M * Cout * (K1 + K2), "bytes": 4.0 * (M * (K1 + K2) + w_cat.numel()) + out.numel() * out.element_size(),
                            "conv": (1, stride, K1 + K2, Cout, OH, OW)}
    with _dev(x):
        call("agrl_conv1x1_dual_split16", ptr(x), ptr(x2), ptr(w_cat), ptr(bias), ptr(out), N, H, W, stride, K1, K2, Cout,
             1 if relu else 0, float(w_cat.agrl_unscale), 1 if x2_presplit else 0, int(out_planes), _stream(x))
    return out


def split16_prescale(w):
    """fp32 weights -> the same tensor times 2^k, k chosen so that max |w| 2^k lies in [2^13, 2^14) (exact; fp16's largest finite
    value is 65504 = ~2^16), with ``.agrl_unscale`` = 2^-k riding on the tensor object: what agrl_conv2d_bn_act_split16 expects.
    With the largest weight at ~2^13.5 a weight 2^-13 of it still has a NORMAL fp16 low half, and anything smaller is off by at
    most 2^-25 absolute = 2^-38 of the largest."""
    import math
    amax = float(w.detach().abs().max())
    if not (amax > 0.0 and math.isfinite(amax)):
        k = 0
    else:
        k = 13 - math.frexp(amax)[1] + 1       # frexp: amax = m 2^e, m in [0.5, 1) -> amax 2^k in [2^13, 2^14)
    ws = (w.detach().float() * (2.0 ** k)).contiguous()
    ws.agrl_unscale = 2.0 ** (-k)
    return ws


def split16_inloop_weights(w):
    """fp32 weights (.., K), K % 32 == 0 -> the weight operand of the kernels that split in the k-loop (conv_bn_act on 'fp16x3' weights,
    conv1x1_dual_split16, graph_linear_mix): split16_prescale, then the fp16 high / low halves of every 32-value k-tile side by side
    (the layout of agrl_split16_weights_inloop: the same shape and bytes, NOT readable as fp32 any more). ``.agrl_unscale`` = 2^-k and
    ``.agrl_presplit`` ride on the tensor; ``.agrl_scaled`` keeps the scaled fp32 tensor for packers that derive other forms from it
    (split16_true_weights)."""
    ws = split16_prescale(w)
    K = ws.shape[-1]
    assert K % 32 == 0, "the in-loop split needs whole 32-element k-tiles (K = %d)" % K
    # pack-time host logic in torch ops (any device; a C caller has agrl_split16_weights_inloop, byte-identical:
    # tests/test_gpu_kernels.py::test_split16_inloop_weight_layout): (rows, tile, c, half, e) = lane group c's eight values of a k-tile
    t = ws.view(-1, K // 32, 2, 4, 4).permute(0, 1, 3, 2, 4)
    hi = t.to(torch.float16)
    lo = (t - hi.float()).to(torch.float16)
    out = torch.stack([hi, lo], dim=2).contiguous().view(torch.float32).view(ws.shape)
    out.agrl_unscale = ws.agrl_unscale
    out.agrl_presplit = True
    out.agrl_scaled = ws
    return out


def split16_true_weights(t):
    """The fp32 weights a split16_prescale / split16_inloop_weights tensor was made from (the power-of-two scale undone: exact)."""
    src = t.agrl_scaled if getattr(t, 'agrl_presplit', False) else t
    return src * t.agrl_unscale


# ---- split-fp16 PLANES (round 6): the conforming mode at speed -- include/agrl_hip.h, "Split-fp16 PLANES" ----------------------
def pack_stem_weights_split16(w_ohwi):
    """(64,7,7,3) fp32 OHWI (BN folded) -> (wh_packed, wl_packed, w_unscale): the fp16 high / low halves of w 2^k, each in the 16-bit
    stem's (64, 240) layout (per filter row 8 taps x 4 channels, zero padded, 480-byte rows)."""
    assert tuple(w_ohwi.shape) == (64, 7, 7, 3)
    ws = split16_prescale(w_ohwi)
    w4 = torch.zeros((64, 7, 8, 4), dtype=torch.float32, device=w_ohwi.device)
    w4[:, :, :7, :3] = ws
    wh = w4.to(torch.float16)
    wl = (w4 - wh.float()).to(torch.float16)
    out = []
    for t in (wh, wl):
        packed = torch.zeros((64, 240), dtype=torch.float16, device=w_ohwi.device)
        packed[:, :224] = t.view(64, 224)
        out.append(packed.contiguous())
    return out[0], out[1], ws.agrl_unscale


def stem_split16(x_nchw, wh_packed, wl_packed, unscale, bias):
    """Split-fp16 stem: (N,3,H,W) fp32 NCHW -> (N,PH,PW,64) fp32 NHWC (agrl_stem_split16). vmgn.py:281-284."""
    assert x_nchw.dtype == torch.float32 and x_nchw.dim() == 4 and x_nchw.size(1) == 3
    x_nchw = x_nchw.contiguous()
    N, _, H, W = x_nchw.shape
    CH, CW = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    PH, PW = (CH + 2 - 3) // 2 + 1, (CW + 2 - 3) // 2 + 1
    out = torch.empty((N, PH, PW, 64), dtype=torch.float32, device=x_nchw.device)
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * N * CH * CW * 64 * 147, "bytes": 4.0 * x_nchw.numel() + 4.0 * out.numel() + 4.0 * wh_packed.numel(),
                            "conv": (7, 2, 3, 64, PH, PW)}
    with _dev(x_nchw):
        call("agrl_stem_split16", ptr(x_nchw), ptr(wh_packed), ptr(wl_packed), ptr(bias), ptr(out), N, H, W, float(unscale), _stream(x_nchw))
    return out


def split16_planes_available():
    """The plane kernels exist in the fp16 build of the library only."""
    return LP_NAME == 'fp16'


def split16_plane_weights(w, segments=None, pair_first=False):
    """fp32 weights (Cout, K) / OHWI (Cout, R, S, Cin) -> (fp16 tensor with 3 x the innermost axis = [wh | wh 2^-11 | wl] of w 2^k,
    2^-k): the operand the plane kernels multiply with activation planes [xh | xl 2^11 | xh] -- xh wh + xl wh + xh wl in one
    accumulator. k as in split16_prescale (max |w| 2^k in [2^13, 2^14)); ONE k for the whole tensor, also for the two-source form
    (``segments`` = [K1, K2]: the innermost axis is [W1 | W2] and each source gets its own plane triple, [W1 triple | W2 triple]).
    ``pair_first``: the first source is a plane PAIR [xh | xl 2^11] (include/agrl_hip.h, "Plane PAIRS"): its columns run per
    128-channel slab c as [wh_c | wl_c | wh_c 2^-11], against the kernel's slab reads [xh_c | xh_c | xl_c]."""
    ws = split16_prescale(w)
    parts = []
    for i, seg in enumerate(torch.split(ws, segments or [ws.shape[-1]], dim=-1)):
        wh = seg.to(torch.float16)
        wl = (seg - wh.float()).to(torch.float16)
        wh_s = (wh.float() * (2.0 ** -11)).to(torch.float16)
        if pair_first and i == 0:
            assert seg.shape[-1] % 128 == 0
            lead = tuple(seg.shape[:-1])
            slabs = [t.reshape(lead + (-1, 128)) for t in (wh, wl, wh_s)]
            parts.append(torch.stack(slabs, dim=-2).reshape(lead + (3 * seg.shape[-1],)))
        else:
            parts += [wh, wh_s, wl]
    return torch.cat(parts, dim=-1).contiguous(), ws.agrl_unscale


def to_split16_planes(x, nplanes=3):
    """fp32 (..., C) -> fp16 (..., 3 C) = [hi | lo 2^11 | hi], or with ``nplanes`` = 2 the pair (..., 2 C) = [hi | lo 2^11]
    (agrl_split16_planes)."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] % 4 == 0 and nplanes in (2, 3)
    Cc = x.shape[-1]
    out = torch.empty(tuple(x.shape[:-1]) + (nplanes * Cc,), dtype=torch.float16, device=x.device)
    with _dev(x):
        call("agrl_split16_planes", ptr(x), ptr(out), x.numel() // Cc, Cc, nplanes, _stream(x))
    return out


def to_split16_weight_planes(x, scale):
    """fp32 (rows, C) -> fp16 (rows, 3 C) = [h | h 2^-11 | l] of x * scale (a power of two), on the device (agrl_split16_weight_planes):
    the gallery side of distmat_split16."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2 and x.shape[1] % 4 == 0
    out = torch.empty((x.shape[0], 3 * x.shape[1]), dtype=torch.float16, device=x.device)
    with _dev(x):
        call("agrl_split16_weight_planes", ptr(x), ptr(out), x.shape[0], x.shape[1], float(scale), _stream(x))
    return out


def distmat_split16(q3, g3, metric, g_unscale, qn=None, gn=None, out=None):
    """q3 (m, 3D) query planes, g3 (n, 3D) gallery weight planes -> fp32 (m, n) distance matrix in the split-fp16 arithmetic
    (agrl_distmat_split16). distance.py:59-89."""
    m, D3 = q3.shape
    n = g3.shape[0]
    assert g3.shape[1] == D3 and q3.dtype == torch.float16 and g3.dtype == torch.float16
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=q3.device)
    code = METRIC_EUCLIDEAN if metric == "euclidean" else METRIC_COSINE
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * m * n * D3 / 3, "mfma_flops": 2.0 * m * n * D3, "bytes": 2.0 * (m + n) * D3 + 4.0 * m * n}
    ws = None
    if (-(-m // 64)) * (-(-n // 128)) < 256:
        ws = torch.empty((8 * m * n,), dtype=torch.float32, device=q3.device)
    with _dev(q3):
        call("agrl_distmat_split16", ptr(q3), ptr(g3), ptr(qn), ptr(gn), out.data_ptr(), m, n, D3, out.stride(0), code, float(g_unscale),
             ptr(ws), 0 if ws is None else ws.numel() * 4, _stream(q3))
    return out


def from_split16_planes(x3, nplanes=3):
    """planes (..., 3 C) (or a pair, (..., 2 C)) -> fp32 (..., C) = hi + lo 2^-11 (exact): tests and the stage-by-stage parity hooks
    (torch arithmetic: not on the product path)."""
    Cc = x3.shape[-1] // nplanes
    return x3[..., :Cc].float() + x3[..., Cc:2 * Cc].float() * (2.0 ** -11)


def conv1x1_split16(x3, packed, unscale, bias, Cout, residual3=None, relu=True, x2=None, layout=0):
    """1x1 conv on planes: act(w_unscale [x3 | x2] @ W3^T + bias + residual) -> planes (N,H,W,3 Cout). vmgn.py:48-50, :56-64.
    ``layout`` bit 0: x3 is a plane pair (N,H,W,2 K) with weights from split16_plane_weights(pair_first=True); bit 1: the residual
    and the result are pairs (N,H,W,2 Cout). x2 is always a triple."""
    N, H, W, Kx = x3.shape
    K3 = Kx // 2 * 3 if layout & 1 else Kx     # the k-loop's columns
    K23 = 0 if x2 is None else x2.shape[3]
    M = N * H * W
    no = 2 if layout & 2 else 3
    assert x3.dtype == torch.float16 and x3.is_contiguous() and packed.numel() == 2 * (K3 + K23) * Cout
    assert residual3 is None or (residual3.dtype == torch.float16 and residual3.is_contiguous() and tuple(residual3.shape) == (N, H, W, no * Cout))
    out = torch.empty((N, H, W, no * Cout), dtype=torch.float16, device=x3.device)
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * M * Cout * (K3 + K23) / 3, "mfma_flops": 2.0 * M * Cout * (K3 + K23),
                            "bytes": 2.0 * (x3.numel() + (0 if x2 is None else x2.numel()) + out.numel() + (0 if residual3 is None else residual3.numel() * 2 // no)) + packed.numel()}
    with _dev(x3):
        if x2 is None:
            call("agrl_conv1x1_split16", ptr(x3), ptr(packed), ptr(bias), ptr(residual3), ptr(out), M, K3, Cout, 1 if relu else 0,
                 float(unscale), int(layout), _stream(x3))
        else:
            assert residual3 is None and x2.dtype == torch.float16 and x2.is_contiguous() and tuple(x2.shape[:3]) == (N, H, W)
            call("agrl_conv1x1_split16_dual", ptr(x3), ptr(x2), ptr(packed), ptr(bias), ptr(out), M, K3, K23, Cout, 1 if relu else 0,
                 float(unscale), int(layout), _stream(x3))
    return out


def conv1x1_split16_pool(x3, packed, unscale, bias, Cout, residual3, splits, mean, relu=True, layout=0):
    """Last conv of a layer-4 branch on planes with the frame pooling in the epilogue (the unrounded fp32 values are pooled; no map).
    -> pooled fp32 (F, P, Cout). vmgn.py:56-64 + :298-308."""
    N, H, W, K3 = x3.shape
    assert x3.dtype == torch.float16 and x3.is_contiguous() and (H, W) == (16, 8) and packed.numel() == 2 * K3 * Cout
    assert not (layout & 1), "the pooled conv's input is conv2's output: a triple"
    assert residual3 is None or (residual3.dtype == torch.float16 and residual3.is_contiguous()
                                 and tuple(residual3.shape) == (N, H, W, (2 if layout & 2 else 3) * Cout))
    P = int(sum(splits))
    pooled = torch.empty((N, P, Cout), dtype=torch.float32, device=x3.device)
    arr = (C.c_int * len(splits))(*[int(s_) for s_ in splits])
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * N * H * W * Cout * K3 / 3, "mfma_flops": 2.0 * N * H * W * Cout * K3,
                            "bytes": 2.0 * (x3.numel() + (residual3.numel() if residual3 is not None else 0)) + packed.numel() + 4.0 * pooled.numel()}
    with _dev(x3):
        call("agrl_conv1x1_split16_pool", ptr(x3), ptr(packed), ptr(bias), ptr(residual3), ptr(pooled), N, H, W, K3, Cout,
             1 if relu else 0, arr, len(splits), 1 if mean else 0, float(unscale), int(layout), _stream(x3))
    return pooled


def conv3x3_split16(x3, packed, unscale, bias, Cout, relu=True):
    """relu(conv3x3(x) + bias), stride 1 / pad 1, on planes -> planes (N,H,W,3 Cout). vmgn.py:52-54."""
    N, H, W, Cin3 = x3.shape
    assert x3.dtype == torch.float16 and x3.is_contiguous() and packed.numel() == 2 * 9 * Cin3 * Cout
    out = torch.empty((N, H, W, 3 * Cout), dtype=torch.float16, device=x3.device)
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * N * H * W * Cout * 9 * Cin3 / 3, "mfma_flops": 2.0 * N * H * W * Cout * 9 * Cin3,
                            "bytes": 2.0 * (x3.numel() + out.numel()) + packed.numel()}
    with _dev(x3):
        call("agrl_conv3x3_packed_split16", ptr(x3), ptr(packed), ptr(bias), ptr(out), N, H, W, Cin3, Cout, 1 if relu else 0,
             float(unscale), _stream(x3))
    return out


def conv1x1_dual_supported(x1, x2, w_cat):
    """The two-source pointwise GEMM exists for bf16, K1 == 2 K2, K1 % 64 == 0, Cout % 256 == 0 (layer-4 first blocks)."""
    K1, K2 = x1.shape[-1], x2.shape[-1]
    return (x1.dtype == LP_DTYPE and x2.dtype == LP_DTYPE and K1 == 2 * K2 and K1 % 64 == 0
            and w_cat.shape[0] % 256 == 0 and x1.shape[:-1] == x2.shape[:-1] and switch_on('AGRL_HIP_FUSE_DS'))


def conv1x1_dual(x1, x2, w_cat, bias, relu=True):
    """relu([x1 | x2] @ w_cat^T + bias): a first Bottleneck's conv3 + its 1x1 stride-1 downsample conv in one GEMM
    (vmgn.py:56-64). x1 (N,H,W,K1) block input, x2 (N,H,W,K2) conv2 output, w_cat (Cout, K1+K2) -> (N,H,W,Cout) bf16."""
    N, H, W, K1 = x1.shape
    K2 = x2.shape[-1]
    Cout = w_cat.shape[0]
    assert tuple(w_cat.shape) == (Cout, K1 + K2) and w_cat.dtype == x1.dtype
    out = torch.empty((N, H, W, Cout), dtype=x1.dtype, device=x1.device)
    M = N * H * W
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * M * Cout * (K1 + K2), "bytes": 2.0 * (x1.numel() + x2.numel() + w_cat.numel() + out.numel()),
                            "conv": (1, 1, K1 + K2, Cout, H, W)}
    with _dev(x1):
        call("agrl_conv1x1_dual_bn_act", ptr(x1), ptr(x2), ptr(w_cat), ptr(bias), ptr(out), M, K1, K2, Cout, 1 if relu else 0,
             _stream(x1))
    return out


def conv1x1_bn_act_pool(x, w_ohwi, bias, residual, splits, mean, want_lp, relu=True):
    """Last 1x1 conv of a layer4 branch with the frame pooling fused in (bf16, 128-pixel frames): the 2048-channel map
    is never written. -> pooled fp32 (F, P, Cout) [, bf16 copy]. vmgn.py:45-65 + :298-308."""
    N, H, W, Cin = x.shape
    Cout = w_ohwi.shape[0]
    assert x.dtype == LP_DTYPE and tuple(w_ohwi.shape[1:3]) == (1, 1) and H * W == 128
    P = int(sum(splits))
    pooled = torch.empty((N, P, Cout), dtype=torch.float32, device=x.device)
    pooled_lp = torch.empty((N, P, Cout), dtype=LP_DTYPE, device=x.device) if want_lp else None
    arr = (C.c_int * len(splits))(*[int(s) for s in splits])
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * N * H * W * Cout * Cin,
                            "bytes": 2.0 * (x.numel() + w_ohwi.numel() + (residual.numel() if residual is not None else 0)) + 4.0 * pooled.numel()}
    with _dev(x):
        call("agrl_conv1x1_bn_act_pool", ptr(x), ptr(w_ohwi), ptr(bias), ptr(residual), None, ptr(pooled), ptr(pooled_lp),
             N, H, W, Cin, Cout, 1 if relu else 0, arr, len(splits), 1 if mean else 0, _stream(x))
    return pooled, pooled_lp


# (Cmid, Cout, Cnext) of the layer-3 / layer-4 seams csrc/bottleneck_seam.hip is built for (the third: layer 3 -> a layer-4 branch)
SEAM_SHAPES = ((256, 1024, 256), (512, 2048, 512), (256, 1024, 512))


def seam_enabled():
    """AGRL_HIP_FUSE_SEAM=0 runs the layer-3 seams as two launches (A/B)."""
    return switch_on('AGRL_HIP_FUSE_SEAM')


def bottleneck_seam_supported(w3, w1_next, pixels=None):
    """conv3 + residual -> next conv1 back to back (agrl_bottleneck_seam): 16-bit weights of a layer-3 / layer-4 seam, and --
    when ``pixels`` is given -- a pixel count made of whole 128-pixel tiles (16 x 8 frames)."""
    return (w3.dtype == LP_DTYPE and w1_next.dtype == LP_DTYPE and w3.dim() == 4 and tuple(w3.shape[1:3]) == (1, 1)
            and (w3.shape[3], w3.shape[0], w1_next.shape[0]) in SEAM_SHAPES and tuple(w1_next.shape[1:]) == (1, 1, w3.shape[0])
            and (pixels is None or pixels % 128 == 0) and switch_on('AGRL_HIP_FUSE_SEAM'))


def bottleneck_seam_pack(w3, w1_next):
    """The two (static) weight matrices of a seam re-ordered once into the per-wave MFMA fragment streams the kernel loads
    straight into registers -> one uint8 tensor of agrl_bottleneck_seam_packed_bytes bytes."""
    assert bottleneck_seam_supported(w3, w1_next)
    Cout, Cmid, Cnext = w3.shape[0], w3.shape[3], w1_next.shape[0]
    w3, w1_next = w3.contiguous(), w1_next.contiguous()
    nbytes = int(_hip.lib().agrl_bottleneck_seam_packed_bytes(Cmid, Cout, Cnext))
    assert nbytes == 2 * (w3.numel() + w1_next.numel())
    packed = torch.empty((nbytes,), dtype=torch.uint8, device=w3.device)
    with _dev(w3):
        call("agrl_bottleneck_seam_pack", ptr(w3), ptr(w1_next), ptr(packed), Cmid, Cout, Cnext, _stream(w3))
    return packed


def bottleneck_seam(y2, packed, b3, residual, b1_next, dims):
    """out = relu(conv3(y2) + residual), z = relu(conv1_next(out)) in one pass over 128-pixel tiles; ``packed`` from
    bottleneck_seam_pack, ``dims`` = (Cmid, Cout, Cnext). vmgn.py:56-64 (block i) + :48-50 (block i+1).
    -> out (N,H,W,Cout), z (N,H,W,Cnext) 16-bit NHWC."""
    Cmid, Cout, Cnext = dims
    N, H, W, C = y2.shape
    M = N * H * W
    assert y2.dtype == LP_DTYPE and C == Cmid and residual.shape == (N, H, W, Cout) and residual.dtype == y2.dtype and M % 128 == 0
    assert y2.is_contiguous() and residual.is_contiguous()
    out = torch.empty((N, H, W, Cout), dtype=y2.dtype, device=y2.device)
    z = torch.empty((N, H, W, Cnext), dtype=y2.dtype, device=y2.device)
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * M * (Cmid * Cout + Cout * Cnext),
                            "bytes": 2.0 * (y2.numel() + residual.numel() + out.numel() + z.numel()) + packed.numel(),
                            "conv": (1, 1, Cmid, Cout, H, W)}
    with _dev(y2):
        call("agrl_bottleneck_seam", ptr(y2), ptr(packed), ptr(b3), ptr(residual), ptr(out), ptr(b1_next), ptr(z), M, Cmid, Cout,
             Cnext, _stream(y2))
    return out, z


def conv3x3_packed_enabled():
    """AGRL_HIP_CONV3X3_PACKED=0 runs the layer-3 / layer-4 3x3 convs through conv_bn_act (A/B; bit-identical results)."""
    return switch_on('AGRL_HIP_CONV3X3_PACKED')


def conv3x3_packed_supported(w_ohwi, H=None, W=None):
    """3x3 / stride 1 conv through the four-wave kernel with a pre-packed weight stream (csrc/conv3x3_fat.hip): 16-bit weights,
    Cin % 64 == 0 (>= 128), Cout % 256 == 0 -- or Cout == 128 (layer 2: packed as the lower half of a 256-channel tile) -- and, when
    given, maps made of whole 16 x 8 blocks."""
    return (w_ohwi.dtype == LP_DTYPE and w_ohwi.dim() == 4 and tuple(w_ohwi.shape[1:3]) == (3, 3) and w_ohwi.shape[3] % 64 == 0
            and w_ohwi.shape[3] >= 128 and (w_ohwi.shape[0] % 256 == 0 or w_ohwi.shape[0] == 128) and (H is None or (H % 16 == 0 and W % 8 == 0)))


def conv3x3_pack(w_ohwi):
    """OHWI 3x3 weights re-ordered once into per-wave MFMA fragment streams (agrl_conv3x3_pack) -> uint8 tensor."""
    assert conv3x3_packed_supported(w_ohwi)
    Cout, Cin = w_ohwi.shape[0], w_ohwi.shape[3]
    if Cout == 128:   # the lower half of one 256-channel tile; the upper half (zeros) is never read by the launch
        w_ohwi = torch.cat([w_ohwi, torch.zeros_like(w_ohwi)], dim=0)
        Cout = 256
    w_ohwi = w_ohwi.contiguous()
    nbytes = int(_hip.lib().agrl_conv3x3_packed_bytes(Cin, Cout))
    assert nbytes == 2 * w_ohwi.numel()
    packed = torch.empty((nbytes,), dtype=torch.uint8, device=w_ohwi.device)
    with _dev(w_ohwi):
        call("agrl_conv3x3_pack", ptr(w_ohwi), ptr(packed), Cin, Cout, _stream(w_ohwi))
    return packed


def conv3x3_packed(x, packed, bias, Cout, relu=True):
    """relu(conv3x3(x) + bias), stride 1 / pad 1, weights from conv3x3_pack. vmgn.py:52-54. -> (N,H,W,Cout) 16-bit NHWC."""
    N, H, W, Cin = x.shape
    assert x.dtype == LP_DTYPE and x.is_contiguous() and packed.numel() == 2 * 9 * Cin * max(Cout, 256)
    out = torch.empty((N, H, W, Cout), dtype=x.dtype, device=x.device)
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * N * H * W * 9 * Cin * Cout, "bytes": 2.0 * (x.numel() + out.numel()) + 2.0 * 9 * Cin * Cout,
                            "conv": (3, 1, Cin, Cout, H, W)}
    with _dev(x):
        call("agrl_conv3x3_packed_bn_act", ptr(x), ptr(packed), ptr(bias), ptr(out), N, H, W, Cin, Cout, 1 if relu else 0, _stream(x))
    return out


def conv1x1_packed_enabled():
    """AGRL_HIP_CONV1X1_PACKED=0 runs the layer-3 / layer-4 1x1 convs through conv_bn_act / conv1x1_dual (A/B; bit-identical)."""
    return switch_on('AGRL_HIP_CONV1X1_PACKED')


def conv1x1_packed_supported(w):
    """1x1 / stride 1 conv through the four-wave kernel with a pre-packed weight stream (csrc/conv1x1_fat.hip): 16-bit weights
    (Cout, 1, 1, K) or (Cout, K) with K % 128 == 0 and Cout % 256 == 0."""
    w2 = w.reshape(w.shape[0], -1) if w.dim() == 4 and tuple(w.shape[1:3]) == (1, 1) else w
    return w2.dtype == LP_DTYPE and w2.dim() == 2 and w2.shape[1] % 128 == 0 and w2.shape[0] % 256 == 0


def conv1x1_pack(w):
    """(Cout, K) 16-bit weights (K = K1 + K2 for the two-source form: [W1 | W2]) re-ordered once into per-wave MFMA fragment
    streams (agrl_conv1x1_pack) -> uint8 tensor."""
    assert conv1x1_packed_supported(w)
    w = w.reshape(w.shape[0], -1).contiguous()
    Cout, K = w.shape
    nbytes = int(_hip.lib().agrl_conv1x1_packed_bytes(K, Cout))
    assert nbytes == 2 * w.numel()
    packed = torch.empty((nbytes,), dtype=torch.uint8, device=w.device)
    with _dev(w):
        call("agrl_conv1x1_pack", ptr(w), ptr(packed), K, Cout, _stream(w))
    return packed


def conv1x1_packed(x, packed, bias, Cout, relu=True, x2=None, duo=False):
    """act([x | x2] @ W^T + bias) over pixel rows, weights from conv1x1_pack. vmgn.py:48-50 (conv1), :56-64 with x2 (conv3 +
    downsample conv of a first block as one GEMM: x = the block input, x2 = conv2's output). -> (N,H,W,Cout) 16-bit NHWC."""
    N, H, W, K1 = x.shape
    K2 = 0 if x2 is None else x2.shape[3]
    M = N * H * W
    assert x.dtype == LP_DTYPE and x.is_contiguous() and packed.numel() == 2 * (K1 + K2) * Cout
    assert x2 is None or (x2.dtype == x.dtype and x2.is_contiguous() and tuple(x2.shape[:3]) == (N, H, W))
    out = torch.empty((N, H, W, Cout), dtype=x.dtype, device=x.device)
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * M * (K1 + K2) * Cout,
                            "bytes": 2.0 * (x.numel() + (0 if x2 is None else x2.numel()) + out.numel()) + packed.numel(),
                            "conv": (1, 1, K1 + K2, Cout, H, W)}
    with _dev(x):
        if duo and x2 is not None:   # the two-workgroups-per-CU kernel (csrc/conv1x1_duo.hip), same packed weights, bit-identical
            call("agrl_conv1x1_packed_dual_duo", ptr(x), ptr(x2), ptr(packed), ptr(bias), ptr(out), M, K1, K2, Cout, 1 if relu else 0, _stream(x))
        else:
            call("agrl_conv1x1_packed_bn_act", ptr(x), ptr(x2), ptr(packed), ptr(bias), ptr(out), M, K1, K2, Cout, 1 if relu else 0, _stream(x))
    return out


def conv1x1_packed_dual_strided(x, x2, packed, bias, Cout, stride, relu=True):
    """relu(W3 x2 + b3 + Wds x[:, ::stride, ::stride] + bds) of the FIRST block of a strided layer (vmgn.py:56-64 with a stride-2
    downsample conv) as one GEMM over [x sampled | x2]: x (N,Hi,Wi,K1) the block input, x2 (N,Ho,Wo,K2) conv2's output, packed =
    conv1x1_pack([Wds | W3]), bias = bds + b3. The (N,Ho,Wo,Cout) shortcut map never exists. -> (N,Ho,Wo,Cout) 16-bit NHWC."""
    N, Hi, Wi, K1 = x.shape
    Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    K2 = x2.shape[3]
    assert x.dtype == LP_DTYPE and x2.dtype == LP_DTYPE and x.is_contiguous() and x2.is_contiguous()
    assert tuple(x2.shape[:3]) == (N, Ho, Wo) and packed.numel() == 2 * (K1 + K2) * Cout
    out = torch.empty((N, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
    M = N * Ho * Wo
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * M * (K1 + K2) * Cout, "bytes": 2.0 * (M * K1 + x2.numel() + out.numel()) + packed.numel(),
                            "conv": (1, 1, K1 + K2, Cout, Ho, Wo)}
    with _dev(x):
        call("agrl_conv1x1_packed_dual_strided", ptr(x), ptr(x2), ptr(packed), ptr(bias), ptr(out), N, Hi, Wi, stride, K1, K2, Cout,
             1 if relu else 0, _stream(x))
    return out


def conv1x1_duo_enabled():
    """AGRL_HIP_CONV1X1_DUO=0 runs the pool-fused last conv of a layer-4 branch through conv1x1_bn_act_pool and layer 4's conv1s through
    conv1x1_packed / conv_bn_act (A/B; bit-identical)."""
    return switch_on('AGRL_HIP_CONV1X1_DUO')


def conv1x1_packed_res(x, packed, bias, Cout, residual, relu=True):
    """act(x @ W^T + bias + residual) over pixel rows, weights from conv1x1_pack: conv3 / bn3 + identity shortcut + ReLU of a
    Bottleneck (vmgn.py:56-64), or with ``residual=None`` a plain conv1 / bn1 / relu (vmgn.py:48-50), through the two-workgroups-per-CU kernel (csrc/conv1x1_duo.hip; layer 4's conv1s are faster here back to back -- 2048 -> 512 69 us against conv1x1_fat_kernel's 73 -- but not inside the step, and stay where they were; with a residual 115.6 against conv_bn_act's 122.4 us back to back, 117 / 119.5 inside a block: also routed here, AGRL_HIP_CONV1X1_DUO_RES=0 = off). -> (N,H,W,Cout) 16-bit NHWC."""
    N, H, W, K = x.shape
    M = N * H * W
    assert x.dtype == LP_DTYPE and x.is_contiguous() and packed.numel() == 2 * K * Cout
    assert residual is None or (residual.dtype == x.dtype and residual.is_contiguous() and tuple(residual.shape) == (N, H, W, Cout))
    out = torch.empty((N, H, W, Cout), dtype=x.dtype, device=x.device)
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * M * K * Cout,
                            "bytes": 2.0 * (x.numel() + out.numel() * (2 if residual is not None else 1)) + packed.numel(),
                            "conv": (1, 1, K, Cout, H, W)}
    with _dev(x):
        call("agrl_conv1x1_packed_res_bn_act", ptr(x), ptr(packed), ptr(bias), ptr(residual), ptr(out), M, K, Cout, 1 if relu else 0,
             _stream(x))
    return out


def conv1x1_packed_res_pool(x, packed, bias, Cout, residual, splits, mean, want_lp, relu=True):
    """conv1x1_bn_act_pool through the two-workgroups-per-CU kernel (csrc/conv1x1_duo.hip: 9 % ahead of igemm_wide_kernel's pooled form) with weights from conv1x1_pack: last conv of a layer-4 branch,
    16 x 8 frames, the 2048-channel map never written. -> pooled fp32 (F, P, Cout) [, 16-bit copy]. vmgn.py:56-64 + :298-308."""
    N, H, W, K = x.shape
    assert x.dtype == LP_DTYPE and x.is_contiguous() and (H, W) == (16, 8) and packed.numel() == 2 * K * Cout
    assert residual is None or (residual.dtype == x.dtype and residual.is_contiguous() and tuple(residual.shape) == (N, H, W, Cout))
    P = int(sum(splits))
    pooled = torch.empty((N, P, Cout), dtype=torch.float32, device=x.device)
    pooled_lp = torch.empty((N, P, Cout), dtype=LP_DTYPE, device=x.device) if want_lp else None
    arr = (C.c_int * len(splits))(*[int(s) for s in splits])
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * N * H * W * Cout * K,
                            "bytes": 2.0 * (x.numel() + (residual.numel() if residual is not None else 0)) + packed.numel() + 4.0 * pooled.numel()}
    with _dev(x):
        call("agrl_conv1x1_packed_res_pool", ptr(x), ptr(packed), ptr(bias), ptr(residual), ptr(pooled), ptr(pooled_lp),
             N, H, W, K, Cout, 1 if relu else 0, arr, len(splits), 1 if mean else 0, _stream(x))
    return pooled, pooled_lp


def bottleneck_tail_supported(y2, w3, w1_next, shortcut_conv=None):
    """The fused conv3(+residual) -> next conv1 kernel exists for the layer-1 and layer-2 shapes in bf16. ``shortcut_conv`` =
    (weight, stride) of the block's downsample conv when the residual is to be computed in the same pass."""
    if (y2.dtype == LP_DTYPE and tuple(w3.shape) == (512, 1, 1, 128) and tuple(w1_next.shape) == (128, 1, 1, 512)
            and shortcut_conv is None and switch_on('AGRL_HIP_FUSE_TAIL_L2')):
        return True  # layer-2 form (weights resident in registers)
    ok = (y2.dtype == LP_DTYPE and tuple(w3.shape) == (256, 1, 1, 64)
          and tuple(w1_next.shape) in ((64, 1, 1, 256), (128, 1, 1, 256)))
    if shortcut_conv is not None:
        ok = ok and tuple(shortcut_conv[0].shape) == (256, 1, 1, 64) and shortcut_conv[1] == 1 and w1_next.shape[0] == 64
    return ok


def bottleneck_tail(y2, w3, b3, residual, w1_next, b1_next, shortcut=None):
    """out = relu(conv3(y2) + R), z = relu(conv1_next(out)) in one pass (out never re-read from HBM); R = ``residual``
    or, with ``shortcut`` = (x, w_ds, b_ds), the block's 1x1 stride-1 downsample conv of x computed in the same pass.
    vmgn.py:57-64 (block i) + :48-50 (block i+1). -> out (N,H,W,256), z (N,H,W,64 | 128) bf16 NHWC."""
    N, H, W, Cmid = y2.shape
    Cout, Cnext = w3.shape[0], w1_next.shape[0]
    assert y2.dtype == LP_DTYPE and (residual is None) != (shortcut is None)
    xs = ws = bs = None
    Cshort = 0
    if shortcut is not None:
        xs, ws, bs = shortcut
        assert xs.shape[:3] == y2.shape[:3] and xs.dtype == y2.dtype
        Cshort = xs.shape[3]
    else:
        assert residual.shape == (N, H, W, Cout) and residual.dtype == y2.dtype
    out = torch.empty((N, H, W, Cout), dtype=y2.dtype, device=y2.device)
    z = torch.empty((N, H, W, Cnext), dtype=y2.dtype, device=y2.device)
    M = N * H * W
    if _hip.PROFILE is not None:
        rd = (residual.numel() if residual is not None else xs.numel() + ws.numel())
        _hip.PROFILE_TAG = {"flops": 2.0 * M * ((Cmid + Cshort) * Cout + Cout * Cnext),
                            "bytes": 2.0 * (y2.numel() + rd + out.numel() + z.numel() + w3.numel() + w1_next.numel())}
    with _dev(y2):
        call("agrl_bottleneck_tail", ptr(y2), ptr(w3), ptr(b3), ptr(residual), ptr(xs), ptr(ws), ptr(bs), ptr(out),
             ptr(w1_next), ptr(b1_next), ptr(z), M, Cmid, Cout, Cnext, Cshort, _stream(y2))
    return out, z


def bottleneck_block_supported(z, w2, stride, w3, w1_next, shortcut_conv=None):
    """The fused 3x3 -> conv3(+shortcut) -> next conv1 kernel exists for the layer-1 shapes in bf16 (stride-1 3x3,
    8 x 8-divisible maps). ``shortcut_conv`` as in bottleneck_tail_supported."""
    if (not switch_on('AGRL_HIP_FUSE_BLOCK')):
        return False
    ok = (z.dtype == LP_DTYPE and stride == 1 and tuple(w2.shape) == (64, 3, 3, 64) and tuple(w3.shape) == (256, 1, 1, 64)
          and tuple(w1_next.shape) in ((64, 1, 1, 256), (128, 1, 1, 256)) and z.shape[1] % 8 == 0 and z.shape[2] % 8 == 0)
    if shortcut_conv is not None:
        ok = ok and tuple(shortcut_conv[0].shape) == (256, 1, 1, 64) and shortcut_conv[1] == 1 and w1_next.shape[0] == 64
    return ok


def bottleneck_block(z, w2, b2, w3, b3, residual, w1_next, b1_next, shortcut=None):
    """y2 = relu(conv3x3(z)), out = relu(conv3(y2) + R), z_next = relu(conv1_next(out)) in one pass: y2 never leaves the
    CU, out is written once. R = ``residual`` or, with ``shortcut`` = (x, w_ds, b_ds), the block's 1x1 stride-1
    downsample conv. vmgn.py:52-64 (block i) + :48-50 (block i+1). -> out (N,H,W,256), z_next (N,H,W,64 | 128)."""
    N, H, W, Cmid = z.shape
    Cout, Cnext = w3.shape[0], w1_next.shape[0]
    assert z.dtype == LP_DTYPE and (residual is None) != (shortcut is None)
    xs = ws = bs = None
    if shortcut is not None:
        xs, ws, bs = shortcut
        assert tuple(xs.shape) == (N, H, W, 64) and xs.dtype == z.dtype
    else:
        assert tuple(residual.shape) == (N, H, W, Cout) and residual.dtype == z.dtype
    out = torch.empty((N, H, W, Cout), dtype=z.dtype, device=z.device)
    zn = torch.empty((N, H, W, Cnext), dtype=z.dtype, device=z.device)
    M = N * H * W
    if _hip.PROFILE is not None:
        rd = residual.numel() if residual is not None else xs.numel() + ws.numel()
        _hip.PROFILE_TAG = {"flops": 2.0 * M * (9 * Cmid * Cmid + (Cmid + (64 if shortcut is not None else 0)) * Cout + Cout * Cnext),
                            "bytes": 2.0 * (z.numel() + rd + out.numel() + zn.numel() + w2.numel() + w3.numel() + w1_next.numel())}
    with _dev(z):
        call("agrl_bottleneck_block", ptr(z), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(residual), ptr(xs), ptr(ws), ptr(bs),
             ptr(out), ptr(w1_next), ptr(b1_next), ptr(zn), N, H, W, Cmid, Cout, Cnext, _stream(z))
    return out, zn


def linear_nobias(x, w):
    """(M,K) @ (N,K)^T -> fp32 (M,N). vmgn.py:148."""
    M, K = x.shape
    Nout, K2 = w.shape
    assert K == K2 and x.dtype == w.dtype
    y = torch.empty((M, Nout), dtype=torch.float32, device=x.device)
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * M * K * Nout, "bytes": x.element_size() * (x.numel() + w.numel()) + 4 * y.numel()}
    with _dev(x):
        call("agrl_linear_nobias", ptr(x), ptr(w), ptr(y), M, K, Nout, _gemm_code(x.dtype), _stream(x))
    return y


def part_pool(x4_1, x4_2, splits, want_lp):
    """-> gsum (F,C), nodes (F,P,C) fp32 [, nodes_lp bf16]. vmgn.py:298-308."""
    F_, h, w, Cc = x4_1.shape
    assert x4_2.shape == x4_1.shape and x4_1.dtype == x4_2.dtype
    P = int(sum(splits))
    gsum = torch.empty((F_, Cc), dtype=torch.float32, device=x4_1.device)
    nodes = torch.empty((F_, P, Cc), dtype=torch.float32, device=x4_1.device)
    nodes_lp = torch.empty((F_, P, Cc), dtype=LP_DTYPE, device=x4_1.device) if want_lp else None
    arr = (C.c_int * len(splits))(*[int(s) for s in splits])
    with _dev(x4_1):
        call("agrl_part_pool", ptr(x4_1), ptr(x4_2), ptr(gsum), ptr(nodes), ptr(nodes_lp), F_, h, w, Cc, arr,
             len(splits), dtype_code(x4_1.dtype), _stream(x4_1))
    return gsum, nodes, nodes_lp


GRAM_CSLICE = 128


def graph_matrix(f, adj, use_pose, learn_graph, mask_diag=False):
    """f (B,V,C) fp32, adj (B,V,V) fp32 -> G (B,V,V). vmgn.py:114-120, :155-166; ``mask_diag``: ganet.py:259-268."""
    B, V, Cc = f.shape
    G = torch.empty((B, V, V), dtype=torch.float32, device=f.device)
    gram = None
    nz = 0
    with _dev(f):
        if learn_graph:
            nz = Cc // GRAM_CSLICE
            gram = torch.empty((B, nz, V, V), dtype=torch.float32, device=f.device)
            call("agrl_graph_gram", ptr(f), ptr(gram), B, V, Cc, GRAM_CSLICE, _stream(f))
        if use_pose and adjacency_is_packed(adj):
            assert tuple(adj.shape) == (B, V, (V + 31) // 32)
            call("agrl_graph_finalize_bits", ptr(gram), nz, ptr(adj.contiguous()), ptr(G), B, V, 1, 1 if learn_graph else 0,
                 1 if mask_diag else 0, _stream(f))
            return G
        if use_pose:
            assert adj is not None and tuple(adj.shape) == (B, V, V) and adj.dtype == torch.float32
            adj = adj.contiguous()
        call("agrl_graph_finalize", ptr(gram), nz, ptr(adj) if use_pose else None, ptr(G), B, V,
             1 if use_pose else 0, 1 if learn_graph else 0, 1 if mask_diag else 0, _stream(f))
    return G


def adjacency_is_packed(adj):
    """The bit-packed adjacency is an int32 tensor (B, V, ceil(V/32)); the reference's form is fp32 (B, V, V)."""
    return adj is not None and adj.dtype == torch.int32


def adjacency_pack(adj):
    """fp32 {0,1} adjacency (B,V,V) on the device -> bit-packed int32 (B, V, ceil(V/32)): bit j & 31 of word j >> 5 of row i."""
    B, V, V2 = adj.shape
    assert V == V2 and adj.dtype == torch.float32
    bits = torch.empty((B, V, (V + 31) // 32), dtype=torch.int32, device=adj.device)
    with _dev(adj):
        call("agrl_adjacency_pack", ptr(adj.contiguous()), ptr(bits), B, V, _stream(adj))
    return bits


def adjacency_pack_host(adj):
    """The same packing on the HOST (numpy): what a loader does before the upload -- 448 bytes per 56-node tracklet cross PCIe
    instead of 12.5 KB. adj: CPU tensor / array (B,V,V) -> CPU int32 tensor (B, V, ceil(V/32))."""
    import numpy as np
    a = np.asarray(adj.detach().cpu() if hasattr(adj, "detach") else adj) != 0
    B, V, _ = a.shape
    W = (V + 31) // 32
    pad = np.zeros((B, V, W * 32), dtype=bool)
    pad[:, :, :V] = a
    words = (pad.reshape(B, V, W, 32).astype(np.uint64) << np.arange(32, dtype=np.uint64)).sum(axis=3).astype(np.uint32)
    return torch.from_numpy(words.view(np.int32).copy())


def graph_propagate(f, h, G, bn_scale, bn_shift, gamma, slope, want_lp, keep=None):
    """out = keep f + gamma lrelu(bn(G h)); keep defaults to the reference's Python-float (1 - gamma). vmgn.py:168-172;
    ganet.py:278-283 passes keep = 1."""
    if keep is None:
        keep = 1.0 - float(gamma)
    B, V, Cc = f.shape
    out = torch.empty_like(f)
    out_lp = torch.empty((B, V, Cc), dtype=LP_DTYPE, device=f.device) if want_lp else None
    if _hip.PROFILE is not None:  # read f + read h + read G(adj-sized) + write out
        _hip.PROFILE_TAG = {"flops": 2.0 * B * V * V * Cc, "bytes": 4.0 * (3 * B * V * Cc + B * V * V)}
    with _dev(f):
        call("agrl_graph_propagate", ptr(f), ptr(h), ptr(G), ptr(bn_scale), ptr(bn_shift), float(keep), float(gamma),
             float(slope), ptr(out), ptr(out_lp), B, V, Cc, _stream(f))
    return out, out_lp


def graph_apply_presplit_supported(f):
    B, V, Cc = f.shape
    return V <= 64 and V % 4 == 0 and Cc % 128 == 0


def graph_apply_operand(G, f, out_dtype, presplit=False):
    """P = G f, (B,V,V) x (B,V,C) fp32 -> (B,V,C) in ``out_dtype`` (fp32 / bf16): the message pass applied to the layer INPUT,
    written once as the operand of ``graph_linear_mix``. vmgn.py:168 with the Linear commuted behind it. ``presplit`` ('fp16x3'): the
    fp32-sized rows hold the fp16 halves the GEMM's k-loop would form (graph_linear_mix(..., p_presplit=True))."""
    B, V, Cc = f.shape
    assert f.dtype == torch.float32 and G.dtype == torch.float32 and tuple(G.shape) == (B, V, V)
    assert not presplit or (out_dtype == torch.float32 and graph_apply_presplit_supported(f))
    if V <= 64 and V % 4 == 0 and Cc % 128 == 0:
        out = torch.empty((B, V, Cc), dtype=out_dtype, device=f.device)
        if _hip.PROFILE is not None:
            _hip.PROFILE_TAG = {"flops": 2.0 * B * V * V * Cc, "bytes": 4.0 * (B * V * Cc + B * V * V) + out.element_size() * B * V * Cc}
        with _dev(f):
            call("agrl_graph_apply", ptr(G.contiguous()), ptr(f.contiguous()), ptr(out), _hip.F32H3P if presplit else dtype_code(out_dtype), B, V, Cc,
                 _stream(f))
        return out
    # other node counts (V > 64: seq_len 16; V % 4 != 0): the general message-pass kernels with a unit BatchNorm
    key = (f.device, Cc)
    if key not in _UNIT:
        _UNIT[key] = (torch.ones((Cc,), dtype=torch.float32, device=f.device), torch.zeros((Cc,), dtype=torch.float32, device=f.device))
    one, zero = _UNIT[key]
    out, out_lp = graph_propagate(f, f, G, one, zero, 1.0, 1.0, want_lp=out_dtype == LP_DTYPE, keep=0.0)
    return out_lp if out_dtype == LP_DTYPE else out


TRACKLET_FORM_MIN_B = int(os.environ.get('AGRL_HIP_GCN_TRACKLET_MIN_B', '224'))


def graph_tracklet_operand_supported(f):
    """One workgroup per tracklet needs enough tracklets to fill the chip (B >= 224 by default: measured 90 us against 119 us for the three launches at 256 tracklets, slower below ~190) and the streaming shapes."""
    B, V, Cc = f.shape
    return B >= TRACKLET_FORM_MIN_B and V <= 64 and V % 4 == 0 and Cc % 512 == 0


def graph_tracklet_operand(f, adj, use_pose, learn_graph, out_dtype, want_graph=False, mask_diag=False):
    """graph_matrix + graph_apply_operand in one launch, one workgroup per tracklet: -> P = G f (B,V,C) in ``out_dtype``, G (B,V,V) or
    None. vmgn.py:114-120, :155-168."""
    B, V, Cc = f.shape
    assert f.dtype == torch.float32
    P = torch.empty((B, V, Cc), dtype=out_dtype, device=f.device)
    G = torch.empty((B, V, V), dtype=torch.float32, device=f.device) if want_graph else None
    packed = use_pose and adjacency_is_packed(adj)
    if use_pose:
        assert adj is not None and (tuple(adj.shape) == (B, V, (V + 31) // 32) if packed else (tuple(adj.shape) == (B, V, V) and adj.dtype == torch.float32))
        adj = adj.contiguous()
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 4.0 * B * V * V * Cc, "bytes": 4.0 * (B * V * Cc + B * V * V) + P.element_size() * B * V * Cc}
    with _dev(f):
        call("agrl_graph_tracklet_operand", ptr(f.contiguous()), ptr(adj) if use_pose else None, 1 if packed else 0, ptr(G), ptr(P),
             dtype_code(out_dtype), B, V, Cc, 1 if use_pose else 0, 1 if learn_graph else 0, 1 if mask_diag else 0, _stream(f))
    return P, G


def graph_linear_mix(p_op, w, f, bn_scale, bn_shift, gamma, slope, keep=None, p_presplit=False):
    """out = keep f + gamma lrelu(bn((G f) W^T)): the Linear of a GraphLayer as ONE GEMM over P = G f with BatchNorm1d, LeakyReLU
    and the residual mix in its epilogue (vmgn.py:148, :168-172). p_op (B,V,K) fp32 / bf16, w (N,K) same dtype, f (B,V,N) fp32."""
    if keep is None:
        keep = 1.0 - float(gamma)
    B, V, K = p_op.shape
    Nout = w.shape[0]
    assert w.shape[1] == K and p_op.dtype == w.dtype and f.dtype == torch.float32 and tuple(f.shape) == (B, V, Nout)
    out = torch.empty_like(f)
    if _hip.PROFILE is not None:
        _hip.PROFILE_TAG = {"flops": 2.0 * B * V * K * Nout, "bytes": p_op.element_size() * (p_op.numel() + w.numel()) + 8.0 * f.numel()}
    code = _gemm_code(p_op.dtype)
    unscale = getattr(w, 'agrl_unscale', None)
    if unscale is not None:   # 'fp16x3': w pre-scaled by a power of two and pre-split (split16_inloop_weights); bn_scale must already carry the 2^-k
        assert p_op.dtype == torch.float32 and getattr(bn_scale, 'agrl_folded_unscale', None) == unscale and getattr(w, 'agrl_presplit', False)
        code = _hip.F32H3P if p_presplit else _hip.F32H3
    else:
        assert not p_presplit
    with _dev(f):
        call("agrl_graph_linear_mix", ptr(p_op.contiguous()), ptr(w), ptr(f.contiguous()), ptr(bn_scale), ptr(bn_shift), float(keep), float(gamma),
             float(slope), ptr(out), B * V, K, Nout, code, _stream(f))
    return out


def pam_pool(x, qk, splits):
    """ganet's position-attention part nodes, first half: x (F,h,w,C) NHWC, qk (F,h,w,2*Cq) NHWC stacked query / key conv
    output (None when the module's gamma is 0) -> xbar (F,P,C) fp32 (None without qk), xmean (F,P,C) fp32. ganet.py:98-136,
    :384-400."""
    F_, h, w, Cc = x.shape
    P = int(sum(splits))
    xmean = torch.empty((F_, P, Cc), dtype=torch.float32, device=x.device)
    xbar = None
    Cq = 0
    if qk is not None:
        assert qk.dtype == x.dtype and tuple(qk.shape[:3]) == (F_, h, w) and qk.shape[3] % 2 == 0
        Cq = qk.shape[3] // 2
        xbar = torch.empty((F_, P, Cc), dtype=torch.float32, device=x.device)
    arr = (C.c_int * len(splits))(*[int(s) for s in splits])
    with _dev(x):
        call("agrl_pam_pool", ptr(x), ptr(qk), ptr(xbar), ptr(xmean), F_, h, w, Cc, Cq, arr, len(splits), dtype_code(x.dtype),
             _stream(x))
    return xbar, xmean


def pam_combine(y, bv, xmean, gamma, want_lp):
    """nodes = gamma (y + bv) + 2 xmean (y = Wv xbar), ganet.py:394-399 -> nodes fp32 like xmean [, bf16 copy]."""
    nodes = torch.empty_like(xmean)
    nodes_lp = torch.empty(xmean.shape, dtype=LP_DTYPE, device=xmean.device) if want_lp else None
    rows, Cc = xmean.numel() // xmean.shape[-1], xmean.shape[-1]
    with _dev(xmean):
        call("agrl_pam_combine", ptr(y), ptr(bv), ptr(xmean), float(gamma), ptr(nodes), ptr(nodes_lp), rows, Cc, _stream(xmean))
    return nodes, nodes_lp


def clip_pool(feats, num_clips, mode="avg"):
    """(T*num_clips, D) fp32 -> (T, D): mean / max over each tracklet's clips. train_vidreid_xent_htri.py:471-476."""
    assert feats.dtype == torch.float32 and feats.dim() == 2 and feats.size(0) % num_clips == 0 and mode in ("avg", "max")
    feats = feats.contiguous()
    T, D = feats.size(0) // num_clips, feats.size(1)
    out = torch.empty((T, D), dtype=torch.float32, device=feats.device)
    with _dev(feats):
        call("agrl_clip_pool", ptr(feats), ptr(out), T, num_clips, D, 0 if mode == "avg" else 1, _stream(feats))
    return out


def row_sqnorm(x):
    R, Cc = x.shape
    out = torch.empty((R,), dtype=torch.float32, device=x.device)
    with _dev(x):
        call("agrl_row_sqnorm", ptr(x), ptr(out), R, Cc, dtype_code(x.dtype), _stream(x))
    return out


def attn_pool_bnneck(nodes, sqn, gsum, g_scale, g_shift, a_scale, a_shift, B, S, P, hw, want_feats=False):
    """-> out (B,2C) [, g_f, att_f (B,C)]. vmgn.py:270-278, :299-301, :313-321."""
    Cc = nodes.shape[-1]
    out = torch.empty((B, 2 * Cc), dtype=torch.float32, device=nodes.device)
    g_f = torch.empty((B, Cc), dtype=torch.float32, device=nodes.device) if want_feats else None
    att_f = torch.empty((B, Cc), dtype=torch.float32, device=nodes.device) if want_feats else None
    with _dev(nodes):
        call("agrl_attn_pool_bnneck", ptr(nodes), ptr(sqn), ptr(gsum), ptr(g_scale), ptr(g_shift), ptr(a_scale),
             ptr(a_shift), ptr(out), ptr(g_f), ptr(att_f), B, S, P, Cc, hw, _stream(nodes))
    if want_feats:
        return out, g_f, att_f
    return out


def attn_tail_supported(S, P, Cc, B=None):
    """The one-launch tail (agrl_attn_tail): C % 4 == 0 and S * P + 2 C floats within 64 KB of LDS. With ``B`` given: also whether the
    model should TAKE it -- one workgroup per tracklet fills the chip only from ~224 tracklets per GPU (measured, tools/
    attn_tail_bench.py, one launch / three launches: 37.6 / 35.7 us at 32 tracklets, 43.6 / 40.2 at 128, 53.7 / 58.2 at 256);
    AGRL_HIP_FUSE_ATTN_TAIL=1 / 0 forces it on / off."""
    ok = Cc % 4 == 0 and (((S * P + 3) & ~3) + 2 * Cc + 4) * 4 <= 64 * 1024
    if B is None or not ok:
        return ok
    force = switch('AGRL_HIP_FUSE_ATTN_TAIL', '')
    if force != '':
        return force != '0'
    return B >= int(switch('AGRL_HIP_ATTN_TAIL_MIN_B', '224'))


def attn_tail(nodes, gsum, g_scale, g_shift, a_scale, a_shift, B, S, P, hw, want_feats=False, query_dtype=None, want_node_sqn=False):
    """row_sqnorm(nodes) + attn_pool_bnneck + the distance matrix's query operand in ONE launch (bit-identical to the separate
    calls). -> out (B,2C), feats (g_f, att_f) or None, query dict or None: {'sqn': ||out||^2 (B,), 'normalized': out / ||out||
    in ``query_dtype`` (LP_DTYPE or torch.float32)}, node_sqn (B*S*P,) or None. vmgn.py:270-278, :313-321; distance.py:70-71, :86-87."""
    Cc = nodes.shape[-1]
    dev = nodes.device
    out = torch.empty((B, 2 * Cc), dtype=torch.float32, device=dev)
    g_f = torch.empty((B, Cc), dtype=torch.float32, device=dev) if want_feats else None
    att_f = torch.empty((B, Cc), dtype=torch.float32, device=dev) if want_feats else None
    node_sqn = torch.empty((B * S * P,), dtype=torch.float32, device=dev) if want_node_sqn else None
    q_lp = q_f32 = out_sqn = None
    if query_dtype is not None:
        out_sqn = torch.empty((B,), dtype=torch.float32, device=dev)
        if query_dtype == LP_DTYPE:
            q_lp = torch.empty((B, 2 * Cc), dtype=LP_DTYPE, device=dev)
        else:
            q_f32 = torch.empty((B, 2 * Cc), dtype=torch.float32, device=dev)
    with _dev(nodes):
        call("agrl_attn_tail", ptr(nodes), ptr(gsum), ptr(g_scale), ptr(g_shift), ptr(a_scale), ptr(a_shift), ptr(out), ptr(g_f),
             ptr(att_f), ptr(node_sqn), ptr(out_sqn), ptr(q_lp), ptr(q_f32), B, S, P, Cc, hw, _stream(nodes))
    query = None if query_dtype is None else {'sqn': out_sqn, 'normalized': q_lp if q_lp is not None else q_f32}
    return out, ((g_f, att_f) if want_feats else None), query, node_sqn


class QueryOperandCache:
    """What agrl_attn_tail left beside an embedding batch: its rows' squared norms and L2-normalised copy in the distance matrix's
    operand type. ``lookup(emb, dtype)`` answers only for THE tensor object the forward returned -- identity through a weak
    reference, not its address: once that tensor is freed the caching allocator may hand the same address to another (B, D) fp32
    tensor whose ``_version`` is 0 as well (raw kernels never bump it), and an address-keyed cache would then serve the previous
    batch's rows. The version is still compared so that an in-place edit of the live tensor misses."""

    def __init__(self, emb, query):
        self.ref = weakref.ref(emb)
        self.version = emb._version
        self.query = query

    def lookup(self, emb, dtype):
        if self.ref() is not emb or self.version != emb._version or self.query['normalized'].dtype != dtype:
            return None
        return self.query


def query_operands(model, emb, metric, dtype):
    """The query side of hip_distmat for one batch of embeddings ``emb`` (B,D) fp32 -> (operand, sq-norms or None): the cosine
    operand / the euclidean norms straight from the tail kernel when ``emb`` is the tensor ``model`` just returned (no extra
    launch), else agrl_row_l2_normalize / agrl_row_sqnorm. distance.py:59-89."""
    cache = getattr(model, '_hip_query', None)
    q = cache.lookup(emb, dtype) if cache is not None else None
    if metric == 'cosine':
        return (q['normalized'] if q is not None else row_l2_normalize(emb, True, dtype)), None
    qn = q['sqn'] if q is not None else row_sqnorm(emb)
    return (row_l2_normalize(emb, False, dtype) if dtype != torch.float32 else emb), qn


def k_multiple(dtype):
    """K granularity of the GEMM kernel: one 128-byte k-tile."""
    return 64 if dtype == LP_DTYPE else 32


def row_l2_normalize(x, normalize, out_dtype, pad_to=1):
    """fp32 (R,C) -> (R, Cpad) out_dtype, Cpad = C rounded up to ``pad_to``, padding columns zero."""
    R, Cc = x.shape
    assert x.dtype == torch.float32
    ld = -(-Cc // pad_to) * pad_to
    y = torch.empty((R, ld), dtype=out_dtype, device=x.device)
    with _dev(x):
        call("agrl_row_l2_normalize", ptr(x), ptr(y), R, Cc, ld, 1 if normalize else 0, dtype_code(out_dtype),
             _stream(x))
    return y


def distmat(q, g, metric, qn=None, gn=None, out=None):
    """q (m,D), g (n,D) prepared operands (see metrics.distance) -> fp32 (m,n). distance.py:59-89."""
    m, D = q.shape
    n, D2 = g.shape
    assert D == D2 and q.dtype == g.dtype
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=q.device)
    assert out.stride(1) == 1 and out.dtype == torch.float32
    code = METRIC_EUCLIDEAN if metric == "euclidean" else METRIC_COSINE
    if _hip.PROFILE is not None:  # (m+n)*D*e + m*n*4
        _hip.PROFILE_TAG = {"flops": 2.0 * m * n * D, "bytes": q.element_size() * (m + n) * D + 4.0 * m * n}
    ws = None
    # few output tiles (one eval batch against a long gallery, or the gathered queries of an 8-GPU step against a gallery
    # shard): hand the kernel scratch for split-K partials; the m <= 64 streaming kernels ignore it
    if (-(-m // 64)) * (-(-n // 128)) < 256:
        ws = torch.empty((8 * m * n,), dtype=torch.float32, device=q.device)
    with _dev(q):
        _hip.call("agrl_distmat", ptr(q), ptr(g), ptr(qn), ptr(gn), out.data_ptr(), m, n, D, out.stride(0), code,
                  dtype_code(q.dtype), ptr(ws), 0 if ws is None else ws.numel() * 4, _stream(q))
    return out


def rank_topk(dist, k, idx_offset=0):
    """dist (m,n) fp32 -> idx int32 (m,k), val fp32 (m,k), ascending (distance, index). rank.py:170-172."""
    m, n = dist.shape
    assert dist.dtype == torch.float32 and dist.stride(1) == 1
    idx = torch.empty((m, k), dtype=torch.int32, device=dist.device)
    val = torch.empty((m, k), dtype=torch.float32, device=dist.device)
    with _dev(dist):
        _hip.call("agrl_rank_topk", dist.data_ptr(), m, n, dist.stride(0), k, idx_offset, ptr(idx), ptr(val),
                  _stream(dist))
    return idx, val


def distmat_topk(q, g, metric, k, qn=None, gn=None, idx_offset=0, workspace_bytes=None):
    """q (m,D), g (n,D) prepared operands (as for ``distmat``) -> idx int32 (m,k), val fp32 (m,k): the k nearest gallery rows
    of every query in ascending (distance, index) order, without the (m,n) matrix (distance.py:59-89 + rank.py:171-172).
    Bit-identical to ``rank_topk(distmat(q, g, ...), k)`` when ONE query block covers m (the default workspace at the MARS sizes);
    with several blocks (a small ``workspace_bytes``, very many queries) a block's row count picks the GEMM's kernel family and
    split-K, so a distance can differ from the full-matrix path in its last bit and near-ties may swap: equal up to that
    (tests/test_gpu_kernels.py::test_distmat_topk_blocks_tie_aware)."""
    m, D = q.shape
    n, D2 = g.shape
    assert D == D2 and q.dtype == g.dtype
    code = METRIC_EUCLIDEAN if metric == "euclidean" else METRIC_COSINE
    nbytes = int(_hip.lib().agrl_distmat_topk_workspace(m, n)) if workspace_bytes is None else int(workspace_bytes)
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=q.device)
    rows = max(1, min(m, nbytes // (4 * (-(-n // 4) * 4))))
    gws = None
    if (-(-rows // 64)) * (-(-n // 128)) < 256:   # few output tiles per block: split-K scratch, as in ``distmat``
        gws = torch.empty((8 * rows * n,), dtype=torch.float32, device=q.device)
    idx = torch.empty((m, k), dtype=torch.int32, device=q.device)
    val = torch.empty((m, k), dtype=torch.float32, device=q.device)
    if _hip.PROFILE is not None:  # (m+n) D e in, m k 8 out
        _hip.PROFILE_TAG = {"flops": 2.0 * m * n * D, "bytes": q.element_size() * (m + n) * D + 8.0 * m * k}
    with _dev(q):
        _hip.call("agrl_distmat_topk", ptr(q), ptr(g), ptr(qn), ptr(gn), m, n, D, code, dtype_code(q.dtype), int(k), int(idx_offset),
                  ptr(idx), ptr(val), ptr(ws), ws.numel() * 4, ptr(gws), 0 if gws is None else gws.numel() * 4, _stream(q))
    return idx, val


RANK_ARGSORT_MAX_N = 16384


def rank_argsort(dist):
    """dist (m,n) fp32, n <= 16384 -> int32 (m,n): the stable ascending order of every row (NaN last). rank.py:45-47."""
    m, n = dist.shape
    assert dist.dtype == torch.float32 and dist.stride(1) == 1 and n <= RANK_ARGSORT_MAX_N
    idx = torch.empty((m, n), dtype=torch.int32, device=dist.device)
    with _dev(dist):
        _hip.call("agrl_rank_argsort", dist.data_ptr(), m, n, dist.stride(0), ptr(idx), _stream(dist))
    return idx


def rank_mars(topk_idx, q_pids, q_camids, g_pids, g_camids):
    """-> ap fp64 (m), cmc fp32 (m,k). rank.py:160-212."""
    m, k = topk_idx.shape
    n = g_pids.numel()
    ap = torch.empty((m,), dtype=torch.float64, device=topk_idx.device)
    cmc = torch.empty((m, k), dtype=torch.float32, device=topk_idx.device)
    for t in (topk_idx, q_pids, q_camids, g_pids, g_camids):
        assert t.dtype == torch.int32
    with _dev(topk_idx):
        call("agrl_rank_mars", ptr(topk_idx), ptr(q_pids), ptr(q_camids), ptr(g_pids), ptr(g_camids), m, n, k, ptr(ap),
             ptr(cmc), _stream(topk_idx))
    return ap, cmc


def pose_adjacency(poses, detected, height, num_split=4, pyramid_part=True, threshold=0.1, packed=False):
    """AlphaPose keypoints (B,S,18,3) fp32 + per-frame detection flags (B,S) -> adjacency (B,V,V) fp32 on the device, or with
    ``packed`` the bit-packed int32 (B, V, ceil(V/32)) form the graph kernels also take. dataset_loader.py:218-388
    (generate_graph + adj_graph)."""
    B, S = poses.shape[:2]
    assert poses.dtype == torch.float32 and tuple(poses.shape[2:]) == (18, 3)
    det = detected.to(torch.uint8).contiguous()
    assert tuple(det.shape) == (B, S)
    P = 2 * num_split - 1 if pyramid_part else num_split
    if packed:
        V = S * P
        bits = torch.empty((B, V, (V + 31) // 32), dtype=torch.int32, device=poses.device)
        with _dev(poses):
            call("agrl_pose_adjacency_bits", ptr(poses.contiguous()), ptr(det), ptr(bits), B, S, int(num_split), 1 if pyramid_part else 0,
                 float(height), float(threshold), _stream(poses))
        return bits
    adj = torch.empty((B, S * P, S * P), dtype=torch.float32, device=poses.device)
    with _dev(poses):
        call("agrl_pose_adjacency", ptr(poses.contiguous()), ptr(det), ptr(adj), B, S, int(num_split), 1 if pyramid_part else 0,
             float(height), float(threshold), _stream(poses))
    return adj


def rank_market1501(dist, q_pids, q_camids, g_pids, g_camids, max_rank):
    """-> ap fp64 (m) (NaN when invalid), cmc fp32 (m,max_rank), valid int32 (m). rank.py:95-150."""
    m, n = dist.shape
    assert dist.dtype == torch.float32 and dist.stride(1) == 1
    for t in (q_pids, q_camids, g_pids, g_camids):
        assert t.dtype == torch.int32
    ap = torch.empty((m,), dtype=torch.float64, device=dist.device)
    cmc = torch.empty((m, max_rank), dtype=torch.float32, device=dist.device)
    valid = torch.empty((m,), dtype=torch.int32, device=dist.device)
    with _dev(dist):
        _hip.call("agrl_rank_market1501", dist.data_ptr(), m, n, dist.stride(0), ptr(q_pids), ptr(q_camids), ptr(g_pids),
                  ptr(g_camids), max_rank, ptr(ap), ptr(cmc), ptr(valid), _stream(dist))
    return ap, cmc, valid


def re_ranking(q_g, q_q, g_g, k1=20, k2=6, lambda_value=0.3):
    """k-reciprocal re-ranking on the device: fp32 CUDA (m,n), (m,m), (n,n) -> fp32 (m,n). utils/re_ranking.py:30-95."""
    m, n = q_g.shape
    for t, shp in ((q_g, (m, n)), (q_q, (m, m)), (g_g, (n, n))):
        assert t.dtype == torch.float32 and tuple(t.shape) == shp
    q_g, q_q, g_g = q_g.contiguous(), q_q.contiguous(), g_g.contiguous()
    out = torch.empty((m, n), dtype=torch.float32, device=q_g.device)
    nbytes = int(_hip.lib().agrl_re_ranking_workspace(m, n, int(k1)))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=q_g.device)
    with _dev(q_g):
        call("agrl_re_ranking", ptr(q_g), ptr(q_q), ptr(g_g), m, n, int(k1), int(k2), float(lambda_value), ptr(out), n,
             ptr(ws), nbytes, _stream(q_g))
    return out


def triplet_hard_mine(x, pids):
    """x (n,d) fp32, pids int32 (n) -> dist_ap, dist_an fp32 (n), idx_ap, idx_an int32 (n)."""
    n, d = x.shape
    assert x.dtype == torch.float32 and pids.dtype == torch.int32
    dap = torch.empty((n,), dtype=torch.float32, device=x.device)
    dan = torch.empty((n,), dtype=torch.float32, device=x.device)
    iap = torch.empty((n,), dtype=torch.int32, device=x.device)
    ian = torch.empty((n,), dtype=torch.int32, device=x.device)
    with _dev(x):
        call("agrl_triplet_hard_mine", ptr(x), ptr(pids), n, d, ptr(dap), ptr(dan), ptr(iap), ptr(ian), _stream(x))
    return dap, dan, iap, ian


def read_stream(buf, nbytes=None, workgroups=2048):
    """Measurement yardstick (bench.py): one pure read pass over the first ``nbytes`` of ``buf`` on the current stream."""
    total = buf.numel() * buf.element_size()
    nbytes = total if nbytes is None else min(int(nbytes), total)
    sink = torch.empty((1,), dtype=torch.float32, device=buf.device)
    with _dev(buf):
        call("agrl_diag_read_stream", ptr(buf), nbytes & ~15, ptr(sink), int(workgroups), _stream(buf))
    return sink


# ---- train step of the conv trunk (include/agrl_hip.h, "train step" section) ----------------------------------------------
def _bn_ws(M, Cc, device):
    nbytes = int(_hip.lib().agrl_bn_workspace(int(M), int(Cc)))
    return torch.empty((nbytes // 8,), dtype=torch.float64, device=device), nbytes


def bn_stats(y2d):
    """(M,C) fp32 -> mean (C), biased var (C): BatchNorm2d batch statistics. vmgn.py:49-61 under model.train()."""
    M, Cc = y2d.shape
    assert y2d.dtype == torch.float32 and y2d.is_contiguous()
    mean = torch.empty((Cc,), dtype=torch.float32, device=y2d.device)
    var = torch.empty((Cc,), dtype=torch.float32, device=y2d.device)
    ws, nbytes = _bn_ws(M, Cc, y2d.device)
    with _dev(y2d):
        call("agrl_bn_stats", ptr(y2d), ptr(mean), ptr(var), M, Cc, ptr(ws), nbytes, _stream(y2d))
    return mean, var


def conv_stats(x, w_ohwi, stride, pad):
    """fp32 NHWC conv (no bias / activation) + the batch statistics of its output: -> y (N,OH,OW,Cout), mean (Cout), var (Cout,
    biased). The per-tile sums come out of the conv's epilogue; nothing re-reads y."""
    N, H, W, Cin = x.shape
    Cout, R, S, Cin2 = w_ohwi.shape
    assert Cin == Cin2 and x.dtype == torch.float32 and w_ohwi.dtype == torch.float32
    OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    M = N * OH * OW
    rows = -(-M // 64)
    out = torch.empty((N, OH, OW, Cout), dtype=torch.float32, device=x.device)
    partial = torch.empty((rows * 2 * Cout,), dtype=torch.float32, device=x.device)
    mean = torch.empty((Cout,), dtype=torch.float32, device=x.device)
    var = torch.empty((Cout,), dtype=torch.float32, device=x.device)
    with _dev(x):
        call("agrl_conv2d_stats", ptr(x), ptr(w_ohwi), ptr(out), ptr(partial), partial.numel() * 4, N, H, W, Cin, Cout, R, S, stride, pad,
             _gemm_code(torch.float32), _stream(x))
        ws, nbytes = _bn_ws(rows, 2 * Cout, x.device)
        call("agrl_bn_stats_from_partials", ptr(partial), rows, Cout, M, ptr(mean), ptr(var), ptr(ws), nbytes, _stream(x))
    return out, mean, var


def bn_fold_train(mean, var, gamma, beta, eps, momentum, n, running_mean=None, running_var=None, num_batches_tracked=None):
    """-> scale, shift, invstd (C): the folded BatchNorm of the batch statistics, and -- in the same launch -- the running-statistics
    update nn.BatchNorm applies in train mode (momentum, unbiased variance, num_batches_tracked += 1). vmgn.py:49-63, :169."""
    Cc = mean.numel()
    scale, shift, invstd = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
    if num_batches_tracked is not None:
        assert num_batches_tracked.dtype == torch.int64
    with _dev(mean):
        call("agrl_bn_fold_train", ptr(mean), ptr(var), ptr(gamma), ptr(beta), float(eps), float(momentum), int(n), ptr(running_mean),
             ptr(running_var), ptr(num_batches_tracked), ptr(scale), ptr(shift), ptr(invstd), Cc, _stream(mean))
    return scale, shift, invstd


def bn_apply(y2d, scale, shift, residual, relu, slope=0.0, want_mask=False):
    """relu: activation on; slope 0 = ReLU, > 0 = LeakyReLU(slope). -> out, mask: with ``want_mask`` (and relu) the sign bits of
    the pre-activation, one bit per element (uint8, M*C/8 bytes) -- what ``bn_backward`` needs instead of the output."""
    M, Cc = y2d.shape
    out = torch.empty_like(y2d)
    mask = torch.empty(((M * Cc + 7) // 8,), dtype=torch.uint8, device=y2d.device) if (want_mask and relu) else None
    with _dev(y2d):
        call("agrl_bn_apply", ptr(y2d), ptr(scale), ptr(shift), ptr(residual), ptr(out), ptr(mask), M, Cc, 1 if relu else 0, float(slope),
             _stream(y2d))
    return out, mask


def bn_backward(dout, out, y2d, mean, invstd, gamma, relu, want_dz, slope=0.0, mask=None):
    """-> dy (M,C), dz (M,C) or None, dgamma (C), dbeta (C). With relu either the forward output ``out`` or the sign ``mask``
    of ``bn_apply``."""
    M, Cc = y2d.shape
    dy = torch.empty_like(y2d)
    dz = torch.empty_like(y2d) if want_dz else None
    dgamma = torch.empty((Cc,), dtype=torch.float32, device=y2d.device)
    dbeta = torch.empty((Cc,), dtype=torch.float32, device=y2d.device)
    ws, nbytes = _bn_ws(M, Cc, y2d.device)
    with _dev(y2d):
        call("agrl_bn_backward", ptr(dout), ptr(out) if (relu and mask is None) else None, ptr(mask) if relu else None, ptr(y2d), ptr(mean),
             ptr(invstd), ptr(gamma), 1 if relu else 0,
             float(slope), ptr(dy), ptr(dz), ptr(dgamma), ptr(dbeta), M, Cc, ptr(ws), nbytes, _stream(y2d))
    return dy, dz, dgamma, dbeta


def im2col_t(x, R, S, stride, pad):
    """x (F,H,W,C) fp32 NHWC -> (R*S*C, Mpad) fp32, Mpad = F*OH*OW rounded up to the GEMM's k-tile (zero columns):
    tap-expanded channel-major transpose (weight-gradient operand)."""
    F_, H, W, Cc = x.shape
    OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    ld = -(-(F_ * OH * OW) // 32) * 32
    T = torch.empty((R * S * Cc, ld), dtype=torch.float32, device=x.device)
    with _dev(x):
        call("agrl_im2col_t", ptr(x), ptr(T), ld, F_, H, W, Cc, R, S, stride, pad, _stream(x))
    return T


def gemm_nt_splitk(x, w):
    """(M,K) @ (N,K)^T -> fp32 (M,N), K split over workgroups (weight gradients)."""
    M, K = x.shape
    Nout, K2 = w.shape
    assert K == K2 and x.dtype == w.dtype
    y = torch.empty((M, Nout), dtype=torch.float32, device=x.device)
    # enough for the split the entry point will choose: it stops at >= 1024 workgroups, i.e. at most 1024 / tiles slices
    tiles = (-(-M // 64)) * (-(-Nout // (64 if Nout <= 64 else 128)))
    ks_cap = 1
    while ks_cap < 256 and tiles * ks_cap < 1024:
        ks_cap *= 2
    ws = torch.empty((ks_cap * M * Nout,), dtype=torch.float32, device=x.device)
    with _dev(x):
        call("agrl_gemm_nt_splitk", ptr(x), ptr(w), ptr(y), M, K, Nout, _gemm_code(x.dtype), ptr(ws), ws.numel() * 4, _stream(x))
    return y


def im2col_rows(x, R, S, stride, pad, ld, nchw=False):
    """Pixel-major patches of an NHWC (or NCHW) fp32 tensor: -> (F*OH*OW, ld), columns (r, s, c), zero padded to ``ld``."""
    if nchw:
        F_, Cc, H, W = x.shape
    else:
        F_, H, W, Cc = x.shape
    OH, OW = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    x = x.contiguous()
    P = torch.empty((F_ * OH * OW, ld), dtype=torch.float32, device=x.device)
    with _dev(x):
        call("agrl_im2col_rows", ptr(x), ptr(P), ld, F_, H, W, Cc, R, S, stride, pad, 1 if nchw else 0, _stream(x))
    return P, OH, OW


def conv_wgrad_supported(Cin, Cout):
    return Cin % 4 == 0 and Cout % 4 == 0


def conv_wgrad(x, dy, wshape, stride, pad):
    """Weight gradient of a conv from the NHWC activations: x (F,H,W,Cin), dy (F,OH,OW,Cout) fp32 -> dw (Cout,Cin,R,S) fp32
    (the nn.Conv2d.weight.grad layout); exact fp32 or, under ``f32_split``, the split-bf16 arithmetic."""
    Cout, Cin, R, S = (int(v) for v in wshape)
    F_, H, W, Cc = x.shape
    assert Cc == Cin and dy.shape[-1] == Cout and x.dtype == torch.float32 and dy.dtype == torch.float32
    x, dy = x.contiguous(), dy.contiguous()
    dw = torch.empty((Cout, Cin, R, S), dtype=torch.float32, device=x.device)
    with _dev(x):
        nbytes = _hip.lib().agrl_conv_wgrad_workspace(F_, H, W, Cin, Cout, R, S, stride, pad)
        ws = torch.empty((max(int(nbytes) // 4, 4),), dtype=torch.float32, device=x.device)
        call("agrl_conv_wgrad", ptr(x), ptr(dy), ptr(dw), F_, H, W, Cin, Cout, R, S, stride, pad, _gemm_code(torch.float32), ptr(ws),
             ws.numel() * 4, _stream(x))
    return dw


def maxpool3x3s2(x):
    F_, H, W, Cc = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((F_, OH, OW, Cc), dtype=torch.float32, device=x.device)
    idx = torch.empty((F_, OH, OW, Cc), dtype=torch.uint8, device=x.device)
    with _dev(x):
        call("agrl_maxpool3x3s2", ptr(x), ptr(out), ptr(idx), F_, H, W, Cc, _stream(x))
    return out, idx


def maxpool3x3s2_backward(dout, idx, H, W):
    F_, OH, OW, Cc = dout.shape
    dx = torch.empty((F_, H, W, Cc), dtype=torch.float32, device=dout.device)
    with _dev(dout):
        call("agrl_maxpool3x3s2_backward", ptr(dout), ptr(idx), ptr(dx), F_, H, W, Cc, _stream(dout))
    return dx


def triplet_loss(x, pids, margin, soft):
    """Batch-hard triplet loss value + feature gradient in one native call (no host sync). -> loss (1,), grad (n,d)."""
    n, d = x.shape
    assert x.dtype == torch.float32 and pids.dtype == torch.int32
    dev = x.device
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    grad = torch.empty((n, d), dtype=torch.float32, device=dev)
    dap, dan = torch.empty((n,), dtype=torch.float32, device=dev), torch.empty((n,), dtype=torch.float32, device=dev)
    iap, ian = torch.empty((n,), dtype=torch.int32, device=dev), torch.empty((n,), dtype=torch.int32, device=dev)
    coeff = torch.empty((2 * n,), dtype=torch.float32, device=dev)
    with _dev(x):
        call("agrl_triplet_loss", ptr(x), ptr(pids), n, d, float(margin), 1 if soft else 0, ptr(loss), ptr(grad), ptr(dap), ptr(dan),
             ptr(iap), ptr(ian), ptr(coeff), _stream(x))
    return loss, grad


# ---- train step of the tail (include/agrl_hip.h, "train step of the tail" section) ------------------------------------------
def axpby(a, x, b=0.0, y=None):
    out = torch.empty_like(x)
    with _dev(x):
        call("agrl_axpby", ptr(x.contiguous()), ptr(y.contiguous()) if y is not None else None, float(a), float(b), ptr(out), x.numel(), _stream(x))
    return out


def part_pool_backward(dg, dnodes, S, h, w, splits):
    """dg (B,C) | None, dnodes (F,P,C) -> dx1 (F,h,w,C) | None, dx2 (F,h,w,C): backward of part_pool. vmgn.py:298-308."""
    F_, P, Cc = dnodes.shape
    dx2 = torch.empty((F_, h, w, Cc), dtype=torch.float32, device=dnodes.device)
    dx1 = torch.empty_like(dx2) if dg is not None else None
    arr = (C.c_int * len(splits))(*[int(v) for v in splits])
    with _dev(dnodes):
        call("agrl_part_pool_backward", ptr(dg.contiguous()) if dg is not None else None, ptr(dnodes.contiguous()), ptr(dx1), ptr(dx2), F_, S, h, w,
             Cc, arr, len(splits), _stream(dnodes))
    return dx1, dx2


def attn_pool_backward(nodes, datt):
    """nodes (B,S,P,C), datt (B,C) -> dnodes (B,S,P,C). vmgn.py:270-278, :313-317."""
    B, S, P, Cc = nodes.shape
    dn = torch.empty_like(nodes)
    with _dev(nodes):
        call("agrl_attn_pool_backward", ptr(nodes.contiguous()), ptr(datt.contiguous()), ptr(dn), B, S, P, Cc, _stream(nodes))
    return dn


def graph_gram(f):
    """f (B,V,C) fp32 -> Gram partials (B, C/128, V, V) (first half of graph_matrix, kept for the backward pass)."""
    B, V, Cc = f.shape
    nz = Cc // GRAM_CSLICE
    gram = torch.empty((B, nz, V, V), dtype=torch.float32, device=f.device)
    with _dev(f):
        call("agrl_graph_gram", ptr(f), ptr(gram), B, V, Cc, GRAM_CSLICE, _stream(f))
    return gram


PAIR_PRODUCT_MAX_V = 144


def graph_pair_product(a, b):
    """a, b (B,V,C) fp32 -> (B,V,V) with out[t] = a[t] b[t]^T (d loss / d G of the message pass, vmgn.py:168)."""
    B, V, Cc = a.shape
    assert a.shape == b.shape and a.dtype == b.dtype == torch.float32 and Cc % GRAM_CSLICE == 0 and V <= PAIR_PRODUCT_MAX_V
    part = torch.empty((B, Cc // GRAM_CSLICE, V, V), dtype=torch.float32, device=a.device)
    out = torch.empty((B, V, V), dtype=torch.float32, device=a.device)
    with _dev(a):
        call("agrl_graph_pair_product", ptr(a.contiguous()), ptr(b.contiguous()), ptr(part), ptr(out), B, V, Cc, _stream(a))
    return out


def graph_finalize(gram, adj, B, V, use_pose, learn_graph, mask_diag=False):
    G = torch.empty((B, V, V), dtype=torch.float32, device=(gram if gram is not None else adj).device)
    nz = gram.shape[1] if gram is not None else 0
    with _dev(G):
        call("agrl_graph_finalize", ptr(gram), nz, ptr(adj.contiguous()) if use_pose else None, ptr(G), B, V, 1 if use_pose else 0,
             1 if learn_graph else 0, 1 if mask_diag else 0, _stream(G))
    return G


def graph_matrix_backward(gram, dG, use_pose, mask_diag=False):
    """Gram partials (B,nz,V,V), dG (B,V,V) -> M (B,V,V) with d loss / d f = M f. vmgn.py:114-120, :155-166."""
    B, nz, V, _ = gram.shape
    M = torch.empty((B, V, V), dtype=torch.float32, device=gram.device)
    with _dev(gram):
        call("agrl_graph_matrix_backward", ptr(gram), nz, ptr(dG.contiguous()), ptr(M), B, V, 1 if use_pose else 0, 1 if mask_diag else 0, _stream(gram))
    return M


def graph_apply(G, h):
    """G (B,V,V) @ h (B,V,C) on the message-pass kernel (unit BatchNorm, no activation, no residual)."""
    Cc = h.shape[-1]
    key = (h.device, Cc)
    if key not in _UNIT:
        _UNIT[key] = (torch.ones((Cc,), dtype=torch.float32, device=h.device), torch.zeros((Cc,), dtype=torch.float32, device=h.device))
    one, zero = _UNIT[key]
    out, _ = graph_propagate(h, h, G, one, zero, 1.0, 1.0, want_lp=False, keep=0.0)
    return out


_UNIT = {}


def xent_label_smooth(logits, targets, eps):
    """-> loss (1,), dlogits (n,K): CrossEntropyLabelSmooth value + gradient in one call. cross_entropy_loss.py:26-37."""
    n, K = logits.shape
    loss = torch.empty((1,), dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits)
    rows = torch.empty((n,), dtype=torch.float32, device=logits.device)
    with _dev(logits):
        call("agrl_xent_label_smooth", ptr(logits), ptr(targets), n, K, float(eps), ptr(loss), ptr(dl), ptr(rows), _stream(logits))
    return loss, dl
