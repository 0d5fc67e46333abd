"""ctypes binding of libagrl_hip.so -- the C-ABI declared in include/agrl_hip.h.

The library is the product: there is NO fallback. ``lib()`` raises ``HipLibraryError`` when the
shared object is missing or does not export a declared entry point, and every wrapper raises
``HipKernelError`` carrying ``agrl_last_error()`` when an entry point returns non-zero.

Only device pointers, sizes and the current HIP stream cross this boundary; torch is used by the
callers for device memory and streams, never for the arithmetic of the hot path.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

F32 = 0
LP16 = 1   # the library's 16-bit storage type: fp16 (libagrl_hip.so) or bfloat16 (libagrl_hip_bf16.so)
F32X3 = 2  # fp32 tensors, split-bf16 MFMA arithmetic (include/agrl_hip.h)
F32H3P = 4  # F32H3 with the activation operand pre-split (graph_apply -> graph_linear_mix)
F32H3 = 3  # fp32 tensors, split-FP16 MFMA arithmetic with pre-scaled weights (agrl_conv2d_bn_act_split16)
METRIC_EUCLIDEAN = 0
METRIC_COSINE = 1

# The 16-bit type is a property of the LIBRARY (the same sources built twice, include/agrl_hip.h): AGRL_HIP_LP16 = fp16
# (default: 8 x smaller rounding error at the same MFMA rate, the path's outputs stay within the north star's 1e-3 of the
# fp32 oracle) or bf16. It is fixed for the process when this module is imported.
LP_NAME = os.environ.get("AGRL_HIP_LP16")
if LP_NAME is None:
    # not chosen explicitly: a 16-bit AGRL_HIP_PRECISION names the type (round-2 scripts exported AGRL_HIP_PRECISION=bf16 alone and
    # would otherwise fail at the first forward, after the model is built and the data loaded)
    LP_NAME = os.environ.get("AGRL_HIP_PRECISION") if os.environ.get("AGRL_HIP_PRECISION") in ("fp16", "bf16") else "fp16"
if LP_NAME not in ("fp16", "bf16"):
    raise ValueError("AGRL_HIP_LP16 must be fp16 or bf16, not %r" % LP_NAME)
LP_DTYPE = torch.float16 if LP_NAME == "fp16" else torch.bfloat16

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get(
    "AGRL_HIP_LIB", os.path.normpath(os.path.join(_HERE, "..", "lib", "libagrl_hip.so" if LP_NAME == "fp16" else "libagrl_hip_bf16.so"))
)


class HipLibraryError(RuntimeError):
    """libagrl_hip.so is missing or incomplete."""


class HipKernelError(RuntimeError):
    """An entry point of libagrl_hip.so returned a non-zero status."""


_p = C.c_void_p
_i = C.c_int
_f = C.c_float

# name -> argtypes; must list every function include/agrl_hip.h declares (tests/test_cabi.py checks)
SIGNATURES = {
    "agrl_stem_conv_bn_relu_maxpool": [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    "agrl_stem_conv_bn_relu_maxpool_lp16": [_p, _p, _p, _p, _i, _i, _i, _p],
    "agrl_conv2d_bn_act": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "agrl_conv1x1_dual_split16": [_p, _p, _p, _p, _p] + [_i] * 8 + [_f, _i, _i, _p],
    "agrl_stem_split16": [_p, _p, _p, _p, _p, _i, _i, _i, _f, _p],
    "agrl_split16_planes": [_p, _p, C.c_longlong, _i, _i, _p],
    "agrl_split16_weight_planes": [_p, _p, C.c_longlong, _i, _f, _p],
    "agrl_split16_weights_inloop": [_p, _p, C.c_longlong, _i, _p],
    "agrl_conv1x1_split16": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p],
    "agrl_conv1x1_split16_dual": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _p],
    "agrl_conv1x1_split16_pool": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, C.POINTER(_i), _i, _i, _f, _i, _p],
    "agrl_conv3x3_packed_split16": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p],
    "agrl_conv2d_bn_act_split16": [_p, _p, _p, _p, _p] + [_i] * 10 + [_f, _i, _i, _p],
    "agrl_bottleneck_tail": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "agrl_bottleneck_seam_packed_bytes": [_i, _i, _i],   # returns long long (restype patched after loading)
    "agrl_bottleneck_seam_pack": [_p, _p, _p, _i, _i, _i, _p],
    "agrl_bottleneck_seam": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "agrl_conv3x3_packed_bytes": [_i, _i],               # returns long long
    "agrl_conv1x1_packed_bytes": [_i, _i],               # returns long long
    "agrl_conv1x1_pack": [_p, _p, _i, _i, _p],
    "agrl_conv1x1_packed_bn_act": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "agrl_conv1x1_packed_res_bn_act": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "agrl_conv1x1_packed_res_pool": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, C.POINTER(_i), _i, _i, _p],
    "agrl_conv1x1_packed_dual_duo": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "agrl_conv1x1_packed_dual_strided": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "agrl_conv3x3_pack": [_p, _p, _i, _i, _p],
    "agrl_conv3x3_packed_bn_act": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "agrl_bottleneck_block": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "agrl_conv1x1_dual_bn_act": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "agrl_conv1x1_bn_act_pool": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, C.POINTER(_i), _i, _i, _p],
    "agrl_linear_nobias": [_p, _p, _p, _i, _i, _i, _i, _p],
    "agrl_part_pool": [_p, _p, _p, _p, _p, _i, _i, _i, _i, C.POINTER(_i), _i, _i, _p],
    "agrl_clip_pool": [_p, _p, _i, _i, _i, _i, _p],
    "agrl_graph_gram": [_p, _p, _i, _i, _i, _i, _p],
    "agrl_graph_pair_product": [_p, _p, _p, _p, _i, _i, _i, _p],
    "agrl_graph_finalize": [_p, _i, _p, _p, _i, _i, _i, _i, _i, _p],
    "agrl_graph_propagate": [_p, _p, _p, _p, _p, _f, _f, _f, _p, _p, _i, _i, _i, _p],
    "agrl_pam_pool": [_p, _p, _p, _p, _i, _i, _i, _i, _i, C.POINTER(_i), _i, _i, _p],
    "agrl_pam_combine": [_p, _p, _p, _f, _p, _p, _i, _i, _p],
    "agrl_graph_apply": [_p, _p, _p, _i, _i, _i, _i, _p],
    "agrl_graph_tracklet_operand": [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "agrl_graph_finalize_bits": [_p, _i, _p, _p, _i, _i, _i, _i, _i, _p],
    "agrl_pose_adjacency_bits": [_p, _p, _p, _i, _i, _i, _i, C.c_float, C.c_float, _p],
    "agrl_adjacency_pack": [_p, _p, _i, _i, _p],
    "agrl_graph_linear_mix": [_p, _p, _p, _p, _p, _f, _f, _f, _p, _i, _i, _i, _i, _p],
    "agrl_row_sqnorm": [_p, _p, _i, _i, _i, _p],
    "agrl_attn_pool_bnneck": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "agrl_attn_tail": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "agrl_row_l2_normalize": [_p, _p, _i, _i, _i, _i, _i, _p],
    "agrl_distmat_split16": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p, C.c_size_t, _p],
    "agrl_distmat": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, C.c_size_t, _p],
    "agrl_pose_adjacency": [_p, _p, _p, _i, _i, _i, _i, C.c_float, C.c_float, _p],
    "agrl_re_ranking_workspace": [_i, _i, _i],   # returns size_t (restype patched after loading)
    "agrl_re_ranking": [_p, _p, _p, _i, _i, _i, _i, C.c_double, _p, _i, _p, C.c_size_t, _p],
    "agrl_rank_topk": [_p, _i, _i, _i, _i, _i, _p, _p, _p],
    "agrl_distmat_topk_workspace": [_i, _i],   # returns size_t
    "agrl_distmat_topk": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, C.c_size_t, _p, C.c_size_t, _p],
    "agrl_rank_argsort": [_p, _i, _i, _i, _p, _p],
    "agrl_rank_mars": [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p],
    "agrl_rank_market1501": [_p, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p],
    "agrl_triplet_hard_mine": [_p, _p, _i, _i, _p, _p, _p, _p, _p],
    "agrl_axpby": [_p, _p, _f, _f, _p, C.c_size_t, _p],
    "agrl_part_pool_backward": [_p, _p, _p, _p, _i, _i, _i, _i, _i, C.POINTER(_i), _i, _p],
    "agrl_attn_pool_backward": [_p, _p, _p, _i, _i, _i, _i, _p],
    "agrl_graph_matrix_backward": [_p, _i, _p, _p, _i, _i, _i, _i, _p],
    "agrl_xent_label_smooth": [_p, _p, _i, _i, _f, _p, _p, _p, _p],
    "agrl_triplet_loss": [_p, _p, _i, _i, _f, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    "agrl_bn_workspace": [_i, _i],   # returns size_t
    "agrl_bn_stats": [_p, _p, _p, _i, _i, _p, C.c_size_t, _p],
    "agrl_conv2d_stats": [_p, _p, _p, _p, C.c_size_t] + [_i] * 10 + [_p],
    "agrl_bn_stats_from_partials": [_p, _i, _i, _i, _p, _p, _p, C.c_size_t, _p],
    "agrl_bn_fold_train": [_p, _p, _p, _p, _f, _f, C.c_longlong, _p, _p, _p, _p, _p, _p, _i, _p],
    "agrl_bn_apply": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p],
    "agrl_bn_backward": [_p, _p, _p, _p, _p, _p, _p, _i, _f, _p, _p, _p, _p, _i, _i, _p, C.c_size_t, _p],
    "agrl_im2col_t": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    "agrl_gemm_nt_splitk": [_p, _p, _p, _i, _i, _i, _i, _p, C.c_size_t, _p],
    "agrl_im2col_rows": [_p, _p] + [_i] * 10 + [_p],
    "agrl_conv_wgrad_workspace": [_i] * 9,   # returns size_t
    "agrl_conv_wgrad": [_p, _p, _p] + [_i] * 10 + [_p, C.c_size_t, _p],
    "agrl_maxpool3x3s2": [_p, _p, _p, _i, _i, _i, _i, _p],
    "agrl_maxpool3x3s2_backward": [_p, _p, _p, _i, _i, _i, _i, _p],
    "agrl_diag_read_stream": [_p, C.c_size_t, _p, _i, _p],
}

_lock = threading.Lock()
_lib = None


def lib():
    """Load (once) and return the ctypes handle; raise loudly if it cannot be used."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                "libagrl_hip.so not found at %s -- build it with `python __graft_entry__.py` "
                "(or `make -C agrl.pytorch_amd/csrc`); the HIP path has no fallback" % LIB_PATH
            )
        try:
            h = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover - depends on the host
            raise HipLibraryError("cannot load %s: %s" % (LIB_PATH, e))
        for name, argtypes in SIGNATURES.items():
            try:
                fn = getattr(h, name)
            except AttributeError:
                raise HipLibraryError("%s does not export %s" % (LIB_PATH, name))
            fn.argtypes = argtypes
            fn.restype = _i
        h.agrl_re_ranking_workspace.restype = C.c_size_t
        h.agrl_bn_workspace.restype = C.c_size_t
        h.agrl_conv_wgrad_workspace.restype = C.c_size_t
        h.agrl_distmat_topk_workspace.restype = C.c_size_t
        h.agrl_bottleneck_seam_packed_bytes.restype = C.c_longlong
        h.agrl_conv3x3_packed_bytes.restype = C.c_longlong
        h.agrl_conv1x1_packed_bytes.restype = C.c_longlong
        for name in ("agrl_reload_options", "agrl_built_with_ablation", "agrl_lp16_is_f16"):
            getattr(h, name).argtypes = []
            getattr(h, name).restype = _i
        h.agrl_version.restype = _i
        h.agrl_last_error.restype = C.c_char_p
        if bool(h.agrl_lp16_is_f16()) != (LP_NAME == "fp16"):
            raise HipLibraryError("%s stores 16-bit data as %s but AGRL_HIP_LP16 asks for %s" % (
                LIB_PATH, "fp16" if h.agrl_lp16_is_f16() else "bf16", LP_NAME))
        _lib = h
    return _lib


RELOAD_HOOKS = []   # host-side caches of environment switches (hip_ops.switch) register their invalidation here


def reload_options():
    """Make the library re-read its AGRL_* tuning switches from the environment (they are read once at load time), and drop the
    host side's cached switches with them."""
    lib().agrl_reload_options()
    for hook in RELOAD_HOOKS:
        hook()


def available() -> bool:
    try:
        lib()
        return True
    except HipLibraryError:
        return False


def _check(name, status):
    if status != 0:
        msg = lib().agrl_last_error()
        raise HipKernelError("%s failed (%d): %s" % (name, status, msg.decode() if msg else "?"))


def stream_ptr(device=None):
    """The current HIP stream of the calling thread on ``device`` as an integer handle."""
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL). The tensor must be contiguous."""
    if t is None:
        return None
    assert t.is_cuda, "HIP entry points take device tensors"
    assert t.is_contiguous(), "HIP entry points take contiguous tensors"
    return t.data_ptr()


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == LP_DTYPE:
        return LP16
    raise TypeError("unsupported dtype %s (float32 / %s only: the loaded library's 16-bit type is %s)" % (dt, LP_DTYPE, LP_NAME))


# Optional launch timing: set ``PROFILE`` to a list and every entry-point call appends
# (name, start_event, end_event, tag); the events are recorded on the stream the kernel is launched on
# (the last argument of every entry point). Used by bench.py for the live roofline numbers.
PROFILE = None
PROFILE_TAG = None


def call(name, *args):
    """Invoke an entry point (its last argument is the HIP stream handle)."""
    fn = getattr(lib(), name)
    if PROFILE is None:
        _check(name, fn(*args))
        return
    stream = torch.cuda.ExternalStream(args[-1]) if args[-1] else torch.cuda.current_stream()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record(stream)
    _check(name, fn(*args))
    end.record(stream)
    global PROFILE_TAG
    PROFILE.append((name, start, end, PROFILE_TAG))
    PROFILE_TAG = None
