"""GPU parity tests, kernel by kernel, through the C-ABI (libagrl_hip.so) against the CPU oracle.

fp32 (exact-fp32 MFMA) must agree with the oracle to ~1e-5 relative; the north-star bar is 1e-3.
The 16-bit type of the loaded build (lp16.LP_DTYPE: fp16 by default, bf16 with AGRL_HIP_LP16=bf16) is the throughput mode:
operands rounded to it, fp32 accumulation -> tolerance 2e-2 of the output scale (set for bf16; fp16 sits 8 x below it).
"""
import os

import numpy as np
import pytest
import torch

from lp16 import LP16, LP_DTYPE
import torch.nn.functional as F

from oracle import vmgn_oracle as O
from torchreid import _hip

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = {torch.float32: 2e-5, LP_DTYPE: 2e-2}


def rel_err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return ((got - ref).abs().max() / ref.abs().max().clamp(min=1e-30)).item()


def nhwc(x, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


CONV_CASES = [
    # N, H, W, Cin, Cout, R, stride, residual, relu
    (3, 10, 7, 64, 64, 1, 1, False, True),      # ragged M (210 pixels), small N tile
    (2, 16, 8, 64, 256, 1, 1, True, True),      # conv3 + residual
    (2, 16, 8, 128, 128, 3, 1, False, True),    # 3x3 pad 1
    (2, 16, 8, 128, 128, 3, 2, False, True),    # 3x3 stride 2
    (2, 16, 8, 256, 512, 1, 2, False, False),   # downsample 1x1 stride 2, no relu
    (1, 9, 5, 512, 200, 3, 1, True, False),     # ragged N (200 channels), odd spatial
    (4, 16, 8, 1024, 512, 1, 1, False, True),   # deep K
    (2, 32, 16, 64, 64, 3, 1, False, True),     # 3x3 LDS-patch kernel: 4 pixel blocks per frame, narrow N
    (1, 64, 32, 128, 128, 3, 1, False, True),   # 3x3 patch, 16 blocks per frame, 2 channel slabs
    (3, 16, 8, 512, 200, 3, 1, False, False),   # 3x3 patch, whole-frame tile, ragged N, 8 slabs
]


@pytest.mark.parametrize("dtype", [torch.float32, LP_DTYPE])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_bn_act(case, dtype):
    from torchreid import hip_ops as ops
    N, H, W, Cin, Cout, R, stride, use_res, relu = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, R, R), generator=g) / np.sqrt(Cin * R * R)
    b = torch.randn((Cout,), generator=g)
    pad = R // 2
    if dtype == LP_DTYPE:  # the kernel sees bf16-rounded operands; so does the reference
        x, w = x.to(LP_DTYPE).float(), w.to(LP_DTYPE).float()
    ref = F.conv2d(x, w, bias=b, stride=stride, padding=pad)
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g)
        if dtype == LP_DTYPE:
            res = res.to(LP_DTYPE).float()
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    out = ops.conv_bn_act(nhwc(x, dtype), w.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV), b.to(DEV), stride, pad,
                          relu, residual=None if res is None else nhwc(res, dtype))
    torch.cuda.synchronize()
    got = out.float().permute(0, 3, 1, 2)
    e = rel_err(got, ref)
    print("conv", case, dtype, "rel err %.3e" % e)
    assert e < (1e-5 if dtype == torch.float32 else 1e-2)


WIDE_CASES = [
    # N, H, W, Cin, Cout, residual, relu  -- the 256 x 256 tile kernel (igemm_wide.hip), forced through AGRL_IGEMM_WIDE=1
    (4, 16, 8, 512, 512, True, True),     # two M tiles x two N tiles, residual halves through both ring slots
    (3, 10, 7, 256, 256, False, True),    # ragged M (210 pixels < one tile)
    (5, 16, 8, 128, 768, True, False),    # 2.5 M tiles, 3 N tiles, 2 k-tiles, no relu
    (2, 16, 8, 2048, 256, False, True),   # 32 k-tiles
]


@pytest.mark.parametrize("tile", ["2", "3"])  # AGRL_IGEMM_WIDE: 2 = 256 x 256 tile, 3 = 256 x 128 tile
@pytest.mark.parametrize("case", WIDE_CASES + [(3, 16, 8, 1024, 384, True, True),
                                  # more tiles than CUs: the persistent form (a workgroup walks several tiles, the next
                                  # tile's first k-tile is requested in front of the stores); ragged last M tile; 1 / 3 k-tiles
                                  (81, 16, 8, 64, 2048, False, True), (90, 16, 8, 192, 1024, False, False)])
def test_conv_wide_tile(case, tile, monkeypatch):
    from torchreid import hip_ops as ops
    N, H, W, Cin, Cout, use_res, relu = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = (torch.randn((N, Cin, H, W), generator=g)).to(LP_DTYPE).float()
    w = (torch.randn((Cout, Cin, 1, 1), generator=g) / np.sqrt(Cin)).to(LP_DTYPE).float()
    b = torch.randn((Cout,), generator=g)
    ref = F.conv2d(x, w, bias=b)
    res = None
    if use_res:
        res = torch.randn(ref.shape, generator=g).to(LP_DTYPE).float()
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    args = (nhwc(x, LP_DTYPE), w.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV), b.to(DEV), 1, 0, relu)
    kw = dict(residual=None if res is None else nhwc(res, LP_DTYPE))
    if tile == "2" and Cout % 256:
        pytest.skip("256-channel tiles need Cout % 256 == 0")
    monkeypatch.setenv("AGRL_IGEMM_WIDE", tile)
    _hip.reload_options()
    wide = ops.conv_bn_act(*args, **kw)
    monkeypatch.setenv("AGRL_IGEMM_WIDE", "0")
    _hip.reload_options()
    narrow = ops.conv_bn_act(*args, **kw)
    torch.cuda.synchronize()
    e = rel_err(wide.float().permute(0, 3, 1, 2), ref)
    print("wide conv", case, "rel err %.3e" % e)
    assert e < 1e-2
    # same fp32 accumulation order per output (k ascending in 32-deep MFMA steps) -> identical bf16 results
    assert torch.equal(wide, narrow)


@pytest.mark.parametrize("cnext", [64, 128])
@pytest.mark.parametrize("shape", [(2, 64, 32), (1, 16, 8), (3, 10, 7), (5, 64, 32)])
def test_bottleneck_tail(shape, cnext):
    """conv3 + residual + relu of a layer-1 block fused with the next block's conv1 (bottleneck_tail.hip) against the
    two separate igemm launches (bitwise: same fp32 accumulation order, same roundings) and the fp32 reference."""
    from torchreid import hip_ops as ops
    N, H, W = shape
    g = torch.Generator().manual_seed(N * H)
    y2 = torch.randn((N, 64, H, W), generator=g).to(LP_DTYPE).float()
    res = torch.randn((N, 256, H, W), generator=g).to(LP_DTYPE).float()
    w3 = (torch.randn((256, 64, 1, 1), generator=g) / 8).to(LP_DTYPE).float()
    w1 = (torch.randn((cnext, 256, 1, 1), generator=g) / 16).to(LP_DTYPE).float()
    b3, b1 = torch.randn(256, generator=g), torch.randn(cnext, generator=g)
    out_ref = F.relu(F.conv2d(y2, w3, bias=b3) + res)
    z_ref = F.relu(F.conv2d(out_ref.to(LP_DTYPE).float(), w1, bias=b1))
    dy2, dres = nhwc(y2, LP_DTYPE), nhwc(res, LP_DTYPE)
    dw3 = w3.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV)
    dw1 = w1.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV)
    assert ops.bottleneck_tail_supported(dy2, dw3, dw1)
    out, z = ops.bottleneck_tail(dy2, dw3, b3.to(DEV), dres, dw1, b1.to(DEV))
    out2 = ops.conv_bn_act(dy2, dw3, b3.to(DEV), 1, 0, True, residual=dres)
    z2 = ops.conv_bn_act(out2, dw1, b1.to(DEV), 1, 0, True)
    torch.cuda.synchronize()
    e1, e2 = rel_err(out.float().permute(0, 3, 1, 2), out_ref), rel_err(z.float().permute(0, 3, 1, 2), z_ref)
    print("bottleneck tail", shape, "out %.3e z %.3e" % (e1, e2))
    assert e1 < 1e-2 and e2 < 1e-2
    assert torch.equal(out, out2) and torch.equal(z, z2)
    if cnext != 64:
        return
    # first-block form: the shortcut is the block's 1x1 downsample conv of x0, computed in the same pass
    x0 = torch.randn((N, 64, H, W), generator=g).to(LP_DTYPE).float()
    ws = (torch.randn((256, 64, 1, 1), generator=g) / 8).to(LP_DTYPE).float()
    bs = torch.randn(256, generator=g)
    out_ref = F.relu(F.conv2d(y2, w3, bias=b3) + F.conv2d(x0, ws, bias=bs))
    z_ref = F.relu(F.conv2d(out_ref.to(LP_DTYPE).float(), w1, bias=b1))
    out, z = ops.bottleneck_tail(dy2, dw3, b3.to(DEV), None, dw1, b1.to(DEV),
                                 shortcut=(nhwc(x0, LP_DTYPE), ws.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV), bs.to(DEV)))
    torch.cuda.synchronize()
    e1, e2 = rel_err(out.float().permute(0, 3, 1, 2), out_ref), rel_err(z.float().permute(0, 3, 1, 2), z_ref)
    print("bottleneck tail + downsample", shape, "out %.3e z %.3e" % (e1, e2))
    assert e1 < 1e-2 and e2 < 1e-2


@pytest.mark.parametrize("cnext", [64, 128])
@pytest.mark.parametrize("shape", [(2, 64, 32), (1, 16, 8), (3, 8, 24), (9, 64, 32)])
def test_bottleneck_block(shape, cnext, monkeypatch):
    """3x3 conv + conv3 + residual + relu of a layer-1 block fused with the next block's conv1 (bottleneck_block.hip)
    against the three separate launches (bitwise: same fp32 accumulation order, same roundings) and the fp32 reference;
    (9, 64, 32) = 288 tiles, more than one per workgroup; also the first-block form with the downsample conv."""
    from torchreid import hip_ops as ops
    N, H, W = shape
    g = torch.Generator().manual_seed(N * H + cnext)
    z = torch.randn((N, 64, H, W), generator=g).to(LP_DTYPE).float()
    res = torch.randn((N, 256, H, W), generator=g).to(LP_DTYPE).float()
    w2 = (torch.randn((64, 64, 3, 3), generator=g) / 24).to(LP_DTYPE).float()
    w3 = (torch.randn((256, 64, 1, 1), generator=g) / 8).to(LP_DTYPE).float()
    w1 = (torch.randn((cnext, 256, 1, 1), generator=g) / 16).to(LP_DTYPE).float()
    b2, b3, b1 = torch.randn(64, generator=g), torch.randn(256, generator=g), torch.randn(cnext, generator=g)
    y2_ref = F.relu(F.conv2d(z, w2, bias=b2, padding=1)).to(LP_DTYPE).float()
    out_ref = F.relu(F.conv2d(y2_ref, w3, bias=b3) + res)
    z_ref = F.relu(F.conv2d(out_ref.to(LP_DTYPE).float(), w1, bias=b1))
    dz, dres = nhwc(z, LP_DTYPE), nhwc(res, LP_DTYPE)
    ohwi = lambda w: w.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV)
    dw2, dw3, dw1 = ohwi(w2), ohwi(w3), ohwi(w1)
    assert ops.bottleneck_block_supported(dz, dw2, 1, dw3, dw1)
    out, zn = ops.bottleneck_block(dz, dw2, b2.to(DEV), dw3, b3.to(DEV), dres, dw1, b1.to(DEV))
    y2s = ops.conv_bn_act(dz, dw2, b2.to(DEV), 1, 1, True)
    out2 = ops.conv_bn_act(y2s, dw3, b3.to(DEV), 1, 0, True, residual=dres)
    z2 = ops.conv_bn_act(out2, dw1, b1.to(DEV), 1, 0, True)
    torch.cuda.synchronize()
    e1, e2 = rel_err(out.float().permute(0, 3, 1, 2), out_ref), rel_err(zn.float().permute(0, 3, 1, 2), z_ref)
    print("bottleneck block", shape, cnext, "out %.3e z %.3e" % (e1, e2))
    assert e1 < 1e-2 and e2 < 1e-2
    assert torch.equal(out, out2) and torch.equal(zn, z2)
    if cnext != 64:
        return
    x0 = torch.randn((N, 64, H, W), generator=g).to(LP_DTYPE).float()
    ws = (torch.randn((256, 64, 1, 1), generator=g) / 8).to(LP_DTYPE).float()
    bs = torch.randn(256, generator=g)
    out_ref = F.relu(F.conv2d(y2_ref, w3, bias=b3) + F.conv2d(x0, ws, bias=bs))
    z_ref = F.relu(F.conv2d(out_ref.to(LP_DTYPE).float(), w1, bias=b1))
    out, zn = ops.bottleneck_block(dz, dw2, b2.to(DEV), dw3, b3.to(DEV), None, dw1, b1.to(DEV),
                                   shortcut=(nhwc(x0, LP_DTYPE), ohwi(ws), bs.to(DEV)))
    out3, z3 = ops.bottleneck_tail(y2s, dw3, b3.to(DEV), None, dw1, b1.to(DEV), shortcut=(nhwc(x0, LP_DTYPE), ohwi(ws), bs.to(DEV)))
    torch.cuda.synchronize()
    e1, e2 = rel_err(out.float().permute(0, 3, 1, 2), out_ref), rel_err(zn.float().permute(0, 3, 1, 2), z_ref)
    print("bottleneck block + downsample", shape, "out %.3e z %.3e" % (e1, e2))
    assert e1 < 1e-2 and e2 < 1e-2
    assert torch.equal(out, out3) and torch.equal(zn, z3)


@pytest.mark.parametrize("shape", [(2, 32, 16), (1, 16, 8), (3, 10, 7), (40, 32, 16)])
def test_bottleneck_tail_layer2(shape):
    """Layer-2 form of the fused tail (conv3 128 -> 512 + residual + relu, next conv1 512 -> 128; weights resident in
    registers) against the two separate igemm launches (bitwise) and the fp32 reference; (40, 32, 16) = 320 tiles, more
    than one per workgroup (prefetch ring, counted waits), (3, 10, 7) a ragged last tile."""
    from torchreid import hip_ops as ops
    N, H, W = shape
    g = torch.Generator().manual_seed(N * H + 7)
    y2 = torch.randn((N, 128, H, W), generator=g).to(LP_DTYPE).float()
    res = torch.randn((N, 512, H, W), generator=g).to(LP_DTYPE).float()
    w3 = (torch.randn((512, 128, 1, 1), generator=g) / 11).to(LP_DTYPE).float()
    w1 = (torch.randn((128, 512, 1, 1), generator=g) / 22).to(LP_DTYPE).float()
    b3, b1 = torch.randn(512, generator=g), torch.randn(128, generator=g)
    out_ref = F.relu(F.conv2d(y2, w3, bias=b3) + res)
    z_ref = F.relu(F.conv2d(out_ref.to(LP_DTYPE).float(), w1, bias=b1))
    dy2, dres = nhwc(y2, LP_DTYPE), nhwc(res, LP_DTYPE)
    dw3 = w3.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV)
    dw1 = w1.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV)
    assert ops.bottleneck_tail_supported(dy2, dw3, dw1)
    out, z = ops.bottleneck_tail(dy2, dw3, b3.to(DEV), dres, dw1, b1.to(DEV))
    out2 = ops.conv_bn_act(dy2, dw3, b3.to(DEV), 1, 0, True, residual=dres)
    z2 = ops.conv_bn_act(out2, dw1, b1.to(DEV), 1, 0, True)
    torch.cuda.synchronize()
    e1, e2 = rel_err(out.float().permute(0, 3, 1, 2), out_ref), rel_err(z.float().permute(0, 3, 1, 2), z_ref)
    print("bottleneck tail layer 2", shape, "out %.3e z %.3e" % (e1, e2))
    assert e1 < 1e-2 and e2 < 1e-2
    assert torch.equal(out, out2) and torch.equal(z, z2)


@pytest.mark.parametrize("frames", [1, 3, 40])
@pytest.mark.parametrize("dims", [(256, 1024, 256), (512, 2048, 512), (256, 1024, 512)])
def test_bottleneck_seam(dims, frames):
    """conv3 + residual + relu of a layer-3 / layer-4 block back to back with the next block's conv1 (bottleneck_seam.hip:
    128-pixel tiles, 256-channel chunks handed over through LDS, the pre-packed weights streamed into register rings) against
    the fp32 reference and against the two separate launches. Not bitwise: a chunk's accumulators start as residual + bias, so
    the fp32 summation order differs from (acc + bias) + residual -- the 16-bit outputs may differ by one rounding. 40 frames =
    40 tiles, 1 frame a single workgroup; every call twice (bit-for-bit repeatable: no race between the rings' counted waits)."""
    from torchreid import hip_ops as ops
    cmid, cout, cnext = dims
    H, W = 16, 8
    g = torch.Generator().manual_seed(frames * cmid + cnext)
    y2 = torch.randn((frames, cmid, H, W), generator=g).to(LP_DTYPE).float()
    res = torch.randn((frames, cout, H, W), generator=g).to(LP_DTYPE).float()
    w3 = (torch.randn((cout, cmid, 1, 1), generator=g) / np.sqrt(cmid)).to(LP_DTYPE).float()
    w1 = (torch.randn((cnext, cout, 1, 1), generator=g) / np.sqrt(cout)).to(LP_DTYPE).float()
    b3, b1 = torch.randn(cout, generator=g), torch.randn(cnext, generator=g)
    out_ref = F.relu(F.conv2d(y2, w3, bias=b3) + res)
    z_ref = F.relu(F.conv2d(out_ref.to(LP_DTYPE).float(), w1, bias=b1))
    dy2, dres = nhwc(y2, LP_DTYPE), nhwc(res, LP_DTYPE)
    dw3 = w3.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV)
    dw1 = w1.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV)
    assert ops.bottleneck_seam_supported(dw3, dw1, frames * H * W)
    packed = ops.bottleneck_seam_pack(dw3, dw1)
    out, z = ops.bottleneck_seam(dy2, packed, b3.to(DEV), dres, b1.to(DEV), dims)
    out_b, z_b = ops.bottleneck_seam(dy2, packed, b3.to(DEV), dres, b1.to(DEV), dims)
    out2 = ops.conv_bn_act(dy2, dw3, b3.to(DEV), 1, 0, True, residual=dres)
    z2 = ops.conv_bn_act(out2, dw1, b1.to(DEV), 1, 0, True)
    torch.cuda.synchronize()
    e1, e2 = rel_err(out.float().permute(0, 3, 1, 2), out_ref), rel_err(z.float().permute(0, 3, 1, 2), z_ref)
    d1 = (out.float() - out2.float()).abs().max().item() / out2.float().abs().max().item()
    d2 = (z.float() - z2.float()).abs().max().item() / z2.float().abs().max().item()
    print("bottleneck seam", dims, frames, "out %.3e z %.3e | vs two launches out %.3e z %.3e" % (e1, e2, d1, d2))
    tol = 3e-3 if LP_DTYPE == torch.float16 else 2e-2
    assert e1 < tol and e2 < tol
    assert d1 < tol and d2 < tol
    assert torch.equal(out, out_b) and torch.equal(z, z_b)
    # a pixel count that is not a whole number of 128-pixel tiles is not taken (10 x 6 frames): the caller runs the two convs
    assert not ops.bottleneck_seam_supported(dw3, dw1, 60)
    with pytest.raises(_hip.HipKernelError):
        ops.call("agrl_bottleneck_seam", ops.ptr(dy2), ops.ptr(packed), ops.ptr(b3.to(DEV)), ops.ptr(dres), ops.ptr(out), ops.ptr(b1.to(DEV)),
                 ops.ptr(z), 60, cmid, cout, cnext, None)


@pytest.mark.parametrize("case", [(1, 16, 8, 128, 256, True), (3, 16, 8, 256, 256, True), (5, 32, 16, 192, 512, False),
                                  (250, 16, 8, 512, 512, True), (231, 16, 8, 256, 256, True), (7, 32, 16, 128, 128, True), (64, 32, 16, 128, 128, False)])
def test_conv3x3_packed(case, monkeypatch):
    """3x3 conv through the four-wave kernel with the pre-packed weight stream (conv3x3_fat.hip) against the fp32 reference and
    against conv_bn_act (same summation order: equal bit for bit). 1 frame = a single workgroup with an absent second block;
    5 frames of 32 x 16 = blocks with real neighbours on all sides (halo rows / columns from the map, zeros at the border);
    250 frames x 512 channels and 231 (an odd block count) also run forced into the two-blocks-per-workgroup form and the half-width form; 128 -> 128 on 32 x 16 maps (layer
    2) runs as one half-width workgroup per block on the lower half of a 256-channel weight tile. Every call twice."""
    from torchreid import hip_ops as ops
    N, H, W, Cin, Cout, relu = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn((N, Cin, H, W), generator=g).to(LP_DTYPE).float()
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) / np.sqrt(9 * Cin)).to(LP_DTYPE).float()
    b = torch.randn((Cout,), generator=g)
    dx = nhwc(x, LP_DTYPE)
    dw = w.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV)
    assert ops.conv3x3_packed_supported(dw, H, W)
    packed = ops.conv3x3_pack(dw)
    out = ops.conv3x3_packed(dx, packed, b.to(DEV), Cout, relu)
    out_b = ops.conv3x3_packed(dx, packed, b.to(DEV), Cout, relu)
    other = ops.conv_bn_act(dx, dw, b.to(DEV), 1, 1, relu)
    torch.cuda.synchronize()
    ref = F.conv2d(x.to(DEV), w.to(DEV), bias=b.to(DEV), padding=1)
    ref = F.relu(ref) if relu else ref
    e = rel_err(out.float().permute(0, 3, 1, 2).cpu(), ref.cpu())
    d = (out.float() - other.float()).abs().max().item()
    print("conv3x3 packed", case, "vs fp32 %.3e | max |packed - conv_bn_act| %.3g" % (e, d))
    assert e < (3e-3 if LP_DTYPE == torch.float16 else 2e-2)
    assert torch.equal(out, out_b)
    assert torch.equal(out, other)
    assert not ops.conv3x3_packed_supported(dw, 10, 6)
    if Cout % 256 == 0:   # the other forms of the same kernel family: two blocks per workgroup (the default until late round 5), half-width workgroups
        for var, val in (("AGRL_CONV3X3_FAT_PB", "2"), ("AGRL_CONV3X3_HALF", "1"), ("AGRL_CONV3X3_HALF", "0")):
            monkeypatch.setenv(var, val)
            _hip.reload_options()
            form = ops.conv3x3_packed(dx, packed, b.to(DEV), Cout, relu)
            torch.cuda.synchronize()
            assert torch.equal(out, form), (var, val)
            monkeypatch.delenv(var)
        _hip.reload_options()
    with pytest.raises(_hip.HipKernelError):
        ops.call("agrl_conv3x3_packed_bn_act", ops.ptr(dx), ops.ptr(packed), ops.ptr(b.to(DEV)), ops.ptr(out), N, 10, 6, Cin, Cout, 1, None)


@pytest.mark.parametrize("case", [(1, 16, 8, 128, 0, 256, True), (3, 10, 6, 256, 0, 512, False), (40, 16, 8, 2048, 0, 512, True),
                                  (37, 16, 8, 1024, 512, 2048, True), (2, 16, 8, 256, 128, 256, True), (250, 16, 8, 1024, 0, 512, True)])
def test_conv1x1_packed(case):
    """1x1 conv through the four-wave kernel with the pre-packed weight stream (conv1x1_fat.hip), one source and two (K axis
    concatenated: conv3 + downsample conv of a first block), against the fp32 reference and against conv_bn_act /
    conv1x1_dual (same summation order: equal bit for bit). 1 frame = half a pixel tile; 3 frames of 10 x 6 = 180 rows (ragged
    tile, no ReLU); 37 frames = an odd tile count with the layer-4 first-block shape. Every call twice."""
    from torchreid import hip_ops as ops
    N, H, W, K1, K2, Cout, relu = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    x = torch.randn((N, K1, H, W), generator=g).to(LP_DTYPE).float()
    w = (torch.randn((Cout, K1 + K2, 1, 1), generator=g) / np.sqrt(K1 + K2)).to(LP_DTYPE).float()
    b = torch.randn((Cout,), generator=g)
    dx = nhwc(x, LP_DTYPE)
    dw = w.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV)
    assert ops.conv1x1_packed_supported(dw)
    packed = ops.conv1x1_pack(dw)
    if K2:
        x2 = torch.randn((N, K2, H, W), generator=g).to(LP_DTYPE).float()
        dx2 = nhwc(x2, LP_DTYPE)
        ref = F.conv2d(torch.cat([x, x2], dim=1).to(DEV), w.to(DEV), bias=b.to(DEV))
    else:
        dx2 = None
        ref = F.conv2d(x.to(DEV), w.to(DEV), bias=b.to(DEV))
    ref = F.relu(ref) if relu else ref
    out = ops.conv1x1_packed(dx, packed, b.to(DEV), Cout, relu, x2=dx2)
    out_b = ops.conv1x1_packed(dx, packed, b.to(DEV), Cout, relu, x2=dx2)
    torch.cuda.synchronize()
    e = rel_err(out.float().permute(0, 3, 1, 2).cpu(), ref.cpu())
    assert e < (3e-3 if LP_DTYPE == torch.float16 else 2e-2), e
    assert torch.equal(out, out_b)
    if K2 == 0:
        other = ops.conv_bn_act(dx, dw, b.to(DEV), 1, 0, relu)
    elif ops.conv1x1_dual_supported(dx, dx2, dw.view(Cout, -1)):
        other = ops.conv1x1_dual(dx, dx2, dw.view(Cout, -1).contiguous(), b.to(DEV), relu)
    else:
        other = None
    d = None if other is None else (out.float() - other.float()).abs().max().item()
    print("conv1x1 packed", case, "vs fp32 %.3e | max |packed - other launch|" % e, d)
    assert other is None or torch.equal(out, other)
    if K2:   # the two-source form through the two-workgroups-per-CU kernel (conv1x1_duo.hip): the same bits
        assert torch.equal(ops.conv1x1_packed(dx, packed, b.to(DEV), Cout, relu, x2=dx2, duo=True), out)
    else:
        assert torch.equal(ops.conv1x1_packed_res(dx, packed, b.to(DEV), Cout, None, relu), out)
    with pytest.raises(_hip.HipKernelError):
        ops.call("agrl_conv1x1_packed_bn_act", ops.ptr(dx), None, ops.ptr(packed), ops.ptr(b.to(DEV)), ops.ptr(out), N * H * W, K1 + 64, 0, Cout, 1, None)


STRIDED_DUAL_CASES = [(8, 64, 32, 256, 128, 512, 2), (8, 32, 16, 512, 256, 1024, 2), (3, 7, 5, 128, 128, 256, 2), (2, 9, 6, 128, 256, 256, 3),
                      (2, 6, 4, 128, 256, 256, 1)]


@pytest.mark.parametrize("case", STRIDED_DUAL_CASES)
def test_conv1x1_packed_dual_strided(case):
    """First block of a strided layer (layers 2 / 3): conv3 + the stride-s 1x1 downsample conv as ONE GEMM over [block input sampled
    at the stride | conv2's output] (conv1x1_duo.hip, agrl_conv1x1_packed_dual_strided) against the fp32 reference (two convs + add +
    ReLU), bit for bit against the same kernel fed an explicitly gathered copy of the block input, and -- to the 16-bit rounding of
    the shortcut map that the fused form no longer makes -- against the two launches it replaces. The trunk's two shapes, odd map
    sizes with a ragged last tile (stride 2 and 3), stride 1. Every call twice."""
    from torchreid import hip_ops as ops
    N, Hi, Wi, K1, K2, Cout, s = case
    Ho, Wo = (Hi - 1) // s + 1, (Wi - 1) // s + 1
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn((N, K1, Hi, Wi), generator=g).relu().to(LP_DTYPE).float()
    y = torch.randn((N, K2, Ho, Wo), generator=g).relu().to(LP_DTYPE).float()
    wds = (torch.randn((Cout, K1, 1, 1), generator=g) / np.sqrt(K1)).to(LP_DTYPE).float()
    w3 = (torch.randn((Cout, K2, 1, 1), generator=g) / np.sqrt(K2)).to(LP_DTYPE).float()
    bds, b3 = torch.randn((Cout,), generator=g), torch.randn((Cout,), generator=g)
    ref = F.relu(F.conv2d(x.to(DEV), wds.to(DEV), bias=bds.to(DEV), stride=s) + F.conv2d(y.to(DEV), w3.to(DEV), bias=b3.to(DEV)))
    assert tuple(ref.shape) == (N, Cout, Ho, Wo)
    dx, dy = nhwc(x, LP_DTYPE), nhwc(y, LP_DTYPE)
    dwds = wds.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV)
    dw3 = w3.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV)
    dual = torch.cat([dwds.view(Cout, -1), dw3.view(Cout, -1)], dim=1).contiguous()
    packed = ops.conv1x1_pack(dual)
    bias = (bds + b3).to(DEV)
    out = ops.conv1x1_packed_dual_strided(dx, dy, packed, bias, Cout, s, True)
    out_b = ops.conv1x1_packed_dual_strided(dx, dy, packed, bias, Cout, s, True)
    gathered = ops.conv1x1_packed(dx[:, ::s, ::s].contiguous(), packed, bias, Cout, True, x2=dy, duo=True)
    shortcut = ops.conv_bn_act(dx, dwds, bds.to(DEV), s, 0, False)
    two = ops.conv_bn_act(dy, dw3, b3.to(DEV), 1, 0, True, residual=shortcut)
    torch.cuda.synchronize()
    e = rel_err(out.float().permute(0, 3, 1, 2).cpu(), ref.cpu())
    e2 = rel_err(out.float().cpu(), two.float().cpu())
    print("strided dual", case, "vs fp32 %.3e | vs the two launches %.3e" % (e, e2))
    assert e < (3e-3 if LP_DTYPE == torch.float16 else 2e-2), e
    assert e2 < (4e-3 if LP_DTYPE == torch.float16 else 3e-2), e2
    assert torch.equal(out, out_b) and torch.equal(out, gathered)
    with pytest.raises(_hip.HipKernelError):
        ops.call("agrl_conv1x1_packed_dual_strided", ops.ptr(dx), ops.ptr(dy), ops.ptr(packed), ops.ptr(bias), ops.ptr(out), N, Hi, Wi, s,
                 K1 + 64, K2, Cout, 1, None)


DUO_CASES = [(256, 16, 8, 512, 2048, True, True), (1, 16, 8, 512, 2048, True, True), (3, 10, 6, 128, 256, True, False),
             (37, 16, 8, 256, 1024, True, True), (5, 16, 8, 384, 512, False, True)]


@pytest.mark.parametrize("case", DUO_CASES)
def test_conv1x1_packed_res(case):
    """conv3 + identity shortcut + ReLU through the two-workgroups-per-CU kernel (conv1x1_duo.hip) against the fp32 reference and
    against conv_bn_act(residual=...) (same summation order, same order of bias / residual / ReLU / rounding: equal bit for bit).
    The full layer-4 shape (2048 tiles: four resident rounds), one frame (8 tiles), a ragged 180-row map without ReLU, an odd tile
    count, no residual with three slabs. Every call twice."""
    from torchreid import hip_ops as ops
    N, H, W, K, Cout, use_res, relu = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn((N, K, H, W), generator=g).relu().to(LP_DTYPE).float()
    w = (torch.randn((Cout, K, 1, 1), generator=g) / np.sqrt(K)).to(LP_DTYPE).float()
    b = torch.randn((Cout,), generator=g)
    res = torch.randn((N, Cout, H, W), generator=g).to(LP_DTYPE).float() if use_res else None
    dx = nhwc(x, LP_DTYPE)
    dw = w.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV)
    dres = None if res is None else nhwc(res, LP_DTYPE)
    packed = ops.conv1x1_pack(dw)
    ref = F.conv2d(x.to(DEV), w.to(DEV), bias=b.to(DEV))
    if res is not None:
        ref = ref + res.to(DEV)
    ref = F.relu(ref) if relu else ref
    out = ops.conv1x1_packed_res(dx, packed, b.to(DEV), Cout, dres, relu)
    out_b = ops.conv1x1_packed_res(dx, packed, b.to(DEV), Cout, dres, relu)
    other = ops.conv_bn_act(dx, dw, b.to(DEV), 1, 0, relu, residual=dres)
    torch.cuda.synchronize()
    e = rel_err(out.float().permute(0, 3, 1, 2).cpu(), ref.cpu())
    print("conv1x1 duo", case, "vs fp32 %.3e | max |duo - conv_bn_act| %.3e" % (e, (out.float() - other.float()).abs().max().item()))
    assert e < (3e-3 if LP_DTYPE == torch.float16 else 2e-2), e
    assert torch.equal(out, out_b)
    assert torch.equal(out, other)
    with pytest.raises(_hip.HipKernelError):
        ops.call("agrl_conv1x1_packed_res_bn_act", ops.ptr(dx), ops.ptr(packed), ops.ptr(b.to(DEV)), None, ops.ptr(out), N * H * W, K + 64, Cout, 1, None)


@pytest.mark.parametrize("case", ["res", "plain_k2048", "plain_k1024", "dual", "strided_l2", "strided_l3", "pool_parts", "pool_sum", "ragged", "two_slabs"])
def test_conv1x1_duo_persistent_form_is_bit_identical(case, monkeypatch):
    """The default dispatch since round 6 (conv1x1_duo_persist_kernel; AGRL_DUO_PERSIST=0 is the one-shot form it is compared with) -- two persistent workgroups per CU, the next tile's first slab and weight
    ring requested during the current tile's last slabs, the epilogue in two 64-row passes inside ONE pixel buffer) against the one-shot
    form on every shape the step sends through conv1x1_duo.hip, and a ragged tile count with an odd number of slabs (the buffer parity
    flips from tile to tile): equal BIT FOR BIT, every call twice (the second run starts with warm caches: other timing, same answer)."""
    from torchreid import hip_ops as ops
    g = torch.Generator().manual_seed(len(case))

    def rnd(shape, scale=1.0, relu=False):
        t = torch.randn(shape, generator=g) * scale
        return (t.relu() if relu else t).to(LP_DTYPE).to(DEV)

    def run():
        if case == "res":          # conv3 + residual stored (512 -> 2048, 2048 tiles)
            x, w, b, r = rnd((256, 16, 8, 512), relu=True), rnd((2048, 512), 0.05), rnd((2048,)).float(), rnd((256, 16, 8, 2048))
            pk = ops.conv1x1_pack(w)
            return [ops.conv1x1_packed_res(x, pk, b, 2048, r, True)]
        if case in ("plain_k2048", "plain_k1024"):   # layer 4's conv1s (no residual; 512 tiles: exactly one per slot -> engages only above 512)
            K = 2048 if case == "plain_k2048" else 1024
            x, w, b = rnd((520, 16, 8, K), relu=True), rnd((512, K), 0.03), rnd((512,)).float()
            pk = ops.conv1x1_pack(w)
            return [ops.conv1x1_packed_res(x, pk, b, 512, None, True)]
        if case == "dual":         # conv3 + downsample of layer 4's first block over [1024 | 512]
            x, y, w, b = rnd((256, 16, 8, 1024), relu=True), rnd((256, 16, 8, 512), relu=True), rnd((2048, 1536), 0.03), rnd((2048,)).float()
            pk = ops.conv1x1_pack(w)
            return [ops.conv1x1_packed(x, pk, b, 2048, True, x2=y, duo=True)]
        if case in ("strided_l2", "strided_l3"):
            N, Hi, Wi, K1, K2, Cout = (64, 64, 32, 256, 128, 512) if case == "strided_l2" else (128, 32, 16, 512, 256, 1024)
            x, y = rnd((N, Hi, Wi, K1), relu=True), rnd((N, Hi // 2, Wi // 2, K2), relu=True)
            w, b = rnd((Cout, K1 + K2), 0.04), rnd((Cout,)).float()
            pk = ops.conv1x1_pack(w)
            return [ops.conv1x1_packed_dual_strided(x, y, pk, b, Cout, 2, True)]
        if case in ("pool_parts", "pool_sum"):
            x, w, b, r = rnd((256, 16, 8, 512), relu=True), rnd((2048, 512), 0.05), rnd((2048,)).float(), rnd((256, 16, 8, 2048))
            pk = ops.conv1x1_pack(w)
            splits, mean = ([4, 2, 1], True) if case == "pool_parts" else ([1], False)
            return list(ops.conv1x1_packed_res_pool(x, pk, b, 2048, r, splits, mean, True))
        if case == "two_slabs":    # K = 256: the shortest pipeline the persistent form takes (the next tile's slab 0 is requested during slab 0)
            x, w, b, r = rnd((300, 16, 8, 256), relu=True), rnd((1024, 256), 0.06), rnd((1024,)).float(), rnd((300, 16, 8, 1024))
            pk = ops.conv1x1_pack(w)
            return [ops.conv1x1_packed_res(x, pk, b, 1024, r, True)]
        # ragged: 131 frames x 100 px (M = 13100: a partial last tile), K = 384 (three slabs: odd), Cout = 1280 (five channel tiles)
        x, w, b, r = rnd((131, 10, 10, 384), relu=True), rnd((1280, 384), 0.05), rnd((1280,)).float(), rnd((131, 10, 10, 1280))
        pk = ops.conv1x1_pack(w)
        return [ops.conv1x1_packed_res(x, pk, b, 1280, r, False)]

    monkeypatch.setenv("AGRL_DUO_PERSIST", "0")
    _hip.reload_options()
    g.manual_seed(len(case))
    ref = run()
    monkeypatch.delenv("AGRL_DUO_PERSIST")   # the default since the A/B of round 6: persistent
    _hip.reload_options()
    g.manual_seed(len(case))
    got = run()
    g.manual_seed(len(case))
    got2 = run()
    torch.cuda.synchronize()
    for a, b_, c in zip(ref, got, got2):
        assert torch.isfinite(a.float()).all()
        assert torch.equal(a, b_) and torch.equal(a, c), (case, (a.float() - b_.float()).abs().max().item())


@pytest.mark.parametrize("cfg", [(256, 512, 2048, [4, 2, 1], True), (256, 512, 2048, [1], False), (3, 128, 256, [4, 2, 1], True),
                                 (9, 256, 512, [2, 1], True)])
def test_conv1x1_packed_res_pool(cfg):
    """The pool-fused last conv of a layer-4 branch through the two-workgroups-per-CU kernel: pooled sums / means and the 16-bit
    copy equal BIT FOR BIT to agrl_conv1x1_bn_act_pool's (igemm_wide_kernel's pooled epilogue: same rounded activations, same order
    of the quarter sums) on the layer-4 shapes."""
    from torchreid import hip_ops as ops
    N, Cin, Cout, splits, mean = cfg
    g = torch.Generator().manual_seed(N + Cin)
    x = torch.randn((N, 16, 8, Cin), generator=g).relu().to(LP_DTYPE).to(DEV)
    w = (torch.randn((Cout, 1, 1, Cin), generator=g) / np.sqrt(Cin)).to(LP_DTYPE).to(DEV)
    b = torch.randn((Cout,), generator=g).to(DEV)
    res = torch.randn((N, 16, 8, Cout), generator=g).to(LP_DTYPE).to(DEV)
    packed = ops.conv1x1_pack(w)
    pooled, pooled_lp = ops.conv1x1_packed_res_pool(x, packed, b, Cout, res, splits, mean, True)
    pooled_b, _ = ops.conv1x1_packed_res_pool(x, packed, b, Cout, res, splits, mean, False)
    ref, ref_lp = ops.conv1x1_bn_act_pool(x, w, b, res, splits, mean, True)
    act = ops.conv_bn_act(x, w, b, 1, 0, True, residual=res)
    torch.cuda.synchronize()
    a = act.float().cpu()
    parts = []
    for n in splits:
        for j in range(n):
            lo, hi = (j * 16) // n, -(-((j + 1) * 16) // n)
            blk = a[:, lo:hi].reshape(N, -1, Cout)
            parts.append(blk.mean(1) if mean else blk.sum(1))
    e = rel_err(pooled, torch.stack(parts, 1))
    print("duo pool conv", cfg, "rel err vs pooled conv_bn_act map %.3e" % e)
    assert e < 1e-5
    assert torch.equal(pooled, pooled_b)
    if N >= 64:   # agrl_conv1x1_bn_act_pool through igemm_wide_kernel (small maps take the persistent igemm: another order of the sums)
        assert torch.equal(pooled, ref) and torch.equal(pooled_lp, ref_lp)
    else:
        assert rel_err(pooled, ref) < 1e-6 and rel_err(pooled_lp.float(), ref_lp.float()) < 2e-3
    import ctypes as C
    with pytest.raises(_hip.HipKernelError):   # bins that are not whole quarters
        arr3 = (C.c_int * 1)(3)
        ops.call("agrl_conv1x1_packed_res_pool", ops.ptr(x), ops.ptr(packed), ops.ptr(b), ops.ptr(res), ops.ptr(pooled), None,
                 N, 16, 8, Cin, Cout, 1, arr3, 1, 1, None)


@pytest.mark.parametrize("tile", ["2", "3"])
@pytest.mark.parametrize("case", [(3, 16, 8, 256, 512), (2, 32, 16, 512, 256), (1, 10, 6, 128, 256)])
def test_conv_wide_tile_strided(case, tile, monkeypatch):
    """1x1 stride-2 downsample convs through the wide igemm (gathered pixel rows) against the generic kernel."""
    from torchreid import hip_ops as ops
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn((N, Cin, H, W), generator=g).to(LP_DTYPE).float()
    w = (torch.randn((Cout, Cin, 1, 1), generator=g) / np.sqrt(Cin)).to(LP_DTYPE).float()
    b = torch.randn((Cout,), generator=g)
    ref = F.conv2d(x, w, bias=b, stride=2)
    args = (nhwc(x, LP_DTYPE), w.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV), b.to(DEV), 2, 0, False)
    monkeypatch.setenv("AGRL_IGEMM_WIDE", tile)
    _hip.reload_options()
    wide = ops.conv_bn_act(*args)
    monkeypatch.setenv("AGRL_IGEMM_WIDE", "0")
    _hip.reload_options()
    narrow = ops.conv_bn_act(*args)
    torch.cuda.synchronize()
    assert rel_err(wide.float().permute(0, 3, 1, 2), ref) < 1e-2
    assert torch.equal(wide, narrow)


DUAL_CASES = [(256, 16, 8, 1024, 512, 2048), (3, 16, 8, 1024, 512, 2048), (90, 16, 8, 128, 64, 512), (81, 16, 8, 512, 256, 256)]


@pytest.mark.parametrize("case", DUAL_CASES)
def test_conv1x1_dual_source(case):
    """agrl_conv1x1_dual_bn_act: a first Bottleneck's conv3 + its 1x1 stride-1 downsample conv as ONE GEMM over the
    concatenated K axis (vmgn.py:56-64), against the fp32 reference of the sum and against the two separate launches (which
    round the shortcut map to bf16 first): full-size layer-4 shape (1024 tiles, persistent walk), ragged M, 3 / 6 / 12 k-tiles,
    the source switch inside the ring."""
    from torchreid import hip_ops as ops
    N, H, W, K1, K2, Cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn((N, K1, H, W), generator=g).to(LP_DTYPE).float()
    y2 = torch.randn((N, K2, H, W), generator=g).relu().to(LP_DTYPE).float()
    wd = (torch.randn((Cout, K1, 1, 1), generator=g) / np.sqrt(K1)).to(LP_DTYPE).float()
    w3 = (torch.randn((Cout, K2, 1, 1), generator=g) / np.sqrt(K2)).to(LP_DTYPE).float()
    bd, b3 = torch.randn((Cout,), generator=g), torch.randn((Cout,), generator=g)
    ref = F.relu(F.conv2d(x, wd, bias=bd) + F.conv2d(y2, w3, bias=b3))
    xd, yd = nhwc(x, LP_DTYPE), nhwc(y2, LP_DTYPE)
    wcat = torch.cat([wd.view(Cout, K1), w3.view(Cout, K2)], dim=1).to(LP_DTYPE).to(DEV).contiguous()
    assert ops.conv1x1_dual_supported(xd, yd, wcat)
    out = ops.conv1x1_dual(xd, yd, wcat, (bd + b3).to(DEV), True)
    sc = ops.conv_bn_act(xd, wd.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV), bd.to(DEV), 1, 0, False)
    sep = ops.conv_bn_act(yd, w3.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV), b3.to(DEV), 1, 0, True, residual=sc)
    again = ops.conv1x1_dual(xd, yd, wcat, (bd + b3).to(DEV), True)
    torch.cuda.synchronize()
    e, es = rel_err(out.float().permute(0, 3, 1, 2), ref), rel_err(sep.float().permute(0, 3, 1, 2), ref)
    print("dual conv", case, "rel err %.3e (separate launches: %.3e)" % (e, es))
    assert e < 6e-3 and e <= es * 1.05 + 1e-6   # one bf16 rounding of the sum instead of two
    assert torch.equal(out, again)
    with pytest.raises(Exception):   # K1 != 2 K2 is not built: loud rejection, the caller keeps the two-launch form
        ops.conv1x1_dual(xd, xd, torch.cat([wcat[:, :K1], wcat[:, :K1]], 1).contiguous(), (bd + b3).to(DEV), True)


@pytest.mark.parametrize("path", ["wide", "persistent"])
@pytest.mark.parametrize("cfg", [(6, 512, 512, [4, 2, 1], True), (5, 256, 768, [1], False), (4, 2048, 256, [4, 2, 1], True)])
def test_conv1x1_pool_fused(cfg, path, monkeypatch):
    """Last conv of a layer4 branch with the frame pooling fused into its epilogue (the 2048-channel map is never
    written) against the unfused pair conv_bn_act -> part_pool, both igemm forms."""
    from torchreid import hip_ops as ops
    N, Cin, Cout, splits, mean = cfg
    g = torch.Generator().manual_seed(N + Cin)
    x = torch.randn((N, 16, 8, Cin), generator=g).to(LP_DTYPE).to(DEV)
    w = (torch.randn((Cout, 1, 1, Cin), generator=g) / np.sqrt(Cin)).to(LP_DTYPE).to(DEV)
    b = torch.randn((Cout,), generator=g).to(DEV)
    res = torch.randn((N, 16, 8, Cout), generator=g).to(LP_DTYPE).to(DEV)
    monkeypatch.setenv("AGRL_POOL_PERSIST", "1" if path == "persistent" else "0")
    _hip.reload_options()
    if path == "wide":
        monkeypatch.setenv("AGRL_IGEMM_WIDE", "1")
        _hip.reload_options()
    pooled, pooled_lp = ops.conv1x1_bn_act_pool(x, w, b, res, splits, mean, True)
    monkeypatch.delenv("AGRL_IGEMM_WIDE", raising=False)
    _hip.reload_options()
    act = ops.conv_bn_act(x, w, b, 1, 0, True, residual=res)
    torch.cuda.synchronize()
    a = act.float().cpu()  # (N,16,8,C): pool the bf16 activations exactly as the reference pools its map
    parts = []
    for n in splits:
        for j in range(n):
            lo, hi = (j * 16) // n, -(-((j + 1) * 16) // n)
            blk = a[:, lo:hi].reshape(N, -1, Cout)
            parts.append(blk.mean(1) if mean else blk.sum(1))
    ref = torch.stack(parts, 1)
    e = rel_err(pooled, ref)
    print("fused pool conv", cfg, path, "rel err %.3e" % e)
    assert e < 1e-5
    assert rel_err(pooled_lp.float(), ref) < 5e-3


@pytest.mark.parametrize("case", [(2, 32, 16, True), (3, 16, 8, False), (1, 64, 32, True), (5, 16, 24, True)])
def test_conv3x3_c64_resident_weights(case, monkeypatch):
    """Layer-1 3x3 (64 -> 64) persistent kernel with all nine taps' weights resident in LDS vs the one-block kernel."""
    from torchreid import hip_ops as ops
    N, H, W, relu = case
    g = torch.Generator().manual_seed(N * H + W)
    x = torch.randn((N, 64, H, W), generator=g).to(LP_DTYPE).float()
    w = (torch.randn((64, 64, 3, 3), generator=g) / 24).to(LP_DTYPE).float()
    b = torch.randn((64,), generator=g)
    ref = F.conv2d(x, w, bias=b, padding=1)
    if relu:
        ref = F.relu(ref)
    args = (nhwc(x, LP_DTYPE), w.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV), b.to(DEV), 1, 1, relu)
    monkeypatch.setenv("AGRL_CONV3X3_C64", "1")
    _hip.reload_options()
    fast = ops.conv_bn_act(*args)
    monkeypatch.setenv("AGRL_CONV3X3_C64", "0")
    _hip.reload_options()
    base = ops.conv_bn_act(*args)
    torch.cuda.synchronize()
    assert rel_err(fast.float().permute(0, 3, 1, 2), ref) < 1e-2
    assert torch.equal(fast, base)


WIDE3_CASES = [
    # N, H, W, Cin, Cout, relu -- the two-block 3x3 kernel (conv3x3_wide.hip), forced through AGRL_CONV3X3_WIDE=1
    (3, 16, 8, 128, 128, True),     # odd number of pixel blocks: the last workgroup has one valid block
    (1, 64, 32, 128, 256, True),    # 16 blocks per frame, two channel tiles, 2 slabs
    (3, 16, 8, 512, 200, False),    # ragged N, 8 slabs (72 tap-steps), no relu
    (4, 16, 8, 64, 128, True),      # a single slab: no next-slab patch pieces in the queue
]


@pytest.mark.parametrize("case", WIDE3_CASES)
def test_conv3x3_wide_tile(case, monkeypatch):
    from torchreid import hip_ops as ops
    N, H, W, Cin, Cout, relu = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn((N, Cin, H, W), generator=g).to(LP_DTYPE).float()
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) / np.sqrt(9 * Cin)).to(LP_DTYPE).float()
    b = torch.randn((Cout,), generator=g)
    ref = F.conv2d(x, w, bias=b, padding=1)
    if relu:
        ref = F.relu(ref)
    args = (nhwc(x, LP_DTYPE), w.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV), b.to(DEV), 1, 1, relu)
    monkeypatch.setenv("AGRL_CONV3X3_WIDE", "1")
    _hip.reload_options()
    wide = ops.conv_bn_act(*args)
    monkeypatch.setenv("AGRL_CONV3X3_WIDE", "0")
    _hip.reload_options()
    narrow = ops.conv_bn_act(*args)
    torch.cuda.synchronize()
    e = rel_err(wide.float().permute(0, 3, 1, 2), ref)
    print("wide 3x3", case, "rel err %.3e" % e)
    assert e < 1e-2
    assert torch.equal(wide, narrow)  # same k order per output -> identical bf16 results


@pytest.mark.parametrize("case", [(385, 16, 8, 128, 512, True), (192, 32, 8, 64, 256, False), (400, 16, 8, 512, 768, True)])
def test_conv3x3_wide_256_channel_tiles(case, monkeypatch):
    """The 256-channel-tile form of the two-block 3x3 kernel (64 x 128 wave tiles, two weight slots, out tile over patches +
    weight ring: layer 4's 512 -> 512 convs at the bench size) is taken when Cout % 256 == 0 and >= 192 such tiles exist:
    odd block count (the last workgroup has one valid block), two blocks per frame, three channel tiles -- against the
    reference conv and BITWISE against the 128-channel-tile form (same k order per output)."""
    from torchreid import hip_ops as ops
    N, H, W, Cin, Cout, relu = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    x = torch.randn((N, Cin, H, W), generator=g).to(LP_DTYPE).float()
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) / np.sqrt(9 * Cin)).to(LP_DTYPE).float()
    b = torch.randn((Cout,), generator=g)
    args = (nhwc(x, LP_DTYPE), w.permute(0, 2, 3, 1).contiguous().to(LP_DTYPE).to(DEV), b.to(DEV), 1, 1, relu)
    monkeypatch.setenv("AGRL_CONV3X3_WIDE", "1")
    monkeypatch.delenv("AGRL_CONV3X3_N128", raising=False)
    _hip.reload_options()
    t256 = ops.conv_bn_act(*args)
    monkeypatch.setenv("AGRL_CONV3X3_N128", "1")
    _hip.reload_options()
    t128 = ops.conv_bn_act(*args)
    monkeypatch.delenv("AGRL_CONV3X3_N128")
    _hip.reload_options()
    torch.cuda.synchronize()
    rows = torch.randperm(N, generator=g)[:24]          # the reference conv of a sample of frames (CPU time)
    ref = F.conv2d(x[rows], w, bias=b, padding=1)
    if relu:
        ref = F.relu(ref)
    e = rel_err(t256[rows.to(DEV)].float().permute(0, 3, 1, 2), ref)
    print("3x3, 256-channel tiles", case, "rel err %.3e" % e)
    assert e < 1e-2
    assert torch.equal(t256, t128)


@pytest.mark.parametrize("dtype", [torch.float32, LP_DTYPE])
@pytest.mark.parametrize("shape", [(2, 256, 128), (3, 64, 48), (1, 37, 29)])
def test_stem(shape, dtype):
    from torchreid import hip_ops as ops
    N, H, W = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn((N, 3, H, W), generator=g)
    w = torch.randn((64, 3, 7, 7), generator=g) * 0.1
    b = torch.randn((64,), generator=g) * 0.1
    ref = F.max_pool2d(F.relu(F.conv2d(x, w, bias=b, stride=2, padding=3)), 3, 2, 1)
    out = ops.stem(x.to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV), b.to(DEV), dtype)
    torch.cuda.synchronize()
    e = rel_err(out.float().permute(0, 3, 1, 2), ref)
    print("stem", shape, dtype, "rel err %.3e" % e)
    assert e < (1e-5 if dtype == torch.float32 else 8e-3)


@pytest.mark.parametrize("shape", [(2, 256, 128), (3, 64, 48), (1, 37, 29), (1, 224, 112)])
def test_stem_bf16_mfma(shape):
    from torchreid import hip_ops as ops
    N, H, W = shape
    g = torch.Generator().manual_seed(6)
    x = torch.randn((N, 3, H, W), generator=g)
    w = torch.randn((64, 3, 7, 7), generator=g) * 0.1
    b = torch.randn((64,), generator=g) * 0.1
    # the kernel rounds pixels and weights to bf16 and accumulates in fp32: so does the reference here
    ref = F.max_pool2d(F.relu(F.conv2d(x.to(LP_DTYPE).float(), w.to(LP_DTYPE).float(), bias=b, stride=2, padding=3)), 3, 2, 1)
    wpk = ops.pack_stem_weights_lp16(w.permute(0, 2, 3, 1).contiguous().to(DEV))
    out = ops.stem_lp16(x.to(DEV), wpk, b.to(DEV))
    torch.cuda.synchronize()
    e = rel_err(out.float().permute(0, 3, 1, 2), ref)
    print("stem bf16 mfma", shape, "rel err %.3e" % e)
    assert e < 5e-3  # one bf16 rounding of the output


@pytest.mark.parametrize("shape", [(8, 64, 48), (10, 64, 48), (33, 256, 128), (64, 128, 64)])
def test_stem_16bit_tile_order_and_lds_forms(shape, monkeypatch):
    """The 16-bit stem walks its tiles frame by frame per XCD from 8 frames on (round 5: HBM fetch 300 -> 101 MB per launch) and keeps
    the patch and the conv tile in separate LDS regions (two barriers per tile): frame counts that give every XCD the same number of
    frames, uneven ones (10, 33: XCDs with one frame more), a grid above the 512 persistent workgroups -- against the fp32 reference,
    and bit for bit against the launch-order / overlaid-region forms (AGRL_STEM_XCD_MAP=0, AGRL_STEM_SPLIT_LDS=0)."""
    from torchreid import hip_ops as ops
    N, H, W = shape
    g = torch.Generator().manual_seed(N + H)
    x = torch.randn((N, 3, H, W), generator=g)
    w = torch.randn((64, 3, 7, 7), generator=g) * 0.1
    b = torch.randn((64,), generator=g) * 0.1
    ref = F.max_pool2d(F.relu(F.conv2d(x.to(LP_DTYPE).float(), w.to(LP_DTYPE).float(), bias=b, stride=2, padding=3)), 3, 2, 1)
    wpk = ops.pack_stem_weights_lp16(w.permute(0, 2, 3, 1).contiguous().to(DEV))
    dx, db = x.to(DEV), b.to(DEV)
    out = ops.stem_lp16(dx, wpk, db)
    again = ops.stem_lp16(dx, wpk, db)
    torch.cuda.synchronize()
    e = rel_err(out.float().permute(0, 3, 1, 2), ref)
    print("stem 16-bit", shape, "rel err %.3e" % e)
    assert e < 5e-3 and torch.equal(out, again)
    for var in ("AGRL_STEM_XCD_MAP", "AGRL_STEM_SPLIT_LDS"):
        monkeypatch.setenv(var, "0")
        _hip.reload_options()
        other = ops.stem_lp16(dx, wpk, db)
        torch.cuda.synchronize()
        assert torch.equal(out, other), var
        monkeypatch.delenv(var)
        _hip.reload_options()


@pytest.mark.parametrize("dtype", [torch.float32, LP_DTYPE])
@pytest.mark.parametrize("cfg", [(2, 4, 16, 8, 2048, [4, 2, 1]), (1, 3, 14, 7, 512, [4, 2, 1]), (2, 2, 16, 8, 256, [8, 4, 2, 1]), (1, 2, 16, 8, 256, [4])])
def test_part_pool(cfg, dtype):
    from torchreid import hip_ops as ops
    B, S, h, w, C, splits = cfg
    g = torch.Generator().manual_seed(11)
    x41 = torch.rand((B * S, C, h, w), generator=g)
    x42 = torch.rand((B * S, C, h, w), generator=g)
    if dtype == LP_DTYPE:
        x41, x42 = x41.to(LP_DTYPE).float(), x42.to(LP_DTYPE).float()
    gsum, nodes, nodes_lp = ops.part_pool(nhwc(x41, dtype), nhwc(x42, dtype), splits, want_lp=True)
    torch.cuda.synchronize()
    ref_nodes = O.part_nodes(x42, B, S, splits)
    ref_g = O.global_feature(x41, B, S)
    e1 = rel_err(nodes.view(B, -1, C), ref_nodes)
    e2 = rel_err(gsum.view(B, S, C).sum(1) / (S * h * w), ref_g)
    e3 = rel_err(nodes_lp.float().view(B, -1, C), ref_nodes)
    print("part_pool", cfg, dtype, "%.3e %.3e %.3e" % (e1, e2, e3))
    assert e1 < 1e-5 and e2 < 1e-5 and e3 < 5e-3


@pytest.mark.parametrize("V", [28, 56, 112, 20])
@pytest.mark.parametrize("mode", [(True, True), (True, False), (False, True)])
def test_graph_layer(V, mode):
    from torchreid import hip_ops as ops
    from recipe import synthetic_adj
    use_pose, learn_graph = mode
    B, C = 3, 2048
    g = torch.Generator().manual_seed(V)
    # node features: correlated positive vectors, some pairs close enough that the learned graph is not the identity
    base = torch.rand((B, 1, C), generator=g)
    f = base + 0.02 * torch.randn((B, V, C), generator=g)
    if V % 7 == 0:
        adj = synthetic_adj(B, V // 7, seed=V)
    else:
        adj = (torch.rand((B, V, V), generator=g) > 0.5).float()
        adj[0, 3] = 0  # an all-zero row
    W = torch.randn((C, C), generator=g) * 0.02
    sd = {"gl.linear.weight": W, "gl.bn.weight": 0.8 + 0.4 * torch.rand(C, generator=g), "gl.bn.bias": 0.1 * torch.randn(C, generator=g),
          "gl.bn.running_mean": 0.1 * torch.randn(C, generator=g), "gl.bn.running_var": 0.5 + torch.rand(C, generator=g)}
    ref = O.graph_layer(f, adj, sd, "gl", use_pose, learn_graph)
    refG = O.graph_matrix(f, adj, use_pose, learn_graph)
    fd, adjd = f.to(DEV), adj.to(DEV)
    h = ops.linear_nobias(fd.view(B * V, C), W.to(DEV)).view(B, V, C)
    G = ops.graph_matrix(fd, adjd, use_pose, learn_graph)
    scale = sd["gl.bn.weight"] / torch.sqrt(sd["gl.bn.running_var"] + 1e-5)
    shift = sd["gl.bn.bias"] - sd["gl.bn.running_mean"] * scale
    out, out_lp = ops.graph_propagate(fd, h, G, scale.to(DEV), shift.to(DEV), 0.1, 0.1, want_lp=True)
    torch.cuda.synchronize()
    # fp64 ground truth: in fp32 the diagonal d2_ii = n_i + n_i - 2 g_ii is pure cancellation noise (its value
    # depends on the BLAS summation order), and sqrt() amplifies it to a ~1e-2 relative change of sim_ii.
    # The kernel takes n_i from the Gram diagonal (d2_ii == 0 exactly), i.e. the exact-arithmetic value.
    sd64 = {k: v.double() for k, v in sd.items()}
    ref64 = O.graph_layer(f.double(), adj.double(), sd64, "gl", use_pose, learn_graph)
    refG64 = O.graph_matrix(f.double(), adj.double(), use_pose, learn_graph)
    eh = rel_err(h, f.double() @ W.double().t())
    eG, eG32 = rel_err(G, refG64), rel_err(G, refG)
    eo, eo32 = rel_err(out, ref64), rel_err(out, ref)
    em = rel_err(out.cpu().double() - 0.9 * f.double(), ref64 - 0.9 * f.double())
    print("graph V=%d mode=%s  h %.3e | vs fp64: G %.3e out %.3e msg %.3e | vs fp32 oracle: G %.3e out %.3e | oracle32 vs 64: G %.3e | offdiag mass %.3e" % (
        V, mode, eh, eG, eo, em, eG32, eo32, rel_err(refG, refG64), (refG.sum(2) - refG.diagonal(dim1=1, dim2=2)).mean().item()))
    # G: node features are deliberately close (d2 ~ 1.6 from norms ~ 680), so fp32 d2 itself carries ~1e-4 rel
    assert eh < 1e-5 and eG < 5e-4 and eo < 1e-5 and em < 1e-4
    assert eo32 < 1e-3  # the north-star bar against the fp32 reference arithmetic
    assert rel_err(out_lp.float(), ref) < 5e-3
    if learn_graph:
        # G against the fp32 oracle itself, OFF the diagonal: the one documented deviation (d2_ii taken as exactly 0 instead
        # of the reference's fp32 cancellation noise) changes sim_ii and, through the row-L1 normalisation, rescales the
        # whole row; the off-diagonal PROFILE of the learned part -- each row's off-diagonal entries over their own sum --
        # does not depend on the diagonal at all and must agree with the reference arithmetic at the north-star bar.
        def offdiag_profile(Gm):
            Gm = Gm.detach().cpu().float()
            learned = 2 * Gm - torch.nn.functional.normalize(adj, p=1, dim=2) if use_pose else Gm
            learned = learned - torch.diag_embed(learned.diagonal(dim1=1, dim2=2))
            return learned / learned.sum(dim=2, keepdim=True).clamp(min=1e-30)
        ep = rel_err(offdiag_profile(G), offdiag_profile(refG))
        print("   off-diagonal profile of the learned graph vs the fp32 oracle: %.3e (whole G incl. diagonal: %.3e)" % (ep, eG32))
        assert ep < 1e-3
        assert eG32 < 3e-2   # the diagonal noise of the fp32 reference, not a kernel error (fp64 check above: eG < 5e-4)


@pytest.mark.parametrize("mode", [(True, True), (True, False), (False, True)])
@pytest.mark.parametrize("shape", [(3, 28, 2048), (32, 56, 2048), (2, 112, 2048), (2, 128, 2048), (2, 240, 1024), (5, 60, 256), (3, 49, 512)])
def test_graph_layer_commuted(shape, mode):
    """The default eval GraphLayer: graph -> P = G f (agrl_graph_apply) -> ONE GEMM with BatchNorm1d + LeakyReLU + residual mix
    in its epilogue (agrl_graph_linear_mix), i.e. (G f) W^T instead of G (f W^T) (vmgn.py:142-172), against the fp32 oracle of the
    reference's order of operations: exact-fp32 mode <= 1e-5 (the two orders differ by fp32 roundoff only), split-bf16 mode
    <= 2e-4, bf16 mode <= 1e-2; V in {28, 56} x the bench batch, 112 / 128 (the LDS-resident message-pass kernel), 240 (the
    general kernel), 60 (a width the streaming kernel takes with a different fragment count), 49 (V % 4 != 0)."""
    from torchreid import hip_ops as ops
    from recipe import synthetic_adj
    B, V, C = shape
    use_pose, learn_graph = mode
    g = torch.Generator().manual_seed(B * 1000 + V + C)
    base = torch.rand((B, 1, C), generator=g)
    f = base + 0.02 * torch.randn((B, V, C), generator=g)
    adj = synthetic_adj(B, V // 7, seed=V) if V % 7 == 0 else (torch.rand((B, V, V), generator=g) > 0.5).float()
    sd = {"gl.linear.weight": torch.randn((C, C), generator=g) * 0.02, "gl.bn.weight": 0.8 + 0.4 * torch.rand(C, generator=g),
          "gl.bn.bias": 0.1 * torch.randn(C, generator=g), "gl.bn.running_mean": 0.1 * torch.randn(C, generator=g),
          "gl.bn.running_var": 0.5 + torch.rand(C, generator=g)}
    ref = O.graph_layer(f, adj, sd, "gl", use_pose, learn_graph)
    ref64 = O.graph_layer(f.double(), adj.double(), {k: v.double() for k, v in sd.items()}, "gl", use_pose, learn_graph)
    scale = (sd["gl.bn.weight"] / torch.sqrt(sd["gl.bn.running_var"] + 1e-5)).to(DEV)
    shift = (sd["gl.bn.bias"] - sd["gl.bn.running_mean"] * scale.cpu()).to(DEV)
    fd, adjd = f.to(DEV), adj.to(DEV)
    G = ops.graph_matrix(fd, adjd, use_pose, learn_graph)
    errs = {}
    for name, dt, split in (("fp32", torch.float32, False), ("bf16x3", torch.float32, True), (LP16, LP_DTYPE, False)):
        with ops.f32_split(split):
            P = ops.graph_apply_operand(G, fd, dt)
            out = ops.graph_linear_mix(P, sd["gl.linear.weight"].to(dt).to(DEV), fd, scale, shift, 0.1, 0.1)
        torch.cuda.synchronize()
        assert P.dtype == dt and tuple(P.shape) == (B, V, C)
        # the message term alone (out - 0.9 f) carries the whole layer arithmetic at a tenth of the weight
        errs[name] = (rel_err(out, ref), rel_err(out.cpu().double() - 0.9 * f.double(), ref64 - 0.9 * f.double()))
    print("commuted graph layer", shape, mode, {k: "%.2e / msg %.2e" % v for k, v in errs.items()})
    assert errs["fp32"][0] < 1e-5 and errs["fp32"][1] < 2e-4
    assert errs["bf16x3"][0] < 2e-4
    assert errs[LP16][0] < 1e-2


@pytest.mark.parametrize("cfg", [(5, 56, 2048, True, True, False), (300, 28, 512, True, True, False), (3, 64, 512, False, True, False),
                                 (4, 20, 1024, True, False, False), (6, 56, 512, True, True, True)])
def test_graph_tracklet_operand(cfg):
    """agrl_graph_tracklet_operand (Gram -> graph -> P = G f, one workgroup per tracklet, one launch: the form the model takes at
    >= 224 tracklets per GPU) against the three-launch form gram + finalize + apply and against the fp64 oracle: graph to fp32
    roundoff (the Gram is summed in another order), P in fp32 and bf16; more tracklets than CUs, V = 64 (full fragments), V = 20,
    pose-only / learned-only graphs, ganet's masked diagonal; bitwise repeatable."""
    from torchreid import hip_ops as ops
    from recipe import synthetic_adj
    B, V, C, use_pose, learn_graph, masked = cfg
    g = torch.Generator().manual_seed(B + V + C)
    f = torch.rand((B, 1, C), generator=g) + 0.02 * torch.randn((B, V, C), generator=g)
    adj = synthetic_adj(B, V // 7, seed=V) if V % 7 == 0 else (torch.rand((B, V, V), generator=g) > 0.5).float()
    fd, adjd = f.to(DEV), adj.to(DEV)
    G3 = ops.graph_matrix(fd, adjd, use_pose, learn_graph, mask_diag=masked)
    P3 = ops.graph_apply_operand(G3, fd, torch.float32)
    P, G = ops.graph_tracklet_operand(fd, adjd, use_pose, learn_graph, torch.float32, want_graph=True, mask_diag=masked)
    P2, _ = ops.graph_tracklet_operand(fd, adjd, use_pose, learn_graph, torch.float32, mask_diag=masked)
    Plp, _ = ops.graph_tracklet_operand(fd, adjd, use_pose, learn_graph, LP_DTYPE, mask_diag=masked)
    torch.cuda.synchronize()
    if not masked:
        G64 = O.graph_matrix(f.double(), adj.double(), use_pose, learn_graph)
        assert rel_err(G, G64) < 5e-4 and rel_err(P, torch.bmm(G64, f.double())) < 1e-5
    eG, eP = rel_err(G, G3), rel_err(P, P3)
    print("tracklet form", cfg, "G vs three-launch form %.2e, P %.2e" % (eG, eP))
    assert eG < (5e-4 if learn_graph else 1e-7) and eP < 1e-5
    assert torch.equal(P, P2) and rel_err(Plp.float(), P) < 5e-3


@pytest.mark.parametrize("dtype", [torch.float32, LP_DTYPE])
@pytest.mark.parametrize("shape", [(5, 16, 8, 256, 32, [4, 2, 1]), (3, 12, 8, 512, 64, [4, 2, 1]), (2, 16, 8, 128, 32, [2]), (2, 6, 4, 64, 32, [4])])
def test_pam_pool(shape, dtype):
    """agrl_pam_pool + the folded value conv (one Linear on the attention-weighted slice mean) + agrl_pam_combine against
    the oracle's literal PAM_Module (ganet.py:98-136: value conv on every position, attention, pooling), per pyramid slice
    incl. h // n slices that drop remainder rows (h = 6, n = 4) and the gamma = 0 form."""
    from torchreid import hip_ops as ops
    Fr, h, w, C, Cq, splits = shape
    g = torch.Generator().manual_seed(Fr + h + C)
    x = torch.randn((Fr, C, h, w), generator=g)
    sd = {"pam.query_conv.weight": 0.3 * torch.randn((Cq, C, 1, 1), generator=g) / np.sqrt(C) * 4, "pam.query_conv.bias": 0.1 * torch.randn(Cq, generator=g),
          "pam.key_conv.weight": 0.3 * torch.randn((Cq, C, 1, 1), generator=g) / np.sqrt(C) * 4, "pam.key_conv.bias": 0.1 * torch.randn(Cq, generator=g),
          "pam.value_conv.weight": torch.randn((C, C, 1, 1), generator=g) / np.sqrt(C), "pam.value_conv.bias": 0.1 * torch.randn(C, generator=g),
          "pam.gamma": torch.tensor([0.7])}
    if dtype == LP_DTYPE:   # the same rounded operands on both sides
        x = x.to(LP_DTYPE).float()
        for k in ("pam.query_conv.weight", "pam.key_conv.weight", "pam.value_conv.weight"):
            sd[k] = sd[k].to(LP_DTYPE).float()
    ref = O.ganet_nodes(x, {k.replace("pam.", "pam_layer."): v for k, v in sd.items()}, Fr, 1, splits).view(Fr, sum(splits), C)
    xd = nhwc(x, dtype)
    qk_w = torch.cat([sd["pam.query_conv.weight"], sd["pam.key_conv.weight"]], 0).permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)
    qk_b = torch.cat([sd["pam.query_conv.bias"], sd["pam.key_conv.bias"]], 0).to(DEV)
    qk = ops.conv_bn_act(xd, qk_w, qk_b, 1, 0, False)
    xbar, xmean = ops.pam_pool(xd, qk, splits)
    y = ops.linear_nobias(xbar.to(dtype).view(-1, C), sd["pam.value_conv.weight"].view(C, C).to(dtype).to(DEV))
    nodes, nodes_lp = ops.pam_combine(y, sd["pam.value_conv.bias"].to(DEV), xmean, 0.7, want_lp=True)
    _, xmean0 = ops.pam_pool(xd, None, splits)
    nodes0, _ = ops.pam_combine(None, None, xmean0, 0.0, want_lp=False)
    torch.cuda.synchronize()
    e = rel_err(nodes, ref)
    sd0 = {k.replace("pam.", "pam_layer."): v for k, v in sd.items()}
    sd0["pam_layer.gamma"] = torch.zeros(1)
    e0 = rel_err(nodes0, O.ganet_nodes(x, sd0, Fr, 1, splits).view(Fr, sum(splits), C))
    print("pam_pool", shape, dtype, "rel err %.3e (gamma = 0 form %.3e)" % (e, e0))
    assert e < (2e-2 if dtype == LP_DTYPE else 1e-4) and e0 < 1e-5
    assert torch.equal(xmean, xmean0) and rel_err(nodes_lp.float(), nodes) < 5e-3


def test_attention_tail():
    from torchreid import hip_ops as ops
    B, S, P, C, hw = 3, 8, 7, 2048, 128
    g = torch.Generator().manual_seed(3)
    nodes = torch.rand((B, S, P, C), generator=g)
    nodes[1, 2] = 0  # a frame whose nodes are all zero
    gsum = torch.rand((B * S, C), generator=g) * hw
    sd = {}
    for name in ("global_bottleneck", "att_bottleneck"):
        sd[name + ".weight"] = 0.8 + 0.4 * torch.rand(C, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(C, generator=g)
        sd[name + ".running_mean"] = 0.1 * torch.randn(C, generator=g)
        sd[name + ".running_var"] = 0.5 + torch.rand(C, generator=g)
    att_f = O.attention_pool(nodes)
    g_f = gsum.view(B, S, C).sum(1) / (S * hw)
    ref = torch.cat([O._bn(g_f, sd, "global_bottleneck"), O._bn(att_f, sd, "att_bottleneck")], 1)

    def fold(n):
        sc = sd[n + ".weight"] / torch.sqrt(sd[n + ".running_var"] + 1e-5)
        return sc.to(DEV), (sd[n + ".bias"] - sd[n + ".running_mean"] * sc).to(DEV)

    nd = nodes.to(DEV)
    sqn = ops.row_sqnorm(nd.view(B * S * P, C))
    gs, gsh = fold("global_bottleneck")
    as_, ash = fold("att_bottleneck")
    out, gf, af = ops.attn_pool_bnneck(nd, sqn, gsum.to(DEV), gs, gsh, as_, ash, B, S, P, hw, want_feats=True)
    torch.cuda.synchronize()
    print("tail", rel_err(out, ref), rel_err(gf, g_f), rel_err(af, att_f))
    assert rel_err(out, ref) < 1e-5 and rel_err(gf, g_f) < 1e-5 and rel_err(af, att_f) < 1e-5


@pytest.mark.parametrize("cfg", [(32, 8, 7, 2048, 128), (3, 8, 7, 2048, 128), (2, 16, 7, 2048, 128), (5, 4, 3, 512, 60), (1, 1, 7, 256, 128)])
def test_attention_tail_one_launch(cfg):
    """agrl_attn_tail = agrl_row_sqnorm (node norms) + agrl_attn_pool_bnneck + the distance matrix's query operand (agrl_row_sqnorm and
    agrl_row_l2_normalize over the embedding rows, both operand types) in ONE launch: every output BIT-IDENTICAL to the separate
    launches (each sum runs in the order of the kernel it replaces) -- bench shape, a zero frame, seq_len 16, small widths, one frame."""
    from torchreid import hip_ops as ops
    B, S, P, C, hw = cfg
    g = torch.Generator().manual_seed(B * 7 + S)
    nodes = torch.rand((B, S, P, C), generator=g)
    if S > 2:
        nodes[B // 2, 2] = 0  # a frame whose nodes are all zero
    nd = nodes.to(DEV)
    gsum = (torch.rand((B * S, C), generator=g) * hw).to(DEV)
    gs, gsh = (0.8 + 0.4 * torch.rand(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
    as_, ash = (0.8 + 0.4 * torch.rand(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
    assert ops.attn_tail_supported(S, P, C)
    sqn = ops.row_sqnorm(nd.view(B * S * P, C))
    out0, gf0, af0 = ops.attn_pool_bnneck(nd, sqn, gsum, gs, gsh, as_, ash, B, S, P, hw, want_feats=True)
    for qdt in (LP_DTYPE, torch.float32):
        out, feats, query, node_sqn = ops.attn_tail(nd, gsum, gs, gsh, as_, ash, B, S, P, hw, want_feats=True, query_dtype=qdt, want_node_sqn=True)
        torch.cuda.synchronize()
        assert torch.equal(node_sqn, sqn)
        assert torch.equal(out, out0) and torch.equal(feats[0], gf0) and torch.equal(feats[1], af0)
        assert torch.equal(query['sqn'], ops.row_sqnorm(out0))
        assert torch.equal(query['normalized'], ops.row_l2_normalize(out0, True, qdt))
    out_plain, feats_none, q_none, _ = ops.attn_tail(nd, gsum, gs, gsh, as_, ash, B, S, P, hw)
    assert torch.equal(out_plain, out0) and feats_none is None and q_none is None
    with pytest.raises(_hip.HipKernelError):
        ops.call("agrl_attn_tail", ops.ptr(nd), ops.ptr(gsum), ops.ptr(gs), ops.ptr(gsh), ops.ptr(as_), ops.ptr(ash), ops.ptr(out), None, None,
                 None, None, None, None, B, S, P, C + 2, hw, None)


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
@pytest.mark.parametrize("shape", [(37, 101, 4096), (5, 300, 96), (130, 257, 2048), (32, 12180, 4096), (8, 3000, 1024), (50, 2500, 512), (20, 2077, 256), (64, 4099, 128)])
def test_distmat(shape, metric):
    from torchreid.metrics.distance import hip_distmat_device
    m, n, D = shape
    g = torch.Generator().manual_seed(m)
    q = torch.randn((m, D), generator=g)
    gal = torch.randn((n, D), generator=g)
    ref = O.euclidean_squared(q.double(), gal.double()) if metric == "euclidean" else O.cosine(q.double(), gal.double())
    got = hip_distmat_device(q.to(DEV), gal.to(DEV), metric, "fp32")
    got_lp = hip_distmat_device(q.to(DEV), gal.to(DEV), metric, LP16)
    torch.cuda.synchronize()
    e, elp = rel_err(got, ref), rel_err(got_lp, ref)
    print("distmat", shape, metric, "fp32 %.3e bf16 %.3e" % (e, elp))
    assert e < 1e-5 and elp < 1e-2


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
@pytest.mark.parametrize("shape", [(2000, 7428, 512), (1793, 8190 + 2, 576)])
def test_distmat_wide_tile_f32out(shape, metric, monkeypatch):
    """The full query x gallery form through the 256 x 256 tile with fp32 output (igemm_wide_kernel<65536>: ragged M and N, both
    metrics' epilogues) against the tiled igemm form it replaces (AGRL_DISTMAT_TILED=1) and the fp64 oracle."""
    from torchreid.metrics.distance import hip_distmat_device
    m, n, D = shape
    g = torch.Generator().manual_seed(n)
    q, gal = torch.randn((m, D), generator=g), torch.randn((n, D), generator=g)
    ref = O.euclidean_squared(q.double(), gal.double()) if metric == "euclidean" else O.cosine(q.double(), gal.double())
    qd, gd = q.to(DEV), gal.to(DEV)
    wide = hip_distmat_device(qd, gd, metric, LP16)
    monkeypatch.setenv("AGRL_DISTMAT_TILED", "1")
    _hip.reload_options()
    tiled = hip_distmat_device(qd, gd, metric, LP16)
    monkeypatch.delenv("AGRL_DISTMAT_TILED")
    _hip.reload_options()
    torch.cuda.synchronize()
    e = rel_err(wide, ref)
    d = (wide - tiled).abs().max().item() / tiled.abs().max().item()
    print("distmat wide f32out", shape, metric, "vs fp64 %.3e, vs tiled %.3e" % (e, d))
    assert e < 1e-2 and d < 1e-6 and torch.isfinite(wide).all()


@pytest.mark.parametrize("cfg", [(8, 16, 8, 512, 512, 3, 1, 1, False), (8, 16, 8, 2048, 512, 1, 1, 0, False), (6, 16, 8, 512, 2048, 1, 1, 0, True),
                                 (4, 32, 16, 256, 256, 3, 2, 1, False), (3, 64, 32, 64, 64, 1, 1, 0, False), (2, 9, 5, 32, 96, 3, 1, 1, True)])
def test_conv_split_fp16_three_products(cfg):
    """agrl_conv2d_bn_act_split16 (round 6; vmgn.py:45-65 in fp32 tensors, every product as three FP16 MFMAs on fp16 high / low
    halves, weights pre-scaled by a power of two, un-scaled in the epilogue) against F.conv2d in FLOAT64 -- beside the exact-fp32
    kernel and the split-bf16 one on the same operands. Weights at BatchNorm-folded magnitude (1e-2 / sqrt(fan-in) .. with two
    channels a thousand times smaller), activations post-ReLU-like with a long tail of tiny values: the cases where an fp16 low
    half goes subnormal. Bars: within 4 x the exact-fp32 kernel's own distance from float64 + 2^-22 (the dropped low x low
    term), and at least 5 x closer than split-bf16."""
    from torchreid import hip_ops as ops
    N, H, W, Cin, Cout, R, stride, pad, with_res = cfg
    g = torch.Generator().manual_seed(sum(cfg[:8]))
    x = torch.randn((N, H, W, Cin), generator=g).clamp(min=0) * torch.exp(2.0 * torch.randn((N, H, W, 1), generator=g))   # 1e-3 .. 1e2
    x[0, 0, 0, :8] = torch.tensor([3e-5, 1e-6, 6e-8, 0.0, 1e-3, 2e-4, 7.0, 1200.0])
    w = torch.randn((Cout, R, R, Cin), generator=g) * (0.5 / np.sqrt(Cin * R * R))
    w[1] *= 1e-3
    w[Cout // 2] *= 1e-3
    b = 0.1 * torch.randn(Cout, generator=g)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    res = None
    if with_res:
        res = torch.randn(ref.shape, generator=g)
        ref = ref + res.double()
    ref = ref.clamp(min=0)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    rd = None if res is None else res.to(DEV)
    exact = ops.conv_bn_act(xd, wd, bd, stride, pad, True, residual=rd)
    with ops.f32_split(True):
        b3 = ops.conv_bn_act(xd, wd, bd, stride, pad, True, residual=rd)
    ws = ops.split16_prescale(wd)
    k = np.log2(ws.agrl_unscale)
    assert k == int(k) and 2 ** 13 <= float(ws.abs().max()) < 2 ** 14 and torch.equal(ws * ws.agrl_unscale, wd)   # exact pre-scale
    wp = ops.split16_inloop_weights(wd)
    assert wp.agrl_unscale == ws.agrl_unscale and torch.equal(ops.split16_true_weights(wp), wd)
    h3 = ops.conv_bn_act(xd, wp, bd, stride, pad, True, residual=rd)
    torch.cuda.synchronize()
    den = ref.abs().max().item()
    e = {name: ((t.double().cpu() - ref).abs().max().item() / den) for name, t in (("fp32", exact), ("bf16x3", b3), ("fp16x3", h3))}
    # the two scaled-down channels on their own scale (the per-tensor pre-scale leaves them 2^-10 below the rest)
    small = [1, Cout // 2]
    es = {name: ((t[..., small].double().cpu() - ref[..., small]).abs().max().item() / ref[..., small].abs().max().clamp(min=1e-30).item())
          for name, t in (("fp32", exact), ("fp16x3", h3))}
    print("conv split-fp16", cfg, "vs float64: exact fp32 %.2e, fp16x3 %.2e, bf16x3 %.2e | 1e-3-scaled channels: fp32 %.2e fp16x3 %.2e" % (
        e["fp32"], e["fp16x3"], e["bf16x3"], es["fp32"], es["fp16x3"]))
    assert torch.isfinite(h3).all()
    # (measured, K = 4608: exact fp32 1.5e-6 -- its accumulation error -- fp16x3 1.0e-6, bf16x3 1.4e-5)
    assert e["fp16x3"] < 4 * e["fp32"] + 2.4e-7 and e["fp16x3"] * 5 < e["bf16x3"]
    assert es["fp16x3"] < 1e-4     # bias-dominated outputs: loose; the strict bar is the whole-tensor one above


def test_fp16_mfma_keeps_subnormal_operands():
    """What the split-fp16 modes rest on: v_mfma_f32_*_f16 does NOT flush subnormal fp16 operands (a low half below 2^-14 keeps its
    absolute precision of 2^-24). x = 2^-20 everywhere (a subnormal fp16, exactly representable), w = 1024: every product is 2^-10,
    K = 64 of them sum to 2^-4; a flushing matrix pipe would return 0."""
    if LP16 != "fp16":
        pytest.skip("fp16 build only")
    from torchreid import hip_ops as ops
    x = torch.full((2, 16, 8, 64), 2.0 ** -20, dtype=torch.float16, device=DEV)
    w = torch.full((64, 1, 1, 64), 1024.0, dtype=torch.float16, device=DEV)
    assert float(x[0, 0, 0, 0]) == 2.0 ** -20
    out = ops.conv_bn_act(x, w, torch.zeros(64, device=DEV), 1, 0, False)
    torch.cuda.synchronize()
    print("fp16 MFMA on subnormal operands: sum of 64 x (2^-20 x 1024) = %.6f (expected 0.0625)" % float(out[0, 0, 0, 0]))
    assert torch.all(out.float() == 0.0625)


@pytest.mark.parametrize("shape", [(3, 256, 128), (2, 64, 32), (1, 37, 23), (2, 130, 70)])
def test_stem_split16_matches_float64(shape):
    """agrl_stem_split16 (round 6: the stem as three fp16 MFMAs per product, fp32 out; vmgn.py:281-284) against conv 7x7/2 + ReLU +
    maxpool 3x3/2 in FLOAT64 and beside the exact-fp32 stem (agrl_stem_conv_bn_relu_maxpool): image borders, ragged tiles, a weight row
    a thousand times smaller than the rest."""
    from torchreid import hip_ops as ops
    N, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn((N, 3, H, W), generator=g)
    w = torch.randn((64, 7, 7, 3), generator=g) * 0.08
    w[5] *= 1e-3
    b = 0.2 * torch.randn(64, generator=g)
    y = F.conv2d(x.double(), w.permute(0, 3, 1, 2).double(), b.double(), stride=2, padding=3).clamp(min=0)
    ref = F.max_pool2d(y, 3, 2, 1).permute(0, 2, 3, 1)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    exact = ops.stem(xd, wd, bd, torch.float32)
    wh, wl, u = ops.pack_stem_weights_split16(wd)
    got = ops.stem_split16(xd, wh, wl, u, bd)
    torch.cuda.synchronize()
    den = ref.abs().max().item()
    e32 = (exact.double().cpu() - ref).abs().max().item() / den
    e16 = (got.double().cpu() - ref).abs().max().item() / den
    print("stem split16", shape, "vs float64: exact fp32 %.2e, split fp16 %.2e" % (e32, e16))
    assert got.shape == ref.shape and torch.isfinite(got).all()
    assert e16 < 4 * e32 + 2.4e-7


def test_split16_entry_points_validate_their_arguments():
    """The plane entry points refuse what they cannot compute instead of computing something else: the bf16 build (the planes are fp16),
    channel counts that are not whole plane triples of whole slabs, an un-scaling factor that is not a power of two."""
    from torchreid import hip_ops as ops
    x = torch.rand((4, 16, 8, 128), device=DEV)
    if LP16 != "fp16":
        with pytest.raises(_hip.HipKernelError):
            ops.to_split16_planes(x)
        with pytest.raises(_hip.HipKernelError):
            ops.to_split16_weight_planes(x.view(-1, 128), 8.0)
        return
    x3 = ops.to_split16_planes(x)
    w3, u = ops.split16_plane_weights(torch.randn((256, 128), device=DEV) * 0.05)
    pk = ops.conv1x1_pack(w3)
    b = torch.zeros(256, device=DEV)
    out = ops.conv1x1_split16(x3, pk, u, b, 256)
    assert tuple(out.shape) == (4, 16, 8, 768)
    with pytest.raises(_hip.HipKernelError):   # 0.3 is not a power of two
        ops.conv1x1_split16(x3, pk, 0.3, b, 256)
    with pytest.raises(_hip.HipKernelError):   # K3 = 320 is not a whole number of plane triples of 128-channel slabs
        ops.call("agrl_conv1x1_split16", ops.ptr(x3), ops.ptr(pk), ops.ptr(b), None, ops.ptr(out), 512, 320, 256, 1, 1.0, 0, None)
    with pytest.raises(_hip.HipKernelError):   # layout bits beyond 3
        ops.call("agrl_conv1x1_split16", ops.ptr(x3), ops.ptr(pk), ops.ptr(b), None, ops.ptr(out), 512, 384, 256, 1, 1.0, 4, None)
    with pytest.raises(_hip.HipKernelError):   # scale of the weight planes must be a power of two
        ops.to_split16_weight_planes(x.view(-1, 128), 3.0)
    with pytest.raises(_hip.HipKernelError):   # D3 must be a multiple of 3
        q3 = torch.zeros((8, 128), dtype=torch.float16, device=DEV)
        ops.call("agrl_distmat_split16", ops.ptr(q3), ops.ptr(q3), None, None, ops.ptr(torch.zeros((8, 8), device=DEV)), 8, 8, 128, 8, 1, 1.0, None, 0, None)


def _planes_ref(x3):
    from torchreid import hip_ops as ops
    return ops.from_split16_planes(x3).double().cpu()


def test_split16_planes_round_trip():
    """agrl_split16_planes: fp32 -> [hi | lo 2^11 | hi]; hi + lo 2^-11 gives the value back to 2^-22 relative (<= 2^-25 absolute in fp16's
    subnormal range), planes 0 and 2 are equal, zeros and the sign survive."""
    if LP16 != "fp16":
        pytest.skip("fp16 build only")
    from torchreid import hip_ops as ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn((300, 64), generator=g) * torch.exp(3.0 * torch.randn((300, 1), generator=g))
    x[0, :6] = torch.tensor([0.0, -0.0, 3e-5, -1e-6, 6e-8, 60000.0])
    x3 = ops.to_split16_planes(x.to(DEV))
    torch.cuda.synchronize()
    assert x3.dtype == torch.float16 and tuple(x3.shape) == (300, 192) and torch.equal(x3[:, :64], x3[:, 128:])
    back = _planes_ref(x3)
    err = (back - x.double()).abs()
    bound = torch.maximum(x.double().abs() * 2.0 ** -21, torch.full_like(err, 2.0 ** -24))
    print("split16 planes round trip: worst relative error %.2e" % float((err / x.double().abs().clamp(min=1e-3)).max()))
    assert bool((err <= bound).all())
    assert torch.equal(x3[:, :64].cpu(), x.half())


@pytest.mark.parametrize("cfg", [(256 * 4, 512, 2048, "res"), (128 * 6, 2048, 512, "plain"), (128 * 5 + 40, 1024, 256, "plain"),
                                 (128 * 4, 256, 1024, "res"), (128 * 6, (1024, 512), 2048, "dual"), (5, 512, 2048, "pool")])
def test_conv1x1_split16_planes(cfg):
    """agrl_conv1x1_split16 / _dual / _pool (round 6: conv1x1_duo_kernel on split-fp16 planes; vmgn.py:48-50, :56-64, :298-308) against
    float64 on the values the planes hold: plain, + residual planes, the two-source first-block form, the pooled last conv; ragged M.
    Bar: the result planes reproduce float64 to 2e-6 of the largest output (fp32 accumulation over K <= 2048 + the dropped lo x lo term;
    the 16-bit kernel on the same data sits at ~5e-4)."""
    if LP16 != "fp16":
        pytest.skip("fp16 build only")
    from torchreid import hip_ops as ops
    M, K, Cout, kind = cfg
    g = torch.Generator().manual_seed(M + Cout)
    Ks = list(K) if isinstance(K, tuple) else [K]
    if kind == "pool":
        N, H, W = M, 16, 8
        M = N * 128
    xs = [(torch.randn((M, k_), generator=g).clamp(min=0) * torch.exp(1.5 * torch.randn((M, 1), generator=g))) for k_ in Ks]
    w = torch.randn((Cout, sum(Ks)), generator=g) * (0.7 / np.sqrt(sum(Ks)))
    w[3] *= 1e-3
    b = 0.1 * torch.randn(Cout, generator=g)
    x3 = [ops.to_split16_planes(x_.to(DEV)) for x_ in xs]
    xv = torch.cat([_planes_ref(t) for t in x3], dim=1)        # what the planes hold (22 bits)
    w3, unscale = ops.split16_plane_weights(w.to(DEV), segments=Ks if len(Ks) > 1 else None)
    # the weight triples hold w 2^k to 22 bits as well
    segs, off = [], 0
    for k_ in Ks:
        t = w3[:, off:off + 3 * k_].double().cpu()
        segs.append((t[:, :k_] + t[:, 2 * k_:]) * unscale)
        assert torch.equal(t[:, k_:2 * k_] * 2048.0, t[:, :k_]) or float((t[:, k_:2 * k_] * 2048.0 - t[:, :k_]).abs().max()) < 2.0 ** -10
        off += 3 * k_
    wv = torch.cat(segs, dim=1)
    assert float((wv - w.double()).abs().max() / w.abs().max()) < 2.0 ** -21
    packed = ops.conv1x1_pack(w3)
    ref = xv @ wv.t() + b.double()
    bd = b.to(DEV)
    if kind == "res":
        r = torch.randn((M, Cout), generator=g)
        r3 = ops.to_split16_planes(r.to(DEV))
        ref = (ref + _planes_ref(r3)).clamp(min=0)
        out = ops.conv1x1_split16(x3[0].view(1, M, 1, -1), packed, unscale, bd, Cout, residual3=r3.view(1, M, 1, -1))
    elif kind == "dual":
        ref = ref.clamp(min=0)
        out = ops.conv1x1_split16(x3[0].view(1, M, 1, -1), packed, unscale, bd, Cout, x2=x3[1].view(1, M, 1, -1))
    elif kind == "pool":
        r = torch.randn((M, Cout), generator=g).clamp(min=0)
        r3 = ops.to_split16_planes(r.to(DEV))
        full = (ref + _planes_ref(r3)).clamp(min=0).view(N, 16, 8, Cout)
        splits = [4, 2, 1]
        bins = []
        for n_ in splits:
            for j in range(n_):
                bins.append(full[:, j * 16 // n_:(j + 1) * 16 // n_].mean(dim=(1, 2)))
        ref = torch.stack(bins, dim=1)
        got = ops.conv1x1_split16_pool(x3[0].view(N, 16, 8, -1), packed, unscale, bd, Cout, r3.view(N, 16, 8, -1), splits, True)
        gsum = ops.conv1x1_split16_pool(x3[0].view(N, 16, 8, -1), packed, unscale, bd, Cout, r3.view(N, 16, 8, -1), [1], False)
        torch.cuda.synchronize()
        e = float((got.double().cpu() - ref).abs().max() / ref.abs().max())
        e2 = float((gsum.double().cpu()[:, 0] - full.sum(dim=(1, 2))).abs().max() / full.sum(dim=(1, 2)).abs().max())
        print("conv1x1 split16 pooled", cfg, "means %.2e sums %.2e" % (e, e2))
        assert e < 2e-6 and e2 < 2e-6
        return
    else:
        ref = ref.clamp(min=0)
        out = ops.conv1x1_split16(x3[0].view(1, M, 1, -1), packed, unscale, bd, Cout)
    torch.cuda.synchronize()
    out = out.view(M, 3 * Cout)
    assert torch.equal(out[:, :Cout], out[:, 2 * Cout:])
    e = float((_planes_ref(out) - ref).abs().max() / ref.abs().max())
    print("conv1x1 split16", cfg, "vs float64 %.2e" % e)
    assert e < 2e-6


@pytest.mark.parametrize("cfg", [(128 * 6, 2048, 512, "plain"), (128 * 5 + 40, 1024, 256, "plain"), (256 * 4, 512, 2048, "res"), (128 * 4 + 8, 256, 1024, "res"),
                                 (128 * 6, (1024, 512), 2048, "dual"), (128 * 3 + 77, (1024, 256), 1024, "dual"), (5, 512, 2048, "pool")])
def test_conv1x1_split16_plane_pairs(cfg):
    """The plane-PAIR layout of the wide tensors (include/agrl_hip.h "Plane PAIRS": [hi | lo 2^11], the kernel reads the hi slab twice,
    weights per 128-channel slab [wh | wl | wh 2^-11]) through agrl_conv1x1_split16 / _dual / _pool: conv1 (x a pair -> a triple), conv3 +
    residual (x a triple, residual and result pairs), the two-source first block (x a pair, x2 a triple, result a pair), the pooled last conv
    (residual a pair) -- against float64 on the values the planes hold, and against the all-triples launch (same products, another k order:
    equal to fp32 summation roundoff). A pair IS the first two planes of the triple, checked bytewise."""
    if LP16 != "fp16":
        pytest.skip("fp16 build only")
    from torchreid import hip_ops as ops
    M, K, Cout, kind = cfg
    g = torch.Generator().manual_seed(M + Cout + 1)
    Ks = list(K) if isinstance(K, tuple) else [K]
    if kind == "pool":
        N, H, W = M, 16, 8
        M = N * 128
    xs = [(torch.randn((M, k_), generator=g).clamp(min=0) * torch.exp(1.5 * torch.randn((M, 1), generator=g))).to(DEV) for k_ in Ks]
    w = (torch.randn((Cout, sum(Ks)), generator=g) * (0.7 / np.sqrt(sum(Ks)))).to(DEV)
    bd = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    x3 = [ops.to_split16_planes(x_) for x_ in xs]
    x2p = ops.to_split16_planes(xs[0], 2)
    assert torch.equal(x2p, x3[0][:, :2 * Ks[0]].contiguous())
    xv = torch.cat([_planes_ref(t) for t in x3], dim=1)
    x_pair = kind in ("plain", "dual")
    ro_pair = kind in ("res", "dual", "pool")
    w3t, unscale = ops.split16_plane_weights(w, segments=Ks if len(Ks) > 1 else None)
    w3p, unscale_p = ops.split16_plane_weights(w, segments=Ks if len(Ks) > 1 else None, pair_first=x_pair)
    assert unscale == unscale_p and w3p.shape == w3t.shape
    pk_t, pk_p = ops.conv1x1_pack(w3t), ops.conv1x1_pack(w3p)
    ref = xv @ (w.double().cpu()).t() + bd.double().cpu()
    layout = (1 if x_pair else 0) | (2 if ro_pair else 0)
    xin = (x2p if x_pair else x3[0]).view(1, M, 1, -1)
    r3 = r2 = None
    if kind in ("res", "pool"):
        r = torch.randn((M, Cout), generator=g).clamp(min=0).to(DEV)
        r3, r2 = ops.to_split16_planes(r), ops.to_split16_planes(r, 2)
        ref = ref + _planes_ref(r3)
    ref = ref.clamp(min=0)
    if kind == "pool":
        splits = [4, 2, 1]
        got = ops.conv1x1_split16_pool(x3[0].view(N, 16, 8, -1), pk_p, unscale, bd, Cout, r2.view(N, 16, 8, -1), splits, True, layout=layout)
        tri = ops.conv1x1_split16_pool(x3[0].view(N, 16, 8, -1), pk_t, unscale, bd, Cout, r3.view(N, 16, 8, -1), splits, True)
        torch.cuda.synchronize()
        assert torch.equal(got, tri)      # (the k order is the triple's here: only the residual's row stride differs)
        return
    if kind == "dual":
        got = ops.conv1x1_split16(xin, pk_p, unscale, bd, Cout, x2=x3[1].view(1, M, 1, -1), layout=layout)
        tri = ops.conv1x1_split16(x3[0].view(1, M, 1, -1), pk_t, unscale, bd, Cout, x2=x3[1].view(1, M, 1, -1))
    else:
        got = ops.conv1x1_split16(xin, pk_p, unscale, bd, Cout, residual3=None if r2 is None else r2.view(1, M, 1, -1), layout=layout)
        tri = ops.conv1x1_split16(x3[0].view(1, M, 1, -1), pk_t, unscale, bd, Cout, residual3=None if r3 is None else r3.view(1, M, 1, -1))
    torch.cuda.synchronize()
    no = 2 if ro_pair else 3
    got, tri = got.view(M, no * Cout), tri.view(M, 3 * Cout)
    if not ro_pair:
        assert torch.equal(got[:, :Cout], got[:, 2 * Cout:])
    den = ref.abs().max()
    e = float((ops.from_split16_planes(got, no).double().cpu() - ref).abs().max() / den)
    d = float((ops.from_split16_planes(got, no).double() - ops.from_split16_planes(tri).double()).abs().max().cpu() / den)
    print("conv1x1 split16 pairs", cfg, "vs float64 %.2e, vs the triple launch %.2e" % (e, d))
    assert e < 2e-6 and d < 2e-6
    if kind == "res":      # same k order as the triple: bit-identical planes
        assert torch.equal(got, tri[:, :2 * Cout].contiguous())


@pytest.mark.parametrize("cfg", [(6, 16, 8, 512, 512), (5, 16, 8, 256, 256), (3, 32, 16, 128, 256)])
def test_conv3x3_split16_planes(cfg):
    """agrl_conv3x3_packed_split16 (conv3x3_fat_kernel / conv3x3_half_kernel on split-fp16 planes; vmgn.py:52-54) against F.conv2d in
    float64 on the values the planes hold."""
    if LP16 != "fp16":
        pytest.skip("fp16 build only")
    from torchreid import hip_ops as ops
    N, H, W, Cin, Cout = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    x = torch.randn((N, H, W, Cin), generator=g).clamp(min=0) * torch.exp(1.5 * torch.randn((N, H, W, 1), generator=g))
    w = torch.randn((Cout, 3, 3, Cin), generator=g) * (0.7 / np.sqrt(9 * Cin))
    b = 0.1 * torch.randn(Cout, generator=g)
    x3 = ops.to_split16_planes(x.to(DEV))
    w3, unscale = ops.split16_plane_weights(w.to(DEV))
    t = w3.double().cpu()
    wv = (t[..., :Cin] + t[..., 2 * Cin:]) * unscale
    ref = F.conv2d(_planes_ref(x3).permute(0, 3, 1, 2), wv.permute(0, 3, 1, 2), b.double(), padding=1).permute(0, 2, 3, 1).clamp(min=0)
    out = ops.conv3x3_split16(x3, ops.conv3x3_pack(w3), unscale, b.to(DEV), Cout)
    torch.cuda.synchronize()
    assert torch.equal(out[..., :Cout], out[..., 2 * Cout:])
    e = float((_planes_ref(out) - ref).abs().max() / ref.abs().max())
    print("conv3x3 split16", cfg, "vs float64 %.2e" % e)
    assert e < 2e-6


@pytest.mark.parametrize("cfg", [(4, 64, 32, 64, 64, 256, 1), (3, 64, 32, 256, 128, 512, 2), (5, 32, 16, 512, 256, 1024, 2), (2, 16, 8, 1024, 512, 2048, 1),
                                 (2, 9, 7, 32, 64, 96, 2)])
def test_conv1x1_dual_split16_matches_the_two_convs(cfg):
    """agrl_conv1x1_dual_split16 (round 6): conv3 + the stride-s 1x1 downsample conv of a first Bottleneck as ONE in-loop split GEMM over
    [block input sampled at the stride | conv2's output] (vmgn.py:56-64) against float64, and against the two split convs it replaces
    (downsample conv -> shortcut map -> conv3 + residual): layers 1-4's first-block shapes, an odd map."""
    from torchreid import hip_ops as ops
    N, H, W, K1, K2, Cout, stride = cfg
    g = torch.Generator().manual_seed(sum(cfg))
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    x = torch.randn((N, H, W, K1), generator=g).clamp(min=0)
    y = torch.randn((N, OH, OW, K2), generator=g).clamp(min=0)
    wd = torch.randn((Cout, 1, 1, K1), generator=g) * (0.7 / np.sqrt(K1))
    w3 = torch.randn((Cout, 1, 1, K2), generator=g) * (0.7 / np.sqrt(K2))
    bd, b3 = 0.1 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g)
    xs = x[:, ::stride, ::stride, :]
    ref = (xs.double() @ wd.view(Cout, K1).double().t() + y.double() @ w3.view(Cout, K2).double().t() + (bd + b3).double()).clamp(min=0)
    xd, yd = x.to(DEV), y.to(DEV)
    wcat = ops.split16_inloop_weights(torch.cat([wd.view(Cout, K1), w3.view(Cout, K2)], dim=1).to(DEV))
    got = ops.conv1x1_dual_split16(xd, yd, wcat, (bd + b3).to(DEV), stride, True)
    short = ops.conv_bn_act(xd, ops.split16_inloop_weights(wd.to(DEV)), bd.to(DEV), stride, 0, False)
    two = ops.conv_bn_act(yd, ops.split16_inloop_weights(w3.to(DEV)), b3.to(DEV), 1, 0, True, residual=short)
    torch.cuda.synchronize()
    den = ref.abs().max().item()
    e1 = (got.double().cpu() - ref).abs().max().item() / den
    e2 = (two.double().cpu() - ref).abs().max().item() / den
    print("dual split16", cfg, "vs float64: one GEMM %.2e, two convs %.2e" % (e1, e2))
    assert tuple(got.shape) == (N, OH, OW, Cout) and e1 < 2e-6 and e1 < 2 * e2 + 3e-7


@pytest.mark.parametrize("cfg", [(3, 64, 32, 256, 64, 256, 1, False), (2, 64, 32, 64, 64, 256, 1, True), (3, 32, 16, 512, 128, 512, 1, False),
                                 (2, 64, 32, 256, 128, 512, 2, True), (2, 32, 16, 512, 256, 1024, 2, True), (2, 11, 7, 160, 32, 160, 1, False),
                                 (1, 9, 7, 64, 96, 128, 2, True)])
def test_presplit_activations_are_bit_identical_to_splitting_in_the_loop(cfg):
    """One Bottleneck of the conforming mode with conv1's / conv2's outputs stored PRE-SPLIT (agrl_conv2d_bn_act_split16 x_presplit /
    out_presplit, agrl_conv1x1_dual_split16 x2_presplit) against the same three launches on fp32 tensors: the fp16 halves the epilogue
    stores are those the consumer's k-loop would form, so every output must be equal bit for bit. Layer 1-3 shapes incl. the narrow
    (64-channel) tiles, stride 2, first blocks through the two-source GEMM, ragged pixel counts; and the layout itself is checked against
    the weight packer's (a pre-split activation row IS a pre-split weight row)."""
    from torchreid import hip_ops as ops
    N, H, W, Cin, mid, Cout, stride, first = cfg
    g = torch.Generator().manual_seed(sum(cfg[:7]))
    x = (torch.randn((N, H, W, Cin), generator=g).clamp(min=0) * torch.exp(1.5 * torch.randn((N, H, W, 1), generator=g))).to(DEV)
    mk = lambda co, r, ci: ops.split16_inloop_weights((torch.randn((co, r, r, ci), generator=g) * (0.8 / np.sqrt(ci * r * r))).to(DEV))
    w1, w2, w3 = mk(mid, 1, Cin), mk(mid, 3, mid), mk(Cout, 1, mid)
    b1, b2, b3 = (0.1 * torch.randn(c, generator=g).to(DEV) for c in (mid, mid, Cout))
    outs = {}
    for pre in (False, True):
        y1 = ops.conv_bn_act(x, w1, b1, 1, 0, True, out_presplit=pre)
        y2 = ops.conv_bn_act(y1, w2, b2, stride, 1, True, x_presplit=pre, out_presplit=pre)
        if first:
            wd = (torch.randn((Cout, Cin), generator=torch.Generator().manual_seed(7)) * (0.8 / np.sqrt(Cin))).to(DEV)
            wcat = ops.split16_inloop_weights(torch.cat([wd, ops.split16_true_weights(w3).view(Cout, mid)], dim=1).contiguous())
            out = ops.conv1x1_dual_split16(x, y2, wcat, b3, stride, True, x2_presplit=pre)
        else:
            assert stride == 1 and Cin == Cout
            out = ops.conv_bn_act(y2, w3, b3, 1, 0, True, residual=x, x_presplit=pre)
        outs[pre] = (y1, y2, out)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[False][2]).all() and float(outs[False][2].abs().max()) > 0
    assert torch.equal(outs[True][2], outs[False][2])
    for i in (0, 1):      # the stored halves against the host packer's layout of the fp32 tensor (scale 1 here: compare the raw split)
        t = outs[False][i]
        K = t.shape[-1]
        v = t.reshape(-1, K // 32, 2, 4, 4).permute(0, 1, 3, 2, 4)
        hi = v.to(torch.float16)
        lo = (v - hi.float()).to(torch.float16)
        want = torch.stack([hi, lo], dim=2).contiguous().view(torch.int32).view(-1)
        assert torch.equal(outs[True][i].view(torch.int32).view(-1), want)


@pytest.mark.parametrize("cfg", [(3, 32, 16, 512, 256, 1024, 2), (2, 16, 8, 1024, 512, 2048, 1), (1, 9, 7, 64, 96, 128, 2)])
def test_conv1x1_dual_split16_writes_the_planes_itself(cfg):
    """agrl_conv1x1_dual_split16 with out_planes = 2 / 3 (the seam between the fp32-tensor part of the conforming mode and its plane part):
    the planes must be exactly agrl_split16_planes of the fp32 map the same launch writes with out_planes = 0."""
    if LP16 != "fp16":
        pytest.skip("fp16 build only")
    from torchreid import hip_ops as ops
    N, H, W, K1, K2, Cout, stride = cfg
    g = torch.Generator().manual_seed(sum(cfg) + 3)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    x = torch.randn((N, H, W, K1), generator=g).clamp(min=0).to(DEV)
    y = torch.randn((N, OH, OW, K2), generator=g).clamp(min=0).to(DEV)
    wcat = ops.split16_inloop_weights((torch.randn((Cout, K1 + K2), generator=g) * (0.7 / np.sqrt(K1 + K2))).to(DEV))
    b = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    full = ops.conv1x1_dual_split16(x, y, wcat, b, stride, True)
    for n in (2, 3):
        got = ops.conv1x1_dual_split16(x, y, wcat, b, stride, True, out_planes=n)
        torch.cuda.synchronize()
        assert got.dtype == torch.float16 and tuple(got.shape) == (N, OH, OW, n * Cout)
        assert torch.equal(got, ops.to_split16_planes(full, n))


@pytest.mark.parametrize("shape", [(32, 56, 2048), (3, 28, 2048), (5, 60, 256)])
def test_graph_operand_presplit_is_bit_identical(shape):
    """The GraphLayer's GEMM of the conforming mode with P = G f written PRE-SPLIT by agrl_graph_apply (AGRL_F32H3P) against the same
    GEMM splitting P in its k-loop (AGRL_F32H3): equal bit for bit, and P's bytes are the host packer's layout of the fp32 P."""
    from torchreid import hip_ops as ops
    from recipe import synthetic_adj
    B, V, C = shape
    g = torch.Generator().manual_seed(B + V + C)
    f = (torch.rand((B, 1, C), generator=g) + 0.02 * torch.randn((B, V, C), generator=g)).to(DEV)
    adj = (synthetic_adj(B, V // 7, seed=V) if V % 7 == 0 else (torch.rand((B, V, V), generator=g) > 0.5).float()).to(DEV)
    G = ops.graph_matrix(f, adj, True, True)
    w = ops.split16_inloop_weights((torch.randn((C, C), generator=g) * 0.02).to(DEV))
    scale = ((0.8 + 0.4 * torch.rand(C, generator=g)).to(DEV) * w.agrl_unscale).contiguous()
    scale.agrl_folded_unscale = w.agrl_unscale
    shift = (0.1 * torch.randn(C, generator=g)).to(DEV)
    P = ops.graph_apply_operand(G, f, torch.float32)
    Pp = ops.graph_apply_operand(G, f, torch.float32, presplit=True)
    a = ops.graph_linear_mix(P, w, f, scale, shift, 0.1, 0.1)
    b = ops.graph_linear_mix(Pp, w, f, scale, shift, 0.1, 0.1, p_presplit=True)
    torch.cuda.synchronize()
    v = P.reshape(-1, C // 32, 2, 4, 4).permute(0, 1, 3, 2, 4)
    hi = v.to(torch.float16)
    want = torch.stack([hi, (v - hi.float()).to(torch.float16)], dim=2).contiguous().view(torch.int32).view(-1)
    assert torch.equal(Pp.view(torch.int32).view(-1), want)
    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_conv_split_fp16_rejects_a_scale_that_is_not_a_power_of_two():
    from torchreid import hip_ops as ops
    x = torch.rand((1, 16, 8, 64), device=DEV)
    w = ops.split16_inloop_weights(torch.randn((64, 1, 1, 64), device=DEV) * 0.01)
    w.agrl_unscale = 0.3
    with pytest.raises(_hip.HipKernelError):
        ops.conv_bn_act(x, w, torch.zeros(64, device=DEV), 1, 0, True)
    with pytest.raises(AssertionError):    # scaled but not pre-split: the kernel would read fp32 bits as fp16 pairs
        ops.conv_bn_act(x, ops.split16_prescale(torch.randn((64, 1, 1, 64), device=DEV) * 0.01), torch.zeros(64, device=DEV), 1, 0, True)


def test_split16_inloop_weight_layout():
    """agrl_split16_weights_inloop against its contract (include/agrl_hip.h): every 32-value k-tile of a row becomes
    [hi(k 4c..4c+3, 16+4c..16+4c+3), c = 0..3 | lo in the same order], hi = fp16(w) to nearest, lo = fp16(w - hi) -- bytes compared,
    incl. values whose low half is a subnormal fp16 and zeros; K not a multiple of 32 is refused."""
    from torchreid import hip_ops as ops
    g = torch.Generator().manual_seed(5)
    w = torch.randn((24, 3, 3, 96), generator=g) * torch.exp(3.0 * torch.randn((24, 3, 3, 96), generator=g))
    w[0, 0, 0, :4] = torch.tensor([0.0, 1e-7, 6.1e-5, -3.3e-6])
    host = ops.split16_inloop_weights(w.to(DEV))           # the packers' torch form
    ws = ops.split16_prescale(w)
    wsd = ws.to(DEV)
    got = torch.empty_like(wsd)                            # the C-ABI kernel
    _hip.call("agrl_split16_weights_inloop", wsd.data_ptr(), got.data_ptr(), w.numel() // 96, 96, 0)
    torch.cuda.synchronize()
    assert host.agrl_unscale == ws.agrl_unscale and host.shape == w.shape and host.dtype == torch.float32 and torch.equal(
        host.view(torch.int32), got.view(torch.int32))
    t = ws.view(-1, 3, 2, 4, 4).permute(0, 1, 3, 2, 4).contiguous()      # (rows, tile, c, half, e): the lane group's eight values
    hi = t.to(torch.float16)
    lo = (t - hi.float()).to(torch.float16)
    want = torch.stack([hi, lo], dim=2).contiguous()                       # (rows, tile, {hi, lo}, c, half, e)
    assert torch.equal(got.cpu().view(torch.float16).view(-1), want.view(-1))
    with pytest.raises(AssertionError):
        ops.split16_inloop_weights(torch.randn((8, 48), device=DEV))
    with pytest.raises(_hip.HipKernelError):
        _hip.call("agrl_split16_weights_inloop", got.data_ptr(), got.data_ptr(), 8, 48, 0)


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
@pytest.mark.parametrize("shape", [(1980, 12180, 512), (1000, 12002, 576), (700, 390, 512)])
def test_distmat_wide_tile_192_columns(shape, metric, monkeypatch):
    """Round-5 advice: the 256 x 192 instantiation of the fp32-output tile (igemm_wide_kernel<65536, 192>) had no direct test -- both
    shapes above select the 256-column tile. Here: the MARS shape (1980 x 12 180: 512 tiles of 192 columns = two exact rounds, the
    heuristic's own choice; 12 180 = 63 x 192 + 84, ragged), a second ragged N the heuristic also sends there, and a small matrix
    on which the tile is FORCED (AGRL_DISTMAT_TILE_N=192) -- each against the 256-column tile (bit-identical: same k order per
    output element), the tiled igemm form and the fp64 oracle."""
    from torchreid.metrics.distance import hip_distmat_device
    m, n, D = shape
    g = torch.Generator().manual_seed(n + m)
    q, gal = torch.randn((m, D), generator=g), torch.randn((n, D), generator=g)
    ref = O.euclidean_squared(q.double(), gal.double()) if metric == "euclidean" else O.cosine(q.double(), gal.double())
    qd, gd = q.to(DEV), gal.to(DEV)
    auto = hip_distmat_device(qd, gd, metric, LP16)
    out = {}
    for tile in ("192", "256"):
        monkeypatch.setenv("AGRL_DISTMAT_TILE_N", tile)
        _hip.reload_options()
        out[tile] = hip_distmat_device(qd, gd, metric, LP16)
    monkeypatch.delenv("AGRL_DISTMAT_TILE_N")
    monkeypatch.setenv("AGRL_DISTMAT_TILED", "1")
    _hip.reload_options()
    tiled = hip_distmat_device(qd, gd, metric, LP16)
    monkeypatch.delenv("AGRL_DISTMAT_TILED")
    _hip.reload_options()
    torch.cuda.synchronize()
    e = rel_err(out["192"], ref)
    d = (out["192"] - tiled).abs().max().item() / tiled.abs().max().item()
    print("distmat 256 x 192 tile", shape, metric, "vs fp64 %.3e, vs tiled %.3e, == 256-column tile %s" % (e, d, torch.equal(out["192"], out["256"])))
    assert torch.isfinite(out["192"]).all() and e < 1e-2 and d < 1e-6
    assert torch.equal(out["192"], out["256"])
    assert torch.equal(auto, out["192"])


@pytest.mark.parametrize("metric", ["euclidean", "cosine"])
@pytest.mark.parametrize("shape", [(32, 12180, 4096), (1980, 3000, 4096), (256, 1523, 4096), (7, 60, 64), (37, 101, 128)])
def test_distmat_split16_is_fp32_class(shape, metric):
    """agrl_distmat_split16 (round 6: the 16-bit distance kernels -- streaming, tiled + split-K, the 256-column fp32-output tile -- on
    split-fp16 plane operands, distance.py:59-89) against the fp64 oracle, beside the exact-fp32 kernel and the plain 16-bit one on the
    same rows: fp32-class accuracy. The step's shape, a full-eval slab, an 8-GPU shard, tiny ragged ones."""
    if LP16 != "fp16":
        pytest.skip("fp16 build only")
    from torchreid.metrics.distance import hip_distmat_device
    m, n, D = shape
    g = torch.Generator().manual_seed(m + n)
    centers = torch.randn((16, D), generator=g)
    q = centers[torch.randint(0, 16, (m,), generator=g)] + 0.5 * torch.randn((m, D), generator=g)
    gal = centers[torch.randint(0, 16, (n,), generator=g)] + 0.5 * torch.randn((n, D), generator=g)
    ref = O.euclidean_squared(q.double(), gal.double()) if metric == "euclidean" else O.cosine(q.double(), gal.double())
    qd, gd = q.to(DEV), gal.to(DEV)
    d3 = hip_distmat_device(qd, gd, metric, "fp16x3")
    d32 = hip_distmat_device(qd, gd, metric, "fp32")
    d16 = hip_distmat_device(qd, gd, metric, LP16)
    torch.cuda.synchronize()
    den = ref.abs().max().item()
    e3, e32, e16 = [(t.double().cpu() - ref).abs().max().item() / den for t in (d3, d32, d16)]
    print("distmat split16", shape, metric, "vs fp64: fp16x3 %.2e, exact fp32 %.2e, %s %.2e" % (e3, e32, LP16, e16))
    assert d3.shape == (m, n) and torch.isfinite(d3).all()
    assert e3 < 4 * e32 + 3e-7 and e3 * 5 < e16 + 1e-9    # (measured: 1.6e-6 .. 3.8e-6 against 1.3e-6 .. 3.3e-6 exact and 2.4e-5 .. 2.9e-5 in plain fp16)


def test_distmat_public_api_and_errors():
    from torchreid import metrics
    q, gal = torch.randn(7, 64), torch.randn(60, 64)
    d_cpu_in = metrics.compute_distance_matrix(q, gal, "euclidean")  # CPU in -> CPU out, computed on the GPU
    assert not d_cpu_in.is_cuda and d_cpu_in.shape == (7, 60)
    assert rel_err(d_cpu_in, O.euclidean_squared(q, gal)) < 1e-5
    d_dev = metrics.compute_distance_matrix(q.to(DEV), gal.to(DEV), "cosine")
    assert d_dev.is_cuda and rel_err(d_dev, O.cosine(q, gal)) < 1e-5
    with pytest.raises(ValueError):
        metrics.compute_distance_matrix(q, gal, "manhattan")
    with pytest.raises(AssertionError):
        metrics.compute_distance_matrix(q, gal[:, :32])


@pytest.mark.parametrize("shape", [(30, 300, 50), (4, 64, 64), (3, 12180, 50), (2, 5000, 1000)])
def test_rank_topk_exact(shape):
    from torchreid import hip_ops as ops
    m, n, k = shape
    rng = np.random.RandomState(n)
    d = rng.rand(m, n).astype(np.float32)
    d[0, : n // 2] = np.float32(0.25)          # massive ties
    d[1, 5] = np.nan
    d[1, 7] = -np.inf
    d[2 % m, ::3] = -d[2 % m, ::3]             # negatives
    if m > 3:
        d[3] = np.round(d[3] * 8) / 8           # few distinct values
    idx, val = ops.rank_topk(torch.from_numpy(d).to(DEV), k)
    torch.cuda.synchronize()
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    for r in range(m):
        ref = O.stable_topk(d[r], k)
        assert np.array_equal(idx[r], ref), (r, idx[r][:10], ref[:10])
        assert np.array_equal(val[r], d[r][ref], equal_nan=True)


@pytest.mark.parametrize("shape", [(9, 12180, 50), (5, 1023, 20), (6, 37, 30), (4, 2049, 128), (3, 30000, 1), (3, 4099, 50), (2, 40000, 50)])
def test_rank_topk_single_pass_form(shape):
    """The bandwidth-bound kernel (k <= 128, n <= 32768, aligned rows: the row crosses HBM once, threshold from the per-thread
    minima) on the cases that exercise its branches -- a row of equal values (candidate overflow -> radix path inside the same
    launch), NaN / +-inf / -0.0, more NaNs than the row has room after the top k, an n that is not a multiple of 4, rows shorter
    than k threads' worth of elements -- against the stable argsort (rank.py:171-172), and a row-strided view whose rows are not
    16-byte aligned (radix kernel) giving the same lists. (2, 40000, 50): n beyond the register form."""
    from torchreid import hip_ops as ops
    m, n, k = shape
    rng = np.random.RandomState(m * n + k)
    d = rng.randn(m, n).astype(np.float32)
    d[0] = np.float32(0.5)                       # every value equal: all n are candidates -> overflow path
    d[1, ::7] = np.nan
    d[1, 3] = -np.inf
    d[1, 11 % n] = np.inf
    if m > 2:
        d[2, : n // 2] = np.float32(-0.0)
        d[2, n // 2:] = np.float32(0.0)          # -0.0 == +0.0: pure index order
    if m > 3:
        d[3] = np.sort(d[3])                      # sorted row: the k smallest all sit with the first few threads
    if m > 4:
        d[4, k // 2:] = np.nan                    # fewer than k finite values
    dd = torch.from_numpy(d).to(DEV)
    idx, val = ops.rank_topk(dd, k)
    wide = torch.zeros((m, n + 3), dtype=torch.float32, device=DEV)
    wide[:, 1:n + 1] = dd
    idx_u, val_u = ops.rank_topk(wide[:, 1:n + 1], k)          # base pointer + 4 bytes: not 16-byte aligned
    torch.cuda.synchronize()
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    for r in range(m):
        ref = O.stable_topk(d[r], k)
        assert np.array_equal(idx[r], ref), (r, idx[r][:10], ref[:10])
        assert np.array_equal(val[r], d[r][ref] + np.float32(0.0), equal_nan=True)   # the kernel reports -0.0 as +0.0
    assert torch.equal(idx_u.cpu(), torch.from_numpy(idx)) and np.array_equal(val_u.cpu().numpy(), val, equal_nan=True)


@pytest.mark.parametrize("cfg", [(700, 3000, 256, "fp32"), (700, 3000, 256, LP16), (130, 517, 64, "fp32"), (40, 12180, 4096, LP16)])
def test_distmat_topk_equals_distmat_then_topk(cfg):
    """agrl_distmat_topk (distance rows of one query block at a time in a reused workspace, then the top-k of the block) ==
    agrl_distmat + agrl_rank_topk bit for bit when the queries fit one block (the recommended workspace here), same lists
    with a workspace that forces several blocks; both metrics; ascending (distance, index) order.
    distance.py:59-89 + rank.py:171-172."""
    from torchreid import hip_ops as ops
    from torchreid.metrics.distance import hip_distmat_device, hip_distmat_topk_device
    m, n, D, prec = cfg
    g = torch.Generator().manual_seed(m + n)
    q, gal = torch.randn((m, D), generator=g).to(DEV), torch.randn((n, D), generator=g).to(DEV)
    gal[7] = gal[3]                                            # duplicate gallery rows: exact distance ties
    k = 50
    for metric in ("cosine", "euclidean"):
        dist = hip_distmat_device(q, gal, metric, prec)
        idx0, val0 = ops.rank_topk(dist, k)
        idx1, val1 = hip_distmat_topk_device(q, gal, metric, k, prec)
        assert torch.equal(idx0, idx1) and torch.equal(val0, val1), metric
        assert bool((val1[:, 1:] >= val1[:, :-1]).all())
    dt = LP_DTYPE if prec == LP16 else torch.float32
    qh, gh = ops.row_l2_normalize(q, True, dt, ops.k_multiple(dt)), ops.row_l2_normalize(gal, True, dt, ops.k_multiple(dt))
    dist = ops.distmat(qh, gh, "cosine")
    idx0, val0 = ops.rank_topk(dist, k, idx_offset=1000)
    # a workspace of 300 rows: several query blocks. Every block runs the kernels its own row count selects (a different tile
    # shape keeps the k-order, a different kernel family may differ in the last bit), so: same lists, distances to 1 ulp-ish
    idx2, val2 = ops.distmat_topk(qh, gh, "cosine", k, idx_offset=1000, workspace_bytes=300 * 4 * (-(-n // 4) * 4))
    assert torch.equal(idx0, idx2) and (val0 - val2).abs().max().item() <= 2e-6
    with pytest.raises(Exception):
        ops.distmat_topk(qh, gh, "cosine", n + 1)


@pytest.mark.parametrize("rows", [64, 200, 333])
def test_distmat_topk_blocks_tie_aware(rows):
    """Several query blocks (a workspace of ``rows`` rows, incl. a short tail block that takes another kernel family): a distance
    may differ from the full-matrix path in its last bits, so the lists are compared tie-aware -- distances within 4e-6, and
    wherever the indices differ the two candidates' full-matrix distances are within that same bound (a swap inside a near-tie)."""
    from torchreid import hip_ops as ops
    m, n, D, k = 700, 3000, 256, 50
    g = torch.Generator().manual_seed(rows)
    q, gal = torch.randn((m, D), generator=g).to(DEV), torch.randn((n, D), generator=g).to(DEV)
    gal[11] = gal[5]
    qh, gh = ops.row_l2_normalize(q, True, LP_DTYPE, ops.k_multiple(LP_DTYPE)), ops.row_l2_normalize(gal, True, LP_DTYPE, ops.k_multiple(LP_DTYPE))
    dist = ops.distmat(qh, gh, "cosine")
    idx0, val0 = ops.rank_topk(dist, k)
    idx1, val1 = ops.distmat_topk(qh, gh, "cosine", k, workspace_bytes=rows * 4 * (-(-n // 4) * 4))
    torch.cuda.synchronize()
    tol = 4e-6
    assert (val0 - val1).abs().max().item() <= tol
    diff = idx0 != idx1
    if bool(diff.any()):
        d_at_other = torch.gather(dist, 1, idx1.to(torch.int64))
        assert ((d_at_other - val0).abs()[diff] <= tol).all(), "an index differs outside a near-tie"
    print("distmat_topk in blocks of %d rows: %d of %d positions differ (all inside near-ties)" % (rows, int(diff.sum()), diff.numel()))


def test_rank_mars_bit_exact():
    from torchreid import metrics
    rng = np.random.RandomState(7)
    m, n = 40, 400
    d = rng.rand(m, n).astype(np.float32)
    q_pids = rng.randint(0, 10, m)
    g_pids = rng.randint(0, 10, n)
    g_pids[rng.rand(n) < 0.05] = -1
    q_cam = rng.randint(0, 6, m)
    g_cam = rng.randint(0, 6, n)
    cmc_ref, map_ref, ap_ref, _, _ = O.evaluate_mars(d, q_pids, g_pids, q_cam, g_cam, 50, return_all=True)
    cmc, mAP = metrics.evaluate_rank(d, q_pids, g_pids, q_cam, g_cam, use_metric_mars=True)
    assert mAP == map_ref, (mAP, map_ref)            # fp64, same operation order -> identical
    assert np.array_equal(cmc, cmc_ref)
    d_dev = torch.from_numpy(d).to(DEV)
    cmc2, mAP2 = metrics.evaluate_rank(d_dev, q_pids, g_pids, q_cam, g_cam, use_metric_mars=True)
    assert mAP2 == map_ref and np.array_equal(cmc2, cmc_ref)
    assert metrics.evaluate_rank(d, q_pids, g_pids, q_cam, g_cam) is None
    with pytest.raises(ValueError):
        metrics.evaluate_rank(d[:, :30], q_pids, g_pids[:30], q_cam, g_cam[:30], use_metric_mars=True)
    q_bad = q_pids.copy()
    q_bad[0] = 999  # no match in the gallery
    with pytest.raises(ZeroDivisionError):
        metrics.evaluate_rank(d, q_bad, g_pids, q_cam, g_cam, use_metric_mars=True)


@pytest.mark.parametrize("shape", [(40, 500), (7, 64), (64, 12180)])
def test_rank_market1501_device(shape):
    """agrl_rank_market1501 (rank counting, no argsort) vs the oracle restatement of eval_market1501: CMC and validity
    exact, AP to fp64 rounding; plus the golden fixture produced by the reference's python evaluator."""
    from torchreid import hip_ops as ops, metrics
    m, n = shape
    rng = np.random.RandomState(m + n)
    d = rng.rand(m, n).astype(np.float32)
    d[:, 5] = d[:, 3]                                           # exact ties: stable order decides
    npid = max(4, n // 40)
    q_pids, g_pids = rng.randint(0, npid + 2, m), rng.randint(0, npid, n)   # two identities absent from the gallery
    q_cam, g_cam = rng.randint(0, 6, m), rng.randint(0, 6, n)
    cmc_o, mAP_o, ap_o, first_o = O.eval_market1501(d, q_pids, g_pids, q_cam, g_cam, 50, return_all=True)
    i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(DEV)
    ap, cmc, valid = ops.rank_market1501(torch.from_numpy(d).to(DEV), i32(q_pids), i32(q_cam), i32(g_pids), i32(g_cam), min(50, n))
    torch.cuda.synchronize()
    ap, cmc, valid = ap.cpu().numpy(), cmc.cpu().numpy(), valid.cpu().numpy()
    assert np.array_equal(valid == 1, ~np.isnan(ap_o))
    assert np.array_equal(np.isnan(ap), np.isnan(ap_o))
    ok = valid == 1
    assert np.abs(ap[ok] - ap_o[ok]).max() < 1e-14
    first = np.where(cmc[ok].max(1) > 0, cmc[ok].argmax(1), -1)
    assert np.array_equal(first, np.where(first_o[ok] < min(50, n), first_o[ok], -1))
    cmc2, mAP2 = metrics.evaluate_rank(d, q_pids, g_pids, q_cam, g_cam, max_rank=50, use_metric_market1501=True)
    assert np.array_equal(cmc2, cmc_o) and abs(mAP2 - mAP_o) < 1e-14
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rank_market1501.npz"))
    cmc3, mAP3 = metrics.evaluate_rank(z["dist"], z["q_pids"], z["g_pids"], z["q_camids"], z["g_camids"], use_metric_market1501=True)
    assert np.array_equal(cmc3, z["cmc"]) and abs(mAP3 - float(z["mAP"])) < 1e-14
    # the device kernel against the REFERENCE's own native evaluator (rank_cylib/rank_cy.pyx:154-241): its outputs on exactly
    # these seeded inputs were captured in the build container (tests/golden/make_golden.py F14); it accumulates in fp32
    zc = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rank_market1501_cy.npz"))
    if "cmc_cy_%dx%d" % (m, n) in zc.files:
        assert abs(d.astype(np.float64).sum() - float(zc["dist_checksum_%dx%d" % (m, n)])) < 1e-9
        assert np.allclose(zc["cmc_cy_%dx%d" % (m, n)], cmc2, atol=1e-6) and abs(float(zc["mAP_cy_%dx%d" % (m, n)]) - mAP2) < 1e-6


@pytest.mark.parametrize("shape", [(7, 900), (3, 12180), (5, 16384), (4, 1), (6, 33)])
def test_rank_argsort_full_rows(shape):
    """agrl_rank_argsort == np.argsort(kind='stable') of the whole row (rank.py:45-47), with exact ties, NaN, +-inf, -0.0."""
    from torchreid import hip_ops as ops
    m, n = shape
    rng = np.random.RandomState(m + n)
    d = rng.randn(m, n).astype(np.float32)
    d[0, ::3] = np.float32(0.25)
    if n > 8:
        d[1, 5], d[1, 7], d[1, 2] = np.nan, -np.inf, np.inf
        d[2, : n // 2] = np.float32(-0.0)
        d[2, n // 2:] = np.float32(0.0)
    idx = ops.rank_argsort(torch.from_numpy(d).to(DEV)).cpu().numpy()
    ref = np.argsort(d, axis=1, kind="stable")
    assert np.array_equal(idx, ref)


def test_rank_cuhk03_device():
    """evaluate_rank(use_metric_cuhk03=True) with the ranking and the AP from the device: equal to the reference's python
    evaluator (golden, seeded np.random) and to the oracle on a case with exact distance ties; a device-resident
    distance matrix gives the same answer as a host one."""
    from torchreid import metrics
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rank_cuhk03.npz"))
    args = (z["q_pids"], z["g_pids"], z["q_camids"], z["g_camids"])
    for max_rank in (50, 20):
        for dist in (z["dist"], torch.from_numpy(z["dist"]).to(DEV)):
            np.random.seed(int(z["seed"]))
            cmc, mAP = metrics.evaluate_rank(dist, *args, max_rank=max_rank, use_metric_cuhk03=True)
            assert np.random.randint(0, 1 << 30) == int(z["next_draw_%d" % max_rank])
            assert np.array_equal(cmc, z["cmc_%d" % max_rank]) and abs(mAP - float(z["mAP_%d" % max_rank])) < 1e-14
    rng = np.random.RandomState(77)
    m, n = 50, 900
    d = rng.rand(m, n).astype(np.float32)
    d[:, 7] = d[:, 2]
    q_pids, g_pids = rng.randint(0, 72, m), rng.randint(0, 70, n)
    q_cam, g_cam = rng.randint(0, 3, m), rng.randint(0, 3, n)
    np.random.seed(5)
    cmc_o, mAP_o = O.eval_cuhk03(d, q_pids, g_pids, q_cam, g_cam, 50)
    np.random.seed(5)
    cmc, mAP = metrics.evaluate_rank(d, q_pids, g_pids, q_cam, g_cam, max_rank=50, use_metric_cuhk03=True)
    assert np.array_equal(cmc, cmc_o) and abs(mAP - mAP_o) < 1e-14


def test_conv_and_linear_split_bf16_mode():
    """AGRL_F32X3: fp32 tensors, products as three bf16 MFMAs on high / low operand halves. Against the fp32 reference:
    ~1e-5 relative (the exact-fp32 kernel: ~1e-7), for a 1x1, a 3x3 strided conv with residual and the Linear."""
    from torchreid import hip_ops as ops
    g = torch.Generator().manual_seed(33)
    for (N, H, W, Cin, Cout, R, stride, use_res) in [(2, 16, 8, 256, 128, 1, 1, False), (2, 16, 8, 64, 96, 3, 2, True)]:
        x = torch.randn((N, Cin, H, W), generator=g)
        w = torch.randn((Cout, Cin, R, R), generator=g) / np.sqrt(Cin * R * R)
        b = torch.randn((Cout,), generator=g)
        ref = F.conv2d(x.double(), w.double(), bias=b.double(), stride=stride, padding=R // 2)
        res = torch.randn(ref.shape, generator=g) if use_res else None
        if use_res:
            ref = ref + res.double()
        ref = F.relu(ref)
        args = (nhwc(x, torch.float32), w.permute(0, 2, 3, 1).contiguous().to(DEV), b.to(DEV), stride, R // 2, True)
        kw = dict(residual=None if res is None else nhwc(res, torch.float32))
        exact = ops.conv_bn_act(*args, **kw)
        with ops.f32_split():
            split = ops.conv_bn_act(*args, **kw)
        torch.cuda.synchronize()
        e0 = rel_err(exact.double().permute(0, 3, 1, 2), ref)
        e3 = rel_err(split.double().permute(0, 3, 1, 2), ref)
        print("split conv", (Cin, Cout, R, stride), "exact %.2e split %.2e" % (e0, e3))
        assert e0 < 2e-6 and e3 < 5e-5 and not torch.equal(exact, split)
    xm = torch.randn((300, 512), generator=g)
    wm = torch.randn((256, 512), generator=g) * 0.05
    ref = xm.double() @ wm.double().t()
    with ops.f32_split():
        y = ops.linear_nobias(xm.to(DEV), wm.to(DEV))
    assert rel_err(y.double(), ref) < 5e-5


def test_clip_pool_device():
    """agrl_clip_pool (dense / skipdense test samplers: mean or max over a tracklet's clips) against torch, as the reference
    computes it (train_vidreid_xent_htri.py:471-476); ragged channel count and clip counts 1..11."""
    from torchreid import hip_ops as ops
    g = torch.Generator().manual_seed(5)
    for T, n, D in ((3, 1, 4096), (5, 4, 4096), (2, 11, 2048), (7, 3, 1000)):
        f = torch.randn((T * n, D), generator=g)
        for mode in ("avg", "max"):
            got = ops.clip_pool(f.to(DEV), n, mode).cpu()
            ref = torch.stack([(torch.mean(f[t * n:(t + 1) * n], 0) if mode == "avg" else torch.max(f[t * n:(t + 1) * n], 0)[0]) for t in range(T)])
            if mode == "max":
                assert torch.equal(got, ref)
            else:
                assert (got - ref).abs().max().item() < 1e-6


def test_pose_adjacency_device():
    """agrl_pose_adjacency vs the reference's generate_graph (golden fixture) and vs the oracle on random poses,
    incl. undetected frames, low-confidence keypoints, num_split 8 and the non-pyramid layout."""
    from torchreid import hip_ops as ops
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pose_adjacency.npz"))
    S, width, height, num_split = [int(v) for v in z["meta"]]
    adj = ops.pose_adjacency(torch.from_numpy(z["poses"][None]).to(DEV), torch.from_numpy(z["detected"][None]).to(DEV), height, num_split)
    assert np.array_equal(adj[0].cpu().numpy(), z["adj"])
    rng = np.random.RandomState(17)
    for (B, S, H, ns, pyr) in ((5, 8, 256, 4, True), (3, 16, 256, 4, True), (2, 4, 250, 8, True), (4, 8, 256, 4, False), (2, 6, 100, 2, True)):
        poses = np.stack([rng.rand(B, S, 18) * 128, rng.rand(B, S, 18) * (H + 8) - 4, rng.rand(B, S, 18) * 0.5], axis=-1).astype(np.float32)
        det = rng.rand(B, S) > 0.15
        got = ops.pose_adjacency(torch.from_numpy(poses).to(DEV), torch.from_numpy(det).to(DEV), H, ns, pyr).cpu().numpy()
        for b in range(B):
            sets = [O.pose_part_sets(poses[b, t] if det[b, t] else None, H, ns) for t in range(S)]
            assert np.array_equal(got[b], O.pose_adjacency(sets, ns, pyr)), (B, S, H, ns, pyr, b)


def test_bit_packed_adjacency():
    """The pose graph at 1/32 of the bytes (56 x 56 nodes = 448 B instead of 12.5 KB): agrl_pose_adjacency_bits
    == the bits of agrl_pose_adjacency; agrl_adjacency_pack (device) == adjacency_pack_host (numpy, what a loader does before the
    upload); agrl_graph_finalize_bits and the tracklet form give BITWISE the graph of the fp32 adjacency; V = 56 (two words per
    row), 28 (one), 112 (four, the seq_len-16 graph), 20 (V % 32 != 0 in the last word)."""
    from torchreid import hip_ops as ops
    rng = np.random.RandomState(3)
    for B, S, ns, pyr in ((5, 8, 4, True), (3, 4, 4, True), (2, 16, 4, True), (4, 5, 4, False)):
        H = 256.0
        poses = np.stack([rng.rand(B, S, 18) * 128, rng.rand(B, S, 18) * (H + 8) - 4, rng.rand(B, S, 18) * 0.5], axis=-1).astype(np.float32)
        det = rng.rand(B, S) > 0.15
        pd, dd = torch.from_numpy(poses).to(DEV), torch.from_numpy(det).to(DEV)
        adj = ops.pose_adjacency(pd, dd, H, ns, pyr)
        bits = ops.pose_adjacency(pd, dd, H, ns, pyr, packed=True)
        V = adj.shape[1]
        assert bits.dtype == torch.int32 and tuple(bits.shape) == (B, V, (V + 31) // 32)
        assert torch.equal(bits, ops.adjacency_pack(adj)) and torch.equal(bits.cpu(), ops.adjacency_pack_host(adj.cpu()))
        # unpack on the host and compare element by element
        w = bits.cpu().numpy().view(np.uint32)
        un = ((w[:, :, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(B, V, -1)[:, :, :V].astype(np.float32)
        assert np.array_equal(un, adj.cpu().numpy())
        C = 512
        f = (torch.rand((B, 1, C)) + 0.02 * torch.randn((B, V, C))).to(DEV)
        for use_pose, learn in ((True, True), (True, False)):
            G32 = ops.graph_matrix(f, adj, use_pose, learn)
            Gb = ops.graph_matrix(f, bits, use_pose, learn)
            assert torch.equal(G32, Gb)
            if V <= 64 and V % 4 == 0:
                P32, Gt32 = ops.graph_tracklet_operand(f, adj, use_pose, learn, torch.float32, want_graph=True)
                Pb, Gtb = ops.graph_tracklet_operand(f, bits, use_pose, learn, torch.float32, want_graph=True)
                assert torch.equal(P32, Pb) and torch.equal(Gt32, Gtb)


def test_re_ranking_device():
    """agrl_re_ranking vs the reference (golden fixture) and vs the oracle on a larger random problem; the ranking it
    induces (what evaluate_rank consumes) must be identical."""
    from torchreid import hip_ops as ops
    from torchreid.utils.re_ranking import re_ranking
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "re_ranking.npz"))
    qf, gf = torch.from_numpy(z["qf"]), torch.from_numpy(z["gf"])
    for metric, fn in (("euclidean", O.euclidean_squared), ("cosine", O.cosine)):
        qg, qq, gg = fn(qf, gf).numpy(), fn(qf, qf).numpy(), fn(gf, gf).numpy()
        for tag, kw in (("default", {}), ("k8_3", dict(k1=8, k2=3, lambda_value=0.2)), ("k6_1", dict(k1=6, k2=1, lambda_value=0.5))):
            got = re_ranking(qg, qq, gg, **kw)
            assert got.shape == qg.shape and got.dtype == np.float32
            assert np.abs(got - z[metric + "_" + tag]).max() < 2e-6, (metric, tag)
    g = torch.Generator().manual_seed(9)
    cent = torch.randn((40, 64), generator=g)
    qf = cent[torch.randint(0, 40, (70,), generator=g)] + 0.5 * torch.randn((70, 64), generator=g)
    gf = cent[torch.randint(0, 40, (600,), generator=g)] + 0.5 * torch.randn((600, 64), generator=g)
    qg, qq, gg = (O.euclidean_squared(a, b) for a, b in ((qf, gf), (qf, qf), (gf, gf)))
    ref = O.re_ranking(qg.numpy(), qq.numpy(), gg.numpy())
    got = ops.re_ranking(qg.to(DEV), qq.to(DEV), gg.to(DEV)).cpu().numpy()
    assert np.abs(got - ref).max() < 2e-6
    same = (np.argsort(got, axis=1, kind="stable")[:, :20] == np.argsort(ref, axis=1, kind="stable")[:, :20]).mean()
    assert same > 0.999, same


def test_triplet_mining_and_loss():
    from torchreid import losses, hip_ops as ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn((32, 2048), generator=g)
    pids = torch.arange(8).repeat_interleave(4)
    loss_ref, dap_ref, dan_ref, iap_ref, ian_ref = O.triplet_hard(x, pids, soft=True)
    dap, dan, iap, ian = ops.triplet_hard_mine(x.to(DEV), pids.to(torch.int32).to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(iap.cpu().long(), iap_ref) and torch.equal(ian.cpu().long(), ian_ref)
    assert rel_err(dap, dap_ref) < 1e-5 and rel_err(dan, dan_ref) < 1e-5
    xd = x.to(DEV).requires_grad_(True)
    loss = losses.TripletLoss(soft=True)(xd, pids.to(DEV))
    loss.backward()
    xr = x.clone().requires_grad_(True)
    O.triplet_hard(xr, pids, soft=True)[0].backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-5
    assert rel_err(xd.grad, xr.grad) < 1e-4
    hard = losses.TripletLoss(margin=0.3, soft=False)(x.to(DEV), pids.to(DEV))
    assert abs(hard.item() - O.triplet_hard(x, pids, 0.3, soft=False)[0].item()) < 1e-5
    # the fused native loss (agrl_triplet_loss: mining + value + feature gradient, no host round trip) against the REFERENCE's
    # own TripletLoss (golden fixture: loss and d loss / d features, soft and margin form) ...
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "triplet.npz"))
    n, d, seed = [int(v) for v in z["meta"]]
    feats = torch.randn((n, d), generator=torch.Generator().manual_seed(seed))
    gp = torch.from_numpy(z["pids"])
    for soft, key in ((True, "soft"), (False, "margin")):
        fx = feats.to(DEV).requires_grad_(True)
        val = losses.TripletLoss(margin=0.3, soft=soft)(fx, gp.to(DEV))
        (3.0 * val).backward()   # a non-trivial incoming gradient
        assert abs(val.item() - float(z["loss_" + key])) < 2e-5 * max(1.0, abs(float(z["loss_" + key])))
        assert rel_err(fx.grad / 3.0, torch.from_numpy(z["grad_" + key])) < 1e-4
    # ... an identity with a single sample (its hardest positive is itself: clamped distance, no gradient through it), tied
    # features, and against the oracle with autograd
    x2 = torch.randn((9, 256), generator=g)
    x2[4] = x2[3]
    p2 = torch.tensor([0, 0, 1, 1, 1, 2, 3, 3, 3])
    for soft in (True, False):
        a = x2.to(DEV).requires_grad_(True)
        la = losses.TripletLoss(margin=0.3, soft=soft)(a, p2.to(DEV))
        la.backward()
        b = x2.clone().requires_grad_(True)
        lb = O.triplet_hard(b, p2, 0.3, soft=soft)[0]
        lb.backward()
        assert abs(la.item() - lb.item()) < 2e-5 * max(1.0, abs(lb.item())) and rel_err(a.grad, b.grad) < 1e-4
    # ... and an anchor without any negative: NaN loss instead of the reference's exception (no host sync to raise from)
    assert torch.isnan(losses.TripletLoss()(x2[:3].to(DEV), torch.zeros(3, dtype=torch.long, device=DEV)))
