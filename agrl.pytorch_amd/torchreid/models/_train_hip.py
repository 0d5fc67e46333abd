"""MI355X execution of the conv trunk in TRAIN mode -- forward with batch-statistics BatchNorm and the whole backward -- for
the reference's train step (train_vidreid_xent_htri.py:397-413 driving GSTA.forward under model.train(), vmgn.py:280-290 ->
Bottleneck.forward :45-65). 99 % of a train step's arithmetic is here.

torch.autograd keeps the graph (so the reference's driver, losses and optimizer work unchanged); the nodes are
``torch.autograd.Function``s whose forward AND backward are the gfx950 kernels of libagrl_hip.so, no stock torch kernel:

    conv forward          agrl_conv2d_bn_act (exact-fp32 MFMA implicit GEMM, no bias / activation)
    conv data gradient    1x1: agrl_linear_nobias on dy and W^T; 3x3: agrl_conv2d_bn_act with the flipped, transposed filter
                          (stride 2: on the zero-inserted dy)
    conv weight gradient  agrl_im2col_t (channel-major tap-expanded transposes of x and dy) + agrl_gemm_nt_splitk (K = pixels)
    BatchNorm2d (train)   agrl_bn_stats -> agrl_bn_apply (normalise + shortcut add + ReLU in one pass); running statistics
                          updated as nn.BatchNorm2d does (momentum 0.1, unbiased variance); backward agrl_bn_backward
    max pooling           agrl_maxpool3x3s2 / agrl_maxpool3x3s2_backward

Layout: NHWC fp32 between the nodes (the layout of the eval path); NCHW only at the boundary to the stock-torch tail.
"""
from __future__ import annotations

import torch

from torchreid import hip_ops as ops


def _split_mode():
    """The calling thread's GEMM arithmetic switch (ops.f32_split) at forward time; backward re-enters it."""
    return bool(getattr(ops._MODE, 'split', False))


def _conv_forward(x, w_oihw, stride, pad):
    w = w_oihw.detach().permute(0, 2, 3, 1).contiguous()          # OHWI
    return ops.conv_bn_act(x, w, None, stride, pad, False)


class HipConv2d(torch.autograd.Function):
    """NHWC conv without bias. x (F,H,W,Cin) fp32, weight OIHW (the nn.Conv2d parameter itself) -> (F,OH,OW,Cout)."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.geom = (int(stride), int(pad))
        ctx.split = _split_mode()
        return _conv_forward(x, weight, stride, pad)

    @staticmethod
    def backward(ctx, dy):
        with ops.f32_split(ctx.split):
            return HipConv2d._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, pad = ctx.geom
        dy = dy.contiguous()
        Cout, Cin, R, S = weight.shape
        F_, H, W, _ = x.shape
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if R == 1 and S == 1:
                wt = weight.detach().view(Cout, Cin).t().contiguous()                 # (Cin, Cout): dx = dy @ W
                d = ops.linear_nobias(dy.view(-1, Cout), wt).view(dy.shape[0], dy.shape[1], dy.shape[2], Cin)
                if stride == 1:
                    dx = d
                else:                                                                # strided 1x1: gradient lands on the sampled pixels
                    dx = torch.zeros((F_, H, W, Cin), dtype=dy.dtype, device=dy.device)
                    dx[:, ::stride, ::stride] = d
            else:
                # dx[i] = sum_r dyz[i + r - pad'] w[R-1-r]: a stride-1 conv of the (zero-inserted) dy with the flipped filter,
                # output and input channels exchanged
                wf = weight.detach().flip(2, 3).permute(1, 2, 3, 0).contiguous()       # (Cin, R, S, Cout) OHWI
                if stride == 1:
                    dyz = dy
                else:
                    dyz = torch.zeros((F_, H, W, Cout), dtype=dy.dtype, device=dy.device)
                    dyz[:, ::stride, ::stride][:, :dy.shape[1], :dy.shape[2]] = dy
                dx = ops.conv_bn_act(dyz, wf, None, 1, R - 1 - pad, False)
        if ctx.needs_input_grad[1]:
            xt = ops.im2col_t(x, R, S, stride, pad)                                    # (R*S*Cin, M)
            dyt = ops.im2col_t(dy, 1, 1, 1, 0)                                         # (Cout, M)
            dw = ops.gemm_nt_splitk(dyt, xt).view(Cout, R, S, Cin).permute(0, 3, 1, 2).contiguous()
        return dx, dw, None, None


class HipStemConv(torch.autograd.Function):
    """conv1 of the stem (7x7 / 2, 3 -> 64; vmgn.py:281): forward on the implicit-GEMM kernel with the 3 input channels
    zero-padded to its K granularity (32); the weight gradient from the UNPADDED input (147 = 7*7*3 im2col rows instead of
    1568); no data gradient (the frames are the leaves of the graph)."""

    @staticmethod
    def forward(ctx, x3, weight):
        x3 = x3.contiguous()                                                            # (F,H,W,3)
        ctx.save_for_backward(x3, weight)
        ctx.split = _split_mode()
        xp = torch.nn.functional.pad(x3, (0, 29)).contiguous()
        wp = torch.nn.functional.pad(weight.detach(), (0, 0, 0, 0, 0, 29))
        return _conv_forward(xp, wp, 2, 3)

    @staticmethod
    def backward(ctx, dy):
        x3, weight = ctx.saved_tensors
        dy = dy.contiguous()
        Cout, Cin, R, S = weight.shape
        dw = None
        if ctx.needs_input_grad[1]:
            xt = ops.im2col_t(x3, R, S, 2, 3)                                           # (147, M)
            dyt = ops.im2col_t(dy, 1, 1, 1, 0)                                          # (64, M)
            with ops.f32_split(ctx.split):
                dw = ops.gemm_nt_splitk(dyt, xt).view(Cout, R, S, Cin).permute(0, 3, 1, 2).contiguous()
        return None, dw


class HipBatchNormAct(torch.autograd.Function):
    """BatchNorm2d in train mode (+ shortcut add) (+ ReLU) on NHWC fp32: out = act(bn(y) + residual)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, residual, relu, eps):
        C = y.shape[-1]
        y2 = y.contiguous().view(-1, C)
        mean, var = ops.bn_stats(y2)
        invstd = torch.rsqrt(var + eps)
        scale = (gamma.detach() * invstd).contiguous()
        shift = (beta.detach() - mean * scale).contiguous()
        res2 = None if residual is None else residual.contiguous().view(-1, C)
        out = ops.bn_apply(y2, scale, shift, res2, relu)
        ctx.save_for_backward(y2, out, mean, invstd, gamma)
        ctx.cfg = (bool(relu), residual is not None, tuple(y.shape))
        ctx.mark_non_differentiable(mean, var)
        return out.view(y.shape), mean, var

    @staticmethod
    def backward(ctx, dout, _dmean, _dvar):
        y2, out, mean, invstd, gamma = ctx.saved_tensors
        relu, has_res, shape = ctx.cfg
        C = shape[-1]
        dy, dz, dgamma, dbeta = ops.bn_backward(dout.contiguous().view(-1, C), out, y2, mean, invstd, gamma.detach().contiguous(),
                                                relu, want_dz=has_res)
        return dy.view(shape), dgamma, dbeta, (dz.view(shape) if has_res else None), None, None


class HipMaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        out, idx = ops.maxpool3x3s2(x)
        ctx.save_for_backward(idx)
        ctx.hw = (x.shape[1], x.shape[2])
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        return ops.maxpool3x3s2_backward(dout.contiguous(), idx, ctx.hw[0], ctx.hw[1])


def _bn_act(bn, y, residual, relu):
    """nn.BatchNorm2d(train) semantics around HipBatchNormAct, including the running-statistics update."""
    if not bn.training or not bn.track_running_stats:
        raise RuntimeError('the native train path expects BatchNorm layers in train mode with running statistics')
    out, mean, var = HipBatchNormAct.apply(y, bn.weight, bn.bias, residual, relu, bn.eps)
    with torch.no_grad():
        n = y.numel() // y.shape[-1]
        bn.num_batches_tracked += 1
        m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
        bn.running_var.mul_(1 - m).add_(var * (n / max(n - 1, 1)), alpha=m)
    return out


def _conv(conv, x):
    return HipConv2d.apply(x, conv.weight, conv.stride[0], conv.padding[0])


def bottleneck_train(unit, x):
    """Bottleneck.forward (vmgn.py:45-65) in train mode, NHWC."""
    y = _bn_act(unit.bn1, _conv(unit.conv1, x), None, True)
    y = _bn_act(unit.bn2, _conv(unit.conv2, y), None, True)
    shortcut = x
    if unit.downsample is not None:
        shortcut = _bn_act(unit.downsample[1], _conv(unit.downsample[0], x), None, False)
    return _bn_act(unit.bn3, _conv(unit.conv3, y), shortcut, True)


def stem_train(model, frames_nchw):
    """conv1 7x7/2 + bn1 + relu + maxpool (vmgn.py:281-284)."""
    y = HipStemConv.apply(frames_nchw.permute(0, 2, 3, 1), model.conv1.weight)
    y = _bn_act(model.bn1, y, None, True)
    return HipMaxPool.apply(y)


def featuremaps_train(model, frames_nchw):
    """GSTA.featuremaps (vmgn.py:280-290) under model.train() on the GPU -> x4_1, x4_2 as NCHW views for the tail.
    ``model.hip_train_precision``: 'fp32' = exact-fp32 MFMA (bitwise an fp32 fma chain, the parity mode); 'bf16x3' = fp32 tensors,
    every GEMM product as three bf16 MFMAs on the high / low halves of the operands (~1e-5 relative, 1.6 x the rate)."""
    prec = getattr(model, 'hip_train_precision', 'fp32')
    if prec not in ('fp32', 'bf16x3'):
        raise ValueError("hip_train_precision must be 'fp32' or 'bf16x3', got {!r}".format(prec))
    with ops.f32_split(prec == 'bf16x3'):
        return _featuremaps_train(model, frames_nchw)


def _featuremaps_train(model, frames_nchw):
    a = stem_train(model, frames_nchw)
    for stage in (model.layer1, model.layer2, model.layer3):
        for unit in stage:
            a = bottleneck_train(unit, a)
    outs = []
    for stage in ([model.layer4_1, model.layer4_2] if hasattr(model, 'layer4_1') else [model.layer4]):
        b = a
        for unit in stage:
            b = bottleneck_train(unit, b)
        outs.append(b.permute(0, 3, 1, 2))
    return outs
