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


def _dgrad_3x3_stride2(dy, weight, H, W):
    """Data gradient of a 3x3 / stride 2 / pad 1 conv (the first conv2 of layers 2 and 3) without the zero-inserted dy (four
    times the pixels, three quarters of them zeros): input pixel (iy, ix) only meets the taps whose parity matches,
        iy = 2a     : r = 1 -> dy[a]              iy = 2a + 1 : r = 2 -> dy[a],  r = 0 -> dy[a + 1]
    (columns alike), so dx splits into four parity phases, each a stride-1 conv of dy (one zero row / column appended) with
    a 1x1, 1x2, 2x1 or 2x2 sub-filter: 9 taps in total = the forward's arithmetic."""
    F_, OH, OW, Cout = dy.shape
    Cin = weight.shape[1]
    dyp = torch.nn.functional.pad(dy, (0, 0, 0, 1, 0, 1))                             # (F, OH + 1, OW + 1, Cout)
    dx = torch.empty((F_, H, W, Cin), dtype=dy.dtype, device=dy.device)
    taps = ((1,), (2, 0))                                                            # filter rows met by even / odd input rows
    for py in (0, 1):
        for px in (0, 1):
            nr, nc = (H - py + 1) // 2, (W - px + 1) // 2
            if nr <= 0 or nc <= 0:
                continue
            k = weight[:, :, list(taps[py])][:, :, :, list(taps[px])].permute(1, 2, 3, 0).contiguous()   # (Cin, Rk, Sk, Cout)
            out = ops.conv_bn_act(dyp, k, None, 1, 0, False)                          # (F, OH + 2 - Rk, OW + 2 - Sk, Cin)
            dx[:, py::2, px::2] = out[:, :nr, :nc]
    return dx


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
                dy2 = dy.view(-1, Cout)
                if Cout % 32:                                                        # K granularity of the fp32 GEMM (classifier heads)
                    kpad = 32 - Cout % 32
                    wt, dy2 = torch.nn.functional.pad(wt, (0, kpad)), torch.nn.functional.pad(dy2, (0, kpad))
                d = ops.linear_nobias(dy2.contiguous(), wt.contiguous()).view(dy.shape[0], dy.shape[1], dy.shape[2], Cin)
                if stride == 1:
                    dx = d
                else:                                                                # strided 1x1: gradient lands on the sampled pixels
                    dx = torch.zeros((F_, H, W, Cin), dtype=dy.dtype, device=dy.device)
                    dx[:, ::stride, ::stride] = d
            else:
                # dx[i] = sum_r dyz[i + r - pad'] w[R-1-r]: a stride-1 conv of the (zero-inserted) dy with the flipped filter,
                # output and input channels exchanged
                if (R, S, stride, pad) == (3, 3, 2, 1):
                    dx = _dgrad_3x3_stride2(dy, weight.detach(), H, W)
                else:
                    wf = weight.detach().flip(2, 3).permute(1, 2, 3, 0).contiguous()   # (Cin, R, S, Cout) OHWI
                    if stride == 1:
                        dyz = dy
                    else:
                        dyz = torch.zeros((F_, H, W, Cout), dtype=dy.dtype, device=dy.device)
                        dyz[:, ::stride, ::stride][:, :dy.shape[1], :dy.shape[2]] = dy
                    dx = ops.conv_bn_act(dyz, wf, None, 1, R - 1 - pad, False)
        if ctx.needs_input_grad[1]:
            if ops.conv_wgrad_supported(Cin, Cout):
                dw = ops.conv_wgrad(x, dy, weight.shape, stride, pad)                  # contraction over the pixel axis, in place
            else:                                                                      # classifier widths (702, 625, ...)
                xt = ops.im2col_t(x, R, S, stride, pad)                                # (R*S*Cin, M)
                dyt = ops.im2col_t(dy, 1, 1, 1, 0)                                     # (Cout, M)
                dw = ops.gemm_nt_splitk(dyt, xt).view(Cout, R, S, Cin).permute(0, 3, 1, 2).contiguous()
        return dx, dw, None, None


class HipConv2dStats(torch.autograd.Function):
    """HipConv2d whose forward also returns the batch statistics (mean, biased variance per channel) of its output, taken from
    the conv kernel's epilogue (agrl_conv2d_stats) instead of a second pass over the output -- for the BatchNorm behind it."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.geom = (int(stride), int(pad))
        ctx.split = _split_mode()
        y, mean, var = ops.conv_stats(x, weight.detach().permute(0, 2, 3, 1).contiguous(), int(stride), int(pad))
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar):
        with ops.f32_split(ctx.split):
            return HipConv2d._backward(ctx, dy)


class HipConvFork(torch.autograd.Function):
    """The two consumers of a Bottleneck's input as ONE node, so that their two data gradients are summed inside a GEMM epilogue
    instead of by autograd's accumulation (an extra read-read-write pass over the block's largest tensor, 16 times per step):

        identity shortcut (vmgn.py:58):      x -> conv1(x), x                        backward: dx = dy1 W1 + dshortcut
        downsample shortcut (vmgn.py:60-61): x -> conv1(x), downsample conv(x)       backward: dx = dyd Wd + (dy1 W1)

    both sums as the ``residual`` operand of the data-gradient GEMM (``agrl_conv2d_bn_act``). A strided downsample (first blocks
    of layers 2 / 3) lands on the sampled pixels only: a strided add on a quarter of the tensor."""

    @staticmethod
    def forward(ctx, x, w1, wd, stride_d):
        x = x.contiguous()
        ctx.split = _split_mode()
        ctx.stride_d = int(stride_d)
        y1, m1, v1 = ops.conv_stats(x, w1.detach().permute(0, 2, 3, 1).contiguous(), 1, 0)
        if wd is None:
            ctx.save_for_backward(x, w1)
            ctx.mark_non_differentiable(m1, v1)
            return y1, m1, v1, x
        ctx.save_for_backward(x, w1, wd)
        sc, md, vd = ops.conv_stats(x, wd.detach().permute(0, 2, 3, 1).contiguous(), ctx.stride_d, 0)
        ctx.mark_non_differentiable(m1, v1, md, vd)
        return y1, m1, v1, sc, md, vd

    @staticmethod
    def backward(ctx, dy1, _m1, _v1, dsc, *_rest):
        with ops.f32_split(ctx.split):
            return HipConvFork._backward(ctx, dy1, dsc)

    @staticmethod
    def _backward(ctx, dy1, dsc):
        saved = ctx.saved_tensors
        x, w1 = saved[0], saved[1]
        wd = saved[2] if len(saved) > 2 else None
        dy1, dsc = dy1.contiguous(), dsc.contiguous()
        C1, Cin = w1.shape[0], w1.shape[1]

        def w_t(w):      # dgrad of a 1x1 conv = a 1x1 conv of dy with the transposed matrix: OHWI (Cin, 1, 1, Cout)
            return w.detach().view(w.shape[0], Cin).t().contiguous().view(Cin, 1, 1, w.shape[0])
        dx = dw1 = dwd = None
        if ctx.needs_input_grad[0]:
            if wd is None:
                dx = ops.conv_bn_act(dy1, w_t(w1), None, 1, 0, False, residual=dsc)
            else:
                dx = ops.conv_bn_act(dy1, w_t(w1), None, 1, 0, False)
                if ctx.stride_d == 1:
                    dx = ops.conv_bn_act(dsc, w_t(wd), None, 1, 0, False, residual=dx)
                else:
                    d = ops.conv_bn_act(dsc, w_t(wd), None, 1, 0, False)
                    dx[:, ::ctx.stride_d, ::ctx.stride_d] += d
        if ctx.needs_input_grad[1]:
            dw1 = ops.conv_wgrad(x, dy1, w1.shape, 1, 0)
        if wd is not None and ctx.needs_input_grad[2]:
            dwd = ops.conv_wgrad(x, dsc, wd.shape, ctx.stride_d, 0)
        return dx, dw1, dwd, None


class HipBatchNormAct(torch.autograd.Function):
    """BatchNorm2d in train mode (+ shortcut add) (+ ReLU) on NHWC fp32: out = act(bn(y) + residual)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, residual, relu, eps, slope=0.0, mean=None, var=None, running=None):
        C = y.shape[-1]
        y2 = y.contiguous().view(-1, C)
        if mean is None:
            mean, var = ops.bn_stats(y2)
        else:                                   # handed over by the conv in front (HipConv2dStats / HipConvFork)
            mean, var = mean.detach(), var.detach()
        # folded scale / shift / invstd and the running-statistics update in ONE native launch (was eleven tiny torch kernels)
        rm, rv, nbt, momentum = running if running is not None else (None, None, None, 0.0)
        scale, shift, invstd = ops.bn_fold_train(mean, var, gamma.detach().contiguous(), beta.detach().contiguous(), eps, momentum,
                                                 y2.shape[0], rm, rv, nbt)
        res2 = None if residual is None else residual.contiguous().view(-1, C)
        # with an activation the backward pass needs only the sign of the pre-activation: one bit per element instead of the output
        out, mask = ops.bn_apply(y2, scale, shift, res2, relu, slope, want_mask=True)
        ctx.save_for_backward(y2, mask, mean, invstd, gamma)
        ctx.cfg = (bool(relu), residual is not None, tuple(y.shape), float(slope))
        ctx.mark_non_differentiable(mean, var)
        return out.view(y.shape), mean, var

    @staticmethod
    def backward(ctx, dout, _dmean, _dvar):
        y2, mask, mean, invstd, gamma = ctx.saved_tensors
        relu, has_res, shape, slope = ctx.cfg
        C = shape[-1]
        dy, dz, dgamma, dbeta = ops.bn_backward(dout.contiguous().view(-1, C), None, y2, mean, invstd, gamma.detach().contiguous(),
                                                relu, want_dz=has_res, slope=slope, mask=mask)
        return dy.view(shape), dgamma, dbeta, (dz.view(shape) if has_res else None), None, None, None, None, None, None


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


def _bn_act(bn, y, residual, relu, slope=0.0, stats=None):
    """nn.BatchNorm{1,2}d(train) semantics around HipBatchNormAct, including the running-statistics update. ``slope`` > 0:
    LeakyReLU instead of ReLU. ``stats`` = (mean, biased var) of y when the producing conv already has them."""
    if not bn.training or not bn.track_running_stats:
        raise RuntimeError('the native train path expects BatchNorm layers in train mode with running statistics')
    if bn.momentum is None:   # cumulative moving average: the factor depends on the counter's value -> one host read (not the reference's setting)
        m = 1.0 / float(int(bn.num_batches_tracked) + 1)
    else:
        m = float(bn.momentum)
    mean, var = stats or (None, None)
    out, _, _ = HipBatchNormAct.apply(y, bn.weight, bn.bias, residual, relu, bn.eps, slope, mean, var,
                                      (bn.running_mean, bn.running_var, bn.num_batches_tracked, m))
    return out


def _conv(conv, x):
    return HipConv2d.apply(x, conv.weight, conv.stride[0], conv.padding[0])


def _conv_bn(conv, bn, x, residual, relu):
    """conv -> BatchNorm(train) [+ residual] [-> ReLU]; the statistics come out of the conv's epilogue when its widths allow."""
    if conv.weight.shape[0] % 4 == 0:
        y, mean, var = HipConv2dStats.apply(x, conv.weight, conv.stride[0], conv.padding[0])
        return _bn_act(bn, y, residual, relu, stats=(mean, var))
    return _bn_act(bn, _conv(conv, x), residual, relu)


def bottleneck_train(unit, x):
    """Bottleneck.forward (vmgn.py:45-65) in train mode, NHWC."""
    ds = unit.downsample
    C1, Cin = unit.conv1.weight.shape[:2]
    fork = (x.requires_grad and unit.conv1.stride[0] == 1 and Cin % 32 == 0 and C1 % 32 == 0 and
            (ds is None or (ds[0].weight.shape[0] % 32 == 0 and ds[0].kernel_size == (1, 1))))
    if fork:
        res = HipConvFork.apply(x, unit.conv1.weight, None if ds is None else ds[0].weight, 1 if ds is None else ds[0].stride[0])
        y = _bn_act(unit.bn1, res[0], None, True, stats=(res[1], res[2]))
        shortcut = res[3] if ds is None else _bn_act(ds[1], res[3], None, False, stats=(res[4], res[5]))
    else:                                                                             # odd widths: the plain nodes
        y = _conv_bn(unit.conv1, unit.bn1, x, None, True)
        shortcut = x if ds is None else _conv_bn(ds[0], ds[1], x, None, False)
    y = _conv_bn(unit.conv2, unit.bn2, y, None, True)
    return _conv_bn(unit.conv3, unit.bn3, y, shortcut, True)


def stem_train(model, frames_nchw):
    """conv1 7x7/2 + bn1 + relu + maxpool (vmgn.py:281-284)."""
    w = model.conv1.weight                                                            # (64, 3, 7, 7)
    Cout, Cin, R, S = w.shape
    ncol = R * S * Cin
    ld = -(-ncol // 32) * 32                                                          # 147 -> 160: the GEMM's k granularity
    patches, OH, OW = ops.im2col_rows(frames_nchw.detach(), R, S, 2, 3, ld, nchw=True)  # (F*OH*OW, 160), columns (r, s, c)
    # the weight in the patches' column order; its gradient returns through these (tiny) torch ops
    wp = torch.nn.functional.pad(w.permute(0, 2, 3, 1).reshape(Cout, ncol), (0, ld - ncol))
    y, mean, var = HipConv2dStats.apply(patches.view(1, patches.shape[0], 1, ld), wp.view(Cout, ld, 1, 1), 1, 0)
    y = y.view(frames_nchw.shape[0], OH, OW, Cout)
    y = _bn_act(model.bn1, y, None, True, stats=(mean, var))
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


# =====================================================================================================================
# The tail of the train forward (vmgn.py:296-357) on the same footing: autograd Functions over C-ABI calls.
# =====================================================================================================================
def linear_train(x2d, weight):
    """nn.Linear without bias, forward and backward, as a 1x1 conv node: (M,K) @ (N,K)^T."""
    M, K = x2d.shape
    y = HipConv2d.apply(x2d.contiguous().view(1, M, 1, K), weight.view(weight.shape[0], K, 1, 1), 1, 0)
    return y.view(M, weight.shape[0])


class HipPartPool(torch.autograd.Function):
    """Global feature (mean over S, h, w of x4_1) and part nodes (row-band means of x4_2), vmgn.py:298-308. NHWC in."""

    @staticmethod
    def forward(ctx, x4_1, x4_2, S, splits):
        F_, h, w, C = x4_2.shape
        x4_2 = x4_2.contiguous()
        gsum, nodes, _ = ops.part_pool(x4_1.contiguous(), x4_2, list(splits), want_lp=False)
        ctx.cfg = (int(S), h, w, tuple(splits))
        g_f = gsum.view(F_ // S, S, C).sum(dim=1) / float(S * h * w)
        return g_f, nodes

    @staticmethod
    def backward(ctx, dg, dnodes):
        S, h, w, splits = ctx.cfg
        dx1, dx2 = ops.part_pool_backward(dg.contiguous(), dnodes.contiguous(), S, h, w, list(splits))
        return dx1, dx2, None, None


class HipGraphMatrix(torch.autograd.Function):
    """G = mix(rowL1(adj), rowL1(sim(f))), vmgn.py:114-120, :155-166; gradient w.r.t. f through the similarity."""

    @staticmethod
    def forward(ctx, f, adj, use_pose, learn_graph, mask_diag):
        f = f.contiguous()
        B, V, C = f.shape
        gram = ops.graph_gram(f) if learn_graph else None
        G = ops.graph_finalize(gram, adj, B, V, use_pose, learn_graph, mask_diag)
        ctx.cfg = (bool(use_pose), bool(learn_graph), bool(mask_diag))
        ctx.save_for_backward(f, gram if gram is not None else f.new_zeros(1))
        return G

    @staticmethod
    def backward(ctx, dG):
        use_pose, learn_graph, mask_diag = ctx.cfg
        f, gram = ctx.saved_tensors
        if not learn_graph:
            return None, None, None, None, None
        M = ops.graph_matrix_backward(gram, dG.contiguous(), use_pose, mask_diag)
        return ops.graph_apply(M, f), None, None, None, None


class HipGraphBmm(torch.autograd.Function):
    """msg = bmm(G, h), vmgn.py:168, both gradients."""

    @staticmethod
    def forward(ctx, G, h):
        G, h = G.contiguous(), h.contiguous()
        ctx.save_for_backward(G, h)
        return ops.graph_apply(G, h)

    @staticmethod
    def backward(ctx, dmsg):
        G, h = ctx.saved_tensors
        dmsg = dmsg.contiguous()
        B, V, C = h.shape
        dh = ops.graph_apply(G.transpose(1, 2).contiguous(), dmsg)
        if V <= ops.PAIR_PRODUCT_MAX_V and C % ops.GRAM_CSLICE == 0:
            dG = ops.graph_pair_product(dmsg, h)  # dG[b] = dmsg[b] h[b]^T per tracklet
        else:  # longer clips (V > 144): one GEMM over all (B V) rows, its diagonal V x V blocks are the answer
            full = ops.linear_nobias(dmsg.view(B * V, C), h.view(B * V, C)).view(B, V, B, V)
            idx = torch.arange(B, device=h.device)
            dG = full[idx, :, idx, :].contiguous()
        return dG, dh


class HipAxpby(torch.autograd.Function):
    """a x + b y (the residual mix of vmgn.py:172)."""

    @staticmethod
    def forward(ctx, x, y, a, b):
        ctx.ab = (float(a), float(b))
        return ops.axpby(a, x, b, y)

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.ab
        dout = dout.contiguous()
        return ops.axpby(a, dout), ops.axpby(b, dout), None, None


class HipAttnPool(torch.autograd.Function):
    """GSTA._attention_op + mean over parts, vmgn.py:270-278, :313-317: (B,S,P,C) -> (B,C)."""

    @staticmethod
    def forward(ctx, nodes):
        nodes = nodes.contiguous()
        B, S, P, C = nodes.shape
        ctx.save_for_backward(nodes)
        sqn = ops.row_sqnorm(nodes.view(B * S * P, C))
        key = (nodes.device, C)
        if key not in ops._UNIT:
            ops._UNIT[key] = (torch.ones((C,), dtype=torch.float32, device=nodes.device), torch.zeros((C,), dtype=torch.float32, device=nodes.device))
        one, zero = ops._UNIT[key]
        gsum = torch.zeros((B * S, C), dtype=torch.float32, device=nodes.device)
        _, _, att_f = ops.attn_pool_bnneck(nodes, sqn, gsum, one, zero, one, zero, B, S, P, 1, want_feats=True)
        return att_f

    @staticmethod
    def backward(ctx, datt):
        (nodes,) = ctx.saved_tensors
        return ops.attn_pool_backward(nodes, datt.contiguous())


class HipXent(torch.autograd.Function):
    """CrossEntropyLabelSmooth value + gradient from one native call."""

    @staticmethod
    def forward(ctx, logits, targets, eps):
        loss, dl = ops.xent_label_smooth(logits.detach().float().contiguous(), targets.detach().to(torch.int32).contiguous(), eps)
        ctx.save_for_backward(dl)
        return loss.view(())

    @staticmethod
    def backward(ctx, go):
        (dl,) = ctx.saved_tensors
        return dl * go, None, None


def graph_layer_train(layer, f, adj):
    """GraphLayer.forward (vmgn.py:142-172) in train mode (BatchNorm1d over the N*V rows with batch statistics)."""
    B, V, C = f.shape
    h = linear_train(f.reshape(B * V, C), layer.linear.weight).view(B, V, C)
    G = HipGraphMatrix.apply(f, adj, layer.use_pose, layer.learn_graph, False)
    msg = HipGraphBmm.apply(G, h)
    y = _bn_act(layer.bn, msg.view(B * V, C), None, True, layer.relu.negative_slope).view(B, V, C)
    return HipAxpby.apply(f, y, 1.0 - layer.gamma, layer.gamma)


def tail_train(model, x4_1, x4_2, adj, B, S):
    """Everything of GSTA.forward behind featuremaps under model.train() (vmgn.py:296-357), NHWC maps in ->
    (out_list, f_list) exactly as the module tree returns them."""
    P = model.total_split
    g_f, nodes = HipPartPool.apply(x4_1, x4_2, S, tuple(model.total_split_list))
    C = nodes.shape[-1]
    g_bn = _bn_act(model.global_bottleneck, g_f, None, False)
    f = nodes.view(B, S * P, C)
    adj = adj.detach().to(torch.float32).contiguous()
    for layer in model.graph_layers:
        f = graph_layer_train(layer, f, adj)
    f = f.view(B, S, P, C)
    att_f = HipAttnPool.apply(f)
    att_bn = _bn_act(model.att_bottleneck, att_f, None, False)
    out_list = [linear_train(g_bn, model.global_classifier.weight), linear_train(att_bn, model.att_classifier.weight)]
    f_list = [g_f, att_f]
    if model.consistent_loss:
        assert S >= 5
        for num_frame in [S - 3, S - 2, S - 1]:   # three random frame subsets share the attention head (vmgn.py:327-342)
            pick = torch.sort(torch.randperm(S)[:num_frame])[0].long().to(f.device)
            sub = torch.gather(f, dim=1, index=pick.view(1, num_frame, 1, 1).repeat(B, 1, P, C))
            satt_f = HipAttnPool.apply(sub)
            out_list.append(linear_train(_bn_act(model.att_bottleneck, satt_f, None, False), model.att_classifier.weight))
            f_list.append(satt_f)
    return out_list, f_list


def forward_train(model, x, adj):
    """GSTA.forward under model.train() on the GPU, every arithmetic step a C-ABI call."""
    B, S, C, H, W = x.shape
    prec = getattr(model, 'hip_train_precision', 'fp32')
    if prec not in ('fp32', 'bf16x3'):
        raise ValueError("hip_train_precision must be 'fp32' or 'bf16x3', got {!r}".format(prec))
    with ops.f32_split(prec == 'bf16x3'):
        a = stem_train(model, x.view(B * S, C, H, W))
        for stage in (model.layer1, model.layer2, model.layer3):
            for unit in stage:
                a = bottleneck_train(unit, a)
        maps = []
        for stage in (model.layer4_1, model.layer4_2):
            b = a
            for unit in stage:
                b = bottleneck_train(unit, b)
            maps.append(b)
        return tail_train(model, maps[0], maps[1], adj, B, S)


def forward_train_gsta(model, x, adj):
    """The single-branch sibling ``gsta`` under model.train() on the GPU (reference gsta.py:273-322): the same native nodes as
    vmgn's step -- conv trunk with batch-statistics BatchNorm, part pooling, graph layers, attention pooling, BNNeck, classifier --
    and its consistent loss (one random frame dropped per tracklet, drawn from numpy's global RNG like the reference)."""
    import numpy as np
    B, S, C, H, W = x.shape
    prec = getattr(model, 'hip_train_precision', 'fp32')
    if prec not in ('fp32', 'bf16x3'):
        raise ValueError("hip_train_precision must be 'fp32' or 'bf16x3', got {!r}".format(prec))
    with ops.f32_split(prec == 'bf16x3'):
        a = stem_train(model, x.view(B * S, C, H, W))
        for stage in (model.layer1, model.layer2, model.layer3, model.layer4):
            for unit in stage:
                a = bottleneck_train(unit, a)
        P = model.total_split
        # the part nodes only (gsta has no global branch): the pooling node takes the map for both of its inputs, the unused
        # global feature contributes a zero gradient
        _, nodes = HipPartPool.apply(a, a, S, tuple(model.total_split_list))
        Cf = nodes.shape[-1]
        f = nodes.view(B, S * P, Cf)
        adj = adj.detach().to(torch.float32).contiguous()
        for layer in model.graph_layers:
            f = graph_layer_train(layer, f, adj)
        f = f.view(B, S, P, Cf)
        f_g = HipAttnPool.apply(f)
        bn = _bn_act(model.bottleneck, f_g, None, False)
        sy = sf_g = None
        if model.consistent_loss:
            keep = []
            for _ in range(B):
                idx = list(range(S))
                idx.remove(np.random.randint(S))
                keep.append(idx)
            keep = torch.LongTensor(keep).to(f.device)
            sf = torch.gather(f, dim=1, index=keep.view(B, S - 1, 1, 1).repeat(1, 1, P, Cf))
            sf_g = HipAttnPool.apply(sf)
            sy = linear_train(_bn_act(model.bottleneck, sf_g, None, False), model.classifier.weight)
        y = linear_train(bn, model.classifier.weight)
    if model.loss == {'xent'}:
        return [y, sy] if model.consistent_loss else y
    if model.loss == {'xent', 'htri'}:
        return ([y, sy], [f_g, sf_g]) if model.consistent_loss else (y, f_g)
    raise KeyError('Unsupported loss: {}'.format(model.loss))
