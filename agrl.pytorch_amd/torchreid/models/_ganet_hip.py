"""MI355X execution of the sibling model ``ganet``'s eval forward (reference torchreid/models/ganet.py:378-424) through the
C-ABI of libagrl_hip.so. Same conv kernels and data layout as vmgn (``_vmgn_hip``); what is specific:

    part nodes   ganet.py:384-400: every pyramid slice through the position attention module, ``pam(slice) + slice``
                 average pooled. The value conv (2048 -> 2048 on every position, three pyramid levels) is never evaluated per
                 position -- pooling is linear and attention rows sum to one, so per node
                     avgpool(gamma * value . attention^T + 2 slice) = gamma * (Wv (X abar) + bv) + 2 mean(slice)
                 : one stacked query / key 1x1 conv over the map (``agrl_conv2d_bn_act``, bias, no ReLU), ``agrl_pam_pool``
                 (energies, softmax, abar, X abar, mean), ONE (F*P, C) x (C, C) Linear (``agrl_linear_nobias``) and
                 ``agrl_pam_combine``. With the module's gamma == 0 (its value at construction) only the means are needed.
    graph layers ganet.py:253-283: diagonal-masked graphs (``mask_diag``), ``input + gamma * h'`` (``keep`` = 1); with the
                 constructor's gamma = 0 a layer's output IS its input (0 * h' adds nothing) and its kernels are skipped.
    tail         ganet.py:402-411: outputs concatenated along the channels, attention pooling over (num_gb + 1) * 2048
                 channels, one BNNeck (the vmgn tail kernel; its global half is fed zeros and dropped).
"""
from __future__ import annotations

import torch

from torchreid import hip_ops as ops
from torchreid import _hip
from torchreid.models._vmgn_hip import (_PRECISIONS, run_stem, _fingerprint, _fold_bn1d, _fold_conv_bn, _pack_stage, _run_block, _run_trunk,
                                         check_packed_range)


def pack_weights(model, device, precision):
    ops.check_precision(precision)
    key = (device.index if device.index is not None else torch.cuda.current_device(), precision)
    cached = model._hip_packs.get(key)
    if cached is not None and (model.hip_static_weights or cached['fingerprint'] == _fingerprint(model)):
        return cached
    first = next(model.parameters())
    if first.device != device:
        raise RuntimeError('model parameters live on {} but the input is on {}'.format(first.device, device))
    dtype = _PRECISIONS[precision]
    s16 = precision == 'fp16x3'
    pam = model.pam_layer
    with torch.no_grad():
        stem_w, stem_b = _fold_conv_bn(model.conv1, model.bn1, torch.float32)
        cq = pam.query_conv.weight.shape[0]
        pack = {
            'dtype': dtype,
            'stem': (stem_w, stem_b),
            'stem_lp': ops.pack_stem_weights_lp16(stem_w) if dtype == ops.LP_DTYPE else None,
            'stem_s16': ops.pack_stem_weights_split16(stem_w) if s16 else None,
            # ('fp16x3': the conv trunk in the split-fp16 arithmetic -- pre-scaled weights, agrl_conv2d_bn_act_split16 -- as in vmgn / gsta;
            # the attention module, the graph layers and the tail stay exact fp32)
            'trunk': (_pack_stage(model.layer1, dtype, split16=s16) + _pack_stage(model.layer2, dtype, split16=s16)
                      + _pack_stage(model.layer3, dtype, seam=True, split16=s16)),
            'l4': _pack_stage(model.layer4, dtype, split16=s16),
            # stacked query / key conv as one OHWI weight (2*Cq, 1, 1, C) + bias; value conv as a Linear weight (C, C) + bias
            'qk_w': torch.cat([pam.query_conv.weight, pam.key_conv.weight], 0).detach().float().permute(0, 2, 3, 1).contiguous().to(dtype),
            'qk_b': torch.cat([pam.query_conv.bias, pam.key_conv.bias], 0).detach().float().contiguous(),
            'cq': cq,
            'v_w': pam.value_conv.weight.detach().float().view(pam.value_conv.weight.shape[0], -1).contiguous().to(dtype),
            'v_b': pam.value_conv.bias.detach().float().contiguous(),
            'pam_gamma': float(pam.gamma.detach()),
            'bn': _fold_bn1d(model.bottleneck),
            'graph': [],
        }
        for layer in model.graph_layers:
            scale, shift = _fold_bn1d(layer.bn)
            pack['graph'].append({'w': layer.linear.weight.detach().to(dtype).contiguous(), 'scale': scale, 'shift': shift,
                                  'slope': float(layer.relu.negative_slope), 'use_pose': bool(layer.use_pose),
                                  'learn_graph': bool(layer.learn_graph)})
    if dtype == ops.LP_DTYPE:
        check_packed_range(pack, 'ganet')
    pack['fingerprint'] = _fingerprint(model)
    model._hip_packs[key] = pack
    return pack


def hip_forward_ganet(model, x, adj, stages=None):
    """Eval forward of ``ganet`` on the GPU: (B,S,3,H,W) fp32, (B,V,V) fp32 -> (B, (num_gb + 1) * 2048) fp32."""
    _hip.lib()
    if x.dtype != torch.float32:
        raise TypeError('frames must be float32, got {}'.format(x.dtype))
    B, S, Cc, H, W = x.shape
    P = model.total_split
    V = S * P
    if tuple(adj.shape) != (B, V, V):
        raise ValueError('adj must be {} for S={} and {} parts, got {}'.format((B, V, V), S, P, tuple(adj.shape)))
    pack = pack_weights(model, x.device, model.hip_precision)
    lp = pack['dtype'] == ops.LP_DTYPE
    splits = list(model.total_split_list)
    with torch.no_grad(), ops.f32_split(model.hip_precision == 'bf16x3'):
        frames = x.reshape(B * S, Cc, H, W)
        a = run_stem(frames, pack)
        a = _run_trunk(a, pack['trunk'], model.hip_fuse_tail)
        for blk in pack['l4']:
            a = _run_block(a, blk)
        C = a.shape[-1]
        # ---- position-attention part nodes
        gamma_p = pack['pam_gamma']
        if gamma_p != 0.0:
            qk = ops.conv_bn_act(a, pack['qk_w'], pack['qk_b'], 1, 0, False)
            xbar, xmean = ops.pam_pool(a, qk, splits)
            operand = xbar.to(pack['dtype']) if lp else xbar
            y = ops.linear_nobias(operand.view(B * V, C), pack['v_w'])
            nodes, nodes_lp = ops.pam_combine(y, pack['v_b'], xmean, gamma_p, want_lp=lp)
        else:
            _, xmean = ops.pam_pool(a, None, splits)
            nodes, nodes_lp = ops.pam_combine(None, None, xmean, 0.0, want_lp=lp)
        del a
        nodes = nodes.view(B, V, C)
        if nodes_lp is not None:
            nodes_lp = nodes_lp.view(B, V, C)
        # ---- graph layers, outputs concatenated with their input
        adj32 = adj.detach().to(torch.float32).contiguous()
        outs = [nodes]
        for g, layer in zip(pack['graph'], model.graph_layers):
            gamma_g = float(layer.gamma)   # a plain attribute in the reference (not a parameter): read at call time
            cur = outs[-1]
            if gamma_g == 0.0:
                outs.append(cur)           # input + 0 * h' (ganet.py:283)
                continue
            operand = nodes_lp if lp else cur
            h = ops.linear_nobias(operand.view(B * V, C), g['w']).view(B, V, C)
            G = ops.graph_matrix(cur, adj32, g['use_pose'], g['learn_graph'], mask_diag=True)
            if stages is not None:
                stages.setdefault('G', []).append(G)
            nxt, nodes_lp = ops.graph_propagate(cur, h, G, g['scale'], g['shift'], gamma_g, g['slope'], want_lp=lp, keep=1.0)
            outs.append(nxt)
        cat = torch.cat(outs, dim=2).contiguous()
        Ct = cat.shape[-1]
        if stages is not None:
            stages.update(nodes=nodes, cat=cat)
        # ---- attention pooling + BNNeck over the concatenated channels
        sqn = ops.row_sqnorm(cat.view(B * V, Ct))
        gsum = torch.zeros((B * S, Ct), dtype=torch.float32, device=x.device)
        ident = (torch.ones_like(pack['bn'][0]), torch.zeros_like(pack['bn'][1]))
        out = ops.attn_pool_bnneck(cat, sqn, gsum, ident[0], ident[1], pack['bn'][0], pack['bn'][1], B, S, P, 1)
        return out[:, Ct:].contiguous()
