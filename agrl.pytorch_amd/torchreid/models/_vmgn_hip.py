"""MI355X execution of the VMGN eval forward (reference call site: train_vidreid_xent_htri.py:469/:499
-> GSTA.forward, vmgn.py:292-321) through the C-ABI of libagrl_hip.so.

Data layout in HBM
    frames        fp32 NCHW (B*S,3,H,W)  as handed over by the driver (read once by the stem kernel)
    activations   NHWC (B*S, h, w, C) in the compute dtype (fp32 parity mode / bf16 throughput mode)
    conv weights  OHWI (Cout, R, S, Cin), eval BatchNorm folded in, compute dtype; bias fp32
    part nodes    fp32 (B, V = S*P, 2048) (+ a bf16 copy as the Linear's operand in bf16 mode)
    graph         fp32 (B, V, V)
    embedding     fp32 (B, 4096) = cat(BN(global), BN(attention))

The packed weights are cached per (device, precision) and rebuilt when any parameter / buffer changed
(data pointer or in-place version), unless ``model.hip_static_weights`` is set.
"""
from __future__ import annotations

import os

import torch

from torchreid import hip_ops as ops
from torchreid import _hip

# 'bf16x3': fp32 tensors and layouts of the parity mode, conv / Linear products as three bf16 MFMAs (hip_ops.f32_split)
# 'fp16x3' (round 6): the same fp32 tensors, conv products as three FP16 MFMAs on fp16 high / low halves (22 bits per operand), conv
# weights pre-scaled by a power of two and pre-split at pack time (ops.split16_inloop_weights); GraphLayer, pooling, distance matrix: exact fp32
_PRECISIONS = {'fp32': torch.float32, ops.LP_NAME: ops.LP_DTYPE, 'bf16x3': torch.float32, 'fp16x3': torch.float32}   # ops.LP_NAME: 'fp16' (default build) or 'bf16'


def _fold_conv_bn(conv, bn, dtype, split16=False):
    """conv (no bias) followed by eval BatchNorm2d -> (OHWI weight in dtype, fp32 bias). ``split16``: the fp32 weight pre-scaled by a
    power of two and pre-split for the split-fp16 convolution (ops.split16_inloop_weights; the un-scaling factor rides on the tensor)."""
    w = conv.weight.detach().float()
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    w = (w * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).contiguous()
    if split16:
        assert dtype == torch.float32
        return ops.split16_inloop_weights(w), shift.contiguous()
    return w.to(dtype).contiguous(), shift.contiguous()


def check_packed_range(pack, where="vmgn"):
    """BatchNorm-folded weights are cast to the library's 16-bit type at pack time: with fp16 a checkpoint with a tiny
    running_var or a large gamma can fold to inf (|w| > 65504). Checked once per pack, with the layer named -- the end-of-
    extraction check would blame the activations."""
    def walk(obj, path):
        if torch.is_tensor(obj):
            if obj.dtype == ops.LP_DTYPE and obj.numel() and not bool(torch.isfinite(obj).all()):
                raise FloatingPointError("%s: the BatchNorm-folded weights of %s overflow %s (abs-max of the fp32 fold is beyond its "
                                         "range): load the bf16 build (AGRL_HIP_LP16=bf16) or use hip_precision='fp32'" % (where, path, ops.LP_NAME))
        elif isinstance(obj, dict):
            for k, v in obj.items():
                walk(v, "%s.%s" % (path, k))
        elif isinstance(obj, (list, tuple)):
            for i, v in enumerate(obj):
                walk(v, "%s[%d]" % (path, i))
    walk({k: v for k, v in pack.items() if k != 'fingerprint'}, "pack")


def _fold_bn1d(bn):
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    return scale.contiguous(), shift.contiguous()


def _pack_stage(stage, dtype, seam=False, split16=False):
    blocks = []
    for unit in stage:
        blk = {
            'c1': _fold_conv_bn(unit.conv1, unit.bn1, dtype, split16),
            'c2': _fold_conv_bn(unit.conv2, unit.bn2, dtype, split16),
            'c3': _fold_conv_bn(unit.conv3, unit.bn3, dtype, split16),
            'stride': unit.conv2.stride[0],
            'ds': None,
        }
        if unit.downsample is not None:
            blk['ds'] = _fold_conv_bn(unit.downsample[0], unit.downsample[1], dtype, split16)
            blk['ds_stride'] = unit.downsample[0].stride[0]
            if split16 and blk['ds_stride'] == blk['stride'] and blk['ds'][0].shape[3] % 32 == 0 and blk['c3'][0].shape[3] % 32 == 0:
                # conforming mode: conv3 + downsample as ONE in-loop split GEMM over [block input sampled at the stride | conv2's output]
                # (ops.conv1x1_dual_split16): the fp32 shortcut map -- 537 MB written and read back in layer 1 -- no longer exists
                cout = blk['c3'][0].shape[0]
                wd = ops.split16_true_weights(blk['ds'][0])
                w3 = ops.split16_true_weights(blk['c3'][0])
                blk['dual16'] = (ops.split16_inloop_weights(torch.cat([wd.view(cout, -1), w3.view(cout, -1)], dim=1).contiguous()),
                                 (blk['ds'][1] + blk['c3'][1]).contiguous())
            if dtype == ops.LP_DTYPE and blk['ds_stride'] == 1 and blk['stride'] == 1:
                # conv3 + downsample as ONE GEMM over the concatenated K axis (ops.conv1x1_dual): [w_ds | w3], b_ds + b3
                cout = blk['c3'][0].shape[0]
                blk['dual'] = (torch.cat([blk['ds'][0].view(cout, -1), blk['c3'][0].view(cout, -1)], dim=1).contiguous(),
                               (blk['ds'][1] + blk['c3'][1]).contiguous())
                if (ops.conv1x1_packed_supported(blk['dual'][0]) and blk['dual'][0].shape[1] >= 1536
                        and blk['ds'][0].shape[3] % 128 == 0 and blk['c3'][0].shape[3] % 128 == 0):   # K1, K2 % 128 each
                    blk['dualp'] = ops.conv1x1_pack(blk['dual'][0])     # the same GEMM through the four-wave kernel
            if dtype == ops.LP_DTYPE and blk['ds_stride'] > 1 and blk['ds_stride'] == blk['stride']:
                # first block of layers 2 / 3: the same one-GEMM form with the block input sampled at the stride (ops.conv1x1_packed_dual_strided)
                cout = blk['c3'][0].shape[0]
                dual_w = torch.cat([blk['ds'][0].view(cout, -1), blk['c3'][0].view(cout, -1)], dim=1).contiguous()
                if (ops.conv1x1_packed_supported(dual_w) and blk['ds'][0].shape[3] % 128 == 0 and blk['c3'][0].shape[3] % 128 == 0):
                    blk['dualps'] = ops.conv1x1_pack(dual_w)
                    blk['dualps_bias'] = (blk['ds'][1] + blk['c3'][1]).contiguous()
        if dtype == ops.LP_DTYPE and ops.conv1x1_packed_supported(blk['c1'][0]) and blk['c1'][0].shape[3] >= 1024:
            # layer 4's 2048 -> 512 / 1024 -> 512 convs: the shapes where the packed-weight kernels are ahead of the 8-wave tile
            # (_conv1: ops.conv1x1_packed for 2048 -> 512; ops.conv1x1_packed_res without a residual is an A/B option)
            blk['c1p'] = ops.conv1x1_pack(blk['c1'][0])
        if (dtype == ops.LP_DTYPE and blk['ds'] is None and ops.conv1x1_packed_supported(blk['c3'][0])
                and blk['c3'][0].shape[0] >= 2048):
            # the pool-fused last conv of a layer-4 branch (512 -> 2048 + identity shortcut): two workgroups per CU (ops.conv1x1_packed_res_pool)
            blk['c3p'] = ops.conv1x1_pack(blk['c3'][0])
        if dtype == ops.LP_DTYPE and blk['stride'] == 1 and ops.conv3x3_packed_supported(blk['c2'][0]):
            # layers 3 / 4: the 3x3 weights as per-wave fragment streams for the four-wave kernel (ops.conv3x3_packed)
            blk['c2p'] = ops.conv3x3_pack(blk['c2'][0])
        blocks.append(blk)
    if seam and dtype == ops.LP_DTYPE:
        # conv3 + residual of block i back to back with conv1 of block i + 1 (ops.bottleneck_seam, layer 3): the two static
        # weight matrices re-ordered once into the fragment streams the kernel's waves load straight into registers
        for blk, nxt in zip(blocks[:-1], blocks[1:]):
            if ops.bottleneck_seam_supported(blk['c3'][0], nxt['c1'][0]):
                blk['seam'] = ops.bottleneck_seam_pack(blk['c3'][0], nxt['c1'][0])
                blk['seam_dims'] = (blk['c3'][0].shape[3], blk['c3'][0].shape[0], nxt['c1'][0].shape[0])
    return blocks


def K_OK(w):
    """The in-loop split GEMM needs whole 32-element k-tiles."""
    return w.dim() == 2 and w.shape[1] % 32 == 0


def _fingerprint(model):
    return tuple((t.data_ptr(), t._version) for t in list(model.parameters()) + list(model.buffers()))


def pack_weights(model, device, precision):
    """BN-fold + re-layout every weight of the eval forward for ``device``; cached on the model."""
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
    with torch.no_grad():
        stem_w, stem_b = _fold_conv_bn(model.conv1, model.bn1, torch.float32)
        pack = {
            'dtype': dtype,
            'stem': (stem_w, stem_b),
            'stem_lp': ops.pack_stem_weights_lp16(stem_w) if dtype == ops.LP_DTYPE else None,
            'stem_s16': ops.pack_stem_weights_split16(stem_w) if s16 else None,   # conforming mode: the stem in split-fp16 arithmetic
            'trunk': _pack_stage(model.layer1, dtype, split16=s16) + _pack_stage(model.layer2, dtype, split16=s16) + _pack_stage(model.layer3, dtype, seam=True, split16=s16),
            'graph': [],
        }
        if hasattr(model, 'layer4_1'):   # vmgn: two layer4 branches, two BNNecks
            pack['l4_1'] = _pack_stage(model.layer4_1, dtype, split16=s16)
            pack['l4_2'] = _pack_stage(model.layer4_2, dtype, split16=s16)
            pack['g_bn'] = _fold_bn1d(model.global_bottleneck)
            pack['a_bn'] = _fold_bn1d(model.att_bottleneck)
        else:                            # gsta: one branch, one BNNeck (the unused global half gets an identity)
            pack['l4'] = _pack_stage(model.layer4, dtype, split16=s16)
            pack['a_bn'] = _fold_bn1d(model.bottleneck)
            pack['g_bn'] = (torch.ones_like(pack['a_bn'][0]), torch.zeros_like(pack['a_bn'][1]))
        for layer in model.graph_layers:
            scale, shift = _fold_bn1d(layer.bn)
            gw = layer.linear.weight.detach().to(dtype).contiguous()
            if s16 and K_OK(gw):
                # the GraphLayer's Linear in the split-fp16 arithmetic as well (182 -> ~70 us per layer at 32 tracklets): weight times
                # 2^k, the 2^-k folded into the BatchNorm scale the GEMM's epilogue multiplies the accumulator with (exact)
                gw = ops.split16_inloop_weights(gw)
                scale = (scale * gw.agrl_unscale).contiguous()
                scale.agrl_folded_unscale = gw.agrl_unscale
            pack['graph'].append({
                'w': gw,
                'scale': scale, 'shift': shift,
                'gamma': float(layer.gamma), 'slope': float(layer.relu.negative_slope),
                'use_pose': bool(layer.use_pose), 'learn_graph': bool(layer.learn_graph),
            })
    if s16:
        # the un-scaling factor rides on the weight TENSOR OBJECT (ops.split16_inloop_weights): a copy made behind its back (.to / .contiguous /
        # .clone) would silently run as a 2^k-times-too-large exact-fp32 weight -- checked once per pack
        stages = pack['trunk'] + pack.get('l4_1', []) + pack.get('l4_2', []) + pack.get('l4', [])
        for blk in stages:
            for name in ('c1', 'c2', 'c3', 'ds'):
                if blk.get(name) is not None and not (hasattr(blk[name][0], 'agrl_unscale') and getattr(blk[name][0], 'agrl_presplit', False)):
                    raise RuntimeError("fp16x3 pack: %s lost its pre-scale attributes" % name)
    if s16 and ops.split16_planes_available() and hasattr(model, 'layer4_1') and ops.switch_on('AGRL_HIP_SPLIT16_PLANES'):
        # the conforming mode at speed: behind layer 3's first block every Bottleneck runs on split-fp16 PLANES through the throughput
        # mode's four-wave kernels (ops.conv1x1_split16 / conv3x3_split16); the stem .. layer 3's first block keep fp32 tensors and the
        # in-loop split (agrl_conv2d_bn_act_split16)
        first = len(model.layer1) + len(model.layer2) + 1
        pairs = ops.switch_on('AGRL_HIP_SPLIT16_PAIRS')
        with torch.no_grad():
            ok = all(_pack_planes(blk, pairs) for blk in pack['trunk'][first:] + pack['l4_1'] + pack['l4_2'])
        pack['planes_from'] = first if ok else None
        pack['planes_pairs'] = pairs
    if s16:
        for blk in stages:   # the scaled fp32 copies were for the packers above
            for name in ('c1', 'c2', 'c3', 'ds', 'dual16'):
                if blk.get(name) is not None and hasattr(blk[name][0], 'agrl_scaled'):
                    del blk[name][0].agrl_scaled
    if dtype == ops.LP_DTYPE:
        check_packed_range(pack)
    pack['fingerprint'] = _fingerprint(model)
    model._hip_packs[key] = pack
    return pack


def _pack_planes(blk, pairs=True):
    """The split-fp16 plane operands of one Bottleneck (blk: fp32 folded weights, pre-scaled for the in-loop split) -> blk['p3'];
    False when a shape does not fit the four-wave kernels (the caller then keeps the whole model on the in-loop split).
    ``pairs``: the block's input and output -- the wide tensors the 1x1 kernels are HBM-bound on -- are plane PAIRS [hi | lo 2^11]
    (the third plane repeats the first: ops.split16_plane_weights(pair_first=True)); conv1's and conv2's outputs stay triples."""
    def true_w(pair):
        return ops.split16_true_weights(pair[0])    # undo the per-tensor pre-scale (exact)

    w1, w2, w3 = true_w(blk['c1']), true_w(blk['c2']), true_w(blk['c3'])
    K1, K3c, mid, cout = w1.shape[3], w3.shape[3], w1.shape[0], w3.shape[0]
    if blk['stride'] != 1 or mid % 256 or cout % 256 or K1 % 128 or K3c % 128 or tuple(w2.shape[1:3]) != (3, 3) or w2.shape[3] % 64:
        return False
    p3 = {'pairs': bool(pairs)}
    t, u = ops.split16_plane_weights(w1.view(mid, K1), pair_first=pairs)
    p3['c1'] = (ops.conv1x1_pack(t), u, blk['c1'][1], mid)
    t, u = ops.split16_plane_weights(w2)
    p3['c2'] = (ops.conv3x3_pack(t), u, blk['c2'][1], w2.shape[0])
    if blk['ds'] is not None:
        if blk['ds_stride'] != 1:
            return False
        wd = true_w(blk['ds'])
        Kd = wd.shape[3]
        if Kd % 128:
            return False
        t, u = ops.split16_plane_weights(torch.cat([wd.view(cout, Kd), w3.view(cout, K3c)], dim=1), segments=[Kd, K3c], pair_first=pairs)
        p3['dual'] = (ops.conv1x1_pack(t), u, (blk['ds'][1] + blk['c3'][1]).contiguous(), cout)
    else:
        t, u = ops.split16_plane_weights(w3.view(cout, K3c))
        p3['c3'] = (ops.conv1x1_pack(t), u, blk['c3'][1], cout)
    blk['p3'] = p3
    return True


def _run_block_planes(x3, blk, pool=None):
    """One Bottleneck on split-fp16 planes (x3: (F,h,w,3 C) fp16 = [hi | lo 2^11 | hi]); ``pool`` = (splits, mean): the frame pooling
    in the last conv's epilogue, returns the pooled fp32 tensor instead of the map. vmgn.py:45-65."""
    p3 = blk['p3']
    xp, rop = (1, 2) if p3['pairs'] else (0, 0)   # layout bits: x3 is a pair / residual and result are pairs
    y = ops.conv1x1_split16(x3, p3['c1'][0], p3['c1'][1], p3['c1'][2], p3['c1'][3], layout=xp)
    y = ops.conv3x3_split16(y, p3['c2'][0], p3['c2'][1], p3['c2'][2], p3['c2'][3])
    if 'dual' in p3:
        assert pool is None
        d = p3['dual']
        return ops.conv1x1_split16(x3, d[0], d[1], d[2], d[3], x2=y, layout=xp | rop)
    c = p3['c3']
    if pool is not None:
        return ops.conv1x1_split16_pool(y, c[0], c[1], c[2], c[3], x3, pool[0], pool[1], layout=rop)
    return ops.conv1x1_split16(y, c[0], c[1], c[2], c[3], residual3=x3, layout=rop)


def hip_features_pooled_planes(model, frames, pack, splits):
    """The conv stages of the conforming mode ('fp16x3') at speed: stem .. layer 3's first block on fp32 tensors (in-loop split),
    everything behind it on split-fp16 planes with the pooling fused into the last conv of each layer-4 branch.
    -> gsum (F,C), nodes (F,P,C) fp32, None, hw -- or None when the planes do not apply (frames other than 256 x 128)."""
    first = pack.get('planes_from')
    if first is None or pack['l4_1'][0]['stride'] != 1:
        return None
    H, W = frames.shape[2], frames.shape[3]
    h4, w4 = H, W
    for _ in range(4):
        h4, w4 = (h4 + 1) // 2, (w4 + 1) // 2
    if (h4, w4) != (16, 8):
        return None
    a = ops.stem_split16(frames, pack['stem_s16'][0], pack['stem_s16'][1], pack['stem_s16'][2], pack['stem'][1])
    a = _run_trunk(a, pack['trunk'][:first - 1], False)
    a3 = _run_block_inloop(a, pack['trunk'][first - 1], out_planes=2 if pack.get('planes_pairs') else 3)
    del a
    for blk in pack['trunk'][first:]:
        a3 = _run_block_planes(a3, blk)
    splits = list(splits)
    x = a3
    for blk in pack['l4_1'][:-1]:
        x = _run_block_planes(x, blk)
    gsum = _run_block_planes(x, pack['l4_1'][-1], pool=([1], False))
    x = a3
    for blk in pack['l4_2'][:-1]:
        x = _run_block_planes(x, blk)
    nodes = _run_block_planes(x, pack['l4_2'][-1], pool=(splits, True))
    return gsum.view(gsum.shape[0], gsum.shape[2]), nodes, None, 128


def _conv2(y, blk):
    """conv2 / bn2 / relu of a Bottleneck (vmgn.py:52-54): the packed-weight kernel where it was packed and the map is made of
    whole 16 x 8 blocks, else the general conv."""
    if ('c2p' in blk and y.shape[1] % 16 == 0 and y.shape[2] % 8 == 0 and ops.conv3x3_packed_enabled()
            and (blk['c2'][0].shape[0] != 128 or ops.switch_on('AGRL_HIP_CONV3X3_PACKED_L2'))):   # (layer 2's 128 -> 128: round 5, late)
        return ops.conv3x3_packed(y, blk['c2p'], blk['c2'][1], blk['c2'][0].shape[0], True)
    return ops.conv_bn_act(y, blk['c2'][0], blk['c2'][1], blk['stride'], 1, True)


def _conv1(x, blk):
    """conv1 / bn1 / relu of a Bottleneck (vmgn.py:48-50)."""
    # (layer 4's conv1s through conv1x1_duo_kernel: ahead back to back -- 2048 -> 512 69 us against 73 --; inside the step it measured equal
    # in the middle of round 5 and, on the final tree, 7-13 us ahead per step on two boxes (eight A/B pairs, profiles/r05_ab_conv1_through_duo.txt):
    # on; AGRL_HIP_CONV1X1_DUO_C1=0 = conv1x1_fat_kernel / igemm_wide_kernel)
    if 'c1p' in blk and ops.conv1x1_duo_enabled() and x.is_contiguous() and ops.switch_on('AGRL_HIP_CONV1X1_DUO_C1'):
        return ops.conv1x1_packed_res(x, blk['c1p'], blk['c1'][1], blk['c1'][0].shape[0], None, True)
    if 'c1p' in blk and ops.conv1x1_packed_enabled() and blk['c1'][0].shape[3] >= 2048:
        return ops.conv1x1_packed(x, blk['c1p'], blk['c1'][1], blk['c1'][0].shape[0], True)
    return ops.conv_bn_act(x, blk['c1'][0], blk['c1'][1], 1, 0, True)


def _run_block_inloop(x, blk, out_planes=0):
    """One Bottleneck of the conforming mode on fp32 tensors (weights from ops.split16_inloop_weights): conv1's and conv2's outputs are read
    by one GEMM each and never as numbers, so they are stored with their fp16 halves already formed -- the 3x3 conv's k-loop has no
    VALU work left (bit-identical to splitting in the loop; AGRL_HIP_SPLIT16_PREACT=0 is that form)."""
    pre = ops.switch_on('AGRL_HIP_SPLIT16_PREACT')
    y = ops.conv_bn_act(x, blk['c1'][0], blk['c1'][1], 1, 0, True, out_presplit=pre)
    y = ops.conv_bn_act(y, blk['c2'][0], blk['c2'][1], blk['stride'], 1, True, x_presplit=pre, out_presplit=pre)
    if 'dual16' in blk and ops.switch_on('AGRL_HIP_SPLIT16_DUAL') and x.is_contiguous():
        # conv3 + the downsample conv as ONE GEMM over [x sampled at the stride | y]: no shortcut map in HBM
        # (``out_planes``: the last block in front of the plane kernels writes their input layout itself)
        return ops.conv1x1_dual_split16(x, y, blk['dual16'][0], blk['dual16'][1], blk['stride'], True, x2_presplit=pre, out_planes=out_planes)
    shortcut = x if blk['ds'] is None else ops.conv_bn_act(x, blk['ds'][0], blk['ds'][1], blk['ds_stride'], 0, False)
    out = ops.conv_bn_act(y, blk['c3'][0], blk['c3'][1], 1, 0, True, residual=shortcut, x_presplit=pre)
    return ops.to_split16_planes(out, out_planes) if out_planes else out


def _is_inloop(blk):
    return getattr(blk['c1'][0], 'agrl_presplit', False)


def _run_trunk(a, blocks, fuse_tail=True):
    """layer1..layer3 Bottlenecks. Where the fused kernel exists (layer 1, bf16) the last conv of block i also
    produces the first conv of block i+1 from the tile it still holds in LDS (ops.bottleneck_tail)."""
    z = None  # conv1 output of the current block, when the previous block's tail already computed it
    for i, blk in enumerate(blocks):
        if _is_inloop(blk):
            a = _run_block_inloop(a, blk)
            continue
        nxt = blocks[i + 1] if i + 1 < len(blocks) else None
        y = z if z is not None else ops.conv_bn_act(a, blk['c1'][0], blk['c1'][1], 1, 0, True)
        if fuse_tail and nxt is not None:
            # layer 1: 3x3 + conv3 (+ shortcut) + next conv1 in ONE pass over 8 x 8 pixel tiles (ops.bottleneck_block)
            ds = None if blk['ds'] is None else (blk['ds'][0], blk['ds_stride'])
            if ops.bottleneck_block_supported(y, blk['c2'][0], blk['stride'], blk['c3'][0], nxt['c1'][0], ds) and (
                    ds is None or a.shape[3] == 64):
                if ds is None:
                    a, z = ops.bottleneck_block(y, blk['c2'][0], blk['c2'][1], blk['c3'][0], blk['c3'][1], a,
                                                nxt['c1'][0], nxt['c1'][1])
                else:
                    a, z = ops.bottleneck_block(y, blk['c2'][0], blk['c2'][1], blk['c3'][0], blk['c3'][1], None,
                                                nxt['c1'][0], nxt['c1'][1], shortcut=(a, blk['ds'][0], blk['ds'][1]))
                continue
        y = _conv2(y, blk)
        fusable = fuse_tail and nxt is not None and ops.bottleneck_tail_supported(y, blk['c3'][0], nxt['c1'][0])
        if fusable and blk['ds'] is not None and ops.bottleneck_tail_supported(
                y, blk['c3'][0], nxt['c1'][0], (blk['ds'][0], blk['ds_stride'])) and a.shape[3] == 64:
            # first block of layer 1: the downsample conv rides along as a second k-tile (no shortcut map in HBM)
            a, z = ops.bottleneck_tail(y, blk['c3'][0], blk['c3'][1], None, nxt['c1'][0], nxt['c1'][1],
                                       shortcut=(a, blk['ds'][0], blk['ds'][1]))
            continue
        if ('dualps' in blk and ops.conv1x1_duo_enabled() and ops.switch_on('AGRL_HIP_FUSE_DS_STRIDED')
                and a.is_contiguous() and y.is_contiguous() and a.dtype == y.dtype and a.shape[0] == y.shape[0]
                and tuple(y.shape[1:3]) == tuple((d - 1) // blk['stride'] + 1 for d in a.shape[1:3])):
            # first block of layers 2 / 3: conv3 + the stride-2 downsample conv as ONE GEMM over [a sampled | y]: the shortcut map is
            # neither written nor read back (the next block's conv1 then runs on its own)
            a = ops.conv1x1_packed_dual_strided(a, y, blk['dualps'], blk['dualps_bias'], blk['c3'][0].shape[0], blk['stride'], True)
            z = None
            continue
        shortcut = a if blk['ds'] is None else ops.conv_bn_act(a, blk['ds'][0], blk['ds'][1], blk['ds_stride'], 0, False)
        if fuse_tail and nxt is not None and 'seam' in blk and (y.numel() // y.shape[-1]) % 128 == 0 and ops.seam_enabled():
            a, z = ops.bottleneck_seam(y, blk['seam'], blk['c3'][1], shortcut, nxt['c1'][1], blk['seam_dims'])
            continue
        if fusable:
            a, z = ops.bottleneck_tail(y, blk['c3'][0], blk['c3'][1], shortcut, nxt['c1'][0], nxt['c1'][1])
        else:
            a = ops.conv_bn_act(y, blk['c3'][0], blk['c3'][1], 1, 0, True, residual=shortcut)
            z = None
    return a


def _run_block(x, blk, pool=None):
    """One Bottleneck. ``pool`` = (splits, mean, want_lp): fuse the frame pooling into the last conv's epilogue and
    return the pooled tensors instead of the activation map (which is then never written to HBM)."""
    if _is_inloop(blk):
        assert pool is None
        return _run_block_inloop(x, blk)
    y = _conv1(x, blk)
    y = _conv2(y, blk)
    if (pool is None and 'dualp' in blk and ops.conv1x1_packed_enabled() and ops.switch_on('AGRL_HIP_FUSE_DS')
            and x.shape[:3] == y.shape[:3] and x.dtype == y.dtype and x.is_contiguous() and y.is_contiguous()):
        return ops.conv1x1_packed(x, blk['dualp'], blk['dual'][1], blk['dual'][0].shape[0], True, x2=y, duo=ops.conv1x1_duo_enabled())   # (two workgroups per CU: 167 us against conv1x1_fat_kernel's 185)
    if pool is None and 'dual' in blk and ops.conv1x1_dual_supported(x, y, blk['dual'][0]):
        return ops.conv1x1_dual(x, y, blk['dual'][0], blk['dual'][1], True)
    shortcut = x if blk['ds'] is None else ops.conv_bn_act(x, blk['ds'][0], blk['ds'][1], blk['ds_stride'], 0, False)
    duo = 'c3p' in blk and blk['ds'] is None and ops.conv1x1_duo_enabled() and shortcut.is_contiguous() and y.is_contiguous()
    if pool is not None:
        if duo and tuple(y.shape[1:3]) == (16, 8):
            return ops.conv1x1_packed_res_pool(y, blk['c3p'], blk['c3'][1], blk['c3'][0].shape[0], shortcut, pool[0], pool[1], pool[2])
        return ops.conv1x1_bn_act_pool(y, blk['c3'][0], blk['c3'][1], shortcut, pool[0], pool[1], pool[2])
    if duo and ops.switch_on('AGRL_HIP_CONV1X1_DUO_RES'):   # the same kernel with the map stored (in the step: the layer-4 pointwise family 0.968-0.981 ms with it, 0.984-0.993 without, three A/B pairs on one box)
        return ops.conv1x1_packed_res(y, blk['c3p'], blk['c3'][1], blk['c3'][0].shape[0], shortcut)
    return ops.conv_bn_act(y, blk['c3'][0], blk['c3'][1], 1, 0, True, residual=shortcut)


def run_stem(frames, pack):
    """conv1 / bn1 / relu / maxpool (vmgn.py:281-284) in the pack's arithmetic: the 16-bit MFMA stem, the split-fp16 one of the conforming
    mode, or the exact-fp32 one."""
    if pack['stem_lp'] is not None:
        return ops.stem_lp16(frames, pack['stem_lp'], pack['stem'][1])
    if pack.get('stem_s16') is not None:
        return ops.stem_split16(frames, pack['stem_s16'][0], pack['stem_s16'][1], pack['stem_s16'][2], pack['stem'][1])
    return ops.stem(frames, pack['stem'][0], pack['stem'][1], pack['dtype'])


def hip_featuremaps(model, frames, pack):
    """(F,3,H,W) fp32 NCHW -> x4_1, x4_2 NHWC (F,h,w,2048). reference vmgn.py:280-290."""
    a = run_stem(frames, pack)
    a = _run_trunk(a, pack['trunk'], getattr(model, 'hip_fuse_tail', True))
    x4_1 = a
    for blk in pack['l4_1']:
        x4_1 = _run_block(x4_1, blk)
    x4_2 = a
    for blk in pack['l4_2']:
        x4_2 = _run_block(x4_2, blk)
    return x4_1, x4_2


def hip_features_pooled(model, frames, pack, splits, want_lp=True):
    """Conv stages with the global / part pooling fused into the last conv of each layer4 branch (bf16, 16x8 maps):
    -> gsum (F,C) per-frame sums, nodes (F,P,C) fp32, nodes_lp bf16, hw. None when the fusion does not apply."""
    if pack['dtype'] != ops.LP_DTYPE or pack['l4_1'][0]['stride'] != 1:
        return None
    # applicability is decided from the input size BEFORE anything is launched (a late bail-out would make the caller
    # recompute stem + trunk): the fused epilogue needs 16 x 8 = 128-pixel layer-4 maps, i.e. frames of 256 x 128
    H, W = frames.shape[2], frames.shape[3]
    h4, w4 = H, W
    for _ in range(4):   # stem conv /2, maxpool /2, layer2 /2, layer3 /2 (kernel 7 pad 3 / kernel 3 pad 1: ceil halving)
        h4, w4 = (h4 + 1) // 2, (w4 + 1) // 2
    if (h4, w4) != (16, 8):
        return None
    a = ops.stem_lp16(frames, pack['stem_lp'], pack['stem'][1])
    a = _run_trunk(a, pack['trunk'], getattr(model, 'hip_fuse_tail', True))
    assert a.shape[1] == 16 and a.shape[2] == 8
    splits = list(splits)
    # (the two branches on two HIP streams, and the graph matrix on a side stream under the Linear, were options until round 4:
    # both measured no faster -- every kernel here fills the chip -- and were retired)
    x4_1 = a
    for blk in pack['l4_1'][:-1]:
        x4_1 = _run_block(x4_1, blk)
    gsum, _ = _run_block(x4_1, pack['l4_1'][-1], pool=([1], False, False))
    x4_2 = a
    for blk in pack['l4_2'][:-1]:
        x4_2 = _run_block(x4_2, blk)
    nodes, nodes_lp = _run_block(x4_2, pack['l4_2'][-1], pool=(splits, True, want_lp))
    return gsum.view(gsum.shape[0], gsum.shape[2]), nodes, nodes_lp, 128


def gcn_commute_enabled(model=None):
    """The commuted GraphLayer ((G f) W^T, one GEMM with the BatchNorm / LeakyReLU / residual epilogue) is the default;
    ``model.hip_gcn_commute = False`` or AGRL_HIP_GCN_COMMUTE=0 selects Linear -> message pass (the round-1/2 form)."""
    import os
    if model is not None and hasattr(model, 'hip_gcn_commute'):
        return bool(model.hip_gcn_commute)
    return ops.switch_on('AGRL_HIP_GCN_COMMUTE')


def hip_graph_layers(nodes, nodes_lp, adj, pack, stages=None, commute=True):
    """GraphLayer x num_gb on (B,V,C) fp32 nodes. reference vmgn.py:311-312 -> :142-172."""
    lp = pack['dtype'] == ops.LP_DTYPE
    B, V, C = nodes.shape
    n_layers = len(pack['graph'])
    for i, g in enumerate(pack['graph']):
        if commute:
            # G (f W^T) = (G f) W^T: graph -> P = G f (written once, in the GEMM's operand dtype) -> ONE GEMM whose epilogue
            # applies BatchNorm1d + LeakyReLU + the residual mix. h never exists; f and out cross HBM once each.
            pre = False
            if ops.graph_tracklet_operand_supported(nodes):   # many tracklets per GPU: one workgroup per tracklet, one launch
                P, G = ops.graph_tracklet_operand(nodes, adj, g['use_pose'], g['learn_graph'], pack['dtype'], want_graph=stages is not None)
            else:
                G = ops.graph_matrix(nodes, adj, g['use_pose'], g['learn_graph'])
                # conforming mode: P is read by the GEMM only -- written with its fp16 halves already formed
                pre = (getattr(g['w'], 'agrl_presplit', False) and ops.graph_apply_presplit_supported(nodes)
                       and ops.switch_on('AGRL_HIP_SPLIT16_PREACT'))
                P = ops.graph_apply_operand(G, nodes, pack['dtype'], presplit=pre)
            if stages is not None:
                stages['G%d' % i] = G
            nodes = ops.graph_linear_mix(P, g['w'], nodes, g['scale'], g['shift'], g['gamma'], g['slope'], p_presplit=pre)
            continue
        if lp and nodes_lp is None:   # A/B form entered without the pooled bf16 copy: native conversion kernel
            nodes_lp = ops.row_l2_normalize(nodes.view(B * V, C), False, ops.LP_DTYPE).view(B, V, C)
        operand = nodes_lp if lp else nodes
        h = ops.linear_nobias(operand.view(B * V, C), g['w']).view(B, V, C)
        G = ops.graph_matrix(nodes, adj, g['use_pose'], g['learn_graph'])
        if stages is not None:
            stages['G%d' % i] = G
        nodes, nodes_lp = ops.graph_propagate(nodes, h, G, g['scale'], g['shift'], g['gamma'], g['slope'],
                                              want_lp=lp and i + 1 < n_layers)
    return nodes


def hip_forward(model, x, adj, return_feats=False, stages=None):
    """Eval forward on the GPU: (B,S,3,H,W) fp32, (B,V,V) fp32 -> (B,4096) fp32. ``stages``: an optional dict that
    receives the intermediate tensors of the path (per-frame sums of x4_1, part nodes, graphs, graph output, pre-BN
    features) for the stage-by-stage parity tests."""
    _hip.lib()  # fail loudly before touching anything if the extension is missing
    if x.dtype != torch.float32:
        raise TypeError('frames must be float32, got {}'.format(x.dtype))
    B, S, Cc, H, W = x.shape
    P = model.total_split
    V = S * P
    packed_adj = ops.adjacency_is_packed(adj)   # int32 (B, V, ceil(V/32)): the bit-packed graph (hip_ops.adjacency_pack*)
    if tuple(adj.shape) != ((B, V, (V + 31) // 32) if packed_adj else (B, V, V)):
        raise ValueError('adj must be {} (fp32) or {} (bit-packed int32) for S={} and {} parts, got {}'.format(
            (B, V, V), (B, V, (V + 31) // 32), S, P, tuple(adj.shape)))
    pack = pack_weights(model, x.device, model.hip_precision)
    lp = pack['dtype'] == ops.LP_DTYPE
    with torch.no_grad(), ops.f32_split(model.hip_precision == 'bf16x3'):
        frames = x.reshape(B * S, Cc, H, W)
        commute = gcn_commute_enabled(model)
        fused = hip_features_pooled(model, frames, pack, model.total_split_list, want_lp=not commute) if model.hip_fuse_pool else None
        if fused is None and pack.get('planes_from') is not None:
            fused = hip_features_pooled_planes(model, frames, pack, model.total_split_list)
        if fused is not None:
            gsum, nodes, nodes_lp, hw = fused
            C = nodes.shape[-1]
        else:
            x4_1, x4_2 = hip_featuremaps(model, frames, pack)
            F_, h, w, C = x4_1.shape
            hw = h * w
            gsum, nodes, nodes_lp = ops.part_pool(x4_1, x4_2, model.total_split_list, want_lp=lp and not commute)
            del x4_1, x4_2
        nodes = nodes.view(B, V, C)
        if nodes_lp is not None:
            nodes_lp = nodes_lp.view(B, V, C)
        adj32 = adj.detach().contiguous() if packed_adj else adj.detach().to(torch.float32).contiguous()
        if stages is not None:
            stages.update(gsum=gsum, hw=hw, nodes=nodes)
        nodes = hip_graph_layers(nodes, nodes_lp, adj32, pack, stages, commute=commute)
        model._hip_query = None
        if ops.attn_tail_supported(S, P, C, B):
            # many tracklets per GPU: node norms + attention pooling + BNNeck + the distance matrix's query operand in one launch
            want = return_feats or stages is not None
            out, feats, query, _ = ops.attn_tail(nodes, gsum, pack['g_bn'][0], pack['g_bn'][1], pack['a_bn'][0], pack['a_bn'][1],
                                                 B, S, P, hw, want_feats=want, query_dtype=ops.LP_DTYPE if lp else torch.float32)
            model._hip_query = ops.QueryOperandCache(out, query)
            res = (out, feats[0], feats[1]) if want else out
        else:
            sqn = ops.row_sqnorm(nodes.view(B * V, C))
            res = ops.attn_pool_bnneck(nodes, sqn, gsum, pack['g_bn'][0], pack['g_bn'][1], pack['a_bn'][0],
                                       pack['a_bn'][1], B, S, P, hw, want_feats=return_feats or stages is not None)
        if stages is not None:
            stages.update(nodes_out=nodes, out=res[0], g_f=res[1], att_f=res[2])
            return res if return_feats else res[0]
        return res


def hip_forward_gsta(model, x, adj):
    """Eval forward of the single-branch ``gsta`` on the GPU: (B,S,3,H,W) fp32, (B,V,V) fp32 -> (B,2048) fp32.
    reference gsta.py:273-298. Same kernels as vmgn; the attention tail kernel writes cat(BN(global), BN(attention)) and
    only its second half exists for this model (the global half is fed zeros)."""
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
        hw = (a.shape[1] // pack['l4'][0]['stride']) * (a.shape[2] // pack['l4'][0]['stride'])
        if lp and model.hip_fuse_pool and a.shape[1] * a.shape[2] == 128 and pack['l4'][0]['stride'] == 1:
            for blk in pack['l4'][:-1]:
                a = _run_block(a, blk)
            nodes, nodes_lp = _run_block(a, pack['l4'][-1], pool=(splits, True, True))
            hw = 128
        else:
            for blk in pack['l4']:
                a = _run_block(a, blk)
            hw = a.shape[1] * a.shape[2]
            _, nodes, nodes_lp = ops.part_pool(a, a, splits, want_lp=lp)
        C = nodes.shape[-1]
        nodes = nodes.view(B, V, C)
        if nodes_lp is not None:
            nodes_lp = nodes_lp.view(B, V, C)
        nodes = hip_graph_layers(nodes, nodes_lp, adj.detach().to(torch.float32).contiguous(), pack)
        sqn = ops.row_sqnorm(nodes.view(B * V, C))
        gsum = torch.zeros((B * S, C), dtype=torch.float32, device=x.device)
        out = ops.attn_pool_bnneck(nodes, sqn, gsum, pack['g_bn'][0], pack['g_bn'][1], pack['a_bn'][0], pack['a_bn'][1], B, S, P, hw)
        return out[:, C:].contiguous()
