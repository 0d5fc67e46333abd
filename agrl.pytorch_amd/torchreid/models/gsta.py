"""GSTA (``gsta``): the single-branch predecessor of VMGN -- ResNet50 (last stride 1) + part pooling + the same
pose-guided adaptive ``GraphLayer`` x num_gb + attention temporal pooling + one BNNeck -> (B, 2048).

Drop-in for ``torchreid/models/gsta.py`` of weleen/AGRL.pytorch: same factory signature
(reference gsta.py:339-357), same module tree / state-dict keys (gsta.py:173-222), same call contract and return
conventions (gsta.py:273-336). It shares every kernel with vmgn: CUDA tensors in ``eval()`` run
``_vmgn_hip.hip_forward_gsta`` (stem, implicit-GEMM convs with the part pooling fused into the last one, graph layers,
attention tail); CPU tensors and train mode use the stock-torch module tree below.
"""
from __future__ import absolute_import
from __future__ import division

__all__ = ['gsta']

import os

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from torchreid.utils.reidtools import calc_splits
from .vmgn import Bottleneck, GraphLayer, RESNET50_STAGES, _make_stage


class GSTASingle(nn.Module):
    """The reference names this class GSTA as well (gsta.py:173); renamed here only to keep the two apart."""

    def __init__(self, num_classes, loss, block, layers, num_split, pyramid_part, num_gb, use_pose, learn_graph,
                 consistent_loss, nonlinear='relu', **kwargs):
        super(GSTASingle, self).__init__()
        assert block is Bottleneck
        self.loss = loss
        self.feature_dim = 512 * block.expansion
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        ch = 64
        self.layer1, ch = _make_stage(ch, 64, layers[0], 1)
        self.layer2, ch = _make_stage(ch, 128, layers[1], 2)
        self.layer3, ch = _make_stage(ch, 256, layers[2], 2)
        self.layer4, ch = _make_stage(ch, 512, layers[3], 1)

        self.num_split = num_split
        self.total_split_list = calc_splits(num_split) if pyramid_part else [num_split]
        self.total_split = sum(self.total_split_list)
        self.parts_avgpool = nn.ModuleList(nn.AdaptiveAvgPool2d((n, 1)) for n in self.total_split_list)
        self.num_gb = num_gb
        self.graph_layers = nn.ModuleList(
            GraphLayer(in_features=self.feature_dim, out_features=self.feature_dim, use_pose=use_pose, learn_graph=learn_graph)
            for _ in range(num_gb))
        self.consistent_loss = consistent_loss
        self.bottleneck = nn.BatchNorm1d(self.feature_dim)
        self.bottleneck.bias.requires_grad_(False)
        self.classifier = nn.Linear(self.feature_dim, num_classes, bias=False)
        self._init_params()

        # MI355X path configuration (not part of the state dict)
        self.hip_precision = os.environ.get('AGRL_HIP_PRECISION', 'fp32')
        from torchreid import hip_ops as _ops   # a precision the loaded library cannot serve fails HERE, not at the first forward
        _ops.check_precision(self.hip_precision)
        self.hip_static_weights = False
        self.hip_fuse_pool = os.environ.get('AGRL_HIP_FUSE_POOL', '1') != '0'
        self.hip_fuse_tail = os.environ.get('AGRL_HIP_FUSE_TAIL', '1') != '0'
        self.hip_train = os.environ.get('AGRL_HIP_TRAIN', '1') != '0'   # train-mode forward + backward on the HIP kernels
        self.hip_train_precision = os.environ.get('AGRL_HIP_TRAIN_PRECISION', 'fp32')
        self._hip_packs = {}

    def _init_params(self):
        """reference gsta.py:240-256"""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)

    def featuremaps(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))

    def _attention_op(self, feat):
        att = F.normalize(feat.norm(p=2, dim=3, keepdim=True), p=1, dim=1)
        return (feat * att).sum(dim=1)

    def forward(self, x, adj, *args):
        if x.is_cuda and not self.training:
            from torchreid.models._vmgn_hip import hip_forward_gsta
            return hip_forward_gsta(self, x, adj)
        if x.is_cuda and self.training and self.hip_train:
            if x.dtype != torch.float32:
                raise TypeError('the native train step takes float32 frames, got {}; set model.hip_train = False for the '
                                'stock-torch module tree'.format(x.dtype))
            from torchreid.models._train_hip import forward_train_gsta
            return forward_train_gsta(self, x, adj)
        B, S, C, H, W = x.size()
        fm = self.featuremaps(x.view(B * S, C, H, W))
        c = fm.size(1)
        parts = [pool(fm).view(B, S, c, n) for pool, n in zip(self.parts_avgpool, self.total_split_list)]
        f = torch.cat(parts, dim=3).transpose(2, 3).contiguous().view(B, S * self.total_split, c)
        for layer in self.graph_layers:
            f = layer(f, adj)
        f = f.view(B, S, self.total_split, c)
        f_g = self._attention_op(f).mean(dim=1).view(B, -1)
        bn = self.bottleneck(f_g)
        if self.consistent_loss and self.training:
            # one random frame dropped per tracklet, drawn from numpy's global RNG (reference gsta.py:300-311)
            keep = []
            for _ in range(B):
                idx = list(range(S))
                idx.remove(np.random.randint(S))
                keep.append(idx)
            keep = torch.LongTensor(keep).to(f.device)
            sf = torch.gather(f, dim=1, index=keep.view(B, S - 1, 1, 1).repeat(1, 1, f.size(2), f.size(3)))
            sf_g = self._attention_op(sf).mean(dim=1).view(B, -1)
            sy = self.classifier(self.bottleneck(sf_g))
        if not self.training:
            return bn
        y = self.classifier(bn)
        if self.loss == {'xent'}:
            return [y, sy] if self.consistent_loss else y
        elif self.loss == {'xent', 'htri'}:
            return ([y, sy], [f_g, sf_g]) if self.consistent_loss else (y, f_g)
        raise KeyError('Unsupported loss: {}'.format(self.loss))

    def invalidate_hip_cache(self):
        self._hip_packs.clear()


def gsta(num_classes, loss, last_stride, num_split, num_gb, num_scale, pyramid_part, use_pose, learn_graph,
         pretrained=True, consistent_loss=False, **kwargs):
    """Factory registered as ``'gsta'`` (reference gsta.py:339-357). Never touches the network: ``pretrained`` only takes
    effect through ``AGRL_PRETRAINED_RESNET50`` (a local resnet50-19c8e357.pth)."""
    model = GSTASingle(num_classes=num_classes, loss=loss, block=Bottleneck, layers=list(RESNET50_STAGES), last_stride=last_stride,
                       num_split=num_split, pyramid_part=pyramid_part, num_gb=num_gb, use_pose=use_pose, learn_graph=learn_graph,
                       consistent_loss=consistent_loss, nonlinear='relu', **kwargs)
    path = os.environ.get('AGRL_PRETRAINED_RESNET50', '')
    if pretrained and path and os.path.isfile(path):
        own = model.state_dict()
        picked = {k: v for k, v in torch.load(path, map_location='cpu').items() if k in own and own[k].size() == v.size()}
        own.update(picked)
        model.load_state_dict(own)
    return model
