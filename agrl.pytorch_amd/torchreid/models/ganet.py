"""GANet (``ganet``): single-branch ResNet50 + position-attention part nodes + diagonal-masked pose-guided graph
layers whose outputs are concatenated + attention temporal pooling + one BNNeck -> (B, (num_gb + 1) * 2048).

Drop-in for ``torchreid/models/ganet.py`` of weleen/AGRL.pytorch: same factory signature
(reference ganet.py:458-477, including the required ``knn`` the reference's driver never passes), same module tree /
state-dict keys (ganet.py:285-336: ``pam_layer.{query,key,value}_conv``, ``pam_layer.gamma``, ``cam_layer.gamma``,
``graph_layers.*``, ``bottleneck``, ``classifier``), same call contract and return conventions (ganet.py:378-443).

What differs from vmgn / gsta (and is what the kernels get flags for):
  * part nodes: every pyramid slice (h // n rows, remainder dropped) goes through the position attention module and
    ``pam(slice) + slice`` is average pooled (ganet.py:384-400);
  * GraphLayer: the diagonal of the pose graph and of the learned similarity is zeroed before the row-L1 normalisation,
    and the residual form is ``input + gamma * h'`` with gamma = 0 by default (ganet.py:175, :253-283);
  * the graph layers' outputs are concatenated with their input along the channel axis (ganet.py:402-405).

CUDA tensors in ``eval()`` run ``_ganet_hip.hip_forward_ganet`` (the vmgn conv kernels, ``agrl_pam_pool``, the graph
kernels with the mask flag, the attention tail); CPU tensors and train mode use the stock-torch module tree below.
"""
from __future__ import absolute_import
from __future__ import division

__all__ = ['ganet']

import os

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from torchreid.utils.reidtools import calc_splits
from .vmgn import Bottleneck, RESNET50_STAGES, _make_stage


class PAM_Module(nn.Module):
    """Position attention over the h*w positions of a map (reference ganet.py:98-136): returns
    (gamma * attended + x, attended)."""

    def __init__(self, in_dim):
        super(PAM_Module, self).__init__()
        self.channel_in = in_dim
        self.query_conv = nn.Conv2d(in_dim, in_dim // 8, kernel_size=1)
        self.key_conv = nn.Conv2d(in_dim, in_dim // 8, kernel_size=1)
        self.value_conv = nn.Conv2d(in_dim, in_dim, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x):
        n, c, h, w = x.size()
        query = self.query_conv(x).view(n, -1, h * w).permute(0, 2, 1)
        key = self.key_conv(x).view(n, -1, h * w)
        attention = self.softmax(torch.bmm(query, key))
        value = self.value_conv(x).view(n, -1, h * w)
        attended = torch.bmm(value, attention.permute(0, 2, 1)).view(n, c, h, w)
        return self.gamma * attended + x, attended


class CAM_Module(nn.Module):
    """Channel attention (reference ganet.py:139-169). Constructed (its ``gamma`` is a state-dict key) but not applied by
    the forward pass -- the reference has the call commented out (ganet.py:396)."""

    def __init__(self, in_dim):
        super(CAM_Module, self).__init__()
        self.channel_in = in_dim
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)

    def forward(self, x):
        n, c, h, w = x.size()
        flat = x.view(n, c, -1)
        energy = torch.bmm(flat, flat.permute(0, 2, 1))
        energy = torch.max(energy, -1, keepdim=True)[0].expand_as(energy) - energy
        out = torch.bmm(self.softmax(energy), flat).view(n, c, h, w)
        return self.gamma * out + x


class MaskedGraphLayer(nn.Module):
    """ganet's GraphLayer (reference ganet.py:172-283; the reference names it GraphLayer too): self-loops are masked out
    of both graphs, ``out = input + gamma * LeakyReLU(BN(G @ (input W^T)))`` with gamma = 0 unless set."""

    def __init__(self, in_features, out_features, learn_graph=True, use_pose=True, dist_method='l2', gamma=0, k=4, **kwargs):
        super(MaskedGraphLayer, self).__init__()
        assert use_pose or learn_graph
        if dist_method != 'l2':
            raise NotImplementedError("dist_method={!r}: only 'l2' is on the ganet path".format(dist_method))
        self.in_features = in_features
        self.out_features = out_features
        self.learn_graph = learn_graph
        self.use_pose = use_pose
        self.dist_method = dist_method
        self.gamma = gamma
        self.linear = nn.Linear(in_features, out_features, bias=False)
        self.bn = nn.BatchNorm1d(out_features)
        self.relu = nn.LeakyReLU(0.1)
        nn.init.normal_(self.linear.weight, 0, 0.01)
        nn.init.constant_(self.bn.weight, 1)
        nn.init.constant_(self.bn.bias, 0)

    def get_sim_matrix(self, v_feats):
        sq = v_feats.pow(2).sum(dim=2)
        d2 = sq.unsqueeze(1) + sq.unsqueeze(2) - 2 * torch.bmm(v_feats, v_feats.transpose(1, 2))
        return 2 / (d2.clamp(1e-12).sqrt().exp() + 1)

    def forward(self, input, adj):
        h = self.linear(input)
        n, v, _ = h.size()
        mask = 1 - torch.eye(v, dtype=h.dtype, device=h.device).unsqueeze(0)
        graph = None
        if self.use_pose:
            graph = F.normalize(mask * adj, p=1, dim=2)
        if self.learn_graph:
            learned = F.normalize(mask * self.get_sim_matrix(input), p=1, dim=2)
            graph = learned if graph is None else (graph + learned) / 2
        msg = torch.bmm(graph, h)
        msg = self.relu(self.bn(msg.view(n * v, -1)).view(n, v, -1))
        return input + self.gamma * msg


class GANet(nn.Module):
    """The reference names this class GSTA as well (ganet.py:285)."""

    def __init__(self, num_classes, loss, block, layers, num_split, pyramid_part, num_gb, use_pose, learn_graph,
                 consistent_loss, nonlinear='relu', **kwargs):
        super(GANet, self).__init__()
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
        self.pam_layer = PAM_Module(self.feature_dim)
        self.cam_layer = CAM_Module(self.feature_dim)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.num_gb = num_gb
        self.graph_layers = nn.ModuleList(
            MaskedGraphLayer(in_features=self.feature_dim, out_features=self.feature_dim, use_pose=use_pose, learn_graph=learn_graph)
            for _ in range(num_gb))
        self.consistent_loss = consistent_loss
        self.bottleneck = nn.BatchNorm1d((num_gb + 1) * self.feature_dim)
        self.bottleneck.bias.requires_grad_(False)
        self.classifier = nn.Linear((num_gb + 1) * self.feature_dim, num_classes, bias=False)
        self._init_params()

        # MI355X path configuration (not part of the state dict)
        self.hip_precision = os.environ.get('AGRL_HIP_PRECISION', 'fp32')
        from torchreid import hip_ops as _ops   # a precision the loaded library cannot serve fails HERE, not at the first forward
        _ops.check_precision(self.hip_precision)
        self.hip_static_weights = False
        self.hip_fuse_tail = os.environ.get('AGRL_HIP_FUSE_TAIL', '1') != '0'
        self._hip_packs = {}

    def _init_params(self):
        """reference ganet.py:352-366"""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def featuremaps(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))

    def _attention_op(self, feat):
        att = F.normalize(feat.norm(p=2, dim=3, keepdim=True), p=1, dim=1)
        return (feat * att).sum(dim=1)

    def forward(self, x, adj, *args):
        if x.is_cuda and not self.training:
            from torchreid.models._ganet_hip import hip_forward_ganet
            return hip_forward_ganet(self, x, adj)
        B, S, C, H, W = x.size()
        fm = self.featuremaps(x.view(B * S, C, H, W))
        _, c, h, w = fm.shape
        nodes = []
        for n in self.total_split_list:
            step = h // n
            for i in range(n):
                piece = fm[:, :, step * i: step * (i + 1)]
                pam_f, _ = self.pam_layer(piece)
                nodes.append(self.avgpool(pam_f + piece).view(B * S, c))
        f = torch.stack(nodes, dim=2).transpose(1, 2).contiguous().view(B, S * self.total_split, c)
        outs = [f]
        for layer in self.graph_layers:
            outs.append(layer(outs[-1], adj))
        f = torch.cat(outs, dim=2).view(B, S, self.total_split, (self.num_gb + 1) * c)
        f_g = self._attention_op(f).mean(dim=1).view(B, -1)
        bn = self.bottleneck(f_g)
        if self.consistent_loss and self.training:
            # one random frame dropped per tracklet, drawn from numpy's global RNG (reference ganet.py:413-424)
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


def ganet(num_classes, loss, last_stride, num_split, num_gb, num_scale, knn, pyramid_part, use_pose, learn_graph,
          pretrained=True, consistent_loss=False, **kwargs):
    """Factory registered as ``'ganet'`` (reference ganet.py:458-477; ``knn`` is required and unused there too). Never
    touches the network: ``pretrained`` only takes effect through ``AGRL_PRETRAINED_RESNET50`` (a local
    resnet50-19c8e357.pth)."""
    model = GANet(num_classes=num_classes, loss=loss, block=Bottleneck, layers=list(RESNET50_STAGES), last_stride=last_stride,
                  num_split=num_split, pyramid_part=pyramid_part, num_gb=num_gb, use_pose=use_pose, learn_graph=learn_graph,
                  consistent_loss=consistent_loss, nonlinear='relu', **kwargs)
    path = os.environ.get('AGRL_PRETRAINED_RESNET50', '')
    if pretrained and path and os.path.isfile(path):
        own = model.state_dict()
        picked = {k: v for k, v in torch.load(path, map_location='cpu').items() if k in own and own[k].size() == v.size()}
        own.update(picked)
        model.load_state_dict(own)
    return model
