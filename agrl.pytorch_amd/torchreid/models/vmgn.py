"""VMGN (``vmgn``): two-branch ResNet50 + pose-guided adaptive graph layers + attention temporal pooling.

Drop-in for ``torchreid/models/vmgn.py`` of weleen/AGRL.pytorch: same factory signature
(reference vmgn.py:373-391), same module tree / 402 state-dict keys (reference vmgn.py:215-268), same
call contract ``model(x:(B,S,3,H,W), adj:(B,V,V))`` and return conventions (reference vmgn.py:292-357).

Two execution paths, chosen by where the input lives:

* CUDA tensors, ``model.eval()``  -> the MI355X path: ``_vmgn_hip.hip_forward`` drives the gfx950 kernels of
  libagrl_hip.so (stem, implicit-GEMM convs, part pooling, graph layers, attention tail). There is no
  fallback: a missing library raises.
* CUDA tensors, ``model.train()`` -> the WHOLE train forward and backward -- conv trunk with batch-statistics BatchNorm,
  pooling, graph layers, attention pooling, BNNecks, classifiers -- runs on the gfx950 kernels through ``_train_hip``
  (autograd Functions whose forward and backward are C-ABI calls). ``hip_train_tail = False`` keeps only the trunk
  native, ``hip_train = False`` selects the stock-torch module tree below (bench.py's baseline step).
* CPU tensors (the reference's own CPU-runnable configuration, also what ``compute_model_complexity`` runs
  at start-up) -> the module tree below, evaluated by stock ``torch.nn`` leaf modules so forward hooks, autograd
  and ``nn.DataParallel`` replication behave exactly as they do for the reference.
"""
from __future__ import absolute_import
from __future__ import division

__all__ = ['vmgn']

import copy
import os

import torch
from torch import nn
from torch.nn import functional as F

from torchreid.utils.reidtools import calc_splits
from torchreid.utils.torchtools import weights_init_kaiming, weights_init_classifier

FEATURE_DIM = 2048
RESNET50_STAGES = (3, 4, 6, 3)


class Bottleneck(nn.Module):
    """1x1 -> 3x3(stride) -> 1x1 residual unit (reference vmgn.py:29-65); stride sits on the 3x3."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(Bottleneck, self).__init__()
        width_out = planes * self.expansion
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, width_out, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(width_out)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        y += shortcut
        return self.relu(y)


def _make_stage(inplanes, planes, blocks, stride):
    """One ResNet stage as nn.Sequential of Bottlenecks; returns (stage, out_channels)."""
    out_ch = planes * Bottleneck.expansion
    downsample = None
    if stride != 1 or inplanes != out_ch:
        downsample = nn.Sequential(
            nn.Conv2d(inplanes, out_ch, kernel_size=1, stride=stride, bias=False),
            nn.BatchNorm2d(out_ch),
        )
    units = [Bottleneck(inplanes, planes, stride, downsample)]
    units.extend(Bottleneck(out_ch, planes) for _ in range(1, blocks))
    return nn.Sequential(*units), out_ch


class GraphLayer(nn.Module):
    """Residual graph block over part nodes (reference vmgn.py:68-172).

    ``out = (1-gamma) * f + gamma * LeakyReLU(BN(G @ (f W^T)))`` with
    ``G = (rowL1(adj) + rowL1(sim(f))) / 2``, ``sim = 2 / (exp(||f_i - f_j||) + 1)``.
    """

    def __init__(self, in_features, out_features, learn_graph=True, use_pose=True,
                 dist_method='l2', gamma=0.1, k=4, **kwargs):
        super(GraphLayer, self).__init__()
        assert use_pose or learn_graph
        if dist_method != 'l2':
            # the reference also carries an unused 'dot' variant (vmgn.py:104-107,110-113); vmgn never selects it
            raise NotImplementedError("dist_method={!r}: only 'l2' is on the vmgn path".format(dist_method))
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
        """(b, V, C) -> (b, V, V) similarity from pairwise L2 distances (diagonal not masked)."""
        sq = v_feats.pow(2).sum(dim=2)
        d2 = sq.unsqueeze(1) + sq.unsqueeze(2)
        d2 = d2 - 2 * torch.bmm(v_feats, v_feats.transpose(1, 2))
        dist = d2.clamp(1e-12).sqrt()
        return 2 / (dist.exp() + 1)

    def forward(self, input, adj):
        h = self.linear(input)
        n, v, _ = h.size()
        graph = None
        if self.use_pose:
            graph = F.normalize(adj, p=1, dim=2)
        if self.learn_graph:
            learned = F.normalize(self.get_sim_matrix(input), p=1, dim=2)
            graph = learned if graph is None else (graph + learned) / 2
        msg = torch.bmm(graph, h)
        msg = self.relu(self.bn(msg.view(n * v, -1)).view(n, v, -1))
        return (1 - self.gamma) * input + self.gamma * msg


class GSTA(nn.Module):
    """The VMGN network (the reference names the class GSTA, vmgn.py:214)."""

    def __init__(self, num_classes, loss, block, layers, num_split, pyramid_part, num_gb, use_pose,
                 learn_graph, consistent_loss, nonlinear='relu', **kwargs):
        super(GSTA, self).__init__()
        assert block is Bottleneck
        self.loss = loss
        self.feature_dim = 512 * block.expansion

        # ---- backbone: stem + layer1..3 shared, layer4 duplicated into two branches, last stride 1
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        ch = 64
        self.layer1, ch = _make_stage(ch, 64, layers[0], 1)
        self.layer2, ch = _make_stage(ch, 128, layers[1], 2)
        self.layer3, ch = _make_stage(ch, 256, layers[2], 2)
        self.layer4_1, ch = _make_stage(ch, 512, layers[3], 1)
        _load_local_pretrained(self)
        self.layer4_2 = copy.deepcopy(self.layer4_1)

        # ---- global branch (from layer4_1)
        self.global_avg_pool = nn.AdaptiveAvgPool3d(1)
        self.global_bottleneck = nn.BatchNorm1d(self.feature_dim)
        self.global_bottleneck.bias.requires_grad_(False)
        self.global_classifier = nn.Linear(self.feature_dim, num_classes, bias=False)
        weights_init_kaiming(self.global_bottleneck)
        weights_init_classifier(self.global_classifier)

        # ---- part / graph / attention branch (from layer4_2)
        self.num_split = num_split
        self.total_split_list = calc_splits(num_split) if pyramid_part else [num_split]
        self.total_split = sum(self.total_split_list)
        self.parts_avgpool = nn.ModuleList(nn.AdaptiveAvgPool2d((n, 1)) for n in self.total_split_list)
        self.num_gb = num_gb
        self.graph_layers = nn.ModuleList(
            GraphLayer(in_features=self.feature_dim, out_features=self.feature_dim,
                       use_pose=use_pose, learn_graph=learn_graph)
            for _ in range(num_gb))
        self.consistent_loss = consistent_loss
        self.att_bottleneck = nn.BatchNorm1d(self.feature_dim)
        self.att_bottleneck.bias.requires_grad_(False)
        self.att_classifier = nn.Linear(self.feature_dim, num_classes, bias=False)
        weights_init_kaiming(self.att_bottleneck)
        weights_init_classifier(self.att_classifier)

        # MI355X path configuration (not part of the state dict)
        self.hip_precision = os.environ.get('AGRL_HIP_PRECISION', 'fp32')
        from torchreid import hip_ops as _ops   # a precision the loaded library cannot serve fails HERE, not at the first forward
        _ops.check_precision(self.hip_precision)
        self.hip_static_weights = False
        self.hip_fuse_pool = os.environ.get('AGRL_HIP_FUSE_POOL', '1') != '0'
        self.hip_fuse_tail = os.environ.get('AGRL_HIP_FUSE_TAIL', '1') != '0'
        self.hip_train = os.environ.get('AGRL_HIP_TRAIN', '1') != '0'   # train-mode conv trunk (fwd + bwd) on the HIP kernels
        self.hip_train_precision = os.environ.get('AGRL_HIP_TRAIN_PRECISION', 'fp32')   # 'fp32' exact | 'bf16x3' split-bf16 MFMA
        self.hip_train_tail = os.environ.get('AGRL_HIP_TRAIN_TAIL', '1') != '0'         # tail of the train forward native as well
        self._hip_packs = {}

    # ------------------------------------------------------------------ stock-torch path (CPU / train)
    def _attention_op(self, feat):
        """(b, S, P, c) -> (b, P, c): frames weighted by their L2 norm, L1-normalised over S."""
        att = F.normalize(feat.norm(p=2, dim=3, keepdim=True), p=1, dim=1)
        return (feat * att).sum(dim=1)

    def featuremaps(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer3(self.layer2(self.layer1(x)))
        return self.layer4_1(x), self.layer4_2(x)

    def _pooled_nodes(self, x4_1, x4_2, B, S):
        _, c, h, w = x4_1.shape
        g_f = self.global_avg_pool(x4_1.view(B, S, c, h, w).transpose(1, 2).contiguous()).view(B, -1)
        parts = [pool(x4_2).view(B, S, c, n) for pool, n in zip(self.parts_avgpool, self.total_split_list)]
        f = torch.cat(parts, dim=3).transpose(2, 3).contiguous().view(B, S * self.total_split, c)
        return g_f, f

    def forward(self, x, adj, *args):
        if x.is_cuda and not self.training:
            from torchreid.models._vmgn_hip import hip_forward
            return hip_forward(self, x, adj)

        B, S, C, H, W = x.size()
        if x.is_cuda and self.training and self.hip_train and x.dtype != torch.float32:
            raise TypeError('the native train step takes float32 frames (the reference trains in fp32), got {}; '
                            'set model.hip_train = False for the stock-torch module tree'.format(x.dtype))
        if x.is_cuda and self.training and self.hip_train and self.hip_train_tail:
            # the whole train forward -- trunk, pooling, graph layers, attention pooling, BNNecks, classifiers -- as autograd
            # Functions over C-ABI calls (models/_train_hip.py); the losses have native forms too (torchreid/losses)
            from torchreid.models._train_hip import forward_train
            out_list, f_list = forward_train(self, x, adj)
            if self.loss == {'xent'}:
                return out_list
            elif self.loss == {'xent', 'htri'}:
                return out_list, f_list
            raise KeyError('Unsupported loss: {}'.format(self.loss))
        if x.is_cuda and self.training and self.hip_train:
            # train step on the GPU: the conv trunk -- forward with batch-statistics BatchNorm and its whole backward -- on the
            # gfx950 kernels (models/_train_hip.py); the graph stays torch.autograd's
            from torchreid.models._train_hip import featuremaps_train
            x4_1, x4_2 = [t.contiguous() for t in featuremaps_train(self, x.view(B * S, C, H, W))]
        else:
            x4_1, x4_2 = self.featuremaps(x.view(B * S, C, H, W))
        c = x4_1.size(1)
        g_f, f = self._pooled_nodes(x4_1, x4_2, B, S)
        g_bn = self.global_bottleneck(g_f)
        for layer in self.graph_layers:
            f = layer(f, adj)
        f = f.view(B, S, self.total_split, c)
        att_f = self._attention_op(f).mean(dim=1).view(B, -1)
        att_bn = self.att_bottleneck(att_f)

        if not self.training:
            return torch.cat([g_bn, att_bn], dim=1)

        out_list = [self.global_classifier(g_bn), self.att_classifier(att_bn)]
        f_list = [g_f, att_f]
        if self.consistent_loss:
            # three random frame subsets share the attention head (reference vmgn.py:327-342)
            assert S >= 5
            for num_frame in [S - 3, S - 2, S - 1]:
                pick = torch.sort(torch.randperm(S)[:num_frame])[0].long().to(f.device)
                sub = torch.gather(f, dim=1, index=pick.view(1, num_frame, 1, 1).repeat(B, 1, self.total_split, c))
                satt_f = self._attention_op(sub).mean(dim=1).view(B, -1)
                out_list.append(self.att_classifier(self.att_bottleneck(satt_f)))
                f_list.append(satt_f)

        if self.loss == {'xent'}:
            return out_list
        elif self.loss == {'xent', 'htri'}:
            return out_list, f_list
        raise KeyError('Unsupported loss: {}'.format(self.loss))

    # ------------------------------------------------------------------ MI355X path helpers
    def invalidate_hip_cache(self):
        """Drop the packed (BN-folded, OHWI) weights so the next CUDA eval forward rebuilds them."""
        self._hip_packs.clear()


def _load_local_pretrained(model):
    """ImageNet initialisation without the network.

    The reference downloads resnet50-19c8e357.pth in the constructor (vmgn.py:225, :360-370). This build
    never touches the network: if ``AGRL_PRETRAINED_RESNET50`` names a local copy of that file, the
    name-and-shape-matching tensors are loaded (``layer4.*`` feeding ``layer4_1``); otherwise the random
    initialisation is kept and a checkpoint is expected through ``--load-weights``.
    """
    path = os.environ.get('AGRL_PRETRAINED_RESNET50', '')
    if not path or not os.path.isfile(path):
        return False
    pretrained = torch.load(path, map_location='cpu')
    own = model.state_dict()
    picked = {}
    for key, value in pretrained.items():
        target = key.replace('layer4.', 'layer4_1.', 1) if key.startswith('layer4.') else key
        if target in own and own[target].size() == value.size():
            picked[target] = value
    own.update(picked)
    model.load_state_dict(own)
    print('Initialized model with pretrained weights from {}'.format(path))
    return True


def vmgn(num_classes, loss, last_stride, num_split, num_gb, num_scale,
         pyramid_part, use_pose, learn_graph, consistent_loss=False, **kwargs):
    """Factory registered as ``'vmgn'``. ``last_stride``/``num_scale`` and any extra driver kwargs
    (``num_parts``, ``bnneck``, ``save_dir``) are accepted and ignored, as in the reference."""
    return GSTA(
        num_classes=num_classes,
        loss=loss,
        block=Bottleneck,
        layers=list(RESNET50_STAGES),
        last_stride=last_stride,
        num_split=num_split,
        pyramid_part=pyramid_part,
        num_gb=num_gb,
        use_pose=use_pose,
        learn_graph=learn_graph,
        consistent_loss=consistent_loss,
        nonlinear='relu',
        **kwargs
    )
