"""``torchreid.utils.torchtools``: the weight initialisers the model constructor applies (reference
torchreid/utils/torchtools.py:51-88) and the small training helpers the driver imports from this module
(``set_wd``, ``cur_time``: train_vidreid_xent_htri.py:24; reference torchtools.py:10-49).

This file shadows the reference's module of the same name under the PYTHONPATH overlay (INTEGRATION.md route A),
so it carries every public name of that module, not only the ones the model needs. Modules are matched by
class-name substring like the reference so that any module type it would have touched is touched here too.
"""
from __future__ import absolute_import
from __future__ import division

import gc
import time

import torch
from torch import nn


def cur_time():
    """Local wall-clock time as 'YYYY-mm-dd HH:MM:SS' (the driver's log prefix)."""
    return time.strftime('%Y-%m-%d %H:%M:%S', time.localtime())


def adjust_learning_rate(optimizer, base_lr, epoch, stepsize, gamma=0.1):
    """Step decay: lr = base_lr * gamma ** (epoch // stepsize) on every parameter group."""
    lr = base_lr * gamma ** (epoch // stepsize)
    for group in optimizer.param_groups:
        group['lr'] = lr


def set_bn_to_eval(m):
    """``model.apply`` hook: freeze the running statistics of every BatchNorm (affine parameters stay trainable)."""
    if 'BatchNorm' in type(m).__name__:
        m.eval()


def set_wd(optim, num):
    """Set the weight decay of every parameter group (the driver switches it off after --fixbase, :346)."""
    assert isinstance(num, (int, float)), '{} is not int or float'.format(num)
    for group in optim.param_groups:
        if group['weight_decay'] != num:
            group['weight_decay'] = num


def count_num_param(model):
    """Parameter count in millions, without a ``classifier`` head (unused at test time)."""
    total = sum(p.numel() for p in model.parameters())
    head = getattr(model, 'classifier', None)
    if isinstance(head, nn.Module):
        total -= sum(p.numel() for p in head.parameters())
    return total / 1e6


def flip_tensor(x, dim):
    """Reverse ``x`` along ``dim``."""
    return x.flip(dim)


def _kind(m):
    name = type(m).__name__
    for key in ('Linear', 'Conv', 'BatchNorm'):
        if key in name:
            return key
    return None


def weights_init_kaiming(m):
    kind = _kind(m)
    if kind == 'Linear':
        nn.init.kaiming_normal_(m.weight, a=0, mode='fan_out')
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif kind == 'Conv':
        nn.init.kaiming_normal_(m.weight, a=0, mode='fan_in')
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif kind == 'BatchNorm' and m.affine:
        nn.init.normal_(m.weight, 1.0, 0.001)
        nn.init.constant_(m.bias, 0.0)


def weights_init_xavier(m):
    kind = _kind(m)
    if kind in ('Linear', 'Conv'):
        nn.init.xavier_normal_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif kind == 'BatchNorm' and m.affine:
        nn.init.normal_(m.weight, 1.0, 0.001)
        nn.init.constant_(m.bias, 0.0)


def weights_init_classifier(m):
    if _kind(m) == 'Linear':
        nn.init.normal_(m.weight.data, std=0.001)
        if m.bias is not None:
            nn.init.constant_(m.bias.data, 0.0)


def mem_report():
    """Census of the live tensors' storages on GPU and CPU (debug helper, never called by the driver)."""
    def census(tensors, where):
        seen, numel, mbytes = set(), 0, 0.0
        print('Storage on %s' % where)
        print('-' * 65)
        for t in tensors:
            if t.is_sparse:
                continue
            st = t.untyped_storage()
            if st.data_ptr() in seen:
                continue
            seen.add(st.data_ptr())
            n = st.nbytes() // max(1, t.element_size())
            mb = st.nbytes() / 1024 / 1024
            numel, mbytes = numel + n, mbytes + mb
            print('%s\t\t%s\t\t%.2f' % (type(t).__name__, tuple(t.size()), mb))
        print('-' * 65)
        print('Total Tensors: %d \tUsed Memory Space: %.2f MBytes' % (numel, mbytes))
        print('-' * 65)

    print('=' * 65)
    print('%s\t%s\t\t\t%s' % ('Element type', 'Size', 'Used MEM(MBytes)'))
    tensors = [o for o in gc.get_objects() if torch.is_tensor(o)]
    census([t for t in tensors if t.is_cuda], 'GPU')
    census([t for t in tensors if not t.is_cuda], 'CPU')
    print('=' * 65)
