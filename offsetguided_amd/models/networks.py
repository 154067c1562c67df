"""Model wrapper + checkpoint I/O (reference models/networks.py:12-225)."""
import logging
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from .hourglass_104 import Hourglass104

LOG = logging.getLogger(__name__)


class NetworkWrapper(torch.nn.Module):
    """basenet + headnets; forward returns one tuple of per-stack lists per head:
    [ (hmps[S], bg[S], jo[S]), (offs[S], spreads[S], scales[S]) ]  (networks.py:189-194)."""

    def __init__(self, basenet, headnets):
        super().__init__()
        self.basenet = basenet
        self.headnets = torch.nn.ModuleList(headnets)
        self.head_strides = [hn.stride for hn in headnets]
        self.head_names = [hn.head_name for hn in headnets]

    def forward(self, img_tensor):
        feats = self.basenet(img_tensor)
        return [hn(feats) for hn in self.headnets]


def basenet_factory(basenet_name):
    """-> (network, n_stacks, stride, max_stride, feature channels)."""
    assert basenet_name in ['hourglass104', 'hourglass4stage'], f'{basenet_name} is not implemented.'
    if basenet_name == 'hourglass104':
        return Hourglass104(None, 2), 2, 4, 128, 256
    raise Exception('unknown base network in {}'.format(basenet_name))  # hourglass4stage is dead code upstream


def load_model(model, ckpt_path, *, optimizer=None, drop_layers=True, drop_name='offset_convs',
               resume_optimizer=True, optimizer2cuda=True, load_amp=False):
    """Load `checkpoint['model_state_dict']` tolerant of a 'module.' prefix, missing / extra /
    shape-mismatched entries (reference networks.py:12-123).  Unlike the reference this never
    blocks on input(): a missing file raises FileNotFoundError.
    Returns (model, optimizer, start_epoch, start_loss, amp_state_or_False)."""
    if not os.path.isfile(ckpt_path):
        raise FileNotFoundError(f'checkpoint {ckpt_path} does not exist')
    ckpt = torch.load(ckpt_path, map_location='cpu')
    start_epoch, start_loss = ckpt['epoch'] + 1, ckpt['train_loss']
    amp_state = ckpt['amp'] if (load_amp and 'amp' in ckpt) else False
    own = model.state_dict()
    merged = OrderedDict()
    for key, value in ckpt['model_state_dict'].items():
        if drop_layers and drop_name in key:
            continue
        if key.startswith('module') and not key.startswith('module_list'):
            key = key[7:]
        if key in own and own[key].shape != value.shape:
            LOG.debug('shape mismatch for %s: keep the initialised tensor', key)
            value = own[key]
        merged[key] = value
    for key, value in own.items():
        merged.setdefault(key, value)
    model.load_state_dict(merged, strict=False)
    if optimizer is not None and resume_optimizer and 'optimizer_state_dict' in ckpt:
        optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        if torch.cuda.is_available() and optimizer2cuda:
            for state in optimizer.state.values():
                for k, v in state.items():
                    if torch.is_tensor(v):
                        state[k] = v.cuda()
    return model, optimizer, start_epoch, start_loss, amp_state


def save_model(path, epoch, train_loss, model, optimizer=None, amp_state=None):
    """Checkpoint dict layout of the reference (networks.py:126-144)."""
    net = model.module if hasattr(model, 'module') else model
    data = {'epoch': epoch, 'train_loss': train_loss, 'model_state_dict': net.state_dict()}
    if optimizer is not None:
        data['optimizer_state_dict'] = optimizer.state_dict()
    if amp_state is not None:
        data['amp'] = amp_state
    torch.save(data, path)


def initialize_weights(model):
    """conv ~ N(0, 0.001), zero biases, BN = identity (networks.py:147-173)."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data.normal_(0, 0.001)
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()
        elif isinstance(m, nn.Linear):
            m.weight.data.normal_(0, 0.01)
            m.bias.data.zero_()
    return model
