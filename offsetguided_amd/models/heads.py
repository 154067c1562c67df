"""Head networks on the stride-4 hourglass features (reference models/heads.py:10-223):
1x1 convolutions, no activation.  Class attributes are the global configuration switches,
set by `factory_head` exactly like the reference (heads.py:186-223)."""
import re

import torch


def _convs(flag, inp_dim, out_ch, n_stacks, k, padding, dilation):
    return torch.nn.ModuleList([
        torch.nn.Conv2d(inp_dim, out_ch, k, padding=padding, dilation=dilation) if flag else torch.nn.Sequential()
        for _ in range(n_stacks)])


class HeatMapsHead(torch.nn.Module):
    """Keypoint heatmaps: returns (hmps[S], background[S], jitter offsets[S]); disabled parts are []."""
    stride = 4
    n_keypoints = 17
    include_background = False
    bg_channel = 1
    include_jitter_offset = False
    jo_channel = 2
    include_spread = False

    def __init__(self, head_name, inp_dim, n_stacks, kernel_size=1, padding=0, dilation=1):
        super().__init__()
        self.head_name = head_name
        self.n_stacks = n_stacks
        a = (inp_dim,), (n_stacks, kernel_size, padding, dilation)
        self.hp_convs = _convs(True, *a[0], self.n_keypoints, *a[1])
        self.bghp_convs = _convs(self.include_background, *a[0], self.bg_channel, *a[1])
        self.jitter_convs = _convs(self.include_jitter_offset, *a[0], self.jo_channel, *a[1])

    def forward(self, args):
        assert len(args) == self.n_stacks, 'multiple outputs from BaseNet'
        hm = [conv(x) for conv, x in zip(self.hp_convs, args)]
        bg = [conv(x) if self.include_background else [] for conv, x in zip(self.bghp_convs, args)]
        jo = [conv(x) if self.include_jitter_offset else [] for conv, x in zip(self.jitter_convs, args)]
        return hm, bg, jo


class OffsetMapsHead(torch.nn.Module):
    """Guiding offsets, channel order x0,y0,x1,y1,...: returns (offsets[S], spreads[S], scales[S])."""
    stride = 4
    n_keypoints = HeatMapsHead.n_keypoints
    n_skeleton = 19
    include_spread = False
    include_scale = False

    def __init__(self, head_name, inp_dim, n_stacks, kernel_size=1, padding=0, dilation=1):
        super().__init__()
        self.head_name = head_name
        self.n_stacks = n_stacks
        a = (inp_dim,), (n_stacks, kernel_size, padding, dilation)
        self.reg_convs = _convs(True, *a[0], 2 * self.n_skeleton, *a[1])
        self.spread_convs = _convs(self.include_spread, *a[0], self.n_skeleton, *a[1])
        self.scale_convs = _convs(self.include_scale, *a[0], self.n_keypoints, *a[1])

    def forward(self, args):
        assert len(args) == self.n_stacks, 'multiple outputs from BaseNet'
        off = [conv(x) for conv, x in zip(self.reg_convs, args)]
        spread = [conv(x) if self.include_spread else [] for conv, x in zip(self.spread_convs, args)]
        scale = [conv(x) if self.include_scale else [] for conv, x in zip(self.scale_convs, args)]
        return off, spread, scale


_OMP_LIMBS = {'omp': 19, 'omps': 19, 'offset': 19, 'offsets': 19, 'omp19': 19, 'omp16': 16, 'omp31': 31,
              'omp44': 44, 'omp25': 25, 'omps25': 25}


def factory_head(head_name, inp_dim, n_stacks, stride, include_spread, include_background,
                 include_jitter_offset, include_scale):
    """Configure (class attributes) and build one head (reference heads.py:175-223)."""
    if head_name in ('hmp', 'hmps', 'heatmap', 'heatmaps') or re.match('hmp[s]?([0-9]+)$', head_name):
        m = re.match('hmp[s]?([0-9]+)$', head_name)
        HeatMapsHead.n_keypoints = int(m.group(1)) if m else 17
        HeatMapsHead.stride = stride
        HeatMapsHead.include_spread = include_spread
        HeatMapsHead.include_background = include_background
        HeatMapsHead.include_jitter_offset = include_jitter_offset
        return HeatMapsHead(head_name, inp_dim, n_stacks)
    if head_name in _OMP_LIMBS or re.match('omp[s]?([0-9]+)$', head_name):
        m = re.match('omp[s]?([0-9]+)$', head_name)
        OffsetMapsHead.n_skeleton = int(m.group(1)) if m else _OMP_LIMBS[head_name]
        OffsetMapsHead.stride = stride
        OffsetMapsHead.include_spread = include_spread
        OffsetMapsHead.include_scale = include_scale
        return OffsetMapsHead(head_name, inp_dim, n_stacks)
    raise Exception('unknown head to create a head network: {}'.format(head_name))


def headnets_factory(headnames, n_stacks, strides, inp_dim, include_spread, include_background,
                     include_jitter_offset, include_scale):
    return [factory_head(h, inp_dim, n_stacks, s, include_spread, include_background, include_jitter_offset,
                         include_scale) for h, s in zip(headnames, strides)]
