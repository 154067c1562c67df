"""Hourglass-104 backbone (CornerNet/CenterNet variant used by OffsetGuided), written from the
architecture table -- state_dict-key compatible with the reference's models/hourglass_104.py
(`convolution` :16-30, `residual` :50-79, `kp_module` :132-190, `exkp` :193-298,
`Hourglass104` :307-321) so its checkpoints load unchanged.

    pre   : 7x7/2 conv-BN-ReLU 3->128, residual 128->256 /2                       -> stride 4
    stack : order-5 hourglass, channels 256,256,384,384,384,512, residuals per level 2,2,2,2,2,4;
            down-sampling by stride-2 residuals (no max-pool), nearest x2 up-sampling, additive merge
    out   : per stack a 3x3 conv-BN-ReLU 256->256 feature; stacks are bridged by
            ReLU(1x1conv-BN(inter) + 1x1conv-BN(feature)) -> residual

This eager module is the parameter container / training-shaped definition.  The MI355X
inference path is models/engine.py, which folds BN, runs bf16 channels-last and replays the
whole forward as one HIP graph.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

DIMS = (256, 256, 384, 384, 384, 512)
BLOCKS = (2, 2, 2, 2, 2, 4)


class ConvBlock(nn.Module):
    """k x k conv (+BN) + ReLU; parameters live under `.conv` / `.bn`."""

    def __init__(self, k, cin, cout, stride=1, with_bn=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=(k - 1) // 2, bias=not with_bn)
        self.bn = nn.BatchNorm2d(cout) if with_bn else nn.Sequential()

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


class Residual(nn.Module):
    """3x3(stride) conv-BN-ReLU, 3x3 conv-BN, plus identity or 1x1(stride) conv-BN shortcut, ReLU."""

    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        project = stride != 1 or cin != cout
        self.skip = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False),
                                  nn.BatchNorm2d(cout)) if project else nn.Sequential()

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)), inplace=True)
        y = self.bn2(self.conv2(y))
        return F.relu(y + self.skip(x), inplace=True)


def _residuals(cin, cout, count, first_stride=1, widen_last=False):
    """`count` residuals; the channel change happens in the first (default) or last block."""
    if widen_last:
        chans = [(cin, cin)] * (count - 1) + [(cin, cout)]
    else:
        chans = [(cin, cout)] + [(cout, cout)] * (count - 1)
    return nn.Sequential(*[Residual(a, b, first_stride if i == 0 else 1) for i, (a, b) in enumerate(chans)])


class HourglassLevel(nn.Module):
    """One recursion level: out = up1(x) + upsample2(low3(low2(low1(x))))."""

    def __init__(self, order, dims, blocks):
        super().__init__()
        cur, nxt = dims[0], dims[1]
        self.up1 = _residuals(cur, cur, blocks[0])
        self.low1 = _residuals(cur, nxt, blocks[0], first_stride=2)
        self.low2 = HourglassLevel(order - 1, dims[1:], blocks[1:]) if order > 1 \
            else _residuals(nxt, nxt, blocks[1])
        self.low3 = _residuals(nxt, cur, blocks[0], widen_last=True)

    def forward(self, x):
        low = self.low3(self.low2(self.low1(x)))
        return self.up1(x) + F.interpolate(low, scale_factor=2, mode='nearest')


class Hourglass104(nn.Module):
    """Returns the list of per-stack features [(N,256,H/4,W/4)] * num_stacks."""

    def __init__(self, heads=None, num_stacks=2):
        super().__init__()
        self.nstack = num_stacks
        dim = DIMS[0]
        self.pre = nn.Sequential(ConvBlock(7, 3, 128, stride=2), Residual(128, 256, stride=2))
        self.kps = nn.ModuleList([HourglassLevel(5, DIMS, BLOCKS) for _ in range(num_stacks)])
        self.cnvs = nn.ModuleList([ConvBlock(3, dim, 256) for _ in range(num_stacks)])
        self.inters = nn.ModuleList([Residual(dim, dim) for _ in range(num_stacks - 1)])

        def proj():
            return nn.Sequential(nn.Conv2d(dim, dim, 1, bias=False), nn.BatchNorm2d(dim))
        self.inters_ = nn.ModuleList([proj() for _ in range(num_stacks - 1)])
        self.cnvs_ = nn.ModuleList([proj() for _ in range(num_stacks - 1)])

    def forward(self, image):
        inter = self.pre(image)
        feats = []
        for s in range(self.nstack):
            feat = self.cnvs[s](self.kps[s](inter))
            feats.append(feat)
            if s < self.nstack - 1:
                inter = self.inters[s](F.relu(self.inters_[s](inter) + self.cnvs_[s](feat), inplace=True))
        return feats
