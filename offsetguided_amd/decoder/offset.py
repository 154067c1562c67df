"""Optional heatmap-weighted offset refinement (reference decoder/offset.py:8-43).

Off by default in the reference (`scored_off=False`, decoder/factory.py:52; evaluate.py never
enables it), so it is kept as a thin torch-op formulation on the stride-4 maps rather than a
dedicated kernel.  Differences from the reference: works for batch size 1 (the reference's
`.squeeze()` at :31 breaks there).
"""
import torch
import torch.nn.functional as F


def pack_jtypes(skeleton):
    return [a for a, _ in skeleton], [b for _, b in skeleton]


def scored_offset(hmp, off, jtypes_f, jtypes_t, kernel_size=7):
    """sum_box(hm * off) / (sum_box(hm) + 1e-6) per limb, hm = heatmap of the limb's start joint."""
    n, _, h, w = off.shape
    pad = (kernel_size - 1) // 2
    weight = hmp[:, jtypes_f]                                   # (n, L, h, w)
    pairs = off.view(n, -1, 2, h, w)
    num = F.avg_pool2d((weight.unsqueeze(2) * pairs).view(n, -1, h, w), kernel_size, stride=1, padding=pad,
                       divisor_override=1)
    den = F.avg_pool2d(weight, kernel_size, stride=1, padding=pad, divisor_override=1)
    return (num.view(n, -1, 2, h, w) / (den.unsqueeze(2) + 1e-6)).view(n, -1, h, w)
