"""CPU leg of bench.py's cpu_baseline -- TEST/BASELINE INFRASTRUCTURE ONLY (never the product path).

Times what the reference would do on host cores for ONE image of the benchmark workload:
the eager fp32 PyTorch forward of Hourglass-104 + heads (models/networks.py:189-194) on all
cores, then the decoder (decoder/factory.py:52-96) through the C oracle (single thread)."""
import time

import numpy as np
import torch

from . import oracle as _o


def time_end_to_end(model, hm_lr, off_lr, skeleton, size=640, flags=None, budget_s=25.0):
    """-> dict(backbone_s_per_img, decode_s_per_img, cores, n_decode) on a bounded sample."""
    flags = flags or {}
    model = model.to('cpu').float().eval()
    x = torch.randn(1, 3, size, size)
    t0 = time.perf_counter()
    with torch.no_grad():
        model(x)
    t_backbone = time.perf_counter() - t0
    n_dec, t_dec = 0, 0.0
    for i in range(hm_lr.shape[0]):
        t0 = time.perf_counter()
        _o.decode(hm_lr[i:i + 1], off_lr[i:i + 1], skeleton, **flags)
        t_dec += time.perf_counter() - t0
        n_dec += 1
        if t_backbone + t_dec > budget_s:
            break
    return dict(backbone_s_per_img=t_backbone, decode_s_per_img=t_dec / n_dec, n_decode=n_dec,
                cores=torch.get_num_threads())
