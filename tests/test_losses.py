"""Training losses: torch formulation vs the reference's values/gradients (golden, CPU) and the fused HIP
kernels vs the torch formulation (GPU, tolerance: reduction order differs)."""
import numpy as np
import pytest
import torch

from offsetguided_amd.models import losses
from helpers import GOLDEN



def _run(mod_losses, inp, sqrt_re, off_name, fused=False, device='cpu'):
    hm_pred, hm_gt, off_pred, off_gt, ps, mask = [([t.to(device) for t in x] if isinstance(x, list) else x.to(device)) for x in inp]
    hm_pred = [p.clone().requires_grad_(True) for p in hm_pred]
    off_pred = [p.clone().requires_grad_(True) for p in off_pred]
    hl = mod_losses.HeatMapsLoss('hmp', 2, [1, 3], mod_losses.LossChoice.focal_l2_loss, mod_losses.LossChoice.offset_l1_loss,
                                 sqrt_re, fused)
    ol = mod_losses.OffsetMapsLoss('omp', 2, [1, 3], getattr(mod_losses.LossChoice, off_name),
                                   mod_losses.LossChoice.scale_l1_loss, sqrt_re, fused)
    l1 = hl((hm_pred, [[], []], [[], []]), hm_gt, None, None, mask)
    l2 = ol((off_pred, [[], []], [[], []]), off_gt, None, ps, mask)
    (l1[0] * 1.0 + l2[0] * 100.0).backward()
    return float(l1[0]), float(l2[0]), [p.grad.cpu().numpy() for p in hm_pred], [p.grad.cpu().numpy() for p in off_pred]


def _load_inputs():
    import importlib.util
    import os
    from offsetguided_amd import synth
    rng = synth.HashRng(5)
    n, h, w = 2, 32, 32
    t = lambda c, lo, hi: torch.from_numpy(rng.uniform(n * c * h * w, lo, hi).reshape(n, c, h, w).astype(np.float32))  # noqa: E731
    hm_gt = t(17, 0, 1) * (t(17, 0, 1) > 0.8)
    hm_pred = [t(17, -0.2, 1.1), t(17, -0.2, 1.1)]
    off_gt = t(38, -60, 60)
    off_gt[t(38, 0, 1) > 0.3] = float('inf')
    off_pred = [t(38, -60, 60), t(38, -60, 60)]
    ps = t(1, 20, 300)
    mask = t(1, 0, 1) > 0.15
    return hm_pred, hm_gt, off_pred, off_gt, ps, mask


@pytest.mark.parametrize("sqrt_re", [False, True])
@pytest.mark.parametrize("off_name", ["offset_l1_loss", "offset_instance_l1_loss"])
def test_losses_match_reference_golden(sqrt_re, off_name):
    g = np.load(f"{GOLDEN}/losses.npz")
    tag = f"{off_name}_{int(sqrt_re)}"
    hm, off, ghm, goff = _run(losses, _load_inputs(), sqrt_re, off_name)
    assert np.float32(hm) == g[tag + "_hm"] and np.float32(off) == g[tag + "_off"]
    assert np.array_equal(np.stack(ghm)[:, :, :3, ::4, ::4], g[tag + "_ghm"])
    assert np.array_equal(np.stack(goff)[:, :, :3, ::4, ::4], g[tag + "_goff"])


@pytest.mark.gpu
@pytest.mark.parametrize("sqrt_re", [False, True])
@pytest.mark.parametrize("off_name", ["offset_l1_loss", "offset_instance_l1_loss"])
def test_fused_hip_losses_match_torch(sqrt_re, off_name):
    assert torch.cuda.is_available()
    inp = _load_inputs()
    ref = _run(losses, inp, sqrt_re, off_name, fused=False, device='cuda:0')
    got = _run(losses, inp, sqrt_re, off_name, fused=True, device='cuda:0')
    assert abs(got[0] - ref[0]) <= 1e-5 * abs(ref[0]) and abs(got[1] - ref[1]) <= 1e-5 * abs(ref[1])
    for a, b in zip(ref[2] + ref[3], got[2] + got[3]):
        assert np.allclose(a, b, rtol=1e-5, atol=1e-7)
