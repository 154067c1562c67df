#!/usr/bin/env python3
"""Golden values for the training losses from the imported reference models/losses.py (build
container only).  Inputs come from the portable generator; expected values/gradients are stored."""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
from offsetguided_amd import synth  # noqa: E402
from offsetguided_amd.models import losses as mine  # noqa: E402

spec = importlib.util.spec_from_file_location('ref_losses', '/root/reference/models/losses.py')
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def inputs(seed=5, n=2, h=32, w=32):
    rng = synth.HashRng(seed)
    t = lambda c, lo, hi: torch.from_numpy(rng.uniform(n * c * h * w, lo, hi).reshape(n, c, h, w).astype(np.float32))  # noqa: E731
    hm_gt = t(17, 0, 1) * (t(17, 0, 1) > 0.8)                # sparse targets: most are background (< tau)
    hm_pred = [t(17, -0.2, 1.1), t(17, -0.2, 1.1)]
    off_gt = t(38, -60, 60)
    off_gt[t(38, 0, 1) > 0.3] = float('inf')                 # offsets are defined in patches only
    off_pred = [t(38, -60, 60), t(38, -60, 60)]
    ps = t(1, 20, 300)
    mask = t(1, 0, 1) > 0.15
    return hm_pred, hm_gt, off_pred, off_gt, ps, mask


def run(mod, hm_pred, hm_gt, off_pred, off_gt, ps, mask, sqrt_re, off_name):
    hm_pred = [p.clone().requires_grad_(True) for p in hm_pred]
    off_pred = [p.clone().requires_grad_(True) for p in off_pred]
    hl = mod.HeatMapsLoss('hmp', 2, [1, 3], mod.LossChoice.focal_l2_loss, mod.LossChoice.offset_l1_loss, sqrt_re)
    ol = mod.OffsetMapsLoss('omp', 2, [1, 3], getattr(mod.LossChoice, off_name), mod.LossChoice.scale_l1_loss, sqrt_re)
    l1 = hl((hm_pred, [[], []], [[], []]), hm_gt, None, None, mask)
    l2 = ol((off_pred, [[], []], [[], []]), off_gt, None, ps, mask)
    total = l1[0] * 1.0 + l2[0] * 100.0
    total.backward()
    return (float(l1[0]), float(l2[0]), [p.grad.numpy() for p in hm_pred], [p.grad.numpy() for p in off_pred])


def main():
    out = {}
    inp = inputs()
    for sqrt_re in (False, True):
        for off_name in ('offset_l1_loss', 'offset_instance_l1_loss'):
            r = run(ref, *inp, sqrt_re, off_name)
            m = run(mine, *inp, sqrt_re, off_name)
            assert r[0] == m[0] and r[1] == m[1], (r[:2], m[:2])
            for a, b in zip(r[2] + r[3], m[2] + m[3]):
                assert np.array_equal(a, b)
            tag = f'{off_name}_{int(sqrt_re)}'
            out[tag + '_hm'] = np.float32(r[0])
            out[tag + '_off'] = np.float32(r[1])
            out[tag + '_ghm'] = np.stack(r[2])[:, :, :3, ::4, ::4]          # a slice of the gradients
            out[tag + '_goff'] = np.stack(r[3])[:, :, :3, ::4, ::4]
            print(tag, r[0], r[1])
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'losses.npz'), **out)
    print('losses: torch formulation bit-identical to the reference on CPU; fixture written')


if __name__ == '__main__':
    main()
