#!/usr/bin/env python3
"""tests/golden/resize_cv2.npz: OpenCV's own answers for the two cv2 calls on the §8f paths --
`cv2.resize(image, (w', h'), interpolation=cv2.INTER_CUBIC)` of RescaleLongAbsolute (transforms/scale.py:14-31,75-98) and
the mask shrink of the GT encoders (`cv2.resize(mask_miss, (0, 0), fx=1/4, fy=1/4, INTER_CUBIC) / 255 > 0.7`,
encoder/heatmap.py:56-60).

cv2 is not installed in the build container or on the GPU boxes of this pool, so f2 / the mask shrink are pinned only to
our restatement of OpenCV's published 8-bit algorithm ("parity unpinned").  Run this script on ANY
machine that has `opencv-python` (the reference pins 3.4.5.20) and commit the file it writes: tests/test_transforms.py
then checks the oracle (CPU) and the HIP kernels (GPU) against it; while the file is absent those tests are skipped.

Inputs come from the portable counter-based generator (offsetguided_amd/synth.py:HashRng) and are not stored.  Stored:
for the seven image shapes of tests/test_transforms.py::test_resize_and_fused_chain_bit_exact_vs_restatement the sha256 of
cv2's output plus its top-left 48x48 corner; for seven small shapes (same aspect ratios) the whole output; the three
128x192 mask cases' shrunk masks.

usage: python tools/gen_golden_cv2.py        (exits 2 with a message when cv2 is missing)
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from offsetguided_amd.synth import HashRng  # noqa: E402

CASES = [((427, 640), 640), ((640, 480), 640), ((375, 500), 640), ((1000, 333), 640), ((31, 17), 128), ((640, 640), 640),
         ((240, 320), 512)]
SMALL = [((43, 64), 96), ((64, 48), 96), ((37, 50), 80), ((100, 33), 64), ((31, 17), 128), ((64, 64), 64), ((24, 32), 51)]


def image(seed, h, w):
    """(h, w, 3) uint8, smooth gradients + noise so that the cubic taps both interpolate and overshoot."""
    rng = HashRng(seed)
    noise = rng.integers(h * w * 3, 0, 255).reshape(h, w, 3)
    yy, xx = np.mgrid[:h, :w]
    ramp = ((yy * 3 + xx * 5) % 256)[:, :, None]
    hard = (((yy // 7 + xx // 5) % 2) * 255)[:, :, None]
    sel = rng.integers(h * w, 0, 2).reshape(h, w, 1)
    return np.where(sel == 0, noise, np.where(sel == 1, ramp, hard)).astype(np.uint8)


def masks():
    m = np.full((3, 128, 192), 255, np.uint8)
    m[0, 20:60, 30:100] = 0
    yy, xx = np.mgrid[:128, :192]
    m[1][(yy - 64) ** 2 + (xx - 90) ** 2 < 40 ** 2] = 0
    m[2] = (HashRng(5).uniform(128 * 192).reshape(128, 192) > 0.3).astype(np.uint8) * 255
    return m


def target(h, w, long_edge):   # transforms/scale.py:91-97
    s = long_edge / max(h, w)
    return (int(w * s), long_edge) if h > w else (long_edge, int(h * s))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    try:
        import cv2
    except ImportError:
        print('gen_golden_cv2: cv2 (opencv-python) is not installed here; run this on a machine that has it', file=sys.stderr)
        return 2
    out = {'cv2_version': np.array(cv2.__version__)}
    for i, ((h, w), T) in enumerate(CASES):
        tw, th = target(h, w, T)
        r = cv2.resize(image(100 + i, h, w), (tw, th), interpolation=cv2.INTER_CUBIC)
        out[f'big{i}_sha'], out[f'big{i}_corner'] = np.array(sha(r)), r[:48, :48].copy()
    for i, ((h, w), T) in enumerate(SMALL):
        tw, th = target(h, w, T)
        out[f'small{i}'] = cv2.resize(image(200 + i, h, w), (tw, th), interpolation=cv2.INTER_CUBIC)
    m = masks()
    out['masks'] = np.stack([cv2.resize(m[i], (0, 0), fx=0.25, fy=0.25, interpolation=cv2.INTER_CUBIC).astype(np.float32) / 255 > 0.7
                             for i in range(3)])
    path = os.path.join(ROOT, 'tests', 'golden', 'resize_cv2.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, 'with cv2', cv2.__version__)
    return 0


if __name__ == '__main__':
    sys.exit(main())
