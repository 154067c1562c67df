#!/usr/bin/env python3
"""Generate tests/golden/eval_tail.npz and tests/golden/soft_nms.npz by running the IMPORTED reference
(transforms/preprocess.py:33-63 `Preprocess.annotations_inverse`, decoder/group.py:249-283 `soft_nms`).

Runs only in the build container (needs /root/reference).  Inputs come from the portable counter-based generator
(offsetguided_amd.synth.HashRng) and ARE stored beside the expected outputs: the tests read both from the fixture.
This package's implementations are checked against the reference right here.

usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_tail.py [out_dir]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("OG_REFERENCE", "/root/reference")
GOLD = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden")

from offsetguided_amd import evaluate, synth  # noqa: E402
from offsetguided_amd.decoder import soft_nms as my_soft_nms  # noqa: E402


def load_reference():
    sys.dont_write_bytecode = True
    import importlib.util
    sys.path.insert(0, REF)
    import decoder.group as G
    # the module file itself, not the `transforms` package: its __init__ imports torchvision and cv2, which this image lacks and
    # neither function uses
    spec = importlib.util.spec_from_file_location('_ref_preprocess', os.path.join(REF, 'transforms', 'preprocess.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Preprocess, G


def eval_tail(Preprocess):
    rng = synth.HashRng(9001)
    poses = np.zeros((5, 17, 6), np.float32)
    poses[..., 0] = rng.uniform(85, 0.0, 640.0).reshape(5, 17)
    poses[..., 1] = rng.uniform(85, 0.0, 480.0).reshape(5, 17)
    poses[..., 2] = rng.uniform(85, 0.0, 1.0).reshape(5, 17)
    poses[..., 3] = rng.uniform(85, 1.0, 300.0).reshape(5, 17)
    poses[..., 4] = rng.uniform(85, 0.0, 500.0).reshape(5, 17)
    poses[..., 5] = rng.uniform(85, 0.0, 500.0).reshape(5, 17)
    poses[1, 3] = 0.0                                               # a missing joint stays where the arithmetic puts it
    offset, scale = np.array([-12.0, 37.0]), np.array([1.3125, 1.3125])
    meta = {'offset': offset, 'scale': scale, 'hflip': False, 'image_id': 1}
    keep = poses.copy()
    expected = Preprocess.annotations_inverse(poses, meta)
    assert (poses == keep).all()
    mine = evaluate.annotations_inverse(poses, meta)
    assert mine.dtype == expected.dtype == np.float32 and (mine == expected).all(), 'annotations_inverse differs from the reference'
    np.savez_compressed(os.path.join(GOLD, 'eval_tail.npz'), poses=poses, offset=offset, scale=scale, expected=expected)
    print('eval_tail: annotations_inverse == reference on', poses.shape)


def soft_nms_cases(G):
    rng = synth.HashRng(9002)
    counts, pin, pout = [], [], []
    suppressed = 0
    for n in (0, 1, 2, 3, 3, 3, 5, 8):
        poses = []
        for p in range(n):
            a = np.zeros((17, 6), np.float32)
            # persons of one case share a small area: later ones land on cells the earlier ones occupied
            cx, cy = rng.uniform(1, 40.0, 160.0)[0], rng.uniform(1, 40.0, 160.0)[0]
            a[:, 0] = cx + rng.uniform(17, -25.0, 25.0)
            a[:, 1] = cy + rng.uniform(17, -25.0, 25.0)
            a[:, 2] = rng.uniform(17, 0.05, 1.0)
            a[:, 3] = rng.uniform(17, 1.0, 30.0)                    # joint scales: max(10, s) is the occupied half-width
            miss = rng.uniform(17) < 0.15
            a[miss, 2] = -1.0                                       # v == -1: the joint is skipped (decoder/group.py:264)
            poses.append(a)
        given = [q.copy() for q in poses]
        ref = G.soft_nms([q.copy() for q in poses], suppressed_v=0)
        mine = my_soft_nms([q.copy() for q in poses], suppressed_v=0)
        assert len(ref) == len(mine) == n and all(np.array_equal(r, m) for r, m in zip(ref, mine)), 'soft_nms differs from the reference'
        suppressed += sum(int((r[:, 2] != g[:, 2]).sum()) for r, g in zip(ref, given))
        counts.append(n)
        pin += given
        pout += list(ref)
    assert suppressed >= 10, suppressed
    np.savez_compressed(os.path.join(GOLD, 'soft_nms.npz'), counts=np.array(counts), poses_in=np.stack(pin), poses_out=np.stack(pout))
    print(f'soft_nms: {len(counts)} cases, {len(pin)} poses, {suppressed} joints suppressed, == reference')


def main():
    os.makedirs(GOLD, exist_ok=True)
    Preprocess, G = load_reference()
    eval_tail(Preprocess)
    soft_nms_cases(G)


if __name__ == '__main__':
    main()
