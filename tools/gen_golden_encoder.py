#!/usr/bin/env python3
"""Golden vectors for the ground-truth encoder from the imported reference (encoder/heatmap.py, encoder/offset.py;
build container only).  cv2 is stubbed: only HeatMapGenerator / OffsetMapGenerator (pure numpy) are called, not the
mask_miss resize of HeatMaps.__call__ / OffsetMaps.__call__.  Asserts oracle == reference and stores the joints and
the reference outputs."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
import oracle  # noqa: E402
from offsetguided_amd import synth  # noqa: E402
from offsetguided_amd.config.coco_data import COCO_PERSON_SIGMAS, COCO_PERSON_SKELETON  # noqa: E402

sys.modules.setdefault('cv2', types.ModuleType('cv2'))
sys.path.insert(0, '/root/reference')
from encoder.heatmap import HeatMapGenerator  # noqa: E402
from encoder.offset import OffsetMapGenerator  # noqa: E402
sys.path.remove('/root/reference')

GOLD = os.path.join(ROOT, 'tests', 'golden')


def random_joints(seed, persons, size):
    """(P,17,4) fp32 [x, y, v, scale]: people partly outside the image, crowded joints (overlapping windows),
    unannotated joints, scales on both sides of min_jscale."""
    rng = synth.HashRng(seed)
    u = lambda n, lo, hi: rng.uniform(n, lo, hi)  # noqa: E731
    cx, cy = u(persons, -30, size + 30), u(persons, -30, size + 30)
    ext = u(persons, 10, 160)
    j = np.zeros((persons, 17, 4), np.float32)
    j[:, :, 0] = (cx[:, None] + u(persons * 17, -1, 1).reshape(persons, 17) * ext[:, None]).astype(np.float32)
    j[:, :, 1] = (cy[:, None] + u(persons * 17, -1, 1).reshape(persons, 17) * ext[:, None]).astype(np.float32)
    j[:, :, 2] = (u(persons * 17, 0, 1).reshape(persons, 17) > 0.2) * np.round(u(persons * 17, 0.51, 2.49)).reshape(persons, 17)
    j[:, :, 3] = u(persons * 17, 0.2, 40).reshape(persons, 17).astype(np.float32)
    j[:, :, :2] = np.round(j[:, :, :2] * 8) / 8          # eighths: exercises the round-half-even window edges
    return j


def main():
    out = {}
    for name, seed, persons, size in (('a', 11, 6, 256), ('b', 12, 30, 256), ('c', 13, 0, 128), ('d', 14, 12, 512)):
        j = random_joints(seed, persons, size)
        meta = {'joint_num': 17}
        hg = HeatMapGenerator([size, size], 4, 3, 7, 0.01)
        ref_hm = hg.create_heatmaps(j, meta)
        og = OffsetMapGenerator([size, size], 4, 7, 1.0, COCO_PERSON_SKELETON)
        ref_off, ref_sc, ref_ps = og.create_offsetmaps(j, meta)
        ref_jit = hg.create_jitter_offset(j, meta)
        assert np.array_equal(oracle.encode_jitter(j, size, size), ref_jit), name
        hm = oracle.encode_heatmaps(j, size, size)
        off, sc, ps = oracle.encode_offsets(j, COCO_PERSON_SKELETON, COCO_PERSON_SIGMAS, size, size)
        # offsets / scales: bit-exact.  heatmaps: numpy's SIMD float32 exp vs libm expf differ by <= 2 ulp, and a pixel
        # whose value sits at the clip threshold may fall on the other side of it
        assert np.array_equal(off, ref_off) and np.array_equal(sc, ref_sc, equal_nan=True) and np.array_equal(ps, ref_ps), name
        d = np.abs(hm - ref_hm)
        clipped = (np.minimum(hm, ref_hm) == 0) & (np.maximum(hm, ref_hm) < 0.01 * (1 + 1e-5))
        worst = float(d[~clipped].max()) if (~clipped).any() else 0.0
        assert worst <= 1e-6, (name, worst)
        print(f'case {name}: P={persons} size={size}: hm max err {worst:.2e}, '
              f'{int((clipped & (d > 0)).sum())} clip-edge pixels, offsets/scales bit-exact; '
              f'{int(np.isfinite(ref_off).sum())} finite offsets')
        out.update({f'{name}_joints': j, f'{name}_size': np.int64(size), f'{name}_hm': ref_hm, f'{name}_jitter': ref_jit,
                    f'{name}_off': ref_off, f'{name}_scale': ref_sc, f'{name}_pscale': ref_ps})
    np.savez_compressed(os.path.join(GOLD, 'encoder.npz'), **out)


if __name__ == '__main__':
    main()
