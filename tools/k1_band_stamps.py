#!/usr/bin/env python3
"""Time points inside the two launches of K1 (library built with -DOG_K1_STAMPS, tools/build_variants.sh): per-workgroup
wall-clock stamps of band_topk_kernel (entry, set-up done, stream done, wave list compacted, band list stored) and of
merge_collect_kernel (entry, planes merged, limb rows written), relative to the first band workgroup's entry."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offsetguided_amd import _lib, synth  # noqa: E402
from offsetguided_amd.config import coco_data as cd  # noqa: E402
from tools.k1_bench import load  # noqa: E402


def main():
    lib = load(sys.argv[1])
    lib.og_k1_band_stamps.argtypes = [C.c_void_p]
    dev = torch.device('cuda:0')
    _lib.load()
    sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    n, c, k, L, h, w = 8, 17, 32, 19, 640, 640
    hm, off = synth.synth_batch(0, n, h, w)
    lrs = [torch.from_numpy(hm).to(dev) * (1.0 - 0.01 * r) for r in range(3)]
    t_off = torch.from_numpy(off).to(dev)
    hrs = [torch.empty((n, c, h, w), device=dev) for _ in range(3)]
    for i in range(3):
        _lib.check(lib.og_upsample_bicubic4_f32(_lib.ptr(lrs[i]), n * c, h // 4, w // 4, _lib.ptr(hrs[i]), sp), lib)
    jf = _lib.int_table([x for x, _ in cd.COCO_PERSON_SKELETON], dev)
    jt = _lib.int_table([y for _, y in cd.COCO_PERSON_SKELETON], dev)
    ws = torch.zeros(lib.og_generate_limbs_workspace_bytes(n, c, h, w, k), dtype=torch.uint8, device=dev)
    limbs = torch.empty((n, L, k, 13), device=dev)
    sc = torch.empty((n, c, k), device=dev)
    ix = torch.empty((n, c, k), dtype=torch.int64, device=dev)
    nb = 1088
    for it in range(10):
        _lib.check(lib.og_generate_limbs_f32(_lib.ptr(hrs[it % 3]), _lib.ptr(t_off), 1, 2, None, 0, None, 0, n, c, h, w,
                                             _lib.ptr(jf), _lib.ptr(jt), L, k, 0.04, 0.5, 1.0, _lib.ptr(sc), _lib.ptr(ix),
                                             _lib.ptr(limbs), 0, _lib.ptr(ws), ws.numel(), sp), lib)
        torch.cuda.synchronize()
        buf = np.zeros(2048 * 8, np.int64)
        lib.og_k1_band_stamps(buf.ctypes.data)
        if it < 4:
            continue
        st = buf.reshape(2048, 8).astype(np.float64)
        band, mrg = st[:nb], st[1100:1100 + n * L]
        t0 = band[:, 0].min()
        b = (band[:, :5] - t0) / 100.0
        m = (mrg[:, :3] - t0) / 100.0
        q = lambda v: f'min {v.min():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}'
        print(f'launch {it}:')
        for j, nm in enumerate(['band entry', 'set-up done', 'stream done', 'compacted', 'list stored']):
            print(f'   {nm:13s} {q(b[:, j])}')
        print(f'   stream duration {q(b[:, 2] - b[:, 1])};  compaction {q(b[:, 3] - b[:, 2])};  merge+store {q(b[:, 4] - b[:, 3])}')
        for j, nm in enumerate(['merge entry', 'planes merged', 'rows written']):
            print(f'   {nm:13s} {q(m[:, j])}')
        print(f'   merge duration {q(m[:, 1] - m[:, 0])};  pairing {q(m[:, 2] - m[:, 1])};  gap last band -> first merge {m[:, 0].min() - b[:, 4].max():.2f}')
        sys.stdout.flush()


if __name__ == '__main__':
    main()
