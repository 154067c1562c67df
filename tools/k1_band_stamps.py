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


def report(buf, n_limb_wgs, title, wbuf=None, launches=1):
    st = buf.reshape(2048, 8).astype(np.float64)
    nb = int((st[:1100, 0] > 0).sum())   # workgroups of the band kernel
    band, mrg = st[:nb], st[1100:1100 + n_limb_wgs]
    t0 = band[:, 0].min()
    b = (band[:, :5] - t0) / 100.0
    m = (mrg[:, :3] - t0) / 100.0
    mi = (mrg[:, 3:6] - t0) / 100.0   # inside merge_plane: keys in LDS, bound known, filtered
    nf = mrg[:, 7]
    nz = mrg[:, 6]
    q = lambda v: f'min {v.min():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}'
    print(title)
    for j, nm in enumerate(['band entry', 'set-up done', 'stream done', 'compacted', 'list stored']):
        print(f'   {nm:13s} {q(b[:, j])}')
    allw, hlp = (band[:, 5] - t0) / 100.0, (band[:, 6] - t0) / 100.0
    print(f'   all waves done {q(allw)};  helper left {q(hlp)};  last wave behind wave 0: {q(allw - b[:, 4])}')
    if wbuf is not None:
        wv = wbuf.reshape(2048, 4, 4)[:nb, :3].astype(np.float64)   # [wg][streaming wave][stream done, end, pushes, compactions]
        wdone, wend = (wv[:, :, 0] - t0) / 100.0, (wv[:, :, 1] - t0) / 100.0
        slow = np.argsort(wend.max(axis=1))[-6:]
        for g_ in slow:
            print(f'   late workgroup {g_}: set-up done {b[g_, 1]:.1f}; per wave stream done {np.round(wdone[g_], 1)}, end {np.round(wend[g_], 1)}, pushes {wv[g_, :, 2].astype(int)}, compactions {wv[g_, :, 3].astype(int)}')
        sd = (wdone - b[:, 1:2]).ravel()
        print(f'   per wave: stream time {q(sd)};  pushes {q(wv[:, :, 2].ravel())};  corr(pushes, stream time) = {np.corrcoef(wv[:, :, 2].ravel(), sd)[0, 1]:.2f};  end - stream done {q((wend - wdone).ravel())}')
    dur, done, blk = b[:, 2] - b[:, 1], b[:, 2], np.arange(nb)
    print('   stream duration by XCD (blockIdx % 8):', np.round([dur[blk % 8 == x].mean() for x in range(8)], 1), ' stream done:', np.round([done[blk % 8 == x].mean() for x in range(8)], 1), ' max:', np.round([done[blk % 8 == x].max() for x in range(8)], 1))
    j = blk // 8   # dispatch order within the XCD = plane-major work item index (og_xcd_remap)
    grp = nb // 8 // 16
    print(f'   {nb} workgroups; by dispatch order (16 groups of {grp} per XCD): duration', np.round([dur[(j // grp) == pl].mean() for pl in range(16)], 1))
    print('                                          done    ', np.round([done[(j // grp) == pl].mean() for pl in range(16)], 1))
    print('   by position in a round of 32 (j % 32, groups of 4): duration', np.round([dur[(j % 32) // 4 == bd].mean() for bd in range(8)], 1), ' done', np.round([done[(j % 32) // 4 == bd].mean() for bd in range(8)], 1))
    print(f'   corr(entry, duration) = {np.corrcoef(b[:, 0], dur)[0, 1]:.2f}')
    print(f'   stream duration {q(b[:, 2] - b[:, 1])};  compaction {q(b[:, 3] - b[:, 2])};  merge+store {q(b[:, 4] - b[:, 3])}')
    for j, nm in enumerate(['merge entry', 'planes merged', 'rows written']):
        print(f'   {nm:13s} {q(m[:, j])}')
    print(f'   merge: entry -> keys in LDS {q(mi[:, 0] - m[:, 0])};  -> bound {q(mi[:, 1] - mi[:, 0])};  -> filtered {q(mi[:, 2] - mi[:, 1])};  -> ranked + emitted {q(m[:, 1] - mi[:, 2])};  keys past the filter {q(nf)};  non-zero keys of the plane (accumulated over launches / launches) {q(nz / max(1, launches))}')
    print(f'   merge duration {q(m[:, 1] - m[:, 0])};  pairing {q(m[:, 2] - m[:, 1])};  gap last band -> first merge {m[:, 0].min() - b[:, 4].max():.2f}')


def main():
    lib = load(sys.argv[1])
    lib.og_k1_band_stamps.argtypes = [C.c_void_p]
    dev = torch.device('cuda:0')
    _lib.load()
    sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    n, c, k, L, h, w = 8, 17, 32, 19, 640, 640
    if '--bench-inputs' in sys.argv:
        from tools.k1_bench import bench_inputs
        bi = bench_inputs(dev, 3, n, h)
        lrs = [x[0] for x in bi]
        t_off = bi[0][1]
    else:
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
    if len(sys.argv) > 2:
        buf0 = np.zeros(2048 * 8, np.int64)
    for it in range(10):
        _lib.check(lib.og_generate_limbs_f32(_lib.ptr(hrs[it % 3]), _lib.ptr(t_off), 1, 2, None, 0, None, 0, n, c, h, w,
                                             _lib.ptr(jf), _lib.ptr(jt), L, k, 0.04, 0.5, 1.0, _lib.ptr(sc), _lib.ptr(ix),
                                             _lib.ptr(limbs), 0, _lib.ptr(ws), ws.numel(), sp), lib)
        torch.cuda.synchronize()
        buf = np.zeros(2048 * 8, np.int64)
        lib.og_k1_band_stamps(buf.ctypes.data)
        if it < 4:
            continue
        wbuf = np.zeros(2048 * 16, np.int64)
        lib.og_k1_wave_stamps.argtypes = [C.c_void_p]
        lib.og_k1_wave_stamps(wbuf.ctypes.data)
        report(buf, n * L, f'launch {it}:', wbuf, it + 1)
        sys.stdout.flush()


if __name__ == '__main__':
    main()
