#!/usr/bin/env python3
"""Phase timeline of the single-launch K1 (library built with -DOG_K1_STAMPS): per-workgroup wall-clock stamps."""
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
    path = sys.argv[1]
    lib = load(path)
    lib.og_k1_debug_stamps.argtypes = [C.c_void_p]
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
    for it in range(14):
        _lib.check(lib.og_generate_limbs_f32(_lib.ptr(hrs[it % 3]), _lib.ptr(t_off), 1, 2, None, 0, None, 0, n, c, h, w,
                                             _lib.ptr(jf), _lib.ptr(jt), L, k, 0.04, 0.5, 1.0, None, None, _lib.ptr(limbs),
                                             1, _lib.ptr(ws), ws.numel(), sp), lib)
        torch.cuda.synchronize()
        buf = np.zeros(1024 * 16, np.int64)
        lib.og_k1_debug_stamps(buf.ctypes.data)
        st = buf.reshape(1024, 16)[:256].astype(np.float64)
        t0 = st[:, 0].min()
        us = (st[:, :6] - t0) / 100.0
        names = ['entry', 'init done', 'stream done', 'lists stored', 'ticket done', 'exit']
        rows = np.diff(ws[61440:65536].view(torch.int32).cpu().numpy()[16:16 + 257])
        print(f'launch {it}: rows per workgroup min/median/max', rows.min(), int(np.median(rows)), rows.max(), ' by XCD', np.round([rows[x::8].mean() for x in range(8)], 0))
        for j, nm in enumerate(names):
            print(f'   {nm:13s} min {us[:, j].min():7.2f}  median {np.median(us[:, j]):7.2f}  max {us[:, j].max():7.2f} us')
        sd = us[:, 2]
        print('   entry by XCD:', np.round([us[x::8, 0].mean() for x in range(8)], 1), ' stream duration by XCD:', np.round([(us[x::8, 2] - us[x::8, 1]).mean() for x in range(8)], 1))
        print('   stream-done by XCD (wg % 8):', np.round([sd[x::8].mean() for x in range(8)], 1))
        print('   stream-done by wg/32 block:', np.round([sd[b * 32:(b + 1) * 32].mean() for b in range(8)], 1))
        order = np.argsort(sd)
        print('   slowest wgs:', order[-12:], np.round(sd[order[-12:]], 1), ' fastest:', order[:8], np.round(sd[order[:8]], 1))
        fin = st[:, 15] > 0
        print('   finisher ticket->merged', np.round((st[fin, 6] - st[fin, 4]) / 100.0, 1), ' merged->exit', np.round((st[fin, 5] - st[fin, 6]) / 100.0, 1))
        print('   wave0: ticket->loads issued', np.round((st[fin, 9] - st[fin, 4]) / 100.0, 1), ' ->plane A merged', np.round((st[fin, 10] - st[fin, 9]) / 100.0, 1), ' ->plane B merged', np.round((st[fin, 11] - st[fin, 10]) / 100.0, 1), ' ->barrier', np.round((st[fin, 6] - st[fin, 11]) / 100.0, 1))
        print('   collect: ->offsets gathered', np.round((st[fin, 12] - st[fin, 6]) / 100.0, 1), ' ->argmin done', np.round((st[fin, 13] - st[fin, 12]) / 100.0, 1), ' ->exit', np.round((st[fin, 5] - st[fin, 13]) / 100.0, 1))
        if st[fin, 7].max() > 0:
            print('   SECOND pass: merge', np.round((st[fin, 8] - st[fin, 7]) / 100.0, 1), ' collect', np.round((st[fin, 5] - st[fin, 8]) / 100.0, 1), ' first pass collect', np.round((st[fin, 7] - st[fin, 6]) / 100.0, 1))
        print(f'   finishers: {int(fin.sum())}; their exit: {np.sort(us[fin, 5]).round(1)}; ticket->exit {np.round(us[fin, 5] - us[fin, 4], 1)}')


if __name__ == '__main__':
    main()
