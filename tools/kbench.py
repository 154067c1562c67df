#!/usr/bin/env python3
"""Per-kernel timing of the decoder stages on synthetic bs8 640x640 maps (HIP events on the
launch stream).  Prints GB/s against algorithmic bytes.  `--rotate R` cycles R distinct input
batches so reads come from HBM rather than the 256 MiB Infinity Cache."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offsetguided_amd import decoder, synth  # noqa: E402
from offsetguided_amd.decoder.factory import upsample4  # noqa: E402


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn(0)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    ev[0].record()
    for i in range(iters):
        fn(i)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(iters)]) * 1e3  # us
    return float(np.median(ts)), float(ts.min())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--rotate', type=int, default=3)
    ap.add_argument('--k', type=int, default=32)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    hm, off = synth.synth_batch(0, a.batch, a.size, a.size)
    lr = [torch.from_numpy(hm).to(dev) + 0.0 * r for r in range(a.rotate)]
    offs = torch.from_numpy(off).to(dev)
    hr = [upsample4(x, 'bicubic') for x in lr]
    n, c, H, W = hr[0].shape
    if os.environ.get('OG_DUMP_HR'):
        hr[0].cpu().numpy().tofile(os.environ['OG_DUMP_HR'])
    nbytes = n * c * H * W * 4
    from offsetguided_amd.decoder.collect import LimbsCollect
    from offsetguided_amd.decoder.group import GreedyGroup
    col = LimbsCollect(4, 4, topk=a.k, thre_hmp=0.04, min_len=0.5)
    grp = GreedyGroup(0.04, dist_max=40.0)
    res = {}
    res['bicubic4 (w %.0f MB)' % (nbytes / 1e6)] = (timeit(lambda i: upsample4(lr[i % a.rotate], 'bicubic'), a.iters), nbytes * (1 + 1 / 16))
    res['hmp_NMS (r+w)'] = (timeit(lambda i: decoder.hmp_NMS(hr[i % a.rotate]), a.iters), 2 * nbytes)
    res['joint_dets = NMS+topk (r)'] = (timeit(lambda i: decoder.joint_dets(hr[i % a.rotate], a.k), a.iters), nbytes)
    res['joint_dets_lowres (K1-fused)'] = (timeit(lambda i: decoder.joint_dets_lowres(lr[i % a.rotate], a.k), a.iters), nbytes / 16)
    limbs = col.generate_limbs_lowres(hr[0], offs)
    if os.environ.get('OG_DUMP_LIMBS'):
        limbs.cpu().numpy().tofile(os.environ['OG_DUMP_LIMBS'])
    res['generate_limbs_lowres (K1+K2)'] = (timeit(lambda i: col.generate_limbs_lowres(hr[i % a.rotate], offs), a.iters), nbytes)
    res['K1a+K1+K2 back to back'] = (timeit(lambda i: col.generate_limbs_lowres(upsample4(lr[i % a.rotate], 'bicubic'), offs), a.iters), 2 * nbytes)
    res['group_device (K3)'] = (timeit(lambda i: grp.group_device(limbs), a.iters), 0)
    for k, ((med, mn), b) in res.items():
        print(f'{k:34s} median {med:9.1f} us  min {mn:9.1f} us  {b / med / 1e6 if b else 0:8.2f} TB/s(med)')
    t0 = time.time()
    poses = grp.group_batch(limbs)
    print('poses/img', [len(p) for p in poses], 'group_batch wall %.1f us' % ((time.time() - t0) * 1e6))


if __name__ == '__main__':
    main()
