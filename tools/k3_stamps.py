#!/usr/bin/env python3
"""Phase breakdown of K3 (greedy_group_kernel) on the bs8 640x640 synthetic batch, from a -DOG_K3_STAMPS build:
  tools/build_variants.sh group.hip k3s "-DOG_K3_STAMPS=1";  python tools/k3_stamps.py tools/build/libog_k3s.so"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if __name__ == '__main__':
    os.environ['OG_DECODER_LIB'] = os.path.abspath(sys.argv[1])      # (before offsetguided_amd._lib is imported: it reads the variable then)
from offsetguided_amd import _lib, decoder, models, synth  # noqa: E402
import argparse  # noqa: E402

NAMES = ['U4 stage the surviving rows', 'match', 'apply', 'merge search + merge', 'new-row slots', 'fill + reset', 'final (score, sort, copy)', 'U1 load + validity filter', 'U2 rank + duplicate to-index', 'U3 rank among survivors']


def main():
    dev = torch.device('cuda:0')
    p = argparse.ArgumentParser()
    models.net_cli(p)
    decoder.decoder_cli(p)
    a = p.parse_args(['--no-pretrain', '--topk', '32', '--thre-hmp', '0.04', '--person-thre', '0.04', '--dist-max', '40'])
    a.batch_size = 8
    proc = decoder.decoder_factory(a)
    hm, off = synth.synth_batch(0, 8, 640, 640)
    feats = [([None, torch.from_numpy(hm).to(dev)], [[], []], [[], []]), ([None, torch.from_numpy(off).to(dev)], [[], []], [[], []])]
    limbs = proc.generate_limbs(feats)
    for _ in range(3):
        proc.limb_group.group_device(limbs)
    torch.cuda.synchronize()
    evs = []
    for _ in range(20):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        proc.limb_group.group_device(limbs)
        e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    t = np.median([s.elapsed_time(e) for s, e in evs]) * 1e3
    lib = _lib.load()
    if not hasattr(lib, 'og_k3_debug_stamps'):
        print(f'K3 per batch (events): {t:.1f} us  (library without -DOG_K3_STAMPS: no phase breakdown)')
        return
    buf = (C.c_ulonglong * 16)()
    lib.og_k3_debug_stamps(buf)
    c = np.array(buf[:10], dtype=np.float64)
    print(f'K3 per batch (events): {t:.1f} us;  image 0 by phase (share of its {c.sum():.0f} cycles):')
    for n, v in zip(NAMES, c):
        print(f'   {n:42s} {100 * v / c.sum():5.1f} %   ~{t * v / c.sum():5.1f} us')


if __name__ == '__main__':
    main()
