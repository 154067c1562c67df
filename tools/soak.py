#!/usr/bin/env python3
"""Soak: (1) bench.py's loop for --steps steps with two batches in flight, (2) evaluate.run_images with --fixed-height over --batches
batches of varying widths (engine churn: builds, evictions, stream release / reuse) -- throughput per block of batches, the number of
HIP streams the process made, and GPU memory at the end.  usage: soak.py [--batches 300]"""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from offsetguided_amd import _lib, evaluate, models  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', type=int, default=300)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    args = evaluate.evaluate_cli(['--no-pretrain', '--initialize-whole', 'False', '--topk', '32', '--thre-hmp', '0.04', '--person-thre', '0.04',
                                  '--dist-max', '40', '--long-edge', '256', '--batch-size', '1', '--print-freq', '100000', '--fixed-height'])
    model, _ = models.model_factory(args)
    rng = np.random.default_rng(3)
    widths = [90, 150, 210, 260, 330, 400, 470, 520]          # -> eight padded widths, ENGINE_CACHE = 4: constant churn
    raw = [rng.integers(0, 256, (100, w, 3), dtype=np.uint8) for w in widths]

    def loader(n):
        for i in range(n):
            k = int(rng.integers(0, len(raw))) if i % 3 else i % len(raw)
            yield [raw[k]], [None], [{'image_id': i + 1}]
    for block in range(3):
        t0 = time.perf_counter()
        res, ids = evaluate.run_images(args, data_loader=loader(a.batches // 3), model=model)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        made = sum(1 for _ in _lib._own_streams) + sum(len(v) for v in _lib._free_streams.values())
        print(f'block {block}: {len(ids)} images in {dt:.1f} s, {len(res)} results; own streams keyed {len(_lib._own_streams)}, on the free list '
              f'{sum(len(v) for v in _lib._free_streams.values())}; memory allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB, '
              f'reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB', flush=True)
    print('soak ok')


if __name__ == '__main__':
    main()
