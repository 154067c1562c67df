#!/usr/bin/env python3
"""Where evaluate.run_images spends its host time (bench.py's `harness` block under cProfile) and the GPU time of its kernels."""
import cProfile
import io
import os
import pstats
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from offsetguided_amd import models  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    import argparse
    p = argparse.ArgumentParser()
    models.net_cli(p)
    margs = p.parse_args(['--no-pretrain'])
    model, _ = models.model_factory(margs)
    bench.bench_init(model, 1234)
    a = types.SimpleNamespace(batch=8, size=640)
    print('warm:', bench.harness_block(a, model, dev, n_batches=6)['value'], 'img/s')
    pr = cProfile.Profile()
    pr.enable()
    out = bench.harness_block(a, model, dev, n_batches=24)
    pr.disable()
    print(out['value'], 'img/s', out['ms_per_batch'], 'ms per batch')
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
    print(s.getvalue()[:6000])


if __name__ == '__main__':
    main()
