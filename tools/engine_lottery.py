#!/usr/bin/env python3
"""Does the forward's time depend on WHEN in a process an engine is built?  Engines built one after another (the earlier ones kept alive
or dropped), each timed over 20 graph replays: a probe for the stream -> hardware-queue assignment of the captured graph's branches."""
import argparse
import os
import sys

os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from offsetguided_amd import models  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--keep', action='store_true', help='keep every engine alive (default: drop it before the next is built)')
    ap.add_argument('--extra-streams', type=int, default=0, help='create this many unrelated streams before every build')
    ap.add_argument('--n', type=int, default=8)
    a = ap.parse_args()
    p = argparse.ArgumentParser()
    models.net_cli(p)
    model, _ = models.model_factory(p.parse_args(['--no-pretrain']))
    bench.bench_init(model, 1234)
    dev = torch.device('cuda:0')
    x = torch.randn(8, 3, 640, 640, device=dev)
    keep, junk = [], []
    for i in range(a.n):
        junk += [torch.cuda.Stream(dev) for _ in range(a.extra_streams)]
        eng = models.InferenceEngine(model, 8, 640, 640, device=dev)
        for _ in range(3):
            eng(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            eng(x)
        e1.record()
        torch.cuda.synchronize()
        print(f'engine {i}: {e0.elapsed_time(e1) / 20:.3f} ms per forward', flush=True)
        if a.keep:
            keep.append(eng)
        del eng


if __name__ == '__main__':
    main()
