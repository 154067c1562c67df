#!/usr/bin/env python3
"""Does the replay time of the bs8 640x640 forward depend on the capture / instantiation it came from?  Builds the engine R times in one
process (same layers, fresh HIP graph each) and times 20 replays of each, interleaved over two rounds.
usage: capture_variance.py [R] [attr=v0,v1,...]   -- with attr: engine i is captured with models.engine.<attr> = v[i % len(v)] (switches that
are read at capture time, e.g. SKIP_FORK=0,1,2): an A/B inside one process, free of the box-to-box and process-to-process spread."""
import sys
import time

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from offsetguided_amd import models  # noqa: E402
import argparse  # noqa: E402


def main():
    r = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    dev = torch.device('cuda', 0)
    p = argparse.ArgumentParser()
    models.net_cli(p)
    model, _ = models.model_factory(p.parse_args(['--no-pretrain']))
    bench.bench_init(model, 1234)
    x = torch.randn(8, 3, 640, 640, device=dev)
    from offsetguided_amd.models import engine as eng_mod
    attr, vals = None, [None]
    if len(sys.argv) > 2:
        attr, vs = sys.argv[2].split('=')
        vals = [int(v) for v in vs.split(',')]          # (negative values allowed: stream priorities)
    engines = []
    for i in range(r):
        if attr:
            setattr(eng_mod, attr, vals[i % len(vals)])
        engines.append(models.InferenceEngine(model, 8, 640, 640, dtype=torch.float16, device=dev))
    if attr:
        print('engines:', ' '.join(f'{attr}={vals[i % len(vals)]}' for i in range(r)))
    for rnd in range(3):
        line = []
        for e in engines:
            for _ in range(3):
                e.forward_raw(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                e.forward_raw(x)
            torch.cuda.synchronize()
            line.append((time.perf_counter() - t0) / 20 * 1e3)
        print('round', rnd, ' '.join(f'{v:.3f}' for v in line), flush=True)


if __name__ == '__main__':
    main()
