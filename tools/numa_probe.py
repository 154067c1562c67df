#!/usr/bin/env python3
"""Why the harness figure is bimodal on the shared boxes (diagnosis): evaluate.run_images timed repeatedly in one process, with the host
time of the pinned packing, of the whole preprocessing call and of the engine call accumulated per pass."""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from offsetguided_amd import models, transforms  # noqa: E402
from offsetguided_amd.transforms import scale  # noqa: E402


def main():
    import argparse
    p = argparse.ArgumentParser()
    models.net_cli(p)
    margs = p.parse_args(['--no-pretrain'])
    model, _ = models.model_factory(margs)
    bench.bench_init(model, 1234)
    dev = torch.device('cuda:0')
    a = types.SimpleNamespace(batch=8, size=640)
    acc = {'pack': 0.0, 'pre': 0.0, 'eng': 0.0}
    orig_pack, orig_call, orig_eng = scale.EvalPreprocess.pack, scale.EvalPreprocess.__call__, models.InferenceEngine.__call__

    def timed(key, fn):
        def w(*args, **kw):
            t = time.perf_counter()
            r = fn(*args, **kw)
            acc[key] += time.perf_counter() - t
            return r
        return w
    scale.EvalPreprocess.pack = timed('pack', orig_pack)
    scale.EvalPreprocess.__call__ = timed('pre', orig_call)
    models.InferenceEngine.__call__ = timed('eng', orig_eng)
    for i in range(8):
        for k in acc:
            acc[k] = 0.0
        h = bench.harness_block(a, model, dev, n_batches=24)
        print(f'pass {i}: {h["value"]:7.1f} img/s  {h["ms_per_batch"]:6.2f} ms per batch; host per batch (26 batches): pack {acc["pack"] / 26 * 1e3:5.2f} ms, '
              f'preprocess call {acc["pre"] / 26 * 1e3:5.2f}, engine call {acc["eng"] / 26 * 1e3:5.2f}', flush=True)


if __name__ == '__main__':
    main()
