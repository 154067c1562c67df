#!/usr/bin/env python3
"""bench.py's harness block (evaluate.run_images from raw host images) eight times in one process: img/s per pass, host time per stage
(packing on the worker thread, preprocessing call, engine call, decoder submit, waiting for the previous batch's poses) and the device-side
pitch between consecutive engine launches.  (Found that the 'bimodal' harness figures of round 4 were a pinned allocation inside the timed
region.)"""
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
    from offsetguided_amd.decoder import factory
    acc = {}
    gpu = {'pre': [], 'eng': []}

    def timed(key, fn, events=None):
        def w(*args, **kw):
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            t = time.perf_counter()
            r = fn(*args, **kw)
            acc[key] = acc.get(key, 0.0) + time.perf_counter() - t
            if events is not None:
                e1.record()
                events.append((e0, e1))
            return r
        return w
    scale.EvalPreprocess.pack = timed('pack', scale.EvalPreprocess.pack)
    scale.EvalPreprocess.__call__ = timed('pre', scale.EvalPreprocess.__call__, gpu['pre'])
    models.InferenceEngine.__call__ = timed('eng', models.InferenceEngine.__call__, gpu['eng'])
    factory.PostProcess.submit = timed('submit', factory.PostProcess.submit)
    factory.PendingPoses.result = timed('result', factory.PendingPoses.result)
    for i in range(8):
        acc.clear()
        for v in gpu.values():
            v.clear()
        h = bench.harness_block(a, model, dev, n_batches=24)
        torch.cuda.synchronize()
        g = {k: sum(e0.elapsed_time(e1) for e0, e1 in v[3:]) / max(len(v) - 3, 1) for k, v in gpu.items()}
        # GPU time between the START of consecutive engine calls (the batch pitch on the device)
        starts = [e0 for e0, _ in gpu['eng'][3:]]
        pitch = [starts[j].elapsed_time(starts[j + 1]) for j in range(len(starts) - 1)]
        print(f'pass {i}: {h["value"]:7.1f} img/s {h["ms_per_batch"]:6.2f} ms/batch | host ms/batch: ' +
              ' '.join(f'{k} {v / 26 * 1e3:.2f}' for k, v in acc.items()) +
              f' | device ms: preprocess {g["pre"]:.2f} engine {g["eng"]:.2f} pitch {sum(pitch) / len(pitch):.2f} (max {max(pitch):.2f})', flush=True)


if __name__ == '__main__':
    main()
