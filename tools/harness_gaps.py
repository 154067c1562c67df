#!/usr/bin/env python3
"""GPU timeline of evaluate.run_images (bench.py's harness block) from a rocprofv3 kernel trace: pitch between consecutive forwards
(stem kernel to stem kernel), what runs between the last kernel of one forward's decoder and the next stem, and the idle gaps.
  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/harness_gaps.py --run      (runs two harness passes)
  python3 tools/harness_gaps.py <dir>/*/*_kernel_trace.csv                                          (analyses the trace)"""
import csv
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    import bench
    from offsetguided_amd import models
    import argparse
    p = argparse.ArgumentParser()
    models.net_cli(p)
    model, _ = models.model_factory(p.parse_args(['--no-pretrain']))
    bench.bench_init(model, 1234)
    a = types.SimpleNamespace(batch=8, size=640)
    dev = torch.device('cuda:0')
    for _ in range(2):
        print(bench.harness_block(a, model, dev, n_batches=24), file=sys.stderr)


def analyse(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    stems = [i for i, r in enumerate(rows) if 'stem7x7' in r['Kernel_Name']]
    stems = stems[-24:]                                   # the last pass's timed batches
    pitches = []
    for a, b in zip(stems[:-1], stems[1:]):
        seg = rows[a:b]
        t0, t1 = int(seg[0]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
        busy, cur_end = 0, t0
        gaps = []
        for r in seg:
            s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
            if s > cur_end:
                gaps.append(((s - cur_end) / 1e3, r['Kernel_Name'][:50]))
            busy += max(0, e - max(s, cur_end))
            cur_end = max(cur_end, e)
        if t1 > cur_end:
            gaps.append(((t1 - cur_end) / 1e3, 'next stem'))
        pitches.append(((t1 - t0) / 1e3, busy / 1e3, sorted(gaps, reverse=True)[:4]))
    for p, b, g in pitches[:8]:
        print(f'pitch {p:8.1f} us, some kernel running {b:8.1f} us, largest gaps: ' + '; '.join(f'{x:.0f} us before {n}' for x, n in g))
    import statistics
    print('median pitch', statistics.median(p for p, _, _ in pitches), 'median busy', statistics.median(b for _, b, _ in pitches))
    # kernels between the heads kernel of a forward and the next stem
    a, b = stems[4], stems[5]
    heads = max(i for i in range(a, b) if 'conv1x1_tiled_kernel<64' in rows[i]['Kernel_Name'])
    t_h = int(rows[heads]['End_Timestamp'])
    print('after the heads kernel of one forward:')
    for r in rows[heads + 1:b + 1]:
        print(f"   +{(int(r['Start_Timestamp']) - t_h) / 1e3:8.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  q{r['Queue_Id']}  {r['Kernel_Name'][:70]}")


if __name__ == '__main__':
    if sys.argv[1] == '--run':
        run()
    else:
        analyse(sys.argv[1])
