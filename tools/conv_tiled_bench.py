#!/usr/bin/env python3
"""og_conv3x3_tiled_* (two 4-wave workgroups per CU, pre-tiled weights) against MIOpen's convolution + og_bias_act_* on the
large-level 3x3 shapes, in ONE process, interleaved rounds (cdna_hip_programming.md rule 24): random operands, rotating
activations, `reps` back-to-back launches per HIP graph; median / minimum us per launch, TFLOP/s of the median, and the largest
difference between the two results.  --s2: the stride-2 kernel on the down-sampling layers.  (The round-2 halo kernel and the
OG_TILED_VAR / OG_TILED_KSPLIT arms this tool used to time left the library in round 4: profiles/r03_*.)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offsetguided_amd import _lib  # noqa: E402

SHAPES = [(8, 160, 160, 256, 256), (8, 80, 80, 256, 256), (8, 40, 40, 384, 384), (16, 160, 160, 256, 256),
          (8, 40, 40, 384, 256), (8, 40, 40, 256, 256), (8, 20, 20, 384, 384)]


def main():
    import torch.nn.functional as F
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=8)
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--only', type=int, nargs='*', default=[])
    ap.add_argument('--dtype', choices=['bf16', 'f16'], default='bf16')
    ap.add_argument('--no-miopen', action='store_true', help='time the tiled kernel only')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = _lib.load()
    dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float16
    cl = torch.channels_last
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(0)
    for si, (n, h, w, cin, cout) in enumerate(SHAPES):
        if a.only and si not in a.only:
            continue
        xs = [torch.randn(n, cin, h, w, device=dev).to(dt).contiguous(memory_format=cl) for _ in range(3)]
        skip = torch.randn(n, cout, h, w, device=dev).to(dt).contiguous(memory_format=cl)
        out = torch.empty_like(skip)
        wt = (torch.randn(cout, cin, 3, 3, device=dev) * (1.0 / (9 * cin)) ** 0.5).to(dt).contiguous(memory_format=cl)
        bias = torch.randn(cout, device=dev) * 0.1
        packed = torch.empty(wt.numel(), dtype=dt, device=dev)
        _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wt), cin, cout, 0, _lib.ptr(packed), _lib.stream_ptr(dev)), lib)
        tws = torch.zeros(max(int(lib.og_conv3x3_tiled_workspace_bytes(n, h, w, cin, cout)), 256), dtype=torch.uint8, device=dev)
        new_fn, ba_fn = _lib.lp(lib, 'og_conv3x3_tiled', dt), _lib.lp(lib, 'og_bias_act', dt)
        holder = {}

        def miopen(i):
            y = F.conv2d(xs[i % 3], wt, None, 1, 1)
            _lib.check(ba_fn(_lib.ptr(y), _lib.ptr(bias), _lib.ptr(skip), n * h * w, cout, 1, _lib.stream_ptr(dev)), lib)
            holder['y'] = y

        def new(i):
            _lib.check(new_fn(_lib.ptr(xs[i % 3]), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(skip), _lib.ptr(out), n, h, w,
                              cin, cout, 1, _lib.ptr(tws), tws.numel(), _lib.stream_ptr(dev)), lib)

        arms = [('tiled', new)] + ([] if a.no_miopen else [('miopen+epilogue', miopen)])
        for _, fn in arms:
            fn(0)
        torch.cuda.synchronize()
        diff = (holder['y'].float() - out.float()).abs().max().item() if not a.no_miopen else float('nan')
        graphs = {}
        for name, fn in arms:
            for i in range(a.reps):
                fn(i)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(a.reps):
                    fn(i)
            graphs[name] = g
        times = {k: [] for k in graphs}
        for _ in range(a.rounds):
            for name, g in graphs.items():
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                g.replay()
                e.record()
                torch.cuda.synchronize()
                times[name].append(s.elapsed_time(e) * 1e3 / a.reps)
        flop = 2.0 * n * h * w * cout * 9 * cin
        line = f'[{si}] {n}x{h}x{w} {cin}->{cout} {a.dtype}:'
        for name, t in times.items():
            t = sorted(t)
            med = t[len(t) // 2]
            line += f'  {name} {med:7.1f} us (min {t[0]:7.1f}) = {flop / med / 1e6:6.0f} TFLOP/s'
        print(line + f'  | max |miopen - tiled| = {diff:.4f}', flush=True)


def main_s2():
    """--s2: og_conv3x3s2_tiled_bf16 against MIOpen's stride-2 convolution + og_bias_act_bf16 on the network's large
    down-sampling layers, interleaved rounds."""
    import torch.nn.functional as F
    ap = argparse.ArgumentParser()
    ap.add_argument('--s2', action='store_true')
    ap.add_argument('--reps', type=int, default=8)
    ap.add_argument('--rounds', type=int, default=7)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = _lib.load()
    torch.backends.cudnn.benchmark = True
    cl = torch.channels_last
    dt = torch.bfloat16
    for si, (n, h, w, cin, cout) in enumerate([(8, 320, 320, 128, 256), (8, 160, 160, 256, 256), (8, 80, 80, 256, 384)]):
        xs = [torch.randn(n, cin, h, w, device=dev).to(dt).contiguous(memory_format=cl) for _ in range(3)]
        wt = (torch.randn(cout, cin, 3, 3, device=dev) * (1.0 / (9 * cin)) ** 0.5).to(dt).contiguous(memory_format=cl)
        bias = torch.randn(cout, device=dev) * 0.1
        packed = torch.empty(wt.numel(), dtype=dt, device=dev)
        _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wt), cin, cout, 1, _lib.ptr(packed), _lib.stream_ptr(dev)), lib)
        out = torch.empty((n, cout, h // 2, w // 2), dtype=dt, device=dev, memory_format=cl)
        holder = {}

        def miopen(i):
            y = F.conv2d(xs[i % 3], wt, None, 2, 1)
            _lib.check(lib.og_bias_act_bf16(_lib.ptr(y), _lib.ptr(bias), None, n * (h // 2) * (w // 2), cout, 1, _lib.stream_ptr(dev)), lib)
            holder['y'] = y

        def ours(i):
            _lib.check(lib.og_conv3x3s2_tiled_bf16(_lib.ptr(xs[i % 3]), _lib.ptr(packed), _lib.ptr(bias), None, _lib.ptr(out), n, h, w,
                                                   cin, cout, 1, _lib.stream_ptr(dev)), lib)

        miopen(0)
        ours(0)
        torch.cuda.synchronize()
        diff = (holder['y'].float() - out.float()).abs().max().item()
        graphs = {}
        for name, fn in [('miopen+epilogue', miopen), ('tiled_s2', ours)]:
            for i in range(a.reps):
                fn(i)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(a.reps):
                    fn(i)
            graphs[name] = g
        times = {k: [] for k in graphs}
        for _ in range(a.rounds):
            for name, g in graphs.items():
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                g.replay()
                e.record()
                torch.cuda.synchronize()
                times[name].append(s.elapsed_time(e) * 1e3 / a.reps)
        flop = 2.0 * n * (h // 2) * (w // 2) * cout * 9 * cin
        line = f'[s2 {si}] {n}x{h}x{w} {cin}->{cout}:'
        for name, t in times.items():
            t = sorted(t)
            med = t[len(t) // 2]
            line += f'  {name} {med:7.1f} us (min {t[0]:7.1f}) = {flop / med / 1e6:6.0f} TFLOP/s'
        print(line + f'  | max |miopen - tiled| = {diff:.4f}', flush=True)


if __name__ == '__main__':
    if '--s2' in sys.argv:
        main_s2()
    else:
        main()
