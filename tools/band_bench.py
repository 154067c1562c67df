#!/usr/bin/env python3
"""The small hourglass levels, layer by layer: og_conv_band_* (csrc/conv_band.hip) against the kernels it replaces
(og_conv2d_* split-K, og_conv3x3_tiled_* with its K split at 20x20), as a CHAIN of `reps` dependent layers in one HIP graph
(layer l reads what layer l-1 wrote, distinct weights per layer: the situation inside the network), HBM-cold and warm.
Prints us per layer."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offsetguided_amd import _lib  # noqa: E402

SHAPES = [(8, 5, 5, 512, 512), (8, 10, 10, 384, 384), (8, 20, 20, 384, 384), (16, 5, 5, 512, 512), (16, 10, 10, 384, 384)]
_flush = None


def graph_time(fn, reps, rounds=7, cold=True):
    global _flush
    if cold and _flush is None:
        _flush = torch.empty(1 << 30, dtype=torch.uint8, device='cuda:0')
    for l in range(reps):
        fn(l)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for l in range(reps):
            fn(l)
    g.replay()
    torch.cuda.synchronize()
    times = []
    for _ in range(rounds):
        if cold:
            _flush.fill_(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        times.append(s.elapsed_time(e) * 1e3 / reps)
    return sorted(times)[len(times) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=16, help='layers per chain (<= 16)')
    ap.add_argument('--dtype', default='f16', choices=['f16', 'bf16'])
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = _lib.load()
    dt = torch.float16 if a.dtype == 'f16' else torch.bfloat16
    cl = torch.channels_last
    torch.manual_seed(0)
    for n, h, w, cin, cout in SHAPES:
        assert cin == cout
        acts = [torch.randn(n, cin, h, w, device=dev).to(dt).contiguous(memory_format=cl) for _ in range(3)]
        wts = [(torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (9 * cin)) ** 0.5).to(dt).contiguous(memory_format=cl)
               for _ in range(a.reps)]
        bias = torch.zeros(cout, device=dev)
        row = f'{n}x{h}x{w} {cin}->{cout}:'
        # band kernel
        if lib.og_conv_band_supported(n, h, w, cin, cout, 1, 0, 0, 0, 0):
            packed = []
            for wt in wts:
                p = torch.empty(wt.numel(), dtype=dt, device=dev)
                _lib.check(lib.og_conv_band_pack_w16(_lib.ptr(wt), None, cin, cout, 0, _lib.ptr(p), _lib.stream_ptr(dev)), lib)
                packed.append(p)
            fn = _lib.lp(lib, 'og_conv_band', dt)

            def band(l):
                _lib.check(fn(_lib.ptr(acts[l % 3]), _lib.ptr(packed[l]), _lib.ptr(bias), _lib.ptr(acts[(l + 2) % 3]), None,
                              _lib.ptr(acts[(l + 1) % 3]), n, h, w, cin, cout, 1, 1, 0, 0, 0, 1, _lib.stream_ptr(dev)), lib)
            row += f'  band {graph_time(band, a.reps):6.2f} cold {graph_time(band, a.reps, cold=False):6.2f} warm'
        # split-K kernel
        ws = torch.zeros(lib.og_conv2d_workspace_bytes(n, h, w, cin, cout, 3, 1), dtype=torch.uint8, device=dev)
        fn2 = _lib.lp(lib, 'og_conv2d', dt)

        def splitk(l):
            _lib.check(fn2(_lib.ptr(acts[l % 3]), _lib.ptr(wts[l]), _lib.ptr(bias), _lib.ptr(acts[(l + 2) % 3]), _lib.ptr(acts[(l + 1) % 3]),
                           n, h, w, cin, cout, 3, 1, 1, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), lib)
        row += f'  | split-K {graph_time(splitk, a.reps):6.2f} cold {graph_time(splitk, a.reps, cold=False):6.2f} warm'
        if lib.og_conv3x3_tiled_supported(n, h, w, cin, cout):
            tp = []
            for wt in wts:
                p = torch.empty(wt.numel(), dtype=dt, device=dev)
                _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wt), cin, cout, 0, _lib.ptr(p), _lib.stream_ptr(dev)), lib)
                tp.append(p)
            need = lib.og_conv3x3_tiled_workspace_bytes(n, h, w, cin, cout)
            tws = torch.zeros(max(need, 256), dtype=torch.uint8, device=dev)
            fn3 = _lib.lp(lib, 'og_conv3x3_tiled', dt)

            def tiled(l):
                _lib.check(fn3(_lib.ptr(acts[l % 3]), _lib.ptr(tp[l]), _lib.ptr(bias), _lib.ptr(acts[(l + 2) % 3]), _lib.ptr(acts[(l + 1) % 3]),
                               n, h, w, cin, cout, 1, _lib.ptr(tws) if need else None, tws.numel() if need else 0, _lib.stream_ptr(dev)), lib)
            row += f'  | tiled {graph_time(tiled, a.reps):6.2f} cold {graph_time(tiled, a.reps, cold=False):6.2f} warm'
        print(row, flush=True)


if __name__ == '__main__':
    main()
