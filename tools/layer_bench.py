#!/usr/bin/env python3
"""Every large-level layer of the forward ALONE (its tiled kernel, back-to-back launches on rotating activations): us per launch,
TFLOP/s and the fraction of the 2.5 PFLOP/s dense peak -- which layers of Hourglass-104 (bs8, 640x640) sit how far from the matrix
pipe's rate when nothing runs beside them.  The in-network durations are in tools/forward_timeline.py's output."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offsetguided_amd import _lib  # noqa: E402

# (kind, N, Hin, Cin, Cout, launches of this shape in one forward of the 2-stack network)
LAYERS = [('s1', 8, 160, 256, 256, 13), ('s1', 8, 80, 256, 256, 12), ('s1', 8, 80, 256, 384, 0), ('s1', 8, 40, 384, 384, 14),
          ('s1', 8, 40, 384, 256, 2), ('s1', 8, 20, 384, 384, 20),
          ('s2', 8, 320, 128, 256, 1), ('s2', 8, 160, 256, 256, 2), ('s2', 8, 80, 256, 384, 2), ('s2', 8, 40, 384, 384, 2),
          ('pw', 8, 160, 256, 256, 2), ('pw', 8, 160, 128, 256, 1), ('hd', 8, 160, 256, 64, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--dtype', choices=['bf16', 'f16'], default='f16')
    ap.add_argument('--only', default='', help="comma list of kind:Hin to run, e.g. s1:20,s1:40 (default: every layer)")
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = _lib.load()
    dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float16
    cl = torch.channels_last
    only = set(filter(None, a.only.split(',')))
    for kind, n, hin, cin, cout, count in LAYERS:
        if only and f'{kind}:{hin}' not in only:
            continue
        taps = 1 if kind in ('pw', 'hd') else 9
        hout = hin // 2 if kind == 's2' else hin
        xs = [torch.randn(n, cin, hin, hin, device=dev).to(dt).contiguous(memory_format=cl) for _ in range(3)]
        outs = [torch.empty(n, cout, hout, hout, device=dev, dtype=dt).contiguous(memory_format=cl) for _ in range(3)]
        wt = (torch.randn(cout, cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1, device=dev) * (1.0 / (taps * cin)) ** 0.5).to(dt).contiguous(memory_format=cl)
        packed = torch.empty(wt.numel(), dtype=dt, device=dev)
        order = {'s1': 0, 's2': 1, 'pw': 2, 'hd': 3}[kind]
        _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wt), cin, cout, order, _lib.ptr(packed), _lib.stream_ptr(dev)), lib)
        bias = torch.zeros(cout, device=dev)
        if kind == 's1':
            if not lib.og_conv3x3_tiled_supported(n, hin, hin, cin, cout):
                print(kind, n, hin, cin, cout, 'not served')
                continue
            ws = torch.zeros(max(int(lib.og_conv3x3_tiled_workspace_bytes(n, hin, hin, cin, cout)), 256), dtype=torch.uint8, device=dev)
            fn = _lib.lp(lib, 'og_conv3x3_tiled', dt)

            def once(i):
                _lib.check(fn(_lib.ptr(xs[i % 3]), _lib.ptr(packed), _lib.ptr(bias), None, _lib.ptr(outs[i % 3]), n, hin, hin, cin, cout, 1,
                              _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), lib)
        elif kind == 's2':
            if not lib.og_conv3x3s2_tiled_supported(n, hin, hin, cin, cout):
                print(kind, n, hin, cin, cout, 'not served')
                continue
            fn = _lib.lp(lib, 'og_conv3x3s2_tiled', dt)

            def once(i):
                _lib.check(fn(_lib.ptr(xs[i % 3]), _lib.ptr(packed), _lib.ptr(bias), None, _lib.ptr(outs[i % 3]), n, hin, hin, cin, cout, 1,
                              _lib.stream_ptr(dev)), lib)
        elif kind == 'hd':     # the heads of the decoded stack: 17 + 38 channels as fp32 NCHW tensors
            import ctypes as C
            fn = _lib.lp(lib, 'og_conv1x1_heads', dt)
            houts = [torch.empty((n, ch, hin, hin), dtype=torch.float32, device=dev) for ch in (17, 38)]
            chans = (C.c_int * 2)(17, 38)
            ptrs = (C.c_void_p * 2)(*[o.data_ptr() for o in houts])

            def once(i):
                _lib.check(fn(_lib.ptr(xs[i % 3]), cin, _lib.ptr(packed), _lib.ptr(bias), n, hin, hin, cout, 2, chans, ptrs, _lib.stream_ptr(dev)), lib)
        else:
            fn = _lib.lp(lib, 'og_conv1x1_tiled', dt)

            def once(i):
                _lib.check(fn(_lib.ptr(xs[i % 3]), cin, hin, hin, 1, None, 0, 0, 0, 1, _lib.ptr(packed), _lib.ptr(bias), None, _lib.ptr(outs[i % 3]),
                              n, hin, hin, cout, 1, _lib.stream_ptr(dev)), lib)
        for i in range(60):         # (five launches are not enough: the first layer measured read 226 us where it runs in 190 once the chip has
            once(i)                 # settled -- round 5, found with a second tool that happened to warm up longer)
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.reps):
                once(i)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / a.reps)
        us = sorted(ts)[len(ts) // 2]
        flop = 2.0 * n * hout * hout * cout * cin * taps
        mb = (n * hin * hin * cin + n * hout * hout * cout) * 2 / 1e6      # MB / us = TB/s
        print(f'{kind} {n}x{hin}x{hin} {cin:3d}->{cout:3d}  {us:7.1f} us  {flop / us / 1e6:7.1f} TFLOP/s = {flop / us / 1e6 / 2500:5.3f} of peak  '
              f'{mb / us:5.2f} TB/s in+out   x{count} per forward = {us * count:7.1f} us', flush=True)


if __name__ == '__main__':
    main()
