#!/usr/bin/env python3
"""How og_conv3x3_tiled_* scales with the number of workgroups against the 512 slots of the chip (two per CU): 256 -> 256
channels on (N, 128, 128) activations = 128 workgroups per image, plus the product shape (8, 160, 160) = 1 600.  Prints us per
launch (median of rounds of back-to-back launches on rotating activations) and us per 512 workgroups."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offsetguided_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--dtype', choices=['bf16', 'f16'], default='f16')
    ap.add_argument('--shapes', type=str, default='4x128,6x128,8x128,10x128,12x128,13x128,14x128,16x128,8x160,6x160,5x160')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = _lib.load()
    dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float16
    cl = torch.channels_last
    fn = _lib.lp(lib, 'og_conv3x3_tiled', dt)
    wt = (torch.randn(256, 256, 3, 3, device=dev) * (1.0 / 2304) ** 0.5).to(dt).contiguous(memory_format=cl)
    packed = torch.empty(wt.numel(), dtype=dt, device=dev)
    _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wt), 256, 256, 0, _lib.ptr(packed), _lib.stream_ptr(dev)), lib)
    bias = torch.zeros(256, device=dev)
    for spec in a.shapes.split(','):
        n, hw = (int(v) for v in spec.split('x'))
        xs = [torch.randn(n, 256, hw, hw, device=dev).to(dt).contiguous(memory_format=cl) for _ in range(9)]     # inputs | residual operands | outputs: no feedback (overflow)
        ws_bytes = max(int(lib.og_conv3x3_tiled_workspace_bytes(n, hw, hw, 256, 256)), 256)
        ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)

        def once(i):
            _lib.check(fn(_lib.ptr(xs[i % 3]), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(xs[3 + i % 3]), _lib.ptr(xs[6 + i % 3]),
                          n, hw, hw, 256, 256, 1, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), lib)
        for i in range(5):
            once(i)
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
        ts.sort()
        wgs = n * (hw // 16) ** 2 * 2
        us = ts[len(ts) // 2]
        tf = 2.0 * n * hw * hw * 256 * 2304 / us / 1e6
        print(f'{spec:8s} workgroups {wgs:5d} = {wgs / 512:5.3f} x 512   {us:7.1f} us   {us / (wgs / 512):6.1f} us per 512   {tf:7.1f} TFLOP/s', flush=True)


if __name__ == '__main__':
    main()
