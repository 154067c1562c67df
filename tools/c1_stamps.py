#!/usr/bin/env python3
"""Where a step of og_conv3x3_tiled_bf16's main loop spends its cycles (library built with -DOG_TILED_STAMPS:
tools/build_variants.sh conv3x3.hip c1stamps -DOG_TILED_STAMPS; run with OG_DECODER_LIB=tools/build/libog_c1stamps.so).
Per wave the kernel sums the shader clocks of the three phases of every step -- counted vmcnt wait | barrier | fragment reads +
LDS-DMA issue + 32 MFMA -- and stamps its start / end; prints cycles per step and phase (mean over waves, and the spread), the
in-kernel clock and the workgroup durations, for a lone workgroup per CU (N = 2, 128x128: 256 workgroups) and the product shape."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offsetguided_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', type=str, default='2x128,4x128,8x160')
    ap.add_argument('--warm', type=int, default=200)
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = _lib.load()
    dt = torch.bfloat16
    cl = torch.channels_last
    fn = _lib.lp(lib, 'og_conv3x3_tiled', dt)
    for spec in a.shapes.split(','):
        shape, _, chans = spec.partition(':')              # NxHW[:CINxCOUT], e.g. 8x160 or 8x40:384x384
        n, hw = (int(v) for v in shape.split('x'))
        cin, cout = (int(v) for v in chans.split('x')) if chans else (256, 256)
        wt = (torch.randn(cout, cin, 3, 3, device=dev) * (1.0 / (9 * cin)) ** 0.5).to(dt).contiguous(memory_format=cl)
        packed = torch.empty(wt.numel(), dtype=dt, device=dev)
        _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wt), cin, cout, 0, _lib.ptr(packed), _lib.stream_ptr(dev)), lib)
        bias = torch.zeros(cout, device=dev)
        xs = [torch.randn(n, cin, hw, hw, device=dev).to(dt).contiguous(memory_format=cl) for _ in range(3)]
        outs = [torch.empty(n, cout, hw, hw, device=dev, dtype=dt).contiguous(memory_format=cl) for _ in range(3)]
        sks = [torch.randn(n, cout, hw, hw, device=dev).to(dt).contiguous(memory_format=cl) for _ in range(3)]      # (fixed residual operands:
        # fed back from the outputs, the activations overflow after a few dozen launches and the chip runs inf / NaN at a higher clock)
        tw, th = (16, 16) if hw % 16 == 0 else (hw, 4)
        items = n * (hw // th) * (hw // tw) * (cout // 128)
        chunks = cin // 32
        ksplit = 1 if items >= 200 or tw == 16 else (2 if chunks % 4 == 0 else 1) if tw == 20 else (3 if chunks % 6 == 0 else 1)   # tiled_ksplit
        wgs = items * ksplit
        stamps = torch.zeros(wgs * 4 * 16, dtype=torch.int64, device=dev)
        ws = torch.zeros(max(int(lib.og_conv3x3_tiled_workspace_bytes(n, hw, hw, cin, cout)), 256), dtype=torch.uint8, device=dev)

        def once(i):
            _lib.check(fn(_lib.ptr(xs[i % 3]), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(sks[i % 3]), _lib.ptr(outs[i % 3]),
                          n, hw, hw, cin, cout, 1, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), lib)
        lib.og_conv3x3_debug_stamps(None)
        for i in range(a.warm):          # the clock the chip holds under this load
            once(i)
        lib.og_conv3x3_debug_stamps(C.c_void_p(stamps.data_ptr()))
        once(0)
        lib.og_conv3x3_debug_stamps(None)
        torch.cuda.synchronize()
        s = stamps.view(wgs, 4, 16).double().cpu()
        s[..., 8] = torch.where(s[..., 8] == 0, s[..., 6], s[..., 8])      # K slices that are not the last arriver leave behind the loop
        steps = s[..., 3]
        per = lambda k: (s[..., k] / steps)
        loop_us = (s[..., 6] - s[..., 5]) / 100.0
        clk = (s[..., 12] - s[..., 4]) / (s[..., 6] - s[..., 5]) * 100.0       # MHz
        t0 = s[..., 7].min()
        entry, first, l0, l1, end = ((s[..., k] - t0) / 100.0 for k in (7, 9, 5, 6, 8))
        print(f'{spec}: {wgs} workgroups, {int(steps[0, 0])} steps; first entry -> last end {end.max():.1f} us; in-kernel clock {clk.mean():.0f} MHz '
              f'(min {clk.min():.0f}, max {clk.max():.0f})')
        for name, v in (('entry -> operands of step 0 landed', first - entry), ('main loop (incl. that wait)', l1 - l0),
                        ('main loop without it', l1 - first), ('epilogue (loop end -> stores issued)', end - l1), ('workgroup lifetime', end - entry)):
            f = v.flatten()
            print(f'   {name:38s} {f.mean():6.2f} us  (p10 {f.quantile(0.1):6.2f}, p90 {f.quantile(0.9):6.2f})')
        tot = per(0) + per(1) + per(2)
        for name, k in (('wait', 0), ('barrier', 1), ('reads+dma+mfma', 2)):
            v = per(k)
            print(f'   {name:16s} {v.mean():7.1f} cycles per step  (p10 {v.flatten().quantile(0.1):7.1f}, p90 {v.flatten().quantile(0.9):7.1f})')
        print(f'   {"step":16s} {tot.mean():7.1f} cycles  (MFMAs alone: 512 at 16 x 16, 320 at 40 x 4, 160 at 20 x 4)')
        # slot occupancy per CU: workgroups grouped by (XCC, SE, CU) from HW_ID; the time a CU holds fewer than two workgroups
        hw = s[:, 0, 10].long()
        xcc = s[:, 0, 11].long() & 15
        cu = (xcc << 8) | (((hw >> 13) & 7) << 4) | ((hw >> 8) & 15)       # SE_ID [15:13], CU_ID [11:8]
        e0, e1 = entry[:, 0], end.max(dim=1).values
        import collections
        groups = collections.defaultdict(list)
        for i in range(wgs):
            groups[int(cu[i])].append((float(e0[i]), float(e1[i])))
        span = float(end.max())
        res = []
        for c, iv in groups.items():
            ev = sorted([(a_, 1) for a_, _ in iv] + [(b_, -1) for _, b_ in iv])
            occ, last, t = 0, 0.0, [0.0, 0.0, 0.0, 0.0]
            for tt, d in ev:
                t[min(occ, 3)] += tt - last
                last, occ = tt, occ + d
            t[0] += span - last
            res.append(t)
        r = torch.tensor(res)
        print(f'   {len(groups)} CUs seen; of the {span:.1f} us a CU holds 0 / 1 / 2 / >2 workgroups for {r[:, 0].mean():.1f} / {r[:, 1].mean():.1f} / '
              f'{r[:, 2].mean():.1f} / {r[:, 3].mean():.1f} us on average', flush=True)


if __name__ == '__main__':
    main()
