#!/usr/bin/env python3
"""Time points inside og_conv1x1_tiled_bf16 (library built with -DOG_PW_STAMPS: tools/build_variants.sh conv3x3.hip pwstamps
-DOG_PW_STAMPS; run with OG_DECODER_LIB=tools/build/libog_pwstamps.so): per wave entry | loads issued | operands of step 0 landed
| K loop done | epilogue done, and how many workgroups a CU-slot timeline holds.  160x160, 256 -> 256, batch 8."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offsetguided_amd import _lib  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    lib = _lib.load()
    dt, cl = torch.bfloat16, torch.channels_last
    n, hw, cin, cout = 8, 160, 256, 256
    xs = [torch.randn(n, cin, hw, hw, device=dev).to(dt).contiguous(memory_format=cl) for _ in range(3)]
    outs = [torch.empty(n, cout, hw, hw, device=dev, dtype=dt).contiguous(memory_format=cl) for _ in range(3)]
    wt = (torch.randn(cout, cin, 1, 1, device=dev) * (1.0 / cin) ** 0.5).to(dt).contiguous(memory_format=cl)
    packed = torch.empty(wt.numel(), dtype=dt, device=dev)
    _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wt), cin, cout, 2, _lib.ptr(packed), _lib.stream_ptr(dev)), lib)
    bias = torch.zeros(cout, device=dev)
    fn = _lib.lp(lib, 'og_conv1x1_tiled', dt)
    wgs = (n * hw * hw + 255) // 256 * (cout // 128)
    stamps = torch.zeros(wgs * 4 * 8, dtype=torch.int64, device=dev)

    def once(i):
        _lib.check(fn(_lib.ptr(xs[i % 3]), cin, hw, hw, 1, None, 0, 0, 0, 1, _lib.ptr(packed), _lib.ptr(bias), None, _lib.ptr(outs[i % 3]),
                      n, hw, hw, cout, 1, _lib.stream_ptr(dev)), lib)
    lib.og_conv3x3_debug_stamps(None)
    for i in range(50):
        once(i)
    lib.og_conv3x3_debug_stamps(C.c_void_p(stamps.data_ptr()))
    once(0)
    lib.og_conv3x3_debug_stamps(None)
    torch.cuda.synchronize()
    s = stamps.view(wgs, 4, 8).double().cpu()
    t0 = s[..., 0].min()
    t = (s - t0) / 100.0
    print(f'{wgs} workgroups; first entry -> last end {t[..., 5].max():.1f} us')
    names = ['entry -> loads issued', 'loads issued -> operands of step 0 landed', 'K loop', 'drain of the tail loads', 'epilogue', 'lifetime']
    spans = [t[..., 1] - t[..., 0], t[..., 2] - t[..., 1], t[..., 3] - t[..., 2], t[..., 4] - t[..., 3], t[..., 5] - t[..., 4], t[..., 5] - t[..., 0]]
    for nm, v in zip(names, spans):
        f = v.flatten()
        print(f'   {nm:44s} {f.mean():6.2f} us  (p10 {f.quantile(0.1):6.2f}, p90 {f.quantile(0.9):6.2f})')
    starts = t[:, 0, 0].sort().values
    print('   workgroups started by us 2 / 5 / 10 / 20 / 40:', [int((starts <= x).sum()) for x in (2, 5, 10, 20, 40)], flush=True)


if __name__ == '__main__':
    main()
