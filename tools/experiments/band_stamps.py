#!/usr/bin/env python3
"""Time points inside og_conv_band_bf16 (library built with -DOG_BAND_STAMPS: tools/build_variants.sh conv_band.hip stamps
-DOG_BAND_STAMPS; run with OG_DECODER_LIB=tools/build/libog_stamps.so): a chain of dependent layers in one HIP graph, per
workgroup s_memrealtime marks (100 MHz): entry, loads landed + LDS written, barrier, MFMAs done, partials exchanged, stores
issued, stores drained -- relative to the first workgroup's entry of each launch, and the launch-to-launch gaps."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offsetguided_amd import _lib  # noqa: E402


def main():
    lib = _lib.load()
    lib.og_conv_band_debug_stamps.argtypes = [C.c_void_p]
    lib.og_conv_band_debug_stamps.restype = None
    dev = torch.device('cuda:0')
    dt, cl, reps = torch.bfloat16, torch.channels_last, 12
    flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    for n, h, w, c in [(8, 5, 5, 512), (8, 10, 10, 384)]:
        acts = [torch.randn(n, c, h, w, device=dev).to(dt).contiguous(memory_format=cl) for _ in range(3)]
        bias = torch.zeros(c, device=dev)
        packed = []
        for _ in range(reps):
            wt = (torch.randn(c, c, 3, 3, device=dev) * (2.0 / (9 * c)) ** 0.5).to(dt).contiguous(memory_format=cl)
            p = torch.empty(wt.numel(), dtype=dt, device=dev)
            _lib.check(lib.og_conv_band_pack_w16(_lib.ptr(wt), None, c, c, 0, _lib.ptr(p), _lib.stream_ptr(dev)), lib)
            packed.append(p)
        stamps = torch.zeros(reps * 4096 * 8, dtype=torch.int64, device=dev)

        def layer(l):
            _lib.check(lib.og_conv_band_bf16(_lib.ptr(acts[l % 3]), _lib.ptr(packed[l]), _lib.ptr(bias), _lib.ptr(acts[(l + 2) % 3]), None,
                                             _lib.ptr(acts[(l + 1) % 3]), n, h, w, c, c, 1, 1, 0, 0, 0, 1, _lib.stream_ptr(dev)), lib)
        lib.og_conv_band_debug_stamps(None)
        for l in range(reps):
            layer(l)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        lib.og_conv_band_debug_stamps(C.c_void_p(stamps.data_ptr()))
        with torch.cuda.graph(g):
            for l in range(reps):
                layer(l)
        lib.og_conv_band_debug_stamps(None)
        # the same layers as one chained launch
        cacts = [acts[0]] + [torch.empty_like(acts[0]) for _ in range(reps)]
        descs = [_lib.BandLayerDesc(_lib.ptr(cacts[l]), _lib.ptr(packed[l]), _lib.ptr(bias), _lib.ptr(cacts[l - 1]) if l else None, None,
                                    _lib.ptr(cacts[l + 1]), n, h, w, c, c, 1, 1, 0, 0, 0, 1) for l in range(reps)]
        cws = torch.zeros(lib.og_conv_band_chain_workspace_bytes(), dtype=torch.uint8, device=dev)
        cstamps = torch.zeros(reps * 4096 * 8, dtype=torch.int64, device=dev)
        _lib.band_chain(descs, dt, cws, dev)
        torch.cuda.synchronize()
        gc = torch.cuda.CUDAGraph()
        lib.og_conv_band_debug_stamps(C.c_void_p(cstamps.data_ptr()))
        with torch.cuda.graph(gc):
            _lib.band_chain(descs, dt, cws, dev)
        lib.og_conv_band_debug_stamps(None)
        for cold in (True, False):
            if cold:
                flush.fill_(1)
            cstamps.zero_()
            gc.replay()
            torch.cuda.synchronize()
            st = cstamps.cpu().numpy().reshape(reps, 4096, 8).astype(np.float64)
            t00 = st[0][st[0][:, 0] > 0][:, 0].min()
            print(f'{n}x{h}x{w} {c}->{c} CHAINED, {"HBM-cold" if cold else "warm"}: us since the launch\'s first entry (min / median / max over roles); error word {int(cws.view(torch.int32)[-32].item())}')
            prev = None
            for l in range(reps):
                s_ = st[l]
                s_ = s_[s_[:, 0] > 0]
                r = (s_[:, :7] - t00) / 100.0
                q = lambda v: f'{v.min():6.2f}/{np.median(v):6.2f}/{v.max():6.2f}'   # noqa: E731
                d = '' if prev is None else f'  layer time (last signal to last signal) {r[:, 6].max() - prev:5.2f}'
                print(f'  layer {l:2d} ({len(s_)} roles): entry {q(r[:, 0])}  dep seen {q(r[:, 1])}  staged {q(r[:, 2])}  mfma {q(r[:, 3])}  '
                      f'exchanged {q(r[:, 4])}  stored {q(r[:, 5])}  signalled {q(r[:, 6])}{d}')
                prev = r[:, 6].max()
        for cold in (True, False):
            if cold:
                flush.fill_(1)
            stamps.zero_()
            g.replay()
            torch.cuda.synchronize()
            st = stamps.cpu().numpy().reshape(reps, 4096, 8).astype(np.float64)
            print(f'{n}x{h}x{w} {c}->{c}, {"HBM-cold" if cold else "warm"} replay: us relative to the launch\'s first entry (min / median / max over workgroups)')
            prev_end = None
            for l in range(reps):
                s = st[l]
                live = s[:, 0] > 0
                s = s[live]
                t0 = s[:, 0].min()
                r = (s[:, :7] - t0) / 100.0
                q = lambda v: f'{v.min():5.2f}/{np.median(v):5.2f}/{v.max():5.2f}'   # noqa: E731
                gap = '' if prev_end is None else f'  gap after previous launch\'s last drain {(t0 - prev_end) / 100.0:5.2f}'
                print(f'  layer {l:2d} ({live.sum()} wgs): entry {q(r[:, 0])}  dep {q(r[:, 1])}  staged {q(r[:, 2])}  mfma {q(r[:, 3])}  '
                      f'exchanged {q(r[:, 4])}  stored {q(r[:, 5])}  drained {q(r[:, 6])}{gap}')
                prev_end = s[:, 6].max()


if __name__ == '__main__':
    main()
