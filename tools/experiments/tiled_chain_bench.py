#!/usr/bin/env python3
"""og_conv3x3_tiled_chain_* against the same layers as separate og_conv3x3_tiled_* launches: two residual blocks (conv1 -> conv2 +
skip, twice = four dependent layers) at the shapes of the large levels, results compared bit for bit, us per group of four layers
(back-to-back groups on rotating inputs, median of rounds)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offsetguided_amd import _lib  # noqa: E402

SHAPES = [(8, 160, 160, 256), (8, 80, 80, 256), (8, 80, 80, 384), (8, 40, 40, 384)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=6)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--layers', type=int, default=4)
    ap.add_argument('--dtype', choices=['bf16', 'f16'], default='f16')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = _lib.load()
    dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float16
    cl = torch.channels_last
    fn = _lib.lp(lib, 'og_conv3x3_tiled', dt)
    ws = torch.zeros(int(lib.og_conv3x3_tiled_chain_workspace_bytes()), dtype=torch.uint8, device=dev)
    for n, h, w, c in SHAPES:
        torch.manual_seed(0)
        xs = [torch.randn(n, c, h, w, device=dev).to(dt).contiguous(memory_format=cl) for _ in range(3)]
        wts, bias = [], []
        for _ in range(a.layers):
            wt = (torch.randn(c, c, 3, 3, device=dev) * (1.0 / (9 * c)) ** 0.5).to(dt).contiguous(memory_format=cl)
            packed = torch.empty(wt.numel(), dtype=dt, device=dev)
            _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wt), c, c, 0, _lib.ptr(packed), _lib.stream_ptr(dev)), lib)
            wts.append(packed)
            bias.append(torch.randn(c, device=dev) * 0.1)
        bufs = [[torch.empty_like(xs[0]) for _ in range(a.layers)] for _ in range(2)]     # [separate | chained] outputs per layer

        def layer_io(i, which):       # layer l: input, skip (odd layers close a residual: skip = the block's input), output
            out, cur, blk_in = [], xs[i % 3], xs[i % 3]
            for l in range(a.layers):
                skip = blk_in if l % 2 == 1 else None
                out.append((cur, skip, bufs[which][l]))
                cur = bufs[which][l]
                if l % 2 == 1:
                    blk_in = cur
            return out

        def separate(i):
            for l, (x, skip, y) in enumerate(layer_io(i, 0)):
                _lib.check(fn(_lib.ptr(x), _lib.ptr(wts[l]), _lib.ptr(bias[l]), _lib.ptr(skip) if skip is not None else None, _lib.ptr(y),
                              n, h, w, c, c, 1, None, 0, _lib.stream_ptr(dev)), lib)

        def chained(i):
            descs = [_lib.TiledLayerDesc(_lib.ptr(x), _lib.ptr(wts[l]), _lib.ptr(bias[l]), _lib.ptr(skip) if skip is not None else None,
                                         _lib.ptr(y), None, n, h, w, c, c, 1) for l, (x, skip, y) in enumerate(layer_io(i, 1))]
            _lib.tiled_chain(descs, dt, ws, dev)

        if not _lib.tiled_chain_supported([_lib.TiledLayerDesc(_lib.ptr(x), _lib.ptr(wts[l]), _lib.ptr(bias[l]), None, _lib.ptr(y), None, n, h, w, c, c, 1)
                                           for l, (x, _, y) in enumerate(layer_io(0, 1))]):
            print(f'{n}x{h}x{w} {c}: not served')
            continue
        ok = True
        for i in range(3):
            separate(i)
            chained(i)
            torch.cuda.synchronize()
            ok = ok and all(torch.equal(p, q) for p, q in zip(bufs[0], bufs[1]))
        clean = int(ws.view(torch.int32).abs().sum().item()) == 0

        def timed(f):
            ts = []
            for _ in range(a.rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(a.reps):
                    f(i)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / a.reps)
            return sorted(ts)[len(ts) // 2]
        t_sep, t_ch = timed(separate), timed(chained)
        t_sep2, t_ch2 = timed(separate), timed(chained)
        print(f'{n}x{h}x{w} {c}->{c} x {a.layers} layers: separate {t_sep:7.1f} / {t_sep2:7.1f} us   chained {t_ch:7.1f} / {t_ch2:7.1f} us   '
              f'({100 * (min(t_ch, t_ch2) / min(t_sep, t_sep2) - 1):+.1f} %)   identical {ok}   workspace back to zero {clean}', flush=True)


if __name__ == '__main__':
    main()
