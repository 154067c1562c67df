#!/usr/bin/env python3
"""og_conv3x3_bf16 vs MIOpen conv + og_bias_act_bf16 on the hourglass's 3x3 shapes: numerics against an fp32
torch convolution and GPU time per call (both captured in HIP graphs of `reps` calls)."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offsetguided_amd import _lib  # noqa: E402

SHAPES = [(8, 5, 5, 512, 512), (8, 10, 10, 384, 384), (8, 20, 20, 384, 384), (8, 40, 40, 384, 384),
          (8, 5, 5, 384, 512), (8, 40, 40, 256, 384), (8, 80, 80, 256, 256), (16, 20, 20, 384, 384), (16, 5, 5, 512, 512),
          (8, 160, 160, 256, 256), (16, 160, 160, 256, 256), (16, 80, 80, 256, 256)]


# the hourglass's other convolutions at bs8 640x640: (N, Hin, Win, Cin, Cout, ksize, stride)
SHAPES2 = [(8, 320, 320, 128, 256, 3, 2), (8, 320, 320, 128, 256, 1, 2), (8, 160, 160, 256, 256, 1, 1),
           (8, 160, 160, 256, 256, 3, 2), (8, 160, 160, 256, 256, 1, 2), (8, 80, 80, 256, 384, 3, 2),
           (8, 80, 80, 256, 384, 1, 2), (8, 40, 40, 384, 256, 1, 1), (8, 40, 40, 384, 384, 3, 2),
           (8, 40, 40, 384, 384, 1, 2), (8, 20, 20, 384, 384, 3, 2), (8, 20, 20, 384, 384, 1, 2),
           (8, 10, 10, 384, 512, 3, 2), (8, 10, 10, 384, 512, 1, 2), (8, 5, 5, 512, 384, 1, 1),
           (8, 160, 160, 256, 64, 1, 1)]

_flush = None


def graph_time(fn, reps, rounds=5, cold=True):
    """us per call of fn(l), l = 0..reps-1 (distinct weights/activations per l), captured in one graph.  cold: every
    timed replay starts behind a 1 GiB fill, so the weights come from HBM as they do inside the network."""
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
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--only', type=int, nargs='*', default=[])
    ap.add_argument('--stamps', action='store_true', help='print in-kernel timeline (us) of one cold launch per plan')
    ap.add_argument('--warm', action='store_true', help='no cache flush between timed replays (weights stay in the Infinity Cache)')
    ap.add_argument('--conv2d', action='store_true', help='og_conv2d_bf16 on the 1x1 / stride-2 shapes instead')
    a = ap.parse_args()
    if a.conv2d:
        return main_conv2d(a)
    dev = torch.device('cuda:0')
    lib = _lib.load()
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(0)
    for si, (n, h, w, cin, cout) in enumerate(SHAPES):
        if a.only and si not in a.only:
            continue
        cl = torch.channels_last
        xs = [torch.randn(n, cin, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=cl) for _ in range(a.reps)]
        wts = [(torch.randn(cout, cin, 3, 3, device=dev) * (1.0 / (9 * cin)) ** 0.5).to(torch.bfloat16)
               .contiguous(memory_format=cl) for _ in range(a.reps)]
        bias = torch.randn(cout, device=dev) * 0.1
        skip = torch.randn(n, cout, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=cl)
        ref = F.relu(F.conv2d(xs[-1].float(), wts[-1].float(), bias, 1, 1) + skip.float())
        out = torch.empty_like(skip)
        need = lib.og_conv3x3_workspace_bytes(n * h * w, cin, cout)
        ws = torch.zeros(max(need, 256) * 8, dtype=torch.uint8, device=dev)  # room for any split override

        def ours(l=-1):
            _lib.check(lib.og_conv3x3_bf16(_lib.ptr(xs[l]), _lib.ptr(wts[l]), _lib.ptr(bias), _lib.ptr(skip), _lib.ptr(out),
                                           n, h, w, cin, cout, 1, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), lib)

        def miopen(l=-1):
            y = F.conv2d(xs[l], wts[l], None, 1, 1)
            _lib.check(lib.og_bias_act_bf16(_lib.ptr(y), _lib.ptr(bias), _lib.ptr(skip), n * h * w, cout, 1,
                                            _lib.stream_ptr(dev)), lib)
            return y

        t_ref = graph_time(miopen, a.reps, cold=not a.warm)
        gflop = 2 * n * h * w * cout * 9 * cin / 1e9
        line = f'{n}x{h}x{w} {cin}->{cout} ({gflop:.2f} GF): miopen+epilogue {t_ref:7.1f} us'
        for plan in ['']:      # (the OG_CONV_PLAN arms of rounds 1-3 left the library: the plan is the library's own)
            out.zero_()
            try:
                ours()
            except Exception as ex:  # noqa: BLE001
                line += f' | {plan or "auto"}: {str(ex)[:40]}'
                continue
            torch.cuda.synchronize()
            err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
            if a.stamps:
                st = torch.zeros(32768 * 8, dtype=torch.int64, device=dev)
                _flush.fill_(1)
                torch.cuda.synchronize()
                lib.og_conv3x3_debug_stamps(_lib.ptr(st))
                for _ in range(8):   # back to back: the last launch (its marks stay) runs at the clock the chip holds under this load
                    ours(0)
                torch.cuda.synchronize()
                lib.og_conv3x3_debug_stamps(None)
                v = st.view(-1, 8).cpu().numpy()
                later = np.nonzero((v[:, 0] == 0) & (v[:, 2] > 0) & (v[:, 5] > 0))[0]   # halo kernel: items after a workgroup's first
                if len(later):
                    prev_end = v[later - 1, 5]
                    print('   later items of a workgroup (us, median): hand-over (previous stores issued -> loop) %.2f | loop %.2f | epilogue %.2f   (n=%d)'
                          % (np.median((v[later, 2] - prev_end) / 100.0), np.median((v[later, 3] - v[later, 2]) / 100.0),
                             np.median((v[later, 5] - v[later, 3]) / 100.0), len(later)))
                if os.environ.get('OG_EPI_STAMPS') and len(later):
                    e = v[later]
                    seq = np.stack([e[:, 4] - e[:, 3], e[:, 6] - e[:, 4], e[:, 7] - e[:, 6], e[:, 5] - e[:, 7]], 1) / 100.0
                    print('   hand-over epilogue (us, median): loop end -> half 0 staged %.2f | stores 0 issued %.2f | half 1 staged %.2f | stores 1 issued %.2f' % tuple(np.median(seq, 0)))
                    v[:, 6:8] = 0
                v = v[v[:, 0] > 0]
                t0 = v[:, 0].min()
                rel = (v - t0) / 100.0
                d = (v[:, [1, 2, 3, 5]] - v[:, [0, 1, 2, 3]]) / 100.0
                ok = (v[:, 5] > 0) & (v[:, 3] > 0)
                if (v[ok, 7] > 0).any():   # halo kernel: shader-clock counter beside the loop marks
                    ghz = (v[ok, 7] - v[ok, 6]) / ((v[ok, 3] - v[ok, 2]) * 10.0)
                    print('   shader clock over the main loop: median %.2f GHz (min %.2f, max %.2f)' % (np.median(ghz), ghz.min(), ghz.max()))
                print('   per-workgroup phases (us, median): setup+issue %.2f | first data %.2f | loop %.2f | epilogue %.2f | total %.2f; kernel span %.1f'
                      % (*np.median(d[ok], 0), np.median((v[ok, 5] - v[ok, 0]) / 100.0), (v[ok, 5].max() - t0) / 100.0))
                names = ['start', 'issued', 'first data', 'loop end', 'ticket', 'end(last)']
                print(f'   [{plan or "auto"}] {len(v)} workgroups; us since first start (median / max):')
                for i, nm in enumerate(names):
                    col = rel[:, i][v[:, i] > 0]
                    if len(col):
                        print(f'      {nm:11s} {np.median(col):7.2f} {col.max():7.2f}   (n={len(col)})')
            t = graph_time(ours, a.reps, cold=not a.warm)
            line += f' | {plan or "auto"}: {t:6.1f} us ({gflop / t * 1e-3:5.0f} TF) err {err:.1e}'
        print(line, flush=True)


def main_conv2d(a):
    dev = torch.device('cuda:0')
    lib = _lib.load()
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(0)
    cl = torch.channels_last
    for si, (n, h, w, cin, cout, k, st) in enumerate(SHAPES2):
        if a.only and si not in a.only:
            continue
        reps = a.reps if n * h * w * cin < (1 << 24) else max(3, a.reps // 4)
        xs = [torch.randn(n, cin, h, w, device=dev).to(torch.bfloat16).contiguous(memory_format=cl) for _ in range(reps)]
        wts = [(torch.randn(cout, cin, k, k, device=dev) * (1.0 / (k * k * cin)) ** 0.5).to(torch.bfloat16)
               .contiguous(memory_format=cl) for _ in range(reps)]
        bias = torch.randn(cout, device=dev) * 0.1
        ref = F.relu(F.conv2d(xs[-1].float(), wts[-1].float(), bias, st, k // 2))
        ho, wo = ref.shape[2:]
        out = torch.empty((n, cout, ho, wo), dtype=torch.bfloat16, device=dev).contiguous(memory_format=cl)
        need = lib.og_conv2d_workspace_bytes(n, h, w, cin, cout, k, st)
        ws = torch.zeros(max(need, 256) * 8 + (1 << 20), dtype=torch.uint8, device=dev)

        def ours(l=-1):
            _lib.check(lib.og_conv2d_bf16(_lib.ptr(xs[l]), _lib.ptr(wts[l]), _lib.ptr(bias), None, _lib.ptr(out),
                                          n, h, w, cin, cout, k, st, 1, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), lib)

        def miopen_raw(l=-1):
            return F.conv2d(xs[l], wts[l], None, st, k // 2)

        def miopen(l=-1):
            y = F.conv2d(xs[l], wts[l], None, st, k // 2)
            _lib.check(lib.og_bias_act_bf16(_lib.ptr(y), _lib.ptr(bias), None, n * ho * wo, cout, 1,
                                            _lib.stream_ptr(dev)), lib)
            return y

        t_raw = graph_time(miopen_raw, reps)
        t_ref = graph_time(miopen, reps)
        gflop = 2 * n * ho * wo * cout * k * k * cin / 1e9
        line = f'[{si}] {n}x{h}x{w} {cin}->{cout} k{k} s{st} ({gflop:.2f} GF): miopen {t_raw:7.1f} +epilogue {t_ref:7.1f} us'
        for plan in ['']:      # (the OG_CONV_PLAN arms of rounds 1-3 left the library: the plan is the library's own)
            out.zero_()
            try:
                ours()
            except Exception as ex:  # noqa: BLE001
                line += f' | {plan or "auto"}: {str(ex)[-50:]}'
                continue
            torch.cuda.synchronize()
            err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
            t = graph_time(ours, reps)
            line += f' | {plan or "auto"}: {t:6.1f} us ({gflop / t * 1e-3:5.0f} TF) err {err:.1e}'
        print(line, flush=True)


if __name__ == '__main__':
    main()
