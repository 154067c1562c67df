#!/usr/bin/env python3
"""Do HIP stream priorities change how two concurrent convolution streams share the chip?  (Round 6: the question behind "priorities inside the
captured forward -- not available": a graph's kernel nodes take no priority, but separate launches on prioritised streams might.)

Stream A runs the 160x160 256->256 layer back to back (the up1 branch's situation: 1 600 workgroups per launch); 150 us later stream B runs ten
80x80 256->256 layers (the trunk's situation below the fork: 400 workgroups per launch).  B's ten layers are timed with events on B, A's whole run
on A, for B's priority in {normal, high} and A's in {normal, low}.  Alone, a 80x80 layer takes ~53 us."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from offsetguided_amd import _lib  # noqa: E402


def make_stream(dev, prio):
    hip = _lib._hip_runtime()
    h = C.c_void_p()
    rc = hip.hipStreamCreateWithPriority(C.byref(h), C.c_uint(1), C.c_int(prio))
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device=dev)


def main():
    dev = torch.device('cuda:0')
    lib = _lib.load()
    hip = _lib._hip_runtime()
    lo, hi = C.c_int(), C.c_int()
    hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
    print('priority range: least', lo.value, 'greatest', hi.value)
    cl, lp = torch.channels_last, torch.float16

    def layer(hw):
        xs = [torch.randn(8, 256, hw, hw, device=dev).to(lp).contiguous(memory_format=cl) for _ in range(3)]
        outs = [torch.empty_like(xs[0]) for _ in range(3)]
        wt = (torch.randn(256, 256, 3, 3, device=dev) * (1.0 / 2304) ** 0.5).to(lp).contiguous(memory_format=cl)
        packed = torch.empty(wt.numel(), dtype=lp, device=dev)
        _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wt), 256, 256, 0, _lib.ptr(packed), _lib.stream_ptr(dev)), lib)
        cb = torch.zeros(256, device=dev)

        def run(i, stream):
            _lib.check(lib.og_conv3x3_tiled_f16(_lib.ptr(xs[i % 3]), _lib.ptr(packed), _lib.ptr(cb), None, _lib.ptr(outs[i % 3]), 8, hw, hw, 256, 256, 1,
                                                None, 0, C.c_void_p(stream.cuda_stream)), lib)
        return run

    big, small = layer(160), layer(80)
    torch.cuda.synchronize()
    for pa, pb in ((0, 0), (0, hi.value), (lo.value, 0), (lo.value, hi.value), (0, 0), (lo.value, hi.value)):
        sa, sb = make_stream(dev, pa), make_stream(dev, pb)
        res = []
        for rep in range(6):
            torch.cuda.synchronize()
            a0, a1, b0, b1 = (_lib.TimingEvent() for _ in range(4))
            a0.record(sa)
            for i in range(8):
                big(i, sa)
            a1.record(sa)
            with torch.cuda.stream(sb):
                torch.cuda._sleep(300_000)          # ~150 us: B starts when A's first layer is under way
            b0.record(sb)
            for i in range(10):
                small(i, sb)
            b1.record(sb)
            torch.cuda.synchronize()
            res.append((b0.elapsed_time(b1) * 1e3 / 10, a0.elapsed_time(a1) * 1e3))
        r = np.array(res[1:])
        print(f'A priority {pa:2d}, B priority {pb:2d}:  B (80x80) {np.median(r[:, 0]):6.1f} us per layer beside A   A (8 x 160x160) {np.median(r[:, 1]):7.1f} us in all', flush=True)
    sb = make_stream(dev, 0)
    torch.cuda.synchronize()
    b0, b1 = _lib.TimingEvent(), _lib.TimingEvent()
    for i in range(20):
        small(i, sb)
    b0.record(sb)
    for i in range(10):
        small(i, sb)
    b1.record(sb)
    torch.cuda.synchronize()
    print(f'B alone: {b0.elapsed_time(b1) * 1e3 / 10:6.1f} us per layer')


if __name__ == '__main__':
    main()
