#!/usr/bin/env python3
"""og_stem7x7_bf16 on the bs8 640x640 batch, several library builds in one process (tools/build_variants.sh stem.hip ...):
HIP events around `reps` back-to-back launches on rotating inputs."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offsetguided_amd import _lib  # noqa: E402
from tools.k1_bench import load  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    _lib.load()
    n, h, w = 8, 640, 640
    xs = [torch.randn(n, 3, h, w, device=dev) for _ in range(3)]
    packed = (torch.randn(128, 7, 8, 4, device=dev) * 0.08).to(torch.bfloat16).contiguous()
    bias = torch.randn(128, device=dev) * 0.1
    outs = [torch.empty((n, 128, h // 2, w // 2), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(3)]
    st = _lib.stream_ptr(dev)
    for path in sys.argv[1:] or [_lib.LIB_PATH]:
        lib = load(path)
        def run(i):
            _lib.check(lib.og_stem7x7_bf16(_lib.ptr(xs[i % 3]), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(outs[i % 3]), n, h, w, 1, st), lib)
        for i in range(6):
            run(i)
        torch.cuda.synchronize()
        ts = []
        for r in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(12):
                run(i)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) * 1e3 / 12)
        ts.sort()
        print(f'{os.path.basename(path):28s} median {ts[3]:6.1f} us  min {ts[0]:6.1f} us')


if __name__ == '__main__':
    main()
