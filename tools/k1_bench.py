#!/usr/bin/env python3
"""K1 at the generate_limbs boundary (og_generate_limbs_f32 vs og_nms_topk_f32 + og_collect_limbs_full_f32) on the bs8
640x640 synthetic batch: HIP events on the launch stream, several library builds in ONE process (same box, same clocks).

  python tools/k1_bench.py [--libs a.so b.so ...] [--iters 40]

For every library: (1) results of og_generate_limbs_f32 == the separate entry points (bitwise), (2) HBM-cold timing (3 rotating
hi-res batches, 669 MB > Infinity Cache), (3) timing directly behind K1a (the hi-res batch has just been written: the
decode pipeline's situation), (4) the three-launch form in both situations."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from offsetguided_amd import _lib, synth  # noqa: E402
from offsetguided_amd.config import coco_data as cd  # noqa: E402


def load(path):
    lib = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


def bench_inputs(dev, n_rot=3, batch=8, size=640):
    """The decoder inputs of bench.py's timed region: head outputs of the random-init network (bench_init) on random images +
    the synthetic GT-like maps.  (The head outputs are NOT small -- |hm| is 0.14 on average: every background pixel lies above
    the admission threshold's starting point, which the synthetic maps alone do not reproduce; K1 takes 7-8 us longer.)"""
    import argparse as ap_
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import bench
    from offsetguided_amd import decoder, models
    p = ap_.ArgumentParser()
    models.net_cli(p)
    decoder.decoder_cli(p)
    margs = p.parse_args(['--no-pretrain', '--topk', '32', '--thre-hmp', '0.04', '--person-thre', '0.04', '--dist-max', '40'])
    margs.batch_size = batch
    model, _ = models.model_factory(margs)
    bench.bench_init(model, 1234)
    engine = models.InferenceEngine(model, batch, size, size, dtype=torch.bfloat16, device=dev, use_graph=False)
    out = []
    for r in range(n_rot):
        img = torch.randn(batch, 3, size, size, device=dev, generator=torch.Generator(dev).manual_seed(r))
        hm_o, off_o = engine.forward_raw(img)
        hm, off = synth.synth_batch(r, batch, size, size)
        out.append(((hm_o + torch.from_numpy(hm).to(dev)).contiguous(), (off_o + torch.from_numpy(off).to(dev)).contiguous()))
    del engine, model
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--bench-inputs', action='store_true', help='decoder inputs as in bench.py (network head outputs + synthetic maps) instead of the synthetic maps alone')
    ap.add_argument('--forms', nargs='*', default=['two', 'three'], help='which forms of K1 to time (PMC passes: --forms two = the default form only)')
    ap.add_argument('--libs', nargs='*', default=[_lib.LIB_PATH])
    ap.add_argument('--iters', type=int, default=40)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--k', type=int, default=32)
    ap.add_argument('--burst', type=int, default=0, help='dense bf16 GEMMs (about this many ms) queued before every timed call: the clock / power state the decoder sees behind the backbone')
    ap.add_argument('--rotate', type=int, default=3, help='hi-res batches cycled through (1 = the decode pipeline: one buffer)')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    _lib.load()
    st = torch.cuda.current_stream(dev)
    sp = C.c_void_p(st.cuda_stream)
    n, c, k, L = a.batch, 17, a.k, 19
    h = w = a.size
    if a.bench_inputs:
        bi = bench_inputs(dev, a.rotate, n, h)
        lrs = [x[0] for x in bi]
        t_off = bi[0][1]
    else:
        hm, off = synth.synth_batch(0, n, h, w)
        lrs = [torch.from_numpy(hm).to(dev) * (1.0 - 0.01 * r) for r in range(a.rotate)]
        t_off = torch.from_numpy(off).to(dev)
    hrs = [torch.empty((n, c, h, w), device=dev) for _ in range(a.rotate)]
    jf = _lib.int_table([x for x, _ in cd.COCO_PERSON_SKELETON], dev)
    jt = _lib.int_table([y for _, y in cd.COCO_PERSON_SKELETON], dev)
    nbytes = n * c * h * w * 4 + n * L * k * (8 + 52)

    for path in a.libs:
        lib = load(path)
        tag = os.path.basename(path)

        def k1a(i):
            _lib.check(lib.og_upsample_bicubic4_f32(_lib.ptr(lrs[i % a.rotate]), n * c, h // 4, w // 4, _lib.ptr(hrs[i % a.rotate]), sp), lib)

        for i in range(a.rotate):
            k1a(i)
        ws1 = torch.zeros(lib.og_generate_limbs_workspace_bytes(n, c, h, w, k), dtype=torch.uint8, device=dev)
        ws3 = torch.zeros(lib.og_topk_workspace_bytes(n * c, h, w, k), dtype=torch.uint8, device=dev)
        limbs3 = torch.empty((n, L, k, 13), device=dev)
        sc = torch.empty((n, c, k), device=dev)
        ix = torch.empty((n, c, k), dtype=torch.int64, device=dev)

        limbs2 = torch.empty((n, L, k, 13), device=dev)
        sc2 = torch.empty((n, c, k), device=dev)
        ix2 = torch.empty((n, c, k), dtype=torch.int64, device=dev)

        def two(i):   # the default form: band kernel + merge-and-pair kernel
            _lib.check(lib.og_generate_limbs_f32(_lib.ptr(hrs[i % a.rotate]), _lib.ptr(t_off), 1, 2, None, 0, None, 0, n, c, h, w,
                                                 _lib.ptr(jf), _lib.ptr(jt), L, k, 0.04, 0.5, 1.0, _lib.ptr(sc2), _lib.ptr(ix2),
                                                 _lib.ptr(limbs2), 0, _lib.ptr(ws1), ws1.numel(), sp), lib)

        def three(i):
            _lib.check(lib.og_nms_topk_f32(_lib.ptr(hrs[i % a.rotate]), n * c, h, w, k, _lib.ptr(sc), _lib.ptr(ix), _lib.ptr(ws3),
                                           ws3.numel(), sp), lib)
            _lib.check(lib.og_collect_limbs_full_f32(_lib.ptr(sc), _lib.ptr(ix), _lib.ptr(t_off), 1, 2, None, 0, None, 0, n, c, h,
                                                     w, _lib.ptr(jf), _lib.ptr(jt), L, k, 0.04, 0.5, 1.0, _lib.ptr(limbs3), sp), lib)

        ok = True
        for i in range(a.rotate if set(a.forms) >= {'two', 'three'} else 0):
            three(i)
            two(i)
            torch.cuda.synchronize()
            ok = ok and torch.equal(limbs2, limbs3) and torch.equal(sc2, sc) and torch.equal(ix2, ix)
        tick = int(ws1[:61440].view(torch.int32).abs().sum())

        def timed(fn, pre=None):
            for i in range(4):
                if pre:
                    pre(i)
                fn(i)
            torch.cuda.synchronize()
            evs = []
            for i in range(a.iters):
                if pre:
                    pre(i)
                s, e = _lib.TimingEvent(), _lib.TimingEvent()
                s.record(st)
                fn(i)
                e.record(st)
                evs.append((s, e))
            torch.cuda.synchronize()
            t = np.array([s.elapsed_time(e) for s, e in evs]) * 1e3
            return float(np.median(t)), float(t.min())

        def fused(i):   # the production form: x4 bicubic inside the band kernel, merge + pairing in one launch (no hi-res tensor)
            _lib.check(lib.og_generate_limbs_fused_f32(_lib.ptr(lrs[i % a.rotate]), _lib.ptr(t_off), 2, None, 0, None, 0, n, c, h // 4, w // 4,
                                                       _lib.ptr(jf), _lib.ptr(jt), L, k, 0.04, 0.5, 1.0, _lib.ptr(sc2), _lib.ptr(ix2),
                                                       _lib.ptr(limbs2), _lib.ptr(ws1), ws1.numel(), sp), lib)

        fns = {'two': two, 'three': three, 'fused': fused}
        res = {f'{f} cold': timed(fns[f]) for f in a.forms}
        res.update({f'{f} after K1a': timed(fns[f], k1a) for f in a.forms})
        if a.burst:
            ga = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
            gb = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ga @ gb
            e1.record()
            torch.cuda.synchronize()
            reps = max(1, int(a.burst / (e0.elapsed_time(e1) / 10)))

            def burst_then_k1a(i):
                for _ in range(reps):
                    ga @ gb
                k1a(i)
            res[f'two after {a.burst} ms of GEMM + K1a'] = timed(two, burst_then_k1a)
            res[f'three after {a.burst} ms of GEMM + K1a'] = timed(three, burst_then_k1a)
        print(f'== {tag}: two launches == three entry points: {ok}; state words zero: {tick == 0}')
        for name, (med, mn) in res.items():
            print(f'   {name:36s} median {med:7.1f} us  min {mn:7.1f} us   {nbytes / med / 1e6:5.2f} TB/s  frac {nbytes / med / 1e6 / 8:.3f}')
        sys.stdout.flush()


if __name__ == '__main__':
    main()
