#!/usr/bin/env python3
"""Why the decoder's kernels take 2.6-2.9x longer beside the other batch's convolutions (bench.py --inflight 2: stage_us K1-fused 125 us, K3
189 us against 48 / 65 us alone): HIP-event durations AND in-kernel wall-clock stamps of the same launches, alone and while a second
engine replays its forward on another stream.

  event window  = what bench.py's stage_us reports (launch-stream events around the C call)
  kernel window = first workgroup's entry -> last workgroup's exit (s_memrealtime inside the kernels, 100 MHz)
  workgroup life = exit - entry of one workgroup
  event - kernel = time the launch waits for CU slots / behind the dispatcher, outside any of its workgroups

Needs a library built with -DOG_K1_STAMPS (nms_topk.hip) and -DOG_K3_STAMPS (group.hip): tools/r06_run2.sh builds it.
  OG_DECODER_LIB=tools/build/libog_stamps.so python tools/decoder_contention.py"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
import bench  # noqa: E402
from offsetguided_amd import _lib, decoder, models, synth  # noqa: E402


def stamps(lib):
    b = np.zeros(2048 * 8, np.int64)
    lib.og_k1_band_stamps(b.ctypes.data)
    k = np.zeros(128, np.uint64)
    lib.og_k3_wall_stamps(k.ctypes.data)
    return b.reshape(2048, 8).astype(np.float64) / 100.0, k.reshape(64, 2).astype(np.float64) / 100.0


def main():
    dev = torch.device('cuda:0')
    lib = _lib.load()
    for name in ('og_k1_band_stamps', 'og_k3_wall_stamps'):
        getattr(lib, name).argtypes = [C.c_void_p]
    p = argparse.ArgumentParser()
    models.net_cli(p)
    decoder.decoder_cli(p)
    margs = p.parse_args(['--no-pretrain', '--topk', '32', '--thre-hmp', '0.04', '--person-thre', '0.04', '--dist-max', '40'])
    margs.batch_size = 8
    model, _ = models.model_factory(margs)
    bench.bench_init(model, 1234)
    engines = [models.InferenceEngine(model, 8, 640, 640, device=dev) for _ in range(2)]
    proc = decoder.decoder_factory(margs)
    lanes = _lib.lane_streams(dev, 2)
    img = torch.randn(8, 3, 640, 640, device=dev)
    hm, off = synth.synth_batch(0, 8, 640, 640)
    hm_o, off_o = engines[0].forward_raw(img)
    fixed = [([None, (hm_o + torch.from_numpy(hm).to(dev)).contiguous()], [[], []], [[], []]),
             ([None, (off_o + torch.from_numpy(off).to(dev)).contiguous()], [[], []], [[], []])]
    torch.cuda.synchronize()
    n_band, n_merge = None, 8 * 19

    def one(delay_ms, beside):
        """one decoder call on lane 0; beside: three forwards of the second engine queued on lane 1 first, the decoder `delay_ms` later"""
        torch.cuda.synchronize()
        if beside:
            with torch.cuda.stream(lanes[1]):
                for _ in range(3):
                    engines[1].forward_raw(img)
        _lib.profile_start()
        with torch.cuda.stream(lanes[0]):
            if delay_ms:
                torch.cuda._sleep(int(delay_ms * 2.0e6))          # ~cycles at 2 GHz: where in the other forward the decoder lands
            h = proc.submit(fixed)
        h.result()
        ev = _lib.profile_stop()
        b, k = stamps(lib)
        return ev, b, k

    rows = {False: [], True: []}
    for beside in (False, True):
        for i in range(4):
            one(0.3 * i, beside)
        for i in range(24):
            ev, b, k = one(0.05 + 0.22 * i if beside else 0.0, beside)
            nb = int((b[:1100, 0] > 0).sum()) if n_band is None else n_band
            n_band = nb
            band, mrg, k3 = b[:nb], b[1100:1100 + n_merge], k[:8]
            rows[beside].append(dict(
                k1f_event=float(np.mean(ev['k1f_fused_limbs'])), k3_event=float(np.mean(ev['k3_group'])),
                band_window=band[:, 5].max() - band[:, 0].min(), band_life=float(np.median(band[:, 5] - band[:, 0])),
                band_entry_spread=float(np.percentile(band[:, 0], 99) - band[:, 0].min()),
                k1f_window=mrg[:, 2].max() - band[:, 0].min(), merge_life=float(np.median(mrg[:, 2] - mrg[:, 0])),
                gap=mrg[:, 0].min() - band[:, 5].max(),
                k3_window=k3[:, 1].max() - k3[:, 0].min(), k3_life=float(np.median(k3[:, 1] - k3[:, 0])),
                k3_entry_spread=float(k3[:, 0].max() - k3[:, 0].min())))
    keys = list(rows[False][0])
    print(f'{n_band} band workgroups, {n_merge} merge workgroups, 8 K3 workgroups; microseconds, median [min .. max] over 24 calls')
    for beside in (False, True):
        print('== decoder alone ==' if not beside else '== beside the other batch\'s forward (second engine replaying on another stream) ==')
        for kk in keys:
            v = np.array([r[kk] for r in rows[beside]])
            print(f'   {kk:18s} {np.median(v):7.1f}  [{v.min():7.1f} .. {v.max():7.1f}]')
    a, bz = rows[False], rows[True]
    med = lambda rs, kk: float(np.median([r[kk] for r in rs]))   # noqa: E731
    print('summary: K1-fused event %.0f -> %.0f us; in-kernel window %.0f -> %.0f us; band workgroup life %.1f -> %.1f us' %
          (med(a, 'k1f_event'), med(bz, 'k1f_event'), med(a, 'k1f_window'), med(bz, 'k1f_window'), med(a, 'band_life'), med(bz, 'band_life')))
    print('         K3 event %.0f -> %.0f us; in-kernel window %.0f -> %.0f us; workgroup life %.1f -> %.1f us; entry spread %.1f -> %.1f us' %
          (med(a, 'k3_event'), med(bz, 'k3_event'), med(a, 'k3_window'), med(bz, 'k3_window'), med(a, 'k3_life'), med(bz, 'k3_life'),
           med(a, 'k3_entry_spread'), med(bz, 'k3_entry_spread')))


if __name__ == '__main__':
    main()
