#!/usr/bin/env python3
"""tools/k1_band_stamps.py inside the decode pipeline: backbone (HIP graph) -> heads + synthetic maps -> K1a -> K1 -> K3,
software-pipelined as bench.py does; the stamps of the LAST K1 launches are reported.

  OG_DECODER_LIB=$PWD/tools/build/libog_stamps.so python tools/k1_pipeline_stamps.py
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')
from offsetguided_amd import _lib, decoder, models, synth  # noqa: E402
from tools.k1_band_stamps import report  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    lib = _lib.load()
    lib.og_k1_band_stamps.argtypes = [C.c_void_p]
    p = argparse.ArgumentParser()
    models.net_cli(p)
    decoder.decoder_cli(p)
    margs = p.parse_args(['--no-pretrain', '--topk', '32', '--thre-hmp', '0.04', '--person-thre', '0.04', '--dist-max', '40'])
    margs.batch_size = 8
    model, _ = models.model_factory(margs)
    bench.bench_init(model, 1234)
    engine = models.InferenceEngine(model, 8, 640, 640, dtype=torch.bfloat16, device=dev, use_graph=os.environ.get('OG_STAMPS_NO_GRAPH') != '1')
    proc = decoder.decoder_factory(margs)
    images = [torch.randn(8, 3, 640, 640, device=dev) for _ in range(3)]
    maps = []
    for r in range(3):
        hm, off = synth.synth_batch(r, 8, 640, 640)
        maps.append((torch.from_numpy(hm).to(dev), torch.from_numpy(off).to(dev)))
    # OG_STAMPS_WARM=1/2: a dummy K1 on other buffers right before the real decode (1: the decoder's own workspace is not
    # touched, 2: neither are the offsets) -- separates cold code / descriptors from cold data pages
    warm = int(os.environ.get('OG_STAMPS_WARM', '0'))
    if warm:
        from offsetguided_amd.config import coco_data as cd
        sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        d_hr = torch.randn(8, 17, 640, 640, device=dev).abs_()
        d_off = torch.randn(8, 38, 160, 160, device=dev)
        d_ws = torch.zeros(lib.og_generate_limbs_workspace_bytes(8, 17, 640, 640, 32), dtype=torch.uint8, device=dev)
        d_limbs = torch.empty((8, 19, 32, 13), device=dev)
        jf = _lib.int_table([x for x, _ in cd.COCO_PERSON_SKELETON], dev)
        jt = _lib.int_table([y for _, y in cd.COCO_PERSON_SKELETON], dev)
    pending = None
    sync_each = os.environ.get('OG_STAMPS_SYNC_EACH') == '1'
    # OG_STAMPS_PARTS: e = engine forward, a = the "+ maps" additions, k = K3 + D2H (PostProcess.submit); default all
    parts = os.environ.get('OG_STAMPS_PARTS', 'eak')
    hm_o, off_o = engine.forward_raw(images[0])
    # without "a": fixed inputs -- the synthetic maps alone, or (c in parts) with the head outputs added once
    fixed = [[([None, maps[r][0] + (hm_o if 'c' in parts else 0)], [[], []], [[], []]), ([None, maps[r][1] + (off_o if 'c' in parts else 0)], [[], []], [[], []])] for r in range(3)]
    print('head outputs: |hm| max', float(hm_o.abs().max()), 'mean', float(hm_o.abs().mean()), ' |off| max', float(off_o.abs().max()))
    for i in range(12):
        if 'e' in parts:
            hm_o, off_o = engine.forward_raw(images[i % 3])
        if 'a' in parts:
            feats = [([None, hm_o + maps[i % 3][0]], [[], []], [[], []]), ([None, off_o + maps[i % 3][1]], [[], []], [[], []])]
        else:
            feats = fixed[i % 3]
        if sync_each:
            torch.cuda.synchronize()
        if warm == 3:   # touch every 4 KiB page of the decoder's own K1 workspace (address translations only)
            wsr = _lib.workspace(dev, lib.og_generate_limbs_workspace_bytes(8, 17, 640, 640, 32), 'limbs')
            touch = wsr[::4096].sum()
        elif warm:
            _lib.check(lib.og_generate_limbs_f32(_lib.ptr(d_hr), _lib.ptr(d_off), 1, 2, None, 0, None, 0, 8, 17, 640, 640,
                                                 _lib.ptr(jf), _lib.ptr(jt), 19, 32, 0.04, 0.5, 1.0, None, None, _lib.ptr(d_limbs),
                                                 0, _lib.ptr(d_ws), d_ws.numel(), sp), lib)
        if 'k' in parts:
            nxt = proc.submit(feats, flip_test=False)
            if pending is not None:
                pending.result()
            pending = nxt
        else:
            limbs = proc.generate_limbs(feats, flip_test=False)
        if i >= 8:
            torch.cuda.synchronize()
            buf = np.zeros(2048 * 8, np.int64)
            lib.og_k1_band_stamps(buf.ctypes.data)
            report(buf, 8 * 19, f'pipeline step {i}:')
            sys.stdout.flush()
    if pending is not None:
        pending.result()


if __name__ == '__main__':
    main()
