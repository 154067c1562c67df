#!/usr/bin/env python3
"""tests/golden/bilinear256*.npz: the IMPORTED reference decoder run with `--resize-mode bilinear`
(decoder/factory.py:151-153 -> PostProcess.inter_mode: heatmaps and keypoint-scale maps are resized x4 bilinearly,
:74-75 and :80-82), with and without the keypoint-scale head (scales_mode 3 = scale head on, use_scale True) and flip-test.
The oracle is checked against the reference output here; the fixtures hold seeds, input hashes and expected poses.

Build container only (needs /root/reference).   usage: PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_bilinear.py
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import FLAGS, GOLD, check_poses, load_reference, sha  # noqa: E402

import oracle  # noqa: E402
from offsetguided_amd import synth  # noqa: E402
from offsetguided_amd.config.coco_data import (COCO_KEYPOINTS, COCO_PERSON_SKELETON,  # noqa: E402
                                               heatmap_hflip, offset_hflip)


def case(decoder, name, seed, batch, size, flip, with_scale):
    hm, off = synth.synth_batch(seed, batch, size, size, flip=flip, n_persons=8)
    nb = hm.shape[0]
    scl = (synth.noise_batch(seed + 5, (nb, 17, size // 4, size // 4)) * 20 + 25).astype(np.float32) if with_scale else None
    p = argparse.ArgumentParser()
    decoder.decoder_cli(p)
    dist_max = 6.0 if with_scale else FLAGS['dist_max']
    a = p.parse_args(['--resize-mode', 'bilinear', '--topk', str(FLAGS['topk']), '--thre-hmp', str(FLAGS['thre_hmp']),
                      '--person-thre', str(FLAGS['person_thre']), '--dist-max', str(dist_max), '--min-len', str(FLAGS['min_len']),
                      '--use-scale', str(bool(with_scale))])
    a.headnets, a.strides, a.batch_size = ['hmp', 'omp'], [4, 4], batch
    a.include_scale, a.include_jitter_offset = bool(with_scale), False
    proc = decoder.decoder_factory(a)
    assert proc.inter_mode == 'bilinear'
    t = torch.from_numpy
    sc_feat = [t(scl) * 0, t(scl)] if with_scale else [[], []]
    feats = [([t(hm) * 0, t(hm)], [[], []], [[], []]), ([t(off) * 0, t(off)], [[], []], sc_feat)]
    poses = proc.generate_poses(feats, flip_test=flip)
    proc.worker_pool.close()
    fl = None
    if flip:
        perm, rev = offset_hflip(COCO_KEYPOINTS, COCO_PERSON_SKELETON)
        fl = (heatmap_hflip(COCO_KEYPOINTS), perm, rev)
    o_poses, _ = oracle.decode(hm, off, COCO_PERSON_SKELETON, topk_k=FLAGS['topk'], thre_hmp=FLAGS['thre_hmp'],
                               min_len=FLAGS['min_len'], person_thre=FLAGS['person_thre'], dist_max=dist_max,
                               use_scale=bool(with_scale), flip=fl, scales_lr=scl, inter_mode='bilinear')
    dp = check_poses(poses, o_poses, name)
    if with_scale:
        for r, m in zip(poses, o_poses):
            assert (r[..., 3] == m[..., 3]).all()
    shas = [sha(hm), sha(off)] + ([sha(scl)] if with_scale else [])
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), seed=seed, batch=batch, size=size, flip=int(flip),
                        with_scale=int(with_scale), dist_max=dist_max, in_sha=np.array(shas),
                        n_poses=np.array([len(q) for q in poses]),
                        poses=np.concatenate(poses, 0) if sum(len(q) for q in poses) else np.zeros((0, 17, 6), np.float32))
    print(f'{name}: --resize-mode bilinear, poses/img {[len(q) for q in poses]}, pose ls err {dp:.2e}')


def main():
    decoder = load_reference()
    case(decoder, 'bilinear256', 71, 2, 256, False, False)
    case(decoder, 'bilinear256_flip', 72, 2, 256, True, False)
    case(decoder, 'bilinear256_scale', 73, 2, 256, False, True)
    case(decoder, 'bilinear256_scale_flip', 74, 2, 256, True, True)


if __name__ == '__main__':
    main()
