#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the IMPORTED reference decoder.

Runs only in the build container (needs /root/reference; the GPU box never sees it).
For every case the C oracle (oracle/) is checked against the reference output right here
(indices / coordinates / grouping bit-exact, scores <= 1e-4), and the expected outputs are
stored as small fixtures.  Inputs come from the portable generator
offsetguided_amd/synth.py and are NOT stored; a sha256 of the input bytes is, so that a
drifting generator is reported as such instead of as a parity failure.

Reference pinning: the reference targets torch 1.3.1 where
`int64 / int` floors; decoder.heatmap.topK_channel is patched to floor-divide.

usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import argparse
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("OG_REFERENCE", "/root/reference")
GOLD = os.environ.get("OG_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")   # OG_GOLDEN_OUT: regenerate into a scratch directory

import oracle  # noqa: E402
from offsetguided_amd import synth  # noqa: E402
from offsetguided_amd.config.coco_data import (COCO_KEYPOINTS, COCO_PERSON_SKELETON,  # noqa: E402
                                               heatmap_hflip, offset_hflip)

FLAGS = dict(topk=32, thre_hmp=0.04, person_thre=0.04, dist_max=40.0, min_len=0.5)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import decoder  # noqa
    import decoder.heatmap as H

    def _topk_floor(scores, K=40):
        n, c, h, w = scores.shape
        s, i = torch.topk(scores.view(n, c, -1), K)
        return s, i, torch.div(i, w, rounding_mode='floor'), i % w

    H.topK_channel = _topk_floor
    return decoder


def ref_processor(decoder, batch, headnet='omp', **over):
    p = argparse.ArgumentParser()
    decoder.decoder_cli(p)
    f = dict(FLAGS, **over)
    a = p.parse_args(['--topk', str(f['topk']), '--thre-hmp', str(f['thre_hmp']), '--person-thre',
                      str(f['person_thre']), '--dist-max', str(f['dist_max']), '--min-len', str(f['min_len'])])
    a.headnets = ['hmp', headnet]
    a.strides = [4, 4]
    a.batch_size = batch
    a.include_scale = False
    a.include_jitter_offset = False
    return decoder.decoder_factory(a)


def features(hm, off):
    """models/networks.py:192 nesting; two stacks, only the last is consumed."""
    hm, off = torch.from_numpy(hm), torch.from_numpy(off)
    return [([hm * 0, hm], [[], []], [[], []]), ([off * 0, off], [[], []], [[], []])]


def check_limbs(ref, mine, tag):
    """idx/xy/dist/len exact; scores 1e-4."""
    exact_cols = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12]
    bad = (ref[..., exact_cols] != mine[..., exact_cols]).sum()
    ds = np.abs(ref[..., 10] - mine[..., 10]).max()
    assert bad == 0, f'{tag}: {bad} limb fields differ'
    assert ds <= 1e-4, f'{tag}: limb score err {ds}'
    return float(ds)


def check_poses(ref, mine, tag):
    assert len(ref) == len(mine), tag
    worst = 0.0
    for r, m in zip(ref, mine):
        assert r.shape == m.shape, f'{tag}: pose count {r.shape} vs {m.shape}'
        if r.size == 0:
            continue
        assert (r[..., [0, 1, 2, 3, 5]] == m[..., [0, 1, 2, 3, 5]]).all(), f'{tag}: pose fields differ'
        worst = max(worst, float(np.abs(r[..., 4] - m[..., 4]).max()))
    assert worst <= 1e-4, f'{tag}: pose limb-score err {worst}'
    return worst


def pipeline_case(decoder, name, seed, batch, size, flip, n_persons, store_stage=True, cat=False, headnet='omp', topk=None):
    """One pipeline fixture.  `headnet` selects the skeleton the way the reference's CLI does (decoder/factory.py:211-225:
    omp = omp19 = COCO_PERSON_SKELETON, omp16, omp31, omp44, omp25); `topk` overrides FLAGS['topk'] (the reference's CLI
    default is 48, decoder/factory.py:154)."""
    FLAGS = dict(globals()['FLAGS'], topk=topk or globals()['FLAGS']['topk'])
    SKEL = [tuple(c) for c in decoder.factory.parse_heads(headnet, 4)['skeleton']]   # the REFERENCE's table
    from offsetguided_amd.decoder.factory import parse_heads as my_parse_heads
    assert [tuple(c) for c in my_parse_heads(headnet, 4)['skeleton']] == SKEL, f'{headnet}: skeleton table differs'
    hm, off = synth.synth_batch(seed, batch, size, size, flip=flip, n_persons=n_persons, skeleton=SKEL)
    proc = ref_processor(decoder, batch, headnet, topk=FLAGS['topk'])
    # --- the reference, stage by stage (decoder/factory.py:52-96) ---
    thm, toff = torch.from_numpy(hm), torch.from_numpy(off)
    if flip:
        m_hm, _, m_off, _, nd = proc.flip_augment(thm, [], toff, [], cat, 2)
    else:
        m_hm, m_off, nd = thm, toff, 2
    hr = torch.nn.functional.interpolate(m_hm, scale_factor=4, mode='bicubic')
    ohr = torch.nn.functional.interpolate(m_off, scale_factor=4, mode='bilinear')
    dets = decoder.joint_dets(hr, FLAGS['topk'])
    limbs = proc.limb_collect.generate_limbs(hr, [], ohr, [], nd).numpy()
    poses = proc.generate_poses(features(hm, off), flip_test=flip, cat_flip_offs=cat)
    proc.worker_pool.close()
    poses_serial = [proc.limb_group.group_skeletons(l) for l in limbs]
    for a, b in zip(poses, poses_serial):
        assert a.shape == b.shape and (a == b).all()

    # --- the oracle on the same inputs ---
    fl = None
    if flip:
        perm, rev = offset_hflip(COCO_KEYPOINTS, SKEL)
        fl = (heatmap_hflip(COCO_KEYPOINTS), perm, rev)
        o_hm, o_off = (oracle.flip_cat if cat else oracle.flip_merge)(hm, off, *fl)
        # the reference hands the cat form on as a (2N, 2L, h, w) VIEW of the same (N, 4L, h, w) memory
        assert (o_hm == m_hm.numpy()).all() and (o_off.ravel() == m_off.numpy().ravel()).all(), f'{name}: flip merge'
        m_off = m_off.reshape(o_off.shape)
    o_poses, o_mid = oracle.decode(hm, off, SKEL, topk_k=FLAGS['topk'], thre_hmp=FLAGS['thre_hmp'],
                                   min_len=FLAGS['min_len'], person_thre=FLAGS['person_thre'],
                                   dist_max=FLAGS['dist_max'], flip=fl, cat_flip_offs=cat)
    assert (o_mid['hm_hr'] == hr.numpy()).all(), f'{name}: bicubic not bit-exact'
    o_ohr = oracle.bilinear4(m_off.numpy())
    assert (o_ohr.ravel() == ohr.numpy().ravel()).all(), f'{name}: bilinear not bit-exact'
    sc, idx = dets[0].numpy(), dets[1].numpy()
    pos = sc > 0  # entries that are real peaks are fully specified; zero filler order is not
    assert (o_mid['scores'][pos] == sc[pos]).all() and (o_mid['inds'][pos] == idx[pos]).all(), f'{name}: topk'
    assert (o_mid['scores'][~pos] == 0).all()
    all_pos = bool(pos.all())
    ds = 0.0
    if all_pos:
        ds = check_limbs(limbs, o_mid['limbs'], name)
        o_l2 = oracle.collect_limbs(o_mid['scores'], o_mid['inds'], o_ohr, False, hr.shape[-2:], SKEL,
                                    FLAGS['thre_hmp'], FLAGS['min_len'], vector_nd=nd)
        assert (o_l2 == o_mid['limbs']).all(), f'{name}: low-res sampling != hi-res gather'
    dp = check_poses(poses, o_poses, name)
    # oracle grouping fed with the REFERENCE limbs must agree exactly (isolates a12)
    for i in range(batch):
        g = oracle.greedy_group(limbs[i], SKEL, 17, FLAGS['person_thre'], FLAGS['dist_max'])
        assert g.shape == poses[i].shape and (g == poses[i]).all(), f'{name}: grouping on ref limbs'

    out = dict(headnet=np.array(headnet), topk=FLAGS['topk'], seed=seed, batch=batch, size=size, flip=int(flip), cat=int(cat), n_persons=-1 if n_persons is None else n_persons,
               in_sha=np.array([sha(hm), sha(off)]), hm_hr_sha=np.array(sha(hr.numpy())),
               off_hr_sha=np.array(sha(ohr.numpy())), all_positive=int(all_pos),
               scores=sc, inds=idx, limbs=limbs, n_poses=np.array([len(p) for p in poses]),
               poses=np.concatenate(poses, 0) if sum(len(p) for p in poses) else np.zeros((0, 17, 6), np.float32))
    if flip:
        out['merged_sha'] = np.array([sha(m_hm.numpy()), sha(m_off.numpy())])
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **out)
    print(f'{name}: poses/img {[len(p) for p in poses]}, all_pos={all_pos}, limb score err {ds:.2e}, pose ls err {dp:.2e}')


def scale_case(decoder, name, seed, batch, size, flip):
    """Keypoint-scale head (include_scale): scale maps ride in features[omp][2], LimbsCollect gathers them into limbs
    columns 11/12 and GreedyGroup(use_scale=True) uses max(dist_max, scale_to) as the rejection radius."""
    hm, off = synth.synth_batch(seed, batch, size, size, flip=flip, n_persons=8)
    nb = hm.shape[0]
    scl = (synth.noise_batch(seed + 5, (nb, 17, size // 4, size // 4)) * 20 + 25).astype(np.float32)   # 5 .. 45 px
    p = argparse.ArgumentParser()
    decoder.decoder_cli(p)
    a = p.parse_args(['--topk', str(FLAGS['topk']), '--thre-hmp', str(FLAGS['thre_hmp']), '--person-thre',
                      str(FLAGS['person_thre']), '--dist-max', '6', '--min-len', str(FLAGS['min_len']), '--use-scale', 'True'])
    a.headnets, a.strides, a.batch_size = ['hmp', 'omp'], [4, 4], batch
    a.include_scale, a.include_jitter_offset = True, False
    proc = decoder.decoder_factory(a)
    t = torch.from_numpy
    feats = [([t(hm) * 0, t(hm)], [[], []], [[], []]), ([t(off) * 0, t(off)], [[], []], [t(scl) * 0, t(scl)])]
    poses = proc.generate_poses(feats, flip_test=flip)
    proc.worker_pool.close()
    fl = None
    if flip:
        perm, rev = offset_hflip(COCO_KEYPOINTS, COCO_PERSON_SKELETON)
        fl = (heatmap_hflip(COCO_KEYPOINTS), perm, rev)
    o_poses, o_mid = oracle.decode(hm, off, COCO_PERSON_SKELETON, topk_k=FLAGS['topk'], thre_hmp=FLAGS['thre_hmp'],
                                   min_len=FLAGS['min_len'], person_thre=FLAGS['person_thre'], dist_max=6.0,
                                   use_scale=True, flip=fl, scales_lr=scl)
    dp = check_poses(poses, o_poses, name)
    # the scale columns of the poses come straight from the gathered maps: exact
    for r, m in zip(poses, o_poses):
        assert (r[..., 3] == m[..., 3]).all()
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), seed=seed, batch=batch, size=size, flip=int(flip),
                        in_sha=np.array([sha(hm), sha(off), sha(scl)]), n_poses=np.array([len(q) for q in poses]),
                        poses=np.concatenate(poses, 0) if sum(len(q) for q in poses) else np.zeros((0, 17, 6), np.float32))
    print(f'{name}: poses/img {[len(q) for q in poses]} with the scale head, pose ls err {dp:.2e}')


def jitter_case(decoder, name, seed, flip):
    """Jitter-offset head (include_jitter_offset / use_jitter_offset): jitter maps ride in features[hmp][2]; the guide
    points and the end points of the limbs are refined by them (collect.py:127-138, :154-165, :210-214)."""
    batch, size = 2, 256
    hm, off = synth.synth_batch(seed, batch, size, size, flip=flip, n_persons=7)
    nb = hm.shape[0]
    jit = ((synth.noise_batch(seed + 9, (nb, 2, size // 4, size // 4)) - 0.5) * 3.0).astype(np.float32)   # +-1.5 px
    p = argparse.ArgumentParser()
    decoder.decoder_cli(p)
    a = p.parse_args(['--topk', str(FLAGS['topk']), '--thre-hmp', str(FLAGS['thre_hmp']), '--person-thre',
                      str(FLAGS['person_thre']), '--dist-max', str(FLAGS['dist_max']), '--min-len', str(FLAGS['min_len']),
                      '--use-jitter-offset', 'True'])
    a.headnets, a.strides, a.batch_size = ['hmp', 'omp'], [4, 4], batch
    a.include_scale, a.include_jitter_offset = False, True
    proc = decoder.decoder_factory(a)
    t = torch.from_numpy
    feats = [([t(hm) * 0, t(hm)], [[], []], [t(jit) * 0, t(jit)]), ([t(off) * 0, t(off)], [[], []], [[], []])]
    poses = proc.generate_poses(feats, flip_test=flip)
    proc.worker_pool.close()
    fl = None
    if flip:
        perm, rev = offset_hflip(COCO_KEYPOINTS, COCO_PERSON_SKELETON)
        fl = (heatmap_hflip(COCO_KEYPOINTS), perm, rev)
    o_poses, _ = oracle.decode(hm, off, COCO_PERSON_SKELETON, topk_k=FLAGS['topk'], thre_hmp=FLAGS['thre_hmp'],
                               min_len=FLAGS['min_len'], person_thre=FLAGS['person_thre'], dist_max=FLAGS['dist_max'],
                               flip=fl, jitter_lr=jit)
    dp = check_poses(poses, o_poses, name)
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), seed=seed, batch=batch, size=size, flip=int(flip),
                        in_sha=np.array([sha(hm), sha(off), sha(jit)]), n_poses=np.array([len(q) for q in poses]),
                        poses=np.concatenate(poses, 0))
    print(f'{name}: poses/img {[len(q) for q in poses]} with the jitter head, pose ls err {dp:.2e}')


def scored_case(decoder):
    """Optional heatmap-weighted offsets (decoder/offset.py:8-43, generate_poses(scored_off=True)): the function itself
    (bit-exact: same torch ops) and the poses it leads to."""
    from offsetguided_amd.decoder.offset import scored_offset as mine
    hm, off = synth.synth_batch(501, 2, 256, 256, n_persons=6)
    jf, jt = decoder.offset.pack_jtypes(COCO_PERSON_SKELETON)
    ref = decoder.scored_offset(torch.from_numpy(hm), torch.from_numpy(off), jf, jt, kernel_size=3)
    got = mine(torch.from_numpy(hm), torch.from_numpy(off), jf, jt, kernel_size=3)
    assert torch.equal(ref, got), 'scored_offset differs from the reference'
    proc = ref_processor(decoder, 2)
    poses = proc.generate_poses(features(hm, off), scored_off=True)
    proc.worker_pool.close()
    o_poses, _ = oracle.decode(hm, ref.numpy(), COCO_PERSON_SKELETON, topk_k=FLAGS['topk'], thre_hmp=FLAGS['thre_hmp'],
                               min_len=FLAGS['min_len'], person_thre=FLAGS['person_thre'], dist_max=FLAGS['dist_max'])
    dp = check_poses(poses, o_poses, 'scored256')
    np.savez_compressed(os.path.join(GOLD, 'scored256.npz'), seed=501, batch=2, size=256, in_sha=np.array([sha(hm), sha(off)]),
                        scored_sha=np.array(sha(ref.numpy())), n_poses=np.array([len(q) for q in poses]),
                        poses=np.concatenate(poses, 0))
    print(f'scored256: scored_offset bit-exact, poses/img {[len(q) for q in poses]}, pose ls err {dp:.2e}')


def adversarial_limbs(rng, K, hw=4096, skeleton=COCO_PERSON_SKELETON):
    """(19,K,13) limbs with many index collisions.

    mode 0: endpoints drawn at random from tiny per-joint candidate pools.
    mode 1: a few "persons" whose joints come from pools smaller than the number of persons
            (shared keypoints -> merges, >=3-joint crossings), limbs mostly person-consistent
            (-> redundant-limb phase A) with some cross-person rows and a wide score range
            (-> non-replacing matches, {-1,+1} columns).
    """
    L = len(skeleton)
    mode = int(rng.integers(1, 0, 1)[0])
    pool = int(rng.integers(1, 2, 6)[0]) if mode == 0 else int(rng.integers(1, 1, 3)[0])
    cand_xy = rng.uniform(17 * pool * 2, 1.0, 200.0).reshape(17, pool, 2).round()
    cand_v = rng.uniform(17 * pool, 0.05, 1.0).reshape(17, pool)
    limbs = np.zeros((L, K, 13), np.float32)
    P = int(rng.integers(1, 2, 5)[0])
    pid = rng.integers(P * 17, 0, pool - 1).reshape(P, 17)
    for l, (a, b) in enumerate(skeleton):
        if mode == 0:
            nrow = int(rng.integers(1, 0, min(K, pool))[0])
            fsel = np.argsort(rng.uniform(pool))[:nrow]            # from-candidates are distinct peaks
            tsel = rng.integers(nrow, 0, pool - 1)                 # to-candidates may repeat
        else:
            order = np.argsort(rng.uniform(P))
            take = rng.uniform(P) < 0.8
            cross = rng.uniform(P) < 0.3
            other = rng.integers(P, 0, P - 1)
            fl, tl = [], []
            for p in order:
                if not take[p] or pid[p, a] in fl:
                    continue
                fl.append(pid[p, a])
                tl.append(pid[other[p], b] if cross[p] else pid[p, b])
            fsel, tsel = np.array(fl[:K], int), np.array(tl[:K], int)
            nrow = len(fsel)
        sc = rng.uniform(K, 0.01, 1.0) * (1.0 if mode == 0 else 10.0 ** -float(rng.integers(1, 0, 2)[0]))
        dist = rng.uniform(K, 0.0, 60.0 if mode == 0 else 45.0)
        for k in range(K):
            if k < nrow:
                f, t = fsel[k], tsel[k]
                limbs[l, k] = [cand_xy[a, f, 0], cand_xy[a, f, 1], cand_v[a, f],
                               cand_xy[b, t, 0], cand_xy[b, t, 1], cand_v[b, t],
                               a * hw + f, b * hw + t, dist[k], 10.0, sc[k], 4.0, 4.0]
            else:  # sub-threshold filler rows as produced by _channel_dets (collect.py:253)
                limbs[l, k] = [-99990.0, -99980.0, 0.001, -99970.0, -99960.0, 0.002,
                               a * hw + pool + k, b * hw + pool + k, 5.0, 10.0, 1e-6 * (k + 1), 4.0, 4.0]
    return limbs


GROUP_SKELETONS = ['COCO_PERSON_SKELETON', 'COCO_PERSON_WITH_REDUNDANT_SKELETON', 'DENSER_COCO_PERSON_SKELETON',
                   'KINEMATIC_TREE_SKELETON']
# (person_thre, dist_max, use_scale, sort_dim)
GROUP_CFGS = [(FLAGS['person_thre'], FLAGS['dist_max'], 0, 2), (0.02, 25.0, 1, 4)]


def grouping_cases(decoder, n_fuzz, n_store):
    """Fuzz GreedyGroup.group_skeletons vs the oracle on every skeleton the reference defines."""
    import contextlib
    import io
    from offsetguided_amd.config import coco_data as cd
    out = dict(cfg_table=np.array(GROUP_CFGS, np.float64), skeleton_names=np.array(GROUP_SKELETONS))
    for si, sk_name in enumerate(GROUP_SKELETONS):
        sk = getattr(cd, sk_name)
        groupers = [decoder.GreedyGroup(c[0], sort_dim=c[3], dist_max=c[1], use_scale=bool(c[2]),
                                        keypoints=COCO_KEYPOINTS, skeleton=sk) for c in GROUP_CFGS]
        store_l, store_p, store_n, store_cfg = [], [], [], []
        oracle.group_stats(True)
        for i in range(n_fuzz):
            rng = synth.HashRng(777000 + 100000 * si + i)
            K = int(rng.integers(1, 2, 12)[0])
            limbs = adversarial_limbs(rng, K, skeleton=sk)
            cfg = i % 2
            g = groupers[cfg]
            with contextlib.redirect_stdout(io.StringIO()):
                ref = g.group_skeletons(limbs.copy())
            mine = oracle.greedy_group(limbs, sk, 17, g.person_thre, g.dist_max, g.use_scale, g.sort_dim)
            assert ref.shape == mine.shape and (ref == mine).all(), f'grouping fuzz case {sk_name}/{i}'
            if i < n_store:
                pad = np.zeros((len(sk), 12, 13), np.float32)
                pad[:, :K] = limbs
                store_l.append(pad)
                store_p.append(ref)
                store_n.append((K, len(ref)))
                store_cfg.append(cfg)
        out[f'limbs_{si}'] = np.stack(store_l)
        out[f'kn_{si}'] = np.array(store_n)
        out[f'cfg_{si}'] = np.array(store_cfg)
        out[f'poses_{si}'] = np.concatenate(store_p, 0)
        print(f'grouping/{sk_name}: {n_fuzz} fuzz cases bit-exact, {n_store} stored; coverage {oracle.group_stats()}')
    np.savez_compressed(os.path.join(GOLD, 'grouping_adversarial.npz'), **out)


def stage_units():
    """bicubic / bilinear / NMS / top-k on plain noise (incl. negative peaks, borders)."""
    x = synth.noise_batch(11, (2, 3, 24, 40))
    yb = torch.nn.functional.interpolate(torch.from_numpy(x), scale_factor=4, mode='bicubic').numpy()
    yl = torch.nn.functional.interpolate(torch.from_numpy(x), scale_factor=4, mode='bilinear').numpy()
    assert (oracle.bicubic4(x) == yb).all() and (oracle.bilinear4(x) == yl).all()
    import decoder
    z = synth.noise_batch(12, (2, 3, 37, 53))
    nm = decoder.hmp_NMS(torch.from_numpy(z)).numpy()
    o = oracle.hmp_nms(z)
    assert (o == nm).all() and (np.signbit(o) == np.signbit(nm)).all()
    s, i, ys, xs = decoder.heatmap.topK_channel(torch.from_numpy(z), K=9)  # the floor-division shim
    os_, oi, oy, ox = oracle.topk(z, 9)
    assert (os_ == s.numpy()).all() and (oi == i.numpy()).all() and (oy == ys.numpy()).all() and (ox == xs.numpy()).all()
    np.savez_compressed(os.path.join(GOLD, 'stage_units.npz'),
                        in_sha=np.array([sha(x), sha(z)]), bicubic_sha=np.array(sha(yb)), bilinear_sha=np.array(sha(yl)),
                        nms_sha=np.array(sha(nm)), topk_scores=s.numpy(), topk_inds=i.numpy())
    print('stage units: bicubic/bilinear/NMS/top-k bit-exact')


def skeleton_cases(decoder):
    """The head configurations the drop-in surface accepts besides the default (decoder/factory.py:211-225): K1 -> K2 -> K3
    with and without the flip merge (limb_perm / reserve of THAT skeleton, config/coco_data.py:130-153) and with
    cat_flip_offs; omp44 also at the CLI's default --topk 48 (decoder/factory.py:154)."""
    for i, hn in enumerate(['omp16', 'omp31', 'omp44', 'omp25']):
        # (seed 703 has two equal sub-threshold peaks in one plane: torch.topk's order of ties is unspecified, the strict
        # check below refuses such a fixture)
        pipeline_case(decoder, f'pipe256_{hn}_p6', (700 if i < 3 else 740) + i, 2, 256, False, 6, headnet=hn)
        pipeline_case(decoder, f'pipe256_{hn}_flip_p6', 710 + i, 2, 256, True, 6, headnet=hn)
        pipeline_case(decoder, f'pipe256_{hn}_flipcat_p6', 720 + i, 2, 256, True, 6, cat=True, headnet=hn)
    pipeline_case(decoder, 'pipe256_omp44_k48_p20', 730, 2, 256, False, 20, headnet='omp44', topk=48)
    pipeline_case(decoder, 'pipe256_omp44_k48_flip_p20', 731, 2, 256, True, 20, headnet='omp44', topk=48)
    pipeline_case(decoder, 'pipe640_omp31_k48_flip', 732, 1, 640, True, None, headnet='omp31', topk=48)
    pipeline_case(decoder, 'pipe256_omp19_k48_p6', 733, 2, 256, False, 6, headnet='omp19', topk=48)


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    decoder = load_reference()
    pipeline_case(decoder, 'pipe256_flipcat_p6', 306, 2, 256, True, 6, cat=True)
    pipeline_case(decoder, 'pipe640_flipcat', 642, 2, 640, True, None, cat=True)
    jitter_case(decoder, 'jitter256', 601, False)
    jitter_case(decoder, 'jitter256_flip', 602, True)
    scored_case(decoder)
    scale_case(decoder, 'scale256', 401, 2, 256, False)
    scale_case(decoder, 'scale256_flip', 402, 2, 256, True)
    if '--cat-only' in sys.argv:
        return
    skeleton_cases(decoder)
    if '--skeletons-only' in sys.argv:
        return
    stage_units()
    grouping_cases(decoder, n_fuzz=1500, n_store=60)
    for P in (0, 1, 6, 20):
        pipeline_case(decoder, f'pipe256_p{P}', 100 + P, 2, 256, False, P)
    pipeline_case(decoder, 'pipe256_flip_p6', 206, 2, 256, True, 6)
    pipeline_case(decoder, 'pipe256_flip_p20', 220, 2, 256, True, 20)
    pipeline_case(decoder, 'pipe640', 640, 2, 640, False, None)
    pipeline_case(decoder, 'pipe640_flip', 641, 2, 640, True, None)


if __name__ == '__main__':
    main()
