"""Reference-shaped CPU decoder in Python -- TEST / BASELINE INFRASTRUCTURE ONLY (never imported by the product).

BASELINE.md section 4.1 / SURVEY.md 8(d) define the CPU baseline as "our Python restatement of
decoder/factory.py:52-96: torch-CPU upsample -> pad + max_pool NMS -> topk -> gather / pair -> numpy greedy grouping in
a Pool(batch)", timed on all host cores.  This module is that restatement, written from the specification in SURVEY.md
Appendix A (it shares no text with the reference; the reference's own files never travel to the GPU box).  It uses the
same torch / numpy operations the reference calls, so its cost profile is the reference's:

    upsample        decoder/factory.py:74-78    F.interpolate x4, bicubic (heatmaps) / bilinear (offsets)
    hmp_nms         decoder/heatmap.py:15-35    F.pad + F.max_pool2d + (hmax == heat)
    topk_channel    decoder/heatmap.py:38-49    torch.topk per (n, c) plane
    generate_limbs  decoder/collect.py:62-236   gathers, guide points, pairwise distances, first argmin
    group_skeletons decoder/group.py:39-240     serial over limb types, numpy
    flip_augment    decoder/factory.py:98-146

Parity: pinned against the committed golden vectors (outputs of the imported reference) by
tests/test_oracle_golden.py::test_restatement_* -- indices, coordinates and grouping bit-exact, scores <= 1e-4.
"""
import multiprocessing
import time

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------- stages
def upsample(hm_lr, off_lr, mode='bicubic'):
    hm = F.interpolate(hm_lr, scale_factor=4, mode=mode, align_corners=False)
    off = F.interpolate(off_lr, scale_factor=4, mode='bilinear', align_corners=False)
    return hm, off


def hmp_nms(heat):
    pooled = F.max_pool2d(F.pad(heat, (1, 1, 1, 1)), kernel_size=3, stride=1)
    return heat * (pooled == heat).float()


def topk_channel(scores, k):
    n, c, h, w = scores.shape
    vals, inds = torch.topk(scores.reshape(n, c, h * w), k)
    return vals, inds, inds // w, inds % w


def flip_augment(hm, off, kp_perm, limb_perm, keep_original):
    """[images, mirrored images] -> merged maps."""
    n = hm.shape[0] // 2
    hm = (hm[:n] + torch.flip(hm[n:], [-1])[:, kp_perm]) / 2
    L = off.shape[1] // 2
    o = off.reshape(2 * n, L, 2, *off.shape[-2:])
    mirrored = torch.flip(o[n:], [-1]).clone()
    mirrored[:, :, 0] *= -1
    merged = (o[:n] + mirrored[:, limb_perm]) / 2
    merged[:, keep_original] = o[:n, keep_original]
    return hm, merged.reshape(n, 2 * L, *off.shape[-2:])


def generate_limbs(hm_hr, off_hr, skeleton, k, thre, min_len):
    """(N,C,H,W) heatmaps + (N,2L,H,W) offsets -> (N,L,k,13) candidate limbs."""
    n, c, h, w = hm_hr.shape
    jf = [a for a, _ in skeleton]
    jt = [b for _, b in skeleton]
    L = len(skeleton)
    s, idx, ys, xs = topk_channel(hmp_nms(hm_hr), k)
    low = s < thre
    xs = torch.where(low, xs - 100000, xs).float()
    ys = torch.where(low, ys - 100000, ys).float()
    s_f, s_t = s[:, jf], s[:, jt]
    i_f, i_t = idx[:, jf], idx[:, jt]
    xy_f = torch.stack((xs[:, jf], ys[:, jf]), -1)                         # (N,L,k,2)
    xy_t = torch.stack((xs[:, jt], ys[:, jt]), -1)
    o = off_hr.reshape(n, L, 2, h * w)
    vec = torch.stack((o[:, :, 0].gather(2, i_f), o[:, :, 1].gather(2, i_f)), -1)
    guide = xy_f + vec * 1.0
    dist = (guide[:, :, :, None, :] - xy_t[:, :, None, :, :]).norm(dim=-1)   # (N,L,k,k)
    d_min, m = dist.min(dim=-1)
    xy_m = xy_t.gather(2, m[..., None].expand(-1, -1, -1, 2))
    s_m, i_m = s_t.gather(2, m), i_t.gather(2, m)
    length = (xy_f - xy_m).norm(dim=-1).clamp(min=min_len)
    score = s_f * s_m * torch.exp(-d_min / length)
    plane = h * w
    ind1 = (i_f + torch.tensor(jf).view(1, L, 1) * plane).float()
    ind2 = (i_m + torch.tensor(jt).view(1, L, 1) * plane).float()
    four = torch.full_like(score, 4.0)
    return torch.stack((xy_f[..., 0], xy_f[..., 1], s_f, xy_m[..., 0], xy_m[..., 1], s_m, ind1, ind2, d_min, length,
                        score, four, four), -1)


def _f32_sum(v):
    """numpy's float32 sum order for <= 17 elements."""
    return np.float32(np.sum(np.asarray(v, np.float32)))


def group_skeletons(limbs, skeleton, n_kp=17, person_thre=0.04, dist_max=40.0, use_scale=False, sort_dim=2):
    """(L,k,13) limbs of one image -> (M,n_kp,6) poses (the per-row 'last matching column' form)."""
    limbs = np.asarray(limbs, np.float32)
    rows = []                                                     # each: float32 (n_kp, 6), -1 = unset
    for l, (jf, jt) in enumerate(skeleton):
        cand = limbs[l]
        limit = np.maximum(np.float32(dist_max), cand[:, 12]) if use_scale else np.float32(dist_max)
        ok = (cand[:, 8] < limit) & (cand[:, 0] > 0) & (cand[:, 4] > 0) & (cand[:, 3] > 0) & (cand[:, 1] > 0)
        cand = cand[ok]
        cand = cand[np.argsort(-cand[:, 10], kind='stable')]
        _, first = np.unique(cand[:, 7].astype(np.int64), return_index=True)
        cand = cand[np.sort(first)]
        kk, mm = len(cand), len(rows)
        if kk == 0:
            continue
        i1, i2, sc = cand[:, 6].astype(np.int64), cand[:, 7].astype(np.int64), cand[:, 10]
        if mm:
            sub = np.stack(rows)
            hit = (sub[:, jf, 5].astype(np.int64)[:, None] == i1[None]).astype(np.int64) + \
                  (sub[:, jt, 5].astype(np.int64)[:, None] == i2[None])
            better = (sc[None] > sub[:, jt, 4][:, None]) | (sc[None] > sub[:, jf, 4][:, None])
            both = (hit == 2) & better
            if both.any():                                        # phase A: limb already known, keep the better score
                pre = sub.copy()
                for m_, c_ in zip(*np.nonzero(both)):
                    sub[m_, jf, 4] = max(sc[c_], pre[m_, jf, 4])
                    sub[m_, jt, 4] = max(sc[c_], pre[m_, jt, 4])
                hit[hit == 2] = -1
            one = (hit == 1) & better
            if one.any():                                         # phase B: extend a skeleton by the new joint
                pre = sub.copy()
                for m_, c_ in zip(*np.nonzero(one)):              # row-major: the last column wins per row
                    sub[m_, jf, 5], sub[m_, jt, 5] = cand[c_, 6], cand[c_, 7]
                    sub[m_, jf, 0:3], sub[m_, jf, 3] = cand[c_, 0:3], cand[c_, 11]
                    sub[m_, jt, 0:3], sub[m_, jt, 3] = cand[c_, 3:6], cand[c_, 12]
                    sub[m_, jf, 4] = max(sc[c_], pre[m_, jf, 4])
                    sub[m_, jt, 4] = max(sc[c_], pre[m_, jt, 4])
                hit[hit == 1] = -1
            gone = set()
            if mm >= 2:                                           # phase C: two skeletons sharing exactly two joints merge
                ids = sub[:, :, 5].astype(np.int64)
                pre = sub.copy()
                for a in range(mm):
                    for b in range(a + 1, mm):
                        if int(((ids[a] == ids[b]) & (ids[a] != -1)).sum()) == 2:
                            sub[a] = np.maximum(pre[a], pre[b])
                            gone.add(b)
            orphan = hit.sum(0) == 0
            rows = [sub[m_] for m_ in range(mm) if m_ not in gone]
        else:
            orphan = np.ones(kk, bool)
        for c_ in np.nonzero(orphan)[0]:                          # phase D: unmatched limbs start new skeletons
            r = np.full((n_kp, 6), -1, np.float32)
            r[jf, 0:3], r[jf, 3], r[jf, 4], r[jf, 5] = cand[c_, 0:3], cand[c_, 11], sc[c_], cand[c_, 6]
            r[jt, 0:3], r[jt, 3], r[jt, 4], r[jt, 5] = cand[c_, 3:6], cand[c_, 12], sc[c_], cand[c_, 7]
            rows.append(r)
    scored = []
    for r in rows:
        vals = r[:, sort_dim][r[:, sort_dim] > 0]
        mean = np.float64(_f32_sum(vals)) / np.float64(len(vals)) if len(vals) else np.float64('nan')
        if not mean < person_thre:
            scored.append((mean, r))
    order = sorted(range(len(scored)), key=lambda i: scored[i][0], reverse=True)   # stable
    out = np.stack([scored[i][1] for i in order]) if scored else np.zeros((0, n_kp, 6), np.float32)
    out[out == -1] = 0
    return out.astype(np.float32)


# ---------------------------------------------------------------------------------- pipeline
def _group_worker(args):
    return group_skeletons(*args)


class Decoder:
    """PostProcess.generate_poses (decoder/factory.py:52-96) with a Pool(batch) for the grouping, like the reference."""

    def __init__(self, skeleton, batch, k=32, thre_hmp=0.04, min_len=0.5, person_thre=0.04, dist_max=40.0, flip=None):
        self.skeleton, self.k, self.thre, self.min_len = skeleton, k, thre_hmp, min_len
        self.person_thre, self.dist_max, self.flip = person_thre, dist_max, flip
        self.pool = multiprocessing.get_context('fork').Pool(batch) if batch > 1 else None

    def close(self):
        if self.pool is not None:
            self.pool.terminate()
            self.pool = None

    def generate_poses(self, hm_lr, off_lr):
        hm_lr, off_lr = torch.as_tensor(hm_lr), torch.as_tensor(off_lr)
        with torch.no_grad():
            if self.flip is not None:
                hm_lr, off_lr = flip_augment(hm_lr, off_lr, *self.flip)
            hm_hr, off_hr = upsample(hm_lr, off_lr)
            limbs = generate_limbs(hm_hr, off_hr, self.skeleton, self.k, self.thre, self.min_len).numpy()
        jobs = [(limbs[i], self.skeleton, hm_lr.shape[1], self.person_thre, self.dist_max) for i in range(len(limbs))]
        poses = self.pool.map(_group_worker, jobs) if self.pool is not None else [group_skeletons(*j) for j in jobs]
        return poses, limbs


def time_decoder(hm_lr, off_lr, skeleton, flags, repeats=5):
    """Median wall time of generate_poses on all host cores: 1 warm-up + `repeats` runs -> (seconds per batch, cores)."""
    dec = Decoder(skeleton, len(hm_lr) // (2 if flags.get('flip') else 1), k=flags.get('topk_k', 32),
                  thre_hmp=flags.get('thre_hmp', 0.04), min_len=flags.get('min_len', 0.5),
                  person_thre=flags.get('person_thre', 0.04), dist_max=flags.get('dist_max', 40.0), flip=flags.get('flip'))
    try:
        dec.generate_poses(hm_lr, off_lr)
        ts = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            dec.generate_poses(hm_lr, off_lr)
            ts.append(time.perf_counter() - t0)
    finally:
        dec.close()
    time_decoder.last_runs = [float(t) for t in ts]        # the individual runs: bench.py reports their spread
    return float(np.median(ts)), torch.get_num_threads()


def time_backbone(model, size=640, repeats=5, budget_s=60.0):
    """Median wall time of the eager fp32 forward of one image on all host cores (1 warm-up + up to `repeats` runs within
    the time budget) -> (seconds per image, runs)."""
    model = model.to('cpu').float().eval()
    x = torch.randn(1, 3, size, size)
    ts = []
    with torch.no_grad():
        t_start = time.perf_counter()
        model(x)
        for _ in range(repeats):
            t0 = time.perf_counter()
            model(x)
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget_s and len(ts) >= 1:
                break
    time_backbone.last_runs = [float(t) for t in ts]
    return float(np.median(ts)), len(ts)
