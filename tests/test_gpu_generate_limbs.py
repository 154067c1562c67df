"""og_generate_limbs_f32 -- joint_dets + limb pairing in ONE call (decoder/heatmap.py:15-59 + decoder/collect.py:62-236): two queued
launches (band top-k; merge + pairing) -- through the C ABI, against the oracle and against the separate entry points
(og_nms_topk_f32 + og_collect_limbs_full_f32).

Bit-exact candidate lists (scores, flat indices) and limb rows; the limb score within 1e-4 of the oracle (the device exp()).
Exercised under repetition, with workspace reuse across shapes (shapes whose merge stage takes the fallback interleaved) and on
the full bs8 640x640 batch."""
import numpy as np
import pytest
import torch

import oracle
from offsetguided_amd import _lib, synth
from offsetguided_amd.config import coco_data as cd
from helpers import assert_limbs_match

pytestmark = pytest.mark.gpu
SK = cd.COCO_PERSON_SKELETON
JF, JT = [a for a, _ in SK], [b for _, b in SK]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no HIP device is visible")
    _lib.load()
    return torch.device("cuda:0")


def run_single(hr, off, k, dev, ws=None, want_lists=True, thre=0.04, min_len=0.5, off_lowres=True, single=0):
    """-> limbs (N,L,k,13), scores (N,C,k), inds (N,C,k), workspace"""
    lib = _lib.load()
    n, c, h, w = hr.shape
    L = len(SK)
    if ws is None:
        ws = torch.zeros(lib.og_generate_limbs_workspace_bytes(n, c, h, w, k), dtype=torch.uint8, device=dev)
    assert ws.numel() >= lib.og_generate_limbs_workspace_bytes(n, c, h, w, k)
    limbs = torch.full((n, L, k, 13), float('nan'), device=dev)
    sc = torch.full((n, c, k), float('nan'), device=dev) if want_lists else None
    ix = torch.full((n, c, k), -1, dtype=torch.int64, device=dev) if want_lists else None
    jf, jt = _lib.int_table(JF, dev), _lib.int_table(JT, dev)
    _lib.check(lib.og_generate_limbs_f32(_lib.ptr(hr), _lib.ptr(off), int(off_lowres), 2, None, 0, None, 0, n, c, h, w,
                                         _lib.ptr(jf), _lib.ptr(jt), L, k, thre, min_len, 1.0,
                                         _lib.ptr(sc) if want_lists else None, _lib.ptr(ix) if want_lists else None,
                                         _lib.ptr(limbs), int(single), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), lib)
    return limbs, sc, ix, ws


def run_three(hr, off, k, dev, thre=0.04, min_len=0.5):
    lib = _lib.load()
    n, c, h, w = hr.shape
    L = len(SK)
    ws = torch.zeros(lib.og_topk_workspace_bytes(n * c, h, w, k), dtype=torch.uint8, device=dev)
    sc = torch.empty((n, c, k), device=dev)
    ix = torch.empty((n, c, k), dtype=torch.int64, device=dev)
    limbs = torch.empty((n, L, k, 13), device=dev)
    jf, jt = _lib.int_table(JF, dev), _lib.int_table(JT, dev)
    _lib.check(lib.og_nms_topk_f32(_lib.ptr(hr), n * c, h, w, k, _lib.ptr(sc), _lib.ptr(ix), _lib.ptr(ws), ws.numel(),
                                   _lib.stream_ptr(dev)), lib)
    _lib.check(lib.og_collect_limbs_full_f32(_lib.ptr(sc), _lib.ptr(ix), _lib.ptr(off), 1, 2, None, 0, None, 0, n, c, h, w,
                                             _lib.ptr(jf), _lib.ptr(jt), L, k, thre, min_len, 1.0, _lib.ptr(limbs),
                                             _lib.stream_ptr(dev)), lib)
    return limbs, sc, ix


def maps(seed, n, h, w, dev, persons=4):
    hm, off = synth.synth_batch(seed, n, h, w, n_persons=persons)
    hr = oracle.bicubic4(hm)
    return hr, off, torch.from_numpy(hr).to(dev), torch.from_numpy(off).to(dev)


@pytest.mark.parametrize("n,size,k", [(2, (256, 256), 32), (1, (640, 640), 32), (3, (384, 512), 48), (2, (128, 640), 16),
                                       (8, (256, 256), 64), (1, (64, 64), 9), (5, (512, 256), 1),
                                       (2, (320, 320), 100)])
def test_single_launch_matches_oracle_and_three_launch(dev, n, size, k):
    hr, off, t_hr, t_off = maps(500 + n + k, n, size[0], size[1], dev)
    limbs, sc, ix, _ = run_single(t_hr, t_off, k, dev)
    rs, ri, _, _ = oracle.nms_topk(hr, k)
    assert (sc.cpu().numpy() == rs).all() and (ix.cpu().numpy() == ri).all()
    ref = oracle.collect_limbs(rs, ri, off, True, hr.shape[2:], SK, 0.04, 0.5)
    assert_limbs_match(ref, limbs.cpu().numpy(), 1e-4)
    l3, s3, i3 = run_three(t_hr, t_off, k, dev)
    assert torch.equal(l3, limbs) and torch.equal(s3, sc) and torch.equal(i3, ix)      # same code, same bits
    l_only, _, _, _ = run_single(t_hr, t_off, k, dev, want_lists=False)             # lists optional
    assert torch.equal(l_only, limbs)
    again, s_again, i_again, w2 = run_single(t_hr, t_off, k, dev)                     # a second call on a fresh workspace: same bits
    assert torch.equal(again, limbs) and torch.equal(s_again, sc) and torch.equal(i_again, ix)
    assert int(w2[:61440].view(torch.int32).abs().sum()) == 0, "the reserved head of the workspace must stay zero"


def test_degenerate_planes(dev):
    """All-zero, constant, negative-only, plateau and first-row planes: the zero-filler rule inside the finisher."""
    H, W, k = 64, 64, 32
    planes = np.zeros((2, 17, H, W), np.float32)
    planes[0, 1] = 0.5
    planes[0, 2] = -1.0
    planes[0, 3, 10:14, 20:30] = 0.7
    planes[0, 4] = -np.abs(synth.noise_batch(3, (H, W)))
    planes[0, 5, 0, :5] = [0.3, 0.0, 0.2, 0.0, 0.1]
    planes[1] = synth.noise_batch(4, (17, H, W)) - 0.45           # a handful of positive peaks per plane: < k
    off = synth.noise_batch(6, (2, 38, H // 4, W // 4)).astype(np.float32) * 8
    t_hr, t_off = torch.from_numpy(planes).to(dev), torch.from_numpy(off).to(dev)
    limbs, sc, ix, _ = run_single(t_hr, t_off, k, dev)
    rs, ri, _, _ = oracle.nms_topk(planes, k)
    assert (sc.cpu().numpy() == rs).all() and (ix.cpu().numpy() == ri).all()
    l3, _, _ = run_three(t_hr, t_off, k, dev)
    assert torch.equal(l3, limbs)
    lf, sf, xf, _ = run_single(t_hr, t_off, k, dev)   # the zero-filler rule in the merge kernel, again
    assert torch.equal(lf, limbs) and torch.equal(sf, sc) and torch.equal(xf, ix)


def test_workspace_reuse_across_shapes_and_fallback(dev):
    """One zero-filled workspace, many calls: large shapes, shapes that take other paths (k > 64, W % 4 != 0, tiny planes) and
    back -- the reserved head of the workspace stays zero, every call finds the workspace ready."""
    lib = _lib.load()
    seq = [(2, 256, 256, 32), (1, 96, 128, 100), (2, 256, 256, 32), (1, 40, 50, 8), (3, 128, 128, 48), (1, 16, 16, 4),
           (2, 256, 256, 32)]
    nbytes = max(lib.og_generate_limbs_workspace_bytes(n, 17, h, w, k) for n, h, w, k in seq)
    ws = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    for j, (n, h, w, k) in enumerate(seq):
        if w % 4 == 0:
            hr, off, t_hr, t_off = maps(700 + j, n, h, w, dev, persons=2)
            off_lr = True
        else:   # hi-res offsets gathered: no divisibility needed
            hr = synth.noise_batch(700 + j, (n, 17, h, w)) - 0.3
            off = synth.noise_batch(800 + j, (n, 38, h, w)) * 5
            t_hr, t_off, off_lr = torch.from_numpy(hr).to(dev), torch.from_numpy(off).to(dev), False
        limbs, sc, ix, _ = run_single(t_hr, t_off, k, dev, ws=ws, off_lowres=off_lr)
        rs, ri, _, _ = oracle.nms_topk(hr, k)
        assert (sc.cpu().numpy() == rs).all() and (ix.cpu().numpy() == ri).all(), (j, n, h, w, k)
        ref = oracle.collect_limbs(rs, ri, off, off_lr, (h, w), SK, 0.04, 0.5)
        assert_limbs_match(ref, limbs.cpu().numpy(), 1e-4)
        assert int(ws[:65536].view(torch.int32).abs().sum()) == 0, "the reserved head of the workspace must stay zero"


def test_full_size_batch_repeated(dev):
    """bs8 640x640 (BASELINE configs[1]): 30 back-to-back launches on rotating inputs give identical bits every time
    (hand-off under load), equal to the three-launch form; two images against the oracle."""
    hm, off = synth.synth_batch(7, 8, 640, 640)
    lr = torch.from_numpy(hm).to(dev)
    t_off = torch.from_numpy(off).to(dev)
    from offsetguided_amd.decoder.factory import upsample4
    t_hr = upsample4(lr, 'bicubic')
    others = [upsample4(lr.flip(0), 'bicubic'), upsample4(lr * 0.5, 'bicubic')]
    l3, s3, i3 = run_three(t_hr, t_off, 32, dev)
    limbs, sc, ix, ws = run_single(t_hr, t_off, 32, dev)
    assert torch.equal(l3, limbs) and torch.equal(s3, sc) and torch.equal(i3, ix)
    for r in range(30):
        run_single(others[r % 2], t_off, 32, dev, ws=ws)            # different data in between, same workspace
        l2, s2, i2, _ = run_single(t_hr, t_off, 32, dev, ws=ws)
        assert torch.equal(l2, limbs) and torch.equal(s2, sc) and torch.equal(i2, ix), r
    rs, ri, _, _ = oracle.nms_topk(oracle.bicubic4(hm[:2]), 32)
    assert (sc[:2].cpu().numpy() == rs).all() and (ix[:2].cpu().numpy() == ri).all()
    ref = oracle.collect_limbs(rs, ri, off[:2], True, (640, 640), SK, 0.04, 0.5)
    assert_limbs_match(ref, limbs[:2].cpu().numpy(), 1e-4)


def test_bs1_and_large_batch(dev):
    """configs[0]-sized input (1 image: many workgroups per plane) and a batch whose ranges span several planes."""
    for n, h, w in [(1, 640, 640), (40, 128, 128)]:
        hr, off, t_hr, t_off = maps(900 + n, n, h, w, dev, persons=3)
        limbs, sc, ix, _ = run_single(t_hr, t_off, 32, dev)
        l3, s3, i3 = run_three(t_hr, t_off, 32, dev)
        assert torch.equal(l3, limbs) and torch.equal(s3, sc) and torch.equal(i3, ix)


def test_errors(dev):
    lib = _lib.load()
    t = torch.zeros(1, 17, 64, 64, device=dev)
    o = torch.zeros(1, 38, 16, 16, device=dev)
    limbs = torch.zeros(1, 19, 32, 13, device=dev)
    jf, jt = _lib.int_table(JF, dev), _lib.int_table(JT, dev)
    ws = torch.zeros(1024, dtype=torch.uint8, device=dev)
    rc = lib.og_generate_limbs_f32(_lib.ptr(t), _lib.ptr(o), 1, 2, None, 0, None, 0, 1, 17, 64, 64, _lib.ptr(jf), _lib.ptr(jt),
                                   19, 32, 0.04, 0.5, 1.0, None, None, _lib.ptr(limbs), 0, _lib.ptr(ws), ws.numel(), None)
    assert rc == _lib.OG_ENOSPC and b"workspace" in lib.og_last_error()
    rc = lib.og_generate_limbs_f32(_lib.ptr(t), _lib.ptr(o), 1, 3, None, 0, None, 0, 1, 17, 64, 64, _lib.ptr(jf), _lib.ptr(jt),
                                   19, 32, 0.04, 0.5, 1.0, None, None, _lib.ptr(limbs), 0, _lib.ptr(ws), ws.numel(), None)
    assert rc == _lib.OG_EUNSUPPORTED


def test_random_shapes_all_forms_agree(dev):
    """Random (N, H, W, k) incl. degenerate planes: og_generate_limbs_f32 (band top-k + merge-and-pair) against the separate
    entry points, bit for bit; a reserved `flags` value changes nothing."""
    rng = np.random.default_rng(7)
    done = 0
    for it in range(40):
        n, h, w = int(rng.integers(1, 7)), int(rng.integers(4, 80)) * 4, int(rng.integers(4, 80)) * 4
        k = int(rng.choice([1, 3, 8, 17, 32, 40, 64, 100]))
        if h * w < k or 2 * (h + w) - 4 < k:
            continue
        hm = (synth.noise_batch(1000 + it, (n, 17, h, w)) - rng.uniform(0.2, 0.6)).astype(np.float32)
        if it % 5 == 0:
            hm[:, ::3] = 0.0
        off = (synth.noise_batch(2000 + it, (n, 38, h // 4, w // 4)) * 6).astype(np.float32)
        t_hr, t_off = torch.from_numpy(hm).to(dev), torch.from_numpy(off).to(dev)
        l3, s3, i3 = run_three(t_hr, t_off, k, dev)
        for flags in (0, 3):
            l, s, i, ws = run_single(t_hr, t_off, k, dev, single=flags)
            assert torch.equal(l, l3) and torch.equal(s, s3) and torch.equal(i, i3), (it, n, h, w, k, flags)
            assert int(ws[:61440].view(torch.int32).abs().sum()) == 0, (it, flags)
        done += 1
    assert done >= 30


@pytest.mark.parametrize("knobs", [{"OG_K1_WAVE_LISTS": "0"}, {"OG_NMS_ROWS": "48"}])
def test_band_layout_knobs(dev, knobs):
    """The kept A/B layouts of the band kernel (read once per process): one list per band instead of one per streaming wave, another
    band height -- the random-shape and the full-size tests again in a child process, bit for bit against the separate entry points."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, **knobs)
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k",
                        "test_random_shapes_all_forms_agree or test_full_size_batch_repeated or test_degenerate_planes"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert " passed" in r.stdout


def test_offset_planes_full_size(dev):
    """Planes that sit on a large positive or negative offset with large-scale structure (what a random-init network's heads
    add to the maps in bench.py): every pixel above the starting threshold, or fewer than k positive peaks in the plane (the
    merge's zero-fill path) -- bs8 640x640, flags 0 against the separate entry points, bit for bit, three times."""
    n, h, w, k = 8, 640, 640, 32
    hm, off = synth.synth_batch(11, n, h, w)
    yy, xx = np.meshgrid(np.linspace(-1, 1, h // 4, dtype=np.float32), np.linspace(-1, 1, w // 4, dtype=np.float32), indexing='ij')
    rng = np.random.default_rng(3)
    for i in range(n):
        for c in range(17):
            a, b, o = rng.uniform(-0.4, 0.4, 3)
            hm[i, c] += (o + a * yy + b * xx).astype(np.float32)   # some planes end up negative nearly everywhere
    hr = torch.from_numpy(oracle.bicubic4(hm)).to(dev)
    t_off = torch.from_numpy(off).to(dev)
    l3, s3, i3 = run_three(hr, t_off, k, dev)
    ws = None
    for _ in range(3):
        l, s, i, ws = run_single(hr, t_off, k, dev, ws=ws, single=0)
        assert torch.equal(l, l3) and torch.equal(s, s3) and torch.equal(i, i3)
    few = int(((torch.from_numpy(oracle.hmp_nms(hr.cpu().numpy()[:1])) > 0).sum(dim=(2, 3)) < k).sum())
    assert few >= 1, "the case is meant to hold planes with fewer than k positive peaks"
