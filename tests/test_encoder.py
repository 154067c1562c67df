"""Ground-truth encoder: the C oracle against the golden vectors taken from the imported reference
(tools/gen_golden_encoder.py) on the CPU; the HIP kernels against both on the GPU."""
import argparse
import os

import numpy as np
import pytest
import torch

import oracle
from offsetguided_amd.config import coco_data as cd
from helpers import GOLDEN

CASES = ["a", "b", "c", "d"]


def load():
    return np.load(os.path.join(GOLDEN, "encoder.npz"))


def heatmaps_match(got, ref, clip=0.01):
    """<= 1e-6 (float32 exp implementations differ by an ulp or two); a pixel whose value sits at the clip threshold may
    be 0 in one and ~clip in the other."""
    d = np.abs(got - ref)
    edge = (np.minimum(got, ref) == 0) & (np.maximum(got, ref) < clip * (1 + 1e-5))
    return bool((d[~edge] <= 1e-6).all()) and int((edge & (d > 0)).sum()) <= 2


@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_golden(case):
    g = load()
    j, size = g[f"{case}_joints"], int(g[f"{case}_size"])
    assert heatmaps_match(oracle.encode_heatmaps(j, size, size), g[f"{case}_hm"])
    off, sc, ps = oracle.encode_offsets(j, cd.COCO_PERSON_SKELETON, cd.COCO_PERSON_SIGMAS, size, size)
    assert np.array_equal(off, g[f"{case}_off"]) and np.array_equal(ps, g[f"{case}_pscale"])
    assert np.array_equal(sc, g[f"{case}_scale"], equal_nan=True)
    assert np.array_equal(oracle.encode_jitter(j, size, size), g[f"{case}_jitter"])


def test_encoder_factory_names():
    from offsetguided_amd import encoder
    p = argparse.ArgumentParser()
    encoder.encoder_cli(p)
    a = p.parse_args([])
    a.headnets, a.square_length = ['hmp', 'omp'], 512
    a.include_background, a.include_jitter_offset, a.include_scale = True, False, False
    encs = encoder.encoder_factory(a, [4, 4], device='cpu')
    assert [type(e).__name__ for e in encs] == ['HeatMaps', 'OffsetMaps']
    assert encs[0].input_size == [512, 512] and encs[0].stride == 4 and encoder.OffsetMaps.skeleton == cd.COCO_PERSON_SKELETON
    assert len(encoder.factory_head('omp44', 512, 4, 'cpu').skeleton) == len(cd.DENSER_COCO_PERSON_SKELETON)
    with pytest.raises(Exception, match='unknown head'):
        encoder.factory_head('paf', 512, 4, 'cpu')
    with pytest.raises(Exception, match='unknown skeleton'):
        encoder.factory_head('omp7', 512, 4, 'cpu')


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_gpu_encoder_matches_reference_golden(case):
    from offsetguided_amd import encoder
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no HIP device is visible")
    g = load()
    j, size = g[f"{case}_joints"], int(g[f"{case}_size"])
    encoder.HeatMaps.include_jitter_offset = True
    encoder.HeatMaps.include_background = True
    encoder.OffsetMaps.include_scale = True
    encoder.OffsetMaps.skeleton = cd.COCO_PERSON_SKELETON
    hm_enc, off_enc = encoder.HeatMaps(size, 4), encoder.OffsetMaps(size, 4)
    # batch of 3: the case, the case with its persons reversed and padded (n_persons masks the padding), no person
    P = max(j.shape[0], 1) + 2
    batch = np.zeros((3, P, 17, 4), np.float32)
    batch[0, :j.shape[0]] = j
    batch[1, :j.shape[0]] = j[::-1]
    batch[1, j.shape[0]:] = 7.0                      # garbage rows behind n_persons must be ignored
    n_persons = np.array([j.shape[0], j.shape[0], 0], np.int32)
    hm, bg, jit, mask = hm_enc.encode_batch(batch, n_persons)
    off, sc, ps, mask2 = off_enc.encode_batch(batch, n_persons)
    assert np.array_equal(jit[0].cpu().numpy(), g[f"{case}_jitter"]) and bool(torch.isinf(jit[2]).all())
    assert mask.dtype == torch.bool and bool(mask.all()) and mask.shape == (3, 1, size // 4, size // 4)
    hm, bg, off, sc, ps = (t.cpu().numpy() for t in (hm, bg, off, sc, ps))
    assert heatmaps_match(hm[0], g[f"{case}_hm"]) and heatmaps_match(hm[1], g[f"{case}_hm"])     # max: order-free
    assert np.array_equal(bg[0, 0], 1 - hm[0].max(0))
    assert np.array_equal(off[0], g[f"{case}_off"]) and np.array_equal(ps[0], g[f"{case}_pscale"])
    assert np.array_equal(sc[0], g[f"{case}_scale"], equal_nan=True)
    r_off, r_sc, r_ps = oracle.encode_offsets(j[::-1], cd.COCO_PERSON_SKELETON, cd.COCO_PERSON_SIGMAS, size, size)
    assert np.array_equal(off[1], r_off) and np.array_equal(ps[1], r_ps) and np.array_equal(sc[1], r_sc, equal_nan=True)
    assert (hm[2] == 0).all() and (bg[2] == 1).all() and np.isinf(off[2]).all() and np.isnan(sc[2]).all() and (ps[2] == 1).all()
    # the per-image call of the reference API
    one = hm_enc(j, {'width_height': [size, size], 'joint_num': 17}, None)
    assert heatmaps_match(one[0].cpu().numpy(), g[f"{case}_hm"]) and one[3].shape == (1, size // 4, size // 4)
    # the reference's mask_miss input: uint8 at input resolution, shrunk on the device (encoder/heatmap.py:56-60)
    full = np.full((3, size, size), 255, np.uint8)
    full[1, : size // 2] = 0
    m = hm_enc.encode_batch(batch, n_persons, full)[3].cpu().numpy()
    assert m.shape == (3, 1, size // 4, size // 4) and m[0].all() and m[2].all()
    assert not m[1, 0, : size // 8 - 1].any() and m[1, 0, size // 8 + 1:].all()
    with pytest.raises(ValueError):
        hm_enc.encode_batch(batch, n_persons, np.zeros((3, size, size), np.float32))
