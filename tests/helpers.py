"""Shared parity helpers (oracle side = checker only)."""
import hashlib
import os

import numpy as np

from offsetguided_amd import synth
from offsetguided_amd.config.coco_data import (COCO_KEYPOINTS, COCO_PERSON_SKELETON, heatmap_hflip,
                                               offset_hflip)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FLAGS = dict(topk=32, thre_hmp=0.04, person_thre=0.04, dist_max=40.0, min_len=0.5)
PIPE_CASES = ["pipe256_p0", "pipe256_p1", "pipe256_p6", "pipe256_p20", "pipe256_flip_p6", "pipe256_flip_p20",
              "pipe640", "pipe640_flip", "pipe256_flipcat_p6", "pipe640_flipcat"]
# head configurations the drop-in surface accepts besides the default (decoder/factory.py:211-225), no flip / flip / cat_flip_offs
# each; omp44 and omp19 also at the CLI's default --topk 48 (decoder/factory.py:154)
SKELETON_CASES = [f"pipe256_{hn}{v}_p6" for hn in ("omp16", "omp31", "omp44", "omp25") for v in ("", "_flip", "_flipcat")] + \
    ["pipe256_omp44_k48_p20", "pipe256_omp44_k48_flip_p20", "pipe640_omp31_k48_flip", "pipe256_omp19_k48_p6"]
PIPE_CASES = PIPE_CASES + SKELETON_CASES
EXACT_LIMB_COLS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12]
EXACT_POSE_COLS = [0, 1, 2, 3, 5]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def flip_tables(skeleton=COCO_PERSON_SKELETON):
    perm, rev = offset_hflip(COCO_KEYPOINTS, skeleton)
    return heatmap_hflip(COCO_KEYPOINTS), perm, rev


def case_headnet(g):
    """Name of the offset head of a pipeline fixture ('omp' for the fixtures that predate the field)."""
    return str(g["headnet"]) if "headnet" in g.files else "omp"


def case_skeleton(g):
    from offsetguided_amd.decoder.factory import parse_heads
    return parse_heads(case_headnet(g), 4)["skeleton"]


def case_flags(g):
    """FLAGS of a pipeline fixture (topk is a field of the newer ones)."""
    return dict(FLAGS, topk=int(g["topk"])) if "topk" in g.files else dict(FLAGS)


def is_cat(g):
    """cat_flip_offs case (decoder/factory.py:115-127)?  Older fixtures predate the field."""
    return bool(int(g["cat"])) if "cat" in g.files else False


def scale_case_inputs(g):
    """Inputs of a keypoint-scale-head fixture (tools/gen_golden.py:scale_case), sha-guarded."""
    batch, size, flip, seed = int(g["batch"]), int(g["size"]), bool(g["flip"]), int(g["seed"])
    hm, off = synth.synth_batch(seed, batch, size, size, flip=flip, n_persons=8)
    scl = (synth.noise_batch(seed + 5, (hm.shape[0], 17, size // 4, size // 4)) * 20 + 25).astype(np.float32)
    assert [sha(hm), sha(off), sha(scl)] == list(g["in_sha"]), "synthetic input generator drifted (not a parity failure)"
    return hm, off, scl


def jitter_case_inputs(g):
    """Inputs of a jitter-head fixture (tools/gen_golden.py:jitter_case), sha-guarded."""
    batch, size, flip, seed = int(g["batch"]), int(g["size"]), bool(g["flip"]), int(g["seed"])
    hm, off = synth.synth_batch(seed, batch, size, size, flip=flip, n_persons=7)
    jit = ((synth.noise_batch(seed + 9, (hm.shape[0], 2, size // 4, size // 4)) - 0.5) * 3.0).astype(np.float32)
    assert [sha(hm), sha(off), sha(jit)] == list(g["in_sha"]), "synthetic input generator drifted (not a parity failure)"
    return hm, off, jit


def bilinear_case_inputs(g):
    """Inputs of a --resize-mode bilinear fixture (tools/gen_golden_bilinear.py), sha-guarded; scl is None without the
    keypoint-scale head."""
    batch, size, flip, seed = int(g["batch"]), int(g["size"]), bool(g["flip"]), int(g["seed"])
    hm, off = synth.synth_batch(seed, batch, size, size, flip=flip, n_persons=8)
    scl = None
    if int(g["with_scale"]):
        scl = (synth.noise_batch(seed + 5, (hm.shape[0], 17, size // 4, size // 4)) * 20 + 25).astype(np.float32)
    shas = [sha(hm), sha(off)] + ([sha(scl)] if scl is not None else [])
    assert shas == list(g["in_sha"]), "synthetic input generator drifted (not a parity failure)"
    return hm, off, scl


def load_case(name):
    """Golden case + regenerated inputs (sha-guarded against generator drift)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    n_persons = int(g["n_persons"])
    hm, off = synth.synth_batch(int(g["seed"]), int(g["batch"]), int(g["size"]), int(g["size"]),
                                flip=bool(g["flip"]), n_persons=None if n_persons < 0 else n_persons,
                                skeleton=case_skeleton(g))
    assert [sha(hm), sha(off)] == list(g["in_sha"]), "synthetic input generator drifted (not a parity failure)"
    return g, hm, off


def split_poses(g):
    out, o = [], 0
    for n in g["n_poses"]:
        out.append(g["poses"][o:o + n])
        o += n
    return out


def assert_limbs_match(ref, got, tol=1e-4, valid_only_thre=None):
    """Indices / coordinates / dist / len bit-exact, limb score within tol."""
    ref, got = np.asarray(ref), np.asarray(got)
    assert ref.shape == got.shape
    if valid_only_thre is not None:  # rows with a sub-threshold endpoint are don't-care
        ok = (ref[..., 2] >= valid_only_thre) & (ref[..., 5] >= valid_only_thre)
        ref, got = ref[ok], got[ok]
    assert (ref[..., EXACT_LIMB_COLS] == got[..., EXACT_LIMB_COLS]).all()
    assert np.abs(ref[..., 10] - got[..., 10]).max(initial=0.0) <= tol


def assert_poses_match(ref_list, got_list, tol=1e-4):
    assert len(ref_list) == len(got_list)
    for r, m in zip(ref_list, got_list):
        assert r.shape == m.shape, f"pose count {r.shape} vs {m.shape}"
        if r.size:
            assert (r[..., EXACT_POSE_COLS] == m[..., EXACT_POSE_COLS]).all()
            assert np.abs(r[..., 4] - m[..., 4]).max() <= tol
