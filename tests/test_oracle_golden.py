"""The CPU oracle against the committed golden vectors (generated from the imported reference
by tools/gen_golden.py).  CPU-only; pins the checker that the GPU parity tests rely on."""
import numpy as np
import pytest

import oracle
from offsetguided_amd import synth
from offsetguided_amd.config import coco_data as cd
from helpers import (FLAGS, case_flags, case_skeleton, GOLDEN, PIPE_CASES, assert_limbs_match, assert_poses_match, flip_tables, is_cat, load_case, jitter_case_inputs, scale_case_inputs,
                     sha, split_poses)


def test_stage_units():
    g = np.load(f"{GOLDEN}/stage_units.npz")
    x = synth.noise_batch(11, (2, 3, 24, 40))
    z = synth.noise_batch(12, (2, 3, 37, 53))
    assert [sha(x), sha(z)] == list(g["in_sha"])
    assert sha(oracle.bicubic4(x)) == str(g["bicubic_sha"])
    assert sha(oracle.bilinear4(x)) == str(g["bilinear_sha"])
    assert sha(oracle.hmp_nms(z)) == str(g["nms_sha"])
    s, i, ys, xs = oracle.topk(z, 9)
    assert (s == g["topk_scores"]).all() and (i == g["topk_inds"]).all()
    assert (ys == i // 53).all() and (xs == i % 53).all()


@pytest.mark.parametrize("name", PIPE_CASES)
def test_pipeline_case(name):
    g, hm, off = load_case(name)
    FLAGS, skel = case_flags(g), case_skeleton(g)
    flip = flip_tables(skel) if int(g["flip"]) else None
    cat = is_cat(g)
    merge = oracle.flip_cat if cat else oracle.flip_merge
    if flip:
        mh, mo = merge(hm, off, *flip)
        assert [sha(mh), sha(mo)] == list(g["merged_sha"])
    poses, mid = oracle.decode(hm, off, skel, topk_k=FLAGS["topk"], thre_hmp=FLAGS["thre_hmp"],
                               min_len=FLAGS["min_len"], person_thre=FLAGS["person_thre"],
                               dist_max=FLAGS["dist_max"], flip=flip, cat_flip_offs=cat)
    assert sha(mid["hm_hr"]) == str(g["hm_hr_sha"])
    assert (mid["scores"] == g["scores"]).all() and (mid["inds"] == g["inds"]).all()
    assert_limbs_match(g["limbs"], mid["limbs"])
    assert_poses_match(split_poses(g), poses)
    # materialised x4 bilinear offsets + plain gather give the same limbs (decoder/factory.py:77-78)
    merged_off = merge(hm, off, *flip)[1] if flip else off
    ohr = oracle.bilinear4(merged_off)
    assert sha(ohr) == str(g["off_hr_sha"])
    l2 = oracle.collect_limbs(mid["scores"], mid["inds"], ohr, False, mid["hm_hr"].shape[-2:],
                              skel, FLAGS["thre_hmp"], FLAGS["min_len"], vector_nd=4 if cat else 2)
    assert (l2 == mid["limbs"]).all()


def test_grouping_adversarial():
    g = np.load(f"{GOLDEN}/grouping_adversarial.npz")
    oracle.group_stats(True)
    for si, sk_name in enumerate(g["skeleton_names"]):
        sk = getattr(cd, str(sk_name))
        o = 0
        for limbs, (K, M), cfg in zip(g[f"limbs_{si}"], g[f"kn_{si}"], g[f"cfg_{si}"]):
            thre, dmax, use_scale, sort_dim = g["cfg_table"][cfg]
            got = oracle.greedy_group(limbs[:, :K], sk, 17, thre, dmax, bool(use_scale), int(sort_dim))
            ref = g[f"poses_{si}"][o:o + M]
            o += M
            assert got.shape == ref.shape and (got == ref).all()
    st = oracle.group_stats()
    # the stored cases exercise every branch of the grouping restatement
    for k in ("phaseA", "phaseB", "phaseB_dup_row", "merges", "cross3", "dup_a_merge", "merged_row_deleted"):
        assert st[k] > 0, k


def test_topk_tie_rule_and_errors():
    z = np.zeros((1, 1, 4, 5), np.float32)
    z[0, 0, 2, 3] = 1.0
    s, i, _, _ = oracle.nms_topk(z, 4)
    assert i.tolist() == [[[13, 0, 1, 2]]] and s.tolist() == [[[1.0, 0.0, 0.0, 0.0]]]
    with pytest.raises(RuntimeError):
        oracle.topk(z, 21)


@pytest.mark.parametrize("name", ["scale256", "scale256_flip"])
def test_scale_head_case(name):
    """Keypoint-scale head (include_scale / use_scale): poses incl. the exact scale column vs the reference's."""
    g = np.load(f"{GOLDEN}/{name}.npz")
    hm, off, scl = scale_case_inputs(g)
    flip = flip_tables() if int(g["flip"]) else None
    poses, _ = oracle.decode(hm, off, cd.COCO_PERSON_SKELETON, topk_k=FLAGS["topk"], thre_hmp=FLAGS["thre_hmp"],
                             min_len=FLAGS["min_len"], person_thre=FLAGS["person_thre"], dist_max=6.0, use_scale=True,
                             flip=flip, scales_lr=scl)
    assert_poses_match(split_poses(g), poses)
    for r, m in zip(split_poses(g), poses):
        assert (r[..., 3] == m[..., 3]).all()


BILINEAR_CASES = ["bilinear256", "bilinear256_flip", "bilinear256_scale", "bilinear256_scale_flip"]


@pytest.mark.parametrize("name", BILINEAR_CASES)
def test_resize_mode_bilinear_case(name):
    """--resize-mode bilinear (decoder/factory.py:151-153; heatmaps :74-75 and scale maps :80-82 resized bilinearly), with
    and without the keypoint-scale head and flip-test: the oracle vs poses of the reference run with that flag."""
    from helpers import bilinear_case_inputs
    g = np.load(f"{GOLDEN}/{name}.npz")
    hm, off, scl = bilinear_case_inputs(g)
    flip = flip_tables() if int(g["flip"]) else None
    poses, _ = oracle.decode(hm, off, cd.COCO_PERSON_SKELETON, topk_k=FLAGS["topk"], thre_hmp=FLAGS["thre_hmp"],
                             min_len=FLAGS["min_len"], person_thre=FLAGS["person_thre"], dist_max=float(g["dist_max"]),
                             use_scale=scl is not None, flip=flip, scales_lr=scl, inter_mode='bilinear')
    assert_poses_match(split_poses(g), poses)
    if scl is not None:
        for r, m in zip(split_poses(g), poses):
            assert (r[..., 3] == m[..., 3]).all()


def test_scored_offset_matches_reference_golden():
    """decoder/offset.py:8-43 (optional, off by default): our torch formulation is bit-identical to the reference's on
    the CPU (asserted at generation time and re-checked here through the stored hash); poses with scored_off=True."""
    import torch
    from offsetguided_amd.decoder.offset import pack_jtypes, scored_offset
    g = np.load(f"{GOLDEN}/scored256.npz")
    hm, off = synth.synth_batch(int(g["seed"]), int(g["batch"]), int(g["size"]), int(g["size"]), n_persons=6)
    assert [sha(hm), sha(off)] == list(g["in_sha"])
    jf, jt = pack_jtypes(cd.COCO_PERSON_SKELETON)
    scored = scored_offset(torch.from_numpy(hm), torch.from_numpy(off), jf, jt, kernel_size=3).numpy()
    assert sha(scored) == str(g["scored_sha"])
    poses, _ = oracle.decode(hm, scored, cd.COCO_PERSON_SKELETON, topk_k=FLAGS["topk"], thre_hmp=FLAGS["thre_hmp"],
                             min_len=FLAGS["min_len"], person_thre=FLAGS["person_thre"], dist_max=FLAGS["dist_max"])
    assert_poses_match(split_poses(g), poses)


@pytest.mark.parametrize("name", ["jitter256", "jitter256_flip"])
def test_jitter_head_case(name):
    """Jitter-offset head (include_jitter_offset / use_jitter_offset): refined guide points and end points, incl. the
    reference's [x][y] indexing of the refinement read; poses (fractional coordinates) exact vs the reference's."""
    g = np.load(f"{GOLDEN}/{name}.npz")
    hm, off, jit = jitter_case_inputs(g)
    flip = flip_tables() if int(g["flip"]) else None
    poses, _ = oracle.decode(hm, off, cd.COCO_PERSON_SKELETON, topk_k=FLAGS["topk"], thre_hmp=FLAGS["thre_hmp"],
                             min_len=FLAGS["min_len"], person_thre=FLAGS["person_thre"], dist_max=FLAGS["dist_max"],
                             flip=flip, jitter_lr=jit)
    assert_poses_match(split_poses(g), poses)


# ---- the reference-shaped Python restatement (bench.py's cpu_baseline, kind "restatement") ----
@pytest.mark.parametrize("name", ["pipe256_p0", "pipe256_p1", "pipe256_p6", "pipe256_p20", "pipe256_flip_p6", "pipe640", "pipe640_flip",
                                  "pipe256_omp16_flip_p6", "pipe256_omp44_k48_p20", "pipe256_omp25_p6", "pipe640_omp31_k48_flip"])
def test_restatement_pipeline_case(name):
    """oracle/restatement.py (torch-CPU ops + numpy grouping in a Pool, decoder/factory.py:52-96 shaped) against the
    reference's own outputs: candidate lists, limbs and poses."""
    import torch
    from oracle import restatement as rs
    g, hm, off = load_case(name)
    FLAGS, skel = case_flags(g), case_skeleton(g)
    flip = flip_tables(skel) if int(g["flip"]) else None
    dec = rs.Decoder(skel, 2, k=FLAGS["topk"], thre_hmp=FLAGS["thre_hmp"], min_len=FLAGS["min_len"],
                     person_thre=FLAGS["person_thre"], dist_max=FLAGS["dist_max"],
                     flip=(flip[0], flip[1], sorted(flip[2])) if flip else None)
    try:
        poses, limbs = dec.generate_poses(hm, off)
    finally:
        dec.close()
    assert_limbs_match(g["limbs"], limbs, valid_only_thre=FLAGS["thre_hmp"])
    assert_poses_match(split_poses(g), poses)
    t_hm = torch.from_numpy(hm)
    if flip:
        t_hm = rs.flip_augment(t_hm, torch.from_numpy(off), flip[0], flip[1], sorted(flip[2]))[0]
    s, i, _, _ = rs.topk_channel(rs.hmp_nms(rs.upsample(t_hm, torch.from_numpy(off)[:len(t_hm)])[0]), FLAGS["topk"])
    pos = g["scores"] > 0                                        # ties among the zero fillers are torch's business
    assert (s.numpy()[pos] == g["scores"][pos]).all() and (i.numpy()[pos] == g["inds"][pos]).all()


def test_restatement_grouping_adversarial():
    from oracle import restatement as rs
    g = np.load(f"{GOLDEN}/grouping_adversarial.npz")
    n = 0
    for si, sk_name in enumerate(g["skeleton_names"]):
        sk = getattr(cd, str(sk_name))
        o = 0
        for limbs, (K, M), cfg in zip(g[f"limbs_{si}"], g[f"kn_{si}"], g[f"cfg_{si}"]):
            thre, dmax, use_scale, sort_dim = g["cfg_table"][cfg]
            ref = g[f"poses_{si}"][o:o + M]
            o += M
            got = rs.group_skeletons(limbs[:, :K], sk, 17, thre, dmax, bool(use_scale), int(sort_dim))
            assert got.shape == ref.shape and (got == ref).all(), (str(sk_name), n)
            n += 1
    assert n >= 200
