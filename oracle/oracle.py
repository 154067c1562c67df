"""ctypes/numpy front-end of oracle/libog_oracle.so (C restatement, og_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of og_oracle.c.  Parity status:
pinned against the imported reference by tools/gen_golden.py and against
tests/golden/*.npz by tests/test_oracle_golden.py.

The wrappers mirror the reference call sequence of
decoder/factory.py:52-96 (PostProcess.generate_poses) so that a test reads
like the reference's own pipeline.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libog_oracle.so")

__all__ = [
    "build", "bicubic4", "bilinear4", "hmp_nms", "topk", "nms_topk", "collect_limbs",
    "greedy_group", "group_stats", "resize_cubic_u8", "shrink_mask_miss_u8", "flip_merge", "flip_cat", "encode_heatmaps", "encode_offsets", "encode_jitter", "decode",
]

_lib = None
_F = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_I64 = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_I32 = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile the C oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "og_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libog_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.ogo_bicubic4.argtypes = [_F, C.c_long, C.c_int, C.c_int, _F]
        L.ogo_bilinear4.argtypes = [_F, C.c_long, C.c_int, C.c_int, _F]
        L.ogo_hmp_nms.argtypes = [_F, C.c_long, C.c_int, C.c_int, _F]
        L.ogo_topk.argtypes = [_F, C.c_long, C.c_long, C.c_int, _F, _I64]
        L.ogo_topk.restype = C.c_int
        L.ogo_nms_topk.argtypes = [_F, C.c_long, C.c_int, C.c_int, C.c_int, _F, _I64]
        L.ogo_nms_topk.restype = C.c_int
        L.ogo_collect_limbs.argtypes = [_F, _I64, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        _I32, _I32, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, _F]
        L.ogo_collect_limbs_nd.argtypes = [_F, _I64, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           _I32, _I32, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, _F]
        L.ogo_collect_limbs_jit.argtypes = [_F, _I64, _F, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, _I32, _I32, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, _F]
        L.ogo_collect_limbs_ex.argtypes = [_F, _I64, _F, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           _I32, _I32, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, _F]
        L.ogo_flip_cat.argtypes = [_F, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   _I32, _I32, _I32, C.c_int, _F, _F]
        L.ogo_encode_heatmaps.argtypes = [_F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _F]
        L.ogo_encode_jitter.argtypes = [_F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _F]
        L.ogo_encode_offsets.argtypes = [_F, C.c_int, C.c_int, _I32, _I32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_float, _F, _F, _F, _F]
        L.ogo_greedy_group.argtypes = [_F, C.c_int, C.c_int, _I32, _I32, C.c_int, C.c_double, C.c_float,
                                       C.c_int, C.c_int, C.c_int, _F]
        L.ogo_greedy_group.restype = C.c_int
        L.ogo_flip_merge.argtypes = [_F, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     _I32, _I32, _I32, C.c_int, _F, _F]
        L.ogo_resize_cubic_u8.argtypes = [np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS"), C.c_int, C.c_int,
                                          np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS"), C.c_int, C.c_int]
        L.ogo_shrink_mask_miss_u8.argtypes = L.ogo_resize_cubic_u8.argtypes
        L.ogo_group_stats.argtypes = [np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS"), C.c_int]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def bicubic4(x):
    """x4 bicubic upsample of (..., h, w) -- decoder/factory.py:74-75."""
    x = _f32(x)
    h, w = x.shape[-2:]
    out = np.empty(x.shape[:-2] + (4 * h, 4 * w), np.float32)
    lib().ogo_bicubic4(x, x.size // (h * w), h, w, out)
    return out


def bilinear4(x):
    """x4 bilinear upsample of (..., h, w) -- decoder/factory.py:77-78."""
    x = _f32(x)
    h, w = x.shape[-2:]
    out = np.empty(x.shape[:-2] + (4 * h, 4 * w), np.float32)
    lib().ogo_bilinear4(x, x.size // (h * w), h, w, out)
    return out


def hmp_nms(x):
    """decoder/heatmap.py:15-35."""
    x = _f32(x)
    H, W = x.shape[-2:]
    out = np.empty_like(x)
    lib().ogo_hmp_nms(x, x.size // (H * W), H, W, out)
    return out


def topk(scores, k):
    """decoder/heatmap.py:38-49 -> (scores, inds, ys, xs) with floor-division ys."""
    s = _f32(scores)
    n, c, h, w = s.shape
    os_ = np.empty((n, c, k), np.float32)
    oi = np.empty((n, c, k), np.int64)
    if lib().ogo_topk(s, n * c, h * w, k, os_, oi) != 0:
        raise RuntimeError("k out of range")
    return os_, oi, oi // w, oi % w


def nms_topk(hm, k):
    """decoder/heatmap.py:52-59 (joint_dets)."""
    s = _f32(hm)
    n, c, h, w = s.shape
    os_ = np.empty((n, c, k), np.float32)
    oi = np.empty((n, c, k), np.int64)
    if lib().ogo_nms_topk(s, n * c, h, w, k, os_, oi) != 0:
        raise RuntimeError("k out of range")
    return os_, oi, oi // w, oi % w


def collect_limbs(scores, inds, offs, off_lowres, hw_shape, skeleton, thre, min_len, resize=1.0, vector_nd=2,
                  scales_hr=None, jitter_hr=None, use_jitter=True):
    """decoder/collect.py:62-236; `offs` low-res (bilinear-sampled) or hi-res (gathered); vector_nd=4 for the
    cat_flip_offs form (offs then has 4 components per limb)."""
    scores = _f32(scores)
    inds = np.ascontiguousarray(inds, dtype=np.int64)
    offs = _f32(offs)
    n, c, k = scores.shape
    H, W = hw_shape
    jf = np.array([a for a, _ in skeleton], np.int32)
    jt = np.array([b for _, b in skeleton], np.int32)
    L = len(skeleton)
    limbs = np.empty((n, L, k, 13), np.float32)
    if scales_hr is not None:
        scales_hr = _f32(scales_hr)
        assert scales_hr.shape == (n, c, H, W)
    if jitter_hr is not None:
        jitter_hr = _f32(jitter_hr)
        assert jitter_hr.shape == (n, 2, H, W) and H == W, 'the reference indexes the jitter maps [x][y]: square inputs only'
    lib().ogo_collect_limbs_jit(scores, inds, offs, int(bool(off_lowres)),
                                scales_hr.ctypes.data if scales_hr is not None else None,
                                jitter_hr.ctypes.data if jitter_hr is not None else None, int(bool(use_jitter)),
                                n, c, H, W, jf, jt, L, k, thre, min_len, resize, int(vector_nd), limbs)
    return limbs


def greedy_group(limbs, skeleton, n_keypoints, person_thre, dist_max, use_scale=False, sort_dim=2, mmax=None):
    """decoder/group.py:39-185 for one image: (L,K,13) -> (M,17,6)."""
    limbs = _f32(limbs)
    L, K, _ = limbs.shape
    assert L == len(skeleton), 'check the skeleton config and input limbs Tensor'
    jf = np.array([a for a, _ in skeleton], np.int32)
    jt = np.array([b for _, b in skeleton], np.int32)
    mmax = mmax or L * K
    poses = np.zeros((mmax, n_keypoints, 6), np.float32)
    m = lib().ogo_greedy_group(limbs, L, K, jf, jt, n_keypoints, float(person_thre), float(dist_max),
                               int(bool(use_scale)), int(sort_dim), mmax, poses)
    if m < 0:
        raise RuntimeError("mmax too small")
    return poses[:m].copy()


def group_stats(reset=False):
    """Coverage counters of greedy_group since the last reset (see og_oracle.c)."""
    out = np.zeros(9, np.int64)
    lib().ogo_group_stats(out, int(reset))
    names = ["phaseA", "phaseB", "phaseB_dup_row", "merges", "cross3", "nonreplaced_reset", "dup_a_merge",
             "merged_row_deleted", "zero_sum_column"]
    return dict(zip(names, out.tolist()))


def flip_merge(hm, off, kp_perm, limb_perm, reserve):
    """decoder/factory.py:98-146 (vector-addition form)."""
    hm, off = _f32(hm), _f32(off)
    n2, c, h, w = hm.shape
    L = off.shape[1] // 2
    n = n2 // 2
    ho = np.empty((n, c, h, w), np.float32)
    oo = np.empty((n, 2 * L, h, w), np.float32)
    res = np.array(reserve, np.int32)
    lib().ogo_flip_merge(hm, off, n, c, L, h, w, np.array(kp_perm, np.int32), np.array(limb_perm, np.int32),
                         res if len(res) else np.zeros(1, np.int32), len(reserve), ho, oo)
    return ho, oo


def flip_cat(hm, off, kp_perm, limb_perm, reserve):
    """decoder/factory.py:115-127 (cat_flip_offs=True): off_out (N, 4L, h, w)."""
    hm, off = _f32(hm), _f32(off)
    n2, c, h, w = hm.shape
    L = off.shape[1] // 2
    n = n2 // 2
    ho = np.empty((n, c, h, w), np.float32)
    oo = np.empty((n, 4 * L, h, w), np.float32)
    res = np.array(reserve, np.int32)
    lib().ogo_flip_cat(hm, off, n, c, L, h, w, np.array(kp_perm, np.int32), np.array(limb_perm, np.int32),
                       res if len(res) else np.zeros(1, np.int32), len(reserve), ho, oo)
    return ho, oo


def encode_heatmaps(joints, in_w, in_h, stride=4, sigma=7, clip_thre=0.01):
    """encoder/heatmap.py:125-197: joints (P,17,4) fp32 -> (17, h, w) Gaussian heatmaps."""
    joints = _f32(joints)
    p, n_kp = joints.shape[:2]
    hm = np.empty((n_kp, in_h // stride, in_w // stride), np.float32)
    lib().ogo_encode_heatmaps(joints, p, n_kp, in_w, in_h, stride, sigma, clip_thre, hm)
    return hm


def encode_jitter(joints, in_w, in_h, stride=4, fill_size=3):
    """encoder/heatmap.py:199-255: joints (P,17,4) -> (2, h, w) offsets to the nearest keypoint, inf outside."""
    joints = _f32(joints)
    p, n_kp = joints.shape[:2]
    jit = np.empty((2, in_h // stride, in_w // stride), np.float32)
    lib().ogo_encode_jitter(joints, p, n_kp, in_w, in_h, stride, fill_size, jit)
    return jit


def encode_offsets(joints, skeleton, sigmas, in_w, in_h, stride=4, fill_size=7, min_jscale=1.0):
    """encoder/offset.py:98-197: -> offsets (2L,h,w) inf outside, keypoint scales (17,h,w) nan outside, person scales
    (2L,h,w) 1 outside."""
    joints = _f32(joints)
    p, n_kp = joints.shape[:2]
    jf = np.array([a for a, _ in skeleton], np.int32)
    jt = np.array([b for _, b in skeleton], np.int32)
    L = len(skeleton)
    h, w = in_h // stride, in_w // stride
    off, sc, ps = np.empty((2 * L, h, w), np.float32), np.empty((n_kp, h, w), np.float32), np.empty((2 * L, h, w), np.float32)
    lib().ogo_encode_offsets(joints, p, n_kp, jf, jt, L, in_w, in_h, stride, fill_size, min_jscale,
                             np.asarray(sigmas, np.float32), off, sc, ps)
    return off, sc, ps


def resize_cubic_u8(img, new_h, new_w):
    """cv2.resize(img, (new_w, new_h), interpolation=cv2.INTER_CUBIC) for (h, w, 3) uint8 images, restated from OpenCV's
    published fixed-point algorithm (transforms/scale.py:27; PARITY UNPINNED: cv2 is absent, see og_oracle.c)."""
    img = np.ascontiguousarray(img, np.uint8)
    out = np.empty((new_h, new_w, 3), np.uint8)
    lib().ogo_resize_cubic_u8(img, img.shape[0], img.shape[1], out, new_h, new_w)
    return out


def shrink_mask_miss_u8(mask, stride):
    """encoder/heatmap.py:56-60: (h, w) uint8 mask_miss -> bool (h // stride, w // stride) = cv2.resize(fx = 1 / stride, INTER_CUBIC)
    / 255 > 0.7 (PARITY UNPINNED against cv2, like resize_cubic_u8)."""
    mask = np.ascontiguousarray(mask, np.uint8)
    nh, nw = int(round(mask.shape[0] / stride)), int(round(mask.shape[1] / stride))
    out = np.empty((nh, nw), np.uint8)
    lib().ogo_shrink_mask_miss_u8(mask, mask.shape[0], mask.shape[1], out, nh, nw)
    return out.astype(bool)


def decode(hm_lr, off_lr, skeleton, *, topk_k=32, thre_hmp=0.04, min_len=0.5, person_thre=0.04,
           dist_max=40.0, use_scale=False, sort_dim=2, flip=None, materialize_offsets=False, cat_flip_offs=False,
           scales_lr=None, jitter_lr=None, use_jitter=True, inter_mode='bicubic'):
    """PostProcess.generate_poses (decoder/factory.py:52-96) on low-res head outputs.

    inter_mode = --resize-mode (decoder/factory.py:151-153): the x4 resize of the heatmaps (:74-75) and of the
    keypoint-scale maps (:80-82); offsets and jitter maps are always bilinear (:77-78, :87-88).

    flip = (kp_perm, limb_perm, reserve) enables the flip-test merge first.
    Returns (poses list, dict of intermediates).
    """
    assert inter_mode in ('bicubic', 'bilinear')
    up = bicubic4 if inter_mode == 'bicubic' else bilinear4
    nd = 2
    if flip is not None and cat_flip_offs:
        hm_lr, off_lr = flip_cat(hm_lr, off_lr, *flip)
        nd = 4
    elif flip is not None:
        hm_lr, off_lr = flip_merge(hm_lr, off_lr, *flip)
    sc_hr = None
    if scales_lr is not None:  # keypoint-scale head: flip-averaged like the heatmaps (factory.py:141-144), then x4 bicubic
        scales_lr = _f32(scales_lr)
        if flip is not None:
            half = scales_lr.shape[0] // 2
            scales_lr = (scales_lr[:half] + scales_lr[half:, list(flip[0])][..., ::-1]) / np.float32(2)
        sc_hr = up(np.ascontiguousarray(scales_lr))
    jit_hr = None
    if jitter_lr is not None:  # jitter head: flip-averaged with x negated (factory.py:108-113), then x4 bilinear
        jitter_lr = _f32(jitter_lr)
        if flip is not None:
            half = jitter_lr.shape[0] // 2
            fl_j = jitter_lr[half:][..., ::-1].copy()
            fl_j[:, 0::2] *= np.float32(-1)
            jitter_lr = (jitter_lr[:half] + fl_j) / np.float32(2)
        jit_hr = bilinear4(np.ascontiguousarray(jitter_lr))
    hm_hr = up(hm_lr)
    n, c, H, W = hm_hr.shape
    sc, idx, _, _ = nms_topk(hm_hr, topk_k)
    if materialize_offsets:
        limbs = collect_limbs(sc, idx, bilinear4(off_lr), False, (H, W), skeleton, thre_hmp, min_len, vector_nd=nd,
                              scales_hr=sc_hr, jitter_hr=jit_hr, use_jitter=use_jitter)
    else:
        limbs = collect_limbs(sc, idx, off_lr, True, (H, W), skeleton, thre_hmp, min_len, vector_nd=nd, scales_hr=sc_hr,
                              jitter_hr=jit_hr, use_jitter=use_jitter)
    poses = [greedy_group(limbs[i], skeleton, c, person_thre, dist_max, use_scale, sort_dim) for i in range(n)]
    return poses, {"hm_hr": hm_hr, "scores": sc, "inds": idx, "limbs": limbs}
