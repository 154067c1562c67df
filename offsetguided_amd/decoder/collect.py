"""Candidate limb collection (reference decoder/collect.py:21-273) on the HIP kernels K1+K2."""
import logging
import os

import torch

from .. import _lib
from ..config.coco_data import COCO_KEYPOINTS, COCO_PERSON_SKELETON
from .heatmap import _topk_raw

LOG = logging.getLogger(__name__)


class LimbsCollect(object):
    """Pairs top-k keypoint candidates into limbs along the guiding offsets.

    Same constructor and `generate_limbs` contract as the reference class
    (decoder/collect.py:37-67).  The output rows are
    [x1, y1, v1, x2, y2, v2, ind1, ind2, len_delta, len_limb, limb_score, scale1, scale2].
    Keypoint-scale and jitter-offset heads (off in every published configuration) are supported:
    K2 samples the stride-4 maps at the peaks with the arithmetic of F.interpolate(x4).

    The whole of generate_limbs -- NMS, top-k and the pairing -- is ONE C call: two launches queued back to back (band
    top-k; merge + pairing), on hi-res heatmaps (og_generate_limbs_f32) or straight on the stride-4 head output with the x4
    bicubic inside the band kernel (og_generate_limbs_fused_f32, the production path).
    """

    def __init__(self, hmp_s, off_s, *, topk=40, thre_hmp=0.08, min_len=3,
                 include_jitter_offset=False, include_scale=False, use_jitter_offset=True,
                 keypoints=COCO_KEYPOINTS, skeleton=COCO_PERSON_SKELETON):
        self.hmp_s = hmp_s
        self.off_s = off_s
        self.resize_factor = off_s / hmp_s
        self.keypoints = keypoints
        self.skeleton = skeleton
        self.K = topk
        self.thre_hmp = thre_hmp
        self.min_len = min_len
        self.include_jitter_offset = include_jitter_offset
        self.include_scale = include_scale
        self.use_jitter_offset = use_jitter_offset
        self.jtypes_f, self.jtypes_t = self.pack_jtypes(skeleton)
        LOG.info('%d limbs, keypoint threshold %.4f, offset/heatmap unit ratio %.3f',
                 len(skeleton), thre_hmp, self.resize_factor)

    @staticmethod
    def pack_jtypes(skeleton):
        return [a for a, _ in skeleton], [b for _, b in skeleton]

    def _check_optional_heads(self, jomps_hr, scmps_hr, vector_nd):
        if vector_nd not in (2, 4):
            raise NotImplementedError('guiding offsets have 2 components, or 4 with cat_flip_offs')

    def _jitter(self, jomps):
        """The jitter maps only act when the head exists AND use_jitter_offset (collect.py:154, :212)."""
        return jomps if self.include_jitter_offset and self.use_jitter_offset and isinstance(jomps, torch.Tensor) else None

    def generate_limbs(self, hmps_hr, jomps_hr, offs_hr, scmps_hr, vector_nd=2):
        """(N,C,H,W) heatmaps + (N,2L,H,W) offsets at input resolution -> limbs (N,L,K,13)."""
        assert hmps_hr.shape[-2:] == offs_hr.shape[-2:], 'spatial resolution should be equal'
        self._check_optional_heads(jomps_hr, scmps_hr, vector_nd)
        scales = scmps_hr if self.include_scale and isinstance(scmps_hr, torch.Tensor) else None
        return self._collect(hmps_hr, offs_hr, off_is_lowres=False, vector_nd=vector_nd, scales=scales, scales_mode=1,
                             jitter=self._jitter(jomps_hr), jitter_mode=1)

    def generate_limbs_lowres(self, hmps_hr, offs_lr, vector_nd=2, scmps_lr=None, scale_inter='bicubic', jomps_lr=None):
        """Same result as generate_limbs(hmps_hr, [], F.interpolate(offs_lr, x4, 'bilinear'), [])
        without building the hi-res offset tensor: K2 samples it at the candidate peaks."""
        assert hmps_hr.shape[-2] == 4 * offs_lr.shape[-2] and hmps_hr.shape[-1] == 4 * offs_lr.shape[-1], \
            'spatial resolution should be equal'
        return self._collect(hmps_hr, offs_lr, off_is_lowres=True, vector_nd=vector_nd, scales=scmps_lr,
                             scales_mode=2 if scale_inter == 'bicubic' else 3, jitter=self._jitter(jomps_lr), jitter_mode=3)

    def generate_limbs_flip(self, hmps_hr, offs_pair_lr, limb_perm, reserve_mask):
        """generate_limbs_lowres on the flip-merged offsets WITHOUT merging them first: offs_pair_lr = the stride-4 offset head
        output for [images | mirrored images], (2N, 2L, h, w); every sampled tap is computed as PostProcess.flip_augment would
        have written it (decoder/factory.py:129-138).  2-component offsets, no scale / jitter head."""
        hmps_hr = _lib.require_device(hmps_hr, 'hmps_hr')
        offs = _lib.require_device(offs_pair_lr, 'offs')
        n, c, h, w = hmps_hr.shape
        n_limbs = len(self.skeleton)
        assert tuple(offs.shape) == (2 * n, 2 * n_limbs, h // 4, w // 4), 'offsets of [images | mirrored images] at stride 4'
        dev, lib = hmps_hr.device, _lib.load()
        limbs = torch.empty((n, n_limbs, self.K, 13), dtype=torch.float32, device=dev)
        jf, jt = _lib.int_table(self.jtypes_f, dev), _lib.int_table(self.jtypes_t, dev)
        with _lib.stage_timer('k1_generate_limbs', dev):
            ws = _lib.workspace(dev, lib.og_generate_limbs_workspace_bytes(n, c, h, w, self.K), 'limbs')   # zero-filled
            _lib.check(lib.og_generate_limbs_flip_f32(
                _lib.ptr(hmps_hr), _lib.ptr(offs), _lib.ptr(_lib.int_table(limb_perm, dev)), _lib.ptr(_lib.int_table(reserve_mask, dev)),
                n, c, h, w, _lib.ptr(jf), _lib.ptr(jt), n_limbs, self.K, float(self.thre_hmp), float(self.min_len),
                float(self.resize_factor), None, None, _lib.ptr(limbs), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), lib)
        return limbs

    def generate_limbs_fused_flip(self, hm_pair_lr, offs_pair_lr, kp_perm, limb_perm, reserve_mask):
        """generate_limbs_fused on the flip-merged maps WITHOUT merging them first (og_generate_limbs_fused_flip_f32): the stride-4
        head outputs of [images | mirrored images], (2N, C, h, w) and (2N, 2L, h, w); every heat-map source value and every offset tap
        is computed as PostProcess.flip_augment would have written it (decoder/factory.py:98-146).  2-component offsets, no scale /
        jitter head."""
        hm = _lib.require_device(hm_pair_lr, 'hmps')
        offs = _lib.require_device(offs_pair_lr, 'offs')
        n2, c, h, w = hm.shape
        n, n_limbs = n2 // 2, len(self.skeleton)
        assert n2 == 2 * n and tuple(offs.shape) == (n2, 2 * n_limbs, h, w), 'head outputs of [images | mirrored images] at stride 4'
        dev, lib = hm.device, _lib.load()
        limbs = torch.empty((n, n_limbs, self.K, 13), dtype=torch.float32, device=dev)
        jf, jt = _lib.int_table(self.jtypes_f, dev), _lib.int_table(self.jtypes_t, dev)
        with _lib.stage_timer('k1f_fused_limbs', dev):
            ws = _lib.workspace(dev, lib.og_generate_limbs_workspace_bytes(n, c, 4 * h, 4 * w, self.K), 'limbs')   # zero-filled
            _lib.check(lib.og_generate_limbs_fused_flip_f32(
                _lib.ptr(hm), _lib.ptr(_lib.int_table(kp_perm, dev)), _lib.ptr(offs), _lib.ptr(_lib.int_table(limb_perm, dev)),
                _lib.ptr(_lib.int_table(reserve_mask, dev)), n, c, h, w, _lib.ptr(jf), _lib.ptr(jt), n_limbs, self.K,
                float(self.thre_hmp), float(self.min_len), float(self.resize_factor), None, None, _lib.ptr(limbs), _lib.ptr(ws),
                ws.numel(), _lib.stream_ptr(dev)), lib)
        return limbs

    def generate_limbs_fused(self, hmps_lr, offs_lr, vector_nd=2, scmps_lr=None, scale_inter='bicubic', jomps_lr=None):
        """Same limbs as generate_limbs(F.interpolate(hmps_lr, x4, 'bicubic'), [], F.interpolate(offs_lr, x4,
        'bilinear'), []) with NEITHER hi-res tensor built: K1-fused upsamples inside the NMS kernel."""
        assert hmps_lr.shape[-2:] == offs_lr.shape[-2:], 'spatial resolution should be equal'
        return self._collect(hmps_lr, offs_lr, off_is_lowres=True, hm_is_lowres=True, vector_nd=vector_nd,
                             scales=scmps_lr, scales_mode=2 if scale_inter == 'bicubic' else 3,
                             jitter=self._jitter(jomps_lr), jitter_mode=3)

    def _collect(self, hmps_hr, offs, off_is_lowres, hm_is_lowres=False, vector_nd=2, scales=None, scales_mode=0,
                 jitter=None, jitter_mode=0):
        hmps_hr = _lib.require_device(hmps_hr, 'hmps_hr')
        offs = _lib.require_device(offs, 'offs')
        n, c, h, w = hmps_hr.shape
        if hm_is_lowres:
            h, w = 4 * h, 4 * w
        n_limbs = len(self.skeleton)
        # cat_flip_offs hands the 4-component offsets on as a (2N, 2L, h, w) view (decoder/factory.py:127);
        # like collect.py:73 only the memory order (N, L, vector_nd, h, w) matters
        assert offs.numel() == n * vector_nd * n_limbs * offs.shape[-2] * offs.shape[-1], \
            'offset channels must be vector_nd x number of limbs'
        dev = hmps_hr.device
        lib = _lib.load()
        if scales is None:
            scales_mode = 0
        else:  # keypoint-scale head (collect.py:111-122): maps at input resolution (mode 1) or the stride-4 head output
            scales = _lib.require_device(scales, 'scmps')
            expect = (n, c, h, w) if scales_mode == 1 else (n, c, h // 4, w // 4)
            assert tuple(scales.shape) == expect, f'scale maps {tuple(scales.shape)}, expected {expect}'
        if jitter is None:
            jitter_mode = 0
        else:  # jitter-offset head: two shared channels at input resolution (mode 1) or the stride-4 head output
            jitter = _lib.require_device(jitter, 'jomps')
            expect = (n, 2, h, w) if jitter_mode == 1 else (n, 2, h // 4, w // 4)
            assert tuple(jitter.shape) == expect, f'jitter maps {tuple(jitter.shape)}, expected {expect}'
        if jitter is not None and h != w:   # the reference indexes the refinement maps [x][y] (collect.py:158-165)
            raise NotImplementedError('the jitter-offset head needs square inputs (the reference indexes its maps [x][y])')
        limbs = torch.empty((n, n_limbs, self.K, 13), dtype=torch.float32, device=dev)
        jf, jt = _lib.int_table(self.jtypes_f, dev), _lib.int_table(self.jtypes_t, dev)
        # one bracket round the whole generate_limbs boundary (K1 + K2): what bench.py prices as "K1"
        with _lib.stage_timer('k1f_fused_limbs' if hm_is_lowres else 'k1_generate_limbs', dev):
            if not hm_is_lowres:
                ws = _lib.workspace(dev, lib.og_generate_limbs_workspace_bytes(n, c, h, w, self.K), 'limbs')   # zero-filled
                _lib.check(lib.og_generate_limbs_f32(
                    _lib.ptr(hmps_hr), _lib.ptr(offs), int(off_is_lowres), int(vector_nd),
                    _lib.ptr(scales) if scales is not None else None, int(scales_mode),
                    _lib.ptr(jitter) if jitter is not None else None, int(jitter_mode), n, c, h, w, _lib.ptr(jf), _lib.ptr(jt),
                    n_limbs, self.K, float(self.thre_hmp), float(self.min_len), float(self.resize_factor), None, None,
                    _lib.ptr(limbs), 0, _lib.ptr(ws), ws.numel(),
                    _lib.stream_ptr(dev)), lib)
                return limbs
            # K1-fused: the x4 bicubic runs inside the NMS kernel; ONE C call = two launches (band top-k; merge + pairing)
            ws = _lib.workspace(dev, lib.og_generate_limbs_workspace_bytes(n, c, h, w, self.K), 'limbs')   # zero-filled
            _lib.check(lib.og_generate_limbs_fused_f32(
                _lib.ptr(hmps_hr), _lib.ptr(offs), int(vector_nd), _lib.ptr(scales) if scales is not None else None, int(scales_mode),
                _lib.ptr(jitter) if jitter is not None else None, int(jitter_mode), n, c, h // 4, w // 4, _lib.ptr(jf), _lib.ptr(jt),
                n_limbs, self.K, float(self.thre_hmp), float(self.min_len), float(self.resize_factor), None, None,
                _lib.ptr(limbs), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), lib)
        return limbs
