"""HeatMaps / OffsetMaps (encoder/heatmap.py:11-92, encoder/offset.py:11-69) and encoder_cli / encoder_factory /
factory_heads (encoder/factory.py:15-129).

The reference encodes one image at a time with numpy inside dataloader workers; here a whole batch of annotations
(joints (N,P,17,4) fp32 [x, y, v, scale] in input pixels, padded to P persons) is encoded by two kernel launches on the
training device.  `enc(anns, meta, mask_miss)` keeps the per-image call of the reference (numpy in, tensors out);
`enc.encode_batch(joints, n_persons, mask_miss)` is the batched device form.

mask_miss: the reference's input, the full-resolution uint8 mask (0 / 255), is shrunk on the device like its
cv2.resize(fx = 1 / stride, INTER_CUBIC) / 255 > 0.7 (heatmap.py:56-60; OpenCV's published 8-bit algorithm, parity with cv2
itself unpinned: no cv2 in the image); a boolean mask at OUTPUT resolution (True = labelled) passes through; None =
"everything labelled"."""
import logging
import re

import numpy as np
import torch

from .. import _lib
from ..config.coco_data import (COCO_KEYPOINTS, COCO_PERSON_SIGMAS, COCO_PERSON_SKELETON,
                                COCO_PERSON_WITH_REDUNDANT_SKELETON, DENSER_COCO_PERSON_SKELETON,
                                KINEMATIC_TREE_SKELETON, REDUNDANT_CONNECTIONS)

LOG = logging.getLogger(__name__)


def _batch_joints(joints, n_persons, device):
    joints = torch.as_tensor(joints, dtype=torch.float32)
    if joints.dim() == 3:
        joints = joints[None]
    assert joints.dim() == 4 and joints.shape[-1] == 4, 'joints must be (N, P, n_keypoints, 4) rows [x, y, v, scale]'
    if joints.shape[1] == 0:  # nobody annotated: one all-unlabelled person keeps the buffers non-empty
        joints = torch.zeros((joints.shape[0], 1) + tuple(joints.shape[2:]), dtype=torch.float32)
    joints = _lib.require_device(joints.to(device), 'joints')
    if n_persons is not None:
        n_persons = torch.as_tensor(n_persons, dtype=torch.int32).to(device).contiguous()
        assert n_persons.shape == (joints.shape[0],)
    return joints, n_persons


def _mask(mask_miss, n, h, w, device, stride):
    """mask_miss -> bool (n,1,h,w) on the device.  A boolean mask at output resolution passes through; the reference's input,
    the full-resolution uint8 mask (0 / 255), is shrunk on the device like cv2.resize(fx = 1 / stride, INTER_CUBIC) / 255 >
    0.7 (encoder/heatmap.py:56-60; og_shrink_mask_miss_u8, parity with cv2 unpinned)."""
    if mask_miss is None:
        return torch.ones((n, 1, h, w), dtype=torch.bool, device=device)
    mask = torch.as_tensor(mask_miss)
    if mask.dtype == torch.bool and tuple(mask.shape[-2:]) == (h, w):
        return mask.reshape(n, 1, h, w).to(device)
    if mask.dtype != torch.uint8 or (round(mask.shape[-2] / stride), round(mask.shape[-1] / stride)) != (h, w):
        raise ValueError(f'mask_miss: bool at output resolution ({h}, {w}) or uint8 at input resolution, got {mask.dtype} '
                         f'{tuple(mask.shape)}')
    from .. import _lib
    lib = _lib.load()
    full = mask.reshape(n, mask.shape[-2], mask.shape[-1]).contiguous().to(device)
    out = torch.empty((n, 1, h, w), dtype=torch.uint8, device=device)
    _lib.check(lib.og_shrink_mask_miss_u8(_lib.ptr(full), n, full.shape[1], full.shape[2], int(stride), _lib.ptr(out),
                                          _lib.stream_ptr(device)), lib)
    return out.bool()


class _Encoder:
    def __init__(self, input_size, stride, device=None):
        assert isinstance(input_size, (int, list)), input_size
        assert stride != 0, 'stride can not be zero'
        self.input_size = input_size if isinstance(input_size, list) else [input_size] * 2   # (w, h)
        self.in_out_scale = 1 / stride
        self.stride = stride
        self.device = torch.device(device if device is not None else 'cuda:0')
        if int(stride) != stride:
            raise NotImplementedError('non-integer network strides are not supported')

    def __call__(self, anns, meta, mask_miss=None):
        """Per-image form of the reference (anns (P, n_keypoints, 4) numpy): same tuple, on the device."""
        assert meta['width_height'][0] == self.input_size[0], 'raw data!'
        out = self.encode_batch(np.asarray(anns, np.float32)[None], None, None if mask_miss is None else mask_miss[None],
                                joint_num=meta['joint_num'])
        return tuple(t[0] if t.numel() else t for t in out)


class HeatMaps(_Encoder):
    """Gaussian keypoint heatmaps (+ reversed background) -- encoder/heatmap.py:11-92."""
    clip_thre = 0.01
    sigma = 7
    n_keypoints = 17
    keypoints = COCO_KEYPOINTS
    include_background = True
    include_jitter_offset = True
    fill_jitter_size = 3

    def encode_batch(self, joints, n_persons=None, mask_miss=None, joint_num=None):
        """-> (heatmaps (N,17,h,w), background (N,1,h,w) or empty, jitter maps (N,2,h,w) or empty, mask_miss (N,1,h,w))."""
        joints, n_persons = _batch_joints(joints, n_persons, self.device)
        n, p, n_kp, _ = joints.shape
        assert self.n_keypoints == (joint_num or n_kp) == n_kp, \
            'not implemented! n_keypoints set by command parse args mismatches the COCO config '
        w, h = self.input_size[0] // self.stride, self.input_size[1] // self.stride
        dev = joints.device
        lib = _lib.load()
        hm = torch.empty((n, n_kp, h, w), dtype=torch.float32, device=dev)
        bg = torch.empty((n, 1, h, w), dtype=torch.float32, device=dev) if self.include_background else None
        _lib.check(lib.og_encode_heatmaps_f32(
            _lib.ptr(joints), _lib.ptr(n_persons) if n_persons is not None else None, n, p, n_kp,
            self.input_size[0], self.input_size[1], int(self.stride), int(self.sigma), float(self.clip_thre),
            _lib.ptr(hm), _lib.ptr(bg) if bg is not None else None, _lib.stream_ptr(dev)), lib)
        empty = torch.tensor([], device=dev)
        jit = empty
        if self.include_jitter_offset:   # heatmap.py:52-55
            jit = torch.empty((n, 2, h, w), dtype=torch.float32, device=dev)
            _lib.check(lib.og_encode_jitter_f32(
                _lib.ptr(joints), _lib.ptr(n_persons) if n_persons is not None else None, n, p, n_kp,
                self.input_size[0], self.input_size[1], int(self.stride), int(self.fill_jitter_size), _lib.ptr(jit),
                _lib.stream_ptr(dev)), lib)
        return hm, (bg if bg is not None else empty), jit, _mask(mask_miss, n, h, w, dev, self.stride)


class OffsetMaps(_Encoder):
    """Guiding offsets between adjacent keypoints, keypoint scales, person scales -- encoder/offset.py:11-69."""
    fill_scale_size = 7
    min_jscale = 1.0
    skeleton = COCO_PERSON_SKELETON
    include_scale = True

    def encode_batch(self, joints, n_persons=None, mask_miss=None, joint_num=None):
        """-> (offsets (N,2L,h,w) inf where undefined, keypoint scales (N,17,h,w) nan where undefined or empty,
        person scales (N,2L,h,w), mask_miss (N,1,h,w) bool)."""
        joints, n_persons = _batch_joints(joints, n_persons, self.device)
        n, p, n_kp, _ = joints.shape
        assert (joint_num or n_kp) == n_kp, 'num of joints mismatch'
        w, h = self.input_size[0] // self.stride, self.input_size[1] // self.stride
        dev = joints.device
        lib = _lib.load()
        n_limbs = len(self.skeleton)
        off = torch.empty((n, 2 * n_limbs, h, w), dtype=torch.float32, device=dev)
        ps = torch.empty((n, 2 * n_limbs, h, w), dtype=torch.float32, device=dev)
        sc = torch.empty((n, n_kp, h, w), dtype=torch.float32, device=dev) if self.include_scale else None
        sig = _sigmas(dev)
        _lib.check(lib.og_encode_offsets_f32(
            _lib.ptr(joints), _lib.ptr(n_persons) if n_persons is not None else None, n, p, n_kp,
            _lib.ptr(_lib.int_table([a for a, _ in self.skeleton], dev)),
            _lib.ptr(_lib.int_table([b for _, b in self.skeleton], dev)), n_limbs,
            self.input_size[0], self.input_size[1], int(self.stride), int(self.fill_scale_size), float(self.min_jscale),
            _lib.ptr(sig), _lib.ptr(off), _lib.ptr(sc) if sc is not None else None, _lib.ptr(ps), _lib.stream_ptr(dev)), lib)
        return off, (sc if sc is not None else torch.tensor([], device=dev)), ps, _mask(mask_miss, n, h, w, dev, self.stride)


_sigma_cache = {}


def _sigmas(dev):
    t = _sigma_cache.get(dev.index)
    if t is None:
        t = _sigma_cache[dev.index] = torch.tensor(COCO_PERSON_SIGMAS, dtype=torch.float32, device=dev)
    return t


def encoder_cli(parser):
    """Same flags and defaults as encoder/factory.py:15-37."""
    group = parser.add_argument_group('heatmap encoder')
    group.add_argument('--gaussian-clip-thre', default=HeatMaps.clip_thre, type=float,
                       help='Gaussian responses below this value are cut to zero')
    group.add_argument('--sigma', default=HeatMaps.sigma, type=int, help='standard deviation of the Gaussian peaks')
    group.add_argument('--fill-jitter-size', default=HeatMaps.fill_jitter_size, type=int,
                       help='diameter of the area round a keypoint that holds the jitter refinement offset')
    group = parser.add_argument_group('offsetmap and scalemap encoder')
    group.add_argument('--fill-scale-size', default=OffsetMaps.fill_scale_size, type=int,
                       help='diameter of the area round a keypoint that holds its scale and guiding offset')
    group.add_argument('--min_jscale', default=OffsetMaps.min_jscale, type=float, help='minimum keypoint scale')


def encoder_factory(args, strides=None, device=None):
    """encoder/factory.py:40-61: configure the classes from the parsed flags, build one encoder per head."""
    if not strides:
        strides = [4, 4, 4]
    HeatMaps.clip_thre = args.gaussian_clip_thre
    HeatMaps.sigma = args.sigma
    HeatMaps.include_background = args.include_background
    HeatMaps.include_jitter_offset = args.include_jitter_offset
    HeatMaps.fill_jitter_size = args.fill_jitter_size
    OffsetMaps.fill_scale_size = args.fill_scale_size
    OffsetMaps.min_jscale = args.min_jscale
    OffsetMaps.include_scale = args.include_scale
    return factory_heads(args.headnets, args.square_length, strides, device)


def factory_heads(headnames, square_length, strides, device=None):
    """encoder/factory.py:55-72: one encoder per head name (nested lists = multi-task, as in the reference)."""
    if isinstance(headnames[0], (list, tuple)):
        return [factory_heads(names, square_length, task_strides, device) for names, task_strides in zip(headnames, strides)]
    return [factory_head(name, square_length, stride, device) for name, stride in zip(headnames, strides)]


_OFFSET_SKELETONS = {  # encoder/factory.py:104-121
    'omp': COCO_PERSON_SKELETON, 'omp19': COCO_PERSON_SKELETON, 'omps': COCO_PERSON_SKELETON,
    'offset': COCO_PERSON_SKELETON, 'offsets': COCO_PERSON_SKELETON, 'omp16': KINEMATIC_TREE_SKELETON,
    'omp31': COCO_PERSON_WITH_REDUNDANT_SKELETON, 'omp44': DENSER_COCO_PERSON_SKELETON,
    'omp25': REDUNDANT_CONNECTIONS, 'omps25': REDUNDANT_CONNECTIONS,
}


def factory_head(head_name, square_length, stride, device=None):
    """encoder/factory.py:75-129."""
    m = re.match('hmp[s]?([0-9]+)$', head_name)
    if head_name in ('hmp', 'hmps', 'heatmap', 'heatmaps') or m is not None:
        n_keypoints, keypoints = 17, COCO_KEYPOINTS
        if m is not None:
            n_keypoints, keypoints = int(m.group(1)), None
            assert n_keypoints == 17, f'{n_keypoints} keypoints not supported'
        LOG.info('selected encoder: Heatmap for %s of stride %d computed by %d keypoints', head_name, stride, n_keypoints)
        HeatMaps.n_keypoints = n_keypoints
        HeatMaps.keypoints = keypoints
        return HeatMaps(square_length, stride, device)
    if head_name in ('omp', 'omps', 'offset', 'offsets') or re.match('omp[s]?([0-9]+)$', head_name) is not None:
        if head_name not in _OFFSET_SKELETONS:
            raise Exception('unknown skeleton type of head')
        OffsetMaps.skeleton = _OFFSET_SKELETONS[head_name]
        LOG.info('selected encoder: Offset for %s of stride %d computed by %d limb connections',
                 head_name, stride, len(OffsetMaps.skeleton))
        return OffsetMaps(square_length, stride, device)
    raise Exception('unknown head to create an encoder: {}'.format(head_name))
