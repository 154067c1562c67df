"""PostProcess / decoder_cli / decoder_factory -- the decoder's drop-in boundary
(reference decoder/factory.py:21-267).

generate_poses keeps the reference signature and return type (list of float32 (M,17,6) arrays)
but runs the whole post-processing on the GPU:
    K0 flip merge -> K1a bicubic x4 -> K1 NMS+top-k -> K2 limb collection (offsets sampled from
    the stride-4 map, the hi-res offset tensor is never built) -> K3 greedy grouping,
with a single device->host copy of the finished poses.  No multiprocessing.Pool is created.
"""
import logging
import os
import re

import torch

from .. import _lib
from .. import config
from ..config.coco_data import (COCO_KEYPOINTS, COCO_PERSON_SKELETON, COCO_PERSON_WITH_REDUNDANT_SKELETON,
                                DENSER_COCO_PERSON_SKELETON, KINEMATIC_TREE_SKELETON, REDUNDANT_CONNECTIONS)
from ..utils import boolean_string
from .collect import LimbsCollect
from .group import GreedyGroup
from .offset import pack_jtypes, scored_offset

LOG = logging.getLogger(__name__)


def upsample4_flip(hm_pair, kp_perm):
    """F.interpolate((hm[:N] + flip(hm[N:])[:, kp_perm]) / 2, scale_factor=4, mode='bicubic') in ONE pass over the stride-4 maps of
    [images | mirrored images]: flip_augment's heatmap merge (decoder/factory.py:101-106) rides on the loads of the upsampling."""
    x = _lib.require_device(hm_pair, 'feature map')
    lib = _lib.load()
    n2, c, h, w = x.shape
    out = torch.empty((n2 // 2, c, 4 * h, 4 * w), dtype=torch.float32, device=x.device)
    with _lib.stage_timer('k1a_upsample', x.device):
        _lib.check(lib.og_upsample_bicubic4_flip_f32(_lib.ptr(x), _lib.ptr(_lib.int_table(kp_perm, x.device)), n2 // 2, c, h, w,
                                                     _lib.ptr(out), _lib.stream_ptr(x.device)), lib)
    return out


def upsample4(x, mode):
    """F.interpolate(x, scale_factor=4, mode=mode) with torch-CPU fp32 rounding, as a HIP kernel."""
    x = _lib.require_device(x, 'feature map')
    lib = _lib.load()
    n, c, h, w = x.shape
    out = torch.empty((n, c, 4 * h, 4 * w), dtype=torch.float32, device=x.device)
    fn = {'bicubic': lib.og_upsample_bicubic4_f32, 'bilinear': lib.og_upsample_bilinear4_f32}[mode]
    with _lib.stage_timer('k1a_upsample', x.device):
        _lib.check(fn(_lib.ptr(x), n * c, h, w, _lib.ptr(out), _lib.stream_ptr(x.device)), lib)
    return out


class _HostSlot:
    """One pinned landing area for (poses, counts | status); `busy` from submit() until its handle has been read."""

    def __init__(self, poses_shape, meta_shape):
        self.poses = torch.empty(poses_shape, dtype=torch.float32).pin_memory()
        self.meta = torch.empty(meta_shape, dtype=torch.int32).pin_memory()
        self.busy = False
        self.orphan = None    # event of a handle that was dropped unread while its copy was still in flight


class PendingPoses:
    """Result handle of PostProcess.submit(): the decode is queued on the stream, the finished
    poses travel to pinned host memory asynchronously; result() waits for that copy only.  Handles may be resolved in
    any order and any number may be outstanding: each owns its pinned slot until it has been read (or dropped)."""

    def __init__(self, proc, limbs, poses, meta, slot, event):
        self._proc, self._limbs = proc, limbs
        self._poses, self._meta = poses, meta
        self._slot, self._event = slot, event
        self._out = None

    def result(self):
        if self._out is not None:
            return self._out
        self._event.synchronize()
        n = self._limbs.shape[0]
        meta = self._slot.meta.numpy()
        counts, status = meta[:n], meta[n:]
        if status.any():  # partial-skeleton table overflowed (rare): redo with a larger table
            self._out = self._proc.limb_group.group_batch(self._limbs)
        else:
            host = self._slot.poses.numpy()
            self._out = [host[i, :counts[i]].copy() for i in range(n)]
        self._release()
        return self._out

    def _release(self):
        if self._slot is not None:
            self._slot.busy = False       # the copies above own their data: the slot may take the next batch
            self._slot = None
        self._limbs = self._poses = self._meta = None   # device tensors are no longer needed

    def __del__(self):
        # A handle dropped unread: a finalizer never waits on the GPU (it would stall whichever thread runs the garbage
        # collector, and raise at interpreter shutdown).  If the copy that targets the slot has finished the slot is free again;
        # otherwise the slot keeps the event and submit() takes it back once the event has completed.
        slot = getattr(self, '_slot', None)
        if slot is not None:
            try:
                if self._event.query():
                    slot.busy = False
                else:
                    slot.orphan = self._event
            except Exception:  # noqa: BLE001  torch already torn down
                pass


class PostProcess(torch.nn.Module):
    _made = 0          # instances that took a side stream (which of the process's K3 streams the next one gets)

    def __init__(self, batch_size, hmp_stride, off_stride, inter_mode, keypoints, skeleton,
                 limb_collector, limb_grouper, include_scale=False, include_jitter_offset=False,
                 hmp_index=0, omp_index=1, feat_stage=-1):
        super().__init__()
        if hmp_stride != 4 or off_stride != 4:
            raise NotImplementedError('the HIP decoder is built for the stride-4 heads of Hourglass-104')
        if inter_mode not in ('bicubic', 'bilinear'):
            raise ValueError(f'unknown resize mode {inter_mode}')
        self.batch_size = batch_size
        self.inter_mode = inter_mode
        self.hmp_stride = hmp_stride
        self.off_stride = off_stride
        self.keypoints = keypoints
        self.skeleton = skeleton
        self.limb_collect = limb_collector
        self.limb_group = limb_grouper
        self.hmp_index = hmp_index
        self.omp_index = omp_index
        self.feat_stage = feat_stage
        self.include_scale = include_scale
        self.include_jitter_offset = include_jitter_offset
        self.keypoints_flips = config.heatmap_hflip(keypoints)
        self.limbs_flips = config.offset_hflip(keypoints, skeleton)
        self.worker_pool = None  # grouping runs on the device; kept as an attribute for API parity
        # True (default since round 5): K1-fused -- the x4 bicubic runs inside the NMS kernel, neither hi-res tensor of
        # decoder/factory.py:74-88 is built (identical results, 16x less HBM traffic, ~50 us instead of ~105 us per bs8 batch).
        # False / OG_FUSED_UPSAMPLE=0: K1a materialises the hi-res heatmaps and K1 streams them -- the reference's structure and the
        # HBM-roofline benchmark mode.
        self.fused_upsample = os.environ.get('OG_FUSED_UPSAMPLE', '1') != '0'
        # flip-test (2-component offsets, no scale / jitter head): flip_augment's merge rides on the loads of K1a and on the
        # offset sampling of K1 instead of running as its own pass (K0, og_flip_merge_f32); identical results
        self.fold_flip = os.environ.get('OG_FOLD_FLIP', '1') != '0'
        # submit(): grouping (one workgroup per image, latency-bound) and the pose D2H copy run on their own stream, so
        # the caller's next launches (the following batch's backbone) do not queue behind them
        self.group_on_side_stream = True   # (an attribute, no longer an environment switch: measured best since round 2)
        if 'OG_GROUP_SIDE_STREAM' in os.environ:
            LOG.warning('OG_GROUP_SIDE_STREAM is no longer read (removed in round 5): set PostProcess.group_on_side_stream instead')
        self._side = {}
        self._pinned = {}   # (poses shape, meta shape) -> [_HostSlot]: pinned landing areas of submit()
        LOG.info('decode stage %d features (heatmap head %d, offset head %d), %s heatmap resize, '
                 'device-resident grouping', feat_stage, hmp_index, omp_index, inter_mode)

    # ---- reference API -------------------------------------------------------------------
    def generate_poses(self, features, flip_test=False, cat_flip_offs=False, scored_off=False):
        limbs = self.generate_limbs(features, flip_test, cat_flip_offs, scored_off)
        return self.limb_group.group_batch(limbs)

    def submit(self, features, flip_test=False, cat_flip_offs=False, scored_off=False):
        """Asynchronous generate_poses: returns a PendingPoses; call .result() later.  Lets the
        caller queue the next batch's backbone before blocking on this batch's poses."""
        limbs = self.generate_limbs(features, flip_test, cat_flip_offs, scored_off)
        dev = limbs.device
        cur = torch.cuda.current_stream(dev)
        stream = cur
        if self.group_on_side_stream:
            stream = self._side.get(dev.index)
            if stream is None:
                PostProcess._made += 1       # (a stream of the process's own: torch's pool of 32 repeats, _lib.dedicated_stream)
                stream = self._side[dev.index] = _lib.dedicated_stream(dev, ('k3', PostProcess._made % 8))
            stream.wait_stream(cur)
            limbs.record_stream(stream)
        with torch.cuda.stream(stream):
            poses, meta = self.limb_group.group_device(limbs)
            # a pinned slot nobody is waiting on (the pool grows with the number of outstanding handles: a third submit()
            # before the first result() gets a third slot instead of overwriting the first batch's landing area)
            pool = self._pinned.get((tuple(poses.shape), tuple(meta.shape)))
            if pool is None:    # three slots at once, at the first batch of a shape: a pinned allocation costs 50-60 ms of host time,
                                # and a pipelined caller (run_images: one batch ahead) needs the third one a few batches in
                pool = self._pinned[(tuple(poses.shape), tuple(meta.shape))] = [_HostSlot(poses.shape, meta.shape) for _ in range(3)]
            for sl in pool:     # slots of handles that were dropped unread: free once their copy has landed
                if sl.orphan is not None and sl.orphan.query():
                    sl.busy, sl.orphan = False, None
            slot = next((sl for sl in pool if not sl.busy), None)
            if slot is None:
                slot = _HostSlot(poses.shape, meta.shape)
                pool.append(slot)
            slot.busy = True
            slot.poses.copy_(poses, non_blocking=True)
            slot.meta.copy_(meta, non_blocking=True)
            event = torch.cuda.Event()
            event.record(stream)
        return PendingPoses(self, limbs, poses, meta, slot, event)

    def flip_augment(self, hmps, jomps, offs, scmps, cat_flip_offs, vector_nd):
        """Merge the predictions for [images, mirrored images] (decoder/factory.py:98-146)."""
        hmps = _lib.require_device(hmps, 'hmps')
        offs = _lib.require_device(offs, 'offs')
        n2, c, h, w = hmps.shape
        n, n_limbs = n2 // 2, offs.shape[1] // 2
        dev = hmps.device
        lib = _lib.load()
        keep = [1 if l in self.limbs_flips[1] else 0 for l in range(n_limbs)]
        hm_out = torch.empty((n, c, h, w), dtype=torch.float32, device=dev)
        if cat_flip_offs:  # factory.py:115-127: keep both offsets per limb, (N, L, 4, h, w) in memory
            vector_nd *= 2
        off_out = torch.empty((n, vector_nd * n_limbs, h, w), dtype=torch.float32, device=dev)
        entry = lib.og_flip_cat_f32 if cat_flip_offs else lib.og_flip_merge_f32
        with _lib.stage_timer('k0_flip_merge', dev):
            _lib.check(entry(
                _lib.ptr(hmps), _lib.ptr(offs), n, c, n_limbs, h, w,
                _lib.ptr(_lib.int_table(self.keypoints_flips, dev)),
                _lib.ptr(_lib.int_table(self.limbs_flips[0], dev)),
                _lib.ptr(_lib.int_table(keep, dev)), _lib.ptr(hm_out), _lib.ptr(off_out), _lib.stream_ptr(dev)), lib)
        if cat_flip_offs:
            off_out = off_out.view(n2, -1, h, w)  # the reference's (odd) shape of the same memory, factory.py:127
        if self.include_jitter_offset and isinstance(jomps, torch.Tensor):  # factory.py:108-113, same ops (exact)
            jomps = _lib.require_device(jomps, 'jomps')
            flipped = torch.flip(jomps[n:], [-1])
            flipped[:, ::2] *= -1
            jomps = (jomps[:n] + flipped) / 2
        if self.include_scale and isinstance(scmps, torch.Tensor):  # factory.py:141-144, same ops (exact)
            scmps = _lib.require_device(scmps, 'scmps')
            scmps = (scmps[:n] + torch.flip(scmps[n:], [-1])[:, self.keypoints_flips]) / 2
        return hm_out, jomps, off_out, scmps, vector_nd

    # ---- device-resident pieces ----------------------------------------------------------
    def generate_limbs(self, features, flip_test=False, cat_flip_offs=False, scored_off=False):
        """Everything up to (not including) grouping; returns the (N, L, K, 13) limbs tensor."""
        out_hmps, _, out_jomps = features[self.hmp_index]
        out_offsets, _, out_scales = features[self.omp_index]
        hmps = out_hmps[self.feat_stage]
        jomps = out_jomps[self.feat_stage]
        offs = out_offsets[self.feat_stage]
        scmps = out_scales[self.feat_stage]
        vector_nd = 2
        if (flip_test and self.fold_flip and not cat_flip_offs and not scored_off
                and self.inter_mode == 'bicubic' and not (self.include_scale and isinstance(scmps, torch.Tensor))
                and not (self.include_jitter_offset and isinstance(jomps, torch.Tensor))):
            n_limbs = offs.shape[1] // 2
            keep = [1 if l in self.limbs_flips[1] else 0 for l in range(n_limbs)]
            if self.fused_upsample:
                try:
                    return self.limb_collect.generate_limbs_fused_flip(hmps, offs, self.keypoints_flips, self.limbs_flips[0], keep)
                except _lib.OgError as e:
                    # merge + pairing of the folded form keeps a plane's lists in LDS: beyond k ~ 226 at 640 x 640 (half that for inputs
                    # twice as tall) it does not fit and the C side says OG_EUNSUPPORTED.  The unfolded route below (flip_augment as its own
                    # pass, then K1-fused with its collect-kernel fallback) serves the same request with the same results.
                    if f'(code {_lib.OG_EUNSUPPORTED})' not in str(e):
                        raise
            else:
                hmps_hr = upsample4_flip(hmps, self.keypoints_flips)
                return self.limb_collect.generate_limbs_flip(hmps_hr, offs, self.limbs_flips[0], keep)
        if flip_test:
            hmps, jomps, offs, scmps, vector_nd = self.flip_augment(hmps, jomps, offs, scmps, cat_flip_offs, vector_nd)
        if scored_off:
            if vector_nd != 2:
                raise NotImplementedError('scored_off needs 2-component offsets (the reference fails here as well)')
            jf, jt = pack_jtypes(self.skeleton)
            offs = scored_offset(hmps.float(), offs.float(), jf, jt, kernel_size=3)
        # keypoint-scale head: sampled at the peaks from the stride-4 map, as F.interpolate(mode=inter_mode) would give
        scl = scmps.float().contiguous() if self.include_scale and isinstance(scmps, torch.Tensor) else None
        # jitter-offset head: bilinear x4 (factory.py:84-88), sampled where K2 needs it
        jit = jomps.float().contiguous() if self.include_jitter_offset and isinstance(jomps, torch.Tensor) else None
        if self.fused_upsample and self.inter_mode == 'bicubic':
            return self.limb_collect.generate_limbs_fused(hmps, offs, vector_nd, scl, self.inter_mode, jit)
        hmps_hr = upsample4(hmps, self.inter_mode)
        return self.limb_collect.generate_limbs_lowres(hmps_hr, offs, vector_nd, scl, self.inter_mode, jit)


def decoder_cli(parser):
    """Command-line flags of the decoder (same names and defaults as decoder/factory.py:149-188)."""
    g = parser.add_argument_group('limb collections in post-processing')
    g.add_argument('--resize-mode', default='bicubic', choices=['bilinear', 'bicubic'], type=str,
                   help='interpolation used to bring the keypoint heatmaps to input resolution')
    g.add_argument('--topk', default=48, type=int,
                   help='responses kept per heatmap channel (= candidate limbs per limb type)')
    g.add_argument('--thre-hmp', default=0.06, type=float,
                   help='candidate keypoints below this response are pushed outside the image')
    g.add_argument('--min-len', default=0.5, type=float,
                   help='lower clamp (pixels) on candidate limb length')
    g.add_argument('--feat-stage', default=-1, type=int, help='hourglass stack whose outputs are decoded')
    g = parser.add_argument_group('greedy grouping in post-processing')
    g.add_argument('--person-thre', default=0.06, type=float, help='minimum mean keypoint score of a pose')
    g.add_argument('--sort-dim', default=2, choices=[2, 4], type=int,
                   help='pose field used for scoring/sorting: 2 keypoint score, 4 limb score')
    g.add_argument('--dist-max', default=20, type=float,
                   help='reject limbs whose guided endpoint misses the matched keypoint by more than this')
    g.add_argument('--use-scale', default=True, type=boolean_string,
                   help='with a keypoint-scale head: scale-dependent limb rejection')
    g.add_argument('--use-jitter-offset', default=True, type=boolean_string,
                   help='with a jitter-offset head: refine keypoint positions')


_SKELETONS = {'omp': COCO_PERSON_SKELETON, 'omps': COCO_PERSON_SKELETON, 'omp19': COCO_PERSON_SKELETON,
              'offset': COCO_PERSON_SKELETON, 'offsets': COCO_PERSON_SKELETON,
              'omp16': KINEMATIC_TREE_SKELETON, 'omp31': COCO_PERSON_WITH_REDUNDANT_SKELETON,
              'omp44': DENSER_COCO_PERSON_SKELETON, 'omp25': REDUNDANT_CONNECTIONS, 'omps25': REDUNDANT_CONNECTIONS}


def parse_heads(head_name, stride):
    """Head name -> decoder configuration (decoder/factory.py:191-231).

    Two reference slips are fixed on purpose: 'omps' / 'offset' are accepted individually (a
    missing comma fuses them there, :212-213) and 'hmp17' works (UnboundLocalError there, :201-209)."""
    m = re.match('hmp[s]?([0-9]+)$', head_name)
    if head_name in ('hmp', 'hmps', 'heatmap', 'heatmaps') or m is not None:
        if m is not None:
            assert int(m.group(1)) == 17, f'{m.group(1)} keypoints not supported'
        return {'keypoints': COCO_KEYPOINTS, 'hmp_stride': stride}
    if head_name in ('omp', 'omps', 'offset', 'offsets') or re.match('omp[s]?([0-9]+)$', head_name) is not None:
        if head_name not in _SKELETONS:
            raise Exception('unknown skeleton type of head')
        return {'skeleton': _SKELETONS[head_name], 'omp_stride': stride}
    raise Exception('unknown head to create an encoder: {}'.format(head_name))


def decoder_factory(args):
    """Build a PostProcess from parsed flags (decoder/factory.py:234-267)."""
    cfg = {}
    for name, stride in zip(args.headnets, args.strides):
        cfg.update(parse_heads(name, stride))
    collector = LimbsCollect(cfg['hmp_stride'], cfg['omp_stride'], topk=args.topk, thre_hmp=args.thre_hmp,
                             min_len=args.min_len, include_jitter_offset=args.include_jitter_offset,
                             include_scale=args.include_scale, use_jitter_offset=args.use_jitter_offset,
                             keypoints=cfg['keypoints'], skeleton=cfg['skeleton'])
    grouper = GreedyGroup(args.person_thre, sort_dim=args.sort_dim, dist_max=args.dist_max,
                          use_scale=args.use_scale, keypoints=cfg['keypoints'], skeleton=cfg['skeleton'])
    return PostProcess(args.batch_size, cfg['hmp_stride'], cfg['omp_stride'], args.resize_mode,
                       keypoints=cfg['keypoints'], skeleton=cfg['skeleton'], limb_collector=collector,
                       limb_grouper=grouper, include_scale=args.include_scale,
                       include_jitter_offset=args.include_jitter_offset, feat_stage=args.feat_stage)
