"""RescaleLongAbsolute / RescaleHighAbsolute (transforms/scale.py:75-125) and the whole evaluate.py:157-168 input chain on
the device.

`rescale_size` and `rescale_meta` restate the reference's bookkeeping (target size from the long edge; x/y scale factors
(w' - 1) / (w - 1), keypoint scales * sqrt(sx * sy); meta offset / scale / valid_area updates, transforms/scale.py:33-71).
`EvalPreprocess` is the chain NormalizeAnnotations' meta -> RescaleLongAbsolute -> CenterPad -> ToTensor -> Normalize: the
uint8 images of a batch go to the device through a pinned staging buffer with one copy and ONE launch
(og_rescale_pad_normalize_batch_u8) writes the fp32 batch tensor.  The interpolation is OpenCV's published INTER_CUBIC algorithm for 8-bit images; it is
pinned to the CPU restatement in oracle/, parity with cv2 itself is unpinned (cv2 is not available offline)."""
import ctypes as C
import math

import numpy as np
import torch

from .. import _lib
from ..config import data_mean, data_std
from .pad import FILL


def rescale_size(w, h, long_edge, fixed_height=False):
    """(target_w, target_h) of RescaleLongAbsolute (transforms/scale.py:91-97) / RescaleHighAbsolute (:118-121)."""
    if fixed_height:
        s = long_edge / h
        return int(w * s), long_edge
    s = long_edge / max(h, w)
    return (int(w * s), long_edge) if h > w else (long_edge, int(h * s))


def initial_meta(w, h, image_id=None):
    """The meta NormalizeAnnotations creates for an untouched image (transforms/annotations.py:68-80)."""
    return {'image_id': image_id, 'offset': np.array([0.0, 0.0]), 'scale': np.array([1.0, 1.0]),
            'valid_area': np.array([0.0, 0.0, w, h]), 'hflip': False, 'width_height': np.array([w, h])}


def rescale_meta(meta, anns, w, h, target_w, target_h):
    """Meta / annotation updates of _scale (transforms/scale.py:33-71); returns (meta, anns), inputs untouched."""
    x_scale, y_scale = (target_w - 1) / (w - 1), (target_h - 1) / (h - 1)
    meta = {k: (np.array(v, dtype=np.float64) if isinstance(v, np.ndarray) else v) for k, v in meta.items()}
    factors = np.array((x_scale, y_scale))
    meta['offset'] = meta['offset'] * factors
    meta['scale'] = meta['scale'] * factors
    meta['width_height'] = np.array([target_w, target_h])
    meta['valid_area'] = meta['valid_area'].copy()
    meta['valid_area'][:2] *= factors
    meta['valid_area'][2:] *= factors
    if anns is not None and len(anns):
        anns = np.array(anns, dtype=np.float32, copy=True)
        anns[:, :, 0] *= x_scale
        anns[:, :, 1] *= y_scale
        anns[:, :, 3] *= math.sqrt(x_scale * y_scale)
    return meta, anns


def resize_cubic(image, new_h, new_w, device='cuda:0'):
    """cv2.resize(image, (new_w, new_h), INTER_CUBIC) of one (h, w, 3) uint8 image on the device -> uint8 tensor."""
    lib = _lib.load()
    dev = torch.device(device)
    t = torch.as_tensor(np.ascontiguousarray(image) if isinstance(image, np.ndarray) else image)
    t = _lib.require_device(t.to(dev), 'image', torch.uint8)
    assert t.dim() == 3 and t.shape[2] == 3, 'images are (h, w, 3) uint8 RGB'
    out = torch.empty((new_h, new_w, 3), dtype=torch.uint8, device=dev)
    _lib.check(lib.og_resize_cubic_u8(_lib.ptr(t), int(t.shape[0]), int(t.shape[1]), _lib.ptr(out), new_h, new_w,
                                      _lib.stream_ptr(dev)), lib)
    return out


class EvalPreprocess:
    """images: list of (h, w, 3) uint8 RGB arrays of any size -> ((N, 3, T, T) fp32 on the device, metas) with the metas
    `annotations_inverse` needs (offset, scale) after rescale + centre pad.  Host images are packed into one of three pinned
    buffers and copied asynchronously; batch i+1 can be prepared while batch i is in the network, and its packing -- pure host
    work, a memcpy per image -- may run on another thread (`pack`, used by evaluate.run_images: the reference prepares its
    batches in DataLoader workers, evaluate.py:170-178)."""

    def __init__(self, long_edge, device='cuda:0', mean=data_mean, std=data_std, fill=FILL, fixed_height=False, max_stride=128):
        """fixed_height: RescaleHighAbsolute(long_edge) + RightDownPad(max_stride) (evaluate.py:150-156) instead of
        RescaleLongAbsolute + CenterPad: every image of a batch must then give the same padded size."""
        self.long_edge, self.device, self.fixed_height, self.max_stride = long_edge, torch.device(device), fixed_height, max_stride
        f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])  # noqa: E731
        self._mean, self._std, self._fill = f3(mean), f3(std), f3(fill)
        self._stage, self._turn = [None, None, None], 0
        self._pack_lock = __import__('threading').Lock()   # ONE pack at a time (see pack)
        self.copy_stream = _lib.dedicated_stream(self.device, ('input-chain',))    # (not torch's repeating pool of 32)

    def _staging(self, nbytes):
        buf = self._stage[self._turn]
        if buf is None or buf[0].numel() < nbytes:
            buf = self._stage[self._turn] = [torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8).pin_memory(), None]
        if buf[1] is not None:
            buf[1].synchronize()          # the copies that last used this staging buffer have left the host
        self._turn = (self._turn + 1) % len(self._stage)
        return buf

    def pack(self, images):
        """Host part of a batch: the images copied into one pinned staging buffer.  No device call except the wait for the buffer's
        previous H2D copy; may run on a worker thread while the caller's thread feeds the device.  Invariant: ONE pack at a time
        (the staging ring and its turn counter are not synchronised) -- enforced: a second concurrent pack raises instead of racing."""
        sizes = [(int(im.shape[0]), int(im.shape[1])) for im in images]
        total = sum(h * w * 3 for h, w in sizes)
        if not self._pack_lock.acquire(blocking=False):
            raise RuntimeError('EvalPreprocess.pack: another pack is in progress (one packer at a time: a worker thread OR the caller)')
        try:
            return self._pack_locked(images, sizes, total)
        finally:
            self._pack_lock.release()

    def _pack_locked(self, images, sizes, total):
        stage = self._staging(total)
        o = 0
        stage_np = stage[0].numpy()                    # (a plain memcpy: torch's copy_ into the pinned buffer ran at 0.7 GB/s)
        for im, (h, w) in zip(images, sizes):          # pack the batch into ONE pinned buffer: one H2D copy
            assert im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3, 'images are (h, w, 3) uint8 RGB'
            np.copyto(stage_np[o:o + h * w * 3].reshape(h, w, 3), im)
            o += h * w * 3
        return stage, sizes, total

    def __call__(self, images, image_ids=None, packed=None):
        """packed: the result of pack(images) (same images), or None to pack here."""
        lib = _lib.load()
        T = self.long_edge
        stage, sizes, total = packed if packed is not None else self.pack(images)
        assert len(sizes) == len(images)
        targets = [rescale_size(w, h, T, self.fixed_height) for h, w in sizes]
        if self.fixed_height:   # RightDownPad: up to the next multiple of max_stride (transforms/pad.py:100-103)
            up = lambda v: (v + self.max_stride - 1) // self.max_stride * self.max_stride  # noqa: E731
            padded = {(up(th), up(tw)) for tw, th in targets}
            assert len(padded) == 1, f'--fixed-height: the images of one batch pad to different sizes {sorted(padded)}'
            PH, PW = padded.pop()
        else:
            PH = PW = T
        metas, o = [], 0
        # The whole input chain runs on the COPY stream, behind the H2D copy it depends on: the caller's stream is busy with the
        # previous batch (the host runs a batch ahead), and eight small launches queued there sat between that batch's decoder and this
        # batch's backbone (a kernel trace of round 5: 185 us between the heads kernel and the next stem).  The caller's stream only waits
        # for the finished tensor.  (Neutral on bench.py's harness figure, 1 267 vs 1 271 img/s: the host waits 4.7 of every 6.2 ms for
        # the previous batch, tools/harness_host_times.py -- the pitch is the device's.)
        with torch.cuda.stream(self.copy_stream):
            # allocated on the copy stream's pool: a block of the compute stream's pool may still be in use by kernels
            # queued there (the host runs a batch ahead), and the copy stream does not wait for them
            dev_raw = torch.empty(total, dtype=torch.uint8, device=self.device)
            dev_raw.copy_(stage[0][:total], non_blocking=True)
            stage[1] = torch.cuda.Event()
            stage[1].record(self.copy_stream)
            out = torch.empty((len(images), 3, PH, PW), dtype=torch.float32, device=self.device)
            # ONE launch for the batch (og_rescale_pad_normalize_batch_u8: blockIdx.y = image, the geometry table rides in the
            # kernel arguments); a launch per image cost the host 0.3 ms per batch and the device eight ramps and tails
            n = len(sizes)
            offs, hw4, ltrb = (C.c_long * n)(), (C.c_int * (4 * n))(), (C.c_int * (4 * n))()
            for i, (h, w) in enumerate(sizes):
                offs[i] = o
                hw4[4 * i:4 * i + 4] = [h, w, targets[i][1], targets[i][0]]
                o += h * w * 3
            _lib.check(lib.og_rescale_pad_normalize_batch_u8(_lib.ptr(dev_raw), offs, hw4, n, PH, PW, int(self.fixed_height),
                                                             self._mean, self._std, self._fill, _lib.ptr(out), ltrb,
                                                             _lib.stream_ptr(self.device)), lib)
            for i, (h, w) in enumerate(sizes):
                tw, th = targets[i]
                meta, _ = rescale_meta(initial_meta(w, h, None if image_ids is None else image_ids[i]), None, w, h, tw, th)
                meta['offset'] = meta['offset'] - np.array(ltrb[4 * i:4 * i + 2], np.float64)          # CenterPad, transforms/pad.py:28-31
                meta['valid_area'][:2] += np.array(ltrb[4 * i:4 * i + 2], np.float64)
                meta['width_height'] = np.array([PW, PH])
                metas.append(meta)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_stream(self.copy_stream)
        out.record_stream(cur)
        return out, metas
