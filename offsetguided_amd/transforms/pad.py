"""CenterPad (transforms/pad.py:9-66) fused with ToTensor and Normalize (evaluate.py:163-168)."""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..config import data_mean, data_std

FILL = (124, 116, 104)   # transforms/pad.py:59


def center_pad_ltrb(w, h, target_w, target_h):
    """(left, top, right, bottom) of CenterPad.center_pad (transforms/pad.py:40-55)."""
    left, top = max(int((target_w - w) / 2.0), 0), max(int((target_h - h) / 2.0), 0)
    return left, top, max(target_w - w - left, 0), max(target_h - h - top, 0)


class CenterPadNormalize:
    """images: list of (h, w, 3) uint8 RGB arrays / tensors (h, w <= target) -> (N, 3, T, T) fp32 on the device, plus
    the metas' `offset` / `valid_area` updates of CenterPad.__call__ (transforms/pad.py:24-32)."""

    def __init__(self, target_size, mean=data_mean, std=data_std, fill=FILL, device='cuda:0'):
        self.target_size = (target_size, target_size) if isinstance(target_size, int) else tuple(target_size)  # (w, h)
        self.device = torch.device(device)
        f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])  # noqa: E731
        self._mean, self._std, self._fill = f3(mean), f3(std), f3(fill)

    def __call__(self, images, metas=None):
        lib = _lib.load()
        tw, th = self.target_size
        out = torch.empty((len(images), 3, th, tw), dtype=torch.float32, device=self.device)
        stream = _lib.stream_ptr(self.device)
        keep = []
        for i, img in enumerate(images):
            t = torch.as_tensor(np.ascontiguousarray(img) if isinstance(img, np.ndarray) else img)
            assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3, 'images are (h, w, 3) uint8 RGB'
            t = _lib.require_device(t.to(self.device, non_blocking=True), 'image', torch.uint8)
            keep.append(t)
            h, w = int(t.shape[0]), int(t.shape[1])
            ltrb = (C.c_int * 4)()
            _lib.check(lib.og_center_pad_normalize_u8(_lib.ptr(t), h, w, th, tw, self._mean, self._std, self._fill,
                                                      _lib.ptr(out[i]), ltrb, stream), lib)
            if metas is not None:   # transforms/pad.py:28-31
                metas[i]['offset'] = np.asarray(metas[i]['offset'], np.float64) - np.array(ltrb[:2], np.float64)
                metas[i]['width_height'] = np.array([tw, th])
                metas[i]['valid_area'] = np.asarray(metas[i]['valid_area'], np.float64).copy()
                metas[i]['valid_area'][:2] += np.array(ltrb[:2], np.float64)
        return out
