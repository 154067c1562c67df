"""Input-side transforms (the part without cv2): CenterPad + ToTensor + Normalize in one HIP pass against
the same arithmetic in torch on the CPU (torchvision's ToTensor / Normalize are `div(255)`, `sub_(mean).div_(std)`;
torchvision.transforms.functional.pad with a colour fill is a constant border)."""
import numpy as np
import pytest
import torch

from offsetguided_amd import transforms
from offsetguided_amd.config import data_mean, data_std


def test_center_pad_ltrb_matches_reference_formula():
    # transforms/pad.py:40-55 on a few sizes (odd differences put the extra pixel right / down)
    assert transforms.center_pad_ltrb(640, 427, 640, 640) == (0, 106, 0, 107)
    assert transforms.center_pad_ltrb(480, 640, 640, 640) == (80, 0, 80, 0)
    assert transforms.center_pad_ltrb(333, 501, 640, 640) == (153, 69, 154, 70)
    assert transforms.center_pad_ltrb(640, 640, 640, 640) == (0, 0, 0, 0)


@pytest.mark.gpu
def test_center_pad_normalize_bit_exact():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no HIP device is visible")
    rng = np.random.default_rng(0)
    images = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in ((427, 640), (640, 480), (501, 333), (640, 640), (1, 1))]
    metas = [{'offset': np.array([0.0, 0.0]), 'valid_area': np.array([0.0, 0.0, im.shape[1], im.shape[0]])} for im in images]
    op = transforms.CenterPadNormalize(640)
    out = op(images, metas).cpu()
    mean, std = torch.tensor(data_mean).view(3, 1, 1), torch.tensor(data_std).view(3, 1, 1)
    for im, got, meta in zip(images, out, metas):
        h, w = im.shape[:2]
        left, top, right, bottom = transforms.center_pad_ltrb(w, h, 640, 640)
        canvas = np.empty((640, 640, 3), np.uint8)
        canvas[:] = np.array(transforms.pad.FILL, np.uint8)
        canvas[top:top + h, left:left + w] = im
        ref = (torch.from_numpy(canvas).permute(2, 0, 1).float().div(255) - mean) / std
        assert torch.equal(got, ref)
        assert tuple(meta['offset']) == (-left, -top) and tuple(meta['valid_area'][:2]) == (left, top)
        assert tuple(meta['width_height']) == (640, 640)
