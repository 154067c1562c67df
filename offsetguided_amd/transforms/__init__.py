"""Device-side input transforms (reference package `transforms/`, the evaluate.py chain evaluate.py:157-168).

CenterPadNormalize: CenterPad + ToTensor + Normalize as one HIP pass over an already rescaled uint8 image.
EvalPreprocess: the whole chain incl. RescaleLongAbsolute (cv2.resize INTER_CUBIC restated from OpenCV's published 8-bit
algorithm; pinned to the CPU restatement in oracle/, parity with cv2 itself unpinned: cv2 is not available offline), one
kernel per image writing straight into the fp32 batch tensor, host images staged through pinned double buffers; plus the
meta bookkeeping `annotations_inverse` needs."""
from .pad import CenterPadNormalize, center_pad_ltrb  # noqa: F401
from .scale import EvalPreprocess, initial_meta, rescale_meta, rescale_size, resize_cubic  # noqa: F401
