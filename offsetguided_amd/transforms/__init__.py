"""Device-side input transforms (reference package `transforms/`, the evaluate.py chain evaluate.py:157-168).

Built: CenterPad + ToTensor + Normalize as one HIP pass over an already rescaled uint8 image, and the bookkeeping that
`annotations_inverse` needs.  Not built: RescaleLongAbsolute -- it is cv2.resize (transforms/scale.py), which cannot be
pinned in an image without cv2; callers rescale on the host (or feed images that already have the target long edge)."""
from .pad import CenterPadNormalize, center_pad_ltrb  # noqa: F401
