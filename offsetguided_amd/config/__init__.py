"""Dataset constants of the decoder path (mirrors the reference `config` package surface)."""
from .coco_data import (coco_mean, coco_std, data_mean, data_std,  # noqa: F401
                        heatmap_hflip, offset_hflip)
