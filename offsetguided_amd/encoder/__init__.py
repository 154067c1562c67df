"""Ground-truth encoders on the device (reference package `encoder/`): same class names, class-level settings and
output tuples as encoder/heatmap.py and encoder/offset.py, computed by HIP kernels for whole batches."""
from .factory import HeatMaps, OffsetMaps, encoder_cli, encoder_factory, factory_head, factory_heads  # noqa: F401
