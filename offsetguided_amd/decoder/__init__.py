"""Drop-in for the reference `decoder` package (decoder/__init__.py:1-5): same public names,
every stage a hand-written gfx950 HIP kernel behind libog_decoder.so."""
from .heatmap import hmp_NMS, topK_channel, joint_dets, joint_dets_lowres  # noqa: F401
from .offset import scored_offset  # noqa: F401
from .group import GreedyGroup, soft_nms  # noqa: F401
from .collect import LimbsCollect  # noqa: F401
from .factory import decoder_factory, decoder_cli, PostProcess  # noqa: F401
