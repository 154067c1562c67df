"""Model side of the hot path (reference `models` package surface used by evaluate.py)."""
from .hourglass_104 import Hourglass104  # noqa: F401
from .heads import HeatMapsHead, OffsetMapsHead  # noqa: F401
from .networks import NetworkWrapper, load_model, save_model, initialize_weights  # noqa: F401
from .factory import net_cli, model_factory  # noqa: F401
from .losses import HeatMapsLoss, OffsetMapsLoss, LossChoice, lossfuncs_factory  # noqa: F401
from .engine import InferenceEngine  # noqa: F401
