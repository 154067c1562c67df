from .util import AverageMeter, adjust_learning_rate, boolean_string  # noqa: F401
