from .util import AverageMeter, boolean_string  # noqa: F401
