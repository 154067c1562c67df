"""Small helpers used by the CLI surface (reference utils/util.py:4-24)."""


def boolean_string(s):
    if s not in {'False', 'True'}:
        raise ValueError('Not a valid boolean string')
    return s == 'True'


class AverageMeter(object):
    """Running average of a scalar (batch time, img/s)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def adjust_learning_rate(learning_rate, world_size, optimizer, epoch, step, len_epoch, use_warmup=False):
    """Step-wise learning-rate schedule of the reference (utils/util.py:27-61): lr scales with the
    world size; linear warm-up over the first 15 epochs; x0.2 every 5 epochs from epoch 60; the fixed
    late-phase plateaus (78-92: x1, 92-105: x0.1, 105-110: x0.01) override it."""
    base = learning_rate * world_size
    lr = base if epoch < 60 else base * 0.2 ** ((epoch - 60) // 5)
    if use_warmup and epoch < 15:
        lr = lr * float(1 + step + epoch * len_epoch) / (15. * len_epoch)
    if 78 <= epoch < 92:
        lr = base
    if 92 <= epoch < 105:
        lr = 0.1 * base
    if 105 <= epoch < 110:
        lr = 0.01 * base
    for group in optimizer.param_groups:
        group['lr'] = lr
    return lr
