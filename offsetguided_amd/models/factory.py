"""net_cli / model_factory (reference models/factory.py:10-125), inference-relevant part.

The training losses (models/losses.py) are outside this path: model_factory returns an empty
loss list in their place, keeping the `(model, lossfuncs)` return shape evaluate.py unpacks."""
import logging
import os

from ..utils import boolean_string
from . import heads, networks

LOG = logging.getLogger(__name__)


def net_cli(parser):
    g = parser.add_argument_group('model configuration')
    g.add_argument('--initialize-whole', default=True, type=boolean_string,
                   help='randomly initialize the basenet and headnets')
    g.add_argument('--checkpoint-whole', default=None, type=str,
                   help='checkpoint of the whole model (basenet+headnets)')
    g = parser.add_argument_group('base network configuration')
    g.add_argument('--basenet', default='hourglass104', help='base network')
    g.add_argument('--two-scale', default=False, action='store_true', help='to be implemented')
    g.add_argument('--multi-scale', default=False, action='store_true', help='to be implemented')
    g.add_argument('--no-pretrain', dest='pretrained', default=True, action='store_false',
                   help='create BaseNet without pretraining')
    g.add_argument('--basenet-checkpoint', default='weights/hourglass_104_renamed.pth', type=str,
                   help='pre-trained backbone weights')
    g = parser.add_argument_group('head network configuration')
    g.add_argument('--headnets', default=['hmp', 'omp'], nargs='+', help='head networks')
    g.add_argument('--strides', default=[4, 4], nargs='+', type=int, help='output stride of every head')
    g.add_argument('--max-stride', default=128, type=int, choices=[64, 128],
                   help='largest down-sampling factor inside the network')
    g.add_argument('--include-spread', default=False, action='store_true')
    g.add_argument('--include-background', default=False, action='store_true')
    g.add_argument('--include-jitter-offset', default=False, action='store_true')
    g.add_argument('--include-scale', default=False, action='store_true')


def model_factory(args):
    """Build basenet + heads from the flags -> (NetworkWrapper, lossfuncs)."""
    if 'hourglass' not in args.basenet:
        raise Exception(f'unknown base network: {args.basenet}')
    basenet, n_stacks, stride, max_stride, feature_dim = networks.basenet_factory(args.basenet)
    if args.initialize_whole:
        networks.initialize_weights(basenet)
    if args.pretrained:
        # the reference stops at an interactive prompt when the file is missing
        # (networks.py:31-41); a drop-in must not block: warn and keep the initialisation
        if os.path.isfile(args.basenet_checkpoint):
            networks.load_model(basenet, args.basenet_checkpoint)
        else:
            LOG.warning('backbone checkpoint %s not found: continuing with initialised weights',
                        args.basenet_checkpoint)
    assert stride == args.strides[0], 'strides mismatch'
    assert max_stride == args.max_stride, 'please reset the max_stride based on the network manually'
    headnets = heads.headnets_factory(args.headnets, n_stacks, args.strides, feature_dim, args.include_spread,
                                      args.include_background, args.include_jitter_offset, args.include_scale)
    if args.initialize_whole:
        headnets = [networks.initialize_weights(h) for h in headnets]
    return networks.NetworkWrapper(basenet, headnets), []
