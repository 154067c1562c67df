"""net_cli / model_factory (reference models/factory.py:10-125), inference-relevant part.

model_factory returns `(model, lossfuncs)` like the reference; the loss objects (models/losses.py)
are only used by the training step (train_dist.py)."""
import logging
import os

from ..utils import boolean_string
from . import heads, losses, networks

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
    loss_cli(parser)


def loss_cli(parser):
    g = parser.add_argument_group('loss configuration')
    g.add_argument('--lambdas', default=[1, 1, 100, 100, 0.01], type=float, nargs='+',
                   help='task weights for hmp, bg_hmp, jitter_off, offset and scale losses (multiplied, not averaged)')
    g.add_argument('--stack-weights', default=[1, 1], type=float, nargs='+', help='loss weights of the hourglass stacks')
    g.add_argument('--hmp-loss', default='focal_l2_loss', choices=['l2_loss', 'focal_l2_loss'])
    g.add_argument('--jitter-offset-loss', default='offset_l1_loss',
                   choices=['offset_l1_loss', 'vector_l1_loss', 'offset_laplace_loss'])
    g.add_argument('--offset-loss', default='offset_l1_loss',
                   choices=['offset_l1_loss', 'vector_l1_loss', 'offset_laplace_loss', 'offset_instance_l1_loss'])
    g.add_argument('--sqrt-re', default=False, action='store_true', help='rescale the offset loss with sqrt')
    g.add_argument('--scale-loss', default='scale_l1_loss', choices=['scale_l1_loss'])
    g.add_argument('--ftao', default=losses.TAU, type=float, help='fore/background threshold of the focal L2 loss')
    g.add_argument('--fgamma', default=losses.GAMMA, type=float, help='exponent of the focal L2 scaling factor')
    g.add_argument('--lmargin', default=losses.MARGIN, type=float, help='offset errors below this are not punished')
    g.add_argument('--fused-losses', default=True, type=boolean_string,
                   help='on GPU: one fused HIP kernel per loss (value + gradient) instead of torch mask/gather ops')


def model_factory(args):
    """Build basenet + heads from the flags -> (NetworkWrapper, lossfuncs)."""
    losses.TAU, losses.GAMMA, losses.MARGIN = args.ftao, args.fgamma, args.lmargin
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
    lossfuncs = losses.lossfuncs_factory(args.headnets, n_stacks, args.stack_weights, args.hmp_loss,
                                         args.jitter_offset_loss, args.offset_loss, args.scale_loss, args.sqrt_re,
                                         fused=args.fused_losses)
    return networks.NetworkWrapper(basenet, headnets), lossfuncs
