"""Inference harness with the reference evaluate.py surface (evaluate_cli :49-122, run_images
:125-300, validation :303-328), MI355X-native underneath:

  images -> models.InferenceEngine (fp16 NHWC: the reference's apex-O2 arithmetic, HIP graph) -> decoder.PostProcess.submit (HIP kernels,
  batch i+1's backbone is queued before batch i's poses are collected) -> annotations_inverse ->
  COCO-style result dicts.

COCO data loading and pycocotools are not part of this path: `run_images` takes any iterable of
(images, annos, metas) batches (the reference's collate format, data/factory.py:23-35) and falls back
to synthetic batches; `validation` needs pycocotools and raises if it is absent.
"""
import argparse
import json
import logging
import os
import time

import numpy as np
import torch

from . import _lib, decoder, models
from .utils import AverageMeter

LOG = logging.getLogger(__name__)

ANNOTATIONS_VAL = 'data/link2COCO2017/annotations/person_keypoints_val2017.json'
IMAGE_DIR_VAL = 'data/link2COCO2017/val2017'
ANNOTATIONS_TESTDEV = 'data/link2COCO2017/annotations_trainval_info/image_info_test-dev2017.json'
ANNOTATIONS_TEST = 'data/link2COCO2017/annotations_trainval_info/image_info_test2017.json'
IMAGE_DIR_TEST = 'data/link2COCO2017/test2017/'


def evaluate_cli(argv=None):
    """Same flags as the reference (apex flags are accepted and ignored: bf16 is built in)."""
    parser = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    g = parser.add_argument_group('logging')
    g.add_argument('--debug', default=False, action='store_true')
    g.add_argument('-q', '--quiet', default=False, action='store_true')
    models.net_cli(parser)
    decoder.decoder_cli(parser)
    parser.add_argument('--dump-name', default='hourglass104_mi355x', type=str, help='detection file name')
    parser.add_argument('--dataset', choices=('val', 'test', 'test-dev'), default='val')
    parser.add_argument('--batch-size', default=8, type=int)
    parser.add_argument('--long-edge', default=640, type=int, help='long edge of input images')
    parser.add_argument('--fixed-height', action='store_true', default=False)
    parser.add_argument('--flip-test', action='store_true', default=False, help='flip augmentation during testing')
    parser.add_argument('--cat-flip-offset', action='store_true', default=False)
    parser.add_argument('--loader-workers', default=8, type=int)
    parser.add_argument('--all-images', default=False, action='store_true')
    parser.add_argument('--resume', '-r', action='store_true', default=False, help='load --checkpoint-whole')
    parser.add_argument('--checkpoint-path', '-p', default='link2checkpoints_storage')
    parser.add_argument('--show-detected-poses', action='store_true', default=False)
    g = parser.add_argument_group('apex configuration (accepted for command-line compatibility, unused)')
    g.add_argument('--local_rank', default=0, type=int)
    g.add_argument('--opt-level', type=str, default='O2')
    g.add_argument('--keep-batchnorm-fp32', type=str, default=None)
    g.add_argument('--loss-scale', type=str, default=None)
    g.add_argument('--channels-last', default=False, action='store_true')
    g.add_argument('--print-freq', '-f', default=10, type=int, metavar='N')
    args = parser.parse_args(argv)
    args.image_dir, args.annotation_file = {
        'val': (IMAGE_DIR_VAL, ANNOTATIONS_VAL), 'test': (IMAGE_DIR_TEST, ANNOTATIONS_TEST),
        'test-dev': (IMAGE_DIR_TEST, ANNOTATIONS_TESTDEV)}[args.dataset]
    if args.dataset in ('test', 'test-dev'):
        args.all_images = True
    return args


def annotations_inverse(keypoints, meta):
    """Poses from network-input coordinates back to the original image (transforms/preprocess.py:33-63):
    un-pad (offset), un-scale, keypoint scales / sqrt(sx*sy)."""
    kp = np.array(keypoints, copy=True)
    kp[:, :, 0] += meta['offset'][0]
    kp[:, :, 1] += meta['offset'][1]
    kp[:, :, 0] /= meta['scale'][0]
    kp[:, :, 1] /= meta['scale'][1]
    kp[:, :, 3] /= np.sqrt(np.prod(meta['scale']))
    if meta.get('hflip'):
        raise Exception('this should not happen. please have a check here, not implemented actually!')
    return kp


def poses_to_results(image_poses, image_meta, result_keypoints, result_image_ids):
    """Append one image's COCO keypoint results (evaluate.py:227-265); returns the inverse-mapped poses."""
    subset = annotations_inverse(image_poses, image_meta)
    image_id = image_meta['image_id']
    result_image_ids.append(image_id)
    subset[:, :, :2] = np.around(subset[:, :, :2], 2)
    # (vectorised: the reference's loop over persons x keypoints costs 0.3 ms per image at 40 poses -- 2.4 ms of host time per batch of 8,
    # which made the loop host-bound on crowded images; same values, same types: float x / y, int flag, sequential float sum of the scores)
    if len(subset):
        sub = subset.astype(float)
        n_kp = sub.shape[1]
        flags = ((sub[:, :, 0] > 0) | (sub[:, :, 1] > 0)).astype(int).tolist()
        trip = np.concatenate((sub[:, :, :2], np.zeros(sub.shape[:2] + (1,))), axis=2).reshape(len(sub), 3 * n_kp).tolist()
        # the reference's sum(v) / len(v) (evaluate.py:252) as explicit left-to-right float64 adds, column by column -- what Python's sum()
        # does up to 3.11 (from 3.12 on sum() of floats is Neumaier-compensated: the last ulp of a score may then differ from a reference
        # run under that interpreter; this box and the reference's pins run 3.10 / 3.7)
        total = sub[:, 0, 2].copy()
        for j in range(1, n_kp):
            total += sub[:, j, 2]
        scores = (total / n_kp).tolist()
        for triples, flag, score in zip(trip, flags, scores):
            triples[2::3] = flag
            result_keypoints.append({'image_id': image_id, 'category_id': 1, 'keypoints': triples, 'score': score})
    if not len(subset):
        result_keypoints.append({'image_id': image_id, 'category_id': 1, 'keypoints': np.zeros((17 * 3,)).tolist(),
                                 'score': 0.01})
    return subset


def synthetic_loader(n_batches, batch_size, size, device, seed=0):
    """Stand-in for DataLoader(CocoKeypoints): random normalised images + identity metas."""
    g = torch.Generator(device).manual_seed(seed)
    for b in range(n_batches):
        images = torch.randn(batch_size, 3, size, size, device=device, generator=g)
        metas = [{'image_id': b * batch_size + i, 'offset': np.array([0.0, 0.0]), 'scale': np.array([1.0, 1.0]),
                  'hflip': False} for i in range(batch_size)]
        yield images, [None] * batch_size, metas


ENGINE_CACHE = 8     # input shapes run_images keeps engines for (--fixed-height: one per padded width, a handful on COCO; an engine
                     # holds its activations + graph, ~0.1 GB per image of 640x640, the weights are shared)
# Batches in flight: batch i runs whole (backbone graph + decoder) on HIP stream i % IN_FLIGHT with that lane's PostProcess; the head
# and tail of one forward (stem, final layers, heads, decoder: few workgroups) then run beside the bulk of the next.  A shape whose
# batches follow each other gets a second engine (captured graph + activations; the weights are shared, models/engine.py:_shared_layers)
# so that two of them can run at once.  Measured on bench.py's loop: 2 = +1.6...2.1 %, 3 = nothing more.
IN_FLIGHT = 2


def run_images(args, data_loader=None, model=None, n_synthetic_batches=4, stats=None):
    """The hot loop of evaluate.py:207-298.  Returns (result_keypoints, result_image_ids).
    stats: an optional dict that receives `host_enqueue_s` (per batch: host time to queue the input chain, the forward and the decoder --
    no wait in it), `engines_built` and `torch_conv_calls` (0: every engine is strict, models/engine.py)."""
    if not torch.cuda.is_available():
        raise RuntimeError('run_images needs a HIP device (offsetguided_amd has no CPU path)')
    dev = torch.device('cuda', torch.cuda.current_device())
    result_keypoints, result_image_ids = [], []
    if model is None:
        model, _ = models.model_factory(args)
        if args.resume:
            model, *_ = models.load_model(model, args.checkpoint_whole, optimizer=None, resume_optimizer=False,
                                          drop_layers=False, load_amp=False)
    processors = [decoder.decoder_factory(args) for _ in range(IN_FLIGHT)]
    lanes = _lib.lane_streams(dev, IN_FLIGHT) if IN_FLIGHT > 1 else [torch.cuda.current_stream(dev)]
    if data_loader is None:
        data_loader = synthetic_loader(n_synthetic_batches, args.batch_size, args.long_edge, dev)
    feeder = DeviceFeeder(dev)
    # engines (scratch + captured graph) per input shape, those of the ENGINE_CACHE most recently used shapes kept (--fixed-height:
    # one shape per width); the folded / tiled weights are shared between them (models/engine.py:_shared_layers); a ragged last
    # batch is padded instead of getting an engine of its own.  A shape has one engine until two batches of it follow each other
    # closer than IN_FLIGHT apart: then the one still busy is left alone and another is built (at most IN_FLIGHT per shape)
    import collections
    engines = collections.OrderedDict()                      # shape -> [[engine, index of the last batch it ran, event behind that batch], ...]
    batch_time, end, last_print, pending = AverageMeter(), time.time(), -1, collections.deque()
    full_batch = None
    first_engine = [None]

    def collect(handle):
        poses, metas = handle
        for image_poses, image_meta in zip(poses.result(), metas):   # zip drops the padded images of a ragged batch
            poses_to_results(image_poses, image_meta, result_keypoints, result_image_ids)

    preprocess, packer = [None], [None]

    def ahead(loader):
        """The loader one batch ahead; a batch of raw images is packed into pinned memory on a worker thread meanwhile (host
        memcpys: the reference does this part in DataLoader workers, evaluate.py:170-178)."""
        from concurrent.futures import ThreadPoolExecutor

        def prepare(batch):
            images = batch[0]
            if not isinstance(images, (list, tuple)):
                return batch, None
            if preprocess[0] is None:
                from .transforms import EvalPreprocess
                preprocess[0] = EvalPreprocess(args.long_edge, device=dev, fixed_height=args.fixed_height)
                packer[0] = ThreadPoolExecutor(max_workers=1, thread_name_prefix='og-pack')
            imgs = list(images)
            return (imgs,) + tuple(batch[1:]), packer[0].submit(preprocess[0].pack, imgs)

        it = iter(loader)
        try:
            nxt = prepare(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur = nxt
            try:
                nxt = prepare(next(it))
            except StopIteration:
                nxt = None
            yield cur

    try:
        for batch_idx, ((images, _, metas), packed) in enumerate(ahead(data_loader)):
            t_host = time.perf_counter()
            if packed is not None:
                # raw (h, w, 3) uint8 RGB images of any size: the input chain of evaluate.py:157-168 runs on the device
                # (RescaleLongAbsolute + CenterPad, or with --fixed-height RescaleHighAbsolute + RightDownPad of :150-156, then
                # ToTensor + Normalize; pinned staging packed a batch ahead, one H2D copy); metas are derived here
                images, metas = preprocess[0](images, image_ids=[m['image_id'] for m in metas], packed=packed.result())
            images = feeder(images)
            full_batch = full_batch or images.shape[0]
            if images.shape[0] < full_batch:   # last batch of the dataset: fill up to the engine's batch, results are dropped
                images = torch.cat((images, images[-1:].expand(full_batch - images.shape[0], -1, -1, -1)))
            if args.flip_test:
                images = torch.cat((images, torch.flip(images, [-1])))
            lane = batch_idx % len(lanes)
            of_shape = engines.pop(tuple(images.shape), None)
            if of_shape is None:
                if len(engines) >= ENGINE_CACHE:
                    while pending:               # the batches in flight still read the outputs of the engines that go
                        collect(pending.popleft())
                    engines.popitem(last=False)
                of_shape = []
            engines[tuple(images.shape)] = of_shape          # most recently used last
            slot = min(of_shape, key=lambda e: e[1], default=None)           # the engine of this shape that has rested longest
            if slot is None or (batch_idx - slot[1] < len(lanes) and len(of_shape) < len(lanes)):
                slot = [models.InferenceEngine(model, images.shape[0], images.shape[2], images.shape[3], device=dev,
                                               feat_stage=args.feat_stage, like=first_engine[0]), -len(lanes), None]
                first_engine[0] = first_engine[0] or slot[0]      # the module's weights do not change inside one call
                of_shape.append(slot)
                if stats is not None:
                    stats['engines_built'] = stats.get('engines_built', 0) + 1
                    stats.setdefault('_engines', []).append(slot[0])
            cur = torch.cuda.current_stream(dev)
            if lanes[lane] is not cur:
                lanes[lane].wait_stream(cur)                 # the input chain (H2D copy, rescale / pad / normalize, flip) ran on `cur`
                images.record_stream(lanes[lane])
            if slot[2] is not None:
                lanes[lane].wait_event(slot[2])              # the engine's last batch (maybe on another lane): its decoder has read the outputs
            with torch.cuda.stream(lanes[lane]):
                outputs = slot[0](images)
                handle = (processors[lane].submit(outputs, flip_test=args.flip_test, cat_flip_offs=args.cat_flip_offset), metas)
                slot[1], slot[2] = batch_idx, torch.cuda.Event()
                slot[2].record(lanes[lane])
            pending.append(handle)
            if stats is not None:
                stats.setdefault('host_enqueue_s', []).append(time.perf_counter() - t_host)
            while len(pending) > len(lanes):                 # the oldest batch: its poses are on the host by now (or soon)
                collect(pending.popleft())
            if batch_idx % args.print_freq == 0:
                torch.cuda.synchronize()
                now = time.time()
                per_batch = (now - end) / (batch_idx - last_print)   # batches since the last print (1 at the first), not print_freq
                end, last_print = now, batch_idx
                if batch_idx > 0:                                    # the first batch builds the engine: not a speed sample
                    batch_time.update(per_batch)
                print('==================> [{0}]\tTime {1:.3f} ({2:.3f})\tSpeed {3:.3f} ({4:.3f})'.format(
                    batch_idx, per_batch, batch_time.avg or per_batch, args.batch_size / per_batch,
                    args.batch_size / (batch_time.avg or per_batch)))
        while pending:
            collect(pending.popleft())
    finally:
        # also on an exception (engine build failure, OgError, the --fixed-height assertion): the worker must not outlive the
        # call holding pinned staging buffers, and a pack still queued must not write one while the caller handles the error
        if packer[0] is not None:
            packer[0].shutdown(wait=True, cancel_futures=True)
        if stats is not None:
            stats['torch_conv_calls'] = sum(len(e.torch_conv_calls) for e in stats.pop('_engines', []))
    return result_keypoints, result_image_ids


class DeviceFeeder:
    """Host batch -> device through two pinned staging buffers (the data loader's tensors are pageable: a direct
    `.to(device, non_blocking=True)` from them is a synchronous staged copy).  The copy of batch i+1 is queued on its own
    stream while batch i computes; a batch that already lives on the device passes through."""

    def __init__(self, device):
        self.device = device
        self.stream = _lib.dedicated_stream(device, ('feeder',))
        self.slots, self.turn = [None, None], 0

    def __call__(self, images):
        if images.is_cuda:
            return images
        slot = self.slots[self.turn]
        if slot is None or slot[0].shape != images.shape or slot[0].dtype != images.dtype:
            slot = self.slots[self.turn] = [torch.empty(images.shape, dtype=images.dtype).pin_memory(), None]
        if slot[1] is not None:
            slot[1].synchronize()                # the copy that last used this staging buffer has left the host
        slot[0].copy_(images)
        with torch.cuda.stream(self.stream):
            out = slot[0].to(self.device, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record(self.stream)
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        out.record_stream(torch.cuda.current_stream(self.device))
        self.turn ^= 1
        return out


def validation(args, data_loader=None):
    """run_images + COCO keypoint evaluation (evaluate.py:303-328); needs pycocotools."""
    try:
        from pycocotools.coco import COCO
        from pycocotools.cocoeval import COCOeval
    except ImportError as e:
        raise ImportError('validation() needs pycocotools; run_images() does not') from e
    res_file = 'data/link2COCO2017/results/person_keypoints_%s_%s_results.json' % (args.dataset, args.dump_name)
    os.makedirs(os.path.dirname(res_file), exist_ok=True)
    coco_gt = COCO(args.annotation_file)
    results, ids = run_images(args, data_loader)
    json.dump(results, open(res_file, 'w'))
    coco_eval = COCOeval(coco_gt, coco_gt.loadRes(res_file), iouType='keypoints')
    coco_eval.params.imgIds = ids
    coco_eval.evaluate()
    coco_eval.accumulate()
    coco_eval.summarize()
    return coco_eval


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO)
    a = evaluate_cli()
    kps, ids = run_images(a)
    print(f'{len(ids)} images, {len(kps)} detections')
