"""Data-parallel training step (reference train_dist.py:108-387), MI355X-native:

  apex AMP O1 + dynamic loss scaling      -> torch.autocast(bfloat16) (no scaler needed)
  apex DistributedDataParallel            -> torch DDP over RCCL/xGMI: 25 MB gradient buckets whose
    (delay_allreduce=True: one flat            all-reduce overlaps the rest of backward (xGMI links are
     all-reduce after backward)                point-to-point, so overlap matters more than on NVSwitch)
  apex SyncBatchNorm                      -> torch.nn.SyncBatchNorm (optional, --no-sync-bn)
  apex FusedAdam                          -> torch.optim.Adam(fused=True)
  mask/gather losses                      -> fused HIP loss kernels (models/losses.py, csrc/losses.hip)

One process per GPU (`python -m torch.distributed.run --nproc-per-node N -m offsetguided_amd.train_dist`);
without COCO on disk the loop runs on synthetic encoder-style targets (GT heatmaps, patch offsets with
inf outside the patches, instance scales, mask_miss).
"""
import argparse
import os
import time

import numpy as np
import torch

from . import encoder, models, sharding, synth
from .config import coco_data as cd
from .utils import AverageMeter, adjust_learning_rate


def train_cli(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    models.net_cli(p)
    p.add_argument('--resume', '-r', action='store_true', default=False)
    p.add_argument('--epochs', default=100, type=int)
    p.add_argument('--warmup', action='store_true', default=False, help='warm-up learning rate')
    p.add_argument('--checkpoint-path', '-p', default='link2checkpoints_storage')
    p.add_argument('--batch-size', default=8, type=int, help='per-GPU batch size')
    p.add_argument('--square-length', default=512, type=int, help='training crop size')
    p.add_argument('--steps-per-epoch', default=20, type=int, help='synthetic-data epoch length')
    p.add_argument('--no-sync-bn', dest='sync_bn', action='store_false', default=True)
    p.add_argument('--print-freq', '-f', default=10, type=int)
    p.add_argument('--bench', action='store_true', default=False,
                   help='BASELINE configs[4] measurement: time --bench-steps training steps (after --bench-warmup) on synthetic '
                        'annotations and print ONE JSON line (step ms, images/s over all ranks, gradient all-reduce bus GB/s); '
                        'no checkpoint is written')
    p.add_argument('--bench-steps', default=20, type=int)
    p.add_argument('--bench-warmup', default=5, type=int)
    p.add_argument('--grad-compress', default='bf16', choices=['none', 'bf16'],
                   help='DDP gradient all-reduce payload: fp32 (750.9 MB) or bf16-compressed (375.5 MB)')
    g = p.add_argument_group('optimizer configuration')
    g.add_argument('--optimizer', type=str, default='adam', choices=['sgd', 'adam'])
    g.add_argument('--learning-rate', type=float, default=2.5e-4, help='learning rate for world size 1')
    g.add_argument('--momentum', default=0.9, type=float)
    g.add_argument('--weight-decay', '--wd', default=0, type=float)
    return p.parse_args(argv)


def synthetic_targets(seed, batch, size, device):
    """Encoder-style training targets (encoder/heatmap.py, encoder/offset.py conventions) for one batch:
    annos = [(gt_hmp, gt_bghmp, gt_jomp, mask_miss), (gt_off, gt_scale, gt_ps, mask_miss)]."""
    hm, off = synth.synth_batch(seed, batch, size, size, hm_noise=0.0, off_noise=0.0)
    off = np.where(off == 0.0, np.inf, off).astype(np.float32)          # offsets exist in the patches only
    h = size // 4
    rng = synth.HashRng(seed + 17)
    ps = rng.uniform(batch * h * h, 40.0, 300.0).reshape(batch, 1, h, h).astype(np.float32)
    mask = torch.ones(batch, 1, h, h, dtype=torch.bool, device=device)
    t = lambda a: torch.from_numpy(a).to(device)  # noqa: E731
    return [(t(np.clip(hm, 0, 1)), None, None, mask), (t(off), None, t(ps), mask)]


def synthetic_annotations(seed, batch, size):
    """Annotation-style input of the encoders: joints (N,P,17,4) fp32 [x, y, v, scale] padded to P persons, and the
    person count per image (transforms/annotations.py:46-50 layout; the scale column plays the keypoint scale)."""
    scenes = [synth.make_scene(synth.HashRng(seed * 1000003 + i), size, size) for i in range(batch)]
    p_max = max(xy.shape[0] for xy, _, _ in scenes)
    joints = np.zeros((batch, p_max, 17, 4), np.float32)
    for i, (xy, vis, _) in enumerate(scenes):
        p = xy.shape[0]
        joints[i, :p, :, :2] = xy
        joints[i, :p, :, 2] = vis * 2.0
        extent = (xy[..., 1].max(1) - xy[..., 1].min(1)).astype(np.float32)            # person height in pixels
        joints[i, :p, :, 3] = (extent[:, None] * np.asarray(cd.COCO_PERSON_SIGMAS, np.float32)[None]).astype(np.float32)
    return joints, np.array([xy.shape[0] for xy, _, _ in scenes], np.int32)


def encode_targets(encoders, joints, n_persons):
    """Device-side ground truth (offsetguided_amd.encoder = reference encoder/): the same annos layout as
    synthetic_targets, produced from annotations by the HIP encoder kernels inside the step."""
    hm, bg, _, mask = encoders[0].encode_batch(joints, n_persons)
    off, sc, ps, _ = encoders[1].encode_batch(joints, n_persons)
    return [(hm, bg if bg.numel() else None, None, mask), (off, sc if sc.numel() else None, ps, mask)]


def train_step(model, criterion, optimizer, images, annos, lambdas, autocast_dtype=torch.bfloat16):
    """One optimisation step (train_dist.py:304-342).  Returns (loss, per-head losses)."""
    optimizer.zero_grad(set_to_none=True)
    dev_type = 'cuda' if images.is_cuda else 'cpu'
    with torch.autocast(dev_type, dtype=autocast_dtype, enabled=autocast_dtype is not None):
        outputs = model(images)
    multi_losses = []
    for out, lossfun, anno in zip(outputs, criterion, annos):
        out32 = tuple([o.float() if isinstance(o, torch.Tensor) else o for o in part] for part in out)
        multi_losses += list(lossfun(out32, *anno))
    assert len(multi_losses) <= len(lambdas), 'lambdas is incomplete'
    loss = sum(lam * l for lam, l in zip(lambdas, multi_losses))
    if loss.item() > 1e8:  # gradient explosion: drop the batch (train_dist.py:322-325)
        loss = loss * 0.0
    loss.backward()
    optimizer.step()
    return loss.detach(), [float(l.detach()) if torch.is_tensor(l) else float(l) for l in multi_losses]


def bench_steps(args, model, criterion, optimizer, pool, encoders, dev, rank, world):
    """BASELINE configs[4]: step time of the DDP training step (fused HIP losses, device-side GT encoding, bf16 autocast)
    and the gradient all-reduce's bus bandwidth.  The all-reduce overlaps backward inside the step, so its bandwidth is
    measured on its own: the same payload (one flat buffer of the gradients' size, the bucketed hook's dtype) reduced
    back to back over RCCL; bus GB/s = bytes x 2 (N - 1) / N / time (ring convention).  `exposed_comm_ms` = step time
    minus the same step under no_sync()."""
    import contextlib
    import json
    use_cuda = dev.type == 'cuda'
    dist = torch.distributed

    def run(n, first, sync_grads=True):
        ctx = contextlib.nullcontext() if (sync_grads or world == 1) else model.no_sync()
        with ctx:
            for step in range(first, first + n):
                images, annos = pool[step % len(pool)]
                if encoders is not None:
                    annos = encode_targets(encoders, *annos)
                train_step(model, criterion, optimizer, images, annos, args.lambdas, torch.bfloat16 if use_cuda else None)

    def timed(n, first, **kw):
        sharding.barrier(dev if use_cuda else None)
        t0 = time.perf_counter()
        run(n, first, **kw)
        sharding.barrier(dev if use_cuda else None)
        return sharding.max_over_ranks(time.perf_counter() - t0, dev if use_cuda else None) / n

    model.train()
    run(max(args.bench_warmup, 1), 0)
    step_s = timed(args.bench_steps, args.bench_warmup)
    nosync_s = timed(max(args.bench_steps // 2, 1), 0, sync_grads=False) if world > 1 else step_s
    n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    payload_dtype = torch.bfloat16 if (args.grad_compress == 'bf16' and use_cuda) else torch.float32
    nbytes = n_params * (2 if payload_dtype == torch.bfloat16 else 4)
    comm_ms = bus = None
    if world > 1:
        flat = torch.zeros(n_params, dtype=payload_dtype, device=dev)
        for _ in range(2):
            dist.all_reduce(flat)
        sharding.barrier(dev if use_cuda else None)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            dist.all_reduce(flat)
        sharding.barrier(dev if use_cuda else None)
        comm_s = sharding.max_over_ranks(time.perf_counter() - t0, dev if use_cuda else None) / reps
        comm_ms, bus = round(comm_s * 1e3, 3), round(nbytes * 2 * (world - 1) / world / comm_s / 1e9, 1)
    group = sharding.describe_group(dev if use_cuda else None)   # every rank takes part in the gather
    if rank == 0:
        print(json.dumps({
            'rccl': group,
            'metric': 'training images/sec (DDP step: forward + fused losses + backward + all-reduce + fused Adam)',
            'value': round(world * args.batch_size / step_s, 2), 'unit': 'images/sec', 'n_gpus': world,
            'steps': args.bench_steps, 'warmup': args.bench_warmup, 'ms_per_step': round(step_s * 1e3, 2),
            'config': {'workload': f'train_dist DDP, {args.square_length}x{args.square_length} crops, bs{args.batch_size}/GPU, '
                                   'bf16 autocast, focal-L2 hmp + L1 offset losses (BASELINE configs[4])',
                       'sync_bn': bool(args.sync_bn and world > 1), 'grad_payload': str(payload_dtype).replace('torch.', '')},
            'grad_allreduce': {'bytes': nbytes, 'ms': comm_ms, 'bus_GBps': bus,
                               'exposed_comm_ms': round(max(step_s - nosync_s, 0.0) * 1e3, 2) if world > 1 else 0.0},
            'data': 'synthetic annotations, GT encoded on the device'}))
    if dist.is_initialized():
        dist.destroy_process_group()


def main(argv=None):
    args = train_cli(argv)
    rank, local_rank, world = sharding.env_rank()
    use_cuda = torch.cuda.is_available()
    dev = torch.device('cuda', local_rank) if use_cuda else torch.device('cpu')
    if use_cuda:
        torch.cuda.set_device(dev)
        torch.backends.cudnn.benchmark = True
    sharding.init(device=dev if use_cuda else None)
    model, criterion = models.model_factory(args)
    model = model.to(dev)
    if use_cuda:
        model = model.to(memory_format=torch.channels_last)
    if world > 1 and args.sync_bn:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    params = [p for p in model.parameters() if p.requires_grad]
    if args.optimizer == 'adam':
        optimizer = torch.optim.Adam(params, lr=args.learning_rate * world, weight_decay=args.weight_decay, fused=use_cuda)
    else:
        optimizer = torch.optim.SGD(params, lr=args.learning_rate * world, momentum=args.momentum,
                                    weight_decay=args.weight_decay)
    # continue a run (train_dist.py:219-233): weights, optimizer state and the epoch the LR schedule is at.  Before the DDP
    # wrap, so every rank loads the same state and the first broadcast has nothing to fix.
    start_epoch = 0
    if args.resume:
        if not args.checkpoint_whole:
            raise ValueError('--resume needs --checkpoint-whole <file>')
        model, optimizer, start_epoch, start_loss, _ = models.load_model(
            model, args.checkpoint_whole, optimizer=optimizer, resume_optimizer=True, drop_layers=False, optimizer2cuda=use_cuda)
        if rank == 0:
            print(f'resumed {args.checkpoint_whole}: next epoch {start_epoch}, last train loss {start_loss:.4f}')
    elif args.checkpoint_whole:
        # initialise from a checkpoint without its optimizer / epoch (fine-tuning, as evaluate.py:189-191 loads it); a path
        # that does not exist is an error (load_model raises FileNotFoundError), never a silent random init
        model, *_ = models.load_model(model, args.checkpoint_whole, optimizer=None, resume_optimizer=False, drop_layers=False)
    if world > 1:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank] if use_cuda else None,
                                                          bucket_cap_mb=25, gradient_as_bucket_view=True)
        if args.grad_compress == 'bf16' and use_cuda:   # 375.5 MB instead of 750.9 MB over xGMI per step
            from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
            model.register_comm_hook(None, default_hooks.bf16_compress_hook)
    os.makedirs(args.checkpoint_path, exist_ok=True)
    # a small rotating pool of synthetic batches per rank (generating targets on the host every step
    # would measure numpy, not the training step)
    # On the GPU the pool holds ANNOTATIONS and the targets are encoded on the device inside every step
    # (the reference's numpy encoder manages 17 samples/s per dataloader worker, data/factory.py:284).
    pool, encoders = [], None
    if use_cuda:
        encoder.HeatMaps.include_jitter_offset = False
        encoder.HeatMaps.include_background = False
        encoder.OffsetMaps.include_scale = False
        encoders = encoder.factory_heads(['hmp', 'omp'], args.square_length, [4, 4], dev)
    for i in range(4):
        imgs = torch.randn(args.batch_size, 3, args.square_length, args.square_length, device=dev)
        if use_cuda:
            imgs = imgs.contiguous(memory_format=torch.channels_last)
            joints, n_persons = synthetic_annotations(1000 * rank + i, args.batch_size, args.square_length)
            pool.append((imgs, (torch.from_numpy(joints).to(dev), torch.from_numpy(n_persons).to(dev))))
        else:
            pool.append((imgs, synthetic_targets(1000 * rank + i, args.batch_size, args.square_length, dev)))
    if args.bench:
        return bench_steps(args, model, criterion, optimizer, pool, encoders, dev, rank, world)
    batch_time = AverageMeter()   # over the whole run: the first steps (MIOpen find, allocator warm-up) do not bias an epoch
    # --epochs MORE epochs after a resume, as the reference counts them (train_dist.py:269)
    for epoch in range(start_epoch, start_epoch + args.epochs):
        model.train()
        losses, end, last_print = AverageMeter(), time.time(), -1
        for step in range(args.steps_per_epoch):
            adjust_learning_rate(args.learning_rate, world, optimizer, epoch, step, args.steps_per_epoch, args.warmup)
            images, annos = pool[step % len(pool)]
            if encoders is not None:
                annos = encode_targets(encoders, *annos)
            loss, _ = train_step(model, criterion, optimizer, images, annos, args.lambdas,
                                 torch.bfloat16 if use_cuda else None)
            if step % args.print_freq == 0:
                if world > 1:  # averaged over ranks for logging only (train_dist.py:346-348, :458-466)
                    torch.distributed.all_reduce(loss)
                    loss = loss / world
                if use_cuda:
                    torch.cuda.synchronize()
                now = time.time()
                per_step = (now - end) / (step - last_print)   # steps since the last print (1 at step 0), not print_freq
                end, last_print = now, step
                if epoch > start_epoch or step > 0:            # the very first step is warm-up: not a speed sample
                    batch_time.update(per_step)
                losses.update(float(loss))
                if rank == 0:
                    print(f'epoch {epoch} [{step}/{args.steps_per_epoch}] loss {losses.val:.4f} ({losses.avg:.4f}) '
                          f'speed {world * args.batch_size / per_step:.1f} img/s')
        if rank == 0:
            models.save_model(os.path.join(args.checkpoint_path, f'PoseNet_{epoch}_epoch.pth'), epoch, losses.avg, model,
                              optimizer)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
