"""Batch sharding across GPUs for the inference path.

Images are independent: every rank (one process per GPU, torch.distributed over RCCL/xGMI or
gloo on CPU) decodes its own contiguous slice of the batch with NO data-path collective.  The
only exchanges are control-plane: a barrier + MAX-reduce of the elapsed time for benchmarking
and an optional gather of the finished poses to rank 0 (evaluate.py-style result collection).
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init(backend=None, device=None):
    """Join the process group described by the torchrun environment (no-op for one process)."""
    rank, local_rank, world = env_rank()
    force = os.environ.get('OG_FORCE_DIST') == '1'   # exercise the process-group path with one rank
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        kwargs = {'device_id': device} if (backend == 'nccl' and device is not None) else {}
        dist.init_process_group(backend=backend, **kwargs)
    return rank, local_rank, world


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) slice of n_items for `rank`; sizes differ by at most one."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _on_device(device):
    return device is not None and device.type == 'cuda' and dist.get_backend() == 'nccl'


def barrier(device=None):
    if dist.is_initialized():
        if _on_device(device):
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()
    if device is not None and device.type == 'cuda':
        torch.cuda.synchronize(device)


def max_over_ranks(value, device=None):
    """MAX of a python float over all ranks (the slowest rank defines the job time)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if _on_device(device) else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_to_rank0(local_items):
    """Concatenate per-rank python lists in rank order on rank 0 (None elsewhere)."""
    if not dist.is_initialized():
        return list(local_items)
    out = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
    dist.gather_object(list(local_items), out, dst=0)
    if dist.get_rank() != 0:
        return None
    return [x for part in out for x in part]


def describe_group(device=None):
    """What the process group really is, gathered to rank 0 (None elsewhere): {'world', 'backend', 'device_ids'} -- the
    benchmark lines carry it so that a multi-GPU record proves how many ranks RCCL saw and which device each one used
    (process-group bring-up as train_dist.py:151-152)."""
    mine = None
    if device is not None and device.type == 'cuda':
        mine = torch.cuda.current_device()
    if not dist.is_initialized():
        return {'world': 1, 'backend': None, 'device_ids': [mine]}
    ids = gather_to_rank0([mine])
    if dist.get_rank() != 0:
        return None
    return {'world': dist.get_world_size(), 'backend': dist.get_backend(), 'device_ids': ids}
