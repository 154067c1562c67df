"""N>1 path on CPU: two gloo processes shard a batch, decode their slices independently (the CPU
oracle stands in for the per-GPU HIP decoder), exchange only the timing MAX and the result gather."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

from offsetguided_amd import sharding


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import oracle
    from offsetguided_amd import synth
    from offsetguided_amd.config import coco_data as cd
    r, _, w = sharding.init(backend='gloo')
    assert (r, w) == (rank, world)
    hm, off = synth.synth_batch(55, 4, 128, 128, n_persons=3)        # the global batch, identical on every rank
    lo, hi = sharding.shard_range(len(hm), rank, world)
    poses, _ = oracle.decode(hm[lo:hi], off[lo:hi], cd.COCO_PERSON_SKELETON, topk_k=16)
    sharding.barrier()
    slowest = sharding.max_over_ranks(1.0 + rank)                     # rank-dependent "elapsed time"
    gathered = sharding.gather_to_rank0(poses)
    if rank == 0:
        ref, _ = oracle.decode(hm, off, cd.COCO_PERSON_SKELETON, topk_k=16)
        ok = len(gathered) == len(ref) and all(a.shape == b.shape and (a == b).all() for a, b in zip(gathered, ref))
        q.put((slowest, ok))
    else:
        assert gathered is None
        q.put((slowest, True))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_rank_sharded_decode_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(abs(s - 2.0) < 1e-9 for s, _ in results)              # MAX over ranks of (1 + rank)
    assert all(ok for _, ok in results)
