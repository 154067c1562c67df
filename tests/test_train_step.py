"""Training step: loss goes down on CPU with the reference's loss stack; LR schedule matches the reference's
table; a 2-rank gloo DDP step keeps the replicas' parameters identical."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from offsetguided_amd.utils import adjust_learning_rate


def test_lr_schedule_table():
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    lr = lambda e, s=0, warm=False: adjust_learning_rate(2.5e-4, 4, opt, e, s, 100, warm)  # noqa: E731
    assert lr(0) == 1e-3 and lr(59) == 1e-3
    assert abs(lr(60) - 1e-3) < 1e-12 and abs(lr(65) - 2e-4) < 1e-12 and abs(lr(70) - 4e-5) < 1e-12
    assert lr(80) == 1e-3 and abs(lr(95) - 1e-4) < 1e-12 and abs(lr(107) - 1e-5) < 1e-12
    assert abs(lr(0, 0, True) - 1e-3 / 1500) < 1e-12 and abs(lr(14, 99, True) - 1e-3) < 1e-12
    assert opt.param_groups[0]['lr'] == lr(14, 99, True)


class TinyNet(torch.nn.Module):
    """Stand-in with the NetworkWrapper output nesting (the real backbone is too slow for a CPU unit test)."""

    def __init__(self):
        super().__init__()
        self.body = torch.nn.Conv2d(3, 8, 3, stride=4, padding=1)
        self.hm = torch.nn.ModuleList([torch.nn.Conv2d(8, 17, 1) for _ in range(2)])
        self.off = torch.nn.ModuleList([torch.nn.Conv2d(8, 38, 1) for _ in range(2)])

    def forward(self, x):
        f = torch.relu(self.body(x))
        return [([h(f) for h in self.hm], [[], []], [[], []]), ([o(f) for o in self.off], [[], []], [[], []])]


def _criterion():
    from offsetguided_amd.models import losses
    return losses.lossfuncs_factory(['hmp', 'omp'], 2, [1, 1], 'focal_l2_loss', 'offset_l1_loss',
                                    'offset_instance_l1_loss', 'scale_l1_loss', True)


def test_train_step_reduces_loss_cpu():
    from offsetguided_amd import train_dist
    torch.manual_seed(0)
    net, crit = TinyNet(), _criterion()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    images = torch.randn(2, 3, 64, 64)
    annos = train_dist.synthetic_targets(3, 2, 64, torch.device('cpu'))
    first = last = None
    for _ in range(25):
        loss, parts = train_dist.train_step(net, crit, opt, images, annos, [1, 0, 0, 100, 0.01], autocast_dtype=None)
        first = float(loss) if first is None else first
        last = float(loss)
    assert len(parts) == 5 and np.isfinite(last) and last < 0.7 * first


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _ddp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from offsetguided_amd import sharding, train_dist
    sharding.init(backend='gloo')
    torch.manual_seed(0)
    net = torch.nn.parallel.DistributedDataParallel(TinyNet())
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    crit = _criterion()
    for step in range(3):                                      # different data per rank, like DistributedSampler
        images = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(100 * rank + step))
        annos = train_dist.synthetic_targets(10 * rank + step, 2, 64, torch.device('cpu'))
        train_dist.train_step(net, crit, opt, images, annos, [1, 0, 0, 100, 0.01], autocast_dtype=None)
    flat = torch.cat([p.detach().flatten() for p in net.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    q.put(bool(torch.equal(gathered[0], gathered[1])))
    dist.destroy_process_group()


def test_ddp_step_keeps_replicas_in_sync_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(results)


@pytest.mark.gpu
def test_gpu_train_steps_with_device_encoder(tmp_path):
    """train_dist.main on the GPU: annotations -> HIP encoder -> fused HIP losses -> optimizer, a few steps."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no HIP device is visible")
    from offsetguided_amd import encoder, train_dist
    joints, n_persons = train_dist.synthetic_annotations(3, 2, 256)
    encoder.HeatMaps.include_jitter_offset = False
    encoder.HeatMaps.include_background = False
    encoder.OffsetMaps.include_scale = False
    encs = encoder.factory_heads(['hmp', 'omp'], 256, [4, 4], 'cuda:0')
    annos = train_dist.encode_targets(encs, torch.from_numpy(joints).cuda(), torch.from_numpy(n_persons).cuda())
    hm, off, ps = annos[0][0], annos[1][0], annos[1][2]
    assert hm.shape == (2, 17, 64, 64) and 0.9 < float(hm.max()) <= 1.0 and off.shape == (2, 38, 64, 64)
    assert bool(torch.isfinite(off).any()) and bool(torch.isinf(off).any()) and float(ps.min()) >= 1.0
    train_dist.main(['--no-pretrain', '--square-length', '256', '--batch-size', '2', '--epochs', '1', '--steps-per-epoch', '3',
                     '--print-freq', '1', '--checkpoint-path', str(tmp_path)])
