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
def test_gpu_train_steps_with_device_encoder(tmp_path, monkeypatch):
    """train_dist.main on the GPU: annotations -> HIP encoder -> fused HIP losses -> optimizer, a few steps.  The epoch's
    checkpoint is handed to the test instead of the disk (2.2 GB of Hourglass-104 weights + Adam moments took half of this
    test's time; the file round trip is test_resume_restores_epoch_weights_and_optimizer's)."""
    import torch
    from offsetguided_amd.models import networks
    saved = []
    monkeypatch.setattr(networks.torch, 'save', lambda data, path: saved.append((data, str(path))))
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
    (data, path), = saved
    assert path.startswith(str(tmp_path)) and data['epoch'] == 0 and np.isfinite(data['train_loss'])
    assert all(bool(torch.isfinite(v).all()) for v in data['model_state_dict'].values() if v.is_floating_point())
    steps = [int(st['step']) for st in data['optimizer_state_dict']['state'].values()]
    assert steps and all(n == 3 for n in steps)


def test_resume_restores_epoch_weights_and_optimizer(tmp_path, monkeypatch, capsys):
    """--resume continues from --checkpoint-whole (reference train_dist.py:219-233): weights and optimizer state are
    loaded, training restarts at the NEXT epoch (so the LR schedule does too) and earlier checkpoints are left alone."""
    from offsetguided_amd import models, train_dist
    torch.manual_seed(0)
    monkeypatch.setattr(models, 'model_factory', lambda args: (TinyNet(), _criterion()))
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: False)
    common = ['--no-pretrain', '--square-length', '64', '--batch-size', '2', '--steps-per-epoch', '2', '--print-freq', '1',
              '--checkpoint-path', str(tmp_path), '--lambdas', '1', '0', '0', '100', '0.01']
    train_dist.main(common + ['--epochs', '2'])
    first = sorted(os.listdir(tmp_path))
    assert first == ['PoseNet_0_epoch.pth', 'PoseNet_1_epoch.pth']
    stamp = os.path.getmtime(tmp_path / 'PoseNet_0_epoch.pth')
    ck = torch.load(tmp_path / 'PoseNet_1_epoch.pth', map_location='cpu')
    assert ck['epoch'] == 1 and 'optimizer_state_dict' in ck
    out0 = capsys.readouterr().out
    assert 'epoch 0 [0/2]' in out0 and 'epoch 1 [1/2]' in out0

    # --epochs counts the epochs to train AFTER the resume point, as the reference does (train_dist.py:269)
    train_dist.main(common + ['--epochs', '2', '--resume', '--checkpoint-whole', str(tmp_path / 'PoseNet_1_epoch.pth')])
    out1 = capsys.readouterr().out
    assert 'next epoch 2' in out1 and 'epoch 2 [0/2]' in out1 and 'epoch 0 [' not in out1 and 'epoch 1 [' not in out1
    assert sorted(os.listdir(tmp_path)) == first + ['PoseNet_2_epoch.pth', 'PoseNet_3_epoch.pth']
    assert os.path.getmtime(tmp_path / 'PoseNet_0_epoch.pth') == stamp
    ck3 = torch.load(tmp_path / 'PoseNet_3_epoch.pth', map_location='cpu')
    steps = [int(v['step']) for v in ck3['optimizer_state_dict']['state'].values()]
    assert steps and all(s == 8 for s in steps)                 # 4 epochs x 2 steps: Adam's counters were carried over

    with pytest.raises(ValueError):
        train_dist.main(common + ['--resume'])
    with pytest.raises(FileNotFoundError):       # a mistyped --checkpoint-whole is an error, not a silent random init
        train_dist.main(common + ['--epochs', '1', '--checkpoint-whole', str(tmp_path / 'no_such_file.pth')])


def test_bench_mode_prints_one_json_line_cpu(tmp_path, monkeypatch, capsys):
    import json
    from offsetguided_amd import models, train_dist
    monkeypatch.setattr(models, 'model_factory', lambda args: (TinyNet(), _criterion()))
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: False)
    train_dist.main(['--no-pretrain', '--square-length', '64', '--batch-size', '2', '--checkpoint-path', str(tmp_path), '--bench',
                     '--bench-steps', '3', '--bench-warmup', '1', '--lambdas', '1', '0', '0', '100', '0.01'])
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['value'] > 0 and d['grad_allreduce']['bus_GBps'] is None
    assert os.listdir(tmp_path) == []                            # no checkpoint in bench mode


def _bench8_worker(rank, world, port, tmp, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), OMP_NUM_THREADS='1')
    import contextlib
    import io
    import json
    torch.set_num_threads(1)
    from offsetguided_amd import models, train_dist
    models.model_factory = lambda args: (TinyNet(), _criterion())
    torch.cuda.is_available = lambda: False
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        train_dist.main(['--no-pretrain', '--square-length', '64', '--batch-size', '2', '--checkpoint-path', tmp, '--bench',
                         '--bench-steps', '2', '--bench-warmup', '1', '--lambdas', '1', '0', '0', '100', '0.01'])
    lines = [l for l in buf.getvalue().splitlines() if l.startswith('{')]
    q.put((rank, json.loads(lines[0]) if lines else None))


def test_bench_mode_eight_ranks_gloo(tmp_path):
    """BASELINE configs[4]'s bring-up (reference train_dist.py:151-152, 238-239) with EIGHT ranks on the CPU stand-in model over
    gloo: DDP steps, the stand-alone gradient all-reduce measurement, and exactly one JSON line (rank 0) for the whole job."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench8_worker, args=(r, 8, port, str(tmp_path), q)) for r in range(8)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0] is not None and all(got[r] is None for r in range(1, 8))
    d = got[0]
    assert d['n_gpus'] == 8 and d['steps'] == 2 and d['value'] > 0
    assert d['grad_allreduce']['bus_GBps'] is not None and d['grad_allreduce']['ms'] > 0      # (a tiny payload: the rate rounds to 0.0)
    assert d['rccl']['world'] == 8 and d['rccl']['backend'] == 'gloo'


@pytest.mark.gpu
def test_gpu_train_bench_line():
    """train_dist --bench on the GPU (BASELINE configs[4] hook, one rank): the full Hourglass-104 step at 256x256 bs2."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    r = subprocess.run([sys.executable, '-m', 'offsetguided_amd.train_dist', '--no-pretrain', '--square-length', '256',
                        '--batch-size', '2', '--bench', '--bench-steps', '3', '--bench-warmup', '2'],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['value'] > 0 and d['grad_allreduce']['bytes'] == 2 * 187_729_270 or d['grad_allreduce']['bytes'] > 3e8
