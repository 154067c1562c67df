"""`bench.py --gpus N` must really create N ranks (one process per GPU, reference bring-up train_dist.py:151-152).

CPU: the control plane (`--dry-run`: rendezvous, barriers, MAX-over-ranks, rank-0 JSON) with two gloo ranks.
GPU: the whole two-rank bench flow -- over RCCL with one GPU per rank when the box has two, otherwise with both ranks
on the box's one GPU and the control plane over gloo (OG_BENCH_SHARE_DEVICE=1, a test aid)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'OG_FORCE_DIST')}
    env.update(extra)
    return env


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, f'expected ONE JSON line, got {len(lines)}:\n{stdout}'
    return json.loads(lines[0])


def test_gpus_flag_spawns_that_many_ranks_dry_run():
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run'],
                       capture_output=True, text=True, timeout=300, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line['n_gpus'] == 2 and line['steps'] == 3 and line['warmup'] == 1
    assert line['elapsed_s'] >= 0.02          # MAX over ranks: rank 1 sleeps 20 ms, rank 0 only 10
    # the line proves what the process group was: two ranks, their backend, one device entry per rank
    assert line['rccl'] == {'world': 2, 'backend': 'gloo', 'device_ids': [None, None]}
    assert 'GPU_MAX_HW_QUEUES' in line['knobs'] and all(isinstance(v, str) for v in line['knobs'].values())


def test_eight_ranks_dry_run():
    """The 8-GPU launch of the driver (one process per GPU, reference train_dist.py:151-152), control plane only: eight gloo
    ranks rendezvous, meet at the barriers, MAX-reduce the time and rank 0 prints the one line."""
    r = subprocess.run([sys.executable, BENCH, '--gpus', '8', '--steps', '2', '--warmup', '1', '--dry-run'],
                       capture_output=True, text=True, timeout=600, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    line = _json_line(r.stdout)
    assert line['n_gpus'] == 8 and line['rccl'] == {'world': 8, 'backend': 'gloo', 'device_ids': [None] * 8}
    assert line['elapsed_s'] >= 0.08          # MAX over ranks: rank 7 sleeps 80 ms


def test_rank_pinning_picks_the_numa_node_of_the_gpu(monkeypatch):
    """pin_rank: rank r of 8 goes to NUMA node r * nodes // 8 (GPUs hang off the host's nodes in device order), before any HIP
    call; one node, one rank or OG_BENCH_NUMA=0 leave the affinity alone."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    have = sorted(os.sched_getaffinity(0))
    half = max(1, len(have) // 2)
    nodes = [(0, have[:half]), (1, have[half:] or have[:half])]
    monkeypatch.setattr(bench, 'numa_nodes', lambda: nodes)
    monkeypatch.setattr(bench, 'gpu_numa_nodes', lambda: None)       # no topology in sysfs: the device-order rule
    pinned = []
    monkeypatch.setattr(os, 'sched_setaffinity', lambda pid, cpus: pinned.append(sorted(cpus)))
    assert bench.pin_rank(0, 8) == {'node': 0, 'cpus': len(nodes[0][1]), 'source': 'device-order rule'} and pinned[-1] == sorted(nodes[0][1])
    assert bench.pin_rank(3, 8)['node'] == 0 and bench.pin_rank(4, 8)['node'] == 1 and bench.pin_rank(7, 8)['node'] == 1
    n = len(pinned)
    assert bench.pin_rank(0, 1) is None and len(pinned) == n               # one rank: nothing to separate
    monkeypatch.setenv('OG_BENCH_NUMA', '0')
    assert bench.pin_rank(5, 8) is None and len(pinned) == n
    monkeypatch.delenv('OG_BENCH_NUMA')
    monkeypatch.setattr(bench, 'numa_nodes', lambda: nodes[:1])
    assert bench.pin_rank(5, 8) is None and len(pinned) == n               # one node


def _fake_sysfs(root, gpu_nodes, cpu_nodes=2):
    """A sysfs tree with `cpu_nodes` CPU-only KFD nodes followed by one GPU node per entry of gpu_nodes (its NUMA node), render
    minors counting DOWN (enumeration order != minor order, as on real hosts)."""
    for i in range(cpu_nodes):
        d = root / 'class/kfd/kfd/topology/nodes' / str(i)
        d.mkdir(parents=True)
        (d / 'properties').write_text('cpu_cores_count 64\nsimd_count 0\ndrm_render_minor -1\n')
    for g, numa in enumerate(gpu_nodes):
        d = root / 'class/kfd/kfd/topology/nodes' / str(cpu_nodes + g)
        d.mkdir(parents=True)
        minor = 128 + len(gpu_nodes) - 1 - g
        (d / 'properties').write_text(f'cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {minor}\nlocation_id {g}\n')
        dev = root / f'class/drm/renderD{minor}/device'
        dev.mkdir(parents=True)
        (dev / 'numa_node').write_text(f'{numa}\n')


def test_rank_pinning_reads_the_gpu_numa_node_from_sysfs(tmp_path, monkeypatch):
    """pin_rank looks the GPU's NUMA node up (KFD topology -> DRM render node -> PCI numa_node) instead of assuming that GPUs
    hang off the nodes in device order: an enumeration where they do NOT (0 1 1 0 ...) must be followed; -1 / an unreadable
    tree falls back to the rule and says so; HIP_VISIBLE_DEVICES reorders."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod2', BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    layout = [1, 0, 0, 1, 1, 1, 0, -1]
    _fake_sysfs(tmp_path, layout)
    assert bench.gpu_numa_nodes(str(tmp_path), env={}) == layout
    assert bench.gpu_numa_nodes(str(tmp_path), env={'HIP_VISIBLE_DEVICES': '3,0'}) == [1, 1]
    assert bench.gpu_numa_nodes(str(tmp_path), env={'ROCR_VISIBLE_DEVICES': '1,2,3', 'HIP_VISIBLE_DEVICES': '2'}) == [1]
    assert bench.gpu_numa_nodes(str(tmp_path), env={'HIP_VISIBLE_DEVICES': 'GPU-abcdef'}) is None
    assert bench.gpu_numa_nodes(str(tmp_path / 'nothing'), env={}) is None
    # a container that may open only ONE of the host's GPUs: the other nodes' properties are unreadable and are skipped,
    # exactly as the runtime skips them -- device 0 is that GPU
    solo = tmp_path / 'solo'
    _fake_sysfs(solo, layout)
    for g in range(8):
        if g != 6:
            (solo / 'class/kfd/kfd/topology/nodes' / str(2 + g) / 'properties').unlink()
    assert bench.gpu_numa_nodes(str(solo), env={'HIP_VISIBLE_DEVICES': '0'}) == [layout[6]]
    have = sorted(os.sched_getaffinity(0))
    half = max(1, len(have) // 2)
    nodes = [(0, have[:half]), (1, have[half:] or have[:half])]
    monkeypatch.setattr(bench, 'numa_nodes', lambda: nodes)
    real = bench.gpu_numa_nodes
    monkeypatch.setattr(bench, 'gpu_numa_nodes', lambda: real(str(tmp_path), env={}))
    pinned = []
    monkeypatch.setattr(os, 'sched_setaffinity', lambda pid, cpus: pinned.append(sorted(cpus)))
    for r, want in enumerate(layout[:7]):
        got = bench.pin_rank(r, 8)
        assert got == {'node': want, 'cpus': len(nodes[want][1]), 'source': 'sysfs'} and pinned[-1] == sorted(nodes[want][1])
    assert bench.pin_rank(7, 8) == {'node': 1, 'cpus': len(nodes[1][1]), 'source': 'device-order rule'}      # sysfs says -1


def test_a_rank_that_dies_fails_the_whole_job_quickly():
    """A rank-local failure must surface as a non-zero exit of `bench.py --gpus N`, not as peers waiting at the barrier."""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '3', '--warmup', '1', '--dry-run'],
                       capture_output=True, text=True, timeout=300, env=_clean_env(OG_BENCH_FAIL_RANK='1'))
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]      # and no result line
    assert time.time() - t0 < 120


def test_wrong_results_switches_are_refused():
    """OG_ENGINE_WHATIF (layers return uninitialised tensors) and OG_DECODER_LIB (any ablation build) must not produce a
    benchmark line unless --allow-diagnostic, which marks it."""
    for var in ('OG_ENGINE_WHATIF', 'OG_DECODER_LIB'):
        r = subprocess.run([sys.executable, BENCH, '--dry-run'], capture_output=True, text=True, timeout=120,
                           env=_clean_env(**{var: 'c160'}))
        assert r.returncode != 0 and var in r.stderr and '--allow-diagnostic' in r.stderr
    r = subprocess.run([sys.executable, BENCH, '--dry-run', '--allow-diagnostic'], capture_output=True, text=True, timeout=120,
                       env=_clean_env(OG_ENGINE_WHATIF='c160'))
    assert r.returncode == 0 and _json_line(r.stdout)['knobs']['OG_ENGINE_WHATIF'] == 'c160'


def test_gpus_flag_must_match_the_launcher():
    env = _clean_env(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--dry-run'], capture_output=True, text=True, timeout=120,
                       env=env)
    assert r.returncode != 0 and 'does not match WORLD_SIZE' in r.stderr


def test_more_gpus_than_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 64:
        pytest.skip('box has 64 devices')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '64'], capture_output=True, text=True, timeout=120,
                       env=_clean_env(OG_BENCH_SHARE_DEVICE='0'))
    assert r.returncode != 0 and 'HIP device(s) are visible' in r.stderr


@pytest.mark.gpu
def test_two_rank_bench_on_the_gpu():
    """bench.py --gpus 2 as the driver launches it for the scaling curve, on whatever this box has: two devices over RCCL, or both
    ranks on the one device with gloo as the control plane -- with TWO batches in flight per rank (the default: four lanes, four engines
    on a shared device).  Both ranks make progress at a similar rate, the group really has two ranks, and the line carries what an
    8-rank launch needs to be planned: per-rank HBM footprint, host time per step, the hardware-queue setting."""
    import torch
    two = torch.cuda.device_count() >= 2
    env = _clean_env() if two else _clean_env(OG_BENCH_SHARE_DEVICE='1')
    r = subprocess.run([sys.executable, BENCH, '--gpus', '2', '--steps', '6', '--warmup', '2', '--no-cpu-baseline',
                        '--no-extras', '--inflight', '2'],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = _json_line(r.stdout)
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['config']['batches_in_flight'] == 2
    assert line['config']['control_plane'] == ('rccl' if two else 'gloo (ranks share one device: test aid)')
    assert line['rccl']['world'] == 2 and len(line['rccl']['device_ids']) == 2
    per = line['per_rank_images_per_sec']
    assert line['value'] > 0 and len(per) == 2 and abs(per[0] - per[1]) <= 0.15 * max(per), per
    # whole-job value = all ranks' images over the slowest rank's time
    assert abs(line['value'] - 2 * 8 * 6 / (line['ms_per_step'] * 6e-3)) / line['value'] < 1e-3
    assert line['engine'] == {'strict': True, 'torch_conv_calls': 0}
    hbm = line['per_rank_hbm_bytes']
    assert len(hbm) == 2 and all(1 << 30 < b < 64 << 30 for b in hbm), hbm      # weights + two engines' activations: a few GB of the 288
    assert line['host_us_per_step'] > 0 and line['knobs']['GPU_MAX_HW_QUEUES'] == '4'
