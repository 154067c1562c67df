#!/usr/bin/env python3
"""Headline benchmark: images/s end-to-end (Hourglass-104 backbone + heads + HIP decoder) at
640x640, batch 8 per GPU, synthetic data, plus decoder-only ms/img.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run,
                                                        one rank per GPU, no data-path collective)

A step = one batch of 8 images (16 through the backbone with --flip) resident in HBM:
fp16 (default; --dtype bf16) channels-last backbone replayed as a HIP graph -> head outputs (+ synthetic GT-like
maps, see below) -> K1a bicubic x4 -> K1 NMS+top-k -> K2 limb collection -> K3 greedy grouping
-> poses copied to pinned host memory.  Steps are software-pipelined one deep (the host picks
up batch i-1's poses after queueing batch i); the grouping kernel and the pose copy run on a side stream.

Random-init networks emit ~constant maps (no keypoints), which would leave the decoder with
nothing to do; as SURVEY.md section 8(d) prescribes, synthetic GT-like stride-4 maps (4-20
stick-figure persons per image, encoder conventions, noise) are ADDED to the head outputs inside
the timed region, so the decoder sees a realistic candidate load and still depends on the
backbone's result.

Prints ONE JSON line (rank 0) with the driver's keys plus `roofline` (K1, the HBM-bound
hand-written kernel, timed live with HIP events on the launch stream) and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

# the engine's HIP graph replays its branches on the runtime's hardware queues; 4 (the default) measured best
os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')   # kernel arguments in device memory (see offsetguided_amd/__init__.py)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6290.0    # the guide's measured float4-copy rate (79 % of the spec)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16
FLOP_PER_IMAGE = 732.78e9       # 2 x 366.39 GMAC, convs only, 640x640
K1_BYTES_PER_IMAGE = 27_889_280  # 17*640*640*4 read + offset gathers + limbs write
K1F_BYTES_PER_IMAGE = 5_632_000  # SURVEY.md 8(d), fused production path lower bound: (17 + 38) * 160 * 160 * 4 (x2 with flip-test)
METRIC = 'images/sec end-to-end (backbone+decode) @640x640 bs8; decoder-only ms/img'


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--size', type=int, default=640)
    ap.add_argument('--flip', action='store_true', help='flip-test (BASELINE config 3): 2x images through the backbone')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--dtype', choices=['bf16', 'f16'], default='f16',
                    help='engine arithmetic: f16 (default: the reference evaluates in fp16 through apex O2, evaluate.py:92,198-201) or '
                         'bf16; same MFMA rate, fp16 is 8x closer to the fp32 module')
    ap.add_argument('--no-harness', action='store_true', help='skip the evaluate.run_images figure (raw uint8 host images -> result dicts)')
    ap.add_argument('--no-extras', action='store_true',
                    help='headline region only: skip the decoder-only / backbone-only / conv / HBM-cold / flip measurements')
    ap.add_argument('--allow-diagnostic', action='store_true',
                    help='run although a switch that produces WRONG RESULTS or loads a foreign library is set (OG_ENGINE_WHATIF, OG_BENCH_ZERO_WEIGHTS, '
                         'OG_DECODER_LIB): the JSON line then carries "diagnostic": true and is not a measurement')
    ap.add_argument('--no-alt-dtype', '--no-f16', dest='no_alt_dtype', action='store_true',
                    help='skip the figure for the other 16-bit arithmetic (bf16 beside the fp16 headline, or fp16 beside --dtype bf16)')
    ap.add_argument('--dry-run', action='store_true',
                    help='control plane only (CPU test aid): the ranks rendezvous over gloo, run the barrier / MAX-over-ranks '
                         'protocol and rank 0 prints the JSON skeleton with value null; no kernel runs')
    ap.add_argument('--inflight', type=int, default=2,
                    help='batches in flight (default 2): that many engines (one HIP graph each, shared weights) + decoders, batch i whole on '
                         'HIP stream i %% L -- the head and tail of one forward (stem, final layers, heads, decoder: few workgroups) run beside '
                         'the bulk of the next; 1 = one batch at a time (measured +1.6...2.1 %% for 2, nothing more for 3)')
    ap.add_argument('--overlap', action='store_true',
                    help='run the decoder on a second HIP stream beside the next backbone (measured: no gain, the\n'
                         'backbone already saturates the chip, and K1 then competes with the convolutions for HBM)')
    return ap.parse_args()


def bench_init(model, seed):
    """Variance-preserving random init: activations stay O(1) random data (all-zero / denormal
    operands would let the chip clock higher than real inputs do)."""
    g = torch.Generator().manual_seed(seed)
    for name, m in model.named_modules():
        if isinstance(m, torch.nn.Conv2d):
            fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
            gain = 0.5 if name.endswith('conv2') else 1.0
            m.weight.data.normal_(0, gain * (2.0 / fan_in) ** 0.5, generator=g)
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()
            m.running_mean.zero_()
            m.running_var.fill_(1)
    for head in model.headnets:  # heads: weights x 1e-4, zero biases -- the un-normalised features are large enough that |hm| still
        # averages 0.14 at bs8 640x640, with large-scale structure: the synthetic maps ride on that (tools/k1_bench.py
        # --bench-inputs reproduces it for the decoder alone)
        for m in head.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.weight.data.mul_(1e-4)


DIAGNOSTIC_SWITCHES = ('OG_ENGINE_WHATIF', 'OG_DECODER_LIB', 'OG_BENCH_ZERO_WEIGHTS')   # wrong-results / foreign-library switches of the product path


def knobs():
    """Every environment switch that can change what this process measures: all OG_*, HIP_*, HSA_*, MIOPEN_* variables and
    GPU_MAX_HW_QUEUES, as set when the line is printed (bench.py's own setdefaults included)."""
    pre = ('OG_', 'HIP_', 'HSA_', 'MIOPEN_')
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith(pre) or k in ('GPU_MAX_HW_QUEUES', 'PYTORCH_ROCM_ARCH')}


def diagnostic_switches():
    return [k for k in DIAGNOSTIC_SWITCHES if os.environ.get(k, '') and not (k == 'OG_BENCH_ZERO_WEIGHTS' and os.environ[k] == '0')]


def k1_traffic(kernels):
    """HBM bytes per K1 launch from the PMC passes of profiles/ (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes,
    tools/pmc_traffic.py), or None: the newest tracked file is used only if it lists EVERY kernel `roofline.kernel` names and
    its total is the sum of exactly those kernels."""
    import glob
    for tfile in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]*_k1_traffic.json')), reverse=True):
        try:
            t = json.load(open(tfile))
            per = t['per_kernel']
            total = sum(per[k]['read'] + per[k]['write'] for k in kernels)
        except (KeyError, ValueError, OSError):
            return None, None          # the newest file does not describe these kernels: report nothing rather than a stale figure
        if abs(total - t.get('hbm_bytes_per_launch', -1)) > 1e-6 * total:
            return None, None
        return int(t['hbm_bytes_per_launch']), os.path.relpath(tfile, ROOT)
    return None, None


def k1f_counters():
    """Counter figures of the PRODUCTION decoder kernel (band_topk_kernel<..., fused>) from the newest profiles/r*_k1f_pmc_summary.json
    (tools/r06_run1.sh: rocprofv3 --pmc passes over tools/k1_bench.py --forms fused --bench-inputs, tools/k1f_pmc_summary.py), or {}."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]*_k1f_pmc_summary.json')), reverse=True):
        try:
            per = json.load(open(f))['per_kernel']
            band = next(v for k, v in per.items() if k.startswith('band_topk_kernel'))
            merge = next(v for k, v in per.items() if k.startswith('merge_collect_kernel'))
            return {'valu_issue_frac': band['valu_issue_frac'], 'valu_issue_frac_note': 'band kernel: 4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)',
                    'waves_per_simd': band.get('wave_occupancy_per_simd'),
                    'bytes': int(band['hbm_read_bytes'] + band['hbm_write_bytes'] + merge['hbm_read_bytes'] + merge['hbm_write_bytes']),
                    'bytes_note': 'HBM bytes per launch from FETCH_SIZE (x2, gfx950) + WRITE_SIZE, band + merge kernels',
                    'counters_source': os.path.relpath(f, ROOT)}
        except (KeyError, ValueError, OSError, StopIteration):
            return {}
    return {}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def numa_nodes():
    """[(node id, [cpu ids])] from sysfs, nodes that have CPUs only."""
    import glob
    out = []
    for d in sorted(glob.glob('/sys/devices/system/node/node[0-9]*'), key=lambda p: int(p.rsplit('node', 1)[1])):
        try:
            spec = open(os.path.join(d, 'cpulist')).read().strip()
        except OSError:
            continue
        cpus = []
        for part in filter(None, spec.split(',')):
            lo, _, hi = part.partition('-')
            cpus += list(range(int(lo), int(hi or lo) + 1))
        if cpus:
            out.append((int(d.rsplit('node', 1)[1]), cpus))
    return out


def _read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


def gpu_numa_nodes(sysfs='/sys', env=None):
    """NUMA node of every VISIBLE HIP device, in HIP device order, WITHOUT a HIP call (pinning happens before the process
    touches the GPU): the KFD topology lists the GPUs in the order the runtime enumerates them (nodes with simd_count > 0,
    by node id; nodes whose properties are unreadable belong to other containers and are skipped, as the runtime skips them --
    seen on this pool: an 8-GPU host, one readable GPU node, renderD128..156 on node 0, renderD160..184 on node 1); a node's
    drm_render_minor names its DRM device, whose PCI function carries numa_node.
    ROCR_VISIBLE_DEVICES, then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES select and reorder (index lists only; UUID lists
    leave the answer unknown).  -> list of ints (-1 = sysfs does not say), or None when the topology cannot be read."""
    import glob
    env = os.environ if env is None else env
    gpus = []
    paths = glob.glob(os.path.join(sysfs, 'class/kfd/kfd/topology/nodes/[0-9]*'))
    for d in sorted(paths, key=lambda q: int(os.path.basename(q))):
        props = _read(os.path.join(d, 'properties'))
        if props is None:
            continue      # a GPU this process may not open (device cgroup of a container): the runtime does not enumerate it either
        kv = dict(line.split(None, 1) for line in props.splitlines() if len(line.split(None, 1)) == 2)
        if int(kv.get('simd_count', '0')) <= 0:
            continue                                         # a CPU node
        minor = int(kv.get('drm_render_minor', '-1'))
        node = _read(os.path.join(sysfs, f'class/drm/renderD{minor}/device/numa_node')) if minor >= 0 else None
        gpus.append(int(node) if node not in (None, '') else -1)
    if not gpus:
        return None
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        spec = env.get(var)
        if var == 'CUDA_VISIBLE_DEVICES' and env.get('HIP_VISIBLE_DEVICES') is not None:
            continue                                         # HIP reads one of the two
        if spec is None or spec.strip() == '':
            continue
        try:
            idx = [int(t) for t in spec.split(',')]
        except ValueError:
            return None                                      # UUIDs: not resolvable from here
        gpus = [gpus[i] for i in idx if 0 <= i < len(gpus)]
    return gpus


def pin_rank(local_rank, world):
    """One process per GPU, eight of them on one host with a 6 ms step: keep a rank's threads (launch thread, pinned-copy
    helpers; the CPU baseline's workers stay out of this) on the NUMA node of ITS GPU, read from sysfs (gpu_numa_nodes: KFD
    topology -> DRM device -> PCI numa_node).  Only when sysfs has no answer (-1, containers without the topology) the rule of
    thumb applies: the GPUs of an 8-GPU MI355X node hang off the host's NUMA nodes in device order, rank r -> node
    r * nodes // world; `source` in the result says which.  Runs BEFORE the process makes any HIP call; OG_BENCH_NUMA=0
    switches it off.  -> description or None."""
    if os.environ.get('OG_BENCH_NUMA', '1') == '0' or world <= 1 or not hasattr(os, 'sched_setaffinity'):
        return None
    nodes = numa_nodes()
    if len(nodes) < 2:
        return None
    by_id = dict(nodes)
    of_gpu = gpu_numa_nodes()
    if of_gpu is not None and local_rank < len(of_gpu) and of_gpu[local_rank] in by_id:
        node, cpus, source = of_gpu[local_rank], by_id[of_gpu[local_rank]], 'sysfs'
    else:
        node, cpus = nodes[min(local_rank * len(nodes) // world, len(nodes) - 1)]
        source = 'device-order rule'
    allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
    if not allowed:
        return None
    os.sched_setaffinity(0, allowed)
    return {'node': node, 'cpus': len(allowed), 'source': source}


def launch_ranks(a):
    """`python bench.py --gpus N` outside torchrun: start N ranks (one per GPU) through torch.distributed.run and relay
    rank 0's JSON line.  Runs BEFORE this process makes any HIP call (the parent never touches a GPU; counting devices
    does not initialise one) and starts the ranks as CHILD processes -- a process that has initialised the GPU must
    never be replaced by exec.  Mirrors the reference's one-process-per-GPU bring-up (train_dist.py:151-152)."""
    import subprocess
    share = os.environ.get('OG_BENCH_SHARE_DEVICE') == '1'
    have = torch.cuda.device_count()
    if a.gpus > have and not (share or a.dry_run):
        sys.exit(f'bench.py: --gpus {a.gpus} but only {have} HIP device(s) are visible')
    # --max-restarts 0 + a short monitor interval: a rank that dies (engine build failure, missing library) takes the job
    # down with a non-zero exit code instead of leaving its peers at the barrier
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}', '--max-restarts=0',
           '--monitor-interval=1', '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // a.gpus)))
    return subprocess.call(cmd, env=env)


def dry_run(a, sharding):
    """The N-rank control plane without a GPU: same rendezvous, barriers and MAX-reduce as the real run."""
    rank, local_rank, world = sharding.env_rank()
    numa = pin_rank(local_rank, world)
    rank, _, world = sharding.init(backend='gloo')
    if os.environ.get('OG_BENCH_FAIL_RANK') == str(rank):   # test aid: a rank that dies before the first barrier
        sys.exit(f'bench.py: rank {rank} told to fail (OG_BENCH_FAIL_RANK)')
    sharding.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    sharding.barrier()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0)
    group = sharding.describe_group(None)
    if rank == 0:
        print(json.dumps({'metric': METRIC, 'value': None, 'unit': 'images/sec', 'n_gpus': world, 'steps': a.steps,
                          'warmup': a.warmup, 'ms_per_step': None, 'dry_run': True, 'elapsed_s': round(elapsed, 4),
                          'rccl': group, 'numa': numa, 'knobs': knobs(), **({'diagnostic': True} if diagnostic_switches() else {})}))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    a = parse()
    env_world = int(os.environ.get('WORLD_SIZE', '1'))
    if a.gpus != env_world:
        if 'RANK' in os.environ or 'LOCAL_RANK' in os.environ:   # under a launcher whose world size disagrees with --gpus
            sys.exit(f'bench.py: --gpus {a.gpus} does not match WORLD_SIZE={env_world} of the launcher')
        sys.exit(launch_ranks(a))
    diag = diagnostic_switches()
    if diag and not a.allow_diagnostic:
        sys.exit('bench.py: ' + ', '.join(f'{k}={os.environ[k]}' for k in diag) + ' is set: that switch makes the product path '
                 'return wrong results or load a foreign library; a line measured with it is not a benchmark.  Unset it, or pass '
                 '--allow-diagnostic (the line is then marked "diagnostic": true)')
    from offsetguided_amd import sharding
    if a.dry_run:
        return dry_run(a, sharding)
    rank, local_rank, world = sharding.env_rank()
    numa = pin_rank(local_rank, world)          # before the first HIP call of this process
    assert torch.cuda.is_available(), 'bench.py needs a HIP device (no CPU path)'
    # OG_BENCH_SHARE_DEVICE=1 (test aid for 1-GPU boxes): all ranks on device 0, control plane over gloo
    share = os.environ.get('OG_BENCH_SHARE_DEVICE') == '1'
    dev = torch.device('cuda', 0 if share else local_rank)
    torch.cuda.set_device(dev)
    sharding.init(backend='gloo' if share else 'nccl', device=dev)   # one process per GPU; RCCL only for barrier + timing MAX

    from offsetguided_amd import _lib, decoder, models, synth
    from offsetguided_amd.config import coco_data as cd
    _lib.load()

    # ---- model + decoder ----
    p = argparse.ArgumentParser()
    models.net_cli(p)
    decoder.decoder_cli(p)
    margs = p.parse_args(['--no-pretrain', '--topk', '32', '--thre-hmp', '0.04', '--person-thre', '0.04',
                          '--dist-max', '40'])
    margs.batch_size = a.batch
    model, _ = models.model_factory(margs)
    bench_init(model, 1234)
    if os.environ.get('OG_BENCH_ZERO_WEIGHTS') == '1':
        # timing diagnosis (refused without --allow-diagnostic): all-zero weights -> every MFMA of the forward runs on zeros, the same
        # instruction stream at the lowest switching energy: what the step would take if the chip held its clock
        for prm in model.parameters():
            prm.data.zero_()
    n_rot = 3

    class Pipeline:
        """One configuration of the hot path: engine (HIP graph) + decoder + resident synthetic inputs."""

        def __init__(self, flip, inflight=1, dtype=None):
            self.flip = flip
            self.host_s = []
            self.nb = a.batch * (2 if flip else 1)
            dtype = dtype or a.dtype
            self.engines = [models.InferenceEngine(model, self.nb, a.size, a.size, dtype=torch.float16 if dtype == 'f16' else torch.bfloat16, device=dev,
                                                   use_graph=not a.no_graph) for _ in range(inflight)]
            self.procs = [decoder.decoder_factory(margs) for _ in range(inflight)]
            self.engine, self.proc = self.engines[0], self.procs[0]
            self.lanes = _lib.lane_streams(dev, inflight) if inflight > 1 else None    # the process's own streams, not torch's pool
            self.images = [torch.randn(self.nb, 3, a.size, a.size, device=dev,
                                       generator=torch.Generator(dev).manual_seed(rank * 100 + r)) for r in range(n_rot)]
            self.maps = []
            for r in range(n_rot):
                hm, off = synth.synth_batch(1000 * rank + r, a.batch, a.size, a.size, flip=flip)
                self.maps.append((torch.from_numpy(hm).to(dev), torch.from_numpy(off).to(dev), hm, off))

        def features(self, i, engine=None):
            hm_o, off_o = (engine or self.engine).forward_raw(self.images[i % n_rot])
            hm, off = hm_o + self.maps[i % n_rot][0], off_o + self.maps[i % n_rot][1]
            return [([None, hm], [[], []], [[], []]), ([None, off], [[], []], [[], []])]

        def run_steps(self, n, first=0, serial=False):
            """Backbone + decoder queued per batch; the host picks up batch i-1's poses after queueing batch i (HIP streams,
            event-ordered; no host sync besides the pose pick-up).  serial: one batch at a time on the first engine whatever
            --inflight says (the roofline region: K1 must not share HBM with another batch's convolutions)."""
            if self.lanes is not None and not serial:
                return self.run_steps_lanes(n, first)
            main_stream = torch.cuda.current_stream(dev)
            dec_stream = torch.cuda.Stream(dev) if a.overlap else main_stream
            pending, out = None, None
            for i in range(first, first + n):
                t_host = time.perf_counter()
                feats = self.features(i)
                if a.overlap:
                    ready = torch.cuda.Event()
                    ready.record(main_stream)
                    dec_stream.wait_event(ready)
                    for t in (feats[0][0][1], feats[1][0][1]):
                        t.record_stream(dec_stream)
                with torch.cuda.stream(dec_stream):
                    nxt = self.proc.submit(feats, flip_test=self.flip)
                self.host_s.append(time.perf_counter() - t_host)     # host time to ENQUEUE one step (no wait in it)
                if pending is not None:
                    out = pending.result()
                pending = nxt
            out = pending.result()
            torch.cuda.synchronize(dev)
            return out

        def run_steps_lanes(self, n, first=0):
            """--inflight L: batch i runs whole (backbone graph + decoder) on lane i % L; the host picks up batch i - L."""
            pending, out, L = [], None, len(self.lanes)
            for ln in self.lanes:
                ln.wait_stream(torch.cuda.current_stream(dev))
            for i in range(first, first + n):
                j = i % L
                t_host = time.perf_counter()
                with torch.cuda.stream(self.lanes[j]):
                    nxt = self.procs[j].submit(self.features(i, self.engines[j]), flip_test=self.flip)
                self.host_s.append(time.perf_counter() - t_host)
                pending.append(nxt)
                if len(pending) > L:
                    out = pending.pop(0).result()
            while pending:
                out = pending.pop(0).result()
            torch.cuda.synchronize(dev)
            return out

        def timed_region(self, steps, warmup, serial=False):
            """W untimed steps, barrier, EXACTLY `steps` timed steps, barrier; MAX over ranks."""
            self.run_steps(max(warmup, 1), serial=serial)
            sharding.barrier(dev)
            _lib.profile_start()
            self.host_s = []
            t0 = time.perf_counter()
            poses = self.run_steps(steps, first=warmup, serial=serial)
            sharding.barrier(dev)
            mine = time.perf_counter() - t0
            return poses, mine, sharding.max_over_ranks(mine, dev), _lib.profile_stop()

    pipe = Pipeline(a.flip, a.inflight)
    engine, proc, maps, images, nb = pipe.engine, pipe.proc, pipe.maps, pipe.images, pipe.nb
    poses, elapsed_rank, elapsed, stage_us = pipe.timed_region(a.steps, a.warmup)
    pipe_host = list(pipe.host_s)
    # every convolution of the forward on a hand-written kernel: the engines are strict (a shape none of them serves raises instead of
    # running on torch / MIOpen); the counter is what a strict=False caller would read
    engine_info = {'strict': all(e.strict for e in pipe.engines), 'torch_conv_calls': sum(len(e.torch_conv_calls) for e in pipe.engines)}
    per_rank = sharding.gather_to_rank0([round(a.batch * a.steps / elapsed_rank, 2)])   # control plane only
    # HBM footprint of a rank after the headline region (weights + L engines' activations and graphs + decoder workspaces + inputs): an
    # 8-rank launch's memory is known in advance
    per_rank_hbm = sharding.gather_to_rank0([int(torch.cuda.max_memory_allocated(dev))])
    # The production decoder is K1-fused (PostProcess.fused_upsample): no hi-res tensor, no HBM-streaming K1 in the
    # headline region.  The roofline figure stays defined on K1 at the generate_limbs boundary: the SAME pipeline -- same
    # engine, same inputs, same steps, every rank -- runs a second timed region in its roofline benchmark mode (K1a materialises the
    # hi-res heat maps, K1 streams them) and K1's HIP events of THAT region are what `roofline` reports.
    roof_stage, roof_elapsed = stage_us, None
    if 'k1_generate_limbs' not in stage_us:
        for pr in pipe.procs:
            pr.fused_upsample = False
        _, _, roof_elapsed, roof_stage = pipe.timed_region(a.steps, a.warmup, serial=True)
        for pr in pipe.procs:
            pr.fused_upsample = True

    # what a latency-sensitive caller (--inflight 1, one batch at a time) sees of the PRODUCTION decoder: K1-fused + K3 by their HIP events
    # in a serial region of the same pipeline (with two batches in flight the same launches share the CUs with the other batch's
    # convolutions: stage_us of the headline region; DESIGN 4 "The decoder beside the other batch")
    serial_stage = None
    if not a.no_extras and a.inflight > 1 and 'k1f_fused_limbs' in stage_us:
        _, _, _, serial_stage = pipe.timed_region(a.steps, a.warmup, serial=True)

    def timed(fn, n):
        torch.cuda.synchronize(dev)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(n):
            fn(i)
        e.record()
        torch.cuda.synchronize(dev)
        return s.elapsed_time(e) / n  # ms

    extras = {}
    if serial_stage is not None:
        k1f_1, k3_1 = float(np.mean(serial_stage['k1f_fused_limbs'])), float(np.mean(serial_stage['k3_group']))
        extras['decoder_ms_per_img_inflight1'] = round((k1f_1 + k3_1) * 1e-3 / a.batch, 4)
        extras['decoder_us_per_batch_inflight1'] = {'k1f_fused_limbs': round(k1f_1, 2), 'k3_group': round(k3_1, 2)}
    if not a.no_extras:
        # ---- decoder-only and backbone-only timings (outside the headline region) ----
        fixed = [[([None, m[0]], [[], []], [[], []]), ([None, m[1]], [[], []], [[], []])] for m in maps]
        was_fused = proc.fused_upsample
        proc.fused_upsample = False  # K1a + K1: the hi-res heat maps are materialised and streamed (the roofline benchmark mode)
        dec_ms = timed(lambda i: proc.limb_group.group_device(proc.generate_limbs(fixed[i % n_rot], flip_test=a.flip)), 20)
        bb_ms = timed(lambda i: engine.forward_raw(images[i % n_rot]), 10)
        proc.fused_upsample = True   # K1-fused (production): bicubic inside the NMS kernel, no hi-res tensor (same results)
        dec_fused_ms = timed(lambda i: proc.limb_group.group_device(proc.generate_limbs(fixed[i % n_rot], flip_test=a.flip)), 20)
        proc.fused_upsample = was_fused
        extras.update(decoder_ms_per_img=round(dec_ms / a.batch, 4), decoder_ms_per_img_fused_upsample=round(dec_fused_ms / a.batch, 4),
                      backbone_ms_per_batch=round(bb_ms, 3),
                      backbone_tflops=round(nb * FLOP_PER_IMAGE * (a.size * a.size) / (640 * 640) / (bb_ms * 1e-3) / 1e12, 1))

        # C1, the dominant backbone kernel (13 of the 3x3 convolutions, 54 % of the network's FLOPs): the 160x160 256->256
        # layer through og_conv3x3_tiled_bf16, bias + residual + ReLU epilogue included, HIP events around back-to-back launches
        # on rotating activations (the network hands it activations the previous layer just wrote)
        if a.size == 640:
            lib = _lib.load()
            cl = torch.channels_last
            lp_t = torch.float16 if a.dtype == 'f16' else torch.bfloat16
            # inputs, residual operands and outputs are SEPARATE buffers: fed back into each other (as this block did until round 5)
            # the activations grow by ~1.3x per launch, overflow fp16 after ~40 launches and the rest of the measurement runs on
            # inf / NaN operands -- which the chip multiplies at a higher clock (175 us per launch instead of 217)
            xs = [torch.randn(nb, 256, 160, 160, device=dev).to(lp_t).contiguous(memory_format=cl) for _ in range(9)]
            wt = (torch.randn(256, 256, 3, 3, device=dev) * (1.0 / 2304) ** 0.5).to(lp_t).contiguous(memory_format=cl)
            cb = torch.zeros(256, device=dev)
            packed = torch.empty(wt.numel(), dtype=lp_t, device=dev)
            conv_fn = _lib.lp(lib, 'og_conv3x3_tiled', lp_t)
            _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wt), 256, 256, 0, _lib.ptr(packed), _lib.stream_ptr(dev)), lib)

            def conv_once(i):
                _lib.check(conv_fn(_lib.ptr(xs[i % 3]), _lib.ptr(packed), _lib.ptr(cb), _lib.ptr(xs[3 + i % 3]),
                                                     _lib.ptr(xs[6 + i % 3]), nb, 160, 160, 256, 256, 1, None, 0, _lib.stream_ptr(dev)), lib)
            # the figure is the layer's time on a chip that is already working (60 launches before the 30 timed ones): from idle the
            # first dozen launches run ~15 % slower (the clock ramps UP for ~10 ms: a layer bench that warmed up with five launches
            # read 226 us where the layer takes 198 -- round 5); the 3-launch burst from an idle chip is reported beside it
            timed(conv_once, 60)
            conv_us = timed(conv_once, 30) * 1e3
            time.sleep(0.25)
            burst_us = timed(conv_once, 3) * 1e3
            conv_flop = 2.0 * nb * 160 * 160 * 256 * 2304
            extras['roofline_conv3x3'] = {
                'kernel': 'C1 = og_conv3x3_tiled_' + a.dtype + ' (conv3x3_tiled_kernel<16,16,4>: two workgroups per CU, pre-tiled weights) on the 160x160 '
                          '256->256 layer, residual + bias + ReLU epilogue fused',
                'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': MFMA_BF16_PEAK_TFLOPS, 'us_per_launch': round(conv_us, 1),
                'us_per_launch_burst_of_3': round(burst_us, 1), 
                'achieved': round(conv_flop / (conv_us * 1e-6) / 1e12, 1),
                'frac': round(conv_flop / (conv_us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                'algorithmic_flop_per_launch': conv_flop}
            del xs
        bb_tf = nb * FLOP_PER_IMAGE / (bb_ms * 1e-3) / 1e12
        extras['roofline_backbone'] = {'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': MFMA_BF16_PEAK_TFLOPS,
                                       'achieved': round(bb_tf, 1), 'frac': round(bb_tf / MFMA_BF16_PEAK_TFLOPS, 4)}

        # K1 on HBM-cold inputs: rotate hi-res heatmap batches whose total exceeds the 256 MiB Infinity Cache
        hr = [decoder.factory.upsample4(m[0][:a.batch], 'bicubic') for m in maps]
        _lib.profile_start()
        for i in range(12):
            proc.limb_collect.generate_limbs_lowres(hr[i % n_rot], maps[i % n_rot][1][:a.batch])
        cold = _lib.profile_stop()
        extras['k1_cold_us'] = float(np.mean(cold['k1_generate_limbs'][3:]))
        del hr

    # ---- BASELINE configs[2] (flip-test) in the same process, same steps: 2x images through the backbone, K0 merge ----
    flip_line = None
    if not a.no_extras and not a.flip:
        del pipe, engine, proc
        torch.cuda.empty_cache()
        fpipe = Pipeline(True, a.inflight)
        _, f_rank, f_elapsed, f_stage = fpipe.timed_region(a.steps, a.warmup)
        flip_line = {'value': round(a.batch * a.steps * world / f_elapsed, 2), 'unit': 'images/sec',
                     'ms_per_step': round(f_elapsed / a.steps * 1e3, 3),
                     'k0_us': round(float(np.mean(f_stage['k0_flip_merge'])), 2) if 'k0_flip_merge' in f_stage else 0.0,   # 0: the merge rides on K1a / K1
                     **({'k1a_upsample_us': round(float(np.mean(f_stage['k1a_upsample'])), 2),
                         'k1_generate_limbs_us': round(float(np.mean(f_stage['k1_generate_limbs'])), 2)} if 'k1_generate_limbs' in f_stage else
                        {'k1f_fused_limbs_us': round(float(np.mean(f_stage['k1f_fused_limbs'])), 2)}),
                     'workload': f'bs{a.batch} {a.size}x{a.size} + flip-test: {2 * a.batch} images through the backbone per step, ' +
                                 ('K0 flip merge as its own pass' if 'k0_flip_merge' in f_stage else 'flip merge folded into the loads of ' + ('K1a / K1' if 'k1_generate_limbs' in f_stage else 'K1-fused') + ' (no K0 pass)') + ', full decoder (BASELINE configs[2])'}
        del fpipe

    # ---- the other 16-bit arithmetic timed in the same process, same steps (headline fp16 = the reference's apex-O2 arithmetic,
    # evaluate.py:92,198-201; the block beside it is the bf16 engine) ----
    alt_line, alt_dtype = None, ('bf16' if a.dtype == 'f16' else 'f16')
    if not a.no_extras and not a.no_alt_dtype and not a.flip:
        torch.cuda.empty_cache()
        hpipe = Pipeline(False, a.inflight, dtype=alt_dtype)
        _, _, h_elapsed, h_stage = hpipe.timed_region(a.steps, a.warmup)
        h_bb = timed(lambda i: hpipe.engine.forward_raw(hpipe.images[i % n_rot]), 10)
        alt_line = {'value': round(a.batch * a.steps * world / h_elapsed, 2), 'unit': 'images/sec',
                    'ms_per_step': round(h_elapsed / a.steps * 1e3, 3), 'backbone_ms_per_batch': round(h_bb, 3),
                    **({'k1_generate_limbs_us': round(float(np.mean(h_stage['k1_generate_limbs'])), 2)} if 'k1_generate_limbs' in h_stage else
                       {'k1f_fused_limbs_us': round(float(np.mean(h_stage['k1f_fused_limbs'])), 2)}),
                    'workload': 'the headline workload with the %s engine (og_*_%s kernels: same MFMA rate; fp16 has 3 more mantissa '
                                'bits and is what the reference evaluates in)' % (alt_dtype, alt_dtype)}
        del hpipe

    # ---- the drop-in harness itself: evaluate.run_images fed raw uint8 HWC host images of mixed sizes (pageable memory) --
    # EvalPreprocess (pinned H2D + og_rescale_pad_normalize_u8) -> engine -> PostProcess.submit -> poses_to_results
    harness = None
    if not a.no_extras and not a.no_harness and not a.flip and rank == 0:
        torch.cuda.empty_cache()
        # the figure depends on what else runs on the (shared) host: three passes, the MEDIAN is `value`, all listed, the best one
        # kept as `best` (never quoted as "the" harness rate)
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):      # run_images prints its progress lines: stdout carries the ONE JSON line
            runs = [harness_block(a, model, dev) for _ in range(3)]
        harness = sorted(runs, key=lambda h: h['value'])[1]
        harness['runs'] = [h['value'] for h in runs]
        harness['best'] = max(harness['runs'])

    group = sharding.describe_group(dev)
    host_us = 1e6 * float(np.mean(pipe_host)) if pipe_host else None
    if rank == 0:
        fused_k1 = False
        k1 = float(np.mean(roof_stage['k1_generate_limbs']))
        k1_bytes = a.batch * K1_BYTES_PER_IMAGE * (a.size * a.size) / (640 * 640)
        achieved = k1_bytes / (k1 * 1e-6) / 1e9
        # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes, profiles/README.md): only a figure
        # measured on THIS round's kernels is reported
        k1_kernels = ('band_topk_kernel', 'merge_collect_kernel')
        traffic, traffic_file = (None, None) if fused_k1 else k1_traffic(k1_kernels)
        imgs = a.batch * a.steps * world
        line = {
            'metric': METRIC,
            'value': round(imgs / elapsed, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': round(elapsed / a.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': f'bs{a.batch} {a.size}x{a.size}' + (' + flip-test' if a.flip else '') +
                                   ': Hourglass-104+heads (%s, HIP graph) -> HIP decoder' % a.dtype + ' topk=32, 17 heatmaps, 19 limbs'
                                   ' (BASELINE configs[%d])' % (2 if a.flip else 1),
                       'per_gpu_batch': a.batch, 'batches_in_flight': a.inflight,
                       'parallelism': f'batch-sharded x{world}, no collectives',
                       'control_plane': 'gloo (ranks share one device: test aid)' if share else 'rccl',
                       'decoder_input': 'head outputs + synthetic GT-like maps'},
            'per_rank_images_per_sec': per_rank,
            'per_rank_hbm_bytes': per_rank_hbm,
            'poses_last_batch': [int(len(x)) for x in poses],
            'stage_us': {k: round(float(np.mean(v)), 2) for k, v in stage_us.items()},
            'roofline': {'kernel': 'K1 at the generate_limbs boundary = og_generate_limbs_f32: band_topk_kernel + '
                                   'merge_collect_kernel' + ('' if roof_elapsed is None else
                                   ', HIP events inside a second timed region of the same pipeline in its roofline benchmark mode '
                                   '(fused_upsample=False: K1a + K1, one batch at a time); the headline region runs the production decoder K1-fused '
                                   '(stage_us.k1f_fused_limbs), which never builds the hi-res tensor'),
                         'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 4),
                         # against what a float4 copy reaches on this chip (6.29 TB/s, MI355X_MICROARCH.md "HBM"): the stream
                         # kernel alone runs at ~0.85 of that; the generate_limbs boundary adds the merge / pairing launch
                         'frac_of_achievable': round(achieved / HBM_ACHIEVABLE_GBS, 4),
                         'traffic': traffic, 'traffic_source': traffic_file,
                         'us_per_launch': round(k1, 2), 'algorithmic_bytes_per_launch': int(k1_bytes),
                         **({} if roof_elapsed is None else
                            {'region': {'images_per_sec': round(imgs / roof_elapsed, 2), 'ms_per_step': round(roof_elapsed / a.steps * 1e3, 3),
                                        'stage_us': {k: round(float(np.mean(v)), 2) for k, v in roof_stage.items()}}})},
            **({'k1f': {'kernel': 'K1-fused (production) = og_generate_limbs_fused_f32: band_topk_kernel<fused> (x4 bicubic + NMS + top-k from '
                                  'the stride-4 maps) + merge_collect_kernel<fused>; instruction-issue bound (valu_issue_frac from the PMC passes), no HBM roofline claim (SURVEY 8d)',
                        'us_per_launch': round(float(np.mean(stage_us['k1f_fused_limbs'])), 2),
                        'lowres_bytes_per_launch': int(a.batch * K1F_BYTES_PER_IMAGE * (2 if a.flip else 1) * (a.size * a.size) / (640 * 640)),
                        'effective_GBps': round(a.batch * K1F_BYTES_PER_IMAGE * (2 if a.flip else 1) * (a.size * a.size) / (640 * 640) /
                                                (float(np.mean(stage_us['k1f_fused_limbs'])) * 1e-6) / 1e9, 1),
                        **k1f_counters()}}
               if 'k1f_fused_limbs' in stage_us else {}),
            'engine': engine_info,
            'rccl': group,
            'host_us_per_step': None if host_us is None else round(host_us, 1),   # host time to enqueue one step (8 ranks share one host)
            'numa': numa,
            'knobs': knobs(),
        }
        if harness is not None:
            line['harness'] = harness
        if diag:
            line['diagnostic'] = True
        if 'k1_cold_us' in extras:
            k1_cold = extras.pop('k1_cold_us')
            line['roofline']['hbm_cold'] = {'us_per_launch': round(k1_cold, 2),
                                            'achieved': round(k1_bytes / (k1_cold * 1e-6) / 1e9, 1),
                                            'frac': round(k1_bytes / (k1_cold * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        line.update(extras)
        if flip_line is not None:
            line['flip'] = flip_line
        if alt_line is not None:
            line[alt_dtype] = alt_line
        if world == 1 and not a.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(a, model, maps, cd)
        print(json.dumps(line))
    import torch.distributed as dist
    if dist.is_initialized():
        sharding.barrier(dev)
        dist.destroy_process_group()


def harness_block(a, model, dev, n_batches=24, warm=6):
    """img/s of offsetguided_amd.evaluate.run_images (reference evaluate.py:125-300) from raw host images to COCO result dicts:
    batches of (h, w, 3) uint8 RGB arrays of eight COCO-like sizes in pageable memory; the first `warm` batches build the engine and
    allocate the pinned staging / landing buffers (a hipHostMalloc costs 50-60 ms: one inside the timed region reads as 8.4 instead of
    6.3 ms per batch), the next `n_batches` are timed with the host clock (device synchronised on both sides)."""
    from offsetguided_amd import evaluate
    rng = np.random.default_rng(0)
    sizes = [(480, 640), (427, 640), (640, 480), (375, 500), (500, 375), (640, 640), (333, 500), (612, 612)]
    base = [rng.integers(0, 256, size=hw + (3,), dtype=np.uint8) for hw in sizes]
    args = evaluate.evaluate_cli(['--no-pretrain', '--topk', '32', '--thre-hmp', '0.04', '--person-thre', '0.04', '--dist-max', '40',
                                  '--batch-size', str(a.batch), '--long-edge', str(a.size), '--print-freq', '1000000000'])
    marks = {}

    def loader():
        # run_images reads ONE batch ahead: batch b is requested (and its pack starts on the worker) right before batch b - 1 is
        # processed.  The clock runs from the request of batch `warm` to the request of batch `warm + n_batches`, the device drained at
        # both ends: exactly n_batches packs (warm .. warm + n - 1) and exactly n_batches batches processed from start to finish
        # (warm - 1 .. warm + n - 2) lie inside, plus one refill of the pipeline.  One more batch follows so that the last request exists.
        for b in range(n_batches + warm + 1):
            if b == warm or b == warm + n_batches:
                if b == warm:
                    import gc
                    gc.collect()       # the engine build of the warm-up batches leaves garbage: its collection belongs there, not in the region
                torch.cuda.synchronize(dev)
                marks['t0' if b == warm else 't1'] = time.perf_counter()
            imgs = [base[(b + i) % len(base)] for i in range(a.batch)]
            yield imgs, [None] * a.batch, [{'image_id': b * a.batch + i} for i in range(a.batch)]
    stats = {}
    results, ids = evaluate.run_images(args, loader(), model=model, stats=stats)
    torch.cuda.synchronize(dev)
    dt = marks['t1'] - marks['t0']
    assert len(ids) == (n_batches + warm + 1) * a.batch
    return {'value': round(n_batches * a.batch / dt, 2), 'unit': 'images/sec', 'batches': n_batches,
            'ms_per_batch': round(dt / n_batches * 1e3, 3),
            # host time to queue one batch (input chain + forward + decoder; no wait in it) over the timed batches
            'host_us_per_step': round(1e6 * float(np.mean(stats['host_enqueue_s'][warm:warm + n_batches])), 1),
            'engine': {'strict': True, 'engines_built': stats.get('engines_built'), 'torch_conv_calls': stats.get('torch_conv_calls')},
            'input': f'{a.batch} raw (h, w, 3) uint8 RGB host images per batch, eight COCO-like sizes (333x500 ... 640x640), pageable memory',
            'stages': 'EvalPreprocess (pinned staging + one H2D copy + ONE og_rescale_pad_normalize_batch_u8 launch per batch) -> InferenceEngine -> '
                      'PostProcess.submit -> annotations_inverse + COCO result dicts (evaluate.run_images, reference evaluate.py:125-300)',
            'note': 'head outputs of the random-init network (no synthetic maps added): the decoder sees few candidates here'}


def cpu_baseline(a, model, maps, cd):
    """The reference-shaped CPU path on this box's host cores: oracle/restatement.py = torch-CPU
    upsample / pad+max_pool NMS / topk / gathers + numpy grouping in a Pool(batch), all cores, 1 warm-up + median of 5, on
    one batch of the same synthetic maps; the eager fp32 backbone on one image, 1 warm-up + median.  The scalar C oracle
    (1 thread) is kept as an extra figure.  A reported baseline, not a target."""
    import oracle
    from oracle import restatement as rs
    flags = dict(topk_k=32, thre_hmp=0.04, min_len=0.5, person_thre=0.04, dist_max=40.0)
    if a.flip:
        perm, rev = cd.offset_hflip(cd.COCO_KEYPOINTS, cd.COCO_PERSON_SKELETON)
        flags['flip'] = (cd.heatmap_hflip(cd.COCO_KEYPOINTS), perm, sorted(rev))
    hm_np, off_np = maps[0][2], maps[0][3]
    dec_s, cores = rs.time_decoder(hm_np, off_np, cd.COCO_PERSON_SKELETON, flags, repeats=5)
    bb_s, bb_runs = rs.time_backbone(model, a.size, repeats=5, budget_s=40.0)
    t0 = time.perf_counter()
    n_c = 2
    cflags = dict(flags)
    if a.flip:
        sel = [0, 1, a.batch, a.batch + 1]
        oracle.decode(hm_np[sel], off_np[sel], cd.COCO_PERSON_SKELETON, **cflags)
    else:
        oracle.decode(hm_np[:n_c], off_np[:n_c], cd.COCO_PERSON_SKELETON, **cflags)
    c_ms = (time.perf_counter() - t0) / n_c * 1e3
    per_img = bb_s * (2 if a.flip else 1) + dec_s / a.batch
    # the spread over the individual runs (a 128-thread Pool on a shared host is noisy: the value moved 0.33 ... 0.54 between
    # runs of rounds 2 and 3): images/s from the fastest and the slowest decoder / backbone run
    d_runs, b_runs = getattr(rs.time_decoder, 'last_runs', [dec_s]), getattr(rs.time_backbone, 'last_runs', [bb_s])
    rate = lambda b, d: 1.0 / (b * (2 if a.flip else 1) + d / a.batch)   # noqa: E731
    return {'value': round(1.0 / per_img, 4), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'value_range': [round(rate(max(b_runs), max(d_runs)), 4), round(rate(min(b_runs), min(d_runs)), 4)],
            'port_of': 'decoder/factory.py:52-96 op for op (oracle/restatement.py: torch-CPU interpolate / pad + max_pool2d / topk / '
                       'gather, numpy grouping in a Pool(batch)) + eager fp32 models/networks.py:189-194',
            'sample': f'decoder: one batch of {a.batch} images, 1 warm-up + median of 5 ({dec_s * 1e3:.0f} ms/batch); backbone: '
                      f'1 image, 1 warm-up + median of {bb_runs} ({bb_s:.2f} s/img); {cores} threads',
            'decoder_ms_per_img': round(dec_s / a.batch * 1e3, 2), 'backbone_s_per_img': round(bb_s, 3),
            'c_oracle_decoder_ms_per_img_1thread': round(c_ms, 2)}


if __name__ == '__main__':
    main()
