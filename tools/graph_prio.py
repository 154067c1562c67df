#!/usr/bin/env python3
"""Experiment: kernel-node priorities inside the captured forward.  The engine's HIP graph is kept un-instantiated
(torch.cuda.CUDAGraph(keep_graph=True)), every kernel node with fewer than --below workgroups (the latency-bound kernels of the trunk:
20x20 / 10x10 / 5x5 levels, projections) gets hipKernelNodeAttributePriority = --prio through hipGraphKernelNodeSetAttribute, then the
graph is instantiated; replay time against engines captured the usual way, interleaved in one process.
usage: graph_prio.py [--below 512] [--prio -1] [--engines 6]

RESULT (round 5, ROCm 7.2): hipGraphKernelNodeSetAttribute(node, hipKernelNodeAttributePriority, value) returns hipErrorInvalidValue for
every value in and around hipDeviceGetStreamPriorityRange() = (least 1, greatest -1) -- 0 included -- while the matching Get succeeds
(priority 0): this runtime has no kernel-node priorities, the tool stops at the first refusal.  Capture-stream priorities (trunk on a
priority -1 stream, or the side streams on priority 1 = torch clamps to 0) change nothing either: 5.52-5.60 ms per forward in every arm
of tools/capture_variance.py (profiles/r05_graph_prio.log)."""
import argparse
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from offsetguided_amd import models  # noqa: E402
from offsetguided_amd.models import engine as eng_mod  # noqa: E402


class Dim3(C.Structure):
    _fields_ = [('x', C.c_uint), ('y', C.c_uint), ('z', C.c_uint)]


class KernelNodeParams(C.Structure):
    _fields_ = [('blockDim', Dim3), ('extra', C.c_void_p), ('func', C.c_void_p), ('gridDim', Dim3), ('kernelParams', C.c_void_p),
                ('sharedMemBytes', C.c_uint)]


class AttrValue(C.Union):
    _fields_ = [('pad', C.c_char * 64), ('priority', C.c_int)]


def set_priorities(graph_ptr, below, prio):
    hip = C.CDLL('libamdhip64.so')
    n = C.c_size_t(0)
    assert hip.hipGraphGetNodes(C.c_void_p(graph_ptr), None, C.byref(n)) == 0
    nodes = (C.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(C.c_void_p(graph_ptr), nodes, C.byref(n)) == 0
    kernels = raised = 0
    errs = set()
    for node in nodes:
        t = C.c_int(-1)
        assert hip.hipGraphNodeGetType(C.c_void_p(node), C.byref(t)) == 0
        if t.value != 0:        # hipGraphNodeTypeKernel
            continue
        kernels += 1
        p = KernelNodeParams()
        assert hip.hipGraphKernelNodeGetParams(C.c_void_p(node), C.byref(p)) == 0
        blocks = p.gridDim.x * p.gridDim.y * p.gridDim.z
        if blocks < below:
            v = AttrValue()
            v.priority = prio
            rc = hip.hipGraphKernelNodeSetAttribute(C.c_void_p(node), 8, C.byref(v))      # hipKernelNodeAttributePriority
            if rc:
                errs.add(rc)
            else:
                raised += 1
    print(f'graph: {n.value} nodes, {kernels} kernels, priority {prio} set on {raised} (< {below} workgroups); error codes {sorted(errs)}')
    hip.hipGetLastError()          # a refusal must not surface in torch's next call
    if errs:
        sys.exit('hipGraphKernelNodeSetAttribute refused the priority attribute: nothing to measure')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--below', type=int, default=512)
    ap.add_argument('--prio', type=int, default=-1)
    ap.add_argument('--engines', type=int, default=6)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    p = argparse.ArgumentParser()
    models.net_cli(p)
    model, _ = models.model_factory(p.parse_args(['--no-pretrain']))
    bench.bench_init(model, 1234)
    x = torch.randn(8, 3, 640, 640, device=dev)
    plain_capture = eng_mod.InferenceEngine._capture

    def prio_capture(self):
        real = torch.cuda.CUDAGraph
        torch.cuda.CUDAGraph = lambda: real(keep_graph=True)
        try:
            plain_capture(self)
        finally:
            torch.cuda.CUDAGraph = real
        set_priorities(self._graph.raw_cuda_graph(), a.below, a.prio)
        self._graph.instantiate()

    engines, kinds = [], []
    for i in range(a.engines):
        eng_mod.InferenceEngine._capture = prio_capture if i % 2 else plain_capture
        engines.append(models.InferenceEngine(model, 8, 640, 640, dtype=torch.float16, device=dev))
        kinds.append('prio' if i % 2 else 'plain')
    eng_mod.InferenceEngine._capture = plain_capture
    ref = [o.clone() for o in engines[0].forward_raw(x)]
    for e in engines[1:]:
        assert all(torch.equal(r, o) for r, o in zip(ref, e.forward_raw(x)))
    print('engines:', ' '.join(kinds))
    for rnd in range(3):
        line = []
        for e in engines:
            for _ in range(3):
                e.forward_raw(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                e.forward_raw(x)
            torch.cuda.synchronize()
            line.append((time.perf_counter() - t0) / 20 * 1e3)
        print('round', rnd, ' '.join(f'{v:.3f}' for v in line), flush=True)


if __name__ == '__main__':
    main()
