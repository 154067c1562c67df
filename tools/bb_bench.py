#!/usr/bin/env python3
"""Backbone-only timing under different MIOpen / layout settings (tuning aid)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--benchmark', type=int, default=0)
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--iters', type=int, default=10)
a = ap.parse_args()
torch.backends.cudnn.benchmark = bool(a.benchmark)
from offsetguided_amd import models
import bench
p = argparse.ArgumentParser(); models.net_cli(p)
m, _ = models.model_factory(p.parse_args(['--no-pretrain']))
bench.bench_init(m, 1234)
t0 = time.time()
eng = models.InferenceEngine(m, a.batch, 640, 640, device='cuda:0')
torch.cuda.synchronize(); t_build = time.time() - t0
x = torch.randn(a.batch, 3, 640, 640, device='cuda:0')
for _ in range(3): eng.forward_raw(x)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(a.iters): eng.forward_raw(x)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / a.iters
print(f'benchmark={a.benchmark} MIOPEN_FIND_MODE={os.environ.get("MIOPEN_FIND_MODE")} batch={a.batch}: {ms:.2f} ms/fwd  {a.batch*732.78e9/ms/1e9:.0f} TFLOP/s  (build {t_build:.1f}s)')
