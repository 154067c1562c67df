#!/usr/bin/env python3
"""Gate A of the Winograd F(2x2, 3x3) attempt, CPU only, no GPU minutes.

Emulates the 16-bit engine on the CPU -- every layer: fp32 accumulation over 16-bit operands, ONE rounding of the epilogue's result to
the 16-bit activation type, exactly the arithmetic of the HIP kernels -- once with direct 3x3 convolutions and once with the stride-1
3x3 layers as Winograd F(2x2, 3x3): input transform B^T d B and weight transform G g G^T computed in fp32 and ROUNDED to the 16-bit
type (they are the MFMA operands), 16 position products accumulated in fp32, output transform A^T M A in fp32.  Head outputs against
the eager fp32 module, relative to the head's maximum: the engine gates of tests/test_gpu_backbone.py::ENGINE_GATES
(<= 5e-3 fp16, <= 3e-2 bf16) must hold, unchanged.  (models/hourglass_104.py:16-30,50-79 is what the layers compute.)

  python tools/experiments/wino_gate_a.py [--size 128] [--weights key|bench] [--levels all|top2|top1]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from offsetguided_amd import models  # noqa: E402
from offsetguided_amd.models import engine as E  # noqa: E402

BT = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
AT = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])
GATES = {torch.float16: 5e-3, torch.bfloat16: 3e-2}

MODE = {'wino_min_hw': None, 'count': 0, 'direct': 0}


def wino_conv(x32, w32, lp):
    """F(2x2,3x3) of x32 (N,C,H,W: values already representable in lp) with fp32 folded weights w32 (O,C,3,3); operands rounded to lp."""
    n, c, h, w = x32.shape
    hp, wp = h + (h & 1), w + (w & 1)
    xp = F.pad(x32, (1, 1 + wp - w, 1, 1 + hp - h))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                          # (N, C, Th, Tw, 4, 4)
    v = torch.einsum('ai,nctuij,bj->nctuab', BT, d, BT).to(lp).float()
    u = torch.einsum('ai,ocij,bj->ocab', G, w32, G).to(lp).float()
    m = torch.einsum('nctuab,ocab->notuab', v, u)
    y = torch.einsum('pa,notuab,qb->notupq', AT, m, AT)             # (N, O, Th, Tw, 2, 2)
    y = y.permute(0, 1, 2, 4, 3, 5).reshape(n, -1, hp, wp)
    return y[:, :, :h, :w]


def install(lp):
    orig_init = E._Conv.__init__

    def init(self, conv, bn, relu, dtype, fused):
        orig_init(self, conv, bn, relu, dtype, fused)
        self.w32, _ = E._fold(conv, bn)

    def call(self, x, skip=None):
        x32 = x.float()
        k, st = self.w.shape[2], self.stride[0]
        if k == 3 and st == 1 and MODE['wino_min_hw'] is not None and x.shape[2] >= MODE['wino_min_hw']:
            y = wino_conv(x32, self.w32, lp)
            MODE['count'] += 1
        else:
            y = F.conv2d(x32, self.w.float(), None, self.stride, self.pad)
            MODE['direct'] += (k == 3 and st == 1)
        y = y + self.b32.view(1, -1, 1, 1)
        if skip is not None:
            y = y + skip.float()
        if self.relu:
            y = F.relu(y)
        return y.to(lp)

    def raw(self, x):                       # projections whose bias / activation ride on another layer's epilogue: fp32 until there
        return F.conv2d(x.float(), self.w.float(), None, self.stride, self.pad)

    E._Conv.__init__, E._Conv.__call__, E._Conv.raw = init, call, raw


def head_errors(ref, out):
    errs = []
    for h in (0, 1):
        r, o = ref[h][0][-1].float(), out[h][0][-1].float()
        errs.append(((o - r).abs().max() / r.abs().max()).item())
    return errs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=128)
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--weights', default='key,bench')
    a = ap.parse_args()
    torch.manual_seed(0)
    p = argparse.ArgumentParser()
    models.net_cli(p)
    for weights in a.weights.split(','):
        model, _ = models.model_factory(p.parse_args(['--no-pretrain']))
        if weights == 'key':
            from offsetguided_amd.models.seeding import key_seeded_state
            model.load_state_dict(key_seeded_state(model.state_dict()))
        else:
            import bench
            bench.bench_init(model, 1234)
            for head in model.headnets:
                for m in head.modules():
                    if isinstance(m, torch.nn.Conv2d):
                        m.weight.data.mul_(1e4)
        model = model.eval()
        x = torch.randn(a.batch, 3, a.size, a.size, generator=torch.Generator().manual_seed(0))
        with torch.no_grad():
            ref = model(x)
        top = a.size // 4
        for lp in (torch.float16, torch.bfloat16):
            install(lp)
            for name, min_hw in (('direct', None), ('wino: every stride-1 3x3', 1), (f'wino: levels >= {top // 2} (160 + 80 at 640)', top // 2),
                                 (f'wino: level {top} only (160 at 640)', top)):
                MODE.update(wino_min_hw=min_hw, count=0, direct=0)
                E._layer_cache.clear()
                with torch.no_grad():
                    eng = models.InferenceEngine(model, a.batch, a.size, a.size, dtype=lp, device='cpu', use_graph=False)
                    out = eng(x)
                e = head_errors(ref, out)
                ok = max(e) <= GATES[lp]
                print(f'{weights:5s} {a.batch}x{a.size}x{a.size} {str(lp)[6:]:8s} {name:42s} wino layers {MODE["count"]:3d} direct {MODE["direct"]:3d}  '
                      f'hm {e[0]:.2e} off {e[1]:.2e}  gate {GATES[lp]:.0e}: {"PASS" if ok else "FAIL"}', flush=True)


if __name__ == '__main__':
    main()
