"""Backbone side on the GPU: fused epilogue kernels vs torch ops, and the bf16 HIP-graph engine vs the
eager fp32 module (tolerance-based: bf16 activations through ~100 layers)."""
import argparse

import pytest
import torch

from offsetguided_amd import _lib, models

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no HIP device is visible")
    return torch.device("cuda:0")


def _nhwc(*shape, dev):
    return torch.randn(*shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("relu", [0, 1])
@pytest.mark.parametrize("with_skip", [False, True])
def test_bias_act_matches_torch(dev, relu, with_skip):
    lib = _lib.load()
    x = _nhwc(2, 256, 20, 12, dev=dev)
    skip = _nhwc(2, 256, 20, 12, dev=dev) if with_skip else None
    bias = torch.randn(256, device=dev)
    ref = x.float() + bias.view(1, -1, 1, 1) + (skip.float() if with_skip else 0)
    ref = (torch.relu(ref) if relu else ref).to(torch.bfloat16)              # single rounding, like the kernel
    y = x.clone(memory_format=torch.preserve_format)
    _lib.check(lib.og_bias_act_bf16(_lib.ptr(y), _lib.ptr(bias), _lib.ptr(skip) if with_skip else None,
                                    2 * 20 * 12, 256, relu, _lib.stream_ptr(dev)), lib)
    assert torch.equal(y, ref)


def test_stage_timer_uses_fence_free_events(dev):
    """_lib.stage_timer / TimingEvent (hipEventDisableSystemFence events on the launch stream) give sane durations."""
    x = torch.zeros(1 << 24, device=dev)
    x.add_(0.0)                                               # first launch loads the code object on the host
    torch.cuda.synchronize(dev)
    _lib.profile_start()
    for _ in range(3):
        with _lib.stage_timer('fill', dev):
            x.add_(1.0)
    out = _lib.profile_stop()
    assert len(out['fill']) == 3 and all(1.0 < t < 1e6 for t in out['fill']), out   # microseconds
    assert float(x[0]) == 3.0


def test_upsample2_add_matches_torch(dev):
    lib = _lib.load()
    up, low = _nhwc(2, 384, 10, 20, dev=dev), _nhwc(2, 384, 5, 10, dev=dev)
    ref = (up.float() + torch.nn.functional.interpolate(low.float(), scale_factor=2, mode='nearest')).to(torch.bfloat16)
    _lib.check(lib.og_upsample2_add_bf16(_lib.ptr(up), _lib.ptr(low), 2, 10, 20, 384, _lib.stream_ptr(dev)), lib)
    assert torch.equal(up, ref)


@pytest.mark.parametrize("shape", [(8, 5, 5, 512, 512), (8, 10, 10, 384, 384), (2, 20, 20, 384, 384), (3, 7, 9, 64, 128),
                                   (1, 1, 1, 128, 64), (8, 20, 20, 384, 384), (2, 33, 31, 128, 256)])
def test_conv3x3_matches_torch(dev, shape):
    """og_conv3x3_bf16 (split-K MFMA kernel, last-arriver reduction inside the launch, fused epilogue) vs an fp32 torch
    convolution of the same bf16 operands: differences are the final bf16 rounding (2^-9 relative) plus fp32 summation order."""
    import torch.nn.functional as F
    n, h, w, cin, cout = shape
    lib = _lib.load()
    g = torch.Generator(device='cpu').manual_seed(h * 1000 + cin)
    cl = torch.channels_last
    x = torch.randn(n, cin, h, w, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=cl)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (9 * cin)) ** 0.5).to(dev).to(torch.bfloat16).contiguous(memory_format=cl)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    skip = torch.randn(n, cout, h, w, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=cl)
    ws = torch.zeros(lib.og_conv3x3_workspace_bytes(n * h * w, cin, cout), dtype=torch.uint8, device=dev)
    for use_skip, relu in ((True, 1), (False, 0)):
        ref = F.conv2d(x.float(), wt.float(), bias, 1, 1)
        if use_skip:
            ref = ref + skip.float()
        if relu:
            ref = F.relu(ref)
        out = torch.full_like(skip, float('nan'))
        for _ in range(2):                                   # second call: tickets must be back at zero
            _lib.check(lib.og_conv3x3_bf16(_lib.ptr(x), _lib.ptr(wt), _lib.ptr(bias), _lib.ptr(skip) if use_skip else None,
                                           _lib.ptr(out), n, h, w, cin, cout, relu, _lib.ptr(ws), ws.numel(),
                                           _lib.stream_ptr(dev)), lib)
        err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
        assert err <= 6e-3, f'relative error {err}'
    assert ws[:256].count_nonzero().item() == 0              # the zero page is never written


@pytest.mark.parametrize("shape", [(2, 20, 20, 384, 384, 3, 2), (2, 20, 20, 384, 384, 1, 2), (8, 5, 5, 512, 384, 1, 1),
                                   (8, 10, 10, 384, 512, 3, 2), (8, 10, 10, 384, 512, 1, 2), (3, 7, 9, 64, 128, 3, 2),
                                   (3, 7, 9, 128, 64, 1, 2), (1, 1, 1, 128, 64, 1, 1), (2, 64, 64, 128, 256, 3, 2),
                                   (2, 64, 64, 128, 256, 1, 2), (2, 80, 80, 256, 256, 1, 1), (2, 33, 31, 128, 256, 3, 1),
                                   (1, 96, 96, 256, 384, 3, 2), (4, 160, 160, 256, 64, 1, 1)])
def test_conv2d_matches_torch(dev, shape):
    """og_conv2d_bf16 (1x1 / 3x3, stride 1 / 2, fused epilogue) vs an fp32 torch convolution of the same bf16 operands."""
    import torch.nn.functional as F
    n, h, w, cin, cout, k, st = shape
    lib = _lib.load()
    g = torch.Generator(device='cpu').manual_seed(h * 1000 + cin + k * 7 + st)
    cl = torch.channels_last
    x = torch.randn(n, cin, h, w, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=cl)
    wt = (torch.randn(cout, cin, k, k, generator=g) * (1.0 / (k * k * cin)) ** 0.5).to(dev).to(torch.bfloat16).contiguous(memory_format=cl)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    base = F.conv2d(x.float(), wt.float(), bias, st, k // 2)
    ho, wo = base.shape[2:]
    skip = torch.randn(n, cout, ho, wo, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=cl)
    need = lib.og_conv2d_workspace_bytes(n, h, w, cin, cout, k, st)
    assert need > 0
    ws = torch.zeros(need, dtype=torch.uint8, device=dev)
    for use_skip, relu in ((True, 1), (False, 0)):
        ref = base + skip.float() if use_skip else base
        if relu:
            ref = F.relu(ref)
        out = torch.full_like(skip, float('nan'))
        for _ in range(2):
            _lib.check(lib.og_conv2d_bf16(_lib.ptr(x), _lib.ptr(wt), _lib.ptr(bias), _lib.ptr(skip) if use_skip else None,
                                          _lib.ptr(out), n, h, w, cin, cout, k, st, relu, _lib.ptr(ws), ws.numel(),
                                          _lib.stream_ptr(dev)), lib)
        err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
        assert err <= 6e-3, f'relative error {err}'
    assert ws[:256].count_nonzero().item() == 0


@pytest.mark.parametrize("shape", [(8, 5, 5, 384, 384, 512, 1), (8, 5, 5, 512, 512, 384, 2), (8, 10, 10, 384, 384, 384, 2),
                                   (2, 20, 20, 384, 384, 384, 2), (3, 7, 9, 64, 128, 192, 1), (2, 9, 6, 128, 64, 64, 2)])
def test_conv2d_proj_matches_torch(dev, shape):
    """og_conv2d_proj_bf16 = relu(conv3x3(y) + conv1x1(x, stride) + bias), the tail of a projection residual
    (models/hourglass_104.py:70-79), vs fp32 torch convolutions of the same bf16 operands."""
    import torch.nn.functional as F
    n, h, w, cin, cout, cin2, st2 = shape
    lib = _lib.load()
    g = torch.Generator(device='cpu').manual_seed(h * 100 + cin2 + st2)
    cl = torch.channels_last
    y = torch.randn(n, cin, h, w, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=cl)
    h2, w2 = (h, w) if st2 == 1 else (2 * h - 1, 2 * w)           # odd and even input sizes both give h x w at stride 2
    x = torch.randn(n, cin2, h2, w2, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=cl)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (9 * cin)) ** 0.5).to(dev).to(torch.bfloat16).contiguous(memory_format=cl)
    wp = (torch.randn(cout, cin2, 1, 1, generator=g) * (1.0 / cin2) ** 0.5).to(dev).to(torch.bfloat16).contiguous(memory_format=cl)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    ref = F.relu(F.conv2d(y.float(), wt.float(), bias, 1, 1) + F.conv2d(x.float(), wp.float(), None, st2, 0))
    assert tuple(ref.shape[2:]) == (h, w)
    w_cat = torch.cat([wt.permute(0, 2, 3, 1).reshape(cout, -1), wp.reshape(cout, -1)], 1).contiguous()
    need = lib.og_conv2d_proj_workspace_bytes(n, h, w, cin, cout, 3, 1, cin2)
    assert need > 0
    ws = torch.zeros(need, dtype=torch.uint8, device=dev)
    out = torch.full((n, cout, h, w), float('nan'), dtype=torch.bfloat16, device=dev).contiguous(memory_format=cl)
    for _ in range(2):
        _lib.check(lib.og_conv2d_proj_bf16(_lib.ptr(y), _lib.ptr(w_cat), _lib.ptr(bias), _lib.ptr(x), _lib.ptr(out), n, h, w,
                                           cin, cout, 3, 1, h2, w2, cin2, st2, 1, _lib.ptr(ws), ws.numel(),
                                           _lib.stream_ptr(dev)), lib)
    err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
    assert err <= 6e-3, f'relative error {err}'
    rc = lib.og_conv2d_proj_bf16(_lib.ptr(y), _lib.ptr(w_cat), _lib.ptr(bias), _lib.ptr(x), _lib.ptr(out), n, h, w, cin, cout,
                                 3, 1, h2 + 4, w2, cin2, st2, 1, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    assert rc == _lib.OG_EINVAL                                   # projection input that does not map onto the output


def test_conv2d_rejects_bad_arguments(dev):
    lib = _lib.load()
    x = _nhwc(1, 64, 8, 8, dev=dev)
    wt = _nhwc(64, 64, 3, 3, dev=dev)
    b = torch.zeros(64, device=dev)
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
    out = torch.empty_like(x)
    for k, st in ((5, 1), (3, 3), (2, 1)):
        rc = lib.og_conv2d_bf16(_lib.ptr(x), _lib.ptr(wt), _lib.ptr(b), None, _lib.ptr(out), 1, 8, 8, 64, 64, k, st, 0,
                                _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        assert rc == _lib.OG_EUNSUPPORTED
        assert lib.og_conv2d_workspace_bytes(1, 8, 8, 64, 64, k, st) == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(1, 16, 16, 64, 128), (2, 32, 32, 128, 128), (1, 48, 32, 192, 256), (2, 80, 80, 256, 256),
                                   (1, 4, 40, 64, 128), (2, 40, 40, 384, 384), (1, 12, 40, 128, 256), (3, 16, 32, 64, 384),
                                   (8, 40, 40, 384, 384), (8, 20, 20, 384, 384), (2, 20, 20, 256, 128), (8, 40, 40, 256, 256),
                                   (1, 8, 20, 128, 256)])
def test_conv3x3_tiled_kernel_matches_torch(dev, shape, dtype):
    """og_conv3x3_tiled_* (csrc/conv3x3_tiled.inc: two 4-wave workgroups per CU, pre-tiled weights, 32-channel K steps) vs an
    fp32 torch convolution of the same 16-bit operands, with and without the residual / ReLU epilogue; bit-identical
    results when launched again (no dependence on timing), and the packing is a pure permutation of the weights."""
    import torch.nn.functional as F
    n, h, w, cin, cout = shape
    lib = _lib.load()
    assert lib.og_conv3x3_tiled_supported(n, h, w, cin, cout) > 0
    g = torch.Generator(device='cpu').manual_seed(h * 1000 + cin)
    cl = torch.channels_last
    x = torch.randn(n, cin, h, w, generator=g).to(dev).to(dtype).contiguous(memory_format=cl)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (9 * cin)) ** 0.5).to(dev).to(dtype).contiguous(memory_format=cl)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    skip = torch.randn(n, cout, h, w, generator=g).to(dev).to(dtype).contiguous(memory_format=cl)
    packed = torch.empty(wt.numel(), dtype=dtype, device=dev)
    _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wt), cin, cout, 0, _lib.ptr(packed), _lib.stream_ptr(dev)), lib)
    assert torch.equal(packed.view(torch.int16).sort().values, wt.reshape(-1).view(torch.int16).sort().values)
    fn = _lib.lp(lib, 'og_conv3x3_tiled', dtype)
    tol = 6e-3 if dtype == torch.bfloat16 else 1e-3
    need = lib.og_conv3x3_tiled_workspace_bytes(n, h, w, cin, cout)       # > 0: the launch is split along K (few output tiles)
    ws = torch.zeros(need, dtype=torch.uint8, device=dev) if need else None
    for use_skip, relu in ((True, 1), (False, 0)):
        ref = F.conv2d(x.float(), wt.float(), bias, 1, 1)
        ref = F.relu(ref + skip.float()) if use_skip else ref
        outs = []
        for _ in range(2):
            out = torch.full_like(skip, float('nan'))
            _lib.check(fn(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(skip) if use_skip else None, _lib.ptr(out),
                          n, h, w, cin, cout, relu, _lib.ptr(ws) if need else None, need, _lib.stream_ptr(dev)), lib)
            outs.append(out)
        err = ((outs[0].float() - ref).abs().max() / ref.abs().max()).item()
        assert err <= tol, f'relative error {err}'
        assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(1, 16, 32, 64, 128), (2, 32, 32, 128, 256), (1, 48, 64, 192, 128), (2, 160, 160, 256, 256),
                                   (1, 16, 96, 64, 384), (1, 320, 320, 128, 256), (1, 4, 80, 64, 128), (2, 80, 80, 256, 384),
                                   (3, 12, 80, 128, 128)])
def test_conv3x3s2_tiled_kernel_matches_torch(dev, shape, dtype):
    """og_conv3x3s2_tiled_* (stride 2, pad 1: the four input-parity phase images of a 16 x 8 output tile gathered by LDS-DMA,
    weights pre-tiled in the order the phase groups consume them) vs an fp32 torch convolution of the same 16-bit operands;
    borders (top / left zero padding lives in the odd phases), residual + ReLU epilogue, repeat launches bit-identical."""
    import torch.nn.functional as F
    n, h, w, cin, cout = shape
    lib = _lib.load()
    assert lib.og_conv3x3s2_tiled_supported(n, h, w, cin, cout) == (2 if w == 80 else 1)
    g = torch.Generator(device='cpu').manual_seed(h * 1000 + cin + 7)
    cl = torch.channels_last
    x = torch.randn(n, cin, h, w, generator=g).to(dev).to(dtype).contiguous(memory_format=cl)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (9 * cin)) ** 0.5).to(dev).to(dtype).contiguous(memory_format=cl)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    skip = torch.randn(n, cout, h // 2, w // 2, generator=g).to(dev).to(dtype).contiguous(memory_format=cl)
    packed = torch.empty(wt.numel(), dtype=dtype, device=dev)
    _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wt), cin, cout, 1, _lib.ptr(packed), _lib.stream_ptr(dev)), lib)
    assert torch.equal(packed.view(torch.int16).sort().values, wt.reshape(-1).view(torch.int16).sort().values)
    fn = _lib.lp(lib, 'og_conv3x3s2_tiled', dtype)
    tol = 6e-3 if dtype == torch.bfloat16 else 1e-3
    for use_skip, relu in ((False, 1), (True, 1), (False, 0)):
        ref = F.conv2d(x.float(), wt.float(), bias, 2, 1)
        ref = ref + skip.float() if use_skip else ref
        ref = F.relu(ref) if relu else ref
        outs = []
        for _ in range(2):
            out = torch.full_like(skip, float('nan'))
            _lib.check(fn(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(skip) if use_skip else None, _lib.ptr(out),
                          n, h, w, cin, cout, relu, _lib.stream_ptr(dev)), lib)
            outs.append(out)
        err = ((outs[0].float() - ref).abs().max() / ref.abs().max()).item()
        assert err <= tol, f'relative error {err}'
        assert torch.equal(outs[0], outs[1])
    assert lib.og_conv3x3s2_tiled_supported(1, 30, 32, 64, 128) == 0 and lib.og_conv3x3s2_tiled_supported(1, 16, 32, 64, 64) == 0
    rc = fn(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bias), None, _lib.ptr(outs[0]), 1, 30, 32, 64, 128, 1, _lib.stream_ptr(dev))
    assert rc == _lib.OG_EUNSUPPORTED


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 40, 40, 256, 256, 1, True), (1, 33, 21, 128, 128, 1, False), (2, 32, 48, 128, 256, 2, False),
                                  (1, 160, 160, 256, 256, 1, True), (3, 20, 24, 384, 256, 1, False), (1, 64, 64, 64, 384, 2, False)])
def test_conv1x1_tiled_matches_torch(dev, case, dtype):
    """og_conv1x1_tiled_* vs fp32 torch: one input (stride 1 / 2: the projection `skip` of the residuals, raw and with bias +
    residual + ReLU) and two inputs concatenated along K (the inters_ / cnvs_ junction); tile tails (M not a multiple of 256)."""
    import torch.nn.functional as F
    n, h, w, cin, cout, st, two = case
    lib = _lib.load()
    g = torch.Generator(device='cpu').manual_seed(h * 100 + cin + st)
    cl = torch.channels_last
    mk = lambda *sh: torch.randn(*sh, generator=g).to(dev).to(dtype)   # noqa: E731
    x1 = mk(n, cin, h, w).contiguous(memory_format=cl)
    x2 = mk(n, cin, h, w).contiguous(memory_format=cl) if two else None
    w1 = (mk(cout, cin, 1, 1).float() * (1.0 / cin) ** 0.5).to(dtype)
    w2 = (mk(cout, cin, 1, 1).float() * (1.0 / cin) ** 0.5).to(dtype) if two else None
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
    skip = mk(n, cout, ho, wo).contiguous(memory_format=cl)
    wcat = (torch.cat([w1.reshape(cout, -1), w2.reshape(cout, -1)], 1) if two else w1.reshape(cout, -1)).contiguous()
    packed = torch.empty(wcat.numel(), dtype=dtype, device=dev)
    _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wcat), wcat.shape[1], cout, 2, _lib.ptr(packed), _lib.stream_ptr(dev)), lib)
    fn = _lib.lp(lib, 'og_conv1x1_tiled', dtype)
    tol = 6e-3 if dtype == torch.bfloat16 else 1e-3
    base = F.conv2d(x1.float(), w1.float(), None, st) + (F.conv2d(x2.float(), w2.float(), None, st) if two else 0)
    for use_bias, use_skip, relu in ((False, False, 0), (True, True, 1), (True, False, 1)):
        ref = base + (bias.view(1, -1, 1, 1) if use_bias else 0) + (skip.float() if use_skip else 0)
        ref = F.relu(ref) if relu else ref
        outs = []
        for _ in range(2):
            out = torch.full_like(skip, float('nan'))
            _lib.check(fn(_lib.ptr(x1), cin, h, w, st, _lib.ptr(x2) if two else None, cin if two else 0, h, w, st, _lib.ptr(packed),
                          _lib.ptr(bias) if use_bias else None, _lib.ptr(skip) if use_skip else None, _lib.ptr(out), n, ho, wo, cout,
                          relu, _lib.stream_ptr(dev)), lib)
            outs.append(out)
        err = ((outs[0].float() - ref).abs().max() / ref.abs().max()).item()
        assert err <= tol, f'relative error {err}'
        assert torch.equal(outs[0], outs[1])
    rc = fn(_lib.ptr(x1), 96, h, w, st, None, 0, h, w, st, _lib.ptr(packed), None, None, _lib.ptr(outs[0]), n, ho, wo, cout, 0,
            _lib.stream_ptr(dev))
    assert rc == _lib.OG_EUNSUPPORTED


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", [(2, 40, 40, 256, (17, 38)), (1, 33, 21, 128, (17, 38, 17)), (1, 160, 160, 256, (17, 38)), (2, 16, 16, 64, (17, 38, 17, 2))])
def test_conv1x1_heads_matches_torch(dev, case, dtype):
    """og_conv1x1_heads_*: all heads as one 1x1 convolution, dense fp32 NCHW outputs per head with the bias added in fp32 (no
    16-bit rounding of the result: the error left is the fp32 summation order)."""
    import ctypes as C
    import torch.nn.functional as F
    n, h, w, cin, heads = case
    lib = _lib.load()
    g = torch.Generator(device='cpu').manual_seed(h + cin)
    x = torch.randn(n, cin, h, w, generator=g).to(dev).to(dtype).contiguous(memory_format=torch.channels_last)
    tot = sum(heads)
    cout = (tot + 63) // 64 * 64
    wt = torch.zeros(cout, cin, dtype=dtype, device=dev)
    wt[:tot] = (torch.randn(tot, cin, generator=g) * (1.0 / cin) ** 0.5).to(dev).to(dtype)
    bias = torch.zeros(cout, device=dev)
    bias[:tot] = torch.randn(tot, generator=g).to(dev)
    packed = torch.empty(wt.numel(), dtype=dtype, device=dev)
    _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wt), cin, cout, 3, _lib.ptr(packed), _lib.stream_ptr(dev)), lib)
    outs = [torch.full((n, c, h, w), float('nan'), device=dev) for c in heads]
    chans = (C.c_int * len(heads))(*heads)
    ptrs = (C.c_void_p * len(heads))(*[o.data_ptr() for o in outs])
    _lib.check(_lib.lp(lib, 'og_conv1x1_heads', dtype)(_lib.ptr(x), cin, _lib.ptr(packed), _lib.ptr(bias), n, h, w, cout, len(heads),
                                                     chans, ptrs, _lib.stream_ptr(dev)), lib)
    ref = F.conv2d(x.float(), wt[:tot].float().view(tot, cin, 1, 1), bias[:tot])
    c0 = 0
    for o, c in zip(outs, heads):
        r = ref[:, c0:c0 + c]
        assert ((o - r).abs().max() / r.abs().max()).item() <= 2e-5
        c0 += c


def test_conv3x3_tiled_repeated_full_size(dev):
    """The 160x160 256->256 layer of the network at bs8 (1 600 workgroups, two per CU): 20 launches on rotating inputs, every
    output equal to the first launch of its input (no race between the DMA ring, the barriers and the fragment reads under
    full load), and equal within rounding to the split-K kernel's result for the same layer."""
    lib = _lib.load()
    cl = torch.channels_last
    g = torch.Generator(device='cpu').manual_seed(5)
    xs = [torch.randn(8, 256, 160, 160, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=cl) for _ in range(2)]
    wt = (torch.randn(256, 256, 3, 3, generator=g) * (1.0 / 2304) ** 0.5).to(dev).to(torch.bfloat16).contiguous(memory_format=cl)
    bias = torch.randn(256, generator=g).to(dev) * 0.1
    packed = torch.empty(wt.numel(), dtype=torch.bfloat16, device=dev)
    _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wt), 256, 256, 0, _lib.ptr(packed), _lib.stream_ptr(dev)), lib)
    first = [None, None]
    for it in range(20):
        i = it % 2
        out = torch.empty_like(xs[i])
        _lib.check(lib.og_conv3x3_tiled_bf16(_lib.ptr(xs[i]), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(xs[1 - i]), _lib.ptr(out),
                                             8, 160, 160, 256, 256, 1, None, 0, _lib.stream_ptr(dev)), lib)
        if first[i] is None:
            first[i] = out
        else:
            assert torch.equal(out, first[i]), f'launch {it} differs'
    ws = torch.zeros(lib.og_conv3x3_workspace_bytes(8 * 160 * 160, 256, 256), dtype=torch.uint8, device=dev)
    old = torch.empty_like(xs[0])
    _lib.check(lib.og_conv3x3_bf16(_lib.ptr(xs[0]), _lib.ptr(wt), _lib.ptr(bias), _lib.ptr(xs[1]), _lib.ptr(old), 8, 160, 160, 256,
                                   256, 1, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)), lib)
    d = (old.float() - first[0].float()).abs().max().item()
    assert d <= 2 ** -6 * max(1.0, old.float().abs().max().item()), d     # same operands, different fp32 summation order


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 32, 32, 128, 256), (1, 80, 80, 256, 256), (3, 40, 40, 384, 256), (2, 12, 40, 256, 384),
                                   (8, 20, 20, 384, 384), (2, 16, 48, 64, 128)])
def test_conv3x3_tiled_up2_equals_conv_then_upsample_add(dev, shape, dtype):
    """og_conv3x3_tiled_up2_* (the hourglass merge on the epilogue of the convolution below it, models/hourglass_104.py:170-176)
    == og_conv3x3_tiled_* followed by og_upsample2_add_*, bit for bit -- every tile shape, K-split shapes included, with and
    without the residual, twice."""
    n, h, w, cin, cout = shape
    lib = _lib.load()
    assert lib.og_conv3x3_tiled_supported(n, h, w, cin, cout) > 0
    g = torch.Generator(device='cpu').manual_seed(h * 77 + cout)
    cl = torch.channels_last
    x = torch.randn(n, cin, h, w, generator=g).to(dev).to(dtype).contiguous(memory_format=cl)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (9 * cin)) ** 0.5).to(dev).to(dtype).contiguous(memory_format=cl)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    skip = torch.randn(n, cout, h, w, generator=g).to(dev).to(dtype).contiguous(memory_format=cl)
    up1 = torch.randn(n, cout, 2 * h, 2 * w, generator=g).to(dev).to(dtype).contiguous(memory_format=cl)
    packed = torch.empty(wt.numel(), dtype=dtype, device=dev)
    _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wt), cin, cout, 0, _lib.ptr(packed), _lib.stream_ptr(dev)), lib)
    need = lib.og_conv3x3_tiled_workspace_bytes(n, h, w, cin, cout)
    ws = torch.zeros(need, dtype=torch.uint8, device=dev) if need else None
    wsp, st = (_lib.ptr(ws) if need else None), _lib.stream_ptr(dev)
    for use_skip, relu in ((True, 1), (False, 0)):
        sk = _lib.ptr(skip) if use_skip else None
        low = torch.empty_like(skip)
        _lib.check(_lib.lp(lib, 'og_conv3x3_tiled', dtype)(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bias), sk, _lib.ptr(low), n, h, w,
                                                           cin, cout, relu, wsp, need, st), lib)
        want = up1.clone(memory_format=torch.preserve_format)
        _lib.check(_lib.lp(lib, 'og_upsample2_add', dtype)(_lib.ptr(want), _lib.ptr(low), n, 2 * h, 2 * w, cout, st), lib)
        for _ in range(2):
            got = up1.clone(memory_format=torch.preserve_format)
            _lib.check(_lib.lp(lib, 'og_conv3x3_tiled_up2', dtype)(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bias), sk, _lib.ptr(got), n, h,
                                                                   w, cin, cout, relu, wsp, need, st), lib)
            assert torch.equal(got, want)
        assert not torch.equal(want, up1)


# (N, Hin, Win, Cin, Cout, stride, projection (H2, W2, Cin2, stride2) or None)
BAND_SHAPES = [(8, 5, 5, 512, 512, 1, None), (8, 10, 10, 384, 384, 1, None), (2, 20, 20, 384, 384, 1, None),
               (8, 10, 10, 384, 512, 2, None), (2, 20, 20, 384, 384, 2, None), (8, 5, 5, 512, 512, 1, (10, 10, 384, 2)),
               (8, 5, 5, 384, 384, 1, (5, 5, 512, 1)), (3, 7, 9, 128, 64, 1, None), (1, 4, 4, 64, 64, 1, None),
               (2, 6, 12, 256, 128, 2, None), (1, 3, 40, 64, 48, 1, None), (2, 9, 11, 96, 80, 1, (9, 11, 64, 1))]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", BAND_SHAPES)
def test_conv_band_matches_torch(dev, shape, dtype):
    """og_conv_band_* (band-resident: K split over the waves of a workgroup, partial tiles summed in LDS; csrc/conv_band.hip) vs an
    fp32 torch convolution of the same 16-bit operands (+ the residual's 1x1 projection, models/hourglass_104.py:63-79): with and
    without the residual operand / ReLU, stride 1 and 2, one band and several, channel counts that leave lanes and waves idle."""
    import torch.nn.functional as F
    n, h, w, cin, cout, st, proj = shape
    lib = _lib.load()
    g = torch.Generator(device='cpu').manual_seed(h * 1000 + w * 10 + cin + cout + st)
    cl = torch.channels_last
    x = torch.randn(n, cin, h, w, generator=g).to(dev).to(dtype).contiguous(memory_format=cl)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (9 * cin)) ** 0.5).to(dev).to(dtype).contiguous(memory_format=cl)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    base = F.conv2d(x.float(), wt.float(), bias, st, 1)
    ho, wo = base.shape[2:]
    h2 = w2 = c2 = 0
    st2, x2, wp = 1, None, None
    if proj is not None:
        h2, w2, c2, st2 = proj
        x2 = torch.randn(n, c2, h2, w2, generator=g).to(dev).to(dtype).contiguous(memory_format=cl)
        wp = (torch.randn(cout, c2, generator=g) * (1.0 / c2) ** 0.5).to(dev).to(dtype).contiguous()
        base = base + F.conv2d(x2.float(), wp.float().view(cout, c2, 1, 1), None, st2, 0)
    assert lib.og_conv_band_supported(n, h, w, cin, cout, st, h2, w2, c2, st2) > 0
    packed = torch.empty(cout * (9 * cin + c2), dtype=dtype, device=dev)
    _lib.check(lib.og_conv_band_pack_w16(_lib.ptr(wt), _lib.ptr(wp) if wp is not None else None, cin, cout, c2, _lib.ptr(packed),
                                         _lib.stream_ptr(dev)), lib)
    skip = torch.randn(n, cout, ho, wo, generator=g).to(dev).to(dtype).contiguous(memory_format=cl)
    fn = _lib.lp(lib, 'og_conv_band', dtype)
    tol = 6e-3 if dtype == torch.bfloat16 else 1e-3
    for use_skip, relu in ((True, 1), (False, 0), (False, 1)):
        ref = base + skip.float() if use_skip else base
        if relu:
            ref = F.relu(ref)
        out = torch.full_like(skip, float('nan'))
        for _ in range(2):
            _lib.check(fn(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(skip) if use_skip else None,
                          _lib.ptr(x2) if x2 is not None else None, _lib.ptr(out), n, h, w, cin, cout, st, relu, h2, w2, c2, st2,
                          _lib.stream_ptr(dev)), lib)
        assert not torch.isnan(out.float()).any()
        err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
        assert err <= tol, f'{shape} skip={use_skip} relu={relu}: relative error {err}'


def test_conv_band_rejects_bad_arguments(dev):
    lib = _lib.load()
    assert lib.og_conv_band_supported(8, 5, 5, 48, 64, 1, 0, 0, 0, 0) == 0          # Cin % 32
    assert lib.og_conv_band_supported(8, 5, 5, 1024, 64, 1, 0, 0, 0, 0) == 0        # more chunks than waves
    assert lib.og_conv_band_supported(8, 5, 200, 64, 64, 1, 0, 0, 0, 0) == 0        # a row wider than a band
    assert lib.og_conv_band_supported(8, 5, 5, 64, 64, 3, 0, 0, 0, 0) == 0          # stride
    assert lib.og_conv_band_supported(8, 5, 5, 64, 64, 1, 7, 7, 64, 1) == 0         # projection grid does not match
    x = torch.zeros(1, 64, 4, 4, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.zeros(64, device=dev)
    assert lib.og_conv_band_bf16(_lib.ptr(x), _lib.ptr(x), _lib.ptr(b), None, None, _lib.ptr(x), 1, 4, 4, 48, 64, 1, 1, 0, 0, 0, 1,
                                 _lib.stream_ptr(dev)) == -4
    assert lib.og_conv_band_bf16(None, _lib.ptr(x), _lib.ptr(b), None, None, _lib.ptr(x), 1, 4, 4, 64, 64, 1, 1, 0, 0, 0, 1,
                                 _lib.stream_ptr(dev)) == -1


@pytest.mark.parametrize("shape", [(1, 32, 32), (2, 64, 96), (1, 128, 64)])
def test_stem7x7_matches_torch(dev, shape):
    """og_stem7x7_bf16 (fp32 NCHW in, conv 7x7 s2 p3 + bias + ReLU, bf16 NHWC out) vs an fp32 torch convolution of the
    bf16-rounded operands."""
    import torch.nn.functional as F
    n, h, w = shape
    lib = _lib.load()
    g = torch.Generator(device='cpu').manual_seed(h + w)
    x = torch.randn(n, 3, h, w, generator=g).to(dev)
    wt = (torch.randn(128, 3, 7, 7, generator=g) * (1.0 / 147) ** 0.5).to(dev)
    bias = (torch.randn(128, generator=g) * 0.1).to(dev)
    packed = torch.zeros((128, 7, 8, 4), device=dev)
    packed[:, :, :7, :3] = wt.permute(0, 2, 3, 1)
    packed = packed.to(torch.bfloat16).contiguous()
    out = torch.full((n, 128, h // 2, w // 2), float('nan'), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    _lib.check(lib.og_stem7x7_bf16(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(out), n, h, w, 1, _lib.stream_ptr(dev)), lib)
    ref = F.relu(F.conv2d(x.to(torch.bfloat16).float(), wt.to(torch.bfloat16).float(), bias, 2, 3))
    err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
    assert err <= 6e-3, f'relative error {err}'
    assert lib.og_stem7x7_bf16(_lib.ptr(x), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(out), n, 48, w, 1, _lib.stream_ptr(dev)) == -1


def test_conv3x3_rejects_bad_arguments(dev):
    lib = _lib.load()
    x = torch.zeros(1, 64, 4, 4, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
    b = torch.zeros(64, device=dev)
    args = lambda cin, cout, wsb: (_lib.ptr(x), _lib.ptr(x), _lib.ptr(b), None, _lib.ptr(x), 1, 4, 4, cin, cout, 1,  # noqa: E731
                                   _lib.ptr(ws), wsb, _lib.stream_ptr(dev))
    assert lib.og_conv3x3_bf16(*args(48, 64, ws.numel())) == -4          # OG_EUNSUPPORTED: channels % 64
    assert lib.og_conv3x3_bf16(*args(64, 64, 128)) == -2                 # OG_ENOSPC
    assert lib.og_conv3x3_workspace_bytes(16, 48, 64) == 0


# engine vs the eager fp32 module, per arithmetic: relative to the largest reference value of the head
ENGINE_GATES = {torch.float32: 1e-3, torch.float16: 5e-3, torch.bfloat16: 3e-2}


def _bench_model(seed, dev):
    """model_factory + bench_init (variance-preserving random weights) with the heads at their real magnitude."""
    import bench
    p = argparse.ArgumentParser()
    models.net_cli(p)
    model, _ = models.model_factory(p.parse_args(['--no-pretrain']))
    bench.bench_init(model, seed)
    for head in model.headnets:                      # undo bench_init's head shrink: compare real magnitudes
        for m in head.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.weight.data.mul_(1e4)
    return model.to(dev).eval()


def _check_heads(ref, out, dtype, tag):
    for h in (0, 1):
        r, o = ref[h][0][-1].float(), out[h][0][-1]
        assert o.dtype == torch.float32 and o.is_contiguous() and o.shape == r.shape
        err = (o - r).abs().max().item() / r.abs().max().item()
        print(f'{tag} head {h} {dtype}: relative error {err:.2e}')
        assert err <= ENGINE_GATES[dtype], f"{tag} head {h} {dtype}: relative error {err}"


def test_engine_matches_eager_fp32(dev):
    model = _bench_model(7, dev)
    x = torch.randn(2, 3, 128, 128, device=dev)
    with torch.no_grad():
        ref = model(x)
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        for use_graph in (False, True):
            eng = models.InferenceEngine(model, 2, 128, 128, device=dev, dtype=dtype, use_graph=use_graph)
            out = eng(x)
            _check_heads(ref, out, dtype, f'128x128 graph={use_graph}')
            assert out[0][0][0] is None and out[0][1] == [[], []]     # reference nesting, unused stack skipped


@pytest.mark.parametrize("batch,height,width", [(3, 256, 384), (2, 256, 640), (1, 640, 640), (5, 384, 512)])
def test_engine_other_shapes_match_eager(dev, batch, height, width):
    """Non-square inputs route the layers through every conv path (tiled 16x16 / 40-wide tiles, stride 2, pointwise, split-K)
    and the forked up1 branches; compared with the eager fp32 module on the same weights."""
    model = _bench_model(11, dev)
    x = torch.randn(batch, 3, height, width, device=dev)
    with torch.no_grad():
        ref = model(x)
    for dtype in (torch.bfloat16, torch.float16):
        eng = models.InferenceEngine(model, batch, height, width, device=dev, dtype=dtype)
        out = eng(x)
        first = [out[h][0][-1].clone() for h in (0, 1)]           # (the engine re-uses its output buffers: compare copies)
        _check_heads(ref, out, dtype, f'{batch}x{height}x{width}')
        for _ in range(5):                                        # graph replays: bit-identical results, run after run
            out2 = eng(x)
            for h in (0, 1):
                assert torch.equal(first[h], out2[h][0][-1]), f'{dtype} {batch}x{height}x{width}: a replay differs from the first run'


@pytest.mark.parametrize("shape", [(16, 640, 640), (2, 384, 512)])
def test_engine_replays_are_bit_identical(dev, shape):
    """The same input through the same engine gives the same bits, run after run -- for the flip-test batch (16 images: its 40 -> 20
    stride-2 layer and 20x20 projections used to fall through to MIOpen, whose fp16 pick is not run-to-run exact) and a small input
    (its first 1x1 projection likewise); the split-K kernel's last arriver sums ALL slabs in slice order (adding the others to its own
    registers made the sum depend on who arrived last)."""
    model = _bench_model(5, dev)
    x = torch.randn(shape[0], 3, shape[1], shape[2], device=dev)
    eng = models.InferenceEngine(model, *shape, device=dev, dtype=torch.float16)
    first = [o.clone() for o in eng.forward_raw(x)]
    for _ in range(8):
        out = eng.forward_raw(x)
        assert all(torch.equal(a, b) for a, b in zip(first, out))


@pytest.mark.parametrize("shape", [(8, 640, 640), (16, 640, 640), (24, 640, 640), (1, 640, 384), (1, 640, 512), (1, 640, 896), (1, 128, 128)])
def test_engine_runs_no_torch_convolution(dev, shape):
    """EVERY convolution of the forward (models/hourglass_104.py:16-30,50-79,271-298) runs on a hand-written kernel: bs8 (configs[1]),
    the flip-test batch (configs[2]: 16 images), 24 images, and batch 1 at three --fixed-height widths + the smallest input.  strict
    is the default, so a fallback would already raise; torch_conv_calls is the counter a caller with strict=False would read."""
    model = _bench_model(5, dev)
    # (the routes do not depend on the 16-bit type: both arithmetics on the two smallest shapes, fp16 = the headline on the others)
    for dtype in (torch.float16, torch.bfloat16) if shape[0] * shape[1] * shape[2] <= 640 * 384 else (torch.float16,):
        eng = models.InferenceEngine(model, *shape, device=dev, dtype=dtype)
        assert eng.strict and eng.torch_conv_calls == []
        eng.forward_raw(torch.randn(shape[0], 3, shape[1], shape[2], device=dev))
        assert eng.torch_conv_calls == []
        del eng
        torch.cuda.empty_cache()


def test_engine_strict_refuses_a_torch_convolution(dev, monkeypatch):
    """A layer no hand-written kernel serves raises OgError naming the layer and its shape (here: the tiled kernels switched off, so
    the 1x1 junction layers and the large 3x3 layers have nowhere to go); strict=False runs them on torch and lists them."""
    from offsetguided_amd import _lib
    from offsetguided_amd.models import engine as E
    model = _bench_model(5, dev)
    monkeypatch.setattr(E, 'CONV_TILED', 0)
    monkeypatch.setattr(E, 'CONV_SPLITK_LAST_RESORT', 0)
    with pytest.raises(_lib.OgError, match=r'strict=True\): no hand-written kernel serves basenet\.[\w.]+: \(\d, \d\) conv \d+ -> \d+, stride .*input \(3, \d+, \d+, \d+\) float16'):
        models.InferenceEngine(model, 3, 256, 256, device=dev, use_graph=False)(torch.randn(3, 3, 256, 256, device=dev))
    eng = models.InferenceEngine(model, 3, 256, 256, device=dev, use_graph=False, strict=False)
    x = torch.randn(3, 3, 256, 256, device=dev)
    with torch.no_grad():
        ref = model(x)
    _check_heads(ref, eng(x), torch.float16, 'strict=False on torch convolutions')
    assert len(eng.torch_conv_calls) > 10 and any('basenet.cnvs_.0.0: (1, 1) conv 256 -> 256' in c for c in eng.torch_conv_calls)
    assert any('(3, 3) conv' in c for c in eng.torch_conv_calls)
    # the fp32 checking engine is torch by design: never strict, nothing counted
    e32 = models.InferenceEngine(model, 3, 256, 256, device=dev, dtype=torch.float32, use_graph=False)
    e32(x)
    assert not e32.strict and e32.torch_conv_calls == []


def test_engine_bench_shape_matches_eager(dev):
    """THE benchmarked configuration -- bs8 640x640 through the HIP-graph engine, bench_init weights (heads un-shrunk) -- vs the
    eager fp32 module at that shape (models/networks.py:189-194), both arithmetics."""
    model = _bench_model(1234, dev)
    x = torch.randn(8, 3, 640, 640, device=dev, generator=torch.Generator(dev).manual_seed(0))
    with torch.no_grad():
        ref = model(x)
    ref = [[[None, ref[0][0][-1].clone()]], [[None, ref[1][0][-1].clone()]]]
    torch.cuda.empty_cache()
    for dtype in (torch.bfloat16, torch.float16):
        eng = models.InferenceEngine(model, 8, 640, 640, device=dev, dtype=dtype)
        out = eng(x)
        _check_heads(ref, out, dtype, 'bs8 640x640')
        del eng
        torch.cuda.empty_cache()


def test_engine_scale_head(dev):
    """A model built with the keypoint-scale and jitter-offset heads: the engine returns them as
    features[omp][2][stage] and features[hmp][2][stage]."""
    import bench
    p = argparse.ArgumentParser()
    models.net_cli(p)
    model, _ = models.model_factory(p.parse_args(['--no-pretrain', '--include-scale', '--include-jitter-offset']))
    bench.bench_init(model, 5)
    x = torch.randn(1, 3, 128, 128, device=dev)
    model = model.to(dev).eval()
    with torch.no_grad():
        ref = model(x)
    out = models.InferenceEngine(model, 1, 128, 128, device=dev, use_graph=False)(x)
    r, o = ref[1][2][-1].float(), out[1][2][-1]
    assert o.shape == r.shape == (1, 17, 32, 32) and out[1][2][0] is None
    assert (o - r).abs().max().item() <= 0.05 * r.abs().max().item()
    rj, oj = ref[0][2][-1].float(), out[0][2][-1]
    assert oj.shape == rj.shape == (1, 2, 32, 32) and (oj - rj).abs().max().item() <= 0.05 * rj.abs().max().item()


def test_engine_matches_reference_golden(dev):
    """The reference model's outputs on key-seeded weights (tests/golden/backbone128.npz, generated from the imported
    reference) vs the GPU engine: fp32 engine <= 1e-3 relative, bf16 engine <= 3e-2 (1.5-1.9e-2 measured)."""
    import os
    import numpy as np
    from offsetguided_amd import synth
    from offsetguided_amd.models.seeding import key_seeded_state
    from helpers import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'backbone128.npz'))
    p = argparse.ArgumentParser()
    models.net_cli(p)
    model, _ = models.model_factory(p.parse_args(['--no-pretrain']))
    model.load_state_dict(key_seeded_state(model.state_dict()))
    model = model.to(dev).eval()
    x = torch.from_numpy(synth.noise_batch(int(g['input_seed']), (1, 3, 128, 128))).to(dev)
    for dtype, tol in ((torch.float32, 1e-3), (torch.bfloat16, 0.03)):
        eng = models.InferenceEngine(model, 1, 128, 128, device=dev, dtype=dtype, use_graph=False)
        out = eng(x)
        for h, name in ((0, 'hm'), (1, 'off')):
            ref = g[name]
            err = np.abs(out[h][0][-1].cpu().numpy() - ref).max() / np.abs(ref).max()
            print(f'{name} {dtype}: relative error vs reference model {err:.2e}')
            assert err <= tol, f'{name} {dtype}: relative error {err}'


def test_run_images_synthetic(dev):
    """evaluate.run_images end to end on synthetic batches (engine + pipelined decode + result dicts)."""
    from offsetguided_amd import evaluate
    a = evaluate.evaluate_cli(['--no-pretrain', '--topk', '16', '--thre-hmp', '0.04', '--person-thre', '0.04',
                               '--dist-max', '40', '--long-edge', '128', '--batch-size', '2', '--flip-test',
                               '--print-freq', '1'])
    results, ids = evaluate.run_images(a, n_synthetic_batches=3)
    assert ids == list(range(6))
    assert all(len(r['keypoints']) == 51 and r['category_id'] == 1 for r in results)
    assert {r['image_id'] for r in results} == set(range(6))       # every image reports at least the fallback entry


# ---------------------------------------------------------------------------------- fp16 build of the backbone kernels
@pytest.mark.parametrize("shape", [(8, 5, 5, 512, 512, 3, 1), (2, 20, 20, 384, 384, 3, 2), (2, 160, 160, 256, 256, 3, 1),
                                   (2, 80, 80, 256, 256, 3, 1), (2, 40, 40, 384, 384, 3, 1), (3, 7, 9, 128, 64, 1, 2),
                                   (2, 33, 31, 128, 256, 3, 1)])
def test_conv2d_f16_matches_torch(dev, shape):
    """og_conv2d_f16 (the split-K kernel on small and large shapes, 1x1, stride 2) vs an fp32 torch convolution of the
    same fp16 operands: differences are the final fp16 rounding (2^-11 relative) plus fp32 summation order."""
    import torch.nn.functional as F
    n, h, w, cin, cout, k, st = shape
    lib = _lib.load()
    g = torch.Generator(device='cpu').manual_seed(h * 1000 + cin + k * 7 + st)
    cl = torch.channels_last
    x = torch.randn(n, cin, h, w, generator=g).to(dev).to(torch.float16).contiguous(memory_format=cl)
    wt = (torch.randn(cout, cin, k, k, generator=g) * (1.0 / (k * k * cin)) ** 0.5).to(dev).to(torch.float16).contiguous(memory_format=cl)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    base = F.conv2d(x.float(), wt.float(), bias, st, k // 2)
    ho, wo = base.shape[2:]
    skip = torch.randn(n, cout, ho, wo, generator=g).to(dev).to(torch.float16).contiguous(memory_format=cl)
    ws = torch.zeros(lib.og_conv2d_workspace_bytes(n, h, w, cin, cout, k, st), dtype=torch.uint8, device=dev)
    for use_skip, relu in ((True, 1), (False, 0)):
        ref = base + skip.float() if use_skip else base
        if relu:
            ref = F.relu(ref)
        out = torch.full_like(skip, float('nan'))
        for _ in range(2):
            _lib.check(lib.og_conv2d_f16(_lib.ptr(x), _lib.ptr(wt), _lib.ptr(bias), _lib.ptr(skip) if use_skip else None,
                                         _lib.ptr(out), n, h, w, cin, cout, k, st, relu, _lib.ptr(ws), ws.numel(),
                                         _lib.stream_ptr(dev)), lib)
        err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
        assert err <= 1e-3, f'relative error {err}'


def test_f16_epilogues_and_layout_kernels(dev):
    lib = _lib.load()
    cl = torch.channels_last
    g = torch.Generator(device='cpu').manual_seed(3)
    y = torch.randn(2, 64, 9, 7, generator=g).to(dev).to(torch.float16).contiguous(memory_format=cl)
    skip = torch.randn(2, 64, 9, 7, generator=g).to(dev).to(torch.float16).contiguous(memory_format=cl)
    bias = torch.randn(64, generator=g).to(dev)
    ref = torch.relu(y.float() + bias.view(1, -1, 1, 1) + skip.float()).to(torch.float16)
    _lib.check(lib.og_bias_act_f16(_lib.ptr(y), _lib.ptr(bias), _lib.ptr(skip), 2 * 9 * 7, 64, 1, _lib.stream_ptr(dev)), lib)
    assert torch.equal(y, ref)
    up = torch.randn(2, 384, 10, 20, generator=g).to(dev).to(torch.float16).contiguous(memory_format=cl)
    low = torch.randn(2, 384, 5, 10, generator=g).to(dev).to(torch.float16).contiguous(memory_format=cl)
    ref = (up.float() + torch.nn.functional.interpolate(low.float(), scale_factor=2, mode='nearest')).to(torch.float16)
    _lib.check(lib.og_upsample2_add_f16(_lib.ptr(up), _lib.ptr(low), 2, 10, 20, 384, _lib.stream_ptr(dev)), lib)
    assert torch.equal(up, ref)
    img = torch.randn(2, 3, 32, 64, generator=g).to(dev)
    nhwc = torch.empty((2, 3, 32, 64), dtype=torch.float16, device=dev, memory_format=cl)
    _lib.check(lib.og_nchw_f32_to_nhwc_f16(_lib.ptr(img), _lib.ptr(nhwc), 2, 3, 32, 64, _lib.stream_ptr(dev)), lib)
    assert torch.equal(nhwc, img.to(torch.float16).contiguous(memory_format=cl))
    feat = torch.randn(2, 56, 8, 8, generator=g).to(dev).to(torch.float16).contiguous(memory_format=cl)
    hb = torch.randn(56, generator=g).to(dev)
    out = torch.empty((2, 17, 8, 8), device=dev)
    _lib.check(lib.og_nhwc_f16_to_nchw_f32(_lib.ptr(feat), 56, 3, 17, _lib.ptr(hb), _lib.ptr(out), 2, 8, 8, _lib.stream_ptr(dev)), lib)
    assert torch.equal(out, feat[:, 3:20].float() + hb[3:20].view(1, -1, 1, 1))


def test_engine_f16_matches_reference_golden(dev):
    """fp16 engine (the reference's apex-O2 arithmetic, evaluate.py:92,198-201) vs the REFERENCE model's outputs on
    key-seeded weights: 2.5e-3 relative measured, gated at 4e-3 (bf16: 1.3e-2 ... 1.9e-2), through the HIP graph as well."""
    import os
    import numpy as np
    from offsetguided_amd import synth
    from offsetguided_amd.models.seeding import key_seeded_state
    from helpers import GOLDEN
    g = np.load(os.path.join(GOLDEN, 'backbone128.npz'))
    p = argparse.ArgumentParser()
    models.net_cli(p)
    model, _ = models.model_factory(p.parse_args(['--no-pretrain']))
    model.load_state_dict(key_seeded_state(model.state_dict()))
    x = torch.from_numpy(synth.noise_batch(int(g['input_seed']), (1, 3, 128, 128))).to(dev)
    for use_graph in (False, True):
        eng = models.InferenceEngine(model, 1, 128, 128, device=dev, dtype=torch.float16, use_graph=use_graph)
        out = eng(x)
        for h, name in ((0, 'hm'), (1, 'off')):
            ref = g[name]
            err = np.abs(out[h][0][-1].cpu().numpy() - ref).max() / np.abs(ref).max()
            print(f'{name} fp16 (graph {use_graph}): relative error vs reference model {err:.2e}')
            assert err <= 4e-3, f'{name} fp16: relative error {err}'


@pytest.mark.parametrize("knobs", [{"OG_CONV_UP2": "0", "OG_ENGINE_TRUNK_FIRST": "0"},
                                   {"OG_ENGINE_TRUNK_FIRST": "1", "OG_CONV_BAND_MAX_PIXELS": "0", "OG_ENGINE_DEEP_SHARED": "0"},
                                   {"OG_ENGINE_DEEP_SHARED": "2"}])
def test_engine_schedule_knobs(dev, knobs):
    """The engine's kept A/B switches (read at import), every non-default value in one of three child processes: merges as their own
    launches instead of on the producing convolution's epilogue + the up1 branch captured before the trunk below the fork; trunk-first
    from depth 1 + the small levels back on the split-K kernel with every up1 branch on its own stream; the inner branches sharing a
    stream from depth 2 -- the bench-shape engine test (graph replay against the eager model) again in each."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, **knobs)
    r = subprocess.run([sys.executable, "-m", "pytest", __file__, "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k",
                        "test_engine_bench_shape_matches_eager"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-1000:]
    assert " passed" in r.stdout
