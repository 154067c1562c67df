"""MI355X inference engine for NetworkWrapper(Hourglass104 + hmp/omp heads).

What this file adds to the eager module is the shape the hardware wants:
  * every BatchNorm folded into the preceding convolution (160 BN layers disappear);
  * fp16 (default: the reference's apex-O2 arithmetic) or bf16 activations / weights in channels-last (NHWC) layout;
  * EVERY convolution on a hand-written MFMA kernel, bias + ReLU + residual add fused into its epilogue, fp32 accumulation:
    og_conv3x3_tiled_* / og_conv3x3s2_tiled_* / og_conv1x1_tiled_* for the 160x160 ... 20x20 levels (two 4-wave workgroups
    per CU, pre-tiled weights), the band-resident og_conv_band_* (K split inside a workgroup, no slabs) for 10x10 / 5x5, the
    split-K kernel og_conv2d_* / og_conv2d_proj_* for the stride-2 / projection layers at 20x20 and whatever the band kernel does
    not serve, og_stem7x7_* for the stem, og_conv1x1_heads_* for the heads;
    a shape none of them serves RAISES (InferenceEngine(strict=True), the default on the 16-bit GPU engines); with strict=False it
    runs on torch's convolution + one og_bias_act_* pass and is listed in engine.torch_conv_calls;
  * the hourglass merges (nearest x2 upsample + add) on the epilogue of the convolution below them where that kernel is the
    tiled one, else one og_upsample2_add_* pass;
  * only the decoded stack's heads are evaluated (decoder/factory.py:60-63 reads feat_stage only), all of them as one
    1x1 convolution whose result leaves as dense fp32 NCHW tensors, the layout the HIP decoder kernels stream;
  * the whole forward (~170 launches, the up1 branches of every level forked onto side streams) is captured once into
    a HIP graph and replayed: launch latency, not math, bounds the deep levels.
The result keeps the reference nesting [ (hmps[S], bg[S], jo[S]), (offs[S], spreads[S], scales[S]) ]
(models/networks.py:189-194); stacks that are not decoded hold None.

Building an engine does not touch the caller's module: the weights are folded from its state_dict, the module's
train / eval flag and device stay as they are.

Environment switches (read at import; the measured-best value is the default): OG_CONV_TILED (7: bit 0 / 1 / 2 = 3x3 stride 1 /
3x3 stride 2 / 1x1 layers on the tiled kernels), OG_CONV_UP2 (1: merges on the producing convolution's epilogue),
OG_ENGINE_BRANCHES (1) and OG_ENGINE_BRANCH_MAX_DEPTH (4): the up1 forks, OG_ENGINE_TRUNK_FIRST (2: capture order at the forks of
depth >= 2), OG_ENGINE_DEEP_SHARED (3: the inner up1 branches in fork order on one stream), OG_CONV_BAND_MAX_PIXELS (1024: the
band-resident kernel for 10x10 / 5x5), OG_ENGINE_WHATIF (timing diagnosis, wrong results).  The experiments that lost their A/B (branch delay, shared side stream, stream priorities, ...) are described in
EXPERIMENTS.md and no longer exist as switches.
"""
import os
import threading

import torch
import torch.nn.functional as F

from .. import _lib
from .hourglass_104 import ConvBlock, HourglassLevel, Residual




def _fold(conv, bn):
    """conv (+ optional BatchNorm, folded with its running statistics whatever the module's train / eval flag) ->
    (weight, bias) fp32 on the engine's device.  Reads the module, never modifies or moves it."""
    w = conv.weight.detach().float()
    b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
    if isinstance(bn, torch.nn.BatchNorm2d):
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        w = w * scale[:, None, None, None]
        b = (b - bn.running_mean.detach().float()) * scale + bn.bias.detach().float()
    if _issuer.build_device is not None:   # the engine under construction on this thread
        w, b = w.to(_issuer.build_device), b.to(_issuer.build_device)
    return w, b


def _epilogue(y, bias32, bias_lp, skip, relu, fused):
    """y = act(y + bias (+ skip)) in place.  fused: one hand-written HIP pass (og_bias_act_bf16);
    otherwise plain torch ops (CPU / fp32 checking path)."""
    if fused:
        n, c, h, w = y.shape
        assert y.is_contiguous(memory_format=torch.channels_last) and (skip is None or skip.shape == y.shape)
        if skip is not None and not skip.is_contiguous(memory_format=torch.channels_last):
            skip = skip.contiguous(memory_format=torch.channels_last)
        lib = _lib.load()
        _lib.check(_lib.lp(lib, 'og_bias_act', y.dtype)(_lib.ptr(y), _lib.ptr(bias32), _lib.ptr(skip) if skip is not None else None,
                                        n * h * w, c, int(relu), _lib.stream_ptr(y.device)), lib)
        return y
    y += bias_lp.view(1, -1, 1, 1)
    if skip is not None:
        y += skip
    return F.relu_(y) if relu else y


# Layers with at most this many output pixels (N*H*W) run on the split-K MFMA kernel (og_conv2d_*, epilogue fused): the 10x10 /
# 5x5 levels at batch 8, and the stride-2 layers whose output is that small.
CONV3X3_MAX_PIXELS = 8192      # (4096 until round 5: with 16 images -- flip-test at batch 8 -- the 40 -> 20 stride-2 layer and the 20x20
CONV_S2_MAX_PIXELS = 8192      # projections fell through to MIOpen, whose pick was not run-to-run exact, and slower)
CONV_SPLITK_LAST_RESORT = 1 << 17   # above the preferences: any 3x3 layer up to this size runs on the split-K kernel rather than on MIOpen
# The large levels' layers run on the tiled kernels (og_conv3x3_tiled_* / og_conv3x3s2_tiled_* / og_conv1x1_tiled_*: two workgroups
# per CU, pre-tiled weights) where they serve the shape: bit 0 = 3x3 stride 1, bit 1 = 3x3 stride 2, bit 2 = 1x1 / heads.
CONV_TILED = int(os.environ.get('OG_CONV_TILED', '7'))
CONV_PW_MIN_PIXELS = 1024        # 1x1 layers below this stay on torch's convolution (the deep projections ride on conv2 instead;
                                 # 8192 until round 5: MIOpen's fp16 pick for the first projection of a small input was not run-to-run exact)
CONV_TILED_MIN_PIXELS = 2048     # 20x20 at batch 8 = 3 200 pixels: 20 x 4 tiles split along K
# OG_CONV_BAND_MAX_PIXELS = P (default 1024; 0 = off): 3x3 layers (stride 1 | 2, with or without the residual's 1x1 projection) with
# at most P output pixels run on the band-resident kernel (og_conv_band_*, csrc/conv_band.hip: one 4-wave workgroup per (image,
# row band, 16 output channels), K split over its waves, partial tiles summed in LDS -- no fp32 slabs, no tickets) where it serves
# the shape: 1024 = the 10x10 / 5x5 levels at batch 8.  Measured (round 4, EXPERIMENTS.md): a 5x5 layer alone 12.4 -> 8.2 us, a
# 10x10 layer 13.7 -> 11.7; in the network the kernel is worth 1.3 % of the step TOGETHER with OG_ENGINE_DEEP_SHARED=3 (each alone:
# nothing).  (Chained persistent launches of these layers / of the 160x160 layers: tools/experiments/*_chain.patch.)
CONV_BAND_MAX_PIXELS = int(os.environ.get('OG_CONV_BAND_MAX_PIXELS', '1024'))
# The up1 branch of every hourglass level is independent of the whole pyramid below it (kp_module.forward,
# models/hourglass_104.py:183-190).  With OG_ENGINE_BRANCHES=1 it runs on its own stream, forked and joined inside the
# captured HIP graph, so that the large up1 convolutions fill the CUs the latency-bound 20x20..5x5 levels leave idle.
BRANCHES = int(os.environ.get('OG_ENGINE_BRANCHES', '1'))
BRANCH_MAX_DEPTH = int(os.environ.get('OG_ENGINE_BRANCH_MAX_DEPTH', '4'))   # fork up1 only at levels <= this depth
# OG_ENGINE_TRUNK_FIRST = D (0 = off): at the levels of depth >= D the up1 branch is queued on its side stream AFTER the first kernel
# of the trunk below the fork point (it still depends on the fork point only, through an event recorded there).  The HIP graph
# executor keeps a node on the queue of the parent whose FIRST child it is (capture order): with the branch captured first,
# "fork point -> up1 -> merge" stays on one hardware queue and the latency-critical trunk is moved to a new one -- every merge
# then waits across queues for the trunk (11 us in the forward timeline).  Trunk first: the trunk keeps its queue through forks
# and merges, a fork costs it ~6 us instead (the release behind the fork point).
# Default 2 since round 5 (two batches in flight, faster small levels): 5.859 vs 5.894 ms per step for 1, seven of seven repeats
# (profiles/r05_capture_order_ab.log); 3: 5.91.
TRUNK_FIRST = int(os.environ.get('OG_ENGINE_TRUNK_FIRST', '2'))
# OG_ENGINE_DEEP_SHARED = D (default 3; 0 = off): the up1 branches of the levels of depth >= D share ONE side stream, in the order
# the trunk forks them (depth 3, then depth 4): the graph executor otherwise puts the inner branches on one hardware queue DEEPEST
# first, so the 20x20 branch of depth 3 -- ready when the trunk enters depth 3 -- only starts behind the 10x10 branch of depth 4
# and the trunk waits for it at the depth-3 merge.  A/B over three runs: 6.02 -> 5.95 ms with the band kernel, D = 2 / 4 lose.
DEEP_SHARED = int(os.environ.get('OG_ENGINE_DEEP_SHARED', '3'))
# OG_CONV_WARM_NEXT (default 1): a tiled / band convolution launch is told where the NEXT layer's packed weights lie
# (og_conv_next_weights_hint) and touches them at entry: the layers of the 20x20 / 10x10 / 5x5 levels are bound by first-touch weight
# reads from HBM (the big layers of the same forward have flushed the memory-side cache: 26 us cold vs 19.7 us warm for a 20x20 layer)
CONV_WARM_NEXT = int(os.environ.get('OG_CONV_WARM_NEXT', '1'))
CONV_UP2 = int(os.environ.get('OG_CONV_UP2', '1'))   # the hourglass merge (upsample x2 + add) on the epilogue of the convolution below it
_conv_ws = {}
_WHATIF = set(filter(None, os.environ.get('OG_ENGINE_WHATIF', '').split(',')))
class _Issuer(threading.local):
    """Who is issuing work on THIS host thread (engines may be built / run from several threads and on several
    streams): the engine id and the concurrent branch (0 = trunk, d+1 = up1 branch of level d) select the scratch."""
    engine = 0
    branch = 0
    build_device = None   # device the folded weights of the engine under construction go to
    ws = None             # the issuing engine's scratch buffers (None: a bare _Conv call outside an engine -> the module's _conv_ws)
    eng = None            # the InferenceEngine whose forward is being issued (strict mode, torch_conv_calls)
    names = None          # id(conv module) -> qualified name, while an engine's layers are being built


_issuer = _Issuer()
_live_engines = __import__('weakref').WeakValueDictionary()      # engine id -> InferenceEngine, while it exists


def _torch_conv(name, x, weight, stride=1, padding=0):
    """The ONLY place the engine reaches torch's convolution (MIOpen on the GPU).  The fp32 / CPU checking path lives here; on a
    16-bit GPU engine it means "a shape none of the hand-written kernels serves": counted in `engine.torch_conv_calls` and, in
    strict mode (the default there), refused with the layer and shape -- the flip-test engine ran three layers on MIOpen through
    rounds 1-5 without anybody noticing (models/hourglass_104.py:16-30,50-79 is what must run on our kernels)."""
    eng = _issuer.eng
    if eng is not None and eng.fused:
        what = (f'{name}: {tuple(weight.shape[2:])} conv {weight.shape[1]} -> {weight.shape[0]}, stride {tuple(stride) if not isinstance(stride, int) else stride}, '
                f'input {tuple(x.shape)} {str(x.dtype).replace("torch.", "")}')
        eng.torch_conv_calls.append(what)
        if eng.strict:
            raise _lib.OgError(f'InferenceEngine(strict=True): no hand-written kernel serves {what}; it would run on torch / MIOpen '
                               '(pass strict=False to allow and count it in engine.torch_conv_calls)')
    return F.conv2d(x, weight, None, stride, padding)


_warm_streams = {}
_n_engines = 0


def _conv3x3_workspace(device, nbytes):
    """One zero-initialised scratch per (engine, device, concurrent branch) for og_conv3x3_bf16 (zero page + split-K slabs).
    Layers of one branch run back to back on one stream, so they share it; it only ever grows outside graph capture.  The
    buffers belong to the ENGINE that issues the work (_issuer.ws = InferenceEngine._ws) and go with it: a process-wide cache keyed
    by engine id kept ~17 MB per engine ever built (run_images --fixed-height builds one per new width)."""
    ws = _issuer.ws if _issuer.ws is not None else _conv_ws
    key = (device.index, _issuer.branch)
    buf = ws.get(key)
    if buf is None or buf.numel() < nbytes:
        assert not torch.cuda.is_current_stream_capturing(), 'conv3x3 workspace must be sized before graph capture'
        buf = ws[key] = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
    return buf


class _Conv:
    """Folded conv: weight in the engine dtype (channels-last), bias kept apart for the epilogue."""

    def __init__(self, conv, bn, relu, dtype, fused):
        w, b = _fold(conv, bn)
        self.name = (_issuer.names or {}).get(id(conv), type(conv).__name__)
        self.w = w.to(dtype).contiguous(memory_format=torch.channels_last)
        self.b32 = b.float().contiguous()
        self.b = b.to(dtype)
        self.stride, self.pad, self.relu = conv.stride, conv.padding, relu
        self.w_tiled = None          # weights in og_conv3x3_tiled_*'s layout, made on first use
        self.w_band = None           # weights (+ projection) in og_conv_band_*'s fragment order, made on first use
        self.next_conv = None        # the layer that runs after this one in its chain (set by _link): whose weights to warm
        self.w_alt = None            # conv2 of a projection block on the split-K kernel reads [w | projection] (_Residual.w_cat)
        self.fused = fused and w.shape[0] % 8 == 0
        strides = ((1, 1), (2, 2))
        self.hip3x3 = (self.fused and tuple(w.shape[2:]) == (3, 3) and tuple(conv.stride) in strides
                       and tuple(conv.padding) == (1, 1) and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0)

    def raw(self, x):
        fake = self._whatif_skipped(x)
        if fake is not None:
            return fake
        return _torch_conv(self.name, x, self.w, self.stride, self.pad)

    def _whatif_skipped(self, x):
        """OG_ENGINE_WHATIF=<classes> (diagnostic, WRONG RESULTS): layers of the named classes produce an uninitialised
        output instead of running -- the forward time that disappears is that class's share of the critical path.
        Classes: s2big (large stride-2 3x3), 1x1 (pointwise), chain (the small levels), c160 / c80 / c40 (tiled 3x3 by level)."""
        if not _WHATIF:
            return None
        n, c, h, w = x.shape
        st, k = self.stride[0], self.w.shape[2]
        ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
        pixels = n * ho * wo
        cls = ('1x1' if k == 1 else 'chain' if pixels <= CONV3X3_MAX_PIXELS else 's2big' if st == 2 else
               'c160' if ho >= 160 else 'c80' if ho >= 80 else 'c40')
        if cls not in _WHATIF:
            return None
        return torch.empty((n, self.w.shape[0], ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)

    def __call__(self, x, skip=None):
        fake = self._whatif_skipped(x)
        if fake is not None:
            return fake
        n, c, h, w = x.shape
        if self.hip3x3:
            st = self.stride[0]
            pixels = n * ((h - 1) // st + 1) * ((w - 1) // st + 1)
            if self.band_ok(x):
                return self.band(x, skip)
            # 3x3 stride 1 on the tiled kernel where it serves the shape, the small levels (and the small stride-2 layers) on the
            # split-K kernel
            if (st == 1 and (CONV_TILED & 1) and pixels >= CONV_TILED_MIN_PIXELS and self.w.shape[0] % 128 == 0
                    and _lib.load().og_conv3x3_tiled_supported(n, h, w, c, self.w.shape[0])):
                return self._hip(x, skip)
            if pixels <= (CONV_S2_MAX_PIXELS if st == 2 else CONV3X3_MAX_PIXELS):
                return self._hip(x, skip)
            if (st == 2 and (CONV_TILED & 2) and x.is_contiguous(memory_format=torch.channels_last)
                    and _lib.load().og_conv3x3s2_tiled_supported(n, h, w, c, self.w.shape[0])):
                if skip is not None and not skip.is_contiguous(memory_format=torch.channels_last):
                    skip = skip.contiguous(memory_format=torch.channels_last)
                return self._tiled_s2(x, skip)
            if pixels <= CONV_SPLITK_LAST_RESORT:      # a shape no tiled kernel serves (e.g. the 20-wide level of 24 images): still ours
                return self._hip(x, skip)
        return _epilogue(self.raw(x), self.b32, self.b, skip, self.relu, self.fused)

    def band_ok(self, x, x2=None, proj=None):
        """This 3x3 layer (+ the 1x1 projection `proj` of x2, residual.skip) can run on og_conv_band_* for input x."""
        return x.is_cuda and self.band_ok_shape(tuple(x.shape), tuple(x2.shape) if x2 is not None else None, proj, x.dtype)

    def band_ok_shape(self, x_shape, x2_shape, proj, dtype):
        if not (CONV_BAND_MAX_PIXELS and self.hip3x3 and dtype == self.w.dtype) or _WHATIF:
            return False
        n, c, h, w = x_shape
        st = self.stride[0]
        if c != self.w.shape[1] or n * ((h - 1) // st + 1) * ((w - 1) // st + 1) > CONV_BAND_MAX_PIXELS:
            return False
        h2 = w2 = c2 = 0
        st2 = 1
        if proj is not None:
            if tuple(proj.w.shape[2:]) != (1, 1) or tuple(proj.pad) != (0, 0) or x2_shape is None:
                return False
            _, c2, h2, w2 = x2_shape
            st2 = proj.stride[0]
        # one round of workgroups: a layer cut into more bands than that (stride 2 from 20x20: 576) is slower than the split-K kernel
        return 0 < _lib.load().og_conv_band_supported(n, h, w, c, self.w.shape[0], st, h2, w2, c2, st2) <= 320

    def band_pack(self, device, proj=None):
        """The layer's weights (+ the projection's) in og_conv_band_*'s fragment order, packed once (never inside graph capture)."""
        c2 = proj.w.shape[1] if proj is not None else 0
        # the packed image holds the projection's weights behind the 3x3 ones: a layer is packed for ONE of the two routes
        assert self.w_band is None or self.w_band_c2 == c2, 'band_pack: layer already packed with a different projection'
        if self.w_band is None:
            assert not torch.cuda.is_current_stream_capturing(), 'weights must be packed before graph capture'
            lib = _lib.load()
            cout, c = self.w.shape[0], self.w.shape[1]
            self.w_band_c2 = c2
            self.w_band = torch.empty(cout * (9 * c + c2), dtype=self.w.dtype, device=self.w.device)
            w2 = proj.w.reshape(cout, c2).contiguous() if proj is not None else None
            _lib.check(lib.og_conv_band_pack_w16(_lib.ptr(self.w), _lib.ptr(w2) if w2 is not None else None, c, cout, c2,
                                                 _lib.ptr(self.w_band), _lib.stream_ptr(device)), lib)
        return self.w_band

    def band(self, x, skip=None, x2=None, proj=None):
        """act(conv3x3(x) (+ conv1x1(x2) of `proj`) + bias (+ skip)) on og_conv_band_* (bias = the sum of both folded biases when a
        projection rides along, see _Residual)."""
        n, c, h, w = x.shape
        cout, st = self.w.shape[0], self.stride[0]
        lib = _lib.load()
        self.band_pack(x.device, proj)
        x = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
        if skip is not None and not skip.is_contiguous(memory_format=torch.channels_last):
            skip = skip.contiguous(memory_format=torch.channels_last)
        h2 = w2_ = c2 = 0
        st2 = 1
        if proj is not None:
            x2 = x2 if x2.is_contiguous(memory_format=torch.channels_last) else x2.contiguous(memory_format=torch.channels_last)
            _, c2, h2, w2_ = x2.shape
            st2 = proj.stride[0]
        out = torch.empty((n, cout, (h - 1) // st + 1, (w - 1) // st + 1), dtype=x.dtype, device=x.device,
                          memory_format=torch.channels_last)
        self.warm_next(lib)
        _lib.check(_lib.lp(lib, 'og_conv_band', x.dtype)(
            _lib.ptr(x), _lib.ptr(self.w_band), _lib.ptr(self.b32), _lib.ptr(skip) if skip is not None else None,
            _lib.ptr(x2) if proj is not None else None, _lib.ptr(out), n, h, w, c, cout, st, int(self.relu), h2, w2_, c2, st2,
            _lib.stream_ptr(x.device)), lib)
        return out

    def pointwise_ok(self, x):
        """This 1x1 layer can run on og_conv1x1_tiled_* for input x (large levels, channel multiples the kernel serves)."""
        n, c, h, w = x.shape
        st = self.stride[0]
        return ((CONV_TILED & 4) and self.fused and tuple(self.w.shape[2:]) == (1, 1) and tuple(self.pad) == (0, 0)
                and c % 64 == 0 and self.w.shape[0] % 128 == 0 and x.is_cuda
                and x.is_contiguous(memory_format=torch.channels_last)
                and n * ((h - 1) // st + 1) * ((w - 1) // st + 1) >= CONV_PW_MIN_PIXELS)

    def pointwise(self, x, x2=None, other=None, bias=True, skip=None):
        """act(W x (+ W_other x2) + bias (+ skip)) in one launch; `other` = the second 1x1 layer of the junction (its weight is
        concatenated along K on first use, its bias is already summed into this layer's).  bias=False / relu off = raw conv."""
        n, c, h, w = x.shape
        st, cout = self.stride[0], self.w.shape[0]
        ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
        lib = _lib.load()
        if self.w_tiled is None:
            assert not torch.cuda.is_current_stream_capturing(), 'weights must be tiled before graph capture'
            wcat = self.w.reshape(cout, -1) if other is None else torch.cat([self.w.reshape(cout, -1), other.w.reshape(cout, -1)], 1)
            wcat = wcat.contiguous()
            self.w_tiled = torch.empty(wcat.numel(), dtype=self.w.dtype, device=self.w.device)
            _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wcat), wcat.shape[1], cout, 2, _lib.ptr(self.w_tiled), _lib.stream_ptr(x.device)), lib)
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        if x2 is not None:
            if x2.shape != x.shape:
                raise ValueError(f'pointwise: the second input {tuple(x2.shape)} must have the shape of the first {tuple(x.shape)}')
            if not x2.is_contiguous(memory_format=torch.channels_last):
                x2 = x2.contiguous(memory_format=torch.channels_last)
        if skip is not None and not skip.is_contiguous(memory_format=torch.channels_last):
            skip = skip.contiguous(memory_format=torch.channels_last)
        out = torch.empty((n, cout, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        _lib.check(_lib.lp(lib, 'og_conv1x1_tiled', x.dtype)(
            _lib.ptr(x), c, h, w, st, _lib.ptr(x2) if x2 is not None else None, c if x2 is not None else 0, h, w, st,
            _lib.ptr(self.w_tiled), _lib.ptr(self.b32) if bias else None, _lib.ptr(skip) if skip is not None else None, _lib.ptr(out),
            n, ho, wo, cout, int(self.relu and bias), _lib.stream_ptr(x.device)), lib)
        return out

    def _tiled_s2(self, x, skip):
        """3x3 stride 2 on og_conv3x3s2_tiled_* (the large down-sampling layers: 320 -> 160, 160 -> 80), epilogue fused."""
        n, c, h, w = x.shape
        cout = self.w.shape[0]
        lib = _lib.load()
        if self.w_tiled is None:
            assert not torch.cuda.is_current_stream_capturing(), 'weights must be tiled before graph capture'
            self.w_tiled = torch.empty(self.w.numel(), dtype=self.w.dtype, device=self.w.device)
            _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(self.w), c, cout, 1, _lib.ptr(self.w_tiled), _lib.stream_ptr(x.device)), lib)
        out = torch.empty((n, cout, h // 2, w // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        self.warm_next(lib)
        _lib.check(_lib.lp(lib, 'og_conv3x3s2_tiled', x.dtype)(_lib.ptr(x), _lib.ptr(self.w_tiled), _lib.ptr(self.b32),
                                                             _lib.ptr(skip) if skip is not None else None, _lib.ptr(out), n, h, w, c,
                                                             cout, int(self.relu), _lib.stream_ptr(x.device)), lib)
        return out

    def up2_ok(self, x):
        """This layer can run as og_conv3x3_tiled_up2_* on x: the merge of the hourglass level above rides on its epilogue."""
        n, c, h, w = x.shape
        return (CONV_UP2 and self.hip3x3 and self.stride[0] == 1 and (CONV_TILED & 1) and n * h * w >= CONV_TILED_MIN_PIXELS
                and self.w.shape[0] % 128 == 0 and not _WHATIF and x.is_cuda
                and n * h * w * 4 * self.w.shape[0] < 2 ** 30        # the (N, Cout, 2H, 2W) tensor it updates in place
                and _lib.load().og_conv3x3_tiled_supported(n, h, w, c, self.w.shape[0]))

    def warm_next(self, lib):
        """og_conv_next_weights_hint for the launch that follows: the packed weights of the layer after this one, in whichever layout
        that layer has been run with (none yet during the first warm-up pass: its raw weights then)."""
        nxt = self.next_conv
        if not CONV_WARM_NEXT or nxt is None:
            return
        t = nxt.w_band if nxt.w_band is not None else (nxt.w_tiled if nxt.w_tiled is not None else (nxt.w_alt if nxt.w_alt is not None else nxt.w))
        lib.og_conv_next_weights_hint(_lib.ptr(t), t.numel() * t.element_size())

    def up2(self, x, skip, up):
        """up += nearest_x2(act(conv(x) + bias + skip)) in one launch (up: (N, Cout, 2H, 2W) channels-last, in place)."""
        n, c, h, w = x.shape
        cout = self.w.shape[0]
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        if tuple(up.shape) != (n, cout, 2 * h, 2 * w) or not up.is_contiguous(memory_format=torch.channels_last):
            raise ValueError(f'up2: `up` must be a channels-last {(n, cout, 2 * h, 2 * w)} tensor, got {tuple(up.shape)}')
        if skip is not None and not skip.is_contiguous(memory_format=torch.channels_last):
            skip = skip.contiguous(memory_format=torch.channels_last)
        lib = _lib.load()
        if self.w_tiled is None:
            assert not torch.cuda.is_current_stream_capturing(), 'weights must be tiled before graph capture'
            self.w_tiled = torch.empty(self.w.numel(), dtype=self.w.dtype, device=self.w.device)
            _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(self.w), c, cout, 0, _lib.ptr(self.w_tiled), _lib.stream_ptr(x.device)), lib)
        need = lib.og_conv3x3_tiled_workspace_bytes(n, h, w, c, cout)
        ws = _conv3x3_workspace(x.device, need) if need else None
        self.warm_next(lib)
        _lib.check(_lib.lp(lib, 'og_conv3x3_tiled_up2', x.dtype)(_lib.ptr(x), _lib.ptr(self.w_tiled), _lib.ptr(self.b32),
                                                                 _lib.ptr(skip) if skip is not None else None, _lib.ptr(up), n, h, w,
                                                                 c, cout, int(self.relu), _lib.ptr(ws) if need else None,
                                                                 ws.numel() if need else 0, _lib.stream_ptr(x.device)), lib)

    def _hip(self, x, skip):
        n, c, h, w = x.shape
        cout, st = self.w.shape[0], self.stride[0]
        if not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        if skip is not None and not skip.is_contiguous(memory_format=torch.channels_last):
            skip = skip.contiguous(memory_format=torch.channels_last)
        lib = _lib.load()
        out = torch.empty((n, cout, (h - 1) // st + 1, (w - 1) // st + 1), dtype=x.dtype, device=x.device,
                          memory_format=torch.channels_last)
        if ((CONV_TILED & 1) and st == 1 and n * h * w >= CONV_TILED_MIN_PIXELS
                and lib.og_conv3x3_tiled_supported(n, h, w, c, cout)):
            if self.w_tiled is None:     # tiled once, during the warm-up passes (never inside graph capture)
                assert not torch.cuda.is_current_stream_capturing(), 'weights must be tiled before graph capture'
                self.w_tiled = torch.empty(self.w.numel(), dtype=self.w.dtype, device=self.w.device)
                _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(self.w), c, cout, 0, _lib.ptr(self.w_tiled), _lib.stream_ptr(x.device)), lib)
            need = lib.og_conv3x3_tiled_workspace_bytes(n, h, w, c, cout)      # K-split levels: tickets + fp32 slabs
            ws = _conv3x3_workspace(x.device, need) if need else None
            self.warm_next(lib)
            _lib.check(_lib.lp(lib, 'og_conv3x3_tiled', x.dtype)(_lib.ptr(x), _lib.ptr(self.w_tiled), _lib.ptr(self.b32),
                                                               _lib.ptr(skip) if skip is not None else None, _lib.ptr(out), n, h, w,
                                                               c, cout, int(self.relu), _lib.ptr(ws) if need else None,
                                                               ws.numel() if need else 0, _lib.stream_ptr(x.device)), lib)
            return out
        ws = _conv3x3_workspace(x.device, lib.og_conv2d_workspace_bytes(n, h, w, c, cout, 3, st))
        self.warm_next(lib)
        _lib.check(_lib.lp(lib, 'og_conv2d', x.dtype)(_lib.ptr(x), _lib.ptr(self.w), _lib.ptr(self.b32),
                                      _lib.ptr(skip) if skip is not None else None, _lib.ptr(out), n, h, w, c, cout, 3, st,
                                      int(self.relu), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device)), lib)
        return out


class _Residual:
    def __init__(self, m, dtype, fused):
        self.c1 = _Conv(m.conv1, m.bn1, True, dtype, fused)
        self.c2 = _Conv(m.conv2, m.bn2, True, dtype, fused)       # ReLU after the residual sum
        self.skip = _Conv(m.skip[0], m.skip[1], False, dtype, fused) if len(m.skip) else None
        self.w_cat = None
        if self.skip is not None:  # the projection's bias rides on conv2's epilogue
            self.c2.b32 = (self.c2.b32 + self.skip.b32).contiguous()
            self.c2.b = self.c2.b32.to(dtype)
            # small levels: the 1x1 projection is appended along K of conv2 (og_conv2d_proj_bf16), one launch instead
            # of a separate 1x1 convolution + its output round trip
            if (self.c2.hip3x3 and tuple(self.c2.stride) == (1, 1) and tuple(self.skip.w.shape[2:]) == (1, 1)
                    and self.skip.w.shape[1] % 64 == 0):
                cout = self.c2.w.shape[0]
                self.w_cat = torch.cat([self.c2.w.permute(0, 2, 3, 1).reshape(cout, -1),
                                        self.skip.w.reshape(cout, -1)], 1).contiguous()
                self.c2.w_alt = self.w_cat

    def __call__(self, x, merge_up=None, after_c1=None):
        """-> the block's output; with `merge_up` (the up1 tensor of the hourglass level above, channels-last) and conv2 on
        the tiled kernel: merge_up += nearest_x2(output) in conv2's epilogue, returns None (the output is never written).
        after_c1: called once the first kernel of the block has been launched (fork point of a side branch, _Level)."""
        y = self.c1(x)
        if after_c1 is not None:
            after_c1()
        n, c, h, w = y.shape
        if self.skip is not None and self.c2.band_ok(y, x2=x, proj=self.skip):
            return self.c2.band(y, x2=x, proj=self.skip)      # the projection rides along K; its bias is in c2's
        if self.w_cat is not None and n * h * w <= CONV3X3_MAX_PIXELS and not _WHATIF:
            return self._proj(y, x)
        if self.skip is None:
            shortcut = x
        elif self.skip.pointwise_ok(x) and not _WHATIF:
            shortcut = self.skip.pointwise(x, bias=False)      # raw projection: its bias rides on conv2's epilogue
        else:
            shortcut = self.skip.raw(x)
        if merge_up is not None and self.c2.up2_ok(y):
            self.c2.up2(y, shortcut, merge_up)
            return None
        return self.c2(y, skip=shortcut)

    def _proj(self, y, x):
        n, c, h, w = y.shape
        _, c2, h2, w2 = x.shape
        cout, st2 = self.c2.w.shape[0], self.skip.stride[0]
        assert y.is_contiguous(memory_format=torch.channels_last) and x.is_contiguous(memory_format=torch.channels_last)
        lib = _lib.load()
        out = torch.empty((n, cout, h, w), dtype=y.dtype, device=y.device, memory_format=torch.channels_last)
        ws = _conv3x3_workspace(y.device, lib.og_conv2d_proj_workspace_bytes(n, h, w, c, cout, 3, 1, c2))
        self.c2.warm_next(lib)
        _lib.check(_lib.lp(lib, 'og_conv2d_proj', y.dtype)(_lib.ptr(y), _lib.ptr(self.w_cat), _lib.ptr(self.c2.b32), _lib.ptr(x), _lib.ptr(out),
                                           n, h, w, c, cout, 3, 1, h2, w2, c2, st2, 1, _lib.ptr(ws), ws.numel(),
                                           _lib.stream_ptr(y.device)), lib)
        return out


def _seq(mods, dtype, fused):
    return [_Residual(m, dtype, fused) for m in mods]


def _link(blocks):
    """Successor links along a chain of residual blocks (conv1 -> conv2 -> the next block's conv1): _Conv.next_conv."""
    for i, r in enumerate(blocks):
        r.c1.next_conv = r.c2
        if i + 1 < len(blocks):
            r.c2.next_conv = blocks[i + 1].c1


def _run(seq, x, after_first=None):
    """The blocks of `seq` in order.  after_first: called once the first kernel has been launched (fork point of a side branch,
    _Level)."""
    for i, blk in enumerate(seq):
        x = blk(x, after_c1=after_first) if (i == 0 and after_first is not None) else blk(x)
    return x


class _Level:
    def __init__(self, m, dtype, fused, depth=0):
        self.fused, self.depth, self._side = fused, depth, {}
        self.up1, self.low1, self.low3 = _seq(m.up1, dtype, fused), _seq(m.low1, dtype, fused), _seq(m.low3, dtype, fused)
        self.low2 = (_Level(m.low2, dtype, fused, depth + 1) if isinstance(m.low2, HourglassLevel)
                     else _seq(m.low2, dtype, fused))
        # the trunk's chain through this level (the up1 branch is its own chain): low1 -> [low2 ...] -> low3
        _link(self.up1)
        if isinstance(self.low2, _Level):
            _link(self.low1)
            _link(self.low3)
            self.low1[-1].c2.next_conv = self.low2.low1[0].c1
            self.low2.low3[-1].c2.next_conv = self.low3[0].c1
        else:
            _link(self.low1 + self.low2 + self.low3)

    def _lower(self, x, after_first=None):
        """-> the input of low3's LAST residual (that one runs after the join: the merge may ride on its epilogue)"""
        if not isinstance(self.low2, _Level):      # the bottom of the hourglass
            return _run(self.low1 + self.low2 + self.low3[:-1], x, after_first)
        low = _run(self.low1, x, after_first)
        low = self.low2(low)
        return _run(self.low3[:-1], low)

    def __call__(self, x):
        if BRANCHES and x.is_cuda and self.depth <= BRANCH_MAX_DEPTH:
            cur = torch.cuda.current_stream(x.device)
            # (streams of the process's own, one owner each: _lib.new_stream)
            # side streams belong to the ENGINE that issues the work (the levels themselves are shared by the engines of one model,
            # _shared_layers): two engines in flight must not serialise their branches on one stream
            key = (x.device.index, _issuer.engine)
            side = self._side.get(key)
            if side is None:
                if DEEP_SHARED and self.depth >= DEEP_SHARED:
                    if getattr(_issuer, 'deep_side', None) is None or _issuer.deep_side[0] != key:
                        _issuer.deep_side = (key, _lib.new_stream(x.device))
                    side = _issuer.deep_side[1]
                else:
                    side = _lib.new_stream(x.device)
                if len(self._side) >= 32:       # engines come and go (evaluate.run_images keeps 8 shapes x 2 lanes per model): forget the
                    # oldest entry whose engine no longer exists (the oldest entry may belong to a live engine that runs
                    # eagerly and still needs its stream; with 32 live engines the table simply grows)
                    dead = next((k for k in self._side if k[1] not in _live_engines), None)
                    if dead is not None:
                        _lib.release_stream(self._side.pop(dead))
                self._side[key] = side
            box = {}
            trunk_first = bool(TRUNK_FIRST) and self.depth >= TRUNK_FIRST
            fork_ev = None
            if trunk_first:
                fork_ev = torch.cuda.Event()
                fork_ev.record(cur)                          # the fork point: x is complete here

            def start():
                if fork_ev is not None:
                    side.wait_event(fork_ev)                 # fork: up1 only needs x
                else:
                    side.wait_stream(torch.cuda.current_stream(x.device))
                outer = _issuer.branch
                with torch.cuda.stream(side):
                    _issuer.branch = self.depth + 1
                    box['up'] = _run(self.up1, x)
                    _issuer.branch = outer
                    if DEEP_SHARED and self.depth >= DEEP_SHARED:     # a shared stream: join THIS branch, not what follows it
                        box['done'] = torch.cuda.Event()
                        box['done'].record(side)

            if trunk_first:
                low = self._lower(x, start)                  # the branch is queued behind the first kernel of the trunk below
                if 'up' not in box:
                    start()
            else:
                start()
                low = self._lower(x)
            up = box['up']
            if 'done' in box:
                cur.wait_event(box['done'])
            else:
                cur.wait_stream(side)                        # join before the merge
        else:
            low = self._lower(x)
            up = _run(self.up1, x)
        low = self.low3[-1](low, merge_up=up if self.fused and up.is_contiguous(memory_format=torch.channels_last) else None)
        if low is None:
            return up   # up += nearest_x2(low3(low)) happened in the last convolution's epilogue
        if self.fused:  # up += nearest_x2(low) in one pass
            n, c, h, w = up.shape
            lib = _lib.load()
            _lib.check(_lib.lp(lib, 'og_upsample2_add', up.dtype)(_lib.ptr(up), _lib.ptr(low), n, h, w, c, _lib.stream_ptr(up.device)), lib)
        else:
            up += F.interpolate(low, scale_factor=2, mode='nearest')
        return up


class _Layers:
    """The shape-independent part of an engine: every layer's folded weights (+ the tiled / packed images the kernels want, made
    on first use) and the head tensors.  Shared by the engines of one (model state, dtype, device, decoded stage)."""
    FIELDS = ('pre', 'kps', 'cnvs', 'inters', 'inters_', 'cnvs_', 'hm', 'off', 'scale', 'jitter', 'stem_w', 'heads_w', 'heads_b',
              'head_channels')

    def __init__(self, model, dtype, device, stage, fused):
        dev_model = model
        self.heads_b, self.head_channels = None, None
        net = model.basenet
        self.stage = stage
        # the caller's module is only read: weights are folded from it onto the engine's device (no .to(), no .eval())
        _issuer.build_device = device
        _issuer.names = {id(m): n for n, m in model.named_modules()}
        dev_model = model
        self.pre = [_Conv(net.pre[0].conv, net.pre[0].bn, True, dtype, fused), _Residual(net.pre[1], dtype, fused)]
        self.kps = [_Level(net.kps[s], dtype, fused) for s in range(self.stage + 1)]
        self.cnvs = [_Conv(net.cnvs[s].conv, net.cnvs[s].bn, True, dtype, fused) for s in range(self.stage + 1)]
        self.inters = [_Residual(net.inters[s], dtype, fused) for s in range(self.stage)]
        self.inters_ = [_Conv(net.inters_[s][0], net.inters_[s][1], True, dtype, fused) for s in range(self.stage)]
        self.cnvs_ = [_Conv(net.cnvs_[s][0], net.cnvs_[s][1], False, dtype, fused) for s in range(self.stage)]
        for a_, b_ in zip(self.inters_, self.cnvs_):  # relu(inters_(inter) + cnvs_(feat)): one epilogue, summed biases
            a_.b32 = (a_.b32 + b_.b32).contiguous()
            a_.b = a_.b32.to(dtype)
        hm_head, off_head = dev_model.headnets[0], dev_model.headnets[1]
        self.hm = _Conv(hm_head.hp_convs[self.stage], None, False, dtype, False)
        self.off = _Conv(off_head.reg_convs[self.stage], None, False, dtype, False)
        # optional keypoint-scale head (models/heads.py:112,136): third element of the offset head's output
        self.scale = (_Conv(off_head.scale_convs[self.stage], None, False, dtype, False)
                      if getattr(off_head, 'include_scale', False) else None)
        # optional jitter-offset head (models/heads.py): third element of the heatmap head's output
        self.jitter = (_Conv(hm_head.jitter_convs[self.stage], None, False, dtype, False)
                       if getattr(hm_head, 'include_jitter_offset', False) else None)
        # GPU bf16 path: the stem (7x7 stride 2, 3 -> 128) on og_stem7x7_bf16: weight as [cout][ky][8 taps][4 channels]
        self.stem_w = None
        c0 = net.pre[0].conv
        if (fused and tuple(c0.weight.shape) == (128, 3, 7, 7)
                and tuple(c0.stride) == (2, 2) and tuple(c0.padding) == (3, 3)):
            w7 = self.pre[0].w.float().permute(0, 2, 3, 1)                       # (128, ky, kx, ch), BN folded
            packed = torch.zeros((128, 7, 8, 4), dtype=torch.float32, device=w7.device)
            packed[:, :, :7, :3] = w7
            self.stem_w = packed.to(dtype).contiguous()
        # GPU bf16 path: all heads as ONE 1x1 convolution (output channels padded to a multiple of 8); the maps leave
        # through og_nhwc_bf16_to_nchw_f32 (bias added in fp32, one pass) instead of bias / cast / layout passes each
        self.heads_w, self.heads_tiled = None, None
        if fused:
            parts = [h for h in (self.hm, self.off, self.scale, self.jitter) if h is not None]
            self.head_channels = [h.w.shape[0] for h in parts]
            w = torch.cat([h.w for h in parts], 0)
            b = torch.cat([h.b32 for h in parts], 0)
            pad = (-w.shape[0]) % 8
            if pad:
                w = torch.cat([w, torch.zeros((pad,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)], 0)
                b = torch.cat([b, torch.zeros(pad, dtype=b.dtype, device=b.device)], 0)
            self.heads_w = w.contiguous(memory_format=torch.channels_last)
            self.heads_b = b.contiguous()
        _issuer.build_device = None
        _issuer.names = None


_layer_cache = {}       # key -> (weak reference to the module, _Layers); at most _LAYER_CACHE_MAX entries, oldest dropped first
_LAYER_CACHE_MAX = 2


def _model_signature(model):
    """Identity of a module's CURRENT weights: storage address, version counter, shape and a position-weighted checksum of the BIT
    PATTERNS of every parameter and buffer -- an optimizer step, load_state_dict, .to() or a write through .data changes it, so a
    cached bundle never serves stale weights.  Bit patterns, not values: a NaN parameter hashes like any other number (a float sum
    made the signature differ on every call), and a value-preserving permutation of a tensor's entries is seen.  The checksums of all
    tensors are computed where the tensors live and come back in ONE transfer (one host <-> device sync per device instead of one per
    parameter).  What no checksum can promise: a deliberate collision; callers that edit weights through .data and want certainty call
    invalidate_engine_cache().  (Callers that build several engines for one unchanged module pass the first one's bundle on,
    InferenceEngine(like=...).)"""
    meta, sums = [], {}
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            meta.append((t.data_ptr(), t._version, tuple(t.shape), str(t.dtype), str(t.device)))
            if not t.numel():
                continue
            d = t.detach().contiguous()
            bits = d.view(torch.int32) if d.element_size() == 4 else (d.view(torch.int16) if d.element_size() == 2 else
                                                                      d.view(torch.int64) if d.element_size() == 8 else d.view(torch.int8))
            bits = bits.reshape(-1).to(torch.int64)
            w = torch.arange(1, 2 * bits.numel(), 2, dtype=torch.int64, device=bits.device)      # odd weights: position matters
            sums.setdefault(d.device, []).append((bits * w).sum())                                  # (wraps modulo 2^64: fine for a checksum)
    digest = tuple(tuple(torch.stack(v).tolist()) for _, v in sorted(sums.items(), key=lambda kv: str(kv[0])))
    return hash((tuple(meta), digest))


def invalidate_engine_cache():
    """Forget every cached weight bundle: the next InferenceEngine folds the module's weights afresh (after edits through .data that a
    caller does not want to leave to the checksum)."""
    _layer_cache.clear()


def _shared_layers(model, dtype, device, stage, fused):
    import weakref
    key = (id(model), _model_signature(model), dtype, str(device), stage, fused)
    hit = _layer_cache.pop(key, None)
    if hit is not None and hit[0]() is not model:     # the id of a module that no longer exists
        hit = None
    if hit is None:
        hit = (weakref.ref(model), _Layers(model, dtype, device, stage, fused))
        while len(_layer_cache) >= _LAYER_CACHE_MAX:
            _layer_cache.pop(next(iter(_layer_cache)))
        weakref.finalize(model, _layer_cache.pop, key, None)     # the folded weights (~0.75 GB) go when the module goes
    _layer_cache[key] = hit      # (re-)inserted last: most recently used
    return hit[1]


class InferenceEngine:
    """engine = InferenceEngine(model, batch, H, W); feats = engine(images)  (images fp32 NCHW on device).

    `batch` is the number of images per call (2x the evaluation batch with flip-test).  dtype: torch.float16 (default: the
    reference evaluates in fp16 through apex O2, evaluate.py:92,198-201), torch.bfloat16 (same kernels, same MFMA rate, 8x the
    rounding error) or torch.float32 (plain torch ops: the checking path)."""

    def __init__(self, model, batch, height, width, dtype=torch.float16, device='cuda:0', feat_stage=-1,
                 use_graph=True, like=None, strict=None):
        assert height % 128 == 0 and width % 128 == 0, 'Hourglass-104 needs multiples of max_stride=128'
        global _n_engines
        self._id = _n_engines            # scratch (split-K slabs, tickets) is per engine: two engines may be in flight
        _n_engines += 1
        _live_engines[self._id] = self
        self.device = torch.device(device)
        self.dtype = dtype
        self.shape = (batch, 3, height, width)
        self.n_stacks = model.basenet.nstack
        self.stage = feat_stage % self.n_stacks
        assert isinstance(model.basenet.pre[0], ConvBlock) and isinstance(model.basenet.pre[1], Residual)
        # hand-written HIP kernels on the GPU 16-bit paths (bf16, or fp16 = the reference's apex-O2 arithmetic)
        self.fused = (self.device.type == 'cuda' and dtype in (torch.bfloat16, torch.float16))
        # strict (default on the 16-bit GPU engines; OG_ENGINE_STRICT=0 for the A/B switches that route layers to torch on purpose):
        # a convolution none of the hand-written kernels serves raises OgError naming the layer instead of running on MIOpen;
        # torch_conv_calls lists every such layer met since the engine was made (warm-up passes, capture and eager forwards)
        self.strict = (self.fused and os.environ.get('OG_ENGINE_STRICT', '1') != '0') if strict is None else bool(strict)
        self.torch_conv_calls = []
        # folded / tiled / packed weights do not depend on the input shape: engines of one (model state, dtype, device, stage)
        # share them (evaluate.run_images builds one engine per input shape, --fixed-height: one per width)
        # like = an engine the caller built for this very module a moment ago (evaluate.run_images: one per shape and lane): its
        # bundle is taken as it is, without the 0.2 s look at every weight that decides whether a cached bundle is still current
        if (like is not None and like._model() is model and like.dtype == dtype and like.device == self.device
                and like.stage == self.stage):
            self._layers = like._layers
        else:
            self._layers = _shared_layers(model, dtype, self.device, self.stage, self.fused)
        import weakref
        self._model = weakref.ref(model)
        for name in _Layers.FIELDS:
            setattr(self, name, getattr(self._layers, name))
        self._ws = {}                    # this engine's convolution scratch (_conv3x3_workspace)
        self._static_in = torch.zeros(self.shape, dtype=torch.float32, device=self.device)
        self._graph = None
        self._out = None
        if use_graph:
            self._capture()

    def _forward(self, images):
        _issuer.engine, _issuer.ws, _issuer.eng = self._id, self._ws, self
        try:
            return self._forward_impl(images)
        finally:
            _issuer.eng = None

    def _forward_impl(self, images):
        if self.stem_w is not None:   # fused stem: fp32 NCHW images -> conv 7x7 s2 + BN + ReLU -> bf16 NHWC
            lib = _lib.load()
            images = images.float().contiguous()
            n, c, h, w = images.shape
            x = torch.empty((n, 128, h // 2, w // 2), dtype=self.dtype, device=images.device, memory_format=torch.channels_last)
            _lib.check(_lib.lp(lib, 'og_stem7x7', self.dtype)(_lib.ptr(images), _lib.ptr(self.stem_w), _lib.ptr(self.pre[0].b32), _lib.ptr(x),
                                           n, h, w, 1, _lib.stream_ptr(images.device)), lib)
            inter = self.pre[1](x)
        else:
            if self.fused:   # fp32 NCHW -> bf16 NHWC in one pass
                lib = _lib.load()
                images = images.float().contiguous()
                n, c, h, w = images.shape
                x = torch.empty((n, c, h, w), dtype=self.dtype, device=images.device, memory_format=torch.channels_last)
                _lib.check(_lib.lp(lib, 'og_nchw_f32_to_nhwc', self.dtype)(_lib.ptr(images), _lib.ptr(x), n, c, h, w, _lib.stream_ptr(images.device)), lib)
            else:
                x = images.to(self.dtype).contiguous(memory_format=torch.channels_last)
            inter = _run(self.pre, x)
        feat = None
        for s in range(self.stage + 1):
            feat = self.cnvs[s](self.kps[s](inter))
            if s < self.stage:
                if self.inters_[s].pointwise_ok(inter) and feat.shape == inter.shape and not _WHATIF:
                    # relu(inters_(inter) + cnvs_(feat)) as ONE 1x1 convolution over the concatenated channels
                    inter = _run([self.inters[s]], self.inters_[s].pointwise(inter, x2=feat, other=self.cnvs_[s]))
                else:
                    inter = _run([self.inters[s]], self.inters_[s](inter, skip=self.cnvs_[s].raw(feat)))
        if (self.heads_w is not None and (CONV_TILED & 4) and feat.shape[1] % 64 == 0 and len(self.head_channels) <= 4
                and feat.is_contiguous(memory_format=torch.channels_last)):
            lib = _lib.load()
            n, c, h, w = feat.shape
            if self._layers.heads_tiled is None:
                assert not torch.cuda.is_current_stream_capturing(), 'weights must be tiled before graph capture'
                cout = (sum(self.head_channels) + 63) // 64 * 64
                wpad = torch.zeros((cout, c), dtype=self.heads_w.dtype, device=self.heads_w.device)
                wpad[:self.heads_w.shape[0]] = self.heads_w.reshape(self.heads_w.shape[0], -1)
                bpad = torch.zeros(cout, dtype=torch.float32, device=self.heads_w.device)
                bpad[:self.heads_b.shape[0]] = self.heads_b
                packed = torch.empty(wpad.numel(), dtype=wpad.dtype, device=wpad.device)
                _lib.check(lib.og_conv3x3_pack_w16(_lib.ptr(wpad), c, cout, 3, _lib.ptr(packed), _lib.stream_ptr(feat.device)), lib)
                self._layers.heads_tiled = (packed, bpad, cout)
            packed, bpad, cout = self._layers.heads_tiled
            outs = [torch.empty((n, ch, h, w), dtype=torch.float32, device=feat.device) for ch in self.head_channels]
            import ctypes as C
            chans = (C.c_int * len(outs))(*self.head_channels)
            ptrs = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
            _lib.check(_lib.lp(lib, 'og_conv1x1_heads', feat.dtype)(_lib.ptr(feat), c, _lib.ptr(packed), _lib.ptr(bpad), n, h, w, cout,
                                                                len(outs), chans, ptrs, _lib.stream_ptr(feat.device)), lib)
            return tuple(outs)
        if self.heads_w is not None:
            lib = _lib.load()
            y = _torch_conv('heads', feat, self.heads_w)
            n, cc, h, w = y.shape
            outs, c0 = [], 0
            for ch in self.head_channels:
                o = torch.empty((n, ch, h, w), dtype=torch.float32, device=y.device)
                _lib.check((lib.og_nhwc_f16_to_nchw_f32 if y.dtype == torch.float16 else lib.og_nhwc_bf16_to_nchw_f32)(_lib.ptr(y), cc, c0, ch, _lib.ptr(self.heads_b), _lib.ptr(o), n, h, w,
                                                        _lib.stream_ptr(y.device)), lib)
                outs.append(o)
                c0 += ch
            return tuple(outs)
        hm = self.hm(feat).float().contiguous(memory_format=torch.contiguous_format)
        off = self.off(feat).float().contiguous(memory_format=torch.contiguous_format)
        extra = tuple(h(feat).float().contiguous(memory_format=torch.contiguous_format)
                      for h in (self.scale, self.jitter) if h is not None)
        return (hm, off) + extra

    def _capture(self):
        side = _lib.new_stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):                      # warm-up: weight tiling / packing, workspaces, allocator
                self._forward(self._static_in)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        # the warm-up stream goes back with the side streams of the 32nd engine before this one (every engine MAKES eight streams: that
        # keeps each engine's branches on the hardware queues the first engine's sit on -- _lib.new_stream)
        _warm_streams[(self.device.index, self._id)] = side
        if len(_warm_streams) > 32:
            _lib.release_stream(_warm_streams.pop(next(iter(_warm_streams))))
        self._graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self._graph):
            self._out = self._forward(self._static_in)

    @torch.no_grad()
    def forward_raw(self, images):
        """(hm (N,17,h,w), off (N,38,h,w)[, scale (N,17,h,w)][, jitter (N,2,h,w)]) fp32 NCHW; graph outputs are reused by
        the next call."""
        assert tuple(images.shape) == self.shape, f'engine built for {self.shape}, got {tuple(images.shape)}'
        if self._graph is None:
            return self._forward(images.to(self.device))
        self._static_in.copy_(images, non_blocking=True)
        self._graph.replay()
        return self._out

    def __call__(self, images):
        hm, off, *rest = self.forward_raw(images)
        hms, offs = [None] * self.n_stacks, [None] * self.n_stacks
        hms[self.stage], offs[self.stage] = hm, off
        empty = [[] for _ in range(self.n_stacks)]
        scales, jitters = list(empty), list(empty)
        if self.scale is not None:
            scales = [None] * self.n_stacks
            scales[self.stage] = rest.pop(0)
        if self.jitter is not None:
            jitters = [None] * self.n_stacks
            jitters[self.stage] = rest.pop(0)
        return [(hms, list(empty), jitters), (offs, list(empty), scales)]
