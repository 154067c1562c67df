"""ctypes binding of libog_decoder.so (include/og_decoder.h) + torch plumbing.

There is NO fallback: if the shared library is missing or a tensor is not on a HIP device the
call raises.  PyTorch is used only for device memory, streams and (elsewhere) the dense conv
backbone; every decoder stage below is a hand-written HIP kernel.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libog_decoder.so")
if os.environ.get("OG_DECODER_LIB"):  # alternative build of the same ABI (kernel A/B experiments)
    LIB_PATH = os.environ["OG_DECODER_LIB"]

OG_OK, OG_EINVAL, OG_ENOSPC, OG_EHIP, OG_EUNSUPPORTED = 0, -1, -2, -3, -4
ABI_VERSION = 3

_vp, _i, _l, _f, _d, _sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); mirrors include/og_decoder.h one to one
SIGNATURES = {
    "og_abi_version": (_i, []),
    "og_last_error": (C.c_char_p, []),
    "og_device_count": (_i, []),
    "og_upsample_bicubic4_f32": (_i, [_vp, _l, _i, _i, _vp, _vp]),
    "og_upsample_bilinear4_f32": (_i, [_vp, _l, _i, _i, _vp, _vp]),
    "og_upsample_bicubic4_flip_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "og_generate_limbs_flip_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "og_hmp_nms_f32": (_i, [_vp, _l, _i, _i, _vp, _vp]),
    "og_topk_channel_f32": (_i, [_vp, _l, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "og_nms_topk_f32": (_i, [_vp, _l, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "og_generate_limbs_fused_f32": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "og_generate_limbs_fused_flip_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "og_upsample_nms_topk_f32": (_i, [_vp, _l, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "og_topk_workspace_bytes": (_sz, [_l, _i, _i, _i]),
    "og_collect_limbs_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp]),
    "og_collect_limbs_nd_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp]),
    "og_collect_limbs_ex_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp]),
    "og_collect_limbs_full_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp]),
    "og_generate_limbs_f32": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp, _vp,
                                   _i, _vp, _sz, _vp]),
    "og_generate_limbs_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "og_greedy_group_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _d, _f, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "og_group_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "og_flip_merge_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "og_flip_cat_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "og_bias_act_bf16": (_i, [_vp, _vp, _vp, _l, _i, _i, _vp]),
    "og_upsample2_add_bf16": (_i, [_vp, _vp, _l, _i, _i, _i, _vp]),
    "og_nchw_f32_to_nhwc_bf16": (_i, [_vp, _vp, _l, _i, _i, _i, _vp]),
    "og_nhwc_bf16_to_nchw_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _l, _i, _i, _vp]),
    "og_center_pad_normalize_u8": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "og_resize_cubic_u8": (_i, [_vp, _i, _i, _vp, _i, _i, _vp]),
    "og_shrink_mask_miss_u8": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "og_rescale_pad_normalize_u8": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "og_rescale_pad_normalize_batch_u8": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "og_stem7x7_bf16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "og_conv3x3_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "og_conv3x3_workspace_bytes": (_sz, [_l, _i, _i]),
    "og_conv3x3_workspace_bytes_nhw": (_sz, [_i, _i, _i, _i, _i]),
    "og_conv2d_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "og_conv2d_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "og_conv2d_proj_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "og_conv2d_proj_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "og_conv_band_supported": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "og_conv_band_pack_w16": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "og_conv_band_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "og_conv3x3_tiled_supported": (_i, [_i, _i, _i, _i, _i]),
    "og_conv3x3_pack_w16": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "og_conv3x3s2_tiled_supported": (_i, [_i, _i, _i, _i, _i]),
    "og_conv1x1_tiled_bf16": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "og_conv1x1_heads_bf16": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "og_conv3x3s2_tiled_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "og_conv3x3_tiled_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "og_conv_next_weights_hint": (None, [_vp, _sz]),
    "og_conv3x3_tiled_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "og_conv3x3_tiled_up2_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "og_encode_heatmaps_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "og_encode_jitter_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "og_encode_offsets_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    "og_focal_l2_loss_f32": (_i, [_vp, _vp, _vp, _i, _i, _l, _f, _f, _vp, _vp, _vp]),
    "og_offset_l1_loss_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _l, _f, _i, _vp, _vp, _vp]),
}

# fp16 twins of the 16-bit-type specific entry points (csrc/lp_dtype.h): same signatures
for _name in [n for n in SIGNATURES if n.endswith('_bf16')]:
    SIGNATURES[_name[:-5] + '_f16'] = SIGNATURES[_name]
SIGNATURES['og_nhwc_f16_to_nchw_f32'] = SIGNATURES['og_nhwc_bf16_to_nchw_f32']

_lib = None


def lp(lib, stem, dtype):
    """Entry point of the 16-bit-type specific family for a torch dtype: lp(lib, 'og_conv2d', torch.float16) -> og_conv2d_f16."""
    import torch as _t
    return getattr(lib, stem + ('_f16' if dtype == _t.float16 else '_bf16'))


class OgError(RuntimeError):
    pass


def load():
    """Load (once) and type the C ABI.  Raises if the library is absent -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m offsetguided_amd.build` "
                "(hipcc --offload-arch=gfx950).  offsetguided_amd has no CPU/torch fallback.")
        lib = C.CDLL(LIB_PATH)
        # the version first: a stale library would otherwise fail on a missing symbol with a bare AttributeError
        lib.og_abi_version.restype, lib.og_abi_version.argtypes = _i, []
        if lib.og_abi_version() != ABI_VERSION:
            raise ImportError(f"{LIB_PATH}: ABI version {lib.og_abi_version()} != {ABI_VERSION}; rebuild with "
                              "`python -m offsetguided_amd.build --force`")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc, lib=None):
    if rc != OG_OK:
        msg = (lib or load()).og_last_error().decode(errors="replace")
        raise OgError(f"libog_decoder: {msg} (code {rc})")


def require_device(t, name, dtype=torch.float32):
    """Contiguous device tensor of `dtype` (a detached copy when a conversion is needed)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise OgError(f"{name}: tensor is on {t.device}; the HIP decoder kernels need a GPU tensor "
                      "(offsetguided_amd has no CPU path)")
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr())


_workspaces = {}


def workspace(device, nbytes, tag="ws"):
    """Per-(device, stream, tag) scratch buffer, grown geometrically, never shrunk."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, tag)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.zeros(max(int(nbytes * 1.25), 1 << 16), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


_own_streams = {}


def dedicated_stream(device, key):
    """A HIP stream of this process's own for (device, key) -- hipStreamCreateWithFlags(non-blocking), wrapped as
    torch.cuda.ExternalStream, created on first request and kept.  torch.cuda.Stream() hands out a POOL of 32 streams in turn: after
    enough engines (seven side streams each), copy streams and capture streams have been made, a "new" torch.cuda.Stream() IS one of
    those, and a graph launch on a lane that also sits in that graph's capture history crashed the runtime (segfault in
    hipGraphLaunch, round 5: run_images with two lanes after a few engine builds).  The lanes, the decoder's grouping streams and the
    input-chain streams come from here.  (The engines' capture-time side streams stay torch pool streams: which of them end up on
    which hardware queue decides 3 % of the step, and the pool's order is the measured one -- profiles/r05_inflight_streams.log.)"""
    k = (device.index, key)
    st = _own_streams.get(k)
    if st is None:
        hip = _hip_runtime()
        with torch.cuda.device(device):
            h = C.c_void_p()
            rc = hip.hipStreamCreateWithFlags(C.byref(h), C.c_uint(1))      # hipStreamNonBlocking
            if rc != 0:
                raise OgError(f'hipStreamCreateWithFlags failed ({rc})')
            st = _own_streams[k] = torch.cuda.ExternalStream(h.value, device=device)
    return st


_free_streams = {}


def new_stream(device):
    """A HIP stream of this process's own for ONE owner at a time (an engine's side branch during its capture): like dedicated_stream,
    but handed out per call; release_stream() puts its handle on a free list that serves the next caller first -- torch.cuda.Stream()'s
    pool of 32 repeats whoever still uses what.  Streams are created in the order they are asked for, as the pool's were: which
    capture-time stream sits on which hardware queue is worth 3 % of the step (profiles/r05_inflight_streams.log).  (No finalizer on
    the wrapper objects: weakref.finalize on them crashed the interpreter at exit.)"""
    free = _free_streams.setdefault(device.index, set())
    if free:
        handle = free.pop()
    else:
        hip = _hip_runtime()
        with torch.cuda.device(device):
            h = C.c_void_p()
            rc = hip.hipStreamCreateWithFlags(C.byref(h), C.c_uint(1))      # hipStreamNonBlocking
            if rc != 0:
                raise OgError(f'hipStreamCreateWithFlags failed ({rc})')
            handle = h.value
    return torch.cuda.ExternalStream(handle, device=device)


def release_stream(stream):
    """The owner is done with a new_stream() stream (work already queued on it stays ordered: the next owner queues behind it)."""
    _free_streams.setdefault(stream.device.index, set()).add(stream.cuda_stream)


def lane_streams(device, n):
    """The n streams batches in flight run on (evaluate.run_images, bench.py): dedicated_stream(device, ('lane', i))."""
    return [dedicated_stream(device, ('lane', i)) for i in range(n)]


_const_cache = {}


def int_table(values, device):
    """Small int32 device table (skeleton / permutation), cached per device."""
    key = (tuple(int(v) for v in values), device.index)
    t = _const_cache.get(key)
    if t is None:
        t = torch.tensor(key[0], dtype=torch.int32, device=device)
        _const_cache[key] = t
    return t


# ---- optional per-stage device timing (bench.py) -------------------------------------------
_profile = None  # dict name -> list of (start_event, end_event) while enabled
_hip = None


def _hip_runtime():
    """The HIP runtime torch already loaded (same streams / events), for the calls torch does not expose."""
    global _hip
    if _hip is None:
        paths = [l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l]
        if not paths:
            raise OgError('libamdhip64 is not loaded: no HIP device runtime in this process')
        _hip = C.CDLL(paths[0])
    return _hip


class TimingEvent:
    """HIP event for timing only, created with hipEventDisableSystemFence: a default event performs a system-scope
    fence (L2 write-back + invalidate) when it is recorded, ~5 us each on the stream it times and a cold L2 for the
    kernel behind it.  elapsed_time() needs the stream (or device) synchronised first."""
    DISABLE_SYSTEM_FENCE = 0x20000000

    def __init__(self):
        self.h = C.c_void_p()
        rc = _hip_runtime().hipEventCreateWithFlags(C.byref(self.h), C.c_uint(self.DISABLE_SYSTEM_FENCE))
        if rc != 0:
            raise OgError(f'hipEventCreateWithFlags failed ({rc})')

    def record(self, stream):
        rc = _hip_runtime().hipEventRecord(self.h, C.c_void_p(stream.cuda_stream))
        if rc != 0:
            raise OgError(f'hipEventRecord failed ({rc})')

    def elapsed_time(self, end):
        ms = C.c_float()
        rc = _hip_runtime().hipEventElapsedTime(C.byref(ms), self.h, end.h)
        if rc != 0:
            raise OgError(f'hipEventElapsedTime failed ({rc})')
        return ms.value

    def __del__(self):
        try:
            if self.h and _hip is not None:
                _hip.hipEventDestroy(self.h)
        except Exception:  # noqa: BLE001  interpreter shutdown
            pass


class stage_timer:
    """`with stage_timer('nms_topk', device):` records HIP events on the launch stream when
    profiling is enabled (profile_start()); free otherwise."""

    def __init__(self, name, device):
        self.name, self.device = name, device

    def __enter__(self):
        if _profile is not None:
            self.start = TimingEvent()
            self.start.record(torch.cuda.current_stream(self.device))

    def __exit__(self, *exc):
        if _profile is not None:
            end = TimingEvent()
            end.record(torch.cuda.current_stream(self.device))
            _profile.setdefault(self.name, []).append((self.start, end))


def profile_start():
    global _profile
    _profile = {}


def profile_stop():
    """-> {stage: [durations in microseconds]} (synchronises)."""
    global _profile
    torch.cuda.synchronize()
    out = {k: [a.elapsed_time(b) * 1e3 for a, b in v] for k, v in (_profile or {}).items()}
    _profile = None
    return out
