"""Keypoint candidate detection: 3x3 max-NMS and per-channel top-k on hi-res heatmaps.

API of the reference's decoder/heatmap.py (hmp_NMS :15-35, topK_channel :38-49, joint_dets
:52-59); layout is NCHW (the reference docstring says NHWC but its code pools the last two
dims).  `topk_ys` uses floor division -- the reference pins torch 1.3.1, where int64 / int
floors, and its own demo indexes arrays with the result (demo_batch.py:236-237).
"""
import torch

from .. import _lib


def _planes(t):
    n, c, h, w = t.shape
    return n * c, h, w


def hmp_NMS(heat, kernel=3):
    """Keep kernel x kernel local maxima (zero padding), zero the rest: heat * (maxpool(heat) == heat).

    kernel = 3 (the decoder's window, decoder/heatmap.py:15) is the hand-written HIP kernel; any other odd window runs the
    reference's own three device ops (pad, max_pool2d, multiply: exact comparisons, so the result is the reference's bit for
    bit); an even window fails as it does there (the padded pool no longer returns the input's size)."""
    heat = _lib.require_device(heat, "hmp_NMS(heat)")
    if kernel != 3:
        import torch.nn.functional as F
        pad = (kernel - 1) // 2
        hmax = F.max_pool2d(F.pad(heat, [pad] * 4), (kernel, kernel), stride=1, padding=0)
        return heat * (hmax == heat).float()
    lib = _lib.load()
    planes, h, w = _planes(heat)
    out = torch.empty_like(heat)
    _lib.check(lib.og_hmp_nms_f32(_lib.ptr(heat), planes, h, w, _lib.ptr(out), _lib.stream_ptr(heat.device)), lib)
    return out


def _topk(entry, scores, K):
    out_s, out_i = _topk_raw(entry, scores, K)
    w = scores.shape[-1]
    return out_s, out_i, torch.div(out_i, w, rounding_mode='floor'), out_i % w


def nms_topk_raw(hmps, k):
    """joint_dets without the derived ys/xs tensors: (scores, flat idx), each (N, C, k)."""
    return _topk_raw("og_nms_topk_f32", hmps, k)


def joint_dets_lowres(hmps_lr, k):
    """joint_dets(F.interpolate(hmps_lr, scale_factor=4, mode='bicubic'), k) without ever building the
    hi-res tensor (decoder/factory.py:74-75 + heatmap.py:52-59 in one kernel); indices refer to the
    (4h, 4w) grid."""
    out_s, out_i = _topk_raw("og_upsample_nms_topk_f32", hmps_lr, k, scale=4)
    w = 4 * hmps_lr.shape[-1]
    return out_s, out_i, torch.div(out_i, w, rounding_mode='floor'), out_i % w


def _topk_raw(entry, scores, K, scale=1):
    scores = _lib.require_device(scores, "scores")
    lib = _lib.load()
    n, c, h, w = scores.shape
    if K > h * w * scale * scale:
        raise RuntimeError("selected index k out of range")  # torch.topk's message
    dev = scores.device
    out_s = torch.empty((n, c, K), dtype=torch.float32, device=dev)
    out_i = torch.empty((n, c, K), dtype=torch.int64, device=dev)
    nbytes = lib.og_topk_workspace_bytes(n * c, h * scale, w * scale, K)
    ws = _lib.workspace(dev, nbytes, "topk")
    fn = getattr(lib, entry)
    _lib.check(fn(_lib.ptr(scores), n * c, h, w, K, _lib.ptr(out_s), _lib.ptr(out_i), _lib.ptr(ws), ws.numel(),
                  _lib.stream_ptr(dev)), lib)
    return out_s, out_i


def topK_channel(scores, K=40):
    """Top-K responses of every (n, c) plane: (scores, flat idx, ys, xs), each (N, C, K)."""
    return _topk("og_topk_channel_f32", scores, K)


def joint_dets(hmps, k):
    """topK_channel(hmp_NMS(hmps), k) in ONE pass over the heatmaps (the NMS map is never stored)."""
    return _topk("og_nms_topk_f32", hmps, k)
