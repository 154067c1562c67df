"""Training losses of the heatmap / offset heads (reference models/losses.py:13-299).

Two implementations of the same maths:
  * the plain torch formulation below (boolean-mask gathers, like the reference) -- the
    specification, used on CPU and as the numerics reference;
  * fused HIP kernels (csrc/losses.hip: og_focal_l2_loss_f32 / og_offset_l1_loss_f32), selected with
    `fused=True` on CUDA tensors: one pass over (pred, gt, mask) computes the masked loss SUM and
    writes the gradient, instead of ~6 full-tensor passes + gathers for forward and as many for
    backward.

Element-wise definitions:
  focal_l2(s, s*) = 0.5 (s - s*)^2 |1 - st|^gamma,  st = s if s* >= tau else 1 - s          (:31-36)
  masked by mask_miss and isfinite(gt) (:39-58); heatmap head: sum() * stack_weight, / batch  (:141-197)
  offsets: |pred/ps - gt/ps| (instance-normalised L1, :87-92), entries < MARGIN dropped,
           optional sqrt, sum / (1 + count) * stack_weight, / batch                           (:200-256)
"""
import logging
import re

import torch

LOG = logging.getLogger(__name__)

TAU = 0.01      # fore/background threshold of the focal L2 loss
GAMMA = 1       # focal exponent
MARGIN = 1e-5   # offset errors below this are not punished
MARGIN2 = 0.1   # same for keypoint scales


def l1(x, t):
    return (x - t).abs()


def l2(x, t):
    return 0.5 * (x - t) ** 2


def laplace(norm, logb):
    return logb + norm * torch.exp(-logb)


def focal_l2(s, sxing, tau=None, gamma=None):
    tau = TAU if tau is None else tau
    gamma = GAMMA if gamma is None else gamma
    st = torch.where(sxing >= tau, s, 1. - s)
    return 0.5 * (s - sxing) ** 2 * (1. - st).abs() ** gamma


def tensor_loss(pred, gt, mask_miss, fun):
    """fun(pred, gt) on the labelled (mask_miss) and finite-target elements, as a flat tensor."""
    keep = mask_miss.expand_as(gt)
    p, g = pred[keep], gt[keep]
    ok = torch.isfinite(g)
    return fun(p[ok], g[ok])


class LossChoice(object):
    @staticmethod
    def l2_loss(pred, gt, mask_miss):
        return tensor_loss(pred, gt, mask_miss, l2)

    @staticmethod
    def focal_l2_loss(pred, gt, mask_miss):
        return tensor_loss(pred, gt, mask_miss, focal_l2)

    @staticmethod
    def scale_l1_loss(pred, gt, mask_miss):
        return tensor_loss(pred, gt, mask_miss, l1)

    @staticmethod
    def offset_l1_loss(pred, gt, __, _, mask_miss):
        return tensor_loss(pred, gt, mask_miss, l1)

    @staticmethod
    def offset_instance_l1_loss(pred, gt_off, gt_ps, _, mask_miss):
        return tensor_loss(pred / gt_ps, gt_off / gt_ps, mask_miss, l1)

    @staticmethod
    def vector_l1_loss(pred, gt_off, __, _, mask_miss):
        n, _, h, w = pred.shape
        norm = (pred - gt_off).view(n, -1, 2, h, w).norm(dim=2)
        norm = norm[mask_miss.expand_as(norm)]
        return norm[torch.isfinite(norm)]

    @staticmethod
    def offset_laplace_loss(pred, gt_off, _, logb, mask_miss):
        n, _, h, w = pred.shape
        norm = (pred - gt_off).view(n, -1, 2, h, w).norm(dim=2)
        keep = mask_miss.expand_as(norm)
        norm, logb = norm[keep], logb[keep]
        ok = torch.isfinite(norm)
        return laplace(norm[ok], logb[ok])


class _FocalL2Sum(torch.autograd.Function):
    """sum over labelled, finite elements of focal_l2(pred, gt): fused HIP forward + gradient."""

    @staticmethod
    def forward(ctx, pred, gt, mask_miss, tau, gamma):
        from .. import _lib
        lib = _lib.load()
        pred32 = _lib.require_device(pred, 'pred')
        gt = _lib.require_device(gt, 'gt')
        mask = mask_miss.to(torch.uint8).contiguous()
        n, c, h, w = pred32.shape
        grad = torch.empty_like(pred32)
        total = torch.zeros(1, dtype=torch.float32, device=pred32.device)
        _lib.check(lib.og_focal_l2_loss_f32(_lib.ptr(pred32), _lib.ptr(gt), _lib.ptr(mask), n, c, h * w, float(tau),
                                            float(gamma), _lib.ptr(total), _lib.ptr(grad), _lib.stream_ptr(pred32.device)), lib)
        ctx.save_for_backward(grad)
        ctx.in_dtype = pred.dtype
        return total[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).to(ctx.in_dtype), None, None, None, None


class _OffsetL1Mean(torch.autograd.Function):
    """sum(e) / (1 + count(e)) over labelled, finite elements with e = |pred - gt| / ps >= margin
    (e -> sqrt(e) when sqrt_re): fused HIP forward + gradient."""

    @staticmethod
    def forward(ctx, pred, gt, gt_ps, mask_miss, margin, sqrt_re):
        from .. import _lib
        lib = _lib.load()
        pred32 = _lib.require_device(pred, 'pred')
        gt = _lib.require_device(gt, 'gt')
        ps = _lib.require_device(gt_ps.expand_as(gt) if gt_ps is not None else torch.ones_like(gt), 'gt_ps')
        mask = mask_miss.to(torch.uint8).contiguous()
        n, c, h, w = pred32.shape
        grad = torch.empty_like(pred32)
        acc = torch.zeros(2, dtype=torch.float32, device=pred32.device)     # [sum, count]
        _lib.check(lib.og_offset_l1_loss_f32(_lib.ptr(pred32), _lib.ptr(gt), _lib.ptr(ps), _lib.ptr(mask), n, c, h * w,
                                             float(margin), int(bool(sqrt_re)), _lib.ptr(acc), _lib.ptr(grad),
                                             _lib.stream_ptr(pred32.device)), lib)
        denom = 1.0 + acc[1]
        ctx.save_for_backward(grad, denom)
        ctx.in_dtype = pred.dtype
        return acc[0] / denom

    @staticmethod
    def backward(ctx, g):
        grad, denom = ctx.saved_tensors
        return (grad * (g / denom)).to(ctx.in_dtype), None, None, None, None, None


def _margin_mean(err, margin, sqrt_re):
    err = err[err >= margin]
    if sqrt_re:
        err = torch.sqrt(err)
    return err.sum() / (1 + float(err.numel()))


class HeatMapsLoss(object):
    """(keypoint heatmap loss, background loss, jitter-offset loss), each already / batch."""

    def __init__(self, head_name, n_stacks, stack_weights, hmp_loss, jomp_loss, sqrt_re=False, fused=False):
        self.head_name = head_name + '_loss'
        self.n_stacks = n_stacks
        assert len(stack_weights) >= n_stacks, type(stack_weights)
        self.stack_weights = [w / sum(stack_weights) for w in stack_weights]
        self.hmp_loss = hmp_loss
        self.jomp_loss = jomp_loss
        self.sqrt_re = sqrt_re
        self.fused = fused

    def _hmp_sum(self, pred, gt, mask_miss):
        if self.fused and pred.is_cuda and self.hmp_loss is LossChoice.focal_l2_loss:
            return _FocalL2Sum.apply(pred, gt, mask_miss, TAU, GAMMA)
        return self.hmp_loss(pred, gt, mask_miss).sum()

    def __call__(self, pred_hpms, gt_hpm, gt_bghmp, gt_jomp, mask_miss):
        assert len(pred_hpms[0]) == self.n_stacks, 'BaseNet mismatches HeadNet'
        batch_size = gt_hpm.shape[0]
        hmps, bg_hmps, jomps = pred_hpms
        out1, out2, out3 = [], [], []
        for w, hmp, bg_hmp, jomp in zip(self.stack_weights, hmps, bg_hmps, jomps):
            out1.append(self._hmp_sum(hmp, gt_hpm, mask_miss) * w)
            if len(bg_hmp) > 0:
                out2.append(self._hmp_sum(bg_hmp, gt_bghmp, mask_miss) * w)
            if len(jomp) > 0:
                out3.append(_margin_mean(self.jomp_loss(jomp, gt_jomp, None, None, mask_miss), MARGIN, self.sqrt_re) * w)
        return sum(out1) / batch_size, sum(out2) / batch_size, sum(out3) / batch_size


class OffsetMapsLoss(object):
    """(guiding-offset loss, keypoint-scale loss), each already / batch."""

    def __init__(self, head_name, n_stacks, stack_weights, off_loss, s_loss, sqrt_re=False, fused=False):
        assert len(stack_weights) >= n_stacks, type(stack_weights)
        self.head_name = head_name + '_loss'
        self.n_stacks = n_stacks
        self.stack_weights = [w / sum(stack_weights) for w in stack_weights]
        self.off_loss = off_loss
        self.s_loss = s_loss
        self.sqrt_re = sqrt_re
        self.fused = fused

    def _off_mean(self, pred, gt_off, gt_ps, spread, mask_miss):
        if self.fused and pred.is_cuda and self.off_loss in (LossChoice.offset_l1_loss, LossChoice.offset_instance_l1_loss):
            ps = gt_ps if self.off_loss is LossChoice.offset_instance_l1_loss else None
            return _OffsetL1Mean.apply(pred, gt_off, ps, mask_miss, MARGIN, self.sqrt_re)
        return _margin_mean(self.off_loss(pred, gt_off, gt_ps, spread, mask_miss), MARGIN, self.sqrt_re)

    def __call__(self, preds, gt_off, gt_s, gt_ps, mask_miss):
        assert len(preds[0]) == self.n_stacks
        batch_size = gt_off.shape[0]
        out1, out2 = [], []
        for w, pred_off, pred_spread, pred_s in zip(self.stack_weights, *preds):
            out1.append(self._off_mean(pred_off, gt_off, gt_ps, pred_spread, mask_miss) * w)
            if len(pred_s) > 0:
                out2.append(_margin_mean(self.s_loss(pred_s, gt_s, mask_miss), MARGIN2, self.sqrt_re) * w)
        return sum(out1) / batch_size, sum(out2) / batch_size


def factory_loss(head_name, n_stacks, stack_weights, hmp_loss, jomp_loss, off_loss, s_loss, sqrt_re, fused=False):
    if head_name in ('hmp', 'hmps', 'heatmap', 'heatmaps') or re.match('hmp[s]?([0-9]+)$', head_name):
        return HeatMapsLoss(head_name, n_stacks, stack_weights, hmp_loss, jomp_loss, sqrt_re, fused)
    if head_name in ('omp', 'omps', 'offset', 'offsets') or re.match('omp[s]?([0-9]+)$', head_name):
        return OffsetMapsLoss(head_name, n_stacks, stack_weights, off_loss, s_loss, sqrt_re, fused)
    raise Exception('unknown head to create a lossnet: {}'.format(head_name))


def lossfuncs_factory(headnames, n_stacks, stack_weights, heatmap_loss, jitter_offset_loss, offset_loss, scale_loss,
                      sqrt_re, fused=False):
    choose = lambda name: getattr(LossChoice, name)  # noqa: E731
    return [factory_loss(h, n_stacks, stack_weights, choose(heatmap_loss), choose(jitter_offset_loss), choose(offset_loss),
                         choose(scale_loss), sqrt_re, fused) for h in headnames]
