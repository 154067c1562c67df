// Fused training losses (reference models/losses.py:31-58, :141-256), forward value + gradient in
// ONE pass over (pred, gt, mask_miss).
//
// The reference evaluates each loss with boolean-mask gathers (pred[mask], gt[mask],
// isfinite, where, pow, mul, sum): ~6 full-tensor passes forward and as many backward on
// (N,17+38,128,128) maps.  Here a lane reads 16 B of pred/gt (+ the per-pixel mask byte), accumulates
// the masked loss in fp32, writes d(loss)/d(pred) directly, and the sum leaves through one wave
// reduction + one float atomic per wave (sums are order dependent in the last bits, like torch's).
//   focal_l2:  0.5 (s-s*)^2 |1-st|^g,  st = s if s* >= tau else 1-s
//              d/ds = (s-s*) |1-st|^g + 0.5 (s-s*)^2 g |1-st|^(g-1) * d|1-st|/ds
//   offset l1: e = |p-g| / ps, kept if e >= margin; optional sqrt(e)
//              d/dp = sign(p-g)/ps   (x 0.5/sqrt(e) with sqrt)   -- the caller divides by (1+count)
#include <math.h>

#include "og_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ void __launch_bounds__(256)
focal_l2_kernel(const float *__restrict__ pred, const float *__restrict__ gt, const unsigned char *__restrict__ mask,
                int C, long hw, long total, float tau, float gamma, float *__restrict__ sum, float *__restrict__ grad)
{
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / ((long)C * hw), pix = i % hw;
        const float s = pred[i], t = gt[i];
        float g = 0.f;
        if (mask[n * hw + pix] && isfinite(t)) {
            const bool fg = t >= tau;
            const float om = fg ? 1.f - s : s;          // 1 - st
            const float a = fabsf(om);
            const float d = s - t;
            const float f = (gamma == 1.f) ? a : powf(a, gamma);
            acc += 0.5f * d * d * f;
            // d|1-st|/ds = sign(om) * (fg ? -1 : +1)
            const float da = (om > 0.f ? 1.f : (om < 0.f ? -1.f : 0.f)) * (fg ? -1.f : 1.f);
            const float df = (gamma == 1.f) ? da : gamma * powf(a, gamma - 1.f) * da;
            g = d * f + 0.5f * d * d * df;
        }
        grad[i] = g;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, acc);
}

__global__ void __launch_bounds__(256)
offset_l1_kernel(const float *__restrict__ pred, const float *__restrict__ gt, const float *__restrict__ ps,
                 const unsigned char *__restrict__ mask, int C, long hw, long total, float margin, int sqrt_re,
                 float *__restrict__ acc2, float *__restrict__ grad)
{
    float acc = 0.f, cnt = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / ((long)C * hw), pix = i % hw;
        const float p = pred[i], t = gt[i], sc = ps[i];
        float g = 0.f;
        const float tn = t / sc;                        // the reference normalises both sides first
        if (mask[n * hw + pix] && isfinite(tn)) {
            const float d = p / sc - tn;
            const float e = fabsf(d);
            if (e >= margin) {
                const float sg = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / sc;
                if (sqrt_re) {
                    const float r = sqrtf(e);
                    acc += r;
                    g = sg * 0.5f / r;
                } else {
                    acc += e;
                    g = sg;
                }
                cnt += 1.f;
            }
        }
        grad[i] = g;
    }
    acc = wave_sum(acc);
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { atomicAdd(acc2, acc); atomicAdd(acc2 + 1, cnt); }
}

unsigned loss_grid(long total)
{
    const long blocks = (total + 255) / 256;
    return (unsigned)(blocks < 2048 ? blocks : 2048);
}

}  // namespace

OG_API int og_focal_l2_loss_f32(const float *pred, const float *gt, const unsigned char *mask_miss, int N, int C, long hw,
                                float tau, float gamma, float *sum, float *grad, void *stream)
{
    const char *name = "og_focal_l2_loss_f32";
    OG_REQUIRE(pred && gt && mask_miss && sum && grad, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && C > 0 && hw > 0, OG_EINVAL, "%s: bad shape", name);
    const long total = (long)N * C * hw;
    hipLaunchKernelGGL(focal_l2_kernel, dim3(loss_grid(total)), dim3(256), 0, (hipStream_t)stream, pred, gt, mask_miss, C, hw,
                       total, tau, gamma, sum, grad);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

OG_API int og_offset_l1_loss_f32(const float *pred, const float *gt, const float *gt_ps, const unsigned char *mask_miss, int N,
                                 int C, long hw, float margin, int sqrt_re, float *sum_count, float *grad, void *stream)
{
    const char *name = "og_offset_l1_loss_f32";
    OG_REQUIRE(pred && gt && gt_ps && mask_miss && sum_count && grad, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && C > 0 && hw > 0, OG_EINVAL, "%s: bad shape", name);
    const long total = (long)N * C * hw;
    hipLaunchKernelGGL(offset_l1_kernel, dim3(loss_grid(total)), dim3(256), 0, (hipStream_t)stream, pred, gt, gt_ps, mask_miss,
                       C, hw, total, margin, sqrt_re, sum_count, grad);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}
