// Ground-truth encoder on the device: encoder/heatmap.py:125-197 (Gaussian keypoint heatmaps + reversed
// background) and encoder/offset.py:98-197 (guiding offsets, keypoint scales, person scales).
//
// The reference paints person after person into numpy arrays on a dataloader worker (17 samples/s per worker,
// data/factory.py:284).  Both paintings are order-free or "first best wins" reductions, so they invert into gathers:
// one thread per output pixel walks the persons (joints staged in LDS) and keeps
//   heatmaps: max_p [ window_p(x,y) ] clip( exp(-dy^2/2s^2) * exp(-dx^2/2s^2) )      (max commutes),
//   offsets : the vector to the to-joint of the FIRST person with the strictly shortest vector among those whose
//             fill window round the from-joint covers the pixel  (the reference's sequential `len < current`).
// fp32 arithmetic as in the reference (numpy float32 with weak Python scalars); window edges are Python round() of an
// np.float32 = round-half-even = rintf.  exp() is the device libm: values agree to ~2e-7, a pixel whose value sits
// exactly at clip_thre may fall on the other side of the clip.  Offsets / scales are bit-exact.
#include <math.h>

#include "og_common.h"

namespace {

constexpr int kMaxStage = 4096;  // floats of joint data staged per workgroup

struct Window {
    int x0, x1, y0, y1;
};

__device__ __forceinline__ Window patch(float jx, float jy, int stride, float size)
{
    const float half = size / 2.f, qx = jx / (float)stride, qy = jy / (float)stride;
    Window w;
    w.x0 = max((int)rintf(qx - half), 0);
    w.x1 = (int)rintf(qx + half);
    w.y0 = max((int)rintf(qy - half), 0);
    w.y1 = (int)rintf(qy + half);
    return w;
}

__device__ __forceinline__ float grid(int i, int stride) { return (float)(i * stride + stride / 2.0 - 0.5); }

// grid (pixel blocks, N); thread = pixel; loops channels x persons.  hm (N,n_kp,h,w), bg (N,1,h,w) or null.
__global__ void __launch_bounds__(256)
encode_heatmaps_kernel(const float *__restrict__ joints, const int32_t *__restrict__ n_persons, int P, int n_kp, int w,
                       int h, int stride, float gsize, float ds2, float clip, float *__restrict__ hm, float *__restrict__ bg)
{
    __shared__ float sj[kMaxStage];
    const int n = blockIdx.y, np_ = n_persons ? min(n_persons[n], P) : P;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    const int x = pix % w, y = pix / w;
    const bool live = pix < w * h;
    const float gx = grid(x, stride), gy = grid(y, stride);
    float best_all = 0.f;
    const int per_round = kMaxStage / (4 * n_kp);  // persons staged per round
    for (int c = 0; c < n_kp; ++c) {
        float best = 0.f;
        for (int p0 = 0; p0 < np_; p0 += per_round * n_kp) {  // (channel c of) up to per_round*n_kp persons per round
            const int cnt = min(per_round * n_kp, np_ - p0);
            __syncthreads();
            for (int i = threadIdx.x; i < cnt * 4; i += blockDim.x)
                sj[i] = joints[(((size_t)n * P + p0 + i / 4) * n_kp + c) * 4 + (i & 3)];
            __syncthreads();
            if (!live) continue;
            for (int p = 0; p < cnt; ++p) {
                const float jx = sj[4 * p], jy = sj[4 * p + 1], v = sj[4 * p + 2];
                if (!(v > 0.f)) continue;
                const Window wd = patch(jx, jy, stride, gsize);
                if (x < wd.x0 || x >= wd.x1 || y < wd.y0 || y >= wd.y1) continue;
                const float dx = gx - jx, dy = gy - jy;
                float e = expf(-(dy * dy) / ds2) * expf(-(dx * dx) / ds2);
                if (e < clip) e = 0.f;
                best = fmaxf(best, e);
            }
        }
        if (live) {
            hm[((size_t)n * n_kp + c) * h * w + pix] = best;
            best_all = fmaxf(best_all, best);
        }
    }
    if (live && bg) bg[(size_t)n * h * w + pix] = 1.f - best_all;  // heatmap.py:78
}

// grid (pixel blocks, L, N); thread = pixel of limb l.  off/pscale (N,2L,h,w).
__global__ void __launch_bounds__(256)
encode_offsets_kernel(const float *__restrict__ joints, const int32_t *__restrict__ n_persons, int P, int n_kp,
                      const int32_t *__restrict__ jf, const int32_t *__restrict__ jt, int L, int w, int h, int stride,
                      float fill, const float *__restrict__ sigmas, float *__restrict__ off, float *__restrict__ pscale)
{
    __shared__ float sj[kMaxStage];
    const int n = blockIdx.z, l = blockIdx.y, np_ = n_persons ? min(n_persons[n], P) : P;
    const int fr = jf[l], to = jt[l];
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    const int x = pix % w, y = pix / w;
    const bool live = pix < w * h;
    const float gx = grid(x, stride), gy = grid(y, stride);
    float bx = INFINITY, by = INFINITY, blen = INFINITY, bs = 0.f;
    bool any = false;
    const int per_round = kMaxStage / 8;
    for (int p0 = 0; p0 < np_; p0 += per_round) {
        const int cnt = min(per_round, np_ - p0);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt * 8; i += blockDim.x) {
            const int p = i >> 3, k = i & 7;
            sj[i] = joints[(((size_t)n * P + p0 + p) * n_kp + (k < 4 ? fr : to)) * 4 + (k & 3)];
        }
        __syncthreads();
        if (!live) continue;
        for (int p = 0; p < cnt; ++p) {
            const float *a = sj + 8 * p, *b = a + 4;
            if (!(a[2] > 0.f && b[2] > 0.f)) continue;
            const Window wd = patch(a[0], a[1], stride, fill);
            if (x < wd.x0 || x >= wd.x1 || y < wd.y0 || y >= wd.y1) continue;
            const float ox = b[0] - gx, oy = b[1] - gy;
            const float len = sqrtf(ox * ox + oy * oy);
            if (len < blen) {  // strict: the earlier person keeps a tie (offset.py:190)
                blen = len; bx = ox; by = oy; bs = a[3];
                any = true;
            }
        }
    }
    if (!live) return;
    const size_t o = ((size_t)n * 2 * L + 2 * l) * h * w + pix, hw = (size_t)h * w;
    off[o] = bx;
    off[o + hw] = by;
    const float ps = any ? bs / sigmas[fr] : 1.f;
    pscale[o] = ps;
    pscale[o + hw] = ps;
}

// Jitter offsets (heatmap.py:199-255): two shared channels, vector from the cell centre to the nearest annotated
// keypoint of ANY channel whose fill window covers the cell (channel-major, then person order; strict <).
// grid (pixel blocks, N); thread = pixel.
__global__ void __launch_bounds__(256)
encode_jitter_kernel(const float *__restrict__ joints, const int32_t *__restrict__ n_persons, int P, int n_kp, int w,
                     int h, int stride, float fill, float *__restrict__ jit)
{
    const int n = blockIdx.y, np_ = n_persons ? min(n_persons[n], P) : P;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= w * h) return;
    const int x = pix % w, y = pix / w;
    const float gx = grid(x, stride), gy = grid(y, stride);
    float bx = INFINITY, by = INFINITY, blen = INFINITY;
    for (int c = 0; c < n_kp; ++c)
        for (int p = 0; p < np_; ++p) {
            const float *j = joints + (((size_t)n * P + p) * n_kp + c) * 4;
            if (!(j[2] > 0.f)) continue;
            const Window wd = patch(j[0], j[1], stride, fill);
            if (x < wd.x0 || x >= wd.x1 || y < wd.y0 || y >= wd.y1) continue;
            const float ox = j[0] - gx, oy = j[1] - gy;
            const float len = sqrtf(ox * ox + oy * oy);
            if (len < blen) { blen = len; bx = ox; by = oy; }
        }
    jit[(size_t)n * 2 * h * w + pix] = bx;
    jit[((size_t)n * 2 + 1) * h * w + pix] = by;
}

// Keypoint scale maps: channel c holds the from-joint scale written by the LAST limb (skeleton order) with jf == c
// whose winner covers the pixel (offset.py:193 overwrites limb after limb).  grid (pixel blocks, n_kp, N).
__global__ void __launch_bounds__(256)
encode_scales_kernel(const float *__restrict__ joints, const int32_t *__restrict__ n_persons, int P, int n_kp,
                     const int32_t *__restrict__ jf, const int32_t *__restrict__ jt, int L, int w, int h, int stride,
                     float fill, float min_jscale, float *__restrict__ scale)
{
    const int n = blockIdx.z, c = blockIdx.y, np_ = n_persons ? min(n_persons[n], P) : P;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= w * h) return;
    const int x = pix % w, y = pix / w;
    const float gx = grid(x, stride), gy = grid(y, stride);
    float val = NAN;
    for (int l = 0; l < L; ++l) {
        if (jf[l] != c) continue;
        float blen = INFINITY, bs = 0.f;
        bool any = false;
        for (int p = 0; p < np_; ++p) {
            const float *a = joints + (((size_t)n * P + p) * n_kp + c) * 4, *b = joints + (((size_t)n * P + p) * n_kp + jt[l]) * 4;
            if (!(a[2] > 0.f && b[2] > 0.f)) continue;
            const Window wd = patch(a[0], a[1], stride, fill);
            if (x < wd.x0 || x >= wd.x1 || y < wd.y0 || y >= wd.y1) continue;
            const float ox = b[0] - gx, oy = b[1] - gy;
            const float len = sqrtf(ox * ox + oy * oy);
            if (len < blen) { blen = len; bs = a[3]; any = true; }
        }
        if (any) val = bs >= min_jscale ? bs : NAN;
    }
    scale[((size_t)n * n_kp + c) * h * w + pix] = val;
}

}  // namespace

OG_API int og_encode_heatmaps_f32(const float *joints, const int32_t *n_persons, int N, int P, int n_kp, int in_w,
                                  int in_h, int stride, int sigma, float clip_thre, float *hm, float *bg, void *stream)
{
    const char *name = "og_encode_heatmaps_f32";
    OG_REQUIRE(joints && hm, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && P >= 0 && n_kp > 0 && stride > 0 && sigma > 0 && in_w >= stride && in_h >= stride, OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE(clip_thre > 0.f && clip_thre < 1.f, OG_EINVAL, "%s: clip threshold must be in (0,1)", name);
    OG_REQUIRE(4 * n_kp <= kMaxStage && N <= 65535, OG_EUNSUPPORTED, "%s: too many keypoints / images", name);
    const int w = in_w / stride, h = in_h / stride;
    const double ds2 = 2.0 * sigma * sigma;
    const float gsize = (float)(2 * (int)ceil(sqrt(-ds2 * log((double)clip_thre)) / stride));  // heatmap.py:110-111
    hipLaunchKernelGGL(encode_heatmaps_kernel, dim3((w * h + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, joints,
                       n_persons, P, n_kp, w, h, stride, gsize, (float)ds2, clip_thre, hm, bg);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

OG_API int og_encode_jitter_f32(const float *joints, const int32_t *n_persons, int N, int P, int n_kp, int in_w, int in_h,
                                int stride, int fill_size, float *jit, void *stream)
{
    const char *name = "og_encode_jitter_f32";
    OG_REQUIRE(joints && jit, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && P >= 0 && n_kp > 0 && stride > 0 && fill_size > 0 && in_w >= stride && in_h >= stride && N <= 65535,
               OG_EINVAL, "%s: bad shape", name);
    const int w = in_w / stride, h = in_h / stride;
    hipLaunchKernelGGL(encode_jitter_kernel, dim3((w * h + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, joints, n_persons,
                       P, n_kp, w, h, stride, (float)fill_size, jit);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

OG_API int og_encode_offsets_f32(const float *joints, const int32_t *n_persons, int N, int P, int n_kp, const int32_t *jf,
                                 const int32_t *jt, int L, int in_w, int in_h, int stride, int fill_size,
                                 float min_jscale, const float *sigmas, float *off, float *scale, float *pscale,
                                 void *stream)
{
    const char *name = "og_encode_offsets_f32";
    OG_REQUIRE(joints && jf && jt && sigmas && off && pscale, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && P >= 0 && n_kp > 0 && L > 0 && stride > 0 && fill_size > 0 && in_w >= stride && in_h >= stride,
               OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE(N <= 65535 && L <= 65535, OG_EUNSUPPORTED, "%s: too many images / limbs", name);
    const int w = in_w / stride, h = in_h / stride;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(encode_offsets_kernel, dim3((w * h + 255) / 256, L, N), dim3(256), 0, st, joints, n_persons, P, n_kp,
                       jf, jt, L, w, h, stride, (float)fill_size, sigmas, off, pscale);
    OG_LAUNCH_CHECK(name);
    if (scale) {
        hipLaunchKernelGGL(encode_scales_kernel, dim3((w * h + 255) / 256, n_kp, N), dim3(256), 0, st, joints, n_persons, P,
                           n_kp, jf, jt, L, w, h, stride, (float)fill_size, min_jscale, scale);
        OG_LAUNCH_CHECK(name);
    }
    return OG_OK;
}
