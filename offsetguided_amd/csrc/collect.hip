// K2 -- LimbsCollect.generate_limbs (decoder/collect.py:62-236, _channel_dets :246-254).
//
// One wave per (image, limb type).  The k to-candidates of the limb's end joint are staged in
// LDS; lane i takes from-candidate i, reads its guiding offset (either gathered from hi-res
// offset maps, or bilinearly sampled from the stride-4 head output with the arithmetic of
// F.interpolate(x4, 'bilinear') so the 498 MB hi-res offset tensor is never built), scans the
// to-candidates for the first nearest one and writes its 13-float limb row.  KB-sized,
// latency-bound: ~40 torch launches in the reference, one here.
//
// fp32 arithmetic follows torch-CPU exactly where it decides an index:
//   dist = sqrtf(fma(dy,dy, fl(dx*dx)))   (torch.norm over 2 elements)
//   dist = sqrtf(((dx^2 + dy^2) + dx'^2) + dy'^2), no fma, for the 4-component `cat_flip_offs` form
//          (torch's 4-element reduction rounds differently from its 2-element one)
//   first minimum wins (torch.min tie rule on CPU)
// exp() is the device libm (<= 1 ulp from torch's), so limb scores agree to ~1e-7 relative.
#include <math.h>

#include "bicubic.h"
#include "og_common.h"

namespace {

__device__ __forceinline__ void lin_coord(int dpos, int n, int &i0, int &i1, float &l0, float &l1)
{
    float s = 0.25f * ((float)dpos + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = (i0 + 1 < n) ? i0 + 1 : n - 1;
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

__device__ __forceinline__ float bilinear4_at(const float *__restrict__ p, int h, int w, int Y, int X)
{
    int x0, x1, y0, y1;
    float lx0, lx1, ly0, ly1;
    lin_coord(X, w, x0, x1, lx0, lx1);
    lin_coord(Y, h, y0, y1, ly0, ly1);
    const float a = __builtin_fmaf(p[(size_t)y0 * w + x0], lx0, p[(size_t)y0 * w + x1] * lx1);
    const float b = __builtin_fmaf(p[(size_t)y1 * w + x0], lx0, p[(size_t)y1 * w + x1] * lx1);
    return __builtin_fmaf(a, ly0, b * ly1);
}

template <int ND>
__global__ void __launch_bounds__(64)
collect_limbs_kernel(const float *__restrict__ scores, const int64_t *__restrict__ inds,
                     const float *__restrict__ offs, int off_lowres, int C, int H, int W,
                     const int32_t *__restrict__ jf, const int32_t *__restrict__ jt, int L, int K,
                     float thre, float min_len, float resize, const float *__restrict__ scales, int scale_mode,
                     const float *__restrict__ jitter, int jitter_mode, float *__restrict__ limbs)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Kp = (K + 3) & ~3;                 // to-candidate coordinates interleaved (x,y), padded to 4
    float2 *txy = reinterpret_cast<float2 *>(sm);
    float *ts = sm + 2 * Kp;
    int *ti = reinterpret_cast<int *>(sm + 3 * Kp);
    // jitter-offset head (collect.py:127-138, :154-165, :210-214): two shared channels; mode 1 = maps at input
    // resolution, 3 = the stride-4 head output sampled as F.interpolate(x4, 'bilinear') would.  (row, col) are passed as
    // the reference indexes them -- it reads the guide-point refinement at [x][y].
    auto jitter_at = [&](int comp, int row, int col) -> float {
        const int n_ = blockIdx.x / L;
        if (jitter_mode == 1) return jitter[((size_t)n_ * 2 + comp) * ((long)H * W) + (size_t)row * W + col];
        return bilinear4_at(jitter + ((size_t)n_ * 2 + comp) * (H / 4) * (W / 4), H / 4, W / 4, row, col);
    };
    // keypoint-scale head (collect.py:111-122): the scale map of the joint's channel at the candidate's pixel;
    // mode 1 = hi-res map gathered, 2 / 3 = stride-4 map sampled as F.interpolate(x4, bicubic / bilinear) would
    auto scale_at = [&](int ch, long id, int yy, int xx) -> float {
        if (scale_mode == 0) return 4.f;
        const int n_ = blockIdx.x / L;
        if (scale_mode == 1) return scales[((size_t)n_ * C + ch) * ((long)H * W) + id];
        const float *pl = scales + ((size_t)n_ * C + ch) * (H / 4) * (W / 4);
        return scale_mode == 2 ? og_bicubic4_at(pl, H / 4, W / 4, yy, xx) : bilinear4_at(pl, H / 4, W / 4, yy, xx);
    };
    const int n = blockIdx.x / L, l = blockIdx.x % L, lane = threadIdx.x;
    const int cf = jf[l], ct = jt[l];
    const long HW = (long)H * W;
    const float *sf = scores + ((size_t)n * C + cf) * K, *st = scores + ((size_t)n * C + ct) * K;
    const int64_t *idf = inds + ((size_t)n * C + cf) * K, *idt = inds + ((size_t)n * C + ct) * K;

    for (int m = lane; m < Kp; m += 64) {
        if (m < K) {
            const int64_t id = idt[m];
            int64_t x = id % W, y = id / W;
            const float s = st[m];
            if (s < thre) { x -= 100000; y -= 100000; }  // collect.py:253
            txy[m] = make_float2((float)x, (float)y);
            ts[m] = s;
            ti[m] = (int)id;
        } else {
            txy[m] = make_float2(INFINITY, INFINITY);    // padding never wins the argmin
        }
    }
    __syncthreads();

    for (int k = lane; k < K; k += 64) {
        const int64_t id = idf[k];
        const int xi = (int)(id % W), yi = (int)(id / W);
        const float s1 = sf[k];
        int64_t xs = xi, ys = yi;
        if (s1 < thre) { xs -= 100000; ys -= 100000; }
        const float xf = (float)xs, yf = (float)ys;
        float o4[ND];  // offset at the ORIGINAL flat index (collect.py:143-147)
        if (off_lowres) {
            const int h4 = H / 4, w4 = W / 4;
            const float *px = offs + ((size_t)n * ND * L + ND * l) * h4 * w4;
#pragma unroll
            for (int c = 0; c < ND; ++c) o4[c] = bilinear4_at(px + (size_t)c * h4 * w4, h4, w4, yi, xi);
        } else {
            const float *px = offs + ((size_t)n * ND * L + ND * l) * HW;
#pragma unroll
            for (int c = 0; c < ND; ++c) o4[c] = px[(size_t)c * HW + id];
        }
        float gx = xf + o4[0] * resize, gy = yf + o4[1] * resize;  // collect.py:152
        if (jitter_mode) {  // :158-165: refinement read at the truncated guide point, indexed [x][y]
            const int qx = (int)gx, qy = (int)gy;
            if (qx >= 0 && qx < W && qy >= 0 && qy < H) {
                const float rx = jitter_at(0, qx, qy), ry = jitter_at(1, qx, qy);
                gx += rx;
                gy += ry;
            }
        }
        const float gx2 = ND == 4 ? xf + o4[ND - 2] * resize : 0.f, gy2 = ND == 4 ? yf + o4[ND - 1] * resize : 0.f;
        int best = 0;
        float bd = INFINITY;
        for (int m0 = 0; m0 < Kp; m0 += 4) {  // collect.py:171-177; 4 candidates per pair of wide LDS reads
            const float4 a = *reinterpret_cast<const float4 *>(txy + m0), b = *reinterpret_cast<const float4 *>(txy + m0 + 2);
            const float cx[4] = {a.x, a.z, b.x, b.z}, cy[4] = {a.y, a.w, b.y, b.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dx = gx - cx[j], dy = gy - cy[j];
                float d;
                if (ND == 2) {
                    d = sqrtf(__builtin_fmaf(dy, dy, dx * dx));
                } else {
                    const float dx2 = gx2 - cx[j], dy2 = gy2 - cy[j];
                    d = sqrtf(((dx * dx + dy * dy) + dx2 * dx2) + dy2 * dy2);
                }
                if (d < bd) { bd = d; best = m0 + j; }   // strict <: first minimum wins
            }
        }
        const float x2 = txy[best].x, y2 = txy[best].y, s2 = ts[best];
        const int id2 = ti[best];
        const float sc1 = scale_at(cf, id, yi, xi), sc2 = scale_at(ct, id2, id2 / W, id2 % W);
        const float lx = xf - x2, ly = yf - y2;
        float len = sqrtf(__builtin_fmaf(ly, ly, lx * lx));
        len = len < min_len ? min_len : len;                        // collect.py:204-205
        const float sc = (s1 * s2) * expf(-bd / len);                // collect.py:208
        float *o = limbs + (((size_t)n * L + l) * K + k) * 13;
        float x1o = xf, y1o = yf, x2o = x2, y2o = y2;
        if (jitter_mode) {  // :210-214 (the limb length above used the unmoved end points, :203)
            x1o += jitter_at(0, yi, xi); y1o += jitter_at(1, yi, xi);
            x2o += jitter_at(0, id2 / W, id2 % W); y2o += jitter_at(1, id2 / W, id2 % W);
        }
        o[0] = x1o; o[1] = y1o; o[2] = s1;
        o[3] = x2o; o[4] = y2o; o[5] = s2;
        o[6] = (float)(id + (int64_t)cf * HW);                       // collect.py:194-199, :227-228
        o[7] = (float)((int64_t)ti[best] + (int64_t)ct * HW);
        o[8] = bd; o[9] = len; o[10] = sc; o[11] = sc1; o[12] = sc2;
    }
}

}  // namespace

OG_API int og_collect_limbs_f32(const float *scores, const int64_t *inds, const float *offs, int off_is_lowres,
                                int N, int C, int H, int W, const int32_t *jf, const int32_t *jt, int L, int k,
                                float thre_hmp, float min_len, float resize_factor, float *limbs, void *stream)
{
    return og_collect_limbs_nd_f32(scores, inds, offs, off_is_lowres, 2, N, C, H, W, jf, jt, L, k, thre_hmp, min_len,
                                   resize_factor, limbs, stream);
}

OG_API int og_collect_limbs_nd_f32(const float *scores, const int64_t *inds, const float *offs, int off_is_lowres,
                                   int vector_nd, int N, int C, int H, int W, const int32_t *jf, const int32_t *jt, int L,
                                   int k, float thre_hmp, float min_len, float resize_factor, float *limbs, void *stream)
{
    return og_collect_limbs_ex_f32(scores, inds, offs, off_is_lowres, vector_nd, nullptr, 0, N, C, H, W, jf, jt, L, k,
                                   thre_hmp, min_len, resize_factor, limbs, stream);
}

OG_API int og_collect_limbs_ex_f32(const float *scores, const int64_t *inds, const float *offs, int off_is_lowres,
                                   int vector_nd, const float *scales, int scales_mode, int N, int C, int H, int W,
                                   const int32_t *jf, const int32_t *jt, int L, int k, float thre_hmp, float min_len,
                                   float resize_factor, float *limbs, void *stream)
{
    return og_collect_limbs_full_f32(scores, inds, offs, off_is_lowres, vector_nd, scales, scales_mode, nullptr, 0, N, C, H, W, jf,
                                     jt, L, k, thre_hmp, min_len, resize_factor, limbs, stream);
}

OG_API int og_collect_limbs_full_f32(const float *scores, const int64_t *inds, const float *offs, int off_is_lowres,
                                     int vector_nd, const float *scales, int scales_mode, const float *jitter,
                                     int jitter_mode, int N, int C, int H, int W, const int32_t *jf, const int32_t *jt, int L,
                                     int k, float thre_hmp, float min_len, float resize_factor, float *limbs, void *stream)
{
    const char *name = "og_collect_limbs_full_f32";
    OG_REQUIRE((jitter_mode == 0 || jitter_mode == 1 || jitter_mode == 3) && (jitter_mode == 0) == (jitter == nullptr),
               OG_EINVAL, "%s: jitter_mode 0 (no head), 1 (hi-res maps) or 3 (stride-4 maps), with a map exactly when not 0", name);
    OG_REQUIRE(jitter_mode == 0 || (H == W && vector_nd == 2), OG_EUNSUPPORTED,
               "%s: the jitter refinement indexes its maps [x][y] like the reference: square inputs, 2-component offsets", name);
    OG_REQUIRE(jitter_mode != 3 || H % 4 == 0, OG_EINVAL, "%s: H,W must be multiples of 4", name);
    OG_REQUIRE(scales_mode >= 0 && scales_mode <= 3 && (scales_mode == 0) == (scales == nullptr), OG_EINVAL,
               "%s: scales_mode 0 (no scale head) .. 3, with a map exactly when it is not 0", name);
    OG_REQUIRE(scales_mode < 2 || (H % 4 == 0 && W % 4 == 0), OG_EINVAL, "%s: H,W must be multiples of 4", name);
    OG_REQUIRE(vector_nd == 2 || vector_nd == 4, OG_EUNSUPPORTED, "%s: vector_nd must be 2 or 4", name);
    OG_REQUIRE(scores && inds && offs && jf && jt && limbs, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && L > 0 && k > 0, OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE(!off_is_lowres || (H % 4 == 0 && W % 4 == 0), OG_EINVAL, "%s: H,W must be multiples of 4", name);
    OG_REQUIRE((long)H * W < (1l << 31), OG_EINVAL, "%s: plane too large", name);
    OG_REQUIRE(k <= 2048, OG_EUNSUPPORTED, "%s: k=%d too large", name, k);
    auto kern = vector_nd == 2 ? collect_limbs_kernel<2> : collect_limbs_kernel<4>;
    hipLaunchKernelGGL(kern, dim3(N * L), dim3(64), (size_t)((k + 3) & ~3) * 16, (hipStream_t)stream, scores, inds, offs,
                       off_is_lowres, C, H, W, jf, jt, L, k, thre_hmp, min_len, resize_factor, scales, scales_mode, jitter, jitter_mode, limbs);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}
