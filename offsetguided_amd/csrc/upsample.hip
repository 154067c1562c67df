// K1a -- x4 upsampling of the stride-4 head outputs (decoder/factory.py:74-78).
//
// Bicubic (heatmaps): torch-CPU fp32 arithmetic reproduced bit-for-bit --
//   o = fma(t0,w0, fl(t1*w1)); o = fma(t2,w2,o); o = fma(t3,w3,o),
// x-pass on the source rows first, then the same chain down the rows; A=-0.75,
// align_corners=False, taps index-clamped; the four phase weights are exact dyadics.
// Write-bound (16x more bytes out than in): a lane owns one SOURCE column and walks down the
// source rows with a 5-row window of x-pass results in registers; the +-2 column neighbours
// come from adjacent lanes by DPP wave shifts (60 interior + 2x2 halo lanes per wave), and
// every source row step stores four float4 (4 output rows x 4 output columns): consecutive
// lanes write consecutive 16 B, i.e. 1 KiB coalesced stores.
//
// Bilinear (offsets): the decode path never materialises it (K2 samples at the peaks); the
// kernel exists for API parity and uses the same fma order as torch-CPU.
#include <stdlib.h>

#include "og_common.h"

namespace {

constexpr int kCubicInterior = 60;
constexpr int kCubicMaxWaves = 16;

__constant__ float c_cubic_w[4][4] = {
    {-270.f / 4096.f, 1746.f / 4096.f, 3070.f / 4096.f, -450.f / 4096.f},
    {-42.f / 4096.f, 470.f / 4096.f, 3962.f / 4096.f, -294.f / 4096.f},
    {-294.f / 4096.f, 3962.f / 4096.f, 470.f / 4096.f, -42.f / 4096.f},
    {-450.f / 4096.f, 3070.f / 4096.f, 1746.f / 4096.f, -270.f / 4096.f},
};

__device__ __forceinline__ float cubic_chain(float t0, float t1, float t2, float t3, const float *w)
{
    float o = __builtin_fmaf(t0, w[0], t1 * w[1]);
    o = __builtin_fmaf(t2, w[2], o);
    return __builtin_fmaf(t3, w[3], o);
}

struct Row4 {
    float p[4];  // x-pass results of one source row for output columns 4q..4q+3
};

__global__ void __launch_bounds__(64 * kCubicMaxWaves)
bicubic4_kernel(const float *__restrict__ src, float *__restrict__ dst, int h, int w, int rows, int nbands,
                int panel_cols, int total, int padded, const int32_t *__restrict__ kp_perm, int flip_planes, int C)
{
    const int wid = og_xcd_remap(blockIdx.x, padded);
    if (wid >= total) return;
    const int plane = wid / nbands, band = wid % nbands;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q_first = wave * panel_cols;
    const int q_cnt = min(panel_cols, w - q_first);
    const int q = q_first - 2 + lane;                  // source column of this lane (halo: 2 each side)
    const bool interior = lane >= 2 && lane < 2 + q_cnt;
    const int qc = min(max(q, 0), w - 1);              // index clamp == torch's tap clamp
    const float *s = src + (size_t)plane * h * w;
    // flip-test folded into the loads (PostProcess.flip_augment, decoder/factory.py:101-106): src holds [images | mirrored
    // images]; every source pixel is (a + b[kp_perm[ch]][y][w-1-x]) / 2 -- the value og_flip_merge_f32 would have written
    const float *s2 = nullptr;
    if (kp_perm) {
        const int n = plane / C, ch = plane - n * C;
        s2 = src + ((size_t)flip_planes + (size_t)n * C + kp_perm[ch]) * h * w;
    }
    float *d = dst + (size_t)plane * (4 * h) * (4 * w);
    const int p0 = band * rows, p1 = min(p0 + rows, h);

    float wx[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) wx[r][j] = c_cubic_w[r][j];

    auto xpass = [&](int row) {
        const int rc = min(max(row, 0), h - 1);
        float c0 = s[(size_t)rc * w + qc];
        if (s2) c0 = (c0 + s2[(size_t)rc * w + (w - 1 - qc)]) / 2.f;
        const float cm1 = og_from_lane_below(c0), cm2 = og_from_lane_below(cm1);
        const float cp1 = og_from_lane_above(c0), cp2 = og_from_lane_above(cp1);
        Row4 o;
        o.p[0] = cubic_chain(cm2, cm1, c0, cp1, wx[0]);
        o.p[1] = cubic_chain(cm2, cm1, c0, cp1, wx[1]);
        o.p[2] = cubic_chain(cm1, c0, cp1, cp2, wx[2]);
        o.p[3] = cubic_chain(cm1, c0, cp1, cp2, wx[3]);
        return o;
    };

    Row4 a = xpass(p0 - 2), b = xpass(p0 - 1), c = xpass(p0), e = xpass(p0 + 1);
    for (int p = p0; p < p1; ++p) {
        const Row4 f = xpass(p + 2);
        if (interior) {
            float4 o[4];
            float *op = reinterpret_cast<float *>(o);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                op[0 * 4 + x] = cubic_chain(a.p[x], b.p[x], c.p[x], e.p[x], wx[0]);
                op[1 * 4 + x] = cubic_chain(a.p[x], b.p[x], c.p[x], e.p[x], wx[1]);
                op[2 * 4 + x] = cubic_chain(b.p[x], c.p[x], e.p[x], f.p[x], wx[2]);
                op[3 * 4 + x] = cubic_chain(b.p[x], c.p[x], e.p[x], f.p[x], wx[3]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float4 *dstp = reinterpret_cast<float4 *>(d + (size_t)(4 * p + r) * (4 * w) + 4 * q);
                *dstp = o[r];      // plain store (nt / sc1 policies measured in round 2: no gain)
            }
        }
        a = b; b = c; c = e; e = f;
    }
}

__device__ __forceinline__ void lin_coord(int dpos, int n, int &i0, int &i1, float &l0, float &l1)
{
    float s = 0.25f * ((float)dpos + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    i1 = (i0 + 1 < n) ? i0 + 1 : n - 1;
    l1 = s - (float)i0;
    l0 = 1.f - l1;
}

__global__ void __launch_bounds__(256)
bilinear4_kernel(const float *__restrict__ src, float *__restrict__ dst, int h, int w, long total_vec)
{
    const int W4 = w;  // float4 groups per output row == source columns
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total_vec; t += (long)gridDim.x * blockDim.x) {
        const int q = (int)(t % W4);
        const long rest = t / W4;
        const int Y = (int)(rest % (4 * h));
        const long plane = rest / (4 * h);
        const float *s = src + (size_t)plane * h * w;
        int y0, y1;
        float ly0, ly1;
        lin_coord(Y, h, y0, y1, ly0, ly1);
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int x0, x1;
            float lx0, lx1;
            lin_coord(4 * q + r, w, x0, x1, lx0, lx1);
            const float ta = __builtin_fmaf(s[(size_t)y0 * w + x0], lx0, s[(size_t)y0 * w + x1] * lx1);
            const float tb = __builtin_fmaf(s[(size_t)y1 * w + x0], lx0, s[(size_t)y1 * w + x1] * lx1);
            o[r] = __builtin_fmaf(ta, ly0, tb * ly1);
        }
        *reinterpret_cast<float4 *>(dst + ((size_t)plane * 4 * h + Y) * (4 * w) + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace

static int bicubic4_launch(const char *name, const float *src, long planes, int h, int w, float *dst, const int32_t *kp_perm, int C,
                           void *stream)
{
    OG_REQUIRE(src && dst, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(planes > 0 && h > 0 && w > 0, OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE((uintptr_t)dst % 16 == 0, OG_EINVAL, "%s: dst must be 16-byte aligned", name);
    const int nwaves = (w + kCubicInterior - 1) / kCubicInterior;
    OG_REQUIRE(nwaves <= kCubicMaxWaves, OG_EUNSUPPORTED, "%s: w=%d too wide", name, w);
    const int panel_cols = (w + nwaves - 1) / nwaves;
    const int rows = min(h, 8);
    const int nbands = (h + rows - 1) / rows;
    const long total = planes * nbands;
    OG_REQUIRE(total < (1l << 30), OG_EINVAL, "%s: too many work items", name);
    const int padded = (int)((total + 7) / 8 * 8);
    hipLaunchKernelGGL(bicubic4_kernel, dim3(padded), dim3(64 * nwaves), 0, (hipStream_t)stream, src, dst, h, w, rows,
                       nbands, panel_cols, (int)total, padded, kp_perm, (int)planes, C);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

OG_API int og_upsample_bicubic4_f32(const float *src, long planes, int h, int w, float *dst, void *stream)
{
    return bicubic4_launch("og_upsample_bicubic4_f32", src, planes, h, w, dst, nullptr, 1, stream);
}

// flip_augment's heatmap merge (decoder/factory.py:101-106) + F.interpolate(x4, bicubic) (:74-75) in one pass
OG_API int og_upsample_bicubic4_flip_f32(const float *hm_pair, const int32_t *kp_perm, int N, int C, int h, int w, float *dst, void *stream)
{
    const char *name = "og_upsample_bicubic4_flip_f32";
    OG_REQUIRE(kp_perm, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && C > 0, OG_EINVAL, "%s: bad shape", name);
    return bicubic4_launch(name, hm_pair, (long)N * C, h, w, dst, kp_perm, C, stream);
}

OG_API int og_upsample_bilinear4_f32(const float *src, long planes, int h, int w, float *dst, void *stream)
{
    const char *name = "og_upsample_bilinear4_f32";
    OG_REQUIRE(src && dst, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(planes > 0 && h > 0 && w > 0, OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE((uintptr_t)dst % 16 == 0, OG_EINVAL, "%s: dst must be 16-byte aligned", name);
    const long total_vec = planes * 4l * h * w;
    const long blocks = (total_vec + 255) / 256;
    hipLaunchKernelGGL(bilinear4_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream,
                       src, dst, h, w, total_vec);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}
