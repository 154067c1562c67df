// Backbone epilogues for the bf16 channels-last (NHWC) inference engine.
//
// Stand-alone passes.  Since round 3 every convolution of the engine carries bias + residual + ReLU in its own epilogue
// (conv3x3*.inc / conv_band.hip); what is left here: og_bias_act_* behind a convolution that torch ran (InferenceEngine(strict=False)
// only), og_upsample2_add_* for the hourglass merges of the 20x20 / 10x10 levels, the layout kernels.  They do the epilogue of
//   convolution.forward  (models/hourglass_104.py:26-30):  relu(conv + b)
//   residual.forward     (models/hourglass_104.py:70-79):  relu(conv2 + b2 + skip)
//   kp_module.forward    (models/hourglass_104.py:183-190): up1 + nearest_x2(low3)
// in one in-place pass, 16 B (8 x bf16) per lane, fp32 arithmetic, one rounding to bf16.
#include "lp_dtype.h"
#include "og_common.h"

namespace {

typedef unsigned short v8u16 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf2f(unsigned short u) { return lp2f(u); }            // (bf16, or fp16 in the -DOG_DT_F16 build)
__device__ __forceinline__ unsigned short f2bf(float f) { return f2lp(f); }

template <bool SKIP, bool RELU>
__global__ void __launch_bounds__(256)
bias_act_kernel(unsigned short *__restrict__ x, const float *__restrict__ bias, const unsigned short *__restrict__ skip,
                long groups, int cgroups)
{
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (long)gridDim.x * blockDim.x) {
        const int c0 = (int)(g % cgroups) * 8;
        v8u16 v = *reinterpret_cast<const v8u16 *>(x + g * 8);
        v8u16 s;
        if (SKIP) s = *reinterpret_cast<const v8u16 *>(skip + g * 8);
        const float4 b0 = *reinterpret_cast<const float4 *>(bias + c0), b1 = *reinterpret_cast<const float4 *>(bias + c0 + 4);
        const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = bf2f(v[j]) + b[j];
            if (SKIP) f += bf2f(s[j]);
            if (RELU) f = fmaxf(f, 0.f);
            v[j] = f2bf(f);
        }
        *reinterpret_cast<v8u16 *>(x + g * 8) = v;
    }
}

// up[n][y][x][c] += low[n][y/2][x/2][c]   (nearest x2 upsample fused into the merge add)
__global__ void __launch_bounds__(256)
upsample2_add_kernel(unsigned short *__restrict__ up, const unsigned short *__restrict__ low, long groups, int cgroups,
                     int H, int W)
{
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(g % cgroups);
        long p = g / cgroups;
        const int x = (int)(p % W);
        p /= W;
        const int y = (int)(p % H);
        const long n = p / H;
        const long src = ((n * (H / 2) + y / 2) * (W / 2) + x / 2) * cgroups + cg;
        v8u16 v = *reinterpret_cast<const v8u16 *>(up + g * 8);
        const v8u16 l = *reinterpret_cast<const v8u16 *>(low + src * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) + bf2f(l[j]));
        *reinterpret_cast<v8u16 *>(up + g * 8) = v;
    }
}

unsigned grid_for(long groups)
{
    const long blocks = (groups + 255) / 256;
    return (unsigned)(blocks < 256 * 8 ? blocks : 256 * 8);  // <= 8 blocks per CU, grid-stride beyond
}

}  // namespace

OG_API int OG_LP_NAME(og_bias_act)(void *x, const float *bias, const void *skip, long pixels, int channels, int relu, void *stream)
{
    const char *name = OG_LP_STR("og_bias_act");
    OG_REQUIRE(x && bias, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(pixels > 0 && channels > 0 && channels % 8 == 0, OG_EINVAL, "%s: channels must be a multiple of 8", name);
    OG_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)bias % 16 == 0 && (uintptr_t)skip % 16 == 0, OG_EINVAL,
               "%s: pointers must be 16-byte aligned", name);
    const int cg = channels / 8;
    const long groups = pixels * cg;
    unsigned short *xp = (unsigned short *)x;
    const unsigned short *sp = (const unsigned short *)skip;
    const dim3 grid(grid_for(groups)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (skip && relu) hipLaunchKernelGGL((bias_act_kernel<true, true>), grid, block, 0, st, xp, bias, sp, groups, cg);
    else if (skip) hipLaunchKernelGGL((bias_act_kernel<true, false>), grid, block, 0, st, xp, bias, sp, groups, cg);
    else if (relu) hipLaunchKernelGGL((bias_act_kernel<false, true>), grid, block, 0, st, xp, bias, sp, groups, cg);
    else hipLaunchKernelGGL((bias_act_kernel<false, false>), grid, block, 0, st, xp, bias, sp, groups, cg);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

OG_API int OG_LP_NAME(og_upsample2_add)(void *up, const void *low, long n, int H, int W, int channels, void *stream)
{
    const char *name = OG_LP_STR("og_upsample2_add");
    OG_REQUIRE(up && low, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(n > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && channels % 8 == 0, OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE((uintptr_t)up % 16 == 0 && (uintptr_t)low % 16 == 0, OG_EINVAL, "%s: pointers must be 16-byte aligned", name);
    const int cg = channels / 8;
    const long groups = n * H * W * cg;
    hipLaunchKernelGGL(upsample2_add_kernel, dim3(grid_for(groups)), dim3(256), 0, (hipStream_t)stream,
                       (unsigned short *)up, (const unsigned short *)low, groups, cg, H, W);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

// ---- layout/dtype conversions at the two ends of the engine -------------------------------------------------------
// The reference interface hands the model fp32 NCHW images and expects fp32 NCHW head maps; the engine computes in
// bf16 NHWC.  PyTorch does each conversion as a cast pass plus a layout pass; these do it in one.
namespace {

// images (N,C,H,W) fp32 -> (N,H,W,C) bf16; thread = pixel, C small (3)
template <int C>
__global__ void __launch_bounds__(256)
nchw_to_nhwc_bf16_kernel(const float *__restrict__ src, unsigned short *__restrict__ dst, long hw, long total)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long n = i / hw, p = i % hw;
    const float *s = src + n * C * hw + p;
    unsigned short *d = dst + i * C;
#pragma unroll
    for (int c = 0; c < C; ++c) d[c] = f2bf(s[(long)c * hw]);
}

// head maps: src (N,H,W,src_c) bf16, channels [c0, c0+C) (+ bias) -> dst (N,C,H,W) fp32; thread = pixel
__global__ void __launch_bounds__(256)
nhwc_slice_to_nchw_f32_kernel(const unsigned short *__restrict__ src, int src_c, int c0, int C,
                              const float *__restrict__ bias, float *__restrict__ dst, long hw, long total)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long n = i / hw, p = i % hw;
    const unsigned short *s = src + i * src_c + c0;
    float *d = dst + n * C * hw + p;
    for (int c = 0; c < C; ++c) d[(long)c * hw] = bf2f(s[c]) + (bias ? bias[c0 + c] : 0.f);
}

}  // namespace

#ifdef OG_DT_F16
#define og_nchw_f32_to_nhwc_lp og_nchw_f32_to_nhwc_f16
#define og_nhwc_lp_to_nchw_f32 og_nhwc_f16_to_nchw_f32
#else
#define og_nchw_f32_to_nhwc_lp og_nchw_f32_to_nhwc_bf16
#define og_nhwc_lp_to_nchw_f32 og_nhwc_bf16_to_nchw_f32
#endif
OG_API int og_nchw_f32_to_nhwc_lp(const float *src, void *dst, long N, int C, int H, int W, void *stream)
{
    const char *name = OG_LP_STR("og_nchw_f32_to_nhwc");
    OG_REQUIRE(src && dst, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && H > 0 && W > 0, OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE(C == 3, OG_EUNSUPPORTED, "%s: only 3-channel images", name);
    const long hw = (long)H * W, total = N * hw;
    hipLaunchKernelGGL((nchw_to_nhwc_bf16_kernel<3>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       src, (unsigned short *)dst, hw, total);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

OG_API int og_nhwc_lp_to_nchw_f32(const void *src, int src_channels, int first_channel, int channels, const float *bias,
                                    float *dst, long N, int H, int W, void *stream)
{
    const char *name = "og_nhwc_" OG_LP_STR("") "_to_nchw_f32";
    OG_REQUIRE(src && dst, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && H > 0 && W > 0 && channels > 0 && first_channel >= 0 && first_channel + channels <= src_channels,
               OG_EINVAL, "%s: bad shape", name);
    const long hw = (long)H * W, total = N * hw;
    hipLaunchKernelGGL(nhwc_slice_to_nchw_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short *)src, src_channels, first_channel, channels, bias, dst, hw, total);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

// ---- input side of evaluate.py (the part that can be pinned without cv2) ------------------------------
// CenterPad(fill 124,116,104) + ToTensor + Normalize (transforms/pad.py:40-66, evaluate.py:163-168) on an already
// rescaled uint8 HWC image, in one pass: out[c][Y][X] = (px/255 - mean[c]) / std[c] in fp32 (torchvision's
// div(255), sub_(mean), div_(std)), px = image pixel or the fill colour.  The rescale itself (cv2.resize) is not here.
namespace {

struct PrepArgs {
    float mean[3], stdv[3], fill[3];
};

__global__ void __launch_bounds__(256)
center_pad_normalize_kernel(const unsigned char *__restrict__ img, int h, int w, int left, int top, int TH, int TW,
                            PrepArgs a, float *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)TH * TW) return;
    const int X = (int)(i % TW), Y = (int)(i / TW);
    const int x = X - left, y = Y - top;
    const bool inside = x >= 0 && x < w && y >= 0 && y < h;
    const unsigned char *p = img + ((size_t)(inside ? y : 0) * w + (inside ? x : 0)) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = inside ? (float)p[c] : a.fill[c];
        out[(size_t)c * TH * TW + i] = (v / 255.f - a.mean[c]) / a.stdv[c];
    }
}

}  // namespace

#ifndef OG_DT_F16   // not 16-bit-type specific: exists once, in the bf16 build
OG_API int og_center_pad_normalize_u8(const unsigned char *img, int h, int w, int target_h, int target_w, const float *mean3,
                                      const float *std3, const float *fill3, float *out, int *ltrb, void *stream)
{
    const char *name = "og_center_pad_normalize_u8";
    OG_REQUIRE(img && mean3 && std3 && fill3 && out, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(h > 0 && w > 0 && target_h >= h && target_w >= w, OG_EINVAL, "%s: the image must fit the target", name);
    // transforms/pad.py:43-55: left = int((T - w) / 2.0), top likewise; the rest goes right / down
    const int left = (int)((target_w - w) / 2.0), top = (int)((target_h - h) / 2.0);
    if (ltrb) { ltrb[0] = left; ltrb[1] = top; ltrb[2] = target_w - w - left; ltrb[3] = target_h - h - top; }
    PrepArgs a;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; a.fill[c] = fill3[c]; }
    const long total = (long)target_h * target_w;
    hipLaunchKernelGGL(center_pad_normalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img,
                       h, w, left, top, target_h, target_w, a, out);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}
#endif
