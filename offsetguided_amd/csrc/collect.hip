// K2 -- LimbsCollect.generate_limbs (decoder/collect.py:62-236, _channel_dets :246-254) as its own launch: one wave
// per (image, limb type) over the (N,C,k) candidate lists of og_nms_topk_f32.  The arithmetic lives in collect_body.h
// (shared with the single-launch og_generate_limbs_f32).  KB-sized, latency-bound: ~40 torch launches in the
// reference, one here.
#include "collect_body.h"

namespace {

template <int ND>
__global__ void __launch_bounds__(64)
collect_limbs_kernel(const float *__restrict__ scores, const int64_t *__restrict__ inds, og_collect::Args a)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int n = blockIdx.x / a.L, l = blockIdx.x % a.L;
    const int cf = a.jf[l], ct = a.jt[l];
    og_collect::limb_rows<ND, int64_t>(a, n, l, threadIdx.x, scores + ((size_t)n * a.C + cf) * a.K,
                                       inds + ((size_t)n * a.C + cf) * a.K, scores + ((size_t)n * a.C + ct) * a.K,
                                       inds + ((size_t)n * a.C + ct) * a.K, sm);
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
    const og_collect::Args a{offs, off_is_lowres, C, H, W, jf, jt, L, k, thre_hmp, min_len, resize_factor, scales, scales_mode,
                             jitter, jitter_mode, limbs};
    hipLaunchKernelGGL(kern, dim3(N * L), dim3(64), (size_t)((k + 3) & ~3) * 16, (hipStream_t)stream, scores, inds, a);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}
