// K0 -- PostProcess.flip_augment, vector-addition form (decoder/factory.py:98-146).
//
//   hm  = (hm[:N]  + flipW(hm[N:])[:, kp_perm]) / 2
//   off = (off[:N] + flipW(off[N:] with x components negated)[:, limb_perm]) / 2,
//         except limbs whose mirror image is their own reverse ({2,5,14} for COCO), which keep
//         the un-averaged original.
// One elementwise pass over the stride-4 maps (the reference makes ~8); exact in fp32.
//
// cat_flip_offs form (factory.py:115-127): the mirrored offsets are appended as components 2,3 of a
// (N, L, 4, h, w) tensor instead of being averaged (reserve limbs repeat their own components there).
#include "og_common.h"

namespace {

__global__ void __launch_bounds__(256)
flip_merge_kernel(const float *__restrict__ hm, const float *__restrict__ off, int N, int C, int L, int h, int w,
                  const int32_t *__restrict__ kp_perm, const int32_t *__restrict__ limb_perm,
                  const int32_t *__restrict__ reserve, float *__restrict__ hm_out, float *__restrict__ off_out, int cat)
{
    const int planes_per_img = C + 2 * L;
    const int plane = blockIdx.y;  // (n, channel) over the concatenated [hm | off] channel list
    const int n = plane / planes_per_img, ch = plane % planes_per_img;
    const size_t hw = (size_t)h * w;
    const float *a, *b;
    float *o, *o2 = nullptr;
    float sign = 1.f;
    bool keep = false;
    if (ch < C) {
        a = hm + ((size_t)n * C + ch) * hw;
        b = hm + ((size_t)(n + N) * C + kp_perm[ch]) * hw;
        o = hm_out + ((size_t)n * C + ch) * hw;
    } else {
        const int oc = ch - C, l = oc >> 1, comp = oc & 1;
        a = off + ((size_t)n * 2 * L + oc) * hw;
        b = off + ((size_t)(n + N) * 2 * L + 2 * limb_perm[l] + comp) * hw;
        o = off_out + ((size_t)n * 2 * L + oc) * hw;
        if (cat) {
            o = off_out + ((size_t)n * 4 * L + 4 * l + comp) * hw;
            o2 = o + 2 * hw;
        }
        sign = comp == 0 ? -1.f : 1.f;
        keep = reserve[l] != 0;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / w), x = (int)(i % w);
        const float av = a[i];
        const float fv = b[(size_t)y * w + (w - 1 - x)] * sign;
        if (o2) {
            o[i] = av;
            o2[i] = keep ? av : fv;
        } else {
            o[i] = keep ? av : (av + fv) / 2.f;
        }
    }
}

}  // namespace

static int flip_launch(const char *name, int cat, const float *hm, const float *off, int N, int C, int L, int h, int w,
                       const int32_t *kp_perm, const int32_t *limb_perm, const int32_t *reserve_mask, float *hm_out,
                       float *off_out, void *stream)
{
    OG_REQUIRE(hm && off && kp_perm && limb_perm && reserve_mask && hm_out && off_out, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && C > 0 && L > 0 && h > 0 && w > 0, OG_EINVAL, "%s: bad shape", name);
    const long planes = (long)N * (C + 2 * L);
    OG_REQUIRE(planes <= 65535, OG_EINVAL, "%s: too many planes", name);
    const int bx = (int)(((size_t)h * w + 255) / 256);
    hipLaunchKernelGGL(flip_merge_kernel, dim3(bx < 64 ? bx : 64, (unsigned)planes), dim3(256), 0, (hipStream_t)stream, hm,
                       off, N, C, L, h, w, kp_perm, limb_perm, reserve_mask, hm_out, off_out, cat);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

OG_API int og_flip_merge_f32(const float *hm, const float *off, int N, int C, int L, int h, int w,
                             const int32_t *kp_perm, const int32_t *limb_perm, const int32_t *reserve_mask,
                             float *hm_out, float *off_out, void *stream)
{
    return flip_launch("og_flip_merge_f32", 0, hm, off, N, C, L, h, w, kp_perm, limb_perm, reserve_mask, hm_out, off_out,
                       stream);
}

OG_API int og_flip_cat_f32(const float *hm, const float *off, int N, int C, int L, int h, int w,
                           const int32_t *kp_perm, const int32_t *limb_perm, const int32_t *reserve_mask,
                           float *hm_out, float *off_out, void *stream)
{
    return flip_launch("og_flip_cat_f32", 1, hm, off, N, C, L, h, w, kp_perm, limb_perm, reserve_mask, hm_out, off_out,
                       stream);
}
