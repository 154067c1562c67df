// x4 bicubic (A=-0.75, align_corners=False) with torch-CPU fp32 rounding, shared by K1a
// (materialising upsample) and the fused upsample+NMS+top-k kernel.
//   o = fma(t0,w0, fl(t1*w1)); o = fma(t2,w2,o); o = fma(t3,w3,o)
// x-pass on the source rows first, then the same chain down the rows; the four phase weights are
// exact dyadics (n/4096); taps are index-clamped.
#pragma once
#include "og_common.h"

__device__ __constant__ const float og_cubic_w[4][4] = {
    {-270.f / 4096.f, 1746.f / 4096.f, 3070.f / 4096.f, -450.f / 4096.f},
    {-42.f / 4096.f, 470.f / 4096.f, 3962.f / 4096.f, -294.f / 4096.f},
    {-294.f / 4096.f, 3962.f / 4096.f, 470.f / 4096.f, -42.f / 4096.f},
    {-450.f / 4096.f, 3070.f / 4096.f, 1746.f / 4096.f, -270.f / 4096.f},
};

__device__ __forceinline__ float og_cubic_chain(float t0, float t1, float t2, float t3, const float *w)
{
    float o = __builtin_fmaf(t0, w[0], t1 * w[1]);
    o = __builtin_fmaf(t2, w[2], o);
    return __builtin_fmaf(t3, w[3], o);
}

struct OgRow4 {
    float p[4];  // x-pass results of one source row for output columns 4q..4q+3
};

// x-pass of one source row for the lane's source column; c0 = source value at (row, clamp(q)),
// neighbours q-2..q+2 come from the adjacent lanes (DPP wave shifts).
__device__ __forceinline__ OgRow4 og_cubic_xpass(float c0, const float (*wx)[4])
{
    const float cm1 = og_from_lane_below(c0), cm2 = og_from_lane_below(cm1);
    const float cp1 = og_from_lane_above(c0), cp2 = og_from_lane_above(cp1);
    OgRow4 o;
    o.p[0] = og_cubic_chain(cm2, cm1, c0, cp1, wx[0]);
    o.p[1] = og_cubic_chain(cm2, cm1, c0, cp1, wx[1]);
    o.p[2] = og_cubic_chain(cm1, c0, cp1, cp2, wx[2]);
    o.p[3] = og_cubic_chain(cm1, c0, cp1, cp2, wx[3]);
    return o;
}

// one hi-res pixel of the x4 bicubic upsample of a low-res plane (same rounding as K1a); lrb: the mirrored partner plane of the
// flip-test merge or null -- a source value is then (lr[y][x] + lrb[y][w-1-x]) / 2 (og_flip_merge_f32's arithmetic)
__device__ __forceinline__ float og_bicubic4_at(const float *__restrict__ lr, const float *__restrict__ lrb, int h, int w, int Y, int X)
{
    const int qy = Y >> 2, ry = Y & 3, by = (ry < 2) ? qy - 1 : qy;
    const int qx = X >> 2, rx = X & 3, bx = (rx < 2) ? qx - 1 : qx;
    float rowv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const size_t ro = (size_t)min(max(by - 1 + j, 0), h - 1) * w;
        float t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = min(max(bx - 1 + i, 0), w - 1);
            t[i] = lrb ? (lr[ro + c] + lrb[ro + (w - 1 - c)]) / 2.f : lr[ro + c];
        }
        rowv[j] = og_cubic_chain(t[0], t[1], t[2], t[3], og_cubic_w[rx]);
    }
    return og_cubic_chain(rowv[0], rowv[1], rowv[2], rowv[3], og_cubic_w[ry]);
}

__device__ __forceinline__ float og_bicubic4_at(const float *__restrict__ lr, int h, int w, int Y, int X)
{
    return og_bicubic4_at(lr, nullptr, h, w, Y, X);
}
