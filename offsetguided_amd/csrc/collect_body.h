// Body of K2 (LimbsCollect.generate_limbs pairing, decoder/collect.py:62-236, _channel_dets :246-254), shared by the
// stand-alone collect_limbs_kernel (csrc/collect.hip) and the single-launch K1 (csrc/nms_topk.hip), whose finishing
// workgroup pairs the candidates it has just selected without a trip through global memory.
//
// One wave per (image, limb type).  The k to-candidates of the limb's end joint are staged in LDS; lane i takes
// from-candidate i, reads its guiding offset (gathered from hi-res offset maps, or bilinearly sampled from the stride-4
// head output with the arithmetic of F.interpolate(x4, 'bilinear'), so the 498 MB hi-res offset tensor is never built),
// scans the to-candidates for the first nearest one and writes its 13-float limb row.
//
// fp32 arithmetic follows torch-CPU exactly where it decides an index:
//   dist = sqrtf(fma(dy,dy, fl(dx*dx)))   (torch.norm over 2 elements)
//   dist = sqrtf(((dx^2 + dy^2) + dx'^2) + dy'^2), no fma, for the 4-component `cat_flip_offs` form
//          (torch's 4-element reduction rounds differently from its 2-element one)
//   first minimum wins (torch.min tie rule on CPU)
// exp() is the device libm (<= 1 ulp from torch's), so limb scores agree to ~1e-7 relative.
#pragma once
#include <math.h>

#include "bicubic.h"
#include "og_common.h"

namespace og_collect {


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

// bilinear4_at over a map given by an accessor p(y, x) (the same fma chain)
template <class F>
__device__ __forceinline__ float bilinear4_fn(F p, int h, int w, int Y, int X)
{
    int x0, x1, y0, y1;
    float lx0, lx1, ly0, ly1;
    lin_coord(X, w, x0, x1, lx0, lx1);
    lin_coord(Y, h, y0, y1, ly0, ly1);
    const float a = __builtin_fmaf(p(y0, x0), lx0, p(y0, x1) * lx1);
    const float b = __builtin_fmaf(p(y1, x0), lx0, p(y1, x1) * lx1);
    return __builtin_fmaf(a, ly0, b * ly1);
}

struct Args {
    const float *offs;     // guiding offsets: (N, ND*L, H, W) or, off_lowres, the stride-4 head output (N, ND*L, H/4, W/4)
    int off_lowres;
    int C, H, W;
    const int32_t *jf, *jt;
    int L, K;
    float thre, min_len, resize;
    const float *scales;   // keypoint-scale head (collect.py:111-122) or nullptr
    int scale_mode;        // 0 none, 1 hi-res maps gathered, 2 / 3 stride-4 maps sampled as bicubic / bilinear x4
    const float *jitter;   // jitter-offset head (collect.py:127-138, :154-165, :210-214) or nullptr
    int jitter_mode;       // 0 none, 1 hi-res maps, 3 stride-4 maps sampled as bilinear x4
    float *limbs;          // (N, L, K, 13)
    // flip-test folded into the sampling (PostProcess.flip_augment, decoder/factory.py:129-138; off_lowres, 2 components):
    // offs holds (2N, 2L, H/4, W/4) = [images | mirrored images]; every tap reads (a + sign * b[limb_perm[l]][y][w-1-x]) / 2,
    // sign = -1 for the x component, or a alone for the limbs of `reserve` -- the values og_flip_merge_f32 would have written
    const int32_t *limb_perm = nullptr, *reserve = nullptr;
    int flip_N = 0;
};

// Rows of limb type `l` of image `n` by one GROUP of lanes (a wave, or half a wave when K <= 32; `lane` = 0..GROUP-1
// within the group, the groups of a wave may work on different limb types).  sf/idf and st/idt: the k best (score, flat
// index) of the limb's from- and to-joint planes (any address space; IdxT = int64_t in global memory, int in LDS).
// `sm`: group-private LDS scratch of 16 bytes x ((K + 3) & ~3), 16-byte aligned.
template <int ND, class IdxT, int GROUP = 64>
__device__ __forceinline__ void limb_rows(const Args &a, int n, int l, int lane, const float *sf, const IdxT *idf,
                                          const float *st, const IdxT *idt, float *sm)
{
    const int C = a.C, H = a.H, W = a.W, L = a.L, K = a.K;
    const float thre = a.thre, min_len = a.min_len, resize = a.resize;
    const float *offs = a.offs, *scales = a.scales, *jitter = a.jitter;
    const int off_lowres = a.off_lowres, scale_mode = a.scale_mode, jitter_mode = a.jitter_mode;
    float *limbs = a.limbs;
    const int Kp = (K + 3) & ~3;                 // to-candidate coordinates interleaved (x,y), padded to 4
    float2 *txy = reinterpret_cast<float2 *>(sm);
    float *ts = sm + 2 * Kp;
    int *ti = reinterpret_cast<int *>(sm + 3 * Kp);
    // jitter-offset head: two shared channels; mode 1 = maps at input resolution, 3 = the stride-4 head output sampled
    // as F.interpolate(x4, 'bilinear') would.  (row, col) are passed as the reference indexes them -- it reads the
    // guide-point refinement at [x][y].
    auto jitter_at = [&](int comp, int row, int col) -> float {
        if (jitter_mode == 1) return jitter[((size_t)n * 2 + comp) * ((long)H * W) + (size_t)row * W + col];
        return bilinear4_at(jitter + ((size_t)n * 2 + comp) * (H / 4) * (W / 4), H / 4, W / 4, row, col);
    };
    // keypoint-scale head: the scale map of the joint's channel at the candidate's pixel
    auto scale_at = [&](int ch, long id, int yy, int xx) -> float {
        if (scale_mode == 0) return 4.f;
        if (scale_mode == 1) return scales[((size_t)n * C + ch) * ((long)H * W) + id];
        const float *pl = scales + ((size_t)n * C + ch) * (H / 4) * (W / 4);
        return scale_mode == 2 ? og_bicubic4_at(pl, H / 4, W / 4, yy, xx) : bilinear4_at(pl, H / 4, W / 4, yy, xx);
    };
    const int cf = a.jf[l], ct = a.jt[l];
    const long HW = (long)H * W;

    // The offset gather of the lane's FIRST from-candidate (two dependent round trips: candidate index, then the offset
    // taps in a tensor that is cold by now) is issued before the to-candidates are staged, so the two latencies overlap.
    auto gather_offsets = [&](int64_t id, int yi, int xi, float (&o4)[ND]) {   // offset at the ORIGINAL flat index (collect.py:143-147)
        if (off_lowres && a.flip_N > 0) {
            const int h4 = H / 4, w4 = W / 4;
            const size_t hw4 = (size_t)h4 * w4;
            const bool keep = a.reserve[l] != 0;
            const float *pa = offs + ((size_t)n * ND * L + ND * l) * hw4;
            const float *pb = offs + ((size_t)(n + a.flip_N) * ND * L + ND * a.limb_perm[l]) * hw4;
#pragma unroll
            for (int c = 0; c < ND; ++c) {
                const float sign = (c & 1) == 0 ? -1.f : 1.f;
                const float *qa = pa + (size_t)c * hw4, *qb = pb + (size_t)c * hw4;
                o4[c] = bilinear4_fn([&](int y, int x) {
                    const float av = qa[(size_t)y * w4 + x];
                    if (keep) return av;
                    return (av + qb[(size_t)y * w4 + (w4 - 1 - x)] * sign) / 2.f;
                }, h4, w4, yi, xi);
            }
        } else if (off_lowres) {
            const int h4 = H / 4, w4 = W / 4;
            const float *px = offs + ((size_t)n * ND * L + ND * l) * h4 * w4;
#pragma unroll
            for (int c = 0; c < ND; ++c) o4[c] = bilinear4_at(px + (size_t)c * h4 * w4, h4, w4, yi, xi);
        } else {
            const float *px = offs + ((size_t)n * ND * L + ND * l) * HW;
#pragma unroll
            for (int c = 0; c < ND; ++c) o4[c] = px[(size_t)c * HW + id];
        }
    };
    int64_t id_first = 0;
    float s_first = 0.f, o_first[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c) o_first[c] = 0.f;
    if (lane < K) {
        id_first = (int64_t)idf[lane];
        s_first = sf[lane];
        gather_offsets(id_first, (int)((unsigned)id_first / (unsigned)W), (int)((unsigned)id_first % (unsigned)W), o_first);
    }

    for (int m = lane; m < Kp; m += GROUP) {
        if (m < K) {
            const int id = (int)idt[m];   // flat indices fit 31 bits (checked by the entry points): 32-bit divisions
            int x = (int)((unsigned)id % (unsigned)W), y = (int)((unsigned)id / (unsigned)W);
            const float s = st[m];
            if (s < thre) { x -= 100000; y -= 100000; }  // collect.py:253
            txy[m] = make_float2((float)x, (float)y);
            ts[m] = s;
            ti[m] = (int)id;
        } else {
            txy[m] = make_float2(INFINITY, INFINITY);    // padding never wins the argmin
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    for (int k = lane; k < K; k += GROUP) {
        const bool first = k == lane;
        const int64_t id = first ? id_first : (int64_t)idf[k];
        const int xi = (int)((unsigned)id % (unsigned)W), yi = (int)((unsigned)id / (unsigned)W);
        const float s1 = first ? s_first : sf[k];
        int xs = xi, ys = yi;
        if (s1 < thre) { xs -= 100000; ys -= 100000; }
        const float xf = (float)xs, yf = (float)ys;
        float o4[ND];
        if (first) {
#pragma unroll
            for (int c = 0; c < ND; ++c) o4[c] = o_first[c];
        } else {
            gather_offsets(id, yi, xi, o4);
        }
#ifdef OG_COLLECT_STAMP
        OG_COLLECT_STAMP(12, o4[0]);
#endif
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
        // collect.py:171-177: first minimum of the distance over the to-candidates, 4 per pair of wide LDS reads.  (A
        // variant that compares squared distances and takes the sqrtf only for candidates within 2^-21 of the minimum is
        // exact too, but its two dependent passes over the LDS list measured 2 us SLOWER per launch than this single
        // pass: the loop is bound by LDS latency, the sqrtf hides under it.)
        int best = 0;
        float bd = INFINITY;
        for (int m0 = 0; m0 < Kp; m0 += 4) {
            const float4 a4 = *reinterpret_cast<const float4 *>(txy + m0), b4 = *reinterpret_cast<const float4 *>(txy + m0 + 2);
            const float cx[4] = {a4.x, a4.z, b4.x, b4.z}, cy[4] = {a4.y, a4.w, b4.y, b4.w};
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
#ifdef OG_COLLECT_STAMP
        OG_COLLECT_STAMP(13, bd);
#endif
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

}  // namespace og_collect
