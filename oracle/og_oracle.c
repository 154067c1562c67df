/*
 * og_oracle.c -- CPU restatement of the OffsetGuided decoder hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (offsetguided_amd/,
 * libog_decoder.so) may import, link or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker -- never as the thing shipped or measured as the product.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit (indices,
 * coordinates, grouping) / to 1e-4 (scores) against the imported Python
 * reference (torch 2.10 CPU fp32, floor-division shim for topK_channel) by
 * tools/gen_golden.py in the build container, and against the committed
 * fixtures under tests/golden/ by tests/test_oracle_golden.py everywhere.
 *
 * Each function cites the reference file:line it restates (paths relative to
 * the reference repository root).
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off; FMAs are explicit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OGO_API __attribute__((visibility("default")))

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------- */
/* x4 bicubic, align_corners=False, A=-0.75 (decoder/factory.py:74-75 calling  */
/* torch.nn.functional.interpolate(mode='bicubic')).  torch-CPU arithmetic:   */
/*   o = fma(t0,w0, fl(t1*w1)); o = fma(t2,w2,o); o = fma(t3,w3,o)            */
/* x-pass on each of the 4 source rows, then the same chain over the rows.    */
/* Weights are exact dyadics (n/4096) for the four phases of a x4 upsample.   */
/* ------------------------------------------------------------------------- */
static const float OGO_CUBIC_W[4][4] = {
    {-270.f / 4096.f, 1746.f / 4096.f, 3070.f / 4096.f, -450.f / 4096.f},
    {-42.f / 4096.f, 470.f / 4096.f, 3962.f / 4096.f, -294.f / 4096.f},
    {-294.f / 4096.f, 3962.f / 4096.f, 470.f / 4096.f, -42.f / 4096.f},
    {-450.f / 4096.f, 3070.f / 4096.f, 1746.f / 4096.f, -270.f / 4096.f},
};

static inline float cubic_chain(float t0, float t1, float t2, float t3, const float *w)
{
    float o = fmaf(t0, w[0], t1 * w[1]);
    o = fmaf(t2, w[2], o);
    o = fmaf(t3, w[3], o);
    return o;
}

OGO_API void ogo_bicubic4(const float *src, long planes, int h, int w, float *dst)
{
    const int H = 4 * h, W = 4 * w;
    float *rowbuf = (float *)malloc(sizeof(float) * (size_t)h * W); /* x-pass of every source row */
    int *ix = (int *)malloc(sizeof(int) * 4 * W);
    for (int X = 0; X < W; ++X) {
        int q = X >> 2, r = X & 3, b = (r < 2) ? q - 1 : q;
        for (int j = 0; j < 4; ++j) ix[4 * X + j] = clampi(b - 1 + j, 0, w - 1);
    }
    for (long p = 0; p < planes; ++p) {
        const float *s = src + (size_t)p * h * w;
        float *d = dst + (size_t)p * H * W;
        for (int y = 0; y < h; ++y) {
            const float *sr = s + (size_t)y * w;
            float *rb = rowbuf + (size_t)y * W;
            for (int X = 0; X < W; ++X) {
                const int *i4 = ix + 4 * X;
                rb[X] = cubic_chain(sr[i4[0]], sr[i4[1]], sr[i4[2]], sr[i4[3]], OGO_CUBIC_W[X & 3]);
            }
        }
        for (int Y = 0; Y < H; ++Y) {
            int q = Y >> 2, r = Y & 3, b = (r < 2) ? q - 1 : q;
            const float *r0 = rowbuf + (size_t)clampi(b - 1, 0, h - 1) * W;
            const float *r1 = rowbuf + (size_t)clampi(b, 0, h - 1) * W;
            const float *r2 = rowbuf + (size_t)clampi(b + 1, 0, h - 1) * W;
            const float *r3 = rowbuf + (size_t)clampi(b + 2, 0, h - 1) * W;
            const float *wy = OGO_CUBIC_W[r];
            float *dr = d + (size_t)Y * W;
            for (int X = 0; X < W; ++X) dr[X] = cubic_chain(r0[X], r1[X], r2[X], r3[X], wy);
        }
    }
    free(ix);
    free(rowbuf);
}

/* ------------------------------------------------------------------------- */
/* x4 bilinear, align_corners=False (decoder/factory.py:77-78).               */
/*   s = max(0.25*(d+0.5)-0.5, 0); i0=floor(s); i1=min(i0+1,n-1); l1=s-i0;    */
/*   x-pass a = fma(v[i0], l0, fl(v[i1]*l1)) on rows i0y,i1y; y-pass likewise */
/* ------------------------------------------------------------------------- */
static inline void lin_coord(int d, int n, int *i0, int *i1, float *l0, float *l1)
{
    float s = 0.25f * ((float)d + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    int a = (int)s; /* s >= 0: trunc == floor */
    *i0 = a;
    *i1 = (a + 1 < n) ? a + 1 : n - 1;
    *l1 = s - (float)a;
    *l0 = 1.f - *l1;
}

OGO_API float ogo_bilinear4_at(const float *plane, int h, int w, int Y, int X)
{
    int x0, x1, y0, y1;
    float lx0, lx1, ly0, ly1;
    lin_coord(X, w, &x0, &x1, &lx0, &lx1);
    lin_coord(Y, h, &y0, &y1, &ly0, &ly1);
    const float *ra = plane + (size_t)y0 * w, *rb = plane + (size_t)y1 * w;
    float a = fmaf(ra[x0], lx0, ra[x1] * lx1);
    float b = fmaf(rb[x0], lx0, rb[x1] * lx1);
    return fmaf(a, ly0, b * ly1);
}

OGO_API void ogo_bilinear4(const float *src, long planes, int h, int w, float *dst)
{
    const int H = 4 * h, W = 4 * w;
    for (long p = 0; p < planes; ++p) {
        const float *s = src + (size_t)p * h * w;
        float *d = dst + (size_t)p * H * W;
        for (int Y = 0; Y < H; ++Y)
            for (int X = 0; X < W; ++X) d[(size_t)Y * W + X] = ogo_bilinear4_at(s, h, w, Y, X);
    }
}

/* ------------------------------------------------------------------------- */
/* hmp_NMS (decoder/heatmap.py:15-35): zero-pad 1, 3x3 max, keep where equal;  */
/* out = heat * keep_mask (so suppressed negatives become -0.0).              */
/* ------------------------------------------------------------------------- */
OGO_API void ogo_hmp_nms(const float *in, long planes, int H, int W, float *out)
{
    for (long p = 0; p < planes; ++p) {
        const float *s = in + (size_t)p * H * W;
        float *d = out + (size_t)p * H * W;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float m = -INFINITY;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        int yy = y + dy, xx = x + dx;
                        float v = (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0.f : s[(size_t)yy * W + xx];
                        if (v > m) m = v;
                    }
                float v = s[(size_t)y * W + x];
                d[(size_t)y * W + x] = v * ((m == v) ? 1.f : 0.f);
            }
    }
}

/* ------------------------------------------------------------------------- */
/* topK_channel (decoder/heatmap.py:38-49): per-plane top-K, sorted by value   */
/* descending.  Tie rule (torch leaves it unspecified): lower flat index      */
/* first; -0.0 == +0.0.  ys = idx // w, xs = idx % w are derived by callers   */
/* (floor division: the reference pins torch 1.3.1 integer-division).        */
/* ------------------------------------------------------------------------- */
typedef struct { float s; int64_t i; } ogo_ent;

static inline int ent_worse(const ogo_ent *a, const ogo_ent *b) /* a ranks after b */
{
    return (a->s < b->s) || (a->s == b->s && a->i > b->i);
}

static void heap_sift_down(ogo_ent *hp, int n, int i)
{
    for (;;) { /* min-heap on rank: root = worst kept entry */
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && ent_worse(&hp[l], &hp[m])) m = l;
        if (r < n && ent_worse(&hp[r], &hp[m])) m = r;
        if (m == i) return;
        ogo_ent t = hp[i]; hp[i] = hp[m]; hp[m] = t;
        i = m;
    }
}

static int ent_cmp_desc(const void *pa, const void *pb)
{
    const ogo_ent *a = (const ogo_ent *)pa, *b = (const ogo_ent *)pb;
    if (ent_worse(a, b)) return 1;
    if (ent_worse(b, a)) return -1;
    return 0;
}

OGO_API int ogo_topk(const float *sc, long planes, long hw, int k, float *out_s, int64_t *out_i)
{
    if (k <= 0 || hw < k) return -1;
    ogo_ent *hp = (ogo_ent *)malloc(sizeof(ogo_ent) * (size_t)k);
    for (long p = 0; p < planes; ++p) {
        const float *s = sc + (size_t)p * hw;
        int n = 0;
        for (long i = 0; i < hw; ++i) {
            float v = s[i];
            if (n < k) {
                hp[n].s = v; hp[n].i = i; ++n;
                if (n == k) for (int j = k / 2 - 1; j >= 0; --j) heap_sift_down(hp, k, j);
            } else if (v > hp[0].s) { /* equal value + larger index never wins */
                hp[0].s = v; hp[0].i = i;
                heap_sift_down(hp, k, 0);
            }
        }
        qsort(hp, (size_t)k, sizeof(ogo_ent), ent_cmp_desc);
        for (int j = 0; j < k; ++j) {
            out_s[(size_t)p * k + j] = hp[j].s;
            out_i[(size_t)p * k + j] = hp[j].i;
        }
    }
    free(hp);
    return 0;
}

/* joint_dets (decoder/heatmap.py:52-59) = topK_channel(hmp_NMS(hmps), k).   */
OGO_API int ogo_nms_topk(const float *hm, long planes, int H, int W, int k, float *out_s, int64_t *out_i)
{
    float *tmp = (float *)malloc(sizeof(float) * (size_t)H * W);
    int rc = 0;
    for (long p = 0; p < planes && rc == 0; ++p) {
        ogo_hmp_nms(hm + (size_t)p * H * W, 1, H, W, tmp);
        rc = ogo_topk(tmp, 1, (long)H * W, k, out_s + (size_t)p * k, out_i + (size_t)p * k);
    }
    free(tmp);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* LimbsCollect.generate_limbs (decoder/collect.py:62-236) with              */
/* _channel_dets (:246-254); no scale head / no jitter head (scales = 4).     */
/*   scores,inds : (N,C,K) per-channel top-K (value desc)                     */
/*   offs        : off_lowres ? (N,2L,H/4,W/4) bilinear-sampled (== gather    */
/*                 from the x4 bilinear map, factory.py:77-78) : (N,2L,H,W)   */
/*   limbs       : (N,L,K,13) [x1,y1,v1,x2,y2,v2,ind1,ind2,dist,len,score,s1,s2] */
/* ------------------------------------------------------------------------- */
/* vector_nd = 2: the usual guiding offsets; 4: the `cat_flip_offs` form (decoder/factory.py:115-127), where the
 * mirrored image's offsets ride along as components 2,3 and the match distance is the 4-D norm
 * sqrt(((d0^2 + d1^2) + d2^2) + d3^2) -- torch's rounding for a 4-element reduction (no fma), unlike the
 * 2-element case. */
OGO_API void ogo_collect_limbs_nd(const float *scores, const int64_t *inds, const float *offs, int off_lowres,
                                  int N, int C, int H, int W, const int *jf, const int *jt, int L, int K,
                                  float thre, float min_len, float resize, int vector_nd, float *limbs);

OGO_API void ogo_collect_limbs(const float *scores, const int64_t *inds, const float *offs, int off_lowres,
                               int N, int C, int H, int W, const int *jf, const int *jt, int L, int K,
                               float thre, float min_len, float resize, float *limbs)
{
    ogo_collect_limbs_nd(scores, inds, offs, off_lowres, N, C, H, W, jf, jt, L, K, thre, min_len, resize, 2, limbs);
}

/* scales_hr (N,C,H,W) or NULL: the keypoint-scale head brought to input resolution by F.interpolate(mode=inter_mode)
 * (decoder/factory.py:80-82); limbs columns 11 / 12 = its value in the from / to joint channel at the from-peak / the
 * matched to-peak (decoder/collect.py:111-122, :257-262), the constant 4 without the head. */
OGO_API void ogo_collect_limbs_ex(const float *scores, const int64_t *inds, const float *offs, int off_lowres,
                                  const float *scales_hr, int N, int C, int H, int W, const int *jf, const int *jt, int L,
                                  int K, float thre, float min_len, float resize, int vector_nd, float *limbs);

OGO_API void ogo_collect_limbs_nd(const float *scores, const int64_t *inds, const float *offs, int off_lowres,
                                  int N, int C, int H, int W, const int *jf, const int *jt, int L, int K,
                                  float thre, float min_len, float resize, int vector_nd, float *limbs)
{
    ogo_collect_limbs_ex(scores, inds, offs, off_lowres, NULL, N, C, H, W, jf, jt, L, K, thre, min_len, resize, vector_nd, limbs);
}

/* jitter_hr (N,2,H,W) or NULL: the jitter-offset head at input resolution (F.interpolate bilinear, factory.py:84-88).
 * With it (include_jitter_offset) and use_jitter: the guide point is refined by the jitter vector read at its truncated
 * coordinates -- indexed [x][y] as the reference does (collect.py:158-165: jomps_hr[i, :, xy[0], xy[1]] with xy = (x, y),
 * in range when 0 <= x < w and 0 <= y < h) -- and after matching the end points move by their own jitter vectors
 * (:210-214); the limb length (:203) uses the unmoved coordinates. */
OGO_API void ogo_collect_limbs_jit(const float *scores, const int64_t *inds, const float *offs, int off_lowres,
                                   const float *scales_hr, const float *jitter_hr, int use_jitter, int N, int C, int H,
                                   int W, const int *jf, const int *jt, int L, int K, float thre, float min_len,
                                   float resize, int vector_nd, float *limbs);

OGO_API void ogo_collect_limbs_ex(const float *scores, const int64_t *inds, const float *offs, int off_lowres,
                                  const float *scales_hr, int N, int C, int H, int W, const int *jf, const int *jt, int L,
                                  int K, float thre, float min_len, float resize, int vector_nd, float *limbs)
{
    ogo_collect_limbs_jit(scores, inds, offs, off_lowres, scales_hr, NULL, 0, N, C, H, W, jf, jt, L, K, thre, min_len, resize,
                          vector_nd, limbs);
}

OGO_API void ogo_collect_limbs_jit(const float *scores, const int64_t *inds, const float *offs, int off_lowres,
                                   const float *scales_hr, const float *jitter_hr, int use_jitter, int N, int C, int H,
                                   int W, const int *jf, const int *jt, int L, int K, float thre, float min_len,
                                   float resize, int vector_nd, float *limbs)
{
    const int nd = vector_nd;
    const long HW = (long)H * W;
    const int h4 = H / 4, w4 = W / 4;
    float *tx = (float *)malloc(sizeof(float) * (size_t)K * 2);
    float *ty = tx + K;
    for (int n = 0; n < N; ++n)
        for (int l = 0; l < L; ++l) {
            const float *sf = scores + ((size_t)n * C + jf[l]) * K;
            const float *st = scores + ((size_t)n * C + jt[l]) * K;
            const int64_t *idf = inds + ((size_t)n * C + jf[l]) * K;
            const int64_t *idt = inds + ((size_t)n * C + jt[l]) * K;
            for (int m = 0; m < K; ++m) { /* to-candidates (collect.py:109-110, :246-254) */
                int64_t x = idt[m] % W, y = idt[m] / W;
                if (st[m] < thre) { x -= 100000; y -= 100000; }
                tx[m] = (float)x; ty[m] = (float)y;
            }
            for (int k = 0; k < K; ++k) {
                int64_t xi = idf[k] % W, yi = idf[k] / W;
                int64_t xs = xi, ys = yi;
                if (sf[k] < thre) { xs -= 100000; ys -= 100000; }
                float xf = (float)xs, yf = (float)ys;
                float o4[4] = {0, 0, 0, 0}; /* gather at the ORIGINAL flat index (collect.py:143-147) */
                for (int c = 0; c < nd; ++c) {
                    if (off_lowres) {
                        const float *px = offs + ((size_t)n * nd * L + (size_t)nd * l + c) * h4 * w4;
                        o4[c] = ogo_bilinear4_at(px, h4, w4, (int)yi, (int)xi);
                    } else {
                        o4[c] = offs[((size_t)n * nd * L + (size_t)nd * l + c) * HW + idf[k]];
                    }
                }
                float gx = xf + o4[0] * resize, gy = yf + o4[1] * resize; /* :152 */
                float gx2 = xf + o4[2] * resize, gy2 = yf + o4[3] * resize;
                if (jitter_hr && use_jitter) { /* :158-165 */
                    const int qx = (int)gx, qy = (int)gy; /* .int(): truncation */
                    if (qx >= 0 && qx < W && qy >= 0 && qy < H) {
                        gx += jitter_hr[((size_t)n * 2 + 0) * HW + (size_t)qx * W + qy];
                        gy += jitter_hr[((size_t)n * 2 + 1) * HW + (size_t)qx * W + qy];
                    }
                }
                int best = 0;
                float bd = INFINITY;
                for (int m = 0; m < K; ++m) { /* :171-177, first minimum */
                    float dx = gx - tx[m], dy = gy - ty[m];
                    float d;
                    if (nd == 2) {
                        d = sqrtf(fmaf(dy, dy, dx * dx));
                    } else {
                        float dx2 = gx2 - tx[m], dy2 = gy2 - ty[m];
                        d = sqrtf(((dx * dx + dy * dy) + dx2 * dx2) + dy2 * dy2);
                    }
                    if (d < bd) { bd = d; best = m; }
                }
                float lx = xf - tx[best], ly = yf - ty[best];
                float len = sqrtf(fmaf(ly, ly, lx * lx));
                if (len < min_len) len = min_len; /* :204-205 */
                float sc = (sf[k] * st[best]) * expf(-bd / len); /* :208 */
                float *o = limbs + (((size_t)n * L + l) * K + k) * 13;
                float x1 = xf, y1 = yf, x2 = tx[best], y2 = ty[best];
                if (jitter_hr && use_jitter) { /* :210-214 */
                    x1 += jitter_hr[((size_t)n * 2 + 0) * HW + idf[k]];
                    y1 += jitter_hr[((size_t)n * 2 + 1) * HW + idf[k]];
                    x2 += jitter_hr[((size_t)n * 2 + 0) * HW + idt[best]];
                    y2 += jitter_hr[((size_t)n * 2 + 1) * HW + idt[best]];
                }
                o[0] = x1; o[1] = y1; o[2] = sf[k];
                o[3] = x2; o[4] = y2; o[5] = st[best];
                o[6] = (float)(idf[k] + (int64_t)jf[l] * HW);
                o[7] = (float)(idt[best] + (int64_t)jt[l] * HW);
                o[8] = bd; o[9] = len; o[10] = sc;
                o[11] = scales_hr ? scales_hr[((size_t)n * C + jf[l]) * HW + idf[k]] : 4.f;
                o[12] = scales_hr ? scales_hr[((size_t)n * C + jt[l]) * HW + idt[best]] : 4.f;
            }
        }
    free(tx);
}

/* ------------------------------------------------------------------------- */
/* GreedyGroup.group_skeletons (decoder/group.py:39-185) for ONE image, with   */
/* _delete_reconns (:221-240) and _delete_sort (:187-219).  Literal loop form */
/* of the numpy fancy-assignment semantics: every vectorised statement        */
/* gathers its right-hand side from the state before the statement, then      */
/* scatters in row-major pair order (last write wins).  Sorts are stable.     */
/* Returns number of poses written (<= mmax), or -1 if mmax is too small.     */
/* ------------------------------------------------------------------------- */
#define SUB(m, j, f) sub[((size_t)(m) * nkp + (j)) * 6 + (f)]

/* coverage counters for the fuzz harness: [0] phase-A events, [1] phase-B events,      */
/* [2] phase-B events with a duplicated row, [3] merges, [4] >=3-joint crossings,        */
/* [5] non-replaced ms==2 resets, [6] duplicate-`a` merges, [7] merged-into row deleted, */
/* [8] zero-sum non-empty columns                                                        */
static long g_stats[9];
OGO_API void ogo_group_stats(long *out, int reset)
{
    for (int i = 0; i < 9; ++i) { out[i] = g_stats[i]; if (reset) g_stats[i] = 0; }
}

static float np_sum_f32(const float *a, int n) /* numpy pairwise sum for n <= 128 */
{
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    float r[8];
    int i;
    for (i = 0; i < 8; ++i) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

OGO_API int ogo_greedy_group(const float *limbs, int L, int K, const int *jf_, const int *jt_, int nkp,
                             double person_thre, float dist_max, int use_scale, int sort_dim, int mmax,
                             float *poses)
{
    const size_t rowf = (size_t)nkp * 6;
    int cap = 64, mm = 0;
    float *sub = (float *)malloc(sizeof(float) * rowf * cap);
    int *ord = (int *)malloc(sizeof(int) * K);
    int *uniq = (int *)malloc(sizeof(int) * K);
    int *ms = NULL;
    char *rep = NULL;
    int ok = 1;

    for (int l = 0; l < L; ++l) {
        const int jf = jf_[l], jt = jt_[l];
        const float *cn = limbs + (size_t)l * K * 13;
        /* validity filter (:64-76) */
        int nv = 0;
        for (int k = 0; k < K; ++k) {
            const float *c = cn + (size_t)k * 13;
            float lim = use_scale ? (dist_max > c[12] ? dist_max : c[12]) : dist_max;
            if (c[8] < lim && c[0] > 0.f && c[4] > 0.f && c[3] > 0.f && c[1] > 0.f) ord[nv++] = k;
        }
        /* stable sort by score descending (:232) */
        for (int a = 1; a < nv; ++a) {
            int v = ord[a], b = a - 1;
            while (b >= 0 && cn[(size_t)ord[b] * 13 + 10] < cn[(size_t)v * 13 + 10]) { ord[b + 1] = ord[b]; --b; }
            ord[b + 1] = v;
        }
        /* keep first occurrence per to-index (:233-239) */
        int kk = 0;
        for (int a = 0; a < nv; ++a) {
            int64_t t = (int64_t)cn[(size_t)ord[a] * 13 + 7];
            int dup = 0;
            for (int b = 0; b < kk; ++b)
                if ((int64_t)cn[(size_t)uniq[b] * 13 + 7] == t) { dup = 1; break; }
            if (!dup) uniq[kk++] = ord[a];
        }
        if (kk == 0) continue;
#define CN(c, f) cn[(size_t)uniq[c] * 13 + (f)]
        const int m0 = mm; /* rows before this limb type */
        ms = (int *)realloc(ms, sizeof(int) * (size_t)(m0 ? m0 : 1) * kk);
        rep = (char *)realloc(rep, (size_t)(m0 ? m0 : 1) * kk);
        /* snapshot-based match matrices (:87-109) */
        for (int m = 0; m < m0; ++m)
            for (int c = 0; c < kk; ++c) {
                int64_t idf = (int64_t)SUB(m, jf, 5), idt = (int64_t)SUB(m, jt, 5);
                ms[(size_t)m * kk + c] = (idf == (int64_t)CN(c, 6)) + (idt == (int64_t)CN(c, 7));
                rep[(size_t)m * kk + c] = (CN(c, 10) > SUB(m, jt, 4)) || (CN(c, 10) > SUB(m, jf, 4));
            }
        const size_t np_ = (size_t)m0 * kk;
        float *rhs = (float *)malloc(sizeof(float) * (np_ ? np_ : 1) * 4);
        /* phase A (:114-119) */
        {
            size_t cnt = 0;
            for (size_t e = 0; e < np_; ++e) cnt += (ms[e] == 2 && rep[e]);
            if (cnt) {
                const int jj[2] = {jf, jt};
                ++g_stats[0];
                for (size_t e = 0; e < np_; ++e) g_stats[5] += (ms[e] == 2 && !rep[e]);
                for (int s = 0; s < 2; ++s) {
                    size_t i = 0;
                    for (int m = 0; m < m0; ++m)
                        for (int c = 0; c < kk; ++c)
                            if (ms[(size_t)m * kk + c] == 2 && rep[(size_t)m * kk + c]) {
                                float a = CN(c, 10), b = SUB(m, jj[s], 4);
                                rhs[i++] = a > b ? a : b;
                            }
                    i = 0;
                    for (int m = 0; m < m0; ++m)
                        for (int c = 0; c < kk; ++c)
                            if (ms[(size_t)m * kk + c] == 2 && rep[(size_t)m * kk + c]) SUB(m, jj[s], 4) = rhs[i++];
                }
                for (size_t e = 0; e < np_; ++e) if (ms[e] == 2) ms[e] = -1;
            }
        }
        /* phase B (:124-135): six statements in order */
        {
            size_t cnt = 0;
            for (size_t e = 0; e < np_; ++e) cnt += (ms[e] == 1 && rep[e]);
            if (cnt) {
                ++g_stats[1];
                for (int m = 0; m < m0; ++m) {
                    int r = 0;
                    for (int c = 0; c < kk; ++c) r += (ms[(size_t)m * kk + c] == 1 && rep[(size_t)m * kk + c]);
                    g_stats[2] += (r > 1);
                }
#define FOR_PAIRS for (int m = 0; m < m0; ++m) for (int c = 0; c < kk; ++c) if (ms[(size_t)m * kk + c] == 1 && rep[(size_t)m * kk + c])
                FOR_PAIRS SUB(m, jf, 5) = CN(c, 6);
                FOR_PAIRS SUB(m, jt, 5) = CN(c, 7);
                FOR_PAIRS { SUB(m, jf, 0) = CN(c, 0); SUB(m, jf, 1) = CN(c, 1); SUB(m, jf, 2) = CN(c, 2); SUB(m, jf, 3) = CN(c, 11); }
                FOR_PAIRS { SUB(m, jt, 0) = CN(c, 3); SUB(m, jt, 1) = CN(c, 4); SUB(m, jt, 2) = CN(c, 5); SUB(m, jt, 3) = CN(c, 12); }
                const int jj[2] = {jf, jt};
                for (int s = 0; s < 2; ++s) {
                    size_t i = 0;
                    FOR_PAIRS { float a = CN(c, 10), b = SUB(m, jj[s], 4); rhs[i++] = a > b ? a : b; }
                    i = 0;
                    FOR_PAIRS SUB(m, jj[s], 4) = rhs[i++];
                }
#undef FOR_PAIRS
                for (size_t e = 0; e < np_; ++e) if (ms[e] == 1) ms[e] = -1;
            }
        }
        free(rhs);
        /* phase C: merge rows sharing exactly two keypoints (:140-161) */
        if (m0 >= 2) {
            int npair = 0;
            int *pa = (int *)malloc(sizeof(int) * (size_t)m0 * m0 * 2);
            for (int a = 0; a < m0; ++a)
                for (int b = a + 1; b < m0; ++b) {
                    int cnt = 0;
                    for (int j = 0; j < nkp; ++j) {
                        int64_t ia = (int64_t)SUB(a, j, 5), ib = (int64_t)SUB(b, j, 5);
                        cnt += (ia == ib && ia != -1);
                    }
                    if (cnt == 2) { pa[2 * npair] = a; pa[2 * npair + 1] = b; ++npair; }
                    g_stats[4] += (cnt >= 3);
                }
            if (npair) {
                g_stats[3] += npair;
                for (int p = 0; p < npair; ++p)
                    for (int q = 0; q < npair; ++q) {
                        g_stats[6] += (q > p && pa[2 * q] == pa[2 * p]);
                        g_stats[7] += (pa[2 * q + 1] == pa[2 * p]);
                    }
                float *mx = (float *)malloc(sizeof(float) * rowf * npair);
                for (int p = 0; p < npair; ++p)
                    for (size_t f = 0; f < rowf; ++f) {
                        float a = sub[pa[2 * p] * rowf + f], b = sub[pa[2 * p + 1] * rowf + f];
                        mx[p * rowf + f] = a > b ? a : b; /* no NaNs on this path */
                    }
                for (int p = 0; p < npair; ++p) memcpy(sub + pa[2 * p] * rowf, mx + p * rowf, sizeof(float) * rowf);
                free(mx);
                char *del = (char *)calloc((size_t)m0, 1);
                for (int p = 0; p < npair; ++p) del[pa[2 * p + 1]] = 1;
                int w = 0;
                for (int m = 0; m < m0; ++m)
                    if (!del[m]) { if (w != m) memcpy(sub + w * rowf, sub + m * rowf, sizeof(float) * rowf); ++w; }
                mm = w;
                free(del);
            }
            free(pa);
        }
        /* phase D: unmatched limbs start new skeletons (:166-177); column sums */
        /* run over the m0 rows that existed before the merge.                  */
        for (int c = 0; c < kk; ++c) {
            int s = 0;
            int nz = 0;
            for (int m = 0; m < m0; ++m) { s += ms[(size_t)m * kk + c]; nz += (ms[(size_t)m * kk + c] != 0); }
            if (s != 0) continue;
            g_stats[8] += (nz > 0);
            if (mm == cap) { cap *= 2; sub = (float *)realloc(sub, sizeof(float) * rowf * cap); }
            float *r = sub + mm * rowf;
            for (size_t f = 0; f < rowf; ++f) r[f] = -1.f;
            r[jf * 6 + 5] = CN(c, 6); r[jt * 6 + 5] = CN(c, 7);
            r[jf * 6 + 0] = CN(c, 0); r[jf * 6 + 1] = CN(c, 1); r[jf * 6 + 2] = CN(c, 2); r[jf * 6 + 3] = CN(c, 11);
            r[jt * 6 + 0] = CN(c, 3); r[jt * 6 + 1] = CN(c, 4); r[jt * 6 + 2] = CN(c, 5); r[jt * 6 + 3] = CN(c, 12);
            r[jf * 6 + 4] = CN(c, 10); r[jt * 6 + 4] = CN(c, 10);
            ++mm;
        }
#undef CN
    }

    /* _delete_sort (:187-219) */
    double *ps = (double *)malloc(sizeof(double) * (mm ? mm : 1));
    int *keep = (int *)malloc(sizeof(int) * (mm ? mm : 1));
    int nk = 0;
    float *vals = (float *)malloc(sizeof(float) * nkp);
    for (int m = 0; m < mm; ++m) {
        int n = 0;
        for (int j = 0; j < nkp; ++j)
            if (SUB(m, j, sort_dim) > 0.f) vals[n++] = SUB(m, j, sort_dim);
        double score = (double)np_sum_f32(vals, n) / (double)n; /* 0/0 = NaN is kept, as in the reference */
        if (score < person_thre) continue;
        ps[nk] = score; keep[nk] = m; ++nk;
    }
    /* stable sort, descending (python sorted(reverse=True) keeps ties in order) */
    for (int a = 1; a < nk; ++a) {
        double sv = ps[a]; int kv = keep[a], b = a - 1;
        while (b >= 0 && ps[b] < sv) { ps[b + 1] = ps[b]; keep[b + 1] = keep[b]; --b; }
        ps[b + 1] = sv; keep[b + 1] = kv;
    }
    if (nk > mmax) ok = 0;
    for (int a = 0; a < nk && ok; ++a)
        for (size_t f = 0; f < rowf; ++f) {
            float v = sub[keep[a] * rowf + f];
            poses[a * rowf + f] = (v == -1.f) ? 0.f : v;
        }
    free(vals); free(keep); free(ps); free(ms); free(rep); free(uniq); free(ord); free(sub);
    return ok ? nk : -1;
}
#undef SUB

/* ------------------------------------------------------------------------- */
/* PostProcess.flip_augment (decoder/factory.py:98-146), vector-addition form. */
/*   hm  (2N,C,h,w) -> (N,C,h,w):  (orig + flipW(flip)[kp_perm]) / 2          */
/*   off (2N,2L,h,w) -> (N,2L,h,w): x components of the flipped half negated,  */
/*   limb-permuted, averaged; limbs in `reserve` keep the un-averaged original */
/* ------------------------------------------------------------------------- */
OGO_API void ogo_flip_merge(const float *hm, const float *off, int N, int C, int L, int h, int w,
                            const int *kp_perm, const int *limb_perm, const int *reserve, int n_reserve,
                            float *hm_out, float *off_out)
{
    const size_t hw = (size_t)h * w;
    for (int n = 0; n < N; ++n) {
        for (int c = 0; c < C; ++c) {
            const float *a = hm + ((size_t)n * C + c) * hw;
            const float *b = hm + ((size_t)(n + N) * C + kp_perm[c]) * hw;
            float *o = hm_out + ((size_t)n * C + c) * hw;
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) o[(size_t)y * w + x] = (a[(size_t)y * w + x] + b[(size_t)y * w + (w - 1 - x)]) / 2.f;
        }
        for (int l = 0; l < L; ++l) {
            int keep = 0;
            for (int r = 0; r < n_reserve; ++r) keep |= (reserve[r] == l);
            for (int comp = 0; comp < 2; ++comp) {
                const float *a = off + ((size_t)n * 2 * L + 2 * l + comp) * hw;
                const float *b = off + ((size_t)(n + N) * 2 * L + 2 * limb_perm[l] + comp) * hw;
                float *o = off_out + ((size_t)n * 2 * L + 2 * l + comp) * hw;
                for (int y = 0; y < h; ++y)
                    for (int x = 0; x < w; ++x) {
                        float fv = b[(size_t)y * w + (w - 1 - x)];
                        if (comp == 0) fv = fv * -1.0f;
                        o[(size_t)y * w + x] = keep ? a[(size_t)y * w + x] : (a[(size_t)y * w + x] + fv) / 2.f;
                    }
            }
        }
    }
}

/* PostProcess.flip_augment with cat_flip_offs=True (decoder/factory.py:115-127): the mirrored offsets are
 * appended as components 2,3 instead of being averaged; for the `reserve` limbs (mirror = own reverse)
 * components 2,3 are the original components 0,1.  off_out (N, L, 4, h, w). */
OGO_API void ogo_flip_cat(const float *hm, const float *off, int N, int C, int L, int h, int w,
                          const int *kp_perm, const int *limb_perm, const int *reserve, int n_reserve,
                          float *hm_out, float *off_out)
{
    const size_t hw = (size_t)h * w;
    for (int n = 0; n < N; ++n) {
        for (int c = 0; c < C; ++c) {
            const float *a = hm + ((size_t)n * C + c) * hw;
            const float *b = hm + ((size_t)(n + N) * C + kp_perm[c]) * hw;
            float *o = hm_out + ((size_t)n * C + c) * hw;
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) o[(size_t)y * w + x] = (a[(size_t)y * w + x] + b[(size_t)y * w + (w - 1 - x)]) / 2.f;
        }
        for (int l = 0; l < L; ++l) {
            int keep = 0;
            for (int r = 0; r < n_reserve; ++r) keep |= (reserve[r] == l);
            for (int comp = 0; comp < 2; ++comp) {
                const float *a = off + ((size_t)n * 2 * L + 2 * l + comp) * hw;
                const float *b = off + ((size_t)(n + N) * 2 * L + 2 * limb_perm[l] + comp) * hw;
                float *o0 = off_out + ((size_t)n * 4 * L + 4 * l + comp) * hw;
                float *o1 = off_out + ((size_t)n * 4 * L + 4 * l + 2 + comp) * hw;
                for (int y = 0; y < h; ++y)
                    for (int x = 0; x < w; ++x) {
                        float fv = b[(size_t)y * w + (w - 1 - x)];
                        if (comp == 0) fv = fv * -1.0f;
                        o0[(size_t)y * w + x] = a[(size_t)y * w + x];
                        o1[(size_t)y * w + x] = keep ? a[(size_t)y * w + x] : fv;
                    }
            }
        }
    }
}

OGO_API int ogo_version(void) { return 1; }

/* ===================================================================================================
 * Ground-truth encoder: encoder/heatmap.py:125-197 and encoder/offset.py:98-197.
 * joints (P,17,4) fp32 rows [x, y, v, scale] in input-image pixels (transforms/annotations.py:46-50).
 * All arithmetic is fp32 as in the reference (numpy float32 arrays with weak Python scalars); Python's round()
 * on an np.float32 is round-half-even = rintf.
 * =================================================================================================== */
static int ogo_patch(float c, int stride, float size, int *lo, int *hi)
{
    /* x_min = int(round(c / stride - size / 2)), x_max = int(round(c / stride + size / 2)); negative max: skipped;
     * negative min: clamped (encoder/heatmap.py:154-171) */
    float q = c / (float)stride, half = size / 2.f;
    *lo = (int)rintf(q - half);
    *hi = (int)rintf(q + half);
    if (*hi < 0) return 0;
    if (*lo < 0) *lo = 0;
    return 1;
}

/* encoder/heatmap.py:141-197 (create_heatmaps / put_gaussian_peaks): hm (n_kp, h, w), max over the persons of the
 * separable Gaussian exp_y[y]*exp_x[x] clipped below clip_thre, inside each person's window only. */
OGO_API void ogo_encode_heatmaps(const float *joints, int P, int n_kp, int in_w, int in_h, int stride, int sigma,
                                 float clip_thre, float *hm)
{
    const int w = in_w / stride, h = in_h / stride;
    const double ds2 = 2.0 * sigma * sigma;
    const int gsize = 2 * (int)ceil(sqrt(-ds2 * log((double)clip_thre)) / stride); /* heatmap.py:110-111 */
    const float ds2f = (float)ds2;
    memset(hm, 0, sizeof(float) * (size_t)n_kp * h * w);
    for (int c = 0; c < n_kp; ++c)
        for (int p = 0; p < P; ++p) {
            const float *j = joints + ((size_t)p * n_kp + c) * 4;
            if (!(j[2] > 0)) continue;
            int x0, x1, y0, y1;
            if (!ogo_patch(j[1], stride, (float)gsize, &y0, &y1)) continue; /* y_max < 0 is tested first (:159) */
            if (!ogo_patch(j[0], stride, (float)gsize, &x0, &x1)) continue;
            if (x1 > w) x1 = w; /* slices crop at the array border */
            if (y1 > h) y1 = h;
            for (int y = y0; y < y1; ++y) {
                const float gy = (float)(y * stride + stride / 2.0 - 0.5), dy = gy - j[1];
                const float ey = expf(-(dy * dy) / ds2f);
                for (int x = x0; x < x1; ++x) {
                    const float gx = (float)(x * stride + stride / 2.0 - 0.5), dx = gx - j[0];
                    const float ex = expf(-(dx * dx) / ds2f);
                    float e = ey * ex;
                    if (e < clip_thre) e = 0.f;
                    float *o = hm + ((size_t)c * h + y) * w + x;
                    if (e > *o) *o = e;
                }
            }
        }
}

/* encoder/heatmap.py:199-255 (create_jitter_offset / put_jitter_maps): two shared channels holding, inside a
 * fill_jitter_size window round every annotated keypoint (all channels, channel-major then person order), the vector from
 * the cell centre to that keypoint; overlapping windows keep the shorter vector (strict <).  jit (2,h,w) init inf. */
OGO_API void ogo_encode_jitter(const float *joints, int P, int n_kp, int in_w, int in_h, int stride, int fill_size,
                               float *jit)
{
    const int w = in_w / stride, h = in_h / stride;
    const size_t hw = (size_t)h * w;
    for (size_t i = 0; i < 2 * hw; ++i) jit[i] = INFINITY;
    for (int c = 0; c < n_kp; ++c)
        for (int p = 0; p < P; ++p) {
            const float *j = joints + ((size_t)p * n_kp + c) * 4;
            if (!(j[2] > 0)) continue;
            int x0, x1, y0, y1;
            if (!ogo_patch(j[1], stride, (float)fill_size, &y0, &y1)) continue;
            if (!ogo_patch(j[0], stride, (float)fill_size, &x0, &x1)) continue;
            if (x1 > w) x1 = w;
            if (y1 > h) y1 = h;
            for (int y = y0; y < y1; ++y) {
                const float oy = j[1] - (float)(y * stride + stride / 2.0 - 0.5);
                for (int x = x0; x < x1; ++x) {
                    const float ox = j[0] - (float)(x * stride + stride / 2.0 - 0.5);
                    float *px = jit + (size_t)y * w + x, *py = px + hw;
                    if (sqrtf(ox * ox + oy * oy) < sqrtf(*px * *px + *py * *py)) { *px = ox; *py = oy; }
                }
            }
        }
}

/* encoder/offset.py:98-197 (create_offsetmaps / put_guide_offsets): per limb (fr, to) and person with both joints
 * annotated, a fill_size window round the from-joint holds the vector to the to-joint; where windows overlap the
 * shorter vector wins (strict <: the earlier person keeps ties).  off (2L,h,w) init inf, scale (n_kp,h,w) init nan,
 * pscale (2L,h,w) init 1. */
OGO_API void ogo_encode_offsets(const float *joints, int P, int n_kp, const int *jf, const int *jt, int L, int in_w,
                                int in_h, int stride, int fill_size, float min_jscale, const float *sigmas,
                                float *off, float *scale, float *pscale)
{
    const int w = in_w / stride, h = in_h / stride;
    const size_t hw = (size_t)h * w;
    for (size_t i = 0; i < 2 * L * hw; ++i) { off[i] = INFINITY; pscale[i] = 1.f; }
    for (size_t i = 0; i < n_kp * hw; ++i) scale[i] = NAN;
    for (int l = 0; l < L; ++l)
        for (int p = 0; p < P; ++p) {
            const float *j1 = joints + ((size_t)p * n_kp + jf[l]) * 4, *j2 = joints + ((size_t)p * n_kp + jt[l]) * 4;
            if (!(j1[2] > 0 && j2[2] > 0)) continue;
            int x0, x1, y0, y1;
            if (!ogo_patch(j1[1], stride, (float)fill_size, &y0, &y1)) continue;
            if (!ogo_patch(j1[0], stride, (float)fill_size, &x0, &x1)) continue;
            if (x1 > w) x1 = w;
            if (y1 > h) y1 = h;
            for (int y = y0; y < y1; ++y) {
                const float oy = j2[1] - (float)(y * stride + stride / 2.0 - 0.5);
                for (int x = x0; x < x1; ++x) {
                    const float ox = j2[0] - (float)(x * stride + stride / 2.0 - 0.5);
                    const float len = sqrtf(ox * ox + oy * oy); /* np.linalg.norm over 2 fp32 elements */
                    float *px = off + ((size_t)(2 * l) * h + y) * w + x, *py = px + hw;
                    const float cur = sqrtf(*px * *px + *py * *py);
                    if (len < cur) {
                        *px = ox;
                        *py = oy;
                        scale[((size_t)jf[l] * h + y) * w + x] = j1[3] >= min_jscale ? j1[3] : NAN;
                        const float ps = j1[3] / sigmas[jf[l]];
                        pscale[((size_t)(2 * l) * h + y) * w + x] = ps;
                        pscale[((size_t)(2 * l + 1) * h + y) * w + x] = ps;
                    }
                }
            }
        }
}

/* ---- transforms/scale.py:14-31, :75-98 (RescaleLongAbsolute -> cv2.resize(image, (w, h), INTER_CUBIC)) ----
 * THIRD-PARTY ARITHMETIC, PARITY UNPINNED: cv2 (opencv-python==3.4.5.20, requirment.txt:91) is absent from the build
 * container and not vendored by the reference, so no golden vector exists.  This restates the published algorithm of
 * cv::resize for 8-bit images (OpenCV 3.4 modules/imgproc/src/resize.cpp: resizeGeneric_ with HResizeCubic<uchar,int,short>
 * and VResizeCubic<uchar,int,short,FixedPtCast<int,uchar,22>>):
 *   scale = src / dst (double); per destination index d: f = (float)((d + 0.5) * scale - 0.5), s = floor(f), f -= s;
 *   cubic weights (A = -0.75) in float, fixed-point taps = round-half-even(w * 2048) as short;
 *   horizontal pass in int over taps s-1 .. s+2 (indices clamped to the image = replicated border);
 *   vertical pass over rows s-1 .. s+2 (clamped), result = saturate_u8((sum + 2^21) >> 22).
 * (OpenCV's SSE2 vertical pass evaluates the same sum in float and rounds to nearest even, which can differ from this
 * fixed-point definition by one grey level in rare pixels; neither can be checked here.) */
static void ogo_cubic_taps(float x, short *t)
{
    const float A = -0.75f;
    float c[4];
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
    for (int k = 0; k < 4; ++k) {
        long v = lrintf(c[k] * 2048.f); /* saturate_cast<short>(float): cvRound, round half to even */
        t[k] = (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v);
    }
}

static int ogo_clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

static void ogo_resize_cubic_u8_ch(const unsigned char *src, int h, int w, unsigned char *dst, int nh, int nw, int ch)
{
    const double sx = (double)w / nw, sy = (double)h / nh;
    for (int dy = 0; dy < nh; ++dy) {
        float fy = (float)((dy + 0.5) * sy - 0.5);
        const int y0 = (int)floorf(fy);
        short by[4];
        ogo_cubic_taps(fy - (float)y0, by);
        for (int dx = 0; dx < nw; ++dx) {
            float fx = (float)((dx + 0.5) * sx - 0.5);
            const int x0 = (int)floorf(fx);
            short ax[4];
            ogo_cubic_taps(fx - (float)x0, ax);
            for (int c = 0; c < ch; ++c) {
                int acc = 0;
                for (int r = 0; r < 4; ++r) {
                    const unsigned char *row = src + (size_t)ogo_clampi(y0 - 1 + r, 0, h - 1) * w * ch;
                    int hs = 0;
                    for (int k = 0; k < 4; ++k) hs += row[ogo_clampi(x0 - 1 + k, 0, w - 1) * ch + c] * ax[k];
                    acc += hs * by[r];
                }
                const int v = (acc + (1 << 21)) >> 22;
                dst[((size_t)dy * nw + dx) * ch + c] = (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
        }
    }
}

OGO_API void ogo_resize_cubic_u8(const unsigned char *src, int h, int w, unsigned char *dst, int nh, int nw)
{
    ogo_resize_cubic_u8_ch(src, h, w, dst, nh, nw, 3);
}

/* encoder/heatmap.py:56-60, encoder/offset.py:46-50: the full-resolution uint8 mask_miss (h, w), 0 / 255, shrunk by
 * cv2.resize(fx = fy = 1 / stride, INTER_CUBIC), / 255, > 0.7 -> bool (h / stride, w / stride).  float32(v) / 255 > 0.7
 * <=> v >= 179 for integer v.  Same published 8-bit algorithm as above (parity with cv2 itself unpinned). */
OGO_API void ogo_shrink_mask_miss_u8(const unsigned char *mask, int h, int w, unsigned char *out, int nh, int nw)
{
    ogo_resize_cubic_u8_ch(mask, h, w, out, nh, nw, 1);
    for (long i = 0; i < (long)nh * nw; ++i) out[i] = (float)out[i] / 255.f > 0.7f ? 1 : 0;
}
