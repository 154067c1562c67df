// K1 -- 3x3 max-NMS + per-plane top-k over hi-res fp32 heatmaps, one streaming pass.
//
// Replaces decoder/heatmap.py:15-59 (hmp_NMS -> topK_channel = joint_dets): the reference
// makes ~10 full-tensor passes (pad, max_pool2d, ==, .float(), *, topk); here every element
// is read from HBM once (plus 2 halo rows per band) and nothing but k candidates per band
// is written.
//
// Decomposition (wave-centric, 64-wide):
//   plane (H x W) -> bands of `rows` rows -> one workgroup per band, one wave per column
//   panel.  A lane owns VEC consecutive columns (float4 when W % 4 == 0) and walks down the
//   band keeping a 3-row window of horizontal maxima in registers; left/right neighbours come
//   from the adjacent lanes by DPP wave shifts, so a panel is 62 interior lanes + 2 halo
//   lanes and no LDS or extra loads are needed for the stencil.
//   Loads are buffer loads through a per-plane descriptor: rows above/below the plane and
//   lanes left/right of it get an out-of-range offset and read as 0.0 -- exactly the zero
//   padding of F.pad -- with no branch and no select, so the compiler keeps PREFETCH rows
//   (16 B/lane each) in flight with counted vmcnt waits.  The kernel is instruction-issue
//   sensitive (not just HBM-bound), hence the diet: one integer compare implements
//   "v > 0 and v >= tau" (positive floats order like their bit patterns) and halo/idle lanes
//   carry tau = INT_MAX instead of a separate mask.
//   Candidates (positive 3x3 peaks, or every pixel in plain top-k mode) are compacted with
//   the compare masks themselves (they are the ballots) + mbcnt into a per-wave LDS segment;
//   when a segment fills, the wave keeps its own top-k (rank-by-counting on order-preserving
//   64-bit keys, ping-pong buffers) and raises its admission threshold, so the result is
//   exact for any input.  Each band emits its k best keys; a second tiny kernel (one wave per
//   plane) selects the plane's top-k from the band lists and writes scores/indices.
//   Admission threshold (NMS mode).  Each workgroup also counts its candidates in a 256-bin LDS
//   histogram (bins = exponent + 3 mantissa bits of the score, ds_add); after every PF rows one
//   wave scans it with a DPP prefix sum and raises tau to the lower edge of the bin that holds
//   the band's k-th best so far.  Across bands, a finished workgroup publishes the edge of its
//   t-th best (t = k/4) into a per-plane slot table; a starting workgroup reads the table once
//   and may start from the 4th largest published edge (4 bands x t >= k candidates lie above
//   it).  Every bound is a true lower bound on the plane's k-th best whatever the timing, so no
//   ordering protocol is needed and the top-k stays exact.  Measured lessons (MI355X):
//     * with a perfect threshold the kernel streams at 4.9 TB/s (45 us for 223 MB), with none at
//       3.9 TB/s: candidate handling, not HBM, is the gap;
//     * ANY conditional vector-memory operation inside the streaming loop (histogram atomics,
//       threshold loads) makes the compiler drain the prefetch queue (vmcnt(0)) every iteration,
//       ~5 us each under load -- hence LDS-only refreshes and start/end-only global traffic;
//     * rank-counting loops must read LDS wide (4 x ds_read_b128 per step, og_count_greater).
#include <math.h>

#include "bicubic.h"
#ifdef OG_K1_STAMPS   // tuning harness: time points inside the limb pairing (value-dependent, so they cannot be hoisted)
__device__ long long g_k1_stamps[1024 * 16];
#define OG_COLLECT_STAMP(i, dep) do { if (threadIdx.x == 0 && g_k1_stamps[blockIdx.x * 16 + (i)] == 0 && (dep) == (dep)) g_k1_stamps[blockIdx.x * 16 + (i)] = (long long)wall_clock64(); } while (0)
__device__ long long g_band_stamps[2048 * 8];   // band_topk_kernel / merge_collect_kernel: [workgroup][time point]
#define BAND_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_band_stamps[blockIdx.x * 8 + (i)] = (long long)wall_clock64(); } while (0)
__device__ long long g_wave_stamps[2048 * 16];   // band_topk_kernel: [workgroup][wave 0..3][stream done, end, pushes, compactions]
#define WAVE_STAMP(i, v) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048 && (threadIdx.x >> 6) < 4) g_wave_stamps[blockIdx.x * 16 + (threadIdx.x >> 6) * 4 + (i)] = (long long)(v); } while (0)
#define BAND_STAMP_MAX(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048) atomicMax((unsigned long long *)&g_band_stamps[blockIdx.x * 8 + (i)], (unsigned long long)wall_clock64()); } while (0)
#define MERGE_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 900) g_band_stamps[(1100 + blockIdx.x) * 8 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define BAND_STAMP(i) do { } while (0)
#define BAND_STAMP_MAX(i) do { } while (0)
#define WAVE_STAMP(i, v) do { } while (0)
#define MERGE_STAMP(i) do { } while (0)
#endif
#include "collect_body.h"
#include "og_common.h"

namespace {

#ifndef OG_K1_LOAD_AUX
#define OG_K1_LOAD_AUX 0  // default cache policy.  nt (2) reads an HBM-cold batch 2 us faster (56 vs 58 us for the two launches), but in the
                          // decode pipeline the batch has just been written by K1a and sits in the Infinity Cache: there nt loads cost
                          // 60 us (66 behind the backbone) against 57 -- tools/k1_bench.py, every policy (sc0, sc1, nt, combinations)
#endif
#ifndef OG_K1_BAND_PF
#define OG_K1_BAND_PF 3
#endif
constexpr int kPrefetch = OG_K1_BAND_PF;      // rows in flight per lane: 3 (2 .. 3 measured best on MI355X: 56.7 us for the three launches, 4: 60.4,
                                              // 6 / 8: 63.2, 1: 68; deeper queues lose -- more DRAM rows open at once, not occupancy)
#ifndef OG_K1_BAND_ABL
#define OG_K1_BAND_ABL 0
#endif
constexpr int kBandAbl = OG_K1_BAND_ABL;   // tuning harness, see band_topk_kernel
constexpr int kInterior = 62;     // interior lanes per wave panel
constexpr int kMaxWaves = 16;     // waves per workgroup (panels per row)
constexpr uint32_t kLaneOob = 0x80000000u;  // offset of lanes outside the image
constexpr uint32_t kRowOob = 0x40000000u;   // offset of rows the band must not read
constexpr int kHistBins = 256;              // per-plane score histogram (NMS mode)
#ifndef OG_K1_HIST_MBITS
#define OG_K1_HIST_MBITS 3
#endif
constexpr int kHistMBits = OG_K1_HIST_MBITS;                       // bin = exponent + top kHistMBits mantissa bits
constexpr int kHistShift = 23 - kHistMBits;
constexpr int kHistBase = (127 + 2 - (kHistBins >> kHistMBits)) << kHistMBits;   // the bins end at 2^2; 3 bits: bin 0 starts at 2^-30,
                                                                                  // 4 bits: 2^-14 (everything smaller joins bin 0)
constexpr uint64_t kWsMagic = 0x4f47444543303031ull;  // workspace self-validation word

__device__ __forceinline__ int hist_bin(int bits)
{
    return min(max((bits >> kHistShift) - kHistBase, 0), kHistBins - 1);
}
__device__ __forceinline__ int hist_edge_bits(int bin) { return bin == 0 ? 1 : (bin + kHistBase) << kHistShift; }

template <int VEC>
struct Px {
    float c[VEC];
};

// horizontal 3-max of one row; neighbours of the edge components come from adjacent lanes
template <int VEC>
__device__ __forceinline__ Px<VEC> hmax3(const Px<VEC> &v)
{
    const float left = og_from_lane_below(v.c[VEC - 1]);
    const float right = og_from_lane_above(v.c[0]);
    Px<VEC> h;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float a = (j == 0) ? left : v.c[j - 1];
        const float b = (j == VEC - 1) ? right : v.c[j + 1];
        h.c[j] = og_max3(a, v.c[j], b);
    }
    return h;
}

// flip-test folded into the fused source (FUSED kernels only): `in` holds the stride-4 heat maps of [images | mirrored images]
// (2N x C planes); plane (n, c) is merged on the fly with plane (N + n, kp_perm[c]) read right to left.  kp_perm null: no flip.
struct FlipSrc {
    const int32_t *kp_perm;
    int N, C;
};
__device__ __forceinline__ const float *flip_partner(const FlipSrc &fs, const float *in, int plane, size_t plane_elems)
{
    if (!fs.kp_perm) return nullptr;
    const int n = plane / fs.C, c = plane - n * fs.C;
    return in + ((size_t)(fs.N + n) * fs.C + fs.kp_perm[c]) * plane_elems;
}

struct TileGeom {
    int plane_rows, plane_cols;  // H, W
    int r0, r1;                  // interior rows [r0, r1)
    int col;                     // first column of this lane (may be < 0 or >= W: halo / idle)
    uint32_t lane_off;           // byte offset of the lane's strip inside a row, or kLaneOob
    bool interior;               // lane emits results
};

// Walk rows r0..r1-1 of one panel; emit(row, centre values, 3x3 max incl. zero padding).
struct NoHook {
    __device__ __forceinline__ void operator()(int) const {}
};

// `mid` runs once, between the issue of the first rows' loads and their first use: set-up work that has a memory round trip of
// its own (the band kernel's workspace check) waits beside the rows instead of in front of them.
template <int VEC, int PF, class Emit, class Begin = NoHook, class End = NoHook, class Mid = NoHook>
__device__ __forceinline__ void walk_panel(const float *plane, const TileGeom &g, Emit &&emit, Begin &&iter_begin = NoHook(),
                                           End &&iter_end = NoHook(), Mid &&mid = NoHook())
{
    const int H = g.plane_rows, W = g.plane_cols;
    const uint32_t row_bytes = (uint32_t)W * 4u;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(plane), 0, (int)((uint32_t)H * row_bytes), 0x00020000);
    auto load_row = [&](int row) {
        // row -1 wraps to a huge offset, rows past the band's halo get kRowOob: both read as zeros
        const uint32_t srow = (row <= g.r1) ? (uint32_t)row * row_bytes : kRowOob;
        const int off = (int)(g.lane_off + srow);
        Px<VEC> r;
        if constexpr (VEC == 4) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f t = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, OG_K1_LOAD_AUX));
            r.c[0] = t.x; r.c[1] = t.y; r.c[2] = t.z; r.c[3] = t.w;
        } else {
            r.c[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, off, 0, 0));
        }
        return r;
    };
    // The loads must be ISSUED in the order the rows are consumed: vmcnt counts in-order returns, so a prologue that the
    // scheduler shuffles (the loads are independent) makes the first wait of every iteration a full drain -- vmcnt(0)
    // at the loop head instead of vmcnt(PF-1), seen in the ISA of the PF = 8 build.  sched_barrier pins the order.
    const Px<VEC> top = load_row(g.r0 - 1);
    __builtin_amdgcn_sched_barrier(0);
    Px<VEC> v_b = load_row(g.r0);
    __builtin_amdgcn_sched_barrier(0);
    Px<VEC> q[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        q[u] = load_row(g.r0 + 1 + u);
        __builtin_amdgcn_sched_barrier(0);
    }
    mid(0);
    __builtin_amdgcn_sched_barrier(0);
    Px<VEC> hm_a = hmax3<VEC>(top);
    Px<VEC> hm_b = hmax3<VEC>(v_b);
    for (int r = g.r0; r < g.r1; r += PF) {
        iter_begin(r);
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            // a REAL copy out of the queue slot before it is reloaded: left to the compiler, the row lives on in the slot's
            // registers as v_b, the reload gets other registers and the loop head copies the youngest load back
            // (= waits for it: a drain per iteration)
            Px<VEC> v_c;
#pragma unroll
            for (int j = 0; j < VEC; ++j) asm volatile("v_mov_b32 %0, %1" : "=v"(v_c.c[j]) : "v"(q[u].c[j]));
            __builtin_amdgcn_sched_barrier(0);
            q[u] = load_row(r + u + 1 + PF);
            __builtin_amdgcn_sched_barrier(0);
            const Px<VEC> hm_c = hmax3<VEC>(v_c);  // all lanes take part in the DPP shifts
            if (r + u < g.r1) emit(r + u, v_b, hm_a, hm_b, hm_c);  // centre row + horizontal maxima of rows r-1, r, r+1
            hm_a = hm_b;
            hm_b = hm_c;
            v_b = v_c;
        }
        iter_end(r);
    }
}

// vertical 3-max of the horizontal maxima = 3x3 maximum (zero padding included)
template <int VEC>
__device__ __forceinline__ Px<VEC> vmax3(const Px<VEC> &a, const Px<VEC> &b, const Px<VEC> &c)
{
    Px<VEC> m;
#pragma unroll
    for (int j = 0; j < VEC; ++j) m.c[j] = og_max3(a.c[j], b.c[j], c.c[j]);
    return m;
}

// Fused source: the hi-res rows are not read from memory but produced on the fly from the stride-4
// head output (x4 bicubic, bit-identical to K1a): a lane owns one SOURCE column (= 4 hi-res columns),
// keeps the x-pass results of 5 source rows in registers and emits 4 hi-res rows per source row.  The
// +-2 source-column neighbours and the +-1 hi-res neighbours all come from DPP wave shifts, so a
// panel has 58 interior lanes (3 halo lanes each side).  Rows/columns outside the image are 0.0
// (F.pad), everything downstream is the same emit() as the streaming walker.
//   lr: low-res plane (h x w); the band covers hi-res rows [r0, r1), r0 % 4 == 0.
//   lrb: with flip-test, the plane of the MIRRORED image's partner channel (config.heatmap_hflip) or null: every source value is then
//   (lr[y][x] + lrb[y][w-1-x]) / 2, exactly what og_flip_merge_f32 would have written (decoder/factory.py:104-106).
template <int PF, class Emit, class End = NoHook>
__device__ __forceinline__ void walk_panel_fused(const float *__restrict__ lr, const float *__restrict__ lrb, int h, int w,
                                                 const TileGeom &g, int q, Emit &&emit, End &&iter_end = NoHook())
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int H = 4 * h;
    const int qc = min(max(q, 0), w - 1);       // index clamp == torch's tap clamp
    const bool in_img = q >= 0 && q < w;        // lanes outside the image produce the zero padding
    float wt[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) wt[r][j] = og_cubic_w[r][j];
    auto src = [&](int row) {
        const size_t ro = (size_t)min(max(row, 0), h - 1) * w;
        const float v = lr[ro + qc];
        return lrb ? (v + lrb[ro + (w - 1 - qc)]) / 2.f : v;      // (uniform branch)
    };
    // x-pass of one source row.  Lanes left / right of the image get ZERO x-pass results -- four selects per SOURCE row instead of
    // sixteen per-pixel ones behind the y-pass: the y-pass of zeros is (signed) zero, which is all the 3x3 maximum of an edge pixel
    // needs from its padding (v == m holds for +-0 alike, and a lane outside the image never emits)
    auto xpass = [&](float c) {
        OgRow4 o = og_cubic_xpass(c, wt);
#pragma unroll
        for (int j = 0; j < 4; ++j) o.p[j] = in_img ? o.p[j] : 0.f;
        return o;
    };
    // y-pass of one hi-res row, two pixels per packed instruction (v_pk_mul_f32 / v_pk_fma_f32): the same chain as og_cubic_chain,
    // element for element (o = fma(t0,w0, fl(t1*w1)); o = fma(t2,w2,o); o = fma(t3,w3,o))
    auto hires = [&](const OgRow4 &t0, const OgRow4 &t1, const OgRow4 &t2, const OgRow4 &t3, int phase) {
        Px<4> o;
#pragma unroll
        for (int x = 0; x < 4; x += 2) {
            const v2f a0 = {t0.p[x], t0.p[x + 1]}, a1 = {t1.p[x], t1.p[x + 1]}, a2 = {t2.p[x], t2.p[x + 1]}, a3 = {t3.p[x], t3.p[x + 1]};
            const v2f w0 = {wt[phase][0], wt[phase][0]}, w1 = {wt[phase][1], wt[phase][1]}, w2 = {wt[phase][2], wt[phase][2]},
                      w3 = {wt[phase][3], wt[phase][3]};
            v2f r = __builtin_elementwise_fma(a0, w0, a1 * w1);
            r = __builtin_elementwise_fma(a2, w2, r);
            r = __builtin_elementwise_fma(a3, w3, r);
            o.c[x] = r[0];
            o.c[x + 1] = r[1];
        }
        return o;
    };
    auto zero_row = [] { Px<4> z; z.c[0] = z.c[1] = z.c[2] = z.c[3] = 0.f; return z; };
    const int p0 = g.r0 >> 2, p1 = g.r1 >> 2;   // source rows of the band (r0, r1 are multiples of 4)
    // x-pass of source rows p0-2 .. p0+1: enough for hi-res row r0-1 (= phase 3 of source row p0-1; row -1 is zero padding)
    OgRow4 xa = xpass(src(p0 - 2)), xb = xpass(src(p0 - 1)), xc = xpass(src(p0)), xe = xpass(src(p0 + 1));
    Px<4> hm_a = hmax3<4>(g.r0 > 0 ? hires(xa, xb, xc, xe, 3) : zero_row());
    OgRow4 xf = xpass(src(p0 + 2));   // window of p0 complete: rows p0-2 .. p0+2
    Px<4> v_b = hires(xa, xb, xc, xe, 0);
    Px<4> hm_b = hmax3<4>(v_b);
    float pre[PF];                                  // prefetched source values of rows p+3 .. p+2+PF
#pragma unroll
    for (int u = 0; u < PF; ++u) pre[u] = src(p0 + 3 + u);
    for (int p = p0; p < p1; ++p) {
#pragma unroll
        for (int ph = 1; ph <= 4; ++ph) {          // produce hi-res row 4p+ph, emit row 4p+ph-1
            Px<4> v_c;
            if (ph == 1) v_c = hires(xa, xb, xc, xe, 1);
            else if (ph == 2) v_c = hires(xb, xc, xe, xf, 2);
            else if (ph == 3) v_c = hires(xb, xc, xe, xf, 3);
            else {                                  // first row of source row p+1: shift the window, x-pass of row p+3
                const float nxt = pre[0];
#pragma unroll
                for (int u = 0; u + 1 < PF; ++u) pre[u] = pre[u + 1];
                pre[PF - 1] = src(p + 3 + PF);
                xa = xb; xb = xc; xc = xe; xe = xf;
                xf = xpass(nxt);
                v_c = hires(xa, xb, xc, xe, 0);
                if (4 * p + 4 >= H) v_c = zero_row();       // (uniform) the row below the image: zero padding
            }
            const Px<4> hm_c = hmax3<4>(v_c);
            emit(4 * p + ph - 1, v_b, hm_a, hm_b, hm_c);
            hm_a = hm_b;
            hm_b = hm_c;
            v_b = v_c;
        }
        iter_end(p);
    }
}

// Work items of the band kernel: plane p is cut into b_lo or b_lo + 1 bands (n_hi of the planes, spread evenly, take the extra
// one), so that the launch has a chosen number of workgroups -- a multiple of the CU count: at bs8 640x640 1 024 instead of
// 136 x 8 = 1 088, whose 64 extra workgroups put a fifth one on every fourth CU and set the end of the stream (stamps:
// 36 us against 32.5 us for the others).  Item index = start(p) + band; band boundaries are multiples of 4 rows.
struct BandMap {
    int planes, b_lo, n_hi;
};
// (32-bit arithmetic: make_plan only builds maps whose products fit -- 64-bit divisions cost the band kernel 1 us of set-up)
__host__ __device__ __forceinline__ int bm_start(const BandMap &m, int p) { return p * m.b_lo + (int)(((unsigned)p * (unsigned)m.n_hi) / (unsigned)m.planes); }
__host__ __device__ __forceinline__ int bm_row(int b, int nb, int H)
{
    if (b >= nb) return H;
    if (H < (1 << 20)) return (int)((((unsigned)b * (unsigned)H) / (unsigned)nb) & ~3u);   // (at most 64 bands)
    return (int)((((long)b * H) / nb) & ~3l);
}
__device__ __forceinline__ void bm_locate(const BandMap &m, int total, int wid, int &plane, int &band, int &nb)
{
    if (m.n_hi == 0) {   // equal band counts (any number of planes)
        plane = wid / m.b_lo;
        band = wid - plane * m.b_lo;
        nb = m.b_lo;
        return;
    }
    int p = (int)(((unsigned)wid * (unsigned)m.planes) / (unsigned)total);
    while (p + 1 < m.planes && bm_start(m, p + 1) <= wid) ++p;
    while (p > 0 && bm_start(m, p) > wid) --p;
    plane = p;
    band = wid - bm_start(m, p);
    nb = bm_start(m, p + 1) - bm_start(m, p);
}

__device__ __forceinline__ TileGeom make_geom(int H, int W, int r0, int r1, int panel_strips, int vec)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int strips = (W + vec - 1) / vec;
    const int s_first = wave * panel_strips;                       // first interior strip of the panel
    const int s_cnt = min(panel_strips, strips - s_first);         // interior strips in this panel
    const int s = s_first - 1 + lane;                              // this lane's strip
    TileGeom g;
    g.plane_rows = H;
    g.plane_cols = W;
    g.r0 = r0;
    g.r1 = r1;
    g.col = s * vec;
    g.interior = lane >= 1 && lane <= s_cnt;
    const bool loads = lane <= s_cnt + 1 && s >= 0 && s < strips;
    g.lane_off = loads ? (uint32_t)g.col * 4u : kLaneOob;
    return g;
}

// ---------------------------------------------------------------------------------------
// hmp_NMS materialised (API parity with decoder/heatmap.py:15-35)
// ---------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(64 * kMaxWaves)
nms_map_kernel(const float *__restrict__ in, float *__restrict__ out, int H, int W, int rows, int nbands,
               int panel_strips, int total, int padded)
{
    const int wid = og_xcd_remap(blockIdx.x, padded);
    if (wid >= total) return;
    const int plane = wid / nbands, band = wid % nbands;
    const TileGeom g = make_geom(H, W, band * rows, min(band * rows + rows, H), panel_strips, VEC);
    const float *src = in + (size_t)plane * H * W;
    float *dst = out + (size_t)plane * H * W;
    walk_panel<VEC, kPrefetch>(src, g, [&](int row, const Px<VEC> &v, const Px<VEC> &ha, const Px<VEC> &hb, const Px<VEC> &hc) {
        const Px<VEC> m = vmax3<VEC>(ha, hb, hc);
        if (!g.interior) return;
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = v.c[j] * ((m.c[j] == v.c[j]) ? 1.f : 0.f);
        float *p = dst + (size_t)row * W + g.col;
        if constexpr (VEC == 4) *reinterpret_cast<float4 *>(p) = make_float4(o[0], o[1], o[2], o[3]);
        else *p = o[0];
    });
}

// ---------------------------------------------------------------------------------------
// band kernel: stream + candidate compaction + per-band top-k
// ---------------------------------------------------------------------------------------
// Per-wave candidate segment in LDS: two buffers of `cap` keys (ping-pong).  A compaction
// leaves <= k keys sorted descending, so k + 64 <= cap is required.
struct WaveSeg {
    uint64_t *cur, *alt;  // LDS
    int cap;
    int cnt;              // wave-uniform
    // admission threshold: NMS mode compares float bit patterns as signed ints (>= 1 means
    // strictly positive), plain mode compares floats
    int tau_bits;
    float tau_f;
    // the same threshold as seen by this lane: halo / idle lanes carry "never" so that the hot loop
    // needs no separate lane mask (kept in registers, refreshed only when tau changes)
    bool emits;
    int lane_tau_bits;
    float lane_tau_f;
    int *hist;            // plane histogram (global) or nullptr
    int npush = 0;        // debug statistics (OG_K1_DEBUG)
    int ncompact = 0;

    __device__ __forceinline__ void set_tau(float t)
    {
        tau_f = t;
        tau_bits = __builtin_bit_cast(int, t);
        lane_tau_bits = emits ? tau_bits : 0x7fffffff;
        lane_tau_f = emits ? t : INFINITY;
    }

    // Keep the k largest keys, sorted descending, in cur[0..min(cnt,k)).
    __device__ __forceinline__ void compact(int k)
    {
        const int lane = threadIdx.x & 63;
        ++ncompact;
        // 1) One ballot pass drops the keys below TODAY's admission threshold: it is a lower bound of the plane's k-th best
        //    that has risen since they were admitted (compactions, the helper's plane-wide bound), so they cannot be among
        //    the k best.  Ranking by counting costs cnt^2 / 64 64-bit compares per lane -- 1.7 us for the last compaction of
        //    a wave at bs8 640x640, all of it behind the stream -- and is left with the few keys that matter.
        const uint32_t tau_hi = (uint32_t)(og_make_key(tau_f, 0u) >> 32);
        int kept = 0;
        for (int base = 0; base < cnt; base += 64) {
            const int i = base + lane;
            const uint64_t key = i < cnt ? cur[i] : 0ull;
            const bool keep = i < cnt && (uint32_t)(key >> 32) >= tau_hi;
            const uint64_t m = __builtin_amdgcn_ballot_w64(keep);
            if (keep) alt[kept + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0))] = key;
            kept += __builtin_popcountll(m);
        }
        __builtin_amdgcn_wave_barrier();
        { uint64_t *t = cur; cur = alt; alt = t; }
        cnt = kept;
        // 2) rank the survivors
        for (int i = lane; i < cnt; i += 64) {
            const uint64_t mine = cur[i];
            const int rank = og_count_greater(cur, cnt, mine);  // LDS broadcast reads
            if (rank < k) alt[rank] = mine;
        }
        __builtin_amdgcn_wave_barrier();
        uint64_t *t = cur; cur = alt; alt = t;
        if (cnt >= k) {
            cnt = k;
            set_tau(og_key_value(cur[k - 1]));
        }
    }

    // pred = this lane's candidate passes, mask = its ballot (non-zero).  Kept lean: roughly
    // every row carries a candidate on realistic maps, so this is not a cold path.
    template <bool NMS_MODE>
    __device__ __forceinline__ void push(bool pred, uint64_t mask, float v, uint32_t idx, int k)
    {
        const int n = __builtin_popcountll(mask);
        if (__builtin_expect(cnt + n > cap, 0)) compact(k);
        const int pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
        if (pred) {
            // NMS mode only admits v > 0, whose order-preserving key is just the bit pattern with the top bit set
            const uint64_t key = NMS_MODE ? (((uint64_t)(__builtin_bit_cast(uint32_t, v) | 0x80000000u) << 32) | (uint32_t)~idx)
                                          : og_make_key(v, idx);
            cur[pos] = key;
            if (NMS_MODE && hist) atomicAdd(hist + hist_bin(__builtin_bit_cast(int, v)), 1);
        }
        cnt += n;
        npush += n;
    }
};

// ABL (tuning harness only, -DOG_K1_BAND_ABL): 0 = product; 1 = compute the admission masks but never push; 3 = loads and
// threshold ballots only; 4 = the whole NMS test (3x3 maxima, peak masks) without the candidate handling.
// FUSED: `in` holds the stride-4 head output (planes x H/4 x W/4) and the hi-res rows are produced on
// the fly (walk_panel_fused) instead of being read from a materialised (planes x H x W) tensor.
// One plane's merge by `nthr` consecutive threads (tid = 0..nthr-1 within the group; every thread of the workgroup calls
// this the same number of times: it contains workgroup barriers).  gk: the plane's `nlists` sorted lists of k keys each,
// zero-padded by their writers (a key is never zero).  `all` / `flt`: n_all = nlists * k keys each, in LDS;
// `s_bound` / `s_nf`: the group's own shared words.  emit(rank, score, flat index) receives the k best in any order.
// SC1: the band lists were written in THIS launch by other workgroups (sc1 stores): read them with sc1 loads.
template <bool NMS_MODE, bool FUSED, bool SC1 = false, class Emit>
__device__ __forceinline__ void merge_plane(const uint64_t *__restrict__ gk, uint64_t *all,
                                            uint64_t *flt, uint64_t *s_bound, int *s_nf, int tid, int nthr,
                                            const float *__restrict__ p, int H, int W, int k, int nlists, int t_sub, Emit &&emit,
                                            const float *__restrict__ pb = nullptr)
{
    const int n_all = nlists * k, lane = tid & 63;
    // All of a thread's keys are requested before the first one is used (one memory round trip, not one per key), and
    // land in LDS rank-major -- all[rank * nlists + list] -- so that the subset of step A is the head of the array.
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t *>(gk), 0, n_all * 8, 0x00020000);
    constexpr int U = 4;
    for (int base = 0; base < n_all; base += U * nthr) {
        u32x2 kk[U];
#pragma unroll
        for (int u = 0; u < U; ++u)   // (indices past the end read as zero: buffer range check)
            kk[u] = __builtin_amdgcn_raw_buffer_load_b64(kr, (base + u * nthr + tid) * 8, 0, SC1 ? 16 : 0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * nthr + tid;
            if (i < n_all) all[(i % k) * nlists + i / k] = ((uint64_t)kk[u].y << 32) | kk[u].x;
        }
    }
    if (tid == 0) { *s_bound = 0ull; *s_nf = 0; }
    __syncthreads();
    MERGE_STAMP(3);
    // A) k-th largest of the subset {first t_sub keys of each list}: a lower bound of the plane's k-th best
    const int n_sub = nlists * t_sub;
    for (int i = tid; i < n_sub; i += nthr) {
        const uint64_t mine = all[i];
        if (mine != 0ull && og_count_greater(all, n_sub, mine) == k - 1) *s_bound = mine;
    }
    __syncthreads();
    MERGE_STAMP(4);
    const uint64_t bound = *s_bound;
    // B) keys >= bound (order does not matter: ranks are recomputed)
    for (int i = tid; i < n_all; i += nthr) {
        const uint64_t key = all[i];
        if (key != 0ull && key >= bound) flt[atomicAdd(s_nf, 1)] = key;
    }
    __syncthreads();
    MERGE_STAMP(5);
    const int nf = *s_nf;
#ifdef OG_K1_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 900) g_band_stamps[(1100 + blockIdx.x) * 8 + 7] = nf;
#endif
    // C) rank and emit
    for (int i = tid; i < nf; i += nthr) {
        const uint64_t mine = flt[i];
        const int rank = og_count_greater(flt, nf, mine);
        if (rank < k) emit(rank, og_key_value(mine), (long)og_key_index(mine));
    }
    int t = min(nf, k);
    if (NMS_MODE && t < k && tid < 64) {
        // fewer than k positive peaks: fill with the lowest flat indices whose NMS output is
        // zero (ties at 0.0 broken by index, like every other tie).  One wave walks the plane in index order, 62 pixels per
        // round (lanes 1..62; lanes 0 and 63 are halo): a lane evaluates its COLUMN of three pixels (rows y-1, y, y+1; outside the
        // plane = the zero padding) and takes the neighbouring columns' maxima from the adjacent lanes -- three point
        // evaluations per lane instead of nine (K1-fused: 48 source taps instead of 160 per lane and round: this path cost the
        // fused merge launch 3 us on the bench inputs, where a quarter of the planes have fewer than k positive peaks)
        auto px = [&](int yy, int xx) { return FUSED ? og_bicubic4_at(p, pb, H >> 2, W >> 2, yy, xx) : p[(size_t)yy * W + xx]; };
        const long hw = (long)H * W;
        for (long base = 0; base < hw && t < k; base += 62) {
            const long i = base + lane - 1;
            const bool in = i >= 0 && i < hw;
            const int y = in ? (int)(i / W) : 0, x = in ? (int)(i % W) : 0;
            float c[3];
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) c[dy + 1] = (in && y + dy >= 0 && y + dy < H) ? px(y + dy, x) : 0.f;
            const float cm = og_max3(c[0], c[1], c[2]);
            float left = og_from_lane_below(cm), right = og_from_lane_above(cm);
            left = x == 0 ? 0.f : left;           // the lane below holds the LAST pixel of the row above: zero padding instead
            right = x == W - 1 ? 0.f : right;
            const float m = og_max3(left, cm, right), v = c[1];
            const bool zero = in && lane >= 1 && lane <= 62 && !(v == m && v != 0.f);
            const uint64_t mask = __builtin_amdgcn_ballot_w64(zero);
            const int slot = t + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            if (zero && slot < k) emit(slot, 0.f, i);
            t += __builtin_popcountll(mask);
        }
    }
}

template <int VEC, bool NMS_MODE, int PF, int ABL = 0, bool FUSED = false>
// (amdgpu_waves_per_eu: <= 96 VGPRs -- at bs8 640x640 all 1 088 workgroups must be resident together, 5 per CU)
__global__ void __launch_bounds__(64 * kMaxWaves) __attribute__((amdgpu_waves_per_eu(5)))
band_topk_kernel(const float *__restrict__ in, uint64_t *__restrict__ band_keys,
                 int *__restrict__ hist_all, uint64_t *__restrict__ ws_magic, uint64_t magic,
                 int H, int W, int k, int cap, BandMap bm, int max_bands, int panel_strips, int total, int padded, int helper,
                 int wl, FlipSrc fs)
{
    // all LDS comes from the dynamic region (no static __shared__ in front of it: the base stays
    // 16-byte aligned for ds_read_b128): [2*nwaves key buffers | histogram | per-wave counts/slots | tau]
    extern __shared__ __attribute__((aligned(16))) uint64_t smem[];
    int *s_hist = reinterpret_cast<int *>(smem + (size_t)(blockDim.x >> 6) * 2 * cap);
    int *s_cnt = s_hist + kHistBins;
    int *s_slot = s_cnt + kMaxWaves;   // which of the 2*nwaves key buffers holds wave w's final list
    int *s_tau_p = s_slot + kMaxWaves;    // followed by the streaming-waves-done counter
#define s_tau (*s_tau_p)
    BAND_STAMP(0);
    const int wid = og_xcd_remap(blockIdx.x, padded);
    if (wid >= total) return;
    int plane, band, nbands;   // nbands: bands of THIS plane
    bm_locate(bm, total, wid, plane, band, nbands);
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    TileGeom g = make_geom(H, W, bm_row(band, nbands, H), bm_row(band + 1, nbands, H), panel_strips, VEC);
    int q_lane = 0;
    if constexpr (FUSED) {  // lanes = source columns: 58 interior + 3 halo lanes each side
        const int lane = threadIdx.x & 63, w4 = W >> 2;
        const int q_first = wave * panel_strips, q_cnt = min(panel_strips, w4 - q_first);
        q_lane = q_first - 3 + lane;
        g.col = 4 * q_lane;
        g.interior = lane >= 3 && lane < 3 + q_cnt;
    }
    const float *src = in + (FUSED ? (size_t)plane * (H >> 2) * (W >> 2) : (size_t)plane * H * W);

    WaveSeg seg;
    seg.cur = smem + (size_t)wave * 2 * cap;
    seg.alt = seg.cur + cap;
    seg.cap = cap;
    seg.cnt = 0;
    seg.emits = g.interior;  // (VEC == 1: every interior lane is inside the image as well)
    if (NMS_MODE) seg.set_tau(__builtin_bit_cast(float, 1));  // bits >= 1  <=>  v > +0
    else seg.set_tau(-INFINITY);
    if (ABL == 2) seg.set_tau(0.035f);  // harness: what a perfect plane-wide threshold would buy
    // Plane-wide admission threshold, no atomics on global memory and NO vector-memory operation in
    // the streaming waves' loop (any conditional one makes the compiler drain the row prefetch queue
    // with vmcnt(0) every iteration -- ~5 us each under load).  The streaming waves only ds_add their
    // candidates into an LDS histogram and read the current threshold from LDS.  A HELPER wave (the
    // last wave of the workgroup, present when `helper` is set) loops beside them:
    //   scan the LDS histogram (DPP prefix sum) -> E_k = edge of the bin with the band's k-th best,
    //   E_t = edge of its t-th best (t = k/4) -> publish E_t in the band's slot of a per-plane table
    //   (sc1 store) -> read all slots (sc1 load): the 4th largest published edge bounds the plane's
    //   k-th best (4 bands x t >= k candidates above it) -> s_tau = max(...) in LDS -> sleep.
    // Slots only grow and every bound is a true lower bound whatever the timing, so the top-k stays
    // exact.  The table is trusted only if the workspace carries this geometry's magic word.
    const int hmode = helper;   // bit 0: helper wave present; bit 3: debug statistics (tools/)
    helper &= 1;
    const int nstream = (blockDim.x >> 6) - helper;   // streaming waves
    // The workspace check (one memory round trip) and the LDS set-up behind it run AFTER a streaming wave has issued the
    // loads of its first rows (walk_panel's `mid`): the stream starts one round trip earlier.  Every wave passes through
    // `setup` exactly once (it holds a workgroup barrier).
    int *gslot = nullptr;
    seg.hist = nullptr;
    int *s_done = s_tau_p + 1;
    auto setup = [&](int) {
        gslot = (NMS_MODE && helper && *ws_magic == magic) ? hist_all + (size_t)plane * max_bands : nullptr;
        seg.hist = gslot ? s_hist : nullptr;
        if (gslot) {
            for (int i = threadIdx.x; i < kHistBins; i += blockDim.x) s_hist[i] = 0;
            if (threadIdx.x == 0) { s_tau = 1; *s_done = 0; }
            __syncthreads();
        }
    };
    const int lane_id = threadIdx.x & 63;
    if (helper && wave == nstream) {
        setup(0);
        if (gslot) {
            const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(gslot, 0, nbands * 4, 0x00020000);
            const int t_band = (k + 3) / 4, need = (k + t_band - 1) / t_band;
            int published = 0;
            for (;;) {
                const int done = __hip_atomic_load(s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                // lane l holds local bins 4l..4l+3; inclusive prefix over lanes by DPP, then suffix sums
                int h[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    h[j] = __hip_atomic_load(s_hist + 4 * lane_id + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int mine = h[0] + h[1] + h[2] + h[3];
                int pre = mine;
                pre += __builtin_amdgcn_update_dpp(0, pre, 0x111, 0xf, 0xf, false);  // row_shr:1
                pre += __builtin_amdgcn_update_dpp(0, pre, 0x112, 0xf, 0xf, false);  // row_shr:2
                pre += __builtin_amdgcn_update_dpp(0, pre, 0x114, 0xf, 0xf, false);  // row_shr:4
                pre += __builtin_amdgcn_update_dpp(0, pre, 0x118, 0xf, 0xf, false);  // row_shr:8
                pre += __builtin_amdgcn_update_dpp(0, pre, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
                pre += __builtin_amdgcn_update_dpp(0, pre, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
                const int total = __builtin_amdgcn_readlane(pre, 63);
                const int suf = total - pre + mine;   // candidates in bins >= 4l
                const int above = suf - mine;
                auto edge_of = [&](int want) {        // lower edge (float bits) of the bin holding the want-th best
                    const uint64_t ge = __builtin_amdgcn_ballot_w64(suf >= want);
                    if (ge == 0ull) return 0;
                    const int top = 63 - __builtin_clzll(ge);
                    int bin = 4 * lane_id;
                    if (above + h[3] >= want) bin += 3;
                    else if (above + h[3] + h[2] >= want) bin += 2;
                    else if (above + h[3] + h[2] + h[1] >= want) bin += 1;
                    return __builtin_amdgcn_readlane(hist_edge_bits(bin), top);
                };
                const int e_k = edge_of(k), e_t = edge_of(t_band);
                if (e_t > published) {
                    published = e_t;
                    // PLAIN store (stays in this XCD's L2; the sc1 loads below are L2-served): the bands of a
                    // plane are mapped to one XCD, and a reader on another XCD merely sees an older, weaker
                    // bound.  An sc1 (write-through) store here cost +35 us per launch on MI355X.
                    if (lane_id == 0) __builtin_amdgcn_raw_buffer_store_b32(e_t, srsrc, band * 4, 0, 0);
                }
                // the `need`-th largest published slot (own slot: the value just computed)
                int sv = __builtin_amdgcn_raw_buffer_load_b32(srsrc, lane_id * 4, 0, 16);  // lanes >= nbands read 0
                if (lane_id == band) sv = published;
                int rank = 0;
                for (int b = 0; b < nbands; ++b) {
                    const int o = __builtin_amdgcn_readlane(sv, b);
                    rank += (o > sv) || (o == sv && b < lane_id);
                }
                const uint64_t pick = __builtin_amdgcn_ballot_w64(lane_id < nbands && sv > 0 && rank == need - 1);
                int bound = e_k;
                if (pick != 0ull) bound = max(bound, __builtin_amdgcn_readlane(sv, __builtin_ctzll(pick)));
                if (lane_id == 0 && bound > 1) atomicMax(&s_tau, bound);
                if (done >= nstream) break;
                // (the wait is cut into short naps: the streaming waves' epilogue may hold a workgroup barrier, and the
                // launch ends with its last wave)
                bool stop = false;
                for (int nap = 0; nap < 8 && !stop; ++nap) {
                    __builtin_amdgcn_s_sleep(8);
                    stop = __hip_atomic_load(s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >= nstream;
                }
                if (stop) break;
            }
            // leave no vector-memory operation pending on this path: otherwise the compiler has to
            // assume unknown outstanding counts at the head of the streaming loop below and drains
            // the row prefetch queue (vmcnt(0)) on EVERY iteration of the streaming waves
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        }
        BAND_STAMP_MAX(6);   // helper wave leaves its loop
    }
    // A streaming wave reports itself done a few rows BEFORE its last one: the helper (whose bound could no longer save
    // anything) then leaves ahead of the streaming waves instead of up to one polling period behind them.
    bool reported = false;
    auto iter_begin = [&](int r) {
        if (!NMS_MODE || !gslot) return;
        const int t = s_tau;  // LDS broadcast
        if (t > seg.tau_bits) seg.set_tau(__builtin_bit_cast(float, t));
        if (!reported && (FUSED ? 4 * r : r) + 4 * PF >= g.r1) {
            reported = true;
            if (lane_id == 0) atomicAdd(s_done, 1);
        }
    };

    auto emit_fn = [&](int row, const Px<VEC> &v, const Px<VEC> &ha, const Px<VEC> &hb, const Px<VEC> &hc) {
        // Threshold first: a pixel below tau can never be admitted, so the vertical maxima, the equality
        // tests and the masks are only evaluated for wave-rows that hold a pixel >= tau (the horizontal
        // maxima stay incremental: recomputing all three rows on demand measured slower).  Fewer
        // instructions per row keep the kernel HBM-bound when the chip clocks down behind the backbone.
        // (masks are built from ballots of plain compares and combined on the scalar side: a ballot of `a && b` is
        // lowered through a 0/1 VGPR -- v_cndmask + v_cmp_ne per component)
        // (Comparing the lane's row MAXIMUM first -- one compare and one branch per wave-row, the four compares only behind it -- was
        // measured on the bench inputs, where most wave-rows hold a pixel above the threshold: 57.5 vs 55.3 us for K1, 54.5 vs 49.3 us
        // for K1-fused, profiles/r05_k1_rowmax_ab.log: the extra instructions are paid on every row and save nothing.  Not kept.)
        uint64_t ge[VEC], any = 0;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            ge[j] = __builtin_amdgcn_ballot_w64(NMS_MODE ? (__builtin_bit_cast(int, v.c[j]) >= seg.lane_tau_bits)
                                                         : (v.c[j] >= seg.lane_tau_f));
            any |= ge[j];
        }
        if (ABL == 1) {
            seg.cnt += (int)(any >> 63);  // keep the masks alive
            return;
        }
        if (ABL == 3) return;   // harness: loads + threshold ballots only (the access pattern's own ceiling)
        if (any != 0ull) {
            Px<VEC> m;
            if (NMS_MODE) m = vmax3<VEC>(ha, hb, hc);
            const uint32_t base = (uint32_t)row * (uint32_t)W + (uint32_t)g.col;
            // all four masks before the first push: most wave-rows that reach this point hold pixels above the threshold
            // that are not peaks (the flanks of a blob), and leave through ONE branch instead of four.  A push may raise
            // tau (compaction); the later components are then admitted against the older, lower bound: still exact.
            uint64_t mj[VEC], many = 0;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                mj[j] = NMS_MODE ? (ge[j] & __builtin_amdgcn_ballot_w64(v.c[j] == m.c[j])) : ge[j];
                many |= mj[j];
            }
            if (ABL == 4) {   // harness: the whole NMS test, no candidate handling
                seg.cnt += (int)(many >> 63);
                return;
            }
            if (many != 0ull) {
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    if (mj[j] != 0ull)   // the lane predicate IS the mask (a predicate recomputed here would see a tau that an
                                         // earlier component's compaction has raised, and leave holes in the segment)
                        seg.template push<NMS_MODE>(__builtin_amdgcn_inverse_ballot_w64(mj[j]), mj[j], v.c[j], base + j, k);
            }
        }
    };
    BAND_STAMP(1);
    if (wave < nstream) {  // threshold refresh at the END of each iteration (see walk_panel)
        if constexpr (FUSED) {
            setup(0);
            walk_panel_fused<PF>(src, flip_partner(fs, in, plane, (size_t)(H >> 2) * (W >> 2)), H >> 2, W >> 2, g, q_lane, emit_fn, iter_begin);
        } else {
            walk_panel<VEC, PF>(src, g, emit_fn, NoHook(), iter_begin, setup);
        }
    }
    if (gslot && lane_id == 0 && wave < nstream && !reported) atomicAdd(s_done, 1);
    if ((hmode & 8) && hist_all && lane_id == 0 && wave < nstream) {  // debug: pushes / final threshold / waves, kept in the unused tail of the slot region
        int *dbg = hist_all + (size_t)bm.planes * kHistBins - 16;
        atomicAdd(dbg + 0, seg.npush); atomicMax(dbg + 1, seg.tau_bits); atomicAdd(dbg + 2, 1); atomicAdd(dbg + 3, seg.tau_bits > 1 ? 1 : 0);
    }

    BAND_STAMP(2);
    WAVE_STAMP(0, wall_clock64());
    // per-wave top-k, then merge the waves' lists by rank counting
    seg.compact(k);
    BAND_STAMP(3);
    if (wl > 1) {
        // one list per streaming wave (wl = streaming waves per band): no workgroup barrier, no cross-wave ranking here --
        // the merge launch ranks the plane's nbands * wl sorted lists anyway.  The helper wave has no list.
        if (wave < nstream) {
            const size_t li = (size_t)wid * wl + wave;
            const int c = min(seg.cnt, k);
            for (int i = lane_id; i < k; i += 64) band_keys[li * k + i] = i < c ? seg.cur[i] : 0ull;   // zero-padded
        }
        BAND_STAMP(4);
        BAND_STAMP_MAX(5);   // last wave of the workgroup done
        WAVE_STAMP(1, wall_clock64());
        WAVE_STAMP(2, seg.npush);
        WAVE_STAMP(3, seg.ncompact);
        return;
    }
    if ((threadIdx.x & 63) == 0) { s_cnt[wave] = min(seg.cnt, k); s_slot[wave] = (int)((seg.cur - smem) / cap); }
    __syncthreads();
    int total_keys = 0;
    for (int w = 0; w < nwaves; ++w) total_keys += s_cnt[w];
    uint64_t *out = band_keys + (size_t)wid * k;
    for (int t = threadIdx.x; t < total_keys; t += blockDim.x) {
        int w = 0, o = t;
        while (o >= s_cnt[w]) { o -= s_cnt[w]; ++w; }
        const uint64_t key = smem[(size_t)s_slot[w] * cap + o];
        int rank = 0;
        for (int w2 = 0; w2 < nwaves; ++w2) rank += og_count_greater(smem + (size_t)s_slot[w2] * cap, s_cnt[w2], key);
        if (rank < k) out[rank] = key;
    }
    for (int t = min(total_keys, k) + threadIdx.x; t < k; t += blockDim.x) out[t] = 0ull;   // the list is zero-padded to k keys
    BAND_STAMP(4);
#undef s_tau
}

// ---------------------------------------------------------------------------------------
// merge kernel: one wave per plane selects the plane's top-k from the sorted band lists.
//   A) a lower bound L on the k-th best: the k-th largest among the first `t` keys of every
//      band (a subset of all keys, so the true k-th best is >= L);
//   B) compact the keys >= L (usually just over k of them);
//   C) rank them by counting and write the k best in order.
// ---------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// merge kernel: one workgroup per plane selects the plane's top-k from the sorted band lists.
//   A) a lower bound L on the k-th best: the k-th largest among the first `t` keys of every
//      band (a subset of all keys, so the true k-th best is >= L);
//   B) compact the keys >= L (usually just over k of them);
//   C) rank them by counting and write the k best in order.
// ---------------------------------------------------------------------------------------
template <bool NMS_MODE, bool FUSED = false>
__global__ void __launch_bounds__(256)
merge_bands_kernel(const uint64_t *__restrict__ band_keys, int *__restrict__ hist_all, uint64_t *__restrict__ ws_magic,
                   uint64_t magic, const float *__restrict__ in, int H, int W, int k, BandMap bm, int max_bands, int wl, int t_sub,
                   float *__restrict__ out_scores, int64_t *__restrict__ out_inds, FlipSrc fs)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    const int plane = blockIdx.x, tid = threadIdx.x;
    const int first = bm_start(bm, plane), nlists = (bm_start(bm, plane + 1) - first) * wl;
    __shared__ uint64_t s_bound;
    __shared__ int s_nf;
    float *os = out_scores + (size_t)plane * k;
    int64_t *oi = out_inds + (size_t)plane * k;
    if (hist_all)  // leave the workspace clean (zero slots + this geometry's magic) for the next call
        for (int i = tid; i < max_bands; i += blockDim.x) hist_all[(size_t)blockIdx.x * max_bands + i] = 0;
    if (blockIdx.x == 0 && tid == 0) *ws_magic = hist_all ? magic : 0ull;
    merge_plane<NMS_MODE, FUSED>(band_keys + (size_t)first * wl * k, lds64, lds64 + (size_t)max_bands * wl * k,
                                 &s_bound, &s_nf, tid, 256,
                                 in + (FUSED ? (size_t)plane * (H >> 2) * (W >> 2) : (size_t)plane * H * W), H, W, k, nlists,
                                 t_sub, [&](int rank, float v, long idx) { os[rank] = v; oi[rank] = (int64_t)idx; },
                                 FUSED ? flip_partner(fs, in, plane, (size_t)(H >> 2) * (W >> 2)) : nullptr);
}

// Merge + K2 in one launch (the tail of og_generate_limbs_f32): workgroup (image, limb type) merges the limb's two joint
// planes itself -- 256 threads per plane, both lists stay in LDS -- and its first wave pairs the candidates straight from
// there (collect_body.h); behind them ceil(planes / 2) workgroups write the (N,C,k) lists the caller sees and clean the
// workspace.  A plane is merged by every limb that uses it (2.2 times on the COCO skeleton): 4 KB of L2 reads each, against
// a launch, a launch gap and a round trip of the lists through memory.
// FUSED: `in` = the stride-4 heat maps (the zero-fill path of merge_plane evaluates the x4 bicubic at single points).
template <int ND, bool FUSED = false>
__global__ void __launch_bounds__(512)
merge_collect_kernel(const uint64_t *__restrict__ band_keys, int *__restrict__ hist_all,
                     uint64_t *__restrict__ ws_magic, uint64_t magic, const float *__restrict__ in, BandMap bm, int max_bands,
                     int wl, int t_sub, float *__restrict__ out_scores, int64_t *__restrict__ out_inds, og_collect::Args a, int NL,
                     FlipSrc fs)
{
    extern __shared__ __attribute__((aligned(16))) uint64_t lds64[];
    __shared__ uint64_t s_bound[2];
    __shared__ int s_nf[2];
    MERGE_STAMP(0);
    const int planes = bm.planes, k = a.K, H = a.H, W = a.W, n_all = max_bands * wl * k, Kp = (k + 3) & ~3;   // n_all: LDS carve-up
    const int half = threadIdx.x >> 8, tid = threadIdx.x & 255;
    uint64_t *all = lds64 + (size_t)half * 2 * n_all, *flt = all + n_all;
    float *ls = reinterpret_cast<float *>(lds64 + 4 * (size_t)n_all);   // [2][Kp] scores | [2][Kp] indices | pairing scratch
    int *li = reinterpret_cast<int *>(ls + 2 * Kp);
    float *sm = ls + 4 * Kp;
    const bool limb = (int)blockIdx.x < NL;
    int plane, n = 0, l = 0;
    bool live = true;
    if (limb) {
        n = blockIdx.x / a.L;
        l = blockIdx.x % a.L;
        plane = n * a.C + (half ? a.jt[l] : a.jf[l]);
    } else {
        plane = 2 * ((int)blockIdx.x - NL) + half;
        live = plane < planes;
        plane = live ? plane : planes - 1;   // the odd plane out: merged twice, written once
        if (live && hist_all)
            for (int i = tid; i < max_bands; i += 256) hist_all[(size_t)plane * max_bands + i] = 0;
        if (plane == 0 && tid == 0) *ws_magic = hist_all ? magic : 0ull;
    }
    float *os = limb ? ls + half * Kp : out_scores + (size_t)plane * k;
    int *oi32 = li + half * Kp;
    int64_t *oi64 = out_inds + (size_t)plane * k;
    const int first = bm_start(bm, plane), nlists = (bm_start(bm, plane + 1) - first) * wl;
    const size_t plane_elems = FUSED ? (size_t)(H >> 2) * (W >> 2) : (size_t)H * W;
    merge_plane<true, FUSED>(band_keys + (size_t)first * wl * k, all, flt, &s_bound[half],
                             &s_nf[half], tid, 256, in + (size_t)plane * plane_elems, H, W, k, nlists, t_sub,
                             [&](int rank, float v, long idx) {
                                 if (limb) { os[rank] = v; oi32[rank] = (int)idx; }
                                 else if (live) { os[rank] = v; oi64[rank] = (int64_t)idx; }
                             },
                             FUSED ? flip_partner(fs, in, plane, plane_elems) : nullptr);
    if (!limb) return;
    __syncthreads();
    MERGE_STAMP(1);
    if (threadIdx.x < 64) og_collect::limb_rows<ND, int>(a, n, l, threadIdx.x, ls, li, ls + Kp, li + Kp, sm);
    MERGE_STAMP(2);
}

struct Plan {
    int vec, rows, nbands, panel_strips, nwaves, cap, t_sub;   // rows / nbands: equal bands (og_hmp_nms_f32)
    BandMap bm;       // the band kernel's work items
    int max_bands;    // most bands a plane has (slot table stride, LDS of the merge stage)
    int total;        // work items = workgroups of the band kernel
    int wl;           // sorted k-lists a band hands to the merge: one (cross-wave ranking in the band kernel) or one per streaming wave
    size_t keys_off, hist_off, magic_off, bytes;
    uint64_t magic;
};

int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

inline int device_cu_count() { return og_cu_count(); }

bool make_plan(long planes, int H, int W, int k, bool aligned16, Plan *p)
{
    p->vec = (W % 4 == 0 && aligned16) ? 4 : 1;
    const int strips = (W + p->vec - 1) / p->vec;
    p->nwaves = (strips + kInterior - 1) / kInterior;
    if (p->nwaves > kMaxWaves) return false;
    p->panel_strips = (strips + p->nwaves - 1) / p->nwaves;
    int rows = env_int("OG_NMS_ROWS", 80);
    rows = max(rows, (H + 63) / 64);    // the slot table is scanned by one wave: at most 64 bands
    rows = min(rows, H);
    p->rows = rows;
    p->nbands = (H + rows - 1) / rows;
    // Work items: equal bands of OG_NMS_ROWS rows.  (A balanced map -- exactly n workgroups per CU, bands of 85 / 86 rows -- was 1 us
    // faster stand-alone on HBM-cold data and 0-2 us slower in the decode pipeline: EXPERIMENTS.md; BandMap still describes both.)
    p->bm = BandMap{(int)planes, p->nbands, 0};
    p->max_bands = p->bm.b_lo + (p->bm.n_hi > 0 ? 1 : 0);
    p->total = bm_start(p->bm, (int)planes);
    p->cap = (2 * k + 64 + 63) / 64 * 64;
    if ((size_t)p->nwaves * 2 * p->cap * sizeof(uint64_t) > 60 * 1024) return false;
    // One list per streaming wave when the merge stage can hold them (4 x lists x k keys of LDS for a limb's two planes):
    // the band kernel then ends without a workgroup barrier and without ranking its waves' lists against each other
    // (3.4 us per workgroup at bs8 640x640, all of it exposed behind the last band); OG_K1_WAVE_LISTS=0: one list per band.
    static const int wave_lists = env_int("OG_K1_WAVE_LISTS", 1);
    p->wl = (wave_lists && p->nwaves > 1 && (size_t)4 * p->max_bands * p->nwaves * k * sizeof(uint64_t) <= 40 * 1024) ? p->nwaves : 1;
    // subset depth for the merge's lower bound: lists * t_sub >= k whenever possible
    const int min_lists = p->bm.b_lo * p->wl;
    p->t_sub = min(k, max(2, (k + min_lists - 1) / min_lists + 1));
    // [magic | slot tables | band keys]; the magic word sits at offset 0 for every shape and encodes the shape and the
    // work-item map, so a call with another geometry (or in plain top-k mode) invalidates whatever slot-table state an
    // earlier geometry left behind
    p->magic_off = 0;
    p->hist_off = 256;
    p->keys_off = p->hist_off + og_align_up((size_t)planes * kHistBins * sizeof(int), 256);
    p->bytes = p->keys_off + og_align_up((size_t)p->total * p->wl * k * sizeof(uint64_t), 256);
    p->magic = kWsMagic ^ ((uint64_t)planes * 0x9E3779B97F4A7C15ull + (uint64_t)H * 0x100000001B3ull +
                           (uint64_t)W * 0xC2B2AE3D27D4EB4Full + (uint64_t)k * 0x165667B19E3779F9ull +
                           (uint64_t)p->total * 0x27D4EB2F165667C5ull);
    return true;
}

struct Pairing {   // og_generate_limbs_f32: the merge launch pairs the limbs as well
    og_collect::Args a;
    int nd, N;
};

template <bool NMS_MODE, bool FUSED = false>
int run_topk(const float *in, long planes, int H, int W, int k, float *out_scores, int64_t *out_inds,
             void *workspace, size_t workspace_bytes, hipStream_t stream, const char *name, const Pairing *pair = nullptr,
             FlipSrc fs = FlipSrc{nullptr, 0, 0})
{
    OG_REQUIRE(in && out_scores && out_inds && workspace, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(planes > 0 && H > 0 && W > 0 && k > 0, OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE((long)H * W >= k, OG_EINVAL, "%s: selected index k out of range (k=%d > H*W=%ld)", name, k, (long)H * W);
    OG_REQUIRE((long)H * W < (1l << 27), OG_EINVAL, "%s: plane too large", name);
    if (NMS_MODE) OG_REQUIRE(2l * (H + W) - 4 >= k, OG_EINVAL, "%s: plane border smaller than k", name);
    Plan p;
    OG_REQUIRE(make_plan(planes, H, W, k, (uintptr_t)in % 16 == 0, &p), OG_EUNSUPPORTED, "%s: unsupported W=%d or k=%d",
               name, W, k);
    int wl = p.wl, t_sub = p.t_sub;
    if (FUSED) {  // lanes are source columns (58 interior per wave); bands start on a source row (bm_row: multiples of 4)
        const int w4 = W / 4;
        OG_REQUIRE((w4 + 57) / 58 < kMaxWaves, OG_EUNSUPPORTED, "%s: W=%d too wide", name, W);
        const int plan_waves = p.nwaves;
        p.nwaves = (w4 + 57) / 58;
        p.panel_strips = (w4 + p.nwaves - 1) / p.nwaves;
        if (p.nwaves != plan_waves) {   // (the plan sized the workspace for ITS wave count: one list per band then)
            wl = 1;
            t_sub = min(k, max(2, (k + p.bm.b_lo - 1) / p.bm.b_lo + 1));
        }
    }
    OG_REQUIRE(workspace_bytes >= p.bytes, OG_ENOSPC, "%s: workspace %zu < %zu", name, workspace_bytes, p.bytes);
    OG_REQUIRE((uintptr_t)workspace % 8 == 0, OG_EINVAL, "%s: workspace must be 8-byte aligned", name);
    uint64_t *keys = reinterpret_cast<uint64_t *>((char *)workspace + p.keys_off);
    int *hist = (NMS_MODE && p.max_bands <= 64) ? reinterpret_cast<int *>((char *)workspace + p.hist_off) : nullptr;
    uint64_t *magic = reinterpret_cast<uint64_t *>((char *)workspace + p.magic_off);

    const long total = p.total;
    OG_REQUIRE(total < (1l << 30), OG_EINVAL, "%s: too many work items", name);
    const int padded = (int)((total + 7) / 8 * 8);
    const int helper = (hist != nullptr && p.nwaves < kMaxWaves) ? 1 : 0;   // extra wave: threshold exchange
    const dim3 block(64 * (p.nwaves + (helper & 1)));
    const size_t lds = (size_t)(p.nwaves + (helper & 1)) * 2 * p.cap * sizeof(uint64_t) + (kHistBins + 2 * kMaxWaves + 4) * sizeof(int);
    if (FUSED)
        hipLaunchKernelGGL((band_topk_kernel<4, NMS_MODE, kPrefetch, kBandAbl, true>), dim3(padded), block, lds, stream, in, keys,
                           hist, magic, p.magic, H, W, k, p.cap, p.bm, p.max_bands, p.panel_strips, (int)total, padded,
                           helper, wl, fs);
    else if (p.vec == 4)
        hipLaunchKernelGGL((band_topk_kernel<4, NMS_MODE, kPrefetch, kBandAbl>), dim3(padded), block, lds, stream, in, keys,
                           hist, magic, p.magic, H, W, k, p.cap, p.bm, p.max_bands, p.panel_strips, (int)total, padded, helper,
                           wl, fs);
    else
        hipLaunchKernelGGL((band_topk_kernel<1, NMS_MODE, kPrefetch>), dim3(padded), block, lds, stream, in, keys,
                           hist, magic, p.magic, H, W, k, p.cap, p.bm, p.max_bands, p.panel_strips, (int)total, padded, helper,
                           wl, fs);
    OG_LAUNCH_CHECK(name);
    // dynamic LDS the merge kernels may ask for without raising the 64 KiB default: their static __shared__ words (bounds,
    // counters: 24 B) come on top
    constexpr size_t kDynLdsLimit = 64 * 1024 - 256;
    const size_t mlds = (size_t)2 * p.max_bands * wl * k * sizeof(uint64_t);
    OG_REQUIRE(mlds <= kDynLdsLimit, OG_EUNSUPPORTED, "%s: k*bands too large for the merge stage", name);
    if constexpr (NMS_MODE) {
        const size_t plds = 2 * mlds + (size_t)((k + 3) & ~3) * 32;
        if (pair && plds <= kDynLdsLimit) {
            const int NL = pair->N * pair->a.L;
            auto kern = pair->nd == 2 ? merge_collect_kernel<2, FUSED> : merge_collect_kernel<4, FUSED>;
            hipLaunchKernelGGL(kern, dim3((unsigned)(NL + (planes + 1) / 2)), dim3(512), plds, stream, keys, hist, magic,
                               p.magic, in, p.bm, p.max_bands, wl, t_sub, out_scores, out_inds, pair->a, NL, fs);
            OG_LAUNCH_CHECK(name);
            return 1;   // paired
        }
    }
    hipLaunchKernelGGL((merge_bands_kernel<NMS_MODE, FUSED>), dim3((unsigned)planes), dim3(256), mlds, stream, keys, hist,
                       magic, p.magic, in, H, W, k, p.bm, p.max_bands, wl, t_sub, out_scores, out_inds, fs);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

}  // namespace

OG_API size_t og_topk_workspace_bytes(long planes, int H, int W, int k)
{
    Plan p;
    if (planes <= 0 || H <= 0 || W <= 0 || k <= 0) return 0;
    if (!make_plan(planes, H, W, k, true, &p)) return 0;
    return p.bytes;
}

OG_API int og_nms_topk_f32(const float *hmps, long planes, int H, int W, int k, float *out_scores,
                           int64_t *out_inds, void *workspace, size_t workspace_bytes, void *stream)
{
    return run_topk<true>(hmps, planes, H, W, k, out_scores, out_inds, workspace, workspace_bytes,
                          (hipStream_t)stream, "og_nms_topk_f32");
}

OG_API int og_upsample_nms_topk_f32(const float *hmps_lr, long planes, int h, int w, int k, float *out_scores,
                                    int64_t *out_inds, void *workspace, size_t workspace_bytes, void *stream)
{
    const char *name = "og_upsample_nms_topk_f32";
    OG_REQUIRE(h > 0 && w > 0 && h < (1 << 14) && w < (1 << 14), OG_EINVAL, "%s: bad shape", name);
    return run_topk<true, true>(hmps_lr, planes, 4 * h, 4 * w, k, out_scores, out_inds, workspace, workspace_bytes,
                                (hipStream_t)stream, name);
}

OG_API int og_topk_channel_f32(const float *scores, long planes, int H, int W, int k, float *out_scores,
                               int64_t *out_inds, void *workspace, size_t workspace_bytes, void *stream)
{
    return run_topk<false>(scores, planes, H, W, k, out_scores, out_inds, workspace, workspace_bytes,
                           (hipStream_t)stream, "og_topk_channel_f32");
}

OG_API int og_hmp_nms_f32(const float *heat, long planes, int H, int W, float *out, void *stream)
{
    const char *name = "og_hmp_nms_f32";
    OG_REQUIRE(heat && out, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(planes > 0 && H > 0 && W > 0, OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE((long)H * W < (1l << 27), OG_EINVAL, "%s: plane too large", name);
    Plan p;
    const bool aligned = (uintptr_t)heat % 16 == 0 && (uintptr_t)out % 16 == 0;
    OG_REQUIRE(make_plan(planes, H, W, 1, aligned, &p), OG_EUNSUPPORTED, "%s: unsupported W=%d", name, W);
    const long total = planes * p.nbands;
    OG_REQUIRE(total < (1l << 30), OG_EINVAL, "%s: too many work items", name);
    const int padded = (int)((total + 7) / 8 * 8);
    if (p.vec == 4)
        hipLaunchKernelGGL((nms_map_kernel<4>), dim3(padded), dim3(64 * p.nwaves), 0, (hipStream_t)stream, heat, out, H, W,
                           p.rows, p.nbands, p.panel_strips, (int)total, padded);
    else
        hipLaunchKernelGGL((nms_map_kernel<1>), dim3(padded), dim3(64 * p.nwaves), 0, (hipStream_t)stream, heat, out, H, W,
                           p.rows, p.nbands, p.panel_strips, (int)total, padded);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

// ---- a8+a9+a10 in one launch: LimbsCollect.generate_limbs  decoder/collect.py:62-236 ----
namespace {

// head of every og_generate_limbs_f32 workspace: reserved (the one-launch forms of rounds 2 / 3 kept their tickets there;
// tools/experiments/k1_single.inc); the size stays so that workspaces sized by earlier builds keep working
constexpr size_t kP2TicketBytes = 64 * 1024;

size_t two_step_bytes(int N, int C, int H, int W, int k, bool need_lists)
{
    const size_t topk = og_topk_workspace_bytes((long)N * C, H, W, k);
    if (topk == 0) return 0;
    return kP2TicketBytes + og_align_up(topk, 256) + (need_lists ? og_align_up((size_t)N * C * k * 12, 256) : 0);
}

}  // namespace

OG_API size_t og_generate_limbs_workspace_bytes(int N, int C, int H, int W, int k)
{
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0) return 0;
    return two_step_bytes(N, C, H, W, k, true);
}

// hm_lowres: hmps_hr holds the STRIDE-4 heat maps (N [2N with kp_perm], C, H/4, W/4) and the x4 bicubic runs inside the band kernel
// (K1-fused); kp_perm (with hm_lowres only): flip-test, the heat maps of [images | mirrored images] merged on the fly.
static int generate_limbs_impl(const char *name, const float *hmps_hr, const float *offs, int off_is_lowres, int vector_nd,
                               const float *scales, int scales_mode, const float *jitter, int jitter_mode,
                               int N, int C, int H, int W, const int32_t *jf, const int32_t *jt, int L, int k,
                               float thre_hmp, float min_len, float resize_factor, float *topk_scores,
                               int64_t *topk_inds, float *limbs, const int32_t *limb_perm, const int32_t *reserve_mask,
                               void *workspace, size_t workspace_bytes, void *stream, bool hm_lowres = false,
                               const int32_t *kp_perm = nullptr)
{
    OG_REQUIRE(hmps_hr && offs && jf && jt && limbs && workspace, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE((topk_scores == nullptr) == (topk_inds == nullptr), OG_EINVAL, "%s: topk_scores and topk_inds go together", name);
    OG_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && L > 0 && k > 0, OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE((jitter_mode == 0 || jitter_mode == 1 || jitter_mode == 3) && (jitter_mode == 0) == (jitter == nullptr),
               OG_EINVAL, "%s: jitter_mode 0 (no head), 1 (hi-res maps) or 3 (stride-4 maps), with a map exactly when not 0", name);
    OG_REQUIRE(jitter_mode == 0 || (H == W && vector_nd == 2), OG_EUNSUPPORTED,
               "%s: the jitter refinement indexes its maps [x][y] like the reference: square inputs, 2-component offsets", name);
    OG_REQUIRE(scales_mode >= 0 && scales_mode <= 3 && (scales_mode == 0) == (scales == nullptr), OG_EINVAL,
               "%s: scales_mode 0 (no scale head) .. 3, with a map exactly when it is not 0", name);
    OG_REQUIRE(vector_nd == 2 || vector_nd == 4, OG_EUNSUPPORTED, "%s: vector_nd must be 2 or 4", name);
    OG_REQUIRE(!(off_is_lowres || scales_mode >= 2 || jitter_mode == 3) || (H % 4 == 0 && W % 4 == 0), OG_EINVAL,
               "%s: H,W must be multiples of 4", name);
    OG_REQUIRE((long)H * W >= k, OG_EINVAL, "%s: selected index k out of range (k=%d > H*W=%ld)", name, k, (long)H * W);
    OG_REQUIRE(2l * (H + W) - 4 >= k, OG_EINVAL, "%s: plane border smaller than k", name);
    OG_REQUIRE((uintptr_t)workspace % 16 == 0, OG_EINVAL, "%s: workspace must be 16-byte aligned", name);
    og_collect::Args ca{offs, off_is_lowres, C, H, W, jf, jt, L, k, thre_hmp, min_len, resize_factor, scales, scales_mode,
                        jitter, jitter_mode, limbs};
    if (limb_perm) { ca.limb_perm = limb_perm; ca.reserve = reserve_mask; ca.flip_N = N; }
    // band top-k, then ONE launch that merges the band lists and pairs the limbs
    const bool own_lists = topk_scores == nullptr;
    const size_t need = two_step_bytes(N, C, H, W, k, own_lists);
    OG_REQUIRE(need != 0, OG_EUNSUPPORTED, "%s: unsupported W=%d or k=%d", name, W, k);
    OG_REQUIRE(workspace_bytes >= need, OG_ENOSPC, "%s: workspace %zu < %zu", name, workspace_bytes, need);
    char *ws2 = (char *)workspace + kP2TicketBytes;
    const size_t topk = og_align_up(og_topk_workspace_bytes((long)N * C, H, W, k), 256);
    float *sc = own_lists ? reinterpret_cast<float *>(ws2 + topk + (size_t)N * C * k * 8) : topk_scores;
    int64_t *id = own_lists ? reinterpret_cast<int64_t *>(ws2 + topk) : topk_inds;
    const Pairing pr{ca, vector_nd, N};
    const bool can_pair = (long)H * W < (1l << 31) && k <= 2048;
    OG_REQUIRE(!hm_lowres || (H % 4 == 0 && W % 4 == 0), OG_EINVAL, "%s: H,W must be multiples of 4", name);
    const int rc = hm_lowres ? run_topk<true, true>(hmps_hr, (long)N * C, H, W, k, sc, id, ws2, topk, (hipStream_t)stream, name,
                                                    can_pair ? &pr : nullptr, FlipSrc{kp_perm, N, C})
                             : run_topk<true>(hmps_hr, (long)N * C, H, W, k, sc, id, ws2, topk, (hipStream_t)stream, name,
                                              can_pair ? &pr : nullptr);
    if (rc < 0 || rc == 1) return rc < 0 ? rc : OG_OK;
    // (shapes whose merge + pairing stage does not fit the LDS: the lists are complete, pair them with the collect kernel)
    OG_REQUIRE(!limb_perm, OG_EUNSUPPORTED, "%s: k = %d is too large for the merge-and-pair stage of the flip-folded form", name, k);
    return og_collect_limbs_full_f32(sc, id, offs, off_is_lowres, vector_nd, scales, scales_mode, jitter, jitter_mode, N, C,
                                     H, W, jf, jt, L, k, thre_hmp, min_len, resize_factor, limbs, stream);
}

OG_API int og_generate_limbs_f32(const float *hmps_hr, const float *offs, int off_is_lowres, int vector_nd,
                                 const float *scales, int scales_mode, const float *jitter, int jitter_mode,
                                 int N, int C, int H, int W, const int32_t *jf, const int32_t *jt, int L, int k,
                                 float thre_hmp, float min_len, float resize_factor, float *topk_scores,
                                 int64_t *topk_inds, float *limbs, int flags, void *workspace, size_t workspace_bytes, void *stream)
{
    (void)flags;    // reserved (the one-launch forms of rounds 2 / 3 were selected here): pass 0
    return generate_limbs_impl("og_generate_limbs_f32", hmps_hr, offs, off_is_lowres, vector_nd, scales, scales_mode, jitter,
                               jitter_mode, N, C, H, W, jf, jt, L, k, thre_hmp, min_len, resize_factor, topk_scores, topk_inds,
                               limbs, nullptr, nullptr, workspace, workspace_bytes, stream);
}

// generate_limbs with flip_augment's OFFSET merge (decoder/factory.py:129-138) folded into the offset sampling: offs_pair is the
// stride-4 offset head output for [images | mirrored images], (2N, 2L, H/4, W/4)
OG_API int og_generate_limbs_flip_f32(const float *hmps_hr, const float *offs_pair, const int32_t *limb_perm,
                                      const int32_t *reserve_mask, int N, int C, int H, int W, const int32_t *jf,
                                      const int32_t *jt, int L, int k, float thre_hmp, float min_len, float resize_factor,
                                      float *topk_scores, int64_t *topk_inds, float *limbs, void *workspace,
                                      size_t workspace_bytes, void *stream)
{
    const char *name = "og_generate_limbs_flip_f32";
    OG_REQUIRE(limb_perm && reserve_mask, OG_EINVAL, "%s: null pointer", name);
    return generate_limbs_impl(name, hmps_hr, offs_pair, 1, 2, nullptr, 0, nullptr, 0, N, C, H, W, jf, jt, L, k, thre_hmp, min_len,
                               resize_factor, topk_scores, topk_inds, limbs, limb_perm, reserve_mask, workspace, workspace_bytes,
                               stream);
}

// ---- K1-fused: generate_limbs straight from the STRIDE-4 head outputs.  The x4 bicubic of decoder/factory.py:74-75 runs inside the
// band kernel (bit-identical to og_upsample_bicubic4_f32), the offsets / scale / jitter maps are sampled at the peaks: neither hi-res
// tensor exists.  Two launches, as og_generate_limbs_f32.
OG_API int og_generate_limbs_fused_f32(const float *hmps_lr, const float *offs_lr, int vector_nd, const float *scales_lr,
                                       int scales_mode, const float *jitter_lr, int jitter_mode, int N, int C, int h, int w,
                                       const int32_t *jf, const int32_t *jt, int L, int k, float thre_hmp, float min_len,
                                       float resize_factor, float *topk_scores, int64_t *topk_inds, float *limbs, void *workspace,
                                       size_t workspace_bytes, void *stream)
{
    const char *name = "og_generate_limbs_fused_f32";
    OG_REQUIRE(h > 0 && w > 0 && h < (1 << 14) && w < (1 << 14), OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE(scales_mode == 0 || scales_mode >= 2, OG_EINVAL, "%s: the scale maps are the stride-4 head output (scales_mode 2 / 3)", name);
    OG_REQUIRE(jitter_mode == 0 || jitter_mode == 3, OG_EINVAL, "%s: the jitter maps are the stride-4 head output (jitter_mode 3)", name);
    return generate_limbs_impl(name, hmps_lr, offs_lr, 1, vector_nd, scales_lr, scales_mode, jitter_lr, jitter_mode, N, C, 4 * h, 4 * w,
                               jf, jt, L, k, thre_hmp, min_len, resize_factor, topk_scores, topk_inds, limbs, nullptr, nullptr, workspace,
                               workspace_bytes, stream, true);
}

// ... with flip_augment (decoder/factory.py:98-146, the averaged form) folded into BOTH consumers: hm_pair_lr (2N,C,h,w) and
// offs_pair_lr (2N,2L,h,w) are the head outputs for [images | mirrored images]; no merge pass, no hi-res tensor.
OG_API int og_generate_limbs_fused_flip_f32(const float *hm_pair_lr, const int32_t *kp_perm, const float *offs_pair_lr,
                                            const int32_t *limb_perm, const int32_t *reserve_mask, int N, int C, int h, int w,
                                            const int32_t *jf, const int32_t *jt, int L, int k, float thre_hmp, float min_len,
                                            float resize_factor, float *topk_scores, int64_t *topk_inds, float *limbs,
                                            void *workspace, size_t workspace_bytes, void *stream)
{
    const char *name = "og_generate_limbs_fused_flip_f32";
    OG_REQUIRE(kp_perm && limb_perm && reserve_mask, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(h > 0 && w > 0 && h < (1 << 14) && w < (1 << 14), OG_EINVAL, "%s: bad shape", name);
    return generate_limbs_impl(name, hm_pair_lr, offs_pair_lr, 1, 2, nullptr, 0, nullptr, 0, N, C, 4 * h, 4 * w, jf, jt, L, k, thre_hmp,
                               min_len, resize_factor, topk_scores, topk_inds, limbs, limb_perm, reserve_mask, workspace,
                               workspace_bytes, stream, true, kp_perm);
}

#ifdef OG_K1_STAMPS
OG_API void og_k1_debug_stamps(void *host_out) { (void)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_k1_stamps), sizeof(g_k1_stamps)); }
OG_API void og_k1_wave_stamps(void *host_out) { (void)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wave_stamps), sizeof(g_wave_stamps)); }
OG_API void og_k1_band_stamps(void *host_out) { (void)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_band_stamps), sizeof(g_band_stamps)); }
#endif
