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
//   lanes and no LDS or extra loads are needed for the stencil.  Rows are prefetched
//   PREFETCH deep (16 B/lane each) to keep >= 64 KiB in flight per CU.
//   Candidates (positive 3x3 peaks, or every pixel in plain top-k mode) are compacted with
//   ballot + popcount into a per-wave LDS segment; when a segment fills, the wave keeps its
//   own top-k (rank-by-counting on order-preserving 64-bit keys) and raises its admission
//   threshold, so the result is exact for any input.  Each band emits its k best keys; a
//   second tiny kernel (one wave per plane) k-way merges the bands and writes scores/indices.
#include <math.h>

#include "og_common.h"

namespace {

constexpr int kPrefetch = 8;      // rows in flight per lane
constexpr int kInterior = 62;     // interior lanes per wave panel
constexpr int kMaxWaves = 16;     // waves per workgroup (panels per row)

template <int VEC>
struct Px {
    float c[VEC];
};

template <int VEC>
__device__ __forceinline__ Px<VEC> load_px(const float *p, bool ok)
{
    Px<VEC> r;
    if constexpr (VEC == 4) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) t = *reinterpret_cast<const float4 *>(p);
        r.c[0] = t.x; r.c[1] = t.y; r.c[2] = t.z; r.c[3] = t.w;
    } else {
        r.c[0] = ok ? *p : 0.f;
    }
    return r;
}

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

struct TileGeom {
    int plane_rows, plane_cols;  // H, W
    int r0, r1;                  // interior rows [r0, r1)
    int col;                     // first column of this lane (may be < 0 or >= W: halo / idle)
    bool loads;                  // lane reads memory (interior or halo lane inside the image)
    bool interior;               // lane emits results
};

// Walk rows r0..r1-1 of one panel; emit(row, centre values, 3x3 max incl. zero padding).
template <int VEC, class Emit>
__device__ __forceinline__ void walk_panel(const float *__restrict__ plane, const TileGeom &g, Emit &&emit)
{
    const int H = g.plane_rows, W = g.plane_cols;
    auto load_row = [&](int row) {
        const bool ok = g.loads && row >= 0 && row < H && row <= g.r1;
        return load_px<VEC>(plane + (size_t)(ok ? row : 0) * W + (ok ? g.col : 0), ok);
    };
    Px<VEC> hm_a = hmax3<VEC>(load_row(g.r0 - 1));
    Px<VEC> v_b = load_row(g.r0);
    Px<VEC> hm_b = hmax3<VEC>(v_b);
    Px<VEC> q[kPrefetch];
#pragma unroll
    for (int u = 0; u < kPrefetch; ++u) q[u] = load_row(g.r0 + 1 + u);
    for (int r = g.r0; r < g.r1; r += kPrefetch) {
#pragma unroll
        for (int u = 0; u < kPrefetch; ++u) {
            const Px<VEC> v_c = q[u];
            q[u] = load_row(r + u + 1 + kPrefetch);
            const Px<VEC> hm_c = hmax3<VEC>(v_c);  // all lanes take part in the DPP shifts
            if (r + u < g.r1) {
                Px<VEC> m;
#pragma unroll
                for (int j = 0; j < VEC; ++j) m.c[j] = og_max3(hm_a.c[j], hm_b.c[j], hm_c.c[j]);
                emit(r + u, v_b, m);
            }
            hm_a = hm_b;
            hm_b = hm_c;
            v_b = v_c;
        }
    }
}

__device__ __forceinline__ TileGeom make_geom(int H, int W, int rows, int band, int panel_strips, int vec)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int strips = (W + vec - 1) / vec;
    const int s_first = wave * panel_strips;                       // first interior strip of the panel
    const int s_cnt = min(panel_strips, strips - s_first);         // interior strips in this panel
    const int s = s_first - 1 + lane;                              // this lane's strip
    TileGeom g;
    g.plane_rows = H;
    g.plane_cols = W;
    g.r0 = band * rows;
    g.r1 = min(g.r0 + rows, H);
    g.col = s * vec;
    g.interior = lane >= 1 && lane <= s_cnt;
    g.loads = lane <= s_cnt + 1 && s >= 0 && s < strips;
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
    const TileGeom g = make_geom(H, W, rows, band, panel_strips, VEC);
    const float *src = in + (size_t)plane * H * W;
    float *dst = out + (size_t)plane * H * W;
    walk_panel<VEC>(src, g, [&](int row, const Px<VEC> &v, const Px<VEC> &m) {
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
// Per-wave candidate segment in LDS.  KPL = keys per lane during a compaction, capacity
// 64*KPL; a compaction leaves <= k keys, so k + 64 <= capacity is required.
template <int KPL>
struct WaveSeg {
    uint64_t *keys;  // LDS, 64*KPL entries
    int cnt;         // wave-uniform
    float tau;       // admit v >= tau (wave-uniform)

    // Keep the k largest keys, sorted descending, in keys[0..min(cnt,k)).
    __device__ __forceinline__ void compact(int k)
    {
        const int lane = threadIdx.x & 63;
        uint64_t mine[KPL];
        int rank[KPL];
#pragma unroll
        for (int i = 0; i < KPL; ++i) {
            const int p = lane + 64 * i;
            mine[i] = (p < cnt) ? keys[p] : 0ull;
            rank[i] = 0;
        }
        for (int j = 0; j < cnt; ++j) {
            const uint64_t o = keys[j];  // LDS broadcast
#pragma unroll
            for (int i = 0; i < KPL; ++i) rank[i] += (o > mine[i]);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < KPL; ++i)
            if (lane + 64 * i < cnt && rank[i] < k) keys[rank[i]] = mine[i];
        __builtin_amdgcn_wave_barrier();
        if (cnt >= k) {
            cnt = k;
            tau = og_key_value(keys[k - 1]);
        }
    }

    __device__ __forceinline__ void push(bool pred, uint64_t key, int k)
    {
        const uint64_t mask = __builtin_amdgcn_ballot_w64(pred);
        if (mask == 0) return;
        const int n = __builtin_popcountll(mask);
        if (cnt + n > 64 * KPL) compact(k);
        const int lane = threadIdx.x & 63;
        const int pos = cnt + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (pred) keys[pos] = key;
        cnt += n;
    }
};

// NMS_MODE: candidates are strictly positive 3x3 peaks (zero padding); otherwise every pixel.
template <int VEC, int KPL, bool NMS_MODE>
__global__ void __launch_bounds__(64 * kMaxWaves)
band_topk_kernel(const float *__restrict__ in, uint64_t *__restrict__ band_keys, int *__restrict__ band_cnt,
                 int H, int W, int k, int rows, int nbands, int panel_strips, int total, int padded)
{
    extern __shared__ uint64_t smem[];
    __shared__ int s_cnt[kMaxWaves];
    const int wid = og_xcd_remap(blockIdx.x, padded);
    if (wid >= total) return;
    const int plane = wid / nbands, band = wid % nbands;
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const TileGeom g = make_geom(H, W, rows, band, panel_strips, VEC);
    const float *src = in + (size_t)plane * H * W;

    WaveSeg<KPL> seg;
    seg.keys = smem + (size_t)wave * 64 * KPL;
    seg.cnt = 0;
    seg.tau = NMS_MODE ? 0.f : -INFINITY;

    walk_panel<VEC>(src, g, [&](int row, const Px<VEC> &v, const Px<VEC> &m) {
        bool pred[VEC];
        bool any = false;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const bool inimg = g.interior && (g.col + j < W);
            pred[j] = NMS_MODE ? (inimg && v.c[j] > 0.f && v.c[j] == m.c[j] && v.c[j] >= seg.tau)
                               : (inimg && v.c[j] >= seg.tau);
            any |= pred[j];
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(any) != 0ull, 0)) {
            const uint32_t base = (uint32_t)row * (uint32_t)W + (uint32_t)g.col;
#pragma unroll
            for (int j = 0; j < VEC; ++j) seg.push(pred[j] && v.c[j] >= seg.tau, og_make_key(v.c[j], base + j), k);
        }
    });

    // per-wave top-k, then merge the waves' lists by rank counting
    seg.compact(k);
    if ((threadIdx.x & 63) == 0) s_cnt[wave] = min(seg.cnt, k);
    __syncthreads();
    int total_keys = 0;
    for (int w = 0; w < nwaves; ++w) total_keys += s_cnt[w];
    uint64_t *out = band_keys + ((size_t)plane * nbands + band) * k;
    for (int t = threadIdx.x; t < total_keys; t += blockDim.x) {
        int w = 0, o = t;
        while (o >= s_cnt[w]) { o -= s_cnt[w]; ++w; }
        const uint64_t key = smem[(size_t)w * 64 * KPL + o];
        int rank = 0;
        for (int w2 = 0; w2 < nwaves; ++w2) {
            const uint64_t *kk = smem + (size_t)w2 * 64 * KPL;
            const int c2 = s_cnt[w2];
            for (int j = 0; j < c2; ++j) rank += (kk[j] > key);
        }
        if (rank < k) out[rank] = key;
    }
    if (threadIdx.x == 0) band_cnt[(size_t)plane * nbands + band] = min(total_keys, k);
}

// ---------------------------------------------------------------------------------------
// merge kernel: one wave per plane, k-way tournament over the (sorted) band lists
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, off);
        const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), off);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

template <bool NMS_MODE>
__global__ void __launch_bounds__(64)
merge_bands_kernel(const uint64_t *__restrict__ band_keys, const int *__restrict__ band_cnt,
                   const float *__restrict__ in, int H, int W, int k, int nbands,
                   float *__restrict__ out_scores, int64_t *__restrict__ out_inds)
{
    extern __shared__ uint64_t skeys[];  // nbands * k (when it fits) else unused
    const int plane = blockIdx.x, lane = threadIdx.x;
    const uint64_t *gk = band_keys + (size_t)plane * nbands * k;
    const int *gc = band_cnt + (size_t)plane * nbands;
    float *os = out_scores + (size_t)plane * k;
    int64_t *oi = out_inds + (size_t)plane * k;

    // stage the valid prefix of every band list in LDS (bands beyond 64 are folded per lane)
    for (int b = 0; b < nbands; ++b) {
        const int c = gc[b];
        for (int j = lane; j < c; j += 64) skeys[(size_t)b * k + j] = gk[(size_t)b * k + j];
    }
    __builtin_amdgcn_wave_barrier();
    // lane owns bands lane, lane+64, ...; ptr = how many keys it has consumed from each
    constexpr int kFold = 4;  // up to 256 bands
    int ptr[kFold], cnt[kFold];
#pragma unroll
    for (int f = 0; f < kFold; ++f) {
        const int b = lane + 64 * f;
        ptr[f] = 0;
        cnt[f] = (b < nbands) ? gc[b] : 0;
    }
    int t = 0;
    for (; t < k; ++t) {
        uint64_t best = 0ull;
        int bf = 0;
#pragma unroll
        for (int f = 0; f < kFold; ++f) {
            const int b = lane + 64 * f;
            const uint64_t c = (ptr[f] < cnt[f]) ? skeys[(size_t)b * k + ptr[f]] : 0ull;
            if (c > best) { best = c; bf = f; }
        }
        const uint64_t top = wave_max_u64(best);
        if (top == 0ull) break;
        if (best == top) {  // keys are unique: exactly one lane
#pragma unroll
            for (int f = 0; f < kFold; ++f) ptr[f] += (f == bf);
            os[t] = og_key_value(top);
            oi[t] = (int64_t)og_key_index(top);
        }
    }
    if (NMS_MODE && t < k) {
        // fewer than k positive peaks: fill with the lowest flat indices whose NMS output is
        // zero (ties at 0.0 broken by index, like every other tie)
        const float *p = in + (size_t)plane * H * W;
        const long hw = (long)H * W;
        for (long base = 0; base < hw && t < k; base += 64) {
            const long i = base + lane;
            bool zero = false;
            if (i < hw) {
                const int y = (int)(i / W), x = (int)(i % W);
                const float v = p[i];
                float m = (y == 0 || x == 0 || y == H - 1 || x == W - 1) ? 0.f : -INFINITY;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int yy = y + dy, xx = x + dx;
                        if (yy >= 0 && yy < H && xx >= 0 && xx < W) m = fmaxf(m, p[(size_t)yy * W + xx]);
                    }
                zero = !(v == m && v != 0.f);
            }
            const uint64_t mask = __builtin_amdgcn_ballot_w64(zero);
            const int slot = t + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            if (zero && slot < k) {
                os[slot] = 0.f;
                oi[slot] = i;
            }
            t += __builtin_popcountll(mask);
        }
    }
}

struct Plan {
    int vec, rows, nbands, panel_strips, nwaves, kpl;
    size_t keys_off, cnt_off, bytes;
};

int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

bool make_plan(long planes, int H, int W, int k, const void *base, Plan *p)
{
    p->vec = (W % 4 == 0 && ((uintptr_t)base % 16 == 0)) ? 4 : 1;
    const int strips = (W + p->vec - 1) / p->vec;
    p->nwaves = (strips + kInterior - 1) / kInterior;
    if (p->nwaves > kMaxWaves) return false;
    p->panel_strips = (strips + p->nwaves - 1) / p->nwaves;
    int rows = env_int("OG_NMS_ROWS", 32);
    rows = max(rows, (H + 63) / 64);  // the merge wave folds at most 256 bands; keep it <= 64 normally
    rows = min(rows, H);
    p->rows = rows;
    p->nbands = (H + rows - 1) / rows;
    p->kpl = (k + 64 <= 256) ? 4 : 8;
    if (k + 64 > 64 * p->kpl) return false;
    p->keys_off = 0;
    p->cnt_off = og_align_up((size_t)planes * p->nbands * k * sizeof(uint64_t), 256);
    p->bytes = p->cnt_off + og_align_up((size_t)planes * p->nbands * sizeof(int), 256);
    return true;
}

template <bool NMS_MODE>
int run_topk(const float *in, long planes, int H, int W, int k, float *out_scores, int64_t *out_inds,
             void *workspace, size_t workspace_bytes, hipStream_t stream, const char *name)
{
    OG_REQUIRE(in && out_scores && out_inds && workspace, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(planes > 0 && H > 0 && W > 0 && k > 0, OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE((long)H * W >= k, OG_EINVAL, "%s: selected index k out of range (k=%d > H*W=%ld)", name, k, (long)H * W);
    OG_REQUIRE((long)H * W < (1l << 32), OG_EINVAL, "%s: plane too large", name);
    if (NMS_MODE) OG_REQUIRE(2l * (H + W) - 4 >= k, OG_EINVAL, "%s: plane border smaller than k", name);
    Plan p;
    OG_REQUIRE(make_plan(planes, H, W, k, in, &p), OG_EUNSUPPORTED, "%s: unsupported W=%d or k=%d", name, W, k);
    // planes must share the alignment decision
    if (p.vec == 4 && ((size_t)H * W * sizeof(float)) % 16 != 0) p.vec = 1;
    OG_REQUIRE(workspace_bytes >= p.bytes, OG_ENOSPC, "%s: workspace %zu < %zu", name, workspace_bytes, p.bytes);
    OG_REQUIRE((uintptr_t)workspace % 8 == 0, OG_EINVAL, "%s: workspace must be 8-byte aligned", name);
    OG_REQUIRE(p.nbands <= 256, OG_EUNSUPPORTED, "%s: too many bands", name);
    uint64_t *keys = reinterpret_cast<uint64_t *>((char *)workspace + p.keys_off);
    int *cnts = reinterpret_cast<int *>((char *)workspace + p.cnt_off);

    const long total = planes * p.nbands;
    OG_REQUIRE(total < (1l << 30), OG_EINVAL, "%s: too many work items", name);
    const int padded = (int)((total + 7) / 8 * 8);
    const dim3 block(64 * p.nwaves);
    const size_t lds = (size_t)p.nwaves * 64 * p.kpl * sizeof(uint64_t);
#define OG_BAND(VEC, KPL)                                                                                   \
    hipLaunchKernelGGL((band_topk_kernel<VEC, KPL, NMS_MODE>), dim3(padded), block, lds, stream, in, keys, \
                       cnts, H, W, k, p.rows, p.nbands, p.panel_strips, (int)total, padded)
    if (p.vec == 4 && p.kpl == 4) OG_BAND(4, 4);
    else if (p.vec == 4) OG_BAND(4, 8);
    else if (p.kpl == 4) OG_BAND(1, 4);
    else OG_BAND(1, 8);
#undef OG_BAND
    OG_LAUNCH_CHECK(name);
    const size_t mlds = (size_t)p.nbands * k * sizeof(uint64_t);
    OG_REQUIRE(mlds <= 64 * 1024, OG_EUNSUPPORTED, "%s: k*bands too large for the merge stage", name);
    hipLaunchKernelGGL((merge_bands_kernel<NMS_MODE>), dim3((unsigned)planes), dim3(64), mlds, stream, keys, cnts, in, H,
                       W, k, p.nbands, out_scores, out_inds);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

}  // namespace

OG_API size_t og_topk_workspace_bytes(long planes, int H, int W, int k)
{
    Plan p;
    if (planes <= 0 || H <= 0 || W <= 0 || k <= 0) return 0;
    if (!make_plan(planes, H, W, k, nullptr, &p)) return 0;
    return p.bytes;
}

OG_API int og_nms_topk_f32(const float *hmps, long planes, int H, int W, int k, float *out_scores,
                           int64_t *out_inds, void *workspace, size_t workspace_bytes, void *stream)
{
    return run_topk<true>(hmps, planes, H, W, k, out_scores, out_inds, workspace, workspace_bytes,
                          (hipStream_t)stream, "og_nms_topk_f32");
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
    Plan p;
    OG_REQUIRE(make_plan(planes, H, W, 1, heat, &p), OG_EUNSUPPORTED, "%s: unsupported W=%d", name, W);
    if ((uintptr_t)out % 16 != 0 || ((size_t)H * W * sizeof(float)) % 16 != 0) p.vec = 1;
    if (p.vec == 1) {  // recompute the panel split for scalar lanes
        p.nwaves = (W + kInterior - 1) / kInterior;
        OG_REQUIRE(p.nwaves <= kMaxWaves, OG_EUNSUPPORTED, "%s: W=%d too wide for the unaligned path", name, W);
        p.panel_strips = (W + p.nwaves - 1) / p.nwaves;
    }
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
