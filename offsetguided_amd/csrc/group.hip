// K3 -- GreedyGroup.group_skeletons (decoder/group.py:39-185; _delete_reconns :221-240;
// _delete_sort :187-219), device resident: one 16-wave workgroup (1 024 threads) per image.
//
// The reference ships the limbs to the host (.cpu().numpy(), decoder/factory.py:91) and runs
// numpy in a multiprocessing.Pool; here the partial-skeleton ("subset") table lives in LDS
// (mmax x n_kp x 6 fp32, rows addressed through an order[] indirection so deletions never move
// data).  Everything that does not depend on the table -- validity filter, stable rank sort and
// to-index de-duplication of the K candidates of EVERY limb type -- is done up front in parallel;
// the limb types are then processed serially (that dependence is inherent), each step spread over
// the workgroup's 1 024 lanes:
//   * matching: lanes = (subset row, limb column) cells; per-row results by LDS atomicMax
//     (phase B keeps the LAST matching column: numpy last-write-wins), per-column match counts
//     by LDS atomicAdd;
//   * merge search: lanes = 16x16 tiles of (row a, row b) pairs, 17 index compares each, last
//     partner by LDS atomicMax;
//   * new rows: slot assignment by ballot + popcount in one wave, row fill by everyone.
// Measured on MI355X (bs8, ~25 skeletons/image): 140 us per batch = up-front 21 us, pair search
// 32 us, final scoring/sort/copy 14 us, the rest ~2 us per limb type (cycle stamps, tools/k3_experiments).
//
// numpy fancy-assignment semantics are kept exactly (see oracle/og_oracle.c for the literal
// loop form).  Per subset row they reduce to:
//   phase A: at most one limb column has both endpoints in the row (to-indices are unique);
//            if it is "better" both limb scores become max(score, old);
//   phase B: of all columns sharing exactly one endpoint (and better), only the LAST one
//            sticks; its max() uses the scores after phase A;
//   matrix resets (all ms==2 -> -1 if any A pair fired, all ms==1 -> -1 if any B pair fired)
//            only matter through the column sums that decide which limbs start new rows;
//   merge:   row a takes max(row a, LAST row b sharing exactly two keypoints), computed from
//            the pre-merge state; every such b is deleted, order kept.  Deleted rows are never
//            written, so in-place updates read only pre-merge data;
//   final:   mean of positive entries of column `sort_dim` (numpy pairwise fp32 sum, fp64
//            divide), fp64 threshold, stable descending order, -1 -> 0.
// If more than mmax rows are ever created the image is flagged in status[] (caller retries
// with a larger mmax; the table then lives in the caller's global workspace instead of LDS).
// Capacity: the staged candidate rows (11 floats each) move to the global workspace as well when L*k
// does not leave room for them in LDS (GLIM; e.g. omp44 at --topk 64); what must stay in LDS is
// 16 B per candidate + 108 B per table row: L*k up to ~6 000 at mmax 128, any skeleton the reference defines at k <= 128.
#include <math.h>

#include <type_traits>

#include "og_common.h"

namespace {

#ifndef OG_K3_THREADS
#define OG_K3_THREADS 1024
#endif
constexpr int kThreads = OG_K3_THREADS;   // a multiple of 256
constexpr int kScoreThreads = 256;         // threads that score the rows at the end (17 floats of LDS scratch each)

#ifdef OG_K3_STAMPS
__device__ unsigned long long g_k3_stamps[16];
__device__ unsigned long long g_k3_wall[64 * 2];   // [workgroup][entry, exit] on the 100 MHz wall clock (s_memrealtime)
#define K3_STAMP_INIT unsigned long long t_prev_ = __builtin_amdgcn_s_memtime(), acc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    if (threadIdx.x == 0 && blockIdx.x < 64) g_k3_wall[blockIdx.x * 2] = __builtin_amdgcn_s_memrealtime()
#define K3_STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc_[i] += t_ - t_prev_; t_prev_ = t_; } while (0)
#define K3_STAMP_DUMP do { if (threadIdx.x == 0 && blockIdx.x == 0) for (int i_ = 0; i_ < 10; ++i_) g_k3_stamps[i_] = acc_[i_]; \
    if (threadIdx.x == 0 && blockIdx.x < 64) g_k3_wall[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define K3_STAMP_INIT do { } while (0)
#define K3_STAMP(i) do { } while (0)
#define K3_STAMP_DUMP do { } while (0)
#endif

struct GroupArgs {
    const float *limbs;
    const int32_t *jf, *jt;
    int L, K, nkp, use_scale, sort_dim, mmax;
    double person_thre;
    float dist_max;
    float *poses;
    int32_t *counts, *status;
    float *gsub;  // global subset tables (N x mmax x nkp x 6) or nullptr -> LDS
    float *glim;  // global staged candidate rows (N x L x K x 11) or nullptr -> LDS
};

__device__ __forceinline__ float np_sum17(const float *v, int n)  // numpy pairwise add.reduce, n <= 17
{
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r += v[i];
        return r;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += v[i + j];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += v[i];
    return res;
}

// GSUB is a template flag, not a runtime select: a pointer that may be LDS *or* global is a generic
// pointer, and every access through it becomes a flat_load (several times the latency of ds_read,
// and it ties LDS traffic to vmcnt).  With static address spaces the table is plain ds_*.
//
// Phases.  The per-limb-type candidate preparation (validity filter, stable rank sort, to-index
// de-duplication, staging of the surviving rows) does not depend on the partial-skeleton table, so
// it is done UP FRONT for all limb types at once, spread over (limb, k, j) cells -- 4 barriers in
// total instead of 7 per limb type.  The serial loop then only matches / applies / merges / appends:
// 5 barriers per limb type (3 when the table is still empty).
constexpr int kIdPitch = 20;

template <bool GSUB, bool GLIM>
__global__ void __launch_bounds__(kThreads)
greedy_group_kernel(GroupArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // any-flags and the per-column counters exist twice, used alternately by successive limb types (`par`): a wave that
    // is ahead resets the set of the NEXT limb type while laggards may still be reading the current one
    __shared__ int s_anyA[2], s_anyB[2], s_anyQ[2], s_M, s_kept;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = A.K, L = A.L, nkp = A.nkp, rowf = nkp * 6, mmax = A.mmax, LK = L * K;
    // ---- LDS carve-up (pure pointer arithmetic: an integer round trip for alignment would turn
    // everything carved after it into generic pointers, i.e. flat_load instead of ds_read) ----
    // keypoint ids of every physical row as ints, 20 per row (17 used): the merge search compares two rows with
    // 2 x 5 ds_read_b128 instead of 2 x 17 scattered ds_read_b32 of field 5 of the (row, joint, 6) table
    int *ids = reinterpret_cast<int *>(lds);                   // 16-byte items first: the base is 16-byte aligned
    double *r_score = reinterpret_cast<double *>(ids + (size_t)mmax * kIdPitch);
    float *p = reinterpret_cast<float *>(r_score + mmax);
    const bool wide_ok = (K & 3) == 0 && (mmax & 1) == 0;  // t_score/t_i2/t_dup rows on 16-byte boundaries
    const int LK4 = (LK + 3) & ~3;                         // array lengths rounded to 16 bytes: t_score .. t_dup are read wide
    float *lim_lds = p;          p += GLIM ? 0 : (size_t)((LK * 11 + 3) & ~3);  // per limb type: unique rows x1,y1,v1,s1,x2,y2,v2,s2,score,i1,i2
    float *lim_glb = A.glim + (size_t)img * LK * 11;
#define lim_all (GLIM ? lim_glb : lim_lds)
    float *t_score = p;          p += LK4;                 // up-front pass: score of valid candidates (-inf otherwise)
    int *t_i2 = (int *)p;        p += LK4;                 //                to-index
    int *t_dup = (int *)p;       p += LK4;                 //                a better-ranked row has the same to-index
    int *t_urank = (int *)p;     p += LK;                  //                rank among the valid, non-duplicate ones
    int *kk_arr = (int *)p;      p += L;                   // unique rows per limb type
    int *c_n1_2 = (int *)p;      p += 2 * K;               // rows with ms==1 per column, two alternating sets
    int *c_n2_2 = (int *)p;      p += 2 * K;               // rows with ms==2 per column, two alternating sets
    int *c_newcol = (int *)p;    p += K;                   // column of the i-th new row
    int *order = (int *)p;       p += mmax;                // logical position -> physical row
    int *r_a = (int *)p;         p += mmax;                // phase-A column per logical row
    int *r_b = (int *)p;         p += mmax;                // phase-B last column per logical row
    int *r_p = (int *)p;         p += mmax;                // last merge partner
    int *r_d = (int *)p;         p += mmax;                // deleted by a merge / dropped by the final threshold
    float *s_vals = p;           p += kScoreThreads * 17;   // per-thread scratch for the final mean
    float *sub_lds = p;
    float *sub_glb = A.gsub + (size_t)img * mmax * rowf;
#define sub (GSUB ? sub_glb : sub_lds)
#define SUBP(ph, j, f) sub[((ph) * nkp + (j)) * 6 + (f)]
#define LIM(c, f) lim_l[(c) * 11 + (f)]

    // (staging the image's 31 KB limbs block in LDS with one coalesced pass was measured: no gain, U1 and U4 below take 2 us each)
    const float *limbs = A.limbs + (size_t)img * LK * 13;
    K3_STAMP_INIT;
    // ================= up-front: every limb type at once =================
    // U1. validity filter (:64-76)
    for (int i = tid; i < LK; i += kThreads) {
        const float *c = limbs + (size_t)i * 13;
        const float lim = A.use_scale ? fmaxf(A.dist_max, c[12]) : A.dist_max;
        const bool valid = c[8] < lim && c[0] > 0.f && c[4] > 0.f && c[3] > 0.f && c[1] > 0.f;
        t_score[i] = valid ? c[10] : -INFINITY;
        t_i2[i] = (int)c[7];
        t_dup[i] = 0; t_urank[i] = 0;
    }
    for (int i = tid; i < L; i += kThreads) kk_arr[i] = 0;
    for (int i = tid; i < 2 * K; i += kThreads) { c_n1_2[i] = 0; c_n2_2[i] = 0; }
    if (tid == 0) { s_M = 0; s_anyA[0] = s_anyA[1] = 0; s_anyB[0] = s_anyB[1] = 0; s_anyQ[0] = s_anyQ[1] = 0; }
    __syncthreads();
    K3_STAMP(7);
    // U2. stable descending rank (:232) and "a better row owns my to-index" (:233-239).  One thread per
    // candidate, serial over the K rivals of its limb type (runtime integer divisions per (k, j) cell
    // cost more than the loop)
    // (K == 32, every published configuration: the eight steps are unrolled so that all sixteen wide LDS reads are in flight
    // together -- the rolled loop waits for LDS once per step, 9 us for this pass instead of 3)
    auto rank_pass = [&](auto steps_tag) {
        constexpr int STEPS = decltype(steps_tag)::value;   // K / 4 when K is known at compile time, 0 = generic
        typedef float v4f __attribute__((ext_vector_type(4)));
        typedef int v4i __attribute__((ext_vector_type(4)));
        for (int lk = tid; lk < LK; lk += kThreads) {
            const float s = t_score[lk];
            if (s == -INFINITY) continue;
            const int l = STEPS ? lk / (STEPS * 4) : lk / K, k = lk - l * K, base = l * K, my_i2 = t_i2[lk];
            int rank = 0, dup = 0;
            int j = 0;
            if constexpr (STEPS > 0) {
                v4f o4[STEPS];
                v4i i4[STEPS];
#pragma unroll
                for (int q = 0; q < STEPS; ++q) {
                    o4[q] = *reinterpret_cast<const v4f *>(t_score + base + 4 * q);
                    i4[q] = *reinterpret_cast<const v4i *>(t_i2 + base + 4 * q);
                }
#pragma unroll
                for (int q = 0; q < STEPS; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // bitwise, not short-circuit: `&&` / `||` compile to nested exec-mask branches here (20 instructions and
                        // three branches per rival).  A rival without a score holds -inf: never > s, never == s (s is finite)
                        const int better = (int)(o4[q][e] > s) | ((int)(o4[q][e] == s) & (int)(4 * q + e < k));
                        rank += better;
                        dup |= better & (int)(i4[q][e] == my_i2);
                    }
                j = K;
            } else if (wide_ok) {  // rows start on 16 bytes: four rivals per pair of ds_read_b128
                for (; j < K; j += 4) {
                    const v4f o4 = *reinterpret_cast<const v4f *>(t_score + base + j);
                    const v4i i4 = *reinterpret_cast<const v4i *>(t_i2 + base + j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int better = (int)(o4[e] > s) | ((int)(o4[e] == s) & (int)(j + e < k));
                        rank += better;
                        dup |= better & (int)(i4[e] == my_i2);
                    }
                }
            }
            for (; j < K; ++j) {
                const float o = t_score[base + j];
                const int better = (int)(o > s) | ((int)(o == s) & (int)(j < k));
                rank += better;
                dup |= better & (int)(t_i2[base + j] == my_i2);
            }
            t_dup[lk] = dup;
        }
    };
    if (K == 32 && wide_ok) rank_pass(std::integral_constant<int, 8>());
    else rank_pass(std::integral_constant<int, 0>());
    __syncthreads();
    K3_STAMP(8);
    // U3. position among the surviving rows
    auto urank_pass = [&](auto steps_tag) {
        constexpr int STEPS = decltype(steps_tag)::value;
        typedef float v4f __attribute__((ext_vector_type(4)));
        typedef int v4i __attribute__((ext_vector_type(4)));
        for (int lk = tid; lk < LK; lk += kThreads) {
            const float s = t_score[lk];
            if (s == -INFINITY || t_dup[lk]) continue;
            const int l = STEPS ? lk / (STEPS * 4) : lk / K, k = lk - l * K, base = l * K;
            int urank = 0;
            int j = 0;
            if constexpr (STEPS > 0) {
                v4f o4[STEPS];
                v4i d4[STEPS];
#pragma unroll
                for (int q = 0; q < STEPS; ++q) {
                    o4[q] = *reinterpret_cast<const v4f *>(t_score + base + 4 * q);
                    d4[q] = *reinterpret_cast<const v4i *>(t_dup + base + 4 * q);
                }
#pragma unroll
                for (int q = 0; q < STEPS; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        urank += (int)(d4[q][e] == 0) & ((int)(o4[q][e] > s) | ((int)(o4[q][e] == s) & (int)(4 * q + e < k)));
                j = K;
            } else if (wide_ok) {
                for (; j < K; j += 4) {
                    const v4f o4 = *reinterpret_cast<const v4f *>(t_score + base + j);
                    const v4i d4 = *reinterpret_cast<const v4i *>(t_dup + base + j);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        urank += (int)(d4[e] == 0) & ((int)(o4[e] > s) | ((int)(o4[e] == s) & (int)(j + e < k)));
                }
            }
            for (; j < K; ++j) {
                const float o = t_score[base + j];
                urank += (int)(t_dup[base + j] == 0) & ((int)(o > s) | ((int)(o == s) & (int)(j < k)));
            }
            t_urank[lk] = urank;
        }
    };
    if (K == 32 && wide_ok) urank_pass(std::integral_constant<int, 8>());
    else urank_pass(std::integral_constant<int, 0>());
    __syncthreads();
    K3_STAMP(9);
    // U4. stage the surviving rows in sorted order
    for (int i = tid; i < LK; i += kThreads) {
        if (t_score[i] == -INFINITY || t_dup[i]) continue;
        const int l = i / K;
        const float *r = limbs + (size_t)i * 13;
        float *d = lim_all + ((size_t)l * K + t_urank[i]) * 11;
        d[0] = r[0]; d[1] = r[1]; d[2] = r[2]; d[3] = r[11];
        d[4] = r[3]; d[5] = r[4]; d[6] = r[5]; d[7] = r[12];
        d[8] = r[10]; d[9] = r[6]; d[10] = r[7];
        atomicAdd(&kk_arr[l], 1);
    }
    __syncthreads();

    K3_STAMP(0);
    // ================= serial over limb types =================
    // invariant at the top: r_a = r_b = r_p = -1, r_d = 0 for rows < M; c_n1 = c_n2 = 0 and flags = 0 in the set `par`
    // M_rows (live rows) and P_rows (physical rows ever created) are kept by every wave in registers: they change by
    // wave-uniform arithmetic (new rows) or through s_M behind a barrier (merges)
    int M_rows = 0, P_rows = 0;
    bool overflow = false;
    int par = 1;
    for (int l = 0; l < L; ++l) {
        const int kk = kk_arr[l];
        if (kk == 0) continue;  // uniform
        par ^= 1;
        int *const c_n1 = c_n1_2 + par * K, *const c_n2 = c_n2_2 + par * K;
        const int jf = A.jf[l], jt = A.jt[l];
        const float *lim_l = lim_all + (size_t)l * K * 11;
        const int m0 = M_rows;
        bool anyA = false, anyB = false;
        if (m0 > 0) {
            // ---- match against the subset table (:87-135): lanes = (row, column) cells ----
            const int csh = 32 - __builtin_clz(max(kk - 1, 1) | 1);   // columns padded to 2^csh
            for (int cell = tid; cell < (m0 << csh); cell += kThreads) {
                const int m = cell >> csh, c = cell & ((1 << csh) - 1);
                if (c >= kk) continue;
                const int ph = order[m];
                const int idf = (int)SUBP(ph, jf, 5), idt = (int)SUBP(ph, jt, 5);
                const float lsf = SUBP(ph, jf, 4), lst = SUBP(ph, jt, 4), sc = LIM(c, 8);
                const int ms = (idf == (int)LIM(c, 9)) + (idt == (int)LIM(c, 10));
                const bool rep = (int)(sc > lst) | (int)(sc > lsf);
                if (ms == 2) { atomicAdd(&c_n2[c], 1); if (rep) { r_a[m] = c; s_anyA[par] = 1; } }
                if (ms == 1) { atomicAdd(&c_n1[c], 1); if (rep) { atomicMax(&r_b[m], c); s_anyB[par] = 1; } }
            }
            __syncthreads();
            K3_STAMP(1);
            for (int m = tid; m < m0; m += kThreads) {
                const int cA = r_a[m], cB = r_b[m];
                if (cA < 0 && cB < 0) continue;
                const int ph = order[m];
                float lsf = SUBP(ph, jf, 4), lst = SUBP(ph, jt, 4);
                if (cA >= 0) { lsf = fmaxf(LIM(cA, 8), lsf); lst = fmaxf(LIM(cA, 8), lst); }
                if (cB >= 0) {
                    SUBP(ph, jf, 5) = LIM(cB, 9); SUBP(ph, jt, 5) = LIM(cB, 10);
                    ids[ph * kIdPitch + jf] = (int)LIM(cB, 9); ids[ph * kIdPitch + jt] = (int)LIM(cB, 10);
#pragma unroll
                    for (int f = 0; f < 4; ++f) { SUBP(ph, jf, f) = LIM(cB, f); SUBP(ph, jt, f) = LIM(cB, 4 + f); }
                    lsf = fmaxf(LIM(cB, 8), lsf); lst = fmaxf(LIM(cB, 8), lst);
                }
                SUBP(ph, jf, 4) = lsf; SUBP(ph, jt, 4) = lst;
            }
            __syncthreads();
            K3_STAMP(2);
            anyA = s_anyA[par] != 0;
            anyB = s_anyB[par] != 0;
            // ---- merge rows sharing exactly two keypoints (:140-161): lanes = row pairs ----
            if (m0 >= 2) {
                // 16 x 16 tiles of (a, b) over the upper triangle: no index decoding, no divisions (one pair per lane in row-major
                // order of the triangle needs a third fewer rounds at 30 rows, but its index decoding costs more than that)
                constexpr int TA = kThreads >> 4;   // rows a per tile (16 with 256 threads), 16 rows b
                const int ta_n = (m0 + TA - 1) / TA, tb_n = (m0 + 15) >> 4;
                for (int ta = 0; ta < ta_n; ++ta)
                    for (int tb = (ta * TA) >> 4; tb < tb_n; ++tb) {
                        const int a = ta * TA + (tid >> 4), b = (tb << 4) + (tid & 15);
                        if (a < b && b < m0) {
                            const int pa = order[a], pb = order[b];
                            typedef int v4i __attribute__((ext_vector_type(4)));
                            const v4i *ra = reinterpret_cast<const v4i *>(ids + pa * kIdPitch);
                            const v4i *rb = reinterpret_cast<const v4i *>(ids + pb * kIdPitch);
                            int cnt = 0;
#pragma unroll
                            for (int q = 0; q < 5; ++q) {  // unused tail entries hold -1 in every row
                                const v4i va = ra[q], vb = rb[q];
#pragma unroll
                                for (int e = 0; e < 4; ++e) cnt += (int)(va[e] == vb[e]) & (int)(va[e] != -1);   // bitwise: no exec-mask branches
                            }
                            if (cnt == 2) { atomicMax(&r_p[a], b); r_d[b] = 1; s_anyQ[par] = 1; }
                        }
                    }
                __syncthreads();
                if (s_anyQ[par]) {
                    // a <- max(a, last partner); partners are deleted rows, which are never written
                    for (int a = tid >> 7; a < m0; a += kThreads >> 7) {   // 128 lanes per row (rowf <= 102)
                        const int f = tid & 127, b = r_p[a];
                        if (f < rowf && b >= 0 && !r_d[a]) {
                            const int ia = order[a] * rowf + f;
                            sub[ia] = fmaxf(sub[ia], sub[order[b] * rowf + f]);
                            if (f < nkp) {  // same maximum on the id table (ids are exact in fp32)
                                const int ja = order[a] * kIdPitch + f;
                                ids[ja] = max(ids[ja], ids[order[b] * kIdPitch + f]);
                            }
                        }
                    }
                    __syncthreads();
                    if (wave == 0) {  // np.delete keeps order: compact order[]
                        int newM = 0;
                        for (int mb = 0; mb < m0; mb += 64) {
                            const int m = mb + lane;
                            const bool live = m < m0 && !r_d[m];
                            const int ph = live ? order[m] : 0;
                            const uint64_t mask = __builtin_amdgcn_ballot_w64(live);
                            __builtin_amdgcn_wave_barrier();
                            if (live) order[newM + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = ph;
                            newM += __builtin_popcountll(mask);
                        }
                        if (lane == 0) s_M = newM;
                    }
                    __syncthreads();
                    M_rows = s_M;
                }
            }
        }
        K3_STAMP(3);
        // ---- unmatched limbs start new rows (:166-177).  EVERY wave works out the same slot assignment (ballot + popcount
        // over the kk columns; identical values stored by all four), so no barrier stands between it and the fill ----
        {
            int M = M_rows, P = P_rows, n_tot = 0;
            for (int c0 = 0; c0 < kk; c0 += 64) {
                const int c = c0 + lane;
                bool fresh = false;
                if (c < kk) fresh = (c_n2[c] * (anyA ? -1 : 2) + c_n1[c] * (anyB ? -1 : 1)) == 0;
                const uint64_t mask = __builtin_amdgcn_ballot_w64(fresh);
                const int n_new = __builtin_popcountll(mask);
                if (P + n_new > mmax) { overflow = true; break; }
                const int off = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                if (fresh) {
                    order[M + off] = P + off;
                    c_newcol[n_tot + off] = c;
                }
                M += n_new;
                P += n_new;
                n_tot += n_new;
            }
            if (overflow) break;  // uniform: every wave computed the same
            const int n_new = n_tot, p_first = P_rows;
            M_rows = M;
            P_rows = P;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // this wave reads the c_newcol entries it has just written
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            K3_STAMP(4);
            for (int r = tid >> 7; r < n_new; r += kThreads >> 7) {        // 128 lanes per new row (rowf <= 102)
                const int jf6 = tid & 127;
                if (jf6 >= rowf) continue;
                const int j = jf6 / 6, f = jf6 - j * 6, c = c_newcol[r];
                float v = -1.f;
                if (j == jf) v = (f < 4) ? LIM(c, f) : (f == 4 ? LIM(c, 8) : LIM(c, 9));
                else if (j == jt) v = (f < 4) ? LIM(c, 4 + f) : (f == 4 ? LIM(c, 8) : LIM(c, 10));
                sub[(p_first + r) * rowf + jf6] = v;
                if (jf6 < kIdPitch) ids[(p_first + r) * kIdPitch + jf6] = jf6 == jf ? (int)LIM(c, 9) : (jf6 == jt ? (int)LIM(c, 10) : -1);
            }
            // re-establish the loop invariant for the next limb type.  r_*: last read before a barrier every wave has
            // passed.  Counters and flags: the OTHER set (dirty since the previous limb type, read by nobody since the
            // barrier that ended it); the set of this limb type is still being read above by waves that lag behind
            for (int m = tid; m < M; m += kThreads) { r_a[m] = -1; r_b[m] = -1; r_p[m] = -1; r_d[m] = 0; }
            for (int c = tid; c < K; c += kThreads) { c_n1_2[(par ^ 1) * K + c] = 0; c_n2_2[(par ^ 1) * K + c] = 0; }
            if (tid == 0) { s_anyA[par ^ 1] = 0; s_anyB[par ^ 1] = 0; s_anyQ[par ^ 1] = 0; }
        }
        __syncthreads();
        K3_STAMP(5);
    }

    if (overflow) {
        if (tid == 0) { A.status[img] = 1; A.counts[img] = 0; }
        return;
    }
    // ================= _delete_sort (:187-219) =================
    const int M = M_rows;
    if (tid == 0) s_kept = 0;
    __syncthreads();
    for (int m = tid; m < M && tid < kScoreThreads; m += kScoreThreads) {
        const int ph = order[m];
        float *v = s_vals + (size_t)tid * 17;
        int n = 0;
        for (int j = 0; j < nkp; ++j) {
            const float x = SUBP(ph, j, A.sort_dim);
            if (x > 0.f) v[n++] = x;
        }
        const double score = (double)np_sum17(v, n) / (double)n;  // 0/0 -> NaN -> kept, like the reference
        const bool keep = !(score < A.person_thre);
        r_score[m] = score;
        r_d[m] = keep ? 0 : 1;
        if (keep) atomicAdd(&s_kept, 1);
    }
    __syncthreads();
    float *out = A.poses + (size_t)img * mmax * rowf;
    for (int m = wave; m < M; m += kThreads / 64) {  // one wave per row: coalesced copy
        if (r_d[m]) continue;
        const double s = r_score[m];
        int rank = 0;  // rows ahead of m in the stable descending order: one lane per candidate row, ballot + popcount
        for (int j0 = 0; j0 < M; j0 += 64) {
            const int j = j0 + lane;
            bool ahead = false;
            if (j < M && !r_d[j]) {
                const double o = r_score[j];
                ahead = (int)(o > s) | ((int)(o == s) & (int)(j < m));
            }
            rank += __builtin_popcountll(__builtin_amdgcn_ballot_w64(ahead));
        }
        const float *src = sub + order[m] * rowf;
        for (int f = lane; f < rowf; f += 64) {
            const float v = src[f];
            out[(size_t)rank * rowf + f] = (v == -1.f) ? 0.f : v;
        }
    }
    K3_STAMP(6);
    K3_STAMP_DUMP;
    if (tid == 0) { A.counts[img] = s_kept; A.status[img] = 0; }
#undef SUBP
#undef LIM
#undef sub
#undef lim_all
}

constexpr size_t kLdsLimit = 159 * 1024;  // 160 KiB per CU minus the static __shared__ words

size_t staging_bytes(int L, int K, int mmax, bool glim)   // mirrors the kernel's carve-up
{
    const size_t lk = (size_t)L * K, lk4 = (lk + 3) & ~(size_t)3;
    return (size_t)mmax * kIdPitch * 4 + (size_t)mmax * 8 +
           ((glim ? 0 : ((lk * 11 + 3) & ~(size_t)3)) + 3 * lk4 + lk + L + (size_t)K * 5 + (size_t)mmax * 5 + kScoreThreads * 17) * 4 + 64;
}

size_t table_ws_bytes(int N, int n_kp, int mmax) { return og_align_up((size_t)N * mmax * n_kp * 6 * sizeof(float), 256); }

}  // namespace

OG_API size_t og_group_workspace_bytes(int N, int L, int k, int n_kp, int mmax)
{
    if (N <= 0 || L <= 0 || k <= 0 || n_kp <= 0 || mmax <= 0) return 0;
    return table_ws_bytes(N, n_kp, mmax) + og_align_up((size_t)N * L * k * 11 * sizeof(float), 256);
}

OG_API int og_greedy_group_f32(const float *limbs, int N, int L, int k, const int32_t *jf, const int32_t *jt,
                               int n_kp, double person_thre, float dist_max, int use_scale, int sort_dim, int mmax,
                               float *poses, int32_t *counts, int32_t *status, void *workspace,
                               size_t workspace_bytes, void *stream)
{
    const char *name = "og_greedy_group_f32";
    OG_REQUIRE(limbs && jf && jt && poses && counts && status, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && L > 0 && k > 0 && mmax > 0, OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE(n_kp > 0 && n_kp <= 17, OG_EUNSUPPORTED, "%s: n_kp=%d (max 17)", name, n_kp);
    OG_REQUIRE(sort_dim >= 0 && sort_dim < 6, OG_EINVAL, "%s: sort_dim", name);
    OG_REQUIRE(mmax <= 4096, OG_EUNSUPPORTED, "%s: mmax=%d too large", name, mmax);
    GroupArgs a;
    a.limbs = limbs; a.jf = jf; a.jt = jt; a.L = L; a.K = k; a.nkp = n_kp; a.use_scale = use_scale;
    a.sort_dim = sort_dim; a.mmax = mmax; a.person_thre = person_thre; a.dist_max = dist_max;
    a.poses = poses; a.counts = counts; a.status = status; a.gsub = nullptr; a.glim = nullptr;
    // What lives where: everything in LDS when it fits (the published configuration: 121 KB); else the table moves to the
    // global workspace (GSUB), then the staged candidate rows as well (GLIM).
    const size_t table = (size_t)mmax * n_kp * 6 * sizeof(float);
    bool gsub = false, glim = false;
    size_t lds = staging_bytes(L, k, mmax, false) + table;
    if (lds > kLdsLimit) { gsub = true; lds -= table; }
    if (lds > kLdsLimit) { glim = true; lds = staging_bytes(L, k, mmax, true); }
    OG_REQUIRE(lds <= kLdsLimit, OG_EUNSUPPORTED, "%s: L*k=%d candidates and mmax=%d need %zu bytes of LDS (limit %zu)", name,
               L * k, mmax, lds, kLdsLimit);
    if (gsub) {
        OG_REQUIRE(workspace && workspace_bytes >= og_group_workspace_bytes(N, L, k, n_kp, mmax), OG_ENOSPC,
                   "%s: L=%d k=%d mmax=%d needs a %zu-byte workspace", name, L, k, mmax, og_group_workspace_bytes(N, L, k, n_kp, mmax));
        a.gsub = (float *)workspace;
        if (glim) a.glim = (float *)((char *)workspace + table_ws_bytes(N, n_kp, mmax));
    }
    void (*kern)(GroupArgs) = glim ? greedy_group_kernel<true, true> : gsub ? greedy_group_kernel<true, false> : greedy_group_kernel<false, false>;
    static OgAttrOnce attr_set[3];
    if (attr_set[glim ? 2 : gsub ? 1 : 0].need()) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
        OG_REQUIRE(e == hipSuccess, OG_EHIP, "%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(N), dim3(kThreads), lds, (hipStream_t)stream, a);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

#ifdef OG_K3_STAMPS   // tuning harness (tools/k3_stamps.py): cycle counts per phase of image 0's workgroup
OG_API void og_k3_debug_stamps(void *host_out) { (void)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_k3_stamps), sizeof(g_k3_stamps)); }
OG_API void og_k3_wall_stamps(void *host_out) { (void)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_k3_wall), sizeof(g_k3_wall)); }
#endif
