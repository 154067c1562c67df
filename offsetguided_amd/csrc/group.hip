// K3 -- GreedyGroup.group_skeletons (decoder/group.py:39-185; _delete_reconns :221-240;
// _delete_sort :187-219), device resident: one 4-wave workgroup per image.
//
// The reference ships the limbs to the host (.cpu().numpy(), decoder/factory.py:91) and runs
// numpy in a multiprocessing.Pool; here the partial-skeleton ("subset") table lives in LDS
// (mmax x n_kp x 6 fp32, rows addressed through an order[] indirection so deletions never move
// data) and the limb types are processed serially -- that dependence is inherent -- while
// each step is spread over 256 lanes:
//   * filter / stable rank-sort / to-index de-duplication of the K candidates: lanes = candidates,
//     compaction by ballot + popcount;
//   * matching: lanes = (subset row, limb column) cells; per-row results by LDS atomicMax
//     (phase B keeps the LAST matching column: numpy last-write-wins), per-column match counts
//     by LDS atomicAdd;
//   * merge search: lanes = (row a, row b) pairs, 17 index compares each, last partner by
//     LDS atomicMax.
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
#include <math.h>

#include "og_common.h"

namespace {

constexpr int kThreads = 256;

#ifdef OG_K3_STAMPS
__device__ unsigned long long g_k3_stamps[16];
#define K3_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); g_k3_stamps[i] += t_ - t_prev_; t_prev_ = t_; } } while (0)
#define K3_STAMP_INIT unsigned long long t_prev_ = __builtin_amdgcn_s_memtime()
#else
#define K3_STAMP(i) do { } while (0)
#define K3_STAMP_INIT do { } while (0)
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
    int limbs_in_lds;  // stage the image's whole limbs block (L*K*13 floats) in LDS up front
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

// GSUB / LLDS are template flags, not runtime selects: a pointer that may be LDS *or* global is a
// generic pointer, and every access through it becomes a flat_load (several times the latency of
// ds_read, and it ties LDS traffic to vmcnt).  With static address spaces the table is plain ds_*.
template <bool GSUB, bool LLDS>
__global__ void __launch_bounds__(kThreads)
greedy_group_kernel(GroupArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int s_nv, s_kk, s_anyA, s_anyB, s_anyQ, s_M, s_P, s_overflow, s_kept, s_new0, s_nnew;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = A.K, nkp = A.nkp, rowf = nkp * 6, mmax = A.mmax;
    // ---- LDS carve-up (pure pointer arithmetic: an integer round trip for alignment would turn
    // everything carved after it into generic pointers, i.e. flat_load instead of ds_read) ----
    double *r_score = reinterpret_cast<double *>(lds);        // 8-byte items first: the base is 16-byte aligned
    float *p = reinterpret_cast<float *>(r_score + mmax);
    float *c_lim = p;            p += (size_t)K * 11;      // unique limb rows: x1,y1,v1,s1,x2,y2,v2,s2,score,i1,i2
    float *c_score = p;          p += K;                   // per candidate: score (staging for the sort)
    int *c_i2 = (int *)p;        p += K;                   // per candidate: to-index
    int *c_ord = (int *)p;       p += K;                   // sorted valid candidates
    int *c_uq = (int *)p;        p += K;                   // sorted, de-duplicated
    int *c_n1 = (int *)p;        p += K;                   // rows with ms==1 per column
    int *c_n2 = (int *)p;        p += K;                   // rows with ms==2 per column
    int *c_srt = (int *)p;       p += K;                   // to-index in sorted order
    int *c_dup = (int *)p;       p += K;                   // sorted position repeats an earlier to-index
    int *order = (int *)p;       p += mmax;                // logical position -> physical row
    int *r_a = (int *)p;         p += mmax;                // phase-A column per logical row / merge partner
    int *r_b = (int *)p;         p += mmax;                // phase-B last column per logical row / deleted flag
    float *s_vals = p;           p += kThreads * 17;       // per-thread scratch for the final mean
    float *s_limbs = p;
    if (LLDS) p += (size_t)A.L * K * 13;
    float *sub_lds = p;
    float *sub_glb = A.gsub + (size_t)img * mmax * rowf;
#define sub (GSUB ? sub_glb : sub_lds)
#define SUBP(ph, j, f) sub[((size_t)(ph) * nkp + (j)) * 6 + (f)]
#define LIM(c, f) c_lim[(size_t)(c) * 11 + (f)]

    if (tid == 0) { s_M = 0; s_P = 0; s_overflow = 0; }
    __syncthreads();
    const float *limbs_glb = A.limbs + (size_t)img * A.L * K * 13;
    if (LLDS) {  // one coalesced pass instead of two dependent global reads per limb type
        const int n = A.L * K * 13;
        for (int i = tid; i < n; i += kThreads) s_limbs[i] = limbs_glb[i];
    }
#define limbs (LLDS ? (const float *)s_limbs : limbs_glb)
    __syncthreads();

    K3_STAMP_INIT;
    for (int l = 0; l < A.L; ++l) {
        const int jf = A.jf[l], jt = A.jt[l];
        const float *cn = limbs + (size_t)l * K * 13;
        // ---- 1. validity filter (:64-76) + staging ----
        if (tid == 0) { s_nv = 0; s_anyA = 0; s_anyB = 0; s_anyQ = 0; }
        __syncthreads();
        for (int k = tid; k < K; k += kThreads) {
            const float *c = cn + (size_t)k * 13;
            const float lim = A.use_scale ? fmaxf(A.dist_max, c[12]) : A.dist_max;
            const bool valid = c[8] < lim && c[0] > 0.f && c[4] > 0.f && c[3] > 0.f && c[1] > 0.f;
            c_score[k] = valid ? c[10] : -INFINITY;
            c_i2[k] = (int)c[7];
            if (valid) atomicAdd(&s_nv, 1);
        }
        __syncthreads();
        K3_STAMP(0);
        const int nv = s_nv;
        if (nv == 0) continue;  // uniform
        // ---- 2. stable sort by score descending (:232): rank by counting over (k, j) cells ----
        for (int k = tid; k < K; k += kThreads) c_ord[k] = 0;   // reused as the rank accumulator
        __syncthreads();
        for (int cell = tid; cell < K * K; cell += kThreads) {
            const int k = cell / K, j = cell - k * K;
            const float s = c_score[k], o = c_score[j];
            if (s != -INFINITY && o != -INFINITY && (o > s || (o == s && j < k))) atomicAdd(&c_ord[k], 1);
        }
        __syncthreads();
        int my_rank = -1;
        if (tid < K && c_score[tid] != -INFINITY) my_rank = c_ord[tid];
        __syncthreads();
        if (my_rank >= 0) { c_ord[my_rank] = tid; c_srt[my_rank] = c_i2[tid]; }
        __syncthreads();
        // ---- 3. keep the first occurrence of every to-index (:233-239): (p, q<p) cells, then wave-0 compaction ----
        for (int p0 = tid; p0 < nv; p0 += kThreads) c_dup[p0] = 0;
        __syncthreads();
        for (int cell = tid; cell < nv * nv; cell += kThreads) {
            const int pp = cell / nv, q = cell - pp * nv;
            if (q < pp && c_srt[q] == c_srt[pp]) c_dup[pp] = 1;
        }
        __syncthreads();
        if (wave == 0) {
            int kk = 0;
            for (int p0 = 0; p0 < nv; p0 += 64) {
                const int pp = p0 + lane;
                const bool keep = pp < nv && !c_dup[pp];
                const uint64_t mask = __builtin_amdgcn_ballot_w64(keep);
                if (keep) c_uq[kk + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = c_ord[pp];
                kk += __builtin_popcountll(mask);
            }
            if (lane == 0) s_kk = kk;
        }
        __syncthreads();
        K3_STAMP(2);
        const int kk = s_kk, m0 = s_M;
        // stage the unique limb rows, zero the column counters and the per-row results
        for (int c = tid; c < kk; c += kThreads) {
            const float *r = cn + (size_t)c_uq[c] * 13;
            LIM(c, 0) = r[0]; LIM(c, 1) = r[1]; LIM(c, 2) = r[2]; LIM(c, 3) = r[11];
            LIM(c, 4) = r[3]; LIM(c, 5) = r[4]; LIM(c, 6) = r[5]; LIM(c, 7) = r[12];
            LIM(c, 8) = r[10]; LIM(c, 9) = r[6]; LIM(c, 10) = r[7];
            c_n1[c] = 0;
            c_n2[c] = 0;
        }
        for (int m = tid; m < m0; m += kThreads) { r_a[m] = -1; r_b[m] = -1; }
        __syncthreads();
        K3_STAMP(3);
        // ---- 4. match against the subset table (:87-135): lanes = (row, column) cells ----
        for (int cell = tid; cell < m0 * kk; cell += kThreads) {
            const int m = cell / kk, c = cell - m * kk, ph = order[m];
            const int idf = (int)SUBP(ph, jf, 5), idt = (int)SUBP(ph, jt, 5);
            const float lsf = SUBP(ph, jf, 4), lst = SUBP(ph, jt, 4), sc = LIM(c, 8);
            const int ms = (idf == (int)LIM(c, 9)) + (idt == (int)LIM(c, 10));
            const bool rep = sc > lst || sc > lsf;
            if (ms == 2) { atomicAdd(&c_n2[c], 1); if (rep) { r_a[m] = c; s_anyA = 1; } }
            if (ms == 1) { atomicAdd(&c_n1[c], 1); if (rep) { atomicMax(&r_b[m], c); s_anyB = 1; } }
        }
        __syncthreads();
        for (int m = tid; m < m0; m += kThreads) {
            const int cA = r_a[m], cB = r_b[m];
            if (cA < 0 && cB < 0) continue;
            const int ph = order[m];
            float lsf = SUBP(ph, jf, 4), lst = SUBP(ph, jt, 4);
            if (cA >= 0) { lsf = fmaxf(LIM(cA, 8), lsf); lst = fmaxf(LIM(cA, 8), lst); }
            if (cB >= 0) {
                SUBP(ph, jf, 5) = LIM(cB, 9); SUBP(ph, jt, 5) = LIM(cB, 10);
#pragma unroll
                for (int f = 0; f < 4; ++f) { SUBP(ph, jf, f) = LIM(cB, f); SUBP(ph, jt, f) = LIM(cB, 4 + f); }
                lsf = fmaxf(LIM(cB, 8), lsf); lst = fmaxf(LIM(cB, 8), lst);
            }
            SUBP(ph, jf, 4) = lsf; SUBP(ph, jt, 4) = lst;
        }
        __syncthreads();
        K3_STAMP(4);
        const bool anyA = s_anyA != 0, anyB = s_anyB != 0;
        // ---- 5. merge rows sharing exactly two keypoints (:140-161): lanes = row pairs ----
        if (m0 >= 2) {
            for (int m = tid; m < m0; m += kThreads) { r_a[m] = -1; r_b[m] = 0; }  // partner / deleted
            __syncthreads();
            const int npairs = m0 * (m0 - 1) / 2;
            for (int pi = tid; pi < npairs; pi += kThreads) {
                // pair index -> (a < b), row-major over the strict upper triangle
                int a = (int)((2.f * m0 - 1.f - sqrtf((2.f * m0 - 1.f) * (2.f * m0 - 1.f) - 8.f * pi)) * 0.5f);
                while (a > 0 && a * (2 * m0 - a - 1) / 2 > pi) --a;
                while ((a + 1) * (2 * m0 - a - 2) / 2 <= pi) ++a;
                const int b = a + 1 + (pi - a * (2 * m0 - a - 1) / 2);
                const int pa = order[a], pb = order[b];
                int cnt = 0;
#pragma unroll
                for (int j = 0; j < 17; ++j)  // all 34 LDS reads issue back to back
                    if (j < nkp) {
                        const int ia = (int)SUBP(pa, j, 5), ib = (int)SUBP(pb, j, 5);
                        cnt += (ia == ib && ia != -1);
                    }
                if (cnt == 2) { atomicMax(&r_a[a], b); r_b[b] = 1; s_anyQ = 1; }
            }
            __syncthreads();
            if (s_anyQ) {
                // a <- max(a, last partner); partners are deleted rows, which are never written
                for (int e = tid; e < m0 * rowf; e += kThreads) {
                    const int a = e / rowf, f = e - a * rowf, b = r_a[a];
                    if (b >= 0 && !r_b[a]) {
                        const size_t ia = (size_t)order[a] * rowf + f;
                        sub[ia] = fmaxf(sub[ia], sub[(size_t)order[b] * rowf + f]);
                    }
                }
                __syncthreads();
                if (wave == 0) {  // np.delete keeps order: compact order[]
                    int newM = 0;
                    for (int mb = 0; mb < m0; mb += 64) {
                        const int m = mb + lane;
                        const bool live = m < m0 && !r_b[m];
                        const int ph = live ? order[m] : 0;
                        const uint64_t mask = __builtin_amdgcn_ballot_w64(live);
                        __builtin_amdgcn_wave_barrier();
                        if (live) order[newM + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = ph;
                        newM += __builtin_popcountll(mask);
                    }
                    if (lane == 0) s_M = newM;
                }
                __syncthreads();
            }
        }
        K3_STAMP(5);
        // ---- 6. unmatched limbs start new rows (:166-177): wave 0 assigns slots, everyone fills ----
        if (wave == 0) {
            int M = s_M, P = s_P, n_tot = 0;
            for (int c0 = 0; c0 < kk; c0 += 64) {
                const int c = c0 + lane;
                bool fresh = false;
                if (c < kk) fresh = (c_n2[c] * (anyA ? -1 : 2) + c_n1[c] * (anyB ? -1 : 1)) == 0;
                const uint64_t mask = __builtin_amdgcn_ballot_w64(fresh);
                const int n_new = __builtin_popcountll(mask);
                if (P + n_new > mmax) { if (lane == 0) s_overflow = 1; break; }
                const int off = __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                if (fresh) {
                    order[M + off] = P + off;
                    c_srt[n_tot + off] = c;      // column of the (n_tot+off)-th new row
                }
                M += n_new;
                P += n_new;
                n_tot += n_new;
            }
            if (lane == 0) { s_new0 = s_P; s_nnew = n_tot; s_M = M; s_P = P; }
        }
        __syncthreads();
        {
            const int n_new = s_overflow ? 0 : s_nnew, p_first = s_new0;
            for (int e = tid; e < n_new * rowf; e += kThreads) {
                const int r = e / rowf, jf6 = e - r * rowf, j = jf6 / 6, f = jf6 - j * 6, c = c_srt[r];
                float v = -1.f;
                if (j == jf) v = (f < 4) ? LIM(c, f) : (f == 4 ? LIM(c, 8) : LIM(c, 9));
                else if (j == jt) v = (f < 4) ? LIM(c, 4 + f) : (f == 4 ? LIM(c, 8) : LIM(c, 10));
                sub[(size_t)(p_first + r) * rowf + e - r * rowf] = v;
            }
        }
        __syncthreads();
        K3_STAMP(6);
        if (s_overflow) break;  // uniform
    }

    if (s_overflow) {
        if (tid == 0) { A.status[img] = 1; A.counts[img] = 0; }
        return;
    }
    // ---- _delete_sort (:187-219) ----
    const int M = s_M;
    if (tid == 0) s_kept = 0;
    __syncthreads();
    for (int m = tid; m < M; m += kThreads) {
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
        r_b[m] = keep ? 0 : 1;
        if (keep) atomicAdd(&s_kept, 1);
    }
    __syncthreads();
    float *out = A.poses + (size_t)img * mmax * rowf;
    for (int m = wave; m < M; m += kThreads / 64) {  // one wave per row: coalesced copy
        if (r_b[m]) continue;
        const double s = r_score[m];
        int rank = 0;
        for (int j = 0; j < M; ++j) {
            const double o = r_score[j];
            rank += (!r_b[j]) && (o > s || (o == s && j < m));
        }
        const float *src = sub + (size_t)order[m] * rowf;
        for (int f = lane; f < rowf; f += 64) {
            const float v = src[f];
            out[(size_t)rank * rowf + f] = (v == -1.f) ? 0.f : v;
        }
    }
    K3_STAMP(7);
    if (tid == 0) { A.counts[img] = s_kept; A.status[img] = 0; }
#undef SUBP
#undef LIM
#undef sub
#undef limbs
}

constexpr size_t kLdsLimit = 159 * 1024;  // 160 KiB per CU minus the static __shared__ words

size_t staging_bytes(int K, int mmax)
{
    return ((size_t)K * 11 + (size_t)K * 8 + (size_t)mmax * 3 + kThreads * 17) * 4 + 8 + (size_t)mmax * 8;
}

}  // namespace

OG_API size_t og_group_workspace_bytes(int N, int n_kp, int mmax)
{
    if (N <= 0 || n_kp <= 0 || mmax <= 0) return 0;
    return og_align_up((size_t)N * mmax * n_kp * 6 * sizeof(float), 256);
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
    OG_REQUIRE(k <= kThreads, OG_EUNSUPPORTED, "%s: k=%d too large (max %d)", name, k, kThreads);
    OG_REQUIRE(mmax <= 4096, OG_EUNSUPPORTED, "%s: mmax=%d too large", name, mmax);
    GroupArgs a;
    a.limbs = limbs; a.jf = jf; a.jt = jt; a.L = L; a.K = k; a.nkp = n_kp; a.use_scale = use_scale;
    a.sort_dim = sort_dim; a.mmax = mmax; a.person_thre = person_thre; a.dist_max = dist_max;
    a.poses = poses; a.counts = counts; a.status = status; a.gsub = nullptr;
    size_t lds = staging_bytes(k, mmax);
    const size_t table = (size_t)mmax * n_kp * 6 * sizeof(float);
    const size_t limb_bytes = (size_t)L * k * 13 * sizeof(float);
    a.limbs_in_lds = (lds + limb_bytes + table <= kLdsLimit) || (lds + limb_bytes <= kLdsLimit / 2);
    if (a.limbs_in_lds) lds += limb_bytes;
    if (lds + table <= kLdsLimit) {
        lds += table;
    } else {
        OG_REQUIRE(workspace && workspace_bytes >= og_group_workspace_bytes(N, n_kp, mmax), OG_ENOSPC,
                   "%s: mmax=%d needs a %zu-byte workspace", name, mmax, og_group_workspace_bytes(N, n_kp, mmax));
        OG_REQUIRE(lds <= kLdsLimit, OG_EUNSUPPORTED, "%s: mmax=%d too large", name, mmax);
        a.gsub = (float *)workspace;
    }
    void (*kern)(GroupArgs) = a.gsub ? (a.limbs_in_lds ? greedy_group_kernel<true, true> : greedy_group_kernel<true, false>)
                                     : (a.limbs_in_lds ? greedy_group_kernel<false, true> : greedy_group_kernel<false, false>);
    static bool attr_set[4] = {false, false, false, false};
    const int variant = (a.gsub ? 2 : 0) + (a.limbs_in_lds ? 1 : 0);
    if (!attr_set[variant]) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
        OG_REQUIRE(e == hipSuccess, OG_EHIP, "%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
        attr_set[variant] = true;
    }
    hipLaunchKernelGGL(kern, dim3(N), dim3(kThreads), lds, (hipStream_t)stream, a);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}
