// K3 -- GreedyGroup.group_skeletons (decoder/group.py:39-185; _delete_reconns :221-240;
// _delete_sort :187-219), device resident: one single-wave workgroup per image.
//
// The reference ships the limbs to the host (.cpu().numpy(), decoder/factory.py:91) and runs
// numpy in a multiprocessing.Pool; here the partial-skeleton ("subset") table lives in LDS
// (mmax x n_kp x 6 fp32) and the 19 limb types are processed serially by one wavefront, with
// lanes mapped to limb candidates (filter / stable rank sort / de-duplication), then to
// subset rows (matching, in-place update, pairwise merge search), using ballot+popcount for
// every compaction and for the "did any pair fire" tests.
//
// numpy fancy-assignment semantics are kept exactly (see oracle/og_oracle.c for the literal
// loop form).  Per subset row they reduce to:
//   phase A: at most one limb column has both endpoints in the row (to-indices are unique);
//            if it is "better" both limb scores become max(score, old);
//   phase B: of all columns sharing exactly one endpoint (and better), only the LAST one
//            sticks (last write wins); its max() uses the scores after phase A;
//   matrix resets (all ms==2 -> -1 if any A pair fired, all ms==1 -> -1 if any B pair fired)
//            only matter through the column sums that decide which limbs start new rows;
//   merge:   row a takes max(row a, LAST row b sharing exactly two keypoints), computed from
//            the pre-merge state; every such b is deleted, order kept;
//   final:   mean of positive entries of column `sort_dim` (numpy pairwise fp32 sum, fp64
//            divide), fp64 threshold, stable descending order, -1 -> 0.
// If a table would exceed mmax rows the image is flagged in status[] (caller retries with a
// larger mmax; the table then lives in the caller's global workspace instead of LDS).
#include <math.h>

#include "og_common.h"

namespace {

struct GroupArgs {
    const float *limbs;
    const int32_t *jf, *jt;
    int L, K, nkp, use_scale, sort_dim, mmax;
    double person_thre;
    float dist_max;
    float *poses;
    int32_t *counts, *status;
    float *gsub;  // global subset tables (N x mmax x nkp x 6) or nullptr -> LDS
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

__global__ void __launch_bounds__(64)
greedy_group_kernel(GroupArgs A)
{
    extern __shared__ float lds[];
    const int img = blockIdx.x, lane = threadIdx.x;
    const int K = A.K, nkp = A.nkp, rowf = nkp * 6;
    // ---- LDS carve-up ----
    float *p = lds;
    float *c_lim = p;            p += (size_t)K * 11;      // unique limb rows: x1,y1,v1,s1,x2,y2,v2,s2,score,(i1),(i2)
    float *c_score = p;          p += K;                   // per candidate: score (staging for the sort)
    int *c_i2 = (int *)p;        p += K;                   // per candidate: to-index
    int *c_ord = (int *)p;       p += K;                   // sorted valid candidates
    int *c_uq = (int *)p;        p += K;                   // sorted, de-duplicated
    int *c_n1 = (int *)p;        p += K;                   // rows with ms==1 per column
    int *c_n2 = (int *)p;        p += K;                   // rows with ms==2 per column
    int *r_del = (int *)p;       p += A.mmax;              // row deleted by the merge
    int *r_pos = (int *)p;       p += A.mmax;              // row position after compaction / partner
    float *s_vals = p;           p += 64 * 17;             // per-lane scratch for the final mean (nkp <= 17 -> stride 17)
    double *r_score = (double *)(((uintptr_t)p + 7) & ~(uintptr_t)7);
    p = (float *)(r_score + A.mmax);
    float *sub = A.gsub ? A.gsub + (size_t)img * A.mmax * rowf : p;
#define SUB(m, j, f) sub[((size_t)(m) * nkp + (j)) * 6 + (f)]
#define LIM(c, f) c_lim[(size_t)(c) * 11 + (f)]

    int M = 0;
    bool overflow = false;
    const float *limbs = A.limbs + (size_t)img * A.L * K * 13;

    for (int l = 0; l < A.L && !overflow; ++l) {
        const int jf = A.jf[l], jt = A.jt[l];
        const float *cn = limbs + (size_t)l * K * 13;
        // ---- 1. validity filter (:64-76) + staging ----
        int nv = 0;
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int k = k0 + lane;
            bool valid = false;
            if (k < K) {
                const float *c = cn + (size_t)k * 13;
                const float lim = A.use_scale ? fmaxf(A.dist_max, c[12]) : A.dist_max;
                valid = c[8] < lim && c[0] > 0.f && c[4] > 0.f && c[3] > 0.f && c[1] > 0.f;
                c_score[k] = valid ? c[10] : -INFINITY;
                c_i2[k] = (int)c[7];
            }
            nv += __builtin_popcountll(__builtin_amdgcn_ballot_w64(valid));
        }
        __syncthreads();
        if (nv == 0) continue;
        // ---- 2. stable sort by score descending (:232), rank by counting ----
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int k = k0 + lane;
            if (k < K && c_score[k] != -INFINITY) {
                const float s = c_score[k];
                int rank = 0;
                for (int j = 0; j < K; ++j) {
                    const float o = c_score[j];
                    rank += (o != -INFINITY) && (o > s || (o == s && j < k));
                }
                c_ord[rank] = k;
            }
        }
        __syncthreads();
        // ---- 3. keep the first occurrence of every to-index (:233-239) ----
        int kk = 0;
        for (int p0 = 0; p0 < nv; p0 += 64) {
            const int pp = p0 + lane;
            bool keep = false;
            if (pp < nv) {
                const int t = c_i2[c_ord[pp]];
                keep = true;
                for (int q = 0; q < pp; ++q) keep &= (c_i2[c_ord[q]] != t);
            }
            const uint64_t mask = __builtin_amdgcn_ballot_w64(keep);
            if (keep) c_uq[kk + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = c_ord[pp];
            kk += __builtin_popcountll(mask);
        }
        __syncthreads();
        // stage the unique limb rows, zero the column counters
        for (int c0 = 0; c0 < kk; c0 += 64) {
            const int c = c0 + lane;
            if (c < kk) {
                const float *r = cn + (size_t)c_uq[c] * 13;
                LIM(c, 0) = r[0]; LIM(c, 1) = r[1]; LIM(c, 2) = r[2]; LIM(c, 3) = r[11];
                LIM(c, 4) = r[3]; LIM(c, 5) = r[4]; LIM(c, 6) = r[5]; LIM(c, 7) = r[12];
                LIM(c, 8) = r[10]; LIM(c, 9) = r[6]; LIM(c, 10) = r[7];
                c_n1[c] = 0;
                c_n2[c] = 0;
            }
        }
        __syncthreads();
        const int m0 = M;
        // ---- 4. match against the subset table (:87-135) ----
        bool anyA = false, anyB = false;
        for (int mb = 0; mb < m0; mb += 64) {
            const int m = mb + lane;
            const bool act = m < m0;
            int idf = -2, idt = -2;
            float lsf = 0.f, lst = 0.f;
            if (act) {
                idf = (int)SUB(m, jf, 5); idt = (int)SUB(m, jt, 5);
                lsf = SUB(m, jf, 4); lst = SUB(m, jt, 4);
            }
            int cA = -1, cB = -1;
            for (int c = 0; c < kk; ++c) {
                const int i1 = (int)LIM(c, 9), i2 = (int)LIM(c, 10);
                const float sc = LIM(c, 8);
                const int ms = act ? ((idf == i1) + (idt == i2)) : 0;
                const bool rep = sc > lst || sc > lsf;
                if (ms == 2 && rep) cA = c;
                if (ms == 1 && rep) cB = c;
                const int n1 = __builtin_popcountll(__builtin_amdgcn_ballot_w64(ms == 1));
                const int n2 = __builtin_popcountll(__builtin_amdgcn_ballot_w64(ms == 2));
                if (lane == 0) { c_n1[c] += n1; c_n2[c] += n2; }
            }
            anyA |= __builtin_amdgcn_ballot_w64(cA >= 0) != 0ull;
            anyB |= __builtin_amdgcn_ballot_w64(cB >= 0) != 0ull;
            if (act) {
                if (cA >= 0) { lsf = fmaxf(LIM(cA, 8), lsf); lst = fmaxf(LIM(cA, 8), lst); }
                if (cB >= 0) {
                    SUB(m, jf, 5) = LIM(cB, 9); SUB(m, jt, 5) = LIM(cB, 10);
#pragma unroll
                    for (int f = 0; f < 4; ++f) { SUB(m, jf, f) = LIM(cB, f); SUB(m, jt, f) = LIM(cB, 4 + f); }
                    lsf = fmaxf(LIM(cB, 8), lsf); lst = fmaxf(LIM(cB, 8), lst);
                }
                if (cA >= 0 || cB >= 0) { SUB(m, jf, 4) = lsf; SUB(m, jt, 4) = lst; }
            }
        }
        __syncthreads();
        // ---- 5. merge rows sharing exactly two keypoints (:140-161) ----
        if (m0 >= 2) {
            bool anyQ = false;
            for (int mb = 0; mb < m0; mb += 64) { const int m = mb + lane; if (m < m0) { r_del[m] = 0; r_pos[m] = -1; } }
            __syncthreads();
            for (int ab = 0; ab < m0; ab += 64) {
                const int a = ab + lane;
                int ida[17];
#pragma unroll
                for (int j = 0; j < 17; ++j) ida[j] = (a < m0 && j < nkp) ? (int)SUB(a, j, 5) : -1;
                int partner = -1;
                for (int b = ab + 1; b < m0; ++b) {
                    int cnt = 0;
#pragma unroll
                    for (int j = 0; j < 17; ++j)
                        if (j < nkp) { const int ib = (int)SUB(b, j, 5); cnt += (ida[j] == ib && ida[j] != -1); }
                    if (a < b && a < m0 && cnt == 2) { partner = b; r_del[b] = 1; }
                }
                if (a < m0) r_pos[a] = partner;
                anyQ |= __builtin_amdgcn_ballot_w64(partner >= 0) != 0ull;
            }
            __syncthreads();
            if (anyQ) {
                // a <- max(a, last partner), field by field, reads before writes, chunks ascending
                for (int ab = 0; ab < m0; ab += 64) {
                    const int a = ab + lane;
                    const int b = (a < m0) ? r_pos[a] : -1;
                    for (int f = 0; f < rowf; ++f) {
                        float v = 0.f;
                        if (b >= 0) v = fmaxf(sub[(size_t)a * rowf + f], sub[(size_t)b * rowf + f]);
                        __syncthreads();
                        if (b >= 0) sub[(size_t)a * rowf + f] = v;
                    }
                    __syncthreads();
                }
                // compaction (np.delete keeps order)
                int newM = 0;
                for (int mb = 0; mb < m0; mb += 64) {
                    const int m = mb + lane;
                    const bool live = m < m0 && !r_del[m];
                    const uint64_t mask = __builtin_amdgcn_ballot_w64(live);
                    const int dst = newM + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                    for (int f = 0; f < rowf; ++f) {
                        float v = 0.f;
                        if (live) v = sub[(size_t)m * rowf + f];
                        __syncthreads();
                        if (live) sub[(size_t)dst * rowf + f] = v;
                    }
                    __syncthreads();
                    newM += __builtin_popcountll(mask);
                }
                M = newM;
            }
        }
        // ---- 6. unmatched limbs start new rows (:166-177) ----
        for (int c0 = 0; c0 < kk; c0 += 64) {
            const int c = c0 + lane;
            bool fresh = false;
            if (c < kk) fresh = (c_n2[c] * (anyA ? -1 : 2) + c_n1[c] * (anyB ? -1 : 1)) == 0;
            const uint64_t mask = __builtin_amdgcn_ballot_w64(fresh);
            const int n_new = __builtin_popcountll(mask);
            if (M + n_new > A.mmax) { overflow = true; break; }
            if (fresh) {
                float *r = sub + (size_t)(M + __builtin_popcountll(mask & ((1ull << lane) - 1ull))) * rowf;
                for (int f = 0; f < rowf; ++f) r[f] = -1.f;
                r[jf * 6 + 5] = LIM(c, 9); r[jt * 6 + 5] = LIM(c, 10);
#pragma unroll
                for (int f = 0; f < 4; ++f) { r[jf * 6 + f] = LIM(c, f); r[jt * 6 + f] = LIM(c, 4 + f); }
                r[jf * 6 + 4] = LIM(c, 8); r[jt * 6 + 4] = LIM(c, 8);
            }
            M += n_new;
        }
        __syncthreads();
    }

    if (overflow) {
        if (lane == 0) { A.status[img] = 1; A.counts[img] = 0; }
        return;
    }
    // ---- _delete_sort (:187-219) ----
    int kept = 0;
    for (int mb = 0; mb < M; mb += 64) {
        const int m = mb + lane;
        bool keep = false;
        if (m < M) {
            float *v = s_vals + lane * 17;
            int n = 0;
            for (int j = 0; j < nkp; ++j) {
                const float x = SUB(m, j, A.sort_dim);
                if (x > 0.f) v[n++] = x;
            }
            const double score = (double)np_sum17(v, n) / (double)n;  // 0/0 -> NaN -> kept, like the reference
            keep = !(score < A.person_thre);
            r_score[m] = score;
            r_del[m] = keep ? 0 : 1;
        }
        kept += __builtin_popcountll(__builtin_amdgcn_ballot_w64(keep));
    }
    __syncthreads();
    float *out = A.poses + (size_t)img * A.mmax * rowf;
    for (int mb = 0; mb < M; mb += 64) {
        const int m = mb + lane;
        if (m < M && !r_del[m]) {
            const double s = r_score[m];
            int rank = 0;
            for (int j = 0; j < M; ++j) {
                const double o = r_score[j];
                rank += (!r_del[j]) && (o > s || (o == s && j < m));
            }
            for (int f = 0; f < rowf; ++f) {
                const float v = sub[(size_t)m * rowf + f];
                out[(size_t)rank * rowf + f] = (v == -1.f) ? 0.f : v;
            }
        }
    }
    if (lane == 0) { A.counts[img] = kept; A.status[img] = 0; }
#undef SUB
#undef LIM
}

constexpr size_t kLdsLimit = 160 * 1024;

size_t staging_bytes(int K, int mmax)
{
    return ((size_t)K * 11 + (size_t)K * 6 + (size_t)mmax * 2 + 64 * 17) * 4 + 8 + (size_t)mmax * 8;
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
    OG_REQUIRE(k <= 1024, OG_EUNSUPPORTED, "%s: k=%d too large", name, k);
    GroupArgs a;
    a.limbs = limbs; a.jf = jf; a.jt = jt; a.L = L; a.K = k; a.nkp = n_kp; a.use_scale = use_scale;
    a.sort_dim = sort_dim; a.mmax = mmax; a.person_thre = person_thre; a.dist_max = dist_max;
    a.poses = poses; a.counts = counts; a.status = status; a.gsub = nullptr;
    size_t lds = staging_bytes(k, mmax);
    const size_t table = (size_t)mmax * n_kp * 6 * sizeof(float);
    if (lds + table <= kLdsLimit) {
        lds += table;
    } else {
        OG_REQUIRE(workspace && workspace_bytes >= og_group_workspace_bytes(N, n_kp, mmax), OG_ENOSPC,
                   "%s: mmax=%d needs a %zu-byte workspace", name, mmax, og_group_workspace_bytes(N, n_kp, mmax));
        OG_REQUIRE(lds <= kLdsLimit, OG_EUNSUPPORTED, "%s: mmax=%d too large", name, mmax);
        a.gsub = (float *)workspace;
    }
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)greedy_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kLdsLimit);
        OG_REQUIRE(e == hipSuccess, OG_EHIP, "%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(greedy_group_kernel, dim3(N), dim3(64), lds, (hipStream_t)stream, a);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}
