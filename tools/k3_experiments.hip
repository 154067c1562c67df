// Tuning harness for K3 (not part of the product): phase timing via s_memtime stamps.
// Reads limbs from a raw float file produced by tools/kbench.py --dump-limbs.
#define OG_K3_STAMPS
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../offsetguided_amd/csrc/group.hip"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main(int argc, char **argv)
{
    const int N = 8, L = 19, K = 32, nkp = 17, mmax = 128;
    std::vector<float> h((size_t)N * L * K * 13);
    FILE *f = fopen(argc > 1 ? argv[1] : "tools/build/limbs.bin", "rb");
    if (!f || fread(h.data(), 4, h.size(), f) != h.size()) { printf("cannot read limbs\n"); return 1; }
    fclose(f);
    const int jf_h[19] = {0, 0, 1, 1, 2, 5, 4, 3, 5, 7, 6, 8, 5, 6, 11, 11, 13, 12, 14};
    const int jt_h[19] = {1, 2, 2, 3, 4, 6, 6, 5, 7, 9, 8, 10, 11, 12, 12, 13, 15, 14, 16};
    float *limbs, *poses; int *jf, *jt, *meta;
    CK(hipMalloc(&limbs, h.size() * 4)); CK(hipMemcpy(limbs, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&jf, 76)); CK(hipMalloc(&jt, 76)); CK(hipMemcpy(jf, jf_h, 76, hipMemcpyHostToDevice)); CK(hipMemcpy(jt, jt_h, 76, hipMemcpyHostToDevice));
    CK(hipMalloc(&poses, (size_t)N * mmax * nkp * 6 * 4)); CK(hipMalloc(&meta, 64));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int it = 0; it < 5; ++it) {
        unsigned long long z[16] = {0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_k3_stamps), z, sizeof(z)));
        CK(hipEventRecord(a, 0));
        int rc = og_greedy_group_f32(limbs, N, L, K, jf, jt, nkp, 0.04, 40.f, 0, 2, mmax, poses, meta, meta + N, nullptr, 0, nullptr);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_k3_stamps), sizeof(z)));
        int cnt[8]; CK(hipMemcpy(cnt, meta, 32, hipMemcpyDeviceToHost));
        printf("rc=%d %.1f us  poses %d %d %d..  cycles: upfront %llu match %llu apply %llu pairs %llu slots %llu fill %llu final %llu\n",
               rc, ms * 1e3, cnt[0], cnt[1], cnt[2], z[0], z[1], z[2], z[3], z[4], z[5], z[6]);
    }
    return 0;
}
