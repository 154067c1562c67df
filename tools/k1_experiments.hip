// Tuning harness for K1 (not part of the product): includes the kernel TU directly so template
// parameters (prefetch depth) and launch geometry (rows per band) can be swept in one run.
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I include -I offsetguided_amd/csrc \
//         tools/k1_experiments.hip offsetguided_amd/csrc/abi.cpp -o /tmp/k1exp && /tmp/k1exp
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../offsetguided_amd/csrc/nms_topk.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float *p, size_t n, uint32_t seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = (float)(x >> 8) * (1.0f / 16777216.0f) - 0.3f;  // white noise: ~8% of pixels are positive peaks (worst case)
    }
}

// streaming ceiling of the same access pattern: loads + stencil maxima, one store per lane at the end
template <int PF>
__global__ void __launch_bounds__(64 * kMaxWaves)
stream_only_kernel(const float *__restrict__ in, float *__restrict__ sink, int H, int W, int rows, int nbands,
                   int panel_strips, int total, int padded)
{
    const int wid = og_xcd_remap(blockIdx.x, padded);
    if (wid >= total) return;
    const int plane = wid / nbands, band = wid % nbands;
    const TileGeom g = make_geom(H, W, rows, band, panel_strips, 4);
    float acc = 0.f;
    walk_panel<4, PF>(in + (size_t)plane * H * W, g, [&](int, const Px<4> &v, const Px<4> &ha, const Px<4> &hb, const Px<4> &hc) {
        const Px<4> m = vmax3<4>(ha, hb, hc);
        acc += (v.c[0] == m.c[0]) + (v.c[1] == m.c[1]) + (v.c[2] == m.c[2]) + (v.c[3] == m.c[3]);
    });
    if (acc == -1.f) sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// raw read ceilings: contiguous float4 per lane, UNROLL loads in flight per lane
template <int UNROLL, int AUX>
__global__ void __launch_bounds__(256)
read_sum_kernel(const float *__restrict__ in, float *__restrict__ sink, size_t n4)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, (int)0x7fffffff, 0x00020000);
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const float *p = in + (i + u * stride) * 4;
            if (AUX == 0) v[u] = *reinterpret_cast<const v4f *>(p);
            else v[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == -1.f) sink[threadIdx.x] = acc;
}

template <int UNROLL, int AUX>
float run_read(const std::vector<float *> &bufs, size_t n, void *ws, int iters, int blocks)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float sum = 0;
    for (int it = -2; it < iters; ++it) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((read_sum_kernel<UNROLL, AUX>), dim3(blocks), dim3(256), 0, 0, bufs[(it + 2) % bufs.size()], (float *)ws, n / 4);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it >= 0) sum += ms;
    }
    return sum / iters * 1e3f;
}

template <int PF>
float run_band(const std::vector<float *> &bufs, long planes, int H, int W, int k, int rows, void *ws, int iters, int mode)
{
    const int strips = W / 4, nwaves = (strips + kInterior - 1) / kInterior, panel = (strips + nwaves - 1) / nwaves;
    const int nbands = (H + rows - 1) / rows;
    const long total = planes * nbands;
    const int padded = (int)((total + 7) / 8 * 8);
    uint64_t *keys = (uint64_t *)ws;
    int *cnts = (int *)((char *)ws + og_align_up((size_t)planes * nbands * k * 8, 256));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9, sum = 0;
    for (int it = -2; it < iters; ++it) {
        const float *in = bufs[(it + 2) % bufs.size()];
        CK(hipEventRecord(a, 0));
        if (mode == 1)
            hipLaunchKernelGGL((stream_only_kernel<PF>), dim3(padded), dim3(64 * nwaves), 0, 0, in, (float *)ws, H, W, rows, nbands, panel, (int)total, padded);
        else if (mode == 3)
            hipLaunchKernelGGL((band_topk_kernel<4, true, PF, 2>), dim3(padded), dim3(64 * nwaves), (size_t)nwaves * 2 * 128 * 8 + 2048, 0, in, keys, cnts, (int *)nullptr, (const uint64_t *)keys, 1ull, H, W, k, 128, rows, nbands, panel, (int)total, padded, 0);
        else if (mode == 2)
            hipLaunchKernelGGL((band_topk_kernel<4, true, PF, 1>), dim3(padded), dim3(64 * nwaves), (size_t)nwaves * 2 * 128 * 8 + 2048, 0, in, keys, cnts, (int *)nullptr, (const uint64_t *)keys, 1ull, H, W, k, 128, rows, nbands, panel, (int)total, padded, 0);
        else
            hipLaunchKernelGGL((band_topk_kernel<4, true, PF>), dim3(padded), dim3(64 * nwaves), (size_t)nwaves * 2 * 128 * 8 + 2048, 0, in, keys, cnts, (int *)nullptr, (const uint64_t *)keys, 1ull, H, W, k, 128, rows, nbands, panel, (int)total, padded, 0);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it >= 0) { best = ms < best ? ms : best; sum += ms; }
    }
    return sum / iters * 1e3f;
}

int main(int argc, char **argv)
{
    const long planes = 136; const int H = 640, W = 640, k = 32;
    const size_t n = (size_t)planes * H * W;
    std::vector<float *> bufs(3);
    for (int i = 0; i < 3; ++i) { CK(hipMalloc(&bufs[i], n * 4)); hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, bufs[i], n, 17u * i + 1); }
    void *ws; CK(hipMalloc(&ws, 64 << 20));
    CK(hipDeviceSynchronize());
    if (argc > 1) {  // realistic hi-res heatmaps dumped by tools/kbench.py (OG_DUMP_HR)
        std::vector<float> h(n);
        FILE *f = fopen(argv[1], "rb");
        if (!f || fread(h.data(), 4, n, f) != n) { printf("cannot read %s\n", argv[1]); return 1; }
        fclose(f);
        for (int i = 0; i < 3; ++i) CK(hipMemcpy(bufs[i], h.data(), n * 4, hipMemcpyHostToDevice));
        printf("input: %s\n", argv[1]);
    }
    const double gb = n * 4 / 1e9;
    for (int blocks : {1024, 2048, 4096, 8192}) {
        float t1 = run_read<4, 0>(bufs, n, ws, 20, blocks), t2 = run_read<8, 0>(bufs, n, ws, 20, blocks), t3 = run_read<4, 1>(bufs, n, ws, 20, blocks);
        printf("read_sum blocks=%5d  unroll4 %.1f us %.2f TB/s | unroll8 %.1f us %.2f TB/s | unroll4 nt %.1f us %.2f TB/s\n", blocks,
               t1, gb / t1 * 1e3, t2, gb / t2 * 1e3, t3, gb / t3 * 1e3);
    }
    printf("%-12s %5s %3s %9s %8s\n", "kernel", "rows", "pf", "us", "TB/s");
    const char *names[] = {"band_topk", "stream_only", "masks_only", "ideal_tau"};
    const int rows_list[] = {32, 40, 80};
    for (int mode : {1, 2, 3, 0})
        for (int rows : rows_list) {
            float t4 = run_band<4>(bufs, planes, H, W, k, rows, ws, 20, mode);
            printf("%-12s %5d   4 %9.1f %8.2f\n", names[mode], rows, t4, gb / t4 * 1e3);
        }
    // full entry point (band + merge) at the library's default plan
    float *os; int64_t *oi; CK(hipMalloc(&os, planes * k * 4)); CK(hipMalloc(&oi, planes * k * 8));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(a, 0));
        int rc = og_nms_topk_f32(bufs[it % 3], planes, H, W, k, os, oi, ws, 64 << 20, nullptr);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("og_nms_topk_f32 rc=%d %.1f us\n", rc, ms * 1e3);
    }
    return 0;
}
