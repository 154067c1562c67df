// Main-loop probe for a C1 design that was priced but not built: how fast can a CU run the 3x3 layer's inner
// loop when a wave owns MT x 16 pixels x 128 couts?  Timing only -- operands are random bytes, nothing is checked.
//   MT = 4, two 4-wave workgroups per CU   = today's kernel (64 px x 128 couts per wave, 256 px per workgroup)      [control]
//   MT = 8, ONE 4-wave workgroup per CU    = 128 px x 128 couts per wave on 512 registers, 512 px per workgroup:
//                                            weights LDS-DMA per MFMA halved, fragment reads per MFMA 12/32 -> 16/64
// Per step (one tap of one 32-channel chunk): counted vmcnt | barrier | LDS-DMA of a later weight stage (8 KiB per workgroup) and a
// share of the next halo chunk | the NEXT step's fragments read from LDS (8 x ds_read_b128 weights, MT x 2 x ds_read_b64 pixels)
// while this step's MT x 8 MFMAs run.  Build: hipcc -O3 --offload-arch=gfx950 -o tools/build/c1_wide_probe tools/experiments/c1_wide_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned long long u64;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, unsigned char *l)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)l, 16, voff, soff, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm()
{
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));   // vmcnt(N) lgkmcnt(0)
    asm volatile("" ::: "memory");
}

template <int MT, int WGS_PER_CU>
__global__ void __launch_bounds__(256, WGS_PER_CU)
probe(const unsigned short *w, const unsigned short *x, float *out, int steps, int w_bytes, int x_bytes)
{
    constexpr int NT = 8, kRing = 4, kB = 8192;
    constexpr int kHaloPx = (MT == 8 ? 34 * 18 : MT == 6 ? 26 * 18 : 18 * 18), kA = (kHaloPx * 80 + 1023) / 1024 * 1024;
    constexpr int kHaloPieces = (kA / 1024 + 3) / 4;          // 1-KiB wave blocks per thread and chunk
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char *const bufB = lds, *const bufA = lds + kRing * kB;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int fcol = lane & 15, fk = lane >> 4;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(w), 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(x), 0, x_bytes, 0x00020000);
    const uint32_t voff = (uint32_t)tid * 16;
    const uint32_t item = blockIdx.x * 65536u;

    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 wf[2][NT], pf[2][MT];

    auto issue = [&](int s) {       // weights of stage s (2 pieces per thread) + this step's share of the next halo chunk
        blds16(wr, voff, (item + (uint32_t)s * kB) % (uint32_t)(w_bytes - 2 * kB), bufB + (s % kRing) * kB + wave * 1024);
        blds16(wr, voff + 4096, (item + (uint32_t)s * kB) % (uint32_t)(w_bytes - 2 * kB), bufB + (s % kRing) * kB + 4096 + wave * 1024);
        const int tap = s % 9, q = s / 9;
        if (tap * 4 < kA / 1024) {
#pragma unroll
            for (int i = 0; i < (kHaloPieces + 8) / 9 + (MT >= 6 ? 1 : 0); ++i) {
                const int blk = (tap * ((kHaloPieces + 8) / 9 + (MT >= 6 ? 1 : 0)) + i) * 4 + wave;
                if (blk * 1024 < kA)
                    blds16(xr, voff & 1023, (item * 7 + (uint32_t)(q + 1) * kA + blk * 1024) % (uint32_t)(x_bytes - kA), bufA + ((q + 1) & 1) * kA + blk * 1024);
            }
        }
    };
    auto read_frags = [&](int s, int set) {
        const int tap = s % 9, q = s / 9;
        const unsigned char *wB = bufB + (s % kRing) * kB + fcol * 64 + ((fk ^ ((4 - ((fcol >> 2) & 3)) & 3)) << 4);
#pragma unroll
        for (int n = 0; n < NT; ++n) wf[set][n] = *reinterpret_cast<const f16x8 *>(wB + n * 1024);
        const int hw = MT == 8 ? 34 : MT == 6 ? 26 : 18;
        const unsigned char *pA = bufA + (q & 1) * kA + (((tap / 3) * hw + tap % 3) + (wave * 4) * hw + fcol) * 80 + fk * 8;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const unsigned char *p = pA + (m % 4) * hw * 80 + (m / 4) * 8 * 80;     // (timing only: any conflict-free spread of rows)
            u64 lo = *reinterpret_cast<const u64 *>(p), hi = *reinterpret_cast<const u64 *>(p + 32);
            typedef u64 u64x2 __attribute__((ext_vector_type(2)));
            u64x2 v = {lo, hi};
            pf[set][m] = __builtin_bit_cast(f16x8, v);
        }
    };

    issue(0);
    issue(1);
    issue(2);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0);
#pragma unroll 1
    for (int s0 = 0; s0 < steps; s0 += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int s = s0 + h;
            wait_vm<6>();                      // everything but the two youngest steps' pieces (approximately the product's depth)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            issue(s + 3);
            read_frags(s + 1, h ^ 1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    if constexpr (MT > 4)    // accumulators pinned to AGPRs, in place (left to hipcc, 192+ accumulator registers get rotated
                                             // through v_accvgpr moves: three per MFMA)
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[n][m]) : "v"(wf[h][n]), "v"(pf[h][m]));
                    else
                        acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[h][n], pf[h][m], acc[n][m], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    wait_vm<0>();
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) sum += acc[n][m];
    if (sum[0] + sum[1] + sum[2] + sum[3] == 12345.678f) out[blockIdx.x * 256 + tid] = sum[0];
}

template <int MT, int WPC>
static void run(const char *name, const unsigned short *w, const unsigned short *x, float *out, int w_bytes, int x_bytes, int wgs, int steps)
{
    constexpr int kHaloPx = (MT == 8 ? 34 * 18 : MT == 6 ? 26 * 18 : 18 * 18), kA = (kHaloPx * 80 + 1023) / 1024 * 1024;
    const int lds = 4 * 8192 + 2 * kA;
    CHECK(hipFuncSetAttribute((const void *)probe<MT, WPC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<MT, WPC>), dim3(wgs), dim3(256), lds, 0, w, x, out, steps, w_bytes, x_bytes);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f, tot = 0.f;
    const int reps = 5, launches = 30;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((probe<MT, WPC>), dim3(wgs), dim3(256), lds, 0, w, x, out, steps, w_bytes, x_bytes);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        tot += ms;
        best = ms < best ? ms : best;
    }
    const double flop = 2.0 * wgs * 4.0 * steps * 8 * MT * (16.0 * 16 * 32);
    printf("%-44s %5d workgroups x %d steps: %8.1f us per launch (best %8.1f) = %7.1f TFLOP/s (LDS %d KiB)\n", name, wgs, steps,
           tot / reps / launches * 1e3, best / launches * 1e3, flop / (tot / reps / launches * 1e-3) / 1e12, lds / 1024);
}

int main()
{
    const int w_bytes = 8 << 20, x_bytes = 256 << 20;
    unsigned short *w, *x;
    float *out;
    CHECK(hipMalloc(&w, w_bytes));
    CHECK(hipMalloc(&x, x_bytes));
    CHECK(hipMalloc(&out, 1 << 24));
    std::vector<unsigned short> h(x_bytes / 2);
    for (size_t i = 0; i < h.size(); ++i) {     // random fp16 in (-2, 2): sign, exponent 12..15, random mantissa
        const unsigned r = (unsigned)rand();
        h[i] = (unsigned short)(((r & 1) << 15) | ((12 + ((r >> 1) & 3)) << 10) | ((r >> 3) & 1023));
    }
    CHECK(hipMemcpy(x, h.data(), x_bytes, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(w, h.data(), w_bytes, hipMemcpyHostToDevice));
    // the 160x160 256->256 layer at batch 8: 1 600 workgroups of 72 steps today = 800 of the wide form
    run<4, 2>("MT=4, two workgroups per CU (today)", w, x, out, w_bytes, x_bytes, 1600, 72);
    run<8, 1>("MT=8, one workgroup per CU (512 registers)", w, x, out, w_bytes, x_bytes, 800, 72);
    run<6, 1>("MT=6, one workgroup per CU (96 px x 128 couts per wave)", w, x, out, w_bytes, x_bytes, 1066, 72);
    run<4, 2>("MT=4, 4 x the work per launch", w, x, out, w_bytes, x_bytes, 6400, 72);
    run<8, 1>("MT=8, 4 x the work per launch", w, x, out, w_bytes, x_bytes, 3200, 72);
    run<4, 2>("MT=4, long workgroups (no pro/epilogue share)", w, x, out, w_bytes, x_bytes, 512, 72 * 3);
    run<8, 1>("MT=8, long workgroups", w, x, out, w_bytes, x_bytes, 256, 72 * 3);
    run<6, 1>("MT=6, long workgroups", w, x, out, w_bytes, x_bytes, 256, 72 * 3);
    return 0;
}
