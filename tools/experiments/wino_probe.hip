// Gate B of the Winograd F(2x2, 3x3) attempt: a MAIN-LOOP probe, timing only (random operands, nothing checked),
// in the manner of c1_wide_probe.hip, at the launch shape of the 8 x 160 x 160 256 -> 256 layer.
//
// The only form whose output transform needs no exchange between waves: a wave owns ALL 16 transform positions of its output tile in
// its accumulators -- 256 AGPRs = 16 positions x (32 tiles x 32 couts) -- so a workgroup is 4 waves on 512 registers (ONE workgroup
// per CU), 64 2x2-tiles (a 32 x 8 pixel region) x 64 couts.  K advances in 32-channel chunks; per chunk and wave:
//   T: 2 blocks x (16 ds_read_b128 of the raw 4x4 patches -> B^T d B in registers, 128 v_pk_add_f16) = the MFMA B operands of all
//      16 positions (128 VGPRs: no room for a second set, so T cannot run under the previous chunk's MFMAs in the same wave);
//   M: 4 position groups x (8 ds_read_b128 of the transformed weights U, 16 v_mfma_f32_16x16x32_f16), U by LDS-DMA in 16-KiB stages
//      (4 positions x 64 couts x 32 channels) through a ring of 6, one barrier per stage; the next chunk's raw halo (34 x 10 pixels)
//      by LDS-DMA beside it.
// = 64 MFMAs per chunk and wave for 4 096 outputs x 32 channels: 2.25x fewer than the direct kernel's 144.
// Control: c1_wide_probe.hip's MT = 4 loop (today's kernel) is rebuilt here as `direct` for a same-process comparison.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/build/wino_probe tools/experiments/wino_probe.hip
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

// a - b on eight packed halves as four v_pk_add_f16 with the neg modifiers (hipcc lowers the vector subtraction to scalar v_sub_f16 +
// sdwa + v_pack: three times the VALU instructions)
__device__ __forceinline__ f16x8 pk_sub(f16x8 a, f16x8 b)
{
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned t;
        asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(ua[i]), "v"(ub[i]));
        r[i] = t;
    }
    return __builtin_bit_cast(f16x8, r);
}

// ---------------------------------------------------------------------------------------------------------------- Winograd loop
// XFORM: 1 = the input transform runs (the product form), 0 = ablation: V fragments are plain LDS reads (no VALU work)
template <int XFORM>
__global__ void __launch_bounds__(256, 1)
wino(const unsigned short *w, const unsigned short *x, float *out, int chunks, int w_bytes, int x_bytes)
{
    constexpr int kStage = 16384, kRing = 6, kHaloW = 34, kHaloH = 10, kPitch = 80;
    constexpr int kA = (kHaloW * kHaloH * kPitch + 1023) / 1024 * 1024;          // 27 648
    constexpr int kHaloPieces = kA / 1024;                                         // 27 one-KiB wave pieces per chunk
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char *const bufU = lds, *const bufA = lds + kRing * kStage;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int fcol = lane & 15, fk = lane >> 4;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(w), 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(x), 0, x_bytes, 0x00020000);
    const uint32_t voff = (uint32_t)tid * 16;
    const uint32_t item = blockIdx.x * 65536u;

    f32x4 acc[16][2][2];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[p][m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 V[2][16], U[2][8];

    auto issue = [&](int s) {       // U stage s (16 KiB = 4 pieces per wave) + this stage's share of the next chunk's raw halo
        const uint32_t so = (item + (uint32_t)s * kStage) % (uint32_t)(w_bytes - 2 * kStage);
#pragma unroll
        for (int i = 0; i < 4; ++i) blds16(wr, voff + i * 4096, so, bufU + (s % kRing) * kStage + i * 4096 + wave * 1024);
        const int g = s & 3, q = s >> 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int blk = (g * 2 + i) * 4 + wave;                                // 32 slots for 27 pieces
            if (blk < kHaloPieces)
                blds16(xr, voff & 1023, (item * 7 + (uint32_t)(q + 1) * kA + blk * 1024) % (uint32_t)(x_bytes - kA), bufA + ((q + 1) & 1) * kA + blk * 1024);
        }
    };
    auto read_u = [&](int s, int set) {                // the U fragments of stage s: 4 positions x this wave's 2 cout blocks
        const unsigned char *b = bufU + (s % kRing) * kStage + wn * 2048 + lane * 16;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int n = 0; n < 2; ++n) U[set][p * 2 + n] = *reinterpret_cast<const f16x8 *>(b + p * 4096 + n * 1024);
    };
    auto transform = [&](int q) {                      // V = B^T d B for this wave's two tile rows, all 16 positions
        const unsigned char *a0 = bufA + (q & 1) * kA + fk * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const unsigned char *pa = a0 + (((wm * 2 + m) * 2) * kHaloW + fcol * 2) * kPitch;
            f16x8 d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const f16x8 *>(pa + (i * kHaloW + j) * kPitch);
            if constexpr (XFORM) {
                f16x8 t[4][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    t[0][j] = pk_sub(d[0][j], d[2][j]);
                    t[1][j] = d[1][j] + d[2][j];
                    t[2][j] = pk_sub(d[2][j], d[1][j]);
                    t[3][j] = pk_sub(d[1][j], d[3][j]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    V[m][i * 4 + 0] = pk_sub(t[i][0], t[i][2]);
                    V[m][i * 4 + 1] = t[i][1] + t[i][2];
                    V[m][i * 4 + 2] = pk_sub(t[i][2], t[i][1]);
                    V[m][i * 4 + 3] = pk_sub(t[i][1], t[i][3]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) V[m][i] = d[i >> 2][i & 3];
            }
        }
    };

    // prologue: the first chunk's halo arrives with stage -1's share (here: issued directly), 5 U stages in flight
    {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int blk = i * 4 + wave;
            if (blk < kHaloPieces) blds16(xr, voff & 1023, (item * 7 + blk * 1024) % (uint32_t)(x_bytes - kA), bufA + blk * 1024);
        }
    }
#pragma unroll
    for (int s = 0; s < 5; ++s) issue(s);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    read_u(0, 0);

#pragma unroll 1
    for (int q = 0; q < chunks; ++q) {
        transform(q);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int s = q * 4 + g;
            // stage s + 1 must have landed before its fragments are read below: all but the 4 youngest stages' pieces
            wait_vm<24>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            issue(s + 5);
            read_u(s + 1, (g + 1) & 1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[g * 4 + p][m][n]) : "v"(U[g & 1][p * 2 + n]), "v"(V[m][g * 4 + p]));
            __builtin_amdgcn_s_setprio(0);
        }
    }
    wait_vm<0>();
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) sum += acc[p][m][n];
    if (sum[0] + sum[1] + sum[2] + sum[3] == 12345.678f) out[blockIdx.x * 256 + tid] = sum[0];
}

// ------------------------------------------------------------------------------------------- the direct kernel's loop (control)
__global__ void __launch_bounds__(256, 2)
direct(const unsigned short *w, const unsigned short *x, float *out, int steps, int w_bytes, int x_bytes)
{
    constexpr int MT = 4, NT = 8, kRing = 4, kB = 8192;
    constexpr int kA = (18 * 18 * 80 + 1023) / 1024 * 1024;
    constexpr int kHaloPieces = (kA / 1024 + 3) / 4;
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
    auto issue = [&](int s) {
        blds16(wr, voff, (item + (uint32_t)s * kB) % (uint32_t)(w_bytes - 2 * kB), bufB + (s % kRing) * kB + wave * 1024);
        blds16(wr, voff + 4096, (item + (uint32_t)s * kB) % (uint32_t)(w_bytes - 2 * kB), bufB + (s % kRing) * kB + 4096 + wave * 1024);
        const int tap = s % 9, q = s / 9;
        if (tap * 4 < kA / 1024) {
#pragma unroll
            for (int i = 0; i < (kHaloPieces + 8) / 9; ++i) {
                const int blk = (tap * ((kHaloPieces + 8) / 9) + i) * 4 + wave;
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
        const unsigned char *pA = bufA + (q & 1) * kA + (((tap / 3) * 18 + tap % 3) + (wave * 4) * 18 + fcol) * 80 + fk * 8;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const unsigned char *p = pA + m * 18 * 80;
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
            wait_vm<6>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            issue(s + 3);
            read_frags(s + 1, h ^ 1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[h][n], pf[h][m], acc[n][m], 0, 0, 0);
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

template <typename K>
static double run(const char *name, K kern, int lds, const unsigned short *w, const unsigned short *x, float *out, int w_bytes, int x_bytes, int wgs, int iters,
                  double mfma_per_wg)
{
    CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, 0, w, x, out, iters, w_bytes, x_bytes);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f, tot = 0.f;
    const int reps = 5, launches = 30;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, 0, w, x, out, iters, w_bytes, x_bytes);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        tot += ms;
        best = ms < best ? ms : best;
    }
    const double us = tot / reps / launches * 1e3;
    printf("%-64s %5d workgroups x %3d: %8.1f us per launch (best %8.1f)  %7.1f MFMA-TFLOP/s  (LDS %d KiB)\n", name, wgs, iters, us,
           best / launches * 1e3, 2.0 * wgs * mfma_per_wg * (16.0 * 16 * 32) / (us * 1e-6) / 1e12, lds / 1024);
    return us;
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
    const int lds_w = 6 * 16384 + 2 * 27648, lds_d = 4 * 8192 + 2 * 26624;
    // the 160 x 160 256 -> 256 layer at batch 8: direct = 1 600 workgroups (256 px x 128 couts) x 72 steps;
    // Winograd = 800 pixel tiles x 4 cout tiles = 3 200 workgroups (64 tiles x 64 couts) x 8 chunks
    for (int rep = 0; rep < 2; ++rep) {
        const double d = run("direct (today's loop: 64 px x 128 couts per wave, 2 wg per CU)", direct, lds_d, w, x, out, w_bytes, x_bytes, 1600, 72, 4.0 * 72 * 32);
        const double a = run("winograd F(2x2,3x3): transform in registers + 16 position GEMMs", wino<1>, lds_w, w, x, out, w_bytes, x_bytes, 3200, 8, 4.0 * 8 * 64);
        const double b = run("winograd, ablation: no transform arithmetic (reads only)", wino<0>, lds_w, w, x, out, w_bytes, x_bytes, 3200, 8, 4.0 * 8 * 64);
        printf("   main loop of the layer: direct %.1f us, winograd %.1f us = %.2fx (gate B: >= 1.25x); without the transform's VALU work %.1f us\n", d, a, d / a, b);
    }
    run("direct, long workgroups (no ramp / tail share)", direct, lds_d, w, x, out, w_bytes, x_bytes, 512, 72 * 3, 4.0 * 216 * 32);
    run("winograd, long workgroups", wino<1>, lds_w, w, x, out, w_bytes, x_bytes, 256, 8 * 12, 4.0 * 96 * 64);
    run("winograd, ablation, long workgroups", wino<0>, lds_w, w, x, out, w_bytes, x_bytes, 256, 8 * 12, 4.0 * 96 * 64);
    return 0;
}
