// Main-loop probe for VERDICT r4 item 2(a): the C1 inner loop with the WEIGHT fragments loaded straight from global memory (L2) into
// registers in pre-packed fragment order -- no weight stage in LDS, no weight LDS-DMA pieces (8 of the 11 per workgroup-step), no weight
// ds_reads (8 of the 16 per wave-step) -- against today's structure in the same process.  Timing only: operands are random bytes.
//   today : wave = 64 px x 128 couts; per workgroup-step 8 KiB of weights through a 3-deep LDS ring (LDS-DMA), every wave reads all of it
//   regw  : wave = 128 px x 64 couts (2 x 2 waves); per wave-step 4 x 1 KiB weight fragments by buffer_load_dwordx4 two steps ahead
//           (16 KiB per workgroup-step through the vector L1 instead of 8 KiB of LDS-DMA: the two waves that share couts read the same
//           lines), pixels as today (halo image in LDS, two ds_read_b64 per fragment), 8 pixel fragments per wave-step instead of 4
//   (wave = 64 px x 128 couts with 8 weight fragments per wave-step from global needs 3 x 32 weight registers beside 128 accumulators:
//   89 spilled registers, and 32 KiB per workgroup-step through the vector L1 = its whole bandwidth at two workgroups per CU: not run)
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/build/c1_regw_probe tools/experiments/c1_regw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
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

constexpr int kHaloW = 18, kA = (18 * 18 * 80 + 1023) / 1024 * 1024;   // 26 KiB halo image of a 32-channel chunk (16 x 16 pixels + border)

// ---- today's structure (control) ----
__global__ void __launch_bounds__(256, 2)
probe_today(const unsigned short *w, const unsigned short *x, float *out, int steps, int w_bytes, int x_bytes)
{
    constexpr int NT = 8, MT = 4, kRing = 3, kB = 8192;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char *const bufB = lds, *const bufA = lds + kRing * kB;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int fcol = lane & 15, fk = lane >> 4;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(w), 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(x), 0, x_bytes, 0x00020000);
    const uint32_t voff = (uint32_t)tid * 16;
    const uint32_t wbase = (blockIdx.x & 1) * (uint32_t)(w_bytes / 2), xbase = (blockIdx.x >> 1) * 7919u * 1024u;
    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 wf[NT], pf[2][MT];
    auto issue = [&](int s) {
        const uint32_t so = (wbase + (uint32_t)s * kB) % (uint32_t)(w_bytes - 2 * kB);
        blds16(wr, voff, so, bufB + (s % kRing) * kB + wave * 1024);
        blds16(wr, voff + 4096, so, bufB + (s % kRing) * kB + 4096 + wave * 1024);
        const int tap = s % 9, q = s / 9;
        if (tap < 7) {
            const int blk = tap * 4 + wave;
            if (blk * 1024 < kA)
                blds16(xr, voff & 1023, (xbase + (uint32_t)(q + 1) * kA + blk * 1024) % (uint32_t)(x_bytes - kA), bufA + ((q + 1) & 1) * kA + blk * 1024);
        }
    };
    auto read_px = [&](int s, int set) {
        const int tap = s % 9, q = s / 9;
        // two ds_read_b64 as asm, like the product: left to the compiler the pair becomes ONE ds_read2_b64 (half rate) -- the first
        // version of this probe did that in THIS arm only and reported a 10-12 % win for the register-weight arm that the real kernel
        // does not have (profiles/r05_c1_regw_probe.log, first block)
        const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)bufA + (q & 1) * kA +
                              (((tap / 3) * kHaloW + tap % 3) + (wave * 4) * kHaloW + fcol) * 80 + fk * 8;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            u64 lo, hi;
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "i"(m * kHaloW * 80));
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "i"(m * kHaloW * 80 + 32));
            typedef u64 u64x2 __attribute__((ext_vector_type(2)));
            u64x2 v = {lo, hi};
            pf[set][m] = __builtin_bit_cast(f16x8, v);
        }
    };
    issue(0);
    issue(1);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    read_px(0, 0);
#pragma unroll 1
    for (int s0 = 0; s0 < steps; s0 += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int s = s0 + h;
            wait_vm<3>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            read_px(s + 1, h ^ 1);
            const unsigned char *wB = bufB + (s % kRing) * kB + fcol * 64 + ((fk ^ ((4 - ((fcol >> 2) & 3)) & 3)) << 4);
#pragma unroll
            for (int n = 0; n < NT; ++n) wf[n] = *reinterpret_cast<const f16x8 *>(wB + n * 1024);
            issue(s + 2);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[n], pf[h][m], acc[n][m], 0, 0, 0);
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

// ---- weights straight to registers.  WM x WN waves: a wave owns (256 / WM) pixels x (128 / WN) couts ----
template <int WM>
__global__ void __launch_bounds__(256, 2)
probe_regw(const unsigned short *w, const unsigned short *x, float *out, int steps, int w_bytes, int x_bytes)
{
    constexpr int WN = 4 / WM, NT = 8 / WN, MT = 16 / WM, MH = MT / 2;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char *const bufA = lds;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;
    const int fcol = lane & 15, fk = lane >> 4;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(w), 0, w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(x), 0, x_bytes, 0x00020000);
    const uint32_t voff = (uint32_t)tid * 16;
    const uint32_t wbase = (blockIdx.x & 1) * (uint32_t)(w_bytes / 2), xbase = (blockIdx.x >> 1) * 7919u * 1024u;
    const uint32_t wlane = (uint32_t)(lane * 16 + wn * NT * 1024);      // this wave's cout tiles inside a step's 8 KiB
    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // three weight slots (the loads run two steps ahead), written by asm loads the compiler does not count: explicit vmcnt below.
    // Pixel fragments in two halves, ONE set each (a second full set does not fit beside three weight slots): half B of step s is
    // read at the top of step s, half A of step s + 1 between the two MFMA halves of step s
    u32x4 wg[3][NT];
    f16x8 pfa[MH], pfb[MH];
    auto load_w = [&](int s, int slot) {
        const uint32_t so = (wbase + (uint32_t)s * 8192u) % (uint32_t)(w_bytes - 2 * 8192);
#pragma unroll
        for (int n = 0; n < NT; ++n)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=v"(wg[slot][n]) : "v"(wlane), "s"(wr), "s"(so), "i"(n * 1024) : "memory");
    };
    auto issue_halo = [&](int tap, int q) {     // one piece per wave and step for taps 0..6 (blocks past the image go to a dump KiB)
        const int blk = tap * 4 + wave;
        blds16(xr, voff & 1023, (xbase + (uint32_t)(q + 1) * kA + blk * 1024) % (uint32_t)(x_bytes - kA),
               blk * 1024 < kA ? bufA + ((q + 1) & 1) * kA + blk * 1024 : bufA + 2 * kA + (wave & 1) * 1024);
    };
    auto read_px = [&](int tap, int q, int half, f16x8 (&pf)[MH]) {
        const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)bufA + (q & 1) * kA +
                              (((tap / 3) * kHaloW + tap % 3) + (wm * MT + half * MH) * kHaloW + fcol) * 80 + fk * 8;
#pragma unroll
        for (int m = 0; m < MH; ++m) {
            u64 lo, hi;
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "i"(m * kHaloW * 80));
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "i"(m * kHaloW * 80 + 32));
            typedef u64 u64x2 __attribute__((ext_vector_type(2)));
            u64x2 v = {lo, hi};
            pf[m] = __builtin_bit_cast(f16x8, v);
        }
    };
    load_w(0, 0);
    load_w(1, 1);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    read_px(0, 0, 0, pfa);
#pragma unroll 1
    for (int s0 = 0; s0 < steps; s0 += 9) {      // one chunk per trip: tap and weight slot (9 = 0 mod 3) are compile-time constants
        const int q = s0 / 9;
#pragma unroll
        for (int h = 0; h < 9; ++h) {
            // weights(s) were issued two steps ago; younger: what step s - 1 issued = NT weight loads (+ a halo piece for taps 0..6).
            // lgkmcnt(0): half A of this step is in its registers
            if (h >= 1 && h - 1 < 7) wait_vm<NT + 1>();
            else wait_vm<NT>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            read_px(h, q, 1, pfb);
            load_w(s0 + h + 2, (h + 2) % 3);
            if (h < 7) issue_halo(h, q);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MH; ++m)
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wg[h % 3][n]), pfa[m], acc[n][m], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // half B has landed; half A's registers are free (their MFMAs are issued)
            read_px((h + 1) % 9, q + (h + 1) / 9, 0, pfa);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int m = 0; m < MH; ++m)
                    acc[n][MH + m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wg[h % 3][n]), pfb[m], acc[n][MH + m], 0, 0, 0);
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

template <class K>
static void run(const char *name, K kern, int lds, const unsigned short *w, const unsigned short *x, float *out, int w_bytes, int x_bytes, int wgs, int steps)
{
    CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, 0, w, x, out, steps, w_bytes, x_bytes);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f, tot = 0.f;
    const int reps = 5, launches = 30;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, 0, w, x, out, steps, w_bytes, x_bytes);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        tot += ms;
        best = ms < best ? ms : best;
    }
    const double flop = 2.0 * wgs * 4.0 * steps * 32 * (16.0 * 16 * 32);
    printf("%-58s %5d workgroups x %3d steps: %8.1f us per launch (best %8.1f) = %7.1f TFLOP/s (LDS %d KiB)\n", name, wgs, steps,
           tot / reps / launches * 1e3, best / launches * 1e3, flop / (tot / reps / launches * 1e-3) / 1e12, lds / 1024);
}

int main()
{
    const int w_bytes = 2 * 72 * 8192 + 4 * 8192, x_bytes = 256 << 20;     // two cout tiles of a 256 -> 256 layer: L2-resident, as in the product
    unsigned short *w, *x;
    float *out;
    CHECK(hipMalloc(&w, w_bytes));
    CHECK(hipMalloc(&x, x_bytes));
    CHECK(hipMalloc(&out, 1 << 24));
    std::vector<unsigned short> h(x_bytes / 2);
    for (size_t i = 0; i < h.size(); ++i) {     // random fp16 in (-2, 2)
        const unsigned r = (unsigned)rand();
        h[i] = (unsigned short)(((r & 1) << 15) | ((12 + ((r >> 1) & 3)) << 10) | ((r >> 3) & 1023));
    }
    CHECK(hipMemcpy(x, h.data(), x_bytes, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(w, h.data(), w_bytes, hipMemcpyHostToDevice));
    const int lds_today = 3 * 8192 + 2 * kA, lds_regw = 2 * kA + 2048;
    for (int rep = 0; rep < 2; ++rep) {
        run("today: 64 px x 128 couts per wave, weights via LDS ring", probe_today, lds_today, w, x, out, w_bytes, x_bytes, 1600, 72);
        run("regw : 128 px x 64 couts per wave, weights -> registers", probe_regw<2>, lds_regw, w, x, out, w_bytes, x_bytes, 1600, 72);
        run("today, long workgroups (3 x 72 steps)", probe_today, lds_today, w, x, out, w_bytes, x_bytes, 512, 216);
        run("regw , long workgroups", probe_regw<2>, lds_regw, w, x, out, w_bytes, x_bytes, 512, 216);
    }
    return 0;
}
