// Split-K implicit-GEMM convolution (3x3 pad 1 / 1x1, stride 1 / 2) for the bf16 NHWC inference engine, with the
// bias / residual-add / ReLU epilogue fused (convolution.forward models/hourglass_104.py:26-30,
// residual.forward :70-79).
//
// Why: at batch 8 the inner hourglass levels (20x20, 10x10, 5x5: 70 of the 3x3 convolutions) are GEMMs with
// only M = 3200 / 800 / 200 output pixels against K = 9*Cin = 3456..4608.  Library kernels tile M x N only, so
// they run 32..200 workgroups with a 54..72-step serial K loop each: 23..42 us per layer for 1..8 GFLOP, i.e.
// 2 % of the network's FLOPs in 20 % of its time.  Here the K loop is split over workgroups as well
// (grid = M tiles x N tiles x K splits ~ one or two workgroups per CU), every workgroup streams its K slice
// through a 3/4-deep ring of LDS stages filled by LDS-DMA (global_load_lds, 16 B per lane), and the fp32
// partial tiles meet inside the same launch: the split that arrives last at a tile's ticket counter sums the
// others' slabs into its registers and applies the epilogue (a separate reduce kernel cost 5-6 us per layer).
//
// GEMM view: D[cout][pixel] = sum_k Wt[cout][k] * X[pixel][k],  k = tap*Cin + ci  (tap = 3*dy+dx).
//   "A" operand = weights, OHWI = the memory order of a channels_last (Cout,Cin,3,3) tensor: row = cout,
//   K contiguous.  "B" operand = im2col rows gathered on the fly: row = pixel, 64 consecutive channels of one
//   tap per K step; taps that fall outside the image (and rows past M) are sourced from a zero page.
//   Both LDS tiles are [row][64 bf16] = 128 B per row, fragments by ds_read_b128.  With the weights as the MFMA
//   A operand the accumulator holds 4 consecutive couts per lane (v_mfma_f32_16x16x32_bf16: D row =
//   4*(lane>>4)+reg, col = lane&15), so partials / outputs are stored 16 B / 8 B per lane, channel-contiguous.
// LDS swizzle: 16-B slot index ^= (row>>1)&7 -- conflict-free for ds_read_b128's lane groups
//   ({0-3,12-15,20-27}, ...; MI355X_MICROARCH "LDS").  LDS-DMA writes lane-linear, so the permutation is applied
//   to the per-lane SOURCE address and again to the read address (same involution).
// Generalised (og_conv2d_bf16 / og_conv2d_proj_bf16): the K loop is "taps x Cin/64 steps of one source tensor", so
//   a 1x1 convolution is the centre tap alone, stride 2 only changes the centre pixel (y*stride, x*stride) a row
//   gathers around, and a residual's 1x1 projection is Cin2/64 more steps that read the block input instead (weights
//   appended along K) -- same tiles, ring, split-K hand-off and epilogue.
// Pipeline: STAGES-deep ring, one raw s_barrier per K step, counted vmcnt (never 0 in steady state): the wait
//   that retires step s comes before the barrier, the reads after it; the stage refilled after the barrier is
//   the one whose reads every wave finished (lgkmcnt(0)) before arriving.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "lp_dtype.h"
#include "og_common.h"

namespace {

typedef lp8 bf16x8;   // 8 x 16-bit operands of one MFMA fragment (lp_dtype.h)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

constexpr int kZeroPageBytes = 256;

struct ConvArgs {
    const unsigned short *x;     // (N,H,W,Cin) bf16
    const unsigned short *w;     // (Cout,3,3,Cin) bf16
    const float *bias;           // fp32[Cout]
    const unsigned short *skip;  // (N,H,W,Cout) bf16 or null
    unsigned short *out;         // (N,H,W,Cout) bf16
    unsigned short *up;          // tiled kernel, or null: (N,2H,2W,Cout) updated in place, up += nearest_x2(result); `out` is not written
    float *partial;              // [tile][ksplit][NT*MT][256 lanes] x 4 fp32 slabs (ksplit > 1)
    int *counters;               // [tiles] arrival tickets, zero between launches
    const unsigned short *zero;  // >= 16 B of zeros
    int N, H, W, Cin, Cout, M;   // H, W = OUTPUT height / width; M = N*H*W output pixels
    int Hin, Win, stride, taps;  // input height / width, stride 1|2, taps 9 (3x3, pad 1) | 1 (1x1, pad 0)
    int n_tiles, steps_per_split, ksplit, relu;
    // optional 1x1 projection of a second tensor, summed into the same accumulators (residual.skip): K steps
    // steps_main .. steps_total-1 read x2 (N,H2,W2,Cin2) at (y*stride2, x*stride2); w rows are [taps*Cin | Cin2] wide
    const unsigned short *x2;
    int Cin2, H2, W2, stride2, steps_main, steps_total, w_row;
    uint32_t magic_w, magic_h, magic_chunks;   // ceil(2^32 / d) for d = W, H, Cin/64 (0 when d == 1): q = umulhi(n, magic)
    int x_bytes, w_bytes;        // tensor sizes for the buffer descriptors of the tiled kernels
    unsigned long long *stamps;  // debug: [workgroup][8] s_memrealtime (100 MHz) marks, or null
    OgWarm warm;                 // tiled kernels: the next layer's weights to touch at entry, or {null, 0}
};

// diagnostic builds only (tools/build_variants.sh conv3x3.hip stamps "-DOG_DEBUG_STAMPS"; -DOG_TILED_STAMPS / -DOG_PW_STAMPS imply it):
// the product library has neither the marks nor og_conv3x3_debug_stamps
#if defined(OG_TILED_STAMPS) || defined(OG_PW_STAMPS)
#ifndef OG_DEBUG_STAMPS
#define OG_DEBUG_STAMPS 1
#endif
#endif
#ifdef OG_DEBUG_STAMPS
#define CONV_STAMP(i)                                                                                         \
    do {                                                                                                      \
        if (a.stamps && tid == 0)                                                                             \
            a.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define CONV_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ unsigned short f2bf(float f) { return f2lp(f); }   // (bf16, or fp16 in the -DOG_DT_F16 build)
__device__ __forceinline__ float bf2f(unsigned short u) { return lp2f(u); }

__device__ __forceinline__ void glds16(const void *g, unsigned char *l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// LDS-DMA through a buffer descriptor: 32-bit per-lane offset + scalar offset, and an out-of-range lane offset reads
// as zero -- the convolution's zero padding without a branch, a second base pointer or 64-bit address registers.
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, unsigned char *l)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)l, 16, voff, soff, 0, 0);
}
constexpr uint32_t kOobOffset = 0x80000000u;  // beyond any tensor this library accepts (< 2 GiB)

// 16-byte write-through (sc1) buffer store with a scalar offset.  HAZARD (gfx950, ROCm 7.2): hipcc pads a VALU write of the
// data registers of a > 8-byte buffer store only when the store's soffset is NOT an SGPR (LLVM's createsVALUHazard); with an
// SGPR soffset the next instruction may be a VALU write of those registers -- seen in the K-split slab stores: `buffer_store_
// dwordx4 v[14:17], ..., s16 offen sc1` directly followed by `v_add_u32 v14, ...` -- and the hardware then stores the NEW v14
// for the last four lanes of every 16-lane group (one slab dword per pixel wrong in ~1 of 3 launches).  Two wait states behind
// every such store close it (cdna_hip_programming.md 5.7: the same rule as for inline-asm x3 / x4 stores); the asm store is
// not counted by the compiler: every caller drains it with an explicit `s_waitcnt vmcnt(0)` before the ticket.
__device__ __forceinline__ void store_b128_sc1(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, f32x4 v)
{
    // store and wait states in ONE asm statement: as separate statements hipcc schedules the next address computation
    // (`v_add_u32 v14, ...`, a write of the just-stored register) between them
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1\n\ts_nop 1" ::"v"(v), "v"(voff), "s"(r), "s"(soff) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm_lgkm0()
{
    // vmcnt(N) lgkmcnt(0): simm16 = vmcnt[3:0] | expcnt(7)<<4 | lgkmcnt(0)<<8 | vmcnt[5:4]<<14
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}

// n / d for the divisors a launch fixes (d*n < 2^32): one v_mul_hi instead of the ~25-instruction software division --
// the per-piece set-up and the per-step tap arithmetic of the split-K kernel are latency, not throughput.
__device__ __forceinline__ uint32_t div_magic(uint32_t n, uint32_t magic) { return magic ? __umulhi(n, magic) : n; }

// Epilogue of one BM x BN tile: lane holds couts co..co+3 of pixel pm in acc[n][m].  Two phases so that the bias and
// residual loads of ALL sub-tiles are in flight together (and, after a split-K hand-off, together with the slab
// gather): written as one loop, every load waited for the previous sub-tile's store (`out` may alias `skip` as far as
// the compiler knows) -- eight dependent round trips, 4 of the 13 us of a 5x5 layer.
template <int BM, int BN>
struct EpiloguePre {
    f32x4 bias[BN / 32];
    u16x4 skip[BM / 32][BN / 32];
};

template <int BM, int BN>
__device__ __forceinline__ void epilogue_load(EpiloguePre<BM, BN> &pre, const ConvArgs &a, int m0, int n0, int wave, int lane)
{
    constexpr int MT = BM / 32, NT = BN / 32;
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int n = 0; n < NT; ++n)
        pre.bias[n] = *reinterpret_cast<const f32x4 *>(a.bias + n0 + wn * (BN / 2) + n * 16 + (lane >> 4) * 4);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int pm = m0 + wm * (BM / 2) + m * 16 + (lane & 15);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = n0 + wn * (BN / 2) + n * 16 + (lane >> 4) * 4;
            pre.skip[m][n] = (u16x4){0, 0, 0, 0};
            if (a.skip && pm < a.M) pre.skip[m][n] = *reinterpret_cast<const u16x4 *>(a.skip + (size_t)pm * a.Cout + co);
        }
    }
}

template <int BM, int BN>
__device__ __forceinline__ void epilogue_finish(const f32x4 (&acc)[BN / 32][BM / 32], const EpiloguePre<BM, BN> &pre,
                                                const ConvArgs &a, int m0, int n0, int wave, int lane)
{
    constexpr int MT = BM / 32, NT = BN / 32;
    const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int pm = m0 + wm * (BM / 2) + m * 16 + (lane & 15);
        if (pm >= a.M) continue;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = n0 + wn * (BN / 2) + n * 16 + (lane >> 4) * 4;
            f32x4 v = acc[n][m] + pre.bias[n];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bf2f(pre.skip[m][n][j]);   // zeros without a residual operand
            u16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = f2bf(a.relu ? fmaxf(v[j], 0.f) : v[j]);
            *reinterpret_cast<u16x4 *>(a.out + (size_t)pm * a.Cout + co) = o;
        }
    }
}

// Row-batched form (128-wide tiles: the prefetched operands of all 16 sub-tiles would not fit the register budget):
// the bias once, then per 16-pixel row block all its residual loads together -- MT dependent round trips instead of 2*MT*NT.
template <int BM, int BN>
__device__ __forceinline__ void epilogue_store(const f32x4 (&acc)[BN / 32][BM / 32], const ConvArgs &a, int m0, int n0,
                                               int wave, int lane)
{
    constexpr int MT = BM / 32, NT = BN / 32;
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 bias[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
        bias[n] = *reinterpret_cast<const f32x4 *>(a.bias + n0 + wn * (BN / 2) + n * 16 + (lane >> 4) * 4);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int pm = m0 + wm * (BM / 2) + m * 16 + (lane & 15);
        if (pm >= a.M) continue;
        u16x4 sk[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = n0 + wn * (BN / 2) + n * 16 + (lane >> 4) * 4;
            sk[n] = (u16x4){0, 0, 0, 0};
            if (a.skip) sk[n] = *reinterpret_cast<const u16x4 *>(a.skip + (size_t)pm * a.Cout + co);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = n0 + wn * (BN / 2) + n * 16 + (lane >> 4) * 4;
            f32x4 v = acc[n][m] + bias[n];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bf2f(sk[n][j]);
            u16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = f2bf(a.relu ? fmaxf(v[j], 0.f) : v[j]);
            *reinterpret_cast<u16x4 *>(a.out + (size_t)pm * a.Cout + co) = o;
        }
    }
}

template <int BM, int BN, int STAGES>
__global__ void __launch_bounds__(256)
conv3x3_kernel(ConvArgs a)
{
    constexpr int kRowB = 128;                      // bytes per LDS tile row (64 bf16)
    constexpr int kStage = (BM + BN) * kRowB;
    constexpr int PX = BM / 32, PW = BN / 32;       // 16-B pieces per thread per step: pixels, weights
    constexpr int G = PX + PW;
    constexpr int MT = BM / 32, NT = BN / 32;       // 16-wide sub-tiles per wave (2 x 2 waves)
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // wave: an SGPR
    const unsigned warm_v = og_warm_touch(a.warm, (blockIdx.y * gridDim.x + blockIdx.x) * 256u + (unsigned)tid);   // the next layer's weights (sunk at the end)
    CONV_STAMP(0);
    const int m_tile = blockIdx.x / a.n_tiles, n_tile = blockIdx.x % a.n_tiles, split = blockIdx.y;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int step0 = split * a.steps_per_split;
    const int nsteps = min(a.steps_per_split, a.steps_total - step0);   // the last split may be shorter
    const int chunks = a.Cin >> 6;                  // K steps per tap

    // ---- loader set-up: piece p = tid + 256*i -> tile row p/8, LDS slot p%8, source chunk slot^((row>>1)&7)
    // output pixel m -> centre input pixel (y*stride, x*stride); a 3x3 tap adds (dy, dx) in -1..1, a 1x1 conv has the
    // centre tap only (pad 0)
    const unsigned short *px_ptr[PX], *px2_ptr[PX];
    uint32_t px_mask[PX];
    const int taps = a.taps;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
        const int p = tid + 256 * i, r = p >> 3, c = (p & 7) ^ ((r >> 1) & 7);
        const int m = m0 + r;
        const int t = (int)div_magic((uint32_t)m, a.magic_w), xw = m - t * a.W;
        const int img = (int)div_magic((uint32_t)t, a.magic_h), y = t - img * a.H;
        const int yc = y * a.stride, xc = xw * a.stride;
        uint32_t mask = 0;
        if (m < a.M) {
            if (taps == 1) mask = 1u;
            else {   // separable: 3 valid columns x 3 valid rows
                const uint32_t cols = (xc >= 1 ? 1u : 0u) | 2u | (xc + 1 < a.Win ? 4u : 0u);
                mask = (yc >= 1 ? cols : 0u) | (cols << 3) | (yc + 1 < a.Hin ? cols << 6 : 0u);
            }
        }
        px_mask[i] = mask;
        px_ptr[i] = a.x + (m < a.M ? ((size_t)img * a.Hin + yc) * a.Win + xc : (size_t)0) * a.Cin + c * 8;
        px2_ptr[i] = (a.x2 && m < a.M)
                         ? a.x2 + (((size_t)img * a.H2 + y * a.stride2) * a.W2 + xw * a.stride2) * a.Cin2 + c * 8
                         : a.zero;
    }
    const unsigned short *w_ptr[PW];   // advanced by one K step (64 elements) per issue
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int p = tid + 256 * i, r = p >> 3, c = (p & 7) ^ ((r >> 1) & 7);
        w_ptr[i] = a.w + (size_t)(n0 + r) * a.w_row + c * 8 + (size_t)step0 * 64;
    }
    const int lds_piece = (wave * 64) * 16;  // wave-uniform; + lane*16 is implied by the DMA

    // K steps are issued strictly in order, so (tap, channel offset, tap shift) are running scalars: no division
    int i_step = step0;
    int i_tap = (int)div_magic((uint32_t)step0, a.magic_chunks), i_ci = (step0 - i_tap * chunks) << 6;
    int i_dy = taps == 1 ? 0 : (i_tap * 11 >> 5) - 1;                      // tap / 3 for tap < 9
    int i_dx = taps == 1 ? 0 : i_tap - (i_dy + 1) * 3 - 1;
    long i_shift = ((long)i_dy * a.Win + i_dx) * a.Cin;
    auto issue = [&](int buf) {
        unsigned char *base = lds + buf * kStage + lds_piece;
        if (i_step < a.steps_main) {
            const long shift = i_shift + i_ci;
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                const unsigned short *src = (px_mask[i] >> i_tap) & 1 ? px_ptr[i] + shift : a.zero;
                glds16(src, base + i * 4096);
            }
            i_ci += 64;
            if (i_ci == a.Cin) {
                i_ci = 0;
                ++i_tap;
                if (++i_dx > 1) { i_dx = -1; ++i_dy; }
                i_shift = ((long)i_dy * a.Win + i_dx) * a.Cin;
            }
        } else {
            const int ci0 = (i_step - a.steps_main) << 6;
#pragma unroll
            for (int i = 0; i < PX; ++i) glds16(px_mask[i] ? px2_ptr[i] + ci0 : a.zero, base + i * 4096);
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            glds16(w_ptr[i], base + BM * kRowB + i * 4096);
            w_ptr[i] += 64;
        }
        ++i_step;
    };

    // ---- compute set-up
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fk = lane >> 4;
    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragment byte offsets inside a stage (swizzle applied), one per (sub-tile, k half): the loop below is unrolled
    // over the ring so that the stage base folds into the ds_read immediate
    uint32_t pfa[MT][2], wfa[NT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int r = wm * (BM / 2) + m * 16 + frow;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) pfa[m][kh] = r * kRowB + (((fk + 4 * kh) ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int r = wn * (BN / 2) + n * 16 + frow;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) wfa[n][kh] = BM * kRowB + r * kRowB + (((fk + 4 * kh) ^ ((r >> 1) & 7)) << 4);
    }

    // ---- prologue: STAGES-1 steps in flight (the host guarantees nsteps >= STAGES-1)
#pragma unroll
    for (int j = 0; j < STAGES - 1; ++j) issue(j);
    CONV_STAMP(1);

    for (int s0 = 0; s0 < nsteps; s0 += STAGES) {
#pragma unroll
        for (int b = 0; b < STAGES; ++b) {               // b = ring slot of step s0 + b (compile time)
            const int s = s0 + b;
            if (s >= nsteps) break;
            if (s + STAGES - 2 < nsteps) wait_vm_lgkm0<(STAGES - 2) * G>();
            else wait_vm_lgkm0<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s == 0) CONV_STAMP(2);
            if (s + STAGES - 1 < nsteps) issue((b + STAGES - 1) % STAGES);
            const unsigned char *st = lds + b * kStage;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                bf16x8 pf[MT], wf[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) pf[m] = *reinterpret_cast<const bf16x8 *>(st + pfa[m][kh]);
#pragma unroll
                for (int n = 0; n < NT; ++n) wf[n] = *reinterpret_cast<const bf16x8 *>(st + wfa[n][kh]);
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[n][m] = OG_LP_MFMA(wf[n], pf[m], acc[n][m]);
            }
        }
    }

    CONV_STAMP(3);
    constexpr bool kPrefetchEpi = MT * NT <= 4;
    EpiloguePre<BM, BN> pre;
    // ---- split-K meeting point: every split writes its fp32 tile as a lane-linear slab (1 KiB per wave store),
    // the split that arrives last adds the others to its registers and runs the epilogue.  Hand-off as
    // cdna_hip_programming.md "in-launch split-K reduction", sc1 form: write-through (sc1) slab stores ->
    // vmcnt(0) -> barrier -> relaxed agent-scope ticket; the last arriver reads every slab with sc1 loads.
    // (An agent-scope release fence per workgroup instead writes back the XCD's whole L2 each time: measured
    // 3-10x slower here.)  Correct for any placement of a tile's splits over XCDs.
    if (a.ksplit > 1) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const int tile = blockIdx.x;
        constexpr uint32_t kSlabBytes = NT * MT * 256 * 16;
        // descriptor and scalar offsets pinned to SGPRs (readfirstlane): left to the compiler they sat in VGPRs and every
        // buffer access was wrapped in a waterfall loop
        uint64_t slab_base = reinterpret_cast<uint64_t>(a.partial) + (uint64_t)tile * a.ksplit * kSlabBytes;
        slab_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(slab_base >> 32)) << 32) |
                    (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)slab_base);
        const __amdgpu_buffer_rsrc_t tile_slabs = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char *>(slab_base), 0, __builtin_amdgcn_readfirstlane((int)(a.ksplit * kSlabBytes)), 0x00020000);
        const uint32_t own_off = __builtin_amdgcn_readfirstlane(split * kSlabBytes);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m)
                store_b128_sc1(tile_slabs, (n * MT + m) * 4096 + tid * 16, own_off, acc[n][m]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int *flag = reinterpret_cast<int *>(lds);
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(a.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == a.ksplit - 1;
            if (last) __hip_atomic_store(a.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
            *flag = last;
        }
        __syncthreads();
        CONV_STAMP(4);
        if (!*flag) return;
        if constexpr (kPrefetchEpi) epilogue_load<BM, BN>(pre, a, m0, n0, wave, lane);   // in flight with the gather
        // all loads of a batch are issued before the first add (one memory round trip per batch, not per slab);
        // slots past ksplit are read out of range, which a buffer load returns as 0
        // (64-wide tiles: 8 slabs = every plan's K split in ONE round trip; two dependent batches cost 2 us per layer)
#ifndef OG_GATHER_SLABS
#define OG_GATHER_SLABS 32
#endif
        constexpr int kBatch = NT * MT >= 16 ? 1 : OG_GATHER_SLABS / (NT * MT);
        // the sum runs over ALL slabs in slice order, the own one re-read like the others: the result does not depend on which slice
        // happened to arrive last (fp32 addition is not associative: adding the others to the own registers made the layer -- and with
        // it the whole forward -- differ in the last bit from run to run; as in the tiled kernel's K split)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int sp0 = 0; sp0 < a.ksplit; sp0 += kBatch) {
            f32x4 part[kBatch][NT * MT];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                const int sp = sp0 + b;
                const uint32_t soff = __builtin_amdgcn_readfirstlane(
                    sp < a.ksplit ? sp * kSlabBytes : a.ksplit * kSlabBytes);
#pragma unroll
                for (int i = 0; i < NT * MT; ++i)
                    part[b][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(tile_slabs, i * 4096 + tid * 16, soff, 16));
            }
#pragma unroll
            for (int b = 0; b < kBatch; ++b)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[n][m] += part[b][n * MT + m];
        }
    }

    if constexpr (kPrefetchEpi) {
        if (a.ksplit <= 1) epilogue_load<BM, BN>(pre, a, m0, n0, wave, lane);
        epilogue_finish<BM, BN>(acc, pre, a, m0, n0, wave, lane);
    } else {
        epilogue_store<BM, BN>(acc, a, m0, n0, wave, lane);
    }
    CONV_STAMP(5);
    og_warm_sink(warm_v);
}

struct Plan {
    int bm, bn, stages, ksplit, steps_per_split, m_tiles, n_tiles;
};

// Tile and K split.  Measured inside the network (bs8, whole-forward ms, round 1): 64-wide tiles with ~6 K splits 10.82, 64/3
// 10.89, shape-dependent 128-wide tiles 11.15, 128/3 11.85, no split 11.82; per layer (us, conv + reduction) 20x20 with 2/3/6/9
// splits = 34.9/31.4/36.2/41.1, 10x10 = 19.5/16.2/15.6/18.5, 5x5 = 22.5/17.4/12.7/12.7.  At 20x20 (M >= 2048) a 128 x 64 tile
// (6 fragment reads per 16 MFMA instead of 8 per 8) with 3 splits and 3 stages: 24.7 us against 29.0 for 64 x 64 x 3.  The
// reduction runs inside the launch (last arriver); the plan sweeps of rounds 1-3 are in EXPERIMENTS.md.
bool make_plan(long M, int Cin, int Cout, Plan &p, int taps = 9, int extra_steps = 0)
{
    const int steps = taps * Cin / 64 + extra_steps;
    p.bm = p.bn = 64;
    p.stages = steps < 3 ? 3 : 4;      // 1x1 convolutions with Cin = 128: two K steps
    p.m_tiles = (int)((M + 63) / 64);
    p.n_tiles = Cout / 64;
    // enough tiles to fill the chip several times over (stride-2 / 1x1 layers of larger levels): no K split
    const int max_ks = (long)p.m_tiles * p.n_tiles >= 1024 ? 1 : M >= 2048 ? 3 : 6;
    int best = 1;
    for (int ks = 2; ks <= max_ks; ++ks)
        if (steps % ks == 0 && steps / ks >= 4 && steps / ks >= p.stages - 1) best = ks;
    if (extra_steps && best < max_ks)  // K with a projection appended rarely divides: allow a shorter last split
        for (int ks = best + 1; ks <= max_ks; ++ks) {
            const int sps = (steps + ks - 1) / ks, last = steps - (ks - 1) * sps;
            if (sps >= 4 && last >= p.stages - 1 && last >= 1) best = ks;
        }
    if (M >= 2048 && max_ks > 1) {   // 20x20 at bs8: the rectangular tile
        const int mk = 3, ms = 3, sps = (steps + mk - 1) / mk, last = steps - (mk - 1) * sps;
        if (sps >= 4 && last >= ms - 1 && (steps % mk == 0 || extra_steps)) {
            best = mk;
            p.stages = ms;
            p.bm = 128;
            p.bn = 64;
            p.m_tiles = (int)((M + 127) / 128);
            p.n_tiles = Cout / 64;
        }
    }
    if (steps < p.stages - 1) return false;
    p.ksplit = best;
    p.steps_per_split = (steps + best - 1) / best;
    return true;
}

// [zero page 256 B | tickets int32[kMaxTiles] | slabs], returns the total size.  The ticket area has a FIXED size:
// layers of different shapes share one workspace, and a ticket must never sit where another layer's slabs go.
constexpr size_t kMaxTiles = 16384;
size_t ws_layout(const Plan &p, size_t *counters_off, size_t *slabs_off)
{
    const size_t tiles = (size_t)p.m_tiles * p.n_tiles;
    const size_t c_off = kZeroPageBytes, s_off = c_off + kMaxTiles * sizeof(int);
    if (counters_off) *counters_off = c_off;
    if (slabs_off) *slabs_off = s_off;
    return s_off + (p.ksplit > 1 ? tiles * p.ksplit * (size_t)p.bm * p.bn * sizeof(float) : 0);
}

static unsigned long long *g_stamps = nullptr;

#include "conv3x3_tiled.inc"

}  // namespace

static inline int conv_out_dim(int in, int ksize, int stride) { return (in + 2 * (ksize / 2) - ksize) / stride + 1; }

// entry points that do not depend on the 16-bit type exist once in the library: defined by the bf16 build only
#ifndef OG_DT_F16
#ifdef OG_DEBUG_STAMPS
// Debug aid of the diagnostic builds (tools/conv_bench.py --stamps, tools/c1_stamps.py, tools/pw_stamps.py): device buffer of
// [workgroups][8] u64 that later launches fill with s_memrealtime marks; NULL switches it off.  Not in the product library.
OG_API void og_conv3x3_debug_stamps(void *buf) { g_stamps = (unsigned long long *)buf; }
#endif

OG_API size_t og_conv3x3_workspace_bytes(long pixels, int Cin, int Cout)
{
    Plan p;
    if (pixels <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64 || !make_plan(pixels, Cin, Cout, p)) return 0;
    return ws_layout(p, nullptr, nullptr);
}


// Exact requirement for one layer.
OG_API size_t og_conv2d_workspace_bytes(int N, int Hin, int Win, int Cin, int Cout, int ksize, int stride)
{
    if (N <= 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64) return 0;
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return 0;
    const int H = conv_out_dim(Hin, ksize, stride), W = conv_out_dim(Win, ksize, stride);
    const long M = (long)N * H * W;
    Plan p;
    if (!make_plan(M, Cin, Cout, p, ksize * ksize)) return 0;
    return ws_layout(p, nullptr, nullptr);
}

OG_API size_t og_conv2d_proj_workspace_bytes(int N, int Hin, int Win, int Cin, int Cout, int ksize, int stride, int Cin2)
{
    if (N <= 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64 || Cin2 <= 0 || Cin2 % 64) return 0;
    if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return 0;
    const long M = (long)N * conv_out_dim(Hin, ksize, stride) * conv_out_dim(Win, ksize, stride);
    Plan p;
    if (!make_plan(M, Cin, Cout, p, ksize * ksize, Cin2 / 64)) return 0;
    return ws_layout(p, nullptr, nullptr);
}

OG_API size_t og_conv3x3_workspace_bytes_nhw(int N, int H, int W, int Cin, int Cout)
{
    return og_conv2d_workspace_bytes(N, H, W, Cin, Cout, 3, 1);
}

#endif

struct Proj {  // optional second operand: out += conv1x1(x2, stride2), weights appended along K
    const void *x2 = nullptr;
    int H2 = 0, W2 = 0, Cin2 = 0, stride2 = 1;
};

static int conv_run(const char *name, const void *x, const void *w, const float *bias, const void *skip, void *out, int N,
                    int Hin, int Win, int Cin, int Cout, int ksize, int stride, int relu, void *workspace,
                    size_t workspace_bytes, void *stream, const Proj &pj = Proj())
{
    OG_REQUIRE(x && w && bias && out && workspace, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && Hin > 0 && Win > 0, OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), OG_EUNSUPPORTED,
               "%s: kernel size %d / stride %d (1x1 and 3x3, stride 1 and 2 only)", name, ksize, stride);
    OG_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0 && Cin > 0 && Cout > 0, OG_EUNSUPPORTED,
               "%s: channels must be multiples of 64 (got %d -> %d)", name, Cin, Cout);
    const int H = conv_out_dim(Hin, ksize, stride), W = conv_out_dim(Win, ksize, stride), taps = ksize * ksize;
    const long M = (long)N * H * W, Min = (long)N * Hin * Win;
    OG_REQUIRE(Min * (long)Cin < (1l << 30) && M * (long)Cout < (1l << 30), OG_EUNSUPPORTED,
               "%s: tensor too large (>= 2 GiB)", name);
    OG_REQUIRE((uintptr_t)workspace % 256 == 0, OG_EINVAL, "%s: workspace must be 256-byte aligned", name);
    hipStream_t st = (hipStream_t)stream;
    if (pj.x2) {
        OG_REQUIRE(pj.Cin2 > 0 && pj.Cin2 % 64 == 0 && (pj.stride2 == 1 || pj.stride2 == 2), OG_EUNSUPPORTED,
                   "%s: projection needs Cin2 %% 64 == 0 and stride 1|2 (got %d, %d)", name, pj.Cin2, pj.stride2);
        OG_REQUIRE(conv_out_dim(pj.H2, 1, pj.stride2) == H && conv_out_dim(pj.W2, 1, pj.stride2) == W, OG_EINVAL,
                   "%s: projection input %dx%d / stride %d does not give the %dx%d output", name, pj.H2, pj.W2, pj.stride2, H, W);
        OG_REQUIRE((long)N * pj.H2 * pj.W2 * pj.Cin2 < (1l << 30), OG_EUNSUPPORTED, "%s: tensor too large (>= 2 GiB)", name);
    }
    Plan p;
    OG_REQUIRE(make_plan(M, Cin, Cout, p, taps, pj.x2 ? pj.Cin2 / 64 : 0), OG_EUNSUPPORTED, "%s: no tile plan", name);
    OG_REQUIRE((size_t)p.m_tiles * p.n_tiles <= kMaxTiles, OG_EUNSUPPORTED, "%s: too many tiles", name);
    const size_t need = ws_layout(p, nullptr, nullptr);
    OG_REQUIRE(workspace_bytes >= need, OG_ENOSPC, "%s: workspace %zu < %zu bytes", name, workspace_bytes, need);

    ConvArgs a = {};
    a.x = (const unsigned short *)x;
    a.w = (const unsigned short *)w;
    a.bias = bias;
    a.skip = (const unsigned short *)skip;
    a.out = (unsigned short *)out;
    a.zero = (const unsigned short *)workspace;  // first 256 B: zero page (workspace is zero-initialised by the caller)
    size_t c_off, s_off;
    ws_layout(p, &c_off, &s_off);
    a.counters = (int *)((char *)workspace + c_off);
    a.partial = (float *)((char *)workspace + s_off);
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.M = (int)M;
    a.Hin = Hin; a.Win = Win; a.stride = stride; a.taps = taps;
    a.x2 = (const unsigned short *)pj.x2;
    a.Cin2 = pj.x2 ? pj.Cin2 : 0; a.H2 = pj.H2; a.W2 = pj.W2; a.stride2 = pj.stride2;
    a.steps_main = taps * Cin / 64;
    a.steps_total = a.steps_main + a.Cin2 / 64;
    a.w_row = taps * Cin + a.Cin2;
    auto magic = [](uint32_t d) { return d <= 1 ? 0u : (uint32_t)(((1ull << 32) + d - 1) / d); };
    a.magic_w = magic((uint32_t)W); a.magic_h = magic((uint32_t)H); a.magic_chunks = magic((uint32_t)(Cin / 64));
    a.warm = og_take_warm_hint();
    a.stamps = g_stamps;
    a.n_tiles = p.n_tiles; a.steps_per_split = p.steps_per_split; a.ksplit = p.ksplit; a.relu = relu;
    const dim3 grid((unsigned)(p.m_tiles * p.n_tiles), (unsigned)p.ksplit);
#define CONV_LAUNCH(BM_, BN_, ST_)                                                                              \
    do {                                                                                                        \
        constexpr int lds_ = ST_ * (BM_ + BN_) * 128;                                                           \
        static OgAttrOnce attr_;                                                                                \
        if (attr_.need())                                                                                       \
            (void)hipFuncSetAttribute((const void *)conv3x3_kernel<BM_, BN_, ST_>,                              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_);                        \
        hipLaunchKernelGGL((conv3x3_kernel<BM_, BN_, ST_>), grid, dim3(256), lds_, st, a);                      \
    } while (0)
    if (p.bm == 128) CONV_LAUNCH(128, 64, 3);
    else if (p.stages == 3) CONV_LAUNCH(64, 64, 3);
    else CONV_LAUNCH(64, 64, 4);
#undef CONV_LAUNCH
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

OG_API int OG_LP_NAME(og_conv3x3)(const void *x, const void *w, const float *bias, const void *skip, void *out, int N, int H,
                           int W, int Cin, int Cout, int relu, void *workspace, size_t workspace_bytes, void *stream)
{
    return conv_run(OG_LP_STR("og_conv3x3"), x, w, bias, skip, out, N, H, W, Cin, Cout, 3, 1, relu, workspace, workspace_bytes,
                    stream);
}

OG_API int OG_LP_NAME(og_conv2d)(const void *x, const void *w, const float *bias, const void *skip, void *out, int N, int Hin,
                          int Win, int Cin, int Cout, int ksize, int stride, int relu, void *workspace,
                          size_t workspace_bytes, void *stream)
{
    return conv_run(OG_LP_STR("og_conv2d"), x, w, bias, skip, out, N, Hin, Win, Cin, Cout, ksize, stride, relu, workspace,
                    workspace_bytes, stream);
}

OG_API int OG_LP_NAME(og_conv2d_proj)(const void *x, const void *w_cat, const float *bias, const void *x2, void *out, int N,
                               int Hin, int Win, int Cin, int Cout, int ksize, int stride, int H2, int W2, int Cin2,
                               int stride2, int relu, void *workspace, size_t workspace_bytes, void *stream)
{
    OG_REQUIRE(x2, OG_EINVAL, "og_conv2d_proj: null pointer");
    Proj pj;
    pj.x2 = x2; pj.H2 = H2; pj.W2 = W2; pj.Cin2 = Cin2; pj.stride2 = stride2;
    return conv_run(OG_LP_STR("og_conv2d_proj"), x, w_cat, bias, nullptr, out, N, Hin, Win, Cin, Cout, ksize, stride, relu, workspace,
                    workspace_bytes, stream, pj);
}

// ---- the tiled kernel (conv3x3_tiled.inc): pre-tiled weights, two workgroups per CU ----
#ifndef OG_DT_F16
// stride 2: H, W = INPUT size (even); 1 = output a multiple of 8 rows x 16 columns, 2 = output 40 wide with an even height
OG_API int og_conv3x3s2_tiled_supported(int N, int H, int W, int Cin, int Cout)
{
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (H & 1) || (W & 1)) return 0;
    if ((long)N * H * W * Cin >= (1l << 30) || (long)N * H * W * Cout >= (1l << 32)) return 0;
    if (Cout % 128 || Cin % 64) return 0;
    return ((H / 2) % 8 == 0 && (W / 2) % 16 == 0) ? 1 : ((W / 2) == 40 && (H / 2) % 2 == 0) ? 2 : 0;
}

OG_API int og_conv3x3_tiled_supported(int N, int H, int W, int Cin, int Cout)
{
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    if ((long)N * H * W * Cin >= (1l << 30) || (long)N * H * W * Cout >= (1l << 30)) return 0;
    return tiled_kind(H, W, Cin, Cout);
}

// 0 when the launch needs no K split; otherwise [zero page 256 B | tickets | fp32 slabs], zero-initialised once by the caller
OG_API size_t og_conv3x3_tiled_workspace_bytes(int N, int H, int W, int Cin, int Cout)
{
    const int kind = og_conv3x3_tiled_supported(N, H, W, Cin, Cout);
    if (!kind) return 0;
    const long items = tiled_items(kind, N, H, W, Cout);
    const int ks = tiled_ksplit(kind, items, Cin);
    if (ks <= 1 || items > (long)kMaxTiles) return 0;
    return kZeroPageBytes + kMaxTiles * sizeof(int) + (size_t)items * ks * tiled_slab_bytes(kind);
}

OG_API int og_conv3x3_pack_w16(const void *w, int Cin, int Cout, int order, void *packed, void *stream)
{
    OG_REQUIRE(order >= 0 && order <= 3, OG_EINVAL,
               "og_conv3x3_pack_w16: order is 0 / 1 (3x3, stride 1 / 2) or 2 / 3 (1x1, cout tiles of 128 / 64)");
    OG_REQUIRE(w && packed, OG_EINVAL, "og_conv3x3_pack_w16: null pointer");
    OG_REQUIRE(Cin > 0 && Cout > 0 && Cin % 64 == 0 && Cout % (order == 3 ? 64 : 128) == 0, OG_EUNSUPPORTED,
               "og_conv3x3_pack_w16: Cin must be a multiple of 64 and Cout of %d (got %d -> %d)", order == 3 ? 64 : 128, Cin, Cout);
    const long slots = (long)Cout * (order >= 2 ? 1 : 9) * Cin / 8;
    hipLaunchKernelGGL(conv3x3_pack_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short *)w, (unsigned short *)packed, Cin, Cout, order);
    OG_LAUNCH_CHECK("og_conv3x3_pack_w16");
    return OG_OK;
}
#endif

namespace {
int conv3x3_tiled_impl(const char *name, const void *x, const void *w_packed, const float *bias, const void *skip, void *out, void *up,
                       int N, int H, int W, int Cin, int Cout, int relu, void *workspace, size_t workspace_bytes, void *stream)
{
    const OgWarm warm = og_take_warm_hint();      // (taken even when the launch is refused: a hint never outlives its call)
    OG_REQUIRE(x && w_packed && bias && (out || up), OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(!up || (uintptr_t)up % 16 == 0, OG_EINVAL, "%s: `up` must be 16-byte aligned", name);
    OG_REQUIRE(N > 0 && H > 0 && W > 0, OG_EINVAL, "%s: bad shape", name);
    const long M = (long)N * H * W;
    OG_REQUIRE(M * (long)Cin < (1l << 30) && M * (long)Cout < (1l << 30), OG_EUNSUPPORTED, "%s: tensor too large (>= 2 GiB)", name);
    const int kind = tiled_kind(H, W, Cin, Cout);
    OG_REQUIRE(kind != 0, OG_EUNSUPPORTED, "%s: needs Cout %% 128 == 0, Cin %% 64 == 0 and H, W multiples of 16 or W == 40 with H %% 4 == 0 "
               "(got %dx%d, %d -> %d)", name, H, W, Cin, Cout);
    ConvArgs h = {};
    h.x = (const unsigned short *)x; h.w = (const unsigned short *)w_packed; h.bias = bias;
    h.skip = (const unsigned short *)skip; h.out = (unsigned short *)out; h.up = (unsigned short *)up;
    h.N = N; h.H = H; h.W = W; h.Cin = Cin; h.Cout = Cout; h.M = (int)M; h.n_tiles = Cout / 128; h.relu = relu;
    h.Hin = H; h.Win = W; h.stride = 1; h.taps = 9;
    h.x_bytes = (int)(M * Cin * 2);
    h.w_bytes = Cout * 9 * Cin * 2;
#ifdef OG_TILED_STAMPS
    h.stamps = g_stamps;
#endif
    hipStream_t st = (hipStream_t)stream;
    h.warm = warm;
    const long items = tiled_items(kind, N, H, W, Cout);
    h.ksplit = (items <= (long)kMaxTiles) ? tiled_ksplit(kind, items, Cin) : 1;
    if (h.ksplit > 1) {
        const size_t need = kZeroPageBytes + kMaxTiles * sizeof(int) + (size_t)items * h.ksplit * tiled_slab_bytes(kind);
        OG_REQUIRE(workspace && workspace_bytes >= need, OG_ENOSPC, "%s: workspace %zu < %zu bytes (og_conv3x3_tiled_workspace_bytes)", name,
                   workspace ? workspace_bytes : (size_t)0, need);
        OG_REQUIRE((uintptr_t)workspace % 256 == 0, OG_EINVAL, "%s: workspace must be 256-byte aligned", name);
        h.counters = (int *)((char *)workspace + kZeroPageBytes);
        h.partial = (float *)((char *)workspace + kZeroPageBytes + kMaxTiles * sizeof(int));
    }
#define TILED_LAUNCH(TW_, TH_, WM_, VAR_)                                                                             \
    do {                                                                                                              \
        constexpr int halo_ = ((TW_ + 2) * (TH_ + 2) * 5 * 16 + 1023) / 1024 * 1024;                                  \
        const int lds_ = 2 * halo_ + (TW_ == 20 ? OG_TILED_RING20 : TW_ == 40 ? OG_TILED_RING40 : 3) * 128 * 64 + 1024; \
        static OgAttrOnce attr_;                                                                                      \
        if (attr_.need())                                                                                             \
            (void)hipFuncSetAttribute((const void *)conv3x3_tiled_kernel<TW_, TH_, WM_, VAR_, 4>,                     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                        \
        const long blocks_ = (long)N * (H / TH_) * (W / TW_) * h.n_tiles * h.ksplit;                                  \
        hipLaunchKernelGGL((conv3x3_tiled_kernel<TW_, TH_, WM_, VAR_, 4>), dim3((unsigned)blocks_), dim3(256), lds_, st, h); \
    } while (0)
    // VAR 26 = 16 (next step's weight fragments in rotating register sets; the step's DMA issue and every fragment read BETWEEN its MFMAs,
    // one item per gap: round 5, -2 % of the network step) + 8 + 2 (what 16 replaces: LDS-DMA issue behind the fragment reads; the other
    // tuning variants -- weight-fragment prefetch, 2 x 2 waves, 8 waves, 20 x 4 tiles at 40x40, no XCD remap, the timing-only ablations --
    // are described in EXPERIMENTS.md)
#ifndef OG_TILED_VAR16
#define OG_TILED_VAR16 26
#endif
    if (kind == 1) TILED_LAUNCH(16, 16, 4, OG_TILED_VAR16);
    else if (kind == 3) TILED_LAUNCH(20, 4, 1, 2);
    else TILED_LAUNCH(40, 4, 2, 2);
#undef TILED_LAUNCH
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}
}  // namespace

OG_API int OG_LP_NAME(og_conv3x3_tiled)(const void *x, const void *w_packed, const float *bias, const void *skip, void *out, int N,
                                        int H, int W, int Cin, int Cout, int relu, void *workspace, size_t workspace_bytes,
                                        void *stream)
{
    return conv3x3_tiled_impl(OG_LP_STR("og_conv3x3_tiled"), x, w_packed, bias, skip, out, nullptr, N, H, W, Cin, Cout, relu, workspace,
                              workspace_bytes, stream);
}

// The last convolution below an hourglass merge and the merge itself (kp_module.forward: up2 = upsample(low3); up1 + up2,
// models/hourglass_104.py:170-176) in one launch: up (N,2H,2W,Cout) += nearest_x2(act(conv3x3(x) + bias + skip)), the
// convolution's result rounded to 16 bits first (= og_conv3x3_tiled_* followed by og_upsample2_add_*, bit for bit).
// The (N,H,W,Cout) tensor in between is never written.
OG_API int OG_LP_NAME(og_conv3x3_tiled_up2)(const void *x, const void *w_packed, const float *bias, const void *skip, void *up, int N,
                                            int H, int W, int Cin, int Cout, int relu, void *workspace, size_t workspace_bytes,
                                            void *stream)
{
    const char *name = OG_LP_STR("og_conv3x3_tiled_up2");
    OG_REQUIRE(up, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE((long)N * H * W * 4 * Cout < (1l << 30), OG_EUNSUPPORTED, "%s: tensor too large (>= 2 GiB)", name);
    return conv3x3_tiled_impl(name, x, w_packed, bias, skip, nullptr, up, N, H, W, Cin, Cout, relu, workspace, workspace_bytes, stream);
}

OG_API int OG_LP_NAME(og_conv3x3s2_tiled)(const void *x, const void *w_packed, const float *bias, const void *skip, void *out,
                                          int N, int Hin, int Win, int Cin, int Cout, int relu, void *stream)
{
    const char *name = OG_LP_STR("og_conv3x3s2_tiled");
    OG_REQUIRE(x && w_packed && bias && out, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && Hin > 0 && Win > 0 && !(Hin & 1) && !(Win & 1), OG_EINVAL, "%s: bad shape", name);
    const int H = Hin / 2, W = Win / 2;
    const long M = (long)N * H * W, Min = (long)N * Hin * Win;
    OG_REQUIRE(Min * (long)Cin < (1l << 30) && M * (long)Cout < (1l << 30), OG_EUNSUPPORTED, "%s: tensor too large (>= 2 GiB)", name);
    const int kind = og_conv3x3s2_tiled_supported(N, Hin, Win, Cin, Cout);
    OG_REQUIRE(kind != 0, OG_EUNSUPPORTED,
               "%s: needs Cout %% 128 == 0, Cin %% 64 == 0 and an output of 8k x 16k pixels or 2k x 40 (got %dx%d -> %dx%d, %d -> %d)", name,
               Hin, Win, H, W, Cin, Cout);
    ConvArgs h = {};
    h.warm = og_take_warm_hint();
    h.x = (const unsigned short *)x; h.w = (const unsigned short *)w_packed; h.bias = bias;
    h.skip = (const unsigned short *)skip; h.out = (unsigned short *)out;
    h.N = N; h.H = H; h.W = W; h.Cin = Cin; h.Cout = Cout; h.M = (int)M; h.n_tiles = Cout / 128; h.relu = relu;
    h.Hin = Hin; h.Win = Win; h.stride = 2; h.taps = 9;
    h.x_bytes = (int)(Min * Cin * 2);
    h.w_bytes = Cout * 9 * Cin * 2;
    constexpr int lds_ = 3 * 128 * 64 + 3 * 3 * 4096;
#define S2_LAUNCH(TW_, TH_, WM_, VAR_)                                                                                \
    do {                                                                                                              \
        static OgAttrOnce attr_;                                                                                      \
        if (attr_.need())                                                                                             \
            (void)hipFuncSetAttribute((const void *)conv3x3s2_tiled_kernel<TW_, TH_, WM_, VAR_>,                      \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds_);                              \
        const long blocks_ = (long)N * (H / TH_) * (W / TW_) * h.n_tiles;                                             \
        hipLaunchKernelGGL((conv3x3s2_tiled_kernel<TW_, TH_, WM_, VAR_>), dim3((unsigned)blocks_), dim3(256), lds_,   \
                           (hipStream_t)stream, h);                                                                   \
    } while (0)
    if (kind == 2) S2_LAUNCH(40, 2, 1, 2);
    else S2_LAUNCH(16, 8, 2, 2);       // DMA issue behind the fragment reads: measured 1-4 % faster
#undef S2_LAUNCH
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

// ---- pointwise (1x1) convolutions of the large levels (conv1x1_tiled_kernel) ----
static int pw_fill(const char *name, PwArgs &a, const void *x1, int C1, int H1, int W1, int s1, const void *x2, int C2, int H2, int W2,
                   int s2, const void *w_packed, const float *bias, int N, int H, int W, int Cout, int bn)
{
    OG_REQUIRE(x1 && w_packed, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && H > 0 && W > 0 && (s1 == 1 || s1 == 2) && (!x2 || s2 == 1 || s2 == 2), OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE(C1 > 0 && C1 % 64 == 0 && (!x2 || (C2 > 0 && C2 % 64 == 0)) && Cout % bn == 0, OG_EUNSUPPORTED,
               "%s: input channels must be multiples of 64 and Cout of %d (got %d + %d -> %d)", name, bn, C1, x2 ? C2 : 0, Cout);
    OG_REQUIRE((H - 1) * s1 < H1 && (W - 1) * s1 < W1, OG_EINVAL, "%s: the strided output reaches outside the input", name);
    OG_REQUIRE(!x2 || (C2 == C1 && H2 == H1 && W2 == W1 && s2 == s1), OG_EUNSUPPORTED,
               "%s: a second input must have the shape and stride of the first", name);
    const long M = (long)N * H * W;
    OG_REQUIRE((long)N * H1 * W1 * C1 < (1l << 30) && (!x2 || (long)N * H2 * W2 * C2 < (1l << 30)) && M * Cout < (1l << 30), OG_EUNSUPPORTED,
               "%s: tensor too large (>= 2 GiB)", name);
    a.x1 = (const unsigned short *)x1; a.x2 = (const unsigned short *)x2; a.w = (const unsigned short *)w_packed; a.bias = bias;
    a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.M = (int)M;
    a.C1 = C1; a.H1 = H1; a.W1 = W1; a.s1 = s1;
    a.C2 = x2 ? C2 : 0; a.H2 = H2; a.W2 = W2; a.s2 = s2;
    a.x1_bytes = (int)((long)N * H1 * W1 * C1 * 2);
    a.x2_bytes = x2 ? (int)((long)N * H2 * W2 * C2 * 2) : 0;
    a.w_bytes = Cout * (a.C1 + a.C2) * 2;
    a.n_tiles = Cout / bn;
    a.items = (int)((M + 255) / 256) * a.n_tiles;
#ifdef OG_PW_STAMPS
    a.stamps = g_stamps;
#endif
    return OG_OK;
}

// Workgroups of a pointwise launch: every item its own workgroup, or -- where the items outnumber the chip's slots -- the persistent
// form: OG_PW_PERSIST workgroups per CU (default 2 = the kernel's occupancy; 0 = off), rounded down to a multiple of 8 x n_tiles so that a
// workgroup keeps its XCD's range of items and its cout tile.
static long pw_blocks(const PwArgs &a)
{
    static const int per_cu = [] { const char *e = getenv("OG_PW_PERSIST"); return e ? atoi(e) : 2; }();
    const long unit = 8l * a.n_tiles;
    long cap = (long)per_cu * og_cu_count() / unit * unit;
    if (per_cu <= 0 || cap <= 0 || a.items <= cap || a.items % 8 != 0) return a.items;
    return cap;
}

OG_API int OG_LP_NAME(og_conv1x1_tiled)(const void *x1, int C1, int H1, int W1, int stride1, const void *x2, int C2, int H2, int W2,
                                        int stride2, const void *w_packed, const float *bias, const void *skip, void *out, int N,
                                        int H, int W, int Cout, int relu, void *stream)
{
    const char *name = OG_LP_STR("og_conv1x1_tiled");
    OG_REQUIRE(out, OG_EINVAL, "%s: null pointer", name);
    PwArgs a = {};
    const int rc = pw_fill(name, a, x1, C1, H1, W1, stride1, x2, C2, H2, W2, stride2, w_packed, bias, N, H, W, Cout, 128);
    if (rc != OG_OK) return rc;
    a.skip = (const unsigned short *)skip; a.out = (unsigned short *)out; a.relu = relu;
    constexpr int lds_ = 256 * (2 * 128 + 16);     // the output tile's staging (68 KiB) > the weight ring (48 KiB); two workgroups per CU
    static_assert(lds_ >= 3 * 2 * 128 * 64 && 2 * lds_ <= 160 * 1024, "LDS");
    static OgAttrOnce attr_;
    if (attr_.need()) {
        (void)hipFuncSetAttribute((const void *)conv1x1_tiled_kernel<128, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_);
        (void)hipFuncSetAttribute((const void *)conv1x1_tiled_kernel<128, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_);
    }
    const long blocks = pw_blocks(a);
    if (((a.C1 + a.C2) >> 6) % 2 == 0)     // two pixel register sets in turn: an even number of 64-channel steps
        hipLaunchKernelGGL((conv1x1_tiled_kernel<128, 0, 2>), dim3((unsigned)blocks), dim3(256), lds_, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((conv1x1_tiled_kernel<128, 0, 1>), dim3((unsigned)blocks), dim3(256), lds_, (hipStream_t)stream, a);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

OG_API int OG_LP_NAME(og_conv1x1_heads)(const void *x, int C, const void *w_packed, const float *bias, int N, int H, int W, int Cout,
                                        int n_heads, const int *head_channels, float *const *outs, void *stream)
{
    const char *name = OG_LP_STR("og_conv1x1_heads");
    OG_REQUIRE(head_channels && outs && bias, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(n_heads >= 1 && n_heads <= 4, OG_EUNSUPPORTED, "%s: 1 to 4 heads", name);
    PwArgs a = {};
    const int rc = pw_fill(name, a, x, C, H, W, 1, nullptr, 0, 0, 0, 1, w_packed, bias, N, H, W, Cout, 64);
    if (rc != OG_OK) return rc;
    int c = 0;
    for (int i = 0; i < 4; ++i) {
        a.first[i] = c;
        if (i < n_heads) {
            OG_REQUIRE(head_channels[i] > 0 && outs[i], OG_EINVAL, "%s: bad head %d", name, i);
            c += head_channels[i];
            a.outf[i] = outs[i];
        }
    }
    a.first[4] = c;
    for (int i = n_heads; i < 4; ++i) a.first[i] = c;     // empty ranges
    OG_REQUIRE(c <= Cout, OG_EINVAL, "%s: the heads have %d channels, the packed weight %d", name, c, Cout);
    constexpr int lds_ = 3 * 2 * 64 * 64;
    const long blocks = a.items;        // (the persistent form loses here: 51.5 vs 49.4 us -- no output staging to overlap, a 64-cout tile)
    if ((a.C1 >> 6) % 2 == 0)
        hipLaunchKernelGGL((conv1x1_tiled_kernel<64, 1, 2>), dim3((unsigned)blocks), dim3(256), lds_, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((conv1x1_tiled_kernel<64, 1, 1>), dim3((unsigned)blocks), dim3(256), lds_, (hipStream_t)stream, a);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}
