// Band-resident 3x3 convolution (+ optional 1x1 projection along K) for the small hourglass levels: 20x20, 10x10 and 5x5 at
// batch 8 (convolution.forward models/hourglass_104.py:26-30, residual.forward :70-79, the bottom of kp_module :183-190),
// one layer per launch (og_conv_band_*).
//
// Why another kernel: those layers are GEMMs with M = 25 ... 400 pixels per image against K = 9*Cin = 3456 ... 4608.  The split-K
// kernel of csrc/conv3x3.hip spends 61 % of a 13 us layer outside its K loop (set-up, fp32 slab publish, arrival ticket,
// last-arriver gather: profiles/r03_o_chain_stamps.log).  Here nothing but finished activations is handed between workgroups:
//   * a workgroup owns (image, band of output rows, 16 output channels) and ALL of K;
//   * K is split over the WAVES of the workgroup: four waves (one per SIMD), wave w owns the 32-channel chunks kw*w .. kw*w+kw-1
//     for all nine taps; its MFMA A operands -- 9 * kw fragments of 16 couts x 32 channels -- are 1-KiB contiguous pieces of a
//     pre-packed weight image that go straight from global memory into its registers: one round trip, no LDS, no barrier;
//   * the band's input rows (+ one above / below) sit in LDS once, [pixel][Cin] with a 32-byte pad per pixel, ONE ZERO PIXEL
//     between consecutive rows and a zero row above / below the image: a tap is then a plain address shift for every lane (no
//     validity select), and the 16 columns of an MFMA block are 16 CONSECUTIVE LDS pixels (the zero pixel between two rows is
//     an output position whose result is dropped), so every ds_read_b128 is bank-conflict-free (slots 2p / 2p+1 mod 16; stride 2:
//     16-byte pad, slots p / p+1).  Fragment reads run one step ahead of the MFMAs that consume them;
//   * the four fp32 partial tiles meet in LDS (the staged input is dead by then), every output element is summed once and
//     leaves through the fused bias / residual / ReLU epilogue as 8-byte stores (4 couts of one pixel per lane).
// Work is XCD-aware: the workgroups that share a weight slice (same couts, different images / bands) run on one XCD, so a
// slice crosses the fabric once per layer (speed only: no result depends on the placement).
// (A chained persistent form -- a run of dependent layers as ONE launch with ticket queues and per-image completion counters --
// was built and measured in round 4: 7.4 instead of 8.2 us per 5x5 layer alone, +1..4 % per step in the network because it holds
// all 256 CUs; it lives in tools/experiments/conv_band_chain.patch, not in the product.)
#include <stdio.h>
#include <stdlib.h>

#include "lp_dtype.h"
#include "og_common.h"

namespace {

typedef lp8 frag8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr uint32_t kOob = 0x80000000u;   // per-lane buffer offset beyond any tensor this library accepts (< 2 GiB): reads as 0
constexpr int kWaves = 4;                // one per SIMD
constexpr int kMaxLds = 160 * 1024 - 64; // LDS plan limit

struct BandLayer {
    const unsigned short *x;     // (N,Hin,Win,Cin)
    const unsigned short *w;     // packed: [cout tile of 16][chunk of 32 channels][tap][lane][8], then the projection's [tile][chunk][lane][8]
    const float *bias;           // fp32[Cout]
    const unsigned short *skip;  // (N,H,W,Cout) or null
    const unsigned short *x2;    // projection input (N,H2,W2,Cin2) or null
    unsigned short *out;         // (N,H,W,Cout)
    int N, Hin, Win, Cin, H, W, Cout, stride, relu;
    int Cin2, H2, W2, stride2;
    int bands, band_rows;        // output rows per band (the last band may be shorter)
    int groups, total;           // groups = Cout / 16; total = groups * N * bands work items
    int pitch, pitch2;           // LDS bytes per pixel of the x / x2 images
    int x2_off, sb_off;          // LDS byte offsets of the x2 image and of the epilogue operands (residual slice + biases)
    int w_bytes;
    int kw, kw2;                 // 32-channel chunks per wave: ceil(Cin / 32 / 4), ceil(Cin2 / 32 / 4)
    uint32_t magic_p;            // ceil(2^32 / (Win + 1))
    OgWarm warm;                 // the next layer's weights to touch at entry (og_conv_next_weights_hint), or {null, 0}
};

#ifdef OG_BAND_STAMPS   // tuning builds only (tools/build_variants.sh conv_band.hip stamps -DOG_BAND_STAMPS, tools/band_stamps.py)
#define BAND_STAMP(i)                                                                                              \
    do {                                                                                                           \
        if (stamps && threadIdx.x == 0) stamps[(i)] = __builtin_amdgcn_s_memrealtime();                            \
    } while (0)
#else
#define BAND_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ uint32_t div_magic(uint32_t n, uint32_t magic) { return __umulhi(n, magic); }

__device__ __forceinline__ u32x4 load16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p, int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

// This wave's K slice of cout tile g, straight to registers.  No branches around a wave's work: a chunk past the layer's
// (Cin = 384 on the 16-chunk variant, Cin = 64 on any) loads through an out-of-range lane offset (zeros, no memory traffic)
// and multiplies zeros -- uniform branches here made hipcc carry every accumulator through phi copies and spill.
template <int KW, int KW2>
__device__ __forceinline__ void band_load_weights(const BandLayer &a, int g, u32x4 (&wf)[KW][9], u32x4 (&wp)[KW2 > 0 ? KW2 : 1])
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nch = a.Cin >> 5, nch2 = a.Cin2 >> 5;
    const __amdgpu_buffer_rsrc_t wr = rsrc_of(a.w, a.w_bytes);
#pragma unroll
    for (int k = 0; k < KW; ++k) {
        const int ch = wave * a.kw + k;
        const uint32_t voff = (k < a.kw && ch < nch) ? (uint32_t)(lane << 4) : kOob;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) wf[k][tap] = load16(wr, voff, (uint32_t)(((g * nch + ch) * 9 + tap) << 10));
    }
    if (KW2 > 0) {
        const uint32_t proj_base = (uint32_t)a.Cout * 9u * (uint32_t)a.Cin * 2u;
#pragma unroll
        for (int k = 0; k < KW2; ++k) {
            const int ch = wave * a.kw2 + k;
            const uint32_t voff = (a.x2 && k < a.kw2 && ch < nch2) ? (uint32_t)(lane << 4) : kOob;
            wp[k] = load16(wr, voff, proj_base + (uint32_t)((g * nch2 + ch) << 10));
        }
    }
}

#ifdef OG_BAND_STAMPS
__device__ __forceinline__ int xcc_id_dbg() { return (int)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 15u); }
#endif

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: the weight loads issued at kernel entry
// (~40 KiB per workgroup) would have to land before the first barrier instead of before the first MFMA.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// One work item: layer `a`, cout tile g, (image, band), on a workgroup of FOUR waves (one per SIMD, <= 256 registers: the
// workgroup shares its CU with the kernels that run beside it in the network).  The waves first stage the band -- zero the border
// pixels, the input rows, the projection's pixels, the residual / bias operands of the epilogue -- then run the MFMAs on their K
// slices (weights requested by the caller: band_load_weights), exchange the partial tiles, run the epilogue and store.
// PT: 16-position column blocks per band (2, 4 or 7); KW / KW2: 32-channel chunks per wave of the 3x3 operand / of the
// projection (compile-time upper bounds, a.kw / a.kw2 are the layer's).
template <int PT, int KW, int KW2>
__device__ __forceinline__ void band_role(const BandLayer &a, int g, int item, u32x4 (&wf)[KW][9], u32x4 (&wp)[KW2 > 0 ? KW2 : 1],
                                          unsigned char *lds, unsigned long long *stamps)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ltid = tid, lwave = wave;
    const int nch = a.Cin >> 5, nch2 = a.Cin2 >> 5;
    const int img = item / a.bands, band = item - img * a.bands;
    const int y0 = band * a.band_rows, rows = min(a.band_rows, a.H - y0);
    const int P = a.Win + 1;                       // LDS pixels per image row: the row + one zero pixel
    const int NR = (rows - 1) * a.stride + 3;      // LDS rows: input rows y0*stride-1 .. (y0+rows-1)*stride+1 (zeros outside the image)
    const int ri0 = y0 * a.stride - 1;
    const int Q = rows * P - 1;                    // output positions of the band (the zero pixels between rows included)
    const bool has_proj = KW2 > 0 && a.x2 != nullptr;
#ifdef OG_BAND_STAMPS
#define LSTAMP(i) do { if (stamps && ltid == 0) stamps[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define CSTAMP(i) do { if (stamps && tid == 0) stamps[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LSTAMP(i) do { } while (0)
#define CSTAMP(i) do { } while (0)
#endif
    LSTAMP(0);

    {
        // ---- zero pixels of the LDS image: the leading pixel, the pixel behind every row, the rows outside the image
        const int slots = a.pitch >> 4;            // 16-byte slots per pixel
        for (int s = ltid; s < (NR + 1) * slots; s += kWaves * 64) {
            const int i = s / slots, sl = s - i * slots;
            const int px = i == 0 ? 0 : 1 + (i - 1) * P + a.Win;
            *reinterpret_cast<u32x4 *>(lds + px * a.pitch + (sl << 4)) = (u32x4){0, 0, 0, 0};
        }
        if (ri0 < 0)
            for (int s = ltid; s < a.Win * slots; s += kWaves * 64) *reinterpret_cast<u32x4 *>(lds + a.pitch + (s << 4)) = (u32x4){0, 0, 0, 0};
        if (ri0 + NR - 1 >= a.Hin)
            for (int s = ltid; s < a.Win * slots; s += kWaves * 64)
                *reinterpret_cast<u32x4 *>(lds + (1 + (NR - 1) * P) * a.pitch + (s << 4)) = (u32x4){0, 0, 0, 0};
    }
    LSTAMP(1);

    {
        // ---- activations: whole pixels, one per wave instruction (Cin / 8 <= 64 lanes x 16 B), into the padded LDS image
        {
            const int r_lo = max(ri0, 0), r_hi = min(ri0 + NR - 1, a.Hin - 1);      // input rows that exist
            const int npix = (r_hi - r_lo + 1) * a.Win;
            const unsigned short *src = a.x + (size_t)(img * a.Hin + r_lo) * a.Win * a.Cin;
            const __amdgpu_buffer_rsrc_t xr = rsrc_of(src, npix * a.Cin * 2);
            const bool lane_on = lane < (a.Cin >> 3);
            const uint32_t magic_w = a.Win <= 1 ? 0u : (uint32_t)(((1ull << 32) + a.Win - 1) / a.Win);
            constexpr int SB = 8;
            for (int p0 = lwave; p0 < npix; p0 += SB * kWaves) {
                u32x4 v[SB];
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const int p = p0 + i * kWaves;
                    const uint32_t vo = (lane_on && p < npix) ? (uint32_t)((lane << 4) + p * a.Cin * 2) : kOob;
                    v[i] = load16(xr, vo, 0);
                }
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const int p = p0 + i * kWaves;
                    const int r = magic_w ? (int)__umulhi((uint32_t)p, magic_w) : p, c = p - r * a.Win;
                    const int px = 1 + (r_lo - ri0 + r) * P + c;
                    if (lane_on && p < npix) *reinterpret_cast<u32x4 *>(lds + px * a.pitch + (lane << 4)) = v[i];
                }
            }
        }
        if (has_proj) {   // the projection reads the block input at (y * stride2, x * stride2): staged dense by output position
            const __amdgpu_buffer_rsrc_t x2r = rsrc_of(a.x2, a.N * a.H2 * a.W2 * a.Cin2 * 2);
            const bool lane_on = lane < (a.Cin2 >> 3);
            constexpr int SB = 8;
            for (int q0 = lwave; q0 < Q; q0 += SB * kWaves) {
                u32x4 v[SB];
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const int q = q0 + i * kWaves;
                    const int y = (int)div_magic((uint32_t)q, a.magic_p), x = q - y * P;
                    const bool ok = lane_on && q < Q && x < a.W;
                    const uint32_t vo = ok ? (uint32_t)(((((img * a.H2 + (y0 + y) * a.stride2) * a.W2 + x * a.stride2) * a.Cin2) << 1) + (lane << 4)) : kOob;
                    v[i] = load16(x2r, vo, 0);
                }
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const int q = q0 + i * kWaves;
                    if (lane_on && q < Q) *reinterpret_cast<u32x4 *>(lds + a.x2_off + q * a.pitch2 + (lane << 4)) = v[i];
                }
            }
        }
        // ---- epilogue operands: [position][16 couts] of the residual (32 B each), then the tile's 16 biases (64 B)
        {
            const __amdgpu_buffer_rsrc_t sr = rsrc_of(a.skip, a.skip ? a.N * a.H * a.W * a.Cout * 2 : 0);
            for (int s = ltid; s < PT * 32; s += kWaves * 64) {      // 16-byte piece s: position s / 2, couts 8 * (s & 1) ..
                const int q = s >> 1;
                const int y = (int)div_magic((uint32_t)q, a.magic_p), x = q - y * P;
                const bool ok = a.skip && q < Q && x < a.W;
                const uint32_t vo = ok ? (uint32_t)(((((img * a.H + y0 + y) * a.W + x) * a.Cout + g * 16 + (s & 1) * 8)) << 1) : kOob;
                const u32x4 v = load16(sr, vo, 0);
                *reinterpret_cast<u32x4 *>(lds + a.sb_off + (s << 4)) = v;
            }
            if (ltid < 4) *reinterpret_cast<f32x4 *>(lds + a.sb_off + PT * 512 + (ltid << 4)) = *reinterpret_cast<const f32x4 *>(a.bias + g * 16 + ltid * 4);
        }
    }
    LSTAMP(2);
    lds_barrier();

    // ---- MFMA: position q = 16 t + column reads LDS pixel 1 + P + stride*q + (dy*P + dx).  Fragment reads run D steps ahead of
    // the MFMAs that consume them (sched_barrier: left alone, hipcc sinks every read in front of its MFMA -- one read in flight,
    // 37 cycles per MFMA instead of 16 -- and recomputes the 64-bit address products per read)
    const int col = lane & 15, fk = lane >> 4;
    f32x4 acc[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
        constexpr int D = PT >= 7 ? 1 : PT == 4 ? 2 : 3, NB = D + 1, NS = KW * 9;
        int base[PT];
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            base[t] = (1 + P + a.stride * (t * 16 + col)) * a.pitch + (fk << 4);
            asm volatile("" : "+v"(base[t]));          // a register, not a recipe
        }
        auto step_off = [&](int s) {     // s = k * 9 + tap; a chunk past the layer's multiplies zero weights: any in-range chunk does
            const int k = s / 9, tap = s - k * 9;
            const int ch = min(wave * a.kw + min(k, a.kw - 1), nch - 1);
            return ((tap / 3 - 1) * P + (tap % 3 - 1)) * a.pitch + (ch << 6);
        };
        frag8 b[NB][PT];
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const int off = step_off(s);
#pragma unroll
            for (int t = 0; t < PT; ++t) b[s % NB][t] = *reinterpret_cast<const frag8 *>(lds + base[t] + off);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + D < NS) {
                const int off = step_off(s + D);
#pragma unroll
                for (int t = 0; t < PT; ++t) b[(s + D) % NB][t] = *reinterpret_cast<const frag8 *>(lds + base[t] + off);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < PT; ++t) acc[t] = OG_LP_MFMA(__builtin_bit_cast(frag8, wf[s / 9][s % 9]), b[s % NB][t], acc[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (has_proj) {
#pragma unroll
            for (int k = 0; k < (KW2 > 0 ? KW2 : 1); ++k) {
                const int ch = min(wave * a.kw2 + min(k, a.kw2 - 1), nch2 - 1);
                frag8 bp[PT];
#pragma unroll
                for (int t = 0; t < PT; ++t) bp[t] = *reinterpret_cast<const frag8 *>(lds + a.x2_off + (t * 16 + col) * a.pitch2 + (ch << 6) + (fk << 4));
#pragma unroll
                for (int t = 0; t < PT; ++t) acc[t] = OG_LP_MFMA(__builtin_bit_cast(frag8, wp[k]), bp[t], acc[t]);
            }
        }
    }
    CSTAMP(3);

    // ---- the four partial tiles meet in LDS: [wave][t][lane] float4 over the (dead) staged input
    lds_barrier();
#pragma unroll
    for (int t = 0; t < PT; ++t) *reinterpret_cast<f32x4 *>(lds + (((wave * PT + t) << 6) + lane) * 16) = acc[t];
    lds_barrier();
    CSTAMP(4);
    {
        // epilogue: thread o owns 4 couts of one output position (PT = 7: 448 owners on 256 threads)
#pragma unroll
        for (int rep = 0; rep < (PT > 4 ? 2 : 1); ++rep) {
            const int o = tid + rep * kWaves * 64, l = o & 63, t = o >> 6;
            const int q = t * 16 + (l & 15);
            const int y = (int)div_magic((uint32_t)q, a.magic_p), x = q - y * P;
            if (o < PT * 64 && q < Q && x < a.W) {
                f32x4 v = *reinterpret_cast<const f32x4 *>(lds + a.sb_off + PT * 512 + ((l >> 4) << 4));
#pragma unroll
                for (int w = 0; w < kWaves; ++w) v += *reinterpret_cast<const f32x4 *>(lds + (((w * PT + t) << 6) + l) * 16);
                const u32x2 sk = *reinterpret_cast<const u32x2 *>(lds + a.sb_off + (q << 5) + ((l >> 4) << 3));
                v[0] += lp2f((unsigned short)(sk[0] & 0xffffu));
                v[1] += lp2f((unsigned short)(sk[0] >> 16));
                v[2] += lp2f((unsigned short)(sk[1] & 0xffffu));
                v[3] += lp2f((unsigned short)(sk[1] >> 16));
                if (a.relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                u32x2 ov;
                ov[0] = (uint32_t)f2lp(v[0]) | ((uint32_t)f2lp(v[1]) << 16);
                ov[1] = (uint32_t)f2lp(v[2]) | ((uint32_t)f2lp(v[3]) << 16);
                const size_t oidx = ((size_t)(img * a.H + y0 + y) * a.W + x) * a.Cout + g * 16 + (l >> 4) * 4;
                *reinterpret_cast<u32x2 *>(a.out + oidx) = ov;
            }
        }
    }
    CSTAMP(5);
#ifdef OG_BAND_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CSTAMP(6);
#endif
#undef LSTAMP
#undef CSTAMP
}

template <int PT, int KW, int KW2>
__global__ void __launch_bounds__(kWaves * 64)
conv_band_kernel(BandLayer a, unsigned long long *stamps)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const unsigned warm_v = og_warm_touch(a.warm, blockIdx.x * (unsigned)(kWaves * 64) + threadIdx.x);   // first in the wave's queue
    const int wid = og_xcd_remap(blockIdx.x, gridDim.x);   // grid: a multiple of 8; consecutive work ids share an XCD
    if (wid >= a.total) return;
    const int items = a.N * a.bands;
    const int g = wid / items;
    u32x4 wf[KW][9], wp[KW2 > 0 ? KW2 : 1];
    band_load_weights<KW, KW2>(a, g, wf, wp);
    band_role<PT, KW, KW2>(a, g, wid - g * items, wf, wp, lds, stamps ? stamps + (size_t)blockIdx.x * 8 : nullptr);
    og_warm_sink(warm_v);
}

// OHWI weights (the memory order of a channels_last (Cout,Cin,3,3) tensor) [+ the projection's (Cout,Cin2)] -> fragment order:
// every (cout tile, chunk, tap) is the 1-KiB register image of one wave-wide 16-byte load (lane = cout row + 16 * k octet).
__global__ void __launch_bounds__(256)
conv_band_pack_kernel(const unsigned short *w, const unsigned short *w2, unsigned short *packed, int Cin, int Cout, int Cin2)
{
    const long slot = (long)blockIdx.x * 256 + threadIdx.x;
    const long main_slots = (long)Cout * 9 * Cin / 8, proj_slots = (long)Cout * Cin2 / 8;
    if (slot >= main_slots + proj_slots) return;
    const unsigned short *src;
    if (slot < main_slots) {
        const int lane = (int)(slot & 63);
        long rest = slot >> 6;
        const int tap = (int)(rest % 9);
        rest /= 9;
        const int nch = Cin >> 5, chunk = (int)(rest % nch), g = (int)(rest / nch);
        src = w + ((size_t)(g * 16 + (lane & 15)) * 9 + tap) * Cin + chunk * 32 + (lane >> 4) * 8;
    } else {
        const long s2 = slot - main_slots;
        const int lane = (int)(s2 & 63);
        const long rest = s2 >> 6;
        const int nch2 = Cin2 >> 5, chunk = (int)(rest % nch2), g = (int)(rest / nch2);
        src = w2 + (size_t)(g * 16 + (lane & 15)) * Cin2 + chunk * 32 + (lane >> 4) * 8;
    }
    *reinterpret_cast<u32x4 *>(packed + slot * 8) = *reinterpret_cast<const u32x4 *>(src);
}

struct BandPlan {
    int pt, lds_bytes, grid;
};

// LDS bytes of layer `a` on a kernel with pt column blocks (sets a.x2_off, a.sb_off)
int band_lds(BandLayer &a, int pt)
{
    const int P = a.Win + 1, NR = (a.band_rows - 1) * a.stride + 3;
    // fragment reads of the dropped positions of the last column block reach past the image: the allocation covers them
    const int reach = 1 + P + a.stride * (pt * 16 - 1) + P + 2;
    const int img_px = (1 + NR * P) > reach ? (1 + NR * P) : reach;
    a.x2_off = (img_px * a.pitch + 15) & ~15;
    const int staged = a.x2_off + pt * 16 * a.pitch2;
    // the epilogue operands sit behind the staged images AND behind the partial tiles that overlay them
    a.sb_off = staged > kWaves * pt * 1024 ? staged : kWaves * pt * 1024;
    return a.sb_off + pt * 512 + 64;
}

// Fills the geometry fields of `a` (shapes must be set) and picks the kernel variant; false = this layer is not served.
bool band_plan(BandLayer &a, BandPlan &p)
{
    if (a.Cin % 32 || a.Cin < 64 || a.Cin > 512 || a.Cout % 16 || a.Cout <= 0) return false;
    if (a.stride != 1 && a.stride != 2) return false;
    if (a.x2 && (a.Cin2 % 32 || a.Cin2 < 64 || a.Cin2 > 512 || (a.stride2 != 1 && a.stride2 != 2))) return false;
    if (!a.x2) a.Cin2 = 0;
    a.H = (a.Hin - 1) / a.stride + 1;
    a.W = (a.Win - 1) / a.stride + 1;
    if (a.H <= 0 || a.W <= 0 || a.Win + 1 > 113) return false;
    if (a.x2 && ((a.H2 - 1) / a.stride2 + 1 != a.H || (a.W2 - 1) / a.stride2 + 1 != a.W)) return false;
    if ((long)a.N * a.Hin * a.Win * a.Cin >= (1l << 30) || (long)a.N * a.H * a.W * a.Cout >= (1l << 30)) return false;
    if (a.x2 && (long)a.N * a.H2 * a.W2 * a.Cin2 >= (1l << 30)) return false;
    a.kw = (a.Cin / 32 + kWaves - 1) / kWaves;
    a.kw2 = (a.Cin2 / 32 + kWaves - 1) / kWaves;
    a.pitch = a.Cin * 2 + (a.stride == 1 ? 32 : 16);
    a.pitch2 = a.x2 ? a.Cin2 * 2 + 32 : 0;
    const int P = a.Win + 1;
    // the tallest band of at most 112 output positions (7 MFMA column blocks) whose LDS image fits
    int rows = a.H < 113 / P ? a.H : 113 / P;
    for (; rows >= 1; --rows) {
        const int bands = (a.H + rows - 1) / rows, r = (a.H + bands - 1) / bands;   // balanced
        const int Q = r * P - 1, pt = Q <= 32 ? 2 : Q <= 64 ? 4 : 7;
        if (Q > 112) continue;
        a.bands = bands; a.band_rows = r;
        const int lds = band_lds(a, pt);
        if (lds <= kMaxLds) {
            p.pt = pt; p.lds_bytes = lds;
            break;
        }
    }
    if (rows < 1) return false;
    a.groups = a.Cout / 16;
    a.total = a.groups * a.N * a.bands;
    a.w_bytes = a.Cout * (9 * a.Cin + a.Cin2) * 2;
    a.magic_p = (uint32_t)(((1ull << 32) + P - 1) / P);
    p.grid = (a.total + 7) / 8 * 8;
    return true;
}

#ifdef OG_BAND_STAMPS
unsigned long long *g_band_stamps = nullptr;
int g_band_launch = 0;
#endif

struct BandDesc {
    const void *x, *w_packed;
    const float *bias;
    const void *skip, *x2;
    void *out;
    int N, Hin, Win, Cin, Cout, stride, relu, H2, W2, Cin2, stride2;
};

int fill_layer(const char *name, BandLayer &a, BandPlan &p, const BandDesc &d)
{
    OG_REQUIRE(d.x && d.w_packed && d.bias && d.out, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(d.N > 0 && d.Hin > 0 && d.Win > 0, OG_EINVAL, "%s: bad shape", name);
    a = BandLayer{};
    a.x = (const unsigned short *)d.x; a.w = (const unsigned short *)d.w_packed; a.bias = d.bias; a.skip = (const unsigned short *)d.skip;
    a.x2 = (const unsigned short *)d.x2; a.out = (unsigned short *)d.out;
    a.N = d.N; a.Hin = d.Hin; a.Win = d.Win; a.Cin = d.Cin; a.Cout = d.Cout; a.stride = d.stride; a.relu = d.relu;
    a.H2 = d.H2; a.W2 = d.W2; a.Cin2 = d.x2 ? d.Cin2 : 0; a.stride2 = d.stride2;
    OG_REQUIRE(band_plan(a, p), OG_EUNSUPPORTED,
               "%s: not served (needs 64 <= Cin <= 512 in multiples of 32, Cout %% 16 == 0, stride 1|2, input width <= 112, the "
               "band in 160 KiB of LDS; got %dx%d, %d -> %d, stride %d, projection %d)", name, d.Hin, d.Win, d.Cin, d.Cout, d.stride, a.Cin2);
    return OG_OK;
}

}  // namespace

#ifndef OG_DT_F16
#ifdef OG_BAND_STAMPS
// buf: [launches][4096 workgroups][8] u64; every og_conv_band_* call after this fills the next [4096][8] block
OG_API void og_conv_band_debug_stamps(void *buf) { g_band_stamps = (unsigned long long *)buf; g_band_launch = 0; }
#endif

OG_API int og_conv_band_supported(int N, int Hin, int Win, int Cin, int Cout, int stride, int H2, int W2, int Cin2, int stride2)
{
    if (N <= 0 || Hin <= 0 || Win <= 0) return 0;
    BandLayer a = {};
    BandPlan p;
    a.N = N; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Cout = Cout; a.stride = stride;
    if (Cin2 > 0) { a.x2 = (const unsigned short *)16; a.H2 = H2; a.W2 = W2; a.Cin2 = Cin2; a.stride2 = stride2; }
    if (!band_plan(a, p)) return 0;
    return a.total;
}

OG_API int og_conv_band_pack_w16(const void *w, const void *w2, int Cin, int Cout, int Cin2, void *packed, void *stream)
{
    OG_REQUIRE(w && packed && (Cin2 == 0 || w2), OG_EINVAL, "og_conv_band_pack_w16: null pointer");
    OG_REQUIRE(Cin > 0 && Cin % 32 == 0 && Cout > 0 && Cout % 16 == 0 && Cin2 >= 0 && Cin2 % 32 == 0, OG_EUNSUPPORTED,
               "og_conv_band_pack_w16: Cin, Cin2 must be multiples of 32 and Cout of 16 (got %d + %d -> %d)", Cin, Cin2, Cout);
    const long slots = (long)Cout * (9 * Cin + Cin2) / 8;
    hipLaunchKernelGGL(conv_band_pack_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short *)w, (const unsigned short *)w2, (unsigned short *)packed, Cin, Cout, Cin2);
    OG_LAUNCH_CHECK("og_conv_band_pack_w16");
    return OG_OK;
}
#endif

// kernel variants: PT (column blocks) x chunks per wave (3 = Cin <= 384, 4 = Cin <= 512) x projection chunks per wave (0 / 4)
#define BAND_DISPATCH(KERNEL, pt_, kw_, kw2_)                                                                  \
    do {                                                                                                      \
        if (pt_ == 2) { if (kw_ <= 3) { if (kw2_) KERNEL(2, 3, 4); else KERNEL(2, 3, 0); }                    \
                        else { if (kw2_) KERNEL(2, 4, 4); else KERNEL(2, 4, 0); } }                           \
        else if (pt_ == 4) { if (kw_ <= 3) { if (kw2_) KERNEL(4, 3, 4); else KERNEL(4, 3, 0); }               \
                             else { if (kw2_) KERNEL(4, 4, 4); else KERNEL(4, 4, 0); } }                      \
        else { if (kw_ <= 3) { if (kw2_) KERNEL(7, 3, 4); else KERNEL(7, 3, 0); }                             \
               else { if (kw2_) KERNEL(7, 4, 4); else KERNEL(7, 4, 0); } }                                    \
    } while (0)

OG_API int OG_LP_NAME(og_conv_band)(const void *x, const void *w_packed, const float *bias, const void *skip, const void *x2, void *out,
                                    int N, int Hin, int Win, int Cin, int Cout, int stride, int relu, int H2, int W2, int Cin2,
                                    int stride2, void *stream)
{
    const char *name = OG_LP_STR("og_conv_band");
    const BandDesc d = {x, w_packed, bias, skip, x2, out, N, Hin, Win, Cin, Cout, stride, relu, H2, W2, Cin2, stride2};
    BandLayer a;
    BandPlan p;
    const OgWarm warm = og_take_warm_hint();      // (taken even when the launch is refused: a hint never outlives its call)
    const int rc = fill_layer(name, a, p, d);
    if (rc != OG_OK) return rc;
    a.warm = warm;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *stamps = nullptr;
#ifdef OG_BAND_STAMPS
    if (g_band_stamps) stamps = g_band_stamps + (size_t)(g_band_launch++) * 4096 * 8;
#endif
#define BAND_LAUNCH(PT_, KW_, KW2_)                                                                                   \
    do {                                                                                                              \
        static OgAttrOnce attr_;                                                                                      \
        if (attr_.need())                                                                                             \
            (void)hipFuncSetAttribute((const void *)conv_band_kernel<PT_, KW_, KW2_>,                                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);                           \
        hipLaunchKernelGGL((conv_band_kernel<PT_, KW_, KW2_>), dim3((unsigned)p.grid), dim3(64 * kWaves), p.lds_bytes, st, a, stamps); \
    } while (0)
    BAND_DISPATCH(BAND_LAUNCH, p.pt, a.kw, a.kw2);
#undef BAND_LAUNCH
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}
