// Stem of Hourglass-104: convolution(7, 3, 128, stride=2) + BN + ReLU (models/hourglass_104.py:283, :16-30) on the
// fp32 NCHW images the reference interface hands over, to bf16 NHWC activations -- the input conversion, the
// convolution and its epilogue in one kernel.
//
// MIOpen runs this layer as an im2col GEMM with K = 147 (3 channels x 49 taps) in 148 us and the bias/ReLU pass over
// its 210 MB output takes another ~85 us.  Here a workgroup owns a 16x16 output tile x all 128 channels:
//   * the 37x38-pixel input tile is read from the three fp32 planes once, converted, and kept in LDS as 4-channel
//     bf16 pixels (8 B, channel 3 = 0);
//   * K is laid out per kernel row as 8 taps x 4 channels = 32 (tap 7 and channel 3 are zero weights), so one
//     v_mfma_f32_16x16x32_bf16 covers one kernel row and a B fragment (16 output pixels x 8 k) is one aligned
//     ds_read_b128: two neighbouring input pixels at column 2*(x + fk), row 2*y + ky;
//   * the whole 128 x 224 weight matrix (56 KB) sits in LDS; 7 k-steps, no pipeline: two workgroups per CU overlap
//     each other's load and compute phases;
//   * epilogue as in the 3x3 kernels: bias + ReLU + one rounding in registers, bf16 tile through LDS, 256-B
//     contiguous stores.
#include "lp_dtype.h"
#include "og_common.h"

namespace {

#ifndef OG_STEM_ABL
#define OG_STEM_ABL 0   // timing experiments (wrong results): 1 no output stores, 2 no input loads, 4 no weight loads, 8 no MFMA
#endif

typedef lp8 bf16x8;   // 8 x 16-bit operands of one MFMA fragment (lp_dtype.h)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

constexpr int kCout = 128, kKRow = 32, kK = 7 * kKRow;          // 224
constexpr int kWPitch = kK * 2 + 16;                            // 464 B per cout row (29 16-B slots: odd)
constexpr int kInW = 38, kInH = 37;                             // input tile incl. halo (+1 pad column)
constexpr int kWBytes = kCout * kWPitch, kInBytes = kInH * kInW * 8;
constexpr int kOPitch = 64 * 2 + 16, kOBytes = 16 * kOPitch;     // per wave: one output row, 16 pixels x 64 couts (+ 16 B pad)
static_assert(2 * (kWBytes + kInBytes + 4 * kOBytes) <= 160 * 1024, "two workgroups per CU");

__device__ __forceinline__ unsigned short f2bf(float f) { return f2lp(f); }   // (bf16, or fp16 in the -DOG_DT_F16 build)

// Persistent form: a workgroup loads the 57 KB weight matrix into LDS ONCE and walks over tiles (grid = 2 workgroups per
// CU); the next tile's input pixels are requested into registers before the current tile's MFMAs and go to LDS behind
// them; results leave straight from the accumulators (a lane holds 4 consecutive couts of a pixel: 8-B stores, 32 B
// contiguous per pixel and instruction, the eight fragments of a pixel complete its 256-B row) -- no output staging, so
// the weights never have to be re-staged.  One tile per workgroup (the first form of this kernel) spent its time on
// exactly that: timing builds without loads / stores / MFMA still took 38 of its 85 us.
__global__ void __launch_bounds__(256, 2)
stem7x7_kernel(const float *__restrict__ img, const unsigned short *__restrict__ wp, const float *__restrict__ bias,
               unsigned short *__restrict__ out, int H, int W, int relu, int n_tiles)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char *const wS = lds, *const inS = lds + kWBytes;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int Ho = H >> 1, Wo = W >> 1, tiles_x = Wo >> 4, tiles_y = Ho >> 4;
    const size_t plane = (size_t)H * W;
    constexpr int kInIter = (kInH * kInW + 255) / 256;   // 6
    constexpr int kWIter = kCout * 28 / 256;             // 14
    static_assert(kCout * 28 % 256 == 0, "weight chunks divide among the threads");

    // every load of a thread is issued before the first one is used (as loops with the LDS store inside, the compiler
    // waited for each load: 20 serialised memory round trips)
    float pin_f[kInIter][3];
    auto request_tile = [&](int t) {   // fp32 pixels of tile t's 37 x 38 input window into registers (zero outside the image: pad 3)
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const float *src = img + (size_t)n * 3 * plane;
        const int iy0 = ty * 32 - 3, ix0 = tx * 32 - 3;
#pragma unroll
        for (int u = 0; u < kInIter; ++u) {
            const int i = tid + 256 * u, r = i / kInW, c = i - r * kInW;
            const int y = iy0 + r, x = ix0 + c;
            const bool ok = i < kInH * kInW && y >= 0 && y < H && x >= 0 && x < W && !(OG_STEM_ABL & 2);
            const size_t o = ok ? (size_t)y * W + x : 0;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) pin_f[u][ch] = ok ? src[ch * plane + o] : 0.f;
        }
    };
    auto stage_tile = [&]() {          // -> 4-channel bf16 pixels in LDS
#pragma unroll
        for (int u = 0; u < kInIter; ++u) {
            const int i = tid + 256 * u;
            const u16x4 px = {f2bf(pin_f[u][0]), f2bf(pin_f[u][1]), f2bf(pin_f[u][2]), 0};
            if (i < kInH * kInW) *reinterpret_cast<u16x4 *>(inS + i * 8) = px;
        }
    };
    int t = blockIdx.x;
    if (t >= n_tiles) return;
    request_tile(t);
    {
        u16x8 wreg[kWIter];
#pragma unroll
        for (int u = 0; u < kWIter; ++u) {
            const int i = tid + 256 * u, r = i / 28, c = i - r * 28;
            if (OG_STEM_ABL & 4) wreg[u] = (u16x8){1, 2, 3, 4, 5, 6, 7, (unsigned short)tid};
            else wreg[u] = *reinterpret_cast<const u16x8 *>(wp + (size_t)r * kK + c * 8);
        }
#pragma unroll
        for (int u = 0; u < kWIter; ++u) {    // 128 rows x 28 16-B chunks -> padded rows
            const int i = tid + 256 * u, r = i / 28, c = i - r * 28;
            *reinterpret_cast<u16x8 *>(wS + r * kWPitch + c * 16) = wreg[u];
        }
    }
    const int fc = lane & 15, fk = lane >> 4;
    unsigned char *const oS = inS + kInBytes + wave * kOBytes;   // this wave's output-row scratch
    const unsigned char *pin = inS + ((2 * (wave * 4)) * kInW + 2 * fc + 2 * fk) * 8;
    const unsigned char *pw = wS + fc * kWPitch + fk * 16;
    f32x4 bv[8];
#pragma unroll
    for (int nn = 0; nn < 8; ++nn) bv[nn] = *reinterpret_cast<const f32x4 *>(bias + nn * 16 + fk * 4);

    for (; t < n_tiles; t += gridDim.x) {
        stage_tile();
        __syncthreads();                       // tile t's pixels (and, the first time, the weights) are in LDS
        const int tn = t + gridDim.x;
        if (tn < n_tiles) request_tile(tn);    // in flight during this tile's MFMAs
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const size_t tile_px = ((size_t)n * Ho + ty * 16) * Wo + tx * 16;
        // 7 k-steps (kernel rows); wave = 4 output rows x 16 columns; the 128 couts in two passes of 64 (64 accumulator
        // registers, not 128: two workgroups per CU need the registers)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 acc[4][4];
#pragma unroll
            for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[nn][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                bf16x8 pf[4], wf[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) pf[m] = *reinterpret_cast<const bf16x8 *>(pin + ((2 * m + ky) * kInW) * 8);
#pragma unroll
                for (int nn = 0; nn < 4; ++nn)
                    wf[nn] = *reinterpret_cast<const bf16x8 *>(pw + (half * 4 + nn) * 16 * kWPitch + ky * 64);
#pragma unroll
                for (int nn = 0; nn < 4; ++nn)
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        if (OG_STEM_ABL & 8) acc[nn][m][0] += (float)wf[nn][0] * (float)pf[m][0];
                        else acc[nn][m] = OG_LP_MFMA(wf[nn], pf[m], acc[nn][m]);
                    }
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {      // bias + ReLU + the one rounding; lane = pixel (row wave * 4 + m, column fc)
                // one output row of the wave (16 pixels x 64 couts = 2 KB) through the wave's own LDS scratch: a lane holds 4
                // consecutive couts of a pixel, the stores want 16 B per lane and 128 B contiguous per pixel (straight from
                // the accumulators -- 8-B stores, 32 B per pixel and instruction -- the layer took 88 us, 68 of them stores)
                __builtin_amdgcn_wave_barrier();   // the previous row's reads are done (same wave: LDS operations are in order)
#pragma unroll
                for (int nn = 0; nn < 4; ++nn) {
                    const f32x4 v = acc[nn][m] + bv[half * 4 + nn];
                    u16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = f2bf(relu ? fmaxf(v[j], 0.f) : v[j]);
                    *reinterpret_cast<u16x4 *>(oS + fc * kOPitch + nn * 32 + fk * 8) = o;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const size_t row_px = tile_px + (size_t)(wave * 4 + m) * Wo;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int pc = lane + 64 * j, px = pc >> 3, ch = pc & 7;
                    const u16x8 v8 = *reinterpret_cast<const u16x8 *>(oS + px * kOPitch + ch * 16);
                    const size_t off = (row_px + px) * kCout + half * 64 + ch * 8;
                    if (!(OG_STEM_ABL & 1) || off == 12345) *reinterpret_cast<u16x8 *>(out + off) = v8;
                }
            }
        }
        __syncthreads();                       // everybody is done reading tile t's pixels
    }
}

}  // namespace

OG_API int OG_LP_NAME(og_stem7x7)(const float *images, const void *w_packed, const float *bias, void *out, int N, int H, int W,
                           int relu, void *stream)
{
    const char *name = OG_LP_STR("og_stem7x7");
    OG_REQUIRE(images && w_packed && bias && out, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && H > 0 && W > 0 && H % 32 == 0 && W % 32 == 0, OG_EINVAL, "%s: H, W must be multiples of 32", name);
    OG_REQUIRE((uintptr_t)w_packed % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)bias % 16 == 0, OG_EINVAL,
               "%s: pointers must be 16-byte aligned", name);
    const long blocks = (long)N * (H / 32) * (W / 32);
    OG_REQUIRE(blocks < (1l << 31), OG_EINVAL, "%s: too many tiles", name);
    constexpr int lds = kWBytes + kInBytes + 4 * kOBytes;
    static OgAttrOnce attr;
    if (attr.need()) (void)hipFuncSetAttribute((const void *)stem7x7_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    // two workgroups per CU (LDS: 70.6 KB each), walking over the tiles
    static const int per_launch = 2 * og_cu_count();
    const unsigned grid = (unsigned)(blocks < per_launch ? blocks : per_launch);
    hipLaunchKernelGGL(stem7x7_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, images,
                       (const unsigned short *)w_packed, bias, (unsigned short *)out, H, W, relu, (int)blocks);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}
