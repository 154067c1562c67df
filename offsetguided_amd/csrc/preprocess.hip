// Input side of evaluate.py on the device: RescaleLongAbsolute -> CenterPad -> ToTensor -> Normalize
// (evaluate.py:157-168; transforms/scale.py:14-31,75-98; transforms/pad.py:35-62) for a uint8 HWC image already in HBM,
// as ONE pass that writes the fp32 NCHW network input -- the resized uint8 image is never stored.
//
// The rescale is cv2.resize(..., INTER_CUBIC) in the reference.  cv2 is a third-party dependency that is absent from the
// build container (opencv-python==3.4.5.20, requirment.txt:91), so this kernel follows OpenCV's PUBLISHED algorithm for
// 8-bit images (resize.cpp: fixed-point taps round(w * 2048), int horizontal pass over clamped taps, vertical pass,
// (sum + 2^21) >> 22, saturate) and is pinned bit-exactly to oracle/og_oracle.c:ogo_resize_cubic_u8, the same restatement
// on the CPU -- parity with cv2 itself is UNPINNED (no golden vector can be generated here).
// One thread per output pixel: 16 taps x 3 channels from L2 (the source image is read ~once from HBM), coefficients
// recomputed per thread in the float arithmetic of interpolateCubic (explicit operation order, -ffp-contract=off).
#include <math.h>

#include "og_common.h"

namespace {

struct Taps {
    int i0;       // index of the first tap (s - 1), unclamped
    int t[4];     // fixed-point weights, sum ~ 2048
};

__device__ __forceinline__ Taps cubic_taps(int d, double scale)
{
    const float f = (float)((d + 0.5) * scale - 0.5);
    const int s = (int)floorf(f);
    const float x = f - (float)s;
    const float A = -0.75f;
    float c[4];
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
    Taps r;
    r.i0 = s - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.t[k] = min(max(__float2int_rn(c[k] * 2048.f), -32768), 32767);   // saturate_cast<short>
    return r;
}

// resized pixel (dy, dx) of an (h, w, 3) uint8 image scaled to (nh, nw): three channels
__device__ __forceinline__ void resized_px(const unsigned char *__restrict__ src, int h, int w, double sy, double sx, int dy, int dx,
                                           int (&out)[3])
{
    const Taps ty = cubic_taps(dy, sy), tx = cubic_taps(dx, sx);
    int xi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xi[k] = min(max(tx.i0 + k, 0), w - 1) * 3;
    int acc[3] = {0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned char *row = src + (size_t)min(max(ty.i0 + r, 0), h - 1) * w * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int hs = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) hs += (int)row[xi[k] + c] * tx.t[k];
            acc[c] += hs * ty.t[r];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = min(max((acc[c] + (1 << 21)) >> 22, 0), 255);
}

__global__ void __launch_bounds__(256)
resize_cubic_kernel(const unsigned char *__restrict__ src, int h, int w, unsigned char *__restrict__ dst, int nh, int nw,
                    double sy, double sx)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)nh * nw) return;
    int v[3];
    resized_px(src, h, w, sy, sx, (int)(i / nw), (int)(i % nw), v);
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[i * 3 + c] = (unsigned char)v[c];
}

// encoder/heatmap.py:56-60, encoder/offset.py:46-50: one channel of the same resize, / 255 > 0.7 (<=> value >= 179)
__global__ void __launch_bounds__(256)
shrink_mask_kernel(const unsigned char *__restrict__ mask, int N, int h, int w, unsigned char *__restrict__ out, int nh, int nw,
                   double sy, double sx)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * nh * nw) return;
    const int n = (int)(i / ((long)nh * nw)), r = (int)(i % ((long)nh * nw));
    const int dy = r / nw, dx = r % nw;
    const unsigned char *src = mask + (size_t)n * h * w;
    const Taps ty = cubic_taps(dy, sy), tx = cubic_taps(dx, sx);
    int acc = 0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const unsigned char *row = src + (size_t)min(max(ty.i0 + rr, 0), h - 1) * w;
        int hs = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) hs += (int)row[min(max(tx.i0 + k, 0), w - 1)] * tx.t[k];
        acc += hs * ty.t[rr];
    }
    const int v = min(max((acc + (1 << 21)) >> 22, 0), 255);
    out[i] = (float)v / 255.f > 0.7f ? 1 : 0;
}

struct PrepArgs {
    float mean[3], stdv[3], fill[3];
};

// taps of one output pixel from an LDS image of the source footprint: rows y0 .. , columns x0 .. (already clamped to the image
// when it was staged), `lw` pixels per LDS row -- the integer arithmetic of resized_px, term for term
__device__ __forceinline__ void resized_px_lds(const unsigned char *img, int lw, int y0, int x0, const Taps &ty, const Taps &tx, int (&out)[3])
{
    int acc[3] = {0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned char *row = img + ((ty.i0 + r - y0) * lw + (tx.i0 - x0)) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int hs = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) hs += (int)row[k * 3 + c] * tx.t[k];
            acc[c] += hs * ty.t[r];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = min(max((acc[c] + (1 << 21)) >> 22, 0), 255);
}

// One workgroup per 64 x 4 tile of the padded output (a wave writes 64 consecutive fp32 per channel row).  The source
// footprint of the tile -- (4 sy + 5) rows x (64 sx + 5) pixels, border pixels replicated as the clamped taps of resized_px
// read them -- is staged in LDS with coalesced loads; the 48 tap bytes of a pixel then come from LDS.  (One thread per pixel
// reading its taps from global memory took 1.3-2.5 ms per image: 48 one-byte gathers per thread; the harness of bench.py ran
// at a fifth of the engine's rate.)  Footprints beyond kPrepLds bytes (scale factors above ~4) take the direct path.
constexpr int kPrepTW = 64, kPrepTH = 4, kPrepLds = 24 * 1024;

__device__ __forceinline__ void
rescale_pad_normalize_tile(unsigned char *img, const unsigned char *__restrict__ src, int h, int w, int nh, int nw, double sy, double sx,
                           int left, int top, int TH, int TW, const PrepArgs &a, float *__restrict__ out, int tiles_x, int use_lds)
{
    const int tile_x = blockIdx.x % tiles_x, tile_y = blockIdx.x / tiles_x;
    const int X = tile_x * kPrepTW + (threadIdx.x & 63), Y = tile_y * kPrepTH + (threadIdx.x >> 6);
    const int x = X - left, y = Y - top;
    // resized-image range of the tile and its source footprint (uniform over the workgroup)
    const int xa = max(tile_x * kPrepTW - left, 0), xb = min(tile_x * kPrepTW + kPrepTW - 1 - left, nw - 1);
    const int ya = max(tile_y * kPrepTH - top, 0), yb = min(tile_y * kPrepTH + kPrepTH - 1 - top, nh - 1);
    const bool any = xa <= xb && ya <= yb;
    int x0 = 0, y0 = 0, lw = 0;
    if (any && use_lds) {
        x0 = cubic_taps(xa, sx).i0;
        y0 = cubic_taps(ya, sy).i0;
        const int x1 = cubic_taps(xb, sx).i0 + 3, y1 = cubic_taps(yb, sy).i0 + 3;
        lw = x1 - x0 + 1;
        const int lh = y1 - y0 + 1;
        // rows that lie in the image are copied as runs of bytes (coalesced), the replicated border pixels one by one
        for (int i = threadIdx.x; i < lh * lw * 3; i += 256) {
            const int r = i / (lw * 3), rest = i - r * (lw * 3), cpx = rest / 3, ch = rest - cpx * 3;
            const int gy = min(max(y0 + r, 0), h - 1), gx = min(max(x0 + cpx, 0), w - 1);
            img[i] = src[((size_t)gy * w + gx) * 3 + ch];
        }
    }
    __syncthreads();
    if (X >= TW || Y >= TH) return;
    float v[3] = {a.fill[0], a.fill[1], a.fill[2]};
    if (x >= 0 && x < nw && y >= 0 && y < nh) {
        int p[3];
        if (use_lds) resized_px_lds(img, lw, y0, x0, cubic_taps(y, sy), cubic_taps(x, sx), p);
        else resized_px(src, h, w, sy, sx, y, x, p);
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (float)p[c];
    }
    const size_t i = (size_t)Y * TW + X;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(size_t)c * TH * TW + i] = (v[c] / 255.f - a.mean[c]) / a.stdv[c];   // ToTensor, Normalize
}

__global__ void __launch_bounds__(256)
rescale_pad_normalize_kernel(const unsigned char *__restrict__ src, int h, int w, int nh, int nw, double sy, double sx, int left,
                             int top, int TH, int TW, PrepArgs a, float *__restrict__ out, int tiles_x, int use_lds)
{
    __shared__ unsigned char img[kPrepLds];
    rescale_pad_normalize_tile(img, src, h, w, nh, nw, sy, sx, left, top, TH, TW, a, out, tiles_x, use_lds);
}

// A whole batch in ONE launch (evaluate.py:157-182 hands the network a batch; the harness used to launch once per image):
// blockIdx.y = image, its geometry from a by-value descriptor table in the kernel arguments (no device-side descriptor upload).
constexpr int kPrepBatchMax = 64;
struct PrepBatch {
    long off[kPrepBatchMax];                      // byte offset of the image's (h, w, 3) pixels in the packed uint8 buffer
    int h[kPrepBatchMax], w[kPrepBatchMax], nh[kPrepBatchMax], nw[kPrepBatchMax];
};

__global__ void __launch_bounds__(256)
rescale_pad_normalize_batch_kernel(const unsigned char *__restrict__ raw, PrepBatch b, int corner_pad, int TH, int TW, PrepArgs a,
                                   float *__restrict__ out, int tiles_x)
{
    __shared__ unsigned char img[kPrepLds];
    const int n = blockIdx.y;
    const int h = b.h[n], w = b.w[n], nh = b.nh[n], nw = b.nw[n];
    // the same expressions as og_rescale_pad_normalize_u8 evaluates on the host: identical scales, paddings and LDS decision
    const double sy = (double)h / nh, sx = (double)w / nw;
    const int left = corner_pad ? 0 : (int)((TW - nw) / 2.0), top = corner_pad ? 0 : (int)((TH - nh) / 2.0);
    const long fw = (long)(kPrepTW * sx) + 8, fh = (long)(kPrepTH * sy) + 8;
    const int use_lds = fw * fh * 3 <= kPrepLds;
    rescale_pad_normalize_tile(img, raw + b.off[n], h, w, nh, nw, sy, sx, left, top, TH, TW, a, out + (size_t)n * 3 * TH * TW, tiles_x,
                               use_lds);
}

}  // namespace

OG_API int og_resize_cubic_u8(const unsigned char *src, int h, int w, unsigned char *dst, int new_h, int new_w, void *stream)
{
    const char *name = "og_resize_cubic_u8";
    OG_REQUIRE(src && dst, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(h > 0 && w > 0 && new_h > 0 && new_w > 0 && (long)h * w < (1l << 28) && (long)new_h * new_w < (1l << 28), OG_EINVAL,
               "%s: bad shape", name);
    const long total = (long)new_h * new_w;
    hipLaunchKernelGGL(resize_cubic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, h, w, dst,
                       new_h, new_w, (double)h / new_h, (double)w / new_w);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

OG_API int og_shrink_mask_miss_u8(const unsigned char *mask, int N, int h, int w, int stride, unsigned char *out, void *stream)
{
    const char *name = "og_shrink_mask_miss_u8";
    OG_REQUIRE(mask && out, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(N > 0 && h > 0 && w > 0 && stride > 0 && (long)N * h * w < (1l << 30), OG_EINVAL, "%s: bad shape", name);
    // cv2.resize(dsize = (0, 0), fx = fy = 1 / stride): dsize = round(size * f), coordinates scaled by 1 / f = stride
    const int nh = (int)lrint((double)h / stride), nw = (int)lrint((double)w / stride);
    OG_REQUIRE(nh > 0 && nw > 0, OG_EINVAL, "%s: stride larger than the mask", name);
    const long total = (long)N * nh * nw;
    hipLaunchKernelGGL(shrink_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mask, N, h, w, out,
                       nh, nw, (double)stride, (double)stride);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

OG_API int og_rescale_pad_normalize_u8(const unsigned char *img, int h, int w, int new_h, int new_w, int target_h, int target_w,
                                       int corner_pad, const float *mean3, const float *std3, const float *fill3, float *out,
                                       int *ltrb, void *stream)
{
    const char *name = "og_rescale_pad_normalize_u8";
    OG_REQUIRE(img && mean3 && std3 && fill3 && out, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(h > 0 && w > 0 && new_h > 0 && new_w > 0 && (long)h * w < (1l << 28), OG_EINVAL, "%s: bad shape", name);
    OG_REQUIRE(target_h >= new_h && target_w >= new_w, OG_EINVAL, "%s: the rescaled image must fit the target", name);
    // CenterPad (transforms/pad.py:43-55): left = int((T - w) / 2.0), top likewise, the rest goes right / down;
    // RightDownPad (transforms/pad.py:95-118, the --fixed-height chain of evaluate.py:150-156): everything goes right / down
    const int left = corner_pad ? 0 : (int)((target_w - new_w) / 2.0), top = corner_pad ? 0 : (int)((target_h - new_h) / 2.0);
    if (ltrb) { ltrb[0] = left; ltrb[1] = top; ltrb[2] = target_w - new_w - left; ltrb[3] = target_h - new_h - top; }
    PrepArgs a;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; a.fill[c] = fill3[c]; }
    const double sy = (double)h / new_h, sx = (double)w / new_w;
    const int tiles_x = (target_w + kPrepTW - 1) / kPrepTW, tiles_y = (target_h + kPrepTH - 1) / kPrepTH;
    // the tile's source footprint must fit the LDS image (the tap positions move by at most ceil(scale) per output pixel)
    const long fw = (long)(kPrepTW * sx) + 8, fh = (long)(kPrepTH * sy) + 8;
    const int use_lds = fw * fh * 3 <= kPrepLds;
    hipLaunchKernelGGL(rescale_pad_normalize_kernel, dim3((unsigned)(tiles_x * tiles_y)), dim3(256), 0, (hipStream_t)stream, img,
                       h, w, new_h, new_w, sy, sx, left, top, target_h, target_w, a, out, tiles_x, use_lds);
    OG_LAUNCH_CHECK(name);
    return OG_OK;
}

OG_API int og_rescale_pad_normalize_batch_u8(const unsigned char *raw, const long *offsets, const int *hw4, int n, int target_h,
                                             int target_w, int corner_pad, const float *mean3, const float *std3, const float *fill3,
                                             float *out, int *ltrb, void *stream)
{
    const char *name = "og_rescale_pad_normalize_batch_u8";
    OG_REQUIRE(raw && offsets && hw4 && mean3 && std3 && fill3 && out, OG_EINVAL, "%s: null pointer", name);
    OG_REQUIRE(n > 0 && target_h > 0 && target_w > 0, OG_EINVAL, "%s: bad shape", name);
    PrepArgs a;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; a.fill[c] = fill3[c]; }
    for (int i = 0; i < n; ++i) {
        const int h = hw4[i * 4], w = hw4[i * 4 + 1], new_h = hw4[i * 4 + 2], new_w = hw4[i * 4 + 3];
        OG_REQUIRE(h > 0 && w > 0 && new_h > 0 && new_w > 0 && (long)h * w < (1l << 28) && offsets[i] >= 0, OG_EINVAL,
                   "%s: image %d: bad shape", name, i);
        OG_REQUIRE(target_h >= new_h && target_w >= new_w, OG_EINVAL, "%s: image %d: the rescaled image must fit the target", name, i);
        if (ltrb) {
            const int left = corner_pad ? 0 : (int)((target_w - new_w) / 2.0), top = corner_pad ? 0 : (int)((target_h - new_h) / 2.0);
            ltrb[i * 4] = left; ltrb[i * 4 + 1] = top; ltrb[i * 4 + 2] = target_w - new_w - left; ltrb[i * 4 + 3] = target_h - new_h - top;
        }
    }
    const int tiles_x = (target_w + kPrepTW - 1) / kPrepTW, tiles_y = (target_h + kPrepTH - 1) / kPrepTH;
    for (int first = 0; first < n; first += kPrepBatchMax) {          // at most kPrepBatchMax descriptors ride in one launch's arguments
        const int m = n - first < kPrepBatchMax ? n - first : kPrepBatchMax;
        PrepBatch b;
        for (int i = 0; i < kPrepBatchMax; ++i) {
            const int j = first + (i < m ? i : 0);
            b.off[i] = offsets[j]; b.h[i] = hw4[j * 4]; b.w[i] = hw4[j * 4 + 1]; b.nh[i] = hw4[j * 4 + 2]; b.nw[i] = hw4[j * 4 + 3];
        }
        hipLaunchKernelGGL(rescale_pad_normalize_batch_kernel, dim3((unsigned)(tiles_x * tiles_y), (unsigned)m), dim3(256), 0,
                           (hipStream_t)stream, raw, b, corner_pad, target_h, target_w, a, out + (size_t)first * 3 * target_h * target_w,
                           tiles_x);
        OG_LAUNCH_CHECK(name);
    }
    return OG_OK;
}
