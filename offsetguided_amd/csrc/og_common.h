// Shared helpers of libog_decoder.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "og_decoder.h"

#define OG_API extern "C" __attribute__((visibility("default")))

void og_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define OG_REQUIRE(cond, code, ...)  \
    do {                             \
        if (!(cond)) {               \
            og_set_error(__VA_ARGS__); \
            return (code);           \
        }                            \
    } while (0)

// Launch-error check only: never synchronises.
#define OG_LAUNCH_CHECK(name)                                                      \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            og_set_error("%s: launch failed: %s", (name), hipGetErrorString(e_)); \
            return OG_EHIP;                                                        \
        }                                                                          \
    } while (0)

// "Warm the next layer's weights" hint (og_conv_next_weights_hint, abi.cpp): set by the caller right before a convolution launch,
// taken (and cleared) by that launch.  The kernel's workgroups touch one 128-byte line each of [ptr, ptr + bytes) at entry -- loads whose
// values are not used: they pull the NEXT layer's weights from HBM into the memory-side cache while THIS layer computes.
struct OgWarm {
    const void *ptr;
    unsigned bytes;
};
OgWarm og_take_warm_hint();

// Device side: thread `gtid` of the launch touches line `gtid` of the region (a launch has more threads than the region has lines: one
// load per thread at most).  The loads are the OLDEST vector-memory operations of their wave: every counted wait behind them covers
// them; the value is kept alive until the end of the kernel by og_warm_sink.  (A plain load: the compiler is free to sink it towards
// that use.  It does not -- in every kernel that calls this the load is the first vector-memory instruction of the ISA, 21 lines below
// the entry (checked in round 6: hipcc --offload-device-only -S) -- and the counted vmcnt waits of the tiled kernels would be off by one
// if it ever moved between their DMA issues: re-check after a compiler change.)
__device__ __forceinline__ unsigned og_warm_touch(const OgWarm &w, unsigned gtid)
{
    unsigned v = 0;
    if (w.ptr && gtid < (w.bytes >> 7)) v = reinterpret_cast<const unsigned *>(w.ptr)[(size_t)gtid * 32];
    return v;
}
__device__ __forceinline__ void og_warm_sink(unsigned v) { asm volatile("" ::"v"(v)); }

// Compute units of the current device (256 on MI355X; 256 when the query fails), asked once per process and device.
static inline int og_cu_count()
{
    static int cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

static inline size_t og_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// hipFuncSetAttribute applies to the CURRENT device: a process that drives several devices must apply it on each one.
// `static OgAttrOnce once; if (once.need()) hipFuncSetAttribute(...)` remembers it per device ordinal (a repeated call
// from a racing thread is harmless).
struct OgAttrOnce {
    unsigned long long done = 0;
    bool need()
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return true;
        const unsigned long long bit = 1ull << (dev & 63);
        if (done & bit) return false;
        done |= bit;
        return true;
    }
};

// XCD-aware work mapping: hardware deals consecutive workgroup ids round-robin over the 8
// XCDs, so ids b and b+8 share an L2.  Remap so that CONSECUTIVE work items (bands of one
// plane, which share halo rows) land on one XCD.  `padded` = grid size (multiple of 8).
__device__ __forceinline__ int og_xcd_remap(int bid, int padded)
{
    return (bid & 7) * (padded >> 3) + (bid >> 3);
}

// Whole-wave shifts by one lane (gfx9 DPP wave_shr / wave_shl); lanes without a source get 0.
__device__ __forceinline__ float og_from_lane_below(float v)  // lane i <- lane i-1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));   // bound_ctrl: lane 0 gets 0, no `old` register to clear
}
__device__ __forceinline__ float og_from_lane_above(float v)  // lane i <- lane i+1
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

__device__ __forceinline__ float og_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// Order-preserving 64-bit key: larger key = earlier in (value desc, index asc) order.
__device__ __forceinline__ uint64_t og_make_key(float v, uint32_t idx)
{
    v = (v == 0.f) ? 0.f : v;  // -0.0 ranks with +0.0
    uint32_t u = __builtin_bit_cast(uint32_t, v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((uint64_t)u << 32) | (uint32_t)~idx;
}
__device__ __forceinline__ float og_key_value(uint64_t k)
{
    uint32_t u = (uint32_t)(k >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __builtin_bit_cast(float, u);
}
__device__ __forceinline__ uint32_t og_key_index(uint64_t k) { return ~(uint32_t)k; }

// Number of keys in keys[0..n) (LDS, 16-byte aligned base) that are greater than `mine`.
// Rank-by-counting loops are LDS-latency bound when written one key per iteration (the compiler
// waits for every ds_read before the compare); reading 8 keys per iteration as four independent
// ds_read_b128 keeps the LDS pipe busy and is ~8x faster.
__device__ __forceinline__ int og_count_greater(const uint64_t *keys, int n, uint64_t mine)
{
    typedef unsigned long long v2u __attribute__((ext_vector_type(2)));
    int rank = 0, j = 0;
    for (; j + 8 <= n; j += 8) {
        const v2u a = *reinterpret_cast<const v2u *>(keys + j);
        const v2u b = *reinterpret_cast<const v2u *>(keys + j + 2);
        const v2u c = *reinterpret_cast<const v2u *>(keys + j + 4);
        const v2u d = *reinterpret_cast<const v2u *>(keys + j + 6);
        rank += (a.x > mine) + (a.y > mine) + (b.x > mine) + (b.y > mine) + (c.x > mine) + (c.y > mine) +
                (d.x > mine) + (d.y > mine);
    }
    for (; j < n; ++j) rank += (keys[j] > mine);
    return rank;
}

// Two ranks in one pass over the keys (the LDS reads are the cost).
__device__ __forceinline__ void og_count_greater2(const uint64_t *keys, int n, uint64_t m1, uint64_t m2, int &r1, int &r2)
{
    typedef unsigned long long v2u __attribute__((ext_vector_type(2)));
    int a1 = 0, a2 = 0, j = 0;
    for (; j + 8 <= n; j += 8) {
        const v2u a = *reinterpret_cast<const v2u *>(keys + j);
        const v2u b = *reinterpret_cast<const v2u *>(keys + j + 2);
        const v2u c = *reinterpret_cast<const v2u *>(keys + j + 4);
        const v2u d = *reinterpret_cast<const v2u *>(keys + j + 6);
        a1 += (a.x > m1) + (a.y > m1) + (b.x > m1) + (b.y > m1) + (c.x > m1) + (c.y > m1) + (d.x > m1) + (d.y > m1);
        a2 += (a.x > m2) + (a.y > m2) + (b.x > m2) + (b.y > m2) + (c.x > m2) + (c.y > m2) + (d.x > m2) + (d.y > m2);
    }
    for (; j < n; ++j) { a1 += (keys[j] > m1); a2 += (keys[j] > m2); }
    r1 = a1;
    r2 = a2;
}
