// 16-bit activation / weight type of the backbone kernels (csrc/conv3x3.hip, csrc/epilogue.hip, csrc/stem.hip).
// Each of those files is compiled twice: as it stands for bf16 (the default engine dtype), and with -DOG_DT_F16 for fp16,
// the arithmetic of the reference's apex-O2 evaluation (evaluate.py:92,198-201): v_mfma_f32_16x16x32_f16 runs at the
// bf16 rate on gfx950 and carries 3 more mantissa bits.  The fp16 build exports the same entry points with the suffix
// _f16 instead of _bf16; everything that is not 16-bit-type specific is only defined in the bf16 build.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef OG_DT_F16
typedef _Float16 lp8 __attribute__((ext_vector_type(8)));
#define OG_LP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#define OG_LP_NAME(stem) stem##_f16
#define OG_LP_STR(stem) stem "_f16"
__device__ __forceinline__ unsigned short f2lp(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }   // RNE
__device__ __forceinline__ float lp2f(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }
#else
typedef __bf16 lp8 __attribute__((ext_vector_type(8)));
#define OG_LP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define OG_LP_NAME(stem) stem##_bf16
#define OG_LP_STR(stem) stem "_bf16"
__device__ __forceinline__ unsigned short f2lp(float f)   // round-to-nearest-even on the bit pattern; activations are finite
{
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float lp2f(unsigned short u) { return __builtin_bit_cast(float, (uint32_t)u << 16); }
#endif
