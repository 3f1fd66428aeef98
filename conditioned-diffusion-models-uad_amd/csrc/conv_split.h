// Shared pieces of the split-operand convolution kernels (conv_x6.hip, conv_ov.hip): vector types, the MFMA wrappers per split
// family, the SiLU used while staging and the fp32 -> 16-bit-terms split. See conv_x6.hip for the arithmetic.
#pragma once
#include <hip/hip_runtime.h>

namespace cddpm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int NS> struct SplitT;
template <> struct SplitT<3> {
    typedef bf16x8 v8; typedef bf16x4 v4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct SplitT<2> {
    typedef f16x8 v8; typedef f16x4 v4;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

__device__ __forceinline__ float silu_x6(float v) {
    // identical evaluation to conv_mfma.hip::silu_f (split-product exp2, ~1.5 ulp)
    const float t = fminf(-v * 1.44269502162933349609375f, 126.0f);
    float tl = __builtin_fmaf(-v, 1.44269502162933349609375f, -t);
    tl = __builtin_fmaf(-v, 1.925963033500011e-08f, tl);
    tl = (t < 126.0f) ? tl : 0.0f;
    float e = __builtin_amdgcn_exp2f(t);
    e = __builtin_fmaf(e, tl * 0.693147180559945f, e);
    return v * __builtin_amdgcn_rcpf(1.0f + e);
}

// the same on four values, written on vectors so that the multiplies / fmas / adds issue as packed (v_pk_*_f32) instructions:
// 26 VALU instructions per quad instead of 46. Identical results for x > -87.3; below, where the exponent clamps, the
// correction term is left to run (it only pushes e further towards +inf and the result towards the limit 0).
__device__ __forceinline__ v4f silu_x6_v4(const v4f v) {
    const v4f nl2e = {-1.44269502162933349609375f, -1.44269502162933349609375f, -1.44269502162933349609375f, -1.44269502162933349609375f};
    const v4f nl2e_lo = {-1.925963033500011e-08f, -1.925963033500011e-08f, -1.925963033500011e-08f, -1.925963033500011e-08f};
    const v4f lim = {126.0f, 126.0f, 126.0f, 126.0f};
    const v4f ln2 = {0.693147180559945f, 0.693147180559945f, 0.693147180559945f, 0.693147180559945f};
    const v4f one = {1.0f, 1.0f, 1.0f, 1.0f};
    const v4f t = __builtin_elementwise_min(v * nl2e, lim);
    v4f tl = __builtin_elementwise_fma(v, nl2e, -t);
    tl = __builtin_elementwise_fma(v, nl2e_lo, tl);
    v4f e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y), __builtin_amdgcn_exp2f(t.z), __builtin_amdgcn_exp2f(t.w)};
    e = __builtin_elementwise_fma(e, tl * ln2, e);
    const v4f d = one + e;
    const v4f r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z), __builtin_amdgcn_rcpf(d.w)};
    return v * r;
}

// split of four fp32 values into NS 16-bit quads (8 B each): t[0] = cvt(v), t[1] = cvt(v - t[0]), ...
template <int NS>
__device__ __forceinline__ void split_x4(const v4f v, typename SplitT<NS>::v4 (&t)[NS]) {
    typedef typename SplitT<NS>::v4 q4;
    v4f r = v;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        t[s] = __builtin_convertvector(r, q4);
        r = r - __builtin_convertvector(t[s], v4f);
    }
}

}  // namespace cddpm
