// Shared pieces of the split-operand kernels (csrc/sed_conv_x3.hip, csrc/sed_wgrad_x3.hip): 16-bit piece traits, the split, LDS fragment reads.
#pragma once
#include "conv_common.h"

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(2))) _Float16 half2v;

namespace {

// in-kernel phase stamps (make STAMPS=1 in a scratch copy of the tree: tools/x3_stamp.sh); the product build has none
#ifdef SED_STAMPS
constexpr bool kX3Stamps = true;
#else
constexpr bool kX3Stamps = false;
#endif

template <bool HALF> struct X3;
template <> struct X3<false> {
    typedef bf16x8 vec;
    static constexpr float ILS = 1.f;
    static __device__ __forceinline__ f32x16 mfma(const vec& a, const vec& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct X3<true> {
    typedef half8 vec;
    static constexpr float ILS = 1.f / 2048.f;
    static __device__ __forceinline__ f32x16 mfma(const vec& a, const vec& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

// 8 fp32 values -> hi and lo pieces (round to nearest even; a - hi is exact in fp32).  GRADOP: the operand is a gradient (fp16 only):
// pre-scaled by `pre` = 2^e and clamped into fp16's finite range.
template <bool HALF, bool GRADOP>
__device__ __forceinline__ void split8(const float (&v)[8], sed_u32x4& hw, sed_u32x4& lw, float pre) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float a0 = v[2 * k], a1 = v[2 * k + 1];
        if constexpr (HALF) {
            if constexpr (GRADOP) {
                a0 = __builtin_fminf(__builtin_fmaxf(a0 * pre, -60000.f), 60000.f);
                a1 = __builtin_fminf(__builtin_fmaxf(a1 * pre, -60000.f), 60000.f);
            }
            // 5 instructions per pair: v_cvt_pk_f16_f32, 2 x v_fma_mix_f32 (a - hi with hi read as fp16), v_pk_mul_f32, v_cvt_pk_f16_f32
            const f32x2 pr = {a0, a1};
            const half2v h = __builtin_convertvector(pr, half2v);
            const unsigned hb = __builtin_bit_cast(unsigned, h);
            f32x2 d;
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d[0]) : "v"(hb), "v"(a0));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d[1]) : "v"(hb), "v"(a1));
            d *= 2048.f;
            const half2v l = __builtin_convertvector(d, half2v);
            hw[k] = hb;
            lw[k] = __builtin_bit_cast(unsigned, l);
        } else {
            const f32x2 pr = {a0, a1};
            const bf16x2 h = __builtin_convertvector(pr, bf16x2);
            const unsigned hb = __builtin_bit_cast(unsigned, h);
            const float h0 = __builtin_bit_cast(float, hb << 16), h1 = __builtin_bit_cast(float, hb & 0xffff0000u);
            const f32x2 d = {a0 - h0, a1 - h1};
            const bf16x2 l = __builtin_convertvector(d, bf16x2);
            hw[k] = hb;
            lw[k] = __builtin_bit_cast(unsigned, l);
        }
    }
}

typedef unsigned short u16_t;      // an LDS plane element: a bf16 or fp16 bit pattern

template <typename V>
__device__ __forceinline__ V lds_frag(const u16_t* p) {
    return __builtin_bit_cast(V, *reinterpret_cast<const sed_u32x4*>(p));
}
template <typename V>
__device__ __forceinline__ V lds_frag_tr(const u16_t* p0, const u16_t* p1) {
    return __builtin_bit_cast(V, join_tr(ds_read_tr16_b64(reinterpret_cast<const bf16_t*>(p0)), ds_read_tr16_b64(reinterpret_cast<const bf16_t*>(p1))));
}

}  // namespace
