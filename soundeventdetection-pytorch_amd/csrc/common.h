// Shared device/host helpers for libsed_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/sed_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// ---- error plumbing ---------------------------------------------------------------------------
void sed_set_error(const std::string& s);
#define SED_REQUIRE(cond, msg)                                                             \
    do {                                                                                   \
        if (!(cond)) {                                                                     \
            sed_set_error(std::string(__func__) + ": " + (msg) + " [" #cond "]");          \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)
#define SED_LAUNCH_CHECK()                                                                 \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess) {                                                            \
            sed_set_error(std::string(__func__) + ": launch failed: " + hipGetErrorString(e_)); \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)

// ---- configuration knobs and per-device launch attributes (no hidden per-process state) ---------------------
// SED_* tuning / A-B knobs are read from the environment ONCE per name and cached (sed_config_reload() -- a test hook
// exported from the library -- drops the cache so that a test can flip a knob inside one process).
const char* sed_getenv(const char* name);
#define SED_MAX_DEVICES 64
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute of a kernel: cached per (kernel, device), so a
// process that drives several GPUs sets it on each of them.
template <auto Kernel>
static inline int sed_set_max_lds(size_t lds) {
    static size_t done[SED_MAX_DEVICES] = {};
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SED_MAX_DEVICES) dev = -1;
    if (dev >= 0 && lds <= done[dev]) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { sed_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return 3; }
    if (dev >= 0) done[dev] = lds;
    return 0;
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }

// ---- MFMA element traits (32x32 tiles: bf16 k=16 per instruction, fp32 k=2) -----------------------
template <typename T> struct EL;
template <> struct EL<float> {
    static constexpr int KR = 1;      // consecutive k per lane in a fragment
    static constexpr int KSTEP = 2;   // k per MFMA
    typedef float frag_t;
};
template <> struct EL<bf16_t> {
    static constexpr int KR = 8;
    static constexpr int KSTEP = 16;
    typedef bf16x8 frag_t;
};

__device__ __forceinline__ f32x16 mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma(const float& a, const float& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}


// ---- element helpers --------------------------------------------------------------------------
__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float x) { return (bf16_t)x; }

// 8 consecutive elements <-> 8 floats
template <typename T> struct Vec8;
template <> struct Vec8<float> {
    f32x4 a, b;
    __device__ __forceinline__ float get(int i) const { return i < 4 ? a[i] : b[i - 4]; }
};
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
    bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = a;
}
// relu(scale*x + shift) of 8 bf16 elements -> 8 bf16 elements (the loader waves' BatchNorm + ReLU prologue): eight scalar v_fma_f32,
// four v_cvt_pk_bf16_f32 and the ReLU as four v_pk_max_i16 on the bf16 bit patterns (a negative float is a negative int16; rounding to
// bf16 and the clamp at zero commute, -0 becomes +0): the same bits as fmaxf(0, fmaf(x, scale, shift)) rounded to bf16.
// The FMAs are deliberately NOT packed: with v_pk_fma_f32 (what hipcc's SLP vectorizer makes of adjacent fmaf pairs) the 64 -> 64 forward
// launch took 0.256 instead of 0.226 ms although it has fewer instructions -- packed fp32 arithmetic in a loader wave costs its SIMD's
// MFMA wave more than two scalar instructions do (round 3, tools/ab_build.sh / tools/ab_flags.sh; the files whose loader waves sit beside
// MFMA waves are compiled with -fno-slp-vectorize, csrc/Makefile).  keep = false: zeros (rows outside the image).
typedef __attribute__((ext_vector_type(2))) short sed_i16x2;
typedef __attribute__((ext_vector_type(4))) unsigned sed_u32x4;
__device__ __forceinline__ bf16x8 bnrelu8_bf16(const bf16x8& x, const f32x4& s0, const f32x4& s1, const f32x4& h0, const f32x4& h1,
                                               bool keep = true) {
    const sed_u32x4 xw = __builtin_bit_cast(sed_u32x4, x);
    sed_u32x4 ow;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float x0 = __builtin_bit_cast(float, xw[k] << 16), x1 = __builtin_bit_cast(float, xw[k] & 0xffff0000u);
        const float c0 = k < 2 ? s0[2 * k] : s1[2 * k - 4], c1 = k < 2 ? s0[2 * k + 1] : s1[2 * k - 3];
        const float d0 = k < 2 ? h0[2 * k] : h1[2 * k - 4], d1 = k < 2 ? h0[2 * k + 1] : h1[2 * k - 3];
        float r0, r1;       // (asm: also in translation units that keep the SLP vectorizer)
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(x0), "v"(c0), "v"(d0));
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(x1), "v"(c1), "v"(d1));
#if defined(SED_RELU_I16) && SED_RELU_I16 == 0        // (A/B builds: ReLU as two v_max_f32 before the conversion)
        const f32x2 r = {fmaxf(r0, 0.f), fmaxf(r1, 0.f)};
        const bf16x2 rb = __builtin_convertvector(r, bf16x2);
        ow[k] = keep ? __builtin_bit_cast(unsigned, rb) : 0u;
#else
        const f32x2 r = {r0, r1};
        const bf16x2 rb = __builtin_convertvector(r, bf16x2);          // one v_cvt_pk_bf16_f32 (element-wise casts cost two + a v_perm)
        sed_i16x2 ri = __builtin_bit_cast(sed_i16x2, rb);
        ri = __builtin_elementwise_max(ri, (sed_i16x2){0, 0});
        ow[k] = keep ? __builtin_bit_cast(unsigned, ri) : 0u;
#endif
    }
    return __builtin_bit_cast(bf16x8, ow);
}
template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i];
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
    bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
    f32x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = v[i];
    *reinterpret_cast<f32x4*>(p) = a;
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    bf16x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x4*>(p) = a;
}

// ---- wave64 reductions ------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over each row of 16 lanes; every lane of the row ends with the row total
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    return v;
}
// full wave sum (all 64 lanes), result valid in every lane
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    const int iv = __builtin_bit_cast(int, v);   // readlane is an int builtin: bit-cast, never convert
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) +
                    __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) +
                    __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return a + b;
}

// XCD-aware remap of a 1-D block id (bijective for any grid size; guide T1): blocks that share an
// XCD (id % 8 equal) get a contiguous chunk of the logical index space.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, rmd = nblk & 7, xcd = bid & 7;
    const unsigned base = xcd < rmd ? xcd * (q + 1) : rmd * (q + 1) + (xcd - rmd) * q;
    return base + (bid >> 3);
}
