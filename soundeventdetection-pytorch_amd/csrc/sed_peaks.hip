// Box-measured peaks for bench.py's roofline block ("replace with box-measured peaks ... and state both").
//   sed_peak_mfma_bf16   register-fed v_mfma_f32_32x32x16_bf16 loop on pseudo-random operands, one wave per SIMD on every CU
//                        (tools/micro/mfma_lds.hip's first row as a library entry point): what the matrix pipe issues on THIS part
//                        at the clock it holds under matrix load, no LDS / HBM traffic
//   sed_peak_stream_copy float4 grid-stride copy (read n bytes + write n bytes): what HBM delivers to a 1:1 read / write stream
//   sed_peak_stream_read float4 grid-stride read (n bytes, nothing written): the ceiling of a read-dominated kernel
// Both are plain launches on the caller's stream; bench.py brackets them with HIP events before its timed region.
#include "common.h"

namespace {

constexpr int PEAK_MFMA_PER_ITER = 64;      // MFMAs per loop iteration and wave (4 independent accumulators x 16)

__global__ __launch_bounds__(256) void peak_mfma_bf16_kernel(float* __restrict__ sink, int iters) {
    const int tid = threadIdx.x;
    // pseudo-random bf16 operand bits (data-dependent switching power: all-zero operands would let the part clock higher)
    unsigned s = 0x9E3779B9u * (unsigned)(blockIdx.x * 256 + tid + 1);
    bf16x8 a[4], b[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        sed_u32x4 wa, wb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s = s * 1664525u + 1013904223u;
            // two bf16 values in [-2, 2): sign + exponent 0x3f80 / 0x3f00 region, random mantissa
            wa[k] = (s & 0x807f807fu) | 0x3f003f80u;
            s = s * 1664525u + 1013904223u;
            wb[k] = (s & 0x807f807fu) | 0x3f803f00u;
        }
        a[f] = __builtin_bit_cast(bf16x8, wa);
        b[f] = __builtin_bit_cast(bf16x8, wb);
    }
    f32x16 acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[f][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < PEAK_MFMA_PER_ITER / 4; ++u) {
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(f + u) & 3], b[f], acc[f], 0, 0, 0);
        }
        // (random-sign products of magnitude ~1: the accumulators random-walk to ~1e3 over a 50 ms launch -- no rescaling needed)
    }
    float t = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int i = 0; i < 16; ++i) t += acc[f][i];
    if (t == 12345.678f) sink[0] = t;          // never true: keeps the loop alive without a store per thread
}

__global__ __launch_bounds__(256) void peak_stream_copy_kernel(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {      // four 16-byte loads in flight per thread
        const f32x4 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
        dst[i] = v0; dst[i + stride] = v1; dst[i + 2 * stride] = v2; dst[i + 3 * stride] = v3;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void peak_stream_read_kernel(const f32x4* __restrict__ src, float* __restrict__ sink, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const f32x4 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; i < n16; i += stride) a0 += src[i];
    const f32x4 t = a0 + a1 + a2 + a3;
    if (t[0] + t[1] + t[2] + t[3] == 12345.678f) sink[0] = t[0];      // (never true: the loads must not be dropped)
}

}  // namespace

extern "C" int sed_peak_mfma_bf16(int iters, float* sink, double* flops_out, void* stream) {
    SED_REQUIRE(iters > 0 && sink != nullptr, "iters > 0 and a device float to anchor the loop");
    const int cus = sed_device_cu_count();
    SED_REQUIRE(cus > 0, "no device");
    const int grid = cus * 4;                     // 256-thread workgroups: four per CU = four waves per SIMD
    peak_mfma_bf16_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(sink, iters);
    SED_LAUNCH_CHECK();
    if (flops_out) *flops_out = (double)grid * 4.0 * (double)iters * PEAK_MFMA_PER_ITER * (2.0 * 32 * 32 * 16);
    return 0;
}

extern "C" int sed_peak_stream_copy(const void* src, void* dst, size_t bytes, void* stream) {
    SED_REQUIRE(src != nullptr && dst != nullptr && bytes >= 16 && bytes % 16 == 0, "16-byte multiples");
    const int cus = sed_device_cu_count();
    SED_REQUIRE(cus > 0, "no device");
    peak_stream_copy_kernel<<<cus * 8, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const f32x4*>(src), reinterpret_cast<f32x4*>(dst),
                                                                     bytes / 16);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_peak_stream_read(const void* src, size_t bytes, float* sink, void* stream) {
    SED_REQUIRE(src != nullptr && sink != nullptr && bytes >= 16 && bytes % 16 == 0, "16-byte multiples");
    const int cus = sed_device_cu_count();
    SED_REQUIRE(cus > 0, "no device");
    peak_stream_read_kernel<<<cus * 8, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const f32x4*>(src), sink, bytes / 16);
    SED_LAUNCH_CHECK();
    return 0;
}
