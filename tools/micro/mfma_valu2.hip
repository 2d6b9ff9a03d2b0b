// Micro-benchmark (GPU box): TWO waves per SIMD, the producer / consumer arrangement of the conv kernels -- wave A of a SIMD issues
// back-to-back v_mfma_f32_32x32x16_bf16 (operands in registers), wave B of the same SIMD issues a stream of other instructions
// (v_fma_f32 / v_pk_fma_f32 / ds_read_b128 / s_ instructions).  Question: how much of wave B's stream executes in the shadow of
// wave A's MFMAs, i.e. do the two waves' times overlap or add?  (512-thread workgroups: waves 0-3 = A, waves 4-7 = B; wave i sits
// on SIMD i % 4.)
//   hipcc -O3 --offload-arch=gfx950 -o mfma_valu2 tools/micro/mfma_valu2.hip && ./mfma_valu2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE: 0 v_fma_f32, 1 v_pk_fma_f32, 2 ds_read_b128, 3 s_add_u32, 4 v_cvt_pk_bf16_f32, 5 v_perm/v_and (integer VALU)
// AON / BON: which roles run (the other waves go straight to the end)
template <int MODE, bool AON, bool BON>
__global__ __launch_bounds__(512) void k2(const bf16x8* __restrict__ wsrc, float* out, long long* cyc, int nmfma, int nb) {
    __shared__ __attribute__((aligned(16))) char smem[32768];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 32768 / 16; i += 512) reinterpret_cast<bf16x8*>(smem)[i] = wsrc[i & 1023];
    __syncthreads();
    float s = 0.f;
    long long t0 = 0, t1 = 0;
    if (wave < 4) {
        if (AON) {
            bf16x8 a = wsrc[tid & 1023], b = wsrc[(tid + 64) & 1023];
            f32x16 acc0 = {}, acc1 = {};
            t0 = __builtin_readcyclecounter();
            for (int i = 0; i < nmfma; i += 8) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
                }
            }
            t1 = __builtin_readcyclecounter();
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
        }
    } else {
        if (BON) {
            float f[8];
            f32x2 p2[4];
            unsigned iv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { f[e] = tid * 1e-3f + e; iv[e] = tid * 77u + e; }
#pragma unroll
            for (int e = 0; e < 4; ++e) p2[e] = {f[e], f[e + 4]};
            bf16x8 q[4];
            unsigned sacc = 0;
            const char* lb = smem + (tid & 63) * 16;
            t0 = __builtin_readcyclecounter();
            for (int i = 0; i < nb; ++i) {
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[u & 7]) : "v"(f[(u + 1) & 7]));
                    else if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p2[u & 3]) : "v"(p2[(u + 1) & 3]));
                    else if (MODE == 2) { q[u & 3] = *reinterpret_cast<const volatile bf16x8*>(lb + (u & 15) * 1024); }
                    else if (MODE == 3) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
                    else if (MODE == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(iv[u & 7]) : "v"(f[u & 7]), "v"(f[(u + 1) & 7]));
                    else asm volatile("v_and_b32 %0, %0, %1" : "+v"(iv[u & 7]) : "v"(iv[(u + 1) & 7]));
                }
                if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            t1 = __builtin_readcyclecounter();
#pragma unroll
            for (int e = 0; e < 8; ++e) s += f[e] + (float)iv[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) s += p2[e][0] + p2[e][1] + (float)q[e][0];
            s += (float)sacc;
        }
    }
    out[blockIdx.x * 512 + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, bool AON, bool BON>
void run(const char* name, const bf16x8* w, float* out, long long* cyc, int nmfma, int nb) {
    const int grid = 256;
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL((k2<MODE, AON, BON>), dim3(grid), dim3(512), 0, 0, w, out, cyc, nmfma, nb);
        (void)hipDeviceSynchronize();
    }
    std::vector<long long> h(grid * 8);
    (void)hipMemcpy(h.data(), cyc, grid * 8 * sizeof(long long), hipMemcpyDeviceToHost);
    double ca = 0, cb = 0;
    for (int b = 0; b < grid; ++b)
        for (int w8 = 0; w8 < 8; ++w8) (w8 < 4 ? ca : cb) += (double)h[b * 8 + w8];
    ca /= grid * 4;
    cb /= grid * 4;
    printf("%-44s A: %7.1f ticks/MFMA   B: %6.2f ticks/instr   (A %9.0f ticks for %d MFMAs, B %9.0f ticks for %d instr)\n", name,
           AON ? ca / nmfma : 0.0, BON ? cb / (32.0 * nb) : 0.0, ca, nmfma, cb, 32 * nb);
}

int main() {
    bf16x8* w;
    float* out;
    long long* cyc;
    (void)hipMalloc(&w, 4096 * 16);
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&cyc, 256 * 8 * 8);
    (void)hipMemset(w, 0x3c, 4096 * 16);
    const int NM = 8000;
    printf("s_memtime ticks are 100 MHz-domain-independent shader-clock counts as used in tools/micro/mfma_lds.hip\n");
    run<0, true, false>("A alone (MFMA, operands in registers)", w, out, cyc, NM, 0);
    run<0, false, true>("B alone: v_fma_f32", w, out, cyc, NM, 2000);
    run<1, false, true>("B alone: v_pk_fma_f32", w, out, cyc, NM, 2000);
    run<2, false, true>("B alone: ds_read_b128", w, out, cyc, NM, 2000);
    run<3, false, true>("B alone: s_add_u32", w, out, cyc, NM, 2000);
    run<4, false, true>("B alone: v_cvt_pk_bf16_f32", w, out, cyc, NM, 2000);
    run<5, false, true>("B alone: v_and_b32", w, out, cyc, NM, 2000);
    // together: B sized so that it runs about as long as A (A alone ~ 32 ticks/MFMA -> 256k ticks)
    run<0, true, true>("A + B v_fma_f32 (B as long as A alone)", w, out, cyc, NM, 2000);
    run<0, true, true>("A + B v_fma_f32 (half)", w, out, cyc, NM, 1000);
    run<0, true, true>("A + B v_fma_f32 (quarter)", w, out, cyc, NM, 500);
    run<1, true, true>("A + B v_pk_fma_f32 (half)", w, out, cyc, NM, 1000);
    run<2, true, true>("A + B ds_read_b128 (half)", w, out, cyc, NM, 1000);
    run<3, true, true>("A + B s_add_u32", w, out, cyc, NM, 2000);
    run<4, true, true>("A + B v_cvt_pk_bf16_f32 (half)", w, out, cyc, NM, 1000);
    run<5, true, true>("A + B v_and_b32 (half)", w, out, cyc, NM, 1000);
    return 0;
}
