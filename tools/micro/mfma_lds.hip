// Micro-benchmark (GPU box): cycles per v_mfma_f32_32x32x16_bf16 for one wave per SIMD (256-thread workgroups, one per CU)
// when the A operand comes from VGPRs or AGPRs and the B operand from registers or from a ds_read_b128 ring.
//   hipcc -O3 --offload-arch=gfx950 -o mfma_lds tools/micro/mfma_lds.hip && ./mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        static_for<I + 1, N>(fn);
    }
}

// AG: weights pinned in AGPRs; LDSB: B fragments read from LDS (ring of NXF), else register-resident; NW: distinct weight fragments
// IMM: the ring's LDS addresses are ONE per-iteration base VGPR plus compile-time offsets (no vector address arithmetic per read,
// as in the guide's 32.8-cycle loop); otherwise two VALU instructions per read compute them (round 2's form)
template <bool AG, bool LDSB, int NW, int NXF, int STRIDE, int REUSE = 1, int FILL = 0, bool IMM = false>
__global__ __launch_bounds__(256) void k(const bf16x8* __restrict__ wsrc, float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 65536 / 16; i += 256) reinterpret_cast<bf16x8*>(smem)[i] = wsrc[i & 1023];
    bf16x8 w[NW];
#pragma unroll
    for (int f = 0; f < NW; ++f) {
        w[f] = wsrc[(tid + 64 * f) & 4095];
        if (AG) asm volatile("" : "=a"(w[f]) : "0"(w[f]));
    }
    __syncthreads();
    f32x16 acc = {}, acc2 = {};
    bf16x8 breg = wsrc[tid];
    float fl[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) fl[e] = (float)tid * 1e-3f + e;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        int t = tid;
        asm volatile("" : "+v"(t));
        const int base = (t & 63) * STRIDE + ((t >> 6) & 3) * 1024;
        bf16x8 xf[NXF];
        const char* lbase = smem + ((base + it * 64) & 0x0ff0);
        auto ld = [&](int f) -> bf16x8 {
            if constexpr (IMM) return *reinterpret_cast<const bf16x8*>(lbase + (f % 15) * 4096);
            else return *reinterpret_cast<const bf16x8*>(smem + ((base + f * 4096 + it * 64) & 0xfff0));
        };
        if (LDSB) {
#pragma unroll
            for (int f = 0; f < NXF - 1; ++f) xf[f] = ld(f);
        }
        static_for<0, 72 / REUSE>([&](auto fc) __attribute__((always_inline)) {
            constexpr int f = decltype(fc)::value;
            if constexpr (LDSB && f + NXF - 1 < 72 / REUSE) xf[(f + NXF - 1) % NXF] = ld(f + NXF - 1);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[(REUSE * f) % NW], LDSB ? xf[f % NXF] : breg, acc, 0, 0, 0);
            asm volatile("" : "+a"(acc));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < FILL; ++e) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fl[e % 8]) : "v"(fl[(e + 1) % 8]));
            if constexpr (FILL > 0) __builtin_amdgcn_sched_barrier(0);
            if constexpr (REUSE == 2) {
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[(REUSE * f + 1) % NW], LDSB ? xf[f % NXF] : breg, acc2, 0, 0, 0);
                asm volatile("" : "+a"(acc2));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < FILL; ++e) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fl[e % 8]) : "v"(fl[(e + 1) % 8]));
                if constexpr (FILL > 0) __builtin_amdgcn_sched_barrier(0);
            }
        });
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[e] + acc2[e];
#pragma unroll
    for (int e = 0; e < 8; ++e) s += fl[e];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <bool AG, bool LDSB, int NW, int NXF, int STRIDE, int REUSE = 1, int FILL = 0, bool IMM = false>
void run(const char* name, const bf16x8* w, float* out, long long* cyc) {
    const int iters = 400, grid = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<AG, LDSB, NW, NXF, STRIDE, REUSE, FILL, IMM>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    for (int r = 0; r < 2; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<AG, LDSB, NW, NXF, STRIDE, REUSE, FILL, IMM>), dim3(grid), dim3(256), 98304, 0, w, out, cyc, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    (void)hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= grid;
    const double nm = 72.0 * iters;
    printf("%-58s %7.3f ms  %6.1f ns/MFMA  s_memtime ticks/MFMA %6.1f  -> %6.0f TFLOP/s chip\n", name, ms, ms * 1e6 / nm, avg / nm,
           nm * 4 * 256 * 32768.0 / (ms * 1e-3) / 1e12);
}

// v_mfma_f32_16x16x32_bf16 (half the FLOPs of the 32x32x16 instruction, 16 cycles): 144 per barrier = the same FLOPs per iteration.
// LDSB: one IMM-addressed ds_read_b128 per RPM MFMAs (RPM = 2: the LDS bytes per FLOP of "one read per 32x32x16 MFMA")
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool LDSB, int RPM, int NXF>
__global__ __launch_bounds__(256) void k16(const bf16x8* __restrict__ wsrc, float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 65536 / 16; i += 256) reinterpret_cast<bf16x8*>(smem)[i] = wsrc[i & 1023];
    bf16x8 w[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) w[f] = wsrc[(tid + 64 * f) & 4095];
    __syncthreads();
    f32x4 acc[4] = {};
    bf16x8 breg = wsrc[tid];
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        int t = tid;
        asm volatile("" : "+v"(t));
        const int base = (t & 63) * 16 + ((t >> 6) & 3) * 1024;
        bf16x8 xf[NXF];
        const char* lbase = smem + ((base + it * 64) & 0x0ff0);
        auto ld = [&](int f) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(lbase + (f % 15) * 4096); };
        constexpr int NR = 144 / RPM;
        if (LDSB) {
#pragma unroll
            for (int f = 0; f < NXF - 1; ++f) xf[f] = ld(f);
        }
        static_for<0, NR>([&](auto fc) __attribute__((always_inline)) {
            constexpr int f = decltype(fc)::value;
            if constexpr (LDSB && f + NXF - 1 < NR) xf[(f + NXF - 1) % NXF] = ld(f + NXF - 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < RPM; ++q) {
                acc[(f * RPM + q) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[(f * RPM + q) & 7], LDSB ? xf[f % NXF] : breg, acc[(f * RPM + q) & 3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += acc[a][e];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <bool LDSB, int RPM, int NXF>
void run16(const char* name, const bf16x8* w, float* out, long long* cyc) {
    const int iters = 400, grid = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k16<LDSB, RPM, NXF>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    for (int r = 0; r < 2; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k16<LDSB, RPM, NXF>), dim3(grid), dim3(256), 98304, 0, w, out, cyc, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    (void)hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= grid;
    const double nm = 144.0 * iters;
    printf("%-58s %7.3f ms  %6.1f ns/MFMA  s_memtime ticks/MFMA %6.1f  -> %6.0f TFLOP/s chip\n", name, ms, ms * 1e6 / nm, avg / nm,
           nm * 4 * 256 * 16384.0 / (ms * 1e-3) / 1e12);
}

int main() {
    bf16x8* w;
    float* out;
    long long* cyc;
    (void)hipMalloc(&w, 4096 * 16);
    std::vector<unsigned short> hw(4096 * 8);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 23);     // random-ish bf16 around 0.01
    (void)hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    (void)hipMalloc(&out, 256 * 256 * 4);
    (void)hipMalloc(&cyc, 256 * 8);
    run<false, false, 8, 8, 16>("A: VGPR (8 fragments), B register", w, out, cyc);
    run<true, false, 8, 8, 16>("A: AGPR (8 fragments), B register", w, out, cyc);
    run<true, false, 56, 8, 16>("A: AGPR (56 fragments), B register", w, out, cyc);
    run<false, true, 8, 8, 16>("A: VGPR (8), B: ds_read_b128 ring 8, linear 16 B/lane", w, out, cyc);
    run<true, true, 56, 8, 16>("A: AGPR (56), B: ds_read_b128 ring 8, linear 16 B/lane", w, out, cyc);
    run<true, true, 56, 16, 16>("A: AGPR (56), B: ds_read_b128 ring 16, linear", w, out, cyc);
    run<true, true, 56, 8, 256>("A: AGPR (56), B: ds_read_b128 ring 8, stride 256 B/lane", w, out, cyc);
    run<true, true, 56, 8, 16, 2>("A: AGPR (56), B: ds_read_b128 shared by 2 MFMAs", w, out, cyc);
    run<true, true, 56, 8, 16, 1, 2>("A: AGPR, B: ds_read_b128 per MFMA, + 2 v_fma per gap", w, out, cyc);
    run<true, true, 56, 8, 16, 1, 4>("A: AGPR, B: ds_read_b128 per MFMA, + 4 v_fma per gap", w, out, cyc);
    run<true, true, 56, 8, 16, 1, 6>("A: AGPR, B: ds_read_b128 per MFMA, + 6 v_fma per gap", w, out, cyc);
    run<true, true, 56, 8, 16, 2, 4>("A: AGPR, B shared by 2 MFMAs, + 4 v_fma per gap", w, out, cyc);
    run<true, true, 56, 8, 16, 2, 6>("A: AGPR, B shared by 2 MFMAs, + 6 v_fma per gap", w, out, cyc);
    run<true, false, 56, 8, 16, 1, 4>("A: AGPR, B register, + 4 v_fma per gap", w, out, cyc);
    // round 3: the same read-per-MFMA loops with immediate-offset addressing (no VALU per read)
    run<false, true, 8, 8, 16, 1, 0, true>("IMM A: VGPR (8), B: ds_read_b128 ring 8", w, out, cyc);
    run<true, true, 56, 8, 16, 1, 0, true>("IMM A: AGPR (56), B: ds_read_b128 ring 8", w, out, cyc);
    run<true, true, 56, 4, 16, 1, 0, true>("IMM A: AGPR (56), B: ds_read_b128 ring 4", w, out, cyc);
    run<true, true, 56, 3, 16, 1, 0, true>("IMM A: AGPR (56), B: ds_read_b128 ring 3", w, out, cyc);
    run<true, true, 56, 8, 16, 1, 2, true>("IMM A: AGPR, B: ds_read_b128 per MFMA, + 2 v_fma per gap", w, out, cyc);
    run<true, true, 56, 8, 16, 1, 4, true>("IMM A: AGPR, B: ds_read_b128 per MFMA, + 4 v_fma per gap", w, out, cyc);
    run<true, true, 56, 8, 16, 2, 0, true>("IMM A: AGPR, B shared by 2 MFMAs", w, out, cyc);
    run<true, false, 56, 8, 16, 1, 6>("A: AGPR, B register, + 6 v_fma per gap", w, out, cyc);
    // round 3: the 16x16x32 instruction (same FLOPs per barrier)
    run16<false, 1, 4>("16x16x32: A, B registers", w, out, cyc);
    run16<true, 2, 4>("16x16x32: IMM ds_read_b128 per 2 MFMAs (ring 4)", w, out, cyc);
    run16<true, 4, 4>("16x16x32: IMM ds_read_b128 per 4 MFMAs (ring 4)", w, out, cyc);
    run16<true, 1, 6>("16x16x32: IMM ds_read_b128 per MFMA (ring 6)", w, out, cyc);
    return 0;
}
