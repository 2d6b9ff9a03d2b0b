// Micro-benchmark (GPU box): how many bytes per clock does ONE CU pull through its vector-memory path when every CU streams
//  (a) the same L2-resident 36.8 KB block over and over (the weight chunk of conv_pc_kernel),
//  (b) a private, HBM-resident stream,
// with buffer_load_dwordx4 into registers or with LDS-DMA, 4 or 8 loading waves per CU.
//   hipcc -O3 --offload-arch=gfx950 -o l2bw tools/micro/l2bw.hip && ./l2bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// mode 0: registers (sum kept alive), mode 1: LDS-DMA
template <int MODE, int PER>   // PER loads of 16 B per thread per round
__global__ __launch_bounds__(512) void bw_kernel(const char* __restrict__ src, size_t block_bytes, size_t cu_stride, int rounds, unsigned* out,
                                                 long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const char* base = src + (size_t)blockIdx.x * cu_stride;
    const __amdgpu_buffer_rsrc_t srd = make_srd(base, 0x7fffffff);
    u32x4 acc = {0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    size_t off = 0;
    for (int r = 0; r < rounds; ++r) {
        if (MODE == 0) {
            u32x4 v[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u)
                v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(srd, (unsigned)(off + (size_t)(tid + u * nthr) * 16), 0, 0));
#pragma unroll
            for (int u = 0; u < PER; ++u) acc ^= v[u];
        } else {
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const unsigned lds = (unsigned)(((tid >> 6) * 64 + u * nthr) * 16) & 0xffff;     // wave-uniform base in M0
                const unsigned vo = (unsigned)(off + (size_t)(tid + u * nthr) * 16);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(vo), "s"(__builtin_amdgcn_readfirstlane(lds)), "s"(srd) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        off += (size_t)PER * nthr * 16;
        if (off + (size_t)PER * nthr * 16 > block_bytes) off = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    if (MODE == 1) acc[0] ^= reinterpret_cast<unsigned*>(smem)[tid];
    out[blockIdx.x * nthr + tid] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int PER>
void run(const char* name, const char* src, size_t block_bytes, size_t cu_stride, int nthr, unsigned* out, long long* cyc) {
    const int rounds = 2000, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int it = 0; it < 2; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((bw_kernel<MODE, PER>), dim3(grid), dim3(nthr), 65536, 0, src, block_bytes, cu_stride, rounds, out, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= grid;
    const double bytes = (double)rounds * PER * nthr * 16;
    printf("%-44s threads %4d  per-round %6.1f KB  %8.3f ms  %7.1f GB/s chip  %6.2f B/clk/CU (s_memtime %0.0f cyc, %5.2f B/tick)\n", name, nthr,
           PER * nthr * 16 / 1024.0, ms, bytes * grid / ms / 1e6, bytes / (ms * 1e-3 * 2.4e9), avg, bytes / avg);
}

int main() {
    const size_t total = (size_t)2 << 30;
    char* src;
    unsigned* out;
    long long* cyc;
    hipMalloc(&src, total);
    hipMemset(src, 1, total);
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 256 * 8);
    const size_t chunk = 36864;               // 9 taps x 32 cin x 64 cout x 2 B
    // (a) every CU reads the same 36.8 KB block (L2 hits)
    run<0, 9>("L2-resident 36.8 KB, registers, 4 waves", src, chunk, 0, 256, out, cyc);
    run<0, 9>("L2-resident 36.8 KB, registers, 8 waves", src, chunk * 2, 0, 512, out, cyc);
    run<1, 9>("L2-resident 36.8 KB, LDS-DMA, 4 waves", src, chunk, 0, 256, out, cyc);
    run<1, 9>("L2-resident 36.8 KB, LDS-DMA, 8 waves", src, chunk * 2, 0, 512, out, cyc);
    // (a') a 295 KB block shared by all CUs (all weights of a 128->128 layer)
    run<0, 9>("L2-resident 295 KB, registers, 4 waves", src, chunk * 8, 0, 256, out, cyc);
    run<1, 9>("L2-resident 295 KB, LDS-DMA, 4 waves", src, chunk * 8, 0, 256, out, cyc);
    // (b) private 8 MB stream per CU (HBM)
    run<0, 9>("HBM stream 8 MB/CU, registers, 4 waves", src, (size_t)8 << 20, (size_t)8 << 20, 256, out, cyc);
    run<1, 9>("HBM stream 8 MB/CU, LDS-DMA, 4 waves", src, (size_t)8 << 20, (size_t)8 << 20, 256, out, cyc);
    run<0, 9>("HBM stream 8 MB/CU, registers, 8 waves", src, (size_t)8 << 20, (size_t)8 << 20, 512, out, cyc);
    return 0;
}
