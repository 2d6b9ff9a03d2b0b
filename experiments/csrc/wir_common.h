// Helpers shared by the register-resident-weight convolution kernels (sed_conv_wir.hip, sed_conv_w4.hip): the slot swizzle of
// the row ring, the LDS-DMA instruction, the barrier.
#pragma once
#include "conv_common.h"

namespace {

// 16-byte slot XOR of LDS pixel (column col_lds = image column + 1, virtual row v)
template <int W, int SLOTS>
__device__ __forceinline__ int wir_z(int col_lds, int v) {
    if (SLOTS == 16) return (col_lds + (W == 8 ? 8 * (v & 1) : 0)) & 15;
    return ((col_lds >> 1) + (W == 8 ? 4 * (v & 1) : 0)) & 7;
}

__device__ __forceinline__ void wir_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef __attribute__((address_space(3))) char lds_char;

// LDS-DMA of 16 bytes per lane: LDS[lds_dst + 16*lane] <- buffer[voff] (lds_dst wave-uniform).  Inline asm on purpose:
// for the builtin hipcc models the instruction as a pending LDS write and drains the vector-memory counter (vmcnt(0), or a
// count-down over every older operation) in front of the next ds_read of the array -- the row DMAs of the NEXT step and
// the output stores are exactly what must stay in flight across the k loop.  hipcc does not count an asm load: every
// wait for these is the kernel's own counted s_waitcnt (and nothing else in the step loop may load to a register).
__device__ __forceinline__ void wir_dma16(__amdgpu_buffer_rsrc_t srd, const char* lds_dst, unsigned voff) {
    const unsigned dst = (unsigned)(size_t)(lds_char*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(dst), "s"(srd)
                 : "memory");
}

}  // namespace
