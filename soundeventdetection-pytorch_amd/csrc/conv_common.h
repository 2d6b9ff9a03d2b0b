// Device helpers shared by the convolution kernels (sed_conv.hip, sed_wgrad.hip): LDS swizzle, 16-byte raw
// items, buffer-resource addressing, the tile-invariant halo staging plan and the transpose-read fragments.
#pragma once
#include "common.h"

// element-index XOR applied inside the 32-channel vector of LDS pixel column `col`
template <typename T> __device__ __forceinline__ int swz(int col);
template <> __device__ __forceinline__ int swz<bf16_t>(int col) { return ((col >> 2) & 3) << 3; }
template <> __device__ __forceinline__ int swz<float>(int col) { return col & 31; }

// -------------------------------------------------------------------------------------------------
// Halo-tile staging shared by the forward/data-gradient and the weight-gradient kernels.
// Phase 1 issues EVERY global load of the thread back to back (out-of-image items read a clamped,
// always-valid address and are zeroed afterwards), phase 2 applies the prologue and writes LDS, so a
// thread has all its loads in flight at once instead of one load per branch.
// -------------------------------------------------------------------------------------------------
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { bf16x8 v; };
template <> struct Raw8<float> { f32x4 a, b; };
template <typename T> __device__ __forceinline__ Raw8<T> raw_load8(const T* p);
template <> __device__ __forceinline__ Raw8<bf16_t> raw_load8<bf16_t>(const bf16_t* p) {
    Raw8<bf16_t> r; r.v = *reinterpret_cast<const bf16x8*>(p); return r;
}
template <> __device__ __forceinline__ Raw8<float> raw_load8<float>(const float* p) {
    Raw8<float> r;
    r.a = *reinterpret_cast<const f32x4*>(p);
    r.b = *reinterpret_cast<const f32x4*>(p + 4);
    return r;
}
__device__ __forceinline__ void raw_to_f(const Raw8<bf16_t>& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)r.v[i];
}
__device__ __forceinline__ void raw_to_f(const Raw8<float>& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = r.a[i]; v[4 + i] = r.b[i]; }
}

// -------------------------------------------------------------------------------------------------
// Lean staging ("VALU diet").  The address / bounds arithmetic of the generic helpers above costs
// ~25 VALU per 16-byte load, which made the low-channel layers VALU-bound.  Here:
//   * every image gets its own buffer resource descriptor (base = image start, num_records = image
//     bytes): halo rows above/below the image are out of range, so the hardware returns 0 for the
//     loads and drops the stores -- no row predicates;
//   * the byte offset of each of a thread's items relative to the tile's first halo pixel is tile
//     invariant: computed ONCE (left/right padding columns get an offset that is always out of range);
//     per tile a single scalar is added;
//   * the LDS destination offsets are tile invariant too;
//   * without a prologue the 16 bytes go from the load to the LDS write untouched.
// -------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#define SED_OOB 0x80000000u

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes),
                                             0x00020000);
}
template <typename T> __device__ __forceinline__ Raw8<T> buf_load8(__amdgpu_buffer_rsrc_t r, unsigned voff);
template <> __device__ __forceinline__ Raw8<bf16_t> buf_load8<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    Raw8<bf16_t> o;
    o.v = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
    return o;
}
template <> __device__ __forceinline__ Raw8<float> buf_load8<float>(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    Raw8<float> o;
    o.a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
    o.b = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16, 0, 0));
    return o;
}
template <typename T> __device__ __forceinline__ void buf_store8(__amdgpu_buffer_rsrc_t r, unsigned voff, const float (&v)[8]);
template <> __device__ __forceinline__ void buf_store8<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned voff, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), r, voff, 0, 0);
}
template <> __device__ __forceinline__ void buf_store8<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), r, voff, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, b), r, voff + 16, 0, 0);
}
template <typename T> __device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, const float (&v)[4]);
template <> __device__ __forceinline__ void buf_store4<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned voff, const float (&v)[4]) {
    bf16x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (bf16_t)v[i];
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, a), r, voff, 0, 0);
}
template <> __device__ __forceinline__ void buf_store4<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, const float (&v)[4]) {
    f32x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = v[i];
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, a), r, voff, 0, 0);
}
template <typename T> __device__ __forceinline__ void buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, float (&v)[4]);
template <> __device__ __forceinline__ void buf_load4<bf16_t>(__amdgpu_buffer_rsrc_t r, unsigned voff, float (&v)[4]) {
    const bf16x4 a = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0));
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
}
template <> __device__ __forceinline__ void buf_load4<float>(__amdgpu_buffer_rsrc_t r, unsigned voff, float (&v)[4]) {
    const f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i];
}
template <typename T> __device__ __forceinline__ void lds_store_raw(T* dst, const Raw8<T>& r);
template <> __device__ __forceinline__ void lds_store_raw<bf16_t>(bf16_t* dst, const Raw8<bf16_t>& r) {
    *reinterpret_cast<bf16x8*>(dst) = r.v;
}
template <> __device__ __forceinline__ void lds_store_raw<float>(float* dst, const Raw8<float>& r) {
    *reinterpret_cast<f32x4*>(dst) = r.a;
    *reinterpret_cast<f32x4*>(dst + 4) = r.b;
}

// Tile-invariant plan of one thread's halo-tile items (32 input channels starting at a chunk base).
// PS = LDS pixel stride: 32 (XOR swizzle) or 40 (padded, bf16 only).
template <typename T, int W, int ROWS, int WP, int NTHR, int PS>
struct HaloPlan {
    static constexpr int ITEMS = ROWS * (W + 2) * 4;
    static constexpr int IPT = (ITEMS + NTHR - 1) / NTHR;
    unsigned voff[IPT];   // bytes from the tile's first halo pixel (row h0-1, col -1), chunk channel 0; SED_OOB if padding
    int lds[IPT];         // element offset of the 8-channel group in the LDS tile (bf16) / of the pixel (f32)
    unsigned colmask;     // bit u: item u is a real (non-padding, in-range) column
    Raw8<T> raw[IPT];

    __device__ __forceinline__ void init(int tid, int Cinp) {
        const int cq = tid & 3;
        colmask = 0;
#pragma unroll
        for (int u = 0; u < IPT; ++u) {
            const int it = tid + u * NTHR;
            const int pix = it >> 2;
            const int rowi = pix / (W + 2), coli = pix - rowi * (W + 2);
            const bool ok = (it < ITEMS) && coli >= 1 && coli <= W;
            voff[u] = ok ? (unsigned)(((rowi * W + coli) * Cinp + cq * 8) * (int)sizeof(T)) : SED_OOB;
            if (ok) colmask |= 1u << u;
            if constexpr (sizeof(T) == 2) lds[u] = (rowi * WP + coli) * PS + ((PS == 32) ? ((cq * 8) ^ swz<T>(coli)) : cq * 8);
            else lds[u] = (rowi * WP + coli) * PS;
        }
    }
    // tile_off = (((h0-1)*W - 1)*Cinp + c0)*sizeof(T) as a wrapped unsigned
    __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t img, unsigned tile_off) {
#pragma unroll
        for (int u = 0; u < IPT; ++u) raw[u] = buf_load8<T>(img, voff[u] + tile_off);
    }
    // PRO: SED_PRO_NONE -> raw copy (hardware zeros are already right);
    //      SED_PRO_BNRELU -> relu(scale*x+shift), padding forced back to zero (columns via colmask, rows
    //      only on the first/last tile of an image: row_lo/row_hi = first/last valid halo row index)
    template <int PRO>
    __device__ __forceinline__ void commit(T* __restrict__ xs, int tid, const float* __restrict__ pro_scale,
                                           const float* __restrict__ pro_shift, int c0, int row_lo, int row_hi) const {
        const int cq = tid & 3;
        float sc[8], sh[8];
        if (PRO == SED_PRO_BNRELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = pro_scale[c0 + cq * 8 + e]; sh[e] = pro_shift[c0 + cq * 8 + e]; }
        }
        const bool boundary = (row_lo > 0) || (row_hi < ROWS - 1);
#pragma unroll
        for (int u = 0; u < IPT; ++u) {
            const int it = tid + u * NTHR;
            if (u == IPT - 1 && it >= ITEMS) break;
            if constexpr (sizeof(T) == 2) {
                if (PRO == SED_PRO_NONE) {
                    lds_store_raw<T>(xs + lds[u], raw[u]);
                } else {
                    float v[8];
                    raw_to_f(raw[u], v);
                    bool keep = (colmask >> u) & 1;
                    if (boundary) {
                        const int rowi = (it >> 2) / (W + 2);
                        keep = keep && rowi >= row_lo && rowi <= row_hi;
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = keep ? fmaxf(0.f, fmaf(v[e], sc[e], sh[e])) : 0.f;
                    store8<T>(xs + lds[u], v);
                }
            } else {   // fp32 parity mode: element-wise XOR swizzle, speed irrelevant
                float v[8];
                raw_to_f(raw[u], v);
                const int pix = it >> 2;
                const int rowi = pix / (W + 2), coli = pix - rowi * (W + 2);
                const bool keep = ((colmask >> u) & 1) && rowi >= row_lo && rowi <= row_hi;
                if (PRO == SED_PRO_BNRELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = keep ? fmaxf(0.f, fmaf(v[e], sc[e], sh[e])) : 0.f;
                }
                const int sx = swz<T>(coli);
#pragma unroll
                for (int e = 0; e < 8; ++e) xs[lds[u] + ((cq * 8 + e) ^ sx)] = v[e];
            }
        }
    }
};


typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
// ds_read_b64_tr_b16: per 16-lane group, a 4-row x 16-column block of 16-bit elements delivered
// column-major (lane i gets column i of the 4 rows).  The builtin lets hipcc fold immediates, pair the
// two halves of a fragment into the MFMA operand registers and count lgkmcnt itself.
__device__ __forceinline__ s16x4 ds_read_tr16_b64(const bf16_t* p) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(const __attribute__((address_space(3))) void*)p);
}
__device__ __forceinline__ bf16x8 join_tr(const s16x4& lo, const s16x4& hi) {
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}


// Boundary masks of the loaders' dz production (rows past the last row of an image: only in an image's last tile): as a REAL branch around
// the zeroing (round 5).  hipcc had if-converted the former `if (boundary) v[e] *= m` into eight multiplies per 16-byte item of every stage
// (32 of ~340 vector instructions per loader thread and stage in block 0's backward).  SED_BOUNDARY_BRANCH=0: the multiply form (A/B builds).
#ifndef SED_BOUNDARY_BRANCH
#define SED_BOUNDARY_BRANCH 1
#endif

// ---- weight-gradient kernels: shared parameter block -------------------------------------------------
enum { DZ_GIVEN = 0, DZ_POOL = 1, DZ_BN = 2 };

struct Wgrad2Params {
    const void* x;
    const float* pro_scale;
    const float* pro_shift;
    const void* dz;        // DZ_GIVEN: dz;  DZ_POOL: dy (pooled grad);  DZ_BN: g
    const void* zsrc;      // DZ_POOL / DZ_BN: the pre-BN conv output the coefficients refer to
    const float* scale;    // DZ_POOL
    const float* shift;    // DZ_POOL
    const float* ca;
    const float* cb;
    const float* cc;
    void* dz_out;          // may be NULL
    float* ws;             // [strips][9][Cinp][Coutp]
    int B, H, Cinp, Coutp;
    int tilesPerImg, totalTiles, tpb, strips;
    int pro, pool;
    const float* c1_x;     // SED_PRO_C1: 1-channel fp32 input [B][H][W], its per-column z-score (nullable) and conv1 weights [32][9]
    const float* c1_mean;
    const float* c1_std;
    const float* c1_w;
    int dbg;               // ablation switches (env SED_DBG; profiling only): 1 no dz_out stores, 2 no MFMA loop, 8 no global loads
    unsigned tpi_M, tpi_l; // round 5: tile -> image division as multiply-shift (sed_fastdiv below; filled by the launchers)
    int dzexp;             // SED_F32H3: dz is pre-scaled by 2^dzexp before the fp16 split, the result by 2^-dzexp (bits 8..15 of the dtype argument)
};


// sed_wgrad.hip: producer/consumer weight-gradient kernel (bf16).  Returns -1 when the shape is not covered
// (the caller then falls back to conv_wgrad2_kernel), otherwise 0 / an error code after the launch.
int wgrad3_strips(int B, int H, int W, int Cinp, int Coutp);     // 0 = shape not covered
int launch_wgrad3(int dzmode, Wgrad2Params& p, int W, hipStream_t st);
// sed_wgrad_wide.hip (round 5): the same contract for the wide layers (>= 128 channels on one side, >= 64 on the other; W = 8 / 16 / 32):
// a workgroup owns (128 x 64) or (64 x 128) channels x 9 taps and all eight waves issue MFMAs.  launch_wgrad3 / wgrad3_strips route to
// it where it covers the shape (SED_WGRAD_WIDE=0: the A/B knob)
int wgrad_wide_strips(int B, int H, int W, int Cinp, int Coutp);  // 0 = shape not covered
int launch_wgrad_wide(int dzmode, Wgrad2Params& p, int W, hipStream_t st);

// Run-time ablation switches (SED_DBG bits, wave priorities) exist only in ablation builds (make DEBUG_SWITCHES=1): an untaken
// run-time branch per phase costs the step loops 5-15 %, so the product build compiles them out.
#ifdef SED_DEBUG_SWITCHES
#define SED_DBG(p, bit) ((p).dbg & (bit))
#define SED_SET_PRIO(x) set_wave_prio(x)
#else
#define SED_DBG(p, bit) 0
#define SED_SET_PRIO(x) ((void)0)
#endif

// wave issue priority (s_setprio takes an immediate): the loader waves are dispatched after the MFMA waves, and with equal
// priority the OLDER wave of a SIMD wins every arbitration (MI355X_MICROARCH.md, two waves per SIMD) -- the role that is the
// stage's critical path gets the higher one
__device__ __forceinline__ void set_wave_prio(int pr) {
    switch (pr & 3) {
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        case 3: __builtin_amdgcn_s_setprio(3); break;
        default: break;
    }
}


// ---- forward / data-gradient kernels: shared parameter block ---------------------------------------------
struct ConvParams {
    const void* x;
    const float* pro_scale;
    const float* pro_shift;
    const void* wpack;
    void* z;
    const void* zref;
    const float* epi_scale;
    const float* epi_shift;
    const float* epi_mean;
    const float* epi_invstd;
    float* partial;
    int B, H, Cinp, Coutp;
    int tilesPerImg, totalTiles, tpb, nparts;
    int pro, epi;
    int wres;      // all weight chunks stay resident in LDS (they fit): no per-stage weight staging
    const float* c1_x;     // SED_PRO_C1 / SED_EPI_RELUBWD_C1: 1-channel fp32 input [B][H][W], z-score (nullable), conv1 weights [32][9]
    const float* c1_mean;
    const float* c1_std;
    const float* c1_w;
    void* c1_mask;  // C1 mode: conv1's ReLU decisions, uint16 [B][H][W][2] (written by the forward, read by the data gradient)
    int dbg;       // ablation switches (env SED_DBG; profiling only): 1 no output stores, 2 no MFMA loop, 8 no global loads
    int col_only;  // the 3x3 weights have zero side columns (interleaved Conv1d, W = 8): contract taps 1, 4, 7 only
    const unsigned char* cnt;   // SED_EPI_POOLSTATS: active-pixel counts of the pooled pixels [B][H][W][Coutp] (zref = pooled activation)
    int* flag;                  // SED_EPI_POOLSTATS: raised when a channel's statistics cannot be formed (scale = 0)
    int dry;                    // host only: launch_conv_pc answers "would launch" (0) / "shape not covered" (-1) without launching
    // SED_PRO_DZBN / SED_PRO_DZPOOL: x = zsrc (the pre-BN conv output the coefficients refer to), dz_g = g (DZBN: [B][H][W][Cinp]) or
    // the block's output gradient dy (DZPOOL: [B][H/pool][W/pool][Cinp]); coefficients [Cinp]; dz_out [B][H][W][Cinp] (written once)
    const void* dz_g;
    const float* dz_ca;
    const float* dz_cb;
    const float* dz_cc;
    const float* dz_sc;         // DZPOOL: BN scale / shift of the ReLU decision
    const float* dz_sh;
    void* dz_out;
    int dz_pool;                // DZPOOL: 1 or 2
    // round 5 (host only fills them, launch_pc_n): exact division of the stage bookkeeping's non-negative 31-bit values by tilesPerImg / nchunks
    // as (umulhi(M, n) + n) >> l -- three scalar instructions instead of the ~15 of a run-time division, ~10 divisions per stage and wave
    unsigned tpi_M, tpi_l, nch_M, nch_l;
    int xexp;                   // SED_F32H3: x is pre-scaled by 2^xexp before the fp16 split, the result by 2^-xexp (bits 8..15 of the dtype argument)
};
// q = n / d for 0 <= n < 2^31: l = ceil(log2 d), M = floor(2^32 (2^l - d) / d) + 1 (Granlund / Montgomery; t + n cannot overflow for n < 2^31)
inline void sed_fastdiv_make(unsigned d, unsigned* M, unsigned* l) {
    unsigned ll = 0;
    while ((1ull << ll) < d) ++ll;
    *l = ll;
    *M = (unsigned)((((1ull << ll) - d) << 32) / d + 1);
}
__device__ __forceinline__ int sed_fastdiv(int n, unsigned M, unsigned l) { return (int)((__umulhi((unsigned)n, M) + (unsigned)n) >> l); }

// sed_conv_x3.hip (round 6): dtype SED_F32X3 -- fp32 tensors, split-bf16 (hi + lo) operands, three bf16 MFMAs per product
// (half: 0 = bf16 pieces, SED_F32X3; 1 = fp16 pieces with a scaled lo piece, SED_F32H3)
int launch_conv_x3(int half, ConvParams& p, int W, hipStream_t st);
// sed_conv_x3pc.hip: the split-operand forward / data gradient (fp16 pieces) in producer / consumer form.  -1 = not covered.
int launch_conv_x3pc(ConvParams& p, int W, hipStream_t st);
// sed_wgrad_x3.hip: the split-operand weight gradient (fp16 pieces) in producer / consumer form.  -1 = shape not covered (the caller takes
// launch_wgrad_x3); on entry p.strips = the slab count the caller's workspace holds, on return the count written.
int launch_wgrad_x3pc(int dzmode, Wgrad2Params& p, int W, hipStream_t st);
int launch_wgrad_x3(int half, int dzmode, Wgrad2Params& p, int W, int wn, hipStream_t st);

// sed_conv_pc.hip: bf16 forward / data gradient through the producer/consumer kernel; -1 = shape not covered
// (the caller falls back to conv_igemm_kernel / conv_wreg_kernel), otherwise 0 / an error code after the launch.
int launch_conv_pc(ConvParams& p, int W, hipStream_t st);
// sed_conv_wir.hip: bf16 forward / data gradient with the weights resident in registers (>= 64 input channels, W <= 32);
// -1 = shape not covered
int launch_conv_wir(ConvParams& p, int W, hipStream_t st);
int launch_conv_w4(ConvParams& p, int W, hipStream_t st);

// ---- sed_bwd_fused.hip: weight gradient and data gradient of a layer from one dz tile in LDS (dz never written) -------------
struct BwdFusedParams {
    const void* x;             // the convolution's input [B][H][W][Cinp]: PRO_NONE as stored, PRO_BNRELU: z of the layer before (a = relu(scale*z+shift))
    const float* pro_scale;
    const float* pro_shift;
    const void* gsrc;          // DZ_POOL: dy of the pooled block output [B][H>>1][W>>1][Coutp] (pool 2) / [B][H][W][Coutp] (pool 1);  DZ_BN: g
    const void* zsrc;          // the pre-BN conv output the coefficients refer to [B][H][W][Coutp]
    const float* scale;        // DZ_POOL
    const float* shift;
    const float* ca;
    const float* cb;
    const float* cc;
    const void* wpack_t;       // data-gradient operator (sed_pack_conv_weight, transposed = 1)
    void* dx;                  // [B][H][W][Cinp]
    const void* zref;          // SED_EPI_RELUBWD: ReLU / BN-backward reference (z of the layer before);  SED_EPI_POOLSTATS: pooled activation
    const unsigned char* cnt;  // SED_EPI_POOLSTATS: active-pixel counts
    const float* epi_scale;
    const float* epi_shift;
    const float* epi_mean;
    const float* epi_invstd;
    float* partial;            // [nparts][2][Cinp]
    int* flag;                 // SED_EPI_POOLSTATS
    float* ws;                 // [nwg][9][Cinp][Coutp] weight-gradient slabs
    int B, H, Cinp, Coutp;
    int pool, dzmode, pro, epi;
    int nparts, nwg, tilesPerImg, totalTiles, tpb;
    int dry;                   // host only: answer "would launch" without launching
    int prio;                  // issue priority of the loader waves (0..3)
    int abl;                   // ablation switches (EXPERIMENTS builds only)
};
// 0 = shape / mode not covered; otherwise the number of workgroups (= workspace slabs)
int bwd_fused_nwg(int B, int H, int W, int Cinp, int Coutp, int dzmode, int pro, int epi);
int launch_bwd_fused(BwdFusedParams& p, int W, hipStream_t st);       // -1 = not covered
int bwd_fused_max_nwg(int B, int H, int W, int Cinp, int Coutp);       // over every covered (dzmode, prologue, epilogue) form; 0 = none
int bwd_fused_c1_nwg(int B, int H);
// sed_bwd_fused_cs.hip: the same for 128 output channels at W = 16 / 8, the workgroups of a strip sliced by input channels
int bwd_fused_cs_nstrips(int B, int H, int W, int Cinp, int Coutp, int dzmode, int pro, int epi, int pool);      // 0 = not covered
int launch_bwd_fused_cs(BwdFusedParams& p, int W, hipStream_t st);      // -1 = not covered; p.nwg = strips (= workspace slabs)

// sed_bwd_fused_c1.hip: block 0 (C1 mode), conv2's weight gradient + gated data gradient + [A; sum g] in one launch; -1 = not covered
int launch_bwd_fused_c1(const float* x1, const float* fmean, const float* fstd, const float* w1, const float* sc1, const float* sh1,
                        const void* dy, const void* z2, const float* sc2, const float* sh2, const float* ca, const float* cb,
                        const float* cc, const void* wpack_t, const void* mask, float* a_part, int nparts, float* ws, int B, int H,
                        int* nwg, hipStream_t st);

// ---- "C1 mode": the first ConvBlock without materialising conv1's output -----------------------------------
// z1 = conv3x3(x_norm, w1) has ONE input channel: 9 FMAs per output element re-create it from a 3x3 window of the
// fp32 input, which is 16x smaller than z1.  Loader waves that need z1 (as the next convolution's input after
// BN+ReLU, or as the ReLU / BatchNorm-backward reference) recompute it instead of reading 64 B/pixel from HBM.
// Internal prologue / epilogue codes (beyond the SED_PRO_* / SED_EPI_* of the header):
enum { SED_PRO_C1 = 2, SED_EPI_RELUBWD_C1 = 3 };      // (SED_EPI_POOLSTATS = 4 is public: include/sed_hip.h)
// "dz on load" prologues of the data-gradient call (round 4, sed_conv3x3_dgrad_dz): the loader waves produce the BatchNorm / ReLU /
// avg-pool backward dz = ca*g + cb*z + cc of the layer's output gradient from (g, z) as the weight-gradient kernels do
// (DZ_BN / DZ_POOL), feed it to the convolution AND write it out once for the weight-gradient call that follows with dz given.
enum { SED_PRO_DZBN = 11, SED_PRO_DZPOOL = 12 };

// a loader thread owns image column `col` and the 8 conv1 output channels ch0..ch0+7
struct C1Ctx {
    float w[9][8];
    float mu[3], is[3];     // z-score of columns col-1, col, col+1; is = 0 for a padding column (value -> 0)
};
__device__ __forceinline__ void c1ctx_init(C1Ctx& c, const float* __restrict__ w1, int ch0, const float* __restrict__ fmean,
                                           const float* __restrict__ fstd, int col, int W) {
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) c.w[k][e] = w1[(ch0 + e) * 9 + k];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int cc = col + j - 1;
        const bool ok = cc >= 0 && cc < W;
        c.mu[j] = (ok && fmean) ? fmean[cc] : 0.f;
        c.is[j] = ok ? (fstd ? 1.0f / fstd[cc] : 1.0f) : 0.f;
    }
}
// one window row: raw values -> z-scored (0 outside the image)
__device__ __forceinline__ void c1_norm_row(const C1Ctx& c, const float (&raw)[3], bool row_ok, float (&xn)[3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) xn[j] = row_ok ? (raw[j] - c.mu[j]) * c.is[j] : 0.f;
}
__device__ __forceinline__ void c1_eval(const C1Ctx& c, const float (&r0)[3], const float (&r1)[3], const float (&r2)[3],
                                        float (&z)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = fmaf(r0[j], c.w[j][e], z[e]);
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = fmaf(r1[j], c.w[3 + j][e], z[e]);
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = fmaf(r2[j], c.w[6 + j][e], z[e]);
}
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}

// ---- C1 mode on the matrix pipe --------------------------------------------------------------------------------
// conv1 as a K=16 GEMM: D[ch][pixel] = W1[ch][slot] * P[slot][pixel]: ONE v_mfma_f32_32x32x16_bf16 re-creates 32 pixels x 32
// channels of z1 from a z-scored copy of the 1-channel input in LDS (column 0 = image column -1, zero outside the image).
// The lane holds pixel (lane & 31) and the 16 channels c(i, g) = (i & 3) + 8*(i >> 2) + 4*g, g = lane >> 5.
// K slots (round 4: a patch row is FOUR consecutive input values, so that a bf16 input tile yields it with one 8-byte read):
//   k-group 0: slots 0-2 = taps (0, 0..2), slot 3 = the fourth value of that input row (weight 0), slots 4-6 = taps (1, 0..2), 7: weight 0
//   k-group 1: slots 0-2 = taps (2, 0..2), slot 3: weight 0, slots 4, 5 = the constant 1 (BatchNorm1's shift as hi + lo), 6, 7 = 0
// BatchNorm1's scale and shift ride in the same MFMA: the A fragment holds scale[ch]*w1[ch][tap] and the shift split into two bf16
// terms (hi + lo, exact to 2^-17) -- the accumulator comes out as scale*conv1 + shift and the builder's tail is max / mask only.
struct C1Mma {
    bf16x8 wa;             // A fragment: [ch = lane & 31][slot 8*g + j]
};
__device__ __forceinline__ void c1mma_init(C1Mma& m, const float* __restrict__ w1, const float* __restrict__ scale,
                                           const float* __restrict__ shift, int lane) {
    const int ch = lane & 31, g = lane >> 5;
    const float sc = scale[ch], sh = shift[ch];
    const bf16_t sh_hi = (bf16_t)sh;
    const bf16_t sh_lo = (bf16_t)(sh - (float)sh_hi);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = 2 * g + (j >> 2), dx = j & 3;          // input row of the slot (row 3 = the constants)
        m.wa[j] = (row < 3 && dx < 3) ? (bf16_t)(w1[ch * 9 + 3 * row + dx] * sc) : (row == 3 && dx == 0) ? sh_hi : (row == 3 && dx == 1) ? sh_lo : (bf16_t)0.f;
    }
}
// patch fragment of pixel (halo row rr, column half*32 + (lane & 31)) from an fp32 input tile (XW words per row): the kernels that kept the
// fp32 tile (csrc/sed_wgrad.hip's unfused form) -- same slots, same bits as c1mma_patch_b below
template <int XW>
__device__ __forceinline__ bf16x8 c1mma_patch_f(const float* __restrict__ xt, int rr, int half, int lane) {
    const int col = half * 32 + (lane & 31), g = lane >> 5;
    const float* p0 = xt + (rr + 2 * g) * XW + col;
    float v[8];
#pragma unroll
    for (int t = 0; t < 3; ++t) { v[t] = p0[t]; v[4 + t] = p0[XW + t]; }
    v[3] = 0.f; v[7] = 0.f;
    bf16x8 xb;
#pragma unroll
    for (int j = 0; j < 8; ++j) xb[j] = (bf16_t)v[j];
    u32x4 xw = __builtin_bit_cast(u32x4, xb);
    xw[2] = g ? 0x3F803F80u : xw[2];           // k-group 1: slots 4, 5 = 1.0, 6, 7 = 0
    xw[3] = g ? 0u : xw[3];
    return __builtin_bit_cast(bf16x8, xw);
}
// The bf16 two-copy input tile (round 4): copy A holds x[r][i] at r*XP + i, copy B (XB elements later) x[r][i + 1] at r*XP + i, so the
// four consecutive values x[r][p .. p+3] of a patch row are ONE 4-byte-aligned 8-byte read for every p (even p: copy A at p, odd p: copy
// B at p - 1); columns W+4 .. W+7 of every row of copy A hold the constants {1, 1, 0, 0} (k-group 1's second half: a lane-constant COLUMN
// select instead of a data select).  XP = W + 8; XB chosen = 16 dwords modulo the 32 banks (the even lanes' run of copy A and the odd
// lanes' run of copy B then share two banks at most).  Two ds_read2_b32 per 32-pixel block instead of eight ds_read_b32 + four
// conversions + six mask instructions: the rebuild sits on the consumer waves' critical path (block 0 forward and backward).
typedef unsigned c1_u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
template <int W, int R>
struct C1Tile {
    static constexpr int XP = W + 8;
    static constexpr int XB = ((R * XP / 2 + 15) / 32 * 32 + 16) * 2 >= R * XP ? ((R * XP / 2 + 15) / 32 * 32 + 16) * 2 : ((R * XP / 2 + 15) / 32 * 32 + 48) * 2;
    static constexpr int N = XB + R * XP;          // elements of one tile (both copies)
    static_assert(XB >= R * XP && (XB / 2) % 32 == 16 && XP % 2 == 0, "copy B offset");
};
// lane-constant element offsets of a 32-pixel block's two reads (block column half): .x first read (rows rr / rr + 2), .y second (row rr + 1 /
// the constant column)
template <int W, int R>
__device__ __forceinline__ void c1tile_lane_offsets(int half, int lane, int& o1, int& o2) {
    typedef C1Tile<W, R> TL;
    const int p = half * 32 + (lane & 31), g = lane >> 5;
    const int src = (p & 1) ? TL::XB + p - 1 : p;
    o1 = src + (g ? 2 * TL::XP : 0);
    o2 = g ? TL::XP + W + 4 : TL::XP + src;
}
__device__ __forceinline__ bf16x8 c1mma_patch_b(const bf16_t* __restrict__ row_rr, int o1, int o2) {
    const c1_u32x2_a4 lo = *reinterpret_cast<const c1_u32x2_a4*>(row_rr + o1);
    const c1_u32x2_a4 hi = *reinterpret_cast<const c1_u32x2_a4*>(row_rr + o2);
    const u32x4 xw = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, xw);
}
// zero a tile (both copies: the unused columns are read as slot-3 / slot-7 values with weight 0 and must be finite) and write its constants
template <int W, int R>
__device__ __forceinline__ void c1tile_init(bf16_t* __restrict__ t, int tid, int nthr) {
    typedef C1Tile<W, R> TL;
    unsigned* w = reinterpret_cast<unsigned*>(t);
    for (int i = tid; i < TL::N / 2; i += nthr) {
        const int e = 2 * i, r = e / TL::XP, c = e - r * TL::XP;
        w[i] = (e < R * TL::XP && c == W + 4) ? 0x3F803F80u : 0u;
    }
}
// element x of tile position (row r, halo column c): both copies
template <int W, int R>
__device__ __forceinline__ void c1tile_store(bf16_t* __restrict__ t, int r, int c, float x) {
    typedef C1Tile<W, R> TL;
    const bf16_t b = (bf16_t)x;
    t[r * TL::XP + c] = b;
    if (c >= 1) t[TL::XB + r * TL::XP + c - 1] = b;
}
// a[i] = relu(sc*z1 + sh) of pixel (halo row rr, column half*32 + (lane & 31)); bit i of `mask` = (a[i] > 0)
// (computed only when WANT_MASK).  XW = W + 2 words per row of xt.
// VALU diet (the builder runs between the consumers' MFMAs): lanes of k-group 1 hold tap 8 and seven zeros -- they
// read tap 8 through a per-lane first offset and the other seven elements are cleared with two ANDs on the packed
// registers instead of eight selects; the mask costs two instructions per bit (compare into VCC, add-with-carry).
template <int XW, bool WANT_MASK>
__device__ __forceinline__ void c1mma_block(const C1Mma& m, const float* __restrict__ xt, int rr, int half, int lane,
                                            float (&a)[16], unsigned& mask) {
    const bf16x8 xb = c1mma_patch_f<XW>(xt, rr, half, lane);
    f32x16 d;
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = 0.f;
    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m.wa, xb, d, 0, 0, 0);
    mask = 0;
#pragma unroll
    for (int i = 15; i >= 0; --i) {
        const float y = d[i];                 // = scale*conv1 + shift (c1mma_init)
        a[i] = fmaxf(0.f, y);
        if (WANT_MASK)      // mask = 2*mask + (y > 0): bit i ends at position i
            asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(y) : "vcc");
    }
}

// The same block builder in two phases, so that a wave building several blocks can issue all their LDS reads and
// MFMAs first (independent, pipelined) and run the mask / store tails afterwards.
template <int XW>
__device__ __forceinline__ f32x16 c1mma_block_mfma(const C1Mma& m, const float* __restrict__ xt, int rr, int half, int lane) {
    const bf16x8 xb = c1mma_patch_f<XW>(xt, rr, half, lane);
    f32x16 d;
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = 0.f;
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(m.wa, xb, d, 0, 0, 0);
}
// ... from the bf16 two-copy tile: row_rr = tile + rr * XP, (o1, o2) from c1tile_lane_offsets
__device__ __forceinline__ f32x16 c1mma_block_mfma_b(const C1Mma& m, const bf16_t* __restrict__ row_rr, int o1, int o2) {
    const bf16x8 xb = c1mma_patch_b(row_rr, o1, o2);
    f32x16 d;
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = 0.f;
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(m.wa, xb, d, 0, 0, 0);
}
// The tail on packed bf16 words (round 3): w[k] = bf16 pair (a[2k], a[2k+1]) of relu(d) -- eight v_cvt_pk_bf16_f32 + eight v_pk_max_i16
// (ReLU on the bf16 bit patterns: a negative float is a negative int16) instead of sixteen v_max_f32 + eight conversions; the mask
// (bit i = a[i] > 0) as eight v_pk_min_u16 (each half -> 0 / 1) + eight v_dot4_u32_u8 (the two flags times their bit weights,
// accumulated per byte of the mask) instead of 32 compare / add-with-carry instructions.  Same activation bits; a mask bit differs
// from `d > 0` only for a positive d that rounds to bf16 zero (below 2^-133).
template <bool WANT_MASK>
__device__ __forceinline__ void c1mma_block_tail_pk(const f32x16& d, unsigned (&w)[8], unsigned& mask) {
    unsigned mlo = 0, mhi = 0;
    const unsigned one2 = 0x00010001u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const f32x2 r = {d[2 * k], d[2 * k + 1]};
        const bf16x2 rb = __builtin_convertvector(r, bf16x2);
        sed_i16x2 ri = __builtin_bit_cast(sed_i16x2, rb);
        ri = __builtin_elementwise_max(ri, (sed_i16x2){0, 0});
        w[k] = __builtin_bit_cast(unsigned, ri);
        if (WANT_MASK) {
            unsigned t;
            asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(w[k]), "v"(one2));
            const unsigned wt = (1u << ((2 * k) & 7)) | (1u << (16 + ((2 * k + 1) & 7)));      // byte 0: bit of a[2k], byte 2: bit of a[2k+1]
            if (k < 4) mlo = __builtin_amdgcn_udot4(t, wt, mlo, false);
            else mhi = __builtin_amdgcn_udot4(t, wt, mhi, false);
        }
    }
    mask = mlo | (mhi << 8);
}
template <bool WANT_MASK>
__device__ __forceinline__ void c1mma_block_tail(const C1Mma& m, const f32x16& d, float (&a)[16], unsigned& mask) {
    mask = 0;
#pragma unroll
    for (int i = 15; i >= 0; --i) {
        const float y = d[i];                 // = scale*conv1 + shift (c1mma_init)
        a[i] = fmaxf(0.f, y);
        if (WANT_MASK)
            asm volatile("v_cmp_lt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(y) : "vcc");
    }
}
