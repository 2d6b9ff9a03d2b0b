// Weight gradient of the 3x3 convolutions, producer/consumer form (bf16, gfx950).
//
//   dW[tap][cin][cout] = sum_pixels a[pixel + tap][cin] * dz[pixel][cout]
//
// autograd's conv weight gradient of ConvBlock (/root/reference/models/spectogram_models.py:132-140,
// backward of :155-156), with the BatchNorm / ReLU / avg-pool backward that produces dz fused in.
//
// Why this shape.  The GEMM itself is small next to its operand preparation: every dz element needs two
// loads and ~8 VALU (BN/ReLU/pool backward), every activation a load and the BN+ReLU prologue, both go
// through LDS in the transposed-read layout.  A kernel whose waves all do "load, prepare, barrier, MFMA,
// barrier" runs those phases back to back and leaves the matrix pipe idle most of the time (measured:
// removing the MFMA loop from the previous kernel changed its time by 10 %).  Here one 512-thread
// workgroup per CU splits the roles:
//   * waves 4-7 (one per SIMD) are PRODUCERS: they keep TWO tiles of global loads in flight in registers,
//     turn tile t+1 into the LDS operand images (prologue, dz math, dz_out store) and never touch the
//     matrix pipe;
//   * waves 0-3 (one per SIMD) are CONSUMERS: transposed LDS reads + MFMA only, all 9 taps of a
//     (32 cin x 32 cout) pair per wave (or a tap row when the layer has fewer pairs), accumulators
//     resident for the whole strip;
//   * ONE s_barrier per tile hands a double-buffered LDS stage over; nothing else synchronises.
// VALU work of the producer and MFMA work of the consumer on the same SIMD overlap (separate pipes).
#include "conv_common.h"

// tile -> image in the loaders' per-tile bookkeeping (round 5): exact multiply-shift instead of a run-time division (~15 scalar instructions
// each, several per tile and wave); SED_WG_FASTDIV=0: A/B builds
#ifndef SED_WG_FASTDIV
#define SED_WG_FASTDIV 1
#endif
#if SED_WG_FASTDIV
#define WG_DIV_TPI(n) sed_fastdiv((n), p.tpi_M, p.tpi_l)
#else
#define WG_DIV_TPI(n) ((n) / p.tilesPerImg)
#endif

#include <stdlib.h>

namespace {

constexpr int kWgrad3Blocks = 256;     // one workgroup per CU

enum { MODE_ROW = 0, MODE_PAIR = 1 };

__device__ __forceinline__ void wg_barrier() {
    // LDS writes/reads of this wave are complete; global loads stay in flight across the barrier
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


// C1 mode: one 32-pixel block (halo row rr, column half) of the tile's activation image from the input copy xt
template <int W, int WP, int XTW>
__device__ __forceinline__ void c1_build_block_w(const C1Mma& c1m, const float* __restrict__ xt, bf16_t* __restrict__ st, int bi, int lane,
                                                 int h0, int H, bool live) {
    const int rr = bi >> 1, half = bi & 1, g = lane >> 5;
    const int hr = h0 - 1 + rr;
    float a[16];
    unsigned mk;
    c1mma_block<XTW, false>(c1m, xt, rr, half, lane, a, mk);
    const bool inimg = live && hr >= 0 && hr < H;
    const int coll = half * 32 + (lane & 31) + 1;
    bf16_t* dst = st + (rr * WP + coll) * 32 + g * 4;
    const int sw = (coll >> 2) & 3;
    if (!inimg) {        // wave-uniform
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = 0.f;
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        float v4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = a[4 * g4 + e];
        store4<bf16_t>(dst + ((g4 ^ sw) * 8), v4);
    }
}

// CI_T x CO_T 32-channel tiles per workgroup: (1,1) / (1,2): consumer wave = tap row (MODE_ROW);
// (2,2): consumer wave = (cin tile, cout tile) pair, all 9 taps (MODE_PAIR)
// NPW: loader waves (4, or 8 = two per SIMD where the variant fits three waves per SIMD into the register file: the loader
// waves' dz arithmetic -- ~45 instructions per 16-byte item, already packed f32 -- is what bounds the 32-channel layers)
template <int W, int CI_T, int CO_T, int DZ, int PRO, int NPW = 4>
__global__ __launch_bounds__(256 + 64 * NPW) void conv_wgrad3_kernel(Wgrad2Params p) {
    typedef bf16_t T;
    constexpr int MODE = (CI_T == 2 && CO_T == 2) ? MODE_PAIR : MODE_ROW;
    static_assert(MODE == MODE_PAIR || CI_T == 1, "row mode has one input tile");
    constexpr int BM = (W == 64 && CI_T == 1) ? 256 : 128;   // pixels per tile (whole rows)
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr int ROWS = TH + 2;
    constexpr int XS1 = ROWS * WP * 32;               // one cin tile of the activation halo image (elements)
    constexpr int DZ1 = BM * 32;                      // one cout tile of the dz image
    constexpr int STAGE = CI_T * XS1 + CO_T * DZ1;
    constexpr int NP = 64 * NPW;                      // producer threads
    constexpr int NTHR = 256 + NP;
    constexpr int XPER = ROWS * W * 4;                // 16-byte items of one cin tile (the two padding columns are zeroed once)
    constexpr int XITEMS = CI_T * XPER;
    constexpr int XIPT = (XITEMS + NP - 1) / NP;
    constexpr int IPP = CO_T * 4;                     // 16-byte items per dz pixel
    constexpr int DITEMS = BM * IPP;
    constexpr int DIPT = DITEMS / NP;
    constexpr int DQS = NP / IPP;                     // pixels between two dz items of a thread
    static_assert(DITEMS % NP == 0 && NP % IPP == 0, "dz item geometry");
    constexpr int NACC = (MODE == MODE_PAIR) ? 9 : 3 * CO_T;
    constexpr bool C1PRO = PRO == SED_PRO_C1;         // x operand = relu(bn1(conv1(x1))) recomputed from the 1-channel input
    static_assert(!C1PRO || (W == 64 && CI_T == 1), "C1 mode: W = 64, 32 input channels");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* stage0 = reinterpret_cast<T*>(smem);           // [2][STAGE]: xs[CI_T][ROWS][WP][32] (swizzled), dzs[CO_T][BM][32]
    float* coef = reinterpret_cast<float*>(stage0 + 2 * STAGE);   // [5][CO_T*32]: scale, shift, ca, cb, cc
    float* pcoef = coef + 5 * CO_T * 32;              // [2][CI_T*32]: prologue scale, shift
    constexpr int XTW = W + 2, XTR = ROWS + 2, XTN = XTR * XTW;      // C1 mode: fp32 copy of the 1-channel input of a tile
    float* xt0 = pcoef + 2 * CI_T * 32;                // [2][XTN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, Cinp = p.Cinp, Coutp = p.Coutp;
    const int NCI = Cinp / (32 * CI_T), NCO = Coutp / (32 * CO_T);
    const int NY = NCI * NCO;
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int strip = logical / NY, yb = logical - strip * NY;
    const int cig = yb / NCO;
    const int ci0 = cig * 32 * CI_T, co0 = (yb % NCO) * 32 * CO_T;
    const int psh = p.pool >> 1;                      // pool is 1 or 2
    const int Ho = H >> psh, Wo = W >> psh;

    // ---- coefficients into LDS (both roles; visible after the first barrier) -----------------------------
    if (DZ != DZ_GIVEN) {
        const float inv_pool = psh ? 0.25f : 1.0f;
        for (int i = tid; i < 5 * CO_T * 32; i += NTHR) {
            const int a = i / (CO_T * 32), c = i - a * (CO_T * 32);
            const float* src = (a == 0) ? p.scale : (a == 1) ? p.shift : (a == 2) ? p.ca : (a == 3) ? p.cb : p.cc;
            float v = (src != nullptr) ? src[co0 + c] : 0.f;
            if (a == 2 && DZ == DZ_POOL) v *= inv_pool;   // the 1/pool^2 of the avg-pool backward folded into ca
            coef[i] = v;
        }
    }
    {   // the two padding columns of every halo row stay zero for the whole kernel (both stages)
        constexpr int NPAD = 2 * CI_T * ROWS * 2 * 4;
        for (int i = tid; i < NPAD; i += NTHR) {
            const int c16 = i & 3, side = (i >> 2) & 1, rowi = (i >> 3) % ROWS, ci = ((i >> 3) / ROWS) % CI_T, sg = (i >> 3) / (ROWS * CI_T);
            bf16x8 z8;
#pragma unroll
            for (int e = 0; e < 8; ++e) z8[e] = (bf16_t)0.f;
            *reinterpret_cast<bf16x8*>(stage0 + sg * STAGE + ci * XS1 + (rowi * WP + (side ? W + 1 : 0)) * 32 + c16 * 8) = z8;
        }
    }
    if (PRO == SED_PRO_BNRELU || C1PRO) {
        for (int i = tid; i < 2 * CI_T * 32; i += NTHR) {
            const int a = i / (CI_T * 32), c = i - a * (CI_T * 32);
            pcoef[i] = (a == 0 ? p.pro_scale : p.pro_shift)[ci0 + c];
        }
    }
    __syncthreads();

    const int t_begin = strip * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    const int ntl = t_end > t_begin ? t_end - t_begin : 0;
    if (C1PRO && ntl > 0) {     // the first tile's input copy (later tiles: staged one iteration ahead by the producers)
        const int b = t_begin / p.tilesPerImg, h0 = (t_begin - b * p.tilesPerImg) * TH;
        for (int e = tid; e < XTN; e += NTHR) {
            const int r = e / XTW, c = e - r * XTW;
            const int hy = h0 - 2 + r, wx = c - 1;
            float v = 0.f;
            if (hy >= 0 && hy < H && wx >= 0 && wx < W) {
                v = p.c1_x[((size_t)b * H + hy) * W + wx];
                if (p.c1_mean) v = (v - p.c1_mean[wx]) * (1.0f / p.c1_std[wx]);
            }
            xt0[e] = v;
        }
        __syncthreads();
    }

    if (wave >= 4) {
        // =============================== PRODUCERS =====================================================
        SED_SET_PRIO(p.dbg >> 8);
        const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
        const T* __restrict__ dg = reinterpret_cast<const T*>(p.dz);
        const T* __restrict__ zsg = reinterpret_cast<const T*>(p.zsrc);
        // the dz tile is produced by every cin group of the strip; they take turns (by tile parity) writing it out
        T* __restrict__ dzo = reinterpret_cast<T*>(p.dz_out);
        const int pt = tid - 256;
        const int cq = pt & 3;
        const size_t ximg_ = (size_t)H * W * Cinp, zimg_ = (size_t)H * W * Coutp, pimg_ = (size_t)Ho * Wo * Coutp;

        // tile-invariant item plans
        unsigned xvoff[XIPT];
        int xlds[XIPT];
#pragma unroll
        for (int u = 0; u < XIPT; ++u) {
            const int it = pt + u * NP;
            const int ci = it / XPER, rem = it - ci * XPER;
            const int pix = rem >> 2;
            const int rowi = pix / W, coli = pix - rowi * W + 1;
            const bool ok = it < XITEMS;
            xvoff[u] = ok ? (unsigned)(((rowi * W + coli) * Cinp + ci * 32 + cq * 8) * 2) : SED_OOB;
            xlds[u] = ok ? ci * XS1 + (rowi * WP + coli) * 32 + ((cq * 8) ^ swz<T>(coli)) : -1;
        }
        const int dq0 = pt / IPP, dc8 = (pt - dq0 * IPP) * 8;       // first dz pixel and the (fixed) channel group
        const unsigned dvoff0 = (unsigned)((dq0 * Coutp + co0 + dc8) * 2);
        const int dlds0 = (dc8 >> 5) * DZ1 + dq0 * 32 + (dc8 & 31);
        unsigned pvoff[DIPT];
#pragma unroll
        for (int u = 0; u < DIPT; ++u) {
            const int q = dq0 + u * DQS;
            pvoff[u] = (unsigned)(((((q / W) >> psh) * Wo + ((q % W) >> psh)) * Coutp + co0 + dc8) * 2);
        }
        // the dz coefficients of the thread's fixed channel group live in registers: read from LDS per item they cost ~10
        // ds_read_b128 and two exposed LDS round trips per 16-byte item (measured on csrc/sed_bwd_fused.hip: -8 %)
        float kca[8], kcb[8], kcc[8], ksc[8], ksh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = dc8 + e;
            kca[e] = DZ != DZ_GIVEN ? coef[2 * CO_T * 32 + c] : 0.f; kcb[e] = DZ != DZ_GIVEN ? coef[3 * CO_T * 32 + c] : 0.f;
            kcc[e] = DZ != DZ_GIVEN ? coef[4 * CO_T * 32 + c] : 0.f;
            ksc[e] = DZ == DZ_POOL ? coef[c] : 0.f; ksh[e] = DZ == DZ_POOL ? coef[CO_T * 32 + c] : 0.f;
        }

        constexpr int XTIPT = (XTN + NP - 1) / NP;
        struct RawSet { Raw8<T> x[C1PRO ? 1 : XIPT]; float xr[C1PRO ? XTIPT : 1]; Raw8<T> a[DIPT]; Raw8<T> b[DIPT]; };
        float xtmu[XTIPT], xtis[XTIPT];
        if (C1PRO) {
#pragma unroll
            for (int u2 = 0; u2 < XTIPT; ++u2) {
                const int e = pt + u2 * NP, c = (e % XTW) - 1;
                const bool ok = e < XTN && c >= 0 && c < W;
                xtmu[u2] = (ok && p.c1_mean) ? p.c1_mean[c] : 0.f;
                xtis[u2] = ok ? (p.c1_std ? 1.0f / p.c1_std[c] : 1.0f) : 0.f;
            }
        }
        // C1 mode: the input copy of tile t (rows h0-2 .. h0+TH+1, columns -1 .. W) for xt
        auto issue_x1 = [&](RawSet& r, int tile) {
            const bool live = tile < t_end && !(SED_DBG(p, 8));
            const int b = live ? WG_DIV_TPI(tile) : 0;
            const int h0 = live ? (tile - b * p.tilesPerImg) * TH : 0;
            const size_t img = live ? (size_t)H * W : 0;
            const __amdgpu_buffer_rsrc_t s1 = make_srd(p.c1_x + (size_t)b * img, img * 4);
#pragma unroll
            for (int u2 = 0; u2 < XTIPT; ++u2) {
                const int e = pt + u2 * NP, rr = e / XTW, c = e - rr * XTW - 1;
                const bool ok = e < XTN && c >= 0 && c < W;
                r.xr[C1PRO ? u2 : 0] = buf_load_f32(s1, ok ? (unsigned)(((h0 - 2 + rr) * W + c) * 4) : SED_OOB);
            }
        };
        auto write_xt = [&](const RawSet& r, int tile) {          // z-scored, zero outside the image / strip
            const bool live = tile < t_end;
            const int b = live ? WG_DIV_TPI(tile) : 0;
            const int h0 = live ? (tile - b * p.tilesPerImg) * TH : 0;
            float* xtn = xt0 + ((tile - t_begin) & 1) * XTN;
#pragma unroll
            for (int u2 = 0; u2 < XTIPT; ++u2) {
                const int e = pt + u2 * NP;
                if (u2 == XTIPT - 1 && e >= XTN) break;
                const int hy = h0 - 2 + e / XTW;
                xtn[e] = (live && hy >= 0 && hy < H) ? (r.xr[C1PRO ? u2 : 0] - xtmu[u2]) * xtis[u2] : 0.f;
            }
        };

        // Loads are issued UNCONDITIONALLY (a tile past the strip gets zero-sized descriptors: every lane out of
        // range, zeros returned, no memory traffic): with a conditional issue the compiler's vmcnt bookkeeping
        // must assume the younger set may be missing and waits for vmcnt(0) -- i.e. for BOTH tiles in flight.
        auto issue = [&](RawSet& r, int tile) {
            const bool live = tile < t_end && !(SED_DBG(p, 8));
            const int b = live ? WG_DIV_TPI(tile) : 0;
            const int h0 = live ? (tile - b * p.tilesPerImg) * TH : 0;
            const size_t ximg = live ? ximg_ : 0, zimg = live ? zimg_ : 0, pimg = live ? pimg_ : 0;
            if constexpr (C1PRO) {       // (the 1-channel input copy has its own pipeline: issue_x1 / write_xt)
            } else {
                const __amdgpu_buffer_rsrc_t xsrd = make_srd(xg + (size_t)b * ximg, ximg * 2);
                const unsigned xt = (unsigned)((((h0 - 1) * W - 1) * Cinp + ci0) * 2);
#pragma unroll
                for (int u = 0; u < XIPT; ++u) r.x[C1PRO ? 0 : u] = buf_load8<T>(xsrd, xvoff[u] + xt);
            }
            const unsigned dt = (unsigned)(h0 * W * Coutp * 2);
            const __amdgpu_buffer_rsrc_t zs = make_srd(zsg + (size_t)b * zimg, zimg * 2);
            if (DZ == DZ_POOL) {
                const __amdgpu_buffer_rsrc_t gs = make_srd(dg + (size_t)b * pimg, pimg * 2);
                const unsigned ptq = (unsigned)((h0 >> psh) * Wo * Coutp * 2);
#pragma unroll
                for (int u = 0; u < DIPT; ++u) {
                    r.a[u] = buf_load8<T>(gs, pvoff[u] + ptq);
                    r.b[u] = buf_load8<T>(zs, dvoff0 + (unsigned)(u * DQS * Coutp * 2) + dt);
                }
            } else {
                const __amdgpu_buffer_rsrc_t gs = make_srd(dg + (size_t)b * zimg, zimg * 2);
#pragma unroll
                for (int u = 0; u < DIPT; ++u) {
                    r.a[u] = buf_load8<T>(gs, dvoff0 + (unsigned)(u * DQS * Coutp * 2) + dt);
                    if (DZ == DZ_BN) r.b[u] = buf_load8<T>(zs, dvoff0 + (unsigned)(u * DQS * Coutp * 2) + dt);
                }
            }
        };

        auto commit = [&](const RawSet& r, int tile, T* __restrict__ st) {
            const bool live = tile < t_end;                // the pad tile of an odd strip: all zeros
            const int b = live ? WG_DIV_TPI(tile) : 0;
            const int h0 = live ? (tile - b * p.tilesPerImg) * TH : 0;
            const size_t zimg = zimg_;
            // ---- activations: prologue on load.  Rows outside the image must stay zero (relu(shift) is not):
            //      only the first / last tile of an image (and the pad tile) takes the masked path -----------
            const int row_lo = !live ? ROWS : (h0 == 0 ? 1 : 0);
            const int row_hi = (H - h0 < ROWS - 1) ? (H - h0) : (ROWS - 1);
            const bool boundary = (row_lo > 0) || (row_hi < ROWS - 1);
            if constexpr (C1PRO) {
                // (the consumer waves build the activation image from the input copy: build_next below)
            } else if (PRO == SED_PRO_NONE) {
#pragma unroll
                for (int u = 0; u < XIPT; ++u) {
                    if (u == XIPT - 1 && pt + u * NP >= XITEMS) break;
                    lds_store_raw<T>(st + xlds[u], r.x[C1PRO ? 0 : u]);      // hardware zeros for rows outside the image
                }
            } else {
                const f32x4* pc = reinterpret_cast<const f32x4*>(pcoef);
                auto pro_item = [&](int u, bool masked) {
                    const int it = pt + u * NP;
                    const int ci = it / XPER, rem = it - ci * XPER;
                    const int c4 = (ci * 32 + cq * 8) >> 2;
                    const f32x4 s0 = pc[c4], s1 = pc[c4 + 1], h0v = pc[CI_T * 8 + c4], h1v = pc[CI_T * 8 + c4 + 1];
                    bool keep = true;
                    if (masked) {
                        const int rowi = (rem >> 2) / W;
                        keep = rowi >= row_lo && rowi <= row_hi;
                    }
                    *reinterpret_cast<bf16x8*>(st + xlds[u]) = bnrelu8_bf16(r.x[C1PRO ? 0 : u].v, s0, s1, h0v, h1v, keep);
                };
                if (!boundary) {
#pragma unroll
                    for (int u = 0; u < XIPT; ++u) {
                        if (u == XIPT - 1 && pt + u * NP >= XITEMS) break;
                        pro_item(u, false);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < XIPT; ++u) {
                        if (u == XIPT - 1 && pt + u * NP >= XITEMS) break;
                        pro_item(u, true);
                    }
                }
            }
            // ---- dz: as stored, or produced here (BatchNorm / ReLU / pool backward) and written out -----------
            T* __restrict__ dzs = st + CI_T * XS1;
            const int qmax = live ? (H - h0) * W : 0;      // pixels of the tile inside the image (>= BM except on the last tile)
            const bool writer = live && dzo != nullptr && ((tile + cig) % NCI) == 0;
            const __amdgpu_buffer_rsrc_t os = make_srd(writer ? dzo + (size_t)b * zimg : nullptr, writer ? zimg * 2 : 0);
            const unsigned dt = (unsigned)(h0 * W * Coutp * 2);
#pragma unroll
            for (int u = 0; u < DIPT; ++u) {
                if (DZ == DZ_GIVEN) {
                    lds_store_raw<T>(dzs + dlds0 + u * DQS * 32, r.a[u]);     // rows past the image were read as zeros
                } else {
                    float g[8], z[8], v[8];
                    raw_to_f(r.a[u], g);
                    raw_to_f(r.b[u], z);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float base = fmaf(kcb[i], z[i], kcc[i]);          // cb*z + cc
                        const float full = fmaf(kca[i], g[i], base);            // + ca*g  (g is 0 where the pool floor dropped the pixel)
                        if (DZ == DZ_POOL) v[i] = (fmaf(z[i], ksc[i], ksh[i]) > 0.f) ? full : base;   // ReLU gate on g only
                        else v[i] = full;
                    }
#if SED_BOUNDARY_BRANCH
                    if (qmax < BM) {                              // only the last tile of an image (and the pad tile) has rows past it
                        asm volatile("" ::: "memory");      // (a real branch, see SED_BOUNDARY_BRANCH in conv_common.h)
                        const bool keep = dq0 + u * DQS < qmax;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = keep ? v[e] : 0.f;
                    }
#else
                    if (qmax < BM) {
                        const float m = (dq0 + u * DQS < qmax) ? 1.f : 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= m;
                    }
#endif
                    store8<T>(dzs + dlds0 + u * DQS * 32, v);
                    if (writer && !(SED_DBG(p, 1)))                   // rows past the image: dropped by the range check
                        buf_store8<T>(os, dvoff0 + (unsigned)(u * DQS * Coutp * 2) + dt, v);
                }
            }
        };

        RawSet ra, rb;
        issue(ra, t_begin);
        issue(rb, t_begin + 1);
        if (C1PRO) { issue_x1(ra, t_begin + 1); issue_x1(rb, t_begin + 2); }   // set (i & 1) carries the input copy of tile i+1
        for (int i = 0; i < ntl; i += 2) {           // tiles in pairs (an odd strip is padded with a zero tile)
            commit(ra, t_begin + i, stage0);
            if (C1PRO) { write_xt(ra, t_begin + i + 1); issue_x1(ra, t_begin + i + 3); }
            issue(ra, t_begin + i + 2);
            wg_barrier();
            commit(rb, t_begin + i + 1, stage0 + STAGE);
            if (C1PRO) { write_xt(rb, t_begin + i + 2); issue_x1(rb, t_begin + i + 4); }
            issue(rb, t_begin + i + 3);
            wg_barrier();
        }
    } else {
        // =============================== CONSUMERS =====================================================
        SED_SET_PRIO(p.dbg >> 10);
        const int hh = lane >> 5, r = lane & 31;
        // MODE_PAIR: wave = (cin tile, cout tile), taps 0..8;  MODE_ROW: wave = tap row, cout tiles 0..CO_T-1
        const int wci = (MODE == MODE_PAIR) ? (wave >> 1) : 0;
        const int wco = (MODE == MODE_PAIR) ? (wave & 1) : 0;
        const int wrow = (MODE == MODE_PAIR) ? 0 : wave;
        const bool active = (MODE == MODE_PAIR) || wave < 3;

        f32x16 acc[NACC];
#pragma unroll
        for (int t = 0; t < NACC; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

        // lane-constant parts of the transpose-read addresses: the lane supplies k-row 8*hh + q (+4 for the
        // second half) and the 4 channels 16*gbit + 4*pp .. +3
        int offA[3][2], offB[2];
        {
            const int i16 = lane & 15, gbit = (lane >> 4) & 1;
            const int qq = i16 >> 2, pp = i16 & 3, ch = 16 * gbit + 4 * pp;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int kl = 8 * hh + qq + 4 * half;
                const int rq = kl / W, cqq = kl % W;
#pragma unroll
                for (int tj = 0; tj < 3; ++tj)
                    offA[tj][half] = wci * XS1 + ((rq + wrow) * WP + cqq + tj) * 32 + (ch ^ swz<T>(cqq + tj));
                offB[half] = CI_T * XS1 + wco * DZ1 + kl * 32 + ch;
            }
        }

        // Software-pipelined by hand: the fragments of step s+2 are requested before the MFMAs of step s (a ring
        // of three fragment sets), so an LDS round trip hides behind two steps of MFMAs; the compiler's own
        // schedule drained lgkmcnt to 0 three times per 18 MFMAs.  Fully unrolled: every index is a constant.
        auto compute = [&](const T* __restrict__ st) {
            if (!active || (SED_DBG(p, 2))) return;
            constexpr int KS = BM / 16;
            if constexpr (MODE == MODE_PAIR) {
                constexpr int NS = KS * 3;          // step = (k-step, tap row): 3 MFMAs
                bf16x8 af[3][3], bfr[2];
                auto ld_af = [&](int s2, bf16x8 (&dst)[3]) {
                    const int k0 = (s2 / 3) * 16, ti = s2 % 3;
                    const int ub = ((k0 / W) * WP + (k0 % W)) * 32 + ti * WP * 32;
#pragma unroll
                    for (int tj = 0; tj < 3; ++tj)
                        dst[tj] = join_tr(ds_read_tr16_b64(st + ub + offA[tj][0]), ds_read_tr16_b64(st + ub + offA[tj][1]));
                };
                auto ld_bf = [&](int ks, bf16x8& dst) {
                    dst = join_tr(ds_read_tr16_b64(st + ks * 16 * 32 + offB[0]), ds_read_tr16_b64(st + ks * 16 * 32 + offB[1]));
                };
                ld_bf(0, bfr[0]);
                ld_af(0, af[0]);
                ld_af(1, af[1]);
#pragma unroll
                for (int s2 = 0; s2 < NS; ++s2) {
                    if (s2 + 2 < NS) ld_af(s2 + 2, af[(s2 + 2) % 3]);
                    if (s2 % 3 == 0 && s2 / 3 + 1 < KS) ld_bf(s2 / 3 + 1, bfr[(s2 / 3 + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);      // keep "reads of step s+2, then MFMAs of step s"
#pragma unroll
                    for (int tj = 0; tj < 3; ++tj)
                        acc[(s2 % 3) * 3 + tj] = mfma(af[s2 % 3][tj], bfr[(s2 / 3) & 1], acc[(s2 % 3) * 3 + tj]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                bf16x8 af[3][3], bfr[3][CO_T];      // step = k-step: 3*CO_T MFMAs
                auto ld = [&](int ks, bf16x8 (&a)[3], bf16x8 (&b)[CO_T]) {
                    const int k0 = ks * 16;
                    const int ub = ((k0 / W) * WP + (k0 % W)) * 32;
#pragma unroll
                    for (int tj = 0; tj < 3; ++tj)
                        a[tj] = join_tr(ds_read_tr16_b64(st + ub + offA[tj][0]), ds_read_tr16_b64(st + ub + offA[tj][1]));
#pragma unroll
                    for (int co = 0; co < CO_T; ++co)
                        b[co] = join_tr(ds_read_tr16_b64(st + co * DZ1 + k0 * 32 + offB[0]),
                                        ds_read_tr16_b64(st + co * DZ1 + k0 * 32 + offB[1]));
                };
                ld(0, af[0], bfr[0]);
                ld(1, af[1], bfr[1]);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks + 2 < KS) ld(ks + 2, af[(ks + 2) % 3], bfr[(ks + 2) % 3]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int co = 0; co < CO_T; ++co)
#pragma unroll
                        for (int tj = 0; tj < 3; ++tj)
                            acc[co * 3 + tj] = mfma(af[ks % 3][tj], bfr[ks % 3][co], acc[co * 3 + tj]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };

        // C1 mode: the consumer waves (one of them idle in row mode) build blocks 0..7 of the NEXT tile's activation image
        C1Mma c1mc;
        if (C1PRO) c1mma_init(c1mc, p.c1_w, p.pro_scale, p.pro_shift, lane);
        auto build_next = [&](int i2) __attribute__((always_inline)) {          // i2 = local tile index
            if (!C1PRO) return;
            const int tile = t_begin + i2;
            const bool live = tile < t_end;
            const int b = live ? WG_DIV_TPI(tile) : 0;
            const int h0 = live ? (tile - b * p.tilesPerImg) * TH : 0;
            // Rows 0 and 1 of the halo image are rows TH and TH+1 of the previous tile's image when that tile lies directly above
            // in the same image: copied through the LDS instead of rebuilt (a third of the blocks).
            bf16_t* img = stage0 + (i2 & 1) * STAGE;
            const bool reuse = live && i2 > 0 && h0 > 0;      // (workgroup-uniform)
            if (reuse) {
                const bf16_t* prev = stage0 + ((i2 - 1) & 1) * STAGE + TH * WP * 32;
                constexpr int NIT = 2 * WP * 32 / 8;          // 16-byte items of two rows
#pragma unroll
                for (int it = 0; it < (NIT + 255) / 256; ++it) {
                    const int q = wave * 64 + lane + 256 * it;
                    if (q < NIT) *reinterpret_cast<bf16x8*>(img + q * 8) = *reinterpret_cast<const bf16x8*>(prev + q * 8);
                }
            } else {
                c1_build_block_w<W, WP, XTW>(c1mc, xt0 + (i2 & 1) * XTN, img, wave, lane, h0, H, live);      // blocks 0..3 = rows 0, 1
            }
            // fresh rows 2 .. TH+1 (blocks 4 .. 2*ROWS-1).  Row mode keeps consumer wave 3 free of MFMA work: it takes the larger
            // share (1 / 1 / 1 / 5 of the 8 blocks: 0.472 ms; 2/2/2/2 0.498, 1/2/2/3 0.490, 0/0/1/7 0.521)
            constexpr int NF = 2 * ROWS - 4;
            const int first = 4 + ((MODE == MODE_ROW) ? (wave == 3 ? 3 : wave) : (NF / 4) * wave);
            const int cnt = (MODE == MODE_ROW) ? (wave == 3 ? NF - 3 : 1) : NF / 4;
            for (int blk = 0; blk < cnt; ++blk)
                if (first + blk < 2 * ROWS)
                    c1_build_block_w<W, WP, XTW>(c1mc, xt0 + (i2 & 1) * XTN, img, first + blk, lane, h0, H, live);
        };
        build_next(0);
        for (int i = 0; i < ntl; i += 2) {
            wg_barrier();
            compute(stage0);
            build_next(i + 1);
            wg_barrier();
            compute(stage0 + STAGE);
            build_next(i + 2);
        }

        // each wave stores its own slabs: D row = cin, col (lane) = cout
        if (active) {
            float* out = p.ws + (size_t)strip * 9 * Cinp * Coutp;
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                const int tap = (MODE == MODE_PAIR) ? t : wrow * 3 + (t % 3);
                const int co = (MODE == MODE_PAIR) ? wco : t / 3;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int cin = ci0 + wci * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    out[((size_t)tap * Cinp + cin) * Coutp + co0 + co * 32 + r] = acc[t][i];
                }
            }
        }
    }
}

struct Shape3 { int ci_t, co_t; };

bool shape3(int Cinp, int Coutp, Shape3* s) {
    if (Cinp == 32 && Coutp == 32) { *s = {1, 1}; return true; }
    if (Cinp == 32 && Coutp % 64 == 0) { *s = {1, 2}; return true; }
    if (Cinp % 64 == 0 && Coutp % 64 == 0) { *s = {2, 2}; return true; }
    return false;
}

template <int W, int CI_T, int CO_T, int DZ, int PRO, int NPW = 4>
int launch3n(Wgrad2Params& p, hipStream_t st) {
    constexpr int BM = (W == 64 && CI_T == 1) ? 256 : 128;
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr size_t stage = ((size_t)CI_T * (TH + 2) * WP * 32 + (size_t)CO_T * BM * 32) * sizeof(bf16_t);
    constexpr size_t lds = 2 * stage + (size_t)(5 * CO_T * 32 + 2 * CI_T * 32) * sizeof(float) +
                           (PRO == SED_PRO_C1 ? (size_t)2 * (TH + 4) * (W + 2) * sizeof(float) : 0);
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (int rc_ = sed_set_max_lds<&conv_wgrad3_kernel<W, CI_T, CO_T, DZ, PRO, NPW>>(lds)) return rc_;
    p.tilesPerImg = cdiv(p.H, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    sed_fastdiv_make((unsigned)p.tilesPerImg, &p.tpi_M, &p.tpi_l);
    p.tpb = cdiv(p.totalTiles, p.strips);
    const int ny = (p.Cinp / (32 * CI_T)) * (p.Coutp / (32 * CO_T));
    conv_wgrad3_kernel<W, CI_T, CO_T, DZ, PRO, NPW><<<dim3(p.strips * ny), dim3(256 + 64 * NPW), lds, st>>>(p);
    return 0;
}

template <int W, int CI_T, int CO_T, int DZ, int PRO>
int launch3(Wgrad2Params& p, hipStream_t st) {
#ifdef SED_EXPERIMENTS
    // Round-2 experiment, parity-green and measured NEUTRAL (block-0 weight gradient 0.580 vs 0.584 ms): eight loader waves
    // for the one-cin-tile / one-cout-tile variants (SED_WGRAD_PROD=8)
    if constexpr (CI_T == 1 && CO_T == 1) {
        const char* e = sed_getenv("SED_WGRAD_PROD");
        if (e && e[0] == '8') return launch3n<W, CI_T, CO_T, DZ, PRO, 8>(p, st);
    }
#endif
    return launch3n<W, CI_T, CO_T, DZ, PRO, 4>(p, st);
}

template <int W, int DZ, int PRO>
int dispatch3_shape(Wgrad2Params& p, const Shape3& s, hipStream_t st) {
    if (s.ci_t == 1 && s.co_t == 1) return launch3<W, 1, 1, DZ, PRO>(p, st);
    if (s.ci_t == 1 && s.co_t == 2) return launch3<W, 1, 2, DZ, PRO>(p, st);
    return launch3<W, 2, 2, DZ, PRO>(p, st);
}

template <int DZ, int PRO>
int dispatch3_w(Wgrad2Params& p, int W, const Shape3& s, hipStream_t st) {
    switch (W) {
        case 8: return dispatch3_shape<8, DZ, PRO>(p, s, st);
        case 16: return dispatch3_shape<16, DZ, PRO>(p, s, st);
        case 32: return dispatch3_shape<32, DZ, PRO>(p, s, st);
        case 64: return dispatch3_shape<64, DZ, PRO>(p, s, st);
    }
    return -1;
}

}  // namespace

int wgrad3_strips(int B, int H, int W, int Cinp, int Coutp) {
    Shape3 s;
    if (const int ws = wgrad_wide_strips(B, H, W, Cinp, Coutp)) return ws;      // the wide layers: csrc/sed_wgrad_wide.hip
    if (!(W == 8 || W == 16 || W == 32 || W == 64) || !shape3(Cinp, Coutp, &s)) return 0;
    if (const char* e = sed_getenv("SED_WGRAD_KERNEL")) if (e[0] == '2') return 0;     // A/B runs: force the previous kernel
    const int ny = (Cinp / (32 * s.ci_t)) * (Coutp / (32 * s.co_t));
    const int BM = (W == 64 && s.ci_t == 1) ? 256 : 128;
    const long long tiles = (long long)B * cdiv(H, BM / W);
    long long blocks = kWgrad3Blocks;
    if (const char* e = sed_getenv("SED_WGRAD_BLOCKS")) blocks = atoll(e) > 0 ? atoll(e) : blocks;   // tuning knob
    long long strips = blocks / ny;
    if (strips > tiles) strips = tiles;
    if (strips < 1) strips = 1;
    return (int)strips;
}

int launch_wgrad3(int dzmode, Wgrad2Params& p, int W, hipStream_t st) {
    Shape3 s;
    if (wgrad_wide_strips(p.B, p.H, W, p.Cinp, p.Coutp) > 0) {
        const int rc = launch_wgrad_wide(dzmode, p, W, st);
        if (rc >= 0) return rc;
    }
    if (!shape3(p.Cinp, p.Coutp, &s)) return -1;
    p.strips = wgrad3_strips(p.B, p.H, W, p.Cinp, p.Coutp);
    if (p.strips == 0) return -1;
    if (p.pro == SED_PRO_C1) {         // the conv2 weight gradient of the first block, z1 recomputed from the 1-channel input
        if (W != 64 || p.Cinp != 32 || p.Coutp != 32 || dzmode != DZ_POOL) return -1;   // (32 -> 64 would spill registers)
        return launch3<64, 1, 1, DZ_POOL, SED_PRO_C1>(p, st);
    }
    const bool pro = p.pro == SED_PRO_BNRELU;
    if (dzmode == DZ_GIVEN) return pro ? dispatch3_w<DZ_GIVEN, SED_PRO_BNRELU>(p, W, s, st) : dispatch3_w<DZ_GIVEN, SED_PRO_NONE>(p, W, s, st);
    if (dzmode == DZ_POOL) return pro ? dispatch3_w<DZ_POOL, SED_PRO_BNRELU>(p, W, s, st) : dispatch3_w<DZ_POOL, SED_PRO_NONE>(p, W, s, st);
    return pro ? dispatch3_w<DZ_BN, SED_PRO_BNRELU>(p, W, s, st) : dispatch3_w<DZ_BN, SED_PRO_NONE>(p, W, s, st);
}
