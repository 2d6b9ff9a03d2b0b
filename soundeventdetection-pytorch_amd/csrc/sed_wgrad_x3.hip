// Weight gradient of the 3x3 convolutions in split-operand arithmetic (dtype SED_F32H3, "f16x3"), producer / consumer form (gfx950).
//
//   dW[tap][cin][cout] = sum_pixels a[pixel + tap][cin] * dz[pixel][cout]
//
// autograd's conv weight gradient of ConvBlock (/root/reference/models/spectogram_models.py:132-140, backward of :155-156) with the
// BatchNorm / ReLU / avg-pool backward that produces dz fused in; fp32 tensors, every operand split into two fp16 pieces where it is
// staged, three fp16 MFMAs per product (csrc/x3_common.h, csrc/sed_conv_x3.hip for the arithmetic).
//
// Why a second kernel.  wgrad_x3_kernel (sed_conv_x3.hip) runs "all waves stage, barrier, three waves multiply" with two workgroups per
// CU; its timing ablations (profiles/r06_aa_x3_ablate.txt) show loads + barriers 36 %, staging 27 %, matrix loop 27 % of a launch
// nearly back to back, and every (32 cin x 32 cout) workgroup re-reads and re-splits its activation tile.  Here one 512-thread workgroup
// per CU splits the roles (conv_wgrad3_kernel's structure, csrc/sed_wgrad.hip):
//   * waves 4-7, one per SIMD: PRODUCERS.  Load, BatchNorm+ReLU prologue / dz arithmetic in fp32, split, write the hi / lo planes of
//     tile t+1 (XOR-swizzled transposing-read layout), store dz for the data-gradient call.  ONE register set of loads: an item's
//     registers are re-loaded for the next tile as soon as the item is staged, so every load has a whole tile period to arrive.
//   * waves 0-3: CONSUMERS, transposed LDS reads + MFMA only.  64 output channels: the 18 (tap, cout tile) products split 5 / 4 / 5 / 4
//     over the four waves; 32 output channels: one tap row per wave, wave 3 idles.
//   * ONE s_barrier per tile hands a double-buffered LDS stage over.
// A workgroup owns 64 output channels where the layer has them: the activation tile is loaded and split once per 64.
#include "x3_common.h"

#include <stdlib.h>

namespace {

constexpr int kX3pcBlocks = 256;      // one workgroup per CU

__device__ __forceinline__ void x3_barrier() {
    // LDS writes / reads of this wave are complete; global loads stay in flight across the barrier
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int W, int CO_T, int DZ, int PRO>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_x3pc_kernel(Wgrad2Params p) {
    typedef X3<true> XT;
    typedef typename XT::vec vec;
    constexpr int BM = 128;                           // pixels per tile (whole rows)
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr int ROWS = TH + 2;
    constexpr int XS1 = ROWS * WP * 32;               // one plane of the activation halo image (elements)
    constexpr int DZ1 = BM * 32;                      // one plane of one cout tile of the dz image
    constexpr int STAGE = 2 * XS1 + 2 * CO_T * DZ1;   // xh, xl, dh[CO_T], dl[CO_T]
    constexpr int NP = 256;                           // producer threads
    constexpr int XITEMS = ROWS * W * 4;              // 32-byte items of the activation tile (the two padding columns are zeroed once)
    constexpr int XIPT = (XITEMS + NP - 1) / NP;
    constexpr int IPP = CO_T * 4;                     // items per dz pixel
    constexpr int DITEMS = BM * IPP;
    constexpr int DIPT = DITEMS / NP;
    constexpr int DQS = NP / IPP;                     // pixels between two dz items of a thread
    static_assert(DITEMS % NP == 0 && NP % IPP == 0, "dz item geometry");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    u16_t* stage0 = reinterpret_cast<u16_t*>(smem);   // [2][STAGE]
    float* coef = reinterpret_cast<float*>(stage0 + 2 * STAGE);   // [5][CO_T*32]: scale, shift, ca, cb, cc
    float* pcoef = coef + 5 * CO_T * 32;              // [2][32]: prologue scale, shift

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.H, Cinp = p.Cinp, Coutp = p.Coutp;
    const int NCI = Cinp / 32, NCO = Coutp / (32 * CO_T);
    const int NY = NCI * NCO;
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int strip = logical / NY, yb = logical - strip * NY;
    const int cig = yb / NCO;
    const int ci0 = cig * 32, co0 = (yb % NCO) * 32 * CO_T;
    const int psh = p.pool >> 1;                      // pool is 1 or 2
    const int Ho = H >> psh, Wo = W >> psh;

    if (DZ != DZ_GIVEN) {
        const float inv_pool = psh ? 0.25f : 1.0f;
        for (int i = tid; i < 5 * CO_T * 32; i += 512) {
            const int a = i / (CO_T * 32), c = i - a * (CO_T * 32);
            const float* src = (a == 0) ? p.scale : (a == 1) ? p.shift : (a == 2) ? p.ca : (a == 3) ? p.cb : p.cc;
            float v = (src != nullptr) ? src[co0 + c] : 0.f;
            if (a == 2 && DZ == DZ_POOL) v *= inv_pool;   // the 1/pool^2 of the avg-pool backward folded into ca
            coef[i] = v;
        }
    }
    {   // the two padding columns of every halo row stay zero for the whole kernel (both stages, both planes)
        constexpr int NPAD = 2 * 2 * ROWS * 2 * 4;
        for (int i = tid; i < NPAD; i += 512) {
            const int c16 = i & 3, side = (i >> 2) & 1, rowi = (i >> 3) % ROWS, pl = ((i >> 3) / ROWS) & 1, sg = (i >> 3) / (ROWS * 2);
            const sed_u32x4 z4 = {0u, 0u, 0u, 0u};
            *reinterpret_cast<sed_u32x4*>(stage0 + sg * STAGE + pl * XS1 + (rowi * WP + (side ? W + 1 : 0)) * 32 + c16 * 8) = z4;
        }
    }
    if (PRO == SED_PRO_BNRELU) {
        for (int i = tid; i < 64; i += 512) pcoef[i] = (i < 32 ? p.pro_scale : p.pro_shift)[ci0 + (i & 31)];
    }
    __syncthreads();

    const int t_begin = strip * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    const int ntl = t_end > t_begin ? t_end - t_begin : 0;

    if (wave >= 4) {
        // =============================== PRODUCERS =====================================================
        const float* __restrict__ xg = reinterpret_cast<const float*>(p.x);
        const float* __restrict__ dg = reinterpret_cast<const float*>(p.dz);
        const float* __restrict__ zsg = reinterpret_cast<const float*>(p.zsrc);
        float* __restrict__ dzo = reinterpret_cast<float*>(p.dz_out);
        const int pt = tid - 256;
        const int cq = pt & 3;
        const size_t ximg_ = (size_t)H * W * Cinp, zimg_ = (size_t)H * W * Coutp, pimg_ = (size_t)Ho * Wo * Coutp;
        const float pre = __builtin_ldexpf(1.f, p.dzexp);

        // tile-invariant item plans
        unsigned xvoff[XIPT];
        int xlds[XIPT];
#pragma unroll
        for (int u = 0; u < XIPT; ++u) {
            const int it = pt + u * NP;
            const int pix = it >> 2;
            const int rowi = pix / W, coli = pix - rowi * W + 1;
            const bool ok = it < XITEMS;
            xvoff[u] = ok ? (unsigned)(((rowi * W + coli) * Cinp + cq * 8) * 4) : SED_OOB;
            xlds[u] = ok ? (rowi * WP + coli) * 32 + ((cq * 8) ^ swz<bf16_t>(coli)) : 0;
        }
        const int dq0 = pt / IPP, dc8 = (pt - dq0 * IPP) * 8;       // first dz pixel and the (fixed) channel group
        const unsigned dvoff0 = (unsigned)((dq0 * Coutp + co0 + dc8) * 4);
        const int dlds0 = (dc8 >> 5) * DZ1 + dq0 * 32 + (dc8 & 31);
        unsigned pvoff[DIPT];
#pragma unroll
        for (int u = 0; u < DIPT; ++u) {
            const int q = dq0 + u * DQS;
            pvoff[u] = (unsigned)(((((q / W) >> psh) * Wo + ((q % W) >> psh)) * Coutp + co0 + dc8) * 4);
        }
        // coefficients of the thread's fixed channel groups, in registers
        float kca[8], kcb[8], kcc[8], ksc[8], ksh[8], psc[8], psf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = dc8 + e;
            kca[e] = DZ != DZ_GIVEN ? coef[2 * CO_T * 32 + c] : 0.f; kcb[e] = DZ != DZ_GIVEN ? coef[3 * CO_T * 32 + c] : 0.f;
            kcc[e] = DZ != DZ_GIVEN ? coef[4 * CO_T * 32 + c] : 0.f;
            ksc[e] = DZ == DZ_POOL ? coef[c] : 0.f; ksh[e] = DZ == DZ_POOL ? coef[CO_T * 32 + c] : 0.f;
            psc[e] = PRO == SED_PRO_BNRELU ? pcoef[cq * 8 + e] : 0.f; psf[e] = PRO == SED_PRO_BNRELU ? pcoef[32 + cq * 8 + e] : 0.f;
        }

        // DEPTH register sets of loads in flight: one where a workgroup owns 64 output channels (no registers for more), two at 32 (the
        // HBM-bound 32 -> 32 layer of block 0 wants the bytes in flight: 1.37 ms with one set against 1.26 for the all-waves kernel)
        constexpr int DEPTH = CO_T == 1 ? 2 : 1;
        struct RawSet { Raw8<float> x[XIPT], a[DIPT], b[DIPT]; };
        RawSet rs[DEPTH];
        // descriptors and offsets of tile `tile` (a tile past the strip: zero-sized descriptors -- every lane out of range, zeros, no traffic;
        // loads are issued unconditionally so that the compiler's vmcnt bookkeeping stays exact)
        struct TileSrc { __amdgpu_buffer_rsrc_t xs, gs, zs; unsigned xt, dt, ptq; };
        auto tile_src = [&](int tile) -> TileSrc {
            const bool live = tile < t_end;
            const int b = live ? sed_fastdiv(tile, p.tpi_M, p.tpi_l) : 0;
            const int h0 = live ? (tile - b * p.tilesPerImg) * TH : 0;
            const size_t ximg = live ? ximg_ : 0, zimg = live ? zimg_ : 0, pimg = live ? pimg_ : 0;
            TileSrc s;
            s.xs = make_srd(xg + (size_t)b * ximg, ximg * 4);
            s.xt = (unsigned)((((h0 - 1) * W - 1) * Cinp + ci0) * 4);
            s.dt = (unsigned)(h0 * W * Coutp * 4);
            s.ptq = (unsigned)((h0 >> psh) * Wo * Coutp * 4);
            s.zs = make_srd(zsg + (size_t)b * zimg, (DZ != DZ_GIVEN) ? zimg * 4 : 0);
            s.gs = (DZ == DZ_POOL) ? make_srd(dg + (size_t)b * pimg, pimg * 4) : make_srd(dg + (size_t)b * zimg, zimg * 4);
            return s;
        };
        auto load_x = [&](RawSet& r, const TileSrc& s, int u) __attribute__((always_inline)) { r.x[u] = buf_load8<float>(s.xs, xvoff[u] + s.xt); };
        auto load_d = [&](RawSet& r, const TileSrc& s, int u) __attribute__((always_inline)) {
            const unsigned o = dvoff0 + (unsigned)(u * DQS * Coutp * 4) + s.dt;
            if (DZ == DZ_POOL) { r.a[u] = buf_load8<float>(s.gs, pvoff[u] + s.ptq); r.b[u] = buf_load8<float>(s.zs, o); }
            else if (DZ == DZ_BN) { r.a[u] = buf_load8<float>(s.gs, o); r.b[u] = buf_load8<float>(s.zs, o); }
            else r.a[u] = buf_load8<float>(s.gs, o);
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const TileSrc s0 = tile_src(t_begin + d);
#pragma unroll
            for (int u = 0; u < XIPT; ++u) load_x(rs[d], s0, u);
#pragma unroll
            for (int u = 0; u < DIPT; ++u) load_d(rs[d], s0, u);
        }
        // tile i of the strip: stage it from register set r, re-load each item for tile i + DEPTH as soon as it is staged
        auto stage_tile = [&](int i, RawSet& r) __attribute__((always_inline)) {
            Raw8<float>(&rx)[XIPT] = r.x;
            Raw8<float>(&ra)[DIPT] = r.a;
            Raw8<float>(&rb)[DIPT] = r.b;
            const int tile = t_begin + i;
            u16_t* __restrict__ st = stage0 + (i & 1) * STAGE;
            const int b = sed_fastdiv(tile, p.tpi_M, p.tpi_l);
            const int h0 = (tile - b * p.tilesPerImg) * TH;
            const TileSrc nx = tile_src(tile + DEPTH);
            // ---- activations: prologue on load; rows outside the image must be zero AFTER it --------------------------------------
            const int row_lo = h0 == 0 ? 1 : 0;
            const int row_hi = (H - h0 < ROWS - 1) ? (H - h0) : (ROWS - 1);
#pragma unroll
            for (int u = 0; u < XIPT; ++u) {
                const int it = pt + u * NP;
                if (u == XIPT - 1 && it >= XITEMS) { load_x(r, nx, u); continue; }      // (the re-load keeps the load count per tile fixed)
                float v[8];
                raw_to_f(rx[u], v);
                load_x(r, nx, u);                          // this item's registers are free: the next tile's load has a tile period to arrive
                if (PRO == SED_PRO_BNRELU) {
                    const int rowi = (it >> 2) / W;
                    const float top = (rowi >= row_lo && rowi <= row_hi) ? __builtin_inff() : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(fmaf(v[e], psc[e], psf[e]), 0.f, top);
                }
                if (kX3Stamps && (p.dbg & 4)) continue;      // (STAMPS build, SED_DBG & 4: loads only -- timing ablation)
                sed_u32x4 hw, lw;
                split8<true, false>(v, hw, lw, 1.f);
                *reinterpret_cast<sed_u32x4*>(st + xlds[u]) = hw;
                *reinterpret_cast<sed_u32x4*>(st + XS1 + xlds[u]) = lw;
            }
            // ---- dz: as stored, or produced here (BatchNorm / ReLU / pool backward) and written out for the data-gradient call -------
            u16_t* __restrict__ dzh = st + 2 * XS1;
            const int qmax = (H - h0) * W;              // pixels of the tile inside the image (>= BM except on an image's last tile)
            // the dz tile is produced by every cin group of the strip; they take turns (by tile parity) writing it out
            const bool writer = dzo != nullptr && ((tile + cig) % NCI) == 0;
            const __amdgpu_buffer_rsrc_t os = make_srd(writer ? dzo + (size_t)b * zimg_ : nullptr, writer ? zimg_ * 4 : 0);
            const unsigned dt = (unsigned)(h0 * W * Coutp * 4);
#pragma unroll
            for (int u = 0; u < DIPT; ++u) {
                float v[8];
                if (DZ == DZ_GIVEN) {
                    raw_to_f(ra[u], v);                 // rows past the image were read as zeros
                    load_d(r, nx, u);
                } else {
                    float g[8], z[8];
                    raw_to_f(ra[u], g);
                    raw_to_f(rb[u], z);
                    load_d(r, nx, u);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float base = fmaf(kcb[e], z[e], kcc[e]);          // cb*z + cc
                        const float full = fmaf(kca[e], g[e], base);            // + ca*g
                        if (DZ == DZ_POOL) v[e] = (fmaf(z[e], ksc[e], ksh[e]) > 0.f) ? full : base;   // ReLU gate on g only
                        else v[e] = full;
                    }
                    if (qmax < BM && dq0 + u * DQS >= qmax) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = 0.f;
                    }
                    if (writer) buf_store8<float>(os, dvoff0 + (unsigned)(u * DQS * Coutp * 4) + dt, v);   // rows past the image: dropped by the range check
                }
                if (kX3Stamps && (p.dbg & 4)) continue;
                sed_u32x4 hw, lw;
                split8<true, true>(v, hw, lw, pre);
                *reinterpret_cast<sed_u32x4*>(dzh + dlds0 + u * DQS * 32) = hw;
                *reinterpret_cast<sed_u32x4*>(dzh + CO_T * DZ1 + dlds0 + u * DQS * 32) = lw;
            }
            x3_barrier();
        };
        for (int i = 0; i < ntl; i += DEPTH) {
            stage_tile(i, rs[0]);
            if constexpr (DEPTH == 2) {
                if (i + 1 < ntl) stage_tile(i + 1, rs[1]);
            }
        }
    } else {
        // =============================== CONSUMERS =====================================================
        // CO_T = 1: wave = tap row (3 taps), wave 3 idle.  CO_T = 2: the 18 (tap, cout tile) products split 5 / 4 / 5 / 4 over the four
        // waves -- wave = (cout tile, taps 0..4 | 5..8) -- so that every SIMD multiplies (three tap-row waves of six tiles left one idle).
        const int hh = lane >> 5, r = lane & 31;
        constexpr int NT = CO_T == 2 ? 5 : 3;                      // accumulator tiles of a wave (the 4-tap waves leave one unused)
        const int wco = CO_T == 2 ? (wave >> 1) : 0;
        const int tap0 = CO_T == 2 ? ((wave & 1) ? 5 : 0) : 3 * wave;
        const int ntap = CO_T == 2 ? ((wave & 1) ? 4 : 5) : 3;
        const bool active = CO_T == 2 || wave < 3;
        f32x16 ach[NT], acx[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) { ach[t][i] = 0.f; acx[t][i] = 0.f; }

        // lane-constant parts of the transpose-read addresses: the lane supplies k-row 8*hh + q (+4 for the second half) and the 4
        // channels 16*gbit + 4*pp .. +3; tile t of the wave is tap tap0 + t = (row ti, column tj)
        int offA[NT][2], offB[2];
        {
            const int i16 = lane & 15, gbit = (lane >> 4) & 1;
            const int qq = i16 >> 2, pp = i16 & 3, ch = 16 * gbit + 4 * pp;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int kl = 8 * hh + qq + 4 * half;
                const int rq = kl / W, cqq = kl % W;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int tap = (active && t < ntap) ? tap0 + t : 0;
                    const int ti = tap / 3, tj = tap - 3 * ti;
                    offA[t][half] = ((rq + ti) * WP + cqq + tj) * 32 + (ch ^ swz<bf16_t>(cqq + tj));
                }
                offB[half] = 2 * XS1 + wco * DZ1 + kl * 32 + ch;
            }
        }
        // Eight k-steps of 3 NT MFMAs, software-pipelined without a second fragment set: the products run as ah.bl, ah.bh, al.bh and a
        // group's dead fragments are re-read for the next k-step as soon as the group has issued (>= NT MFMAs before their first use).
        auto compute = [&](const u16_t* __restrict__ st) __attribute__((always_inline)) {
            constexpr int KS = BM / 16;
            auto ld_a = [&](int pl, int ks, int t) __attribute__((always_inline)) {
                const int k0 = ks * 16, ub = pl * XS1 + ((k0 / W) * WP + (k0 % W)) * 32;
                return lds_frag_tr<vec>(st + ub + offA[t][0], st + ub + offA[t][1]);
            };
            auto ld_b = [&](int pl, int ks) __attribute__((always_inline)) {
                const int ub = pl * CO_T * DZ1 + ks * 16 * 32;
                return lds_frag_tr<vec>(st + ub + offB[0], st + ub + offB[1]);
            };
            vec ah[NT], al[NT], bh = ld_b(0, 0), bl = ld_b(1, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) { ah[t] = ld_a(0, 0, t); al[t] = ld_a(1, 0, t); }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bool more = ks + 1 < KS;
#pragma unroll
                for (int t = 0; t < NT; ++t) acx[t] = XT::mfma(ah[t], bl, acx[t]);
                __builtin_amdgcn_sched_barrier(0);
                if (more) bl = ld_b(1, ks + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t) ach[t] = XT::mfma(ah[t], bh, ach[t]);
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) ah[t] = ld_a(0, ks + 1, t);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t) acx[t] = XT::mfma(al[t], bh, acx[t]);
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    bh = ld_b(0, ks + 1);
#pragma unroll
                    for (int t = 0; t < NT; ++t) al[t] = ld_a(1, ks + 1, t);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int i = 0; i < ntl; ++i) {
            x3_barrier();
            if (active && !(kX3Stamps && (p.dbg & 2))) compute(stage0 + (i & 1) * STAGE);      // (STAMPS build, SED_DBG & 2: no matrix loop -- timing ablation)
        }
        if (active) {        // each wave stores its own slabs: D row = cin, col (lane) = cout
            const float post_x = __builtin_ldexpf(XT::ILS, -p.dzexp), post_h = __builtin_ldexpf(1.f, -p.dzexp);
            float* out = p.ws + (size_t)strip * 9 * Cinp * Coutp;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t >= ntap) break;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int cin = ci0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    out[((size_t)(tap0 + t) * Cinp + cin) * Coutp + co0 + wco * 32 + r] = fmaf(acx[t][i], post_x, ach[t][i] * post_h);
                }
            }
        }
    }
}

template <int W, int CO_T, int DZ, int PRO>
int launch_x3pc(Wgrad2Params& p, hipStream_t st) {
    constexpr int TH = 128 / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr size_t stage = ((size_t)2 * (TH + 2) * WP * 32 + (size_t)2 * CO_T * 128 * 32) * 2;
    constexpr size_t lds = 2 * stage + (size_t)(5 * CO_T * 32 + 64) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (int rc_ = sed_set_max_lds<&wgrad_x3pc_kernel<W, CO_T, DZ, PRO>>(lds)) return rc_;
    p.tilesPerImg = cdiv(p.H, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    sed_fastdiv_make((unsigned)p.tilesPerImg, &p.tpi_M, &p.tpi_l);
    const int ny = (p.Cinp / 32) * (p.Coutp / (32 * CO_T));
    // one workgroup per CU; never more strips than the caller's workspace holds slabs for (p.strips on entry: sed_conv_wgrad_ws_floats' count)
    int strips = kX3pcBlocks / ny;
    if (strips > p.strips) strips = p.strips;
    if (strips > p.totalTiles) strips = p.totalTiles;
    if (strips < 1) strips = 1;
    p.strips = strips;
    p.tpb = cdiv(p.totalTiles, p.strips);
    wgrad_x3pc_kernel<W, CO_T, DZ, PRO><<<dim3(p.strips * ny), dim3(512), lds, st>>>(p);
    return 0;
}

template <int W, int DZ, int PRO>
int dispatch_x3pc_co(Wgrad2Params& p, hipStream_t st) {
    return p.Coutp % 64 == 0 ? launch_x3pc<W, 2, DZ, PRO>(p, st) : launch_x3pc<W, 1, DZ, PRO>(p, st);
}

template <int DZ, int PRO>
int dispatch_x3pc_w(Wgrad2Params& p, int W, hipStream_t st) {
    switch (W) {
        case 8: return dispatch_x3pc_co<8, DZ, PRO>(p, st);
        case 16: return dispatch_x3pc_co<16, DZ, PRO>(p, st);
        case 32: return dispatch_x3pc_co<32, DZ, PRO>(p, st);
        case 64: return dispatch_x3pc_co<64, DZ, PRO>(p, st);
    }
    return -1;
}

}  // namespace

// Returns -1 when the shape is not covered (the caller then takes wgrad_x3_kernel), otherwise 0 / an error code after the launch.
// SED_X3_WGRAD=a: the all-waves kernel everywhere (A/B runs).
int launch_wgrad_x3pc(int dzmode, Wgrad2Params& p, int W, hipStream_t st) {
    if (const char* e = sed_getenv("SED_X3_WGRAD")) if (e[0] == 'a') return -1;
    if (!(p.pro == SED_PRO_NONE || p.pro == SED_PRO_BNRELU)) return -1;
    // block 0's 32 -> 32 layer at W = 64 is HBM-bound and has nothing to share: two workgroups per CU of the all-waves kernel keep more
    // bytes in flight (1.19 ms against 1.25 here with two register sets, 1.37 with one: profiles/r06_af_*)
    if (W == 64 && p.Coutp % 64 != 0 && !(sed_getenv("SED_X3_WGRAD") && sed_getenv("SED_X3_WGRAD")[0] == 'p')) return -1;
    const bool pro = p.pro == SED_PRO_BNRELU;
    if (dzmode == DZ_GIVEN) return pro ? dispatch_x3pc_w<DZ_GIVEN, SED_PRO_BNRELU>(p, W, st) : dispatch_x3pc_w<DZ_GIVEN, SED_PRO_NONE>(p, W, st);
    if (dzmode == DZ_POOL) return pro ? dispatch_x3pc_w<DZ_POOL, SED_PRO_BNRELU>(p, W, st) : dispatch_x3pc_w<DZ_POOL, SED_PRO_NONE>(p, W, st);
    return pro ? dispatch_x3pc_w<DZ_BN, SED_PRO_BNRELU>(p, W, st) : dispatch_x3pc_w<DZ_BN, SED_PRO_NONE>(p, W, st);
}
