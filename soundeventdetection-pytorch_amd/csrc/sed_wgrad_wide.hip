// Weight gradient of the 3x3 convolutions for the WIDE layers (>= 128 channels on one side, >= 64 on the other), bf16, gfx950.
//
//   dW[tap][cin][cout] = sum_pixels a[pixel + tap][cin] * dz[pixel][cout]
//
// autograd's conv weight gradient of ConvBlock (/root/reference/models/spectogram_models.py:132-140, backward of :155-156 under
// /root/reference/train.py:102), with the BatchNorm / ReLU / avg-pool backward that produces dz fused in -- the same contract as
// conv_wgrad3_kernel (csrc/sed_wgrad.hip), which keeps the narrow layers.
//
// Why a second form (round 5).  conv_wgrad3_kernel gives a workgroup (64 cin x 64 cout) x 9 taps: four MFMA waves with 144
// accumulator registers each and four loader waves.  On a 128 -> 128 layer that is four workgroups per pixel strip, each streaming
// its half of x and producing its half of dz, and the loader waves -- one per SIMD, issuing beside an MFMA wave at ~15 cycles per
// instruction -- bound the launch at 0.3 of the matrix roof (profiles/r04_d_ab_wgrad_dz_given.txt, r04_l_ab_wgrad_eight_loader_waves.txt).
// Here a workgroup owns (CI_T*32 cin x CO_T*32 cout) x 9 taps with CI_T*CO_T = 8 -- (128 x 64) or (64 x 128) -- and ALL EIGHT waves
// are MFMA issuers (one (32 x 32) x 9-tap pair each) AND loaders:
//   * per 128-pixel tile a wave runs a matrix phase (72 MFMAs, operand fragments by ds_read_b64_tr_b16) and a vector phase (its
//     share of the next tile: BN+ReLU prologue of x, dz = BatchNorm / ReLU / pool backward of (g, z), dz_out store, LDS images) and
//     issues the global loads of the tile after that into registers (zero-sized descriptors past the strip: vmcnt stays exact);
//   * waves 0-3 run matrix -> vector, waves 4-7 vector -> matrix: the two waves of a SIMD are in opposite phases, so the
//     vector work issues at its own rate beside the partner's MFMAs instead of behind a second instruction stream of its kind;
//   * ONE s_barrier per tile hands the double-buffered LDS stage over;
//   * half the strip-redundant work: x is streamed by Cout/(32 CO_T) workgroups and dz produced by Cin/(32 CI_T) (1 x 2 instead of
//     2 x 2 on 128 -> 128; 1 x 1 on 64 -> 128);
//   * operand fragments are reused from registers across the taps: walking the halo rows rho of a 16-pixel column strip, the x
//     fragment of (rho, column shift tj) serves the tap rows ti = 0, 1, 2 against the dz fragments of rows rho, rho - 1, rho - 2
//     (a ring of registers): 76-88 transposed reads per 72 MFMAs instead of 160.  W = 8: a k-step is the row PAIR (s, s + 8), so
//     the same walk applies (and a 32-lane read group covers four consecutive pixels of one row: conflict-free).
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

constexpr int kWideBlocks = 256;     // one workgroup per CU

// in-kernel phase stamps (make STAMPS=1 in a scratch copy of the tree: tools/wide_stamp.sh); the product build has none
#ifdef SED_STAMPS
constexpr bool kWideStamps = true;
#else
constexpr bool kWideStamps = false;
#endif
#ifndef SED_WIDE_PRIO
#define SED_WIDE_PRIO 0              // A/B builds: s_setprio 1 during a wave's vector phase
#endif

__device__ __forceinline__ void wide_barrier() {
    // LDS writes / reads of this wave are complete; global loads stay in flight across the barrier
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int W, int CI_T, int DZ, int PRO>
__global__ __launch_bounds__(512) void conv_wgrad_wide_kernel(Wgrad2Params p) {
    typedef bf16_t T;
    constexpr int CO_T = 8 / CI_T;
    static_assert(CI_T == 2 || CI_T == 4, "eight (32 x 32) pairs per workgroup");
    constexpr int BM = 128;                           // pixels per tile (whole rows)
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr int ROWS = TH + 2;
    constexpr int XS1 = ROWS * WP * 32 + 32;          // one cin tile of the activation halo image; + 64 B: the ci tiles of a pixel on different banks
    constexpr int DZ1 = BM * 32 + 32;                 // one cout tile of the dz image
    constexpr int STAGE = CI_T * XS1 + CO_T * DZ1;
    constexpr int NTHR = 512;
    // x items (16 bytes): a pixel's CI_T*32 channels are LPP consecutive lanes (coalesced), PPR pixels per round of the workgroup
    constexpr int LPP = CI_T * 4, PPR = NTHR / LPP, XPIX = ROWS * W, XIPT = (XPIX + PPR - 1) / PPR;
    static_assert(PPR % W == 0, "a round of x items is whole rows");
    // dz items: IPP per pixel, DQS pixels between two items of a thread
    constexpr int IPP = CO_T * 4, DIPT = BM * IPP / NTHR, DQS = NTHR / IPP;
    static_assert(DQS % W == 0 && BM * IPP % NTHR == 0, "dz item geometry");
    // k-steps: 16 pixels.  W >= 16: 16 columns of one row (NCS column strips per row); W = 8: the rows (s, s + 8)
    constexpr int NCS = W >= 16 ? W / 16 : 1;
    constexpr int NR = W >= 16 ? TH : TH / 2;          // k-steps per column strip
    static_assert(NCS * NR == BM / 16, "k-steps");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* stage0 = reinterpret_cast<T*>(smem);           // [2][STAGE]: xs[CI_T][ROWS][WP][32], dzs[CO_T][BM][32]
    float* coef = reinterpret_cast<float*>(stage0 + 2 * STAGE);   // [5][CO_T*32]: scale, shift, ca, cb, cc
    float* pcoef = coef + 5 * CO_T * 32;              // [2][CI_T*32]: prologue scale, shift

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

    // ---- coefficients into LDS; the two padding columns of every halo row stay zero for the whole kernel (both stages) ----------
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
    {
        constexpr int NPAD = 2 * CI_T * ROWS * 2 * 4;
        for (int i = tid; i < NPAD; i += NTHR) {
            const int c16 = i & 3, side = (i >> 2) & 1, rowi = (i >> 3) % ROWS, ci = ((i >> 3) / ROWS) % CI_T, sg = (i >> 3) / (ROWS * CI_T);
            bf16x8 z8;
#pragma unroll
            for (int e = 0; e < 8; ++e) z8[e] = (bf16_t)0.f;
            *reinterpret_cast<bf16x8*>(stage0 + sg * STAGE + ci * XS1 + (rowi * WP + (side ? W + 1 : 0)) * 32 + c16 * 8) = z8;
        }
    }
    if (PRO == SED_PRO_BNRELU) {
        for (int i = tid; i < 2 * CI_T * 32; i += NTHR) {
            const int a = i / (CI_T * 32), c = i - a * (CI_T * 32);
            pcoef[i] = (a == 0 ? p.pro_scale : p.pro_shift)[ci0 + c];
        }
    }
    __syncthreads();

    const int t_begin = strip * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    const int ntl = t_end > t_begin ? t_end - t_begin : 0;

    // =============================== loader half of every wave ===========================================================
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ dg = reinterpret_cast<const T*>(p.dz);
    const T* __restrict__ zsg = reinterpret_cast<const T*>(p.zsrc);
    T* __restrict__ dzo = reinterpret_cast<T*>(p.dz_out);      // the cin groups of a strip take turns (by tile) writing the dz tile out
    const size_t ximg_ = (size_t)H * W * Cinp, zimg_ = (size_t)H * W * Coutp, pimg_ = (size_t)Ho * Wo * Coutp;

    // tile-invariant item plans: item u of the thread = item 0 + u whole rounds (a uniform byte / LDS step)
    const int xpx0 = tid / LPP, xci = (tid % LPP) >> 2, xcq = tid & 3;
    const int xrow0 = xpx0 / W, xcol = xpx0 - xrow0 * W + 1;
    const unsigned xvoff0 = (unsigned)(((xrow0 * W + xcol) * Cinp + xci * 32 + xcq * 8) * 2);
    const unsigned xustep = (unsigned)(PPR * Cinp * 2);
    constexpr int XLSTEP = (PPR / W) * WP * 32;
    const int xlds0 = xci * XS1 + (xrow0 * WP + xcol) * 32 + xcq * 8;
    const bool xlast_ok = xpx0 + (XIPT - 1) * PPR < XPIX;      // the last round may be partial

    const int dq0 = tid / IPP, dc8 = (tid - dq0 * IPP) * 8;    // first dz pixel and the (fixed) channel group
    const unsigned dvoff0 = (unsigned)((dq0 * Coutp + co0 + dc8) * 2);
    const unsigned dustep = (unsigned)(DQS * Coutp * 2);
    const int dlds0 = (dc8 >> 5) * DZ1 + dq0 * 32 + (dc8 & 31);
    unsigned pvoff[DIPT];
#pragma unroll
    for (int u = 0; u < DIPT; ++u) {
        const int q = dq0 + u * DQS;
        pvoff[u] = (unsigned)(((((q / W) >> psh) * Wo + ((q % W) >> psh)) * Coutp + co0 + dc8) * 2);
    }

    struct RawSet { Raw8<T> x[XIPT]; Raw8<T> a[DIPT]; Raw8<T> b[DIPT]; };

    // Loads are issued UNCONDITIONALLY (a tile past the strip gets zero-sized descriptors: zeros, no traffic), so that the
    // compiler's vmcnt bookkeeping stays exact (csrc/sed_wgrad.hip)
    auto issue = [&](RawSet& r, int tile) __attribute__((always_inline)) {
        const bool live = tile < t_end;
        const int b = live ? WG_DIV_TPI(tile) : 0;
        const int h0 = live ? (tile - b * p.tilesPerImg) * TH : 0;
        const size_t ximg = live ? ximg_ : 0, zimg = live ? zimg_ : 0, pimg = live ? pimg_ : 0;
        {
            const __amdgpu_buffer_rsrc_t xsrd = make_srd(xg + (size_t)b * ximg, ximg * 2);
            const unsigned xt = (unsigned)((((h0 - 1) * W - 1) * Cinp + ci0) * 2);
#pragma unroll
            for (int u = 0; u < XIPT; ++u) {
                unsigned vo = xvoff0 + (xt + (unsigned)u * xustep);
                if (u == XIPT - 1 && XPIX % PPR != 0) vo = xlast_ok ? vo : SED_OOB;
                r.x[u] = buf_load8<T>(xsrd, vo);
            }
        }
        const unsigned dt = (unsigned)(h0 * W * Coutp * 2);
        const __amdgpu_buffer_rsrc_t zs = make_srd(zsg + (size_t)b * zimg, zimg * 2);
        if (DZ == DZ_POOL) {
            const __amdgpu_buffer_rsrc_t gs = make_srd(dg + (size_t)b * pimg, pimg * 2);
            const unsigned ptq = (unsigned)((h0 >> psh) * Wo * Coutp * 2);
#pragma unroll
            for (int u = 0; u < DIPT; ++u) {
                r.a[u] = buf_load8<T>(gs, pvoff[u] + ptq);
                r.b[u] = buf_load8<T>(zs, dvoff0 + (dt + (unsigned)u * dustep));
            }
        } else {
            const __amdgpu_buffer_rsrc_t gs = make_srd(dg + (size_t)b * zimg, zimg * 2);
#pragma unroll
            for (int u = 0; u < DIPT; ++u) {
                r.a[u] = buf_load8<T>(gs, dvoff0 + (dt + (unsigned)u * dustep));
                if (DZ == DZ_BN) r.b[u] = buf_load8<T>(zs, dvoff0 + (dt + (unsigned)u * dustep));
            }
        }
    };

    auto commit = [&](const RawSet& r, int tile, T* __restrict__ st) __attribute__((always_inline)) {
        const bool live = tile < t_end;                // a tile past the strip: all zeros (never multiplied, see the main loop)
        const int b = live ? WG_DIV_TPI(tile) : 0;
        const int h0 = live ? (tile - b * p.tilesPerImg) * TH : 0;
        // ---- activations: prologue on load.  Rows outside the image must stay zero (relu(shift) is not) --------------------
        if (PRO == SED_PRO_NONE) {
#pragma unroll
            for (int u = 0; u < XIPT; ++u) {
                if (u == XIPT - 1 && XPIX % PPR != 0 && !xlast_ok) break;
                lds_store_raw<T>(st + xlds0 + u * XLSTEP, r.x[u]);       // hardware zeros for rows outside the image
            }
        } else {
            const int row_lo = !live ? ROWS : (h0 == 0 ? 1 : 0);
            const int row_hi = (H - h0 < ROWS - 1) ? (H - h0) : (ROWS - 1);
            const f32x4* pc = reinterpret_cast<const f32x4*>(pcoef);
            const int c4 = (xci * 32 + xcq * 8) >> 2;
            const f32x4 s0 = pc[c4], s1 = pc[c4 + 1], h0v = pc[CI_T * 8 + c4], h1v = pc[CI_T * 8 + c4 + 1];
#pragma unroll
            for (int u = 0; u < XIPT; ++u) {
                if (u == XIPT - 1 && XPIX % PPR != 0 && !xlast_ok) break;
                const int rowi = xrow0 + u * (PPR / W);
                const bool keep = rowi >= row_lo && rowi <= row_hi;
                *reinterpret_cast<bf16x8*>(st + xlds0 + u * XLSTEP) = bnrelu8_bf16(r.x[u].v, s0, s1, h0v, h1v, keep);
            }
        }
        // ---- dz: as stored, or produced here (BatchNorm / ReLU / pool backward) and written out ------------------------------
        T* __restrict__ dzs = st + CI_T * XS1;
        if (DZ == DZ_GIVEN) {
#pragma unroll
            for (int u = 0; u < DIPT; ++u) lds_store_raw<T>(dzs + dlds0 + u * DQS * 32, r.a[u]);     // rows past the image were read as zeros
        } else {
            const int qmax = live ? (H - h0) * W : 0;      // pixels of the tile inside the image (>= BM except on the last tile)
            const bool writer = live && dzo != nullptr && ((tile + cig) % NCI) == 0;
            const __amdgpu_buffer_rsrc_t os = make_srd(writer ? dzo + (size_t)b * zimg_ : nullptr, writer ? zimg_ * 2 : 0);
            const unsigned dt = (unsigned)(h0 * W * Coutp * 2);
            const f32x4* cf = reinterpret_cast<const f32x4*>(coef);
            const int d4 = dc8 >> 2;
            float kca[8], kcb[8], kcc[8], ksc[8], ksh[8];
#pragma unroll
            for (int hlf = 0; hlf < 2; ++hlf) {
                const f32x4 a4 = cf[2 * CO_T * 8 + d4 + hlf], b4 = cf[3 * CO_T * 8 + d4 + hlf], c4v = cf[4 * CO_T * 8 + d4 + hlf];
#pragma unroll
                for (int e = 0; e < 4; ++e) { kca[4 * hlf + e] = a4[e]; kcb[4 * hlf + e] = b4[e]; kcc[4 * hlf + e] = c4v[e]; }
                if (DZ == DZ_POOL) {
                    const f32x4 s4 = cf[d4 + hlf], t4 = cf[CO_T * 8 + d4 + hlf];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ksc[4 * hlf + e] = s4[e]; ksh[4 * hlf + e] = t4[e]; }
                }
            }
#pragma unroll
            for (int u = 0; u < DIPT; ++u) {
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
                if (qmax < BM) {                              // only the last tile of an image (and a tile past the strip) has rows past it
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
                if (writer) buf_store8<T>(os, dvoff0 + (dt + (unsigned)u * dustep), v);     // rows past the image: dropped by the range check
            }
        }
    };

    // =============================== matrix half of every wave ===========================================================
    const int hh = lane >> 5, r32 = lane & 31;
    const int wci = wave / CO_T, wco = wave % CO_T;      // the wave's (cin tile, cout tile) pair, taps 0..8

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // lane-constant parts of the transpose-read addresses: the lane supplies k-row kl = 8*hh + q (+4 for the second half) and the 4
    // channels 16*gbit + 4*pp .. +3.  k-row kl of a k-step is column kl of its 16-column strip (W >= 16) or pixel (row 8*(kl >> 3), column kl & 7)
    int offA[2], offB[2];
    {
        const int i16 = lane & 15, gbit = (lane >> 4) & 1;
        const int qq = i16 >> 2, pp = i16 & 3, ch = 16 * gbit + 4 * pp;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int kl = 8 * hh + qq + 4 * half;
            const int krow = W >= 16 ? 0 : 8 * (kl >> 3), kcol = W >= 16 ? kl : (kl & 7);
            offA[half] = wci * XS1 + (krow * WP + kcol) * 32 + ch;
            offB[half] = CI_T * XS1 + wco * DZ1 + (krow * W + kcol) * 32 + ch;
        }
    }

    // Steps (cs, rho): the x fragments of halo row rho (W = 8: rows rho, rho + 8) at the three column shifts, against the dz fragments
    // of k-steps rho, rho - 1, rho - 2 (tap rows 0, 1, 2).  Software-pipelined by hand, everything unrolled (every index a constant):
    // the fragment of column shift tj for the NEXT step is requested as soon as the three MFMAs that use the current one are issued,
    // the dz fragment of the next step at the start of the step.
    auto compute = [&](const T* __restrict__ st) __attribute__((always_inline)) {
        constexpr int NRH = NR + 2;
        constexpr int NS = NCS * NRH;
        bf16x8 af[3], bfr[4];
        auto rd_a = [&](int s, int tj) __attribute__((always_inline)) {
            const int cs = s / NRH, rho = s - cs * NRH;
            const int ub = (rho * WP + 16 * cs + tj) * 32;
            return join_tr(ds_read_tr16_b64(st + ub + offA[0]), ds_read_tr16_b64(st + ub + offA[1]));
        };
        auto rd_b = [&](int s) __attribute__((always_inline)) {          // s = flattened step; its own k-step is (cs, rho) with rho < NR
            const int cs = s / NRH, rho = s - cs * NRH;
            const int ub = (rho * W + 16 * cs) * 32;
            return join_tr(ds_read_tr16_b64(st + ub + offB[0]), ds_read_tr16_b64(st + ub + offB[1]));
        };
        bfr[0] = rd_b(0);
#pragma unroll
        for (int tj = 0; tj < 3; ++tj) af[tj] = rd_a(0, tj);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int cs = s / NRH, rho = s - cs * NRH;
            if (s + 1 < NS && (s + 1) % NRH < NR) bfr[(s + 1) & 3] = rd_b(s + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tj = 0; tj < 3; ++tj) {
#pragma unroll
                for (int ti = 2; ti >= 0; --ti) {
                    const int ks = rho - ti;                     // the dz k-step of tap row ti
                    if (ks >= 0 && ks < NR) acc[ti * 3 + tj] = mfma(af[tj], bfr[(cs * NRH + ks) & 3], acc[ti * 3 + tj]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (s + 1 < NS) af[tj] = rd_a(s + 1, tj);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // =============================== main loop ============================================================================
    // The stage parity is a run-time value on purpose: with the two stages as compile-time constants the second one lies beyond the
    // 64 KB reach of a DS instruction's immediate and hipcc materialises one address register per fragment (70-120 spilled registers).
    RawSet rs;
    issue(rs, t_begin);
    commit(rs, t_begin, stage0);
    issue(rs, t_begin + 1);
    wide_barrier();
    auto stamp = [&]() -> unsigned long long { return kWideStamps ? __builtin_amdgcn_s_memtime() : 0ull; };
    unsigned long long tph[4] = {0, 0, 0, 0};       // matrix phase, vector phase (commit), load issue, barrier wait
    if (wave < 4) {
#pragma unroll 1
        for (int i = 0; i < ntl; ++i) {
            const int par = i & 1;
            const unsigned long long s0 = stamp();
            compute(stage0 + par * STAGE);
            const unsigned long long s1 = stamp();
            if (SED_WIDE_PRIO) __builtin_amdgcn_s_setprio(1);
            commit(rs, t_begin + i + 1, stage0 + (par ^ 1) * STAGE);
            const unsigned long long s2 = stamp();
            issue(rs, t_begin + i + 2);
            if (SED_WIDE_PRIO) __builtin_amdgcn_s_setprio(0);
            const unsigned long long s3 = stamp();
            wide_barrier();
            if (kWideStamps) { tph[0] += s1 - s0; tph[1] += s2 - s1; tph[2] += s3 - s2; tph[3] += stamp() - s3; }
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < ntl; ++i) {
            const int par = i & 1;
            const unsigned long long s0 = stamp();
            if (SED_WIDE_PRIO) __builtin_amdgcn_s_setprio(1);
            commit(rs, t_begin + i + 1, stage0 + (par ^ 1) * STAGE);
            const unsigned long long s1 = stamp();
            issue(rs, t_begin + i + 2);
            if (SED_WIDE_PRIO) __builtin_amdgcn_s_setprio(0);
            const unsigned long long s2 = stamp();
            compute(stage0 + par * STAGE);
            const unsigned long long s3 = stamp();
            wide_barrier();
            if (kWideStamps) { tph[1] += s1 - s0; tph[2] += s2 - s1; tph[0] += s3 - s2; tph[3] += stamp() - s3; }
        }
    }
    if (kWideStamps && blockIdx.x == 8 && lane == 0 && (wave == 1 || wave == 5) && ntl > 0)
        printf("wide W=%d CI_T=%d DZ=%d PRO=%d wave %d: %d tiles; per tile: matrix %llu  vector %llu  issue %llu  barrier %llu ticks\n", W, CI_T, DZ, PRO,
               wave, ntl, tph[0] / ntl, tph[1] / ntl, tph[2] / ntl, tph[3] / ntl);

    // each wave stores its own slabs: D row = cin, col (lane) = cout
    {
        float* out = p.ws + (size_t)strip * 9 * Cinp * Coutp;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int cin = ci0 + wci * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                out[((size_t)t * Cinp + cin) * Coutp + co0 + wco * 32 + r32] = acc[t][i];
            }
        }
    }
}

// (cin tiles, cout tiles) of a workgroup: (4, 2) for layers with a multiple of 128 input channels.  The (2, 4) form the template also
// admits (64 -> 128: x streamed once, dz produced once -- as conv_wgrad3_kernel's two workgroups per strip already do) was built,
// parity-green and 8 % SLOWER on both 64 -> 128 layers (W = 16: 0.163 -> 0.176 ms, W = 32: 0.281 -> 0.305; those launches move
// 0.69 GB at 4.3 TB/s and are bound by HBM, not by their loader waves): not instantiated (profiles/r05_a_ab_wgrad_wide.txt)
bool wide_shape(int Cinp, int Coutp, int* ci_t) {
    if (Cinp % 128 == 0 && Coutp % 64 == 0) { *ci_t = 4; return true; }
    return false;
}

template <int W, int CI_T, int DZ, int PRO>
int launch_wide(Wgrad2Params& p, hipStream_t st) {
    constexpr int CO_T = 8 / CI_T;
    constexpr int TH = 128 / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr size_t stage = ((size_t)CI_T * ((TH + 2) * WP * 32 + 32) + (size_t)CO_T * (128 * 32 + 32)) * sizeof(bf16_t);
    constexpr size_t lds = 2 * stage + (size_t)(5 * CO_T * 32 + 2 * CI_T * 32) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (int rc_ = sed_set_max_lds<&conv_wgrad_wide_kernel<W, CI_T, DZ, PRO>>(lds)) return rc_;
    p.tilesPerImg = cdiv(p.H, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    sed_fastdiv_make((unsigned)p.tilesPerImg, &p.tpi_M, &p.tpi_l);
    p.tpb = cdiv(p.totalTiles, p.strips);
    const int ny = (p.Cinp / (32 * CI_T)) * (p.Coutp / (32 * CO_T));
    conv_wgrad_wide_kernel<W, CI_T, DZ, PRO><<<dim3(p.strips * ny), dim3(512), lds, st>>>(p);
    return 0;
}

template <int W, int CI_T>
int dispatch_wide_mode(int dzmode, Wgrad2Params& p, hipStream_t st) {
    const bool pro = p.pro == SED_PRO_BNRELU;
    if (dzmode == DZ_GIVEN) return pro ? launch_wide<W, CI_T, DZ_GIVEN, SED_PRO_BNRELU>(p, st) : launch_wide<W, CI_T, DZ_GIVEN, SED_PRO_NONE>(p, st);
    if (dzmode == DZ_POOL) return pro ? launch_wide<W, CI_T, DZ_POOL, SED_PRO_BNRELU>(p, st) : launch_wide<W, CI_T, DZ_POOL, SED_PRO_NONE>(p, st);
    return pro ? launch_wide<W, CI_T, DZ_BN, SED_PRO_BNRELU>(p, st) : launch_wide<W, CI_T, DZ_BN, SED_PRO_NONE>(p, st);
}

template <int W>
int dispatch_wide_shape(int dzmode, Wgrad2Params& p, int ci_t, hipStream_t st) {
    return ci_t == 4 ? dispatch_wide_mode<W, 4>(dzmode, p, st) : -1;
}

}  // namespace

// 0 = shape not covered (or switched off: SED_WGRAD_WIDE=0, the A/B knob)
int wgrad_wide_strips(int B, int H, int W, int Cinp, int Coutp) {
    int ci_t;
    if (!(W == 8 || W == 16 || W == 32) || !wide_shape(Cinp, Coutp, &ci_t)) return 0;
    if (const char* e = sed_getenv("SED_WGRAD_WIDE")) if (e[0] == '0') return 0;
    const int ny = (Cinp / (32 * ci_t)) * (Coutp / (32 * (8 / ci_t)));
    const long long tiles = (long long)B * cdiv(H, 128 / W);
    long long blocks = kWideBlocks;
    if (const char* e = sed_getenv("SED_WGRAD_BLOCKS")) blocks = atoll(e) > 0 ? atoll(e) : blocks;   // tuning knob
    long long strips = blocks / ny;
    if (strips > tiles) strips = tiles;
    if (strips < 1) strips = 1;
    return (int)strips;
}

int launch_wgrad_wide(int dzmode, Wgrad2Params& p, int W, hipStream_t st) {
    int ci_t;
    if (p.pro != SED_PRO_NONE && p.pro != SED_PRO_BNRELU) return -1;
    if (!wide_shape(p.Cinp, p.Coutp, &ci_t)) return -1;
    p.strips = wgrad_wide_strips(p.B, p.H, W, p.Cinp, p.Coutp);
    if (p.strips == 0) return -1;
    switch (W) {
        case 8: return dispatch_wide_shape<8>(dzmode, p, ci_t, st);
        case 16: return dispatch_wide_shape<16>(dzmode, p, ci_t, st);
        case 32: return dispatch_wide_shape<32>(dzmode, p, ci_t, st);
    }
    return -1;
}
