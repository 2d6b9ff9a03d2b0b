// Weight gradient AND data gradient of a 3x3 convolution with 128 output channels from ONE dz image in LDS, the workgroups of a
// pixel strip SLICED BY INPUT CHANNELS (bf16, gfx950; W = 16 / 8, 64 -> 128 and 128 -> 128: blocks 2 and 3 of the main network).
//
// autograd through ConvBlock, /root/reference/models/spectogram_models.py:132-160 (backward of :155-158 under train.py:102), for the
// widths of /root/reference/main.py:35.  csrc/sed_bwd_fused.hip does this for block 1 (W = 32, <= 4 (cin tile, cout tile) pairs of
// dW: one workgroup holds them all).  Here a layer has 8 or 16 pairs -- 9*Cin*Cout fp32 accumulators are more than one CU's
// registers -- so a strip is served by Cin/32 workgroups, co-scheduled on one XCD (xcd_remap: their repeated reads of g / z hit
// the L2).  Workgroup `cis` of a strip owns input-channel tile cis:
//   dW[:, cis tile, :]  four (cin tile, cout tile) pairs, one per consumer wave (nine 32x32 accumulators each), and
//   dx[cis tile]        two 32-pixel units x two K halves (cout 0-63 / 64-127) over the consumer waves; the two fp32 partials meet
//                       in the loader waves' epilogue (staging image in fp32);
// and produces the whole dz row ring (all 128 output channels) itself from (g, z): dz never goes to HBM.  The data-gradient
// operator (73.7 KB per cin tile: it does not fit beside the ring) streams from the L2-resident packed image into registers, four
// k-steps ahead (its 36 fragments per K half are the same every stage: the ring is periodic).
//
// Stage = 64 pixels (TH = 64 / W rows); ring, stage bookkeeping, producer / consumer roles, epilogues as in sed_bwd_fused.hip.
//
// STATUS (round 4): parity-green (the kernel-level GPU tests under tests/, 77 cases incl. pool 1 and strips that cross images) and NOT
// the default: interleaved A/B at B = 32 (tools/ab_fused_cs.py) 0.358 / 0.644 / 0.175 / 0.164 ms against 0.266 / 0.501 / 0.133 /
// 0.140 ms of the two-kernel form (b2c1, b2c2, b3c1, b3c2), PMC traffic 1.96 against ~3.4 GB.  Ablations (tools/ab_cs_abl.sh): with
// the operator stream compiled out 0.237 / 0.573 / 0.120 / 0.150 ms -- a 1 KB operator fragment from L2 feeds ONE MFMA of its wave
// (the workgroup's dx is only 64 pixels x 32 channels wide), four k-steps of prefetch do not cover the L2 latency and the nine
// accumulators leave no registers for more; the 4x redundant dz arithmetic costs 4 %.  An LDS-resident operator does not fit
// beside the four-image ring (92 + 74 + 45 KB).  Built only with make EXPERIMENTS=1, selected with SED_BWD_FUSED_CS=1.
#include "conv_common.h"

#include <stdlib.h>

namespace {

constexpr int kCsBlocks = 256;          // one workgroup per CU

__device__ __forceinline__ void cs_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int cs_xswz(int col) { return (col >> 2) & 3; }

template <int W, int DZ, int PRO, int EPI>
__global__ __launch_bounds__(512) void conv_bwd_fused_cs_kernel(BwdFusedParams p) {
    typedef bf16_t T;
    constexpr int CI = 32, CO = 128, CO_T = 4;
    constexpr int BM = 64, TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3, ROWE = WP * 32;
    constexpr int DZIMG = (4 * TH + 2) * ROWE;          // four ring positions + the last window's tail, per 32-channel image
    constexpr int A1 = BM * 32;
    constexpr int OP = CI + 4, OSZ = BM * OP;           // fp32 staging image of one K half (row pitch 36 floats)
    constexpr int NP = 256, NTHR = 512;
    constexpr int KSW = BM / 16;                        // k-steps (16 pixels) of a stage's weight-gradient contraction
    constexpr int RPK = 16 / W;                         // image rows per k-step (W = 16: 1, W = 8: 2)
    constexpr int RPU = 32 / W;                         // image rows per 32-pixel data-gradient unit
    static_assert((W == 16 || W == 8) && KSW == 4, "geometry: W = 16 or 8, 64-pixel stages");
    constexpr bool RELUBWD = EPI == SED_EPI_RELUBWD, PSTATS = EPI == SED_EPI_POOLSTATS;
    // producer item geometry: dz -- thread = (row drow0 of a group of RPI rows, column dcol, channel group dc8), item u = row u*RPI + drow0
    constexpr int IPP = CO / 8, DQS = NP / IPP, DIPT = BM * IPP / NP, RPI = DQS / W;
    static_assert(DQS % W == 0 && TH == DIPT * RPI, "a thread's dz items are rows of one column");
    constexpr int IPX = CI / 8, XQS = NP / IPX;
    static_assert(BM * IPX == NP && XQS == BM, "one activation / output item per loader thread");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* dzr = reinterpret_cast<T*>(smem);               // [CO_T][4 * TH + 2][WP][32]  swizzled 16-byte slots
    T* ab = dzr + CO_T * DZIMG;                        // [2][BM][32]
    float* osf = reinterpret_cast<float*>(ab + 2 * A1);     // [2][2 K halves][BM][OP]  fp32 partial data gradients

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, CIF = p.Cinp, NSL = CIF >> 5;
    const int logical = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int strip = logical / NSL, cis = logical - strip * NSL, nstrips = gridDim.x / NSL;
    const int c0 = cis * 32;                           // this workgroup's input channels
    const int psh = p.pool >> 1;
    const int Ho = H >> psh, Wo = W >> psh;
    const int NTI = p.tilesPerImg;                     // output tiles per image = ceil((H + 1) / TH)
    const int t_begin = strip * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    const int ntl = t_end > t_begin ? t_end - t_begin : 0;
    const int pre = (ntl > 0 && (t_begin % NTI) != 0) ? 1 : 0;          // producer-only first stage (chunk j-1 of the first tile)
    const int NS = ntl + pre;
    constexpr int D = 4;                               // stages of global loads in flight per loader thread (~36 KB per stage)
    const int NI = (NS + 2 + D - 1) / D * D;

    // ---- one-time LDS setup ------------------------------------------------------------------------------------------
    {
        bf16x8 z8;
#pragma unroll
        for (int e = 0; e < 8; ++e) z8[e] = (bf16_t)0.f;
        for (int i = tid; i < CO_T * DZIMG / 8; i += NTHR) *reinterpret_cast<bf16x8*>(dzr + i * 8) = z8;     // padding columns stay zero
    }
    __syncthreads();

    struct StInfo { int b, j, pos; bool live, mainst; };
    auto st_first = [&]() -> StInfo {
        StInfo t;
        const int b0 = t_begin / NTI, j0 = t_begin - b0 * NTI;
        t.live = NS > 0;
        t.b = t.live ? b0 : 0;
        t.j = t.live ? (pre ? j0 - 1 : j0) : 0;
        t.mainst = t.live && !pre;
        t.pos = 0;
        return t;
    };
    auto st_next = [&](const StInfo& c, int s_next) -> StInfo {
        StInfo n;
        int j = c.j + 1, b = c.b;
        if (j == NTI) { j = 0; b += 1; }
        n.live = s_next < NS;
        n.b = n.live ? b : 0;
        n.j = n.live ? j : 0;
        n.mainst = n.live;
        n.pos = (c.pos + (j == 0 ? 2 : 1)) & 3;      // top of an image: skip a position (its prefix rows are free to be zeroed)
        return n;
    };
    const StInfo st_dead = {0, 0, 0, false, false};

    if (wave >= 4) {
        float S[8], Q[8];                                   // (statistics sums live in the loader waves only: the consumers have no register to spare)
#pragma unroll
        for (int e = 0; e < 8; ++e) { S[e] = 0.f; Q[e] = 0.f; }
        // =============================== PRODUCERS =====================================================
        const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
        const T* __restrict__ gg = reinterpret_cast<const T*>(p.gsrc);
        const T* __restrict__ zsg = reinterpret_cast<const T*>(p.zsrc);
        T* __restrict__ dxg = reinterpret_cast<T*>(p.dx);
        const int pt = tid - 256;
        const size_t ximg_ = (size_t)H * W * CIF, zimg_ = (size_t)H * W * CO, pimg_ = (size_t)Ho * Wo * CO;

        const int dq0 = pt / IPP, dc8 = (pt - dq0 * IPP) * 8;
        const int drow0 = dq0 / W, dcol = dq0 - drow0 * W;
        unsigned dvoff[DIPT], pvoff[DIPT];
        int drow[DIPT];
#pragma unroll
        for (int u = 0; u < DIPT; ++u) {
            drow[u] = u * RPI + drow0;
            dvoff[u] = (unsigned)(((drow[u] * W + dcol) * CO + dc8) * 2);
            pvoff[u] = (unsigned)((((drow[u] >> psh) * Wo + (dcol >> psh)) * CO + dc8) * 2);
        }
        const int dlds0 = (dc8 >> 5) * DZIMG + (dcol + 1) * 32 + ((((dc8 & 31) >> 3) ^ cs_xswz(dcol + 1)) * 8);
        // activation / output item: pixel xq0 of the tile, channel group xc8 of this workgroup's slice
        const int xq0 = pt / IPX, xc8 = (pt - xq0 * IPX) * 8;
        const unsigned xvoff0 = (unsigned)((xq0 * CIF + c0 + xc8) * 2);
        const int xlds0 = xq0 * 32 + xc8;
        float kca[8], kcb[8], kcc[8], ksc[8], ksh[8], qsc[8], qsh[8];
        {
            const float inv_pool = (DZ == DZ_POOL && psh) ? 0.25f : 1.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                kca[e] = p.ca[dc8 + e] * inv_pool; kcb[e] = p.cb[dc8 + e]; kcc[e] = p.cc[dc8 + e];
                ksc[e] = DZ == DZ_POOL ? p.scale[dc8 + e] : 0.f; ksh[e] = DZ == DZ_POOL ? p.shift[dc8 + e] : 0.f;
                qsc[e] = PRO == SED_PRO_BNRELU ? p.pro_scale[c0 + xc8 + e] : 0.f; qsh[e] = PRO == SED_PRO_BNRELU ? p.pro_shift[c0 + xc8 + e] : 0.f;
            }
        }
        float cem[8];
        if (RELUBWD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) cem[e] = p.epi_mean[c0 + xc8 + e];
        }

        struct RawSet { Raw8<T> x; Raw8<T> a[DIPT]; Raw8<T> b[DIPT]; };
        Raw8<T> zkeep[2];                              // RELUBWD: the ReLU / BN1 reference of flush(s + 2) is the z tile the prologue consumed
        u32x2 craw;

        // every load is issued unconditionally: a dead stage gets zero-sized descriptors (zeros, no traffic): exact vmcnt bookkeeping
        auto issue = [&](RawSet& r, const StInfo& si) {
            const bool live = si.live, mainst = si.mainst;
            const int b = si.b, j = si.j;
            const size_t ximg = (live && mainst) ? ximg_ : 0, zimg = live ? zimg_ : 0, pimg = live ? pimg_ : 0;
            const __amdgpu_buffer_rsrc_t xsrd = make_srd(xg + (size_t)b * ximg, ximg * 2);
            const unsigned xt = (unsigned)((TH * j - 1) * W * CIF * 2);       // wraps for the row above the image: out of range -> 0
            r.x = buf_load8<T>(xsrd, xvoff0 + xt);
            const unsigned dt = (unsigned)(TH * j * W * CO * 2);
            const __amdgpu_buffer_rsrc_t zs = make_srd(zsg + (size_t)b * zimg, zimg * 2);
            if (DZ == DZ_POOL) {
                const __amdgpu_buffer_rsrc_t gs = make_srd(gg + (size_t)b * pimg, pimg * 2);
                const unsigned ptq = (unsigned)(((TH * j) >> psh) * Wo * CO * 2);
#pragma unroll
                for (int u = 0; u < DIPT; ++u) {
                    r.a[u] = buf_load8<T>(gs, pvoff[u] + ptq);
                    r.b[u] = buf_load8<T>(zs, dvoff[u] + dt);
                }
            } else {
                const __amdgpu_buffer_rsrc_t gs = make_srd(gg + (size_t)b * zimg, zimg * 2);
#pragma unroll
                for (int u = 0; u < DIPT; ++u) {
                    r.a[u] = buf_load8<T>(gs, dvoff[u] + dt);
                    r.b[u] = buf_load8<T>(zs, dvoff[u] + dt);
                }
            }
        };

        auto commit = [&](const RawSet& r, const StInfo& si, int s) {
            const bool live = si.live, mainst = si.mainst;
            const int j = si.j;
            if (!live) return;
            // ---- dz chunk j (image rows TH*j .. TH*j + TH - 1) -> window rows 2 .. TH + 1 of ring position si.pos ------------
            T* __restrict__ dst = dzr + dlds0 + (si.pos * TH + 2) * ROWE;
            const bool dup = si.pos == 3;                   // rows TH-2, TH-1 of this chunk are also the prefix of position 0
            if (j == 0) {                                   // top of an image: the two prefix rows are the convolution's zero padding
                bf16x8 z8;
#pragma unroll
                for (int e = 0; e < 8; ++e) z8[e] = (bf16_t)0.f;
                if (drow0 == 0) {
                    *reinterpret_cast<bf16x8*>(dst - 2 * ROWE) = z8;
                    *reinterpret_cast<bf16x8*>(dst - ROWE) = z8;
                }
            }
            const int rows_in = H - TH * j;                 // rows of the chunk inside the image (pool floor: g = 0 by the range check)
#pragma unroll
            for (int u = 0; u < DIPT; ++u) {
                float g[8], z[8], v[8];
                raw_to_f(r.a[u], g);
                raw_to_f(r.b[u], z);
#if defined(SED_CS_ABL) && (SED_CS_ABL & 2)
                if (cis != 0) { store8<T>(dst + drow[u] * ROWE, z); continue; }      // ablation: only slice 0 pays for the dz arithmetic
#endif
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float base = fmaf(kcb[i], z[i], kcc[i]);
                    const float full = fmaf(kca[i], g[i], base);
                    if (DZ == DZ_POOL) v[i] = (fmaf(z[i], ksc[i], ksh[i]) > 0.f) ? full : base;
                    else v[i] = full;
                }
                if (rows_in < TH) {                          // (uniform: only the last chunks of an image)
                    const float m = (drow[u] < rows_in) ? 1.f : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= m;
                }
                store8<T>(dst + drow[u] * ROWE, v);
                if (dup && drow[u] >= TH - 2) store8<T>(dzr + dlds0 + (drow[u] - (TH - 2)) * ROWE, v);
            }
            // ---- activation tile j (image rows TH*j - 1 .. TH*j + TH - 2), this workgroup's 32 channels -> buffer s & 1 ---------
            if (!mainst) return;
            T* __restrict__ adst = ab + (s & 1) * A1 + xlds0;
            if (PRO == SED_PRO_NONE) {
                lds_store_raw<T>(adst, r.x);                 // hardware zeros outside the image
            } else {
                const int row = TH * j - 1 + xq0 / W;
                const bool keep = row >= 0 && row < H;       // rows outside the image stay zero (relu(shift) is not)
                const f32x4 qs0 = {qsc[0], qsc[1], qsc[2], qsc[3]}, qs1 = {qsc[4], qsc[5], qsc[6], qsc[7]};
                const f32x4 qh0 = {qsh[0], qsh[1], qsh[2], qsh[3]}, qh1 = {qsh[4], qsh[5], qsh[6], qsh[7]};
                *reinterpret_cast<bf16x8*>(adst) = bnrelu8_bf16(r.x.v, qs0, qs1, qh0, qh1, keep);
            }
        };

        // pooled-tensor statistics: active-pixel counts of the output tile of the NEXT iteration's flush (stage s - 1)
        auto issue_c = [&](const StInfo& si) {
            if constexpr (PSTATS) {
                const size_t rimg = (si.live && si.mainst) ? ximg_ : 0;
                const unsigned tq = (unsigned)((TH * si.j - 1) * W * CIF * 2);
                const __amdgpu_buffer_rsrc_t cs = make_srd(p.cnt + (size_t)si.b * rimg, rimg);
                craw = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(cs, (xvoff0 + tq) >> 1, 0, 0));
            }
        };
        // the output tile of stage s - 2 sits complete in its staging image
        auto flush = [&](const StInfo& si, int s) {     // si = stage s - 2
            const bool live = si.live, mainst = si.mainst;
            const int b = si.b, j = si.j;
            if (!live || !mainst) return;
            const float* osb = osf + (s & 1) * 2 * OSZ + xq0 * OP + xc8;
            const T* aref = ab + (s & 1) * A1 + xlds0;          // activation tile of stage s - 2 (PSTATS: the pooled activation itself)
            const __amdgpu_buffer_rsrc_t ds = make_srd(dxg + (size_t)b * ximg_, ximg_ * 2);
            const unsigned tq = (unsigned)((TH * j - 1) * W * CIF * 2);
            bf16x8 raw;                                     // the two K halves summed, rounded to bf16 as stored
            {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(osb), a1 = *reinterpret_cast<const f32x4*>(osb + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(osb + OSZ), b1 = *reinterpret_cast<const f32x4*>(osb + OSZ + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { raw[e] = (bf16_t)(a0[e] + b0[e]); raw[4 + e] = (bf16_t)(a1[e] + b1[e]); }
            }
            const int row = TH * j - 1 + xq0 / W;
            const bool valid = row >= 0 && row < H;
            const unsigned off = valid ? xvoff0 + tq : SED_OOB;
            if (RELUBWD) {
                float v[8], z[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (float)raw[e];
                raw_to_f(zkeep[s & 1], z);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gate = (valid && fmaf(z[e], qsc[e], qsh[e]) > 0.f) ? v[e] : 0.f;
                    v[e] = gate;
                    S[e] += gate;
                    Q[e] = fmaf(gate, z[e] - cem[e], Q[e]);
                }
                buf_store8<T>(ds, off, v);
            } else {
                if constexpr (PSTATS) {       // S = sum dy*cnt, Q = sum dy*y_pooled (rows outside the image: 0)
                    float ya[8];
                    Raw8<T> yr; yr.v = *reinterpret_cast<const bf16x8*>(aref);
                    raw_to_f(yr, ya);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float dyv = (float)raw[e];
                        const float cf = (float)((craw[e >> 2] >> (8 * (e & 3))) & 0xffu);
                        S[e] = fmaf(dyv, cf, S[e]);
                        Q[e] = fmaf(dyv, ya[e], Q[e]);
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, raw), ds, off, 0, 0);
            }
        };

        RawSet r0, r1, r2, r3;
        StInfo sm2 = st_dead, sm1 = st_dead, sc = st_first();
        StInfo sf = sc;                                     // the stage whose loads are issued next (s + D)
        issue(r0, sf); sf = st_next(sf, 1);
        issue(r1, sf); sf = st_next(sf, 2);
        issue(r2, sf); sf = st_next(sf, 3);
        issue(r3, sf); sf = st_next(sf, 4);
        auto iter = [&](int s, RawSet& r) {
            if constexpr (PSTATS) flush(sm2, s);            // (its reference is the activation tile that commit(s) overwrites)
            commit(r, sc, s);
            if constexpr (!PSTATS) flush(sm2, s);
            if constexpr (RELUBWD) zkeep[s & 1] = r.x;      // the reference of flush(s + 2)
            issue_c(sm1);
            issue(r, sf);
            sm2 = sm1; sm1 = sc; sc = st_next(sc, s + 1); sf = st_next(sf, s + D + 1);
            cs_barrier();
        };
        for (int s = 0; s < NI; s += D) {
            iter(s, r0);
            iter(s + 1, r1);
            iter(s + 2, r2);
            iter(s + 3, r3);
        }
        cs_barrier();                                       // (the consumers are past their last LDS read)
        if (RELUBWD || PSTATS) {
            float* red = reinterpret_cast<float*>(smem);   // [NP][16]
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[pt * 16 + e] = S[e]; red[pt * 16 + 8 + e] = Q[e]; }
        }
    } else {
        // =============================== CONSUMERS =====================================================
        const int r = lane & 31, hh = lane >> 5;
        f32x16 accw[9];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) accw[t][i] = 0.f;
        // ---- data-gradient unit: 32 pixels pu of the tile, K half kh (output channels 64*kh .. 64*kh + 63) -----------------------
        const int pu = wave & 1, kh = wave >> 1;
        const int prow = r / W, pcol = r - prow * W;
        int xoff[3][2];                                     // [tj][ks]: lane part of the dz fragment address (halo column pcol + tj)
#pragma unroll
        for (int tj = 0; tj < 3; ++tj)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xoff[tj][ks] = (prow * WP + pcol + tj) * 32 + (((ks * 2 + hh) ^ cs_xswz(pcol + tj)) * 8);
        const int ostg = kh * OSZ + (pu * 32 + r) * OP + 4 * hh;
        // operator fragment (chunk c of this K half, tap, ks) of this lane: 16 bytes of wpack_t [cout chunk][tap][kq][CIF][8] at
        // kq = 2*ks + hh, input channel c0 + r
        const __amdgpu_buffer_rsrc_t wsrd = make_srd(reinterpret_cast<const T*>(p.wpack_t) + (size_t)(2 * kh) * 36 * CIF * 8, (size_t)72 * CIF * 8 * 2);
        const unsigned wstep = (unsigned)(CIF * 8 * 2);     // bytes between consecutive kq of the packed image
        const unsigned wlane = (unsigned)(((hh * CIF + c0 + r) * 8) * 2);
        // ---- weight-gradient pair: (this workgroup's cin tile, cout tile `wave`) -------------------------------------------------
        int offA[2], offB[3][2];
        {
            const int i16 = lane & 15, gbit = (lane >> 4) & 1;
            const int qq = i16 >> 2, pp = i16 & 3, ch = 16 * gbit + 4 * pp;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int kl = 8 * hh + qq + 4 * half;      // pixel of the 16-pixel k-step
                const int krow = kl / W, kcol = kl - krow * W;
                offA[half] = kl * 32 + ch;
#pragma unroll
                for (int sj = 0; sj < 3; ++sj) offB[sj][half] = wave * DZIMG + (krow * WP + kcol + sj) * 32 + (ch ^ swz<T>(kcol + sj));
            }
        }

        // the operator ring: fragment k (of 36 per stage) sits in slot k % RD; fragments k + RD.. are requested while k is used
        constexpr int RD = 4, NK = 36;
        static_assert(NK % RD == 0, "the ring is periodic over a stage");
        bf16x8 wf[RD];
        auto ld_w = [&](int k, bf16x8& dst) {               // k = (chunk c of this half, tap, ks)
            const int c = k / 18, kk = k % 18, tap = kk >> 1, ks = kk & 1;
            dst = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wsrd, wlane + (unsigned)(c * 36 + tap * 4 + ks * 2) * wstep, 0, 0));
        };
#pragma unroll
        for (int k = 0; k < RD; ++k) ld_w(k, wf[k]);

        StInfo csi = st_first();
        auto citer = [&](int s) {
            cs_barrier();
            const StInfo cs = csi;
            csi = st_next(csi, s + 1);
            if (!cs.live || !cs.mainst) return;
            // the stage's window: TH + 2 consecutive ring rows from row pos*TH (row hr of it = image row TH*j - 2 + hr)
            const T* __restrict__ win = dzr + cs.pos * TH * ROWE;
            const T* __restrict__ abuf = ab + (s & 1) * A1;

            // ---- weight gradient: accw[si*3+sj] += a[k-step] (x) dz[k-step shifted by (si, sj)] -----------------------------
            {
                constexpr int NSTEP = KSW * 3;           // step = (k-step, shift row): 3 MFMAs
                bf16x8 bfr[2][3], afr[2];
                auto ld_a = [&](int kk, bf16x8& dst) {
                    dst = join_tr(ds_read_tr16_b64(abuf + kk * 16 * 32 + offA[0]), ds_read_tr16_b64(abuf + kk * 16 * 32 + offA[1]));
                };
                auto ld_b = [&](int st, bf16x8 (&dst)[3]) {
                    const int kk = st / 3, si = st % 3;
                    const int imm = (kk * RPK + si) * ROWE;
#pragma unroll
                    for (int sj = 0; sj < 3; ++sj)
                        dst[sj] = join_tr(ds_read_tr16_b64(win + imm + offB[sj][0]), ds_read_tr16_b64(win + imm + offB[sj][1]));
                };
                // (two-deep fragment ring here: the operator ring of the data-gradient loop stays live across this phase and the nine
                //  accumulators leave 112 registers for everything else)
                ld_a(0, afr[0]);
                ld_b(0, bfr[0]);
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) {
                    if (st + 1 < NSTEP) ld_b(st + 1, bfr[(st + 1) & 1]);
                    if (st % 3 == 0 && st / 3 + 1 < KSW) ld_a(st / 3 + 1, afr[(st / 3 + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int sj = 0; sj < 3; ++sj)
                        accw[(st % 3) * 3 + sj] = mfma(afr[(st / 3) & 1], bfr[st & 1][sj], accw[(st % 3) * 3 + sj]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- data gradient: D[cin][pixel] over this wave's K half (two 32-channel chunks x 9 taps x 2 k-steps) -------------------
            f32x16 accd;
#pragma unroll
            for (int i = 0; i < 16; ++i) accd[i] = 0.f;
            {
                bf16x8 xf[3];
                const T* __restrict__ dbase = win + (pu * RPU) * ROWE + (2 * kh) * DZIMG;
                auto ld_x = [&](int k, bf16x8& xd) {
                    const int c = k / 18, kk = k % 18, tap = kk >> 1, ks = kk & 1, ti = tap / 3, tj = tap % 3;
                    xd = *reinterpret_cast<const bf16x8*>(dbase + (c * DZIMG + ti * ROWE) + xoff[tj][ks]);
                };
                ld_x(0, xf[0]);
                ld_x(1, xf[1]);
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    if (k + 2 < NK) ld_x(k + 2, xf[(k + 2) % 3]);
                    __builtin_amdgcn_sched_barrier(0);
                    accd = mfma(wf[k % RD], xf[k % 3], accd);
                    __builtin_amdgcn_sched_barrier(0);
#if !defined(SED_CS_ABL) || !(SED_CS_ABL & 1)
                    ld_w((k + RD) % NK, wf[k % RD]);        // (the last RD requests are the next stage's first fragments)
#endif
                }
            }
            float* osb = osf + (s & 1) * 2 * OSZ + ostg;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {accd[4 * g], accd[4 * g + 1], accd[4 * g + 2], accd[4 * g + 3]};
                *reinterpret_cast<f32x4*>(osb + 8 * g) = v;
            }
        };
        for (int s = 0; s < NI; s += 2) {
            citer(s);
            citer(s + 1);
        }
        cs_barrier();                                       // (the producers join: nothing reads the stage buffers any more)
        // ---- weight-gradient slab rows of this workgroup: [9][cin tile cis][128] of the strip's slab ----------------------------
        {
            float* out = p.ws + (size_t)strip * 9 * CIF * CO;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int tap = (2 - t / 3) * 3 + (2 - t % 3);         // shift (si, sj) = (2 - ti, 2 - tj)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int cin = c0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    out[((size_t)tap * CIF + cin) * CO + wave * 32 + r] = accw[t][i];
                }
            }
        }
    }

    // ---- statistics partial of this workgroup (fixed-order sum over the producer threads of a channel group) --------------------
    if (RELUBWD || PSTATS) {
        float* red = reinterpret_cast<float*>(smem);       // [NP][16], written by the loader waves above
        __syncthreads();
        if (tid < 2 * CI) {
            const int stat = tid / CI, cn = tid % CI;
            const int cg = cn >> 3, e = cn & 7;
            const int ch = c0 + cn;
            float tot = 0.f;
            for (int k = 0; k < XQS; ++k) tot += red[(cg + IPX * k) * 16 + stat * 8 + e];
            if (RELUBWD && stat) tot *= p.epi_invstd[ch];
            if constexpr (PSTATS) {       // (as sed_conv_pc.hip: sum g = sum dy*cnt / 4, sum g*xhat = (sum dy*y - beta/4 sum dy*cnt) / gamma)
                float sraw = tot;
                if (stat) {
                    sraw = 0.f;
                    for (int k = 0; k < XQS; ++k) sraw += red[(cg + IPX * k) * 16 + e];
                }
                const float sc = p.epi_scale[ch], is = p.epi_invstd[ch];
                const float beta = fmaf(p.epi_mean[ch], sc, p.epi_shift[ch]);
                const bool ill = fabsf(beta) * is > 8.0f * fabsf(sc);
                if (!stat) {
                    tot = 0.25f * sraw;
                } else if (sc != 0.f && !ill) {
                    tot = (tot - 0.25f * beta * sraw) * (is / sc);
                } else {
                    if (tot != 0.f || sraw != 0.f) atomicOr(p.flag, 1);
                    tot = 0.f;
                }
            }
            p.partial[((size_t)strip * 2 + stat) * CIF + ch] = tot;
            for (int row = strip + nstrips; row < p.nparts; row += nstrips) p.partial[((size_t)row * 2 + stat) * CIF + ch] = 0.f;
        }
    }
}

template <int W, int DZ, int PRO, int EPI>
int launch_cs(BwdFusedParams& p, hipStream_t st) {
    constexpr int TH = 64 / W, WP = (W + 2 + 3) & ~3;
    constexpr size_t lds = ((size_t)4 * (4 * TH + 2) * WP * 32 + (size_t)2 * 64 * 32) * sizeof(bf16_t) + (size_t)2 * 2 * 64 * 36 * sizeof(float);
    static_assert(lds <= 160 * 1024 && lds >= 256 * 16 * sizeof(float), "LDS budget (the statistics reduction reuses it)");
    if (p.dry) return 0;
    if (int rc_ = sed_set_max_lds<&conv_bwd_fused_cs_kernel<W, DZ, PRO, EPI>>(lds)) return rc_;
    conv_bwd_fused_cs_kernel<W, DZ, PRO, EPI><<<dim3(p.nwg * (p.Cinp / 32)), dim3(512), lds, st>>>(p);
    return 0;
}

}  // namespace

// strips (= weight-gradient slabs) the cin-sliced fused backward launches for this problem; 0 = shape / mode not covered
int bwd_fused_cs_nstrips(int B, int H, int W, int Cinp, int Coutp, int dzmode, int pro, int epi, int pool) {
    if (!(W == 16 || W == 8) || Coutp != 128 || !(Cinp == 64 || Cinp == 128)) return 0;
    const bool c1 = dzmode == DZ_BN && pro == SED_PRO_NONE && (epi == SED_EPI_POOLSTATS || epi == SED_EPI_STORE);
    const bool c2 = dzmode == DZ_POOL && pro == SED_PRO_BNRELU && epi == SED_EPI_RELUBWD && (pool == 1 || pool == 2);
    if (!c1 && !c2) return 0;
    if (const char* e = sed_getenv("SED_BWD_FUSED")) if (e[0] == '0') return 0;
    // opt-in (make EXPERIMENTS=1 and SED_BWD_FUSED_CS=1): measured 30 % SLOWER than the two-kernel backward of these layers (round 4,
    // tools/ab_fused_cs.py: 1.34 vs 1.04 ms for the four layers at B = 32) although it halves their HBM traffic (1.96 vs 3.4 GB)
    {
        const char* e = sed_getenv("SED_BWD_FUSED_CS");
        if (!e || e[0] != '1') return 0;
    }
    const int TH = 64 / W, nsl = Cinp / 32;
    const long long tiles = (long long)B * cdiv(H + 1, TH);
    long long n = kCsBlocks;
    if (const char* e = sed_getenv("SED_BWD_FUSED_BLOCKS")) n = atoll(e) > 0 ? atoll(e) : n;      // tuning knob (total workgroups)
    n /= nsl;
    if (n > tiles) n = tiles;
    return (int)(n < 1 ? 1 : n);
}

int launch_bwd_fused_cs(BwdFusedParams& p, int W, hipStream_t st) {
    p.nwg = bwd_fused_cs_nstrips(p.B, p.H, W, p.Cinp, p.Coutp, p.dzmode, p.pro, p.epi, p.pool);
    if (p.nwg == 0) return -1;
    if (p.epi != SED_EPI_STORE && p.nwg > p.nparts) p.nwg = p.nparts;
    const int TH = 64 / W;
    p.tilesPerImg = cdiv(p.H + 1, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    p.tpb = cdiv(p.totalTiles, p.nwg);
#define SED_CS_DISPATCH(WW)                                                                                                    \
    do {                                                                                                                      \
        if (p.dzmode == DZ_BN) {                                                                                              \
            if (p.epi == SED_EPI_POOLSTATS) return launch_cs<WW, DZ_BN, SED_PRO_NONE, SED_EPI_POOLSTATS>(p, st);              \
            return launch_cs<WW, DZ_BN, SED_PRO_NONE, SED_EPI_STORE>(p, st);                                                  \
        }                                                                                                                     \
        return launch_cs<WW, DZ_POOL, SED_PRO_BNRELU, SED_EPI_RELUBWD>(p, st);                                                \
    } while (0)
    if (W == 16) SED_CS_DISPATCH(16);
    SED_CS_DISPATCH(8);
#undef SED_CS_DISPATCH
}
