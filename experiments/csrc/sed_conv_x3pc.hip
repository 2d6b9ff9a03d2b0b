// 3x3 convolution forward / data gradient in split-operand arithmetic (dtype SED_F32H3, "f16x3"), producer / consumer form (gfx950).
//
// ConvBlock's nn.Conv2d forward and autograd's data gradient (/root/reference/models/spectogram_models.py:132-140,155-156) on fp32 NHWC
// tensors, every operand split into two fp16 pieces where it is staged, three fp16 MFMAs per product (csrc/x3_common.h; the arithmetic,
// prologue and epilogues are conv_x3_kernel's, csrc/sed_conv_x3.hip).
//
// Why a second kernel.  conv_x3_kernel runs "all waves stage, barrier, all waves multiply" with two workgroups per CU: its timing ablations
// (profiles/r06_aa_x3_ablate.txt) show half of a launch in its skeleton -- loads, barriers, epilogue -- and only partial overlap of the
// two workgroups' phases.  Here one 512-thread workgroup per CU splits the roles (conv_pc_kernel's structure, csrc/sed_conv_pc.hip):
//   * waves 4-7, one per SIMD: PRODUCERS.  Halo tile of stage s+1: load (one register set, re-loaded item by item as it is staged), BN+ReLU
//     prologue in fp32, split, write the hi / lo planes; the operator chunk of stage s+1 when the operator is not resident; the epilogue
//     of the tile that finished at stage s-1 (staging image -> whole-line stores, BatchNorm statistics / the ReLU-backward gate and sums).
//   * waves 0-3: CONSUMERS: one 32-pixel x 32-channel tile each, fragment reads + MFMA only; after a tile's last chunk the accumulators
//     go to the staging image.
//   * ONE s_barrier per stage (tile, 32-channel chunk) hands the double-buffered planes over.
// 160 KB of LDS for one workgroup: the operator stays resident up to Cin = 64 (the two-workgroup form restaged it every stage from Cin = 64).
#include "x3_common.h"

#include <stdlib.h>

namespace {

constexpr int kX3pcConvBlocks = 256;      // one workgroup per CU

__device__ __forceinline__ void x3c_barrier() {
    // LDS writes / reads of this wave are complete; global loads stay in flight across the barrier
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// NPW loader waves (8: two per SIMD -- the staging code is latency-bound, four waves cannot keep the four MFMA waves fed)
#ifndef SED_X3PC_NPW
#define SED_X3PC_NPW 8
#endif
template <int W, int PRO, int EPI>
__global__ __launch_bounds__(256 + 64 * SED_X3PC_NPW) __attribute__((amdgpu_waves_per_eu(SED_X3PC_NPW == 8 ? 3 : 2, SED_X3PC_NPW == 8 ? 3 : 2))) void conv_x3pc_kernel(ConvParams p) {
    typedef X3<true> XT;
    typedef typename XT::vec vec;
    constexpr int BM = 128, BN = 32, NP = 64 * SED_X3PC_NPW, NTHR = 256 + NP;
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr int ROWS = TH + 2;
    constexpr int PS = 40;                   // padded-linear 80-byte pixels: conflict-free ds_read_b128 for all nine taps
    constexpr int XS = ROWS * WP * PS;       // elements per plane
    constexpr int WS = 9 * 32 * BN;          // elements per plane and 32-channel chunk
    constexpr int BNP = BN + 4;
    constexpr int XITEMS = ROWS * W * 4;     // 32-byte items of a halo tile (the two padding columns are zeroed once)
    constexpr int XIPT = (XITEMS + NP - 1) / NP;
    constexpr int WITEMS = 2 * WS / 8;       // 16-byte items of an operator chunk (hi image, lo image)
    constexpr int WIPT = (WITEMS + NP - 1) / NP;
    constexpr int IPR = BN / 8, FIPT = BM * IPR / NP, FQS = NP / IPR;
    constexpr bool GRADOP = EPI != SED_EPI_STATS;          // (the exponent is 0 for forward calls: the scale is then 1)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool wres = p.wres != 0;
    const int nchunks = p.Cinp >> 5;
    const int nwb = wres ? nchunks : 2;      // operator buffers
    const int nos = nchunks == 1 ? 2 : 1;    // single-chunk layers finish a tile every stage: two staging images
    u16_t* planes = reinterpret_cast<u16_t*>(smem);                  // [2 stages][hi, lo][XS]
    u16_t* wop = planes + 4 * XS;                                    // [nwb][hi, lo][WS]
    float* os = reinterpret_cast<float*>(wop + nwb * 2 * WS);        // [nos][BM][BNP]
    float* ecoef = os + nos * BM * BNP;                              // [3][BN]
    float* pcoef = ecoef + 3 * BN;                                   // [2][Cinp]: prologue scale, shift (read per stage: a global load
                                                                     // there exposed an L2 round trip in every stage)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NY = p.Coutp / BN;
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int by = logical % NY, bx = logical / NY;
    const int nbx = gridDim.x / NY;
    const int n0 = by * BN;
    const int H = p.H, Cinp = p.Cinp, Coutp = p.Coutp;
    const float* __restrict__ xg = reinterpret_cast<const float*>(p.x);
    const u16_t* __restrict__ wg = reinterpret_cast<const u16_t*>(p.wpack);
    float* __restrict__ zg = reinterpret_cast<float*>(p.z);
    const float* __restrict__ zr = reinterpret_cast<const float*>(p.zref);
    const size_t wchunk_bytes = (size_t)(9 * 4) * Coutp * 8 * 2;          // one 32-input-channel chunk of one image
    const size_t wimg_bytes = wchunk_bytes * nchunks;
    const size_t ximg = (size_t)H * W * Cinp, zimg = (size_t)H * W * Coutp;

    const int t_begin = bx * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    const int ntl = t_end > t_begin ? t_end - t_begin : 0;
    const int nst = ntl * nchunks;

    // ---- one-time LDS setup: padding columns, epilogue coefficients, the resident operator -----------------------------------------
    {
        constexpr int NPAD = 2 * 2 * ROWS * 2 * 5;       // stage, piece, row, side, 16-byte piece of the 80-byte pixel
        for (int i = tid; i < NPAD; i += NTHR) {
            const int c16 = i % 5, side = (i / 5) & 1, rowi = (i / 10) % ROWS, pl = (i / (10 * ROWS)) & 3;
            const sed_u32x4 z4 = {0u, 0u, 0u, 0u};
            *reinterpret_cast<sed_u32x4*>(planes + pl * XS + (rowi * WP + (side ? W + 1 : 0)) * PS + c16 * 8) = z4;
        }
    }
    if (PRO == SED_PRO_BNRELU) {
        for (int i = tid; i < 2 * Cinp; i += NTHR) pcoef[i] = (i < Cinp ? p.pro_scale : p.pro_shift)[i < Cinp ? i : i - Cinp];
    }
    if (EPI == SED_EPI_RELUBWD) {
        for (int i = tid; i < 3 * BN; i += NTHR) {
            const int a = i / BN, c = i - a * BN;
            ecoef[i] = (a == 0 ? p.epi_scale : a == 1 ? p.epi_shift : p.epi_mean)[n0 + c];
        }
    }
    // operator item `it` of chunk kc: piece image, row (tap, kq) of BN*8 contiguous elements; source row stride Coutp*8
    auto w_src = [&](int it) -> unsigned {
        const int rem = it % (WS / 8), rowi = rem / (BN * 8 / 8), off = (rem - rowi * (BN * 8 / 8)) * 8;
        return (unsigned)(((rowi * Coutp + n0) * 8 + off) * 2);
    };
    const __amdgpu_buffer_rsrc_t wsrd_h = make_srd(wg, wimg_bytes);
    const __amdgpu_buffer_rsrc_t wsrd_l = make_srd(reinterpret_cast<const char*>(wg) + wimg_bytes, wimg_bytes);
    if (wres && nst > 0) {
        for (int c = 0; c < nchunks; ++c)
            for (int it = tid; it < WITEMS; it += NTHR) {
                const Raw8<bf16_t> v = buf_load8<bf16_t>(it < WS / 8 ? wsrd_h : wsrd_l, w_src(it) + (unsigned)(c * wchunk_bytes));
                *reinterpret_cast<bf16x8*>(wop + c * 2 * WS + it * 8) = v.v;
            }
    }
    __syncthreads();

    auto coords = [&](int s, int& b, int& h0, int& kc) {
        const int tl = s / nchunks;
        kc = s - tl * nchunks;
        const int tile = t_begin + tl;
        b = tile / p.tilesPerImg;
        h0 = (tile - b * p.tilesPerImg) * TH;
    };

    if (wave >= 4) {
        // =============================== PRODUCERS =====================================================
        const int pt = tid - 256;      // 0 .. NP-1
        const int cq = pt & 3;
        const float pre = __builtin_ldexpf(1.f, p.xexp);
        unsigned xvoff[XIPT];
        int xlds[XIPT];
#pragma unroll
        for (int u = 0; u < XIPT; ++u) {
            const int it = pt + u * NP;
            const int pix = it >> 2;
            const int rowi = pix / W, coli = pix - rowi * W + 1;
            const bool ok = it < XITEMS;
            xvoff[u] = ok ? (unsigned)(((rowi * W + coli) * Cinp + cq * 8) * 4) : SED_OOB;
            xlds[u] = ok ? (rowi * WP + coli) * PS + cq * 8 : 0;
        }
        float psc[8], psf[8];
        // TWO register sets of loads in flight (stage parity): a stage lasts about as long as a load takes to arrive, so a one-stage lead
        // -- what the two workgroups per CU of conv_x3_kernel have between them -- left the loaders waiting for memory every stage
        struct RawSet { Raw8<float> x[XIPT]; Raw8<bf16_t> w[WIPT]; };
        RawSet rs0, rs1;
        // stage s' halo tile / operator chunk sources (a stage past the strip: empty descriptors -- zeros, no traffic; loads are issued
        // unconditionally so that the compiler's vmcnt bookkeeping stays exact)
        struct StageSrc { __amdgpu_buffer_rsrc_t xs; unsigned xt, wo; bool live; int kc; };
        auto stage_src = [&](int s) -> StageSrc {
            StageSrc r;
            r.live = s < nst;
            int b = 0, h0 = 0, kc = 0;
            if (r.live) coords(s, b, h0, kc);
            const size_t img = r.live ? ximg : 0;
            r.xs = make_srd(xg + (size_t)b * img, img * 4);
            r.xt = (unsigned)((((h0 - 1) * W - 1) * Cinp + kc * 32) * 4);
            r.wo = (unsigned)(kc * wchunk_bytes);
            r.kc = kc;
            return r;
        };
        auto load_x = [&](RawSet& r, const StageSrc& s, int u) __attribute__((always_inline)) { r.x[u] = buf_load8<float>(s.xs, xvoff[u] + s.xt); };
        auto load_w = [&](RawSet& r, const StageSrc& s, int u) __attribute__((always_inline)) {
            const int it = pt + u * NP;
            const bool lo = it >= WS / 8;       // (wave-uniform: 64 divides WS / 8)
            const __amdgpu_buffer_rsrc_t srd = (s.live && it < WITEMS) ? (lo ? wsrd_l : wsrd_h) : make_srd(wg, 0);
            r.w[u] = buf_load8<bf16_t>(srd, w_src(it < WITEMS ? it : 0) + s.wo);
        };
        // halo tile (and, when streamed, operator chunk) of stage s from register set r into stage buffer s & 1; every item is re-loaded
        // for stage s + 2 as soon as it is staged
        auto commit = [&](int s, RawSet& r) __attribute__((always_inline)) {
            Raw8<float>(&rx)[XIPT] = r.x;
            Raw8<bf16_t>(&rw)[WIPT] = r.w;
            int b, h0, kc;
            coords(s, b, h0, kc);
            const StageSrc nx = stage_src(s + 2);
            u16_t* __restrict__ xh = planes + (s & 1) * 2 * XS;
            u16_t* __restrict__ xl = xh + XS;
            if (PRO == SED_PRO_BNRELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { psc[e] = pcoef[kc * 32 + cq * 8 + e]; psf[e] = pcoef[Cinp + kc * 32 + cq * 8 + e]; }
            }
            const int row_lo = h0 == 0 ? 1 : 0;
            const int row_hi = (H - h0 < ROWS - 1) ? (H - h0) : (ROWS - 1);
#pragma unroll
            for (int u = 0; u < XIPT; ++u) {
                const int it = pt + u * NP;
                if (u == XIPT - 1 && it >= XITEMS) { load_x(r, nx, u); continue; }      // (keeps the load count per stage fixed)
                float v[8];
                raw_to_f(rx[u], v);
                load_x(r, nx, u);
                if (PRO == SED_PRO_BNRELU) {     // padding rows must be zero AFTER the prologue: ReLU and the mask in one v_med3_f32
                    const int rowi = (it >> 2) / W;
                    const float top = (rowi >= row_lo && rowi <= row_hi) ? __builtin_inff() : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(fmaf(v[e], psc[e], psf[e]), 0.f, top);
                }
                sed_u32x4 hw, lw;
                split8<true, GRADOP>(v, hw, lw, pre);
                *reinterpret_cast<sed_u32x4*>(xh + xlds[u]) = hw;
                *reinterpret_cast<sed_u32x4*>(xl + xlds[u]) = lw;
            }
            if (!wres) {
                u16_t* __restrict__ wb = wop + (s & 1) * 2 * WS;
#pragma unroll
                for (int u = 0; u < WIPT; ++u) {
                    if (pt + u * NP < WITEMS) *reinterpret_cast<bf16x8*>(wb + (pt + u * NP) * 8) = rw[u].v;
                    load_w(r, nx, u);
                }
            }
        };

        // ---- coalesced epilogue (conv_x3_kernel's): staging image -> 32-byte items, whole lines; statistics on a thread's 8 channels ----
        const int fcg = pt % IPR, fq0 = pt / IPR;
        const int fl_lds0 = fq0 * BNP + fcg * 8;
        const unsigned fl_off0 = (unsigned)((fq0 * Coutp + n0 + fcg * 8) * 4);
        const unsigned fl_step = (unsigned)(FQS * Coutp * 4);
        Raw8<float> zraw[FIPT];
        float S8[8], Q8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { S8[e] = 0.f; Q8[e] = 0.f; }
        auto load_zref = [&](int tl) __attribute__((always_inline)) {      // the ReLU-backward gate's z of local tile tl (past the strip: zeros)
            const bool live = tl < ntl;
            const int tile = t_begin + (live ? tl : 0);
            const int b = tile / p.tilesPerImg, h0 = (tile - b * p.tilesPerImg) * TH;
            const __amdgpu_buffer_rsrc_t rs = make_srd(zr + (size_t)b * zimg, live ? zimg * 4 : 0);
            const unsigned tq = (unsigned)(h0 * W * Coutp * 4);
#pragma unroll
            for (int u = 0; u < FIPT; ++u) zraw[u] = buf_load8<float>(rs, fl_off0 + u * fl_step + tq);
        };
        auto flush = [&](int tl) __attribute__((always_inline)) {          // local tile tl sits complete in staging image tl & (nos - 1)
            const int tile = t_begin + tl;
            const int b = tile / p.tilesPerImg, h0 = (tile - b * p.tilesPerImg) * TH;
            const float* __restrict__ osb = os + (tl & (nos - 1)) * BM * BNP;
            const __amdgpu_buffer_rsrc_t zs = make_srd(zg + (size_t)b * zimg, zimg * 4);
            const unsigned tq = (unsigned)(h0 * W * Coutp * 4);
#pragma unroll
            for (int u = 0; u < FIPT; ++u) {
                float v[8];
                load8<float>(osb + fl_lds0 + u * FQS * BNP, v);
                const bool valid = h0 + (fq0 + u * FQS) / W < H;
                if (EPI == SED_EPI_STATS) {
                    if (valid) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { S8[e] += v[e]; Q8[e] = fmaf(v[e], v[e], Q8[e]); }
                    }
                }
                if (EPI == SED_EPI_RELUBWD) {
                    float z[8], ces[8], cet[8], cem[8];
                    raw_to_f(zraw[u], z);
                    load8<float>(ecoef + fcg * 8, ces);
                    load8<float>(ecoef + BN + fcg * 8, cet);
                    load8<float>(ecoef + 2 * BN + fcg * 8, cem);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gate = (valid && fmaf(z[e], ces[e], cet[e]) > 0.f) ? v[e] : 0.f;
                        v[e] = gate;
                        S8[e] += gate;
                        Q8[e] = fmaf(gate, z[e] - cem[e], Q8[e]);
                    }
                }
                buf_store8<float>(zs, fl_off0 + u * fl_step + tq, v);      // rows past the image: dropped by the descriptor's range check
            }
            if (EPI == SED_EPI_RELUBWD) load_zref(tl + 1);
        };

        {
            const StageSrc s0 = stage_src(0), s1 = stage_src(1);
#pragma unroll
            for (int u = 0; u < XIPT; ++u) load_x(rs0, s0, u);
            if (!wres) {
#pragma unroll
                for (int u = 0; u < WIPT; ++u) load_w(rs0, s0, u);
            }
#pragma unroll
            for (int u = 0; u < XIPT; ++u) load_x(rs1, s1, u);
            if (!wres) {
#pragma unroll
                for (int u = 0; u < WIPT; ++u) load_w(rs1, s1, u);
            }
            if (EPI == SED_EPI_RELUBWD) load_zref(0);
        }
        if (nst > 0) commit(0, rs0);
        x3c_barrier();                                       // stage 0 is staged
        // interval s: the epilogue of the tile whose last chunk was stage s - 1 (it sits complete in its staging image: the consumers
        // passed the barrier behind it), then stage s + 1 from the register set of its parity
        auto interval = [&](int s, RawSet& r) __attribute__((always_inline)) {
            if (!(kX3Stamps && (p.dbg & 4))) {          // (STAMPS build, SED_DBG & 4: idle loaders -- timing ablation, wrong results)
                if (s > 0 && (s % nchunks) == 0) flush(s / nchunks - 1);
                if (s + 1 < nst) commit(s + 1, r);
            }
            x3c_barrier();
        };
        for (int s = 0; s < nst; s += 2) {
            interval(s, rs1);
            if (s + 1 < nst) interval(s + 1, rs0);
        }
        if (nst > 0) flush(ntl - 1);

        // thread t accumulated channels 8*(t % IPR) .. +7 over its pixels: fixed-order sum over the FQS threads of each channel group
        // (SED_EPI_RELUBWD: Q was accumulated as gate*(z - mean), the 1/std factor is applied here)
        if (EPI == SED_EPI_STATS || EPI == SED_EPI_RELUBWD) {
            __syncthreads();                               // (both roles: the consumers are done with the planes the scratch aliases)
            float* red = reinterpret_cast<float*>(smem);   // [NP][16]
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[pt * 16 + e] = S8[e]; red[pt * 16 + 8 + e] = Q8[e]; }
            __syncthreads();
            if (pt < 2 * BN) {
                const int stat = pt / BN, cn = pt % BN;
                const int cg = cn >> 3, e = cn & 7;
                float tot = 0.f;
                for (int k = 0; k < FQS; ++k) tot += red[(cg + IPR * k) * 16 + stat * 8 + e];
                if (EPI == SED_EPI_RELUBWD && stat) tot *= p.epi_invstd[n0 + cn];
                p.partial[((size_t)bx * 2 + stat) * Coutp + n0 + cn] = tot;
                // rows of `partial` beyond the launched strips are zeroed (the finalize kernels read nparts rows)
                for (int row = bx + nbx; row < p.nparts; row += nbx) p.partial[((size_t)row * 2 + stat) * Coutp + n0 + cn] = 0.f;
            }
        }
    } else {
        // =============================== CONSUMERS =====================================================
        const int r = lane & 31, hh = lane >> 5;
        const float post_x = __builtin_ldexpf(XT::ILS, -p.xexp), post_h = __builtin_ldexpf(1.f, -p.xexp);
        const int q = wave * 32 + r;
        const int prow = q / W;
        const int rot = (W == 16) ? 12 * (prow & 1) : (W == 8) ? 4 * ((((prow & 3) + 1) >> 1) & 1) : 0;
        const int pcol = (q % W + rot) % W;
        const int xbase = (prow * WP + pcol) * PS;
        const int ostg = (prow * W + pcol) * BNP + 4 * hh;
        f32x16 ach, acx;
        x3c_barrier();                                       // stage 0 is staged
        for (int s = 0; s < nst; ++s) {
            const int tl = s / nchunks, kc = s - tl * nchunks;
            const u16_t* __restrict__ xh = planes + (s & 1) * 2 * XS;
            const u16_t* __restrict__ xl = xh + XS;
            const u16_t* __restrict__ whc = wop + (wres ? kc : (s & 1)) * 2 * WS;
            const u16_t* __restrict__ wlc = whc + WS;
            if (kc == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) { ach[i] = 0.f; acx[i] = 0.f; }
            }
            // 18 k-steps (tap, 16-channel half), software-pipelined by hand: the four fragment reads of step i + 2 are issued BEFORE the
            // three MFMAs of step i
            vec ah[3], al[3], bh[3], bl[3];                  // (a ring of three fragment sets: one wave per SIMD multiplies, nobody else covers the LDS round trip)
            auto frags = [&](int step, int buf) __attribute__((always_inline)) {
                const int tap = step >> 1, ks = step & 1;
                const int ti = tap / 3, tj = tap - 3 * ti;
                const int kb = ks * 16 + hh * 8;
                const int wo = ((tap * 4 + kb / 8) * BN + r) * 8;
                ah[buf] = lds_frag<vec>(whc + wo);
                al[buf] = lds_frag<vec>(wlc + wo);
                const int xo = xbase + (ti * WP + tj) * PS + kb;
                bh[buf] = lds_frag<vec>(xh + xo);
                bl[buf] = lds_frag<vec>(xl + xo);
            };
            if (!(kX3Stamps && (p.dbg & 2))) {               // (STAMPS build, SED_DBG & 2: no matrix loop -- timing ablation, wrong results)
                frags(0, 0);
                frags(1, 1);
#pragma unroll
                for (int step = 0; step < 18; ++step) {
                    const int cur = step % 3;
                    if (step + 2 < 18) frags(step + 2, (step + 2) % 3);
                    __builtin_amdgcn_sched_barrier(0);
                    acx = XT::mfma(al[cur], bh[cur], acx);
                    ach = XT::mfma(ah[cur], bh[cur], ach);
                    acx = XT::mfma(ah[cur], bl[cur], acx);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (kc == nchunks - 1) {                         // the tile is complete: results to its staging image (flushed by the producers next stage)
                float* __restrict__ osb = os + (tl & (nos - 1)) * BM * BNP;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(acx[4 * g + e], post_x, ach[4 * g + e] * post_h);
                    store4<float>(osb + ostg + 8 * g, v);
                }
            }
            x3c_barrier();
        }
        if (EPI == SED_EPI_STATS || EPI == SED_EPI_RELUBWD) {
            __syncthreads();
            __syncthreads();
        }
    }
}

template <int W, int PRO, int EPI>
int launch_conv_x3pc_t(ConvParams& p, hipStream_t st) {
    constexpr int TH = 128 / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr size_t lds_x = (size_t)4 * (TH + 2) * WP * 40 * 2;       // two stages of two planes
    constexpr size_t lds_w1 = (size_t)2 * 9 * 32 * 32 * 2;             // one operator chunk (hi + lo)
    constexpr size_t lds_o1 = (size_t)128 * 36 * 4;
    const int nchunks = p.Cinp / 32;
    const size_t lds_e = (size_t)3 * 32 * 4 + (PRO == SED_PRO_BNRELU ? (size_t)2 * p.Cinp * 4 : 0);
    const size_t lds_o = (nchunks == 1 ? 2 : 1) * lds_o1;
    // the whole operator resident when it fits beside the planes and the staging image; otherwise two chunk buffers
    p.wres = (lds_x + nchunks * lds_w1 + lds_o + lds_e <= (size_t)160 * 1024) ? 1 : 0;
    const size_t lds = lds_x + (p.wres ? nchunks : 2) * lds_w1 + lds_o + lds_e;
    if (lds > (size_t)160 * 1024) return -1;
    if (int rc_ = sed_set_max_lds<&conv_x3pc_kernel<W, PRO, EPI>>(lds)) return rc_;
    p.tilesPerImg = cdiv(p.H, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    const int ny = p.Coutp / 32;
    int nbx = kX3pcConvBlocks / ny;
    if (nbx > p.nparts) nbx = p.nparts;           // `partial` has nparts rows
    if (nbx > p.totalTiles) nbx = p.totalTiles;
    if (nbx < 1) nbx = 1;
    p.tpb = cdiv(p.totalTiles, nbx);
    conv_x3pc_kernel<W, PRO, EPI><<<dim3(nbx * ny), dim3(256 + 64 * SED_X3PC_NPW), lds, st>>>(p);
    return 0;
}

template <int W>
int dispatch_conv_x3pc_pe(ConvParams& p, hipStream_t st) {
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_STATS) return launch_conv_x3pc_t<W, SED_PRO_NONE, SED_EPI_STATS>(p, st);
    if (p.pro == SED_PRO_BNRELU && p.epi == SED_EPI_STATS) return launch_conv_x3pc_t<W, SED_PRO_BNRELU, SED_EPI_STATS>(p, st);
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_STORE) return launch_conv_x3pc_t<W, SED_PRO_NONE, SED_EPI_STORE>(p, st);
    if (p.pro == SED_PRO_BNRELU && p.epi == SED_EPI_STORE) return launch_conv_x3pc_t<W, SED_PRO_BNRELU, SED_EPI_STORE>(p, st);
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_RELUBWD) return launch_conv_x3pc_t<W, SED_PRO_NONE, SED_EPI_RELUBWD>(p, st);
    return -1;
}

}  // namespace

// Returns -1 when the shape / mode is not covered (the caller then takes conv_x3_kernel), otherwise 0 / an error code after the launch.
// EXPERIMENTS build only, selected with SED_X3_CONV=p.  Parity-green (tests/test_gpu_x3.py passes through it) and measured SLOWER than
// conv_x3_kernel on every layer (profiles/r06_ai_x3pc_conv_by_layer.txt: b1c2 fwd 0.90 against 0.82 ms, b2c2 0.90 against 0.72): the loader
// waves alone take 0.67-0.71 ms (profiles/r06_aj_*) -- four waves stage what eight waves stage in the two-workgroup form.
int launch_conv_x3pc(ConvParams& p, int W, hipStream_t st) {
    switch (W) {
        case 8: return dispatch_conv_x3pc_pe<8>(p, st);
        case 16: return dispatch_conv_x3pc_pe<16>(p, st);
        case 32: return dispatch_conv_x3pc_pe<32>(p, st);
        case 64: return dispatch_conv_x3pc_pe<64>(p, st);
    }
    return -1;
}
