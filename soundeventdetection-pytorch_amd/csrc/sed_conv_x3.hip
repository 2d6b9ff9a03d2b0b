// Split-operand convolutions: fp32 tensors, 16-bit pieces on the 16-bit matrix pipe.
//
// The reference-identity gate (frame logits within 1e-3 of the reference's fp32 CPU path, decisions / onsets bit-exact:
// /root/reference/models/spectogram_models.py:185-205, train.py:44) was only met by precision="fp32", whose convolutions run on
// v_mfma_f32_32x32x2f32 (157 TF dense peak: 39.8 ms per step).  Here every fp32 operand is split ONCE, where it is staged,
//     a = hi + lo / LS,   hi = r16(a),  lo = r16((a - hi) * LS)
// and a product runs as THREE 16-bit MFMAs with fp32 accumulation into two accumulators,
//     acc_h += hi(a).hi(b);   acc_x += lo(a).hi(b) + hi(a).lo(b);   result = acc_h + acc_x / LS
// (dropped: lo.lo and the two residuals).  Two piece types, one kernel:
//   SED_F32X3 "bf16x3": r16 = bf16, LS = 1.    |a - hi - lo| <= 2^-17 |a|: ~1e-5 relative per product.  Logits / decisions / onsets hold
//                       the gate; parameter gradients do NOT reach the fp32 kernels' noise level (BatchNorm's backward cancels ~3x per
//                       layer: percent-level error on the first layer of tiny cases) -- measured, recorded, kept for the A/B.
//   SED_F32H3 "f16x3":  r16 = fp16, LS = 2^11.  |a - hi - lo/LS| <= 2^-22 |a|: ~5e-7 per product, within a small factor of the fp32 MFMA.
//                       fp16 has five exponent bits: forward operands (z-scored features, BatchNorm'd activations, weights) sit in its
//                       normal range as they are; GRADIENT operands (~1/(B*T) per element) are pre-scaled by 2^e before the split and
//                       the accumulators by 2^-e afterwards -- e rides in bits 8..15 of the dtype argument (sed_hip.h), the host picks
//                       it from the shape (engine.py) -- and clamped to +-60000 (an outlier 2^20 above the typical gradient saturates
//                       instead of becoming inf).
// Tensors stay fp32 in HBM (element-wise kernels, BatchNorm statistics, the first layer and the head are the fp32 mode's); only the two
// GEMM-shaped kernels differ:
//   conv_x3_kernel   forward / data gradient.  One 256-thread workgroup per CU; a wave owns MT 32-pixel tiles x NT 32-channel output
//                    tiles of a (128 MT)-pixel stage: (MT, NT) = (2, 1), or (1, 2) when Cout % 64 == 0 (the halo tile is then split
//                    and staged once per 64 output channels).  Either way a k-step reads 6 fragments for 6 MFMAs.  Halo tile = two
//                    16-bit planes (padded-linear 80-byte pixels: conflict-free ds_read_b128 for all nine taps), operator = two packed
//                    16-bit images; BatchNorm+ReLU prologue in fp32 BEFORE the split; the epilogue (conv_igemm_kernel's, sed_conv.hip)
//                    stages the fp32 results in LDS and writes whole lines one stage later; BatchNorm statistics / the ReLU-backward
//                    gate and sums run in that pass on a thread's fixed 8 channels.
//   wgrad_x3_kernel  weight gradient with dz produced on load (DZ_POOL / DZ_BN / given): conv_wgrad2_kernel's structure with both
//                    operands as hi / lo planes read through ds_read_b64_tr_b16.
// The operator images come from sed_pack_conv_weight(s_batch) with the same dtype: [hi image][lo image], each in the bf16 layout.
#include "x3_common.h"

namespace {

// Tile-invariant plan of one thread's halo-tile items (32 input channels of an fp32 NHWC tensor) -> two 16-bit LDS planes.
// PS = LDS pixel stride in elements: 40 (padded-linear, forward / data gradient) or 32 (XOR-swizzled, weight gradient).
template <int W, int ROWS, int WP, int NTHR, int PS>
struct HaloPlanX3 {
    static constexpr int ITEMS = ROWS * (W + 2) * 4;
    static constexpr int IPT = (ITEMS + NTHR - 1) / NTHR;
    unsigned voff[IPT];
    int lds[IPT];
    unsigned colmask;
    Raw8<float> raw[IPT];

    __device__ __forceinline__ void init(int tid, int Cinp) {
        const int cq = tid & 3;
        colmask = 0;
#pragma unroll
        for (int u = 0; u < IPT; ++u) {
            const int it = tid + u * NTHR;
            const int pix = it >> 2;
            const int rowi = pix / (W + 2), coli = pix - rowi * (W + 2);
            const bool ok = (it < ITEMS) && coli >= 1 && coli <= W;
            voff[u] = ok ? (unsigned)(((rowi * W + coli) * Cinp + cq * 8) * 4) : SED_OOB;
            if (ok) colmask |= 1u << u;
            lds[u] = (rowi * WP + coli) * PS + ((PS == 32) ? ((cq * 8) ^ swz<bf16_t>(coli)) : cq * 8);
        }
    }
    __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t img, unsigned tile_off) {
#pragma unroll
        for (int u = 0; u < IPT; ++u) raw[u] = buf_load8<float>(img, voff[u] + tile_off);
    }
    template <bool HALF, bool GRADOP, int PRO>
    __device__ __forceinline__ void commit(u16_t* __restrict__ xh, u16_t* __restrict__ xl, int tid, const float* __restrict__ pro_scale,
                                           const float* __restrict__ pro_shift, int c0, int row_lo, int row_hi, float pre) const {
        const int cq = tid & 3;
        float sc[8], sh[8];
        if (PRO == SED_PRO_BNRELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = pro_scale[c0 + cq * 8 + e]; sh[e] = pro_shift[c0 + cq * 8 + e]; }
        }
#pragma unroll
        for (int u = 0; u < IPT; ++u) {
            const int it = tid + u * NTHR;
            if (u == IPT - 1 && it >= ITEMS) break;
            float v[8];
            raw_to_f(raw[u], v);
            if (PRO == SED_PRO_BNRELU) {     // padding must be zero AFTER the prologue: columns via colmask, rows on an image's first / last tile
                const int rowi = (it >> 2) / (W + 2);
                const bool keep = ((colmask >> u) & 1) && rowi >= row_lo && rowi <= row_hi;
                const float top = keep ? __builtin_inff() : 0.f;          // ReLU and the padding mask in one v_med3_f32: clamp to [0, top]
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(fmaf(v[e], sc[e], sh[e]), 0.f, top);
            }
            sed_u32x4 hw, lw;
            split8<HALF, GRADOP>(v, hw, lw, pre);
            *reinterpret_cast<sed_u32x4*>(xh + lds[u]) = hw;
            *reinterpret_cast<sed_u32x4*>(xl + lds[u]) = lw;
        }
    }
};

// =================================================================================================
// forward / data gradient
// =================================================================================================
// SM ("small"): one 32-pixel x 32-channel tile per wave, 128-pixel stages, the epilogue's staging image aliased onto the halo planes:
// <= 80 KB of LDS and <= 256 registers, so TWO workgroups share a CU and one's split / staging / flush phases run under the other's
// MFMA loop (the single-workgroup forms serialise them: ~35 % matrix-pipe utilisation).  Costs 8 instead of 6 fragment reads per 6 MFMAs.
// PAIR (small form, layers whose operator does not stay resident: Cin >= 64): the stages of TWO consecutive pixel tiles interleave --
// (tile A, chunk 0), (tile B, chunk 0), (tile A, chunk 1), ... -- so that an operator chunk is staged once per two stages; a second
// accumulator set (32 registers) carries tile B.  The operator chunks are 37 KB against a 23-35 KB halo tile: restaging them every stage
// was more than half of the bytes a workgroup pulls from L2 (profiles/r06_ac_x3_ablate_operator.txt: -12 % with no restaging at all).
template <bool HALF, int W, int NT, int PRO, int EPI, bool SM = false, bool PAIR = false>
__global__ __launch_bounds__(256, SM ? 2 : 1) void conv_x3_kernel(ConvParams p) {
    static_assert(!PAIR || SM, "tile pairs belong to the small form");
    typedef X3<HALF> XT;
    typedef typename XT::vec vec;
    constexpr int MT = SM ? 1 : 3 - NT, BM = 128 * MT, BN = 32 * NT, NTHR = 256;
    static_assert(NT == 1 || NT == 2, "one or two output tiles per wave");
    static_assert(!SM || NT == 1, "the small form owns one output tile per wave");
    constexpr int TH = BM / W;
    constexpr int WP = (SM && W == 64) ? 66 : (W + 2 + 3) & ~3;     // (SM at W = 64: an unpadded pitch keeps planes + operator under 80 KB)
    constexpr int ROWS = TH + 2;
    constexpr int PS = 40;
    constexpr int XS = ROWS * WP * PS;       // elements per plane
    constexpr int WS = 9 * 32 * BN;          // elements per plane and 32-channel chunk
    constexpr int WITEMS = WS / 8;
    constexpr int WIPT = (WITEMS + NTHR - 1) / NTHR;
    constexpr int BNP = BN + 4;
    static_assert(BM % W == 0, "tile shape");
    typedef HaloPlanX3<W, ROWS, WP, NTHR, PS> XPlan;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool wres = p.wres != 0;
    const int nchunks = p.Cinp >> 5;
    u16_t* xh = reinterpret_cast<u16_t*>(smem);
    u16_t* xl = xh + XS;
    u16_t* wh = xl + XS;                                   // [wres ? nchunks : 1][WS]
    u16_t* wl = wh + (wres ? nchunks : 1) * WS;
    // [BM][BNP]: output staging of the coalesced epilogue (SM: in the halo planes, which nobody reads between a tile's last MFMA and the
    // next stage's commit)
    float* os = SM ? reinterpret_cast<float*>(smem) : reinterpret_cast<float*>(wl + (wres ? nchunks : 1) * WS);
    static_assert(!SM || (size_t)BM * BNP * 4 <= (size_t)2 * XS * 2, "staging image must fit in the halo planes");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int NY = p.Coutp / BN;
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int by = logical % NY, bx = logical / NY;
    const int n0 = by * BN;
    const int H = p.H, Cinp = p.Cinp, Coutp = p.Coutp;
    const float* __restrict__ xg = reinterpret_cast<const float*>(p.x);
    const u16_t* __restrict__ wg = reinterpret_cast<const u16_t*>(p.wpack);
    float* __restrict__ zg = reinterpret_cast<float*>(p.z);
    const float* __restrict__ zr = reinterpret_cast<const float*>(p.zref);
    constexpr int epi = EPI;
    constexpr bool GRADOP = HALF && EPI != SED_EPI_STATS;         // (the exponent is 0 for forward calls: the scale is then 1)
    const float pre = __builtin_ldexpf(1.f, p.xexp);
    const float post_x = __builtin_ldexpf(XT::ILS, -p.xexp), post_h = __builtin_ldexpf(1.f, -p.xexp);

    int prow[MT], xbase[MT], ostg[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (wave * MT + mt) * 32 + r;
        prow[mt] = q / W;
        const int rot = (W == 16) ? 12 * (prow[mt] & 1) : (W == 8) ? 4 * ((((prow[mt] & 3) + 1) >> 1) & 1) : 0;
        const int pcol = (q % W + rot) % W;
        xbase[mt] = (prow[mt] * WP + pcol) * PS;
        ostg[mt] = (prow[mt] * W + pcol) * BNP + 4 * hh;
    }
    XPlan xp;
    xp.init(tid, Cinp);
    // operator chunk items: row (tap, kq) of BN*8 contiguous elements in LDS; source row stride Coutp*8; the lo image follows the hi image
    unsigned wsrc[WIPT];
    int wdst[WIPT];
    Raw8<bf16_t> wrh[WIPT], wrl[WIPT];
    {
        constexpr int ROWLEN = BN * 8;
        constexpr int ITEMS_PER_ROW = ROWLEN / 8;
#pragma unroll
        for (int u = 0; u < WIPT; ++u) {
            const int it = tid + u * NTHR;
            const int rowi = it / ITEMS_PER_ROW, off = (it - rowi * ITEMS_PER_ROW) * 8;
            const bool ok = it < WITEMS;
            wsrc[u] = ok ? (unsigned)(((rowi * Coutp + n0) * 8 + off) * 2) : SED_OOB;
            wdst[u] = ok ? rowi * ROWLEN + off : 0;
        }
    }
    const size_t wchunk_bytes = (size_t)(9 * 4) * Coutp * 8 * 2;          // one 32-input-channel chunk of one image
    const size_t wimg_bytes = wchunk_bytes * nchunks;
    const __amdgpu_buffer_rsrc_t wsrd_h = make_srd(wg, wimg_bytes);
    const __amdgpu_buffer_rsrc_t wsrd_l = make_srd(reinterpret_cast<const char*>(wg) + wimg_bytes, wimg_bytes);
    const size_t ximg = (size_t)H * W * Cinp, zimg = (size_t)H * W * Coutp;

    // ---- coalesced epilogue (conv_igemm_kernel's): results -> fp32 staging image -> 32-byte items, whole lines -----------------
    constexpr int IPR = BN / 8;
    constexpr int FIPT = BM * IPR / NTHR;
    constexpr int FQS = NTHR / IPR;
    static_assert((BM * IPR) % NTHR == 0 && NTHR % IPR == 0, "flush geometry");
    const int fcg = tid % IPR, fq0 = tid / IPR;
    const int fl_lds0 = fq0 * BNP + fcg * 8;
    const unsigned fl_off0 = (unsigned)((fq0 * Coutp + n0 + fcg * 8) * 4);
    const unsigned fl_step = (unsigned)(FQS * Coutp * 4);
    Raw8<float> zraw[FIPT];
    float S8[8], Q8[8];                  // statistics of this thread's 8 channels over the pixels it flushes
#pragma unroll
    for (int e = 0; e < 8; ++e) { S8[e] = 0.f; Q8[e] = 0.f; }
    // ReLU-backward epilogue: scale / shift / mean of the workgroup's BN channels, in LDS behind the operator ([3][BN] floats; as 24
    // registers per thread they were what kept the tile-pair form from fitting at W = 32)
    float* ecoef = reinterpret_cast<float*>(wl + (wres ? nchunks : 1) * WS) + (SM ? 0 : BM * BNP);
    if (epi == SED_EPI_RELUBWD) {
        for (int i = tid; i < 3 * BN; i += NTHR) {
            const int a = i / BN, c = i - a * BN;
            ecoef[i] = (a == 0 ? p.epi_scale : a == 1 ? p.epi_shift : p.epi_mean)[n0 + c];
        }
    }
    int fb = 0, fh0 = 0;
    bool pending = false;
    auto flush = [&]() {
        const __amdgpu_buffer_rsrc_t zs = make_srd(zg + (size_t)fb * zimg, zimg * 4);
        const unsigned tq = (unsigned)(fh0 * W * Coutp * 4);
#pragma unroll
        for (int u = 0; u < FIPT; ++u) {
            float v[8];
            load8<float>(os + fl_lds0 + u * FQS * BNP, v);
            if (epi == SED_EPI_STATS) {
                if (fh0 + (fq0 + u * FQS) / W < H) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { S8[e] += v[e]; Q8[e] = fmaf(v[e], v[e], Q8[e]); }
                }
            }
            if (epi == SED_EPI_RELUBWD) {
                float z[8], ces[8], cet[8], cem[8];
                raw_to_f(zraw[u], z);
                load8<float>(ecoef + fcg * 8, ces);
                load8<float>(ecoef + BN + fcg * 8, cet);
                load8<float>(ecoef + 2 * BN + fcg * 8, cem);
                const bool valid = fh0 + (fq0 + u * FQS) / W < H;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gate = (valid && fmaf(z[e], ces[e], cet[e]) > 0.f) ? v[e] : 0.f;
                    v[e] = gate;
                    S8[e] += gate;
                    Q8[e] = fmaf(gate, z[e] - cem[e], Q8[e]);
                }
            }
            buf_store8<float>(zs, fl_off0 + u * fl_step + tq, v);
        }
    };

    const int t_begin = bx * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    const int ntl = t_end > t_begin ? (t_end - t_begin) : 0;
    const int nst = (PAIR ? ((ntl + 1) & ~1) : ntl) * nchunks;      // (PAIR: an odd strip's last pair has stages without a tile: skipped)

    // (stage -> tile / chunk / image by exact multiply-shift divisions, sed_fastdiv: three scalar instructions instead of the ~20 of a run-time
    //  division, six divisions per stage and wave -- the scalar bookkeeping was a quarter of the kernel's instructions)
    const int sdiv = PAIR ? 2 * nchunks : nchunks;          // p.nch_M / nch_l: the magic of this divisor (launch_x3)
    auto stage_tile = [&](int s) -> int {
        const int g = sed_fastdiv(s, p.nch_M, p.nch_l);
        return PAIR ? t_begin + 2 * g + (s & 1) : t_begin + g;
    };
    auto coords = [&](int s, int& b, int& h0, int& kc) {
        const int g = sed_fastdiv(s, p.nch_M, p.nch_l);
        kc = PAIR ? (s - g * sdiv) >> 1 : s - g * sdiv;
        const int tile = PAIR ? t_begin + 2 * g + (s & 1) : t_begin + g;
        b = sed_fastdiv(tile, p.tpi_M, p.tpi_l);
        h0 = (tile - b * p.tilesPerImg) * TH;
    };
    auto issue = [&](int s, bool with_w) {
        int b, h0, kc;
        coords(s, b, h0, kc);
        xp.issue(make_srd(xg + (size_t)b * ximg, ximg * 4), (unsigned)((((h0 - 1) * W - 1) * Cinp + kc * 32) * 4));
        if (with_w) {
            const unsigned wo = (unsigned)(kc * wchunk_bytes);
#pragma unroll
            for (int u = 0; u < WIPT; ++u) { wrh[u] = buf_load8<bf16_t>(wsrd_h, wsrc[u] + wo); wrl[u] = buf_load8<bf16_t>(wsrd_l, wsrc[u] + wo); }
        }
    };
    auto commit = [&](int s, bool with_w) {
        int b, h0, kc;
        coords(s, b, h0, kc);
        const int row_hi = (H - h0 < ROWS - 1) ? (H - h0) : (ROWS - 1);
        xp.template commit<HALF, GRADOP, PRO>(xh, xl, tid, p.pro_scale, p.pro_shift, kc * 32, h0 == 0 ? 1 : 0, row_hi, pre);
        if (with_w) {
#pragma unroll
            for (int u = 0; u < WIPT; ++u) {
                if (u == WIPT - 1 && tid + u * NTHR >= WITEMS) break;
                *reinterpret_cast<bf16x8*>(wh + wdst[u]) = wrh[u].v;
                *reinterpret_cast<bf16x8*>(wl + wdst[u]) = wrl[u].v;
            }
        }
    };
    if (wres && nst > 0) {
        for (int c = 0; c < nchunks; ++c) {
            const unsigned wo = (unsigned)(c * wchunk_bytes);
#pragma unroll
            for (int u = 0; u < WIPT; ++u) { wrh[u] = buf_load8<bf16_t>(wsrd_h, wsrc[u] + wo); wrl[u] = buf_load8<bf16_t>(wsrd_l, wsrc[u] + wo); }
#pragma unroll
            for (int u = 0; u < WIPT; ++u) {
                if (u == WIPT - 1 && tid + u * NTHR >= WITEMS) break;
                *reinterpret_cast<bf16x8*>(wh + c * WS + wdst[u]) = wrh[u].v;
                *reinterpret_cast<bf16x8*>(wl + c * WS + wdst[u]) = wrl[u].v;
            }
        }
    }
    // operator staging: every stage of a multi-chunk layer that does not keep all chunks resident, once for a single-chunk layer
    const bool stage_w_each = !wres && nchunks > 1 && !(kX3Stamps && (p.dbg & 8));     // (STAMPS build, SED_DBG & 8: operator staged once -- timing ablation, wrong results)

    f32x16 ach[NT][MT], acx[NT][MT];
    f32x16 ach1[NT][MT], acx1[NT][MT];                             // PAIR: the second tile of the pair (never touched otherwise)
    int cur_set = 0;
    if (nst > 0) issue(0, !wres);
    for (int s = 0; s < nst; ++s) {
        if (PAIR && stage_tile(s) >= t_end) continue;      // (the tile-less stages of an odd strip's last pair)
        const int sub = PAIR ? (s & 1) : 0;
        int b, h0, kc;
        coords(s, b, h0, kc);
        __syncthreads();                                   // previous stage's readers of the planes (and of the staging image) are done
        if (!SM && pending) { flush(); pending = false; }
        const bool need_w = PAIR ? (sub == 0) : (stage_w_each || (!wres && s == 0));
        if (!(kX3Stamps && (p.dbg & 4))) commit(s, need_w);   // (STAMPS build, SED_DBG & 4: no split / staging -- timing ablation)
        __syncthreads();
        {
            int nx = s + 1;
            if (PAIR && nx < nst && stage_tile(nx) >= t_end) nx += 1;
            if (nx < nst) issue(nx, PAIR ? ((nx & 1) == 0) : stage_w_each);
        }
        const unsigned tq = (unsigned)(h0 * W * Coutp * 4);
        if (epi == SED_EPI_RELUBWD && kc == nchunks - 1) {
            const __amdgpu_buffer_rsrc_t rs = make_srd(zr + (size_t)b * zimg, zimg * 4);
#pragma unroll
            for (int u = 0; u < FIPT; ++u) zraw[u] = buf_load8<float>(rs, fl_off0 + u * fl_step + tq);
        }
        const u16_t* __restrict__ whc = wh + (wres ? kc * WS : 0);
        const u16_t* __restrict__ wlc = wl + (wres ? kc * WS : 0);
        // one stage's products into accumulator set (AH, AX); the last chunk's stage also moves the results to the staging image
        auto stage_mm = [&](f32x16 (&AH)[NT][MT], f32x16 (&AX)[NT][MT]) __attribute__((always_inline)) {
        if (kc == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) { AH[nt][mt][i] = 0.f; AX[nt][mt][i] = 0.f; }
        }
        // 18 k-steps (tap, 16-channel half), software-pipelined by hand: the six fragment reads of step i + 1 are issued BEFORE the six
        // MFMAs of step i (one wave per SIMD: nobody else covers the LDS round trip; left alone hipcc sinks the reads next to their use)
        vec ah[2][NT], al[2][NT], bh[2][MT], bl[2][MT];
        auto frags = [&](int step, int buf) __attribute__((always_inline)) {
            const int tap = step >> 1, ks = step & 1;
            const int ti = tap / 3, tj = tap - 3 * ti;
            const int kb = ks * 16 + hh * 8;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int wo = ((tap * 4 + kb / 8) * BN + nt * 32 + r) * 8;
                ah[buf][nt] = lds_frag<vec>(whc + wo);
                al[buf][nt] = lds_frag<vec>(wlc + wo);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int xo = xbase[mt] + (ti * WP + tj) * PS + kb;
                bh[buf][mt] = lds_frag<vec>(xh + xo);
                bl[buf][mt] = lds_frag<vec>(xl + xo);
            }
        };
        if (!(kX3Stamps && (p.dbg & 2))) frags(0, 0);         // (STAMPS build, SED_DBG & 2: no matrix loop -- timing ablation, wrong results)
#pragma unroll
        for (int step = 0; step < 18; ++step) {
            if (kX3Stamps && (p.dbg & 2)) break;
            const int cur = step & 1;
            if (step + 1 < 18) frags(step + 1, cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            // six MFMAs: two uses of an accumulator are two (different-accumulator) MFMAs apart
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) AX[nt][mt] = XT::mfma(al[cur][nt], bh[cur][mt], AX[nt][mt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) AH[nt][mt] = XT::mfma(ah[cur][nt], bh[cur][mt], AH[nt][mt]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) AX[nt][mt] = XT::mfma(ah[cur][nt], bl[cur][mt], AX[nt][mt]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kc != nchunks - 1) return;
        if (SM) __syncthreads();                           // every wave is done reading the halo planes the staging image aliases
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(AX[nt][mt][4 * g + e], post_x, AH[nt][mt][4 * g + e] * post_h);
                    store4<float>(os + ostg[mt] + nt * 32 + 8 * g, v);
                }
            }
        };
        if constexpr (PAIR) {       // the accumulators of the tile this stage belongs to move into (ach, acx): 32 v_swap_b32 per stage
            if (sub != cur_set) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const f32x16 th = ach[nt][mt], tx = acx[nt][mt];
                        ach[nt][mt] = ach1[nt][mt]; acx[nt][mt] = acx1[nt][mt];
                        ach1[nt][mt] = th; acx1[nt][mt] = tx;
                    }
                cur_set = sub;
            }
        }
        stage_mm(ach, acx);
        if (kc != nchunks - 1) continue;
        fb = b; fh0 = h0;
        if (SM) {
            __syncthreads();
            flush();                                       // (the loop-top barrier keeps the next commit off the staging image until every thread has read it)
        } else {
            pending = true;
        }
    }
    if (!SM && pending) {
        __syncthreads();
        flush();
    }

    // thread t accumulated channels 8*(t % IPR) .. +7 over its pixels: fixed-order sum over the FQS threads of each channel group
    // (SED_EPI_RELUBWD: Q was accumulated as gate*(z - mean), the 1/std factor is applied here)
    if (epi == SED_EPI_STATS || epi == SED_EPI_RELUBWD) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);   // [NTHR][16] (reuses the tile buffers)
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = S8[e]; red[tid * 16 + 8 + e] = Q8[e]; }
        __syncthreads();
        if (tid < 2 * BN) {
            const int stat = tid / BN, cn = tid % BN;
            const int cg = cn >> 3, e = cn & 7;
            float tot = 0.f;
            for (int k = 0; k < FQS; ++k) tot += red[(cg + IPR * k) * 16 + stat * 8 + e];
            if (epi == SED_EPI_RELUBWD && stat) tot *= p.epi_invstd[n0 + cn];
            p.partial[((size_t)bx * 2 + stat) * Coutp + n0 + cn] = tot;
        }
    }
}

// =================================================================================================
// weight gradient (conv_wgrad2_kernel's structure): dW[tap][cin][cout] = sum_pix a[pix + tap][cin] * dz[pix][cout]
// =================================================================================================
// (fp16 pieces at 32 output channels: a FOURTH wave that only loads, splits and stages -- with 256 threads a thread's prefetch registers
//  shrink from 3 + 6 to 2 + 5 items and the kernel fits two workgroups per SIMD beside its 96 accumulator registers: 1.8 -> 1.35 ms on
//  block 0's layer.  An unscaled-lo single accumulator, 48 registers, was tried first: it loses the lo piece of every element 2^9 below
//  the typical one -- the matrix pipe flushes fp16 subnormals -- and G8's six-step Adam trajectory left its gate.)
template <bool HALF, int WN> struct WgX3Threads { static constexpr int N = (HALF && WN == 1) ? 256 : 192 * WN; };

template <bool HALF, int W, int WN, int DZ, int PRO>
__global__ __launch_bounds__((WgX3Threads<HALF, WN>::N)) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_x3_kernel(Wgrad2Params p) {
    typedef X3<HALF> XT;
    typedef typename XT::vec vec;
    constexpr int NTHR = WgX3Threads<HALF, WN>::N;
    constexpr int MWAVES = 3 * WN;                     // waves that issue MFMAs (the rest only stage)
    constexpr int BM = 128;
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr int ROWS = TH + 2;
    constexpr int XS = ROWS * WP * 32;
    constexpr int CO = 32 * WN;
    constexpr int IPP = CO / 8;
    constexpr int DITEMS = BM * IPP;
    constexpr int DIT = (DITEMS + NTHR - 1) / NTHR;
    typedef HaloPlanX3<W, ROWS, WP, NTHR, 32> XPlan;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    u16_t* xh = reinterpret_cast<u16_t*>(smem);
    u16_t* xl = xh + XS;
    u16_t* dh = xl + XS;                               // [WN][BM][32]
    u16_t* dl = dh + WN * BM * 32;
    float* coef = reinterpret_cast<float*>(dl + WN * BM * 32);   // [5][CO]: scale, shift, ca, cb, cc

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool mwave = wave < MWAVES;
    const int wt = mwave ? wave % 3 : 0, wn = mwave ? wave / 3 : 0;
    const int hh = lane >> 5;
    const int H = p.H, Cinp = p.Cinp, Coutp = p.Coutp;
    const int NCO = Coutp / CO;
    const int NY = (Cinp >> 5) * NCO;
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int strip = logical / NY, yb = logical - strip * NY;
    const int ci_tile = yb / NCO;
    const int ci0 = ci_tile * 32, co0 = (yb % NCO) * CO;
    const float* __restrict__ xg = reinterpret_cast<const float*>(p.x);
    const float* __restrict__ dg = reinterpret_cast<const float*>(p.dz);
    const float* __restrict__ zsg = reinterpret_cast<const float*>(p.zsrc);
    float* __restrict__ dzo = (ci_tile == 0) ? reinterpret_cast<float*>(p.dz_out) : nullptr;
    const int psh = p.pool >> 1;
    const int Ho = H >> psh, Wo = W >> psh;
    const float inv_pool = psh ? 0.25f : 1.0f;
    const float pre = __builtin_ldexpf(1.f, p.dzexp);
    const float post_x = __builtin_ldexpf(XT::ILS, -p.dzexp), post_h = __builtin_ldexpf(1.f, -p.dzexp);

    if (DZ != DZ_GIVEN) {
        for (int i = tid; i < 5 * CO; i += NTHR) {
            const int a = i / CO, c = i - a * CO;
            const float* src = (a == 0) ? p.scale : (a == 1) ? p.shift : (a == 2) ? p.ca : (a == 3) ? p.cb : p.cc;
            float v = (src != nullptr) ? src[co0 + c] : 0.f;
            if (a == 2 && DZ == DZ_POOL) v *= inv_pool;
            coef[i] = v;
        }
    }

    f32x16 ach[3], acx[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { ach[t][i] = 0.f; acx[t][i] = 0.f; }

    int offA[3][2], offB[2];
    {
        const int i16 = lane & 15, gbit = (lane >> 4) & 1;
        const int qq = i16 >> 2, pp = i16 & 3, ch = 16 * gbit + 4 * pp;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int kl = 8 * hh + qq + 4 * half;
            const int rq = kl / W, cq = kl % W;
#pragma unroll
            for (int tj = 0; tj < 3; ++tj) offA[tj][half] = ((rq + wt) * WP + cq + tj) * 32 + (ch ^ swz<bf16_t>(cq + tj));
            offB[half] = (wn * BM + kl) * 32 + ch;
        }
    }

    XPlan xp;
    xp.init(tid, Cinp);
    unsigned dvoff[DIT], pvoff[DIT];
    int dlds[DIT], dq[DIT];
    Raw8<float> da[DIT], db[DIT];
#pragma unroll
    for (int u = 0; u < DIT; ++u) {
        const int it = tid + u * NTHR;
        const int q = it / IPP, c8 = (it - q * IPP) * 8;
        const bool ok = it < DITEMS;
        dq[u] = ok ? q : BM;
        dvoff[u] = ok ? (unsigned)((q * Coutp + co0 + c8) * 4) : SED_OOB;
        pvoff[u] = ok ? (unsigned)(((((q / W) >> psh) * Wo + ((q % W) >> psh)) * Coutp + co0 + c8) * 4) : SED_OOB;
        dlds[u] = ((c8 >> 5) * BM + (ok ? q : 0)) * 32 + (c8 & 31);
    }
    const size_t ximg = (size_t)H * W * Cinp, zimg = (size_t)H * W * Coutp, pimg = (size_t)Ho * Wo * Coutp;

    auto issue = [&](int tile) {
        const int b = sed_fastdiv(tile, p.tpi_M, p.tpi_l);
        const int h0 = (tile - b * p.tilesPerImg) * TH;
        xp.issue(make_srd(xg + (size_t)b * ximg, ximg * 4), (unsigned)((((h0 - 1) * W - 1) * Cinp + ci0) * 4));
        const unsigned dt = (unsigned)(h0 * W * Coutp * 4);
        if (DZ == DZ_POOL) {
            const __amdgpu_buffer_rsrc_t gs = make_srd(dg + (size_t)b * pimg, pimg * 4);
            const __amdgpu_buffer_rsrc_t zs = make_srd(zsg + (size_t)b * zimg, zimg * 4);
            const unsigned pt = (unsigned)((h0 >> psh) * Wo * Coutp * 4);
#pragma unroll
            for (int u = 0; u < DIT; ++u) { da[u] = buf_load8<float>(gs, pvoff[u] + pt); db[u] = buf_load8<float>(zs, dvoff[u] + dt); }
        } else if (DZ == DZ_BN) {
            const __amdgpu_buffer_rsrc_t gs = make_srd(dg + (size_t)b * zimg, zimg * 4);
            const __amdgpu_buffer_rsrc_t zs = make_srd(zsg + (size_t)b * zimg, zimg * 4);
#pragma unroll
            for (int u = 0; u < DIT; ++u) { da[u] = buf_load8<float>(gs, dvoff[u] + dt); db[u] = buf_load8<float>(zs, dvoff[u] + dt); }
        } else {
            const __amdgpu_buffer_rsrc_t gs = make_srd(dg + (size_t)b * zimg, zimg * 4);
#pragma unroll
            for (int u = 0; u < DIT; ++u) da[u] = buf_load8<float>(gs, dvoff[u] + dt);
        }
    };

    auto commit = [&](int tile) {
        const int b = sed_fastdiv(tile, p.tpi_M, p.tpi_l);
        const int h0 = (tile - b * p.tilesPerImg) * TH;
        const int row_hi = (H - h0 < ROWS - 1) ? (H - h0) : (ROWS - 1);
        xp.template commit<HALF, false, PRO>(xh, xl, tid, p.pro_scale, p.pro_shift, ci0, h0 == 0 ? 1 : 0, row_hi, 1.f);
        const int qmax = (H - h0) * W;
        const __amdgpu_buffer_rsrc_t os = make_srd(dzo ? dzo + (size_t)b * zimg : nullptr, dzo ? zimg * 4 : 0);
        const unsigned dt = (unsigned)(h0 * W * Coutp * 4);
#pragma unroll
        for (int u = 0; u < DIT; ++u) {
            if (u == DIT - 1 && dq[u] >= BM) break;
            float v[8];
            if (DZ == DZ_GIVEN) {
                raw_to_f(da[u], v);                  // rows past the image were read as zeros
            } else {
                const int c8 = (dlds[u] & 31) + 32 * (dlds[u] / (BM * 32));
                float g[8], z[8];
                raw_to_f(da[u], g);
                raw_to_f(db[u], z);
                const f32x4* cf = reinterpret_cast<const f32x4*>(coef);
#pragma unroll
                for (int e4 = 0; e4 < 2; ++e4) {
                    const int ci4 = (c8 >> 2) + e4;
                    const f32x4 a4 = cf[2 * (CO / 4) + ci4], b4 = cf[3 * (CO / 4) + ci4], c4 = cf[4 * (CO / 4) + ci4];
                    f32x4 s4, t4;
                    if (DZ == DZ_POOL) { s4 = cf[0 * (CO / 4) + ci4]; t4 = cf[1 * (CO / 4) + ci4]; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = e4 * 4 + e;
                        const float base = fmaf(b4[e], z[i], c4[e]);
                        const float full = fmaf(a4[e], g[i], base);
                        if (DZ == DZ_POOL) v[i] = (fmaf(z[i], s4[e], t4[e]) > 0.f) ? full : base;
                        else v[i] = full;
                    }
                }
                if (qmax < BM && dq[u] >= qmax) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = 0.f;
                }
                if (dzo != nullptr) buf_store8<float>(os, dvoff[u] + dt, v);
            }
            sed_u32x4 hw, lw;
            split8<HALF, HALF>(v, hw, lw, pre);
            *reinterpret_cast<sed_u32x4*>(dh + dlds[u]) = hw;
            *reinterpret_cast<sed_u32x4*>(dl + dlds[u]) = lw;
        }
    };

    const int t_begin = strip * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    auto stamp = [&]() -> unsigned long long { return kX3Stamps ? __builtin_amdgcn_s_memtime() : 0ull; };
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};       // barrier A, wait for loads, commit, barrier B, issue, matrix loop
    if (t_begin < t_end) issue(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        const unsigned long long s0 = stamp();
        __syncthreads();
        const unsigned long long s1 = stamp();
        if (kX3Stamps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long s2 = stamp();
        if (!(kX3Stamps && (p.dbg & 4))) commit(tile);      // (STAMPS build, SED_DBG & 4: no split / staging -- timing ablation)
        const unsigned long long s3 = stamp();
        __syncthreads();
        const unsigned long long s4 = stamp();
        if (tile + 1 < t_end) issue(tile + 1);
        const unsigned long long s5 = stamp();
        if (kX3Stamps) { tph[0] += s1 - s0; tph[1] += s2 - s1; tph[2] += s3 - s2; tph[3] += s4 - s3; tph[4] += s5 - s4; }
        if (mwave && !(kX3Stamps && (p.dbg & 2))) {        // (STAMPS build, SED_DBG & 2: no matrix loop -- timing ablation, wrong results)
            auto ld_b = [&](const u16_t* pl, int ks) __attribute__((always_inline)) {
                return lds_frag_tr<vec>(pl + ks * 16 * 32 + offB[0], pl + ks * 16 * 32 + offB[1]);
            };
            auto ld_a = [&](const u16_t* pl, int ks, int tj) __attribute__((always_inline)) {
                const int k0 = ks * 16, ub = ((k0 / W) * WP + (k0 % W)) * 32;
                return lds_frag_tr<vec>(pl + ub + offA[tj][0], pl + ub + offA[tj][1]);
            };
            if constexpr (HALF) {
                // Eight k-steps of nine MFMAs, software-pipelined WITHOUT a second fragment set (the 96 accumulator + 64 prefetch registers
                // leave no room for one): the three products run in the order ah.bl, ah.bh, al.bh, and each group's dead fragments are
                // re-read for the next k-step as soon as the group has issued -- every fragment is in flight for >= 3 MFMAs (96 cycles)
                // before its first use.  Read-all / wait / multiply exposed the LDS round trip once per k-step: 56 cycles per MFMA.
                vec bh = ld_b(dh, 0), bl = ld_b(dl, 0), ah[3], al[3];
#pragma unroll
                for (int tj = 0; tj < 3; ++tj) { ah[tj] = ld_a(xh, 0, tj); al[tj] = ld_a(xl, 0, tj); }
#pragma unroll
                for (int ks = 0; ks < BM / 16; ++ks) {
                    const bool more = ks + 1 < BM / 16;
#pragma unroll
                    for (int tj = 0; tj < 3; ++tj) acx[tj] = XT::mfma(ah[tj], bl, acx[tj]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) bl = ld_b(dl, ks + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int tj = 0; tj < 3; ++tj) ach[tj] = XT::mfma(ah[tj], bh, ach[tj]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
#pragma unroll
                        for (int tj = 0; tj < 3; ++tj) ah[tj] = ld_a(xh, ks + 1, tj);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int tj = 0; tj < 3; ++tj) acx[tj] = XT::mfma(al[tj], bh, acx[tj]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (more) {
                        bh = ld_b(dh, ks + 1);
#pragma unroll
                        for (int tj = 0; tj < 3; ++tj) al[tj] = ld_a(xl, ks + 1, tj);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {        // bf16 pieces carry no lo scale: one accumulator (48 registers instead of 96: two workgroups per SIMD)
#pragma unroll 2
                for (int ks = 0; ks < BM / 16; ++ks) {
                    const vec bh = ld_b(dh, ks), bl = ld_b(dl, ks);
                    vec ah[3], al[3];
#pragma unroll
                    for (int tj = 0; tj < 3; ++tj) { ah[tj] = ld_a(xh, ks, tj); al[tj] = ld_a(xl, ks, tj); }
#pragma unroll
                    for (int tj = 0; tj < 3; ++tj) ach[tj] = XT::mfma(al[tj], bh, ach[tj]);
#pragma unroll
                    for (int tj = 0; tj < 3; ++tj) ach[tj] = XT::mfma(ah[tj], bl, ach[tj]);
#pragma unroll
                    for (int tj = 0; tj < 3; ++tj) ach[tj] = XT::mfma(ah[tj], bh, ach[tj]);
                }
            }
        }
        if (kX3Stamps) tph[5] += stamp() - s5;
    }
    if (kX3Stamps && (p.dbg & 16) && logical == 300 && lane == 0 && t_end > t_begin) {
        const unsigned long long n = t_end - t_begin;
        printf("wgrad_x3 W=%d WN=%d DZ=%d wave %d: %llu tiles; per tile: barrierA %llu  loads %llu  commit %llu  barrierB %llu  issue %llu  matrix %llu ticks\n",
               W, WN, DZ, wave, n, tph[0] / n, tph[1] / n, tph[2] / n, tph[3] / n, tph[4] / n, tph[5] / n);
    }

    if (!mwave) return;
    float* out = p.ws + (size_t)strip * 9 * Cinp * Coutp;
#pragma unroll
    for (int tj = 0; tj < 3; ++tj) {
        const int tap = wt * 3 + tj;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int cin = ci0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            out[((size_t)tap * Cinp + cin) * Coutp + co0 + wn * 32 + (lane & 31)] = fmaf(acx[tj][i], post_x, ach[tj][i] * post_h);
        }
    }
}

// ---- launchers ---------------------------------------------------------------------------------------------------------------
template <bool HALF, int W, int NT, int PRO, int EPI, bool SM = false, bool PAIR = false>
int launch_x3(ConvParams& p, hipStream_t st) {
    constexpr int BM = SM ? 128 : 128 * (3 - NT);
    constexpr int TH = BM / W;
    constexpr int WP = (SM && W == 64) ? 66 : (W + 2 + 3) & ~3;
    constexpr size_t lds_x = (size_t)2 * (TH + 2) * WP * 40 * 2;
    constexpr size_t lds_w1 = (size_t)2 * 9 * 32 * 32 * NT * 2;
    constexpr size_t lds_o = (SM ? 0 : (size_t)BM * (32 * NT + 4) * 4) + (EPI == SED_EPI_RELUBWD ? (size_t)3 * 32 * NT * 4 : 0);
    static_assert(lds_x + lds_w1 + lds_o <= (SM ? 80 : 160) * 1024, "LDS budget");
    const int nchunks = p.Cinp / 32;
    p.wres = (nchunks > 1 && lds_x + nchunks * lds_w1 + lds_o <= (SM ? 78 : 150) * 1024) ? 1 : 0;
    const size_t lds = lds_x + (p.wres ? nchunks : 1) * lds_w1 + lds_o;
    if constexpr (SM && !PAIR) {      // an operator that is restaged every stage: tile pairs halve that (SED_X3_PAIR=0: off, A/B runs)
        const char* e = sed_getenv("SED_X3_PAIR");
        // (the ReLU-backward epilogue at W >= 32 has no 32 registers to spare: it would spill)
        constexpr bool fits = !(EPI == SED_EPI_RELUBWD && W >= 64);   // (W = 64: would spill)
        if (fits && nchunks > 1 && !p.wres && !(e && e[0] == '0')) return launch_x3<HALF, W, NT, PRO, EPI, SM, fits>(p, st);
    }
    if (int rc_ = sed_set_max_lds<&conv_x3_kernel<HALF, W, NT, PRO, EPI, SM, PAIR>>(lds)) return rc_;
    p.tilesPerImg = cdiv(p.H, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    sed_fastdiv_make((unsigned)p.tilesPerImg, &p.tpi_M, &p.tpi_l);
    sed_fastdiv_make((unsigned)(PAIR ? 2 * nchunks : nchunks), &p.nch_M, &p.nch_l);
    // (SM: two workgroups per CU -- twice the strips, so that every CU holds two; the partial-statistics rows stay p.nparts:
    //  strip bx writes row bx, and sed_conv_nparts' count is what the caller's buffers hold, so the strip count cannot exceed it)
    p.tpb = cdiv(p.totalTiles, p.nparts);
    conv_x3_kernel<HALF, W, NT, PRO, EPI, SM, PAIR><<<dim3(p.nparts * (p.Coutp / (32 * NT))), dim3(256), lds, st>>>(p);
    return 0;
}

template <bool HALF, int W, int NT, bool SM = false>
int dispatch_x3_pe(ConvParams& p, hipStream_t st) {
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_STATS) return launch_x3<HALF, W, NT, SED_PRO_NONE, SED_EPI_STATS, SM>(p, st);
    if (p.pro == SED_PRO_BNRELU && p.epi == SED_EPI_STATS) return launch_x3<HALF, W, NT, SED_PRO_BNRELU, SED_EPI_STATS, SM>(p, st);
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_STORE) return launch_x3<HALF, W, NT, SED_PRO_NONE, SED_EPI_STORE, SM>(p, st);
    if (p.pro == SED_PRO_BNRELU && p.epi == SED_EPI_STORE) return launch_x3<HALF, W, NT, SED_PRO_BNRELU, SED_EPI_STORE, SM>(p, st);
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_RELUBWD) return launch_x3<HALF, W, NT, SED_PRO_NONE, SED_EPI_RELUBWD, SM>(p, st);
    sed_set_error("sed_conv3x3_fwd (split operands): unsupported prologue/epilogue combination");
    return 1;
}

template <bool HALF, int W>
int dispatch_x3_nt(ConvParams& p, hipStream_t st) {
    // SED_X3_FORM (A/B knob): s = the two-workgroups-per-CU small form (default), 2 = two output tiles per wave where Cout % 64 == 0,
    // 1 = one output tile per wave and 256-pixel stages
    const char* e = sed_getenv("SED_X3_FORM");
    const char form = e ? e[0] : 's';
    if (form == 's' || (form == 'S' && W <= 32)) return dispatch_x3_pe<HALF, W, 1, true>(p, st);      // (S: the small form at W <= 32 only, the round's first version)
    const bool nt2 = p.Coutp % 64 == 0 && form != '1';
    return nt2 ? dispatch_x3_pe<HALF, W, 2>(p, st) : dispatch_x3_pe<HALF, W, 1>(p, st);
}

template <bool HALF>
int dispatch_x3_w(ConvParams& p, int W, hipStream_t st) {
    switch (W) {
        case 8: return dispatch_x3_nt<HALF, 8>(p, st);
        case 16: return dispatch_x3_nt<HALF, 16>(p, st);
        case 32: return dispatch_x3_nt<HALF, 32>(p, st);
        case 64: return dispatch_x3_nt<HALF, 64>(p, st);
    }
    sed_set_error("sed_conv3x3_fwd (split operands): W must be one of 8,16,32,64");
    return 1;
}

template <bool HALF, int W, int WN, int DZ, int PRO>
int launch_wg_x3(Wgrad2Params& p, hipStream_t st) {
    constexpr int TH = 128 / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr size_t lds = ((size_t)2 * (TH + 2) * WP * 32 + (size_t)2 * WN * 128 * 32) * 2 + (size_t)5 * 32 * WN * sizeof(float);
    if (int rc_ = sed_set_max_lds<&wgrad_x3_kernel<HALF, W, WN, DZ, PRO>>(lds)) return rc_;
    p.tilesPerImg = cdiv(p.H, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    sed_fastdiv_make((unsigned)p.tilesPerImg, &p.tpi_M, &p.tpi_l);
    p.tpb = cdiv(p.totalTiles, p.strips);
    const int ny = (p.Cinp / 32) * (p.Coutp / (32 * WN));
    wgrad_x3_kernel<HALF, W, WN, DZ, PRO><<<dim3(p.strips * ny), dim3(WgX3Threads<HALF, WN>::N), lds, st>>>(p);
    return 0;
}

// (64 output channels per workgroup at most: the 128-channel form's twelve waves would share a SIMD three ways and spill; the host's
//  strip count does not depend on it -- a strip's slab is written by however many (cin tile, cout tile) workgroups serve the strip)
template <bool HALF, int DZ>
int dispatch_wg_x3(Wgrad2Params& p, int W, int wn, hipStream_t st) {
    // 32 output channels per workgroup (4 waves, two workgroups per CU at two waves per SIMD) beat 64 (6 waves: one workgroup per CU, two
    // SIMDs with a single wave) on every layer: profiles/r06_r_ab_x3_wgrad_wn.txt.  SED_X3_WGWN=2 restores the wide form (A/B runs).
    { const char* e = sed_getenv("SED_X3_WGWN"); if (!(e && atoi(e) >= 2)) wn = 1; }
#define SED_CASE(WW)                                                                                          \
    case WW:                                                                                                  \
        if (p.pro == SED_PRO_BNRELU) {                                                                        \
            if (wn >= 2) return launch_wg_x3<HALF, WW, 2, DZ, SED_PRO_BNRELU>(p, st);                         \
            return launch_wg_x3<HALF, WW, 1, DZ, SED_PRO_BNRELU>(p, st);                                      \
        }                                                                                                     \
        if (wn >= 2) return launch_wg_x3<HALF, WW, 2, DZ, SED_PRO_NONE>(p, st);                               \
        return launch_wg_x3<HALF, WW, 1, DZ, SED_PRO_NONE>(p, st);
    switch (W) {
        SED_CASE(8)
        SED_CASE(16)
        SED_CASE(32)
        SED_CASE(64)
    }
#undef SED_CASE
    sed_set_error("sed_conv3x3_wgrad (split operands): W must be one of 8,16,32,64");
    return 1;
}

template <bool HALF>
int dispatch_wg_x3_dz(int dzmode, Wgrad2Params& p, int W, int wn, hipStream_t st) {
    return dzmode == DZ_GIVEN ? dispatch_wg_x3<HALF, DZ_GIVEN>(p, W, wn, st)
           : dzmode == DZ_POOL ? dispatch_wg_x3<HALF, DZ_POOL>(p, W, wn, st)
                               : dispatch_wg_x3<HALF, DZ_BN>(p, W, wn, st);
}

}  // namespace

int launch_conv_x3(int half, ConvParams& p, int W, hipStream_t st) {
#ifdef SED_EXPERIMENTS     // (make EXPERIMENTS=1, SED_X3_CONV=p: the producer / consumer form, experiments/csrc/sed_conv_x3pc.hip -- measured 8-25 % SLOWER:
    // with equal wave counts two time-sharing workgroups need (staging + matrix) / 2 per stage, role-split waves max(staging, matrix))
    if (half) {
        const char* e = sed_getenv("SED_X3_CONV");
        const int rc = (e && e[0] == 'p') ? launch_conv_x3pc(p, W, st) : -1;
        if (rc >= 0) return rc;
    }
#endif
    return half ? dispatch_x3_w<true>(p, W, st) : dispatch_x3_w<false>(p, W, st);
}

int launch_wgrad_x3(int half, int dzmode, Wgrad2Params& p, int W, int wn, hipStream_t st) {
    return half ? dispatch_wg_x3_dz<true>(dzmode, p, W, wn, st) : dispatch_wg_x3_dz<false>(dzmode, p, W, wn, st);
}
