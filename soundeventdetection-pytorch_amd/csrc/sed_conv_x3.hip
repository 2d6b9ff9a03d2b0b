// "bf16x3": fp32 storage, split-bf16 arithmetic on the bf16 matrix pipe (dtype SED_F32X3).
//
// The reference-identity gate (frame logits within 1e-3 of the reference's fp32 CPU path, decisions / onsets bit-exact:
// /root/reference/models/spectogram_models.py:185-205, train.py:44) was only met by precision="fp32", whose convolutions run on
// v_mfma_f32_32x32x2f32 (157 TF dense peak: 39.8 ms per step).  Here every fp32 operand is split once, where it is staged,
//     a = hi(a) + lo(a),   hi = bf16(a),  lo = bf16(a - hi)            (|a - hi - lo| <= 2^-17 |a|)
// and a product runs as THREE v_mfma_f32_32x32x16_bf16 with fp32 accumulation,
//     a.b ~= lo(a).hi(b) + hi(a).lo(b) + hi(a).hi(b)                   (dropped: lo.lo and the residuals, <= 3 * 2^-18 |a.b|),
// i.e. 5.3x the matrix rate of the fp32 MFMA at ~1e-5 relative per product.  Tensors stay fp32 in HBM (the element-wise kernels,
// BatchNorm statistics, the first layer and the head are the fp32 mode's); only the two GEMM-shaped kernels differ:
//   conv_x3_kernel   forward / data gradient: conv_igemm_kernel's structure (sed_conv.hip) with the halo tile as two bf16 planes
//                    (hi, lo; padded-linear 80-byte pixels: conflict-free ds_read_b128 for all nine taps) and the packed operator as
//                    two bf16 images; BatchNorm+ReLU prologue in fp32 BEFORE the split; statistics / ReLU-backward epilogue on the fp32
//                    accumulators exactly as the fp32 kernel's
//   wgrad_x3_kernel  weight gradient with dz produced on load (DZ_POOL / DZ_BN / given): conv_wgrad2_kernel's structure with both
//                    operands as hi / lo planes read through ds_read_b64_tr_b16
// The operator images come from sed_pack_conv_weight(s_batch) with dtype SED_F32X3: [hi image][lo image], each in the bf16 layout.
#include "conv_common.h"

namespace {

// 8 fp32 values -> their bf16 hi and lo parts (v_cvt_pk_bf16_f32 rounds to nearest even; the difference a - hi is exact in fp32)
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
    sed_u32x4 hw, lw;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x2 pr = {v[2 * k], v[2 * k + 1]};
        const bf16x2 h = __builtin_convertvector(pr, bf16x2);
        const unsigned hb = __builtin_bit_cast(unsigned, h);
        const float h0 = __builtin_bit_cast(float, hb << 16), h1 = __builtin_bit_cast(float, hb & 0xffff0000u);
        const f32x2 d = {v[2 * k] - h0, v[2 * k + 1] - h1};
        const bf16x2 l = __builtin_convertvector(d, bf16x2);
        hw[k] = hb;
        lw[k] = __builtin_bit_cast(unsigned, l);
    }
    hi = __builtin_bit_cast(bf16x8, hw);
    lo = __builtin_bit_cast(bf16x8, lw);
}

__device__ __forceinline__ f32x16 mfma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);      // small terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}

// Tile-invariant plan of one thread's halo-tile items (32 input channels of an fp32 NHWC tensor) -> two bf16 LDS planes.
// PS = LDS pixel stride in bf16 elements: 40 (padded-linear, forward / data gradient) or 32 (XOR-swizzled, weight gradient).
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
    template <int PRO>
    __device__ __forceinline__ void commit(bf16_t* __restrict__ xh, bf16_t* __restrict__ xl, int tid, const float* __restrict__ pro_scale,
                                           const float* __restrict__ pro_shift, int c0, int row_lo, int row_hi) const {
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
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = keep ? fmaxf(0.f, fmaf(v[e], sc[e], sh[e])) : 0.f;
            }
            bf16x8 hi, lo;
            split8(v, hi, lo);
            *reinterpret_cast<bf16x8*>(xh + lds[u]) = hi;
            *reinterpret_cast<bf16x8*>(xl + lds[u]) = lo;
        }
    }
};

// =================================================================================================
// forward / data gradient
// =================================================================================================
template <int W, int BM, int PRO, int EPI>
__global__ __launch_bounds__(256) void conv_x3_kernel(ConvParams p) {
    constexpr int BN = 32, NTHR = 256;
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr int ROWS = TH + 2;
    constexpr int PS = 40;
    constexpr int XS = ROWS * WP * PS;       // bf16 elements per plane
    constexpr int WS = 9 * 32 * BN;          // bf16 elements per plane and 32-channel chunk
    constexpr int MT = BM / 128;
    constexpr int WITEMS = WS / 8;
    constexpr int WIPT = (WITEMS + NTHR - 1) / NTHR;
    static_assert(BM % 128 == 0 && BM % W == 0, "tile shape");
    typedef HaloPlanX3<W, ROWS, WP, NTHR, PS> XPlan;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool wres = p.wres != 0;
    const int nchunks = p.Cinp >> 5;
    bf16_t* xh = reinterpret_cast<bf16_t*>(smem);
    bf16_t* xl = xh + XS;
    bf16_t* wh = xl + XS;                                   // [wres ? nchunks : 1][WS]
    bf16_t* wl = wh + (wres ? nchunks : 1) * WS;
    float* os = reinterpret_cast<float*>(wl + (wres ? nchunks : 1) * WS);     // [BM][BN + 4]: output staging of the coalesced epilogue

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int NY = p.Coutp / BN;
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int by = logical % NY, bx = logical / NY;
    const int n0 = by * BN;
    const int H = p.H, Cinp = p.Cinp, Coutp = p.Coutp;
    const float* __restrict__ xg = reinterpret_cast<const float*>(p.x);
    const bf16_t* __restrict__ wg = reinterpret_cast<const bf16_t*>(p.wpack);
    float* __restrict__ zg = reinterpret_cast<float*>(p.z);
    const float* __restrict__ zr = reinterpret_cast<const float*>(p.zref);
    constexpr int epi = EPI;

    int prow[MT], xbase[MT], ostg[MT];
    constexpr int BNP = BN + 4;
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
    // operator chunk items: row (tap, kq) of BN*8 contiguous bf16 in LDS; source row stride Coutp*8; the lo image follows the hi image
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

    float S[16], Q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { S[i] = 0.f; Q[i] = 0.f; }

    // ---- coalesced epilogue (conv_igemm_kernel's): accumulators -> fp32 staging image -> 32-byte items, whole lines -------------
    constexpr int IPR = BN / 8;
    constexpr int FIPT = BM * IPR / NTHR;
    constexpr int FQS = NTHR / IPR;
    static_assert((BM * IPR) % NTHR == 0 && NTHR % IPR == 0, "flush geometry");
    const int fcg = tid % IPR, fq0 = tid / IPR;
    const int fl_lds0 = fq0 * BNP + fcg * 8;
    const unsigned fl_off0 = (unsigned)((fq0 * Coutp + n0 + fcg * 8) * 4);
    const unsigned fl_step = (unsigned)(FQS * Coutp * 4);
    Raw8<float> zraw[FIPT];
    float ces[8], cet[8], cem[8];
    if (epi == SED_EPI_RELUBWD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ces[e] = p.epi_scale[n0 + fcg * 8 + e];
            cet[e] = p.epi_shift[n0 + fcg * 8 + e];
            cem[e] = p.epi_mean[n0 + fcg * 8 + e];
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
            if (epi == SED_EPI_RELUBWD) {
                float z[8];
                raw_to_f(zraw[u], z);
                const bool valid = fh0 + (fq0 + u * FQS) / W < H;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gate = (valid && fmaf(z[e], ces[e], cet[e]) > 0.f) ? v[e] : 0.f;
                    v[e] = gate;
                    S[e] += gate;
                    Q[e] = fmaf(gate, z[e] - cem[e], Q[e]);
                }
            }
            buf_store8<float>(zs, fl_off0 + u * fl_step + tq, v);
        }
    };

    const int t_begin = bx * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    const int nst = (t_end > t_begin ? (t_end - t_begin) : 0) * nchunks;

    auto coords = [&](int s, int& b, int& h0, int& kc) {
        const int tl = s / nchunks;
        kc = s - tl * nchunks;
        const int tile = t_begin + tl;
        b = tile / p.tilesPerImg;
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
        xp.template commit<PRO>(xh, xl, tid, p.pro_scale, p.pro_shift, kc * 32, h0 == 0 ? 1 : 0, row_hi);
        if (with_w) {
#pragma unroll
            for (int u = 0; u < WIPT; ++u) {
                if (u == WIPT - 1 && tid + u * NTHR >= WITEMS) break;
                lds_store_raw<bf16_t>(wh + wdst[u], wrh[u]);
                lds_store_raw<bf16_t>(wl + wdst[u], wrl[u]);
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
                lds_store_raw<bf16_t>(wh + c * WS + wdst[u], wrh[u]);
                lds_store_raw<bf16_t>(wl + c * WS + wdst[u], wrl[u]);
            }
        }
    }
    const bool stage_w_each = !wres && nchunks > 1;

    f32x16 acc[MT];
    if (nst > 0) issue(0, !wres);
    for (int s = 0; s < nst; ++s) {
        int b, h0, kc;
        coords(s, b, h0, kc);
        __syncthreads();
        if (pending) { flush(); pending = false; }
        const bool need_w = stage_w_each || (!wres && s == 0);
        commit(s, need_w);
        __syncthreads();
        if (s + 1 < nst) issue(s + 1, stage_w_each);
        if (kc == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
        }
        const unsigned tq = (unsigned)(h0 * W * Coutp * 4);
        if (epi == SED_EPI_RELUBWD && kc == nchunks - 1) {
            const __amdgpu_buffer_rsrc_t rs = make_srd(zr + (size_t)b * zimg, zimg * 4);
#pragma unroll
            for (int u = 0; u < FIPT; ++u) zraw[u] = buf_load8<float>(rs, fl_off0 + u * fl_step + tq);
        }
        const bf16_t* __restrict__ whc = wh + (wres ? kc * WS : 0);
        const bf16_t* __restrict__ wlc = wl + (wres ? kc * WS : 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ti = tap / 3, tj = tap % 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int kb = ks * 16 + hh * 8;
                const int wo = ((tap * 4 + kb / 8) * BN + r) * 8;
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(whc + wo);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(wlc + wo);
                bf16x8 bh[MT], bl[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int xo = xbase[mt] + (ti * WP + tj) * PS + kb;
                    bh[mt] = *reinterpret_cast<const bf16x8*>(xh + xo);
                    bl[mt] = *reinterpret_cast<const bf16x8*>(xl + xo);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma3(ah, al, bh[mt], bl[mt], acc[mt]);
            }
        }
        if (kc != nchunks - 1) continue;

#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const bool valid = h0 + prow[mt] < H;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mt][4 * g + e];
                if (epi == SED_EPI_STATS && valid) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { S[4 * g + e] += v[e]; Q[4 * g + e] = fmaf(v[e], v[e], Q[4 * g + e]); }
                }
                store4<float>(os + ostg[mt] + 8 * g, v);
            }
        }
        fb = b; fh0 = h0; pending = true;
    }
    if (pending) {
        __syncthreads();
        flush();
    }

    if (epi == SED_EPI_STATS) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);   // [wave][quarter][stat][16]
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float sv = row16_sum(S[i]);
            const float qv = row16_sum(Q[i]);
            if ((lane & 15) == 0) {
                const int quarter = lane >> 4;
                red[((wave * 4 + quarter) * 2 + 0) * 16 + i] = sv;
                red[((wave * 4 + quarter) * 2 + 1) * 16 + i] = qv;
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int stat = tid / BN, within = tid % BN;
            const int hhh = (within >> 2) & 1;
            const int reg = (within & 3) + 4 * (within >> 3);
            float tot = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) tot += red[((wv * 4 + 2 * hhh + qq) * 2 + stat) * 16 + reg];
            p.partial[((size_t)bx * 2 + stat) * Coutp + n0 + within] = tot;
        }
    } else if (epi == SED_EPI_RELUBWD) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);   // [NTHR][16]
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = S[e]; red[tid * 16 + 8 + e] = Q[e]; }
        __syncthreads();
        if (tid < 2 * BN) {
            const int stat = tid / BN, cn = tid % BN;
            const int cg = cn >> 3, e = cn & 7;
            float tot = 0.f;
            for (int k = 0; k < FQS; ++k) tot += red[(cg + IPR * k) * 16 + stat * 8 + e];
            if (stat) tot *= p.epi_invstd[n0 + cn];
            p.partial[((size_t)bx * 2 + stat) * Coutp + n0 + cn] = tot;
        }
    }
}

// =================================================================================================
// weight gradient (conv_wgrad2_kernel's structure): dW[tap][cin][cout] = sum_pix a[pix + tap][cin] * dz[pix][cout]
// =================================================================================================
template <int W, int WN, int DZ, int PRO>
__global__ __launch_bounds__(192 * WN) void wgrad_x3_kernel(Wgrad2Params p) {
    constexpr int NTHR = 192 * WN;
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
    bf16_t* xh = reinterpret_cast<bf16_t*>(smem);
    bf16_t* xl = xh + XS;
    bf16_t* dh = xl + XS;                              // [WN][BM][32]
    bf16_t* dl = dh + WN * BM * 32;
    float* coef = reinterpret_cast<float*>(dl + WN * BM * 32);   // [5][CO]: scale, shift, ca, cb, cc

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave % 3, wn = wave / 3;
    const int r = lane & 31, hh = lane >> 5;
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

    if (DZ != DZ_GIVEN) {
        for (int i = tid; i < 5 * CO; i += NTHR) {
            const int a = i / CO, c = i - a * CO;
            const float* src = (a == 0) ? p.scale : (a == 1) ? p.shift : (a == 2) ? p.ca : (a == 3) ? p.cb : p.cc;
            float v = (src != nullptr) ? src[co0 + c] : 0.f;
            if (a == 2 && DZ == DZ_POOL) v *= inv_pool;
            coef[i] = v;
        }
    }

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

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
    (void)r;

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
        const int b = tile / p.tilesPerImg;
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
        const int b = tile / p.tilesPerImg;
        const int h0 = (tile - b * p.tilesPerImg) * TH;
        const int row_hi = (H - h0 < ROWS - 1) ? (H - h0) : (ROWS - 1);
        xp.template commit<PRO>(xh, xl, tid, p.pro_scale, p.pro_shift, ci0, h0 == 0 ? 1 : 0, row_hi);
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
            bf16x8 hi, lo;
            split8(v, hi, lo);
            *reinterpret_cast<bf16x8*>(dh + dlds[u]) = hi;
            *reinterpret_cast<bf16x8*>(dl + dlds[u]) = lo;
        }
    };

    const int t_begin = strip * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    if (t_begin < t_end) issue(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();
        commit(tile);
        __syncthreads();
        if (tile + 1 < t_end) issue(tile + 1);
#pragma unroll 2
        for (int k0 = 0; k0 < BM; k0 += 16) {
            const int ub = ((k0 / W) * WP + (k0 % W)) * 32;
            const bf16x8 bh = join_tr(ds_read_tr16_b64(dh + k0 * 32 + offB[0]), ds_read_tr16_b64(dh + k0 * 32 + offB[1]));
            const bf16x8 bl = join_tr(ds_read_tr16_b64(dl + k0 * 32 + offB[0]), ds_read_tr16_b64(dl + k0 * 32 + offB[1]));
            bf16x8 ah[3], al[3];
#pragma unroll
            for (int tj = 0; tj < 3; ++tj) {
                ah[tj] = join_tr(ds_read_tr16_b64(xh + ub + offA[tj][0]), ds_read_tr16_b64(xh + ub + offA[tj][1]));
                al[tj] = join_tr(ds_read_tr16_b64(xl + ub + offA[tj][0]), ds_read_tr16_b64(xl + ub + offA[tj][1]));
            }
#pragma unroll
            for (int tj = 0; tj < 3; ++tj) acc[tj] = mfma3(ah[tj], al[tj], bh, bl, acc[tj]);
        }
    }

    float* out = p.ws + (size_t)strip * 9 * Cinp * Coutp;
#pragma unroll
    for (int tj = 0; tj < 3; ++tj) {
        const int tap = wt * 3 + tj;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int cin = ci0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            out[((size_t)tap * Cinp + cin) * Coutp + co0 + wn * 32 + (lane & 31)] = acc[tj][i];
        }
    }
}

// ---- launchers ---------------------------------------------------------------------------------------------------------------
template <int W, int PRO, int EPI>
int launch_x3(ConvParams& p, hipStream_t st) {
    constexpr int BM = 256;
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr size_t lds_x = (size_t)2 * (TH + 2) * WP * 40 * 2;
    constexpr size_t lds_w1 = (size_t)2 * 9 * 32 * 32 * 2;
    constexpr size_t lds_o = (size_t)BM * 36 * 4;
    const int nchunks = p.Cinp / 32;
    p.wres = (nchunks > 1 && lds_x + nchunks * lds_w1 + lds_o <= 150 * 1024) ? 1 : 0;
    const size_t lds = lds_x + (p.wres ? nchunks : 1) * lds_w1 + lds_o;
    if (int rc_ = sed_set_max_lds<&conv_x3_kernel<W, BM, PRO, EPI>>(lds)) return rc_;
    p.tilesPerImg = cdiv(p.H, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    p.tpb = cdiv(p.totalTiles, p.nparts);
    conv_x3_kernel<W, BM, PRO, EPI><<<dim3(p.nparts * (p.Coutp / 32)), dim3(256), lds, st>>>(p);
    return 0;
}

template <int W>
int dispatch_x3_pe(ConvParams& p, hipStream_t st) {
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_STATS) return launch_x3<W, SED_PRO_NONE, SED_EPI_STATS>(p, st);
    if (p.pro == SED_PRO_BNRELU && p.epi == SED_EPI_STATS) return launch_x3<W, SED_PRO_BNRELU, SED_EPI_STATS>(p, st);
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_STORE) return launch_x3<W, SED_PRO_NONE, SED_EPI_STORE>(p, st);
    if (p.pro == SED_PRO_BNRELU && p.epi == SED_EPI_STORE) return launch_x3<W, SED_PRO_BNRELU, SED_EPI_STORE>(p, st);
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_RELUBWD) return launch_x3<W, SED_PRO_NONE, SED_EPI_RELUBWD>(p, st);
    sed_set_error("sed_conv3x3_fwd (bf16x3): unsupported prologue/epilogue combination");
    return 1;
}

template <int W, int WN, int DZ, int PRO>
int launch_wg_x3(Wgrad2Params& p, hipStream_t st) {
    constexpr int TH = 128 / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr size_t lds = ((size_t)2 * (TH + 2) * WP * 32 + (size_t)2 * WN * 128 * 32) * 2 + (size_t)5 * 32 * WN * sizeof(float);
    if (int rc_ = sed_set_max_lds<&wgrad_x3_kernel<W, WN, DZ, PRO>>(lds)) return rc_;
    p.tilesPerImg = cdiv(p.H, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    p.tpb = cdiv(p.totalTiles, p.strips);
    const int ny = (p.Cinp / 32) * (p.Coutp / (32 * WN));
    wgrad_x3_kernel<W, WN, DZ, PRO><<<dim3(p.strips * ny), dim3(192 * WN), lds, st>>>(p);
    return 0;
}

// (64 output channels per workgroup at most: the 128-channel form's twelve waves would share a SIMD three ways and spill; the host's
//  strip count does not depend on it -- a strip's slab is written by however many (cin tile, cout tile) workgroups serve the strip)
template <int DZ>
int dispatch_wg_x3(Wgrad2Params& p, int W, int wn, hipStream_t st) {
#define SED_CASE(WW)                                                                                          \
    case WW:                                                                                                  \
        if (p.pro == SED_PRO_BNRELU) {                                                                        \
            if (wn >= 2) return launch_wg_x3<WW, 2, DZ, SED_PRO_BNRELU>(p, st);                               \
            return launch_wg_x3<WW, 1, DZ, SED_PRO_BNRELU>(p, st);                                            \
        }                                                                                                     \
        if (wn >= 2) return launch_wg_x3<WW, 2, DZ, SED_PRO_NONE>(p, st);                                     \
        return launch_wg_x3<WW, 1, DZ, SED_PRO_NONE>(p, st);
    switch (W) {
        SED_CASE(8)
        SED_CASE(16)
        SED_CASE(32)
        SED_CASE(64)
    }
#undef SED_CASE
    sed_set_error("sed_conv3x3_wgrad (bf16x3): W must be one of 8,16,32,64");
    return 1;
}

}  // namespace

int launch_conv_x3(ConvParams& p, int W, hipStream_t st) {
    switch (W) {
        case 8: return dispatch_x3_pe<8>(p, st);
        case 16: return dispatch_x3_pe<16>(p, st);
        case 32: return dispatch_x3_pe<32>(p, st);
        case 64: return dispatch_x3_pe<64>(p, st);
    }
    sed_set_error("sed_conv3x3_fwd (bf16x3): W must be one of 8,16,32,64");
    return 1;
}

int launch_wgrad_x3(int dzmode, Wgrad2Params& p, int W, int wn, hipStream_t st) {
    return dzmode == DZ_GIVEN ? dispatch_wg_x3<DZ_GIVEN>(p, W, wn, st)
           : dzmode == DZ_POOL ? dispatch_wg_x3<DZ_POOL>(p, W, wn, st)
                               : dispatch_wg_x3<DZ_BN>(p, W, wn, st);
}
