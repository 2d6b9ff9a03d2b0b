// Weight gradient AND data gradient of a 3x3 convolution from ONE dz tile in LDS (bf16, gfx950).
//
//   dz          = BatchNorm / ReLU / avg-pool backward of the layer's output gradient, produced on load (never written to HBM)
//   dW[tap]     = sum_pixels a[pixel + tap] (x) dz[pixel]                (sed_wgrad.hip's contraction)
//   dx[pixel]   = sum_taps  dz[pixel - tap] . W[tap]^T  (+ epilogue)     (sed_conv_pc.hip's contraction on the transposed operator)
//
// autograd through ConvBlock, /root/reference/models/spectogram_models.py:132-160 (backward of :155-158 under train.py:102).
// The two-kernel form (sed_conv3x3_wgrad_fused writes dz, the data-gradient call reads it back) moves dz through HBM twice and,
// for conv2, reads z1 twice: 3 of 6.25 (conv2) / 2 of 5.75 (conv1) tensor passes of a layer's backward.  Block 1 of the main
// network sits at the HBM floor of that dataflow; here dz exists only in LDS.
//
// Structure (one 512-thread workgroup per CU, like sed_conv_pc.hip / sed_wgrad.hip):
//   * waves 4-7 PRODUCE: two stages of global loads in flight; dz = ca*g + cb*z + cc (DZ_BN) or the pool / ReLU / BN2 backward
//     (DZ_POOL) into a ROW RING of the swizzled halo image -- a workgroup walks its strip of an image top to bottom, every dz
//     row is produced ONCE (the 4-row halo tiles of the two-kernel form load 6 rows per 4); the activation tile (BN+ReLU
//     prologue on load); the whole epilogue of the tile before last (staging image -> whole-line stores, ReLU gate + BN1
//     backward sums, or the pooled-tensor statistics of the previous block);
//   * waves 0-3 CONSUME: each holds the nine 32x32 accumulators of one (cin tile, cout tile) pair of dW (transposed LDS reads of
//     the activation tile and of the SHIFTED dz image) and one (row, cin tile) unit of dx (ds_read_b128 of the same dz image,
//     operator resident in LDS): 36 + 36 MFMAs per stage;
//   * one s_barrier per stage; three ring slots, two activation tiles, two staging images.
// Output tile j of an image = rows [TH*j - 1, TH*j + TH - 1): it needs dz rows TH*j - 2 .. TH*j + TH - 1, i.e. the last two rows
// of chunk j-1 and chunk j -- available as soon as chunk j is in the ring.  A strip that starts inside an image spends one
// producer-only stage on chunk j-1; at the top of an image the two rows above come from a constant zero region.
#include "conv_common.h"

#include <stdlib.h>

namespace {

constexpr int kBfBlocks = 256;          // one workgroup per CU

__device__ __forceinline__ void bf_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int bf_xswz(int col) { return (col >> 2) & 3; }

template <int W, int CI_T, int CO_T, int DZ, int PRO, int EPI>
__global__ __launch_bounds__(512) void conv_bwd_fused_kernel(BwdFusedParams p) {
    typedef bf16_t T;
    constexpr int CI = 32 * CI_T, CO = 32 * CO_T;
    constexpr int NPAIR = CI_T * CO_T, KSPLIT = 4 / NPAIR;
    static_assert(NPAIR == 2 || NPAIR == 4, "two or four (cin tile, cout tile) pairs per workgroup");
    constexpr int TH = 4 / CI_T;                       // rows per tile: TH * CI_T = 4 data-gradient units, one per consumer wave
    constexpr int BM = TH * W, WP = (W + 2 + 3) & ~3, ROWE = WP * 32;
    constexpr int RING = 3 * TH, DZIMG = (RING + 2) * ROWE;      // ring rows + two constant zero rows, per 32-channel image
    constexpr int A1 = BM * 32, ABUF = CI_T * A1;
    constexpr int WSZ = CO_T * 36 * CI * 8;
    constexpr int BNP = CI + 8, OSZ = BM * BNP;
    constexpr int NP = 256, NTHR = 512;
    constexpr int KSW = BM / 16 / KSPLIT;              // k-steps (16 pixels) of a wave's weight-gradient share
    static_assert(W == 32 && KSW == 4, "geometry: W = 32, four k-steps per wave");
    constexpr bool RELUBWD = EPI == SED_EPI_RELUBWD, PSTATS = EPI == SED_EPI_POOLSTATS;
    // producer item geometry
    constexpr int IPP = CO / 8, DITEMS = BM * IPP, DIPT = DITEMS / NP, DQS = NP / IPP;
    static_assert(DQS == W && DIPT == TH, "a thread's dz items are the rows of one column");
    constexpr int IPX = CI / 8, XITEMS = BM * IPX, XIPT = XITEMS / NP, XQS = NP / IPX;
    static_assert(XITEMS % NP == 0 && NP % IPX == 0, "activation item geometry");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* dzr = reinterpret_cast<T*>(smem);               // [CO_T][RING + 2][WP][32]  swizzled 16-byte slots
    T* ab = dzr + CO_T * DZIMG;                        // [2][CI_T][BM][32]
    T* wsm = ab + 2 * ABUF;                            // [CO_T][9][4][CI][8]       the data-gradient operator, resident
    T* os = wsm + WSZ;                                 // [2][BM][BNP]
    float* coef = reinterpret_cast<float*>(os + 2 * OSZ);      // [5][CO]: scale, shift, ca, cb, cc
    float* pcoef = coef + 5 * CO;                      // [2][CI]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // (provably wave-uniform: the roles' row / slot arithmetic stays scalar)
    const int H = p.H;
    const int bx = (int)xcd_remap(blockIdx.x, gridDim.x), nbx = gridDim.x;
    const int psh = p.pool >> 1;
    const int Ho = H >> psh, Wo = W >> psh;
    const int NTI = p.tilesPerImg;                     // output tiles per image = ceil((H + 1) / TH)
    const int t_begin = bx * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    const int ntl = t_end > t_begin ? t_end - t_begin : 0;
    const int pre = (ntl > 0 && (t_begin % NTI) != 0) ? 1 : 0;          // producer-only first stage (chunk j-1 of the first tile)
    const int NS = ntl + pre;
    const int NI = (NS + 2 + 1) & ~1;

    // ---- one-time LDS setup ------------------------------------------------------------------------------------------
    {
        bf16x8 z8;
#pragma unroll
        for (int e = 0; e < 8; ++e) z8[e] = (bf16_t)0.f;
        for (int i = tid; i < CO_T * DZIMG / 8; i += NTHR) *reinterpret_cast<bf16x8*>(dzr + i * 8) = z8;     // padding columns, zero rows
        const T* __restrict__ wg = reinterpret_cast<const T*>(p.wpack_t);
        for (int i = tid; i < WSZ / 8; i += NTHR) *reinterpret_cast<bf16x8*>(wsm + i * 8) = *reinterpret_cast<const bf16x8*>(wg + i * 8);
        const float inv_pool = psh ? 0.25f : 1.0f;
        for (int i = tid; i < 5 * CO; i += NTHR) {
            const int a = i / CO, c = i - a * CO;
            const float* src = (a == 0) ? p.scale : (a == 1) ? p.shift : (a == 2) ? p.ca : (a == 3) ? p.cb : p.cc;
            float v = (src != nullptr) ? src[c] : 0.f;
            if (a == 2 && DZ == DZ_POOL) v *= inv_pool;
            coef[i] = v;
        }
        if (PRO == SED_PRO_BNRELU)
            for (int i = tid; i < 2 * CI; i += NTHR) pcoef[i] = (i < CI ? p.pro_scale[i] : p.pro_shift[i - CI]);
    }
    __syncthreads();

    // stage s of this workgroup: image b, chunk / tile j, and whether the consumers have a tile to compute
    auto stage_of = [&](int s, bool& live, bool& mainst, int& b, int& j) {
        live = s >= 0 && s < NS;
        const int t = t_begin + (live ? s : 0) - pre;
        if (live && pre && s == 0) { b = t_begin / NTI; j = t_begin % NTI - 1; mainst = false; }
        else { const int tt = live ? t : t_begin; b = live ? tt / NTI : 0; j = live ? tt - b * NTI : 0; mainst = live; }
    };

    float S[8], Q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { S[e] = 0.f; Q[e] = 0.f; }

    if (wave >= 4) {
        // =============================== PRODUCERS =====================================================
        const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
        const T* __restrict__ gg = reinterpret_cast<const T*>(p.gsrc);
        const T* __restrict__ zsg = reinterpret_cast<const T*>(p.zsrc);
        const T* __restrict__ zr = reinterpret_cast<const T*>(p.zref);
        T* __restrict__ dxg = reinterpret_cast<T*>(p.dx);
        const int pt = tid - 256;
        const size_t ximg_ = (size_t)H * W * CI, zimg_ = (size_t)H * W * CO, pimg_ = (size_t)Ho * Wo * CO;

        // dz items: thread = (column dq0, channel group dc8), item u = chunk row u
        const int dq0 = pt / IPP, dc8 = (pt - dq0 * IPP) * 8;
        const unsigned dvoff0 = (unsigned)((dq0 * CO + dc8) * 2);
        const int dlds0 = (dc8 >> 5) * DZIMG + (dq0 + 1) * 32 + ((((dc8 & 31) >> 3) ^ bf_xswz(dq0 + 1)) * 8);
        unsigned pvoff[DIPT];
#pragma unroll
        for (int u = 0; u < DIPT; ++u) pvoff[u] = (unsigned)((((u >> psh) * Wo + (dq0 >> psh)) * CO + dc8) * 2);
        // activation / output items: thread = (pixel xq0 + u * XQS, channel group xc8)
        const int xq0 = pt / IPX, xc8 = (pt - xq0 * IPX) * 8;
        const unsigned xvoff0 = (unsigned)((xq0 * CI + xc8) * 2);
        constexpr unsigned xvstep = (unsigned)(XQS * CI * 2);
        const int xlds0 = (xc8 >> 5) * A1 + xq0 * 32 + (xc8 & 31);
        float ces[8], cet[8], cem[8];
        if (RELUBWD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { ces[e] = p.epi_scale[xc8 + e]; cet[e] = p.epi_shift[xc8 + e]; cem[e] = p.epi_mean[xc8 + e]; }
        }

        struct RawSet { Raw8<T> x[XIPT]; Raw8<T> a[DIPT]; Raw8<T> b[DIPT]; };
        Raw8<T> zraw[XIPT];
        u32x2 craw[PSTATS ? XIPT : 1];

        // every load is issued unconditionally: a dead stage gets zero-sized descriptors (zeros, no traffic), so hipcc's vmcnt
        // bookkeeping is exact and two stages stay in flight
        auto issue = [&](RawSet& r, int s) {
            bool live, mainst; int b, j;
            stage_of(s, live, mainst, b, j);
            const size_t ximg = (live && mainst) ? ximg_ : 0, zimg = live ? zimg_ : 0, pimg = live ? pimg_ : 0;
            const __amdgpu_buffer_rsrc_t xsrd = make_srd(xg + (size_t)b * ximg, ximg * 2);
            const unsigned xt = (unsigned)((TH * j - 1) * W * CI * 2);        // wraps for the row above the image: out of range -> 0
#pragma unroll
            for (int u = 0; u < XIPT; ++u) r.x[u] = buf_load8<T>(xsrd, xvoff0 + (unsigned)u * xvstep + xt);
            const unsigned dt = (unsigned)(TH * j * W * CO * 2);
            const __amdgpu_buffer_rsrc_t zs = make_srd(zsg + (size_t)b * zimg, zimg * 2);
            if (DZ == DZ_POOL) {
                const __amdgpu_buffer_rsrc_t gs = make_srd(gg + (size_t)b * pimg, pimg * 2);
                const unsigned ptq = (unsigned)(((TH * j) >> psh) * Wo * CO * 2);
#pragma unroll
                for (int u = 0; u < DIPT; ++u) {
                    r.a[u] = buf_load8<T>(gs, pvoff[u] + ptq);
                    r.b[u] = buf_load8<T>(zs, dvoff0 + (unsigned)(u * W * CO * 2) + dt);
                }
            } else {
                const __amdgpu_buffer_rsrc_t gs = make_srd(gg + (size_t)b * zimg, zimg * 2);
#pragma unroll
                for (int u = 0; u < DIPT; ++u) {
                    r.a[u] = buf_load8<T>(gs, dvoff0 + (unsigned)(u * W * CO * 2) + dt);
                    r.b[u] = buf_load8<T>(zs, dvoff0 + (unsigned)(u * W * CO * 2) + dt);
                }
            }
        };

        auto commit = [&](const RawSet& r, int s) {
            bool live, mainst; int b, j;
            stage_of(s, live, mainst, b, j);
            if (!live) return;
            // ---- dz chunk j (image rows TH*j .. TH*j + TH - 1) -> ring slot s % 3 ---------------------------------------
            T* __restrict__ dst = dzr + dlds0 + (s % 3) * TH * ROWE;
            const int rows_in = H - TH * j;                 // rows of the chunk inside the image (pool floor: g = 0 by the range check)
            const f32x4* cf = reinterpret_cast<const f32x4*>(coef);
            constexpr int C4 = CO / 4;
#pragma unroll
            for (int u = 0; u < DIPT; ++u) {
                float g[8], z[8], v[8];
                raw_to_f(r.a[u], g);
                raw_to_f(r.b[u], z);
#pragma unroll
                for (int e4 = 0; e4 < 2; ++e4) {
                    const int ci4 = (dc8 >> 2) + e4;
                    const f32x4 a4 = cf[2 * C4 + ci4], b4 = cf[3 * C4 + ci4], c4 = cf[4 * C4 + ci4];
                    f32x4 s4, t4;
                    if (DZ == DZ_POOL) { s4 = cf[ci4]; t4 = cf[C4 + ci4]; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = e4 * 4 + e;
                        const float base = fmaf(b4[e], z[i], c4[e]);
                        const float full = fmaf(a4[e], g[i], base);
                        if (DZ == DZ_POOL) v[i] = (fmaf(z[i], s4[e], t4[e]) > 0.f) ? full : base;
                        else v[i] = full;
                    }
                }
                if (rows_in < TH) {                          // (uniform: only the last chunks of an image)
                    const float m = (u < rows_in) ? 1.f : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= m;
                }
                store8<T>(dst + u * ROWE, v);
            }
            // ---- activation tile j (image rows TH*j - 1 .. TH*j + TH - 2) -> buffer s & 1 -----------------------------------
            if (!mainst) return;
            T* __restrict__ adst = ab + (s & 1) * ABUF + xlds0;
            if (PRO == SED_PRO_NONE) {
#pragma unroll
                for (int u = 0; u < XIPT; ++u) lds_store_raw<T>(adst + u * XQS * 32, r.x[u]);      // hardware zeros outside the image
            } else {
                const int r0 = TH * j - 1;
                const bool boundary = r0 < 0 || r0 + TH > H;
                const f32x4* pc = reinterpret_cast<const f32x4*>(pcoef);
                const int c4 = xc8 >> 2;
                const f32x4 s0 = pc[c4], s1 = pc[c4 + 1], h0v = pc[(CI >> 2) + c4], h1v = pc[(CI >> 2) + c4 + 1];
#pragma unroll
                for (int u = 0; u < XIPT; ++u) {
                    float v[8];
                    raw_to_f(r.x[u], v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = fmaxf(0.f, fmaf(v[e], s0[e], h0v[e]));
                        v[4 + e] = fmaxf(0.f, fmaf(v[4 + e], s1[e], h1v[e]));
                    }
                    if (boundary) {                          // rows outside the image stay zero (relu(shift) is not)
                        const int row = r0 + (xq0 + u * XQS) / W;
                        const float m = (row >= 0 && row < H) ? 1.f : 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= m;
                    }
                    store8<T>(adst + u * XQS * 32, v);
                }
            }
        };

        // reference tile of the flush of the NEXT iteration (output tile of stage s - 1)
        auto issue_z = [&](int s) {
            if (!RELUBWD && !PSTATS) return;
            bool live, mainst; int b, j;
            stage_of(s - 1, live, mainst, b, j);
            const size_t rimg = (live && mainst) ? ximg_ : 0;
            const __amdgpu_buffer_rsrc_t rs = make_srd(zr + (size_t)b * rimg, rimg * 2);
            const unsigned tq = (unsigned)((TH * j - 1) * W * CI * 2);
#pragma unroll
            for (int u = 0; u < XIPT; ++u) zraw[u] = buf_load8<T>(rs, xvoff0 + (unsigned)u * xvstep + tq);
            if constexpr (PSTATS) {
                const __amdgpu_buffer_rsrc_t cs = make_srd(p.cnt + (size_t)b * rimg, rimg);
#pragma unroll
                for (int u = 0; u < XIPT; ++u)
                    craw[u] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(cs, (xvoff0 + (unsigned)u * xvstep + tq) >> 1, 0, 0));
            }
        };
        // the output tile of stage s - 2 sits complete in its staging image
        auto flush = [&](int s) {
            bool live, mainst; int b, j;
            stage_of(s - 2, live, mainst, b, j);
            if (!live || !mainst) return;
            const T* osb = os + ((s - 2) & 1) * OSZ;
            const __amdgpu_buffer_rsrc_t ds = make_srd(dxg + (size_t)b * ximg_, ximg_ * 2);
            const unsigned tq = (unsigned)((TH * j - 1) * W * CI * 2);
            const int r0 = TH * j - 1;
#pragma unroll
            for (int u = 0; u < XIPT; ++u) {
                const int q = xq0 + u * XQS;
                const bf16x8 raw = *reinterpret_cast<const bf16x8*>(osb + q * BNP + xc8);
                const int row = r0 + q / W;
                const bool valid = row >= 0 && row < H;
                const unsigned off = valid ? xvoff0 + (unsigned)u * xvstep + tq : SED_OOB;
                if (RELUBWD) {
                    float v[8], z[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (float)raw[e];
                    raw_to_f(zraw[u], z);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gate = (valid && fmaf(z[e], ces[e], cet[e]) > 0.f) ? v[e] : 0.f;
                        v[e] = gate;
                        S[e] += gate;
                        Q[e] = fmaf(gate, z[e] - cem[e], Q[e]);
                    }
                    buf_store8<T>(ds, off, v);
                } else {
                    if constexpr (PSTATS) {       // S = sum dy*cnt, Q = sum dy*y_pooled (reference loads of rows outside the image: 0)
                        float ya[8];
                        raw_to_f(zraw[u], ya);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float dyv = (float)raw[e];
                            const float cf = (float)((craw[u][e >> 2] >> (8 * (e & 3))) & 0xffu);
                            S[e] = fmaf(dyv, cf, S[e]);
                            Q[e] = fmaf(dyv, ya[e], Q[e]);
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, raw), ds, off, 0, 0);
                }
            }
        };

        RawSet ra, rb;
        issue(ra, 0);
        issue(rb, 1);
        auto iter = [&](int s, RawSet& r) {
            commit(r, s);
            flush(s);
            issue_z(s);
            issue(r, s + 2);
            bf_barrier();
        };
        for (int s = 0; s < NI; s += 2) {
            iter(s, ra);
            iter(s + 1, rb);
        }
        bf_barrier();                                       // (the consumers' slab reduction reuses the LDS from here on)
        if (KSPLIT == 2) bf_barrier();
    } else {
        // =============================== CONSUMERS =====================================================
        const int r = lane & 31, hh = lane >> 5;
        f32x16 accw[9];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) accw[t][i] = 0.f;
        // ---- data-gradient unit: output row drow of the tile, cin tile dcit --------------------------------------------------
        const int drow = wave % TH, dcit = wave / TH;
        int xoff[3][2];                                     // [tj][ks]: lane part of the dz fragment address (halo column r + tj)
#pragma unroll
        for (int tj = 0; tj < 3; ++tj)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xoff[tj][ks] = (r + tj) * 32 + (((ks * 2 + hh) ^ bf_xswz(r + tj)) * 8);
        const int woff = (hh * CI + dcit * 32 + r) * 8;
        const int ostg = (drow * W + r) * BNP + 4 * hh + dcit * 32;
        // ---- weight-gradient unit: pair (wcit, wcot), k share kw ---------------------------------------------------------------
        const int pw = wave % NPAIR, wcit = pw / CO_T, wcot = pw % CO_T, kw = wave / NPAIR;
        int offA[2], offB[3][2];
        {
            const int i16 = lane & 15, gbit = (lane >> 4) & 1;
            const int qq = i16 >> 2, pp = i16 & 3, ch = 16 * gbit + 4 * pp;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int kl = 8 * hh + qq + 4 * half;
                offA[half] = wcit * A1 + kl * 32 + ch;
#pragma unroll
                for (int sj = 0; sj < 3; ++sj) offB[sj][half] = wcot * DZIMG + (kl + sj) * 32 + (ch ^ swz<T>(kl + sj));
            }
        }

        auto citer = [&](int s) {
            bf_barrier();
            bool live, mainst; int b, j;
            stage_of(s, live, mainst, b, j);
            if (!live || !mainst) return;
            // halo row hr of the stage's window (image row TH*j - 2 + hr): rows 0, 1 = the previous chunk's last two rows (or the
            // constant zero rows at the top of an image), rows 2 .. TH+1 = this chunk
            const int cur = (s % 3) * TH * ROWE;
            const int prv = (j == 0) ? (RING - (TH - 2)) * ROWE : ((s + 2) % 3) * TH * ROWE;
            auto rowbase = [&](int hr) -> int { return hr < 2 ? prv + (TH - 2 + hr) * ROWE : cur + (hr - 2) * ROWE; };
            const T* __restrict__ abuf = ab + (s & 1) * ABUF;

            // ---- weight gradient: accw[si*3+sj] += a[k-step] (x) dz[k-step shifted by (si, sj)] -----------------------------
            {
                constexpr int NSTEP = KSW * 3;           // step = (k-step, shift row): 3 MFMAs
                bf16x8 bfr[3][3], afr[2];
                auto ld_a = [&](int kk, bf16x8& dst) {
                    const int k0 = (kw * KSW + kk) * 16;
                    dst = join_tr(ds_read_tr16_b64(abuf + k0 * 32 + offA[0]), ds_read_tr16_b64(abuf + k0 * 32 + offA[1]));
                };
                auto ld_b = [&](int st, bf16x8 (&dst)[3]) {
                    const int kk = st / 3, si = st % 3;
                    const int k0 = (kw * KSW + kk) * 16;
                    int rb = rowbase(k0 / W + si) + (k0 % W) * 32;
                    asm volatile("" : "+s"(rb));           // per-use address arithmetic: hoisted, the 36 (row, lane part) sums spill
                    const T* rowp = dzr + rb;
#pragma unroll
                    for (int sj = 0; sj < 3; ++sj)
                        dst[sj] = join_tr(ds_read_tr16_b64(rowp + offB[sj][0]), ds_read_tr16_b64(rowp + offB[sj][1]));
                };
                ld_a(0, afr[0]);
                ld_b(0, bfr[0]);
                ld_b(1, bfr[1]);
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) {
                    if (st + 2 < NSTEP) ld_b(st + 2, bfr[(st + 2) % 3]);
                    if (st % 3 == 0 && st / 3 + 1 < KSW) ld_a(st / 3 + 1, afr[(st / 3 + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int sj = 0; sj < 3; ++sj)
                        accw[(st % 3) * 3 + sj] = mfma(afr[(st / 3) & 1], bfr[st % 3][sj], accw[(st % 3) * 3 + sj]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- data gradient: D[cin][pixel] over (cout chunk, tap, 16-channel half) -----------------------------------------
            f32x16 accd;
#pragma unroll
            for (int i = 0; i < 16; ++i) accd[i] = 0.f;
            {
                constexpr int NK = CO_T * 18;
                bf16x8 xf[3], wf[3];
                auto ld = [&](int k, bf16x8& xd, bf16x8& wd) {
                    const int c = k / 18, kk = k % 18, tap = kk >> 1, ks = kk & 1, ti = tap / 3, tj = tap % 3;
                    int rb = c * DZIMG + rowbase(drow + ti);
                    asm volatile("" : "+s"(rb));
                    xd = *reinterpret_cast<const bf16x8*>(dzr + rb + xoff[tj][ks]);
                    wd = *reinterpret_cast<const bf16x8*>(wsm + woff + ((c * 36 + tap * 4 + ks * 2) * CI) * 8);
                };
                ld(0, xf[0], wf[0]);
                ld(1, xf[1], wf[1]);
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    if (k + 2 < NK) ld(k + 2, xf[(k + 2) % 3], wf[(k + 2) % 3]);
                    __builtin_amdgcn_sched_barrier(0);
                    accd = mfma(wf[k % 3], xf[k % 3], accd);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            T* osb = os + (s & 1) * OSZ;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = accd[4 * g + e];
                store4<T>(osb + ostg + 8 * g, v);
            }
        };
        for (int s = 0; s < NI; s += 2) {
            citer(s);
            citer(s + 1);
        }
        // ---- weight-gradient slab of this workgroup: the k shares of a pair are summed through LDS in a fixed order ------------
        bf_barrier();                                       // (the producers join below: nothing reads the stage buffers any more)
        float* red = reinterpret_cast<float*>(smem);       // [NPAIR][9][16][64]
        if (KSPLIT == 2) {
            if (kw == 1) {
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) red[((pw * 9 + t) * 16 + i) * 64 + lane] = accw[t][i];
            }
            bf_barrier();
        }
        if (kw == 0) {
            float* out = p.ws + (size_t)bx * 9 * CI * CO;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int tap = (2 - t / 3) * 3 + (2 - t % 3);         // shift (si, sj) = (2 - ti, 2 - tj)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = accw[t][i];
                    if (KSPLIT == 2) v += red[((pw * 9 + t) * 16 + i) * 64 + lane];
                    const int cin = wcit * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    out[((size_t)tap * CI + cin) * CO + wcot * 32 + r] = v;
                }
            }
        }
    }

    // ---- statistics partial of this workgroup (fixed-order sum over the producer threads of a channel group) --------------------
    if (RELUBWD || PSTATS) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);       // [NP][16]
        if (wave >= 4) {
            const int pt = tid - 256;
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[pt * 16 + e] = S[e]; red[pt * 16 + 8 + e] = Q[e]; }
        }
        __syncthreads();
        if (tid < 2 * CI) {
            const int stat = tid / CI, cn = tid % CI;
            const int cg = cn >> 3, e = cn & 7;
            float tot = 0.f;
            for (int k = 0; k < XQS; ++k) tot += red[(cg + IPX * k) * 16 + stat * 8 + e];
            if (RELUBWD && stat) tot *= p.epi_invstd[cn];
            if constexpr (PSTATS) {       // (as sed_conv_pc.hip: sum g = sum dy*cnt / 4, sum g*xhat = (sum dy*y - beta/4 sum dy*cnt) / gamma)
                float sraw = tot;
                if (stat) {
                    sraw = 0.f;
                    for (int k = 0; k < XQS; ++k) sraw += red[(cg + IPX * k) * 16 + e];
                }
                const float sc = p.epi_scale[cn], is = p.epi_invstd[cn];
                const float beta = fmaf(p.epi_mean[cn], sc, p.epi_shift[cn]);
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
            p.partial[((size_t)bx * 2 + stat) * CI + cn] = tot;
            for (int row = bx + nbx; row < p.nparts; row += nbx) p.partial[((size_t)row * 2 + stat) * CI + cn] = 0.f;
        }
    }
}

template <int W, int CI_T, int CO_T, int DZ, int PRO, int EPI>
int launch_bf(BwdFusedParams& p, hipStream_t st) {
    constexpr int CI = 32 * CI_T, CO = 32 * CO_T, TH = 4 / CI_T, BM = TH * W, WP = (W + 2 + 3) & ~3;
    constexpr size_t lds = ((size_t)CO_T * (3 * TH + 2) * WP * 32 + (size_t)2 * CI_T * BM * 32 + (size_t)CO_T * 36 * CI * 8 +
                            (size_t)2 * BM * (CI + 8)) * sizeof(bf16_t) + (size_t)(5 * CO + 2 * CI) * sizeof(float);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert(CI_T * CO_T == 4 || lds >= (size_t)CI_T * CO_T * 9 * 16 * 64 * 4, "the k-share reduction at the end reuses the LDS");
    if (p.dry) return 0;
    if (int rc_ = sed_set_max_lds<&conv_bwd_fused_kernel<W, CI_T, CO_T, DZ, PRO, EPI>>(lds)) return rc_;
    p.tilesPerImg = cdiv(p.H + 1, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    conv_bwd_fused_kernel<W, CI_T, CO_T, DZ, PRO, EPI><<<dim3(p.nwg), dim3(512), lds, st>>>(p);
    return 0;
}

}  // namespace

// workgroups the fused backward kernel launches for this problem (= slabs of its weight-gradient workspace); 0 = shape not covered
int bwd_fused_nwg(int B, int H, int W, int Cinp, int Coutp, int dzmode, int pro, int epi) {
    if (W != 32) return 0;
    const bool c1 = Cinp == 32 && Coutp == 64 && dzmode == DZ_BN && pro == SED_PRO_NONE && (epi == SED_EPI_POOLSTATS || epi == SED_EPI_STORE);
    const bool c2 = Cinp == 64 && Coutp == 64 && dzmode == DZ_POOL && pro == SED_PRO_BNRELU && epi == SED_EPI_RELUBWD;
    if (!c1 && !c2) return 0;
    if (const char* e = sed_getenv("SED_BWD_FUSED")) if (e[0] == '0') return 0;
    const int TH = Cinp == 32 ? 4 : 2;
    const long long tiles = (long long)B * cdiv(H + 1, TH);
    long long n = kBfBlocks;
    if (const char* e = sed_getenv("SED_BWD_FUSED_BLOCKS")) n = atoll(e) > 0 ? atoll(e) : n;      // tuning knob
    if (n > tiles) n = tiles;
    return (int)(n < 1 ? 1 : n);
}

int launch_bwd_fused(BwdFusedParams& p, int W, hipStream_t st) {
    p.nwg = bwd_fused_nwg(p.B, p.H, W, p.Cinp, p.Coutp, p.dzmode, p.pro, p.epi);
    if (p.nwg == 0) return -1;
    if (p.epi != SED_EPI_STORE && p.nwg > p.nparts) p.nwg = p.nparts;
    const int TH = p.Cinp == 32 ? 4 : 2;
    p.tpb = cdiv((long long)p.B * cdiv(p.H + 1, TH), p.nwg);
    if (p.Cinp == 32) {
        if (p.epi == SED_EPI_POOLSTATS) return launch_bf<32, 1, 2, DZ_BN, SED_PRO_NONE, SED_EPI_POOLSTATS>(p, st);
        return launch_bf<32, 1, 2, DZ_BN, SED_PRO_NONE, SED_EPI_STORE>(p, st);
    }
    return launch_bf<32, 2, 2, DZ_POOL, SED_PRO_BNRELU, SED_EPI_RELUBWD>(p, st);
}
