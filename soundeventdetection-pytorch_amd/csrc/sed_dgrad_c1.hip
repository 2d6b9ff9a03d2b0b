// Block 0, C1 mode: conv2's data gradient FUSED with everything that consumes it (bf16, gfx950, W = 64, 32 -> 32).
//
// Reference: autograd through ConvBlock, /root/reference/models/spectogram_models.py:132-156 (train.py:102):
//   g  = relu'(bn1(z1)) * conv2^T(dz2)                       (gradient at BN1's output)
//   BN1 backward needs  sum g  and  sum g*z1;  conv1's weight gradient needs  sum_px dz1 (x) patch(x).
// conv1 has ONE input channel, so with A[tap][c] = sum_px g[px][c] * xz[px + tap] (the plain first-layer weight gradient
// of g):  sum g*z1 = w1 . A  and  dW1 = ca*A + cb*(w1.G) + cc*sx  (sed_bn_bwd_finalize_c1, sed_conv3x3_c1_wgrad_combine).
// g itself is needed by NOTHING else (block 0 has no data gradient), so this kernel never writes it: the consumer waves
// gate their accumulators with conv1's ReLU bit mask and contract them over the pixels on the matrix pipe,
//   A^T[tap][c] += P[tap][px] * g[px][c]     (tap 9 = all ones -> sum g),
// straight from the accumulator registers.  The main MFMAs run with the operands swapped (D[pixel][channel]): a lane then
// holds 16 PIXELS of one channel, which is exactly the B-operand layout of the second contraction (K = pixels) -- no
// transpose, no LDS staging image, no output store.  Replaces sed_conv3x3_dgrad_c1 + sed_conv3x3_c1_wgrad (786 MB written
// and read back at B = 32) and leaves one streaming read of dz2.
//
// Structure as sed_conv_pc.hip: one 512-thread workgroup per CU; waves 4-7 load (two stages in flight, dead stages get
// zero-sized descriptors), build the swizzled LDS halo image of dz2, the z-scored fp32 input tile and the mask tile;
// waves 0-3 run the MFMAs; one s_barrier per tile.  TH output rows per tile (8: 1.25x halo re-reads instead of 1.5x).
#include "conv_common.h"

#include <stdlib.h>

namespace {

struct DgradC1Params {
    const void* dz;          // [B][H][64][32] bf16
    const void* wpack_t;     // conv2 weights packed for the data gradient: [36][32][8] bf16
    const float* x1;         // [B][H][64] fp32
    const float* fmean;      // [64] or NULL
    const float* fstd;
    const unsigned* mask;    // [B][H][64] one word per pixel: conv1's ReLU decisions
    float* out;              // [nparts][10][32]
    void* g_dbg;             // test hook (NULL in production): the gated data gradient g [B][H][64][32] bf16
    int B, H;
    int tilesPerImg, totalTiles, tpb, nparts;
};

__device__ __forceinline__ void wg_barrier2() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int xswz2(int col) { return (col >> 2) & 3; }

// GDBG: the test-hook instantiation that also writes g (sed_conv3x3_dgrad_c1_stats_g); the product instantiation carries no
// trace of it in its loop
template <int TH, bool GDBG = false>
__global__ __launch_bounds__(512) void dgrad_c1a_kernel(DgradC1Params p) {
    typedef bf16_t T;
    constexpr int W = 64, ROWS = TH + 2, WP = 68, XS = ROWS * WP * 32, WS = 9 * 32 * 32, BM = TH * W, MT = TH / 2, NP = 256;
    constexpr int XTW = W + 2, XTN = ROWS * XTW, XTIPT = (XTN + NP - 1) / NP, MIPT = BM / NP;
    static_assert(TH % 2 == 0 && BM % NP == 0, "geometry");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs0 = reinterpret_cast<T*>(smem);                       // [2][XS]   dz2 halo image (16-byte slots XOR-swizzled)
    T* ws = xs0 + 2 * XS;                                      // [WS]      resident weights
    float* xt0 = reinterpret_cast<float*>(ws + WS);            // [2][XTN]  z-scored input tile: row 0 = image row h0-1, col 0 = image col -1
    unsigned* mk0 = reinterpret_cast<unsigned*>(xt0 + 2 * XTN);   // [2][BM]   mask words of the tile's pixels
    float* cst0 = reinterpret_cast<float*>(mk0 + 2 * BM);         // [2][XTN]  constant "input tiles": all ones (tap 9 -> sum g), all zeros (taps 10..31)

    const int H = p.H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bx = (int)xcd_remap(blockIdx.x, gridDim.x), nbx = gridDim.x;
    const int t_begin = bx * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    const int nst = t_end > t_begin ? t_end - t_begin : 0;
    const int NI = (nst + 1) & ~1;

    {   // padding columns of both stages (zero for the whole kernel), resident weights
        constexpr int NPAD = 2 * ROWS * 2 * 4;
        bf16x8 z8;
#pragma unroll
        for (int e = 0; e < 8; ++e) z8[e] = (bf16_t)0.f;
        for (int i = tid; i < NPAD; i += 512) {
            const int c16 = i & 3, side = (i >> 2) & 1, rowi = (i >> 3) % ROWS, sg = (i >> 3) / ROWS;
            *reinterpret_cast<bf16x8*>(xs0 + sg * XS + (rowi * WP + (side ? W + 1 : 0)) * 32 + c16 * 8) = z8;
        }
        for (int i = tid; i < 2 * XTN; i += 512) cst0[i] = i < XTN ? 1.0f : 0.0f;
        const T* __restrict__ wg = reinterpret_cast<const T*>(p.wpack_t);
        for (int i = tid; i < WS / 8; i += 512)
            *reinterpret_cast<bf16x8*>(ws + i * 8) = *reinterpret_cast<const bf16x8*>(wg + i * 8);
    }
    __syncthreads();

    f32x16 accA;
#pragma unroll
    for (int i = 0; i < 16; ++i) accA[i] = 0.f;

    if (wave >= 4) {
        // =============================== PRODUCERS =====================================================
        const T* __restrict__ xg = reinterpret_cast<const T*>(p.dz);
        const int pt = tid - 256, cq = pt & 3, xcol = (pt >> 2) + 1;       // item u = halo row u of LDS column xcol
        const size_t ximg_ = (size_t)H * W * 32, x1img_ = (size_t)H * W;
        const unsigned xvoff0 = (unsigned)(((xcol - 1) * 32 + cq * 8) * 2);
        constexpr unsigned xvstep = (unsigned)(W * 32 * 2);
        const int xlds0 = xcol * 32 + ((cq ^ xswz2(xcol)) * 8);

        float xtmu[XTIPT], xtis[XTIPT];
#pragma unroll
        for (int u = 0; u < XTIPT; ++u) {
            const int e = pt + u * NP, c = (e % XTW) - 1;
            const bool ok = e < XTN && c >= 0 && c < W;
            xtmu[u] = (ok && p.fmean) ? p.fmean[c] : 0.f;
            xtis[u] = ok ? (p.fstd ? 1.0f / p.fstd[c] : 1.0f) : 0.f;
        }
        struct Set { Raw8<T> x[ROWS]; float xr[XTIPT]; unsigned m[MIPT]; };

        auto stage_of = [&](int j, bool& live, int& b, int& h0) {
            live = j >= 0 && j < nst;
            const int tile = t_begin + (live ? j : 0);
            b = live ? tile / p.tilesPerImg : 0;
            h0 = live ? (tile - b * p.tilesPerImg) * TH : 0;
        };
        // every load is issued unconditionally; a dead stage gets zero-sized descriptors (zeros, no traffic)
        auto issue = [&](Set& r, int j) {
            bool live; int b, h0;
            stage_of(j, live, b, h0);
            const size_t ximg = live ? ximg_ : 0, x1img = live ? x1img_ : 0;
            const __amdgpu_buffer_rsrc_t xsrd = make_srd(xg + (size_t)b * ximg, ximg * 2);
            const unsigned xt = (unsigned)((h0 - 1) * W * 32 * 2);         // wraps for the row above the image: out of range -> 0
#pragma unroll
            for (int u = 0; u < ROWS; ++u) r.x[u] = buf_load8<T>(xsrd, xvoff0 + (unsigned)u * xvstep + xt);
            const __amdgpu_buffer_rsrc_t s1 = make_srd(p.x1 + (size_t)b * x1img, x1img * 4);
#pragma unroll
            for (int u = 0; u < XTIPT; ++u) {
                const int e = pt + u * NP, rr = e / XTW, c = e - rr * XTW - 1;
                const bool ok = e < XTN && c >= 0 && c < W;
                r.xr[u] = buf_load_f32(s1, ok ? (unsigned)(((h0 - 1 + rr) * W + c) * 4) : SED_OOB);
            }
            const __amdgpu_buffer_rsrc_t sm = make_srd(p.mask + (size_t)b * x1img, x1img * 4);
#pragma unroll
            for (int u = 0; u < MIPT; ++u)       // rows past the image: out of range -> 0 -> every channel gated off
                r.m[u] = __builtin_amdgcn_raw_buffer_load_b32(sm, (unsigned)((h0 * W + pt + u * NP) * 4), 0, 0);
        };
        auto commit = [&](const Set& r, int j) {
            if (j >= nst) return;
            bool live; int b, h0;
            stage_of(j, live, b, h0);
            T* xsb = xs0 + (j & 1) * XS;
#pragma unroll
            for (int u = 0; u < ROWS; ++u) lds_store_raw<T>(xsb + xlds0 + u * WP * 32, r.x[u]);
            float* xtb = xt0 + (j & 1) * XTN;
#pragma unroll
            for (int u = 0; u < XTIPT; ++u) {
                const int e = pt + u * NP;
                if (u == XTIPT - 1 && e >= XTN) break;
                const int hy = h0 - 1 + e / XTW;                         // rows outside the image are zero AFTER the z-score
                xtb[e] = (hy >= 0 && hy < H) ? (r.xr[u] - xtmu[u]) * xtis[u] : 0.f;
            }
            unsigned* mkb = mk0 + (j & 1) * BM;
#pragma unroll
            for (int u = 0; u < MIPT; ++u) mkb[pt + u * NP] = r.m[u];
        };

        Set ra, rb;
        issue(ra, 0);
        issue(rb, 1);
        for (int j = 0; j < NI; j += 2) {
            commit(ra, j);
            issue(ra, j + 2);
            wg_barrier2();
            commit(rb, j + 1);
            issue(rb, j + 3);
            wg_barrier2();
        }
    } else {
        // =============================== CONSUMERS =====================================================
        const int r = lane & 31, hh = lane >> 5;
        int xoff[MT][3][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int q = (wave * MT + mt) * 32 + r;
            const int prow = q / W, pcol = q % W;
#pragma unroll
            for (int tj = 0; tj < 3; ++tj)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    xoff[mt][tj][ks] = (prow * WP + pcol) * 32 + (((ks * 2 + hh) ^ xswz2(pcol + tj)) * 8);
        }
        const int woff = (hh * 32 + r) * 8;
        // second contraction: this lane is channel r of g (B operand) and tap r of the patch matrix (A operand)
        const unsigned bitpos = 16 * ((r >> 2) & 1) + (r & 3) + 4 * (r >> 3);    // mask bit of channel r
        const int tap = r < 9 ? r : 0;
        const int ptap = (tap / 3) * XTW + (tap % 3) + 4 * hh;
        // lanes 9 (sum g) and 10..31 (unused rows of A^T) read their patch fragment from a constant tile instead of masking it
        const float* cbase = cst0 + (r == 9 ? 0 : XTN);

        auto citer = [&](int j) {
            wg_barrier2();
            if (j >= nst) return;
            const T* __restrict__ xsb = xs0 + (j & 1) * XS;
            const float* __restrict__ xtb = xt0 + (j & 1) * XTN;
            const unsigned* __restrict__ mkb = mk0 + (j & 1) * BM;
            f32x16 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
            {
                bf16x8 xf[3][MT], wf[3];
                auto ld = [&](int k, bf16x8 (&xd)[MT], bf16x8& wd) {
                    const int tp = k >> 1, ks = k & 1, ti = tp / 3, tj = tp % 3;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        xd[mt] = *reinterpret_cast<const bf16x8*>(xsb + xoff[mt][tj][ks] + (ti * WP + tj) * 32);
                    wd = *reinterpret_cast<const bf16x8*>(ws + woff + ((tp * 4 + ks * 2) * 32) * 8);
                };
                ld(0, xf[0], wf[0]);
                ld(1, xf[1], wf[1]);
#pragma unroll
                for (int k = 0; k < 18; ++k) {
                    if (k + 2 < 18) ld(k + 2, xf[(k + 2) % 3], wf[(k + 2) % 3]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma(xf[k % 3][mt], wf[k % 3], acc[mt]);     // D[pixel][channel]
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // gate with conv1's ReLU decisions, contract over the block's 32 pixels
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int blk = wave * MT + mt, prow = blk >> 1, half = blk & 1;
                const unsigned* mrow = mkb + prow * W + half * 32 + 4 * hh;
                const float* prow_x = (r < 9 ? xtb : cbase) + prow * XTW + half * 32 + ptap;
                unsigned gv[16];
#pragma unroll
                for (int i4 = 0; i4 < 4; ++i4) {
                    const u32x4 m4 = *reinterpret_cast<const u32x4*>(mrow + 8 * i4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int t = __builtin_amdgcn_sbfe((int)m4[e], bitpos, 1u);      // 0 / -1 in one v_bfe_i32
                        const float av = acc[mt][4 * i4 + e];      // (a copy: __builtin_bit_cast straight on the vector-element lvalue reads element 0)
                        gv[4 * i4 + e] = __builtin_bit_cast(unsigned, av) & (unsigned)t;
                    }
                }
                if constexpr (GDBG) {
                    const int tile = t_begin + j, b = tile / p.tilesPerImg, h0 = (tile - b * p.tilesPerImg) * TH;
                    if (h0 + prow < H) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int pm = (i & 3) + 8 * (i >> 2) + 4 * hh;
                            reinterpret_cast<T*>(p.g_dbg)[((((size_t)b * H + h0 + prow) * W) + half * 32 + pm) * 32 + r] =
                                (bf16_t)__builtin_bit_cast(float, gv[i]);
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    bf16x8 gf, pf;
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        gf[jj] = (bf16_t)__builtin_bit_cast(float, gv[8 * s + jj]);
                        pf[jj] = (bf16_t)prow_x[16 * s + 8 * (jj >> 2) + (jj & 3)];
                    }
                    accA = mfma(pf, gf, accA);         // A^T[tap][channel]
                }
            }
        };
        for (int j = 0; j < NI; j += 2) {
            citer(j);
            citer(j + 1);
        }
    }

    // ---- per-workgroup partial: fixed-order sum over the four consumer waves; rows beyond the launched workgroups are zeroed
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);       // [4][16][64]
    if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = accA[i];
    }
    __syncthreads();
    if (tid < 320) {
        const int k = tid >> 5, c = tid & 31;          // tap k (9 = sum g), channel c: register 4*(k>>3) + (k&3) of lane c + 32*((k>>2)&1)
        const int i = 4 * (k >> 3) + (k & 3), ln = c + 32 * ((k >> 2) & 1);
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) tot += red[(w * 16 + i) * 64 + ln];
        p.out[((size_t)bx * 10 + k) * 32 + c] = tot;
        for (int row = bx + nbx; row < p.nparts; row += nbx) p.out[((size_t)row * 10 + k) * 32 + c] = 0.f;
    }
}

template <int TH, bool GDBG = false>
int launch_dgrad_c1a(DgradC1Params& p, hipStream_t st) {
    constexpr int ROWS = TH + 2;
    const size_t lds = ((size_t)2 * ROWS * 68 * 32 + 9 * 32 * 32) * sizeof(bf16_t) + (size_t)4 * ROWS * 66 * sizeof(float) +
                       (size_t)2 * TH * 64 * sizeof(unsigned);
    if (int rc_ = sed_set_max_lds<&dgrad_c1a_kernel<TH, GDBG>>(lds)) return rc_;
    p.tilesPerImg = cdiv(p.H, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    int nbx = p.nparts < p.totalTiles ? p.nparts : p.totalTiles;
    if (nbx < 1) nbx = 1;
    p.tpb = cdiv(p.totalTiles, nbx);
    dgrad_c1a_kernel<TH, GDBG><<<dim3(nbx), dim3(512), lds, st>>>(p);
    return 0;
}

}  // namespace

extern "C" int sed_conv_dgrad_c1_nparts(void) { return 256; }        // one workgroup per CU

static int dgrad_c1_stats_impl(int dtype, const void* dz, const void* wpack_t, const float* x1, const float* fmean,
                                const float* fstd, const void* relu_mask, float* a_partial, void* g_dbg, int B, int H, int W,
                                void* stream) {
    SED_REQUIRE(dtype == SED_BF16 && W == 64, "covered: bf16, W = 64, 32 -> 32 channels");
    SED_REQUIRE(dz && wpack_t && x1 && relu_mask && a_partial && B > 0 && H > 0, "operands");
    SED_REQUIRE((fmean == nullptr) == (fstd == nullptr), "mean/std must both be given or both NULL");
    SED_REQUIRE((double)H * W * 32 * 2 < 2147483648.0, "one image must stay below 2 GiB");
    DgradC1Params p = {};
    p.dz = dz; p.wpack_t = wpack_t; p.x1 = x1; p.fmean = fmean; p.fstd = fstd;
    p.mask = reinterpret_cast<const unsigned*>(relu_mask); p.out = a_partial; p.g_dbg = g_dbg; p.B = B; p.H = H;
    p.nparts = sed_conv_dgrad_c1_nparts();
    int th = 8;
    if (const char* e = sed_getenv("SED_DGRAD_TH")) th = atoi(e) == 4 ? 4 : 8;     // tuning knob
    const int rc = g_dbg ? (th == 4 ? launch_dgrad_c1a<4, true>(p, (hipStream_t)stream) : launch_dgrad_c1a<8, true>(p, (hipStream_t)stream))
                         : (th == 4 ? launch_dgrad_c1a<4>(p, (hipStream_t)stream) : launch_dgrad_c1a<8>(p, (hipStream_t)stream));
    if (rc) return rc;
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_conv3x3_dgrad_c1_stats(int dtype, const void* dz, const void* wpack_t, const float* x1, const float* fmean,
                                          const float* fstd, const void* relu_mask, float* a_partial, int B, int H, int W,
                                          void* stream) {
    return dgrad_c1_stats_impl(dtype, dz, wpack_t, x1, fmean, fstd, relu_mask, a_partial, nullptr, B, H, W, stream);
}
// test hook: additionally writes the gated data gradient g [B][H][64][32] bf16 (what sed_conv3x3_dgrad_c1 stores)
extern "C" int sed_conv3x3_dgrad_c1_stats_g(int dtype, const void* dz, const void* wpack_t, const float* x1, const float* fmean,
                                            const float* fstd, const void* relu_mask, float* a_partial, void* g_out, int B, int H,
                                            int W, void* stream) {
    SED_REQUIRE(g_out, "g_out");
    return dgrad_c1_stats_impl(dtype, dz, wpack_t, x1, fmean, fstd, relu_mask, a_partial, g_out, B, H, W, stream);
}
