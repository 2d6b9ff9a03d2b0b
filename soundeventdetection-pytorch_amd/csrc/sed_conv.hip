// 3x3 convolution kernels for gfx950 (CDNA4): implicit GEMM on MFMA over LDS-staged NHWC tiles.
//
// Replaces nn.Conv2d(3x3, s1, p1, bias=False) forward / autograd backward of ConvBlock
// (/root/reference/models/spectogram_models.py:132-140,155-156).
//
// Orientation: D[cout][pixel] += Wfrag[cout][k] * Xfrag[k][pixel]   (k = input channels of a tap)
//   - MFMA 32x32x16 bf16 (or 32x32x2 f32 in the fp32-accurate mode): A operand = weights,
//     B operand = activations, so the accumulator has the PIXEL on the lane and 4 consecutive
//     output channels in consecutive registers -> NHWC stores of 8/16 B per lane, and per-channel
//     BatchNorm statistics accumulate per lane across tiles and are reduced once per workgroup.
//   - activations: LDS image [rows+2][W+2 (pitch WP)][32 ch], XOR-swizzled per pixel column so that
//     the 32 pixels of a fragment read hit distinct banks; the 9 taps are 9 shifted reads of it.
//   - weights: pre-packed by sed_pack_conv_weight() to [chunk][tap][32/KR][Coutp][KR] so both the
//     global->LDS copy and the fragment read are linear.
#include "conv_common.h"

#include <stdlib.h>
#include <algorithm>

// =================================================================================================
// weight packing
// =================================================================================================
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin,
                                   int POp, int PIp, int tf) {
    constexpr int KR = EL<T>::KR;
    const size_t total = (size_t)PIp * 9 * POp;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        // idx = (((chunk*9 + tap)*(32/KR) + kq)*POp + po)*KR + kr
        size_t t = idx;
        const int kr = t % KR; t /= KR;
        const int po = t % POp; t /= POp;
        const int kq = t % (32 / KR); t /= (32 / KR);
        const int tap = t % 9;
        const int chunk = t / 9;
        const int pi = chunk * 32 + kq * KR + kr;
        float v = 0.f;
        if (!tf) {
            if (po < Cout && pi < Cin) v = w[((size_t)po * Cin + pi) * 9 + tap];
        } else {  // packed-out = conv Cin, packed-in = conv Cout, taps flipped
            if (po < Cin && pi < Cout) v = w[((size_t)pi * Cin + po) * 9 + (8 - tap)];
        }
        out[idx] = from_f<T>(v);
    }
}

// One launch for every conv layer of a step (forward and data-gradient operators): desc[i] = {w, out, Cout, Cin, POp, PIp,
// tf, first_block} as eight 64-bit words; block b serves 1024 elements of the descriptor whose block range holds b.
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(const long long* __restrict__ desc, int n) {
    constexpr int KR = EL<T>::KR;
    int d = 0;
    for (int i = 1; i < n; ++i)
        if ((int)desc[i * 8 + 7] <= (int)blockIdx.x) d = i;
    const long long* e = desc + d * 8;
    const float* __restrict__ w = reinterpret_cast<const float*>(e[0]);
    T* __restrict__ out = reinterpret_cast<T*>(e[1]);
    const int Cout = (int)e[2], Cin = (int)e[3], POp = (int)e[4], PIp = (int)e[5], tf = (int)e[6];
    const size_t total = (size_t)PIp * 9 * POp;
    const size_t base = (size_t)((int)blockIdx.x - (int)e[7]) * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t idx = base + u * 256 + threadIdx.x;
        if (idx >= total) break;
        size_t t = idx;
        const int kr = t % KR; t /= KR;
        const int po = t % POp; t /= POp;
        const int kq = t % (32 / KR); t /= (32 / KR);
        const int tap = t % 9;
        const int chunk = t / 9;
        const int pi = chunk * 32 + kq * KR + kr;
        float v = 0.f;
        if (!tf) {
            if (po < Cout && pi < Cin) v = w[((size_t)po * Cin + pi) * 9 + tap];
        } else {
            if (po < Cin && pi < Cout) v = w[((size_t)pi * Cin + po) * 9 + (8 - tap)];
        }
        out[idx] = from_f<T>(v);
    }
}

// dtype SED_F32X3: the operator as two bf16 images in the bf16 layout, [hi = bf16(w)][lo = bf16(w - hi)] (sed_conv_x3.hip)
typedef _Float16 sed_half_t;
__device__ __forceinline__ void x3_pieces(float v, int half, unsigned short& hi, unsigned short& lo) {
    if (half) {                 // fp16 pieces, lo scaled by 2^11 (sed_conv_x3.hip)
        const sed_half_t h = (sed_half_t)v;
        const sed_half_t l = (sed_half_t)((v - (float)h) * 2048.f);
        hi = __builtin_bit_cast(unsigned short, h);
        lo = __builtin_bit_cast(unsigned short, l);
    } else {
        const bf16_t h = (bf16_t)v;
        const bf16_t l = (bf16_t)(v - (float)h);
        hi = __builtin_bit_cast(unsigned short, h);
        lo = __builtin_bit_cast(unsigned short, l);
    }
}
__global__ void pack_weight_x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout, int Cin, int POp, int PIp, int tf,
                                      int half) {
    const size_t total = (size_t)PIp * 9 * POp;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t t = idx;
        const int kr = t % 8; t /= 8;
        const int po = t % POp; t /= POp;
        const int kq = t % 4; t /= 4;
        const int tap = t % 9;
        const int chunk = t / 9;
        const int pi = chunk * 32 + kq * 8 + kr;
        float v = 0.f;
        if (!tf) {
            if (po < Cout && pi < Cin) v = w[((size_t)po * Cin + pi) * 9 + tap];
        } else {
            if (po < Cin && pi < Cout) v = w[((size_t)pi * Cin + po) * 9 + (8 - tap)];
        }
        x3_pieces(v, half, out[idx], out[total + idx]);
    }
}
__global__ __launch_bounds__(256) void pack_weight_batch_x3_kernel(const long long* __restrict__ desc, int n, int half) {
    int d = 0;
    for (int i = 1; i < n; ++i)
        if ((int)desc[i * 8 + 7] <= (int)blockIdx.x) d = i;
    const long long* e = desc + d * 8;
    const float* __restrict__ w = reinterpret_cast<const float*>(e[0]);
    unsigned short* __restrict__ out = reinterpret_cast<unsigned short*>(e[1]);
    const int Cout = (int)e[2], Cin = (int)e[3], POp = (int)e[4], PIp = (int)e[5], tf = (int)e[6];
    const size_t total = (size_t)PIp * 9 * POp;
    const size_t base = (size_t)((int)blockIdx.x - (int)e[7]) * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t idx = base + u * 256 + threadIdx.x;
        if (idx >= total) break;
        size_t t = idx;
        const int kr = t % 8; t /= 8;
        const int po = t % POp; t /= POp;
        const int kq = t % 4; t /= 4;
        const int tap = t % 9;
        const int chunk = t / 9;
        const int pi = chunk * 32 + kq * 8 + kr;
        float v = 0.f;
        if (!tf) {
            if (po < Cout && pi < Cin) v = w[((size_t)po * Cin + pi) * 9 + tap];
        } else {
            if (po < Cin && pi < Cout) v = w[((size_t)pi * Cin + po) * 9 + (8 - tap)];
        }
        x3_pieces(v, half, out[idx], out[total + idx]);
    }
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cout, int Cin,
                                    int Coutp, int Cinp) {
    const int total = Cout * Cin * 9;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int tap = idx % 9;
        const int ci = (idx / 9) % Cin;
        const int co = idx / (9 * Cin);
        dw[idx] = dwp[((size_t)tap * Cinp + ci) * Coutp + co];
    }
}

// xs[(rowi*WP + coli)*32 + swizzled channel] <- pro(x[b][h0-1+rowi][coli-1][c0 + ..32 channels])
template <typename T, int W, int ROWS, int NTHR> struct HaloRegs {
    static constexpr int ITEMS = ROWS * (W + 2) * 4;
    static constexpr int IPT = (ITEMS + NTHR - 1) / NTHR;
    Raw8<T> raw[IPT];
    bool ok[IPT];
};

// phase 1: every global load of the thread, back to back
template <typename T, int W, int ROWS, int NTHR>
__device__ __forceinline__ void halo_issue(HaloRegs<T, W, ROWS, NTHR>& hr, const T* __restrict__ xg, int b, int h0,
                                           int H, int Cinp, int c0, int tid) {
    typedef HaloRegs<T, W, ROWS, NTHR> HR;
    const int cq = tid & 3;
#pragma unroll
    for (int u = 0; u < HR::IPT; ++u) {
        const int it = tid + u * NTHR;
        const int pix = it >> 2;
        const int rowi = pix / (W + 2), coli = pix - rowi * (W + 2);
        const int h = h0 - 1 + rowi, w = coli - 1;
        hr.ok[u] = (it < HR::ITEMS) && h >= 0 && h < H && w >= 0 && w < W;
        const size_t off = hr.ok[u] ? ((((size_t)b * H + h) * W + w) * Cinp + c0 + cq * 8) : (size_t)(c0 + cq * 8);
        hr.raw[u] = raw_load8<T>(xg + off);
    }
}

// phase 2: prologue + LDS writes.  PS = LDS pixel stride in elements: 32 -> XOR-swizzled channels,
// 40 (bf16 only) -> linear with 16 B of padding per pixel (80 B stride: 5p mod 16 visits every 16-B
// slot, so ds_read_b128 fragment reads are conflict-free AND tap shifts are plain immediates).
template <typename T, int W, int ROWS, int WP, int NTHR, int PS = 32>
__device__ __forceinline__ void halo_commit(const HaloRegs<T, W, ROWS, NTHR>& hr, T* __restrict__ xs, int c0, int pro,
                                            const float* __restrict__ pro_scale, const float* __restrict__ pro_shift,
                                            int tid) {
    typedef HaloRegs<T, W, ROWS, NTHR> HR;
    const int cq = tid & 3;
    float sc[8], sh[8];
    if (pro == SED_PRO_BNRELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = pro_scale[c0 + cq * 8 + e]; sh[e] = pro_shift[c0 + cq * 8 + e]; }
    }
#pragma unroll
    for (int u = 0; u < HR::IPT; ++u) {
        const int it = tid + u * NTHR;
        if (it < HR::ITEMS) {
            const int pix = it >> 2;
            const int rowi = pix / (W + 2), coli = pix - rowi * (W + 2);
            float v[8];
            raw_to_f(hr.raw[u], v);
            if (pro == SED_PRO_BNRELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(0.f, fmaf(v[e], sc[e], sh[e]));
            }
            if (!hr.ok[u]) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            T* dst = xs + (rowi * WP + coli) * PS;
            if constexpr (PS != 32) {
                store8<T>(dst + cq * 8, v);
            } else {
                const int sx = swz<T>(coli);
                if constexpr (sizeof(T) == 2) {
                    store8<T>(dst + ((cq * 8) ^ sx), v);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) dst[(cq * 8 + e) ^ sx] = v[e];
                }
            }
        }
    }
}

template <typename T, int W, int ROWS, int WP, int NTHR>
__device__ __forceinline__ void stage_halo_tile(T* __restrict__ xs, const T* __restrict__ xg, int b, int h0, int H,
                                                int Cinp, int c0, int pro, const float* __restrict__ pro_scale,
                                                const float* __restrict__ pro_shift, int tid) {
    HaloRegs<T, W, ROWS, NTHR> hr;
    halo_issue<T, W, ROWS, NTHR>(hr, xg, b, h0, H, Cinp, c0, tid);
    halo_commit<T, W, ROWS, WP, NTHR>(hr, xs, c0, pro, pro_scale, pro_shift, tid);
}

// =================================================================================================
// generic implicit-GEMM conv (forward and data gradient)
// =================================================================================================

template <typename T, int W, int BM, int WN, int PRO, int EPI>
__global__ __launch_bounds__(256 * WN, 2) void conv_igemm_kernel(ConvParams p) {
    constexpr int BN = 32 * WN;          // output channels per workgroup: one 32-wide N tile per wave column
    constexpr int NTHR = 256 * WN;
    typedef typename EL<T>::frag_t frag_t;
    constexpr int KR = EL<T>::KR, KSTEP = EL<T>::KSTEP;
    constexpr int ES = (int)sizeof(T);
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr int ROWS = TH + 2;
    constexpr int PS = (sizeof(T) == 2) ? 40 : 32;   // LDS pixel stride: padded-linear (bf16) / XOR-swizzled (f32)
    constexpr int XS = ROWS * WP * PS;   // elements
    constexpr int WS = 9 * 32 * BN;      // elements
    constexpr int MT = BM / 128;         // 32-pixel tiles per wave (4 waves along M)
    constexpr int WITEMS = WS / 8;       // 8-element items of one weight chunk
    constexpr int WIPT = (WITEMS + NTHR - 1) / NTHR;
    static_assert(BM % 128 == 0 && BM % W == 0, "tile shape");
    typedef HaloPlan<T, W, ROWS, WP, NTHR, PS> XPlan;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool wres = p.wres != 0;
    const int nchunks_ = p.Cinp >> 5;
    T* xs = reinterpret_cast<T*>(smem);
    T* ws = xs + XS;                                     // [wres ? nchunks : 1][WS]
    T* os = ws + (wres ? nchunks_ : 1) * WS;             // [BM][BN + pad]: output staging of the coalesced epilogue

    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, wn = tid >> 8;
    const int r = lane & 31, hh = lane >> 5;
    const int NY = p.Coutp / BN;
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int by = logical % NY, bx = logical / NY;
    const int n0 = by * BN;
    const int H = p.H, Cinp = p.Cinp, Coutp = p.Coutp;
    const int nchunks = Cinp >> 5;
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ wg = reinterpret_cast<const T*>(p.wpack);
    T* __restrict__ zg = reinterpret_cast<T*>(p.z);
    const T* __restrict__ zr = reinterpret_cast<const T*>(p.zref);
    constexpr int epi = EPI;             // compile-time: straight-line prologue/epilogue code

    // ---- tile-invariant per-lane coordinates / offsets ----------------------------------------------------
    int prow[MT], xbase[MT];
    unsigned eoff[MT];                   // byte offset of the lane's first output channel inside the image, tile row 0
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (wave * MT + mt) * 32 + r;
        prow[mt] = q / W;
        const int rot = (PS == 32) ? 0 : (W == 16) ? 12 * (prow[mt] & 1) : (W == 8) ? 4 * ((((prow[mt] & 3) + 1) >> 1) & 1) : 0;
        const int pcol = (q % W + rot) % W;
        xbase[mt] = (prow[mt] * WP + pcol) * PS;
        eoff[mt] = (unsigned)(((prow[mt] * W + pcol) * Coutp + n0 + wn * 32 + 4 * hh) * ES);
    }
    XPlan xp;
    xp.init(tid, Cinp);
    // weight chunk items: row (tap, kq) of BN*KR contiguous elements in LDS; source row stride Coutp*KR
    unsigned wsrc[WIPT];
    int wdst[WIPT];
    Raw8<T> wraw[WIPT];
    {
        constexpr int ROWLEN = BN * KR;
        constexpr int ITEMS_PER_ROW = ROWLEN / 8;
#pragma unroll
        for (int u = 0; u < WIPT; ++u) {
            const int it = tid + u * NTHR;
            const int rowi = it / ITEMS_PER_ROW, off = (it - rowi * ITEMS_PER_ROW) * 8;
            const bool ok = it < WITEMS;
            wsrc[u] = ok ? (unsigned)(((rowi * Coutp + n0) * KR + off) * ES) : SED_OOB;
            wdst[u] = ok ? rowi * ROWLEN + off : 0;
        }
    }
    const size_t wchunk_bytes = (size_t)(9 * 32 / KR) * Coutp * KR * ES;     // one 32-input-channel chunk of wpack
    const __amdgpu_buffer_rsrc_t wsrd = make_srd(wg, wchunk_bytes * nchunks);
    const size_t ximg = (size_t)H * W * Cinp, zimg = (size_t)H * W * Coutp;

    float S[16], Q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { S[i] = 0.f; Q[i] = 0.f; }

    // ---- coalesced epilogue ---------------------------------------------------------------------------------
    // The accumulator has the pixel on the lane and 4 consecutive output channels per register group: stored
    // straight from there, every lane of a store instruction hits a different 128-byte line (8 bytes each).
    // Instead the tile is written to an LDS staging image [pixel][BN (+16 B pad)] and, after the next barrier
    // of the stage loop, read back 16 bytes per lane in pixel-major order: each store instruction then writes
    // whole lines.  The fused ReLU-mask / BatchNorm-backward statistics run in that pass, on coalesced loads of
    // the reference tile (requested one stage ahead): a thread's items all belong to the same 8 channels.
    constexpr int BNP = BN + 16 / ES;
    constexpr int IPR = BN / 8;                   // 8-channel items per staged pixel
    constexpr int FIPT = BM * IPR / NTHR;         // items per thread
    constexpr int FQS = NTHR / IPR;               // pixels between two items of a thread
    static_assert((BM * IPR) % NTHR == 0 && NTHR % IPR == 0, "flush geometry");
    int ostg[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (wave * MT + mt) * 32 + r;
        const int rot = (PS == 32) ? 0 : (W == 16) ? 12 * (prow[mt] & 1) : (W == 8) ? 4 * ((((prow[mt] & 3) + 1) >> 1) & 1) : 0;
        ostg[mt] = (prow[mt] * W + (q % W + rot) % W) * BNP + wn * 32 + 4 * hh;
    }
    const int fcg = tid % IPR, fq0 = tid / IPR;
    const int fl_lds0 = fq0 * BNP + fcg * 8;
    const unsigned fl_off0 = (unsigned)((fq0 * Coutp + n0 + fcg * 8) * ES);
    const unsigned fl_step = (unsigned)(FQS * Coutp * ES);
    Raw8<T> zraw[FIPT];
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
        const __amdgpu_buffer_rsrc_t zs = make_srd(zg + (size_t)fb * zimg, zimg * ES);
        const unsigned tq = (unsigned)(fh0 * W * Coutp * ES);
#pragma unroll
        for (int u = 0; u < FIPT; ++u) {
            float v[8];
            load8<T>(os + fl_lds0 + u * FQS * BNP, v);
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
            if (!(SED_DBG(p, 1))) buf_store8<T>(zs, fl_off0 + u * fl_step + tq, v);
        }
    };

    const int t_begin = bx * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    const int nst = (t_end > t_begin ? (t_end - t_begin) : 0) * nchunks;   // stages = (tile, chunk)

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
        if (SED_DBG(p, 8)) return;
        xp.issue(make_srd(xg + (size_t)b * ximg, ximg * ES), (unsigned)((((h0 - 1) * W - 1) * Cinp + kc * 32) * ES));
        if (with_w) {
            const unsigned wo = (unsigned)(kc * wchunk_bytes);
#pragma unroll
            for (int u = 0; u < WIPT; ++u) wraw[u] = buf_load8<T>(wsrd, wsrc[u] + wo);
        }
    };
    auto commit = [&](int s, bool with_w) {
        int b, h0, kc;
        coords(s, b, h0, kc);
        const int row_hi = (H - h0 < ROWS - 1) ? (H - h0) : (ROWS - 1);
        xp.template commit<PRO>(xs, tid, p.pro_scale, p.pro_shift, kc * 32, h0 == 0 ? 1 : 0, row_hi);
        if (with_w) {
#pragma unroll
            for (int u = 0; u < WIPT; ++u) {
                if (u == WIPT - 1 && tid + u * NTHR >= WITEMS) break;
                lds_store_raw<T>(ws + wdst[u], wraw[u]);
            }
        }
    };
    if (wres && nst > 0) {      // one-time: every chunk of this N slice into LDS
        for (int c = 0; c < nchunks; ++c) {
            const unsigned wo = (unsigned)(c * wchunk_bytes);
#pragma unroll
            for (int u = 0; u < WIPT; ++u) wraw[u] = buf_load8<T>(wsrd, wsrc[u] + wo);
#pragma unroll
            for (int u = 0; u < WIPT; ++u) {
                if (u == WIPT - 1 && tid + u * NTHR >= WITEMS) break;
                lds_store_raw<T>(ws + c * WS + wdst[u], wraw[u]);
            }
        }
    }
    const bool stage_w_each = !wres && nchunks > 1;

    f32x16 acc[MT];
    if (nst > 0) issue(0, !wres);
    for (int s = 0; s < nst; ++s) {
        int b, h0, kc;
        coords(s, b, h0, kc);
        __syncthreads();                                   // previous stage's readers of xs/ws are done; staging is complete
        if (pending) { flush(); pending = false; }         // previous tile: LDS -> whole-line global stores
        const bool need_w = stage_w_each || (!wres && s == 0);   // single-chunk layers: staged once, stay resident
        commit(s, need_w);
        __syncthreads();
        if (s + 1 < nst) issue(s + 1, stage_w_each);       // next stage's loads fly during the MFMAs below
        if (kc == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
        }
        const unsigned tq = (unsigned)(h0 * W * Coutp * ES);
        if (epi == SED_EPI_RELUBWD && kc == nchunks - 1) {   // reference tile of the flush one stage from now
            const __amdgpu_buffer_rsrc_t rs = make_srd(zr + (size_t)b * zimg, zimg * ES);
#pragma unroll
            for (int u = 0; u < FIPT; ++u) zraw[u] = buf_load8<T>(rs, fl_off0 + u * fl_step + tq);
        }
        // ---- 9 taps x (32/KSTEP) k-steps of MFMA ------------------------------------------------------------
        const T* __restrict__ wsc = ws + (wres ? kc * WS : 0);
        if (!(SED_DBG(p, 2)))
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ti = tap / 3, tj = tap % 3;
#pragma unroll 4
            for (int ks = 0; ks < 32 / KSTEP; ++ks) {
                const int kb = ks * KSTEP + hh * KR;   // first channel of this lane's fragment
                const frag_t wf = *reinterpret_cast<const frag_t*>(wsc + ((tap * (32 / KR) + kb / KR) * BN + wn * 32 + r) * KR);
                frag_t xf[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if constexpr (PS == 32) {
                        const int col = (xbase[mt] / PS) % WP + tj;
                        xf[mt] = *reinterpret_cast<const frag_t*>(xs + xbase[mt] + (ti * WP + tj) * PS + (kb ^ swz<T>(col)));
                    } else {
                        xf[mt] = *reinterpret_cast<const frag_t*>(xs + xbase[mt] + (ti * WP + tj) * PS + kb);
                    }
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma(wf, xf[mt], acc[mt]);
            }
        }
        if (kc != nchunks - 1) continue;

        // ---- tile done: forward statistics from the fp32 accumulators, then stage the tile for the flush ------
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
                store4<T>(os + ostg[mt] + 8 * g, v);
            }
        }
        fb = b; fh0 = h0; pending = true;
    }
    if (pending) {
        __syncthreads();
        flush();
    }

    // ---- per-workgroup statistics partial ------------------------------------------------------
    if (epi == SED_EPI_STATS) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);   // [wn][wave][quarter][stat][16] (reuses the tile buffers)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float sv = row16_sum(S[i]);
            const float qv = row16_sum(Q[i]);
            if ((lane & 15) == 0) {
                const int quarter = lane >> 4;
                red[(((wn * 4 + wave) * 4 + quarter) * 2 + 0) * 16 + i] = sv;
                red[(((wn * 4 + wave) * 4 + quarter) * 2 + 1) * 16 + i] = qv;
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int stat = tid / BN, cn = tid % BN;
            const int wcol = cn >> 5, within = cn & 31;
            const int hhh = (within >> 2) & 1;
            const int reg = (within & 3) + 4 * (within >> 3);
            float tot = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
                    tot += red[(((wcol * 4 + wv) * 4 + 2 * hhh + qq) * 2 + stat) * 16 + reg];
            p.partial[((size_t)bx * 2 + stat) * Coutp + n0 + cn] = tot;
        }
    } else if (epi == SED_EPI_RELUBWD) {
        // thread t accumulated channels 8*(t % IPR) .. +7 over its pixels: fixed-order sum over the FQS threads
        // of each channel group; Q was accumulated as gate*(z - mean), the 1/std factor is applied here
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
// bf16 fast path: weights stationary in REGISTERS.
//   Each wave owns 32 output channels (wn) and MT 32-pixel tiles (wm): its 18 A-fragments of the
//   current 32-channel chunk (9 taps x 2 k-steps, 72 VGPRs) are loaded straight from the packed
//   weights in L2 (coalesced 16 B/lane) -- once per workgroup when Cin = 32, otherwise prefetched
//   for the next chunk into a second register set while the current chunk computes.  Only the
//   activation halo tile goes through LDS (double buffered; the next stage's global loads are in
//   flight during the MFMAs, its LDS writes follow them; ONE barrier per stage), and every MFMA
//   needs exactly one ds_read_b128 (the B-fragment), half of what the LDS-weights tiling needs.
//   One wave per SIMD (up to 512 registers): the overlap is explicit, not by occupancy.
// =================================================================================================
template <int W, int WM, int WN, int PRO, int EPI>
__global__ __launch_bounds__(256) void conv_wreg_kernel(ConvParams p) {
    typedef bf16_t T;
    constexpr int BM = 256;
    constexpr int MT = BM / (32 * WM);
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr int ROWS = TH + 2;
    constexpr int PS = 40;               // padded LDS pixel stride (elements)
    constexpr int XS = ROWS * WP * PS;
    constexpr int BN = 32 * WN;
    static_assert(WM * WN == 4, "four waves");
    typedef HaloRegs<T, W, ROWS, 256> HR;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs0 = reinterpret_cast<T*>(smem);
    T* xs1 = xs0 + XS;
    constexpr int BNP = BN + 8;          // output staging row (elements): BN channels + 16 bytes of padding
    T* os = xs1 + XS;                    // [BM][BNP]: output staging of the coalesced epilogue (see conv_igemm_kernel)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int r = lane & 31, hh = lane >> 5;
    const int NY = p.Coutp / BN;
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int by = logical % NY, bx = logical / NY;
    const int n0 = by * BN, cw = n0 + wn * 32;
    const int H = p.H, Cinp = p.Cinp, Coutp = p.Coutp;
    const int nchunks = Cinp >> 5;
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const bf16x8* __restrict__ wg8 = reinterpret_cast<const bf16x8*>(p.wpack);
    T* __restrict__ zg = reinterpret_cast<T*>(p.z);
    constexpr int pro = PRO, epi = EPI;

    int prow[MT], pcol[MT], xbase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (wm * MT + mt) * 32 + r;
        prow[mt] = q / W;
        // A 32-pixel MFMA tile spans 32/W rows; rotating the columns of some rows (found by exhaustive
        // search over the ds_read_b128 lane groups) keeps the padded layout bank-conflict free:
        // W=16: odd rows +12; W=8: rows 1,2 (mod 4) +4.  Any lane->pixel bijection is legal.
        const int rot = (W == 16) ? 12 * (prow[mt] & 1) : (W == 8) ? 4 * ((((prow[mt] & 3) + 1) >> 1) & 1) : 0;
        pcol[mt] = (q % W + rot) % W;
        xbase[mt] = (prow[mt] * WP + pcol[mt]) * PS + hh * 8;   // + (ti*WP + tj)*PS + ks*16: immediates
    }

    f32x16 acc[MT];

    // 18 A-fragments of chunk kc: wpack[kc][tap][kq = 2*ks + hh][Coutp][8]
    auto load_w = [&](bf16x8 (&wf)[18], int kc) {
#pragma unroll
        for (int t = 0; t < 18; ++t)
            wf[t] = wg8[(size_t)((kc * 9 + (t >> 1)) * 4 + (t & 1) * 2 + hh) * Coutp + cw + r];
    };

    // 18 groups (tap, k-step) of MT MFMAs.  Both operand sets are refilled IN PLACE: the B-fragment of
    // M tile mt is re-read from LDS for group g+1 right after its MFMA of group g has issued (it is
    // needed MT MFMAs later), and -- when the next stage uses another chunk (`refill`) -- weight fragment
    // g is re-loaded from L2 after its last MFMA (needed a whole stage later).  One fragment set each;
    // the fences stop the scheduler from hoisting everything and spilling.
    auto xfrag = [&](const T* __restrict__ xs, int g, int mt) -> bf16x8 {
        const int tap = g >> 1, ks = g & 1;
        const int ti = tap / 3, tj = tap % 3;
        return *reinterpret_cast<const bf16x8*>(xs + xbase[mt] + (ti * WP + tj) * PS + ks * 16);
    };
    auto compute = [&](bf16x8 (&wf)[18], const T* __restrict__ xs, bool refill, int kc_next) {
        bf16x8 xf[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[mt] = xfrag(xs, 0, mt);
#pragma unroll
        for (int g = 0; g < 18; ++g) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[g], xf[mt], acc[mt], 0, 0, 0);
                if (g + 1 < 18) xf[mt] = xfrag(xs, g + 1, mt);
            }
            if (refill) wf[g] = wg8[(size_t)((kc_next * 9 + (g >> 1)) * 4 + (g & 1) * 2 + hh) * Coutp + cw + r];
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // tile done: it goes to the LDS staging image [pixel][BN + pad] and is written out with whole-line stores
    // by flush() after the stage's barrier.  The forward BatchNorm statistics are taken there too, from the
    // values as stored (bf16), 8 fixed channels per thread, accumulated in a thread-owned LDS slot: this
    // kernel has no registers left for 32 persistent accumulators (they spilled to scratch).
    static_assert(EPI != SED_EPI_RELUBWD, "the fused ReLU/BN-backward epilogue runs in conv_igemm_kernel");
    auto epilogue = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int ob = (prow[mt] * W + pcol[mt]) * BNP + wn * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[mt][4 * g + e];
                store4<T>(os + ob + 8 * g, v);
            }
        }
    };
    constexpr int IPR = BN / 8, FIPT = BM * IPR / 256, FQS = 256 / IPR;
    const int fcg = tid % IPR, fq0 = tid / IPR;
    const size_t zimg = (size_t)H * W * Coutp;
    float* sslot = reinterpret_cast<float*>(os + BM * BNP) + tid * 16;   // [256][S 0..7, Q 0..7]
    if (epi == SED_EPI_STATS) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sslot[e] = 0.f;
    }
    int fb = 0, fh0 = 0;
    bool pending = false;
    auto flush = [&]() {
        const __amdgpu_buffer_rsrc_t zs = make_srd(zg + (size_t)fb * zimg, zimg * 2);
        const unsigned base = (unsigned)(((fh0 * W + fq0) * Coutp + n0 + fcg * 8) * 2);
        float ts[8], tq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { ts[e] = 0.f; tq[e] = 0.f; }
#pragma unroll
        for (int u = 0; u < FIPT; ++u) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(os + (fq0 + u * FQS) * BNP + fcg * 8);
            if (epi == SED_EPI_STATS && fh0 + (fq0 + u * FQS) / W < H) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; ts[e] += f; tq[e] = fmaf(f, f, tq[e]); }
            }
            if (!(SED_DBG(p, 1)))    // rows past the image: dropped by the descriptor's range check
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), zs, base + (unsigned)(u * FQS * Coutp * 2), 0, 0);
        }
        if (epi == SED_EPI_STATS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sslot[e] += ts[e]; sslot[8 + e] += tq[e]; }
        }
    };

    const int t_begin = bx * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    const int nst = (t_end > t_begin ? (t_end - t_begin) : 0) * nchunks;   // stages = (tile, chunk) pairs

    auto stage_coords = [&](int s, int& b, int& h0, int& kc) {
        const int tile = t_begin + s / nchunks;
        kc = s - (s / nchunks) * nchunks;
        b = tile / p.tilesPerImg;
        h0 = (tile - b * p.tilesPerImg) * TH;
    };

    bf16x8 wf[18];
    HR hr;
    if (nst > 0) {
        int b, h0, kc;
        stage_coords(0, b, h0, kc);
        load_w(wf, 0);
        halo_issue<T, W, ROWS, 256>(hr, xg, b, h0, H, Cinp, 0, tid);
        halo_commit<T, W, ROWS, WP, 256, PS>(hr, xs0, 0, pro, p.pro_scale, p.pro_shift, tid);
    }
    __syncthreads();

    // one stage: issue stage s+1's activation loads (-> hr), compute stage s from xs[s&1] (refilling
    // the weight fragments for stage s+1 on the way), then write stage s+1's activations into the
    // other LDS buffer; ONE barrier per stage.
    for (int s = 0; s < nst; ++s) {
        int b, h0, kc;
        stage_coords(s, b, h0, kc);
        const bool more = (s + 1 < nst);
        int b1 = 0, h1 = 0, kc1 = 0;
        if (pending) { flush(); pending = false; }   // staged by every wave before the barrier that ended the last stage
        if (more) {
            stage_coords(s + 1, b1, h1, kc1);
            halo_issue<T, W, ROWS, 256>(hr, xg, b1, h1, H, Cinp, kc1 * 32, tid);
        }
        if (kc == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
        }
        if (!(SED_DBG(p, 2))) compute(wf, (s & 1) ? xs1 : xs0, more && nchunks > 1, kc1);
        // (the next write of the staging image is >= 2 stages = one more barrier away: nchunks >= 2 here)
        if (kc == nchunks - 1) { epilogue(); fb = b; fh0 = h0; pending = true; }
        if (more) halo_commit<T, W, ROWS, WP, 256, PS>(hr, (s & 1) ? xs0 : xs1, kc1 * 32, pro, p.pro_scale, p.pro_shift, tid);
        __syncthreads();
    }
    if (pending) flush();

    // ---- per-workgroup statistics partial ------------------------------------------------------
    if (epi == SED_EPI_STATS) {
        // fixed-order sum over the FQS threads that own each channel group
        __syncthreads();
        const float* slots = reinterpret_cast<const float*>(os + BM * BNP);
        if (tid < 2 * BN) {
            const int stat = tid / BN, cn = tid % BN;
            const int cg = cn >> 3, e = cn & 7;
            float tot = 0.f;
            for (int k = 0; k < FQS; ++k) tot += slots[(cg + IPR * k) * 16 + stat * 8 + e];
            p.partial[((size_t)bx * 2 + stat) * Coutp + n0 + cn] = tot;
        }
    }
}

// =================================================================================================
// weight gradient: dW[tap][cin][cout] = sum_pix a[pix+tap][cin] * dz[pix][cout]
//   D[cin][cout] += A[cin][k=pixel] * B[k=pixel][cout]; both operands are "k-strided" in NHWC, so
//   bf16 fragments come from ds_read_b64_tr_b16 (hardware transpose read), f32 fragments are
//   single elements.
// =================================================================================================
struct WgradParams {
    const void* x;
    const float* pro_scale;
    const float* pro_shift;
    const void* dz;
    float* ws;   // [strips][9][Cinp][Coutp]
    int B, H, Cinp, Coutp;
    int tilesPerImg, totalTiles, tpb, strips;
    int pro;
};

template <typename T, int W, int WN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
    typedef typename EL<T>::frag_t frag_t;
    constexpr int KR = EL<T>::KR, KSTEP = EL<T>::KSTEP;
    constexpr int BM = 128;
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr int ROWS = TH + 2;
    constexpr int XS = ROWS * WP * 32;
    constexpr int CO = 32 * WN;
    constexpr int WK = 4 / WN;           // waves along the pixel (K) axis
    constexpr int PXW = BM / WK;         // pixels per wave per tile
    constexpr int DZS = BM * CO;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = reinterpret_cast<T*>(smem);
    T* dzs = xs + XS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wn = wave % WN, wk = wave / WN;
    const int H = p.H, Cinp = p.Cinp, Coutp = p.Coutp;
    const int NCO = Coutp / CO;
    const int strip = blockIdx.x;
    const int ci0 = (blockIdx.y / NCO) * 32, co0 = (blockIdx.y % NCO) * CO;
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ dg = reinterpret_cast<const T*>(p.dz);
    const int pro = p.pro;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // lane-constant parts of the transpose-read addresses (bf16): the lane supplies k-row
    // 8*hh + q (+4 for the second half) and the 4 channels 16*gbit + 4*pp .. +3
    int offA[3][2], offB[2];
    {
        const int i16 = lane & 15, gbit = (lane >> 4) & 1;
        const int qq = i16 >> 2, pp = i16 & 3, ch = 16 * gbit + 4 * pp;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int kl = 8 * hh + qq + 4 * half;
            const int rq = kl / W, cq = kl % W;
#pragma unroll
            for (int tj = 0; tj < 3; ++tj) offA[tj][half] = (rq * WP + cq + tj) * 32 + (ch ^ swz<bf16_t>(cq + tj));
            offB[half] = kl * CO + wn * 32 + ch;
        }
    }

    const int t_begin = strip * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int b = tile / p.tilesPerImg;
        const int h0 = (tile - b * p.tilesPerImg) * TH;
        __syncthreads();
        // stage activations (32-channel chunk ci0) with halo, prologue fused
        stage_halo_tile<T, W, ROWS, WP, 256>(xs, xg, b, h0, H, Cinp, ci0, pro, p.pro_scale, p.pro_shift, tid);
        // stage dz tile [BM pixels][CO]; rows past H are zero
        {
            constexpr int IPP = CO / 8;
            constexpr int DIT = BM * IPP / 256;
            Raw8<T> raw[DIT];
            bool ok[DIT];
#pragma unroll
            for (int u = 0; u < DIT; ++u) {
                const int it = tid + u * 256;
                const int q = it / IPP, c8 = (it % IPP) * 8;
                const int h = h0 + q / W, w = q % W;
                ok[u] = h < H;
                raw[u] = raw_load8<T>(dg + (ok[u] ? (((size_t)b * H + h) * W + w) * Coutp : 0) + co0 + c8);
            }
#pragma unroll
            for (int u = 0; u < DIT; ++u) {
                const int it = tid + u * 256;
                const int q = it / IPP, c8 = (it % IPP) * 8;
                float v[8];
                raw_to_f(raw[u], v);
                if (!ok[u]) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = 0.f;
                }
                store8<T>(dzs + q * CO + c8, v);
            }
        }
        __syncthreads();

#pragma unroll 2
        for (int k0 = wk * PXW; k0 < (wk + 1) * PXW; k0 += KSTEP) {
            frag_t bf;
            frag_t af[9];
            if constexpr (sizeof(T) == 2) {
                // uniform part of the addresses: this k-step's first pixel (k0 is a multiple of 16, so it
                // never changes the swizzle bits of the lane-constant part)
                const int ub = ((k0 / W) * WP + (k0 % W)) * 32;
                bf = join_tr(ds_read_tr16_b64(dzs + k0 * CO + offB[0]), ds_read_tr16_b64(dzs + k0 * CO + offB[1]));
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int ti = tap / 3, tj = tap % 3;
                    af[tap] = join_tr(ds_read_tr16_b64(xs + ub + ti * WP * 32 + offA[tj][0]),
                                      ds_read_tr16_b64(xs + ub + ti * WP * 32 + offA[tj][1]));
                }
            } else {
                const int k = k0 + hh;
                bf = dzs[k * CO + wn * 32 + r];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int ti = tap / 3, tj = tap % 3;
                    const int rr = k / W + ti, cc = k % W + tj;
                    af[tap] = xs[(rr * WP + cc) * 32 + (r ^ swz<T>(cc))];
                }
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) acc[tap] = mfma(af[tap], bf, acc[tap]);
        }
    }

    // reduce the WK pixel-waves and write this strip's partial: D row = cin, col (lane) = cout
    float* red = reinterpret_cast<float*>(smem);   // [WK][WN][16][64]
    float* out = p.ws + (size_t)strip * 9 * Cinp * Coutp;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) red[((wk * WN + wn) * 16 + i) * 64 + lane] = acc[tap][i];
        __syncthreads();
        for (int e = tid; e < WN * 16 * 64; e += 256) {
            const int l = e & 63, i = (e >> 6) & 15, n = e >> 10;
            float tot = 0.f;
#pragma unroll
            for (int kk = 0; kk < WK; ++kk) tot += red[((kk * WN + n) * 16 + i) * 64 + l];
            const int cin = ci0 + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
            const int cout = co0 + n * 32 + (l & 31);
            out[((size_t)tap * Cinp + cin) * Coutp + cout] = tot;
        }
    }
}

// =================================================================================================
// weight gradient v2.
//   dW[tap][cin][cout] = sum_pix a[pix+tap][cin] * dz[pix][cout]
//   * the 9 taps are split over waves by tap ROW: wave (wt, wn) owns taps (wt, 0..2) x 32 cin x 32 cout
//     (48 accumulator registers; no cross-wave reduction, each wave stores its own slab);
//   * both operands are k(=pixel)-strided in NHWC: bf16 fragments come from ds_read_b64_tr_b16;
//   * the next tile's global loads are issued before the current tile's MFMAs (register prefetch);
//   * dz can be PRODUCED here (fused BatchNorm/ReLU/pool backward), and is then also written out by
//     the cin-tile-0 workgroups for the data-gradient kernel:
//       DZ_GIVEN : dz read as stored
//       DZ_POOL  : dz = ca*g + cb*z + cc, g = up(dy)/pool^2 * [scale*z+shift > 0]   (z = z2 of the block)
//       DZ_BN    : dz = ca*g + cb*z + cc, g stored (data-gradient epilogue output), z = z1
// =================================================================================================
template <typename T, int W, int WN, int DZ, int PRO>
__global__ __launch_bounds__(192 * WN) void conv_wgrad2_kernel(Wgrad2Params p) {
    typedef typename EL<T>::frag_t frag_t;
    constexpr int KSTEP = EL<T>::KSTEP;
    constexpr int NTHR = 192 * WN;
    constexpr int BM = 128;
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr int ROWS = TH + 2;
    constexpr int XS = ROWS * WP * 32;
    constexpr int CO = 32 * WN;
    constexpr int IPP = CO / 8;                       // 8-channel items per pixel of the dz tile
    constexpr int DITEMS = BM * IPP;
    constexpr int DIT = (DITEMS + NTHR - 1) / NTHR;
    constexpr int ES = (int)sizeof(T);
    typedef HaloPlan<T, W, ROWS, WP, NTHR, 32> XPlan;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = reinterpret_cast<T*>(smem);
    T* dzs = xs + XS;                                  // [WN][BM][32]
    float* coef = reinterpret_cast<float*>(dzs + WN * BM * 32);   // [5][CO]: scale, shift, ca, cb, cc

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave % 3, wn = wave / 3;
    const int r = lane & 31, hh = lane >> 5;
    const int H = p.H, Cinp = p.Cinp, Coutp = p.Coutp;
    const int NCO = Coutp / CO;
    // 1-D grid, XCD-aware: the NY = (Cinp/32)*NCO workgroups of one strip read the same dz sources (and the
    // same 128-byte lines of x), so they get consecutive logical ids = the same XCD's L2, close in time.
    const int NY = (Cinp >> 5) * NCO;
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int strip = logical / NY, yb = logical - strip * NY;
    const int ci_tile = yb / NCO;
    const int ci0 = ci_tile * 32, co0 = (yb % NCO) * CO;
    const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ dg = reinterpret_cast<const T*>(p.dz);
    const T* __restrict__ zsg = reinterpret_cast<const T*>(p.zsrc);
    T* __restrict__ dzo = (ci_tile == 0) ? reinterpret_cast<T*>(p.dz_out) : nullptr;
    const int psh = p.pool >> 1;                      // pool is 1 or 2
    const int Ho = H >> psh, Wo = W >> psh;
    const float inv_pool = psh ? 0.25f : 1.0f;

    if (DZ != DZ_GIVEN) {
        for (int i = tid; i < 5 * CO; i += NTHR) {
            const int a = i / CO, c = i - a * CO;
            const float* src = (a == 0) ? p.scale : (a == 1) ? p.shift : (a == 2) ? p.ca : (a == 3) ? p.cb : p.cc;
            float v = (src != nullptr) ? src[co0 + c] : 0.f;
            if (a == 2 && DZ == DZ_POOL) v *= inv_pool;       // the 1/pool^2 of the avg-pool backward folded into ca
            coef[i] = v;
        }
    }

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // lane-constant parts of the transpose-read addresses (bf16): the lane supplies k-row
    // 8*hh + q (+4 for the second half) and the 4 channels 16*gbit + 4*pp .. +3
    int offA[3][2], offB[2];
    {
        const int i16 = lane & 15, gbit = (lane >> 4) & 1;
        const int qq = i16 >> 2, pp = i16 & 3, ch = 16 * gbit + 4 * pp;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int kl = 8 * hh + qq + 4 * half;
            const int rq = kl / W, cq = kl % W;
#pragma unroll
            for (int tj = 0; tj < 3; ++tj)
                offA[tj][half] = ((rq + wt) * WP + cq + tj) * 32 + (ch ^ swz<bf16_t>(cq + tj));
            offB[half] = (wn * BM + kl) * 32 + ch;
        }
    }

    // ---- tile-invariant staging plans ---------------------------------------------------------------------
    XPlan xp;
    xp.init(tid, Cinp);
    unsigned dvoff[DIT], pvoff[DIT];
    int dlds[DIT], dq[DIT];
    Raw8<T> da[DIT], db[DIT];
#pragma unroll
    for (int u = 0; u < DIT; ++u) {
        const int it = tid + u * NTHR;
        const int q = it / IPP, c8 = (it - q * IPP) * 8;
        const bool ok = it < DITEMS;
        dq[u] = ok ? q : BM;                           // BM = "never valid"
        dvoff[u] = ok ? (unsigned)((q * Coutp + co0 + c8) * ES) : SED_OOB;
        pvoff[u] = ok ? (unsigned)(((((q / W) >> psh) * Wo + ((q % W) >> psh)) * Coutp + co0 + c8) * ES) : SED_OOB;
        dlds[u] = ((c8 >> 5) * BM + (ok ? q : 0)) * 32 + (c8 & 31);
    }
    const size_t ximg = (size_t)H * W * Cinp, zimg = (size_t)H * W * Coutp, pimg = (size_t)Ho * Wo * Coutp;

    auto issue = [&](int tile) {
        const int b = tile / p.tilesPerImg;
        const int h0 = (tile - b * p.tilesPerImg) * TH;
        if (SED_DBG(p, 8)) return;
        xp.issue(make_srd(xg + (size_t)b * ximg, ximg * ES), (unsigned)((((h0 - 1) * W - 1) * Cinp + ci0) * ES));
        const unsigned dt = (unsigned)(h0 * W * Coutp * ES);
        if (DZ == DZ_POOL) {
            const __amdgpu_buffer_rsrc_t gs = make_srd(dg + (size_t)b * pimg, pimg * ES);
            const __amdgpu_buffer_rsrc_t zs = make_srd(zsg + (size_t)b * zimg, zimg * ES);
            const unsigned pt = (unsigned)((h0 >> psh) * Wo * Coutp * ES);
#pragma unroll
            for (int u = 0; u < DIT; ++u) { da[u] = buf_load8<T>(gs, pvoff[u] + pt); db[u] = buf_load8<T>(zs, dvoff[u] + dt); }
        } else if (DZ == DZ_BN) {
            const __amdgpu_buffer_rsrc_t gs = make_srd(dg + (size_t)b * zimg, zimg * ES);
            const __amdgpu_buffer_rsrc_t zs = make_srd(zsg + (size_t)b * zimg, zimg * ES);
#pragma unroll
            for (int u = 0; u < DIT; ++u) { da[u] = buf_load8<T>(gs, dvoff[u] + dt); db[u] = buf_load8<T>(zs, dvoff[u] + dt); }
        } else {
            const __amdgpu_buffer_rsrc_t gs = make_srd(dg + (size_t)b * zimg, zimg * ES);
#pragma unroll
            for (int u = 0; u < DIT; ++u) da[u] = buf_load8<T>(gs, dvoff[u] + dt);
        }
    };

    auto commit = [&](int tile) {
        const int b = tile / p.tilesPerImg;
        const int h0 = (tile - b * p.tilesPerImg) * TH;
        const int row_hi = (H - h0 < ROWS - 1) ? (H - h0) : (ROWS - 1);
        xp.template commit<PRO>(xs, tid, p.pro_scale, p.pro_shift, ci0, h0 == 0 ? 1 : 0, row_hi);
        const int qmax = (H - h0) * W;                 // pixels of the tile inside the image (>= BM except on the last tile)
        const __amdgpu_buffer_rsrc_t os = make_srd(dzo ? dzo + (size_t)b * zimg : nullptr, dzo ? zimg * ES : 0);
        const unsigned dt = (unsigned)(h0 * W * Coutp * ES);
#pragma unroll
        for (int u = 0; u < DIT; ++u) {
            if (u == DIT - 1 && dq[u] >= BM) break;
            if (DZ == DZ_GIVEN) {
                lds_store_raw<T>(dzs + dlds[u], da[u]);     // rows past the image were read as zeros
            } else {
                const int c8 = (dlds[u] & 31) + 32 * (dlds[u] / (BM * 32));
                float g[8], z[8], v[8];
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
                        const float base = fmaf(b4[e], z[i], c4[e]);        // cb*z + cc
                        const float full = fmaf(a4[e], g[i], base);         // + ca*g  (g is 0 where the pool floor dropped the pixel)
                        if (DZ == DZ_POOL) v[i] = (fmaf(z[i], s4[e], t4[e]) > 0.f) ? full : base;   // ReLU gate on g only
                        else v[i] = full;
                    }
                }
                if (qmax < BM && dq[u] >= qmax) {     // only the last tile of an image has rows past it
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = 0.f;
                }
                store8<T>(dzs + dlds[u], v);
                if (dzo != nullptr && !(SED_DBG(p, 1))) buf_store8<T>(os, dvoff[u] + dt, v);   // rows past the image: dropped by the range check
            }
        }
    };

    const int t_begin = strip * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    if (t_begin < t_end) issue(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();                       // previous tile's readers are done (and coef is visible)
        commit(tile);
        __syncthreads();
        if (tile + 1 < t_end) issue(tile + 1);
        if (!(SED_DBG(p, 2)))
#pragma unroll 2
        for (int k0 = 0; k0 < BM; k0 += KSTEP) {
            frag_t bf;
            frag_t af[3];
            if constexpr (sizeof(T) == 2) {
                const int ub = ((k0 / W) * WP + (k0 % W)) * 32;
                bf = join_tr(ds_read_tr16_b64(dzs + k0 * 32 + offB[0]), ds_read_tr16_b64(dzs + k0 * 32 + offB[1]));
#pragma unroll
                for (int tj = 0; tj < 3; ++tj)
                    af[tj] = join_tr(ds_read_tr16_b64(xs + ub + offA[tj][0]), ds_read_tr16_b64(xs + ub + offA[tj][1]));
            } else {
                const int k = k0 + hh;
                bf = dzs[(wn * BM + k) * 32 + r];
#pragma unroll
                for (int tj = 0; tj < 3; ++tj) {
                    const int rr = k / W + wt, cc = k % W + tj;
                    af[tj] = xs[(rr * WP + cc) * 32 + (r ^ swz<T>(cc))];
                }
            }
#pragma unroll
            for (int tj = 0; tj < 3; ++tj) acc[tj] = mfma(af[tj], bf, acc[tj]);
        }
    }

    // each wave stores its own 3 taps x 32 cin x 32 cout slab: D row = cin, col (lane) = cout
    float* out = p.ws + (size_t)strip * 9 * Cinp * Coutp;
#pragma unroll
    for (int tj = 0; tj < 3; ++tj) {
        const int tap = wt * 3 + tj;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int cin = ci0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
            out[((size_t)tap * Cinp + cin) * Coutp + co0 + wn * 32 + r] = acc[tj][i];
        }
    }
}

// out[i] = sum_s ws[s][i]: a 1024-thread workgroup owns 64 consecutive outputs; its 16 waves each walk
// every 16th strip (coalesced 256-byte rows, 8 loads in flight), then a fixed-order LDS reduction.
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                            int strips, size_t n, float* __restrict__ dw = nullptr, int Cout = 0,
                                                            int Cin = 0, int Cinp = 0, int Coutp = 0) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    float t = 0.f;
    if (i < n) {
        int sidx = wv;
        for (; sidx + 16 * 7 < strips; sidx += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(sidx + 16 * u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; sidx < strips; sidx += 16) t += ws[(size_t)sidx * n + i];
    }
    red[wv][lane] = t;
    __syncthreads();
    if (wv == 0 && i < n) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += red[k][lane];
        out[i] = tot;
        if (dw != nullptr) {          // the same value in torch's [Cout][Cin][3][3] layout (what sed_unpack_conv_wgrad writes)
            const int co = (int)(i % Coutp), ci = (int)((i / Coutp) % Cinp), tap = (int)(i / ((size_t)Coutp * Cinp));
            if (co < Cout && ci < Cin) dw[((size_t)co * Cin + ci) * 9 + tap] = tot;
        }
    }
}

// Round 6: the reduction can be DEFERRED.  An entry point called with dwpack == NULL leaves its per-workgroup slabs in the caller's
// workspace, reports their count through sed_wgrad_last_slabs() (per calling thread, like sed_last_error) and launches nothing; the
// caller reduces later -- sed_wgrad_reduce for one layer, sed_wgrad_reduce_batch for several layers in ONE launch (the weight gradients
// feed only the optimizer / the gradient all-reduce: seven dependent 10 us launches of a train step become one at its end).
static thread_local int g_last_slabs = 0;
static int reduce_or_defer(const float* ws, float* dwpack, int slabs, size_t n, float* dw, int Cout, int Cin, int Cinp, int Coutp, hipStream_t st) {
    g_last_slabs = slabs;
    if (dwpack == nullptr) return 0;
    wgrad_reduce_kernel<<<cdiv(n, 64), 1024, 0, st>>>(ws, dwpack, slabs, n, dw, Cout, Cin, Cinp, Coutp);
    return 0;
}
extern "C" int sed_wgrad_last_slabs(void) { return g_last_slabs; }

// desc[i] = {ws, dwpack, dw, slabs, n, Cout, Cin, Cinp, Coutp, first_block} as ten 64-bit words; block b serves 64 outputs of the
// descriptor whose block range holds b (same arithmetic and summation order as wgrad_reduce_kernel: bit-identical results)
__global__ __launch_bounds__(1024) void wgrad_reduce_batch_kernel(const long long* __restrict__ desc, int nd) {
    __shared__ float red[16][64];
    int d = 0;
    for (int i = 1; i < nd; ++i)
        if ((int)desc[i * 10 + 9] <= (int)blockIdx.x) d = i;
    const long long* e = desc + d * 10;
    const float* __restrict__ ws = reinterpret_cast<const float*>(e[0]);
    float* __restrict__ out = reinterpret_cast<float*>(e[1]);
    float* __restrict__ dw = reinterpret_cast<float*>(e[2]);
    const int strips = (int)e[3];
    const size_t n = (size_t)e[4];
    const int Cout = (int)e[5], Cin = (int)e[6], Cinp = (int)e[7], Coutp = (int)e[8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t i = (size_t)((int)blockIdx.x - (int)e[9]) * 64 + lane;
    float t = 0.f;
    if (i < n) {
        int sidx = wv;
        for (; sidx + 16 * 7 < strips; sidx += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(sidx + 16 * u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; sidx < strips; sidx += 16) t += ws[(size_t)sidx * n + i];
    }
    red[wv][lane] = t;
    __syncthreads();
    if (wv == 0 && i < n) {
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) tot += red[k][lane];
        if (out != nullptr) out[i] = tot;
        if (dw != nullptr) {
            const int co = (int)(i % Coutp), ci = (int)((i / Coutp) % Cinp), tap = (int)(i / ((size_t)Coutp * Cinp));
            if (co < Cout && ci < Cin) dw[((size_t)co * Cin + ci) * 9 + tap] = tot;
        }
    }
}

// =================================================================================================
// first layer (Cin = 1): direct, bandwidth bound
// =================================================================================================
// A workgroup walks bands of C1_TR rows grid-stride; per band the C1_TR + 2 input lines are staged in LDS
// (z-scored on the way in, zero padded) behind ONE barrier pair, then thread (w, cg) produces 8 output
// channels of pixel w in each row of the band.  All index math is 32-bit and per band.
constexpr int C1_TR = 8;

template <typename T>
__global__ __launch_bounds__(256) void conv_c1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ stdv, const float* __restrict__ w,
                                                          T* __restrict__ z, float* __restrict__ partial, int B,
                                                          int H, int W, int Cout, int Coutp, int G, int PPB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wsm = reinterpret_cast<float*>(smem);          // [9][Coutp]
    float* xrow = wsm + 9 * Coutp;                        // [C1_TR + 2][W+2]
    float* red = xrow + (C1_TR + 2) * (W + 2);            // [PPB][2][Coutp]
    const int tid = threadIdx.x;
    for (int i = tid; i < 9 * Coutp; i += blockDim.x) {
        const int tap = i / Coutp, c = i % Coutp;
        wsm[i] = c < Cout ? w[c * 9 + tap] : 0.f;
    }
    __syncthreads();
    const int cg = tid % G, pl = tid / G;                 // fixed channel group per thread
    float wr[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) wr[t][e] = wsm[t * Coutp + cg * 8 + e];
    float S[8], Q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { S[e] = 0.f; Q[e] = 0.f; }
    const int WP2 = W + 2;
    const int bands = (H + C1_TR - 1) / C1_TR;
    for (int band = blockIdx.x; band < B * bands; band += gridDim.x) {
        const int b = band / bands, h0 = (band - b * bands) * C1_TR;
        __syncthreads();
        for (int i = tid; i < (C1_TR + 2) * WP2; i += blockDim.x) {
            const int rr = i / WP2, cc = i - rr * WP2;
            const int hy = h0 + rr - 1, wx = cc - 1;
            float v = 0.f;
            if (hy >= 0 && hy < H && wx >= 0 && wx < W) {
                v = x[((size_t)b * H + hy) * W + wx];
                if (mean) v = (v - mean[wx]) / stdv[wx];
            }
            xrow[i] = v;
        }
        __syncthreads();
        if (pl < PPB) {
            for (int wq = pl; wq < W; wq += PPB) {
                // sliding 3x3 window down the band: three new inputs per row
                float x0[3], x1[3], x2[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) { x0[j] = xrow[wq + j]; x1[j] = xrow[WP2 + wq + j]; }
#pragma unroll
                for (int r = 0; r < C1_TR; ++r) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) x2[j] = xrow[(r + 2) * WP2 + wq + j];
                    if (h0 + r < H) {
                        float a[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) a[e] = 0.f;
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) a[e] = fmaf(x0[j], wr[j][e], a[e]);
                        }
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) a[e] = fmaf(x1[j], wr[3 + j][e], a[e]);
                        }
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) a[e] = fmaf(x2[j], wr[6 + j][e], a[e]);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) { S[e] += a[e]; Q[e] = fmaf(a[e], a[e], Q[e]); }
                        store8<T>(z + (((size_t)b * H + h0 + r) * W + wq) * Coutp + cg * 8, a);
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) { x0[j] = x1[j]; x1[j] = x2[j]; }
                }
            }
        }
    }
    if (partial) {
        __syncthreads();
        if (pl < PPB) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(pl * 2 + 0) * Coutp + cg * 8 + e] = S[e];
                red[(pl * 2 + 1) * Coutp + cg * 8 + e] = Q[e];
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * Coutp; i += blockDim.x) {
            float t = 0.f;
            for (int q = 0; q < PPB; ++q) t += red[q * 2 * Coutp + i];
            partial[(size_t)blockIdx.x * 2 * Coutp + i] = t;
        }
    }
}

// With zsrc != NULL the layer's dz is produced on load: dz = ca*g + cb*z + cc (g = `dz` argument = output of
// the data-gradient epilogue, z = the layer's pre-BN output); nothing is written back -- block 0 has no
// data gradient, so its dz1 never needs to exist in memory.
template <typename T, bool FUSED>
__global__ __launch_bounds__(256) void conv_c1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ stdv, const T* __restrict__ dz,
                                                            const T* __restrict__ zsrc, const float* __restrict__ ca,
                                                            const float* __restrict__ cb, const float* __restrict__ cc,
                                                            float* __restrict__ partial, int B, int H, int W,
                                                            int Coutp, int G, int PPB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xrow = reinterpret_cast<float*>(smem);         // [C1_TR + 2][W+2]
    float* red = xrow + (C1_TR + 2) * (W + 2);            // [PPB][Coutp] per tap
    const int tid = threadIdx.x;
    const int cg = tid % G, pl = tid / G;
    const int WP2 = W + 2;
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    float a8[8], b8[8], c8[8];
    if (FUSED) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { a8[e] = ca[cg * 8 + e]; b8[e] = cb[cg * 8 + e]; c8[e] = cc[cg * 8 + e]; }
    }
    const int bands = (H + C1_TR - 1) / C1_TR;
    // the band's input lines are fetched one band ahead into registers (the load -> LDS -> barrier -> compute chain of the
    // first version exposed a full memory latency per band: 31 bands x ~2.5 us per workgroup)
    constexpr int NPF = 4;                                   // staged values per thread: (C1_TR + 2) * (W + 2) <= 4 * 256
    const int nstage = (C1_TR + 2) * WP2;
    float pf[NPF];
    auto fetch = [&](int band) {
        const int b = band / bands, h0 = (band - b * bands) * C1_TR;
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int i = tid + u * 256;
            const int rr = i / WP2, cc2 = i - rr * WP2;
            const int hy = h0 + rr - 1, wx = cc2 - 1;
            float v = 0.f;
            if (i < nstage && band < B * bands && hy >= 0 && hy < H && wx >= 0 && wx < W) {
                v = x[((size_t)b * H + hy) * W + wx];
                if (mean) v = (v - mean[wx]) / stdv[wx];
            }
            pf[u] = v;
        }
    };
    // (host-checked: (C1_TR + 2) * (W + 2) <= NPF * 256)
    fetch(blockIdx.x);
    for (int band = blockIdx.x; band < B * bands; band += gridDim.x) {
        const int b = band / bands, h0 = (band - b * bands) * C1_TR;
        (void)b;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NPF; ++u)
            if (tid + u * 256 < nstage) xrow[tid + u * 256] = pf[u];
        __syncthreads();
        fetch(band + gridDim.x);
        if (pl < PPB) {
            for (int wq = pl; wq < W; wq += PPB) {
                float x0[3], x1[3], x2[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) { x0[j] = xrow[wq + j]; x1[j] = xrow[WP2 + wq + j]; }
                // the band's loads first (independent addresses), then the math
                float d[C1_TR][8], zz[FUSED ? C1_TR : 1][8];
#pragma unroll
                for (int r = 0; r < C1_TR; ++r) {
                    const int h = (h0 + r < H) ? h0 + r : H - 1;       // clamped: the row is skipped below
                    const size_t off = (((size_t)b * H + h) * W + wq) * Coutp + cg * 8;
                    load8<T>(dz + off, d[r]);
                    if (FUSED) load8<T>(zsrc + off, zz[FUSED ? r : 0]);
                }
#pragma unroll
                for (int r = 0; r < C1_TR; ++r) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) x2[j] = xrow[(r + 2) * WP2 + wq + j];
                    if (h0 + r < H) {
                        if (FUSED) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) d[r][e] = fmaf(a8[e], d[r][e], fmaf(b8[e], zz[FUSED ? r : 0][e], c8[e]));
                        }
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                acc[j][e] = fmaf(x0[j], d[r][e], acc[j][e]);
                                acc[3 + j][e] = fmaf(x1[j], d[r][e], acc[3 + j][e]);
                                acc[6 + j][e] = fmaf(x2[j], d[r][e], acc[6 + j][e]);
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 3; ++j) { x0[j] = x1[j]; x1[j] = x2[j]; }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
        if (pl < PPB) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[pl * Coutp + cg * 8 + e] = acc[t][e];
        }
        __syncthreads();
        for (int i = tid; i < Coutp; i += blockDim.x) {
            float sacc = 0.f;
            for (int q = 0; q < PPB; ++q) sacc += red[q * Coutp + i];
            partial[((size_t)blockIdx.x * 9 + t) * Coutp + i] = sacc;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// First-layer weight gradient without the layer's pre-BN output.  With dz1 = ca*g + cb*z1 + cc and
// z1[c] = sum_j w1[c][j]*xp[j] (xp = the 3x3 patch of the z-scored, zero-padded input),
//     dW1[c][k] = sum_px dz1[c]*xp[k] = ca[c]*A[c][k] + cb[c]*sum_j w1[c][j]*G[j][k] + cc[c]*sx[k]
// where A = sum_px g[c]*xp[k] is the plain first-layer weight gradient of g, and G[j][k] = sum_px xp[j]*xp[k],
// sx[k] = sum_px xp[k] depend on the input alone: z1 is never read (and is exact instead of bf16-rounded).
// conv_c1_gram_kernel: partial[block][54] = 45 products (j <= k, row-major upper triangle) then the 9 sums.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_c1_gram_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ stdv, float* __restrict__ partial,
                                                           int B, int H, int W) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xrow = reinterpret_cast<float*>(smem);         // [C1_TR + 2][W+2]
    float* red = xrow + (C1_TR + 2) * (W + 2);            // [4][54]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int WP2 = W + 2;
    float acc[54];
#pragma unroll
    for (int i = 0; i < 54; ++i) acc[i] = 0.f;
    const int bands = (H + C1_TR - 1) / C1_TR;
    // a thread's staging items are the same (row, column) of every band: index arithmetic and the z-score constants are hoisted out of
    // the band loop (they were more than half of the kernel's instructions); z-score as (v - mean) * (1 / std), the form of the
    // convolution kernels' input copy (conv_common.h / sed_conv_pc.hip)
    constexpr int SIT = 4;                                  // items per thread: (C1_TR + 2) * (W + 2) <= 4 * 256 (checked by the launcher)
    int srow[SIT], scol[SIT];
    float smu[SIT], sinv[SIT];
#pragma unroll
    for (int u = 0; u < SIT; ++u) {
        const int i = tid + u * 256;
        const int rr = i / WP2, cc2 = i - rr * WP2;
        const bool ok = i < (C1_TR + 2) * WP2 && cc2 >= 1 && cc2 <= W;
        srow[u] = i < (C1_TR + 2) * WP2 ? rr - 1 : (1 << 28);      // past the staged lines: never inside an image
        scol[u] = ok ? cc2 - 1 : -1;
        smu[u] = (ok && mean) ? mean[cc2 - 1] : 0.f;
        sinv[u] = (ok && mean) ? 1.0f / stdv[cc2 - 1] : 1.f;
    }
    const int npix = C1_TR * W;
    // the next band's lines are fetched into registers while this band's products run (the kernel was bound by one exposed memory
    // latency per band)
    float nraw[SIT];
    unsigned nvalid = 0;
    auto fetch = [&](int band) {
        nvalid = 0;
        if (band >= B * bands) return;
        const int b = band / bands, h0 = (band - b * bands) * C1_TR;
#pragma unroll
        for (int u = 0; u < SIT; ++u) {
            const int hy = h0 + srow[u];
            const bool ok = hy >= 0 && hy < H && scol[u] >= 0;
            nraw[u] = ok ? x[((size_t)b * H + hy) * W + scol[u]] : 0.f;
            nvalid |= ok ? (1u << u) : 0u;
        }
    };
    fetch(blockIdx.x);
    for (int band = blockIdx.x; band < B * bands; band += gridDim.x) {
        const int b = band / bands, h0 = (band - b * bands) * C1_TR;
        (void)b;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < SIT; ++u) {
            const int i = tid + u * 256;
            if (i >= (C1_TR + 2) * WP2) break;
            xrow[i] = ((nvalid >> u) & 1u) ? (nraw[u] - smu[u]) * sinv[u] : 0.f;
        }
        fetch(band + gridDim.x);
        __syncthreads();
        for (int pix = tid; pix < npix; pix += blockDim.x) {
            const int r = pix / W, wq = pix - r * W;
            if (h0 + r >= H) continue;
            float xp[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) xp[t] = xrow[(r + t / 3) * WP2 + wq + t % 3];
#pragma unroll
            for (int j = 0; j < 9; ++j)
#pragma unroll
                for (int k = j; k < 9; ++k) {
                    constexpr int dummy = 0; (void)dummy;
                    const int o = j * 9 - j * (j - 1) / 2 + (k - j);      // constant after unrolling (a running index went to scratch)
                    acc[o] = fmaf(xp[j], xp[k], acc[o]);
                }
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[45 + k] += xp[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 54; ++i) {
        const float t = wave_sum(acc[i]);
        if (lane == 0) red[wave * 54 + i] = t;
    }
    __syncthreads();
    if (tid < 54) partial[(size_t)blockIdx.x * 54 + tid] = red[tid] + red[54 + tid] + red[108 + tid] + red[162 + tid];
}

// BatchNorm statistics of z1 = conv1(x_norm) from the Gram statistics of the input patches:
//   sum z1[c] = sum_k w[c][k]*sx[k],  sum z1[c]^2 = sum_jk w[c][j]*w[c][k]*G[j][k]   (same outputs as bn_train_finalize)
__global__ __launch_bounds__(1024) void bn_train_finalize_c1_kernel(const float* __restrict__ gram, int nparts, double count,
                                                                    const float* __restrict__ w, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float* __restrict__ rmean,
                                                                    float* __restrict__ rvar, float momentum, float eps,
                                                                    float* __restrict__ scale, float* __restrict__ shift,
                                                                    float* __restrict__ mean_o, float* __restrict__ invstd_o, int C,
                                                                    int Cp, double* __restrict__ gsum_out = nullptr) {
    __shared__ double G[54];
    __shared__ double Gp[16][64];
    const int tid = threadIdx.x;
    {
        const int v = tid & 63, g = tid >> 6;
        double s = 0.0;
        if (v < 54) {       // eight independent loads in flight, summed in the same fixed order as a plain loop
            int i = g;
            for (; i + 16 * 7 < nparts; i += 16 * 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = gram[(size_t)(i + 16 * u) * 54 + v];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += (double)t[u];
            }
            for (; i < nparts; i += 16) s += (double)gram[(size_t)i * 54 + v];
        }
        Gp[g][v] = s;
    }
    __syncthreads();
    if (tid < 54) {
        double s = 0.0;
        for (int g = 0; g < 16; ++g) s += Gp[g][tid];
        G[tid] = s;
        if (gsum_out != nullptr) gsum_out[tid] = s;       // the reduced Gram statistics, kept for the backward's tail kernel (sed_c1_bwd_tail)
    }
    __syncthreads();
    for (int c = tid; c < Cp; c += blockDim.x) {
        if (c >= C) { scale[c] = 0.f; shift[c] = 0.f; mean_o[c] = 0.f; invstd_o[c] = 0.f; continue; }
        double s1 = 0.0, s2 = 0.0;
        for (int j = 0; j < 9; ++j) {
            s1 += (double)w[c * 9 + j] * G[45 + j];
            for (int k2 = 0; k2 < 9; ++k2) {
                const int a = j < k2 ? j : k2, b2 = j < k2 ? k2 : j;
                s2 += (double)w[c * 9 + j] * (double)w[c * 9 + k2] * G[a * 9 - a * (a - 1) / 2 + (b2 - a)];
            }
        }
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * invstd;
        scale[c] = sc;
        shift[c] = beta[c] - (float)mean * sc;
        mean_o[c] = (float)mean;
        invstd_o[c] = invstd;
        if (rmean) {
            const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
    }
}

// BatchNorm-1 backward coefficients in C1 mode: sum g from the data-gradient epilogue, sum g*z1 = sum_k w1[c][k]*A[k][c]
// with A = the plain first-layer weight gradient of g (z1 itself is never read)
__global__ __launch_bounds__(256) void bn_bwd_finalize_c1_kernel(const float* __restrict__ partial, int nparts, double count,
                                                                 const float* __restrict__ A, const float* __restrict__ w,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, float* __restrict__ ca,
                                                                 float* __restrict__ cb, float* __restrict__ cc, int C, int Cp) {
    __shared__ double sm[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    double s = 0.0;
    for (int i = tid; i < nparts; i += 256) s += (double)partial[((size_t)i * 2 + 0) * Cp + c];
    sm[tid] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sm[tid] += sm[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        if (c >= C) { ca[c] = 0.f; cb[c] = 0.f; cc[c] = 0.f; return; }
        const double sg = sm[0];
        double sgz = 0.0;
        for (int k2 = 0; k2 < 9; ++k2) sgz += (double)w[c * 9 + k2] * (double)A[k2 * Cp + c];
        const double g = gamma[c], is = invstd[c], mu = mean[c];
        const double q = is * (sgz - mu * sg);              // sum g * xhat
        dbeta[c] = (float)sg;
        dgamma[c] = (float)q;
        const double mg = sg / count, mgx = q / count;
        ca[c] = (float)(g * is);
        cb[c] = (float)(-g * is * is * mgx);
        cc[c] = (float)(-g * is * (mg - mu * is * mgx));
    }
}

__global__ __launch_bounds__(1024) void conv_c1_wgrad_combine_kernel(const float* __restrict__ A, const float* __restrict__ gram,
                                                                    int nparts, const float* __restrict__ w,
                                                                    const float* __restrict__ ca, const float* __restrict__ cb,
                                                                    const float* __restrict__ cc, float* __restrict__ dw,
                                                                    int Cout, int Coutp, float* __restrict__ dw_torch = nullptr) {
    __shared__ double G[54];
    __shared__ double Gp[16][64];
    const int tid = threadIdx.x;
    {   // thread (value v, group g of 16): every 16th partial row, then a fixed-order 16-way sum
        const int v = tid & 63, g = tid >> 6;
        double s = 0.0;
        if (v < 54) {       // eight independent loads in flight, summed in the same fixed order as a plain loop
            int i = g;
            for (; i + 16 * 7 < nparts; i += 16 * 8) {
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = gram[(size_t)(i + 16 * u) * 54 + v];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += (double)t[u];
            }
            for (; i < nparts; i += 16) s += (double)gram[(size_t)i * 54 + v];
        }
        Gp[g][v] = s;
    }
    __syncthreads();
    if (tid < 54) {
        double s = 0.0;
        for (int g = 0; g < 16; ++g) s += Gp[g][tid];
        G[tid] = s;
    }
    __syncthreads();
    for (int idx = tid; idx < 9 * Coutp; idx += blockDim.x) {
        const int k = idx / Coutp, c = idx - k * Coutp;
        float out = 0.f;
        if (c < Cout) {
            double wg = 0.0;
            for (int j = 0; j < 9; ++j) {
                const int a = j < k ? j : k, b2 = j < k ? k : j;          // symmetric: G[a][b2], a <= b2
                wg += (double)w[c * 9 + j] * G[a * 9 - a * (a - 1) / 2 + (b2 - a)];
            }
            out = (float)((double)ca[c] * (double)A[idx] + (double)cb[c] * wg + (double)cc[c] * G[45 + k]);
        }
        dw[idx] = out;
        if (dw_torch != nullptr && c < Cout) dw_torch[c * 9 + k] = out;      // torch layout [Cout][1][3][3]
    }
}

// Block 0's conv1 backward tail in ONE launch (round 5; C1 mode with the fused data gradient, no SyncBN): the three dependent
// one-workgroup-scale kernels sed_sum_partials ([A; sum g] partial rows) -> sed_bn_bwd_finalize_c1 -> sed_conv3x3_c1_wgrad_combine
// (which reduced the forward's Gram partial rows a second time: up to 2048 x 54 floats through one CU) took ~22 us of dependent
// launches per step.  Here: the [A; sum g] rows are summed (fixed order, double), BatchNorm-1's backward coefficients follow, and
// dW1 = ca*A + cb*(w1.G) + cc*sx takes the Gram statistics ALREADY REDUCED by the forward's sed_bn_train_finalize_c1_g (54 doubles).
// Same formulas, same rounding points as the three kernels (a10, ca / cb / cc are rounded to fp32 where they were stored).
__global__ __launch_bounds__(1024) void c1_bwd_tail_kernel(const float* __restrict__ a_part, int a_nparts, const double* __restrict__ gsum,
                                                           double count, const float* __restrict__ w, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ ca,
                                                           float* __restrict__ cb, float* __restrict__ cc, float* __restrict__ a10_out,
                                                           float* __restrict__ dw, int Cout, float* __restrict__ dw_torch) {
    constexpr int Cp = 32, NV = 10 * Cp, NG = 3;
    __shared__ double As[NG][NV];
    __shared__ float a10[NV];
    __shared__ float coef[3][Cp];
    __shared__ double G[54];
    const int tid = threadIdx.x;
    if (tid < 54) G[tid] = gsum[tid];
    if (tid < NG * NV) {        // thread (value v, group g): rows g, g + 3, ..., eight loads in flight, one fixed order
        const int v = tid % NV, g = tid / NV;
        double s = 0.0;
        int i = g;
        for (; i + NG * 7 < a_nparts; i += NG * 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = a_part[(size_t)(i + NG * u) * NV + v];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)t[u];
        }
        for (; i < a_nparts; i += NG) s += (double)a_part[(size_t)i * NV + v];
        As[g][v] = s;
    }
    __syncthreads();
    if (tid < NV) {
        const float t = (float)(As[0][tid] + As[1][tid] + As[2][tid]);
        a10[tid] = t;
        a10_out[tid] = t;
    }
    __syncthreads();
    if (tid < Cp) {             // BatchNorm-1 backward (bn_bwd_finalize_c1_kernel): sum g = row 9, sum g*z1 = w1 . A
        const int c = tid;
        float fa = 0.f, fb = 0.f, fc = 0.f;
        if (c < Cout) {
            const double sg = (double)a10[9 * Cp + c];
            double sgz = 0.0;
            for (int k2 = 0; k2 < 9; ++k2) sgz += (double)w[c * 9 + k2] * (double)a10[k2 * Cp + c];
            const double g = gamma[c], is = invstd[c], mu = mean[c];
            const double q = is * (sgz - mu * sg);
            dbeta[c] = (float)sg;
            dgamma[c] = (float)q;
            const double mg = sg / count, mgx = q / count;
            fa = (float)(g * is);
            fb = (float)(-g * is * is * mgx);
            fc = (float)(-g * is * (mg - mu * is * mgx));
        }
        ca[c] = fa; cb[c] = fb; cc[c] = fc;
        coef[0][c] = fa; coef[1][c] = fb; coef[2][c] = fc;
    }
    __syncthreads();
    for (int idx = tid; idx < 9 * Cp; idx += blockDim.x) {      // conv_c1_wgrad_combine_kernel
        const int k = idx / Cp, c = idx - k * Cp;
        float out = 0.f;
        if (c < Cout) {
            double wg = 0.0;
            for (int j = 0; j < 9; ++j) {
                const int a = j < k ? j : k, b2 = j < k ? k : j;
                wg += (double)w[c * 9 + j] * G[a * 9 - a * (a - 1) / 2 + (b2 - a)];
            }
            out = (float)((double)coef[0][c] * (double)a10[idx] + (double)coef[1][c] * wg + (double)coef[2][c] * G[45 + k]);
        }
        dw[idx] = out;
        if (dw_torch != nullptr && c < Cout) dw_torch[c * 9 + k] = out;
    }
}

// =================================================================================================
// host launchers (C ABI)
// =================================================================================================
static const int kMaxParts = 1024;

extern "C" int sed_conv_nparts(int B, int H, int W) {
    const long long tiles = (long long)B * cdiv((long long)H * W, 256);
    return (int)(tiles < kMaxParts ? tiles : kMaxParts);
}
extern "C" int sed_conv_c1_nparts(int B, int H, int W) {
    (void)W;
    // 768 = 3 resident 256-thread workgroups on each of the 256 CUs: one full round, no 1/3-occupancy tail
    const long long rows = (long long)B * H;
    return (int)(rows < 768 ? rows : 768);
}

extern "C" int sed_pack_conv_weight(int dtype, const float* w, void* wpack, int Cout, int Cin, int Coutp,
                                    int Cinp, int transpose_flip, void* stream) {
    SED_REQUIRE(Coutp % 32 == 0 && Cinp % 32 == 0 && Coutp >= Cout && Cinp >= Cin, "padded channels must be multiples of 32");
    hipStream_t st = (hipStream_t)stream;
    // packed-out / packed-in padded sizes
    const int POp = transpose_flip ? Cinp : Coutp, PIp = transpose_flip ? Coutp : Cinp;
    const size_t total = (size_t)PIp * 9 * POp;
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (dtype == SED_BF16)
        pack_weight_kernel<bf16_t><<<grid, 256, 0, st>>>(w, (bf16_t*)wpack, Cout, Cin, POp, PIp, transpose_flip);
    else if (dtype == SED_F32)
        pack_weight_kernel<float><<<grid, 256, 0, st>>>(w, (float*)wpack, Cout, Cin, POp, PIp, transpose_flip);
    else if (dtype == SED_F32X3 || dtype == SED_F32H3)
        pack_weight_x3_kernel<<<grid, 256, 0, st>>>(w, (unsigned short*)wpack, Cout, Cin, POp, PIp, transpose_flip, dtype == SED_F32H3);
    else
        SED_REQUIRE(false, "bad dtype");
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_unpack_conv_wgrad(const float* dwpack, float* dw, int Cout, int Cin, int Coutp, int Cinp,
                                     void* stream) {
    const int total = Cout * Cin * 9;
    unpack_wgrad_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(dwpack, dw, Cout, Cin, Coutp, Cinp);
    SED_LAUNCH_CHECK();
    return 0;
}

// the (prologue, epilogue) pairs the training / inference paths use
#define SED_PE_DISPATCH(CALL)                                                                         \
    do {                                                                                              \
        if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_STATS) return CALL(SED_PRO_NONE, SED_EPI_STATS);       \
        if (p.pro == SED_PRO_BNRELU && p.epi == SED_EPI_STATS) return CALL(SED_PRO_BNRELU, SED_EPI_STATS);   \
        if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_STORE) return CALL(SED_PRO_NONE, SED_EPI_STORE);       \
        if (p.pro == SED_PRO_BNRELU && p.epi == SED_EPI_STORE) return CALL(SED_PRO_BNRELU, SED_EPI_STORE);   \
        if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_RELUBWD) return CALL(SED_PRO_NONE, SED_EPI_RELUBWD);   \
        sed_set_error("sed_conv3x3_fwd: unsupported prologue/epilogue combination");                  \
        return 1;                                                                                     \
    } while (0)

template <typename T, int W, int BM, int WN, int PRO, int EPI>
static int launch_conv(ConvParams& p, hipStream_t st) {
    constexpr int BN = 32 * WN;
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr int PS = (sizeof(T) == 2) ? 40 : 32;
    constexpr size_t lds_x = (size_t)(TH + 2) * WP * PS * sizeof(T);
    constexpr size_t lds_w1 = (size_t)9 * 32 * BN * sizeof(T);
    constexpr size_t lds_o = (size_t)BM * (BN + 16 / sizeof(T)) * sizeof(T);     // output staging (coalesced epilogue)
    const int nchunks = p.Cinp / 32;
    // multi-chunk layers keep every weight chunk resident when that fits beside the activation tile
    p.wres = (nchunks > 1 && lds_x + nchunks * lds_w1 + lds_o <= 150 * 1024) ? 1 : 0;
    const size_t lds = lds_x + (p.wres ? nchunks : 1) * lds_w1 + lds_o;
    if (int rc_ = sed_set_max_lds<&conv_igemm_kernel<T, W, BM, WN, PRO, EPI>>(lds)) return rc_;
    p.tilesPerImg = cdiv(p.H, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    p.tpb = cdiv(p.totalTiles, p.nparts);
    const int ny = p.Coutp / BN;
    conv_igemm_kernel<T, W, BM, WN, PRO, EPI><<<dim3(p.nparts * ny), dim3(256 * WN), lds, st>>>(p);
    return 0;
}

template <typename T, int W, int BM>
static int dispatch_conv_pe(ConvParams& p, hipStream_t st) {
    if (p.Coutp % 64 == 0) {
#define SED_CALL(P_, E_) launch_conv<T, W, BM, 2, P_, E_>(p, st)
        SED_PE_DISPATCH(SED_CALL);
#undef SED_CALL
    } else {
#define SED_CALL(P_, E_) launch_conv<T, W, BM, 1, P_, E_>(p, st)
        SED_PE_DISPATCH(SED_CALL);
#undef SED_CALL
    }
}

template <typename T, int BM>
static int dispatch_conv_w(ConvParams& p, int W, hipStream_t st) {
    switch (W) {
        case 8: return dispatch_conv_pe<T, 8, BM>(p, st);
        case 16: return dispatch_conv_pe<T, 16, BM>(p, st);
        case 32: return dispatch_conv_pe<T, 32, BM>(p, st);
        case 64: return dispatch_conv_pe<T, 64, BM>(p, st);
    }
    sed_set_error("sed_conv3x3_fwd: W must be one of 8,16,32,64");
    return 1;
}

template <int W, int WM, int WN, int PRO, int EPI>
static int launch_wreg(ConvParams& p, hipStream_t st) {
    constexpr int BM = 256;
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr size_t lds = (size_t)2 * (TH + 2) * WP * 40 * sizeof(bf16_t) + (size_t)BM * (32 * WN + 8) * sizeof(bf16_t) +
                           (EPI == SED_EPI_STATS ? 256 * 16 * sizeof(float) : 0);
    if (int rc_ = sed_set_max_lds<&conv_wreg_kernel<W, WM, WN, PRO, EPI>>(lds)) return rc_;
    p.tilesPerImg = cdiv(p.H, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    p.tpb = cdiv(p.totalTiles, p.nparts);
    const int ny = p.Coutp / (32 * WN);
    conv_wreg_kernel<W, WM, WN, PRO, EPI><<<dim3(p.nparts * ny), dim3(256), lds, st>>>(p);
    return 0;
}

template <int W>
static int dispatch_wreg_pe(ConvParams& p, hipStream_t st) {
    // only the 128-output-channel configuration is dispatched to this kernel (see sed_conv3x3_fwd)
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_STATS) return launch_wreg<W, 1, 4, SED_PRO_NONE, SED_EPI_STATS>(p, st);
    if (p.pro == SED_PRO_BNRELU && p.epi == SED_EPI_STATS) return launch_wreg<W, 1, 4, SED_PRO_BNRELU, SED_EPI_STATS>(p, st);
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_STORE) return launch_wreg<W, 1, 4, SED_PRO_NONE, SED_EPI_STORE>(p, st);
    if (p.pro == SED_PRO_BNRELU && p.epi == SED_EPI_STORE) return launch_wreg<W, 1, 4, SED_PRO_BNRELU, SED_EPI_STORE>(p, st);
    sed_set_error("sed_conv3x3_fwd: unsupported prologue/epilogue combination for the register-weights kernel");
    return 1;
}

static int dispatch_wreg(ConvParams& p, int W, hipStream_t st) {
    switch (W) {
        case 8: return dispatch_wreg_pe<8>(p, st);
        case 16: return dispatch_wreg_pe<16>(p, st);
        case 32: return dispatch_wreg_pe<32>(p, st);
        case 64: return dispatch_wreg_pe<64>(p, st);
    }
    sed_set_error("sed_conv3x3_fwd: W must be one of 8,16,32,64");
    return 1;
}

static int conv3x3_fwd_impl(int dtype, int pro, int epi, const void* x, const float* pro_scale,
                            const float* pro_shift, const void* wpack, void* z, const void* zref,
                            const float* epi_scale, const float* epi_shift, const float* epi_mean,
                            const float* epi_invstd, float* partial, int B, int H, int W, int Cinp, int Coutp,
                            void* stream, int col_only) {
    // SED_F32H3: bits 8..15 of dtype = signed power-of-two exponent applied to the streamed operand x before the fp16 split (gradients)
    const int xexp = (int)(signed char)((dtype >> 8) & 0xff);
    dtype &= 0xff;
    SED_REQUIRE(xexp == 0 || dtype == SED_F32H3, "an operand exponent belongs to dtype SED_F32H3");
    SED_REQUIRE(Cinp % 32 == 0 && Coutp % 32 == 0 && Cinp > 0 && Coutp > 0, "channels must be padded to 32");
    SED_REQUIRE(B > 0 && H > 0, "empty input");
    SED_REQUIRE(pro == SED_PRO_NONE || (pro == SED_PRO_BNRELU && pro_scale && pro_shift), "prologue operands");
    SED_REQUIRE(epi == SED_EPI_STORE || partial, "epilogue needs a partial buffer");
    SED_REQUIRE(epi != SED_EPI_RELUBWD || (zref && epi_scale && epi_shift && epi_mean && epi_invstd), "epilogue operands");
    // buffer addressing: one descriptor per image, 32-bit byte offsets inside it
    SED_REQUIRE((double)H * W * (Cinp > Coutp ? Cinp : Coutp) * (dtype == SED_BF16 ? 2 : 4) < 2147483648.0,
                "one image (H*W*C elements) must stay below 2 GiB");
    ConvParams p = {};
    p.x = x; p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.wpack = wpack; p.z = z; p.zref = zref;
    p.epi_scale = epi_scale; p.epi_shift = epi_shift; p.epi_mean = epi_mean; p.epi_invstd = epi_invstd;
    p.partial = partial; p.B = B; p.H = H; p.Cinp = Cinp; p.Coutp = Coutp; p.pro = pro; p.epi = epi; p.wres = 0;
    p.col_only = col_only;
    p.xexp = xexp;
    { const char* d = sed_getenv("SED_DBG"); p.dbg = d ? atoi(d) : 0; }
    p.nparts = sed_conv_nparts(B, H, W);
    int rc;
    // bf16: the register-stationary-weights kernel wins when a workgroup covers 128 output channels
    // (MFMA-bound layers); the LDS-weights kernel (2 workgroups/CU) wins on the low-channel,
    // memory-bound layers.  SED_CONV_KERNEL=lds|wreg forces one of them (A/B runs).
    const char* force = sed_getenv("SED_CONV_KERNEL");
    // (its fused ReLU/BN-backward epilogue variant does not fit the register file with MT = 8: that
    // one always takes the LDS-weights kernel)
    // register-resident weights (sed_conv_wir.hip) for the >= 64-channel layers it covers; SED_CONV_KERNEL=r forces it
    // where it applies, =p keeps the producer/consumer kernel everywhere (A/B runs)
#ifdef SED_EXPERIMENTS     // (make EXPERIMENTS=1: the two opt-in resident-weight kernels, measured <= the producer/consumer kernel)
    if (dtype == SED_BF16 && force && force[0] == 'r') {
        const int rc_w = launch_conv_wir(p, W, (hipStream_t)stream);
        if (rc_w > 0) return rc_w;
        if (rc_w == 0) { SED_LAUNCH_CHECK(); return 0; }
    }
    // one wave per SIMD, all weights of its 32 output channels in registers (sed_conv_w4.hip); SED_CONV_KERNEL=4
    if (dtype == SED_BF16 && force && force[0] == '4') {
        const int rc_w = launch_conv_w4(p, W, (hipStream_t)stream);
        if (rc_w > 0) return rc_w;
        if (rc_w == 0) { SED_LAUNCH_CHECK(); return 0; }
    }
#else
    SED_REQUIRE(!(force && (force[0] == 'r' || force[0] == '4')), "SED_CONV_KERNEL=r/4 need a library built with make EXPERIMENTS=1");
#endif
    if (dtype == SED_BF16 && !(force && force[0] != 'p' && force[0] != 'r' && force[0] != '4')) {      // producer/consumer kernel (sed_conv_pc.hip) where it covers the shape
        const int rc_pc = launch_conv_pc(p, W, (hipStream_t)stream);
        if (rc_pc > 0) return rc_pc;
        if (rc_pc == 0) { SED_LAUNCH_CHECK(); return 0; }
    }
    const bool want_wreg = (Coutp % 128 == 0) && epi != SED_EPI_RELUBWD && Cinp >= 64 && (force ? (force[0] == 'w') : true);
    if (dtype == SED_BF16 && want_wreg) rc = dispatch_wreg(p, W, (hipStream_t)stream);
    else if (dtype == SED_BF16) rc = dispatch_conv_w<bf16_t, 256>(p, W, (hipStream_t)stream);
    else if (dtype == SED_F32) rc = dispatch_conv_w<float, 128>(p, W, (hipStream_t)stream);
    else if (dtype == SED_F32X3 || dtype == SED_F32H3) rc = launch_conv_x3(dtype == SED_F32H3, p, W, (hipStream_t)stream);
    else { sed_set_error("sed_conv3x3_fwd: bad dtype"); return 1; }
    if (rc) return rc;
    SED_LAUNCH_CHECK();
    return 0;
}

// Data gradient whose epilogue also accumulates the pool + ReLU + BatchNorm backward statistics of the block that produced
// its output's forward twin (include/sed_hip.h).  bf16, shapes of the producer/consumer kernel only.
extern "C" int sed_dgrad_poolstats_supported(int dtype, int W, int Cinp, int Coutp) {
    if (!(dtype == SED_BF16 && (W == 8 || W == 16 || W == 32 || W == 64) && Cinp % 32 == 0 && Coutp % 32 == 0 && Cinp > 0 && Coutp > 0))
        return 0;
    ConvParams p = {};      // ask the producer/consumer dispatcher itself (its LDS budget decides for wide layers)
    p.B = 1; p.H = 64; p.Cinp = Cinp; p.Coutp = Coutp; p.pro = SED_PRO_NONE; p.epi = SED_EPI_POOLSTATS; p.nparts = 1; p.dry = 1;
    return launch_conv_pc(p, W, nullptr) == 0;
}

extern "C" int sed_conv3x3_dgrad_poolstats(int dtype, const void* dz, const void* wpack_t, void* dy, const void* y_pooled,
                                           const void* cnt, const float* scale, const float* shift, const float* mean,
                                           const float* invstd, float* partial, int nparts, int* flag, int B, int H, int W,
                                           int Cinp, int Coutp, void* stream) {
    SED_REQUIRE(sed_dgrad_poolstats_supported(dtype, W, Cinp, Coutp), "sed_conv3x3_dgrad_poolstats: bf16, W in {8,16,32,64}, channels padded to 32");
    SED_REQUIRE(B > 0 && H > 0 && dz && wpack_t && dy && y_pooled && cnt && scale && shift && mean && invstd && partial && flag, "operands");
    SED_REQUIRE((double)H * W * (Cinp > Coutp ? Cinp : Coutp) * 2 < 2147483648.0, "one image (H*W*C elements) must stay below 2 GiB");
    ConvParams p = {};
    p.x = dz; p.wpack = wpack_t; p.z = dy; p.zref = y_pooled; p.cnt = reinterpret_cast<const unsigned char*>(cnt); p.flag = flag;
    p.epi_scale = scale; p.epi_shift = shift; p.epi_mean = mean; p.epi_invstd = invstd;
    p.partial = partial; p.B = B; p.H = H; p.Cinp = Cinp; p.Coutp = Coutp; p.pro = SED_PRO_NONE; p.epi = SED_EPI_POOLSTATS;
    { const char* d = sed_getenv("SED_DBG"); p.dbg = d ? atoi(d) : 0; }
    const int own = sed_conv_nparts(B, H, W);
    SED_REQUIRE(nparts >= own, "partial needs at least sed_conv_nparts(B, H, W) rows");
    p.nparts = nparts;
    const int rc = launch_conv_pc(p, W, (hipStream_t)stream);
    SED_REQUIRE(rc >= 0, "sed_conv3x3_dgrad_poolstats: shape not covered by the producer/consumer kernel");
    if (rc > 0) return rc;
    SED_LAUNCH_CHECK();
    return 0;
}

// Data gradient that PRODUCES dz on load (csrc/sed_conv_pc.hip, SED_PRO_DZBN / SED_PRO_DZPOOL) and writes it out once for the
// weight-gradient call with dz given (include/sed_hip.h).  bf16, W = 16 / 8, dz channels (the layer's outputs) a multiple of 32.
static int dgrad_dz_setup(ConvParams& p, int dzmode, int epi, int B, int H, int C, int Cx, int pool) {
    p.B = B; p.H = H; p.Cinp = C; p.Coutp = Cx; p.epi = epi;
    p.pro = dzmode == SED_DZ_POOL ? SED_PRO_DZPOOL : SED_PRO_DZBN;
    p.dz_pool = dzmode == SED_DZ_POOL ? pool : 1;
    return 0;
}
extern "C" int sed_conv3x3_dgrad_dz_supported(int dtype, int W, int C, int Cx, int dzmode, int epi, int pool) {
    if (!(dtype == SED_BF16 && (W == 8 || W == 16) && C % 32 == 0 && Cx % 32 == 0 && C > 0 && Cx > 0)) return 0;
    if (!(dzmode == SED_DZ_BN || (dzmode == SED_DZ_POOL && (pool == 1 || pool == 2)))) return 0;
    if (dzmode == SED_DZ_BN && !(epi == SED_EPI_STORE || epi == SED_EPI_POOLSTATS)) return 0;
    if (dzmode == SED_DZ_POOL && epi != SED_EPI_RELUBWD) return 0;
#ifndef SED_EXPERIMENTS
    return 0;          // (measured slower than the round-3 order, csrc/sed_conv_pc.hip: built with make EXPERIMENTS=1 only)
#endif
    ConvParams p = {};
    dgrad_dz_setup(p, dzmode, epi, 1, 64, C, Cx, pool);
    p.nparts = 1; p.dry = 1;
    return launch_conv_pc(p, W, nullptr) == 0;
}
extern "C" int sed_conv3x3_dgrad_dz(int dtype, int dzmode, const void* gsrc, const void* zsrc, const float* scale, const float* shift,
                                    const float* ca, const float* cb, const float* cc, int pool, const void* wpack_t, void* dz_out,
                                    void* dx, int epi, const void* zref, const void* cnt, const float* epi_scale, const float* epi_shift,
                                    const float* epi_mean, const float* epi_invstd, float* partial, int nparts, int* flag, int B, int H,
                                    int W, int C, int Cx, void* stream) {
    SED_REQUIRE(sed_conv3x3_dgrad_dz_supported(dtype, W, C, Cx, dzmode, epi, pool),
                "covered: bf16, W = 16 / 8; SED_DZ_BN with STORE / POOLSTATS, SED_DZ_POOL (pool 1 / 2) with RELUBWD");
    SED_REQUIRE(B > 0 && H > 0 && gsrc && zsrc && ca && cb && cc && wpack_t && dz_out && dx, "operands");
    SED_REQUIRE(dzmode != SED_DZ_POOL || (scale && shift), "pool-backward operands");
    SED_REQUIRE(epi == SED_EPI_STORE || (zref && epi_scale && epi_shift && epi_mean && epi_invstd && partial), "epilogue operands");
    SED_REQUIRE(epi != SED_EPI_POOLSTATS || (cnt && flag), "pooled-tensor statistics operands");
    SED_REQUIRE((double)H * W * (C > Cx ? C : Cx) * 2 < 2147483648.0, "one image (H*W*C elements) must stay below 2 GiB");
    ConvParams p = {};
    dgrad_dz_setup(p, dzmode, epi, B, H, C, Cx, pool);
    p.x = zsrc; p.dz_g = gsrc; p.dz_ca = ca; p.dz_cb = cb; p.dz_cc = cc; p.dz_sc = scale; p.dz_sh = shift; p.dz_out = dz_out;
    p.wpack = wpack_t; p.z = dx; p.zref = zref; p.cnt = reinterpret_cast<const unsigned char*>(cnt); p.flag = flag;
    p.epi_scale = epi_scale; p.epi_shift = epi_shift; p.epi_mean = epi_mean; p.epi_invstd = epi_invstd; p.partial = partial;
    { const char* d = sed_getenv("SED_DBG"); p.dbg = d ? atoi(d) : 0; }
    const int own = sed_conv_nparts(B, H, W);
    SED_REQUIRE(epi == SED_EPI_STORE || nparts >= own, "partial needs at least sed_conv_nparts(B, H, W) rows");
    p.nparts = epi == SED_EPI_STORE ? own : nparts;
    const int rc = launch_conv_pc(p, W, (hipStream_t)stream);
    SED_REQUIRE(rc >= 0, "sed_conv3x3_dgrad_dz: shape not covered by the producer/consumer kernel");
    if (rc > 0) return rc;
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_conv3x3_fwd(int dtype, int pro, int epi, const void* x, const float* pro_scale,
                               const float* pro_shift, const void* wpack, void* z, const void* zref,
                               const float* epi_scale, const float* epi_shift, const float* epi_mean,
                               const float* epi_invstd, float* partial, int B, int H, int W, int Cinp, int Coutp,
                               void* stream) {
    return conv3x3_fwd_impl(dtype, pro, epi, x, pro_scale, pro_shift, wpack, z, zref, epi_scale, epi_shift, epi_mean, epi_invstd,
                            partial, B, H, W, Cinp, Coutp, stream, 0);
}

// the same call for 3x3 weights whose side columns are zero (a k = 3 Conv1d over interleaved frames, W = 8): the covered
// kernels skip the six zero taps, every other path computes them (same result)
extern "C" int sed_conv3x3_fwd_col(int dtype, int pro, int epi, const void* x, const float* pro_scale,
                                   const float* pro_shift, const void* wpack, void* z, const void* zref,
                                   const float* epi_scale, const float* epi_shift, const float* epi_mean,
                                   const float* epi_invstd, float* partial, int B, int H, int W, int Cinp, int Coutp,
                                   void* stream) {
    return conv3x3_fwd_impl(dtype, pro, epi, x, pro_scale, pro_shift, wpack, z, zref, epi_scale, epi_shift, epi_mean, epi_invstd,
                            partial, B, H, W, Cinp, Coutp, stream, 1);
}

static int wgrad_strips(int B, int H, int W, int Cinp, int Coutp, int* wn_out) {
    const int wn = Coutp % 128 == 0 ? 4 : (Coutp % 64 == 0 ? 2 : 1);
    if (wn_out) *wn_out = wn;
    const int ny = (Cinp / 32) * (Coutp / (32 * wn));
    const int TH = 128 / W;
    const long long tiles = (long long)B * cdiv(H, TH);
    long long target = (ny == 1) ? 1024 : 512;     // measured optimum (tools/bench_layer.py sweep); total workgroups
    if (const char* e = sed_getenv("SED_WGRAD_BLOCKS")) target = atoll(e) > 0 ? atoll(e) : target;   // tuning knob
    long long strips = cdiv(target, ny);
    if (strips > tiles) strips = tiles;
    if (strips < 1) strips = 1;
    return (int)strips;
}

extern "C" size_t sed_conv_wgrad_ws_floats(int B, int H, int W, int Cinp, int Coutp) {
    // one slab per workgroup of whichever kernel runs (bf16: producer/consumer, fp32: v2; the fused backward launches of
    // sed_bwd_fused.hip / sed_bwd_fused_c1.hip cut their strips differently: H + 1 rows, shorter tiles -> more slabs when B*H is small)
    int n = wgrad_strips(B, H, W, Cinp, Coutp, nullptr);
    n = std::max(n, wgrad3_strips(B, H, W, Cinp, Coutp));
    n = std::max(n, bwd_fused_max_nwg(B, H, W, Cinp, Coutp));
    if (W == 64 && Cinp == 32 && Coutp == 32) n = std::max(n, bwd_fused_c1_nwg(B, H));
    return (size_t)n * 9 * Cinp * Coutp;
}

template <typename T, int W, int WN, int DZ, int PRO>
static int launch_wgrad2(Wgrad2Params& p, hipStream_t st) {
    constexpr int TH = 128 / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr size_t lds = ((size_t)(TH + 2) * WP * 32 + (size_t)WN * 128 * 32) * sizeof(T) + (size_t)5 * 32 * WN * sizeof(float);
    if (int rc_ = sed_set_max_lds<&conv_wgrad2_kernel<T, W, WN, DZ, PRO>>(lds)) return rc_;
    p.tilesPerImg = cdiv(p.H, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    p.tpb = cdiv(p.totalTiles, p.strips);
    const int ny = (p.Cinp / 32) * (p.Coutp / (32 * WN));
    conv_wgrad2_kernel<T, W, WN, DZ, PRO><<<dim3(p.strips * ny), dim3(192 * WN), lds, st>>>(p);
    return 0;
}

template <typename T, int DZ>
static int dispatch_wgrad2(Wgrad2Params& p, int W, int wn, hipStream_t st) {
#define SED_CASE(WW)                                                                                          \
    case WW:                                                                                                  \
        if (p.pro == SED_PRO_BNRELU) {                                                                        \
            if (wn == 4) return launch_wgrad2<T, WW, 4, DZ, SED_PRO_BNRELU>(p, st);                           \
            if (wn == 2) return launch_wgrad2<T, WW, 2, DZ, SED_PRO_BNRELU>(p, st);                           \
            return launch_wgrad2<T, WW, 1, DZ, SED_PRO_BNRELU>(p, st);                                        \
        }                                                                                                     \
        if (wn == 4) return launch_wgrad2<T, WW, 4, DZ, SED_PRO_NONE>(p, st);                                 \
        if (wn == 2) return launch_wgrad2<T, WW, 2, DZ, SED_PRO_NONE>(p, st);                                 \
        return launch_wgrad2<T, WW, 1, DZ, SED_PRO_NONE>(p, st);
    switch (W) {
        SED_CASE(8)
        SED_CASE(16)
        SED_CASE(32)
        SED_CASE(64)
    }
#undef SED_CASE
    sed_set_error("sed_conv3x3_wgrad: W must be one of 8,16,32,64");
    return 1;
}

static int wgrad_common(int dtype, int pro, int dzmode, const void* x, const float* pro_scale, const float* pro_shift,
                        const void* dz, const void* zsrc, const float* scale, const float* shift, const float* ca,
                        const float* cb, const float* cc, int pool, void* dz_out, float* dwpack, float* workspace,
                        int B, int H, int W, int Cinp, int Coutp, hipStream_t st, float* dw = nullptr, int Cout = 0, int Cin = 0) {
    const int dzexp = (int)(signed char)((dtype >> 8) & 0xff);      // SED_F32H3: exponent applied to dz before the fp16 split
    dtype &= 0xff;
    if (dzexp != 0 && dtype != SED_F32H3) { sed_set_error("sed_conv3x3_wgrad: an operand exponent belongs to dtype SED_F32H3"); return 1; }
    if ((double)H * W * (Cinp > Coutp ? Cinp : Coutp) * (dtype == SED_BF16 ? 2 : 4) >= 2147483648.0) {
        sed_set_error("sed_conv3x3_wgrad: one image (H*W*C elements) must stay below 2 GiB");
        return 1;
    }
    Wgrad2Params p = {};
    p.dzexp = dzexp;
    int wn;
    p.strips = wgrad_strips(B, H, W, Cinp, Coutp, &wn);
    p.x = x; p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.dz = dz; p.zsrc = zsrc; p.scale = scale;
    p.shift = shift; p.ca = ca; p.cb = cb; p.cc = cc; p.dz_out = dz_out; p.ws = workspace;
    p.B = B; p.H = H; p.Cinp = Cinp; p.Coutp = Coutp; p.pro = pro; p.pool = pool < 1 ? 1 : pool;
    { const char* d = sed_getenv("SED_DBG"); p.dbg = d ? atoi(d) : 0; }
    int rc = 1;
    if (dtype == SED_BF16) {           // producer/consumer kernel (sed_wgrad.hip) where the shape is covered
        rc = launch_wgrad3(dzmode, p, W, st);
    } else {
        rc = -1;
    }
    if (rc < 0) {
        p.strips = wgrad_strips(B, H, W, Cinp, Coutp, &wn);
#define SED_DZ(T_)                                                                         \
    (dzmode == DZ_GIVEN ? dispatch_wgrad2<T_, DZ_GIVEN>(p, W, wn, st)                      \
     : dzmode == DZ_POOL ? dispatch_wgrad2<T_, DZ_POOL>(p, W, wn, st)                      \
                         : dispatch_wgrad2<T_, DZ_BN>(p, W, wn, st))
    if (dtype == SED_BF16) rc = SED_DZ(bf16_t);
    else if (dtype == SED_F32) rc = SED_DZ(float);
    else if (dtype == SED_F32X3 || dtype == SED_F32H3) {
        rc = dtype == SED_F32H3 ? launch_wgrad_x3pc(dzmode, p, W, st) : -1;      // fp16 pieces: the producer / consumer kernel
        if (rc < 0) {
            p.strips = wgrad_strips(B, H, W, Cinp, Coutp, &wn);
            rc = launch_wgrad_x3(dtype == SED_F32H3, dzmode, p, W, wn, st);
        }
    }
    else { sed_set_error("sed_conv3x3_wgrad: bad dtype"); return 1; }
#undef SED_DZ
    }
    if (rc) return rc;
    {
        hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) { sed_set_error(std::string("sed_conv3x3_wgrad: launch failed: ") + hipGetErrorString(e_)); return 2; }
    }
    const size_t n = (size_t)9 * Cinp * Coutp;
    reduce_or_defer(workspace, dwpack, p.strips, n, dw, Cout, Cin, Cinp, Coutp, st);
    {
        hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) { sed_set_error(std::string("sed_conv3x3_wgrad: reduce launch failed: ") + hipGetErrorString(e_)); return 2; }
    }
    return 0;
}

extern "C" int sed_conv3x3_wgrad(int dtype, int pro, const void* x, const float* pro_scale, const float* pro_shift,
                                 const void* dz, float* dwpack, float* workspace, int B, int H, int W, int Cinp,
                                 int Coutp, void* stream) {
    SED_REQUIRE(Cinp % 32 == 0 && Coutp % 32 == 0, "channels must be padded to 32");
    SED_REQUIRE(pro == SED_PRO_NONE || (pro_scale && pro_shift), "prologue operands");
    return wgrad_common(dtype, pro, DZ_GIVEN, x, pro_scale, pro_shift, dz, nullptr, nullptr, nullptr, nullptr, nullptr,
                        nullptr, 1, nullptr, dwpack, workspace, B, H, W, Cinp, Coutp, (hipStream_t)stream);
}

extern "C" int sed_conv3x3_wgrad_u(int dtype, int pro, const void* x, const float* pro_scale, const float* pro_shift, const void* dz,
                                   float* dwpack, float* workspace, int B, int H, int W, int Cinp, int Coutp, float* dw, int Cout,
                                   int Cin, void* stream) {
    SED_REQUIRE(Cinp % 32 == 0 && Coutp % 32 == 0, "channels must be padded to 32");
    SED_REQUIRE(pro == SED_PRO_NONE || (pro_scale && pro_shift), "prologue operands");
    SED_REQUIRE(dw && Cout > 0 && Cin > 0 && Cout <= Coutp && Cin <= Cinp, "unpacked gradient operands");
    return wgrad_common(dtype, pro, DZ_GIVEN, x, pro_scale, pro_shift, dz, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1,
                        nullptr, dwpack, workspace, B, H, W, Cinp, Coutp, (hipStream_t)stream, dw, Cout, Cin);
}

extern "C" int sed_conv3x3_wgrad_fused(int dtype, int pro, const void* x, const float* pro_scale,
                                       const float* pro_shift, int dzmode, const void* gsrc, const void* zsrc,
                                       const float* scale, const float* shift, const float* ca, const float* cb,
                                       const float* cc, int pool, void* dz_out, float* dwpack, float* workspace, int B,
                                       int H, int W, int Cinp, int Coutp, void* stream) {
    SED_REQUIRE(Cinp % 32 == 0 && Coutp % 32 == 0, "channels must be padded to 32");
    SED_REQUIRE(pro == SED_PRO_NONE || (pro_scale && pro_shift), "prologue operands");
    SED_REQUIRE(dzmode == SED_DZ_POOL || dzmode == SED_DZ_BN, "dzmode must be SED_DZ_POOL or SED_DZ_BN");
    SED_REQUIRE(gsrc && zsrc && ca && cb && cc, "fused dz operands");
    SED_REQUIRE(dzmode != SED_DZ_POOL || (scale && shift && (pool == 1 || pool == 2)), "pool-backward operands");
    return wgrad_common(dtype, pro, dzmode, x, pro_scale, pro_shift, gsrc, zsrc, scale, shift, ca, cb, cc, pool, dz_out,
                        dwpack, workspace, B, H, W, Cinp, Coutp, (hipStream_t)stream);
}

extern "C" int sed_conv3x3_wgrad_fused_u(int dtype, int pro, const void* x, const float* pro_scale, const float* pro_shift,
                                         int dzmode, const void* gsrc, const void* zsrc, const float* scale, const float* shift,
                                         const float* ca, const float* cb, const float* cc, int pool, void* dz_out, float* dwpack,
                                         float* workspace, int B, int H, int W, int Cinp, int Coutp, float* dw, int Cout, int Cin,
                                         void* stream) {
    SED_REQUIRE(Cinp % 32 == 0 && Coutp % 32 == 0, "channels must be padded to 32");
    SED_REQUIRE(pro == SED_PRO_NONE || (pro_scale && pro_shift), "prologue operands");
    SED_REQUIRE(dzmode == SED_DZ_POOL || dzmode == SED_DZ_BN, "dzmode must be SED_DZ_POOL or SED_DZ_BN");
    SED_REQUIRE(gsrc && zsrc && ca && cb && cc, "fused dz operands");
    SED_REQUIRE(dzmode != SED_DZ_POOL || (scale && shift && (pool == 1 || pool == 2)), "pool-backward operands");
    SED_REQUIRE(dw && Cout > 0 && Cin > 0 && Cout <= Coutp && Cin <= Cinp, "unpacked gradient operands");
    return wgrad_common(dtype, pro, dzmode, x, pro_scale, pro_shift, gsrc, zsrc, scale, shift, ca, cb, cc, pool, dz_out,
                        dwpack, workspace, B, H, W, Cinp, Coutp, (hipStream_t)stream, dw, Cout, Cin);
}

extern "C" int sed_conv3x3_bwd_fused_supported_pool(int dtype, int W, int Cinp, int Coutp, int dzmode, int pro, int epi, int pool) {
    if (dtype != SED_BF16) return 0;
    if (W == 32) return (dzmode != SED_DZ_POOL || pool == 2) && bwd_fused_nwg(1, 64, W, Cinp, Coutp, dzmode, pro, epi) > 0;
#ifdef SED_EXPERIMENTS
    return bwd_fused_cs_nstrips(1, 64, W, Cinp, Coutp, dzmode, pro, epi, pool) > 0;
#else
    return 0;
#endif
}
extern "C" int sed_conv3x3_bwd_fused_supported(int dtype, int W, int Cinp, int Coutp, int dzmode, int pro, int epi) {
    return sed_conv3x3_bwd_fused_supported_pool(dtype, W, Cinp, Coutp, dzmode, pro, epi, 2);
}

extern "C" int sed_conv3x3_bwd_fused(int dtype, int pro, const void* x, const float* pro_scale, const float* pro_shift, int dzmode,
                                     const void* gsrc, const void* zsrc, const float* scale, const float* shift, const float* ca,
                                     const float* cb, const float* cc, int pool, const void* wpack_t, void* dx, int epi,
                                     const void* zref, const void* cnt, const float* epi_scale, const float* epi_shift,
                                     const float* epi_mean, const float* epi_invstd, float* partial, int nparts, int* flag,
                                     float* dwpack, float* workspace, int B, int H, int W, int Cinp, int Coutp, float* dw, int Cout,
                                     int Cin, void* stream) {
    SED_REQUIRE(sed_conv3x3_bwd_fused_supported_pool(dtype, W, Cinp, Coutp, dzmode, pro, epi, dzmode == SED_DZ_POOL ? pool : 2),
                "covered: bf16; W = 32: 32 -> 64 (DZ_BN, no prologue, STORE / POOLSTATS) or 64 -> 64 (DZ_POOL pool 2, BN+ReLU prologue, "
                "RELUBWD); W = 16 / 8: 64 / 128 -> 128 in the same two forms (DZ_POOL with pool 1 or 2)");
    SED_REQUIRE(B > 0 && H > 0 && x && gsrc && zsrc && ca && cb && cc && wpack_t && dx && workspace, "operands");      // (dwpack == NULL: deferred reduction)
    SED_REQUIRE(pro == SED_PRO_NONE || (pro_scale && pro_shift), "prologue operands");
    SED_REQUIRE(dzmode != SED_DZ_POOL || (scale && shift && (pool == 1 || pool == 2)), "pool-backward operands");
    SED_REQUIRE(epi == SED_EPI_STORE || (zref && epi_scale && epi_shift && epi_mean && epi_invstd && partial && nparts > 0), "epilogue operands");
    SED_REQUIRE(epi != SED_EPI_POOLSTATS || (cnt && flag), "pooled-tensor statistics operands");
    // both covered layers have zref == x (conv2: the ReLU / BN1 reference is the z tensor its prologue reads; conv1: the pooled
    // activation is the convolution's input): the kernel takes the reference from the tile it already holds
    SED_REQUIRE(epi == SED_EPI_STORE || zref == x, "the epilogue reference must be the convolution's input tensor");
    SED_REQUIRE(epi != SED_EPI_RELUBWD || (epi_scale == pro_scale && epi_shift == pro_shift),
                "the ReLU decision of conv2's data gradient uses the prologue's BatchNorm coefficients (same block, BN1)");
    SED_REQUIRE(dw == nullptr || (Cout > 0 && Cin > 0 && Cout <= Coutp && Cin <= Cinp), "unpacked gradient operands");
    SED_REQUIRE((double)H * W * (Cinp > Coutp ? Cinp : Coutp) * 2 < 2147483648.0, "one image (H*W*C elements) must stay below 2 GiB");
    BwdFusedParams p = {};
    p.x = x; p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.gsrc = gsrc; p.zsrc = zsrc; p.scale = scale; p.shift = shift;
    p.ca = ca; p.cb = cb; p.cc = cc; p.wpack_t = wpack_t; p.dx = dx; p.zref = zref; p.cnt = reinterpret_cast<const unsigned char*>(cnt);
    p.epi_scale = epi_scale; p.epi_shift = epi_shift; p.epi_mean = epi_mean; p.epi_invstd = epi_invstd; p.partial = partial;
    p.flag = flag; p.ws = workspace; p.B = B; p.H = H; p.Cinp = Cinp; p.Coutp = Coutp; p.pool = dzmode == SED_DZ_POOL ? pool : 1;
    p.dzmode = dzmode; p.pro = pro; p.epi = epi; p.nparts = nparts;
    const int rc = launch_bwd_fused(p, W, (hipStream_t)stream);
    SED_REQUIRE(rc >= 0, "shape not covered");
    if (rc) return rc;
    SED_LAUNCH_CHECK();
    const size_t n = (size_t)9 * Cinp * Coutp;
    reduce_or_defer(workspace, dwpack, p.nwg, n, dw, Cout, Cin, Cinp, Coutp, (hipStream_t)stream);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_pack_conv_weights_batch(int dtype, const void* desc, int n, int total_blocks, void* stream) {
    SED_REQUIRE(desc && n > 0 && n <= 64 && total_blocks > 0, "descriptor table");
    if (dtype == SED_BF16)
        pack_weight_batch_kernel<bf16_t><<<total_blocks, 256, 0, (hipStream_t)stream>>>((const long long*)desc, n);
    else if (dtype == SED_F32)
        pack_weight_batch_kernel<float><<<total_blocks, 256, 0, (hipStream_t)stream>>>((const long long*)desc, n);
    else if (dtype == SED_F32X3 || dtype == SED_F32H3)
        pack_weight_batch_x3_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>((const long long*)desc, n, dtype == SED_F32H3);
    else
        SED_REQUIRE(false, "bad dtype");
    SED_LAUNCH_CHECK();
    return 0;
}

static void c1_geometry(int Coutp, int* G, int* PPB, int* threads) {
    *G = Coutp / 8;
    *PPB = 256 / *G;
    if (*PPB < 1) *PPB = 1;
    *threads = 256;
}

extern "C" int sed_conv3x3_c1_fwd(int dtype, const float* x, const float* mean, const float* stdv, const float* w,
                                  void* z, float* stats_partial, int B, int H, int W, int Cout, int Coutp,
                                  void* stream) {
    SED_REQUIRE(Coutp % 32 == 0 && Coutp <= 2048 && Cout <= Coutp, "Coutp must be a multiple of 32, <= 2048");
    SED_REQUIRE((mean == nullptr) == (stdv == nullptr), "mean/std must both be given or both NULL");
    int G, PPB, threads;
    c1_geometry(Coutp, &G, &PPB, &threads);
    const int grid = sed_conv_c1_nparts(B, H, W);
    const size_t lds = ((size_t)9 * Coutp + (C1_TR + 2) * (size_t)(W + 2) + (size_t)PPB * 2 * Coutp) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SED_BF16)
        conv_c1_fwd_kernel<bf16_t><<<grid, threads, lds, st>>>(x, mean, stdv, w, (bf16_t*)z, stats_partial, B, H, W, Cout, Coutp, G, PPB);
    else if (dtype == SED_F32)
        conv_c1_fwd_kernel<float><<<grid, threads, lds, st>>>(x, mean, stdv, w, (float*)z, stats_partial, B, H, W, Cout, Coutp, G, PPB);
    else
        SED_REQUIRE(false, "bad dtype");
    SED_LAUNCH_CHECK();
    return 0;
}

static int c1_wgrad_common(int dtype, const float* x, const float* mean, const float* stdv, const void* dz,
                           const void* zsrc, const float* ca, const float* cb, const float* cc, float* dw_partial, int B,
                           int H, int W, int Coutp, void* stream);

extern "C" int sed_conv3x3_c1_wgrad(int dtype, const float* x, const float* mean, const float* stdv, const void* dz,
                                    float* dw_partial, int B, int H, int W, int Coutp, void* stream) {
    return c1_wgrad_common(dtype, x, mean, stdv, dz, nullptr, nullptr, nullptr, nullptr, dw_partial, B, H, W, Coutp, stream);
}

extern "C" int sed_conv3x3_c1_wgrad_fused(int dtype, const float* x, const float* mean, const float* stdv,
                                          const void* g, const void* zsrc, const float* ca, const float* cb,
                                          const float* cc, float* dw_partial, int B, int H, int W, int Coutp,
                                          void* stream) {
    SED_REQUIRE(g && zsrc && ca && cb && cc, "fused dz operands");
    return c1_wgrad_common(dtype, x, mean, stdv, g, zsrc, ca, cb, cc, dw_partial, B, H, W, Coutp, stream);
}

static int c1_wgrad_common(int dtype, const float* x, const float* mean, const float* stdv, const void* dz,
                           const void* zsrc, const float* ca, const float* cb, const float* cc, float* dw_partial, int B,
                           int H, int W, int Coutp, void* stream) {
    SED_REQUIRE(Coutp % 32 == 0 && Coutp <= 2048, "Coutp must be a multiple of 32, <= 2048");
    int G, PPB, threads;
    c1_geometry(Coutp, &G, &PPB, &threads);
    const int grid = sed_conv_c1_nparts(B, H, W);
    const size_t lds = ((C1_TR + 2) * (size_t)(W + 2) + (size_t)PPB * Coutp) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SED_BF16)
        if (zsrc) conv_c1_wgrad_kernel<bf16_t, true><<<grid, threads, lds, st>>>(x, mean, stdv, (const bf16_t*)dz, (const bf16_t*)zsrc, ca, cb, cc, dw_partial, B, H, W, Coutp, G, PPB);
        else conv_c1_wgrad_kernel<bf16_t, false><<<grid, threads, lds, st>>>(x, mean, stdv, (const bf16_t*)dz, nullptr, ca, cb, cc, dw_partial, B, H, W, Coutp, G, PPB);
    else if (dtype == SED_F32)
        if (zsrc) conv_c1_wgrad_kernel<float, true><<<grid, threads, lds, st>>>(x, mean, stdv, (const float*)dz, (const float*)zsrc, ca, cb, cc, dw_partial, B, H, W, Coutp, G, PPB);
        else conv_c1_wgrad_kernel<float, false><<<grid, threads, lds, st>>>(x, mean, stdv, (const float*)dz, nullptr, ca, cb, cc, dw_partial, B, H, W, Coutp, G, PPB);
    else
        SED_REQUIRE(false, "bad dtype");
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_conv_c1_gram_nparts(int B, int H, int W) {
    (void)W;
    // 2048 = 8 resident 256-thread workgroups per CU: the kernel has ~500 cycles of arithmetic per band behind a full
    // memory latency, only more resident workgroups hide it (768 workgroups: 83 us)
    const long long bands = (long long)B * ((H + C1_TR - 1) / C1_TR);
    return (int)(bands < 2048 ? (bands < 1 ? 1 : bands) : 2048);
}

extern "C" int sed_conv3x3_c1_gram(const float* x, const float* mean, const float* stdv, float* gram_partial, int B, int H,
                                   int W, void* stream) {
    SED_REQUIRE((mean == nullptr) == (stdv == nullptr), "mean/std must both be given or both NULL");
    SED_REQUIRE(W >= 1 && (C1_TR + 2) * (W + 2) <= 4 * 256, "W must be <= 100 (one band of input lines is staged by 256 threads)");
    const int grid = sed_conv_c1_gram_nparts(B, H, W); // every row of gram_partial is written (the combine reads nparts rows)
    const size_t lds = ((C1_TR + 2) * (size_t)(W + 2) + 4 * 54) * sizeof(float);
    conv_c1_gram_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(x, mean, stdv, gram_partial, B, H, W);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_conv3x3_c1_wgrad_combine(const float* a_sum, const float* gram_partial, int nparts, const float* w,
                                            const float* ca, const float* cb, const float* cc, float* dwpack, int Cout,
                                            int Coutp, void* stream) {
    SED_REQUIRE(a_sum && gram_partial && w && ca && cb && cc && dwpack && nparts > 0, "operands");
    conv_c1_wgrad_combine_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(a_sum, gram_partial, nparts, w, ca, cb, cc, dwpack, Cout, Coutp);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_conv3x3_c1_wgrad_combine_u(const float* a_sum, const float* gram_partial, int nparts, const float* w,
                                              const float* ca, const float* cb, const float* cc, float* dwpack, int Cout,
                                              int Coutp, float* dw, void* stream) {
    SED_REQUIRE(a_sum && gram_partial && w && ca && cb && cc && dwpack && dw && nparts > 0, "operands");
    conv_c1_wgrad_combine_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(a_sum, gram_partial, nparts, w, ca, cb, cc, dwpack, Cout, Coutp, dw);
    SED_LAUNCH_CHECK();
    return 0;
}

// ---- "C1 mode" entry points: the first ConvBlock without conv1's output in memory (bf16, W = 64, 32 channels) ----
extern "C" int sed_bn_train_finalize_c1(const float* gram_partial, int nparts, double count, const float* w1, const float* gamma,
                                        const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                        float* scale, float* shift, float* mean, float* invstd, int C, int Cp, void* stream) {
    SED_REQUIRE(nparts > 0 && count > 0 && C <= Cp, "bad sizes");
    SED_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running stats must both be given or both NULL");
    bn_train_finalize_c1_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(gram_partial, nparts, count, w1, gamma, beta, running_mean,
                                                                     running_var, momentum, eps, scale, shift, mean, invstd, C, Cp);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_bn_train_finalize_c1_g(const float* gram_partial, int nparts, double count, const float* w1, const float* gamma,
                                          const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                          float* scale, float* shift, float* mean, float* invstd, int C, int Cp, double* gram_sum,
                                          void* stream) {
    SED_REQUIRE(nparts > 0 && count > 0 && C <= Cp && gram_sum, "bad sizes / operands");
    SED_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running stats must both be given or both NULL");
    bn_train_finalize_c1_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(gram_partial, nparts, count, w1, gamma, beta, running_mean,
                                                                     running_var, momentum, eps, scale, shift, mean, invstd, C, Cp, gram_sum);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_c1_bwd_tail(const float* a_partial, int a_nparts, const double* gram_sum, double count, const float* w1,
                               const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta, float* ca,
                               float* cb, float* cc, float* a_sum, float* dwpack, int Cout, int Coutp, float* dw, void* stream) {
    SED_REQUIRE(a_partial && gram_sum && w1 && gamma && mean && invstd && dgamma && dbeta && ca && cb && cc && a_sum && dwpack &&
                a_nparts > 0 && count > 0, "operands");
    SED_REQUIRE(Coutp == 32 && Cout > 0 && Cout <= 32, "covered: 32 (padded) conv1 channels");
    c1_bwd_tail_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(a_partial, a_nparts, gram_sum, count, w1, gamma, mean, invstd, dgamma, dbeta,
                                                            ca, cb, cc, a_sum, dwpack, Cout, dw);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_c1_mode_supported(int dtype, int W, int C1, int Cout2) {
    return dtype == SED_BF16 && W == 64 && C1 == 32 && Cout2 == 32;
}

static int c1_conv_common(ConvParams& p, int W, void* stream) {
    { const char* d = sed_getenv("SED_DBG"); p.dbg = d ? atoi(d) : 0; }
    p.wres = 0;
    p.nparts = sed_conv_nparts(p.B, p.H, W);
    const int rc = launch_conv_pc(p, W, (hipStream_t)stream);
    if (rc < 0) { sed_set_error("C1 mode: shape not covered (needs bf16, W = 64, 32 conv1 channels)"); return 1; }
    if (rc) return rc;
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) { sed_set_error(std::string("C1 mode launch failed: ") + hipGetErrorString(e_)); return 2; }
    return 0;
}

extern "C" int sed_conv3x3_fwd_c1(int dtype, int epi, const float* x1, const float* fmean, const float* fstd, const float* w1,
                                  const float* pro_scale, const float* pro_shift, const void* wpack, void* z, float* partial,
                                  void* relu_mask, int B, int H, int W, int Coutp, void* stream) {
    SED_REQUIRE(dtype == SED_BF16 && x1 && w1 && pro_scale && pro_shift && wpack && z, "operands");
    SED_REQUIRE((fmean == nullptr) == (fstd == nullptr), "mean/std must both be given or both NULL");
    SED_REQUIRE(epi == SED_EPI_STORE || (epi == SED_EPI_STATS && partial), "epilogue");
    ConvParams p = {};
    p.x = nullptr; p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.wpack = wpack; p.z = z; p.partial = partial;
    p.B = B; p.H = H; p.Cinp = 32; p.Coutp = Coutp; p.pro = SED_PRO_C1; p.epi = epi;
    p.c1_x = x1; p.c1_mean = fmean; p.c1_std = fstd; p.c1_w = w1; p.c1_mask = relu_mask;
    return c1_conv_common(p, W, stream);
}

extern "C" int sed_conv3x3_dgrad_c1(int dtype, const void* dz, const void* wpack_t, void* g, const void* relu_mask, float* partial,
                                    int B, int H, int W, int Cinp, void* stream) {
    SED_REQUIRE(dtype == SED_BF16 && dz && wpack_t && g && relu_mask && partial, "operands");
    ConvParams p = {};
    p.x = dz; p.wpack = wpack_t; p.z = g; p.partial = partial;
    p.B = B; p.H = H; p.Cinp = Cinp; p.Coutp = 32; p.pro = SED_PRO_NONE; p.epi = SED_EPI_RELUBWD_C1;
    p.c1_mask = const_cast<void*>(relu_mask);
    return c1_conv_common(p, W, stream);
}

static int wgrad_fused_c1_impl(int dtype, const float* x1, const float* fmean, const float* fstd, const float* w1,
                               const float* pro_scale, const float* pro_shift, const void* gsrc, const void* zsrc,
                               const float* scale, const float* shift, const float* ca, const float* cb,
                               const float* cc, int pool, void* dz_out, float* dwpack, float* workspace, int B, int H,
                               int W, int Coutp, void* stream, float* dw, int Cout, int Cin) {
    SED_REQUIRE(dtype == SED_BF16 && x1 && w1 && pro_scale && pro_shift && gsrc && zsrc && scale && shift && ca && cb && cc,
                "operands");
    Wgrad2Params p = {};
    p.x = nullptr; p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.dz = gsrc; p.zsrc = zsrc; p.scale = scale; p.shift = shift;
    p.ca = ca; p.cb = cb; p.cc = cc; p.dz_out = dz_out; p.ws = workspace;
    p.B = B; p.H = H; p.Cinp = 32; p.Coutp = Coutp; p.pro = SED_PRO_C1; p.pool = pool < 1 ? 1 : pool;
    p.c1_x = x1; p.c1_mean = fmean; p.c1_std = fstd; p.c1_w = w1;
    { const char* d = sed_getenv("SED_DBG"); p.dbg = d ? atoi(d) : 0; }
    const int rc = launch_wgrad3(DZ_POOL, p, W, (hipStream_t)stream);
    if (rc < 0) { sed_set_error("C1 mode weight gradient: shape not covered (needs W = 64, 32 -> 32 channels)"); return 1; }
    if (rc) return rc;
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) { sed_set_error(std::string("C1 mode wgrad launch failed: ") + hipGetErrorString(e_)); return 2; }
    const size_t n = (size_t)9 * 32 * Coutp;
    reduce_or_defer(workspace, dwpack, p.strips, n, dw, Cout, Cin, 32, Coutp, (hipStream_t)stream);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_conv3x3_wgrad_fused_c1(int dtype, const float* x1, const float* fmean, const float* fstd, const float* w1,
                                          const float* pro_scale, const float* pro_shift, const void* gsrc, const void* zsrc,
                                          const float* scale, const float* shift, const float* ca, const float* cb,
                                          const float* cc, int pool, void* dz_out, float* dwpack, float* workspace, int B, int H,
                                          int W, int Coutp, void* stream) {
    return wgrad_fused_c1_impl(dtype, x1, fmean, fstd, w1, pro_scale, pro_shift, gsrc, zsrc, scale, shift, ca, cb, cc, pool, dz_out,
                               dwpack, workspace, B, H, W, Coutp, stream, nullptr, 0, 0);
}

extern "C" int sed_conv3x3_wgrad_fused_c1_u(int dtype, const float* x1, const float* fmean, const float* fstd, const float* w1,
                                            const float* pro_scale, const float* pro_shift, const void* gsrc, const void* zsrc,
                                            const float* scale, const float* shift, const float* ca, const float* cb,
                                            const float* cc, int pool, void* dz_out, float* dwpack, float* workspace, int B, int H,
                                            int W, int Coutp, float* dw, int Cout, int Cin, void* stream) {
    SED_REQUIRE(dw && Cout > 0 && Cin > 0 && Cout <= Coutp && Cin <= 32, "unpacked gradient operands");
    return wgrad_fused_c1_impl(dtype, x1, fmean, fstd, w1, pro_scale, pro_shift, gsrc, zsrc, scale, shift, ca, cb, cc, pool, dz_out,
                               dwpack, workspace, B, H, W, Coutp, stream, dw, Cout, Cin);
}

extern "C" int sed_conv3x3_bwd_fused_c1_supported(int dtype, int W, int Coutp, int pool) {
    if (!(dtype == SED_BF16 && W == 64 && Coutp == 32 && pool == 2)) return 0;
    if (const char* e = sed_getenv("SED_BWD_FUSED_C1")) if (e[0] == '0') return 0;
    return 1;
}

extern "C" int sed_conv3x3_bwd_fused_c1(int dtype, const float* x1, const float* fmean, const float* fstd, const float* w1,
                                        const float* pro_scale, const float* pro_shift, const void* gsrc, const void* zsrc,
                                        const float* scale, const float* shift, const float* ca, const float* cb, const float* cc,
                                        int pool, const void* wpack_t, const void* relu_mask, float* a_partial, float* dwpack,
                                        float* workspace, int B, int H, int W, int Coutp, float* dw, int Cout, int Cin, void* stream) {
    SED_REQUIRE(sed_conv3x3_bwd_fused_c1_supported(dtype, W, Coutp, pool), "covered: bf16, W = 64, 32 -> 32 channels, 2x2 pooling");
    SED_REQUIRE(x1 && w1 && pro_scale && pro_shift && gsrc && zsrc && scale && shift && ca && cb && cc && wpack_t &&
                a_partial && workspace && B > 0 && H > 0, "operands");      // (dwpack == NULL: deferred reduction)
    SED_REQUIRE((fmean == nullptr) == (fstd == nullptr), "mean/std must both be given or both NULL");
    SED_REQUIRE(dw == nullptr || (Cout > 0 && Cin > 0 && Cout <= 32 && Cin <= 32), "unpacked gradient operands");
    SED_REQUIRE((double)H * W * 32 * 2 < 2147483648.0, "one image must stay below 2 GiB");
    int nwg = 0;
    const int rc = launch_bwd_fused_c1(x1, fmean, fstd, w1, pro_scale, pro_shift, gsrc, zsrc, scale, shift, ca, cb, cc, wpack_t, relu_mask,
                                       a_partial, sed_conv_dgrad_c1_nparts(), workspace, B, H, &nwg, (hipStream_t)stream);
    SED_REQUIRE(rc >= 0, "not covered");
    if (rc) return rc;
    SED_LAUNCH_CHECK();
    const size_t n = (size_t)9 * 32 * 32;
    reduce_or_defer(workspace, dwpack, nwg, n, dw, Cout, Cin, 32, 32, (hipStream_t)stream);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_bn_bwd_finalize_c1(const float* partial, int nparts, double count, const float* a_sum, const float* w1,
                                      const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                      float* ca, float* cb, float* cc, int C, int Cp, void* stream) {
    SED_REQUIRE(nparts > 0 && count > 0 && C <= Cp && a_sum && w1, "bad sizes");
    bn_bwd_finalize_c1_kernel<<<Cp, 256, 0, (hipStream_t)stream>>>(partial, nparts, count, a_sum, w1, gamma, mean, invstd, dgamma,
                                                                   dbeta, ca, cb, cc, C, Cp);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_wgrad_reduce(const float* workspace, int nslabs, float* dwpack, float* dw, int Cout, int Cin, int Cinp, int Coutp,
                                void* stream) {
    SED_REQUIRE(workspace && nslabs > 0 && dwpack && Cinp > 0 && Coutp > 0, "operands");
    SED_REQUIRE(dw == nullptr || (Cout > 0 && Cin > 0 && Cout <= Coutp && Cin <= Cinp), "unpacked gradient operands");
    const size_t n = (size_t)9 * Cinp * Coutp;
    wgrad_reduce_kernel<<<cdiv(n, 64), 1024, 0, (hipStream_t)stream>>>(workspace, dwpack, nslabs, n, dw, Cout, Cin, Cinp, Coutp);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_wgrad_reduce_batch(const void* desc, int n, int total_blocks, void* stream) {
    SED_REQUIRE(desc && n > 0 && n <= 64 && total_blocks > 0, "descriptor table");
    wgrad_reduce_batch_kernel<<<total_blocks, 1024, 0, (hipStream_t)stream>>>((const long long*)desc, n);
    SED_LAUNCH_CHECK();
    return 0;
}
