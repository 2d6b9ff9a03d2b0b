// M5's first layer on the matrix pipe (bf16 mode): Conv1d(1, 64, kernel 79, stride 4, padding 39) of
// /root/reference/models/waveform_models.py:13-25 (conv_block1.0), forward and weight gradient.
//
// The fp32 VALU kernels of sed_m5.hip (kept for the fp32 parity mode) ran this layer at 53 TFLOP/s: 4.35 ms forward and
// 4.85 ms weight gradient (+ 1.79 ms for a separate BatchNorm-backward pass) of a 19.5 ms step.  With ONE input channel
// the layer is a K = 79 (padded to 80) contraction:
//   forward   z[c][t]  = sum_k  w[c][k] * x[4t + k - 39]          A = w (64 x 80),  B = im2col patches (80 x positions)
//   wgrad     dW[c][k] = sum_t dz[t][c] * x[4t + k - 39]          A = dz^T (64 x positions), B = patches (positions x 80)
// The patch operand is never materialised: a lane's 8 consecutive taps of one position are 8 consecutive floats of the
// staged input window (16-byte aligned: two ds_read_b128), a lane's 8 consecutive positions of one tap are 8 consecutive
// floats of the window de-interleaved by stride phase.  Both kernels then run at the rate of their one big tensor:
// the forward writes z (2.9 GB at 2880 frames), the weight gradient reads g and z and rebuilds dz = ca*g + cb*z + cc
// (BatchNorm1d backward) on load, so dz is never written.
#include "conv_common.h"

#include <stdlib.h>

namespace {

constexpr int K1 = 79, K1P = 80, S1 = 4, P1 = 39, C1 = 64;
constexpr int TT = 128;                       // outputs per tile (one frame)
constexpr int XWN = S1 * TT + K1P;            // staged input window: xw[i] = x[4*t0 - 39 + i]
constexpr int SP = C1 + 8;                    // staging row pitch (bf16): 64 channels + 16 B pad

// ---- forward -------------------------------------------------------------------------------------------------
// Round 4, "z-free" first block: z1 = conv1(x) is 8x its input (2.9 GB at 2880 frames) and the step wrote it once and read it three
// times (BN + ReLU + MaxPool forward, MaxPool / ReLU backward statistics, weight gradient): 3.5 of its 9.05 ms.  With ONE input
// channel the tile is 10 MFMAs per wave away from the staged input window, so every consumer RECOMPUTES it (bit-identical: the same
// MFMA sequence, rounded to bf16 where the stored tensor was) and z1 is never allocated:
//   MODE 0  store z + BatchNorm statistics (the default; the z-free modes are opt-in, SED_M5_ZFREE=1: measured slower, see below)
//   MODE 1  BatchNorm statistics only
//   MODE 2  relu(scale*z + shift) -> MaxPool1d(4) -> y          (sed_bn_relu_maxpool4_fwd on the tile in LDS)
//   MODE 3  MaxPool / ReLU backward statistics from the pooled dy (sed_maxpool4_relu_bwd with g = NULL on the tile in LDS)
struct M5FwdExtra {
    const float* scale;        // MODE 2, 3
    const float* shift;
    const float* mean;         // MODE 3
    const float* invstd;
    const bf16_t* dy;          // MODE 3: pooled gradient [B/8][L1/4][8][64]
    bf16_t* y;                 // MODE 2: pooled output   [B/8][L1/4][8][64]
};
template <int MODE>
__global__ __launch_bounds__(256) void m5_conv1_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                bf16_t* __restrict__ z, float* __restrict__ partial, int B, int L,
                                                                int L1, int tiles, M5FwdExtra ex) {
    __shared__ __attribute__((aligned(16))) float xw[XWN];
    __shared__ __attribute__((aligned(16))) bf16_t stg[TT * SP];
    static_assert(TT * SP * sizeof(bf16_t) >= 256 * 16 * sizeof(float), "the final reduction buffer overlays the staging image");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, hh = lane >> 5;
    // A fragments for the whole kernel: channel 32*ct + r, taps 16*ks + 8*hh + j
    bf16x8 wa[2][5];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < 5; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int tap = 16 * ks + 8 * hh + j;
                wa[ct][ks][j] = (bf16_t)(tap < K1 ? w[(32 * ct + r) * K1 + tap] : 0.f);
            }
    const int c8 = tid & 7;                   // store pass: this thread's 8 channels
    float S[8], Q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { S[e] = 0.f; Q[e] = 0.f; }
    float sc8[8], sh8[8], mu8[8], is8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc8[e] = MODE >= 2 ? ex.scale[c8 * 8 + e] : 0.f; sh8[e] = MODE >= 2 ? ex.shift[c8 * 8 + e] : 0.f;
        mu8[e] = MODE == 3 ? ex.mean[c8 * 8 + e] : 0.f; is8[e] = MODE == 3 ? ex.invstd[c8 * 8 + e] : 0.f;
    }
    const int Ho = L1 >> 2;                   // pooled length (floor)
    // the input window of the NEXT tile is fetched into registers while this tile computes and stores (the plain
    // load -> LDS -> barrier chain exposed a memory latency per tile)
    constexpr int XPT = (XWN + 255) / 256;
    float xn[XPT];
    auto fetch = [&](int tile) {
        const bool live = tile < B * tiles;
        const int b = live ? tile / tiles : 0, t0 = live ? (tile - b * tiles) * TT : 0;
        const float* __restrict__ xb = x + (size_t)b * L;
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int i = tid + 256 * u, src = S1 * t0 - P1 + i;
            xn[u] = (live && i < XWN && src >= 0 && src < L) ? xb[src] : 0.f;
        }
    };
    fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < B * tiles; tile += gridDim.x) {
        const int b = tile / tiles, t0 = (tile - b * tiles) * TT;
        __syncthreads();                      // the previous tile's store pass has read stg
#pragma unroll
        for (int u = 0; u < XPT; ++u)
            if (tid + 256 * u < XWN) xw[tid + 256 * u] = xn[u];
        __syncthreads();
        fetch(tile + gridDim.x);
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
        const float* px = xw + S1 * (32 * wv + r) + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(px + 16 * ks);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(px + 16 * ks + 4);
            bf16x8 pb;
#pragma unroll
            for (int j = 0; j < 4; ++j) { pb[j] = (bf16_t)v0[j]; pb[4 + j] = (bf16_t)v1[j]; }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[ct] = mfma(wa[ct][ks], pb, acc[ct]);       // D[channel][position]
        }
        // lane = position 32*wv + r; register i of tile ct = channel 32*ct + (i&3) + 8*(i>>2) + 4*hh
        bf16_t* srow = stg + (32 * wv + r) * SP + 4 * hh;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float v4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = acc[ct][4 * g4 + e];
                store4<bf16_t>(srow + 32 * ct + 8 * g4, v4);
            }
        __syncthreads();
        if constexpr (MODE <= 1) {
            // whole-line stores (one output step of one frame = 128 contiguous bytes), statistics of the values as stored
#pragma unroll
            for (int u = 0; u < TT * 8 / 256; ++u) {
                const int row = (tid >> 3) + 32 * u, t = t0 + row;
                if (t < L1) {
                    const bf16x8 raw = *reinterpret_cast<const bf16x8*>(stg + row * SP + c8 * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float f = (float)raw[e]; S[e] += f; Q[e] = fmaf(f, f, Q[e]); }
                    if constexpr (MODE == 0)
                        *reinterpret_cast<bf16x8*>(z + ((((size_t)(b >> 3) * L1 + t) * 8 + (b & 7)) * C1 + c8 * 8)) = raw;
                }
            }
        } else {
            // the thread owns pooling window (tid >> 3) of the tile: steps 4*(tid >> 3) .. + 3 (a tile starts on a window boundary)
            static_assert(TT * 8 / 256 == 4, "one pooling window of 4 steps per thread");
            if (MODE == 2 && z != nullptr) {             // two-pass forward with z kept for the backward (statistics came from MODE 1)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = 4 * (tid >> 3) + i, t = t0 + row;
                    if (t < L1)
                        *reinterpret_cast<bf16x8*>(z + ((((size_t)(b >> 3) * L1 + t) * 8 + (b & 7)) * C1 + c8 * 8)) =
                            *reinterpret_cast<const bf16x8*>(stg + row * SP + c8 * 8);
                }
            }
            const int ho = (t0 >> 2) + (tid >> 3);
            if (ho < Ho) {                                   // (steps past the pooling floor: no output / zero gradient)
                float v[4][8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16x8 raw = *reinterpret_cast<const bf16x8*>(stg + (4 * (tid >> 3) + i) * SP + c8 * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[i][e] = (float)raw[e];
                }
                const size_t po = (((size_t)(b >> 3) * Ho + ho) * 8 + (b & 7)) * C1 + c8 * 8;
                if constexpr (MODE == 2) {
                    float m[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) m[e] = 0.f;             // ReLU output is >= 0
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], fmaf(v[i][e], sc8[e], sh8[e]));
                    store8<bf16_t>(ex.y + po, m);
                } else {
                    const bf16x8 draw = *reinterpret_cast<const bf16x8*>(ex.dy + po);
                    float best[8];
                    int am[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { best[e] = fmaxf(0.f, fmaf(v[0][e], sc8[e], sh8[e])); am[e] = 0; }
#pragma unroll
                    for (int i = 1; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float a = fmaxf(0.f, fmaf(v[i][e], sc8[e], sh8[e]));
                            if (a > best[e]) { best[e] = a; am[e] = i; }     // strict: ties keep the first (torch max_pool1d)
                        }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float o = (am[e] == i && best[e] > 0.f) ? (float)draw[e] : 0.f;
                            S[e] += o;
                            Q[e] = fmaf(o, (v[i][e] - mu8[e]) * is8[e], Q[e]);
                        }
                }
            }
        }
    }
    if (partial) {       // fixed-order block reduction over the 32 row lanes of each channel group
        __syncthreads();
        float* r2 = reinterpret_cast<float*>(stg);          // [256][16]
#pragma unroll
        for (int e = 0; e < 8; ++e) { r2[tid * 16 + e] = S[e]; r2[tid * 16 + 8 + e] = Q[e]; }
        __syncthreads();
        if (tid < 2 * C1) {
            const int stat = tid / C1, c = tid % C1;
            float tot = 0.f;
            for (int q = 0; q < 32; ++q) tot += r2[(q * 8 + (c >> 3)) * 16 + stat * 8 + (c & 7)];
            partial[((size_t)blockIdx.x * 2 + stat) * C1 + c] = tot;
        }
    }
}

// ---- weight gradient with the BatchNorm backward produced on load ---------------------------------------------------
// dw_partial[block][k][c] = sum over the block's tiles of dz[b, t, c] * x[b][4t + k - 39],  dz = ca*g + cb*z + cc
constexpr int PHN = TT + K1P / 4 + 1;         // words per stride phase: ph[p][m] = xwin[4m + p]
// pitch of a phase row: = 8 mod 32, so the four phases' 8-word runs that one ds_read_b32 / ds_write_b32 group touches (lane = tap:
// phase = tap & 3, word = tap >> 2) fall on 32 different banks; at the natural 149 the run of phase 3 sat on the run of phase 0 (2-way on
// every patch read -- round 4, SQ_LDS_BANK_CONFLICT = 47 % of the weight-gradient kernel's LDS-array cycles).  SED_M5_LDSPAD=0: A/B.
#ifndef SED_M5_LDSPAD
#define SED_M5_LDSPAD 1
#endif
constexpr int PHL = SED_M5_LDSPAD ? 168 : PHN;
static_assert(PHL >= PHN && (!SED_M5_LDSPAD || PHL % 32 == 8), "phase pitch");
// the two 32-channel slabs of the dz tile sit 64 B apart modulo the 128-byte store row: the 8 lanes of a ds_write_b128 group write one
// position's 128 B, half to each slab
constexpr int DZSL = TT * 32 + (SED_M5_LDSPAD ? 32 : 0);

// POOLG: g itself is rebuilt on load as well -- the MaxPool1d(4) + ReLU backward of sed_maxpool4_relu_bwd (dy scattered to the
// FIRST arg-max of relu(scale*z + shift) over each window of 4 when that maximum is > 0): the thread then owns the four
// consecutive steps of one window, reads z and the pooled dy, and neither g nor dz ever exist in memory.
// RECOMP (round 4, z-free first block): z is not read either -- the tile is recomputed from the staged input window with the forward's
// MFMA sequence (w1 = conv1's weights), staged in LDS as bf16 exactly as the forward would have stored it.
// STATSG (round 4, the algebraic backward of the first block): the contraction runs on g itself (G1[tap][c] = sum_t g[t][c] *
// x[4t + tap - 39]) and the same pass accumulates the pool / ReLU backward statistics (sum g, sum g*xhat: sed_maxpool4_relu_bwd with
// g = NULL) -- one read of z instead of two.  With ONE input channel dW1 = ca*G1 + cb*(w1 . Gram) + cc*Sp follows from the Gram
// matrix / sums of the input patches (m5_conv1_gram_kernel, sed_m5_conv1_wgrad_combine): dz is never formed.
template <bool POOLG, bool RECOMP = false, bool STATSG = false>
__global__ __launch_bounds__(256) void m5_conv1_wgrad_mfma_kernel(const float* __restrict__ x, const bf16_t* __restrict__ g,
                                                                  const bf16_t* __restrict__ zsrc, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, const float* __restrict__ ca,
                                                                  const float* __restrict__ cb, const float* __restrict__ cc,
                                                                  float* __restrict__ partial, int B, int L, int L1, int tiles,
                                                                  const float* __restrict__ w1 = nullptr, const float* __restrict__ mean = nullptr,
                                                                  const float* __restrict__ invstd = nullptr, float* __restrict__ spart = nullptr) {
    static_assert(!RECOMP || POOLG, "the recomputing form is the pooled one");
    static_assert(!STATSG || (POOLG && !RECOMP), "the statistics form reads z and the pooled dy");
    __shared__ __attribute__((aligned(16))) bf16_t dzs[2 * DZSL];      // [channel tile][position][32]
    __shared__ float ph[4 * PHL];
    __shared__ __attribute__((aligned(16))) float xwl[RECOMP ? XWN : 4];             // RECOMP: the linear input window of the forward
    __shared__ __attribute__((aligned(16))) bf16_t zst[RECOMP ? TT * SP : 8];        // RECOMP: the recomputed z tile [position][SP]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int c8 = tid & 7;                   // dz production: this thread's 8 channels
    float a8[8], b8[8], k8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a8[e] = STATSG ? mean[c8 * 8 + e] : ca[c8 * 8 + e];          // STATSG: (mean, invstd) of the statistics, no coefficients yet
        b8[e] = STATSG ? invstd[c8 * 8 + e] : cb[c8 * 8 + e];
        k8[e] = STATSG ? 0.f : cc[c8 * 8 + e];
    }
    float S8[8], Q8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { S8[e] = 0.f; Q8[e] = 0.f; }
    float sc8[8], sh8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc8[e] = POOLG ? scale[c8 * 8 + e] : 0.f; sh8[e] = POOLG ? shift[c8 * 8 + e] : 0.f; }
    const int Ho = L1 >> 2;                   // pooled length (floor)
    // transpose-read offsets (as sed_wgrad.hip): the lane supplies k-row 8*hh + qq (+4 for the second half), channels 16*gbit + 4*pp ..
    int offT[2];
    {
        const int i16 = lane & 15, gbit = (lane >> 4) & 1, qq = i16 >> 2, pp = i16 & 3, ch = 16 * gbit + 4 * pp;
#pragma unroll
        for (int half = 0; half < 2; ++half) offT[half] = (8 * hh + qq + 4 * half) * 32 + ch;
    }
    // patch operand: lane = tap 32*tt + r (taps >= 80 are zero rows), 8 consecutive positions 8*hh + j
    int poff[3];
    unsigned pkeep[3];
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
        const int tap = 32 * tt + r, tc = tap < K1P ? tap : 0;
        poff[tt] = (tc & 3) * PHL + (tc >> 2) + 8 * hh;
        pkeep[tt] = tap < K1 ? 0xFFFFFFFFu : 0u;          // (tap 79 is the zero pad of K1P, taps 80..95 the unused columns)
    }
    f32x16 acc[2][3];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ct][tt][i] = 0.f;
    bf16x8 wa[RECOMP ? 2 : 1][RECOMP ? 5 : 1];          // RECOMP: the forward's A fragments (channel 32*ct + r, taps 16*ks + 8*hh + j)
    if constexpr (RECOMP) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int ks = 0; ks < 5; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int tap = 16 * ks + 8 * hh + j;
                    wa[ct][ks][j] = (bf16_t)(tap < K1 ? w1[(32 * ct + r) * K1 + tap] : 0.f);
                }
    }

    // operands of the NEXT tile are fetched into registers while this tile's MFMAs run
    constexpr int XPT = (4 * PHN + 255) / 256, NIT = TT * 8 / 256;
    float xn[XPT];
    bf16x8 gn[NIT], zn[NIT];
    auto fetch = [&](int tile) {
        const bool live = tile < B * tiles;
        const int b = live ? tile / tiles : 0, t0 = live ? (tile - b * tiles) * TT : 0;
        const float* __restrict__ xb = x + (size_t)b * L;
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int i = tid + 256 * u, src = S1 * t0 - P1 + i;
            xn[u] = (live && i < 4 * PHN && src >= 0 && src < L) ? xb[src] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int row = POOLG ? 4 * (tid >> 3) + u : (tid >> 3) + 32 * u, t = t0 + row;
            const bool ok = live && t < L1;
            const size_t o = ok ? (((size_t)(b >> 3) * L1 + t) * 8 + (b & 7)) * C1 + c8 * 8 : 0;
            if (!POOLG) gn[u] = *reinterpret_cast<const bf16x8*>(g + o);          // (rows past the frame re-read element 0 and are zeroed below)
            if constexpr (!RECOMP) zn[u] = *reinterpret_cast<const bf16x8*>(zsrc + o);
        }
        if (POOLG) {      // the window's pooled gradient (g = the pooled dy here); windows past the pooling floor get 0 below
            const int ho = (t0 >> 2) + (tid >> 3);
            const bool ok = live && ho < Ho;
            const size_t o = ok ? (((size_t)(b >> 3) * Ho + ho) * 8 + (b & 7)) * C1 + c8 * 8 : 0;
            gn[0] = *reinterpret_cast<const bf16x8*>(g + o);
        }
    };
    fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < B * tiles; tile += gridDim.x) {
        const int b = tile / tiles, t0 = (tile - b * tiles) * TT;
        (void)b;
        __syncthreads();                      // the previous tile's MFMA reads are done
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int i = tid + 256 * u;
            if (i < 4 * PHN) ph[(i & 3) * PHL + (i >> 2)] = xn[u];
            if constexpr (RECOMP) if (i < XWN) xwl[i] = xn[u];
        }
        if constexpr (RECOMP) {
            // the forward of this tile (m5_conv1_fwd_mfma_kernel): wave wv, positions 32*wv + r, both channel tiles; staged as bf16
            __syncthreads();
            f32x16 fz[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int i = 0; i < 16; ++i) fz[ct][i] = 0.f;
            const float* px = xwl + S1 * (32 * wv + r) + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < 5; ++ks) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(px + 16 * ks);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(px + 16 * ks + 4);
                bf16x8 pb;
#pragma unroll
                for (int j = 0; j < 4; ++j) { pb[j] = (bf16_t)v0[j]; pb[4 + j] = (bf16_t)v1[j]; }
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) fz[ct] = mfma(wa[ct][ks], pb, fz[ct]);
            }
            bf16_t* srow = zst + (32 * wv + r) * SP + 4 * hh;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float v4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = fz[ct][4 * g4 + e];
                    store4<bf16_t>(srow + 32 * ct + 8 * g4, v4);
                }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NIT; ++u) zn[u] = *reinterpret_cast<const bf16x8*>(zst + (4 * (tid >> 3) + u) * SP + c8 * 8);
        }
        if constexpr (POOLG) {
            static_assert(TT * 8 / 256 == 4, "a thread owns one pooling window of 4 steps");
            const bool win_ok = (t0 >> 2) + (tid >> 3) < Ho;
            float best[8];
            int am[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { best[e] = fmaxf(0.f, fmaf((float)zn[0][e], sc8[e], sh8[e])); am[e] = 0; }
#pragma unroll
            for (int i = 1; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = fmaxf(0.f, fmaf((float)zn[i][e], sc8[e], sh8[e]));
                    if (a > best[e]) { best[e] = a; am[e] = i; }      // strict: ties keep the first (torch max_pool1d)
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 4 * (tid >> 3) + i, t = t0 + row;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gg = (win_ok && am[e] == i && best[e] > 0.f) ? (float)gn[0][e] : 0.f;
                    if constexpr (STATSG) {          // (gg = 0 for steps past the frame: their window is past the pooling floor)
                        v[e] = gg;
                        S8[e] += gg;
                        Q8[e] = fmaf(gg, ((float)zn[i][e] - a8[e]) * b8[e], Q8[e]);
                    } else {
                        v[e] = t < L1 ? fmaf(a8[e], gg, fmaf(b8[e], (float)zn[i][e], k8[e])) : 0.f;
                    }
                }
                store8<bf16_t>(dzs + (c8 >> 2) * DZSL + row * 32 + (c8 & 3) * 8, v);
            }
        } else {
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int row = (tid >> 3) + 32 * u, t = t0 + row;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    v[e] = t < L1 ? fmaf(a8[e], (float)gn[u][e], fmaf(b8[e], (float)zn[u][e], k8[e])) : 0.f;
                store8<bf16_t>(dzs + (c8 >> 2) * DZSL + row * 32 + (c8 & 3) * 8, v);
            }
        }
        __syncthreads();
        fetch(tile + gridDim.x);
        // wave wv contracts positions 32*wv .. 32*wv + 31 (two k-steps of 16) into all six output tiles
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int p0 = 32 * wv + 16 * kk;
            bf16x8 af[2], pf[3];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                af[ct] = join_tr(ds_read_tr16_b64(dzs + ct * DZSL + p0 * 32 + offT[0]),
                                 ds_read_tr16_b64(dzs + ct * DZSL + p0 * 32 + offT[1]));
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
                const float* pp = ph + poff[tt] + p0;
                bf16x8 t8;
#pragma unroll
                for (int j = 0; j < 8; ++j) t8[j] = (bf16_t)pp[j];
                u32x4 tw = __builtin_bit_cast(u32x4, t8);
#pragma unroll
                for (int e = 0; e < 4; ++e) tw[e] &= pkeep[tt];
                pf[tt] = __builtin_bit_cast(bf16x8, tw);
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int tt = 0; tt < 3; ++tt) acc[ct][tt] = mfma(af[ct], pf[tt], acc[ct][tt]);     // D[channel][tap]
        }
    }
    if constexpr (STATSG) {      // statistics partial of this workgroup: fixed-order sum over the 32 window lanes of each channel group
        __syncthreads();
        float* r2 = reinterpret_cast<float*>(dzs);          // [256][16]
#pragma unroll
        for (int e = 0; e < 8; ++e) { r2[tid * 16 + e] = S8[e]; r2[tid * 16 + 8 + e] = Q8[e]; }
        __syncthreads();
        if (tid < 2 * C1) {
            const int stat = tid / C1, c = tid % C1;
            float tot = 0.f;
            for (int q = 0; q < 32; ++q) tot += r2[(q * 8 + (c >> 3)) * 16 + stat * 8 + (c & 7)];
            spart[((size_t)blockIdx.x * 2 + stat) * C1 + c] = tot;
        }
    }
    // per-workgroup partial: fixed-order sum over the four waves, one output tile at a time through LDS
    float* red = reinterpret_cast<float*>(dzs);            // [4][16][64] floats = 16 KB
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; ++i) red[(wv * 16 + i) * 64 + lane] = acc[ct][tt][i];
            __syncthreads();
            // lane l of a wave holds tap 32*tt + (l & 31), register i = channel 32*ct + (i&3) + 8*(i>>2) + 4*(l>>5)
            for (int q = tid; q < 16 * 64; q += 256) {
                const int i = q >> 6, l = q & 63;
                const int tap = 32 * tt + (l & 31), ch = 32 * ct + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
                if (tap < K1P) {
                    float tot = 0.f;
#pragma unroll
                    for (int w4 = 0; w4 < 4; ++w4) tot += red[(w4 * 16 + i) * 64 + l];
                    partial[((size_t)blockIdx.x * K1P + tap) * C1 + ch] = tot;
                }
            }
        }
}


// ---- Gram statistics of the first layer's input patches (round 4) ----------------------------------------------------------------
// G[k'][k] = sum over (frame, step t) of xb[4t + k' - 39] * xb[4t + k - 39], Sp[k] = sum of xb[4t + k - 39]  (xb = the bf16-rounded
// input the matrix pipe sees; taps >= 79 are zero).  One fragment of the weight gradient's patch operand serves as A AND B:
// acc[i][j] += P_i^T P_j for the six upper-triangle 32x32 tiles of the 96 x 96 padded matrix.
// partial [grid][GRAMN]: six tiles in D layout order (lane-major) + 96 sums, reduced by sed_sum_partials, unpacked by the combine kernel.
constexpr int GRAMN = 6 * 16 * 64 + 96;
__global__ __launch_bounds__(256) void m5_conv1_gram_kernel(const float* __restrict__ x, float* __restrict__ partial, int B, int L, int L1,
                                                            int tiles) {
    __shared__ float ph[4 * PHL];
    __shared__ float red[4 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, hh = lane >> 5;
    int poff[3];
    unsigned pkeep[3];
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
        const int tap = 32 * tt + r, tc = tap < K1P ? tap : 0;
        poff[tt] = (tc & 3) * PHL + (tc >> 2) + 8 * hh;
        pkeep[tt] = tap < K1 ? 0xFFFFFFFFu : 0u;
    }
    f32x16 acc[6];
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    float sp[3] = {0.f, 0.f, 0.f};
    constexpr int XPT = (4 * PHN + 255) / 256;
    float xn[XPT];
    auto fetch = [&](int tile) {
        const bool live = tile < B * tiles;
        const int b = live ? tile / tiles : 0, t0 = live ? (tile - b * tiles) * TT : 0;
        const float* __restrict__ xb = x + (size_t)b * L;
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int i = tid + 256 * u, src = S1 * t0 - P1 + i;
            xn[u] = (live && i < 4 * PHN && src >= 0 && src < L) ? xb[src] : 0.f;
        }
    };
    fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < B * tiles; tile += gridDim.x) {
        const int b = tile / tiles, t0 = (tile - b * tiles) * TT;
        (void)b;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < XPT; ++u) {
            const int i = tid + 256 * u;
            if (i < 4 * PHN) ph[(i & 3) * PHL + (i >> 2)] = xn[u];
        }
        __syncthreads();
        fetch(tile + gridDim.x);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int p0 = 32 * wv + 16 * kk;
            bf16x8 pf[3];
#pragma unroll
            for (int tt = 0; tt < 3; ++tt) {
                const float* pp = ph + poff[tt] + p0;
                bf16x8 t8;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool ok = t0 + p0 + 8 * hh + j < L1;            // steps past the frame's last output: not part of any sum
                    t8[j] = (bf16_t)(ok ? pp[j] : 0.f);
                }
                u32x4 tw = __builtin_bit_cast(u32x4, t8);
#pragma unroll
                for (int e = 0; e < 4; ++e) tw[e] &= pkeep[tt];
                pf[tt] = __builtin_bit_cast(bf16x8, tw);
#pragma unroll
                for (int j = 0; j < 8; ++j) sp[tt] += (float)pf[tt][j];
            }
            int q = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = i; j < 3; ++j) { acc[q] = mfma(pf[i], pf[j], acc[q]); ++q; }      // D[tap 32i + ..][tap 32j + ..]
        }
    }
    float* out = partial + (size_t)blockIdx.x * GRAMN;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(wv * 16 + i) * 64 + lane] = acc[q][i];
        __syncthreads();
        for (int e = tid; e < 16 * 64; e += 256) {
            float tot = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) tot += red[w4 * 16 * 64 + e];
            out[q * 16 * 64 + e] = tot;
        }
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) red[(tt * 4 + wv) * 64 + lane] = sp[tt];
    __syncthreads();
    if (tid < 96) {
        const int tt = tid >> 5, rr = tid & 31;
        float tot = 0.f;
        for (int w4 = 0; w4 < 4; ++w4) tot += red[(tt * 4 + w4) * 64 + rr] + red[(tt * 4 + w4) * 64 + 32 + rr];
        out[6 * 16 * 64 + tid] = tot;
    }
}

// dW1[c][k] = ca[c]*G1[k][c] + cb[c] * sum_k' bf16(w1[c][k']) * G[k'][k] + cc[c]*Sp[k]   (one thread per (c, k))
__global__ __launch_bounds__(256) void m5_conv1_wgrad_combine_kernel(const float* __restrict__ g1, const float* __restrict__ gram,
                                                                     const float* __restrict__ w1, const float* __restrict__ ca,
                                                                     const float* __restrict__ cb, const float* __restrict__ cc,
                                                                     float* __restrict__ dw) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= C1 * K1) return;
    const int c = idx / K1, k = idx - c * K1;
    // G[k'][k] from the packed upper-triangle tiles: tile (i, j), i <= j, element (m, n): lane = n + 32*((m >> 2) & 1), register
    // (m & 3) + 4*(m >> 3)  [D layout of the 32x32 MFMA: row m = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5), column n = lane & 31]
    auto G = [&](int a, int b2) -> float {
        int kp = a, kk = b2;
        if ((kp >> 5) > (kk >> 5)) { const int t = kp; kp = kk; kk = t; }        // symmetric: read the stored triangle
        const int i = kp >> 5, j = kk >> 5, m = kp & 31, n = kk & 31;
        const int q = i == 0 ? j : (i == 1 ? 2 + j : 5);
        const int reg = (m & 3) + 4 * (m >> 3), ln = n + 32 * ((m >> 2) & 1);
        return gram[(q * 16 + reg) * 64 + ln];
    };
    double t = 0.0;
    for (int kp = 0; kp < K1; ++kp) t += (double)(float)(bf16_t)w1[c * K1 + kp] * (double)G(kp, k);
    dw[idx] = (float)((double)ca[c] * (double)g1[k * C1 + c] + (double)cb[c] * t + (double)cc[c] * (double)gram[6 * 16 * 64 + k]);
}

}  // namespace

extern "C" int sed_m5_conv1_len(int L);
extern "C" int sed_m5_conv1_nparts(int B, int L);

// bf16 forward on the matrix pipe; returns -1 when disabled (SED_M5_MFMA=0) so that the caller takes the VALU kernel
int launch_m5_conv1_fwd_mfma(const float* x, const float* w, void* z, float* stats_partial, int B, int L, hipStream_t st) {
    if (const char* e = sed_getenv("SED_M5_MFMA")) if (e[0] == '0') return -1;
    const int L1 = sed_m5_conv1_len(L), tiles = cdiv(L1, TT);
    const int grid = sed_m5_conv1_nparts(B, L);
    m5_conv1_fwd_mfma_kernel<0><<<grid, 256, 0, st>>>(x, w, (bf16_t*)z, stats_partial, B, L, L1, tiles, M5FwdExtra{});
    return 0;
}

// ---- the algebraic backward of the first block (round 4) --------------------------------------------------------------------------
extern "C" int sed_m5_alg_supported(int dtype) {
    if (dtype != SED_BF16) return 0;
    if (const char* e = sed_getenv("SED_M5_MFMA")) if (e[0] == '0') return 0;
    // opt-in (SED_M5_ALG=1): parity-green and measured SLOWER as built (round 4, 2880 frames: step 9.49 against 8.70 ms).  The one-pass
    // statistics + G1 kernel does pay (1.68 ms against 0.74 + 1.24 ms of the statistics pass and the dz-forming weight gradient), but
    // the Gram kernel costs 0.92 ms: building the patch fragments from the phase-deinterleaved window (24 scalar LDS reads, 12
    // conversions, masks per 16 steps) is what the weight gradient's own time is made of, and the Gram pays it a second time.
    if (const char* e = sed_getenv("SED_M5_ALG")) return e[0] == '1';
    return 0;
}
extern "C" size_t sed_m5_conv1_gram_floats(void) { return (size_t)GRAMN; }
extern "C" int sed_m5_conv1_gram(const float* x, float* gram_partial, int B, int L, void* stream) {
    SED_REQUIRE(B > 0 && x && gram_partial, "operands");
    const int L1 = sed_m5_conv1_len(L), tiles = cdiv(L1, TT);
    m5_conv1_gram_kernel<<<sed_m5_conv1_nparts(B, L), 256, 0, (hipStream_t)stream>>>(x, gram_partial, B, L, L1, tiles);
    SED_LAUNCH_CHECK();
    return 0;
}
extern "C" int sed_m5_conv1_bwd_stats_g1(int dtype, const float* x, const void* dy, const void* zsrc, const float* scale, const float* shift,
                                         const float* mean, const float* invstd, float* stats_partial, float* g1_partial, int B, int L,
                                         void* stream) {
    SED_REQUIRE(dtype == SED_BF16 && B > 0 && B % 8 == 0, "bf16, batch a multiple of 8");
    SED_REQUIRE(x && dy && zsrc && scale && shift && mean && invstd && stats_partial && g1_partial, "operands");
    const int L1 = sed_m5_conv1_len(L), tiles = cdiv(L1, TT);
    m5_conv1_wgrad_mfma_kernel<true, false, true><<<sed_m5_conv1_nparts(B, L), 256, 0, (hipStream_t)stream>>>(
        x, (const bf16_t*)dy, (const bf16_t*)zsrc, scale, shift, nullptr, nullptr, nullptr, g1_partial, B, L, L1, tiles, nullptr, mean, invstd,
        stats_partial);
    SED_LAUNCH_CHECK();
    return 0;
}
extern "C" int sed_m5_conv1_wgrad_combine(const float* g1, const float* gram, const float* w, const float* ca, const float* cb,
                                          const float* cc, float* dw, void* stream) {
    SED_REQUIRE(g1 && gram && w && ca && cb && cc && dw, "operands");
    m5_conv1_wgrad_combine_kernel<<<cdiv(C1 * K1, 256), 256, 0, (hipStream_t)stream>>>(g1, gram, w, ca, cb, cc, dw);
    SED_LAUNCH_CHECK();
    return 0;
}

// ---- the z-free first block (round 4): the three consumers of z1 recompute it from the input --------------------------------
// two-pass forward of the first block (default for bf16): statistics pass without z, then conv + BN + ReLU + MaxPool in one launch that
// also stores z for the backward -- the separate sed_bn_relu_maxpool4_fwd pass over z (2.9 GB read) is gone.  SED_M5_FWD2=0: round-3 order.
extern "C" int sed_m5_fwd2_supported(int dtype) {
    if (dtype != SED_BF16) return 0;
    if (const char* e = sed_getenv("SED_M5_MFMA")) if (e[0] == '0') return 0;
    if (const char* e = sed_getenv("SED_M5_FWD2")) if (e[0] == '0') return 0;
    return 1;
}
extern "C" int sed_m5_zfree_supported(int dtype) {
    if (dtype != SED_BF16) return 0;
    if (const char* e = sed_getenv("SED_M5_MFMA")) if (e[0] == '0') return 0;
    // opt-in (SED_M5_ZFREE=1): parity-green, the same values as the stored-z path (block 1 gradients to fp32 rounding), but SLOWER as built (round 4, 2880 frames: step 9.73
    // against 8.80 ms) -- the forward half pays (statistics 0.525 + conv/BN/ReLU/pool 0.533 ms against 0.766 + 0.766), the backward
    // half does not: the statistics pass 0.94 against 0.74 ms and the recomputing weight gradient 2.15 against 1.24 ms (a dependent
    // stage -> MFMA -> LDS -> read chain with three barriers per tile at 199 registers = two workgroups per CU instead of four).
    if (const char* e = sed_getenv("SED_M5_ZFREE")) return e[0] == '1';
    return 0;
}
extern "C" int sed_m5_conv1_stats(int dtype, const float* x, const float* w, float* stats_partial, int B, int L, void* stream) {
    SED_REQUIRE(dtype == SED_BF16 && B > 0 && B % 8 == 0 && x && w && stats_partial, "bf16, batch a multiple of 8, operands");
    const int L1 = sed_m5_conv1_len(L), tiles = cdiv(L1, TT);
    m5_conv1_fwd_mfma_kernel<1><<<sed_m5_conv1_nparts(B, L), 256, 0, (hipStream_t)stream>>>(x, w, nullptr, stats_partial, B, L, L1, tiles, M5FwdExtra{});
    SED_LAUNCH_CHECK();
    return 0;
}
extern "C" int sed_m5_conv1_bn_relu_pool_fwd(int dtype, const float* x, const float* w, const float* scale, const float* shift, void* y,
                                             void* z_out, int B, int L, void* stream) {
    SED_REQUIRE(dtype == SED_BF16 && B > 0 && B % 8 == 0 && x && w && scale && shift && y, "bf16, batch a multiple of 8, operands");
    const int L1 = sed_m5_conv1_len(L), tiles = cdiv(L1, TT);
    M5FwdExtra ex{};
    ex.scale = scale; ex.shift = shift; ex.y = (bf16_t*)y;
    m5_conv1_fwd_mfma_kernel<2><<<sed_m5_conv1_nparts(B, L), 256, 0, (hipStream_t)stream>>>(x, w, (bf16_t*)z_out, nullptr, B, L, L1, tiles, ex);
    SED_LAUNCH_CHECK();
    return 0;
}
extern "C" int sed_m5_conv1_pool_bwd_stats(int dtype, const float* x, const float* w, const void* dy, const float* scale, const float* shift,
                                           const float* mean, const float* invstd, float* partial, int B, int L, void* stream) {
    SED_REQUIRE(dtype == SED_BF16 && B > 0 && B % 8 == 0 && x && w && dy && scale && shift && mean && invstd && partial,
                "bf16, batch a multiple of 8, operands");
    const int L1 = sed_m5_conv1_len(L), tiles = cdiv(L1, TT);
    M5FwdExtra ex{};
    ex.scale = scale; ex.shift = shift; ex.mean = mean; ex.invstd = invstd; ex.dy = (const bf16_t*)dy;
    m5_conv1_fwd_mfma_kernel<3><<<sed_m5_conv1_nparts(B, L), 256, 0, (hipStream_t)stream>>>(x, w, nullptr, partial, B, L, L1, tiles, ex);
    SED_LAUNCH_CHECK();
    return 0;
}
extern "C" int sed_m5_conv1_wgrad_fused_pool_x(int dtype, const float* x, const float* w, const void* dy, const float* scale,
                                               const float* shift, const float* ca, const float* cb, const float* cc, float* dw_partial,
                                               int B, int L, void* stream) {
    SED_REQUIRE(dtype == SED_BF16 && B > 0 && B % 8 == 0, "bf16, batch a multiple of 8");
    SED_REQUIRE(x && w && dy && scale && shift && ca && cb && cc && dw_partial, "operands");
    const int L1 = sed_m5_conv1_len(L), tiles = cdiv(L1, TT);
    m5_conv1_wgrad_mfma_kernel<true, true><<<sed_m5_conv1_nparts(B, L), 256, 0, (hipStream_t)stream>>>(
        x, (const bf16_t*)dy, nullptr, scale, shift, ca, cb, cc, dw_partial, B, L, L1, tiles, w);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_m5_conv1_wgrad_fused(int dtype, const float* x, const void* g, const void* zsrc, const float* ca,
                                        const float* cb, const float* cc, float* dw_partial, int B, int L, void* stream) {
    SED_REQUIRE(dtype == SED_BF16, "covered: bf16 (the fp32 parity mode runs sed_bn_bwd_apply + sed_m5_conv1_wgrad)");
    SED_REQUIRE(B > 0 && B % 8 == 0, "the interleaved layout needs a batch that is a multiple of 8");
    SED_REQUIRE(x && g && zsrc && ca && cb && cc && dw_partial, "operands");
    const int L1 = sed_m5_conv1_len(L), tiles = cdiv(L1, TT);
    const int grid = sed_m5_conv1_nparts(B, L);
    m5_conv1_wgrad_mfma_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(x, (const bf16_t*)g, (const bf16_t*)zsrc, nullptr, nullptr, ca,
                                                                            cb, cc, dw_partial, B, L, L1, tiles);
    SED_LAUNCH_CHECK();
    return 0;
}

// the same with g rebuilt from the pooled gradient dy [B/8][L1/4][8][64] and the layer's BatchNorm scale / shift (MaxPool1d(4) + ReLU
// backward on load): pairs with sed_maxpool4_relu_bwd called with g = NULL (statistics only)
extern "C" int sed_m5_conv1_wgrad_fused_pool(int dtype, const float* x, const void* dy, const void* zsrc, const float* scale,
                                             const float* shift, const float* ca, const float* cb, const float* cc, float* dw_partial,
                                             int B, int L, void* stream) {
    SED_REQUIRE(dtype == SED_BF16, "covered: bf16");
    SED_REQUIRE(B > 0 && B % 8 == 0, "the interleaved layout needs a batch that is a multiple of 8");
    SED_REQUIRE(x && dy && zsrc && scale && shift && ca && cb && cc && dw_partial, "operands");
    const int L1 = sed_m5_conv1_len(L), tiles = cdiv(L1, TT);
    const int grid = sed_m5_conv1_nparts(B, L);
    m5_conv1_wgrad_mfma_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(x, (const bf16_t*)dy, (const bf16_t*)zsrc, scale, shift, ca, cb,
                                                                           cc, dw_partial, B, L, L1, tiles);
    SED_LAUNCH_CHECK();
    return 0;
}
