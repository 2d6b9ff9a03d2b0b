// Block 0 (C1 mode) backward of conv2 in ONE launch: dz2 is never written and never read back (bf16, gfx950, W = 64, 32 -> 32).
//
// autograd through the first ConvBlock, /root/reference/models/spectogram_models.py:132-158 under train.py:102:
//   dz2 = BN2 / ReLU / 2x2 avg-pool backward of (dy, z2)                       produced on load into an LDS row ring
//   dW2 = relu(bn1(conv1(x))) (x) dz2                                          (sed_conv3x3_wgrad_fused_c1: activation rebuilt on the
//                                                                               matrix pipe from the 1-channel input, never read)
//   g   = relu'(bn1(z1)) * conv2^T(dz2)                                        never written: gated with conv1's ReLU bit mask in
//   A[tap][c] = sum_px g[px][c] * x[px + tap],  sum g                           registers and contracted over the pixels on the matrix
//                                                                               pipe (sed_conv3x3_dgrad_c1_stats, csrc/sed_dgrad_c1.hip)
// It merges the two kernels named in the right column, which moved dz2 (786 MB at B = 32) through HBM twice; the operands left
// are one streaming read of z2, dy (a quarter of it), the fp32 input and the mask.
//
// Structure as csrc/sed_bwd_fused.hip (row ring with contiguous windows, one s_barrier per stage, waves 4-7 produce, waves 0-3
// consume); per 256-pixel stage a consumer wave
//   * contracts its k share (one tile row) of the weight gradient: 36 MFMAs into nine resident 32x32 accumulators,
//   * computes the data gradient of one tile row with the operands swapped (D[pixel][channel]: the B-operand layout of the second
//     contraction), gates it and contracts it against the patch fragments: 36 + 4 MFMAs,
//   * rebuilds one row of the NEXT stage's activation tile relu(bn1(conv1(x))): 2 MFMAs + their tails.
// Nothing is stored per tile; the workgroup writes its weight-gradient slab and its [A; sum g] partial once at the end.
#include "conv_common.h"

#include <stdlib.h>

namespace {

#ifdef SED_STAMPS
constexpr bool kBcStamps = true;        // make STAMPS=1: phase cycles of one workgroup (tools/bf_stamp.sh)
#else
constexpr bool kBcStamps = false;
#endif
__device__ __forceinline__ void bc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int bc_xswz(int col) { return (col >> 2) & 3; }

struct BwdC1Params {
    const float* x1;         // [B][H][64] fp32 input
    const float* fmean;      // [64] or NULL
    const float* fstd;
    const float* w1;         // conv1 weights [32][9]
    const float* sc1;        // BN1 scale / shift [32]
    const float* sh1;
    const void* dy;          // pooled gradient [B][H/2][32][32] bf16
    const void* z2;          // [B][H][64][32] bf16
    const float* sc2;        // BN2 scale / shift, ca / cb / cc [32]
    const float* sh2;
    const float* ca;
    const float* cb;
    const float* cc;
    const void* wpack_t;     // conv2's data-gradient operator [36][32][8] bf16
    const unsigned* mask;    // conv1's ReLU decisions [B][H][64] (one word per pixel)
    float* a_part;           // [nparts][10][32]
    float* ws;               // [nwg][9][32][32]
    int B, H;
    int tilesPerImg, totalTiles, tpb, nparts;
    int prio;                // issue priority of the loader waves (SED_BC_PRIO, A/B runs)
};

// Activation tile `ab` ([pixel][32 channels], 64 B per pixel): the 8-byte chunk c8 (channels 4*c8 .. 4*c8+3) of pixel column `col` sits
// at chunk position c8 ^ ((col >> 1) & 7).  The builders store a 32x32 accumulator tile as ds_write_b64 (lane = pixel, one chunk per
// instruction; the LDS serves a store in groups of 16 consecutive lanes over 32 banks): unswizzled, the 16 pixels of a group sit 64 B
// apart and hit FOUR banks eight times each (round 4, SQ_LDS_BANK_CONFLICT: a third of this kernel's LDS-array cycles); swizzled, the
// 16 (pixel parity, chunk position) pairs are distinct = all 32 banks once.  The weight gradient's transposed reads take all eight
// chunks of four consecutive pixels per 32-lane group (256 contiguous bytes): any permutation inside a pixel is conflict-free there.
// SED_BC_ABSWZ=0: the linear tile (A/B builds).
#ifndef SED_BC_GATE_AT
#define SED_BC_GATE_AT 17      // k-step of the data-gradient loop at which the derived gate's four transposed reads are requested
#endif
#ifndef SED_BC_ABSWZ
#define SED_BC_ABSWZ 1
#endif
__device__ __forceinline__ int ab_chunk(int c8, int col) { return (SED_BC_ABSWZ ? (c8 ^ ((col >> 1) & 7)) : c8) * 4; }

// TS (make EXPERIMENTS=1, SED_BC_TS=1): the weight-gradient accumulators split over the consumer waves by tap instead of by k share
// LB (SED_BC_LB): the conv1 tile relu(bn1(conv1(x))) of a stage is rebuilt by the LOADER waves (one tile row each, in the iteration
// that stages the stage's dz chunk) instead of by the consumer waves at the end of the stage before: the consumers are this kernel's
// critical path (their stage is weight gradient 1520 + data gradient / gate / contraction 2480 + rebuild 1540 cycles and they never
// wait at the barrier), the loaders have the slack
// DG (round 5, the default form: relu_mask = NULL at the C ABI): the ReLU gate of the data gradient is DERIVED from the activation tile the
// kernel rebuilds for its weight gradient anyway (a1 > 0 <=> the forward's decision bit: same MFMA, same bits) -- the forward then
// neither builds nor stores conv1's bit mask (16 of the ~36 instructions of its rebuild tail per 32-pixel block, 4 B/pixel of stores)
// and this kernel neither loads nor stages it.  The consumer lane of channel r takes four pixels of its channel with one
// ds_read_b64_tr_b16 (the weight gradient's A-operand read pattern) and gates the bf16 PAIRS of g: v_pk_min_u16 (activation bits
// -> 0 / 1) + v_pk_mul_lo_u16 -- two instructions per pair instead of four (bit-field extract + and per value).
template <bool TS, bool LB = false, bool DG = false>
__global__ __launch_bounds__(512) void conv_bwd_fused_c1_kernel(BwdC1Params p) {
    typedef bf16_t T;
    constexpr int W = 64, TH = 4, BM = TH * W, WP = 68, ROWE = WP * 32;
    constexpr int DZIMG = (4 * TH + 2) * ROWE;
    constexpr int ABUF = BM * 32, WS = 9 * 32 * 32;
    // z-scored input tile, XTR rows x (W + 2) columns (row 0 = image row of the first output row - 2), kept as the bf16 two-copy tile of
    // conv_common.h (C1Tile): the conv1 rebuild reads a patch row with one 8-byte read, and so does the second contraction's patch operand
    // (lane = TAP (dy, dx), four consecutive pixels per read: copy A for even dx, copy B for dx = 1; nine different banks)
    constexpr int XTW = W + 2, XTR = TH + 2, XTN = XTR * XTW;
    typedef C1Tile<W, XTR> XTL;
    constexpr int XCN = XTR * XTL::XP;                             // one constant region (ones / zeros), laid out like a tile's copy A
    static_assert((3 * XTL::N * 2) % 16 == 0, "the mask tile behind the input tiles is read with 16-byte loads");
    constexpr int NP = 256, NTHR = 512;
    constexpr int XTIPT = (XTN + NP - 1) / NP;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* dzr = reinterpret_cast<T*>(smem);                     // [4*TH + 2][WP][32]   dz2 row ring (16-byte slots XOR-swizzled)
    T* ab = dzr + DZIMG;                                     // [2][BM][32]          relu(bn1(conv1(x))) of the tile's rows
    T* wsm = ab + 2 * ABUF;                                  // [WS]                 resident data-gradient operator
    T* xt0 = wsm + WS;                                       // [3][XTL::N]          input tiles (bf16, two copies each)
    unsigned* mk0 = reinterpret_cast<unsigned*>(xt0 + 3 * XTL::N);    // [2][BM]
    T* cst0 = reinterpret_cast<T*>(mk0 + 2 * BM);            // [2][XCN]: all ones (tap 9 -> sum g), all zeros (taps 10..31)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, Ho = H >> 1;
    constexpr int Wo = W >> 1;
    const int bx = (int)xcd_remap(blockIdx.x, gridDim.x), nbx = gridDim.x;
    const int NTI = p.tilesPerImg;
    const int t_begin = bx * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    const int ntl = t_end > t_begin ? t_end - t_begin : 0;
    const int pre = (ntl > 0 && (t_begin % NTI) != 0) ? 1 : 0;
    const int NS = ntl + pre;
    const int NI = (NS + 1) & ~1;

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
        n.pos = (c.pos + (j == 0 ? 2 : 1)) & 3;
        return n;
    };

    // ---- one-time LDS setup ------------------------------------------------------------------------------------------
    {
        bf16x8 z8;
#pragma unroll
        for (int e = 0; e < 8; ++e) z8[e] = (bf16_t)0.f;
        for (int i = tid; i < DZIMG / 8; i += NTHR) *reinterpret_cast<bf16x8*>(dzr + i * 8) = z8;
        for (int i = tid; i < 2 * XCN; i += NTHR) cst0[i] = (T)(i < XCN ? 1.0f : 0.0f);
        for (int k = 0; k < 3; ++k) c1tile_init<W, XTR>(xt0 + k * XTL::N, tid, NTHR);
        const T* __restrict__ wg = reinterpret_cast<const T*>(p.wpack_t);
        for (int i = tid; i < WS / 8; i += NTHR) *reinterpret_cast<bf16x8*>(wsm + i * 8) = *reinterpret_cast<const bf16x8*>(wg + i * 8);
        __syncthreads();                                     // (the tiles' zeros / constants before the 2-byte stores below)
        // the input tile of stage 0 (later stages: staged one iteration ahead by the loader waves)
        const StInfo f = st_first();
        for (int e = tid; e < XTN; e += NTHR) {
            const int rr = e / XTW, c = e - rr * XTW;
            const int hy = TH * f.j - 2 + rr, wx = c - 1;
            float v = 0.f;
            if (f.live && hy >= 0 && hy < H && wx >= 0 && wx < W) {
                v = p.x1[((size_t)f.b * H + hy) * W + wx];
                if (p.fmean) v = (v - p.fmean[wx]) * (1.0f / p.fstd[wx]);
            }
            c1tile_store<W, XTR>(xt0, rr, c, v);
        }
    }
    __syncthreads();

    if (wave >= 4) {
        // =============================== PRODUCERS =====================================================
        if (p.prio == 1) __builtin_amdgcn_s_setprio(1);          // SED_BC_PRIO (A/B): issue priority of the loader waves
        else if (p.prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (p.prio == 3) __builtin_amdgcn_s_setprio(3);
        const T* __restrict__ gg = reinterpret_cast<const T*>(p.dy);
        const T* __restrict__ zsg = reinterpret_cast<const T*>(p.z2);
        const int pt = tid - 256;
        const size_t zimg_ = (size_t)H * W * 32, pimg_ = (size_t)Ho * Wo * 32, x1img_ = (size_t)H * W;
        // dz items: thread = (column dq0, channel group cg), item u = chunk row u
        const int dq0 = pt >> 2, cg = pt & 3;
        const unsigned dvoff0 = (unsigned)((dq0 * 32 + cg * 8) * 2);
        const int dlds0 = (dq0 + 1) * 32 + ((cg ^ bc_xswz(dq0 + 1)) * 8);
        // one dy item per 2x2 pooling window: the even column's thread loads it once per row pair, the odd column's thread (4 lanes
        // up: pt = 4*column + channel group) takes it by DPP (row_shr:4 into lanes 4-7 and 12-15 of each row of 16)
        const bool odd = dq0 & 1;
        unsigned pvoff[TH / 2];
#pragma unroll
        for (int u2 = 0; u2 < TH / 2; ++u2) pvoff[u2] = odd ? SED_OOB : (unsigned)(((u2 * Wo + (dq0 >> 1)) * 32 + cg * 8) * 2);
        float kca[8], kcb[8], kcc[8], ksc[8], ksh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            kca[e] = p.ca[cg * 8 + e] * 0.25f; kcb[e] = p.cb[cg * 8 + e]; kcc[e] = p.cc[cg * 8 + e];
            ksc[e] = p.sc2[cg * 8 + e]; ksh[e] = p.sh2[cg * 8 + e];
        }
        float xtmu[XTIPT], xtis[XTIPT];
#pragma unroll
        for (int u = 0; u < XTIPT; ++u) {
            const int e = pt + u * NP, c = (e % XTW) - 1;
            const bool ok = e < XTN && c >= 0 && c < W;
            xtmu[u] = (ok && p.fmean) ? p.fmean[c] : 0.f;
            xtis[u] = ok ? (p.fstd ? 1.0f / p.fstd[c] : 1.0f) : 0.f;
        }
        struct RawSet { Raw8<T> a[TH / 2]; Raw8<T> b[TH]; float xr[XTIPT]; unsigned m; };

        // every load is issued unconditionally (a dead stage gets zero-sized descriptors): exact vmcnt bookkeeping
        auto issue = [&](RawSet& r, const StInfo& si) {
            const bool live = si.live;
            const size_t zimg = live ? zimg_ : 0, pimg = live ? pimg_ : 0, mimg = (live && si.mainst) ? x1img_ : 0;
            const unsigned dt = (unsigned)(TH * si.j * W * 32 * 2);
            const __amdgpu_buffer_rsrc_t zs = make_srd(zsg + (size_t)si.b * zimg, zimg * 2);
            const __amdgpu_buffer_rsrc_t gs = make_srd(gg + (size_t)si.b * pimg, pimg * 2);
            const unsigned ptq = (unsigned)((TH / 2) * si.j * Wo * 32 * 2);
#pragma unroll
            for (int u2 = 0; u2 < TH / 2; ++u2) r.a[u2] = buf_load8<T>(gs, pvoff[u2] + ptq);
#pragma unroll
            for (int u = 0; u < TH; ++u) r.b[u] = buf_load8<T>(zs, dvoff0 + (unsigned)(u * W * 32 * 2) + dt);
            if constexpr (!DG) {
                const __amdgpu_buffer_rsrc_t sm = make_srd(p.mask + (size_t)si.b * mimg, mimg * 4);
                r.m = __builtin_amdgcn_raw_buffer_load_b32(sm, (unsigned)(((TH * si.j - 1) * W + pt) * 4), 0, 0);     // rows outside: 0 -> gated off
            }
        };
        // the input tile of the stage after next rides in the same register set
        auto issue_x1 = [&](RawSet& r, const StInfo& si) {
            const size_t img = (si.live && si.mainst) ? x1img_ : 0;
            const __amdgpu_buffer_rsrc_t s1 = make_srd(p.x1 + (size_t)si.b * img, img * 4);
#pragma unroll
            for (int u = 0; u < XTIPT; ++u) {
                const int e = pt + u * NP, rr = e / XTW, c = e - rr * XTW - 1;
                const bool ok = e < XTN && c >= 0 && c < W;
                r.xr[u] = buf_load_f32(s1, ok ? (unsigned)(((TH * si.j - 2 + rr) * W + c) * 4) : SED_OOB);
            }
        };
        auto write_xt = [&](const RawSet& r, const StInfo& si, int s) {          // si = stage s; z-scored, zero outside the image
            T* xtn = xt0 + (s % 3) * XTL::N;
            const bool ok = si.live && si.mainst;
#pragma unroll
            for (int u = 0; u < XTIPT; ++u) {
                const int e = pt + u * NP;
                if (u == XTIPT - 1 && e >= XTN) break;
                const int er = e / XTW, hy = TH * si.j - 2 + er;
                c1tile_store<W, XTR>(xtn, er, e - er * XTW, (ok && hy >= 0 && hy < H) ? (r.xr[u] - xtmu[u]) * xtis[u] : 0.f);
            }
        };
        auto commit = [&](const RawSet& r, const StInfo& si, int s) {
            if (!si.live) return;
            T* __restrict__ dst = dzr + dlds0 + (si.pos * TH + 2) * ROWE;
            const bool dup = si.pos == 3;
            if (si.j == 0) {
                bf16x8 z8;
#pragma unroll
                for (int e = 0; e < 8; ++e) z8[e] = (bf16_t)0.f;
                *reinterpret_cast<bf16x8*>(dst - 2 * ROWE) = z8;
                *reinterpret_cast<bf16x8*>(dst - ROWE) = z8;
            }
            const int rows_in = H - TH * si.j;
#pragma unroll
            for (int u = 0; u < TH; ++u) {
                float g[8], z[8], v[8];
                u32x4 w4 = __builtin_bit_cast(u32x4, r.a[u / 2].v);
#pragma unroll
                for (int e = 0; e < 4; ++e)      // row_shr:4, banks 1 and 3 only: lanes 4-7 / 12-15 (odd columns) take lanes 0-3 / 8-11
                    w4[e] = (unsigned)__builtin_amdgcn_update_dpp((int)w4[e], (int)w4[e], 0x114, 0xF, 0xA, false);
                Raw8<T> t8; t8.v = __builtin_bit_cast(bf16x8, w4);
                raw_to_f(t8, g);
                raw_to_f(r.b[u], z);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float base = fmaf(kcb[i], z[i], kcc[i]);
                    const float full = fmaf(kca[i], g[i], base);
                    v[i] = (fmaf(z[i], ksc[i], ksh[i]) > 0.f) ? full : base;
                }
#if SED_BOUNDARY_BRANCH
                if (u >= rows_in) {                          // (uniform, only the last chunk of an image: rows past it are zero.  A real branch -- hipcc
                    asm volatile("" ::: "memory");          //  had if-converted the former `v *= m` into eight multiplies per item of EVERY stage)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = 0.f;
                }
#else
                if (rows_in < TH) {
                    const float m = (u < rows_in) ? 1.f : 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= m;
                }
#endif
                store8<T>(dst + u * ROWE, v);
                if (u >= TH - 2 && dup) store8<T>(dzr + dlds0 + (u - (TH - 2)) * ROWE, v);
            }
            if constexpr (!DG) { if (si.mainst) mk0[(s & 1) * BM + pt] = r.m; }
        };

        // LB: row (wave - 4) of stage si's activation tile from the input tile xt (two 32-pixel blocks), as the consumers' build()
        C1Mma c1m;
        int c1o[2][2] = {{0, 0}, {0, 0}};
        if constexpr (LB) {
            c1mma_init(c1m, p.w1, p.sc1, p.sh1, lane);
            c1tile_lane_offsets<W, XTR>(0, lane, c1o[0][0], c1o[0][1]);
            c1tile_lane_offsets<W, XTR>(1, lane, c1o[1][0], c1o[1][1]);
        }
        auto lbuild = [&](const StInfo& si, int s) {
            if constexpr (LB) {
                if (!si.live || !si.mainst) return;
                const int bw = wave - 4, r = lane & 31, hh = lane >> 5;
                const T* xt = xt0 + (s % 3) * XTL::N;
                const int row = TH * si.j - 1 + bw;
                const bool inimg = row >= 0 && row < H;
                T* abuf = ab + (s & 1) * ABUF;
                f32x16 dd[2];
#pragma unroll
                for (int half = 0; half < 2; ++half) dd[half] = c1mma_block_mfma_b(c1m, xt + bw * XTL::XP, c1o[half][0], c1o[half][1]);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    T* dst = abuf + (bw * W + half * 32 + r) * 32;
                    unsigned w8[8], mkd;
                    c1mma_block_tail_pk<false>(dd[half], w8, mkd);
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const u32x2 v2 = {inimg ? w8[2 * g4] : 0u, inimg ? w8[2 * g4 + 1] : 0u};
                        *reinterpret_cast<u32x2*>(dst + ab_chunk(hh + 2 * g4, r)) = v2;
                    }
                }
            }
        };
        RawSet ra, rb;
        StInfo sc = st_first();
        StInfo sn = st_next(sc, 1), sf = st_next(sn, 2);      // sn = stage s + 1, sf = stage s + 2
        issue(ra, sc);
        issue(rb, sn);
        issue_x1(ra, sn);                                     // set (s & 1) carries the input tile of stage s + 1
        issue_x1(rb, sf);
        unsigned long long tp[3] = {0, 0, 0};
        auto pstamp = [&]() -> unsigned long long { return kBcStamps ? __builtin_amdgcn_s_memtime() : 0ull; };
        auto iter = [&](int s, RawSet& r) {
            const unsigned long long s0 = pstamp();
            lbuild(sc, s);                                    // (its input tile was written in iteration s - 1 / by the set-up, before a barrier)
            commit(r, sc, s);
            write_xt(r, sn, s + 1);
            const unsigned long long s1 = pstamp();
            issue(r, sf);
            sc = sn; sn = sf; sf = st_next(sf, s + 3);
            issue_x1(r, sf);                                  // (after the shift: sf = stage s + 3, whose tile set (s & 1) carries next)
            const unsigned long long s2 = pstamp();
            bc_barrier();
            if (kBcStamps) { tp[0] += s1 - s0; tp[1] += s2 - s1; tp[2] += pstamp() - s2; }
        };
        for (int s = 0; s < NI; s += 2) {
            iter(s, ra);
            iter(s + 1, rb);
        }
        if (kBcStamps && blockIdx.x == 8 && lane == 0 && wave == 5)
            printf("bc producer: %d stages; cycles commit %llu issue %llu barrier %llu\n", NI, tp[0], tp[1], tp[2]);
        bc_barrier();
        bc_barrier();
    } else {
        if constexpr (!TS) {
        // =============================== CONSUMERS =====================================================
        const int r = lane & 31, hh = lane >> 5;
        f32x16 accw[9], accA;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) accw[t][i] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) accA[i] = 0.f;
        // data gradient: this wave's tile row, both 32-pixel halves (mt); lane part of the dz fragment address (A operand: pixel r)
        int xoff[3][2];
#pragma unroll
        for (int tj = 0; tj < 3; ++tj)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xoff[tj][ks] = (r + tj) * 32 + (((ks * 2 + hh) ^ bc_xswz(r + tj)) * 8);
        const int woff = (hh * 32 + r) * 8;
        // second contraction: this lane is channel r of g (B operand) and tap r of the patch matrix (A operand)
        const unsigned bitpos = 16 * ((r >> 2) & 1) + (r & 3) + 4 * (r >> 3);
        const int tap = r < 9 ? r : 0;
        const int pdx = tap % 3;
        const int ptap = (tap / 3) * XTL::XP + ((pdx & 1) ? XTL::XB + pdx - 1 : pdx) + 4 * hh;      // (copy B for dx = 1: 4-byte-aligned 8-byte reads)
        const T* cbase = cst0 + (r == 9 ? 0 : XCN);
        // weight gradient: k share = tile row `wave`
        int offA[2], offB[3][2];
        {
            const int i16 = lane & 15, gbit = (lane >> 4) & 1;
            const int qq = i16 >> 2, pp = i16 & 3, ch = 16 * gbit + 4 * pp;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int kl = 8 * hh + qq + 4 * half;
                offA[half] = kl * 32 + ab_chunk(ch >> 2, kl);
#pragma unroll
                for (int sj = 0; sj < 3; ++sj) offB[sj][half] = (kl + sj) * 32 + (ch ^ swz<T>(kl + sj));
            }
        }
        // DG: the gate's transposed reads of the activation tile -- lane (channel r, half hh) takes pixels 8*i4 + 4*hh + 0..3 of its 32-pixel
        // half; the address-supplying lane i16 = 4*qq + pp of a 16-lane group points at (pixel + qq, channels 16*gbit + 4*pp ..)
        // (pixels 8 further: + 256 elements and chunk position ^ 4, i.e. element offset ^ 16 -- formed at the use: one register less)
        int offG0;
        {
            const int i16 = lane & 15, gbit = (lane >> 4) & 1;
            const int qq = i16 >> 2, pp = i16 & 3;
            const int col = 4 * hh + qq;
            offG0 = col * 32 + ab_chunk(4 * gbit + pp, col);
        }
        C1Mma c1m;
        c1mma_init(c1m, p.w1, p.sc1, p.sh1, lane);
        int c1o[2][2];
        c1tile_lane_offsets<W, XTR>(0, lane, c1o[0][0], c1o[0][1]);
        c1tile_lane_offsets<W, XTR>(1, lane, c1o[1][0], c1o[1][1]);
        // row `wave` of stage si's activation tile relu(bn1(conv1(x))) from the input tile xt (two 32-pixel blocks)
        auto build = [&](const StInfo& si, int s) {
            if (!si.live || !si.mainst) return;
            const T* xt = xt0 + (s % 3) * XTL::N;
            const int row = TH * si.j - 1 + wave;
            const bool inimg = row >= 0 && row < H;
            T* abuf = ab + (s & 1) * ABUF;
            f32x16 dd[2];
#pragma unroll
            for (int half = 0; half < 2; ++half) dd[half] = c1mma_block_mfma_b(c1m, xt + wave * XTL::XP, c1o[half][0], c1o[half][1]);      // reads + MFMAs first
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                T* dst = abuf + (wave * W + half * 32 + r) * 32;
#if !defined(SED_C1_PKTAIL) || SED_C1_PKTAIL
                unsigned w8[8], mkd;
                c1mma_block_tail_pk<false>(dd[half], w8, mkd);           // ReLU on the packed bf16 words (conv_common.h)
#if SED_BOUNDARY_BRANCH
                if (!inimg) {        // (wave-uniform, only at the top / bottom of an image: a real branch instead of eight selects per block of every stage)
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int k8 = 0; k8 < 8; ++k8) w8[k8] = 0u;
                }
#else
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) w8[k8] = inimg ? w8[k8] : 0u;
#endif
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const u32x2 v2 = {w8[2 * g4], w8[2 * g4 + 1]};
                    *reinterpret_cast<u32x2*>(dst + ab_chunk(hh + 2 * g4, r)) = v2;
                }
#else
                float a[16];
                unsigned mkd;
                c1mma_block_tail<false>(c1m, dd[half], a, mkd);
                if (!inimg) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) a[i] = 0.f;
                }
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float v4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = a[4 * g4 + e];
                    store4<T>(dst + ab_chunk(hh + 2 * g4, r), v4);
                }
#endif
            }
        };

        unsigned long long tc[5] = {0, 0, 0, 0, 0};
        auto cstamp = [&]() -> unsigned long long { return kBcStamps ? __builtin_amdgcn_s_memtime() : 0ull; };
        StInfo csi = st_first();
        if constexpr (!LB) build(csi, 0);                     // (stage 0's input tile was staged by the whole workgroup)
        auto citer = [&](int s) {
            const unsigned long long c0 = cstamp();
            bc_barrier();
            const unsigned long long c1 = cstamp();
            unsigned long long c2 = c1, c3 = c1;
            const StInfo cs = csi;
            csi = st_next(csi, s + 1);
            if (cs.live && cs.mainst) {
                const T* __restrict__ win = dzr + cs.pos * TH * ROWE;
                const T* __restrict__ abuf = ab + (s & 1) * ABUF;
#if !defined(SED_BC_COLK) || SED_BC_COLK
                // ---- weight gradient: accw[si*3+sj] += a[k-step] (x) dz[k-step shifted by (si, sj)] ----------------------------------
                // k share of this wave = the 16-pixel COLUMN STRIP `wave` of all TH tile rows (k-step = tile row).  The dz fragment of
                // window row rho and column shift sj then serves every tile row `row` with 0 <= rho - row <= 2 (shift row si = rho -
                // row): 18 fragment reads per stage instead of the 36 of a row-wise k share -- the fused backward kernels keep the LDS
                // pipe busier than the matrix pipe (round 4: ~1.5 KB of LDS reads per MFMA, tools/pmc_lds.sh), bytes are what counts.
                {
                    bf16x8 afr[TH], bfr[2][3];
                    const T* __restrict__ abase = abuf + (16 * wave) * 32;
                    const T* __restrict__ wbase = win + (16 * wave) * 32;
                    auto ld_a = [&](int row, bf16x8& dst) {
                        dst = join_tr(ds_read_tr16_b64(abase + row * W * 32 + offA[0]), ds_read_tr16_b64(abase + row * W * 32 + offA[1]));
                    };
                    auto ld_b = [&](int rho, bf16x8 (&dst)[3]) {
#pragma unroll
                        for (int sj = 0; sj < 3; ++sj)
                            dst[sj] = join_tr(ds_read_tr16_b64(wbase + rho * ROWE + offB[sj][0]), ds_read_tr16_b64(wbase + rho * ROWE + offB[sj][1]));
                    };
                    ld_a(0, afr[0]);
                    ld_b(0, bfr[0]);
                    ld_a(1, afr[1]);
#pragma unroll
                    for (int rho = 0; rho < TH + 2; ++rho) {
                        if (rho + 1 < TH + 2) ld_b(rho + 1, bfr[(rho + 1) & 1]);
                        if (rho + 2 < TH) ld_a(rho + 2, afr[rho + 2]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int row = (rho > 2 ? rho - 2 : 0); row <= (rho < TH - 1 ? rho : TH - 1); ++row)
#pragma unroll
                            for (int sj = 0; sj < 3; ++sj)
                                accw[(rho - row) * 3 + sj] = mfma(afr[row], bfr[rho & 1][sj], accw[(rho - row) * 3 + sj]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#else      // (A/B builds, tools/ab_build.sh SED_BC_COLK: the row-wise k share of round 3)
                // ---- weight gradient: accw[si*3+sj] += a[k-step] (x) dz[k-step shifted by (si, sj)], k share = row `wave` ----------
                {
                    constexpr int KSW = 4, NSTEP = KSW * 3;
                    bf16x8 bfr[3][3], afr[2];
                    const T* __restrict__ abase = abuf + wave * W * 32;
                    const T* __restrict__ wbase = win + wave * ROWE;
                    auto ld_a = [&](int kk, bf16x8& dst) {
                        dst = join_tr(ds_read_tr16_b64(abase + kk * 16 * 32 + offA[0]), ds_read_tr16_b64(abase + kk * 16 * 32 + offA[1]));
                    };
                    auto ld_b = [&](int st, bf16x8 (&dst)[3]) {
                        const int kk = st / 3, si = st % 3;
                        const int imm = si * ROWE + kk * 16 * 32;
#pragma unroll
                        for (int sj = 0; sj < 3; ++sj)
                            dst[sj] = join_tr(ds_read_tr16_b64(wbase + imm + offB[sj][0]), ds_read_tr16_b64(wbase + imm + offB[sj][1]));
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
#endif
                c2 = cstamp();
                // ---- data gradient of tile row `wave`, D[pixel][channel]; gate; contract over the pixels ---------------------------
                // (one 32-pixel half at a time: beside the nine weight-gradient accumulators there is no room for both)
                {
                    const T* __restrict__ dbase = win + wave * ROWE;
                    const T* __restrict__ xtb = xt0 + (s % 3) * XTL::N;
                    const unsigned* __restrict__ mkb = mk0 + (s & 1) * BM;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        f32x16 acc;
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                        // gate / patch operands of this half: requested before the k loop, so their LDS round trips hide behind it
                        const unsigned* mrow = mkb + wave * W + mt * 32 + 4 * hh;
                        const T* prow_x = (r < 9 ? xtb : cbase) + wave * XTL::XP + mt * 32 + ptap;
                        u32x4 pfw[2];
#pragma unroll
                        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj) {
                                const c1_u32x2_a4 q4 = *reinterpret_cast<const c1_u32x2_a4*>(prow_x + 16 * sx + 8 * jj);
                                pfw[sx][2 * jj] = q4[0]; pfw[sx][2 * jj + 1] = q4[1];
                            }
                        // DG: the gate operand = this half's activations of the wave's own tile row (the tile the weight gradient just read)
                        // (requested at the LAST k-step, when the fragment ring has drained: eight more registers across the whole loop spill)
                        s16x4 gact[4];
                        const T* grow = abuf + (wave * W + mt * 32) * 32;
                        // fragment ring: SED_BC_DRING k-steps ahead of their MFMAs (A/B builds; round 4: two steps ahead = 256 registers, 0.589-0.597 vs
                        // 0.581 ms with one -- not kept, profiles/r04_f_ab_block0_bwd_ring.txt)
#ifndef SED_BC_DRING
#define SED_BC_DRING 1
#endif
                        constexpr int DR = SED_BC_DRING + 1;
                        bf16x8 xf[DR], wf[DR];
                        auto ld = [&](int k, bf16x8& xd, bf16x8& wd) {
                            const int tp = k >> 1, ks = k & 1, ti = tp / 3, tj = tp % 3;
                            xd = *reinterpret_cast<const bf16x8*>(dbase + (ti * ROWE + mt * 32 * 32) + xoff[tj][ks]);
                            wd = *reinterpret_cast<const bf16x8*>(wsm + woff + ((tp * 4 + ks * 2) * 32) * 8);
                        };
#pragma unroll
                        for (int k = 0; k < DR - 1; ++k) ld(k, xf[k], wf[k]);
#pragma unroll
                        for (int k = 0; k < 18; ++k) {
                            if (k + DR - 1 < 18) ld(k + DR - 1, xf[(k + DR - 1) % DR], wf[(k + DR - 1) % DR]);
                            if constexpr (DG) {
                                if (k == SED_BC_GATE_AT) {
                                    int offG1;
                                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(offG1) : "v"(offG0), "v"(SED_BC_ABSWZ ? 16 : 0));
#pragma unroll
                                    for (int i4 = 0; i4 < 4; ++i4) gact[i4] = ds_read_tr16_b64(grow + (16 * (i4 >> 1) + 8 * (i4 & 1)) * 32 + ((i4 & 1) ? offG1 : offG0));
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            acc = mfma(xf[k % DR], wf[k % DR], acc);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (DG) {
                            if (SED_BC_GATE_AT >= 18) {
                                int offG1;
                                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(offG1) : "v"(offG0), "v"(SED_BC_ABSWZ ? 16 : 0));
#pragma unroll
                                for (int i4 = 0; i4 < 4; ++i4) gact[i4] = ds_read_tr16_b64(grow + (16 * (i4 >> 1) + 8 * (i4 & 1)) * 32 + ((i4 & 1) ? offG1 : offG0));
                            }
                            unsigned gw[8];
#pragma unroll
                            for (int i4 = 0; i4 < 4; ++i4) {
                                const u32x2 aw = __builtin_bit_cast(u32x2, gact[i4]);
#pragma unroll
                                for (int h2 = 0; h2 < 2; ++h2) {
                                    const f32x2 pr = {acc[4 * i4 + 2 * h2], acc[4 * i4 + 2 * h2 + 1]};
                                    const unsigned gb = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2));
                                    unsigned on;
                                    asm("v_pk_min_u16 %0, %1, %2" : "=v"(on) : "v"(aw[h2]), "s"(0x00010001u));
                                    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(gw[2 * i4 + h2]) : "v"(gb), "v"(on));
                                }
                            }
#pragma unroll
                            for (int sx = 0; sx < 2; ++sx) {
                                const u32x4 g4 = {gw[4 * sx], gw[4 * sx + 1], gw[4 * sx + 2], gw[4 * sx + 3]};
                                accA = mfma(__builtin_bit_cast(bf16x8, pfw[sx]), __builtin_bit_cast(bf16x8, g4), accA);
                            }
                        } else {
                        unsigned gv[16];
#pragma unroll
                        for (int i4 = 0; i4 < 4; ++i4) {
                            const u32x4 m4 = *reinterpret_cast<const u32x4*>(mrow + 8 * i4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int t = __builtin_amdgcn_sbfe((int)m4[e], bitpos, 1u);
                                const float av = acc[4 * i4 + e];
                                gv[4 * i4 + e] = __builtin_bit_cast(unsigned, av) & (unsigned)t;
                            }
                        }
#pragma unroll
                        for (int sx = 0; sx < 2; ++sx) {
                            bf16x8 gf;
#pragma unroll
                            for (int jj = 0; jj < 8; ++jj) gf[jj] = (bf16_t)__builtin_bit_cast(float, gv[8 * sx + jj]);
                            accA = mfma(__builtin_bit_cast(bf16x8, pfw[sx]), gf, accA);
                        }
                        }
                    }
                }
            }
            c3 = cstamp();
            if constexpr (!LB) build(csi, s + 1);             // the next stage's activation row (its input tile was written before this barrier)
            if (kBcStamps) { tc[0] += c1 - c0; tc[1] += c2 - c1; tc[2] += c3 - c2; tc[3] += cstamp() - c3; }
        };
        for (int s = 0; s < NI; s += 2) {
            citer(s);
            citer(s + 1);
        }
        if (kBcStamps && blockIdx.x == 8 && lane == 0 && wave == 1)
            printf("bc consumer: %d stages; cycles barrier %llu wgrad %llu dgrad+gate %llu build %llu\n", NI, tc[0], tc[1], tc[2], tc[3]);
        // ---- this workgroup's slabs: the four k shares of dW and the four row partials of [A; sum g], fixed-order sums through LDS --
        bc_barrier();
        float* red = reinterpret_cast<float*>(smem);          // [3][9][16][64] weight-gradient shares of waves 1..3, then [4][16][64]
        if (wave > 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) red[(((wave - 1) * 9 + t) * 16 + i) * 64 + lane] = accw[t][i];
        }
        float* redA = red + 3 * 9 * 16 * 64;
#pragma unroll
        for (int i = 0; i < 16; ++i) redA[(wave * 16 + i) * 64 + lane] = accA[i];
        bc_barrier();
        if (wave == 0) {
            float* out = p.ws + (size_t)bx * 9 * 32 * 32;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int tp = (2 - t / 3) * 3 + (2 - t % 3);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = accw[t][i];
#pragma unroll
                    for (int w = 0; w < 3; ++w) v += red[((w * 9 + t) * 16 + i) * 64 + lane];
                    const int cin = (i & 3) + 8 * (i >> 2) + 4 * hh;
                    out[((size_t)tp * 32 + cin) * 32 + r] = v;
                }
            }
        }
        if (wave == 1) {
            // tap k (9 = sum g), channel c: register 4*(k>>3) + (k&3) of lane c + 32*((k>>2)&1)   (as csrc/sed_dgrad_c1.hip)
            for (int q = lane; q < 320; q += 64) {
                const int k = q >> 5, c = q & 31;
                const int i = 4 * (k >> 3) + (k & 3), ln = c + 32 * ((k >> 2) & 1);
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) tot += redA[(w * 16 + i) * 64 + ln];
                p.a_part[((size_t)bx * 10 + k) * 32 + c] = tot;
                for (int row = bx + nbx; row < p.nparts; row += nbx) p.a_part[((size_t)row * 10 + k) * 32 + c] = 0.f;
            }
        }
            } else {
        // =============================== CONSUMERS, taps split over the waves (TS) ===================================
        // The nine weight-gradient accumulators are divided by TAP (wave 0: shifts 0-2, waves 1-3: two each), every wave walks all 16
        // k-steps of the tile: 48 / 32 accumulator registers instead of 144.  The room is spent on what the k-share form could not
        // afford: both halves of the data gradient together (one weight fragment for two MFMAs), a fragment ring of three, the gate
        // and patch operands requested BEFORE the k loop, and the conv1 rebuild (waves 1-3: 3 / 3 / 2 blocks) issued at the top of the
        // iteration with its tails after the weight-gradient loop.  No cross-wave reduction of dW at the end.
        auto consumer_ts = [&](auto nt_c, auto nb_c) {
        constexpr int NT = decltype(nt_c)::value, NB = decltype(nb_c)::value;
        const int r = lane & 31, hh = lane >> 5;
        f32x16 accw[NT], accA;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) accw[t][i] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) accA[i] = 0.f;
        const int tfirst = wave == 0 ? 0 : 1 + 2 * wave;           // shift index t = si*3 + sj: 0-2 | 3,4 | 5,6 | 7,8
        constexpr bool three = NT == 3;
        int xoff[3][2];
#pragma unroll
        for (int tj = 0; tj < 3; ++tj)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xoff[tj][ks] = (r + tj) * 32 + (((ks * 2 + hh) ^ bc_xswz(r + tj)) * 8);
        const int woff = (hh * 32 + r) * 8;
        const unsigned bitpos = 16 * ((r >> 2) & 1) + (r & 3) + 4 * (r >> 3);
        const int tap = r < 9 ? r : 0;
        const int pdx = tap % 3;
        const int ptap = (tap / 3) * XTL::XP + ((pdx & 1) ? XTL::XB + pdx - 1 : pdx) + 4 * hh;      // (copy B for dx = 1: 4-byte-aligned 8-byte reads)
        const T* cbase = cst0 + (r == 9 ? 0 : XCN);
        int offA[2], offT[NT][2];
        {
            const int i16 = lane & 15, gbit = (lane >> 4) & 1;
            const int qq = i16 >> 2, pp = i16 & 3, ch = 16 * gbit + 4 * pp;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int kl = 8 * hh + qq + 4 * half;
                offA[half] = kl * 32 + ab_chunk(ch >> 2, kl);
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const int t = tfirst + k;
                    const int si = t / 3, sj = t - 3 * si;
                    offT[k][half] = si * ROWE + (kl + sj) * 32 + (ch ^ swz<T>(kl + sj));
                }
            }
        }
        C1Mma c1m;
        c1mma_init(c1m, p.w1, p.sc1, p.sh1, lane);
        int c1o[2][2];
        c1tile_lane_offsets<W, XTR>(0, lane, c1o[0][0], c1o[0][1]);
        c1tile_lane_offsets<W, XTR>(1, lane, c1o[1][0], c1o[1][1]);
        // conv1 rebuild: blocks b = 2*row + half of the next stage's activation tile; wave 1: 0-2, wave 2: 3-5, wave 3: 6, 7
        constexpr int nbld = NB;
        const int bld0 = wave == 0 ? 0 : 3 * (wave - 1);
        f32x16 dd[NB > 0 ? NB : 1];
        auto build_issue = [&](const StInfo& si, int s) {
            if (!si.live || !si.mainst) return;
            const T* xt = xt0 + (s % 3) * XTL::N;
#pragma unroll
            for (int k = 0; k < nbld; ++k)
                dd[k] = c1mma_block_mfma_b(c1m, xt + ((bld0 + k) >> 1) * XTL::XP, c1o[(bld0 + k) & 1][0], c1o[(bld0 + k) & 1][1]);
        };
        auto build_finish = [&](const StInfo& si, int s) {
            if (!si.live || !si.mainst) return;
            T* abuf = ab + (s & 1) * ABUF;
#pragma unroll
            for (int k = 0; k < nbld; ++k) {
                const int b = bld0 + k, brow = b >> 1, half = b & 1;
                const int row = TH * si.j - 1 + brow;
                const bool inimg = row >= 0 && row < H;
                float a[16];
                unsigned mkd;
                c1mma_block_tail<false>(c1m, dd[k], a, mkd);
                if (!inimg) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) a[i] = 0.f;
                }
                T* dst = abuf + (brow * W + half * 32 + r) * 32;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float v4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = a[4 * g4 + e];
                    store4<T>(dst + ab_chunk(hh + 2 * g4, r), v4);
                }
            }
        };

        StInfo csi = st_first();
        build_issue(csi, 0);
        build_finish(csi, 0);
        auto citer = [&](int s) {
            bc_barrier();
            const StInfo cs = csi;
            csi = st_next(csi, s + 1);
            build_issue(csi, s + 1);                          // next stage's conv1 blocks: reads + MFMAs now, tails after the weight gradient
            if (cs.live && cs.mainst) {
                const T* __restrict__ win = dzr + cs.pos * TH * ROWE;
                const T* __restrict__ abuf = ab + (s & 1) * ABUF;
                // ---- weight gradient: this wave's taps over all 16 k-steps of the tile ----------------------------------------
                {
                    constexpr int NKS = BM / 16;
                    bf16x8 afr[3], bfr[3][NT];
                    auto ldk = [&](int kk, bf16x8& a, bf16x8 (&b)[NT]) {
                        const int aimm = kk * 16 * 32;
                        const int dimm = (kk / 4) * ROWE + (kk % 4) * 16 * 32;
                        a = join_tr(ds_read_tr16_b64(abuf + aimm + offA[0]), ds_read_tr16_b64(abuf + aimm + offA[1]));
                        b[0] = join_tr(ds_read_tr16_b64(win + dimm + offT[0][0]), ds_read_tr16_b64(win + dimm + offT[0][1]));
                        b[1] = join_tr(ds_read_tr16_b64(win + dimm + offT[1][0]), ds_read_tr16_b64(win + dimm + offT[1][1]));
                        if constexpr (three) b[2] = join_tr(ds_read_tr16_b64(win + dimm + offT[2][0]), ds_read_tr16_b64(win + dimm + offT[2][1]));
                    };
                    ldk(0, afr[0], bfr[0]);
                    ldk(1, afr[1], bfr[1]);
#pragma unroll
                    for (int kk = 0; kk < NKS; ++kk) {
                        if (kk + 2 < NKS) ldk(kk + 2, afr[(kk + 2) % 3], bfr[(kk + 2) % 3]);
                        __builtin_amdgcn_sched_barrier(0);
                        accw[0] = mfma(afr[kk % 3], bfr[kk % 3][0], accw[0]);
                        accw[1] = mfma(afr[kk % 3], bfr[kk % 3][1], accw[1]);
                        if constexpr (three) accw[2] = mfma(afr[kk % 3], bfr[kk % 3][2], accw[2]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                build_finish(csi, s + 1);
                // ---- data gradient of tile row `wave` (both 32-pixel halves), D[pixel][channel]; gate; contract over the pixels ----------
                {
                    const T* __restrict__ dbase = win + wave * ROWE;
                    const T* __restrict__ xtb = xt0 + (s % 3) * XTL::N;
                    const unsigned* __restrict__ mkb = mk0 + (s & 1) * BM;
                    f32x16 acc[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
                    bf16x8 xf[3][2], wf[3];
                    auto ld = [&](int k, bf16x8 (&xd)[2], bf16x8& wd) {
                        const int tp = k >> 1, ks = k & 1, ti = tp / 3, tj = tp % 3;
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            xd[mt] = *reinterpret_cast<const bf16x8*>(dbase + (ti * ROWE + mt * 32 * 32) + xoff[tj][ks]);
                        wd = *reinterpret_cast<const bf16x8*>(wsm + woff + ((tp * 4 + ks * 2) * 32) * 8);
                    };
                    ld(0, xf[0], wf[0]);
                    ld(1, xf[1], wf[1]);
#pragma unroll
                    for (int k = 0; k < 18; ++k) {
                        if (k + 2 < 18) ld(k + 2, xf[(k + 2) % 3], wf[(k + 2) % 3]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) acc[mt] = mfma(xf[k % 3][mt], wf[k % 3], acc[mt]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // gate / patch operands: requested before the k loop
                    u32x4 m4v[2][4];
                    u32x4 pfw[2][2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const unsigned* mrow = mkb + wave * W + mt * 32 + 4 * hh;
                        const T* prow_x = (r < 9 ? xtb : cbase) + wave * XTL::XP + mt * 32 + ptap;
#pragma unroll
                        for (int i4 = 0; i4 < 4; ++i4) m4v[mt][i4] = *reinterpret_cast<const u32x4*>(mrow + 8 * i4);
#pragma unroll
                        for (int sx = 0; sx < 2; ++sx)
#pragma unroll
                            for (int jj = 0; jj < 2; ++jj) {
                                const c1_u32x2_a4 q4 = *reinterpret_cast<const c1_u32x2_a4*>(prow_x + 16 * sx + 8 * jj);
                                pfw[mt][sx][2 * jj] = q4[0]; pfw[mt][sx][2 * jj + 1] = q4[1];
                            }
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        unsigned gv[16];
#pragma unroll
                        for (int i4 = 0; i4 < 4; ++i4)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int t = __builtin_amdgcn_sbfe((int)m4v[mt][i4][e], bitpos, 1u);
                                const float av = acc[mt][4 * i4 + e];
                                gv[4 * i4 + e] = __builtin_bit_cast(unsigned, av) & (unsigned)t;
                            }
#pragma unroll
                        for (int sx = 0; sx < 2; ++sx) {
                            bf16x8 gf;
#pragma unroll
                            for (int jj = 0; jj < 8; ++jj) gf[jj] = (bf16_t)__builtin_bit_cast(float, gv[8 * sx + jj]);
                            accA = mfma(__builtin_bit_cast(bf16x8, pfw[mt][sx]), gf, accA);
                        }
                    }
                }
            } else {
                build_finish(csi, s + 1);                     // (a loader-only stage)
            }
        };
        for (int s = 0; s < NI; s += 2) {
            citer(s);
            citer(s + 1);
        }
        // ---- this workgroup's slabs: every wave owns its taps of dW; the four row partials of [A; sum g] are summed through LDS ----
        bc_barrier();
        {
            float* out = p.ws + (size_t)bx * 9 * 32 * 32;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int t = tfirst + k, tp = (2 - t / 3) * 3 + (2 - t % 3);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int cin = (i & 3) + 8 * (i >> 2) + 4 * hh;
                    out[((size_t)tp * 32 + cin) * 32 + r] = accw[k][i];
                }
            }
        }
        float* redA = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int i = 0; i < 16; ++i) redA[(wave * 16 + i) * 64 + lane] = accA[i];
        bc_barrier();
        if (wave == 1) {
            for (int q = lane; q < 320; q += 64) {
                const int k = q >> 5, c = q & 31;
                const int i = 4 * (k >> 3) + (k & 3), ln = c + 32 * ((k >> 2) & 1);
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) tot += redA[(w * 16 + i) * 64 + ln];
                p.a_part[((size_t)bx * 10 + k) * 32 + c] = tot;
                for (int row = bx + nbx; row < p.nparts; row += nbx) p.a_part[((size_t)row * 10 + k) * 32 + c] = 0.f;
            }
        }
        };
        if (wave == 0) consumer_ts(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
        else if (wave == 3) consumer_ts(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
        else consumer_ts(std::integral_constant<int, 2>{}, std::integral_constant<int, 3>{});
        }
    }
}

}  // namespace

int bwd_fused_c1_nwg(int B, int H) {
    const long long tiles = (long long)B * cdiv(H + 1, 4);
    long long n = 256;
    if (const char* e = sed_getenv("SED_BWD_FUSED_BLOCKS")) n = atoll(e) > 0 ? atoll(e) : n;      // tuning knob
    if (n > tiles) n = tiles;
    return (int)(n < 1 ? 1 : n);
}

// -1 = not covered (SED_BWD_FUSED_C1=0 keeps the two-kernel form); workgroups launched are returned through *nwg
int launch_bwd_fused_c1(const float* x1, const float* fmean, const float* fstd, const float* w1, const float* sc1, const float* sh1,
                        const void* dy, const void* z2, const float* sc2, const float* sh2, const float* ca, const float* cb,
                        const float* cc, const void* wpack_t, const void* mask, float* a_part, int nparts, float* ws, int B, int H,
                        int* nwg, hipStream_t st) {
    if (const char* e = sed_getenv("SED_BWD_FUSED_C1")) if (e[0] == '0') return -1;
    BwdC1Params p = {};
    p.x1 = x1; p.fmean = fmean; p.fstd = fstd; p.w1 = w1; p.sc1 = sc1; p.sh1 = sh1; p.dy = dy; p.z2 = z2; p.sc2 = sc2; p.sh2 = sh2;
    p.ca = ca; p.cb = cb; p.cc = cc; p.wpack_t = wpack_t; p.mask = reinterpret_cast<const unsigned*>(mask); p.a_part = a_part;
    p.ws = ws; p.B = B; p.H = H; p.nparts = nparts;
    if (const char* e = sed_getenv("SED_BC_PRIO")) p.prio = atoi(e);
    int n = bwd_fused_c1_nwg(B, H);
    if (n > nparts) n = nparts;
    *nwg = n;
    p.tilesPerImg = cdiv(H + 1, 4);
    p.totalTiles = B * p.tilesPerImg;
    p.tpb = cdiv(p.totalTiles, n);
    constexpr size_t lds = ((size_t)18 * 68 * 32 + (size_t)2 * 256 * 32 + 9 * 32 * 32) * sizeof(bf16_t) + (size_t)(3 * C1Tile<64, 6>::N + 2 * 6 * C1Tile<64, 6>::XP) * sizeof(bf16_t) +
                           (size_t)2 * 256 * sizeof(unsigned);
    static_assert(lds <= 160 * 1024 && lds >= (size_t)(3 * 9 + 4) * 16 * 64 * 4, "LDS budget (the final reductions reuse it)");
#ifdef SED_EXPERIMENTS
    // tap-split consumers (SED_BC_TS=1): parity-green, 0.73 vs 0.61 ms (30 spilled registers, 88 instead of 78 MFMAs on the busiest wave)
    if (const char* e = sed_getenv("SED_BC_TS"); e && e[0] == '1') {
        if (int rc_ = sed_set_max_lds<&conv_bwd_fused_c1_kernel<true>>(lds)) return rc_;
        conv_bwd_fused_c1_kernel<true><<<dim3(n), dim3(512), lds, st>>>(p);
        return 0;
    }
#endif
    bool lb = false;
    if (const char* e = sed_getenv("SED_BC_LB")) lb = e[0] == '1';
    if (lb && mask != nullptr) {
        if (int rc_ = sed_set_max_lds<&conv_bwd_fused_c1_kernel<false, true>>(lds)) return rc_;
        conv_bwd_fused_c1_kernel<false, true><<<dim3(n), dim3(512), lds, st>>>(p);
        return 0;
    }
    if (mask == nullptr) {       // derived gate (round 5, what the engine runs)
        if (int rc_ = sed_set_max_lds<&conv_bwd_fused_c1_kernel<false, false, true>>(lds)) return rc_;
        conv_bwd_fused_c1_kernel<false, false, true><<<dim3(n), dim3(512), lds, st>>>(p);
        return 0;
    }
    if (int rc_ = sed_set_max_lds<&conv_bwd_fused_c1_kernel<false>>(lds)) return rc_;
    conv_bwd_fused_c1_kernel<false><<<dim3(n), dim3(512), lds, st>>>(p);
    return 0;
}
