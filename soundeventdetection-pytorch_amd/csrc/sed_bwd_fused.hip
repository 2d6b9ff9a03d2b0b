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
//   * one s_barrier per stage; four ring positions, two activation tiles, two staging images.
// Output tile j of an image = rows [TH*j - 1, TH*j + TH - 1): it needs dz rows TH*j - 2 .. TH*j + TH - 1, i.e. the last two rows
// of chunk j-1 and chunk j -- available as soon as chunk j is in the ring.  A strip that starts inside an image spends one
// producer-only stage on chunk j-1.
// Ring: 4*TH + 2 rows; position k puts a stage's WINDOW (TH + 2 rows: two prefix rows + the chunk) at rows k*TH .. k*TH + TH + 1,
// so the window is CONTIGUOUS and every fragment address of a stage is one per-stage base plus compile-time offsets (the
// first version selected between two slot bases per read: ~170 scalar + ~120 vector instructions per stage of a consumer wave,
// and the stage time of this kernel is the instruction total of the two waves of a SIMD).  Positions advance 0, 1, 2, 3, 0 ...;
// the chunk at position 3 writes its last two rows a second time to rows 0, 1 (the prefix of position 0).  At the top of an
// image the prefix must be zero rows: that stage SKIPS a position (k + 2), whose prefix rows nobody is reading, and zeroes them.
#include "conv_common.h"

#include <stdlib.h>

namespace {

constexpr int kBfBlocks = 256;          // one workgroup per CU
#ifndef SED_BF_WREGS
#define SED_BF_WREGS 0      // operator fragments of the 64 -> 64 data gradient kept in registers (8: 246, 10: 254 registers, no spills): measured neutral
#endif                      // (profiles/r05_h_ab_block1_bwd_operator_fragments_in_registers.txt), off
#ifdef SED_EXPERIMENTS
#define BF_ABL(p, bit) ((p).abl & (bit))      // ablation switches (SED_BF_ABL; make EXPERIMENTS=1 only): 1 dead loads, 2 no dz / activation
#else                                         // arithmetic, 4 no epilogue, 8 no MFMA loops, 16 no dz / activation LDS stores
#define BF_ABL(p, bit) 0
#endif
#ifdef SED_STAMPS
constexpr bool kBfStamps = true;        // make STAMPS=1: s_memtime around the phases of a stage, printed by one workgroup (tools/bf_stamp.sh)
#else
constexpr bool kBfStamps = false;
#endif

__device__ __forceinline__ void bf_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int bf_xswz(int col) { return (col >> 2) & 3; }

// VAR (A/B builds, SED_BF_VAR): bit 0 = epilogue references from registers / LDS instead of a second global read,
// bit 1 = one dy load per 2x2 pooling window (DPP to the odd column) instead of four,
// bit 2 (round 5, RELUBWD with the BN+ReLU prologue) = the ReLU gate of the data gradient from the ACTIVATION tile still in LDS:
//   a = relu(bn1(z1)) != 0 <=> the decision the forward made, so the loaders gate the staged bf16 PAIRS of dx with v_pk_min_u16
//   (activation bits -> 0 / 1) + v_pk_mul_lo_u16 and store them as they are -- instead of converting dx to fp32, re-evaluating
//   fma(z1, scale, shift) > 0 per value, selecting and converting back (the loader waves are this launch's critical role and their
//   instructions cost ~15 ticks each beside an MFMA wave; profiles/r04_l_block1_bwd_phase_stamps.txt).  Same bits as before unless a
//   positive pre-activation rounds to bf16 zero (< 2^-133).  The BN1-backward sums still use z1 (kept registers): no new algebra.
template <int W, int CI_T, int CO_T, int DZ, int PRO, int EPI, int VAR = 7>
__global__ __launch_bounds__(512) void conv_bwd_fused_kernel(BwdFusedParams p) {
    typedef bf16_t T;
    constexpr bool REGREF = VAR & 1, DYDUP = VAR & 2;
    constexpr bool AGATE = (VAR & 4) && EPI == SED_EPI_RELUBWD && PRO == SED_PRO_BNRELU && REGREF;
    constexpr int CI = 32 * CI_T, CO = 32 * CO_T;
    constexpr int NPAIR = CI_T * CO_T, KSPLIT = 4 / NPAIR;
    static_assert(NPAIR == 2 || NPAIR == 4, "two or four (cin tile, cout tile) pairs per workgroup");
    constexpr int TH = 4 / CI_T;                       // rows per tile: TH * CI_T = 4 data-gradient units, one per consumer wave
    constexpr int BM = TH * W, WP = (W + 2 + 3) & ~3, ROWE = WP * 32;
    constexpr int DZIMG = (4 * TH + 2) * ROWE;          // four ring positions + the last window's tail, per 32-channel image
    constexpr int A1 = BM * 32, ABUF = CI_T * A1;
    constexpr int WSZ = CO_T * 36 * CI * 8;
    constexpr int BNP = CI + 8, OSZ = BM * BNP;
    constexpr int NP = 256, NTHR = 512;
    constexpr int KSW = BM / 16 / KSPLIT;              // k-steps (16 pixels) of a wave's weight-gradient share
    static_assert(W == 32 && KSW == 4 && (KSW * 16) % W == 0, "geometry: W = 32, four k-steps (two rows) per wave");
    constexpr bool RELUBWD = EPI == SED_EPI_RELUBWD, PSTATS = EPI == SED_EPI_POOLSTATS;
    // producer item geometry
    constexpr int IPP = CO / 8, DITEMS = BM * IPP, DIPT = DITEMS / NP, DQS = NP / IPP;
    static_assert(DQS == W && DIPT == TH, "a thread's dz items are the rows of one column");
    constexpr int IPX = CI / 8, XITEMS = BM * IPX, XIPT = XITEMS / NP, XQS = NP / IPX;
    static_assert(XITEMS % NP == 0 && NP % IPX == 0, "activation item geometry");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* dzr = reinterpret_cast<T*>(smem);               // [CO_T][4 * TH + 2][WP][32]  swizzled 16-byte slots
    T* ab = dzr + CO_T * DZIMG;                        // [2][CI_T][BM][32]
    T* wsm = ab + 2 * ABUF;                            // [CO_T][9][4][CI][8]       the data-gradient operator, resident
    T* os = wsm + WSZ;                                 // [2][BM][BNP]

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
    // Prefetch depth D: stages of global loads in flight per loader thread.  The 64-pixel tiles of the 64 -> 64 layer put only
    // ~20 KB per stage in flight; with two stages the kernel ran latency-bound at 2.3 TB/s (commit waited ~3000 cycles per stage
    // for loads issued two stages earlier), the 128-pixel tiles of the 32 -> 64 layer (88 KB in flight) at the HBM rate.
    constexpr int D = TH == 2 ? 4 : 2;
    const int NI = (NS + 2 + D - 1) / D * D;

    // ---- one-time LDS setup ------------------------------------------------------------------------------------------
    {
        bf16x8 z8;
#pragma unroll
        for (int e = 0; e < 8; ++e) z8[e] = (bf16_t)0.f;
        for (int i = tid; i < CO_T * DZIMG / 8; i += NTHR) *reinterpret_cast<bf16x8*>(dzr + i * 8) = z8;     // padding columns stay zero; every row finite
        const T* __restrict__ wg = reinterpret_cast<const T*>(p.wpack_t);
        for (int i = tid; i < WSZ / 8; i += NTHR) *reinterpret_cast<bf16x8*>(wsm + i * 8) = *reinterpret_cast<const bf16x8*>(wg + i * 8);
    }
    __syncthreads();

    // Stage bookkeeping, incremental (no divisions in the loops): image b, chunk / tile j, ring position, whether the stage exists
    // and whether the consumers have a tile to compute (a strip's producer-only first stage has none)
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
    auto st_next = [&](const StInfo& c, int s_next) -> StInfo {       // the stage after c; s_next = its index
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

    float S[8], Q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { S[e] = 0.f; Q[e] = 0.f; }

    if (wave >= 4) {
        // =============================== PRODUCERS =====================================================
        // The loader waves are dispatched after the MFMA waves and lose every issue arbitration against the older wave of their
        // SIMD (MI355X_MICROARCH.md, two waves per SIMD) -- and they are this kernel's critical path (stamps: the consumers wait
        // ~40 % of a stage at the barrier): static priority for the younger half, no per-phase flips.  SED_BF_PRIO (EXPERIMENTS).
        if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
        else if (p.prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (p.prio == 3) __builtin_amdgcn_s_setprio(3);
        const T* __restrict__ xg = reinterpret_cast<const T*>(p.x);
        const T* __restrict__ gg = reinterpret_cast<const T*>(p.gsrc);
        const T* __restrict__ zsg = reinterpret_cast<const T*>(p.zsrc);
        T* __restrict__ dxg = reinterpret_cast<T*>(p.dx);
        const int pt = tid - 256;
        const size_t ximg_ = (size_t)H * W * CI, zimg_ = (size_t)H * W * CO, pimg_ = (size_t)Ho * Wo * CO;

        // dz items: thread = (column dq0, channel group dc8), item u = chunk row u
        const int dq0 = pt / IPP, dc8 = (pt - dq0 * IPP) * 8;
        const unsigned dvoff0 = (unsigned)((dq0 * CO + dc8) * 2);
        const int dlds0 = (dc8 >> 5) * DZIMG + (dq0 + 1) * 32 + ((((dc8 & 31) >> 3) ^ bf_xswz(dq0 + 1)) * 8);
        // DZ_POOL, pool 2: the four pixels of a pooled pixel share one dy item -- the thread of the EVEN column loads it once per row
        // pair, the odd column's thread (8 lanes up in the same 16-lane row: pt = 8*column + channel group) takes it by DPP
        const bool dy_dup = DYDUP && DZ == DZ_POOL && psh == 1 && (dq0 & 1);
        unsigned pvoff[DIPT];
#pragma unroll
        for (int u = 0; u < DIPT; ++u)
            pvoff[u] = (dy_dup || (DYDUP && psh == 1 && (u & 1))) ? SED_OOB : (unsigned)((((u >> psh) * Wo + (dq0 >> psh)) * CO + dc8) * 2);
        // activation / output items: thread = (pixel xq0 + u * XQS, channel group xc8)
        const int xq0 = pt / IPX, xc8 = (pt - xq0 * IPX) * 8;
        const unsigned xvoff0 = (unsigned)((xq0 * CI + xc8) * 2);
        constexpr unsigned xvstep = (unsigned)(XQS * CI * 2);
        const int xlds0 = (xc8 >> 5) * A1 + xq0 * 32 + (xc8 & 31);
        // per-channel coefficients of the thread's fixed channel groups live in REGISTERS: read from LDS per item they cost
        // ~10 ds_read_b128 + two exposed LDS round trips per 16-byte item (the loader waves have the registers to spare)
        float kca[8], kcb[8], kcc[8], ksc[8], ksh[8], qsc[8], qsh[8];
        {
            const float inv_pool = (DZ == DZ_POOL && psh) ? 0.25f : 1.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                kca[e] = p.ca[dc8 + e] * inv_pool; kcb[e] = p.cb[dc8 + e]; kcc[e] = p.cc[dc8 + e];
                ksc[e] = DZ == DZ_POOL ? p.scale[dc8 + e] : 0.f; ksh[e] = DZ == DZ_POOL ? p.shift[dc8 + e] : 0.f;
                qsc[e] = PRO == SED_PRO_BNRELU ? p.pro_scale[xc8 + e] : 0.f; qsh[e] = PRO == SED_PRO_BNRELU ? p.pro_shift[xc8 + e] : 0.f;
            }
        }
        // (RELUBWD: the ReLU decision's scale / shift ARE the prologue's -- BN1 of the same block; the C entry point checks it)
        float cem[8];
        if (RELUBWD) {
#pragma unroll
            for (int e = 0; e < 8; ++e) cem[e] = p.epi_mean[xc8 + e];
        }

        // (DZ_POOL is pool 2 here: the two rows of a pooling window share their dy item, one load per row pair)
        constexpr int AIT = (DZ == DZ_POOL && DYDUP) ? DIPT / 2 : DIPT;
        struct RawSet { Raw8<T> x[XIPT]; Raw8<T> a[AIT]; Raw8<T> b[DIPT]; };
        // Epilogue references without a second trip through the CU's memory pipe (the kernel runs at its ~10 B/clk): the ReLU /
        // BN1-backward reference of conv2's data gradient is the very z tile the prologue consumed two iterations earlier (its
        // raw registers are kept: zkeep); the pooled activation of the pooled-tensor statistics is the raw activation tile
        // still sitting in LDS (read before this iteration's commit overwrites the buffer -- same thread, same items).
        Raw8<T> zkeep[2][RELUBWD ? XIPT : 1];
        Raw8<T> zraw[XIPT];                           // (!REGREF: the reference re-read from global memory)
        const T* __restrict__ zr = reinterpret_cast<const T*>(p.zref);
        u32x2 craw[PSTATS ? XIPT : 1];

        // every load is issued unconditionally: a dead stage gets zero-sized descriptors (zeros, no traffic), so hipcc's vmcnt
        // bookkeeping is exact and two stages stay in flight
        auto issue = [&](RawSet& r, const StInfo& si) {
            const bool live = si.live, mainst = si.mainst;
            const int b = si.b, j = si.j;
            const bool ld = live && !BF_ABL(p, 1);
            const size_t ximg = (ld && mainst) ? ximg_ : 0, zimg = ld ? zimg_ : 0, pimg = ld ? pimg_ : 0;
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
                    if (AIT == DIPT || !(u & 1)) r.a[AIT == DIPT ? u : u / 2] = buf_load8<T>(gs, pvoff[u] + ptq);
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
                *reinterpret_cast<bf16x8*>(dst - 2 * ROWE) = z8;
                *reinterpret_cast<bf16x8*>(dst - ROWE) = z8;
            }
            const int rows_in = H - TH * j;                 // rows of the chunk inside the image (pool floor: g = 0 by the range check)
#pragma unroll
            for (int u = 0; u < DIPT; ++u) {
                float g[8], z[8], v[8];
                if (DYDUP && DZ == DZ_POOL) {       // rows 2k, 2k+1 and columns 2c, 2c+1 share the item of (k, c)
                    u32x4 w4 = __builtin_bit_cast(u32x4, r.a[u / 2].v);
#pragma unroll
                    for (int e = 0; e < 4; ++e)        // row_shr:8 -- lanes 8..15 of a row take lanes 0..7, lanes 0..7 keep their own
                        w4[e] = (unsigned)__builtin_amdgcn_update_dpp((int)w4[e], (int)w4[e], 0x118, 0xF, 0xF, false);
                    Raw8<T> t8; t8.v = __builtin_bit_cast(bf16x8, w4);
                    raw_to_f(t8, g);
                } else {
                    raw_to_f(r.a[u], g);
                }
                raw_to_f(r.b[u], z);
                if (BF_ABL(p, 2)) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = z[i] + g[i];
                } else
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float base = fmaf(kcb[i], z[i], kcc[i]);
                    const float full = fmaf(kca[i], g[i], base);
                    if (DZ == DZ_POOL) v[i] = (fmaf(z[i], ksc[i], ksh[i]) > 0.f) ? full : base;
                    else v[i] = full;
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
                if (!BF_ABL(p, 16)) store8<T>(dst + u * ROWE, v);
                if (u >= TH - 2 && dup) store8<T>(dzr + dlds0 + (u - (TH - 2)) * ROWE, v);
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
                const f32x4 qs0 = {qsc[0], qsc[1], qsc[2], qsc[3]}, qs1 = {qsc[4], qsc[5], qsc[6], qsc[7]};
                const f32x4 qh0 = {qsh[0], qsh[1], qsh[2], qsh[3]}, qh1 = {qsh[4], qsh[5], qsh[6], qsh[7]};
#pragma unroll
                for (int u = 0; u < XIPT; ++u) {
                    bool keep = true;
                    if (boundary) {                          // rows outside the image stay zero (relu(shift) is not)
                        const int row = r0 + (xq0 + u * XQS) / W;
                        keep = row >= 0 && row < H;
                    }
                    *reinterpret_cast<bf16x8*>(adst + u * XQS * 32) = bnrelu8_bf16(r.x[u].v, qs0, qs1, qh0, qh1, keep);
                }
            }
        };

        // reference tile of the flush of the NEXT iteration (output tile of stage s - 1)
        auto issue_z = [&](const StInfo& si) {          // si = stage s - 1
            if (!RELUBWD && !PSTATS) return;
            const bool live = si.live, mainst = si.mainst;
            const int b = si.b, j = si.j;
            const size_t rimg = (live && mainst) ? ximg_ : 0;
            const unsigned tq = (unsigned)((TH * j - 1) * W * CI * 2);
            if constexpr (!REGREF) {
                const __amdgpu_buffer_rsrc_t rs = make_srd(zr + (size_t)b * rimg, rimg * 2);
#pragma unroll
                for (int u = 0; u < XIPT; ++u) zraw[u] = buf_load8<T>(rs, xvoff0 + (unsigned)u * xvstep + tq);
            }
            if constexpr (PSTATS) {
                const __amdgpu_buffer_rsrc_t cs = make_srd(p.cnt + (size_t)b * rimg, rimg);
#pragma unroll
                for (int u = 0; u < XIPT; ++u)
                    craw[u] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(cs, (xvoff0 + (unsigned)u * xvstep + tq) >> 1, 0, 0));
            }
        };
        // the output tile of stage s - 2 sits complete in its staging image
        auto flush = [&](const StInfo& si, int s) {     // si = stage s - 2
            const bool live = si.live, mainst = si.mainst;
            const int b = si.b, j = si.j;
            if (!live || !mainst || BF_ABL(p, 4)) return;
            const T* osb = os + ((s - 2) & 1) * OSZ;
            const T* aref = ab + (s & 1) * ABUF + xlds0;        // activation tile of stage s - 2 (PSTATS: the pooled activation itself)
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
                if constexpr (AGATE) {
                    // (rows outside the image hold zero activations: gated off, and their store offset is out of range anyway)
                    const u32x4 aw = __builtin_bit_cast(u32x4, *reinterpret_cast<const bf16x8*>(aref + u * XQS * 32));
                    const u32x4 rw = __builtin_bit_cast(u32x4, raw);
                    u32x4 gw;
                    float z[8];
                    raw_to_f(zkeep[s & 1][u], z);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        unsigned on, gq;
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(on) : "v"(aw[d]), "s"(0x00010001u));
                        asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(gq) : "v"(rw[d]), "v"(on));
                        gw[d] = gq;
                        const float g0 = __builtin_bit_cast(float, gq << 16), g1 = __builtin_bit_cast(float, gq & 0xffff0000u);
                        S[2 * d] += g0;
                        S[2 * d + 1] += g1;
                        // Q = sum g*z1 here; the mean comes off ONCE per workgroup in the final reduction (in double): sum g*(z1 - mean) =
                        // sum g*z1 - mean * sum g -- eight subtractions less per item
                        Q[2 * d] = fmaf(g0, z[2 * d], Q[2 * d]);
                        Q[2 * d + 1] = fmaf(g1, z[2 * d + 1], Q[2 * d + 1]);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(gw, ds, off, 0, 0);
                } else if (RELUBWD) {
                    float v[8], z[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (float)raw[e];
                    if constexpr (REGREF) raw_to_f(zkeep[s & 1][RELUBWD ? u : 0], z);
                    else raw_to_f(zraw[u], z);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gate = (valid && fmaf(z[e], qsc[e], qsh[e]) > 0.f) ? v[e] : 0.f;
                        v[e] = gate;
                        S[e] += gate;
                        Q[e] = fmaf(gate, z[e] - cem[e], Q[e]);
                    }
                    buf_store8<T>(ds, off, v);
                } else {
                    if constexpr (PSTATS) {       // S = sum dy*cnt, Q = sum dy*y_pooled (reference loads of rows outside the image: 0)
                        float ya[8];
                        Raw8<T> yr; yr.v = *reinterpret_cast<const bf16x8*>(aref + u * XQS * 32);
                        if constexpr (REGREF) raw_to_f(yr, ya);
                        else raw_to_f(zraw[u], ya);
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

        RawSet r0, r1, r2, r3;
        StInfo sm2 = st_dead, sm1 = st_dead, sc = st_first();
        StInfo sf = sc;                                     // the stage whose loads are issued next (s + D)
        issue(r0, sf); sf = st_next(sf, 1);
        issue(r1, sf); sf = st_next(sf, 2);
        if constexpr (D == 4) {
            issue(r2, sf); sf = st_next(sf, 3);
            issue(r3, sf); sf = st_next(sf, 4);
        }
        unsigned long long tp[4] = {0, 0, 0, 0};
        auto stamp = [&]() -> unsigned long long { return kBfStamps ? __builtin_amdgcn_s_memtime() : 0ull; };
        constexpr bool FLUSH_FIRST = (PSTATS && REGREF) || AGATE;      // (its reference is the activation tile that commit(s) overwrites)
        auto iter = [&](int s, RawSet& r) {
            const unsigned long long s0 = stamp();
            if constexpr (FLUSH_FIRST) flush(sm2, s);
            const unsigned long long s1 = stamp();
            commit(r, sc, s);
            const unsigned long long s2 = stamp();
            if constexpr (!FLUSH_FIRST) flush(sm2, s);
            if constexpr (RELUBWD && REGREF) {
#pragma unroll
                for (int u = 0; u < XIPT; ++u) zkeep[s & 1][u] = r.x[u];       // the reference of flush(s + 2)
            }
            issue_z(sm1);
            issue(r, sf);
            sm2 = sm1; sm1 = sc; sc = st_next(sc, s + 1); sf = st_next(sf, s + D + 1);
            const unsigned long long s3 = stamp();
            bf_barrier();
            if (kBfStamps) { tp[0] += s1 - s0; tp[1] += s2 - s1; tp[2] += s3 - s2; tp[3] += stamp() - s3; }
        };
        for (int s = 0; s < NI; s += D) {
            iter(s, r0);
            iter(s + 1, r1);
            if constexpr (D == 4) {
                iter(s + 2, r2);
                iter(s + 3, r3);
            }
        }
        if (kBfStamps && blockIdx.x == 8 && lane == 0 && wave == 5)
            printf("bf producer: %d stages; cycles flush(REGREF) %llu commit %llu flush+issue %llu barrier %llu\n", NI, tp[0], tp[1], tp[2], tp[3]);
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

        // Round 5: this launch sits on the LDS bandwidth of its tiling (64 -> 64: ~100 KB of fragment reads per wave and 64-pixel stage, two
        // operands of every data-gradient MFMA from LDS; with the loaders' images ~440 KB per stage = ~3440 cycles at 128 B/clk against a
        // ~3500-cycle stage).  The first NWREG of the 36 operator fragments of the wave's data-gradient unit stay in the registers the
        // kernel has left (SED_BF_WREGS, A/B builds).
        constexpr int NWREG = (CI_T == 2 && CO_T == 2) ? SED_BF_WREGS : 0;
        bf16x8 wreg[NWREG > 0 ? NWREG : 1];
        if constexpr (NWREG > 0) {
#pragma unroll
            for (int k = 0; k < NWREG; ++k) {
                const int c = k / 18, kk = k % 18, tap = kk >> 1, ks = kk & 1;
                wreg[k] = *reinterpret_cast<const bf16x8*>(wsm + woff + ((c * 36 + tap * 4 + ks * 2) * CI) * 8);
            }
        }
        unsigned long long tc[4] = {0, 0, 0, 0};
        auto cstamp = [&]() -> unsigned long long { return kBfStamps ? __builtin_amdgcn_s_memtime() : 0ull; };
        StInfo csi = st_first();
        auto citer = [&](int s) {
            const unsigned long long c0 = cstamp();
            bf_barrier();
            const unsigned long long c1 = cstamp();
            tc[0] += c1 - c0;
            const StInfo cs = csi;
            csi = st_next(csi, s + 1);
            if (!cs.live || !cs.mainst || BF_ABL(p, 8)) return;
            // the stage's window: TH + 2 consecutive ring rows from row pos*TH (row hr of it = image row TH*j - 2 + hr)
            const T* __restrict__ win = dzr + cs.pos * TH * ROWE;
            const T* __restrict__ abuf = ab + (s & 1) * ABUF;

#if !defined(SED_BF_COLK) || SED_BF_COLK
            // ---- weight gradient: accw[si*3+sj] += a[k-step] (x) dz[k-step shifted by (si, sj)] -----------------------------
            // The wave's k share (KSW k-steps = RW tile rows from row kw*RW) is walked as 16-pixel COLUMN STRIPS: the dz fragment of
            // window row rho / column shift sj of a strip serves every tile row `row` of the share with 0 <= rho - row <= 2 (shift
            // row si = rho - row): 3*(RW + 2) fragment reads per strip instead of 9*RW (24 instead of 36 per stage) -- these kernels
            // keep the LDS pipe busier than the matrix pipe, the bytes read are what counts (round 4, tools/pmc_lds.sh).
            {
                constexpr int RW = KSW * 16 / W, NSTRIP = W / 16;          // 2 rows x 2 strips
                bf16x8 afr[NSTRIP][RW], bfr[2][3];
                const T* __restrict__ abase = abuf + kw * KSW * 16 * 32;
                const T* __restrict__ wbase = win + kw * RW * ROWE;
                auto ld_a = [&](int strip, int row, bf16x8& dst) {
                    const int imm = (row * W + strip * 16) * 32;
                    dst = join_tr(ds_read_tr16_b64(abase + imm + offA[0]), ds_read_tr16_b64(abase + imm + offA[1]));
                };
                auto ld_b = [&](int strip, int rho, bf16x8 (&dst)[3]) {
                    const int imm = rho * ROWE + strip * 16 * 32;
#pragma unroll
                    for (int sj = 0; sj < 3; ++sj)
                        dst[sj] = join_tr(ds_read_tr16_b64(wbase + imm + offB[sj][0]), ds_read_tr16_b64(wbase + imm + offB[sj][1]));
                };
                constexpr int NR = RW + 2, NST = NSTRIP * NR;             // step = (strip, window row rho)
                ld_b(0, 0, bfr[0]);
#pragma unroll
                for (int strip = 0; strip < NSTRIP; ++strip)
#pragma unroll
                    for (int row = 0; row < RW; ++row) ld_a(strip, row, afr[strip][row]);
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const int strip = st / NR, rho = st % NR;
                    if (st + 1 < NST) ld_b((st + 1) / NR, (st + 1) % NR, bfr[(st + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int row = (rho > 2 ? rho - 2 : 0); row <= (rho < RW - 1 ? rho : RW - 1); ++row)
#pragma unroll
                        for (int sj = 0; sj < 3; ++sj)
                            accw[(rho - row) * 3 + sj] = mfma(afr[strip][row], bfr[st & 1][sj], accw[(rho - row) * 3 + sj]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#else      // (A/B builds, tools/ab_build.sh SED_BF_COLK: the k-step-wise walk of round 3)
            // ---- weight gradient: accw[si*3+sj] += a[k-step] (x) dz[k-step shifted by (si, sj)] -----------------------------
            {
                constexpr int NSTEP = KSW * 3;           // step = (k-step, shift row): 3 MFMAs
                bf16x8 bfr[3][3], afr[2];
                // this wave's k share starts KSW*16 pixels = 2 rows further down per kw: folded into the two per-stage bases
                const T* __restrict__ abase = abuf + kw * KSW * 16 * 32;
                const T* __restrict__ wbase = win + kw * (KSW * 16 / W) * ROWE;
                auto ld_a = [&](int kk, bf16x8& dst) {
                    dst = join_tr(ds_read_tr16_b64(abase + kk * 16 * 32 + offA[0]), ds_read_tr16_b64(abase + kk * 16 * 32 + offA[1]));
                };
                auto ld_b = [&](int st, bf16x8 (&dst)[3]) {
                    const int kk = st / 3, si = st % 3;
                    constexpr int dummy = 0; (void)dummy;
                    const int imm = ((kk * 16) / W + si) * ROWE + ((kk * 16) % W) * 32;
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
            const unsigned long long c2 = cstamp();
            // ---- data gradient: D[cin][pixel] over (cout chunk, tap, 16-channel half) -----------------------------------------
            f32x16 accd;
#pragma unroll
            for (int i = 0; i < 16; ++i) accd[i] = 0.f;
            {
                constexpr int NK = CO_T * 18;
                bf16x8 xf[3], wf[3];
                const T* __restrict__ dbase = win + drow * ROWE;
                auto ld = [&](int k, bf16x8& xd, bf16x8& wd) {
                    const int c = k / 18, kk = k % 18, tap = kk >> 1, ks = kk & 1, ti = tap / 3, tj = tap % 3;
                    xd = *reinterpret_cast<const bf16x8*>(dbase + (c * DZIMG + ti * ROWE) + xoff[tj][ks]);
                    if (k < NWREG) wd = wreg[k < NWREG ? k : 0];          // (operator fragments kept in registers: no LDS read)
                    else wd = *reinterpret_cast<const bf16x8*>(wsm + woff + ((c * 36 + tap * 4 + ks * 2) * CI) * 8);
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
            const unsigned long long c3 = cstamp();
            T* osb = os + (s & 1) * OSZ;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = accd[4 * g + e];
                store4<T>(osb + ostg + 8 * g, v);
            }
            if (kBfStamps) { tc[1] += c2 - c1; tc[2] += c3 - c2; tc[3] += cstamp() - c3; }
        };
        for (int s = 0; s < NI; s += 2) {
            citer(s);
            citer(s + 1);
        }
        if (kBfStamps && blockIdx.x == 8 && lane == 0 && wave == 1)
            printf("bf consumer: %d stages; cycles barrier %llu wgrad %llu dgrad %llu staging %llu\n", NI, tc[0], tc[1], tc[2], tc[3]);
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
            if constexpr (AGATE) {
                double ts = 0.0, tq = 0.0;
                for (int k = 0; k < XQS; ++k) { ts += (double)red[(cg + IPX * k) * 16 + e]; tq += (double)red[(cg + IPX * k) * 16 + 8 + e]; }
                tot = stat ? (float)((tq - (double)p.epi_mean[cn] * ts) * (double)p.epi_invstd[cn]) : (float)ts;
            } else {
                for (int k = 0; k < XQS; ++k) tot += red[(cg + IPX * k) * 16 + stat * 8 + e];
                if (RELUBWD && stat) tot *= p.epi_invstd[cn];
            }
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

template <int W, int CI_T, int CO_T, int DZ, int PRO, int EPI, int VAR = 7>
int launch_bf_v(BwdFusedParams& p, hipStream_t st) {
    constexpr int CI = 32 * CI_T, TH = 4 / CI_T, BM = TH * W, WP = (W + 2 + 3) & ~3;
    constexpr size_t lds = ((size_t)CO_T * (4 * TH + 2) * WP * 32 + (size_t)2 * CI_T * BM * 32 + (size_t)CO_T * 36 * CI * 8 +
                            (size_t)2 * BM * (CI + 8)) * sizeof(bf16_t);
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert(CI_T * CO_T == 4 || lds >= (size_t)CI_T * CO_T * 9 * 16 * 64 * 4, "the k-share reduction at the end reuses the LDS");
    if (p.dry) return 0;
    if (int rc_ = sed_set_max_lds<&conv_bwd_fused_kernel<W, CI_T, CO_T, DZ, PRO, EPI, VAR>>(lds)) return rc_;
    p.tilesPerImg = cdiv(p.H + 1, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    conv_bwd_fused_kernel<W, CI_T, CO_T, DZ, PRO, EPI, VAR><<<dim3(p.nwg), dim3(512), lds, st>>>(p);
    return 0;
}
template <int W, int CI_T, int CO_T, int DZ, int PRO, int EPI>
int launch_bf(BwdFusedParams& p, hipStream_t st) {
#ifdef SED_EXPERIMENTS
    if (const char* e = sed_getenv("SED_BF_VAR")) {
        if (e[0] == '0') return launch_bf_v<W, CI_T, CO_T, DZ, PRO, EPI, 0>(p, st);
        if (e[0] == '1') return launch_bf_v<W, CI_T, CO_T, DZ, PRO, EPI, 1>(p, st);
        if (e[0] == '2') return launch_bf_v<W, CI_T, CO_T, DZ, PRO, EPI, 2>(p, st);
    }
#endif
    if constexpr (EPI == SED_EPI_RELUBWD) {        // SED_BF_AGATE=0: the round-4 epilogue (gate re-evaluated from z1), for the A/B
        if (const char* e = sed_getenv("SED_BF_AGATE"); e && e[0] == '0') return launch_bf_v<W, CI_T, CO_T, DZ, PRO, EPI, 3>(p, st);
        return launch_bf_v<W, CI_T, CO_T, DZ, PRO, EPI, 7>(p, st);
    }
    return launch_bf_v<W, CI_T, CO_T, DZ, PRO, EPI, 3>(p, st);
}

}  // namespace

// workgroups the fused backward kernel launches for this problem (= slabs of its weight-gradient workspace); 0 = shape not covered
int bwd_fused_nwg(int B, int H, int W, int Cinp, int Coutp, int dzmode, int pro, int epi) {
#ifdef SED_EXPERIMENTS
    if (W != 32) return bwd_fused_cs_nstrips(B, H, W, Cinp, Coutp, dzmode, pro, epi, 2);       // (pool 1 / 2: the same strip count)
#endif
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

int bwd_fused_max_nwg(int B, int H, int W, int Cinp, int Coutp) {
    const int a = bwd_fused_nwg(B, H, W, Cinp, Coutp, DZ_BN, SED_PRO_NONE, SED_EPI_STORE);
    const int b = bwd_fused_nwg(B, H, W, Cinp, Coutp, DZ_POOL, SED_PRO_BNRELU, SED_EPI_RELUBWD);
    return a > b ? a : b;
}

int launch_bwd_fused(BwdFusedParams& p, int W, hipStream_t st) {
#ifdef SED_EXPERIMENTS
    if (W != 32) return launch_bwd_fused_cs(p, W, st);
#endif
    p.prio = 0;
    if (const char* e = sed_getenv("SED_BF_PRIO")) p.prio = atoi(e);
    p.abl = 0;
    if (const char* e = sed_getenv("SED_BF_ABL")) p.abl = atoi(e);
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
