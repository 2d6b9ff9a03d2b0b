// 3x3 convolution forward / data gradient, ALL weights of a wave's 32 output channels resident in registers, ONE wave per
// SIMD (bf16, gfx950).
//
// Same mathematics and epilogues as conv_pc_kernel (sed_conv_pc.hip) -- nn.Conv2d(3x3, s1, p1, bias=False) of ConvBlock,
// /root/reference/models/spectogram_models.py:132-140,155-156.  Successor of conv_wir_kernel (sed_conv_wir.hip), whose two
// waves per SIMD split the input channels and paid ~900 cycles per step for exchanging half accumulators through the LDS
// with the matrix pipe idle.  Here a 256-thread workgroup runs one wave per SIMD with the full
// 512-register budget:
//   * wave (cb, ph) keeps the 9 * CIN/16 MFMA A-fragments of output channels [32 cb, 32 cb + 32) in registers (288 at
//     CIN = 128) and computes the 32-pixel block ph of every step: no k split, no exchange, no second barrier;
//   * the input walks through a ring of image rows in LDS filled by LDS-DMA (one shared zero row between consecutive
//     images, every row fetched once; swizzle on the source address; wir_common.h), BatchNorm+ReLU prologue = 16-byte
//     read-modify-write of the rows a wave fetched itself;
//   * ALL side work of a step sits in slots between the MFMAs of its k loop (the matrix pipe runs 32 cycles per MFMA, a
//     lone wave can issue ~5 other instructions in that time): DMA of row group i+2 | counted vmcnt | prologue of row
//     group i+1 | bf16 staging of step i-1's accumulators | whole-line stores + statistics of step i-2;
//   * one s_barrier per step.
#include "conv_common.h"
#include "wir_common.h"

#include <stdlib.h>

#include <type_traits>

namespace {

constexpr int kW4Blocks = 256;
// in-kernel phase stamps (make STAMPS=1; printed by one workgroup when SED_DBG & 16)
#ifdef SED_STAMPS
constexpr bool kW4Stamps = true;
#else
constexpr bool kW4Stamps = false;
#endif

template <int W, int CIN, int COUT>
struct W4Geom {
    static constexpr int NCB = COUT / 32;             // 32-channel output blocks = waves along the channels
    static constexpr int NPH = 4 / NCB;               // 32-pixel blocks per step
    static constexpr int RB = 32 / W;                 // image rows per 32-pixel block
    static constexpr int SR = NPH * RB;               // rows per step
    static constexpr int WP = W + 2;
    static constexpr int PIX = CIN * 2;               // bytes per pixel
    static constexpr int ROWB = WP * PIX;
    // row groups in flight ahead of the MFMAs: the DMA of iteration i fetches row group i + LA.  LA = 3 (its rows are needed two
    // iterations later: ~3 us against an HBM round trip of 1-2 us under load) where the deeper ring fits the LDS, else 2
    static constexpr int need3 = 4 * SR + 2, R3 = need3 <= 8 ? 8 : need3 <= 16 ? 16 : 32;
    static constexpr int LA = ((size_t)R3 * ROWB <= 96 * 1024) ? 3 : 2;
    static constexpr int need = (LA + 1) * SR + 2;
    static constexpr int R = need <= 8 ? 8 : need <= 16 ? 16 : 32;                    // ring rows (power of two)
    static constexpr int SLOTS = CIN / 8;             // 16-byte slots per pixel
    static constexpr int Q = CIN / 16;                // k16-steps per tap
    static constexpr int FR = 9 * Q;                  // A fragments (= MFMAs per 32 pixels) per wave
    static constexpr int CHP = 1024 / PIX;            // pixels per DMA chunk (one wave-instruction)
    static constexpr int CPR = W / CHP;               // chunks per row
    static constexpr int NCH = SR * CPR;              // chunks per row group
    static constexpr int CPW = (NCH + 3) / 4;         // chunks per wave
    static constexpr int OP = COUT + 8;               // staging pitch (elements)
    static constexpr int NPX = NPH * 32;              // pixels per step
    static constexpr int IPR = COUT / 8;              // 16-byte items per pixel row of the output
    static constexpr size_t RING_B = (size_t)R * ROWB;
    static constexpr size_t OST_B = (size_t)2 * NPX * OP * 2;
    static constexpr size_t COEF_B = (size_t)(2 * CIN + 3 * COUT) * 4;
    static constexpr size_t ZST_B = (size_t)3 * 512 * 16;       // RELUBWD: three reference tiles, two 16-byte items per thread
    static constexpr size_t LDS = RING_B + OST_B + COEF_B + ZST_B;
    static_assert(NCB >= 1 && NCB <= 4 && 4 % NCB == 0, "waves = (channel block, pixel block)");
    static_assert(W <= 32 && 32 % W == 0 && SR >= 2 && SR % 2 == 0 && CPR >= 1 && NCH % 4 == 0 && NPX * IPR == 512, "geometry");
    static_assert(need <= R, "ring depth");
    // one barrier in the MIDDLE of a step instead of one at its end: the fragment reads then run on across the step boundary
    // (no pipeline drain / refill around a barrier); the slowest wave may still read step i-1's rows while the fastest fetches
    // row group i + LA, which the ring must absorb
    static constexpr bool MIDB = R >= (LA + 2) * SR + 2;
    static_assert(RING_B + OST_B >= 256 * 16 * 4, "the statistics reduction overlays the ring and the staging images");
    static_assert(256 % IPR == 0, "a thread's two flush items share their channel group");
};

// compile-time loop: the k loop's body differs per MFMA slot (`if constexpr` on the slot number); `#pragma unroll` gives up on
// bodies beyond LLVM's pragma-unroll threshold and then indexes the weight fragments dynamically (scratch + select chains)
template <int I, int N, class F>
__device__ __forceinline__ void w4_static_for(F&& fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        w4_static_for<I + 1, N>(fn);
    }
}

// gap of vector piece pc: the p_fl pre-barrier pieces spread over gaps [1, g_split), the n_fl flush pieces over [g_split, fr - 1)
constexpr int w4_piece_gap(int pc, int p_fl, int n_fl, int g_split, int fr) {
    return pc < p_fl ? 1 + pc * (g_split - 1) / p_fl : g_split + (pc - p_fl) * (fr - 1 - g_split) / n_fl;
}

template <int N>
__device__ __forceinline__ void w4_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int W, int CIN, int COUT, int PRO, int EPI>
__global__ __launch_bounds__(256) void conv_w4_kernel(ConvParams p) {
    typedef bf16_t T;
    typedef W4Geom<W, CIN, COUT> G;
    constexpr int NCB = G::NCB, RB = G::RB, SR = G::SR, R = G::R, PIX = G::PIX, ROWB = G::ROWB, SLOTS = G::SLOTS;
    constexpr int LA = G::LA;
    constexpr int Q = G::Q, FR = G::FR, CHP = G::CHP, CPR = G::CPR, CPW = G::CPW, OP = G::OP, NPX = G::NPX, IPR = G::IPR;
    constexpr bool RELUBWD = EPI == SED_EPI_RELUBWD;
    constexpr int NPIN = FR < 60 ? FR : 60;            // weight fragments pinned in AGPRs (4 registers each; 16 AGPRs = accumulators)
    constexpr int DEAD = 1 << 24;                      // row offset of a step outside this workgroup's strip: no row is "real"

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    T* ost = reinterpret_cast<T*>(smem + G::RING_B);                                   // [2][NPX][OP]
    float* coef = reinterpret_cast<float*>(smem + G::RING_B + G::OST_B);               // [2][CIN] pro, [3][COUT] epi
    char* zst = smem + G::RING_B + G::OST_B + G::COEF_B;                                // [3][512] 16-byte items

    const unsigned long long tk0 = kW4Stamps ? __builtin_amdgcn_s_memtime() : 0ull;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (scalar: everything derived from it stays in SGPRs)
    const int H = p.H;
    const int SPI = p.tilesPerImg, HV = SPI * SR;      // an image = HV virtual rows: row 0 zero, rows 1..H the image, the rest zero
    const int cb = wave % NCB, ph = wave / NCB;

    // step s produces virtual rows [s*SR, (s+1)*SR) = rows r0 .. r0+SR-1 of image b (s = b*SPI + r0/SR) and reads rows
    // r0-1 .. r0+SR; row group g = rows r0+1 .. r0+SR of the same (b, r0) -- the last rows step g needs
    const int NS = p.totalTiles;                       // = B * SPI
    const int s_begin = blockIdx.x * p.tpb;
    const int s_end = min(NS, s_begin + p.tpb);

    // ---- one-time setup: zero the ring (padding columns stay zero for good), coefficients, the resident weights ----------
    {
        const bf16x8 z8 = {};
        for (int i = tid; i < (int)(G::RING_B / 16); i += 256) reinterpret_cast<bf16x8*>(ring)[i] = z8;
        if (PRO == SED_PRO_BNRELU)
            for (int i = tid; i < 2 * CIN; i += 256) coef[i] = i < CIN ? p.pro_scale[i] : p.pro_shift[i - CIN];
        if (RELUBWD)
            for (int i = tid; i < 3 * COUT; i += 256)
                coef[2 * CIN + i] = i < COUT ? p.epi_scale[i] : i < 2 * COUT ? p.epi_shift[i - COUT] : p.epi_mean[i - 2 * COUT];
    }
    bf16x8 wreg[FR];
    {
        // wpack: [Cin/32][tap][4][Coutp][8] (k quarter kq = 8 input channels): fragment (tap, q) = input channels 16 q .. 16 q + 15
        const T* __restrict__ wg = reinterpret_cast<const T*>(p.wpack);
        const int n0 = tid & 31, hh0 = (tid >> 5) & 1;
        // pin the fragments in the accumulation half of the register file (the MFMA reads its A operand from there directly):
        // 240 AGPRs + the 256 VGPRs hold all 288 weight registers; left to itself hipcc parks some fragments in AGPRs as
        // spill slots and copies them back (v_accvgpr_read x 4) in front of their MFMA, every step.  The pin consumes the
        // loaded value, so the loads go out in batches of 36 (one memory round trip per batch, not per fragment).
        constexpr int WB = 36;
#pragma unroll
        for (int f0 = 0; f0 < FR; f0 += WB) {
#pragma unroll
            for (int f = f0; f < f0 + WB && f < FR; ++f) {
                const int tap = f / Q, q = f % Q;
                const int c = q >> 1, kq = 2 * (q & 1) + hh0;
                wreg[f] = *reinterpret_cast<const bf16x8*>(wg + ((size_t)((c * 9 + tap) * 4 + kq) * COUT + cb * 32 + n0) * 8);
            }
#pragma unroll
            for (int f = f0; f < f0 + WB && f < FR; ++f)
                if (f < NPIN) asm volatile("" : "=a"(wreg[f]) : "0"(wreg[f]));
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xsrd = make_srd(p.x, (size_t)p.B * H * W * PIX);
    const __amdgpu_buffer_rsrc_t zsrd = make_srd(p.z, (size_t)p.B * H * W * COUT * 2);
    const __amdgpu_buffer_rsrc_t rsrd = make_srd(RELUBWD ? p.zref : p.z, (size_t)p.B * H * W * COUT * 2);

    float S[8], Qs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { S[e] = 0.f; Qs[e] = 0.f; }
    f32x16 acc = {};

    // (b*H, r0) of the steps / row groups i-2 .. i+LA of the current iteration (entry k <-> index i + k - 2): scalars, shifted
    // once per iteration.  An index outside [s_begin-1, s_end) carries r0 = DEAD: none of its rows is an image row, so its
    // loads read out of range (zeros), its prologue writes zeros and its flush stores nothing -- no separate "live"
    // predicates anywhere.
    constexpr int NT = LA + 3;
    int tbH[NT], tr0[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) { tbH[k] = 0; tr0[k] = DEAD; }
    int nbH, nr0;                                      // running (b*H, r0) of index i + LA + 1
    {
        const int g = s_begin - 1;
        const int b = g >= 0 ? g / SPI : 0;
        tbH[NT - 1] = b * H;
        tr0[NT - 1] = g >= 0 ? (g - b * SPI) * SR : DEAD;
        nbH = g >= 0 ? tbH[NT - 1] : 0;
        nr0 = g >= 0 ? tr0[NT - 1] : -SR;
    }
    int zw = 0;                                        // reference-tile buffer written this iteration (step i); step i-2's = (zw + 1) % 3
    constexpr int NZ = RELUBWD ? 2 : 0;                // reference-tile DMAs per iteration
    constexpr bool MIDB = G::MIDB;
    constexpr int NXF = FR % 8 == 0 ? 8 : 9;           // fragment ring depth (divides FR: the ring runs on across steps)
    static_assert(FR % NXF == 0, "fragment f of every step sits in ring slot f % NXF");

    // ---- the step's side work, cut into PIECES of a few instructions, one or two per MFMA gap ------------------------------
    // A lone wave hides only what fits beside the matrix pipe's 32 cycles per MFMA: the MFMA's own issue (8 cycles), the
    // fragment read (~10) and two or three vector instructions (tools/micro/mfma_lds.hip: 38.5 cycles per MFMA with the read,
    // 46.9 with four more v_fma, 55.7 with six) -- a 60-instruction block between two MFMAs is paid in full.
    //   gap 0: the DMAs of row group i+LA (+ reference tile of step i)       gap 1: counted vmcnt
    //   LDS-read track (RPG reads per gap from gap 2): raw rows of group i+1, prologue coefficients, staged tile of step
    //                  i-2, its reference tile
    //   vector track:  next step's fragment addresses (9) | step i-1's accumulators -> bf16 staging image (4) | prologue of row
    //                  group i+1 per element, write-back | -- barrier (gap GB) -- | flush of step i-2: statistics / ReLU gate
    //                  per element, stores
    //   from gap FR - NXF + 1 on the fragment reads are those of step i+1
    constexpr bool BNPRO = PRO == SED_PRO_BNRELU;
    constexpr int EPP = FR >= 72 ? 1 : 2;              // elements per flush / prologue piece
    constexpr int NEP = 8 / EPP;                       // element pieces per 16-byte item
    constexpr int N_ADR = 9, N_FIN = 4;
    constexpr int N_FL = 2 * (NEP + 1);                // per item: element pieces + store
    constexpr int N_PRO = CPW * (BNPRO ? NEP + 1 : 1); // per chunk: element pieces + write-back
    // vector pieces in program order: addresses, staging, prologue (all that other waves wait for: before the barrier), flush
    constexpr int P_FIN = N_ADR, P_PRO = P_FIN + N_FIN, P_FL = P_PRO + N_PRO, NVP = P_FL + N_FL;
    constexpr int W_PRE = N_ADR + N_FIN + N_PRO * (BNPRO ? 2 : 1), W_FL = N_FL * (RELUBWD ? 3 : 1);     // rough instruction weights
    constexpr int G_SPLIT0 = 1 + (FR - 2) * W_PRE / (W_PRE + W_FL);
    constexpr int G_SPLIT = MIDB && G_SPLIT0 > FR - NXF - 9 ? FR - NXF - 9 : G_SPLIT0;      // pre pieces: gaps [1, G_SPLIT), flush: [G_SPLIT, FR - 1)
    constexpr int GB = G_SPLIT + 8;                    // the barrier's gap: every LDS write of the pre pieces is >= NXF LDS operations old
    static_assert(!MIDB || (GB <= FR - NXF && G_SPLIT > 4), "barrier before the first read of the next step's rows");
    // LDS-read track: raw rows of group i+1, prologue coefficients, staged tile of step i-2, its reference tile
    constexpr int L_PRAW = 0, L_COEF = L_PRAW + CPW, L_FRAW = L_COEF + (BNPRO ? 4 * CPW : 0), L_FZR = L_FRAW + 2, NL = L_FZR + (RELUBWD ? 2 : 0);
    constexpr int RPG = (NL <= 12 && FR >= 72) ? 1 : 2, GL0 = 2;
    static_assert(GL0 + (NL + RPG - 1) / RPG < FR, "LDS-read track");
    static_assert(1 + P_PRO * (G_SPLIT - 1) / P_FL >= GL0 + (L_FRAW - 1) / RPG + 3 || !BNPRO, "the prologue pieces follow their LDS reads by >= 3 gaps");
    static_assert(G_SPLIT >= GL0 + (NL - 1) / RPG + 3, "the flush pieces follow their LDS reads by >= 3 gaps");

    // loop-invariant lane terms (a handful of registers; everything that changes with the step is scalar)
    const int lane = tid & 63, n = tid & 31, hh = (tid >> 5) & 1;
    const int prow = n / W, pcol = n % W;
    unsigned dma_lane[CPW];                            // source offset inside the row: pixel + swizzled slot
    int pro_ci[CPW];                                   // f32x4 index of this lane's 8 prologue coefficients
#pragma unroll
    for (int u = 0; u < CPW; ++u) {
        const int qc = wave + 4 * u;
        const int j = qc / CPR, c0 = (qc % CPR) * CHP;
        const int col = c0 + lane / SLOTS;
        const int sslot = (lane % SLOTS) ^ wir_z<W, SLOTS>(col + 1, 1 + j);     // (SR is even: row parity of v = parity of 1 + j)
        dma_lane[u] = (unsigned)(col * PIX + sslot * 16);
        pro_ci[u] = sslot * 2;
    }
    const int fcg = tid % IPR;
    const int fpx0 = tid / IPR, fpx1 = fpx0 + 256 / IPR;                       // flush items tid and tid + 256 of the step's tile
    const int o_lane = (ph * 32 + n) * OP + cb * 32 + 4 * hh;                  // staging image: this lane's pixel / channels
    f32x4 ecs[2], ect[2], ecm[2];                      // RELUBWD: BN scale / shift / mean of the thread's 8 channels
    if (RELUBWD) {
        const f32x4* ec = reinterpret_cast<const f32x4*>(coef + 2 * CIN);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) { ecs[hf] = ec[fcg * 2 + hf]; ect[hf] = ec[COUT / 4 + fcg * 2 + hf]; ecm[hf] = ec[COUT / 2 + fcg * 2 + hf]; }
    }

    // fragment addresses of a step: (col term ^ k offset) + ring row base, one v_xad_u32 per fragment; those of step i+1 are
    // computed inside iteration i, so that the first reads follow the barrier at once
    int rb[3], ct[3][3];
    auto addr_piece = [&](int step, int ti, int tj, int (&rbo)[3], int (&cto)[3][3]) __attribute__((always_inline)) {
        const int vin = step * SR + ph * RB + prow + ti - 1;
        if (tj == 0) rbo[ti] = (vin & (R - 1)) * ROWB;
        const int cl = pcol + tj;
        cto[ti][tj] = cl * PIX + ((hh ^ wir_z<W, SLOTS>(cl, vin)) << 4);
    };
#pragma unroll
    for (int ti = 0; ti < 3; ++ti)
#pragma unroll
        for (int tj = 0; tj < 3; ++tj) addr_piece(s_begin - 1 - LA, ti, tj, rb, ct);

    auto up = [](bf16_t v) -> float { return (float)v; };
    unsigned long long tph[5] = {0, 0, 0, 0, 0};
    auto stamp = [&]() -> unsigned long long { return kW4Stamps ? __builtin_amdgcn_s_memtime() : 0ull; };
    unsigned long long ts1 = 0, ts2 = 0, ts3 = 0;

    if (s_begin < s_end) {
        bf16x8 xf[NXF];
        auto ldf = [&](int f, const int (&rbx)[3], const int (&ctx)[3][3]) -> bf16x8 {
            const int tap = f / Q, q = f % Q;
            return *reinterpret_cast<const bf16x8*>(ring + ((ctx[tap / 3][tap % 3] ^ (q << 5)) + rbx[tap / 3]));
        };
        if (MIDB) {
#pragma unroll
            for (int f = 0; f < NXF - 1; ++f) xf[f] = ldf(f, rb, ct);
        }
        for (int i = s_begin - 1 - LA; i < s_end + 2; ++i) {
            const int zr = zw == 2 ? 0 : zw + 1;
            const unsigned long long ts0 = stamp();
            if (!MIDB) {
#pragma unroll
                for (int f = 0; f < NXF - 1; ++f) xf[f] = ldf(f, rb, ct);
            }
            const f32x16 prev = acc;                   // step i-1's accumulators: staged by the FIN pieces
            bf16x8 fraw[2], fzr[2], praw[CPW], fo[2];
            f32x4 cf[CPW][4];
            u32x4 pw[CPW];
            int nrb[3], nct[3][3];
            bool fvalid[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = tr0[0] + (k ? fpx1 : fpx0) / W;
                fvalid[k] = (unsigned)(r - 1) < (unsigned)H && i - 2 >= s_begin;
            }
            w4_static_for<0, FR>([&](auto fc) __attribute__((always_inline)) {
                constexpr int f = decltype(fc)::value;
                if constexpr (f + NXF - 1 < FR) xf[(f + NXF - 1) % NXF] = ldf(f + NXF - 1, rb, ct);
                else if constexpr (MIDB) xf[(f + NXF - 1 - FR) % NXF] = ldf(f + NXF - 1 - FR, nrb, nct);        // step i+1 (after the barrier)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (f == 0) {
                    const f32x16 zero = {};
                    acc = mfma(wreg[0], xf[0], zero);
                } else {
                    acc = mfma(wreg[f], xf[f % NXF], acc);
                }
                // (an MFMA has no side effect: instruction selection may linearize it anywhere between its operands and its use,
                // across the fences; the empty asm ties it to the program order of the fences and the LDS reads)
                asm volatile("" : "+a"(acc));
                __builtin_amdgcn_sched_barrier(0);

                if constexpr (f == 0) {
                    // row group i+2 -> ring: this wave's chunks (unconditional; an unreal row reads out of range)
#pragma unroll
                    for (int u = 0; u < CPW; ++u) {
                        const int qc = wave + 4 * u;
                        const int j = qc / CPR, c0 = (qc % CPR) * CHP;
                        const int r = tr0[NT - 1] + 1 + j, v = (i + LA) * SR + 1 + j;
                        const unsigned rowoff = ((unsigned)r <= (unsigned)H && !(SED_DBG(p, 8))) ? (unsigned)(tbH[NT - 1] + r - 1) * (unsigned)(W * PIX) : SED_OOB;
                        wir_dma16(xsrd, ring + (v & (R - 1)) * ROWB + (c0 + 1) * PIX, rowoff + dma_lane[u]);
                    }
                    if (RELUBWD) {     // reference tile of step i: this thread's own two items (read by its flush two iterations on)
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const int r = tr0[2] + (k ? fpx1 : fpx0) / W;
                            const bool ok = (unsigned)(r - 1) < (unsigned)H && i >= s_begin;
                            const unsigned off = ok ? (unsigned)(tbH[2] + tr0[2] - 1) * (unsigned)(W * COUT * 2) + (unsigned)((tid + 256 * k) * 16) : SED_OOB;
                            wir_dma16(rsrd, zst + (zw * 512 + 256 * k + wave * 64) * 16, off);
                        }
                    }
                }
                // vector-memory order of an iteration: DMA x CPW, reference DMA x NZ, 2 stores.  Everything issued LA - 1 iterations
                // ago has landed: row group i+1, the reference tile of step i-2
                if constexpr (f == 1) w4_wait_vm<(LA - 1) * (CPW + NZ) + (LA - 2) * 2>();
                if constexpr (f == 1 && kW4Stamps) ts1 = stamp();
                if constexpr (f == FR / 2 && kW4Stamps) ts2 = stamp();
                if constexpr (f == FR - 1 && kW4Stamps) ts3 = stamp();

                // the step's barrier: the staging image of step i-1 and row group i+1 are complete in every wave (their LDS writes are
                // at least NXF LDS operations old: the counted wait covers them and leaves the newest fragment reads in flight)
                if constexpr (MIDB && f == GB) asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" ::"n"(NXF - 1) : "memory");

                // ---- LDS-read track ----------------------------------------------------------------------------------------
                w4_static_for<0, NL>([&](auto rc) __attribute__((always_inline)) {
                    constexpr int r = decltype(rc)::value;
                    if constexpr (GL0 + r / RPG == f) {
                        if constexpr (r < L_COEF) {                      // raw rows of group i+1 (this wave's own chunks)
                            constexpr int u = r - L_PRAW;
                            const int qc = wave + 4 * u;
                            const int j = qc / CPR, c0 = (qc % CPR) * CHP;
                            const int v = (i + 1) * SR + 1 + j;
                            praw[u] = *(reinterpret_cast<const bf16x8*>(ring + (v & (R - 1)) * ROWB + (c0 + 1) * PIX) + lane);
                        } else if constexpr (r < L_FRAW) {               // prologue scale / shift of the lane's 8 channels
                            constexpr int u = (r - L_COEF) / 4, w4 = (r - L_COEF) % 4;
                            const f32x4* pc = reinterpret_cast<const f32x4*>(coef);
                            cf[u][w4] = pc[(w4 >= 2 ? CIN / 4 : 0) + pro_ci[u] + (w4 & 1)];
                        } else if constexpr (r < L_FZR) {                // staged tile of step i-2 (written before the last barrier)
                            constexpr int k = r - L_FRAW;
                            fraw[k] = *reinterpret_cast<const bf16x8*>(ost + ((i & 1) * NPX + (k ? fpx1 : fpx0)) * OP + fcg * 8);
                        } else {                                         // its reference tile
                            constexpr int k = r - L_FZR;
                            fzr[k] = *reinterpret_cast<const bf16x8*>(zst + (zr * 512 + 256 * k + tid) * 16);
                        }
                    }
                });

                // ---- vector track ------------------------------------------------------------------------------------------
                w4_static_for<0, NVP>([&](auto pcn) __attribute__((always_inline)) {
                    constexpr int pc = decltype(pcn)::value;
                    if constexpr (w4_piece_gap(pc, P_FL, N_FL, G_SPLIT, FR) == f) {
                        if constexpr (pc < P_FIN) {                      // fragment addresses of step i+1
                            addr_piece(i + 1, pc / 3, pc % 3, nrb, nct);
                            asm volatile("" : "+v"(nct[pc / 3][pc % 3]));
                        } else if constexpr (pc < P_PRO) {               // step i-1's accumulators -> bf16 staging image
                            constexpr int g4 = pc - P_FIN;
                            T* o = ost + ((i - 1) & 1) * NPX * OP + o_lane;
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = prev[4 * g4 + e];
                            store4<T>(o + 8 * g4, v);
                        } else if constexpr (pc >= P_FL) {               // flush of step i-2
                            constexpr int k = (pc - P_FL) / (NEP + 1), q = (pc - P_FL) % (NEP + 1);
                            if constexpr (q < NEP) {
#pragma unroll
                                for (int e = q * EPP; e < (q + 1) * EPP; ++e) {
                                    if constexpr (RELUBWD) {
                                        const float zz = fvalid[k] ? up(fzr[k][e]) : 0.f;      // (dead items hold whatever the LDS held: no 0 * NaN)
                                        const float gate = (fvalid[k] && fmaf(zz, ecs[e >> 2][e & 3], ect[e >> 2][e & 3]) > 0.f) ? up(fraw[k][e]) : 0.f;
                                        fo[k][e] = (bf16_t)gate;
                                        S[e] += gate;
                                        Qs[e] = fmaf(gate, zz - ecm[e >> 2][e & 3], Qs[e]);
                                    } else if constexpr (EPI == SED_EPI_STATS) {
                                        const float fv = fvalid[k] ? up(fraw[k][e]) : 0.f;
                                        S[e] += fv;
                                        Qs[e] = fmaf(fv, fv, Qs[e]);
                                    }
                                    if constexpr (EPI != SED_EPI_STORE) asm volatile("" : "+v"(S[e]), "+v"(Qs[e]));
                                }
                            } else {                                     // exactly one store instruction per item
                                const unsigned off = (fvalid[k] && !(SED_DBG(p, 1))) ? (unsigned)(tbH[0] + tr0[0] - 1) * (unsigned)(W * COUT * 2) + (unsigned)((tid + 256 * k) * 16) : SED_OOB;
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, RELUBWD ? fo[k] : fraw[k]), zsrd, off, 0, 0);
                            }
                        } else {                                         // prologue of row group i+1
                            constexpr int u = (pc - P_PRO) / (BNPRO ? NEP + 1 : 1), q = BNPRO ? (pc - P_PRO) % (NEP + 1) : NEP;
                            if constexpr (q < NEP) {
                                bf16x8 o = {};
                                if constexpr (q > 0) o = __builtin_bit_cast(bf16x8, pw[u]);
#pragma unroll
                                for (int e = q * EPP; e < (q + 1) * EPP; ++e)
                                    o[e] = (bf16_t)fmaxf(0.f, fmaf(up(praw[u][e]), cf[u][e >> 2][e & 3], cf[u][2 + (e >> 2)][e & 3]));
                                pw[u] = __builtin_bit_cast(u32x4, o);
                                asm volatile("" : "+v"(pw[u]));
                            } else {                                     // write back relu(bn(.)) -- or zeros for an unreal row
                                const int qc = wave + 4 * u;
                                const int j = qc / CPR, c0 = (qc % CPR) * CHP;
                                const int r = tr0[3] + 1 + j, v = (i + 1) * SR + 1 + j;
                                const bool real = (unsigned)r <= (unsigned)H;
                                u32x4 ow = BNPRO ? pw[u] : __builtin_bit_cast(u32x4, praw[u]);
#pragma unroll
                                for (int e = 0; e < 4; ++e) ow[e] = real ? ow[e] : 0u;
                                *(reinterpret_cast<u32x4*>(ring + (v & (R - 1)) * ROWB + (c0 + 1) * PIX) + lane) = ow;
                            }
                        }
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
#pragma unroll
            for (int ti = 0; ti < 3; ++ti) {
                rb[ti] = nrb[ti];
#pragma unroll
                for (int tj = 0; tj < 3; ++tj) ct[ti][tj] = nct[ti][tj];
            }
            // next iteration's step table
#pragma unroll
            for (int k = 0; k + 1 < NT; ++k) { tbH[k] = tbH[k + 1]; tr0[k] = tr0[k + 1]; }
            nr0 += SR;
            if (nr0 >= HV) { nr0 = 0; nbH += H; }
            tbH[NT - 1] = nbH;
            tr0[NT - 1] = (i + LA + 1 < s_end) ? nr0 : DEAD;
            zw = zr;
            const unsigned long long ts4 = stamp();
            if (!MIDB) wir_barrier();
            if (kW4Stamps) {
                const unsigned long long ts5 = stamp();
                tph[0] += ts1 - ts0; tph[1] += ts2 - ts1; tph[2] += ts3 - ts2; tph[3] += ts4 - ts3; tph[4] += ts5 - ts4;
            }
        }
        if (kW4Stamps && (SED_DBG(p, 16)) && (blockIdx.x == 0 || blockIdx.x == 100) && (tid & 63) == 0 && (wave == 0 || wave == 3))
            printf("w4 block %d wave %d: %d iterations; cycles top..gap1 %llu  ..mid %llu  ..last mfma %llu  ..barrier %llu  barrier %llu  kernel so far %llu\n", (int)blockIdx.x,
                   wave, s_end + 2 - (s_begin - 1 - LA), tph[0], tph[1], tph[2], tph[3], tph[4], (unsigned long long)__builtin_amdgcn_s_memtime() - tk0);
    }

    // ---- per-workgroup statistics partial (fixed-order sums; unused rows of `partial` are zeroed) --------------------------
    if (EPI != SED_EPI_STORE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);     // [256][16]
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = S[e]; red[tid * 16 + 8 + e] = Qs[e]; }
        __syncthreads();
        for (int o = tid; o < 2 * COUT; o += 256) {
            const int stat = o / COUT, cn = o % COUT;
            const int cg = cn >> 3, e = cn & 7;
            float tot = 0.f;
            for (int k = 0; k < 256 / IPR; ++k) tot += red[(cg + IPR * k) * 16 + stat * 8 + e];
            if (RELUBWD && stat) tot *= p.epi_invstd[cn];
            const int nb = gridDim.x;
            p.partial[((size_t)blockIdx.x * 2 + stat) * COUT + cn] = tot;
            for (int row = blockIdx.x + nb; row < p.nparts; row += nb) p.partial[((size_t)row * 2 + stat) * COUT + cn] = 0.f;
        }
    }
}

template <int W, int CIN, int COUT, int PRO, int EPI>
int launch_w4(ConvParams& p, hipStream_t st) {
    typedef W4Geom<W, CIN, COUT> G;
    static_assert(G::LDS <= 160 * 1024, "LDS budget");
    if (int rc_ = sed_set_max_lds<&conv_w4_kernel<W, CIN, COUT, PRO, EPI>>(G::LDS)) return rc_;
    p.tilesPerImg = (p.H + 1 + G::SR - 1) / G::SR;                 // steps per image: rows 0 (zero), 1..H, zero fill
    p.totalTiles = p.B * p.tilesPerImg;                            // steps
    int nb = kW4Blocks;
    if (nb > p.nparts && p.epi != SED_EPI_STORE) nb = p.nparts;    // `partial` has nparts rows
    if (nb > p.totalTiles) nb = p.totalTiles;
    if (nb < 1) nb = 1;
    p.tpb = cdiv(p.totalTiles, nb);
    nb = cdiv(p.totalTiles, p.tpb);
    conv_w4_kernel<W, CIN, COUT, PRO, EPI><<<dim3(nb), dim3(256), G::LDS, st>>>(p);
    return 0;
}

template <int W, int CIN, int COUT>
int dispatch_w4_pe(ConvParams& p, hipStream_t st) {
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_STATS) return launch_w4<W, CIN, COUT, SED_PRO_NONE, SED_EPI_STATS>(p, st);
    if (p.pro == SED_PRO_BNRELU && p.epi == SED_EPI_STATS) return launch_w4<W, CIN, COUT, SED_PRO_BNRELU, SED_EPI_STATS>(p, st);
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_STORE) return launch_w4<W, CIN, COUT, SED_PRO_NONE, SED_EPI_STORE>(p, st);
    if (p.pro == SED_PRO_BNRELU && p.epi == SED_EPI_STORE) return launch_w4<W, CIN, COUT, SED_PRO_BNRELU, SED_EPI_STORE>(p, st);
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_RELUBWD) return launch_w4<W, CIN, COUT, SED_PRO_NONE, SED_EPI_RELUBWD>(p, st);
    return -1;
}

}  // namespace

// bf16 forward / data gradient, one wave per SIMD with register-resident weights; -1 = shape not covered (the caller falls
// back to the producer/consumer kernel)
int launch_conv_w4(ConvParams& p, int W, hipStream_t st) {
    if (p.col_only) return -1;
    const size_t xin = (size_t)p.B * p.H * W * p.Cinp * 2, xout = (size_t)p.B * p.H * W * p.Coutp * 2;
    if (xin >= 0x80000000ull || xout >= 0x80000000ull) return -1;          // 32-bit buffer offsets over the whole tensor
    if ((long long)p.B * (p.H + 64) >= (1 << 22)) return -1;               // step indices / row offsets stay far below the DEAD marker
#define SED_W4_CASE(WW, CI, CO) if (W == WW && p.Cinp == CI && p.Coutp == CO) return dispatch_w4_pe<WW, CI, CO>(p, st);
    SED_W4_CASE(16, 128, 128)
    SED_W4_CASE(8, 128, 128)
    SED_W4_CASE(16, 64, 128)
    SED_W4_CASE(16, 128, 64)
    SED_W4_CASE(32, 64, 64)
#undef SED_W4_CASE
    return -1;
}
