// 3x3 convolution forward / data gradient with the WEIGHTS RESIDENT IN REGISTERS (bf16, gfx950).
//
// Same mathematics and epilogues as conv_pc_kernel (sed_conv_pc.hip) -- nn.Conv2d(3x3, s1, p1, bias=False) of ConvBlock,
// /root/reference/models/spectogram_models.py:132-140,155-156 -- for the layers with >= 64 input channels, where the
// producer/consumer kernel is bound by the LDS: its 36.8 KB weight chunk is re-staged for every (tile, 32-channel chunk)
// while the consumers read 128 B/clk of fragments from the same array (0.080 ms of LDS traffic under 0.108 ms
// of MFMA time on the 128 -> 128 layer).  Here nothing but activations ever touches the LDS:
//
//   * one 512-thread workgroup per CU; wave (cb, kh, ph) owns the 32 output channels cb, one HALF of the input channels
//     (all nine taps: 9 * CIN/32 MFMA A-fragments = 144 registers at CIN = 128) for the whole kernel, and the 32-pixel
//     block ph of a step; per 32 pixels a wave issues exactly one ds_read_b128 (the pixels' fragment) per MFMA;
//   * the two waves (cb, 0, ph) / (cb, 1, ph) exchange half an accumulator through the LDS after the k loop (each
//     finishes 16 of the 32 output channels), 4 KB per 32 pixels x 32 channels;
//   * the images are walked as ONE sequence of rows with a single shared zero row between consecutive images (the 3x3
//     zero padding below image b and above image b+1): the input lives in a ring of image rows in LDS, every row is
//     fetched from HBM exactly once (no halo re-reads) by LDS-DMA (buffer_load ... lds, 1 KB per wave-instruction, no
//     registers, no ds_write), the XOR swizzle that makes all nine taps' fragment reads conflict-free
//     (tools/wir_bank_check.py) is applied on the SOURCE address; the BatchNorm+ReLU prologue is a 16-byte
//     read-modify-write of the rows a wave fetched itself;
//   * software pipeline over steps (one barrier per step): DMA of step i+2 | prologue of step i+1 | MFMAs of step i |
//     half-accumulator exchange + bf16 staging of step i-1 | whole-line stores + statistics of step i-2.  Waves 0-3 run
//     "memory work, then MFMAs", waves 4-7 "MFMAs, then memory work": the two waves of a SIMD alternate on the matrix
//     pipe instead of meeting there (MI355X_MICROARCH.md, two waves per SIMD, item 9).
#include "conv_common.h"
#include "wir_common.h"

#include <stdlib.h>

namespace {

// in-kernel phase stamps (s_memtime around the phases of a step, printed by one workgroup when SED_DBG & 16): compiled in
// only with -DSED_STAMPS -- even an untaken run-time branch per phase costs the step loop 5-15 %
#ifdef SED_STAMPS
constexpr bool kStamps = true;
#else
constexpr bool kStamps = false;
#endif

constexpr int kWirBlocks = 256;
// Two instruction orders (waves 0-3 "memory work, then MFMAs", waves 4-7 the reverse) would let the two waves of a SIMD
// alternate on the matrix pipe, but hipcc then needs ~35 more registers (two copies of the step body share one allocation)
// and spills weight fragments into the k loop: off.


template <int W, int CIN, int COUT>
struct WirGeom {
    static constexpr int NCB = COUT / 32;             // 32-channel output blocks
    static constexpr int NPH = 8 / (2 * NCB);         // 32-pixel blocks per step
    static constexpr int RB = 32 / W;                 // image rows per 32-pixel block
    static constexpr int SR = NPH * RB;               // rows per step
    static constexpr int R = (3 * SR + 2 <= 8) ? 8 : (3 * SR + 2 <= 16) ? 16 : 32;   // ring rows (power of two)
    static constexpr int WP = W + 2;
    static constexpr int PIX = CIN * 2;               // bytes per pixel
    static constexpr int ROWB = WP * PIX;
    static constexpr int SLOTS = CIN / 8;             // 16-byte slots per pixel
    static constexpr int QH = CIN / 32;               // k16-steps per tap in a wave's half
    static constexpr int FR = 9 * QH;                 // A fragments per wave
    static constexpr int CHP = 1024 / PIX;            // pixels per DMA chunk (one wave-instruction)
    static constexpr int CPR = W / CHP;               // chunks per row
    static constexpr int NCH = SR * CPR;              // chunks per row group
    static constexpr int CPW = (NCH + 7) / 8;         // chunks per wave
    static constexpr int OP = COUT + 8;               // staging pitch (elements)
    static constexpr int NPX = NPH * 32;              // pixels per step
    static constexpr int IPR = COUT / 8;              // 16-byte items per pixel row of the output
    static constexpr size_t RING_B = (size_t)R * ROWB;
    static constexpr size_t PART_B = (size_t)2 * 8 * 2048;
    static constexpr size_t OST_B = (size_t)2 * NPX * OP * 2;
    static constexpr size_t COEF_B = (size_t)(2 * CIN + 3 * COUT) * 4;
    static constexpr size_t ZST_B = (size_t)2 * 512 * 16;       // RELUBWD: two reference tiles, one 16-byte item per thread
    static constexpr size_t LDS = RING_B + PART_B + OST_B + COEF_B + ZST_B;
    static_assert(W <= 32 && 32 % W == 0 && SR >= 2 && NPH >= 1 && CPR >= 1 && NPX * IPR == 512, "geometry");
    static_assert(3 * SR + 2 <= R, "ring depth");
    static_assert(PART_B >= 512 * 16 * 4, "statistics reduction overlays the exchange buffers");
};

template <int W, int CIN, int COUT, int PRO, int EPI>
__global__ __launch_bounds__(512) void conv_wir_kernel(ConvParams p) {
    typedef bf16_t T;
    typedef WirGeom<W, CIN, COUT> G;
    constexpr int NCB = G::NCB, RB = G::RB, SR = G::SR, R = G::R, PIX = G::PIX, ROWB = G::ROWB, SLOTS = G::SLOTS;
    constexpr int QH = G::QH, FR = G::FR, CHP = G::CHP, CPR = G::CPR, NCH = G::NCH, CPW = G::CPW, OP = G::OP, NPX = G::NPX;
    constexpr int IPR = G::IPR;
    constexpr bool RELUBWD = EPI == SED_EPI_RELUBWD;
    constexpr int DEAD = 1 << 24;                      // row offset of a step outside this workgroup's strip: no row is "real"

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* part = reinterpret_cast<float*>(smem + G::RING_B);                 // [2][8 waves][2][64][4]
    T* ost = reinterpret_cast<T*>(smem + G::RING_B + G::PART_B);              // [2][NPX][OP]
    float* coef = reinterpret_cast<float*>(smem + G::RING_B + G::PART_B + G::OST_B);     // [2][CIN] pro, [3][COUT] epi
    char* zst = smem + G::RING_B + G::PART_B + G::OST_B + G::COEF_B;                      // [2][512] 16-byte items

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (scalar: everything derived from it stays in SGPRs)
    const int H = p.H;
    const int SPI = p.tilesPerImg, HV = SPI * SR;      // an image = HV virtual rows: row 0 zero, rows 1..H the image, the rest zero
    const int cb = wave % NCB, kh = (wave / NCB) & 1, ph = wave / (2 * NCB);
    const int pwave = wave ^ NCB;                      // the other k-half of the same (cb, ph)

    // step s produces virtual rows [s*SR, (s+1)*SR) = rows r0 .. r0+SR-1 of image b (s = b*SPI + r0/SR) and reads rows
    // r0-1 .. r0+SR; row group g = rows r0+1 .. r0+SR of the same (b, r0) -- the last rows step g needs
    const int NS = p.totalTiles;                       // = B * SPI
    const int s_begin = blockIdx.x * p.tpb;
    const int s_end = min(NS, s_begin + p.tpb);

    // ---- one-time setup: zero the ring (padding columns stay zero for good), coefficients, the resident weights ----------
    {
        const bf16x8 z8 = {};
        for (int i = tid; i < (int)(G::RING_B / 16); i += 512) reinterpret_cast<bf16x8*>(ring)[i] = z8;
        if (PRO == SED_PRO_BNRELU)
            for (int i = tid; i < 2 * CIN; i += 512) coef[i] = i < CIN ? p.pro_scale[i] : p.pro_shift[i - CIN];
        if (RELUBWD)
            for (int i = tid; i < 3 * COUT; i += 512)
                coef[2 * CIN + i] = i < COUT ? p.epi_scale[i] : i < 2 * COUT ? p.epi_shift[i - COUT] : p.epi_mean[i - 2 * COUT];
    }
    bf16x8 wreg[FR];
    {
        // wpack: [Cin/32][tap][4][Coutp][8]; this wave's k-half = input channels [kh*CIN/2, (kh+1)*CIN/2).  MFMA row r of wave
        // kh holds output channel r ^ 16*kh: accumulator registers 0..7 are then ALWAYS the 16 channels this wave finishes
        // (cb*32 + 16*kh + ..) and registers 8..15 the partner's -- no register selection by a run-time k-half
        const T* __restrict__ wg = reinterpret_cast<const T*>(p.wpack);
        const int n0 = tid & 31, hh0 = (tid >> 5) & 1;
#pragma unroll
        for (int f = 0; f < FR; ++f) {
            const int tap = f / QH, q = f % QH;
            const int ch16 = kh * QH + q;              // 16-channel group of the input
            const int c = ch16 >> 1, kq = 2 * (ch16 & 1) + hh0;
            wreg[f] = *reinterpret_cast<const bf16x8*>(wg + ((size_t)((c * 9 + tap) * 4 + kq) * COUT + cb * 32 + (n0 ^ (16 * kh))) * 8);
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xsrd = make_srd(p.x, (size_t)p.B * H * W * PIX);
    const __amdgpu_buffer_rsrc_t zsrd = make_srd(p.z, (size_t)p.B * H * W * COUT * 2);
    const __amdgpu_buffer_rsrc_t rsrd = make_srd(RELUBWD ? p.zref : p.z, (size_t)p.B * H * W * COUT * 2);

    float S[8], Q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { S[e] = 0.f; Q[e] = 0.f; }
    f32x16 acc = {};

    // (b*H, r0) of the steps / row groups i+2 .. i-2 of the current iteration: scalars, shifted once per iteration.  An index
    // outside [s_begin-1, s_end) carries r0 = DEAD: none of its rows is an image row, so its loads read out of range
    // (zeros), its prologue writes zeros and its flush stores nothing -- no separate "live" predicates anywhere.
    int bH2, r02, bH1 = 0, r01 = DEAD, bH0 = 0, r00 = DEAD, bHm1 = 0, r0m1 = DEAD, bHm2 = 0, r0m2 = DEAD;
    int nbH, nr0;                                      // running (b*H, r0) of index i+3
    {
        const int g = s_begin - 1;
        const int b = g >= 0 ? g / SPI : 0;
        bH2 = b * H;
        r02 = g >= 0 ? (g - b * SPI) * SR : DEAD;
        nbH = g >= 0 ? bH2 : 0;
        nr0 = g >= 0 ? r02 : -SR;
    }
    constexpr int NZ = RELUBWD ? 1 : 0;
    constexpr int NXF = CIN == 128 ? (RELUBWD ? 4 : 6) : 8;      // fragment ring depth (what the register file leaves)
    // side work of an iteration sits BETWEEN the MFMAs of its k loop; slot = MFMA index
    constexpr int S_DMA = 0, S_FL_LD = 3, S_FL_ST = 6, S_WAIT = FR / 2, S_PRO_ST = FR / 2 + 3;
    static_assert(S_FL_ST < S_WAIT && S_PRO_ST < FR, "slot order");
    unsigned long long tph[4] = {0, 0, 0, 0};
    auto stamp = [&]() -> unsigned long long { return kStamps ? __builtin_amdgcn_s_memtime() : 0ull; };

    if (s_begin < s_end) {
        for (int i = s_begin - 3; i < s_end + 2; ++i) {
            // Lane-derived indices are RE-DERIVED every iteration from an opaque copy of the thread id: left alone, hipcc
            // hoists some 70 loop-invariant address registers out of the step loop and spills weight fragments into the k loop
            int t = tid;
            asm volatile("" : "+v"(t));
            const int lane = t & 63, n = t & 31, hh = (t >> 5) & 1;
            const int prow = n / W, pcol = n % W;
            const unsigned long long t0 = stamp();

            // ---- finish(step i-1): the partner's half (written before the last barrier) + the own half still in
            // accumulator registers 0..7 -> bf16 staging image (a dead step stages garbage nobody stores)
            f32x4 pv[2];
            {
                const float* src = part + ((((i - 1) & 1) * 8 + pwave) * 2) * 256 + lane * 4;
                pv[0] = *reinterpret_cast<const f32x4*>(src);
                pv[1] = *reinterpret_cast<const f32x4*>(src + 256);
            }
            // fragment addresses of step i: (col term ^ k offset) + ring row base, one v_xad_u32 per fragment
            const int sb = kh * (SLOTS / 2) + hh;      // slot of this lane's first 8 channels inside a tap
            int rb[3], ct[3][3];
#pragma unroll
            for (int ti = 0; ti < 3; ++ti) {
                const int vin = i * SR + ph * RB + prow + ti - 1;
                rb[ti] = (vin & (R - 1)) * ROWB;
#pragma unroll
                for (int tj = 0; tj < 3; ++tj) {
                    const int cl = pcol + tj;
                    ct[ti][tj] = cl * PIX + ((sb ^ wir_z<W, SLOTS>(cl, vin)) << 4);
                }
            }
            // fragment ring: NXF - 1 reads in flight ahead of the MFMA that consumes them (the LDS round trip is ~250 cycles
            // with eight waves reading); issued BEFORE the finish arithmetic so that their latency overlaps it
            bf16x8 xf[NXF];
            auto ld = [&](int f) -> bf16x8 {
                const int tap = f / QH, q = f % QH;
                return *reinterpret_cast<const bf16x8*>(ring + ((ct[tap / 3][tap % 3] ^ (q << 5)) + rb[tap / 3]));
            };
#pragma unroll
            for (int f = 0; f < NXF - 1; ++f) xf[f] = ld(f);
            {
                T* o = ost + (((i - 1) & 1) * NPX + ph * 32 + n) * OP + cb * 32 + 16 * kh + 4 * hh;
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[4 * g2 + e] + pv[g2][e];
                    store4<T>(o + 8 * g2, v);
                }
            }
            const unsigned long long t1 = stamp();
            // flush item of this thread: pixel fpx of a step, channels fcg*8 .. +8 = 16-byte item number t of the step's tile
            const int fpx = t / IPR, fcg = t % IPR, frow = fpx / W;
            bf16x8 fraw, fzr, praw[CPW];
#pragma unroll
            for (int f = 0; f < FR; ++f) {
                if (f + NXF - 1 < FR) xf[(f + NXF - 1) % NXF] = ld(f + NXF - 1);
                __builtin_amdgcn_sched_barrier(0);
                if (f == 0) {
                    const f32x16 zero = {};
                    acc = mfma(wreg[0], xf[0], zero);
                } else {
                    acc = mfma(wreg[f], xf[f % NXF], acc);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (f == S_DMA) {
                    // row group i+2 -> ring: this wave's chunks (unconditional; an unreal row reads out of range)
#pragma unroll
                    for (int u = 0; u < CPW; ++u) {
                        const int qc = wave + 8 * u;
                        if (qc >= NCH) break;
                        const int j = qc / CPR, c0 = (qc % CPR) * CHP;
                        const int r = r02 + 1 + j, v = (i + 2) * SR + 1 + j;
                        const unsigned rowoff = ((unsigned)r <= (unsigned)H && !(SED_DBG(p, 8))) ? (unsigned)(bH2 + r - 1) * (unsigned)(W * PIX) : SED_OOB;
                        const int col = c0 + lane / SLOTS;
                        const int sslot = (lane % SLOTS) ^ wir_z<W, SLOTS>(col + 1, 1 + j);     // (SR is even whenever the row matters)
                        wir_dma16(xsrd, ring + (v & (R - 1)) * ROWB + (c0 + 1) * PIX, rowoff + (unsigned)(col * PIX + sslot * 16));
                    }
                    if (RELUBWD) {     // reference tile of step i-1: this wave's own 64 items (read by its flush next iteration)
                        const int r = r0m1 + frow;
                        const bool ok = (unsigned)(r - 1) < (unsigned)H && i - 1 >= s_begin;
                        const unsigned off = ok ? (unsigned)(bHm1 + r0m1 - 1) * (unsigned)(W * COUT * 2) + (unsigned)(t * 16) : SED_OOB;
                        wir_dma16(rsrd, zst + (((i - 1) & 1) * 512 + wave * 64) * 16, off);
                    }
                }
                if (f == S_FL_LD) {
                    fraw = *reinterpret_cast<const bf16x8*>(ost + ((i & 1) * NPX + fpx) * OP + fcg * 8);       // step i-2
                    if (RELUBWD) fzr = *reinterpret_cast<const bf16x8*>(zst + ((i & 1) * 512 + t) * 16);
                }
                if (f == S_FL_ST) {       // flush(step i-2): always exactly one store instruction
                    const int r = r0m2 + frow;
                    const bool valid = (unsigned)(r - 1) < (unsigned)H && i - 2 >= s_begin;
                    const unsigned off = (valid && !(SED_DBG(p, 1))) ? (unsigned)(bHm2 + r0m2 - 1) * (unsigned)(W * COUT * 2) + (unsigned)(t * 16) : SED_OOB;
                    if (RELUBWD) {
                        const f32x4* ec = reinterpret_cast<const f32x4*>(coef + 2 * CIN);
                        bf16x8 o;
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            const f32x4 es = ec[fcg * 2 + hf], et = ec[COUT / 4 + fcg * 2 + hf], em = ec[COUT / 2 + fcg * 2 + hf];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float zz = valid ? (float)fzr[4 * hf + e] : 0.f;      // (dead items hold whatever the LDS held: no 0 * NaN)
                                const float gate = (valid && fmaf(zz, es[e], et[e]) > 0.f) ? (float)fraw[4 * hf + e] : 0.f;
                                o[4 * hf + e] = (bf16_t)gate;
                                S[4 * hf + e] += gate;
                                Q[4 * hf + e] = fmaf(gate, zz - em[e], Q[4 * hf + e]);
                            }
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), zsrd, off, 0, 0);
                    } else {
                        if (EPI == SED_EPI_STATS) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float fv = valid ? (float)fraw[e] : 0.f;
                                S[e] += fv;
                                Q[e] = fmaf(fv, fv, Q[e]);
                            }
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, fraw), zsrd, off, 0, 0);
                    }
                }
                if (f == S_WAIT) {
                    // vector-memory order per iteration: DMA x CPW, [reference DMA], store: everything of the PREVIOUS
                    // iteration has landed, this iteration's operations stay in flight
                    if (CPW + NZ + 1 == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else if (CPW + NZ + 1 == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    else if (CPW + NZ + 1 == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (f == S_WAIT || f == S_PRO_ST) {
                    // prologue of row group i+1 on the chunks this wave fetched itself: S_WAIT reads them (an unreal row is
                    // cleared: the zero row between two images, rows past the strip), S_PRO_ST writes relu(bn(.)) back
#pragma unroll
                    for (int u = 0; u < CPW; ++u) {
                        const int qc = wave + 8 * u;
                        if (qc >= NCH) break;
                        const int j = qc / CPR, c0 = (qc % CPR) * CHP;
                        const int r = r01 + 1 + j, v = (i + 1) * SR + 1 + j;
                        bf16x8* it = reinterpret_cast<bf16x8*>(ring + (v & (R - 1)) * ROWB + (c0 + 1) * PIX) + lane;
                        if ((unsigned)r > (unsigned)H) {
                            if (f == S_WAIT) { const bf16x8 z8 = {}; *it = z8; }
                        } else if (PRO == SED_PRO_BNRELU) {
                            if (f == S_WAIT) {
                                praw[u] = *it;
                            } else {
                                const int sslot = (lane % SLOTS) ^ wir_z<W, SLOTS>(c0 + lane / SLOTS + 1, 1 + j);
                                const f32x4* pc = reinterpret_cast<const f32x4*>(coef);
                                const f32x4 s0 = pc[sslot * 2], s1 = pc[sslot * 2 + 1], h0 = pc[CIN / 4 + sslot * 2], h1 = pc[CIN / 4 + sslot * 2 + 1];
                                bf16x8 o;
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    o[e] = (bf16_t)fmaxf(0.f, fmaf((float)praw[u][e], s0[e], h0[e]));
                                    o[4 + e] = (bf16_t)fmaxf(0.f, fmaf((float)praw[u][4 + e], s1[e], h1[e]));
                                }
                                *it = o;
                            }
                        }
                    }
                }
                if (f == S_DMA || f == S_FL_LD || f == S_FL_ST || f == S_WAIT || f == S_PRO_ST) __builtin_amdgcn_sched_barrier(0);
            }
            const unsigned long long t2 = stamp();
            // the half the partner finishes goes to the exchange buffer; the own half stays in registers 0..7
            {
                float* dst = part + (((i & 1) * 8 + wave) * 2) * 256 + lane * 4;
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[8 + 4 * g2 + e];
                    *reinterpret_cast<f32x4*>(dst + g2 * 256) = v;
                }
            }
            // next iteration's step table
            bHm2 = bHm1; r0m2 = r0m1; bHm1 = bH0; r0m1 = r00; bH0 = bH1; r00 = r01; bH1 = bH2; r01 = r02;
            nr0 += SR;
            if (nr0 >= HV) { nr0 = 0; nbH += H; }
            bH2 = nbH;
            r02 = (i + 3 < s_end) ? nr0 : DEAD;
            if (kStamps) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const unsigned long long t3 = stamp();
                tph[0] += t1 - t0; tph[1] += t2 - t1; tph[2] += t3 - t2;
            }
            const unsigned long long tb = stamp();
            wir_barrier();
            if (kStamps) tph[3] += stamp() - tb;
        }
        if (kStamps && (SED_DBG(p, 16)) && (blockIdx.x == 0 || blockIdx.x == 100) && (tid & 63) == 0 && (wave == 0 || wave == 5))
            printf("wir block %d wave %d: %d iterations; cycles pre %llu loop %llu exch %llu barrier %llu\n", (int)blockIdx.x, wave,
                   s_end + 2 - (s_begin - 3), tph[0], tph[1], tph[2], tph[3]);
    }

    // ---- per-workgroup statistics partial (fixed-order sums; unused rows of `partial` are zeroed) --------------------------
    if (EPI != SED_EPI_STORE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float* red = part;                               // [512][16]
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[tid * 16 + e] = S[e]; red[tid * 16 + 8 + e] = Q[e]; }
        __syncthreads();
        if (tid < 2 * COUT) {
            const int stat = tid / COUT, cn = tid % COUT;
            const int cg = cn >> 3, e = cn & 7;
            float tot = 0.f;
            for (int k = 0; k < NPX; ++k) tot += red[(cg + IPR * k) * 16 + stat * 8 + e];
            if (RELUBWD && stat) tot *= p.epi_invstd[cn];
            const int nb = gridDim.x;
            p.partial[((size_t)blockIdx.x * 2 + stat) * COUT + cn] = tot;
            for (int row = blockIdx.x + nb; row < p.nparts; row += nb) p.partial[((size_t)row * 2 + stat) * COUT + cn] = 0.f;
        }
    }
}

template <int W, int CIN, int COUT, int PRO, int EPI>
int launch_wir(ConvParams& p, hipStream_t st) {
    typedef WirGeom<W, CIN, COUT> G;
    static_assert(G::LDS <= 160 * 1024, "LDS budget");
    if (int rc_ = sed_set_max_lds<&conv_wir_kernel<W, CIN, COUT, PRO, EPI>>(G::LDS)) return rc_;
    p.tilesPerImg = (p.H + 1 + G::SR - 1) / G::SR;                 // steps per image: rows 0 (zero), 1..H, zero fill
    p.totalTiles = p.B * p.tilesPerImg;                            // steps
    int nb = kWirBlocks;
    if (nb > p.nparts && p.epi != SED_EPI_STORE) nb = p.nparts;    // `partial` has nparts rows
    if (nb > p.totalTiles) nb = p.totalTiles;
    if (nb < 1) nb = 1;
    p.tpb = cdiv(p.totalTiles, nb);
    nb = cdiv(p.totalTiles, p.tpb);
    conv_wir_kernel<W, CIN, COUT, PRO, EPI><<<dim3(nb), dim3(512), G::LDS, st>>>(p);
    return 0;
}

template <int W, int CIN, int COUT>
int dispatch_wir_pe(ConvParams& p, hipStream_t st) {
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_STATS) return launch_wir<W, CIN, COUT, SED_PRO_NONE, SED_EPI_STATS>(p, st);
    if (p.pro == SED_PRO_BNRELU && p.epi == SED_EPI_STATS) return launch_wir<W, CIN, COUT, SED_PRO_BNRELU, SED_EPI_STATS>(p, st);
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_STORE) return launch_wir<W, CIN, COUT, SED_PRO_NONE, SED_EPI_STORE>(p, st);
    if (p.pro == SED_PRO_BNRELU && p.epi == SED_EPI_STORE) return launch_wir<W, CIN, COUT, SED_PRO_BNRELU, SED_EPI_STORE>(p, st);
    if (p.pro == SED_PRO_NONE && p.epi == SED_EPI_RELUBWD) return launch_wir<W, CIN, COUT, SED_PRO_NONE, SED_EPI_RELUBWD>(p, st);
    return -1;
}

}  // namespace

// bf16 forward / data gradient with register-resident weights; -1 = shape not covered (the caller falls back to the
// producer/consumer kernel)
int launch_conv_wir(ConvParams& p, int W, hipStream_t st) {
    if (p.col_only) return -1;
    const size_t xin = (size_t)p.B * p.H * W * p.Cinp * 2, xout = (size_t)p.B * p.H * W * p.Coutp * 2;
    if (xin >= 0x80000000ull || xout >= 0x80000000ull) return -1;          // 32-bit buffer offsets over the whole tensor
    if ((long long)p.B * (p.H + 64) >= (1 << 22)) return -1;               // step indices / row offsets stay far below the DEAD marker
#define SED_WIR_CASE(WW, CI, CO) if (W == WW && p.Cinp == CI && p.Coutp == CO) return dispatch_wir_pe<WW, CI, CO>(p, st);
    SED_WIR_CASE(16, 128, 128)
    SED_WIR_CASE(8, 128, 128)
    SED_WIR_CASE(16, 64, 128)
    SED_WIR_CASE(16, 128, 64)
    SED_WIR_CASE(32, 64, 64)
    SED_WIR_CASE(8, 64, 128)
    SED_WIR_CASE(8, 128, 64)
    SED_WIR_CASE(16, 64, 64)
    SED_WIR_CASE(8, 64, 64)
#undef SED_WIR_CASE
    return -1;
}
