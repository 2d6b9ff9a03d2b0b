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

#include <stdlib.h>

namespace {

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

// 16-byte slot XOR of LDS pixel (column col_lds = image column + 1, virtual row v)
template <int W, int SLOTS>
__device__ __forceinline__ int wir_z(int col_lds, int v) {
    if (SLOTS == 16) return (col_lds + (W == 8 ? 8 * (v & 1) : 0)) & 15;
    return ((col_lds >> 1) + (W == 8 ? 4 * (v & 1) : 0)) & 7;
}

__device__ __forceinline__ void wir_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

typedef __attribute__((address_space(3))) char lds_char;

// LDS-DMA of 16 bytes per lane: LDS[lds_dst + 16*lane] <- buffer[voff] (lds_dst wave-uniform).  Inline asm on purpose:
// for the builtin hipcc models the instruction as a pending LDS write and drains the vector-memory counter (vmcnt(0), or a
// count-down over every older operation) in front of the next ds_read of the array -- the row DMAs of the NEXT step and
// the output stores are exactly what must stay in flight across the k loop.  hipcc does not count an asm load: every
// wait for these is the kernel's own counted s_waitcnt (and nothing else in the step loop may load to a register).
__device__ __forceinline__ void wir_dma16(__amdgpu_buffer_rsrc_t srd, const char* lds_dst, unsigned voff) {
    const unsigned dst = (unsigned)(size_t)(lds_char*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(dst), "s"(srd)
                 : "memory");
}

template <int W, int CIN, int COUT, int PRO, int EPI>
__global__ __launch_bounds__(512) void conv_wir_kernel(ConvParams p) {
    typedef bf16_t T;
    typedef WirGeom<W, CIN, COUT> G;
    constexpr int NCB = G::NCB, RB = G::RB, SR = G::SR, R = G::R, PIX = G::PIX, ROWB = G::ROWB, SLOTS = G::SLOTS;
    constexpr int QH = G::QH, FR = G::FR, CHP = G::CHP, CPR = G::CPR, NCH = G::NCH, CPW = G::CPW, OP = G::OP, NPX = G::NPX;
    constexpr int IPR = G::IPR;
    constexpr bool RELUBWD = EPI == SED_EPI_RELUBWD;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* part = reinterpret_cast<float*>(smem + G::RING_B);                 // [2][8 waves][2][64][4]
    T* ost = reinterpret_cast<T*>(smem + G::RING_B + G::PART_B);              // [2][NPX][OP]
    float* coef = reinterpret_cast<float*>(smem + G::RING_B + G::PART_B + G::OST_B);     // [2][CIN] pro, [3][COUT] epi
    char* zst = smem + G::RING_B + G::PART_B + G::OST_B + G::COEF_B;                      // [2][512] 16-byte items

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (scalar: everything derived from it stays in SGPRs)
    // Lane-derived indices are RE-DERIVED at the top of every pipeline iteration from an opaque copy of the thread id
    // (relane() below): left alone, hipcc hoists some 70 loop-invariant address registers out of the step loop and spills
    // weight fragments to scratch inside the k loop -- a few integer instructions per step are free, registers are not.
    int lane = tid & 63;
    const int H = p.H, H1 = H + 1;
    const int Vend = p.B * H1;                         // virtual rows 1 .. Vend-1 hold image rows; multiples of H1 are zero rows
    const float invH1 = 1.0f / (float)H1;
    const int cb = wave % NCB, kh = (wave / NCB) & 1, ph = wave / (2 * NCB);
    int n = lane & 31, hh = lane >> 5;

    // steps: step s produces virtual rows [1 + s*SR, 1 + (s+1)*SR)
    const int NS = p.totalTiles;                       // = ceil((Vend - 1) / SR)
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
#pragma unroll
        for (int f = 0; f < FR; ++f) {
            const int tap = f / QH, q = f % QH;
            const int ch16 = kh * QH + q;              // 16-channel group of the input
            const int c = ch16 >> 1, kq = 2 * (ch16 & 1) + hh;
            wreg[f] = *reinterpret_cast<const bf16x8*>(wg + ((size_t)((c * 9 + tap) * 4 + kq) * COUT + cb * 32 + (n ^ (16 * kh))) * 8);
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xsrd = make_srd(p.x, (size_t)p.B * H * W * PIX);
    const __amdgpu_buffer_rsrc_t zsrd = make_srd(p.z, (size_t)p.B * H * W * COUT * 2);
    const __amdgpu_buffer_rsrc_t rsrd = make_srd(RELUBWD ? p.zref : p.z, (size_t)p.B * H * W * COUT * 2);

    // virtual row -> (is an image row, global image-row index b*H + h)
    auto vrow = [&](int v, bool& real, int& gr) {
        const int b = (int)(((float)v + 0.5f) * invH1);
        real = v > 0 && v < Vend && (v - b * H1) != 0;
        gr = v - b - 1;
    };

    // ---- row group g = virtual rows [2 + g*SR, 2 + (g+1)*SR): DMA of this wave's chunks, then (next iteration) prologue ----
    int dpix = lane / SLOTS, dslot = lane % SLOTS;                    // the lane's pixel / destination slot inside a chunk
    auto chunk_geom = [&](int g, int u, bool& on, int& v, int& c0) {
        const int qc = wave + 8 * u;
        on = qc < NCH;
        v = 2 + g * SR + (on ? qc / CPR : 0);
        c0 = (on ? qc % CPR : 0) * CHP;
    };
    auto dma_group = [&](int g, bool live) {
#pragma unroll
        for (int u = 0; u < CPW; ++u) {
            bool on; int v, c0;
            chunk_geom(g, u, on, v, c0);
            if (!on) break;
            bool real; int gr;
            vrow(v, real, gr);
            const int col = c0 + dpix;
            const int sslot = dslot ^ wir_z<W, SLOTS>(col + 1, v);
            const unsigned voff = (live && real && !(p.dbg & 8)) ? (unsigned)((gr * W + col) * PIX + sslot * 16) : SED_OOB;
            char* dst = ring + (v & (R - 1)) * ROWB + (c0 + 1) * PIX;
            wir_dma16(xsrd, dst, voff);
        }
    };
    bf16x8 praw[CPW];
    auto pro_load = [&](int g, bool live) {        // after this wave's counted vmcnt: the rows it fetched itself
        if (!live) return;
#pragma unroll
        for (int u = 0; u < CPW; ++u) {
            bool on; int v, c0;
            chunk_geom(g, u, on, v, c0);
            if (!on) break;
            bool real; int gr;
            vrow(v, real, gr);
            bf16x8* it = reinterpret_cast<bf16x8*>(ring + (v & (R - 1)) * ROWB + (c0 + 1) * PIX) + lane;
            if (!real) {                          // zero row between two images / rows past the batch (wave-uniform)
                const bf16x8 z8 = {};
                *it = z8;
            } else if (PRO == SED_PRO_BNRELU) {
                praw[u] = *it;
            }
        }
    };
    auto pro_store = [&](int g, bool live) {
        if (!live || PRO != SED_PRO_BNRELU) return;
#pragma unroll
        for (int u = 0; u < CPW; ++u) {
            bool on; int v, c0;
            chunk_geom(g, u, on, v, c0);
            if (!on) break;
            bool real; int gr;
            vrow(v, real, gr);
            if (!real) continue;
            bf16x8* it = reinterpret_cast<bf16x8*>(ring + (v & (R - 1)) * ROWB + (c0 + 1) * PIX) + lane;
            const int sslot = dslot ^ wir_z<W, SLOTS>(c0 + dpix + 1, v);       // the 8 input channels this slot holds
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
    };

    // ---- flush item of this thread: pixel fpx of the step, channels fcg*8 .. +8 (fixed for the whole kernel) ---------------
    int ftid = tid, fpx = tid / IPR, fcg = tid % IPR;
    int frow = fpx / W, fcol = fpx % W;
    float S[8], Q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { S[e] = 0.f; Q[e] = 0.f; }
    auto item_off = [&](int s, bool live, bool& valid) -> unsigned {
        bool real; int gr;
        vrow(1 + s * SR + frow, real, gr);
        valid = live && real;
        return valid ? (unsigned)(((gr * W + fcol) * COUT + fcg * 8) * 2) : SED_OOB;
    };
    // RELUBWD: the reference tile of a step comes in by LDS-DMA as well -- a wave fetches exactly its own threads' items
    // (1 KB, lane-linear), so its own counted vmcnt orders them and no load in this kernel has a register destination
    // (hipcc would wait vmcnt(0) for one, draining the row DMAs that are meant to stay in flight)
    auto issue_zref = [&](int s, bool live) {
        if (!RELUBWD) return;
        bool valid;
        const unsigned off = item_off(s, live, valid);
        wir_dma16(rsrd, zst + ((s & 1) * 512 + wave * 64) * 16, off);
    };
    bf16x8 fraw, fzr;
    auto flush_load = [&](int s) {
        fraw = *reinterpret_cast<const bf16x8*>(ost + ((s & 1) * NPX + fpx) * OP + fcg * 8);
        if (RELUBWD) fzr = *reinterpret_cast<const bf16x8*>(zst + ((s & 1) * 512 + ftid) * 16);
    };
    auto flush_store = [&](int s, bool live) {       // (always exactly one store instruction: dead steps store out of range)
        bool valid;
        const unsigned off = item_off(s, live, valid);
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
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), zsrd, (p.dbg & 1) ? SED_OOB : off, 0, 0);
        } else {
            if (EPI == SED_EPI_STATS && valid) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)fraw[e]; S[e] += f; Q[e] = fmaf(f, f, Q[e]); }
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, fraw), zsrd, (p.dbg & 1) ? SED_OOB : off, 0, 0);
        }
    };

    // ---- the k loop of one step: FR MFMAs, one ds_read_b128 each ----------------------------------------------------------
    f32x16 acc = {};
    int prow = n / W, pcol = n % W;
    int sb = kh * (SLOTS / 2) + hh;                    // slot of this lane's first 8 channels inside a tap
    auto relane = [&]() {
        int t = tid;
        asm volatile("" : "+v"(t));
        lane = t & 63; n = lane & 31; hh = lane >> 5;
        dpix = lane / SLOTS; dslot = lane % SLOTS;
        ftid = t; fpx = t / IPR; fcg = t % IPR; frow = fpx / W; fcol = fpx % W;
        prow = n / W; pcol = n % W; sb = kh * (SLOTS / 2) + hh;
    };
    const int pwave = wave ^ NCB;                      // the other k-half of the same (cb, ph)
    auto in_range = [&](int s) { return s >= s_begin && s < s_end; };
    auto group_live = [&](int g) { return g >= s_begin - 1 && g < s_end; };      // groups s_begin-1 .. s_end-1 feed this strip
    constexpr int NZ = RELUBWD ? 1 : 0;
    constexpr int NXF = CIN == 128 ? (RELUBWD ? 4 : 6) : 8;      // fragment ring depth (what the register file leaves)
    // side work of an iteration, placed BETWEEN the MFMAs of its k loop (a wave issues one MFMA per ~64 cycles -- its SIMD
    // partner takes the other half of the matrix pipe -- so ~10 instruction slots per MFMA are free): slot = MFMA index
    constexpr int S_DMA = 0, S_FL_LD = 3, S_FL_ST = 6, S_WAIT = FR / 2, S_PRO_ST = FR / 2 + 3;
    static_assert(S_FL_ST < S_WAIT && S_PRO_ST < FR, "slot order");

    // ---- one pipeline iteration: k loop of step i with, in its shadow, DMA(row group i+2, reference tile of step i-1) |
    // finish(step i-1) | flush(step i-2) | counted wait for the previous iteration's DMAs -> prologue(group i+1).
    // Every vector-memory instruction is issued unconditionally (dead steps address out of range), so the counted wait is
    // exact and the store of the flush is never waited for.  Fill / drain iterations run the k loop on whatever the ring
    // holds (results discarded): one instruction stream, no second copy of the step body.
    unsigned long long tph[4] = {0, 0, 0, 0};
    auto stamp = [&]() -> unsigned long long { return (p.dbg & 16) ? __builtin_amdgcn_s_memtime() : 0ull; };
    auto iteration = [&](int i) {
        const unsigned long long t0 = stamp();
        // finish(step i-1) first: the partner's half (written before the last barrier) + the own half still in accumulator
        // registers 0..7 -> bf16 staging image; its LDS round trip overlaps the address set-up of this step's k loop
        const bool fin = in_range(i - 1);
        f32x4 pv[2];
        {
            const float* src = part + ((((i - 1) & 1) * 8 + pwave) * 2) * 256 + lane * 4;
            pv[0] = *reinterpret_cast<const f32x4*>(src);
            pv[1] = *reinterpret_cast<const f32x4*>(src + 256);
        }
        const int vout = 1 + i * SR + ph * RB + prow;
        int base9[3][3];
#pragma unroll
        for (int ti = 0; ti < 3; ++ti) {
            const int vin = vout + ti - 1;
            const int rb_ = (vin & (R - 1)) * ROWB;
#pragma unroll
            for (int tj = 0; tj < 3; ++tj) {
                const int cl = pcol + tj;
                base9[ti][tj] = rb_ + cl * PIX + ((sb ^ wir_z<W, SLOTS>(cl, vin)) << 4);
            }
        }
        if (fin) {
            T* o = ost + (((i - 1) & 1) * NPX + ph * 32 + n) * OP + cb * 32 + 16 * kh + 4 * hh;
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[4 * g2 + e] + pv[g2][e];
                store4<T>(o + 8 * g2, v);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        // fragment ring: NXF - 1 reads in flight ahead of the MFMA that consumes them -- with eight waves reading 128 B/clk the
        // LDS round trip is ~250 cycles, and a wave that keeps only two reads in flight issues one MFMA per ~125 cycles
        bf16x8 xf[NXF];
        auto ld = [&](int f) -> bf16x8 {
            const int tap = f / QH, q = f % QH;
            return *reinterpret_cast<const bf16x8*>(ring + (base9[tap / 3][tap % 3] ^ (q << 5)));
        };
#pragma unroll
        for (int f = 0; f < NXF - 1; ++f) xf[f] = ld(f);
        const unsigned long long t1 = stamp();
#pragma unroll
        for (int f = 0; f < FR; ++f) {
            if (f + NXF - 1 < FR) xf[(f + NXF - 1) % NXF] = ld(f + NXF - 1);
            __builtin_amdgcn_sched_barrier(0);
            acc = mfma(wreg[f], xf[f % NXF], acc);       // (no run-time switch here: a branch per MFMA breaks the straight-line k loop)
            __builtin_amdgcn_sched_barrier(0);
            if (f == S_DMA) {
                dma_group(i + 2, group_live(i + 2));
                issue_zref(i - 1, fin);                               // read by the flush of the next iteration
            }
            if (f == S_FL_LD) flush_load(i - 2);
            if (f == S_FL_ST) flush_store(i - 2, in_range(i - 2));
            if (f == S_WAIT) {
                // vector-memory order per iteration: DMA x CPW, [reference DMA], store: wait for the PREVIOUS iteration's
                if (CPW + NZ + 1 == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (CPW + NZ + 1 == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if (CPW + NZ + 1 == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                pro_load(i + 1, group_live(i + 1));
            }
            if (f == S_PRO_ST) pro_store(i + 1, group_live(i + 1));
            if (f == S_DMA || f == S_FL_LD || f == S_FL_ST || f == S_WAIT || f == S_PRO_ST)
                __builtin_amdgcn_sched_barrier(0);
        }
        const unsigned long long t2 = stamp();
        // the half the partner finishes goes to the exchange buffer; the own half stays in registers 0..7
        float* dst = part + (((i & 1) * 8 + wave) * 2) * 256 + lane * 4;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[8 + 4 * g2 + e];
            *reinterpret_cast<f32x4*>(dst + g2 * 256) = v;
        }
        if (p.dbg & 16) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long t3 = stamp();
            tph[0] += t1 - t0; tph[1] += t2 - t1; tph[2] += t3 - t2;
        }
    };

    if (s_begin < s_end) {
        for (int i = s_begin - 3; i < s_end + 2; ++i) {
            relane();
            iteration(i);
            const unsigned long long tb = stamp();
            wir_barrier();
            if (p.dbg & 16) tph[3] += stamp() - tb;
        }
        if ((p.dbg & 16) && (blockIdx.x == 0 || blockIdx.x == 100) && lane == 0 && (wave == 0 || wave == 5))
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
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    static bool attr_done[64] = {};
    if (dev >= 0 && dev < 64 && !attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wir_kernel<W, CIN, COUT, PRO, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS);
        if (e != hipSuccess) { sed_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return 3; }
        attr_done[dev] = true;
    }
    const long long vrows = (long long)p.B * (p.H + 1) - 1;        // virtual rows 1 .. B*(H+1)-1
    p.totalTiles = (int)((vrows + G::SR - 1) / G::SR);             // steps
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
    if ((long long)p.B * (p.H + 1) >= (1 << 21)) return -1;                // exact float division of the virtual row index
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
