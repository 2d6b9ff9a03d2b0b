// Log-mel front-end for gfx950: framing (reflect pad) + window + real FFT + |.|^2 + mel + 10*log10
// (+ z-score), one workgroup per frame, everything between the waveform read and the 64 mel values
// stays in LDS.
//
// Replaces librosa.core.stft / np.abs()**2 / np.dot(., MEL) / librosa.power_to_db as called from
// /root/reference/dataset/spectogram/preprocess.py:21-45 and the (x-mean)/std of
// /root/reference/dataset/spectogram/spectograms_dataset.py:104-108.
//
// FFT: the nfft real samples of a frame are packed as M = nfft/2 complex values z[n] = x[2n]+i x[2n+1],
// loaded in bit-reversed order (coalesced-ish gather from HBM/L2, frames overlap so most hits are
// L2), transformed by an in-place radix-2 DIT FFT in LDS (M*8 bytes: 4 KB for nfft 1024, 128 KB for
// the reference's nfft 32768 -- one workgroup per CU), then split into the nfft/2+1 real-FFT bins.
#include "common.h"

#include <mutex>

#include <stdlib.h>

#include <math.h>

// twiddle table tw[k] = exp(-2*pi*i*k/nfft), k = 0 .. nfft/2-1, computed in double
__global__ void twiddle_kernel(float2* __restrict__ tw, int nfft) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nfft / 2) return;
    double s, c;
    sincospi(-2.0 * (double)k / (double)nfft, &s, &c);
    tw[k] = make_float2((float)c, (float)s);
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

struct FrontParams {
    const float* wave;
    const float* window;
    const float2* tw;
    const float* melT;
    const int* mel_lo;
    const int* mel_hi;
    const float* mean;
    const float* stdv;
    float* out;      // [B][T][n_mels]   (log-mel mode)
    float2* spec;    // [B][T][bins]     (stft mode)
    int B, samples, nfft, hop, T, n_mels, logM;
    int fast_ok;     // the compact mel-weight list fits the nfft=1024 fast path
};

template <bool LOGMEL>
__global__ __launch_bounds__(256) void frontend_kernel(FrontParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* Z = reinterpret_cast<float2*>(smem);
    const int tid = threadIdx.x;
    const int nfft = p.nfft, M = nfft >> 1, logM = p.logM, L = p.samples;
    const int frame = blockIdx.x % p.T;
    const int b = blockIdx.x / p.T;
    const float* __restrict__ wv = p.wave + (size_t)b * L;
    const int start = frame * p.hop - M;   // index into the un-padded signal of padded[frame*hop]

    // ---- framing + window, packed two reals per complex, bit-reversed placement ---------------
    for (int n = tid; n < M; n += 256) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int s = start + 2 * n + e;
            if (s < 0) s = -s;                       // np.pad(mode='reflect'): edge not repeated
            if (s >= L) s = 2 * (L - 1) - s;
            v[e] = wv[s] * p.window[2 * n + e];
        }
        const int j = (int)(__brev((unsigned)n) >> (32 - logM));
        Z[j] = make_float2(v[0], v[1]);
    }
    __syncthreads();

    // ---- in-place radix-2 DIT over LDS --------------------------------------------------------
    for (int s = 0; s < logM; ++s) {
        const int half = 1 << s;
        const int tstep = M >> (s + 1);              // twiddle index stride in units of 1/M turns
        for (int j = tid; j < (M >> 1); j += 256) {
            const int pos = j & (half - 1);
            const int i0 = ((j >> s) << (s + 1)) + pos;
            const int i1 = i0 + half;
            const float2 w = p.tw[2 * pos * tstep];  // exp(-2 pi i pos/(2 half)) = tw[(nfft/(2 half)) pos]
            const float2 a = Z[i0];
            const float2 bb = cmul(Z[i1], w);
            Z[i0] = make_float2(a.x + bb.x, a.y + bb.y);
            Z[i1] = make_float2(a.x - bb.x, a.y - bb.y);
        }
        __syncthreads();
    }

    // ---- split into real-FFT bins k and M-k; power goes back in place (P[k] -> Z[k].x,
    //      P[M] -> Z[0].y) -----------------------------------------------------------------------
    float2* __restrict__ sp = LOGMEL ? nullptr : p.spec + ((size_t)b * p.T + frame) * (M + 1);
    for (int k = tid; k <= (M >> 1); k += 256) {
        if (k == 0) {
            const float2 z0 = Z[0];
            const float x0 = z0.x + z0.y, xm = z0.x - z0.y;
            if (LOGMEL) {
                Z[0] = make_float2(x0 * x0, xm * xm);
            } else {
                sp[0] = make_float2(x0, 0.f);
                sp[M] = make_float2(xm, 0.f);
            }
            continue;
        }
        const int k2 = M - k;
        const float2 a = Z[k], c = Z[k2];
        // E = (Z[k] + conj(Z[M-k]))/2 ; O = -i/2 (Z[k] - conj(Z[M-k]))
        const float2 E = make_float2(0.5f * (a.x + c.x), 0.5f * (a.y - c.y));
        const float2 O = make_float2(0.5f * (a.y + c.y), -0.5f * (a.x - c.x));
        const float2 w = p.tw[k];
        const float2 wo = cmul(w, O);
        const float2 Xk = make_float2(E.x + wo.x, E.y + wo.y);
        // X[M-k] = conj(E) - conj(w*O) ... derived from the same pair: E' = conj(E), O' = conj(O),
        // w' = exp(-2 pi i (M-k)/N) = -conj(w)  ->  X[M-k] = conj(E) - conj(w) conj(O) = conj(E - w O)
        const float2 Xk2 = make_float2(E.x - wo.x, -(E.y - wo.y));
        if (LOGMEL) {
            const float pk = Xk.x * Xk.x + Xk.y * Xk.y;
            const float pk2 = Xk2.x * Xk2.x + Xk2.y * Xk2.y;
            Z[k].x = pk;
            if (k2 != k) Z[k2].x = pk2;
        } else {
            sp[k] = Xk;
            if (k2 != k) sp[k2] = Xk2;
        }
    }
    if (!LOGMEL) return;
    __syncthreads();

    // ---- mel filterbank (sparse triangles), 4 lanes per mel bin, log, z-score -------------------
    const int quad = tid & 3;
    for (int m = tid >> 2; m < p.n_mels; m += 64) {
        const int lo = p.mel_lo[m], hi = p.mel_hi[m];
        const float* __restrict__ row = p.melT + (size_t)m * (M + 1);
        float acc = 0.f;
        for (int k = lo + quad; k < hi; k += 4) {
            const float pw = (k < M) ? Z[k].x : Z[0].y;
            acc = fmaf(row[k], pw, acc);
        }
        acc += dpp_mov<0xB1>(acc);
        acc += dpp_mov<0x4E>(acc);
        if (quad == 0) {
            float v = 10.0f * log10f(fmaxf(1e-10f, acc));
            if (p.mean) v = (v - p.mean[m]) / p.stdv[m];
            p.out[((size_t)b * p.T + frame) * p.n_mels + m] = v;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Large transforms (nfft >= 4096; the reference's committed constants are 48 kHz / nfft 32768 / hop 15840,
// /root/reference/dataset/spectogram/spectogram_configs.py:5-14): one 1024-thread workgroup per frame, the packed nfft/2-point complex
// FFT in place in LDS like frontend_kernel above, but THREE radix-2 stages fused per pass: a thread takes the 8 values
// Z[i0 + j 2^s], j = 0..7, runs stages s, s+1, s+2 on them in registers and writes them back -- 5 LDS round trips and barriers
// instead of 14 for nfft 32768 -- and the mel projection with a wave per filter (the top filters of a 16385-bin spectrum are
// ~1700 bins wide).  Round 6: the radix-2 kernel took 4.3 ms for the 5824 frames of 32 x 60 s clips at the reference's constants
// (190 us per frame and CU), 88 % of that shape's train step (profiles/r06_a_bench_ref_native_T182.json).
// -------------------------------------------------------------------------------------------------
// LDS slot of complex element i.  (Round 6 A/B: one pad slot per 8 and one per 256 elements -- which removes the 8- to 64-way bank
// conflicts of the first two passes and of the bit-reversed placement on paper -- together with 32-byte global segments per lane group
// made the kernel SLOWER, 1.09 -> 1.22 ms for 5824 frames: with one 1024-thread workgroup per CU the frame is bound by its exposed
// global / L2 latencies (samples, twiddles, filter rows), not by LDS cycles.  Identity kept.)
__device__ __forceinline__ int fe_zi(int i) { return i; }

template <int NS>
__device__ __forceinline__ void fe_fused_stages(float2* __restrict__ Z, const float2* __restrict__ tw, int s, int M, int tid, int nthr) {
    constexpr int R = 1 << NS;
    const int groups = M >> NS;
    for (int g = tid; g < groups; g += nthr) {
        const int pos0 = g & ((1 << s) - 1);
        const int i0 = ((g >> s) << (s + NS)) + pos0;
        float2 v[R];
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = Z[fe_zi(i0 + (j << s))];
        // stage s + q pairs elements j and j + 2^q (distance 2^(s+q)); the position inside that stage's half-block is
        // pos0 + (j & (2^q - 1)) 2^s and its twiddle exp(-2 pi i pos / 2^(s+q+1)) = tw[pos (M >> (s + q))]
        float2 w[R - 1];
#pragma unroll
        for (int q = 0; q < NS; ++q)
#pragma unroll
            for (int lb = 0; lb < (1 << q); ++lb) w[(1 << q) - 1 + lb] = tw[(pos0 + (lb << s)) * (M >> (s + q))];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (j & (1 << q)) continue;
                const float2 a = v[j];
                const float2 b = cmul(v[j + (1 << q)], w[(1 << q) - 1 + (j & ((1 << q) - 1))]);
                v[j] = make_float2(a.x + b.x, a.y + b.y);
                v[j + (1 << q)] = make_float2(a.x - b.x, a.y - b.y);
            }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) Z[fe_zi(i0 + (j << s))] = v[j];
    }
}

template <bool LOGMEL>
__global__ __launch_bounds__(1024) void frontend_big_kernel(FrontParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* Z = reinterpret_cast<float2*>(smem);
    const int tid = threadIdx.x;
    const int nfft = p.nfft, M = nfft >> 1, logM = p.logM, L = p.samples;
    const int frame = blockIdx.x % p.T;
    const int b = blockIdx.x / p.T;
    const float* __restrict__ wv = p.wave + (size_t)b * L;
    const int start = frame * p.hop - M;   // index into the un-padded signal of padded[frame*hop]

    // ---- framing + window, two reals per complex value, bit-reversed placement (coalesced 8-byte loads) ----------------------
    const bool interior = start >= 0 && start + nfft <= L && (reinterpret_cast<uintptr_t>(wv + (start > 0 ? start : 0)) & 7) == 0;
    {
        for (int n = tid; n < M; n += 1024) {
            float v0, v1;
            if (interior) {
                const float2 x = *reinterpret_cast<const float2*>(wv + start + 2 * n);
                v0 = x.x; v1 = x.y;
            } else {
                int s0 = start + 2 * n, s1 = s0 + 1;
                if (s0 < 0) s0 = -s0;                    // np.pad(mode='reflect'): edge not repeated
                if (s0 >= L) s0 = 2 * (L - 1) - s0;
                if (s1 < 0) s1 = -s1;
                if (s1 >= L) s1 = 2 * (L - 1) - s1;
                v0 = wv[s0]; v1 = wv[s1];
            }
            const float2 wn = *reinterpret_cast<const float2*>(p.window + 2 * n);
            const int j = (int)(__brev((unsigned)n) >> (32 - logM));
            Z[fe_zi(j)] = make_float2(v0 * wn.x, v1 * wn.y);
        }
    }
    __syncthreads();

    // ---- in-place radix-2 DIT, three stages per LDS round trip -----------------------------------------------------------------
    int s = 0;
    for (; s + 3 <= logM; s += 3) {
        fe_fused_stages<3>(Z, p.tw, s, M, tid, 1024);
        __syncthreads();
    }
    if (logM - s == 2) {
        fe_fused_stages<2>(Z, p.tw, s, M, tid, 1024);
        __syncthreads();
    } else if (logM - s == 1) {
        fe_fused_stages<1>(Z, p.tw, s, M, tid, 1024);
        __syncthreads();
    }

    // ---- split into real-FFT bins k and M-k; power goes back in place (P[k] -> Z[k].x, P[M] -> Z[0].y) --------------------------
    float2* __restrict__ sp = LOGMEL ? nullptr : p.spec + ((size_t)b * p.T + frame) * (M + 1);
    for (int k = tid; k <= (M >> 1); k += 1024) {
        if (k == 0) {
            const float2 z0 = Z[0];
            const float x0 = z0.x + z0.y, xm = z0.x - z0.y;
            if (LOGMEL) {
                Z[0] = make_float2(x0 * x0, xm * xm);
            } else {
                sp[0] = make_float2(x0, 0.f);
                sp[M] = make_float2(xm, 0.f);
            }
            continue;
        }
        const int k2 = M - k;
        const float2 a = Z[fe_zi(k)], c = Z[fe_zi(k2)];
        const float2 E = make_float2(0.5f * (a.x + c.x), 0.5f * (a.y - c.y));
        const float2 O = make_float2(0.5f * (a.y + c.y), -0.5f * (a.x - c.x));
        const float2 wo = cmul(p.tw[k], O);
        const float2 Xk = make_float2(E.x + wo.x, E.y + wo.y);
        const float2 Xk2 = make_float2(E.x - wo.x, -(E.y - wo.y));
        if (LOGMEL) {
            Z[fe_zi(k)].x = Xk.x * Xk.x + Xk.y * Xk.y;
            if (k2 != k) Z[fe_zi(k2)].x = Xk2.x * Xk2.x + Xk2.y * Xk2.y;
        } else {
            sp[k] = Xk;
            if (k2 != k) sp[k2] = Xk2;
        }
    }
    if (!LOGMEL) return;
    __syncthreads();

    // ---- mel filterbank (sparse triangles): a WAVE per filter (coalesced 256-byte reads of the filter row; wave w takes filters w, w + 16,
    //      ...: every wave gets narrow and wide ones), four loads in flight per lane, fixed-order wave sum, log, z-score ----------------
    const int lane = tid & 63;
    for (int m = tid >> 6; m < p.n_mels; m += 16) {
        const int lo = p.mel_lo[m], hi = p.mel_hi[m];
        const float* __restrict__ row = p.melT + (size_t)m * (M + 1);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int k = lo + lane;
        for (; k + 192 < hi; k += 256) {
            const float w0 = row[k], w1 = row[k + 64], w2 = row[k + 128], w3 = row[k + 192];
            a0 = fmaf(w0, Z[fe_zi(k)].x, a0);
            a1 = fmaf(w1, Z[fe_zi(k + 64)].x, a1);
            a2 = fmaf(w2, Z[fe_zi(k + 128)].x, a2);
            a3 = fmaf(w3, (k + 192 < M) ? Z[fe_zi(k + 192)].x : Z[0].y, a3);
        }
        for (; k < hi; k += 64) a0 = fmaf(row[k], (k < M) ? Z[fe_zi(k)].x : Z[0].y, a0);
        const float acc = wave_sum((a0 + a1) + (a2 + a3));
        if (lane == 0) {
            float v = 10.0f * log10f(fmaxf(1e-10f, acc));
            if (p.mean) v = (v - p.mean[m]) / p.stdv[m];
            p.out[((size_t)b * p.T + frame) * p.n_mels + m] = v;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// nfft = 1024 fast path (the 32 kHz / hop 320 bench configuration): ONE WAVE PER FRAME.
// The 512-point complex FFT (real-input trick) is three radix-8 passes held in registers
// (8 complex values per lane) with two exchanges through LDS, instead of nine barrier-separated
// radix-2 passes:  n = 64 n1 + n2, k = k1 + 8 (j1 + 8 j2)
//   pass 1  lane n2       : DFT8 over n1, * w512^(n2 k1)
//   pass 2  lane (k1, m2) : DFT8 over m1 (n2 = 8 m1 + m2), * w64^(m2 j1)
//   pass 3  lane (k1, j1) : DFT8 over m2  ->  X[k1 + 8 j1 + 64 j2]
// then the real-FFT split, |.|^2, one lane per mel filter, log, z-score.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// in-place forward DFT of 8 points, decimation in frequency; v[p] ends up holding X[bitrev3(p)]
__device__ __forceinline__ void dft8_dif(float2 (&v)[8]) {
    const float r = 0.70710678118654752440f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float2 a = v[j], b = v[j + 4];
        v[j] = cadd(a, b);
        const float2 t = csub(a, b);
        // t * exp(-i pi j / 4)
        if (j == 0) v[j + 4] = t;
        else if (j == 1) v[j + 4] = make_float2((t.x + t.y) * r, (t.y - t.x) * r);
        else if (j == 2) v[j + 4] = make_float2(t.y, -t.x);
        else v[j + 4] = make_float2((t.y - t.x) * r, -(t.x + t.y) * r);
    }
#pragma unroll
    for (int base = 0; base < 8; base += 4) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float2 a = v[base + j], b = v[base + j + 2];
            v[base + j] = cadd(a, b);
            const float2 t = csub(a, b);
            v[base + j + 2] = (j == 0) ? t : make_float2(t.y, -t.x);   // * exp(-i pi j / 2)
        }
    }
#pragma unroll
    for (int base = 0; base < 8; base += 2) {
        const float2 a = v[base], b = v[base + 1];
        v[base] = cadd(a, b);
        v[base + 1] = csub(a, b);
    }
}

// Every LDS buffer touched inside the frame loop of frontend1024_kernel is private to one wave, and a wave's LDS
// operations execute in order: the exchange steps need no workgroup barrier, only a compiler fence (a barrier made
// the four independent waves of a workgroup wait for each other seven times per frame).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- packed fp32 complex arithmetic (round 5; MEASURED NEUTRAL, off) -------------------------------------------------------------
// A complex value is a 64-bit register pair, and gfx950's VOP3P fp32 instructions work on both halves with per-half source
// selection / negation (op_sel, op_sel_hi, neg_lo, neg_hi): a complex add is ONE v_pk_add_f32, a + (-i) b is one v_pk_add_f32 with
// swapped halves of b, a complex multiply is v_pk_mul_f32 + v_pk_fma_f32 (instead of 2 / 2 / 4 scalar instructions): the batched
// kernel's vector instructions drop from 1046 to 943 (static), ~100 of ~450 per frame.  Interleaved builds on one box
// (profiles/r05_e_ab_frontend_packed_fft.txt): 0.2514-0.2555 ms packed against 0.2485-0.2587 ms scalar, 0.2478-0.2530 ms packed + the
// in-register first exchange -- nothing moves.  With the phase stamps (profiles/r05_e_frontend_phase_stamps.txt: a wave's batch is
// FFT 3580 + split 1240 + mel 1900 + finalize 1130 + staging 590 + barriers 320 = ~8760 ticks for ~580 instructions) the reading is:
// neither the vector port (-17 % instructions: nothing) nor the LDS (one of the two exchanges removed: nothing) bounds the kernel;
// a wave runs its ~580-instruction dependent chain at ~15 ticks per instruction and four waves per SIMD (128 registers, 71 KB of LDS per
// workgroup) are all there is to overlap.  -DSED_FE_PK=1 builds the packed form (A/B builds, tools/ab_build.sh).
#ifndef SED_FE_PK
#define SED_FE_PK 0
#endif
typedef float fe_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fe_f2 fe_pk(float2 a) { return (fe_f2){a.x, a.y}; }
__device__ __forceinline__ float2 fe_unpk(fe_f2 a) { return make_float2(a[0], a[1]); }
__device__ __forceinline__ fe_f2 pk_add(fe_f2 a, fe_f2 b) { fe_f2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ fe_f2 pk_sub(fe_f2 a, fe_f2 b) { fe_f2 d; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
// a + (-i) b = (a.x + b.y, a.y - b.x);  a - (-i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ fe_f2 pk_add_rot(fe_f2 a, fe_f2 b) { fe_f2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ fe_f2 pk_sub_rot(fe_f2 a, fe_f2 b) { fe_f2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ fe_f2 pk_mul(fe_f2 a, fe_f2 b) { fe_f2 d; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
// complex product: t = (a.x w.x, a.x w.y); d = (a.y * (-w.y) + t.x, a.y * w.x + t.y)
__device__ __forceinline__ fe_f2 pk_cmul(fe_f2 a, fe_f2 w) {
    fe_f2 t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(d) : "v"(a), "v"(w), "v"(t));
    return d;
}
// dft8_dif on register pairs: the same butterflies, every multiplication by -i folded into the swapped-half add / subtract that
// consumes it (28 packed instructions instead of 56 scalar ones); v[p] ends up holding X[bitrev3(p)] as dft8_dif
__device__ __forceinline__ void dft8_dif_pk(fe_f2 (&v)[8]) {
    const fe_f2 rp = {0.70710678118654752440f, 0.70710678118654752440f}, rn = {-0.70710678118654752440f, -0.70710678118654752440f};
    fe_f2 t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const fe_f2 a = v[j], b = v[j + 4];
        v[j] = pk_add(a, b);
        t[j] = pk_sub(a, b);
    }
    const fe_f2 u5 = pk_mul(pk_add_rot(t[1], t[1]), rp);        // t1 * exp(-i pi / 4)  = r (t1 + (-i) t1)
    const fe_f2 u7 = pk_mul(pk_sub_rot(t[3], t[3]), rn);        // t3 * exp(-3 i pi / 4) = -r (t3 - (-i) t3)
    // second stage (t[2] carries a pending multiplication by -i)
    const fe_f2 a0 = pk_add(v[0], v[2]), b0 = pk_sub(v[0], v[2]);
    const fe_f2 a1 = pk_add(v[1], v[3]), d1 = pk_sub(v[1], v[3]);          // d1: pending -i
    const fe_f2 a4 = pk_add_rot(t[0], t[2]), b4 = pk_sub_rot(t[0], t[2]);
    const fe_f2 a5 = pk_add(u5, u7), d5 = pk_sub(u5, u7);                  // d5: pending -i
    // third stage
    v[0] = pk_add(a0, a1);
    v[1] = pk_sub(a0, a1);
    v[2] = pk_add_rot(b0, d1);
    v[3] = pk_sub_rot(b0, d1);
    v[4] = pk_add(a4, a5);
    v[5] = pk_sub(a4, a5);
    v[6] = pk_add_rot(b4, d5);
    v[7] = pk_sub_rot(b4, d5);
}

#define FE_MAXNZ 1152  // capacity of the compact mel-weight list (64 Slaney filters on 513 bins need <= 2*513 + 64 = 1090)
#define FE_XSTRIDE 72   // float2 stride between the 8 rows of an exchange buffer (bank-conflict padding)
// Second exchange: element (k1, j1, m2) at k1*72 + j1*9 + m2 -- written by lane (k1, m2), read by lane (k1, j1); with a pitch of 8 the
// 32 lanes of a ds_read_b64 group (k1 = 0..3, j1 = 0..7) sat on FOUR bank positions (16 (k1 + j1) mod 64: 8-way conflict, round 4:
// SQ_LDS_BANK_CONFLICT = a third of the kernel's LDS-array cycles); with 9, 8 k1 + 9 j1 is distinct mod 32.
#ifndef SED_FE_SWZ
#define SED_FE_SWZ 1      // 0: the round-3 layouts (A/B builds, tools/ab_build.sh)
#endif
#define FE_JSTRIDE (SED_FE_SWZ ? 9 : 8)
#define FE_FSTRIDE 64      // slots between the j2 blocks of the final spectrum
// Final spectrum of the batched kernel: X[k], k = k1 + 8 j1 + 64 j2, sits at slot (j1 ^ 4 (k1 >> 2)) + 8 k1 + 64 j2.  Pass 3's lane
// (k1, j1) = 8 k1 + j1 then stores 16 consecutive lanes into 16 distinct 8-byte bank slots (natural order k1 + 8 j1: four slots, 4-way),
// and the split's reads by k = lane + 64 i (32 lanes: j1 = 0..3, all k1) find k1 and k1 + 4 in different halves of their shared
// 8-slot window -- conflict-free as well; both stay one base register + an immediate (tools/lds_conflicts.py prints the cycle counts).
__device__ __forceinline__ int fe_phi_lo(int k6) { return SED_FE_SWZ ? (((k6 >> 3) ^ ((k6 & 4))) + 8 * (k6 & 7)) : k6; }      // k6 = k mod 64

// Round 5: the FIRST inter-pass exchange of the 3 x radix-8 FFT without the LDS.  After pass 1 lane (a, m2) (a = lane >> 3) holds the
// eight values k1 = 0..7 of its column; pass 2 wants lane (a, m2) to hold m1 = 0..7 of column (k1 = a): value (register k1, lane (m1, m2))
// -> (register m1, lane (k1, m2)) -- a transpose between the register index and lane bits 5..3.  Three commuting swap stages:
// register bit 2 <-> lane bit 5 (v_permlane32_swap: the upper half of x[k] against the lower half of x[k + 4]), register bit 1 <-> lane
// bit 4 (v_permlane16_swap: odd 16-lane rows of x[k] against even rows of x[k + 2]), register bit 0 <-> lane bit 3 (two DPP row_ror:8
// moves with complementary bank masks).  ~36 vector instructions per frame instead of 8 ds_write_b64 + 8 ds_read_b64 and two exposed
// LDS round trips.  MEASURED NEUTRAL (profiles/r05_d_ab_frontend_exchange_in_registers.txt: 0.2526 vs 0.2520 ms over three interleaved
// builds each, outputs identical): the frame loop does not wait for this exchange.  Off; -DSED_FE_X1REG=1 builds it (A/B, tools/ab_build.sh).
#ifndef SED_FE_X1REG
#define SED_FE_X1REG 0
#endif
__device__ __forceinline__ void fe_swap_hi_lo_32(float& a, float& b) {       // a's lanes 32-63 <-> b's lanes 0-31
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];      // (hipcc: __builtin_bit_cast on a vector ELEMENT lvalue reads element 0 -- copy to scalars first)
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void fe_swap_rows_16(float& a, float& b) {        // a's rows 1, 3 (of 16 lanes) <-> b's rows 0, 2
    const auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    const unsigned r0 = r[0], r1 = r[1];      // (hipcc: __builtin_bit_cast on a vector ELEMENT lvalue reads element 0 -- copy to scalars first)
    a = __builtin_bit_cast(float, r0);
    b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void fe_swap_half_rows_8(float& a, float& b) {    // a's lanes 8-15 of every row <-> b's lanes 0-7
    const int ai = __builtin_bit_cast(int, a), bi = __builtin_bit_cast(int, b);
    const int na = __builtin_amdgcn_update_dpp(ai, bi, 0x128, 0xF, 0xC, false);      // row_ror:8 into banks 2, 3
    const int nb = __builtin_amdgcn_update_dpp(bi, ai, 0x128, 0xF, 0x3, false);      // row_ror:8 into banks 0, 1
    a = __builtin_bit_cast(float, na);
    b = __builtin_bit_cast(float, nb);
}
__device__ __forceinline__ void fe_transpose_reg_lane_hi(float2 (&x)[8]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { fe_swap_hi_lo_32(x[k].x, x[k + 4].x); fe_swap_hi_lo_32(x[k].y, x[k + 4].y); }
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (!(k & 2)) { fe_swap_rows_16(x[k].x, x[k + 2].x); fe_swap_rows_16(x[k].y, x[k + 2].y); }
#pragma unroll
    for (int k = 0; k < 8; k += 2) { fe_swap_half_rows_8(x[k].x, x[k + 1].x); fe_swap_half_rows_8(x[k].y, x[k + 1].y); }
}

__global__ __launch_bounds__(256, 5) void frontend1024_kernel(FrontParams p, int nframes_total) {
    // LDS diet: 31.3 KB per workgroup = 5 workgroups (20 waves) per CU instead of 3 -- the kernel is latency-bound (PMC: waves
    // parked 48 % of their cycles, VALU active 19 %): the power spectrum overlays the wave's exchange buffer once every X[k]
    // it can overwrite has been read, and the compact mel-weight list is sized to what 64 Slaney filters need.
    __shared__ float2 tw[512];                      // exp(-2 pi i k / 1024)
    __shared__ float win[1024];
    __shared__ float2 xbuf[4][8 * FE_XSTRIDE];      // per wave: exchange buffer, later X[512], later the power spectrum 0..512
    __shared__ float mw[FE_MAXNZ];                  // non-zero mel weights, filter after filter
    __shared__ int moff[65];                        // start of filter m in mw (n_mels <= 64 on this path)
    const int tid = threadIdx.x, lane = tid & 63, wv_id = tid >> 6;
    for (int i = tid; i < 512; i += 256) tw[i] = p.tw[i];
    for (int i = tid; i < 1024; i += 256) win[i] = p.window[i];
    if (tid == 0) {
        int o = 0;
        for (int m = 0; m < p.n_mels; ++m) { moff[m] = o; o += p.mel_hi[m] - p.mel_lo[m]; }
        moff[p.n_mels] = o;
    }
    __syncthreads();
    const bool mw_ok = moff[p.n_mels] <= FE_MAXNZ;   // an unusually dense filter matrix stays in global memory
    if (mw_ok) {
        for (int m = wv_id; m < p.n_mels; m += 4) {
            const int lo = p.mel_lo[m], cnt = p.mel_hi[m] - lo;
            for (int j = lane; j < cnt; j += 64) mw[moff[m] + j] = p.melT[(size_t)m * 513 + lo + j];
        }
    }
    __syncthreads();
    const bool vec_ok = ((p.samples | p.hop) & 1) == 0 && ((reinterpret_cast<uintptr_t>(p.wave) & 7) == 0);
    auto twid = [&](int k) -> float2 {   // exp(-2 pi i k / 1024), k in [0, 1024)
        const float2 t = tw[k & 511];
        return (k & 512) ? make_float2(-t.x, -t.y) : t;
    };
    constexpr int BR[8] = {0, 4, 2, 6, 1, 5, 3, 7};   // X[k] sits at position BR[k] after dft8_dif
    float2* xb = xbuf[wv_id];
    float* P = reinterpret_cast<float*>(xb);        // (520 floats of the 1152 the buffer holds)
    const int L = p.samples;

    // raw (un-windowed) samples of one frame for this lane: complex n = 64 n1 + lane
    auto load_frame = [&](int fidx_, float2 (&raw)[8]) {
        const int fi_ = fidx_ < nframes_total ? fidx_ : nframes_total - 1;   // dead waves redo the last frame (no stores)
        const int b_ = fi_ / p.T, frame_ = fi_ - b_ * p.T;
        const float* __restrict__ wav = p.wave + (size_t)b_ * L;
        const int start = frame_ * p.hop - 512;
        if (vec_ok && start >= 0 && start + 1024 <= L) {     // interior frame: 8-byte loads, no reflection
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) raw[n1] = *reinterpret_cast<const float2*>(wav + start + 2 * (64 * n1 + lane));
        } else {
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                int s0 = start + 2 * (64 * n1 + lane), s1 = s0 + 1;
                if (s0 < 0) s0 = -s0;
                if (s1 < 0) s1 = -s1;
                if (s0 >= L) s0 = 2 * (L - 1) - s0;
                if (s1 >= L) s1 = 2 * (L - 1) - s1;
                raw[n1] = make_float2(wav[s0], wav[s1]);
            }
        }
    };

    float2 nxt[8];
    if (blockIdx.x * 4 < nframes_total) load_frame(blockIdx.x * 4 + wv_id, nxt);
    for (int f0 = blockIdx.x * 4; f0 < nframes_total; f0 += gridDim.x * 4) {
        const int fidx = f0 + wv_id;
        const bool live = fidx < nframes_total;

        // ---- pass 1: lane = n2; window the prefetched samples, then prefetch the next batch's ----------
        float2 v[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {
            const float2 wv2 = *reinterpret_cast<const float2*>(win + 2 * (64 * n1 + lane));
            v[n1] = make_float2(nxt[n1].x * wv2.x, nxt[n1].y * wv2.y);
        }
        if (f0 + gridDim.x * 4 < nframes_total) load_frame(fidx + gridDim.x * 4, nxt);
        dft8_dif(v);
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) {
            const float2 val = (k1 == 0) ? v[BR[0]] : cmul(v[BR[k1]], twid(2 * lane * k1));   // w512^(n2 k1)
            xb[k1 * FE_XSTRIDE + lane] = val;
        }
        wave_sync();
        // ---- pass 2: lane = (k1, m2) -----------------------------------------------------------------
        const int k1 = lane >> 3, m2 = lane & 7;
#pragma unroll
        for (int m1 = 0; m1 < 8; ++m1) v[m1] = xb[k1 * FE_XSTRIDE + 8 * m1 + m2];
        wave_sync();
        dft8_dif(v);
#pragma unroll
        for (int j1 = 0; j1 < 8; ++j1) {
            const float2 val = (j1 == 0) ? v[BR[0]] : cmul(v[BR[j1]], twid(16 * m2 * j1));        // w64^(m2 j1)
            xb[k1 * FE_XSTRIDE + j1 * FE_JSTRIDE + m2] = val;
        }
        wave_sync();
        // ---- pass 3: lane = (k1, j1) -----------------------------------------------------------------
        const int j1 = lane & 7;
#pragma unroll
        for (int mm = 0; mm < 8; ++mm) v[mm] = xb[k1 * FE_XSTRIDE + j1 * FE_JSTRIDE + mm];
        wave_sync();
        dft8_dif(v);
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) xb[k1 + 8 * j1 + 64 * j2] = v[BR[j2]];
        wave_sync();
        // ---- real-FFT split + power ---------------------------------------------------------------------
        // The power spectrum P[0..512] overlays float2 slots 0..256 of the buffer: X[k] for k <= 256 is read before any P is
        // written; X[512-k] for k < 256 sits in slots 257..511, which are never overwritten (k = 256 pairs with itself).
        float2 xa[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int k = lane + 64 * i;
            if (k > 256) continue;
            xa[i] = xb[k];
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int k = lane + 64 * i;
            if (k > 256) continue;
            if (k == 0) {
                const float2 z0 = xa[i];
                const float x0 = z0.x + z0.y, xm = z0.x - z0.y;
                P[0] = x0 * x0;
                P[512] = xm * xm;
            } else {
                const int k2 = 512 - k;
                const float2 a = xa[i], c = (k == 256) ? xa[i] : xb[k2];
                const float2 E = make_float2(0.5f * (a.x + c.x), 0.5f * (a.y - c.y));
                const float2 O = make_float2(0.5f * (a.y + c.y), -0.5f * (a.x - c.x));
                const float2 wo = cmul(tw[k], O);
                const float2 Xk = cadd(E, wo);
                const float2 Xk2 = make_float2(E.x - wo.x, -(E.y - wo.y));
                P[k] = Xk.x * Xk.x + Xk.y * Xk.y;
                P[k2] = Xk2.x * Xk2.x + Xk2.y * Xk2.y;
            }
        }
        wave_sync();
        // ---- mel: one lane per filter -------------------------------------------------------------------
        for (int m = lane; m < p.n_mels; m += 64) {
            const int lo = p.mel_lo[m], cnt = moff[m + 1] - moff[m];
            const float* __restrict__ pr = P + lo;
            float acc = 0.f;
            if (mw_ok) {
                const float* __restrict__ wr = mw + moff[m];
                int j = 0;
                for (; j + 4 <= cnt; j += 4) {
                    const float w0 = wr[j], w1 = wr[j + 1], w2 = wr[j + 2], w3 = wr[j + 3];
                    const float p0 = pr[j], p1 = pr[j + 1], p2 = pr[j + 2], p3 = pr[j + 3];
                    acc = fmaf(w0, p0, acc); acc = fmaf(w1, p1, acc); acc = fmaf(w2, p2, acc); acc = fmaf(w3, p3, acc);
                }
                for (; j < cnt; ++j) acc = fmaf(wr[j], pr[j], acc);
            } else {
                const float* __restrict__ row = p.melT + (size_t)m * 513 + lo;
                for (int j = 0; j < cnt; ++j) acc = fmaf(row[j], pr[j], acc);
            }
            float val = 10.0f * log10f(fmaxf(1e-10f, acc));
            if (p.mean) val = (val - p.mean[m]) / p.stdv[m];
            if (live) p.out[(size_t)fidx * p.n_mels + m] = val;
        }
        wave_sync();
    }
}

#define FB_MAXHOP 320

typedef __attribute__((ext_vector_type(4))) float fb_f32x4;

// -------------------------------------------------------------------------------------------------
// nfft = 1024, batched kernel (round 3): 8 consecutive frames per 512-thread workgroup (one wave per frame), two workgroups per
// CU, mel projection on the fp32 matrix pipe straight from the power spectra.
//
// History.  Round 1 (frontend1024_kernel above): every frame fetched its own 1024 samples although consecutive frames share 704
// of them, window / twiddles were re-read from LDS per frame, the mel projection ran one lane per filter (0.40 ms at B = 32).
// Round 2: 16 frames per 1024-thread workgroup, the batch's samples staged ONCE in LDS with 16-byte loads (next batch prefetched
// into registers), lane constants in registers, mel as three v_mfma_f32_16x16x32_bf16 on bf16 hi/lo images of the power spectra
// (0.31-0.35 ms; PMC: 680 instructions per frame, 0.20 instructions / cycle / SIMD, waves parked 52 %: one 16-wave workgroup per
// CU meets at three barriers per batch, the mel phase is unbalanced, the hi/lo images cost ~80 instructions and 35 KB of LDS).
// This kernel (0.277 ms, 580 instructions per frame; with ONE workgroup per CU it takes 0.354 ms -- the kernel is bound by the
// latency of its LDS round trips, occupancy is what pays):
//   * the batch's 7*hop + 1024 samples are staged once in LDS (next batch prefetched into registers), window in registers;
//   * the power spectrum P[0..512] stays fp32 where the real-FFT split wrote it (the wave's exchange buffer; row pitch 1156
//     floats: the eight rows of a batch sit on different banks) and D[frame][mel] = P.W runs as v_mfma_f32_16x16x4_f32
//     (fp32 products: no hi/lo split, no conversion pass, no second image);
//   * the (16-mel tile, bin range) schedule is computed once per workgroup from the filter bands: every tile gets at least one
//     wave, the other four go to the tiles with the most k-steps per wave (bench filter bank: 8 / 13 / 2 x 16 / 4 x 21 steps);
//     a wave's B fragments (its filter weights) are loop-invariant and live in registers (up to FC_NBR steps; more: from memory);
//   * 71 KB of LDS per workgroup: two workgroups per CU run their phases independently, so one's barrier waits overlap the
//     other's FFTs; two barriers per batch (the next batch's samples are written behind the second one);
//   * batch / clip indices advance incrementally (no divisions in the loop), normalisation constants are thread constants.
// The FFT itself is the register radix-8 x 3 scheme of frontend1024_kernel.
// -------------------------------------------------------------------------------------------------
#define FC_FR 8                          // frames per batch (one per wave)
#define FC_NSAMP (7 * FB_MAXHOP + 1024)
#define FC_XP (8 * FE_XSTRIDE + 2)       // float2 pitch between the waves' exchange buffers (1156 floats = 4 mod 32 banks)
#define FC_NBR 24                        // k-steps of a wave whose filter weights are kept in registers

__global__ __launch_bounds__(512, 4) void frontend1024c_kernel(FrontParams p, int bpc, int nbatches) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* samp = reinterpret_cast<float*>(smem);                              // [FC_NSAMP]
    float2* xbuf = reinterpret_cast<float2*>(samp + FC_NSAMP);                 // [8][FC_XP]
    float* red = reinterpret_cast<float*>(xbuf + FC_FR * FC_XP);               // [8 waves][8 frames][16 mels]
    int* wsch = reinterpret_cast<int*>(red + 8 * 8 * 16);                      // [8][4]: tile, first bin, steps, -;  [32..39]: first wave / waves of tile t
    float2* twl = reinterpret_cast<float2*>(wsch + 48);                        // [320]: exp(-2 pi i k / 1024), the real-FFT split
    float2* tw2t = twl + 320;                                                  // [8 m2][8 j1]: w64^(m2 j1), pass 2
    float2* tw1t = tw2t + 64;                                                  // [8 k1][64 n2]: w512^(n2 k1), pass 1 (registers are needed elsewhere)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = p.samples, hop = p.hop, T = p.T, nm = p.n_mels;
    const int nsamp = 7 * hop + 1024;                  // samples of a batch
    const int nq = nsamp >> 2;                         // 16-byte groups (hop % 4 == 0); <= 816

    // ---- one-time setup: twiddle tables, the mel schedule -------------------------------------------------------------------
    if (tid < 320) twl[tid] = p.tw[tid];
    if (tid >= 384 && tid < 448) {
        const int k = 16 * ((tid - 384) >> 3) * ((tid - 384) & 7);             // < 1024
        const float2 t = p.tw[k & 511];
        tw2t[tid - 384] = (k & 512) ? make_float2(-t.x, -t.y) : t;
    }
    if (tid == 448) {
        int klo[4], nst[4], nw[4];
        for (int t = 0; t < 4; ++t) {
            int lo = 1 << 30, hi = 0;
            for (int m = 16 * t; m < 16 * t + 16 && m < nm; ++m)
                if (p.mel_hi[m] > p.mel_lo[m]) { lo = min(lo, p.mel_lo[m]); hi = max(hi, p.mel_hi[m]); }
            klo[t] = hi > 0 ? (lo & ~3) : 0;
            nst[t] = hi > 0 ? (hi - klo[t] + 3) >> 2 : 0;
            nw[t] = 1;
        }
        for (int extra = 0; extra < 4; ++extra) {          // the spare waves go where a wave carries the most steps
            int best = 0;
            for (int t = 1; t < 4; ++t)
                if (nst[t] * nw[best] > nst[best] * nw[t]) best = t;
            nw[best] += 1;
        }
        int w = 0;
        for (int t = 0; t < 4; ++t) {
            wsch[32 + 2 * t] = w;
            wsch[33 + 2 * t] = nw[t];
            const int per = (nst[t] + nw[t] - 1) / nw[t];
            for (int i = 0; i < nw[t]; ++i, ++w) {
                const int s0 = min(i * per, nst[t]), s1 = min(s0 + per, nst[t]);
                wsch[4 * w] = t;
                wsch[4 * w + 1] = klo[t] + 4 * s0;
                wsch[4 * w + 2] = s1 - s0;
            }
        }
    }

    // ---- lane constants: window, pass-1 twiddles ----------------------------------------------------------------------------
    float2 win2[8];
    {
        auto twid = [&](int k) -> float2 {   // exp(-2 pi i k / 1024) from the device table p.tw[0..511]
            const float2 t = p.tw[k & 511];
            return (k & 512) ? make_float2(-t.x, -t.y) : t;
        };
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) win2[n1] = *reinterpret_cast<const float2*>(p.window + 2 * (64 * n1 + lane));
        if (tid < 64)
            for (int k1 = 0; k1 < 8; ++k1) tw1t[k1 * 64 + tid] = twid(2 * tid * k1);
    }
    constexpr int BR[8] = {0, 4, 2, 6, 1, 5, 3, 7};   // X[k] sits at position BR[k] after dft8_dif
    float2* xb = xbuf + wv * FC_XP;
    float* P = reinterpret_cast<float*>(xb);
    // finalize role: thread = (frame fr, mel fm)
    const int fr = tid >> 6, fm = tid & 63;
    const float zmean = (p.mean && fm < nm) ? p.mean[fm] : 0.f;
    const float zstd = (p.mean && fm < nm) ? p.stdv[fm] : 1.f;
    const float zinv = 1.0f / zstd;            // (round 5: one reciprocal per thread instead of a 12-instruction division per frame; <= 1 ulp from x / std)

    // batches of this workgroup: index, clip, first frame -- advanced incrementally
    const int G = gridDim.x;
    int bidx = blockIdx.x, cb = bidx / bpc, ct = bidx - cb * bpc;             // current batch
    int nb_ = bidx, nbc = cb, nbt = ct;                                        // the batch `fetch` loads next
    fb_f32x4 pre[2] = {};
    auto fetch = [&]() {                                                       // loads batch (nb_, nbc, nbt), then advances it by G
        const bool ok = nb_ < nbatches;
        const float* __restrict__ wav = p.wave + (size_t)(ok ? nbc : 0) * L;
        const int s0 = (ok ? nbt : 0) * FC_FR * hop - 512;
        if (ok && s0 >= 0 && s0 + nsamp <= L) {          // interior batch: workgroup-uniform
            const fb_f32x4* src = reinterpret_cast<const fb_f32x4*>(wav + s0);
            if (tid < nq) pre[0] = src[tid];
            if (tid + 512 < nq) pre[1] = src[tid + 512];
        } else {
#pragma unroll 1
            for (int u = 0; u < 2; ++u) {
                const int q = tid + 512 * u;
                const int s = s0 + 4 * q;
                fb_f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok && q < nq) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int si = s + e;
                        if (si < 0) si = -si;                       // np.pad(mode='reflect'): edge not repeated
                        if (si >= L) si = 2 * (L - 1) - si;
                        si = si < 0 ? 0 : (si >= L ? L - 1 : si);   // (frames past the clip's last one are never stored)
                        v[e] = wav[si];
                    }
                }
                if (u == 0) pre[0] = v; else pre[1] = v;
            }
        }
        nb_ += G;
        nbt += G;
        while (nbt >= bpc) { nbt -= bpc; nbc += 1; }
    };
    auto stage = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = tid + 512 * u;
            if (q < nq) *reinterpret_cast<fb_f32x4*>(samp + 4 * q) = pre[u];
        }
    };
    fetch();
    __syncthreads();       // (setup writes: tables, schedule)

    // ---- this wave's share of the mel projection: tile, bin range, B fragments ----------------------------------------------
    const int mt = wsch[4 * wv], mk0 = wsch[4 * wv + 1], mst = wsch[4 * wv + 2];
    const int mr = lane & 15, mg = lane >> 4;
    const int mm = 16 * mt + mr;
    auto wval = [&](int s) -> float {
        const int k = mk0 + 4 * s + mg;
        return (s < mst && mm < nm && k < 513) ? p.melT[(size_t)mm * 513 + k] : 0.f;
    };
    float breg[FC_NBR];
#pragma unroll
    for (int s = 0; s < FC_NBR; ++s) breg[s] = wval(s);
    const float* arow = reinterpret_cast<const float*>(xbuf + (mr & 7) * FC_XP) + mk0 + mg;    // rows 8..15 repeat 0..7 (never stored)
    const int tw_first = wsch[32 + 2 * (fm >> 4)], tw_cnt = wsch[33 + 2 * (fm >> 4)];

    stage();
    fetch();
    __syncthreads();
#ifdef SED_STAMPS
    unsigned long long fst[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // make STAMPS=1: phase ticks of one wave (tools/fe_stamp.sh)
    int fsn = 0;
#define FE_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); fst[i] += t_ - fs0; fs0 = t_; } while (0)
#else
#define FE_STAMP(i) do { } while (0)
#endif
    for (; bidx < nbatches; bidx += G) {
#ifdef SED_STAMPS
        unsigned long long fs0 = __builtin_amdgcn_s_memtime();
        ++fsn;
#endif
        const int t0 = ct * FC_FR;
        // ---- FFT of frame t0 + wv (one wave per frame) ---------------------------------------------------------------------
        const int k1 = lane >> 3, m2 = lane & 7, j1 = lane & 7;
#if SED_FE_PK
        {
            fe_f2* xp = reinterpret_cast<fe_f2*>(xb);
            const fe_f2* tw1p = reinterpret_cast<const fe_f2*>(tw1t);
            const fe_f2* tw2p = reinterpret_cast<const fe_f2*>(tw2t);
            fe_f2 w[8];
            const float* fs = samp + wv * hop + 2 * lane;
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) w[n1] = pk_mul(*reinterpret_cast<const fe_f2*>(fs + 128 * n1), fe_pk(win2[n1]));
            dft8_dif_pk(w);
            FE_STAMP(0);
#if SED_FE_X1REG
            {
                float2 u[8];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) u[kk] = fe_unpk((kk == 0) ? w[BR[0]] : pk_cmul(w[BR[kk]], tw1p[kk * 64 + lane]));
                fe_transpose_reg_lane_hi(u);
#pragma unroll
                for (int m1 = 0; m1 < 8; ++m1) w[m1] = fe_pk(u[m1]);
            }
#else
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) xp[kk * FE_XSTRIDE + lane] = (kk == 0) ? w[BR[0]] : pk_cmul(w[BR[kk]], tw1p[kk * 64 + lane]);
            wave_sync();
#pragma unroll
            for (int m1 = 0; m1 < 8; ++m1) w[m1] = xp[k1 * FE_XSTRIDE + 8 * m1 + m2];
            wave_sync();
#endif
            dft8_dif_pk(w);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
                xp[k1 * FE_XSTRIDE + jj * FE_JSTRIDE + m2] = (jj == 0) ? w[BR[0]] : pk_cmul(w[BR[jj]], tw2p[SED_FE_SWZ ? jj * 8 + m2 : m2 * 8 + jj]);
            wave_sync();
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = xp[k1 * FE_XSTRIDE + j1 * FE_JSTRIDE + q];
            wave_sync();
            dft8_dif_pk(w);
#pragma unroll
            for (int j2 = 0; j2 < 8; ++j2) xp[fe_phi_lo(k1 + 8 * j1) + FE_FSTRIDE * j2] = w[BR[j2]];      // X[k] at fe_phi(k)
            wave_sync();
        }
#else
        float2 v[8];
        {
            const float* fs = samp + wv * hop + 2 * lane;
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) {
                const float2 r = *reinterpret_cast<const float2*>(fs + 128 * n1);
                v[n1] = make_float2(r.x * win2[n1].x, r.y * win2[n1].y);
            }
        }
        dft8_dif(v);
        FE_STAMP(0);      // sample reads, window, pass 1
#if SED_FE_X1REG
        {
            float2 u[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) u[kk] = (kk == 0) ? v[BR[0]] : cmul(v[BR[kk]], tw1t[kk * 64 + lane]);
            fe_transpose_reg_lane_hi(u);
#pragma unroll
            for (int m1 = 0; m1 < 8; ++m1) v[m1] = u[m1];
        }
#else
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) xb[kk * FE_XSTRIDE + lane] = (kk == 0) ? v[BR[0]] : cmul(v[BR[kk]], tw1t[kk * 64 + lane]);
        wave_sync();
#pragma unroll
        for (int m1 = 0; m1 < 8; ++m1) v[m1] = xb[k1 * FE_XSTRIDE + 8 * m1 + m2];
        wave_sync();
#endif
        dft8_dif(v);
#pragma unroll
        for (int j1 = 0; j1 < 8; ++j1) xb[k1 * FE_XSTRIDE + j1 * FE_JSTRIDE + m2] = (j1 == 0) ? v[BR[0]] : cmul(v[BR[j1]], tw2t[SED_FE_SWZ ? j1 * 8 + m2 : m2 * 8 + j1]);      // (symmetric table: eight consecutive entries per read instead of a stride of 8)
        wave_sync();
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = xb[k1 * FE_XSTRIDE + j1 * FE_JSTRIDE + q];
        wave_sync();
        dft8_dif(v);
#pragma unroll
        for (int j2 = 0; j2 < 8; ++j2) xb[fe_phi_lo(k1 + 8 * j1) + FE_FSTRIDE * j2] = v[BR[j2]];      // X[k] at fe_phi(k)
        wave_sync();
#endif
        FE_STAMP(1);      // exchange 1, pass 2, exchange 2, pass 3, spectrum store
        // real-FFT split + power: P[0..512] overlays float2 slots 0..256; every X the split needs is read before any P is written
        {
            float2 xa[5], xc[4];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int k = lane + 64 * i;
                if (k <= 256) xa[i] = xb[fe_phi_lo(lane) + FE_FSTRIDE * i];
            }
            // X[512 - k], k = lane + 64 i: 64 (7 - i) + (64 - lane) for lane > 0, 64 (8 - i) for lane 0 (i = 0: k = 0 has no partner --
            // the slot read is the row's padding and the value is not used)
            const int c1 = lane == 0 ? FE_FSTRIDE : fe_phi_lo(64 - lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) xc[i] = xb[c1 + FE_FSTRIDE * (7 - i)];
            wave_sync();
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int k = lane + 64 * i;
                if (k > 256) continue;
                if (k == 0) {
                    const float2 z0 = xa[i];
                    const float x0 = z0.x + z0.y, xm = z0.x - z0.y;
                    P[0] = x0 * x0;
                    P[512] = xm * xm;
                } else {
                    const int k2 = 512 - k;
                    const float2 a = xa[i], c = (i == 4) ? xa[i] : xc[i < 4 ? i : 0];      // (i == 4: k = 256 pairs with itself)
                    const float2 E = make_float2(0.5f * (a.x + c.x), 0.5f * (a.y - c.y));
                    const float2 O = make_float2(0.5f * (a.y + c.y), -0.5f * (a.x - c.x));
                    const float2 wo = cmul(twl[k], O);
                    const float2 Xk = cadd(E, wo);
                    const float2 Xk2 = make_float2(E.x - wo.x, -(E.y - wo.y));
                    P[k] = Xk.x * Xk.x + Xk.y * Xk.y;
                    P[k2] = Xk2.x * Xk2.x + Xk2.y * Xk2.y;
                }
            }
        }
        FE_STAMP(2);      // split + power
        __syncthreads();
        FE_STAMP(3);      // barrier 1
        // ---- the next batch's samples (every wave is past its reads of `samp`) ----------------------------------------------
        stage();
        fetch();
        FE_STAMP(4);      // stage (waits for the prefetched samples) + fetch issue
        // ---- mel projection: this wave's (tile, bin range) partial of D[frame][mel] --------------------------------------------
        {
            fb_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g4 = 0; g4 < FC_NBR / 4; ++g4) {
                if (4 * g4 < mst) {                       // wave-uniform
                    float a4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a4[e] = arow[4 * (4 * g4 + e)];      // (steps past mst: finite X / P words times weight 0)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e], breg[4 * g4 + e], acc, 0, 0, 0);
                }
            }
            for (int s = FC_NBR; s < mst; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * s], wval(s), acc, 0, 0, 0);
            if (lane < 32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) red[(wv * 8 + 4 * mg + e) * 16 + mr] = acc[e];      // D: row (frame) = 4*mg + e, column (mel) = mr
            }
        }
        FE_STAMP(5);      // mel projection
        __syncthreads();
        FE_STAMP(6);      // barrier 2
        // ---- one thread per (frame, mel): fixed-order sum of the tile's partials, log, z-score ------------------------------------
        {
            // (a tile has one to five waves: the same fixed-order sum as a loop over tw_cnt, without the divergent loop's exec-mask bookkeeping)
            const float* rp = red + (tw_first * 8 + fr) * 16 + (fm & 15);
            float s = rp[0];
            if (tw_cnt > 1) s += rp[128];
            if (tw_cnt > 2) s += rp[256];
            if (tw_cnt > 3) s += rp[384];
            if (tw_cnt > 4) s += rp[512];
            float val = 10.0f * log10f(fmaxf(1e-10f, s));
            if (p.mean) val = (val - zmean) * zinv;
            if (t0 + fr < T && fm < nm) p.out[((size_t)cb * T + t0 + fr) * nm + fm] = val;
        }
        ct += G;
        while (ct >= bpc) { ct -= bpc; cb += 1; }
        FE_STAMP(7);      // finalize: sum, log, z-score, store
        // (the next iteration's first barrier separates these reads of `red` from the next writes; the staged samples were made
        //  visible by the barrier above)
    }
#ifdef SED_STAMPS
    if (blockIdx.x == 8 && lane == 0 && (wv == 1 || wv == 6))
        printf("fe wave %d: %d batches; ticks per batch: load+pass1 %llu  x1+pass2+x2+pass3 %llu  split %llu  barrier1 %llu  stage+fetch %llu  mel %llu  barrier2 %llu  finalize %llu\n",
               wv, fsn, fst[0] / fsn, fst[1] / fsn, fst[2] / fsn, fst[3] / fsn, fst[4] / fsn, fst[5] / fsn, fst[6] / fsn, fst[7] / fsn);
#endif
}

// multichannel_complex_to_log_mel (preprocess.py:39-45) for an already computed complex spectrogram
// ("Complex" preprocessing mode, spectograms_dataset.py:104-110): one workgroup per frame.
__global__ __launch_bounds__(256) void complex_to_logmel_kernel(const float2* __restrict__ spec,
                                                                const float* __restrict__ melT,
                                                                const int* __restrict__ mel_lo,
                                                                const int* __restrict__ mel_hi,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ stdv,
                                                                float* __restrict__ out, int bins, int n_mels) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* P = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x;
    const size_t frame = blockIdx.x;
    const float2* __restrict__ row_in = spec + frame * bins;
    for (int k = tid; k < bins; k += 256) {
        const float2 v = row_in[k];
        const float a = sqrtf(v.x * v.x + v.y * v.y);   // np.abs(.) then **2, as the reference does
        P[k] = a * a;
    }
    __syncthreads();
    const int quad = tid & 3;
    for (int m = tid >> 2; m < n_mels; m += 64) {
        const int lo = mel_lo[m], hi = mel_hi[m];
        const float* __restrict__ row = melT + (size_t)m * bins;
        float acc = 0.f;
        for (int k = lo + quad; k < hi; k += 4) acc = fmaf(row[k], P[k], acc);
        acc += dpp_mov<0xB1>(acc);
        acc += dpp_mov<0x4E>(acc);
        if (quad == 0) {
            float v = 10.0f * log10f(fmaxf(1e-10f, acc));
            if (mean) v = (v - mean[m]) / stdv[m];
            out[frame * n_mels + m] = v;
        }
    }
}

extern "C" int sed_complex_to_logmel(const void* spec, const float* melT, const int* mel_lo, const int* mel_hi,
                                     const float* mean, const float* stdv, float* out, size_t nframes, int bins,
                                     int n_mels, void* stream) {
    SED_REQUIRE(bins > 0 && bins <= 32769 && n_mels > 0 && nframes > 0 && nframes < (1u << 31), "bad sizes");
    SED_REQUIRE((mean == nullptr) == (stdv == nullptr), "mean/std must both be given or both NULL");
    const size_t lds = (size_t)bins * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&complex_to_logmel_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { sed_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return 3; }
    }
    complex_to_logmel_kernel<<<(unsigned)nframes, 256, lds, (hipStream_t)stream>>>(
        (const float2*)spec, melT, mel_lo, mel_hi, mean, stdv, out, bins, n_mels);
    SED_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// In-loop "Complex" mode (spectograms_dataset.py:58-78, 104-135): crop gather from the resident
// STFT bank + augment_mix_samples (average of nmix crops) + augment_add_noise (real Gaussian noise)
// + transform (complex z-score, then log-mel), one workgroup per output frame.
// ---------------------------------------------------------------------------------------------
// Counter-based generator for the augmentation noise: splitmix64 of (seed, element counter) -> two
// 24-bit uniforms -> Box-Muller (the tests restate it in numpy; the integer part is bit-exact).
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ float counter_normal(unsigned long long seed, unsigned long long ctr) {
    const unsigned long long h = splitmix64(seed ^ splitmix64(ctr));
    const float u1 = ((float)((h >> 40) & 0xFFFFFFull) + 1.0f) * (1.0f / 16777216.0f);   // (0, 1]
    const float u2 = (float)((h >> 8) & 0xFFFFFFull) * (1.0f / 16777216.0f);            // [0, 1)
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

__global__ __launch_bounds__(256) void complex_augment_logmel_kernel(
    const float2* __restrict__ bank, const int* __restrict__ starts, const int* __restrict__ nmix,
    const float* __restrict__ noise_std, const float* __restrict__ noise, unsigned long long seed,
    const float2* __restrict__ cmean, const float* __restrict__ cstd, const float* __restrict__ melT,
    const int* __restrict__ mel_lo, const int* __restrict__ mel_hi, float* __restrict__ out, int crop, int bins,
    int n_mels) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* P = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x;
    const int b = blockIdx.x / crop, t = blockIdx.x - b * crop;
    const int nm = nmix[b];
    const float inv = 1.0f / (float)nm;
    const float nstd = noise_std ? noise_std[b] : 0.f;
    const size_t frame = (size_t)blockIdx.x;
    const float2* rows[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) rows[j] = bank + ((size_t)starts[b * 4 + (j < nm ? j : 0)] + t) * bins;
    for (int k = tid; k < bins; k += 256) {
        float2 v = rows[0][k];
#pragma unroll
        for (int j = 1; j < 4; ++j)
            if (j < nm) { const float2 w = rows[j][k]; v.x += w.x; v.y += w.y; }     // feature += new_feature (:131)
        if (nm > 1) { v.x *= inv; v.y *= inv; }                                          // feature /= (n + 1)     (:133)
        if (nstd > 0.f) {                                                                // += N(0, noise_var)     (:116)
            const float z = noise ? noise[frame * bins + k] : counter_normal(seed, frame * (size_t)bins + k);
            v.x = fmaf(nstd, z, v.x);
        }
        if (cmean) {                                                                     // (x - mean) / std       (:105)
            const float2 mu = cmean[k];
            const float is = 1.0f / cstd[k];
            v.x = (v.x - mu.x) * is;
            v.y = (v.y - mu.y) * is;
        }
        const float a = sqrtf(v.x * v.x + v.y * v.y);
        P[k] = a * a;
    }
    __syncthreads();
    const int quad = tid & 3;
    for (int m = tid >> 2; m < n_mels; m += 64) {
        const int lo = mel_lo[m], hi = mel_hi[m];
        const float* __restrict__ row = melT + (size_t)m * bins;
        float acc = 0.f;
        for (int k = lo + quad; k < hi; k += 4) acc = fmaf(row[k], P[k], acc);
        acc += dpp_mov<0xB1>(acc);
        acc += dpp_mov<0x4E>(acc);
        if (quad == 0) out[frame * n_mels + m] = 10.0f * log10f(fmaxf(1e-10f, acc));
    }
}

extern "C" int sed_complex_augment_logmel(const void* bank, size_t bank_frames, const int* starts_host,
                                          const int* nmix_host, const int* starts, const int* nmix,
                                          const float* noise_std, const float* noise, unsigned long long seed,
                                          const void* cmean, const float* cstd, const float* melT, const int* mel_lo,
                                          const int* mel_hi, float* out, int B, int crop, int bins, int n_mels,
                                          void* stream) {
    SED_REQUIRE(B > 0 && crop > 0 && bins > 0 && bins <= 32769 && n_mels > 0, "bad sizes");
    SED_REQUIRE((size_t)B * crop < (1u << 31), "too many frames for one launch");
    SED_REQUIRE((cmean == nullptr) == (cstd == nullptr), "mean/std must both be given or both NULL");
    SED_REQUIRE(starts_host && nmix_host && starts && nmix, "crop tables are needed on the host (validation) and on the device");
    for (int b = 0; b < B; ++b) {       // every gathered row must lie inside the bank: checked before the launch
        SED_REQUIRE(nmix_host[b] >= 1 && nmix_host[b] <= 4, "nmix must be 1..4");
        for (int j = 0; j < nmix_host[b]; ++j)
            SED_REQUIRE(starts_host[b * 4 + j] >= 0 && (size_t)starts_host[b * 4 + j] + crop <= bank_frames,
                        "crop outside the spectrogram bank");
    }
    const size_t lds = (size_t)bins * sizeof(float);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&complex_augment_logmel_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { sed_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return 3; }
    }
    complex_augment_logmel_kernel<<<(unsigned)(B * crop), 256, lds, (hipStream_t)stream>>>(
        (const float2*)bank, starts, nmix, noise_std, noise, seed, (const float2*)cmean, cstd, melT, mel_lo, mel_hi, out,
        crop, bins, n_mels);
    SED_LAUNCH_CHECK();
    return 0;
}

// "logMel" mode of SpectogramDataset.__getitem__ + transform (spectograms_dataset.py:66-69, 104-108):
// crop gather from the resident log-mel bank and per-mel z-score, for a whole batch.
__global__ __launch_bounds__(256) void logmel_crops_kernel(const float* __restrict__ bank, const int* __restrict__ starts,
                                                           const float* __restrict__ mean, const float* __restrict__ stdv,
                                                           float* __restrict__ out, int crop, int n_mels, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int m = (int)(i % n_mels);
    const size_t f = i / n_mels;
    const int b = (int)(f / crop), t = (int)(f - (size_t)b * crop);
    float v = bank[((size_t)starts[b] + t) * n_mels + m];
    if (mean) v = (v - mean[m]) / stdv[m];
    out[i] = v;
}

extern "C" int sed_logmel_crops(const float* bank, size_t bank_frames, const int* starts_host, const int* starts,
                                const float* mean, const float* stdv, float* out, int B, int crop, int n_mels,
                                void* stream) {
    SED_REQUIRE(B > 0 && crop > 0 && n_mels > 0 && starts_host && starts, "bad arguments");
    SED_REQUIRE((mean == nullptr) == (stdv == nullptr), "mean/std must both be given or both NULL");
    for (int b = 0; b < B; ++b)
        SED_REQUIRE(starts_host[b] >= 0 && (size_t)starts_host[b] + crop <= bank_frames, "crop outside the feature bank");
    const size_t total = (size_t)B * crop * n_mels;
    logmel_crops_kernel<<<(unsigned)cdivz(total, 256), 256, 0, (hipStream_t)stream>>>(bank, starts, mean, stdv, out, crop,
                                                                                      n_mels, total);
    SED_LAUNCH_CHECK();
    return 0;
}

static int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

extern "C" size_t sed_logmel_ws_bytes(int B, int samples, int nfft, int hop) {
    (void)B; (void)samples; (void)hop;
    return (size_t)(nfft / 2) * sizeof(float2);
}

static int launch_front(bool logmel, FrontParams& p, hipStream_t st) {
    const int M = p.nfft / 2;
    const size_t lds = (size_t)M * sizeof(float2);
    if (lds > 64 * 1024) {
        hipError_t e;
        if (logmel)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&frontend_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&frontend_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { sed_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return 3; }
    }
    // The twiddle table exp(-2 pi i k / nfft), k < nfft / 2, used to be rebuilt into the caller's workspace by a launch of its own in
    // front of every call (4.5 us of a 4.3 ms train step).  Round 5: one library-owned table per (device, nfft), built on first use
    // (that one call synchronises its stream, so later calls on any stream find it complete) and never freed.  While a stream is being
    // captured (no allocation / synchronisation allowed) and for more than four transform sizes per device the workspace path remains.
    {
        static std::mutex tw_mu;
        static struct { int dev, nfft; float2* tab; } tw_cache[64];
        static int tw_n = 0;
        int dev = 0;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
        const float2* tab = nullptr;
        if (!capturing && hipGetDevice(&dev) == hipSuccess) {
            std::lock_guard<std::mutex> lk(tw_mu);
            int per_dev = 0;
            for (int i = 0; i < tw_n; ++i) {
                if (tw_cache[i].dev != dev) continue;
                ++per_dev;
                if (tw_cache[i].nfft == p.nfft) tab = tw_cache[i].tab;
            }
            if (tab == nullptr && per_dev < 4 && tw_n < 64) {
                float2* t = nullptr;
                if (hipMalloc(reinterpret_cast<void**>(&t), (size_t)M * sizeof(float2)) == hipSuccess) {
                    twiddle_kernel<<<cdiv(M, 256), 256, 0, st>>>(t, p.nfft);
                    if (hipStreamSynchronize(st) == hipSuccess) {
                        tw_cache[tw_n].dev = dev; tw_cache[tw_n].nfft = p.nfft; tw_cache[tw_n].tab = t;
                        ++tw_n;
                        tab = t;
                    } else {
                        (void)hipGetLastError();
                        (void)hipFree(t);
                    }
                } else {
                    (void)hipGetLastError();
                }
            }
        }
        if (tab != nullptr) p.tw = tab;
        else twiddle_kernel<<<cdiv(M, 256), 256, 0, st>>>(const_cast<float2*>(p.tw), p.nfft);
    }
    const char* fek = sed_getenv("SED_FE_KERNEL");          // A/B: 1 = the one-wave-per-frame kernel
    if (logmel && p.nfft == 1024 && p.n_mels <= 64 && p.hop <= FB_MAXHOP && p.hop % 4 == 0 && p.samples % 4 == 0 &&
        (reinterpret_cast<uintptr_t>(p.wave) & 15) == 0 && p.samples >= 1024 && !(fek && fek[0] == '1')) {
        const size_t ldc = (size_t)FC_NSAMP * 4 + (size_t)FC_FR * FC_XP * 8 + (size_t)8 * 8 * 16 * 4 + (size_t)48 * 4 + (size_t)(320 + 64 + 512) * 8;
        if (int rc_ = sed_set_max_lds<&frontend1024c_kernel>(ldc)) return rc_;
        const int bpc = cdiv(p.T, FC_FR), nb = p.B * bpc;
        int wgs = 512;                                            // two workgroups per CU
        if (const char* e = sed_getenv("SED_FE_BLOCKS")) wgs = atoi(e) > 0 ? atoi(e) : wgs;     // tuning knob
        frontend1024c_kernel<<<nb < wgs ? nb : wgs, 512, ldc, st>>>(p, bpc, nb);
        return 0;
    }
    if (logmel && p.nfft == 1024 && p.n_mels <= 64) {
        const int nframes = p.B * p.T;
        const int blocks = cdiv(nframes, 4);

        int cap = 8192;         // (measured: 1280 .. 8192 workgroups run the same 0.40-0.41 ms -- the per-workgroup table setup is not what costs)
        if (const char* e = sed_getenv("SED_FE_BLOCKS")) cap = atoi(e) > 0 ? atoi(e) : cap;     // tuning knob
        frontend1024_kernel<<<blocks < cap ? blocks : cap, 256, 0, st>>>(p, nframes);
        return 0;
    }
    const int grid = p.B * p.T;
    if (p.nfft >= 4096 && !(fek && fek[0] == 'g')) {        // (SED_FE_KERNEL=g: the radix-2 kernel, for the A/B)
        const size_t ldb = (size_t)M * sizeof(float2);
        if (logmel) {
            if (int rc_ = sed_set_max_lds<&frontend_big_kernel<true>>(ldb)) return rc_;
            frontend_big_kernel<true><<<grid, 1024, ldb, st>>>(p);
        } else {
            if (int rc_ = sed_set_max_lds<&frontend_big_kernel<false>>(ldb)) return rc_;
            frontend_big_kernel<false><<<grid, 1024, ldb, st>>>(p);
        }
        return 0;
    }
    if (logmel) frontend_kernel<true><<<grid, 256, lds, st>>>(p);
    else frontend_kernel<false><<<grid, 256, lds, st>>>(p);
    return 0;
}

static int front_common_checks(int B, int samples, int nfft, int hop) {
    SED_REQUIRE(B > 0 && samples > 0 && hop > 0, "bad sizes");
    SED_REQUIRE(nfft >= 64 && nfft <= 32768 && (nfft & (nfft - 1)) == 0, "nfft must be a power of two in [64, 32768]");
    SED_REQUIRE(samples > nfft / 2, "reflect padding needs samples > nfft/2");
    return 0;
}

extern "C" int sed_logmel_fwd(const float* wave, const float* window, const float* melT, const int* mel_lo,
                              const int* mel_hi, const float* mean, const float* stdv, float* out, void* workspace,
                              int B, int samples, int nfft, int hop, int n_mels, void* stream) {
    if (int rc = front_common_checks(B, samples, nfft, hop)) return rc;
    SED_REQUIRE((mean == nullptr) == (stdv == nullptr), "mean/std must both be given or both NULL");
    FrontParams p;
    p.wave = wave; p.window = window; p.tw = (const float2*)workspace; p.melT = melT; p.mel_lo = mel_lo;
    p.mel_hi = mel_hi; p.mean = mean; p.stdv = stdv; p.out = out; p.spec = nullptr;
    p.B = B; p.samples = samples; p.nfft = nfft; p.hop = hop; p.T = 1 + samples / hop; p.n_mels = n_mels;
    p.logM = ilog2(nfft / 2);
    // Slaney triangles overlap pairwise: at most 2 non-zero weights per FFT bin -> 2*(nfft/2+1) entries
    p.fast_ok = (2 * (nfft / 2 + 1) + 64 <= FE_MAXNZ) ? 1 : 0;
    if (int rc = launch_front(true, p, (hipStream_t)stream)) return rc;
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_stft_fwd(const float* wave, const float* window, void* spec, void* workspace, int B, int samples,
                            int nfft, int hop, void* stream) {
    if (int rc = front_common_checks(B, samples, nfft, hop)) return rc;
    FrontParams p;
    p.wave = wave; p.window = window; p.tw = (const float2*)workspace; p.melT = nullptr; p.mel_lo = nullptr;
    p.mel_hi = nullptr; p.mean = nullptr; p.stdv = nullptr; p.out = nullptr; p.spec = (float2*)spec;
    p.B = B; p.samples = samples; p.nfft = nfft; p.hop = hop; p.T = 1 + samples / hop; p.n_mels = 0;
    p.logM = ilog2(nfft / 2);
    p.fast_ok = 0;
    if (int rc = launch_front(false, p, (hipStream_t)stream)) return rc;
    SED_LAUNCH_CHECK();
    return 0;
}
