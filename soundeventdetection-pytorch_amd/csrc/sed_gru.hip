// CRNN head (BASELINE config 4): conv_blocks -> mean over mel -> bidirectional
// GRU(hidden Hd) -> Linear(2*Hd, classes) -> interpolate.  torch.nn.GRU semantics (gate order r,z,n):
//     r = sig(W_ir x + b_ir + W_hr h + b_hr)        z = sig(W_iz x + b_iz + W_hz h + b_hz)
//     n = tanh(W_in x + b_in + r * (W_hn h + b_hn)) h' = (1 - z) * n + z * h
//
// Decomposition for the MI355X:
//   * everything that is not sequential is a plain GEMM on MFMA (gemm_nt_kernel): the input projection
//     gi = m.W_ih^T + b_ih for all time steps and both directions, and in the backward pass
//     dW_ih = dgi^T.m, dW_hh = dgh^T.h_prev (split-K over the B*t rows), dm = dgi.W_ih;
//   * the recurrence itself is latency bound (t = 750 dependent steps of a [32 x Hd]x[Hd x 3Hd] product):
//     one workgroup per (direction, 32 batch rows) keeps h in registers (fp32) + LDS (MFMA operand
//     copy) and streams the 3Hd x Hd recurrent matrix from L2 every step in MFMA-fragment order
//     (pre-packed, 1 KB coalesced per fragment); wave w owns hidden units [32w, 32w+32) of all three
//     gates, so the gate math is register-local and each step costs two workgroup barriers.
//     No inter-workgroup synchronisation anywhere (nothing can spin).
#include "conv_common.h"

#include <stdlib.h>

// ---------------------------------------------------------------------------------------------
// mean over the mel axis (spectogram_models.py:193 `torch.mean(x, dim=3)`) and its backward
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mel_mean_fwd_kernel(const T* __restrict__ feat, float* __restrict__ m,
                                                           size_t rows, int Wf, int C, int Cp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * (size_t)C) return;
    const size_t r = i / C;
    const int c = (int)(i - r * C);
    float s = 0.f;
    for (int w = 0; w < Wf; ++w) s += to_f(feat[(r * Wf + w) * Cp + c]);
    m[i] = s / (float)Wf;
}
template <typename T>
__global__ __launch_bounds__(256) void mel_mean_bwd_kernel(const float* __restrict__ dm, T* __restrict__ dfeat,
                                                           size_t rows, int Wf, int C, int Cp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * (size_t)Wf * Cp) return;
    const int c = (int)(i % Cp);
    const size_t r = i / ((size_t)Wf * Cp);
    const float v = c < C ? dm[r * C + c] / (float)Wf : 0.f;
    dfeat[i] = from_f<T>(v);
}

extern "C" int sed_mel_mean_fwd(int dtype, const void* feat, float* m, size_t rows, int Wf, int C, int Cp, void* stream) {
    SED_REQUIRE(rows > 0 && Wf > 0 && C > 0 && C <= Cp, "bad sizes");
    const unsigned grid = (unsigned)cdivz(rows * (size_t)C, 256);
    if (dtype == SED_BF16) mel_mean_fwd_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)feat, m, rows, Wf, C, Cp);
    else if (dtype == SED_F32) mel_mean_fwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)feat, m, rows, Wf, C, Cp);
    else SED_REQUIRE(false, "bad dtype");
    SED_LAUNCH_CHECK();
    return 0;
}
extern "C" int sed_mel_mean_bwd(int dtype, const float* dm, void* dfeat, size_t rows, int Wf, int C, int Cp, void* stream) {
    SED_REQUIRE(rows > 0 && Wf > 0 && C > 0 && C <= Cp, "bad sizes");
    const unsigned grid = (unsigned)cdivz(rows * (size_t)Wf * Cp, 256);
    if (dtype == SED_BF16) mel_mean_bwd_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(dm, (bf16_t*)dfeat, rows, Wf, C, Cp);
    else if (dtype == SED_F32) mel_mean_bwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(dm, (float*)dfeat, rows, Wf, C, Cp);
    else SED_REQUIRE(false, "bad dtype");
    SED_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// C[M][N] (+= over split-K partials) = A[M][K] . B[N][K]^T (+ bias[N]); fp32 in memory, MFMA compute in
// T (bf16 operands / fp32 accumulate, or fp32 32x32x2).  128x128 tile, 4 waves of 64x64, BK = 32.
// ---------------------------------------------------------------------------------------------
template <typename T> struct GemmLds;
template <> struct GemmLds<bf16_t> { static constexpr int STRIDE = 40; };    // 80 B rows: conflict-free 16 B reads
template <> struct GemmLds<float> { static constexpr int STRIDE = 33; };

template <typename T>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Bm,
                                                      int ldb, const float* __restrict__ bias, float* __restrict__ C,
                                                      int ldc, int M, int N, int K, int kchunk, size_t split_stride) {
    constexpr int KR = EL<T>::KR, KSTEP = EL<T>::KSTEP, LS = GemmLds<T>::STRIDE;
    typedef typename EL<T>::frag_t frag_t;
    __shared__ __attribute__((aligned(16))) T As[128 * LS];
    __shared__ __attribute__((aligned(16))) T Bs[128 * LS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int k_begin = blockIdx.z * kchunk;
    const int k_end = (k_begin + kchunk < K) ? k_begin + kchunk : K;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int r = lane & 31, hh = lane >> 5;
    // 128 x 32 of A and B per k-step: 1024 float4 each, 4 per thread -- the NEXT step's values are fetched into registers while this step's
    // MFMAs run (round 4: the plain load -> LDS -> barrier -> MFMA loop exposed a memory latency per 32 columns; these products have K = 24000)
    f32x4 va[4], vb[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int item = tid + u * 256;
            const int row = item >> 3, kq = (item & 7) * 4;
            va[u] = {0.f, 0.f, 0.f, 0.f};
            vb[u] = {0.f, 0.f, 0.f, 0.f};
            const int k = k0 + kq;
            if (m0 + row < M) {
                const float* pa = A + (size_t)(m0 + row) * lda + k;
                if (k + 3 < k_end) va[u] = *reinterpret_cast<const f32x4*>(pa);
                else
                    for (int e = 0; e < 4; ++e) if (k + e < k_end) va[u][e] = pa[e];
            }
            if (n0 + row < N) {
                const float* pb = Bm + (size_t)(n0 + row) * ldb + k;
                if (k + 3 < k_end) vb[u] = *reinterpret_cast<const f32x4*>(pb);
                else
                    for (int e = 0; e < 4; ++e) if (k + e < k_end) vb[u][e] = pb[e];
            }
        }
    };
    if (k_begin < k_end) fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += 32) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int item = tid + u * 256;
            const int row = item >> 3, kq = (item & 7) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                As[row * LS + kq + e] = from_f<T>(va[u][e]);
                Bs[row * LS + kq + e] = from_f<T>(vb[u][e]);
            }
        }
        __syncthreads();
        if (k0 + 32 < k_end) fetch(k0 + 32);
#pragma unroll
        for (int ks = 0; ks < 32 / KSTEP; ++ks) {
            frag_t af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const frag_t*>(&As[(wm * 64 + i * 32 + r) * LS + ks * KSTEP + KR * hh]);
                bf[i] = *reinterpret_cast<const frag_t*>(&Bs[(wn * 64 + i * 32 + r) * LS + ks * KSTEP + KR * hh]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma(af[i], bf[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* __restrict__ Cz = C + (size_t)blockIdx.z * split_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            if (n >= N) continue;
            const float bv = bias ? bias[n] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (m < M) Cz[(size_t)m * ldc + n] = acc[i][j][e] + bv;
            }
        }
}

// out[i] = bias-free sum over splits (fixed order)
__global__ __launch_bounds__(256) void gemm_split_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int ldc,
                                                                int M, int N, int nsplit, size_t split_stride) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * N) return;
    const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += ws[(size_t)z * split_stride + i];
    C[(size_t)m * ldc + n] = s;
}

extern "C" size_t sed_gemm_nt_ws_floats(int M, int N, int ksplit) {
    return ksplit > 1 ? (size_t)ksplit * M * N : 0;
}

extern "C" int sed_gemm_nt(int compute_dtype, const float* A, int lda, const float* B, int ldb, const float* bias, float* C,
                           int ldc, int M, int N, int K, int ksplit, float* workspace, void* stream) {
    SED_REQUIRE(M > 0 && N > 0 && K > 0 && ksplit >= 1, "bad sizes");
    SED_REQUIRE(lda >= K && ldb >= K && ldc >= N, "leading dimensions too small");
    SED_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0, "A and B rows must be 16-byte aligned");
    SED_REQUIRE(ksplit == 1 || (workspace != nullptr && bias == nullptr), "split-K needs a workspace and no bias");
    hipStream_t st = (hipStream_t)stream;
    int kchunk = (int)cdivz(cdivz(K, ksplit), 32) * 32;
    const int nsplit = (int)cdivz(K, kchunk);
    dim3 grid(cdiv(N, 128), cdiv(M, 128), nsplit);
    float* dst = nsplit > 1 ? workspace : C;
    const int ld = nsplit > 1 ? N : ldc;
    const size_t ss = nsplit > 1 ? (size_t)M * N : 0;
    if (compute_dtype == SED_BF16)
        gemm_nt_kernel<bf16_t><<<grid, 256, 0, st>>>(A, lda, B, ldb, bias, dst, ld, M, N, K, kchunk, ss);
    else if (compute_dtype == SED_F32)
        gemm_nt_kernel<float><<<grid, 256, 0, st>>>(A, lda, B, ldb, bias, dst, ld, M, N, K, kchunk, ss);
    else
        SED_REQUIRE(false, "bad dtype");
    SED_LAUNCH_CHECK();
    if (nsplit > 1) {
        gemm_split_reduce_kernel<<<(unsigned)cdivz((size_t)M * N, 256), 256, 0, st>>>(workspace, C, ldc, M, N, nsplit, ss);
        SED_LAUNCH_CHECK();
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Round 6: C[M][N] = sum_k A[k][m] . B[k - shift][n]  -- both operands ROW-major with the reduction index on the rows (the
// weight-gradient products of the recurrence: A = dgi / dgh [B*t][3Hd], B = m / hseq [B*t][C]), so the five transposes that put the
// reduction axis on the columns for gemm_nt_kernel (0.13 ms of the 4.6 ms CRNN step) are not needed: the 32-row k-step of A and B goes
// to LDS as it lies in memory ([k][128] rows, coalesced 512-byte loads) and the MFMA fragments -- eight consecutive k of one column --
// come out of it through ds_read_b64_tr_b16, the conv weight-gradient kernels' read (bf16); fp32: one value per lane, read directly.
// B's rows can be shifted by +-1 INSIDE sequences of `seq` rows (the previous hidden state of a forward / reverse recurrence; a row
// that leaves its sequence is zero), and the column sums of A (the bias gradients) fall out of the loader's registers
// (colsum, nullable; written by the n-tile-0 workgroups, split-K partials like C's).  128 x 128 tile, 4 waves of 64 x 64.
// ---------------------------------------------------------------------------------------------
template <typename T> struct GemmTnLds;
template <> struct GemmTnLds<bf16_t> { static constexpr int STRIDE = 136; };   // 272-byte rows: the four k-rows of a transposing read sit 16 B apart in the bank cycle
template <> struct GemmTnLds<float> { static constexpr int STRIDE = 132; };

// one problem of a batched launch (sed_gemm_tn_batch: the four weight-gradient products of the BPTT tail in ONE launch + ONE reduction)
struct GemmTnProb {
    const float* A; const float* B; float* C; float* colsum;      // C / colsum: the split-K slabs when nsplit > 1
    float* Cout; float* csout;                                     // final destinations (the reduction's outputs)
    int lda, ldb, ldc, ldo, M, N, K, seq, shift, kchunk, gx, gy, nsplit;
    size_t split_stride, cs_stride;
};
constexpr int kGemmTnMax = 8;
struct GemmTnBatch {
    GemmTnProb p[kGemmTnMax];
    int wg0[kGemmTnMax + 1];       // first workgroup of problem i in the product launch
    int rg0[kGemmTnMax + 1];       // ... in the reduction launch
    int n;
};

template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnBatch bt) {
    int pi = 0;
    while (pi + 1 < bt.n && (int)blockIdx.x >= bt.wg0[pi + 1]) ++pi;
    const GemmTnProb& q = bt.p[pi];
    const float* __restrict__ A = q.A;
    const float* __restrict__ Bm = q.B;
    float* __restrict__ C = q.C;
    float* __restrict__ colsum = q.colsum;
    const int lda = q.lda, ldb = q.ldb, ldc = q.ldc, M = q.M, N = q.N, K = q.K, seq = q.seq, shift = q.shift, kchunk = q.kchunk;
    const size_t split_stride = q.split_stride, cs_stride = q.cs_stride;
    const int lwg = (int)blockIdx.x - bt.wg0[pi];
    const int bx = lwg % q.gx, by = (lwg / q.gx) % q.gy, bz = lwg / (q.gx * q.gy);
    constexpr int LS = GemmTnLds<T>::STRIDE;
    typedef typename EL<T>::frag_t frag_t;
    constexpr int KT = 64;                                     // rows per k-step (two loads in flight per operand and thread more than 32 would give)
    __shared__ __attribute__((aligned(16))) T As[KT * LS];
    __shared__ __attribute__((aligned(16))) T Bs[KT * LS];
    __shared__ float cred[8][128];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by * 128, n0 = bx * 128;
    const int k_begin = bz * kchunk;
    const int k_end = (k_begin + kchunk < K) ? k_begin + kchunk : K;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // a k-step = KT rows x 128 columns of A and of B: KT / 8 float4 per operand and thread: thread -> (rows lr + 8 u, columns lc ..)
    const int lc = (tid & 31) * 4, lr = tid >> 5;              // this thread's column group and first row
    constexpr int NU = KT / 8;
    f32x4 va[NU], vb[NU];
    f32x4 csum = {0.f, 0.f, 0.f, 0.f};
    const bool want_cs = colsum != nullptr && bx == 0;
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int k = k0 + lr + 8 * u;
            va[u] = {0.f, 0.f, 0.f, 0.f};
            vb[u] = {0.f, 0.f, 0.f, 0.f};
            if (k < k_end) {
                const int m = m0 + lc;
                const float* pa = A + (size_t)k * lda + m;
                if (m + 3 < M) va[u] = *reinterpret_cast<const f32x4*>(pa);
                else
                    for (int e = 0; e < 4; ++e) if (m + e < M) va[u][e] = pa[e];
                const int pos = k % seq, sp = pos - shift;     // B row k - shift of the same sequence, zero outside it
                if (sp >= 0 && sp < seq) {
                    const int n = n0 + lc;
                    const float* pb = Bm + (size_t)(k - shift) * ldb + n;
                    if (n + 3 < N) vb[u] = *reinterpret_cast<const f32x4*>(pb);
                    else
                        for (int e = 0; e < 4; ++e) if (n + e < N) vb[u][e] = pb[e];
                }
            }
        }
    };
    // fragment addressing: bf16 -> transposing reads (lane supplies k-row 8 hh + qq (+4) and the 4 columns 16 gbit + 4 pp ..), fp32 -> direct
    const int r = lane & 31, hh = lane >> 5;
    const int i16 = lane & 15, gbit = (lane >> 4) & 1, qq = i16 >> 2, pp = i16 & 3;
    if (k_begin < k_end) fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += KT) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int row = lr + 8 * u;
            if (want_cs) csum += va[u];
            if constexpr (sizeof(T) == 2) {
                float a4[4] = {va[u][0], va[u][1], va[u][2], va[u][3]}, b4[4] = {vb[u][0], vb[u][1], vb[u][2], vb[u][3]};
                store4<T>(&As[row * LS + lc], a4);
                store4<T>(&Bs[row * LS + lc], b4);
            } else {
                *reinterpret_cast<f32x4*>(&As[row * LS + lc]) = va[u];
                *reinterpret_cast<f32x4*>(&Bs[row * LS + lc]) = vb[u];
            }
        }
        __syncthreads();
        if (k0 + KT < k_end) fetch(k0 + KT);
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < KT / 16; ++ks) {    // 16 k per MFMA
                frag_t af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ka = (16 * ks + 8 * hh + qq) * LS + 16 * gbit + 4 * pp;
                    af[i] = join_tr(ds_read_tr16_b64(&As[ka + wm * 64 + i * 32]), ds_read_tr16_b64(&As[ka + 4 * LS + wm * 64 + i * 32]));
                    bf[i] = join_tr(ds_read_tr16_b64(&Bs[ka + wn * 64 + i * 32]), ds_read_tr16_b64(&Bs[ka + 4 * LS + wn * 64 + i * 32]));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma(af[i], bf[j], acc[i][j]);
            }
        } else {
#pragma unroll 4
            for (int ks = 0; ks < KT / 2; ++ks) {     // 2 k per MFMA: lane (r, hh) holds column r of k-row 2 ks + hh
                float af[2], bf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = As[(2 * ks + hh) * LS + wm * 64 + i * 32 + r];
                    bf[i] = Bs[(2 * ks + hh) * LS + wn * 64 + i * 32 + r];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = mfma(af[i], bf[j], acc[i][j]);
            }
        }
        __syncthreads();
    }
    float* __restrict__ Cz = C + (size_t)bz * split_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
            if (n >= N) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (m < M) Cz[(size_t)m * ldc + n] = acc[i][j][e];
            }
        }
    if (want_cs) {          // this thread's four columns over its rows -> fixed-order sum over the eight row groups
#pragma unroll
        for (int e = 0; e < 4; ++e) cred[lr][lc + e] = csum[e];
        __syncthreads();
        if (tid < 128 && m0 + tid < M) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) t += cred[g][tid];
            colsum[(size_t)bz * cs_stride + m0 + tid] = t;
        }
    }
}

// fixed-order sums over the split-K slabs of every problem of the batch: C (256 elements per workgroup), then the column sums
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(GemmTnBatch bt) {
    int pi = 0;
    while (pi + 1 < bt.n && (int)blockIdx.x >= bt.rg0[pi + 1]) ++pi;
    const GemmTnProb& q = bt.p[pi];
    const int lb = (int)blockIdx.x - bt.rg0[pi];
    const size_t mn = (size_t)q.M * q.N;
    const int cblocks = (int)((mn + 255) / 256);
    if (lb < cblocks) {
        const size_t i = (size_t)lb * 256 + threadIdx.x;
        if (i >= mn) return;
        const int m = (int)(i / q.N), n = (int)(i - (size_t)m * q.N);
        float s = 0.f;
        for (int z = 0; z < q.nsplit; ++z) s += q.C[(size_t)z * q.split_stride + i];
        q.Cout[(size_t)m * q.ldo + n] = s;
    } else {
        const int i = (lb - cblocks) * 256 + threadIdx.x;
        if (i >= q.M) return;
        float s = 0.f;
        for (int z = 0; z < q.nsplit; ++z) s += q.colsum[(size_t)z * q.cs_stride + i];
        q.csout[i] = s;
    }
}

extern "C" size_t sed_gemm_tn_ws_floats(int M, int N, int ksplit) {
    return ksplit > 1 ? (size_t)ksplit * ((size_t)M * N + M) : 0;
}

extern "C" int sed_gemm_tn_batch(int compute_dtype, const sed_gemm_tn_desc* d, int n, void* stream) {
    SED_REQUIRE(d != nullptr && n >= 1 && n <= kGemmTnMax, "1 .. 8 problems per launch");
    SED_REQUIRE(compute_dtype == SED_BF16 || compute_dtype == SED_F32, "bad dtype");
    GemmTnBatch bt = {};
    bt.n = n;
    bool any_split = false;
    for (int i = 0; i < n; ++i) {
        const sed_gemm_tn_desc& e = d[i];
        SED_REQUIRE(e.M > 0 && e.N > 0 && e.K > 0 && e.ksplit >= 1, "bad sizes");
        SED_REQUIRE(e.lda >= e.M && e.ldb >= e.N && e.ldc >= e.N, "leading dimensions too small");
        SED_REQUIRE(e.lda % 4 == 0 && e.ldb % 4 == 0 && (((uintptr_t)e.A | (uintptr_t)e.B) & 15) == 0, "A and B rows must be 16-byte aligned");
        SED_REQUIRE(e.seq >= 1 && e.K % e.seq == 0 && e.shift >= -1 && e.shift <= 1, "rows come in sequences of `seq`; shift in {-1, 0, +1}");
        SED_REQUIRE(e.ksplit == 1 || e.workspace != nullptr, "split-K needs a workspace");
        GemmTnProb& q = bt.p[i];
        q.kchunk = (int)cdivz(cdivz(e.K, e.ksplit), 64) * 64;
        q.nsplit = (int)cdivz(e.K, q.kchunk);
        const bool split = q.nsplit > 1;
        any_split |= split;
        q.A = e.A; q.B = e.B; q.lda = e.lda; q.ldb = e.ldb; q.M = e.M; q.N = e.N; q.K = e.K; q.seq = e.seq; q.shift = e.shift;
        q.Cout = e.C; q.ldo = e.ldc; q.csout = e.colsum;
        q.C = split ? e.workspace : e.C;
        q.ldc = split ? e.N : e.ldc;
        q.split_stride = split ? (size_t)e.M * e.N : 0;
        q.colsum = e.colsum ? (split ? e.workspace + (size_t)q.nsplit * e.M * e.N : e.colsum) : nullptr;
        q.cs_stride = split ? (size_t)e.M : 0;
        q.gx = cdiv(e.N, 128); q.gy = cdiv(e.M, 128);
        bt.wg0[i + 1] = bt.wg0[i] + q.gx * q.gy * q.nsplit;
        bt.rg0[i + 1] = bt.rg0[i] + (split ? (int)cdivz((size_t)e.M * e.N, 256) + (e.colsum ? cdiv(e.M, 256) : 0) : 0);
    }
    hipStream_t st = (hipStream_t)stream;
    if (compute_dtype == SED_BF16) gemm_tn_kernel<bf16_t><<<bt.wg0[n], 256, 0, st>>>(bt);
    else gemm_tn_kernel<float><<<bt.wg0[n], 256, 0, st>>>(bt);
    SED_LAUNCH_CHECK();
    if (any_split && bt.rg0[n] > 0) {
        gemm_tn_reduce_kernel<<<bt.rg0[n], 256, 0, st>>>(bt);
        SED_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int sed_gemm_tn(int compute_dtype, const float* A, int lda, const float* B, int ldb, float* C, int ldc, float* colsum,
                           int M, int N, int K, int seq, int shift, int ksplit, float* workspace, void* stream) {
    sed_gemm_tn_desc e = {A, B, C, colsum, workspace, lda, ldb, ldc, M, N, K, seq, shift, ksplit};
    return sed_gemm_tn_batch(compute_dtype, &e, 1, stream);
}

// dst[c][r + shift] = src[r][c] (fp32; 32x32 LDS tiles).  `seq`/`shift`: rows are grouped in sequences of
// `seq` consecutive rows and the copy is shifted by `shift` rows INSIDE each sequence, vacated columns
// become 0 (shift = +1: "previous time step" of a forward-running recurrence, -1: of a reverse one).
__global__ __launch_bounds__(256) void transpose_shift_kernel(const float* __restrict__ src, int lds_, float* __restrict__ dst,
                                                              int ldd, int R, int Ccols, int seq, int shift) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int i = ty; i < 32; i += 8) {
        const int rr = r0 + i, cc = c0 + tx;      // dst column rr takes src row rr - shift of the same sequence
        float v = 0.f;
        if (rr < R && cc < Ccols) {
            const int pos = rr % seq, sp = pos - shift;
            if (sp >= 0 && sp < seq) v = src[(size_t)(rr - shift) * lds_ + cc];
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int cc = c0 + i, rr = r0 + tx;
        if (cc < Ccols && rr < R) dst[(size_t)cc * ldd + rr] = tile[tx][i];
    }
}

extern "C" int sed_transpose_shift(const float* src, int ld_src, float* dst, int ld_dst, int R, int C, int seq, int shift,
                                   void* stream) {
    SED_REQUIRE(R > 0 && C > 0 && ld_src >= C && ld_dst >= R && seq >= 1 && R % seq == 0, "bad sizes");
    SED_REQUIRE(shift >= -1 && shift <= 1, "shift must be -1, 0 or +1");
    dim3 grid(cdiv(C, 32), cdiv(R, 32));
    transpose_shift_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(src, ld_src, dst, ld_dst, R, C, seq, shift);
    SED_LAUNCH_CHECK();
    return 0;
}

// out[r] = sum_c src[r][c]: one wave per row, fixed order
__global__ __launch_bounds__(256) void row_sums_kernel(const float* __restrict__ src, int ld, float* __restrict__ out, int R,
                                                       int Ccols) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    const float* __restrict__ rp = src + (size_t)row * ld;
    int c = lane;
    for (; c + 64 * 7 < Ccols; c += 64 * 8) {       // eight loads in flight, added in the same fixed order as a plain loop
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = rp[c + 64 * u];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < Ccols; c += 64) s += rp[c];
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
}
extern "C" int sed_row_sums(const float* src, int ld, float* out, int R, int C, void* stream) {
    SED_REQUIRE(R > 0 && C > 0 && ld >= C, "bad sizes");
    row_sums_kernel<<<cdiv(R, 4), 256, 0, (hipStream_t)stream>>>(src, ld, out, R, C);
    SED_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// recurrent weights in MFMA A-fragment order
//   fwd pack : rows j in [0, 3Hd) (gate*Hd + unit), reduction over k in [0, Hd)
//   bwd pack : rows k in [0, Hd), reduction over j in [0, 3Hd)         (the transposed product)
//   packed[((tile*KS + ks)*64 + lane)*KR + e] = Wop[32*tile + (lane&31)][ks*KSTEP + KR*(lane>>5) + e]
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gru_pack_kernel(const float* __restrict__ whh, T* __restrict__ pf, T* __restrict__ pb,
                                                       int Hd) {
    constexpr int KR = EL<T>::KR, KSTEP = EL<T>::KSTEP;
    const size_t total = (size_t)3 * Hd * Hd;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    {   // forward operator: Wop = W_hh [3Hd][Hd]
        const int KS = Hd / KSTEP;
        size_t t = i;
        const int e = t % KR; t /= KR;
        const int lane = t % 64; t /= 64;
        const int ks = t % KS;
        const int tile = (int)(t / KS);
        pf[i] = from_f<T>(whh[(size_t)(32 * tile + (lane & 31)) * Hd + ks * KSTEP + KR * (lane >> 5) + e]);
    }
    {   // backward operator: Wop = W_hh^T [Hd][3Hd]
        const int KS = 3 * Hd / KSTEP;
        size_t t = i;
        const int e = t % KR; t /= KR;
        const int lane = t % 64; t /= 64;
        const int ks = t % KS;
        const int tile = (int)(t / KS);
        pb[i] = from_f<T>(whh[(size_t)(ks * KSTEP + KR * (lane >> 5) + e) * Hd + 32 * tile + (lane & 31)]);
    }
}

template <typename T> struct SeqLds;
template <> struct SeqLds<bf16_t> { static constexpr int PAD = 8; };
template <> struct SeqLds<float> { static constexpr int PAD = 1; };

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 instructions each): the recurrence's gate math is the largest
// per-step cost once the weights are resident (32 x 256 values x 3 of these per step on ONE CU)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    const float e = __expf(-2.0f * fabsf(x));
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, x);
}

template <typename T> __device__ __forceinline__ void lds_put4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void lds_put4<bf16_t>(bf16_t* p, const float (&v)[4]) {
    bf16x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x4*>(p) = o;
}
template <> __device__ __forceinline__ void lds_put4<float>(float* p, const float (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = v[i];
}

struct GruSeqParams {
    const float* gi;       // [B*t][2*3Hd]  input projection incl. b_ih (fwd) / unused (bwd)
    const float* bhh;      // [2][3Hd]
    const void* wpack;     // [2][3Hd*Hd] packed recurrent operator (fwd or bwd pack), T
    float* hseq;           // [B*t][2*Hd]
    float* saved;          // [B*t][2][4][Hd]   r, z, n, (W_hn h + b_hn)
    const float* dhseq;    // bwd: [B*t][2*Hd]
    float* dgi;            // bwd: [B*t][2*3Hd]
    float* dgh;            // bwd: [B*t][2*3Hd]
    int B, t, Hd;
};

// ---- forward recurrence ------------------------------------------------------------------------
// MFMA orientation: D[batch][unit] (A = the h fragment of 32 batch rows, B = the packed weight fragment), so a LANE IS A HIDDEN
// UNIT and its 16 accumulator registers are batch rows (i&3) + 8*(i>>2) + 4*(lane>>5).  Every load of gi and every store of
// h / saved then covers two full 128-byte lines per instruction.  (The first version had lanes = batch rows: each 16-byte
// access went to a different (b, t) row, 32 partial lines per instruction, and the recurrence ran at 24 GB/s: 14.6 us/step.)
// RESN (bf16, Hd = 256): the n-gate fragments of every wave (16 KB each, 128 KB in all) stay in LDS for the whole sequence,
// so a third of the 393 KB recurrent matrix no longer crosses the CU's memory pipe every step.
template <typename T, bool RESN, int NREG>
__global__ __launch_bounds__(512) void gru_seq_fwd_kernel(GruSeqParams p) {
    constexpr int KR = EL<T>::KR, KSTEP = EL<T>::KSTEP, PAD = SeqLds<T>::PAD;
    typedef typename EL<T>::frag_t frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Hd = p.Hd, HS = Hd + PAD, KS = Hd / KSTEP, NW = Hd / 32;
    T* hs = reinterpret_cast<T*>(smem);                     // [32][Hd + PAD]
    frag_t* wn = reinterpret_cast<frag_t*>(hs + 32 * HS);   // RESN: [waves][KS][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int d = blockIdx.x & 1, bc = blockIdx.x >> 1;
    const int bl = lane & 31, hh = lane >> 5;
    const int unit = 32 * w + bl;                           // this lane's hidden unit
    const int t = p.t;
    for (int i = tid; i < 32 * HS; i += blockDim.x) hs[i] = from_f<T>(0.f);
    const frag_t* __restrict__ wp = reinterpret_cast<const frag_t*>(p.wpack) + (size_t)d * 3 * Hd * Hd / KR;
    const frag_t* __restrict__ wt[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) wt[g] = wp + ((size_t)(g * NW + w) * KS) * 64 + lane;
    if (RESN)
        for (int ks = 0; ks < KS; ++ks) wn[(w * KS + ks) * 64 + lane] = wt[2][(size_t)ks * 64];
    float bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bias[g] = p.bhh[(size_t)d * 3 * Hd + g * Hd + unit];
    float h[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) h[i] = 0.f;
    // buffer addressing with 32-bit byte offsets (host-checked: every tensor < 4 GiB): rows past the batch are out of range,
    // so their loads return 0 and their stores are dropped -- no per-row predicates, no 64-bit pointers per row in registers
    const size_t rows = (size_t)p.B * t;
    const __amdgpu_buffer_rsrc_t gis = make_srd(p.gi, rows * 6 * Hd * 4), hss = make_srd(p.hseq, rows * 2 * Hd * 4);
    const __amdgpu_buffer_rsrc_t svs = make_srd(p.saved, p.saved ? rows * 8 * Hd * 4 : 0);
    const int b0t = (bc * (2 * NREG) + 4 * hh) * t;     // (first batch row of this lane) * t; NREG = 8: 16-row chunks
    auto rowidx = [&](int i, int tt) { return b0t + ((i & 3) + 8 * (i >> 2)) * t + tt; };     // (b*t + tt) of register i
    __syncthreads();
    // NREG = 8: a batch of at most 16 rows (BASELINE config 4: 16 clips per GPU) only uses registers 0..7 of the accumulators
    // (rows 0..15): the gate math, loads and stores of the other eight are compiled out (their h rows stay zero)
    constexpr int nreg = NREG;
    float gin[3][16];
    auto load_gi = [&](int tt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i >= nreg) continue;
            const unsigned o = (unsigned)((rowidx(i, tt) * 6 * Hd + d * 3 * Hd + unit) * 4);
#pragma unroll
            for (int g = 0; g < 3; ++g) gin[g][i] = buf_load_f32(gis, o + (unsigned)(g * Hd * 4));
        }
    };
    load_gi(d == 0 ? 0 : t - 1);
    for (int s = 0; s < t; ++s) {
        const int tt = d == 0 ? s : t - 1 - s;
        f32x16 acc[3];                                       // start from b_hh: acc = W_h* h + b_h*
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[g][i] = bias[g];
        const T* hrow = hs + bl * HS + KR * hh;              // A operand: batch row bl, KR consecutive units of h per k-group
#pragma unroll 4
        for (int ks = 0; ks < KS; ++ks) {
            const frag_t af = *reinterpret_cast<const frag_t*>(hrow + ks * KSTEP);
            acc[0] = mfma(af, wt[0][(size_t)ks * 64], acc[0]);
            acc[1] = mfma(af, wt[1][(size_t)ks * 64], acc[1]);
            acc[2] = mfma(af, RESN ? wn[(w * KS + ks) * 64 + lane] : wt[2][(size_t)ks * 64], acc[2]);
        }
        // gates: register i <-> batch row (i&3) + 8*(i>>2) + 4*hh; stores issued as the values are produced
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i >= nreg) continue;
            const float rr = sigmoidf_(gin[0][i] + acc[0][i]);
            const float zz = sigmoidf_(gin[1][i] + acc[1][i]);
            const float ghn = acc[2][i];
            const float nn = tanhf_(fmaf(rr, ghn, gin[2][i]));
            h[i] = fmaf(zz, h[i] - nn, nn);                 // (1 - z) n + z h
            const int r0 = rowidx(i, tt);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h[i]), hss, (unsigned)((r0 * 2 * Hd + d * Hd + unit) * 4), 0, 0);
            const unsigned so = (unsigned)((r0 * 8 * Hd + d * 4 * Hd + unit) * 4);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rr), svs, so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, zz), svs, so + (unsigned)(Hd * 4), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, nn), svs, so + (unsigned)(2 * Hd * 4), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ghn), svs, so + (unsigned)(3 * Hd * 4), 0, 0);
        }
        if (s + 1 < t) load_gi(d == 0 ? s + 1 : t - 2 - s);
        __syncthreads();                                     // every wave has read hs for this step
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < nreg) hs[((i & 3) + 8 * (i >> 2) + 4 * hh) * HS + unit] = from_f<T>(h[i]);
        __syncthreads();
    }
}

// ---- forward recurrence, bf16 / Hd = 256: the whole recurrent matrix stays on the CU ----------------------------------
// Same orientation as above (lane = hidden unit).  Wave w keeps the r and z gate fragments of its 32 units in REGISTERS
// (32 x 1 KB), the n gate fragments sit in LDS (16 KB per wave): the MFMA loop of a step touches no global memory at all.
// The next step's r / z input projections are fetched straight into the accumulators (acc = b_hh + gi is the sigmoid's
// argument before the recurrent product is added); only gi_n needs registers of its own.
template <int NREG>
__global__ __launch_bounds__(512) void gru_seq_fwd_res_kernel(GruSeqParams p) {
    typedef bf16_t T;
    constexpr int Hd = 256, KS = 16, PAD = SeqLds<T>::PAD, HS = Hd + PAD, NW = Hd / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* hs = reinterpret_cast<T*>(smem);                               // [32][HS]
    bf16x8* wn = reinterpret_cast<bf16x8*>(hs + 32 * HS);             // [8 waves][KS][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int d = blockIdx.x & 1, bc = blockIdx.x >> 1;
    const int bl = lane & 31, hh = lane >> 5;
    const int unit = 32 * w + bl;
    const int t = p.t;
    for (int i = tid; i < 32 * HS; i += 512) hs[i] = (T)0.f;
    const bf16x8* __restrict__ wp = reinterpret_cast<const bf16x8*>(p.wpack) + (size_t)d * 3 * Hd * Hd / 8;
    constexpr int NZR = NREG <= 8 ? 16 : 12;    // z fragments in registers; the last KS - NZR stream from L2 every step (16 spilled)
    bf16x8 wr[KS], wz[NZR];
    const bf16x8* __restrict__ wzg = wp + ((size_t)(1 * NW + w) * KS) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        wr[ks] = wp[((size_t)(0 * NW + w) * KS + ks) * 64 + lane];
        if (ks < NZR) wz[ks] = wzg[(size_t)ks * 64];
        wn[(w * KS + ks) * 64 + lane] = wp[((size_t)(2 * NW + w) * KS + ks) * 64 + lane];
    }
    float bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) bias[g] = p.bhh[(size_t)d * 3 * Hd + g * Hd + unit];
    const size_t rows = (size_t)p.B * t;
    const __amdgpu_buffer_rsrc_t gis = make_srd(p.gi, rows * 6 * Hd * 4), hss = make_srd(p.hseq, rows * 2 * Hd * 4);
    const __amdgpu_buffer_rsrc_t svs = make_srd(p.saved, p.saved ? rows * 8 * Hd * 4 : 0);
    const int b0t = (bc * (2 * NREG) + 4 * hh) * t;
    auto rowidx = [&](int i, int tt) { return b0t + ((i & 3) + 8 * (i >> 2)) * t + tt; };
    float h[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) h[i] = 0.f;
    __syncthreads();
    f32x16 acc[3];
    float ginn[16];
    // the loads of the NEXT step's projections are issued value by value inside the gate loop, into the accumulator registers
    // the gate math has just released; b_hh is added when they are first used (top of the next step), so their latency hides
    // behind the rest of the gate math and the two barriers
    constexpr int nreg = NREG;                              // (see gru_seq_fwd_kernel)
    auto issue_inputs = [&](int i, int tt) {
        if (i >= nreg) { acc[0][i] = 0.f; acc[1][i] = 0.f; ginn[i] = 0.f; return; }
        const unsigned o = (unsigned)((rowidx(i, tt) * 6 * Hd + d * 3 * Hd + unit) * 4);
        acc[0][i] = buf_load_f32(gis, o);
        acc[1][i] = buf_load_f32(gis, o + (unsigned)(Hd * 4));
        ginn[i] = buf_load_f32(gis, o + (unsigned)(2 * Hd * 4));
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) issue_inputs(i, d == 0 ? 0 : t - 1);
    for (int s = 0; s < t; ++s) {
        const int tt = d == 0 ? s : t - 1 - s;
        const int tn = d == 0 ? s + 1 : t - 2 - s;          // next step's time index
        bf16x8 wzs[2];                                       // streamed z fragments: a two-deep ring, fetched six k-steps ahead
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] += bias[0]; acc[1][i] += bias[1]; acc[2][i] = bias[2]; }
        const T* hrow = hs + bl * HS + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (NZR < KS && ks == NZR - 6) wzs[0] = wzg[(size_t)(NZR < KS ? NZR : 0) * 64];
            if (NZR < KS && ks == NZR - 5) wzs[1] = wzg[(size_t)(NZR < KS ? NZR + 1 : 0) * 64];
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(hrow + ks * 16);
            acc[0] = mfma(af, wr[ks], acc[0]);
            acc[1] = mfma(af, ks < NZR ? wz[ks < NZR ? ks : 0] : wzs[(ks - NZR) & 1], acc[1]);
            acc[2] = mfma(af, wn[(w * KS + ks) * 64 + lane], acc[2]);
            if (ks >= NZR && ks + 2 < KS) wzs[(ks - NZR) & 1] = wzg[(size_t)(ks + 2) * 64];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i >= nreg) continue;
            const float rr = sigmoidf_(acc[0][i]);
            const float zz = sigmoidf_(acc[1][i]);
            const float ghn = acc[2][i];
            const float nn = tanhf_(fmaf(rr, ghn, ginn[i]));
            h[i] = fmaf(zz, h[i] - nn, nn);
            const int r0 = rowidx(i, tt);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h[i]), hss, (unsigned)((r0 * 2 * Hd + d * Hd + unit) * 4), 0, 0);
            const unsigned so = (unsigned)((r0 * 8 * Hd + d * 4 * Hd + unit) * 4);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rr), svs, so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, zz), svs, so + (unsigned)(Hd * 4), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, nn), svs, so + (unsigned)(2 * Hd * 4), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ghn), svs, so + (unsigned)(3 * Hd * 4), 0, 0);
            if (s + 1 < t) issue_inputs(i, tn);             // (wave-uniform)
        }
        __syncthreads();                                     // every wave has read hs for this step
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < nreg) hs[((i & 3) + 8 * (i >> 2) + 4 * hh) * HS + unit] = (T)h[i];
        __syncthreads();
    }
}

// ---- backward recurrence (BPTT) ------------------------------------------------------------------
// Same orientation as the forward kernels: D[batch][unit] = dgh[batch][gate unit j] . W_hh[j][unit], a lane is the hidden unit
// whose dh_prev it accumulates AND whose gate gradients it computes, its 16 registers are batch rows; all gate tensors move
// in full 128-byte lines.  The step's inputs (dh, r, z, n, W_hn h + b_hn, h_prev: 96 values per lane) are fetched one step
// ahead, behind the MFMA loop.  NL fragments of the wave's 3Hd/KSTEP recurrent-operator fragments stay in LDS for the whole
// sequence (bf16, Hd = 256: 13 of 48 -- what fits beside the dgh image), the rest streams from L2.
template <typename T, int NREG, int NL, int NR = 0>
__global__ __launch_bounds__(512) void gru_seq_bwd_kernel(GruSeqParams p) {
    constexpr int KR = EL<T>::KR, KSTEP = EL<T>::KSTEP, PAD = SeqLds<T>::PAD;
    typedef typename EL<T>::frag_t frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* dgs = reinterpret_cast<T*>(smem);                    // [DR][3Hd + PAD]: dgh of the current step (DR = 16 rows when NREG = 8)
    constexpr int DR = 2 * NREG;
    const int Hd = p.Hd, GS = 3 * Hd + PAD, KS = 3 * Hd / KSTEP;
    frag_t* wl = reinterpret_cast<frag_t*>(dgs + DR * GS);  // [waves][NL][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int d = blockIdx.x & 1, bc = blockIdx.x >> 1;
    const int bl = lane & 31, hh = lane >> 5;
    const int unit = 32 * w + bl;
    const int t = p.t;
    const frag_t* __restrict__ wt = reinterpret_cast<const frag_t*>(p.wpack) + (size_t)d * 3 * Hd * Hd / KR +
                                    ((size_t)w * KS) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < NL; ++ks) wl[(w * NL + ks) * 64 + lane] = wt[(size_t)ks * 64];
    frag_t wrg[NR > 0 ? NR : 1];                           // NR more fragments in registers
#pragma unroll
    for (int ks = 0; ks < NR; ++ks) wrg[ks] = wt[(size_t)(NL + ks) * 64];
    const size_t rows = (size_t)p.B * t;
    const __amdgpu_buffer_rsrc_t dhs = make_srd(p.dhseq, rows * 2 * Hd * 4), hqs = make_srd(p.hseq, rows * 2 * Hd * 4);
    const __amdgpu_buffer_rsrc_t svs = make_srd(p.saved, rows * 8 * Hd * 4);
    const __amdgpu_buffer_rsrc_t gis = make_srd(p.dgi, rows * 6 * Hd * 4), ghs = make_srd(p.dgh, rows * 6 * Hd * 4);
    const int b0t = (bc * (2 * NREG) + 4 * hh) * t;
    auto rowidx = [&](int i, int tt) { return b0t + ((i & 3) + 8 * (i >> 2)) * t + tt; };     // (b*t + tt) of register i
    float dhc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dhc[i] = 0.f;
    constexpr int nreg = NREG;                              // (see gru_seq_fwd_kernel; the dgh rows of the skipped registers stay zero)
    for (int i = tid; i < DR * GS; i += blockDim.x) dgs[i] = from_f<T>(0.f);
    float in_dh[16], in_r[16], in_z[16], in_n[16], in_g[16], in_hp[16];
    auto fetch = [&](int s) {          // inputs of step s (reverse of the forward order); rows past the batch read 0
        const int tt = d == 0 ? t - 1 - s : s;
        const int tp = d == 0 ? tt - 1 : tt + 1;             // where h_prev of this step lives
        const bool has_prev = tp >= 0 && tp < t;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i >= nreg) continue;
            const int r0 = rowidx(i, tt);
            in_dh[i] = buf_load_f32(dhs, (unsigned)((r0 * 2 * Hd + d * Hd + unit) * 4));
            const unsigned so = (unsigned)((r0 * 8 * Hd + d * 4 * Hd + unit) * 4);
            in_r[i] = buf_load_f32(svs, so);
            in_z[i] = buf_load_f32(svs, so + (unsigned)(Hd * 4));
            in_n[i] = buf_load_f32(svs, so + (unsigned)(2 * Hd * 4));
            in_g[i] = buf_load_f32(svs, so + (unsigned)(3 * Hd * 4));
            in_hp[i] = has_prev ? buf_load_f32(hqs, (unsigned)((rowidx(i, tp) * 2 * Hd + d * Hd + unit) * 4)) : 0.f;
        }
    };
    fetch(0);
    __syncthreads();
    for (int s = 0; s < t; ++s) {
        const int tt = d == 0 ? t - 1 - s : s;
        float dzk[16];      // dh * z: the direct path into dh_prev
        __syncthreads();                                     // previous step's MFMA reads of dgs are done
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i >= nreg) { dzk[i] = 0.f; continue; }
            const float rr = in_r[i], zz = in_z[i], nn = in_n[i], ghn = in_g[i];
            const float dh = in_dh[i] + dhc[i];
            const float dn_pre = dh * (1.f - zz) * (1.f - nn * nn);
            const float dz_pre = dh * (in_hp[i] - nn) * zz * (1.f - zz);
            const float dr_pre = dn_pre * ghn * rr * (1.f - rr);
            const float ghn_r = dn_pre * rr;
            dzk[i] = dh * zz;
            const int r0 = rowidx(i, tt);
            const unsigned go = (unsigned)((r0 * 6 * Hd + d * 3 * Hd + unit) * 4);
            // dgi = gradients of the (r, z, n) pre-activations; dgh = (r, z, n*r)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dr_pre), gis, go, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dz_pre), gis, go + (unsigned)(Hd * 4), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dn_pre), gis, go + (unsigned)(2 * Hd * 4), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dr_pre), ghs, go, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dz_pre), ghs, go + (unsigned)(Hd * 4), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ghn_r), ghs, go + (unsigned)(2 * Hd * 4), 0, 0);
            T* grow_w = dgs + ((i & 3) + 8 * (i >> 2) + 4 * hh) * GS + unit;
            grow_w[0] = from_f<T>(dr_pre);
            grow_w[Hd] = from_f<T>(dz_pre);
            grow_w[2 * Hd] = from_f<T>(ghn_r);
        }
        __syncthreads();
        if (s + 1 < t) fetch(s + 1);                         // flies behind the MFMA loop
        // dh_prev[b][unit k] = sum_j dgh[b][j] W_hh[j][k]
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const T* grow = dgs + (bl & (DR - 1)) * GS + KR * hh;     // A operand: batch row bl, KR consecutive gate units per k-group
                                                                  // (DR = 16: rows 16..31 of D are never used, their lanes re-read rows 0..15)
#pragma unroll
        for (int ks = 0; ks < NL; ++ks) {
            const frag_t af = *reinterpret_cast<const frag_t*>(grow + ks * KSTEP);
            acc = mfma(af, wl[(w * NL + ks) * 64 + lane], acc);
        }
#pragma unroll
        for (int ks = 0; ks < NR; ++ks) {
            const frag_t af = *reinterpret_cast<const frag_t*>(grow + (NL + ks) * KSTEP);
            acc = mfma(af, wrg[ks], acc);
        }
#pragma unroll 8
        for (int ks = NL + NR; ks < KS; ++ks) {
            const frag_t af = *reinterpret_cast<const frag_t*>(grow + ks * KSTEP);
            acc = mfma(af, wt[(size_t)ks * 64], acc);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) dhc[i] = dzk[i] + acc[i];
    }
}


// ---------------------------------------------------------------------------------------------
// Round 4: the 8-row chunks on v_mfma_f32_16x16x32_bf16 (bf16, Hd = 256).
// A 32x32x16 MFMA computes 32 batch rows of which an 8-row chunk uses 8; the 16x16x32 instruction (16 rows x 16 units x 32 k in
// 16 cycles) does the same products in half the matrix-pipe time (a step was 96 MFMAs of 32 cycles per SIMD = 1.5 of its 2.6 us).
// Orientation as before, D[batch][unit]: A = the h (or dgh) fragment, B = the packed operator, a lane is a hidden unit:
//   lane l: n = l & 15 (unit 16*nt + n of the wave's 32), q = l >> 4;  D register i <-> row m = 4q + i
// The chunk's 8 batch rows sit at m = 4*(b >> 1) + (b & 1), so EVERY lane holds two live rows (registers 0, 1 <-> b = 2q + i) of
// its two units: the gate math keeps all 64 lanes busy with 12 values each, as in the 32-wide form.
//   pack16 fwd: [gate][wave][nt][ks (8)][lane] bf16x8:  W_hh[gate*Hd + 32w + 16nt + n][32ks + 8q + e]
//   pack16 bwd: [wave][nt][ks (24)][lane] bf16x8:       W_hh[32ks + 8q + e][32w + 16nt + n]
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float gru_f32x4;
__device__ __forceinline__ gru_f32x4 mfma16(const bf16x8& a, const bf16x8& b, const gru_f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256) void gru_pack16_kernel(const float* __restrict__ whh, bf16_t* __restrict__ pf, bf16_t* __restrict__ pb) {
    constexpr int Hd = 256, NW = 8;
    const size_t total = (size_t)3 * Hd * Hd;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    {   // forward: idx = ((((g*NW + w)*2 + nt)*8 + ks)*64 + lane)*8 + e
        size_t t = i;
        const int e = t % 8; t /= 8;
        const int lane = t % 64; t /= 64;
        const int ks = t % 8; t /= 8;
        const int nt = t % 2; t /= 2;
        const int w = t % NW;
        const int g = (int)(t / NW);
        const int unit = 32 * w + 16 * nt + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + e;
        pf[i] = (bf16_t)whh[(size_t)(g * Hd + unit) * Hd + k];
    }
    {   // backward: idx = (((w*2 + nt)*24 + ks)*64 + lane)*8 + e
        size_t t = i;
        const int e = t % 8; t /= 8;
        const int lane = t % 64; t /= 64;
        const int ks = t % 24; t /= 24;
        const int nt = t % 2;
        const int w = (int)(t / 2);
        const int unit = 32 * w + 16 * nt + (lane & 15), j = 32 * ks + 8 * (lane >> 4) + e;
        pb[i] = (bf16_t)whh[(size_t)j * Hd + unit];
    }
}

// NLIVE live rows per lane: 2 = 8-row chunks (m = 4q + i, b = 2q + i), 1 = 4-row chunks (m = 4q, b = q): twice the workgroups, half the
// gate math (the step's other half besides the MFMAs) per workgroup
template <int NLIVE>
__global__ __launch_bounds__(512) void gru_seq_fwd16_kernel(GruSeqParams p) {
    typedef bf16_t T;
    constexpr int Hd = 256, KS = 8, PAD = SeqLds<T>::PAD, HS = Hd + PAD, NW = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* hs = reinterpret_cast<T*>(smem);                               // [16][HS]   rows m = 4*(b >> 1) + (b & 1) live
    bf16x8* wn = reinterpret_cast<bf16x8*>(hs + 16 * HS);             // [8 waves][2][KS][64 lanes]   n-gate fragments
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int d = blockIdx.x & 1, bc = blockIdx.x >> 1;
    const int n = lane & 15, q = lane >> 4;
    const int t = p.t;
    for (int i = tid; i < 16 * HS; i += 512) hs[i] = (T)0.f;
    const bf16x8* __restrict__ wp = reinterpret_cast<const bf16x8*>(p.wpack) + (size_t)d * 3 * Hd * Hd / 8;
    bf16x8 wr[2][KS], wz[2][KS];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            wr[nt][ks] = wp[((size_t)((0 * NW + w) * 2 + nt) * KS + ks) * 64 + lane];
            wz[nt][ks] = wp[((size_t)((1 * NW + w) * 2 + nt) * KS + ks) * 64 + lane];
            wn[((w * 2 + nt) * KS + ks) * 64 + lane] = wp[((size_t)((2 * NW + w) * 2 + nt) * KS + ks) * 64 + lane];
        }
    int unit[2];
    float bias[3][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        unit[nt] = 32 * w + 16 * nt + n;
#pragma unroll
        for (int g = 0; g < 3; ++g) bias[g][nt] = p.bhh[(size_t)d * 3 * Hd + g * Hd + unit[nt]];
    }
    const size_t rows = (size_t)p.B * t;
    const __amdgpu_buffer_rsrc_t gis = make_srd(p.gi, rows * 6 * Hd * 4), hss = make_srd(p.hseq, rows * 2 * Hd * 4);
    const __amdgpu_buffer_rsrc_t svs = make_srd(p.saved, p.saved ? rows * 8 * Hd * 4 : 0);
    // register i (0, 1) <-> batch row bc*8 + 2q + i (rows past the batch: out of range -> loads 0, stores dropped)
    auto rowidx = [&](int i, int tt) { return (bc * (4 * NLIVE) + NLIVE * q + i) * t + tt; };
    float h[2][NLIVE];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < NLIVE; ++i) h[nt][i] = 0.f;
    __syncthreads();
    gru_f32x4 acc[3][2];
    float ginn[2][NLIVE];
    // the next step's r / z projections go straight into the accumulator registers the gate math has just released
    auto issue_inputs = [&](int nt, int i, int tt) {
        const unsigned o = (unsigned)((rowidx(i, tt) * 6 * Hd + d * 3 * Hd + unit[nt]) * 4);
        acc[0][nt][i] = buf_load_f32(gis, o);
        acc[1][nt][i] = buf_load_f32(gis, o + (unsigned)(Hd * 4));
        ginn[nt][i] = buf_load_f32(gis, o + (unsigned)(2 * Hd * 4));
    };
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[0][nt][i] = 0.f; acc[1][nt][i] = 0.f; }
#pragma unroll
        for (int i = 0; i < NLIVE; ++i) issue_inputs(nt, i, d == 0 ? 0 : t - 1);
    }
    for (int s = 0; s < t; ++s) {
        const int tt = d == 0 ? s : t - 1 - s;
        const int tn = d == 0 ? s + 1 : t - 2 - s;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[0][nt][i] += bias[0][nt]; acc[1][nt][i] += bias[1][nt]; acc[2][nt][i] = bias[2][nt]; }
        }
        const T* hrow = hs + n * HS + 8 * q;                 // A operand: row m = n, 8 consecutive units of h per k-group
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(hrow + ks * 32);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                acc[0][nt] = mfma16(af, wr[nt][ks], acc[0][nt]);
                acc[1][nt] = mfma16(af, wz[nt][ks], acc[1][nt]);
                acc[2][nt] = mfma16(af, wn[((w * 2 + nt) * KS + ks) * 64 + lane], acc[2][nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < NLIVE; ++i) {
                const float rr = sigmoidf_(acc[0][nt][i]);
                const float zz = sigmoidf_(acc[1][nt][i]);
                const float ghn = acc[2][nt][i];
                const float nn = tanhf_(fmaf(rr, ghn, ginn[nt][i]));
                h[nt][i] = fmaf(zz, h[nt][i] - nn, nn);
                const int r0 = rowidx(i, tt);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h[nt][i]), hss, (unsigned)((r0 * 2 * Hd + d * Hd + unit[nt]) * 4), 0, 0);
                const unsigned so = (unsigned)((r0 * 8 * Hd + d * 4 * Hd + unit[nt]) * 4);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rr), svs, so, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, zz), svs, so + (unsigned)(Hd * 4), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, nn), svs, so + (unsigned)(2 * Hd * 4), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ghn), svs, so + (unsigned)(3 * Hd * 4), 0, 0);
                if (s + 1 < t) issue_inputs(nt, i, tn);     // (wave-uniform)
            }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)                       // the dead rows' accumulators restart from 0 (+ bias) every step
#pragma unroll
            for (int i = NLIVE; i < 4; ++i) { acc[0][nt][i] = 0.f; acc[1][nt][i] = 0.f; }
        __syncthreads();                                     // every wave has read hs for this step
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < NLIVE; ++i) hs[(4 * q + i) * HS + unit[nt]] = (T)h[nt][i];
        __syncthreads();
    }
}

// NL of a wave's 48 operator fragments in LDS, the other 48 - NL in registers (as gru_seq_bwd_kernel<bf16, 4, 17, 31>)
template <int NL, int NLIVE>
__global__ __launch_bounds__(512) void gru_seq_bwd16_kernel(GruSeqParams p) {
    typedef bf16_t T;
    constexpr int Hd = 256, PAD = SeqLds<T>::PAD, GS = 3 * Hd + PAD, KS = 24, NF = 2 * KS, NR = NF - NL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* dgs = reinterpret_cast<T*>(smem);                    // [16][GS]: dgh of the current step, rows m = 4*(b >> 1) + (b & 1) live
    bf16x8* wl = reinterpret_cast<bf16x8*>(dgs + 16 * GS);  // [waves][NL][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int d = blockIdx.x & 1, bc = blockIdx.x >> 1;
    const int n = lane & 15, q = lane >> 4;
    const int t = p.t;
    // fragment f = nt*KS + ks of this wave
    const bf16x8* __restrict__ wt = reinterpret_cast<const bf16x8*>(p.wpack) + (size_t)d * 3 * Hd * Hd / 8 + ((size_t)w * NF) * 64 + lane;
#pragma unroll
    for (int f = 0; f < NL; ++f) wl[(w * NL + f) * 64 + lane] = wt[(size_t)f * 64];
    bf16x8 wrg[NR];
#pragma unroll
    for (int f = 0; f < NR; ++f) wrg[f] = wt[(size_t)(NL + f) * 64];
    int unit[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) unit[nt] = 32 * w + 16 * nt + n;
    const size_t rows = (size_t)p.B * t;
    const __amdgpu_buffer_rsrc_t dhs = make_srd(p.dhseq, rows * 2 * Hd * 4), hqs = make_srd(p.hseq, rows * 2 * Hd * 4);
    const __amdgpu_buffer_rsrc_t svs = make_srd(p.saved, rows * 8 * Hd * 4);
    const __amdgpu_buffer_rsrc_t gis = make_srd(p.dgi, rows * 6 * Hd * 4), ghs = make_srd(p.dgh, rows * 6 * Hd * 4);
    auto rowidx = [&](int i, int tt) { return (bc * (4 * NLIVE) + NLIVE * q + i) * t + tt; };
    float dhc[2][NLIVE];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < NLIVE; ++i) dhc[nt][i] = 0.f;
    for (int i = tid; i < 16 * GS; i += 512) dgs[i] = (T)0.f;
    float in_dh[2][NLIVE], in_r[2][NLIVE], in_z[2][NLIVE], in_n[2][NLIVE], in_g[2][NLIVE], in_hp[2][NLIVE];
    auto fetch = [&](int s) {          // inputs of step s (reverse of the forward order); rows past the batch read 0
        const int tt = d == 0 ? t - 1 - s : s;
        const int tp = d == 0 ? tt - 1 : tt + 1;             // where h_prev of this step lives
        const bool has_prev = tp >= 0 && tp < t;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < NLIVE; ++i) {
                const int r0 = rowidx(i, tt);
                in_dh[nt][i] = buf_load_f32(dhs, (unsigned)((r0 * 2 * Hd + d * Hd + unit[nt]) * 4));
                const unsigned so = (unsigned)((r0 * 8 * Hd + d * 4 * Hd + unit[nt]) * 4);
                in_r[nt][i] = buf_load_f32(svs, so);
                in_z[nt][i] = buf_load_f32(svs, so + (unsigned)(Hd * 4));
                in_n[nt][i] = buf_load_f32(svs, so + (unsigned)(2 * Hd * 4));
                in_g[nt][i] = buf_load_f32(svs, so + (unsigned)(3 * Hd * 4));
                in_hp[nt][i] = has_prev ? buf_load_f32(hqs, (unsigned)((rowidx(i, tp) * 2 * Hd + d * Hd + unit[nt]) * 4)) : 0.f;
            }
    };
    fetch(0);
    __syncthreads();
    for (int s = 0; s < t; ++s) {
        const int tt = d == 0 ? t - 1 - s : s;
        float dzk[2][NLIVE];  // dh * z: the direct path into dh_prev
        __syncthreads();                                     // previous step's MFMA reads of dgs are done
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < NLIVE; ++i) {
                const float rr = in_r[nt][i], zz = in_z[nt][i], nn = in_n[nt][i], ghn = in_g[nt][i];
                const float dh = in_dh[nt][i] + dhc[nt][i];
                const float dn_pre = dh * (1.f - zz) * (1.f - nn * nn);
                const float dz_pre = dh * (in_hp[nt][i] - nn) * zz * (1.f - zz);
                const float dr_pre = dn_pre * ghn * rr * (1.f - rr);
                const float ghn_r = dn_pre * rr;
                dzk[nt][i] = dh * zz;
                const int r0 = rowidx(i, tt);
                const unsigned go = (unsigned)((r0 * 6 * Hd + d * 3 * Hd + unit[nt]) * 4);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dr_pre), gis, go, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dz_pre), gis, go + (unsigned)(Hd * 4), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dn_pre), gis, go + (unsigned)(2 * Hd * 4), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dr_pre), ghs, go, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dz_pre), ghs, go + (unsigned)(Hd * 4), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ghn_r), ghs, go + (unsigned)(2 * Hd * 4), 0, 0);
                T* grow_w = dgs + (4 * q + i) * GS + unit[nt];
                grow_w[0] = (T)dr_pre;
                grow_w[Hd] = (T)dz_pre;
                grow_w[2 * Hd] = (T)ghn_r;
            }
        __syncthreads();
        if (s + 1 < t) fetch(s + 1);                         // flies behind the MFMA loop
        // dh_prev[b][unit k] = sum_j dgh[b][j] W_hh[j][k]
        gru_f32x4 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[nt][e] = 0.f;
        const T* grow = dgs + n * GS + 8 * q;                // A operand: row m = n, 8 consecutive gate units per k-group
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(grow + ks * 32);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int f = nt * KS + ks;
                acc[nt] = mfma16(af, f < NL ? wl[(w * NL + (f < NL ? f : 0)) * 64 + lane] : wrg[f >= NL ? f - NL : 0], acc[nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < NLIVE; ++i) dhc[nt][i] = dzk[nt][i] + acc[nt][i];
    }
}

// ---- 2-row chunks with split gate ownership (round 4) ------------------------------------------------------------------------------
// In the 4-row form every lane evaluates the gates of TWO hidden units (nt = 0, 1) of one batch row, and that gate math (six quarter-rate
// transcendentals per value) is the part of a step that does not shrink with the rows.  With two rows per workgroup the rows sit at D
// rows m = 0 and 4 (lanes q = 0, 1) and the lanes q = 2, 3 would idle: v_permlane32_swap hands them the nt = 1 accumulators of lanes
// q = 0, 1 (lane l <-> l + 32, one instruction per gate: the result register holds the nt = 0 value in the lower and the nt = 1 value in
// the upper half-wave), so every lane evaluates ONE unit of one row and keeps that unit's state, inputs and stores to itself.
__device__ __forceinline__ float gru_own(float v_nt0, float v_nt1) {       // lanes 0..31: their nt = 0 value; lanes 32..63: the nt = 1 value of lane - 32
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v_nt0), __builtin_bit_cast(unsigned, v_nt1), false, false);
    return __builtin_bit_cast(float, r[0]);
}
// DB (round 6, SED_GRU_1BAR=1; measured neutral, off): ONE barrier per step.  Only D rows m = 0 and 4 are live, so the h image is three rows (row 0, row 4, one shared
// zero row for the other fourteen lanes' A-operand reads) and fits twice in the space of the 16-row image: step s reads buffer s & 1 and
// writes the new state into the other one -- a wave still reading buffer s & 1 is never overtaken by a write to it, because that write
// belongs to step s + 1 and sits behind the barrier that ends step s.  (The two-barrier form: write-after-read and read-after-write on
// one image.)
template <bool DB = false>
__global__ __launch_bounds__(512) void gru_seq_fwd16h_kernel(GruSeqParams p) {
    typedef bf16_t T;
    constexpr int Hd = 256, KS = 8, PAD = SeqLds<T>::PAD, HS = Hd + PAD, NW = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* hs = reinterpret_cast<T*>(smem);                               // [16][HS]   rows m = 0, 4 live  (DB: [2][3][HS])
    bf16x8* wn = reinterpret_cast<bf16x8*>(hs + 16 * HS);             // [8 waves][2][KS][64 lanes]   n-gate fragments
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int d = blockIdx.x & 1, bc = blockIdx.x >> 1;
    const int n = lane & 15, q = lane >> 4;
    const int t = p.t;
    for (int i = tid; i < 16 * HS; i += 512) hs[i] = (T)0.f;
    const bf16x8* __restrict__ wp = reinterpret_cast<const bf16x8*>(p.wpack) + (size_t)d * 3 * Hd * Hd / 8;
    bf16x8 wr[2][KS], wz[2][KS];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            wr[nt][ks] = wp[((size_t)((0 * NW + w) * 2 + nt) * KS + ks) * 64 + lane];
            wz[nt][ks] = wp[((size_t)((1 * NW + w) * 2 + nt) * KS + ks) * 64 + lane];
            wn[((w * 2 + nt) * KS + ks) * 64 + lane] = wp[((size_t)((2 * NW + w) * 2 + nt) * KS + ks) * 64 + lane];
        }
    const int own = q >> 1, brow = bc * 2 + (q & 1);                   // the unit half (nt) and batch row this lane evaluates
    const int uo = 32 * w + 16 * own + n;
    float bias[3][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 3; ++g) bias[g][nt] = p.bhh[(size_t)d * 3 * Hd + g * Hd + 32 * w + 16 * nt + n];
    const size_t rows = (size_t)p.B * t;
    const __amdgpu_buffer_rsrc_t gis = make_srd(p.gi, rows * 6 * Hd * 4), hss = make_srd(p.hseq, rows * 2 * Hd * 4);
    const __amdgpu_buffer_rsrc_t svs = make_srd(p.saved, p.saved ? rows * 8 * Hd * 4 : 0);
    // (rows past the batch: out of range -> loads 0, stores dropped)
    float h = 0.f, gir, giz, gin;
    auto issue_inputs = [&](int tt) {
        const unsigned o = (unsigned)(((brow * t + tt) * 6 * Hd + d * 3 * Hd + uo) * 4);
        gir = buf_load_f32(gis, o);
        giz = buf_load_f32(gis, o + (unsigned)(Hd * 4));
        gin = buf_load_f32(gis, o + (unsigned)(2 * Hd * 4));
    };
    issue_inputs(d == 0 ? 0 : t - 1);
    __syncthreads();
    for (int s = 0; s < t; ++s) {
        const int tt = d == 0 ? s : t - 1 - s;
        const int tn = d == 0 ? s + 1 : t - 2 - s;
        gru_f32x4 acc[3][2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc[0][nt][i] = bias[0][nt]; acc[1][nt][i] = bias[1][nt]; acc[2][nt][i] = bias[2][nt]; }
        // A operand: row m = n, 8 consecutive units of h per k-group
        const T* hrow = DB ? hs + ((s & 1) * 3 + (n == 0 ? 0 : n == 4 ? 1 : 2)) * HS + 8 * q : hs + n * HS + 8 * q;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(hrow + ks * 32);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                acc[0][nt] = mfma16(af, wr[nt][ks], acc[0][nt]);
                acc[1][nt] = mfma16(af, wz[nt][ks], acc[1][nt]);
                acc[2][nt] = mfma16(af, wn[((w * 2 + nt) * KS + ks) * 64 + lane], acc[2][nt]);
            }
        }
        // D row 4q, register 0: lanes q = 0, 1 hold rows 0 / 1 of the chunk for both unit halves; the upper half-wave takes nt = 1
        // ((b + h W^T) + gi: the 4-row form adds gi first -- same values to fp32 rounding, not the same bits)
        const float rr = sigmoidf_(gir + gru_own(acc[0][0][0], acc[0][1][0]));
        const float zz = sigmoidf_(giz + gru_own(acc[1][0][0], acc[1][1][0]));
        const float ghn = gru_own(acc[2][0][0], acc[2][1][0]);
        const float nn = tanhf_(fmaf(rr, ghn, gin));
        h = fmaf(zz, h - nn, nn);
        const int r0 = brow * t + tt;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h), hss, (unsigned)((r0 * 2 * Hd + d * Hd + uo) * 4), 0, 0);
        const unsigned so = (unsigned)((r0 * 8 * Hd + d * 4 * Hd + uo) * 4);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rr), svs, so, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, zz), svs, so + (unsigned)(Hd * 4), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, nn), svs, so + (unsigned)(2 * Hd * 4), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ghn), svs, so + (unsigned)(3 * Hd * 4), 0, 0);
        if (s + 1 < t) issue_inputs(tn);                     // (wave-uniform)
        if (DB) {
            hs[(((s + 1) & 1) * 3 + (q & 1)) * HS + uo] = (T)h;
            __syncthreads();
        } else {
            __syncthreads();                                 // every wave has read hs for this step
            hs[(4 * (q & 1)) * HS + uo] = (T)h;
            __syncthreads();
        }
    }
}

template <int NL, bool DB = false>
__global__ __launch_bounds__(512) void gru_seq_bwd16h_kernel(GruSeqParams p) {
    typedef bf16_t T;
    constexpr int Hd = 256, PAD = SeqLds<T>::PAD, GS = 3 * Hd + PAD, KS = 24, NF = 2 * KS, NR = NF - NL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* dgs = reinterpret_cast<T*>(smem);                    // [16][GS]: dgh of the current step, rows m = 0, 4 live  (DB: [2][3][GS], one barrier per step)
    bf16x8* wl = reinterpret_cast<bf16x8*>(dgs + 16 * GS);  // [waves][NL][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int d = blockIdx.x & 1, bc = blockIdx.x >> 1;
    const int n = lane & 15, q = lane >> 4;
    const int t = p.t;
    const bf16x8* __restrict__ wt = reinterpret_cast<const bf16x8*>(p.wpack) + (size_t)d * 3 * Hd * Hd / 8 + ((size_t)w * NF) * 64 + lane;
#pragma unroll
    for (int f = 0; f < NL; ++f) wl[(w * NL + f) * 64 + lane] = wt[(size_t)f * 64];
    bf16x8 wrg[NR];
#pragma unroll
    for (int f = 0; f < NR; ++f) wrg[f] = wt[(size_t)(NL + f) * 64];
    const int own = q >> 1, brow = bc * 2 + (q & 1);
    const int uo = 32 * w + 16 * own + n;
    const size_t rows = (size_t)p.B * t;
    const __amdgpu_buffer_rsrc_t dhs = make_srd(p.dhseq, rows * 2 * Hd * 4), hqs = make_srd(p.hseq, rows * 2 * Hd * 4);
    const __amdgpu_buffer_rsrc_t svs = make_srd(p.saved, rows * 8 * Hd * 4);
    const __amdgpu_buffer_rsrc_t gis = make_srd(p.dgi, rows * 6 * Hd * 4), ghs = make_srd(p.dgh, rows * 6 * Hd * 4);
    float dhc = 0.f;
    for (int i = tid; i < 16 * GS; i += 512) dgs[i] = (T)0.f;
    float in_dh, in_r, in_z, in_n, in_g, in_hp;
    auto fetch = [&](int s) {          // inputs of step s (reverse of the forward order); rows past the batch read 0
        const int tt = d == 0 ? t - 1 - s : s;
        const int tp = d == 0 ? tt - 1 : tt + 1;             // where h_prev of this step lives
        const bool has_prev = tp >= 0 && tp < t;
        const int r0 = brow * t + tt;
        in_dh = buf_load_f32(dhs, (unsigned)((r0 * 2 * Hd + d * Hd + uo) * 4));
        const unsigned so = (unsigned)((r0 * 8 * Hd + d * 4 * Hd + uo) * 4);
        in_r = buf_load_f32(svs, so);
        in_z = buf_load_f32(svs, so + (unsigned)(Hd * 4));
        in_n = buf_load_f32(svs, so + (unsigned)(2 * Hd * 4));
        in_g = buf_load_f32(svs, so + (unsigned)(3 * Hd * 4));
        in_hp = has_prev ? buf_load_f32(hqs, (unsigned)(((brow * t + tp) * 2 * Hd + d * Hd + uo) * 4)) : 0.f;
    };
    fetch(0);
    __syncthreads();
    for (int s = 0; s < t; ++s) {
        const int tt = d == 0 ? t - 1 - s : s;
        if (!DB) __syncthreads();                            // previous step's MFMA reads of dgs are done
        const float rr = in_r, zz = in_z, nn = in_n, ghn = in_g;
        const float dh = in_dh + dhc;
        const float dn_pre = dh * (1.f - zz) * (1.f - nn * nn);
        const float dz_pre = dh * (in_hp - nn) * zz * (1.f - zz);
        const float dr_pre = dn_pre * ghn * rr * (1.f - rr);
        const float ghn_r = dn_pre * rr;
        const float dzk = dh * zz;                           // dh * z: the direct path into dh_prev
        {
            const int r0 = brow * t + tt;
            const unsigned go = (unsigned)((r0 * 6 * Hd + d * 3 * Hd + uo) * 4);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dr_pre), gis, go, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dz_pre), gis, go + (unsigned)(Hd * 4), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dn_pre), gis, go + (unsigned)(2 * Hd * 4), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dr_pre), ghs, go, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dz_pre), ghs, go + (unsigned)(Hd * 4), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ghn_r), ghs, go + (unsigned)(2 * Hd * 4), 0, 0);
            T* grow_w = DB ? dgs + ((s & 1) * 3 + (q & 1)) * GS + uo : dgs + (4 * (q & 1)) * GS + uo;
            grow_w[0] = (T)dr_pre;
            grow_w[Hd] = (T)dz_pre;
            grow_w[2 * Hd] = (T)ghn_r;
        }
        __syncthreads();
        if (s + 1 < t) fetch(s + 1);                         // flies behind the MFMA loop
        // dh_prev[b][unit k] = sum_j dgh[b][j] W_hh[j][k]
        gru_f32x4 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[nt][e] = 0.f;
        // A operand: row m = n, 8 consecutive gate units per k-group
        const T* grow = DB ? dgs + ((s & 1) * 3 + (n == 0 ? 0 : n == 4 ? 1 : 2)) * GS + 8 * q : dgs + n * GS + 8 * q;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 af = *reinterpret_cast<const bf16x8*>(grow + ks * 32);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int f = nt * KS + ks;
                acc[nt] = mfma16(af, f < NL ? wl[(w * NL + (f < NL ? f : 0)) * 64 + lane] : wrg[f >= NL ? f - NL : 0], acc[nt]);
            }
        }
        dhc = dzk + gru_own(acc[0][0], acc[1][0]);
    }
}

// which form the 8-row chunks of the bf16 / Hd = 256 recurrence take (pack layout and kernels must agree): SED_GRU_MFMA16=0 keeps
// the 32x32x16 kernels of round 3
// batch rows per workgroup of the 16x16x32 form (SED_GRU16_ROWS = 8 | 4 | 2, default 2: gru_seq_*16h_kernel above, 0.76 / 0.98 ms).  Measured (round 4, B = 32 / t = 750, forward / backward):
// 32x32x16 8-row chunks 1.91 / 2.00 ms, 16x16x32 8-row chunks 1.60 / 1.80 ms, 16x16x32 4-row chunks 1.06 / 1.12 ms -- once the matrix
// work is halved the gate math (six quarter-rate transcendentals per value) is the other half of a step, and it halves with the rows.
static int gru16_rows() {
    if (const char* e = sed_getenv("SED_GRU16_ROWS")) return atoi(e) == 8 ? 8 : atoi(e) == 4 ? 4 : 2;
    return 2;
}
static bool gru_use_mfma16(int dtype, int Hd) {
    if (!(dtype == SED_BF16 && Hd == 256)) return false;
    if (const char* e = sed_getenv("SED_GRU_MFMA16")) if (e[0] == '0') return false;
    if (sed_getenv("SED_GRU_RESIDENT")) return false;                                  // (A/B forms of round 3)
    if (const char* e = sed_getenv("SED_GRU_ROWS")) if (atoi(e) != 8) return false;
    return true;
}

extern "C" size_t sed_gru_pack_elems(int Hd) { return (size_t)2 * 3 * Hd * Hd; }   // per operator: both directions

extern "C" int sed_gru_pack_weights(int dtype, const float* whh_fwd, const float* whh_rev, void* pack_fwd, void* pack_bwd,
                                    int Hd, void* stream) {
    SED_REQUIRE(Hd >= 32 && Hd <= 256 && Hd % 32 == 0, "hidden size must be a multiple of 32 in [32, 256]");
    hipStream_t st = (hipStream_t)stream;
    const size_t per = (size_t)3 * Hd * Hd;
    const unsigned grid = (unsigned)cdivz(per, 256);
    for (int d = 0; d < 2; ++d) {
        const float* w = d == 0 ? whh_fwd : whh_rev;
        if (gru_use_mfma16(dtype, Hd))
            gru_pack16_kernel<<<grid, 256, 0, st>>>(w, (bf16_t*)pack_fwd + d * per, (bf16_t*)pack_bwd + d * per);
        else if (dtype == SED_BF16)
            gru_pack_kernel<bf16_t><<<grid, 256, 0, st>>>(w, (bf16_t*)pack_fwd + d * per, (bf16_t*)pack_bwd + d * per, Hd);
        else if (dtype == SED_F32)
            gru_pack_kernel<float><<<grid, 256, 0, st>>>(w, (float*)pack_fwd + d * per, (float*)pack_bwd + d * per, Hd);
        else
            SED_REQUIRE(false, "bad dtype");
        SED_LAUNCH_CHECK();
    }
    return 0;
}

template <typename T, typename K>
static int set_lds(K kernel, size_t lds) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) { sed_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return 3; }
    }
    return 0;
}

extern "C" int sed_gru_seq_fwd(int dtype, const float* gi, const float* bhh, const void* pack_fwd, float* hseq, float* saved,
                               int B, int t, int Hd, void* stream) {
    SED_REQUIRE(B > 0 && t > 0, "bad sizes");
    SED_REQUIRE(Hd >= 32 && Hd <= 256 && Hd % 32 == 0, "hidden size must be a multiple of 32 in [32, 256]");
    SED_REQUIRE(gi && bhh && pack_fwd && hseq, "null argument");
    SED_REQUIRE((double)B * t * 8 * Hd * 4 < 4294967296.0, "B*t*8*Hd floats must stay below 4 GiB (32-bit buffer offsets)");
    GruSeqParams p{};
    p.gi = gi; p.bhh = bhh; p.wpack = pack_fwd; p.hseq = hseq; p.saved = saved; p.B = B; p.t = t; p.Hd = Hd;
    // 8-row chunks (the 4-register kernels) for every batch size: four times the workgroups of the 32-row form, each with a quarter
    // of the gate math, loads and stores per step and the whole recurrent matrix resident -- B = 32 runs on eight CUs instead of two
    const char* rows_env = sed_getenv("SED_GRU_ROWS");
    int crows = 8;                                                    // rows per chunk (SED_GRU_ROWS = 8 / 16 / 32 overrides; measured at
                                                                      // B = 32: 32 rows 4.65 / 6.5 ms, 16 rows 2.5 / 3.25 ms, 8 rows 1.9 / 2.0 ms)
    if (rows_env) crows = atoi(rows_env) == 32 ? 32 : atoi(rows_env) == 8 ? 8 : 16;
    const bool half = crows == 16, quarter = crows == 8;
    int grid = 2 * cdiv(B, crows);
    const int threads = 64 * (Hd / 32);
    hipStream_t st = (hipStream_t)stream;
    const char* res_env = sed_getenv("SED_GRU_RESIDENT");
#define SED_GRU_FWD(KERNEL, THREADS)                                          \
    do {                                                                      \
        if (int rc = set_lds<bf16_t>(&KERNEL, lds)) return rc;                \
        KERNEL<<<grid, THREADS, lds, st>>>(p);                                \
    } while (0)
    if (gru_use_mfma16(dtype, Hd)) {           // 8-row chunks on the 16x16x32 instruction, recurrent matrix resident (r, z registers; n LDS)
        const size_t lds = (size_t)16 * (Hd + SeqLds<bf16_t>::PAD) * sizeof(bf16_t) + (size_t)8 * 16 * 64 * 16;
        // SED_GRU_1BAR=1: one barrier per step (double-buffered three-row state image).  Measured NEUTRAL (4.4358 / 4.4449 against 4.4382 /
        // 4.4365 ms per CRNN step, profiles/r06_o_ab_gru_one_barrier.txt): the step is the MFMA chain + the gate math, not its barriers.
        const char* b1 = sed_getenv("SED_GRU_1BAR");
        if (gru16_rows() == 2) {
            grid = 2 * cdiv(B, 2);
            if (b1 && b1[0] == '1') SED_GRU_FWD(gru_seq_fwd16h_kernel<true>, 512);
            else SED_GRU_FWD(gru_seq_fwd16h_kernel<false>, 512);
        }
        else if (gru16_rows() == 4) { grid = 2 * cdiv(B, 4); SED_GRU_FWD(gru_seq_fwd16_kernel<1>, 512); }
        else SED_GRU_FWD(gru_seq_fwd16_kernel<2>, 512);
    } else if (dtype == SED_BF16 && Hd == 256 && !(res_env && res_env[0] == '0')) {       // recurrent matrix resident on the CU
        const size_t lds = (size_t)32 * (Hd + SeqLds<bf16_t>::PAD) * sizeof(bf16_t) + (size_t)8 * 16 * 64 * 16;
        if (res_env && res_env[0] == '1') {      // (n gate in LDS, r / z streamed: the intermediate form, kept for A/B runs)
            if (quarter) SED_GRU_FWD((gru_seq_fwd_kernel<bf16_t, true, 4>), threads);
            else if (half) SED_GRU_FWD((gru_seq_fwd_kernel<bf16_t, true, 8>), threads);
            else SED_GRU_FWD((gru_seq_fwd_kernel<bf16_t, true, 16>), threads);
        } else {
            if (quarter) SED_GRU_FWD(gru_seq_fwd_res_kernel<4>, 512);
            else if (half) SED_GRU_FWD(gru_seq_fwd_res_kernel<8>, 512);
            else SED_GRU_FWD(gru_seq_fwd_res_kernel<16>, 512);
        }
    } else if (dtype == SED_BF16) {
        const size_t lds = (size_t)32 * (Hd + SeqLds<bf16_t>::PAD) * sizeof(bf16_t);
        if (quarter) SED_GRU_FWD((gru_seq_fwd_kernel<bf16_t, false, 4>), threads);
        else if (half) SED_GRU_FWD((gru_seq_fwd_kernel<bf16_t, false, 8>), threads);
        else SED_GRU_FWD((gru_seq_fwd_kernel<bf16_t, false, 16>), threads);
    } else if (dtype == SED_F32) {
        const size_t lds = (size_t)32 * (Hd + SeqLds<float>::PAD) * sizeof(float);
        if (quarter) SED_GRU_FWD((gru_seq_fwd_kernel<float, false, 4>), threads);
        else if (half) SED_GRU_FWD((gru_seq_fwd_kernel<float, false, 8>), threads);
        else SED_GRU_FWD((gru_seq_fwd_kernel<float, false, 16>), threads);
#undef SED_GRU_FWD
    } else {
        SED_REQUIRE(false, "bad dtype");
    }
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_gru_seq_bwd(int dtype, const float* dhseq, const float* hseq, const float* saved, const void* pack_bwd,
                               float* dgi, float* dgh, int B, int t, int Hd, void* stream) {
    SED_REQUIRE(B > 0 && t > 0, "bad sizes");
    SED_REQUIRE(Hd >= 32 && Hd <= 256 && Hd % 32 == 0, "hidden size must be a multiple of 32 in [32, 256]");
    SED_REQUIRE(dhseq && hseq && saved && pack_bwd && dgi && dgh, "null argument");
    SED_REQUIRE((double)B * t * 8 * Hd * 4 < 4294967296.0, "B*t*8*Hd floats must stay below 4 GiB (32-bit buffer offsets)");
    GruSeqParams p{};
    p.dhseq = dhseq; p.hseq = const_cast<float*>(hseq); p.saved = const_cast<float*>(saved); p.wpack = pack_bwd;
    p.dgi = dgi; p.dgh = dgh; p.B = B; p.t = t; p.Hd = Hd;
    // 8-row chunks (the 4-register kernels) for every batch size: four times the workgroups of the 32-row form, each with a quarter
    // of the gate math, loads and stores per step and the whole recurrent matrix resident -- B = 32 runs on eight CUs instead of two
    const char* rows_env = sed_getenv("SED_GRU_ROWS");
    int crows = 8;                                                    // rows per chunk (SED_GRU_ROWS = 8 / 16 / 32 overrides; measured at
                                                                      // B = 32: 32 rows 4.65 / 6.5 ms, 16 rows 2.5 / 3.25 ms, 8 rows 1.9 / 2.0 ms)
    if (rows_env) crows = atoi(rows_env) == 32 ? 32 : atoi(rows_env) == 8 ? 8 : 16;
    const bool half = crows == 16, quarter = crows == 8;
    int grid = 2 * cdiv(B, crows);
    const int threads = 64 * (Hd / 32);
    hipStream_t st = (hipStream_t)stream;
    const char* res_env = sed_getenv("SED_GRU_RESIDENT");
#define SED_GRU_BWD(KERNEL)                                                   \
    do {                                                                      \
        if (int rc = set_lds<bf16_t>(&KERNEL, lds)) return rc;                \
        KERNEL<<<grid, threads, lds, st>>>(p);                                \
    } while (0)
    if (gru_use_mfma16(dtype, Hd)) {           // 16x16x32 form: 16 of a wave's 48 operator fragments in LDS (beside the 16-row dgh image), 32 in registers
        const size_t lds = (size_t)16 * (3 * Hd + SeqLds<bf16_t>::PAD) * sizeof(bf16_t) + (size_t)8 * 16 * 64 * 16;
        const char* b1 = sed_getenv("SED_GRU_1BAR");
        if (gru16_rows() == 2) {
            grid = 2 * cdiv(B, 2);
            if (b1 && b1[0] == '1') SED_GRU_BWD((gru_seq_bwd16h_kernel<16, true>));
            else SED_GRU_BWD((gru_seq_bwd16h_kernel<16, false>));
        }
        else if (gru16_rows() == 4) { grid = 2 * cdiv(B, 4); SED_GRU_BWD((gru_seq_bwd16_kernel<16, 1>)); }
        else SED_GRU_BWD((gru_seq_bwd16_kernel<16, 2>));
    } else if (dtype == SED_BF16 && Hd == 256 && !(res_env && res_env[0] == '0')) {       // 13 of a wave's 48 operator fragments in LDS
        const size_t lds = (size_t)crows * (3 * Hd + SeqLds<bf16_t>::PAD) * sizeof(bf16_t) +
                           (size_t)8 * (quarter ? 17 : half ? 16 : 13) * 64 * 16;
        if (res_env && res_env[0] == '1') {
            if (quarter) SED_GRU_BWD((gru_seq_bwd_kernel<bf16_t, 4, 17>));
            else if (half) SED_GRU_BWD((gru_seq_bwd_kernel<bf16_t, 8, 16>));
            else SED_GRU_BWD((gru_seq_bwd_kernel<bf16_t, 16, 13>));
        } else {                                 // + fragments in registers: 23 (32-row chunks), 44 (16 rows), all 48 (8 rows) never leave the CU
            if (quarter) SED_GRU_BWD((gru_seq_bwd_kernel<bf16_t, 4, 17, 31>));
            else if (half) SED_GRU_BWD((gru_seq_bwd_kernel<bf16_t, 8, 16, 28>));
            else SED_GRU_BWD((gru_seq_bwd_kernel<bf16_t, 16, 13, 10>));
        }
    } else if (dtype == SED_BF16) {
        const size_t lds = (size_t)crows * (3 * Hd + SeqLds<bf16_t>::PAD) * sizeof(bf16_t);
        if (quarter) SED_GRU_BWD((gru_seq_bwd_kernel<bf16_t, 4, 0>));
        else if (half) SED_GRU_BWD((gru_seq_bwd_kernel<bf16_t, 8, 0>));
        else SED_GRU_BWD((gru_seq_bwd_kernel<bf16_t, 16, 0>));
    } else if (dtype == SED_F32) {
        const size_t lds = (size_t)crows * (3 * Hd + SeqLds<float>::PAD) * sizeof(float);
        SED_REQUIRE(lds <= 160 * 1024, "hidden size too large for the fp32 recurrence");
        if (quarter) SED_GRU_BWD((gru_seq_bwd_kernel<float, 4, 0>));
        else if (half) SED_GRU_BWD((gru_seq_bwd_kernel<float, 8, 0>));
        else SED_GRU_BWD((gru_seq_bwd_kernel<float, 16, 0>));
#undef SED_GRU_BWD
    } else {
        SED_REQUIRE(false, "bad dtype");
    }
    SED_LAUNCH_CHECK();
    return 0;
}
