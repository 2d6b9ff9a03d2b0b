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
#include "common.h"

template <typename T> struct EL;
template <> struct EL<float> {
    static constexpr int KR = 1;      // consecutive k per lane in a fragment
    static constexpr int KSTEP = 2;   // k per MFMA
    typedef float frag_t;
};
template <> struct EL<bf16_t> {
    static constexpr int KR = 8;
    static constexpr int KSTEP = 16;
    typedef bf16x8 frag_t;
};

// element-index XOR applied inside the 32-channel vector of LDS pixel column `col`
template <typename T> __device__ __forceinline__ int swz(int col);
template <> __device__ __forceinline__ int swz<bf16_t>(int col) { return ((col >> 2) & 3) << 3; }
template <> __device__ __forceinline__ int swz<float>(int col) { return col & 31; }

__device__ __forceinline__ f32x16 mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma(const float& a, const float& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

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

// =================================================================================================
// generic implicit-GEMM conv (forward and data gradient)
// =================================================================================================
struct ConvParams {
    const void* x;
    const float* pro_scale;
    const float* pro_shift;
    const void* wpack;
    void* z;
    const void* zref;
    const float* epi_scale;
    const float* epi_shift;
    const float* epi_mean;
    const float* epi_invstd;
    float* partial;
    int B, H, Cinp, Coutp;
    int tilesPerImg, totalTiles, tpb, nparts;
    int pro, epi;
};

template <typename T, int W, int BM, int BN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p) {
    typedef typename EL<T>::frag_t frag_t;
    constexpr int KR = EL<T>::KR, KSTEP = EL<T>::KSTEP;
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr int ROWS = TH + 2;
    constexpr int XS = ROWS * WP * 32;   // elements
    constexpr int WS = 9 * 32 * BN;      // elements
    constexpr int MT = BM / 128;         // 32-pixel tiles per wave (4 waves along M)
    constexpr int NT = BN / 32;
    static_assert(BM % 128 == 0 && BM % W == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = reinterpret_cast<T*>(smem);
    T* ws = xs + XS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    const int pro = p.pro, epi = p.epi;

    // per-lane pixel coordinates of its column in each M tile
    int prow[MT], pcol[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int q = (wave * MT + mt) * 32 + r;
        prow[mt] = q / W;
        pcol[mt] = q % W;
    }

    float S[NT][16], Q[NT][16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) { S[nt][i] = 0.f; Q[nt][i] = 0.f; }

    auto stage_w = [&](int kc) {
        constexpr int ROWLEN = BN * KR;          // contiguous elements per (tap,kq) row
        constexpr int ITEMS_PER_ROW = ROWLEN / 8;
        constexpr int NROWS = 9 * 32 / KR;
        for (int it = tid; it < NROWS * ITEMS_PER_ROW; it += 256) {
            const int rowi = it / ITEMS_PER_ROW, off = (it % ITEMS_PER_ROW) * 8;
            const T* src = wg + ((size_t)(kc * NROWS + rowi) * Coutp + n0) * KR + off;
            T* dst = ws + rowi * ROWLEN + off;
            if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<bf16x8*>(dst) = *reinterpret_cast<const bf16x8*>(src);
            } else {
                *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(src);
                *reinterpret_cast<f32x4*>(dst + 4) = *reinterpret_cast<const f32x4*>(src + 4);
            }
        }
    };

    const int t_begin = bx * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    if (nchunks == 1 && t_begin < t_end) stage_w(0);   // weights stay resident for the whole strip

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int b = tile / p.tilesPerImg;
        const int h0 = (tile - b * p.tilesPerImg) * TH;

        f32x16 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;

        for (int kc = 0; kc < nchunks; ++kc) {
            __syncthreads();   // previous readers of xs/ws are done
            // ---- stage the activation halo tile (with the fused BN+ReLU prologue) -------------
            {
                const int cq = tid & 3;
                float sc[8], sh[8];
                if (pro == SED_PRO_BNRELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        sc[e] = p.pro_scale[kc * 32 + cq * 8 + e];
                        sh[e] = p.pro_shift[kc * 32 + cq * 8 + e];
                    }
                }
                constexpr int ITEMS = ROWS * (W + 2) * 4;
                for (int it = tid; it < ITEMS; it += 256) {
                    const int pix = it >> 2;
                    const int rowi = pix / (W + 2), coli = pix - rowi * (W + 2);
                    const int h = h0 - 1 + rowi, w = coli - 1;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = 0.f;
                    if (h >= 0 && h < H && w >= 0 && w < W) {
                        load8<T>(xg + (((size_t)b * H + h) * W + w) * Cinp + kc * 32 + cq * 8, v);
                        if (pro == SED_PRO_BNRELU) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = fmaxf(0.f, fmaf(v[e], sc[e], sh[e]));
                        }
                    }
                    T* dst = xs + (rowi * WP + coli) * 32;
                    const int sx = swz<T>(coli);
                    if constexpr (sizeof(T) == 2) {
                        store8<T>(dst + ((cq * 8) ^ sx), v);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) dst[(cq * 8 + e) ^ sx] = v[e];
                    }
                }
            }
            if (nchunks > 1) stage_w(kc);
            __syncthreads();

            // ---- 9 taps x (32/KSTEP) k-steps of MFMA ------------------------------------------
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ti = tap / 3, tj = tap % 3;
                int xoff[MT], xsw[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    xoff[mt] = ((prow[mt] + ti) * WP + pcol[mt] + tj) * 32;
                    xsw[mt] = swz<T>(pcol[mt] + tj);
                }
#pragma unroll 4
                for (int ks = 0; ks < 32 / KSTEP; ++ks) {
                    frag_t wf[NT], xf[MT];
                    const int kb = ks * KSTEP + hh * KR;   // first channel of this lane's fragment
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        wf[nt] = *reinterpret_cast<const frag_t*>(
                            ws + ((tap * (32 / KR) + kb / KR) * BN + nt * 32 + r) * KR);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        xf[mt] = *reinterpret_cast<const frag_t*>(xs + xoff[mt] + (kb ^ xsw[mt]));
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma(wf[nt], xf[mt], acc[mt][nt]);
                }
            }
        }

        // ---- epilogue: lane = pixel, registers 4g..4g+3 = 4 consecutive output channels ---------
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int h = h0 + prow[mt];
            const bool valid = h < H;
            const size_t pixbase = (((size_t)b * H + h) * W + pcol[mt]) * Coutp;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = n0 + nt * 32 + 8 * g + 4 * hh;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][4 * g + e];
                    if (valid) {
                        if (epi == SED_EPI_STATS) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                S[nt][4 * g + e] += v[e];
                                Q[nt][4 * g + e] = fmaf(v[e], v[e], Q[nt][4 * g + e]);
                            }
                        } else if (epi == SED_EPI_RELUBWD) {
                            float zv[4];
                            load4<T>(zr + pixbase + c, zv);
                            const f32x4 es = *reinterpret_cast<const f32x4*>(p.epi_scale + c);
                            const f32x4 et = *reinterpret_cast<const f32x4*>(p.epi_shift + c);
                            const f32x4 em = *reinterpret_cast<const f32x4*>(p.epi_mean + c);
                            const f32x4 ei = *reinterpret_cast<const f32x4*>(p.epi_invstd + c);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float gate = fmaf(zv[e], es[e], et[e]) > 0.f ? v[e] : 0.f;
                                const float xh = (zv[e] - em[e]) * ei[e];
                                v[e] = gate;
                                S[nt][4 * g + e] += gate;
                                Q[nt][4 * g + e] = fmaf(gate, xh, Q[nt][4 * g + e]);
                            }
                        }
                        store4<T>(zg + pixbase + c, v);
                    }
                }
            }
        }
    }

    // ---- per-workgroup statistics partial ------------------------------------------------------
    if (epi != SED_EPI_STORE) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);   // [wave][quarter][stat][nt][16]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float s = row16_sum(S[nt][i]);
                const float q = row16_sum(Q[nt][i]);
                if ((lane & 15) == 0) {
                    const int quarter = lane >> 4;
                    red[(((wave * 4 + quarter) * 2 + 0) * NT + nt) * 16 + i] = s;
                    red[(((wave * 4 + quarter) * 2 + 1) * NT + nt) * 16 + i] = q;
                }
            }
        __syncthreads();
        if (tid < 2 * BN) {
            const int stat = tid / BN, cn = tid % BN;
            const int nt = cn >> 5, within = cn & 31;
            const int hhh = (within >> 2) & 1;
            const int reg = (within & 3) + 4 * (within >> 3);
            float tot = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
                    tot += red[(((wv * 4 + 2 * hhh + qq) * 2 + stat) * NT + nt) * 16 + reg];
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

__device__ __forceinline__ bf16x4 ds_read_tr16_b64(const bf16_t* p) {
    bf16x4 v;
    const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

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

    const int cq = tid & 3;
    float sc[8], sh[8];
    if (pro == SED_PRO_BNRELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = p.pro_scale[ci0 + cq * 8 + e];
            sh[e] = p.pro_shift[ci0 + cq * 8 + e];
        }
    }

    const int t_begin = strip * p.tpb;
    const int t_end = min(p.totalTiles, t_begin + p.tpb);
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int b = tile / p.tilesPerImg;
        const int h0 = (tile - b * p.tilesPerImg) * TH;
        __syncthreads();
        // stage activations (32-channel chunk ci0) with halo, prologue fused
        {
            constexpr int ITEMS = ROWS * (W + 2) * 4;
            for (int it = tid; it < ITEMS; it += 256) {
                const int pix = it >> 2;
                const int rowi = pix / (W + 2), coli = pix - rowi * (W + 2);
                const int h = h0 - 1 + rowi, w = coli - 1;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
                if (h >= 0 && h < H && w >= 0 && w < W) {
                    load8<T>(xg + (((size_t)b * H + h) * W + w) * Cinp + ci0 + cq * 8, v);
                    if (pro == SED_PRO_BNRELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(0.f, fmaf(v[e], sc[e], sh[e]));
                    }
                }
                T* dst = xs + (rowi * WP + coli) * 32;
                const int sx = swz<T>(coli);
                if constexpr (sizeof(T) == 2) {
                    store8<T>(dst + ((cq * 8) ^ sx), v);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) dst[(cq * 8 + e) ^ sx] = v[e];
                }
            }
        }
        // stage dz tile [BM pixels][CO]; rows past H are zero
        {
            constexpr int IPP = CO / 8;
            for (int it = tid; it < BM * IPP; it += 256) {
                const int q = it / IPP, c8 = (it % IPP) * 8;
                const int h = h0 + q / W, w = q % W;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
                if (h < H) load8<T>(dg + (((size_t)b * H + h) * W + w) * Coutp + co0 + c8, v);
                store8<T>(dzs + q * CO + c8, v);
            }
        }
        __syncthreads();

#pragma unroll 1
        for (int k0 = wk * PXW; k0 < (wk + 1) * PXW; k0 += KSTEP) {
            frag_t bf;
            frag_t af[9];
            if constexpr (sizeof(T) == 2) {
                // lane supplies the address of k-row (8*hh + 4*t + q), 4 columns at 16*gbit + 4*pp
                const int i16 = lane & 15, gbit = (lane >> 4) & 1;
                const int qq = i16 >> 2, pp = i16 & 3;
                bf16x4 lo, hi;
                {
                    const int ka = k0 + 8 * hh + qq, kb2 = ka + 4;
                    lo = ds_read_tr16_b64(dzs + ka * CO + wn * 32 + 16 * gbit + 4 * pp);
                    hi = ds_read_tr16_b64(dzs + kb2 * CO + wn * 32 + 16 * gbit + 4 * pp);
                    bf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int ti = tap / 3, tj = tap % 3;
                    const int ka = k0 + 8 * hh + qq, kb2 = ka + 4;
                    const int ra = ka / W + ti, ca = ka % W + tj;
                    const int rb = kb2 / W + ti, cb = kb2 % W + tj;
                    const int ch = 16 * gbit + 4 * pp;
                    lo = ds_read_tr16_b64(xs + (ra * WP + ca) * 32 + (ch ^ swz<T>(ca)));
                    hi = ds_read_tr16_b64(xs + (rb * WP + cb) * 32 + (ch ^ swz<T>(cb)));
                    af[tap] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
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

__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int strips, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float t = 0.f;
        for (int s = 0; s < strips; ++s) t += ws[(size_t)s * n + i];
        out[i] = t;
    }
}

// =================================================================================================
// first layer (Cin = 1): direct, bandwidth bound
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void conv_c1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ stdv, const float* __restrict__ w,
                                                          T* __restrict__ z, float* __restrict__ partial, int B,
                                                          int H, int W, int Cout, int Coutp, int G, int PPB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* wsm = reinterpret_cast<float*>(smem);          // [9][Coutp]
    float* red = wsm + 9 * Coutp;                         // [PPB][2][Coutp] -> reduced by column
    const int tid = threadIdx.x;
    for (int i = tid; i < 9 * Coutp; i += blockDim.x) {
        const int tap = i / Coutp, c = i % Coutp;
        wsm[i] = c < Cout ? w[c * 9 + tap] : 0.f;
    }
    __syncthreads();
    const int cg = tid % G, pl = tid / G;
    float wr[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) wr[t][e] = wsm[t * Coutp + cg * 8 + e];
    float S[8], Q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { S[e] = 0.f; Q[e] = 0.f; }
    const size_t npix = (size_t)B * H * W;
    if (pl < PPB) {
        for (size_t pix = (size_t)blockIdx.x * PPB + pl; pix < npix; pix += (size_t)gridDim.x * PPB) {
            const int wq = pix % W;
            const int h = (pix / W) % H;
            const size_t bimg = pix / ((size_t)W * H) * ((size_t)W * H);
            float a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int hy = h + t / 3 - 1, wx = wq + t % 3 - 1;
                float xv = 0.f;
                if (hy >= 0 && hy < H && wx >= 0 && wx < W) {
                    xv = x[bimg + (size_t)hy * W + wx];
                    if (mean) xv = (xv - mean[wx]) / stdv[wx];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = fmaf(xv, wr[t][e], a[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { S[e] += a[e]; Q[e] = fmaf(a[e], a[e], Q[e]); }
            store8<T>(z + pix * Coutp + cg * 8, a);
        }
    }
    if (partial) {
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

template <typename T>
__global__ __launch_bounds__(256) void conv_c1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ stdv, const T* __restrict__ dz,
                                                            float* __restrict__ partial, int B, int H, int W,
                                                            int Coutp, int G, int PPB) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);   // [PPB][Coutp] per tap
    const int tid = threadIdx.x;
    const int cg = tid % G, pl = tid / G;
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    const size_t npix = (size_t)B * H * W;
    if (pl < PPB) {
        for (size_t pix = (size_t)blockIdx.x * PPB + pl; pix < npix; pix += (size_t)gridDim.x * PPB) {
            const int wq = pix % W;
            const int h = (pix / W) % H;
            const size_t bimg = pix / ((size_t)W * H) * ((size_t)W * H);
            float d[8];
            load8<T>(dz + pix * Coutp + cg * 8, d);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int hy = h + t / 3 - 1, wx = wq + t % 3 - 1;
                float xv = 0.f;
                if (hy >= 0 && hy < H && wx >= 0 && wx < W) {
                    xv = x[bimg + (size_t)hy * W + wx];
                    if (mean) xv = (xv - mean[wx]) / stdv[wx];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[t][e] = fmaf(xv, d[e], acc[t][e]);
            }
        }
    }
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
        if (pl < PPB) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = 0.f;
#pragma unroll
                for (int tt = 0; tt < 9; ++tt) v = (tt == t) ? acc[tt][e] : v;
                red[pl * Coutp + cg * 8 + e] = v;
            }
        }
        __syncthreads();
        for (int i = tid; i < Coutp; i += blockDim.x) {
            float s = 0.f;
            for (int q = 0; q < PPB; ++q) s += red[q * Coutp + i];
            partial[((size_t)blockIdx.x * 9 + t) * Coutp + i] = s;
        }
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
    const long long blocks = cdiv((long long)B * H * W, 64);
    return (int)(blocks < kMaxParts ? blocks : kMaxParts);
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

template <typename T, int W, int BM, int BN>
static int launch_conv(ConvParams& p, hipStream_t st) {
    constexpr int TH = BM / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr size_t lds = ((size_t)(TH + 2) * WP * 32 + 9 * 32 * BN) * sizeof(T);
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<T, W, BM, BN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { sed_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return 3; }
        attr_done = true;
    }
    p.tilesPerImg = cdiv(p.H, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    p.tpb = cdiv(p.totalTiles, p.nparts);
    const int ny = p.Coutp / BN;
    conv_igemm_kernel<T, W, BM, BN><<<dim3(p.nparts * ny), dim3(256), lds, st>>>(p);
    return 0;
}

template <typename T, int BM>
static int dispatch_conv_w(ConvParams& p, int W, hipStream_t st) {
    const bool bn64 = (p.Coutp % 64 == 0);
    switch (W) {
#define SED_CASE(WW)                                                                      \
    case WW:                                                                              \
        return bn64 ? launch_conv<T, WW, BM, 64>(p, st) : launch_conv<T, WW, BM, 32>(p, st);
        SED_CASE(8)
        SED_CASE(16)
        SED_CASE(32)
        SED_CASE(64)
#undef SED_CASE
    }
    sed_set_error("sed_conv3x3_fwd: W must be one of 8,16,32,64");
    return 1;
}

extern "C" int sed_conv3x3_fwd(int dtype, int pro, int epi, const void* x, const float* pro_scale,
                               const float* pro_shift, const void* wpack, void* z, const void* zref,
                               const float* epi_scale, const float* epi_shift, const float* epi_mean,
                               const float* epi_invstd, float* partial, int B, int H, int W, int Cinp, int Coutp,
                               void* stream) {
    SED_REQUIRE(Cinp % 32 == 0 && Coutp % 32 == 0 && Cinp > 0 && Coutp > 0, "channels must be padded to 32");
    SED_REQUIRE(B > 0 && H > 0, "empty input");
    SED_REQUIRE(pro == SED_PRO_NONE || (pro == SED_PRO_BNRELU && pro_scale && pro_shift), "prologue operands");
    SED_REQUIRE(epi == SED_EPI_STORE || partial, "epilogue needs a partial buffer");
    SED_REQUIRE(epi != SED_EPI_RELUBWD || (zref && epi_scale && epi_shift && epi_mean && epi_invstd), "epilogue operands");
    ConvParams p;
    p.x = x; p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.wpack = wpack; p.z = z; p.zref = zref;
    p.epi_scale = epi_scale; p.epi_shift = epi_shift; p.epi_mean = epi_mean; p.epi_invstd = epi_invstd;
    p.partial = partial; p.B = B; p.H = H; p.Cinp = Cinp; p.Coutp = Coutp; p.pro = pro; p.epi = epi;
    p.nparts = sed_conv_nparts(B, H, W);
    int rc;
    if (dtype == SED_BF16) rc = dispatch_conv_w<bf16_t, 256>(p, W, (hipStream_t)stream);
    else if (dtype == SED_F32) rc = dispatch_conv_w<float, 128>(p, W, (hipStream_t)stream);
    else { sed_set_error("sed_conv3x3_fwd: bad dtype"); return 1; }
    if (rc) return rc;
    SED_LAUNCH_CHECK();
    return 0;
}

static int wgrad_strips(int B, int H, int W, int Cinp, int Coutp, int* wn_out) {
    const int wn = Coutp % 128 == 0 ? 4 : (Coutp % 64 == 0 ? 2 : 1);
    if (wn_out) *wn_out = wn;
    const int ny = (Cinp / 32) * (Coutp / (32 * wn));
    const int TH = 128 / W;
    const long long tiles = (long long)B * cdiv(H, TH);
    long long strips = cdiv(768, ny);
    if (strips > tiles) strips = tiles;
    if (strips < 1) strips = 1;
    return (int)strips;
}

extern "C" size_t sed_conv_wgrad_ws_floats(int B, int H, int W, int Cinp, int Coutp) {
    return (size_t)wgrad_strips(B, H, W, Cinp, Coutp, nullptr) * 9 * Cinp * Coutp;
}

template <typename T, int W, int WN>
static int launch_wgrad(WgradParams& p, hipStream_t st) {
    constexpr int TH = 128 / W;
    constexpr int WP = (W + 2 + 3) & ~3;
    constexpr size_t lds_main = ((size_t)(TH + 2) * WP * 32 + 128 * 32 * WN) * sizeof(T);
    constexpr size_t lds_red = (size_t)4 * 16 * 64 * sizeof(float);
    constexpr size_t lds = lds_main > lds_red ? lds_main : lds_red;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<T, W, WN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { sed_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return 3; }
        attr_done = true;
    }
    p.tilesPerImg = cdiv(p.H, TH);
    p.totalTiles = p.B * p.tilesPerImg;
    p.tpb = cdiv(p.totalTiles, p.strips);
    const int ny = (p.Cinp / 32) * (p.Coutp / (32 * WN));
    conv_wgrad_kernel<T, W, WN><<<dim3(p.strips, ny), dim3(256), lds, st>>>(p);
    return 0;
}

template <typename T>
static int dispatch_wgrad(WgradParams& p, int W, int wn, hipStream_t st) {
#define SED_CASE(WW)                                                \
    case WW:                                                        \
        if (wn == 4) return launch_wgrad<T, WW, 4>(p, st);          \
        if (wn == 2) return launch_wgrad<T, WW, 2>(p, st);          \
        return launch_wgrad<T, WW, 1>(p, st);
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

extern "C" int sed_conv3x3_wgrad(int dtype, int pro, const void* x, const float* pro_scale, const float* pro_shift,
                                 const void* dz, float* dwpack, float* workspace, int B, int H, int W, int Cinp,
                                 int Coutp, void* stream) {
    SED_REQUIRE(Cinp % 32 == 0 && Coutp % 32 == 0, "channels must be padded to 32");
    SED_REQUIRE(pro == SED_PRO_NONE || (pro_scale && pro_shift), "prologue operands");
    hipStream_t st = (hipStream_t)stream;
    WgradParams p;
    int wn;
    p.strips = wgrad_strips(B, H, W, Cinp, Coutp, &wn);
    p.x = x; p.pro_scale = pro_scale; p.pro_shift = pro_shift; p.dz = dz; p.ws = workspace;
    p.B = B; p.H = H; p.Cinp = Cinp; p.Coutp = Coutp; p.pro = pro;
    int rc;
    if (dtype == SED_BF16) rc = dispatch_wgrad<bf16_t>(p, W, wn, st);
    else if (dtype == SED_F32) rc = dispatch_wgrad<float>(p, W, wn, st);
    else { sed_set_error("sed_conv3x3_wgrad: bad dtype"); return 1; }
    if (rc) return rc;
    SED_LAUNCH_CHECK();
    const size_t n = (size_t)9 * Cinp * Coutp;
    wgrad_reduce_kernel<<<cdiv(n, 256), 256, 0, st>>>(workspace, dwpack, p.strips, n);
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
    const size_t lds = ((size_t)9 * Coutp + (size_t)PPB * 2 * Coutp) * sizeof(float);
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

extern "C" int sed_conv3x3_c1_wgrad(int dtype, const float* x, const float* mean, const float* stdv, const void* dz,
                                    float* dw_partial, int B, int H, int W, int Coutp, void* stream) {
    SED_REQUIRE(Coutp % 32 == 0 && Coutp <= 2048, "Coutp must be a multiple of 32, <= 2048");
    int G, PPB, threads;
    c1_geometry(Coutp, &G, &PPB, &threads);
    const int grid = sed_conv_c1_nparts(B, H, W);
    const size_t lds = (size_t)PPB * Coutp * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SED_BF16)
        conv_c1_wgrad_kernel<bf16_t><<<grid, threads, lds, st>>>(x, mean, stdv, (const bf16_t*)dz, dw_partial, B, H, W, Coutp, G, PPB);
    else if (dtype == SED_F32)
        conv_c1_wgrad_kernel<float><<<grid, threads, lds, st>>>(x, mean, stdv, (const float*)dz, dw_partial, B, H, W, Coutp, G, PPB);
    else
        SED_REQUIRE(false, "bad dtype");
    SED_LAUNCH_CHECK();
    return 0;
}
