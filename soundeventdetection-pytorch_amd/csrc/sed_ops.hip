// BatchNorm finalisation, fused BN+ReLU+avg-pool (fwd/bwd), head, weighted BCE, Adam-amsgrad and
// layout utilities for gfx950.  All HBM-bound: 16-byte vector accesses, fp32 math, deterministic
// two-stage reductions (per-workgroup partials + fixed-order finalise), no float atomics.
#include "common.h"

#include <math.h>

static thread_local std::string g_last_error;
void sed_set_error(const std::string& s) { g_last_error = s; }

// cached environment knobs (common.h)
#include <map>
#include <mutex>
#include <stdlib.h>
namespace {
std::mutex g_env_mu;
std::map<std::string, std::pair<bool, std::string>> g_env;      // name -> (set?, value)
}
const char* sed_getenv(const char* name) {
    std::lock_guard<std::mutex> lk(g_env_mu);
    auto it = g_env.find(name);
    if (it == g_env.end()) {
        const char* v = getenv(name);
        it = g_env.emplace(name, std::make_pair(v != nullptr, std::string(v ? v : ""))).first;
    }
    return it->second.first ? it->second.second.c_str() : nullptr;
}
extern "C" void sed_config_reload(void) {
    std::lock_guard<std::mutex> lk(g_env_mu);
    g_env.clear();
}
extern "C" const char* sed_last_error(void) { return g_last_error.c_str(); }
extern "C" int sed_abi_version(void) { return SED_ABI_VERSION; }
extern "C" int sed_build_flags(void) {
    int f = 0;
#ifdef SED_EXPERIMENTS
    f |= 1;
#endif
#ifdef SED_DEBUG_SWITCHES
    f |= 2;
#endif
#ifdef SED_STAMPS
    f |= 4;
#endif
    return f;
}
extern "C" int sed_device_cu_count(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return n;
}

// ---------------------------------------------------------------------------------------------
// block-level double reduction helper (256 threads)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_256(double v, double* sm) {
    const int tid = threadIdx.x;
    sm[tid] = v;
    __syncthreads();
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) sm[tid] += sm[tid + s];
        __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm2d training-mode finalise (spectogram_models.py:142-143; Appendix A of SURVEY.md)
// one workgroup per padded channel
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_train_finalize_kernel(const float* __restrict__ partial, int nparts,
                                                                double count, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                float* __restrict__ rmean, float* __restrict__ rvar,
                                                                float momentum, float eps, float* __restrict__ scale,
                                                                float* __restrict__ shift, float* __restrict__ mean_o,
                                                                float* __restrict__ invstd_o, int C, int Cp) {
    __shared__ double sm[256];
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) {
        s += (double)partial[((size_t)i * 2 + 0) * Cp + c];
        q += (double)partial[((size_t)i * 2 + 1) * Cp + c];
    }
    s = block_sum_256(s, sm);
    q = block_sum_256(q, sm);
    if (threadIdx.x == 0) {
        if (c >= C) {
            scale[c] = 0.f; shift[c] = 0.f; mean_o[c] = 0.f; invstd_o[c] = 0.f;
            return;
        }
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float g = gamma[c], b = beta[c];
        const float sc = g * invstd;
        scale[c] = sc;
        shift[c] = b - (float)mean * sc;
        mean_o[c] = (float)mean;
        invstd_o[c] = invstd;
        if (rmean) {
            const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
    }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar,
                                      float eps, float* scale, float* shift, int C, int Cp) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Cp) return;
    if (c >= C) { scale[c] = 0.f; shift[c] = 0.f; return; }
    const float invstd = 1.0f / sqrtf(rvar[c] + eps);
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nparts,
                                                              double count, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ ca, float* __restrict__ cb,
                                                              float* __restrict__ cc, int C, int Cp) {
    __shared__ double sm[256];
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) {
        s += (double)partial[((size_t)i * 2 + 0) * Cp + c];
        q += (double)partial[((size_t)i * 2 + 1) * Cp + c];
    }
    s = block_sum_256(s, sm);
    q = block_sum_256(q, sm);
    if (threadIdx.x == 0) {
        if (c >= C) { ca[c] = 0.f; cb[c] = 0.f; cc[c] = 0.f; return; }
        dbeta[c] = (float)s;
        dgamma[c] = (float)q;
        const double g = gamma[c], is = invstd[c], mu = mean[c];
        const double mg = s / count, mgx = q / count;
        // dz = g*is*(gr - mg - xhat*mgx), xhat = (z-mu)*is
        ca[c] = (float)(g * is);
        cb[c] = (float)(-g * is * is * mgx);
        cc[c] = (float)(-g * is * (mg - mu * is * mgx));
    }
}

// ---------------------------------------------------------------------------------------------
// Row-structured elementwise stages.  One "row" = one (b, h) line of W pixels x G channel groups of
// 8; a workgroup walks rows grid-stride, a thread walks the row's W*G items.  When 256 % G == 0
// (every power-of-two channel count) a thread's channel group is fixed, so the per-channel
// coefficients live in registers for the whole kernel; index math is 32-bit and per row.
// ---------------------------------------------------------------------------------------------
struct RowGeom {
    int B, H, W, Cp, G, items, fixed;   // items = W*G, fixed = (256 % G == 0)
};

__device__ __forceinline__ void load_coef8(const float* __restrict__ p, int cg, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p + cg * 8);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + cg * 8 + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}

// y = avg_pool(relu(scale*z+shift)), pool in {1,2}; rows are OUTPUT rows
// CNT: also store, per pooled element, how many of its POOL*POOL pixels are active (scale*z+shift > 0) as one byte -- the
// backward statistics then need no second pass over z (sed_conv3x3_dgrad_poolstats)
template <typename T, int POOL, bool CNT = false>
__global__ __launch_bounds__(256) void bn_relu_pool_fwd_kernel(const T* __restrict__ z, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, T* __restrict__ y,
                                                               RowGeom g, unsigned char* __restrict__ cnt = nullptr) {
    const int Ho = g.H / POOL, Wo = g.W / POOL, G = g.G, Cp = g.Cp;
    const int items = Wo * G;
    float sc[8], sh[8];
    if (g.fixed) { load_coef8(scale, threadIdx.x % G, sc); load_coef8(shift, threadIdx.x % G, sh); }
    // a row of the BENCH layers has 128 items: two (or more) output rows per pass keep all 256 threads loading
    const int rpi = (items <= 128 && 256 % items == 0) ? 256 / items : 1;
    const int sub = rpi > 1 ? threadIdx.x / items : 0, it0 = rpi > 1 ? threadIdx.x - sub * items : threadIdx.x;
    for (int row0 = blockIdx.x * rpi; row0 < g.B * Ho; row0 += gridDim.x * rpi) {
        const int row = row0 + sub;
        if (row >= g.B * Ho) continue;
        const int b = row / Ho, ho = row - b * Ho;
        const T* __restrict__ zin = z + ((size_t)b * g.H + (size_t)ho * POOL) * g.W * Cp;
        T* __restrict__ yout = y + (size_t)row * Wo * Cp;
        for (int it = it0; it < items; it += 256) {
            const int wo = it / G, cg = it - wo * G;
            if (!g.fixed) { load_coef8(scale, cg, sc); load_coef8(shift, cg, sh); }
            float acc[8];
            unsigned na[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { acc[e] = 0.f; na[e] = 0u; }
#pragma unroll
            for (int dy = 0; dy < POOL; ++dy)
#pragma unroll
                for (int dx = 0; dx < POOL; ++dx) {
                    float v[8];
                    load8<T>(zin + ((size_t)dy * g.W + wo * POOL + dx) * Cp + cg * 8, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float yv = fmaf(v[e], sc[e], sh[e]);
                        acc[e] += fmaxf(0.f, yv);
                        if (CNT) na[e] += yv > 0.f ? 1u : 0u;
                    }
                }
            if (CNT) {
                uint2 pk;
                pk.x = na[0] | (na[1] << 8) | (na[2] << 16) | (na[3] << 24);
                pk.y = na[4] | (na[5] << 8) | (na[6] << 16) | (na[7] << 24);
                *reinterpret_cast<uint2*>(cnt + ((size_t)row * Wo * Cp + (size_t)it * 8)) = pk;
            }
            if (POOL > 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] *= (1.0f / (POOL * POOL));
            }
            store8<T>(yout + (size_t)it * 8, acc);
        }
    }
}

// The 2x2 pooling forward with counts for rows of W*G = 256 items and <= 64 channels (round 5): thread = (INPUT column, channel group), so
// a wave's load covers 1 KB of contiguous memory (in the kernel above a thread owns an OUTPUT column and walks its 2x2 window: with 32
// channels a pixel is 64 B and every load instruction touches 16 half-used 128-B lines).  The two columns of a pooling window sit G lanes
// apart in a 16-lane row: the even column's lane takes its partner's partial sums with DPP row_shl:G adds and stores; two output rows per
// iteration keep four 16-byte loads in flight per thread as before.  Same values: the four relu(bn(z)) terms are added as (row 0 + row 1)
// of a column, then the two columns -- the generic kernel adds them in (dy, dx) order, so the bf16 result can differ in the last bit.
template <int G>
__global__ __launch_bounds__(256) void bn_relu_pool2_cnt_pair_kernel(const bf16_t* __restrict__ z, const float* __restrict__ scale,
                                                                     const float* __restrict__ shift, bf16_t* __restrict__ y,
                                                                     unsigned char* __restrict__ cnt, int B, int H, int W) {
    static_assert(G == 4 || G == 8, "the partner column sits in the same 16-lane row");
    constexpr int Cp = G * 8;
    const int Ho = H >> 1, Wo = W >> 1;
    const int tid = threadIdx.x, px = tid / G, cg = tid % G;
    float sc[8], sh[8];
    load_coef8(scale, cg, sc);
    load_coef8(shift, cg, sh);
    const size_t rowe = (size_t)W * Cp;
    const bool even = !(px & 1);
    const long long nrows = (long long)B * Ho;
    for (long long row0 = (long long)blockIdx.x * 2; row0 < nrows; row0 += (long long)gridDim.x * 2) {
        float v[2][2][8];
        bool live[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long row = row0 + u;
            live[u] = row < nrows;
            const long long rr = live[u] ? row : row0;
            const int b = (int)(rr / Ho), ho = (int)(rr - (long long)b * Ho);
            const bf16_t* zin = z + ((size_t)b * H + 2 * (size_t)ho) * rowe + (size_t)tid * 8;
            load8<bf16_t>(zin, v[u][0]);
            load8<bf16_t>(zin + rowe, v[u][1]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float acc[8];
            unsigned nlo = 0, nhi = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y0 = fmaf(v[u][0][e], sc[e], sh[e]), y1 = fmaf(v[u][1][e], sc[e], sh[e]);
                acc[e] = fmaxf(0.f, y0) + fmaxf(0.f, y1);
                const unsigned n = (y0 > 0.f ? 1u : 0u) + (y1 > 0.f ? 1u : 0u);
                if (e < 4) nlo |= n << (8 * e); else nhi |= n << (8 * (e - 4));
            }
            // partner column (lane + G): row_shl:G = 0x100 + G
#pragma unroll
            for (int e = 0; e < 8; ++e)
                acc[e] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc[e]), 0x100 + G, 0xF, 0xF, true));
            nlo += (unsigned)__builtin_amdgcn_update_dpp(0, (int)nlo, 0x100 + G, 0xF, 0xF, true);
            nhi += (unsigned)__builtin_amdgcn_update_dpp(0, (int)nhi, 0x100 + G, 0xF, 0xF, true);
            if (even && live[u]) {
                const size_t o = ((size_t)(row0 + u) * Wo + (px >> 1)) * Cp + cg * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] *= 0.25f;
                store8<bf16_t>(y + o, acc);
                uint2 pk; pk.x = nlo; pk.y = nhi;
                *reinterpret_cast<uint2*>(cnt + o) = pk;
            }
        }
    }
}

// backward pass 1: statistics of g = up(dy)/pool^2 * relu'(bn(z)); rows are INPUT rows
template <typename T, int POOL>
__global__ __launch_bounds__(256) void pool_relu_bwd_stats_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                                  const float* __restrict__ scale,
                                                                  const float* __restrict__ shift,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  float* __restrict__ partial, RowGeom g,
                                                                  const int* __restrict__ flag = nullptr, int zero_to = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);   // [256/G][2][Cp]
    if (flag != nullptr && *flag == 0) return;      // the pooled-tensor statistics of the data-gradient kernel stand (uniform)
    const int tid = threadIdx.x, G = g.G, Cp = g.Cp, W = g.W;
    const int Ho = g.H / POOL, Wo = W / POOL;
    const int cg = tid % G, pl = tid / G, PPB = 256 / G;
    float S[8], Q[8], sc[8], sh[8], mu[8], is[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { S[e] = 0.f; Q[e] = 0.f; }
    load_coef8(scale, cg, sc); load_coef8(shift, cg, sh); load_coef8(mean, cg, mu); load_coef8(invstd, cg, is);
    auto accum = [&](const float (&zv)[8], const float (&dv)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gg = fmaf(zv[e], sc[e], sh[e]) > 0.f ? dv[e] * (1.0f / (POOL * POOL)) : 0.f;
            S[e] += gg;
            Q[e] = fmaf(gg, (zv[e] - mu[e]) * is[e], Q[e]);
        }
    };
    if (POOL == 2 && pl < PPB) {
        // the two input rows of a pooled row share their dy line: three loads in flight per thread instead of two dependent
        // pairs, and dy is read once (row order within a channel's sum: h, h+1 -- as the one-row loop below)
        for (int prow = blockIdx.x; prow < g.B * Ho; prow += gridDim.x) {
            const int b = prow / Ho, ho = prow - b * Ho;
            const T* __restrict__ z0 = z + ((size_t)b * g.H + 2 * ho) * W * Cp;
            const T* __restrict__ din = dy + (size_t)prow * Wo * Cp;
            for (int w = pl; w < Wo * 2; w += PPB) {
                float za[8], zb[8], dv[8];
                load8<T>(z0 + (size_t)w * Cp + cg * 8, za);
                load8<T>(z0 + ((size_t)W + w) * Cp + cg * 8, zb);
                load8<T>(din + (size_t)(w >> 1) * Cp + cg * 8, dv);
                accum(za, dv);
                accum(zb, dv);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(pl * 2 + 0) * Cp + cg * 8 + e] = S[e];
            red[(pl * 2 + 1) * Cp + cg * 8 + e] = Q[e];
        }
    } else if (pl < PPB) {
        for (int row = blockIdx.x; row < g.B * Ho * POOL; row += gridDim.x) {   // rows dropped by the floor never enter
            const int b = row / (Ho * POOL), h = row - b * (Ho * POOL);
            const T* __restrict__ zin = z + ((size_t)b * g.H + h) * W * Cp;
            const T* __restrict__ din = dy + ((size_t)b * Ho + h / POOL) * Wo * Cp;
            for (int w = pl; w < Wo * POOL; w += PPB) {
                float zv[8], dv[8];
                load8<T>(zin + (size_t)w * Cp + cg * 8, zv);
                load8<T>(din + (size_t)(w / POOL) * Cp + cg * 8, dv);
                accum(zv, dv);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[(pl * 2 + 0) * Cp + cg * 8 + e] = S[e];
            red[(pl * 2 + 1) * Cp + cg * 8 + e] = Q[e];
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * Cp; i += blockDim.x) {
        float t = 0.f;
        for (int q = 0; q < PPB; ++q) t += red[q * 2 * Cp + i];
        partial[(size_t)blockIdx.x * 2 * Cp + i] = t;
        for (int row = blockIdx.x + gridDim.x; row < zero_to; row += gridDim.x) partial[(size_t)row * 2 * Cp + i] = 0.f;
    }
}

// backward pass 2: dz = ca*g + cb*z + cc; rows are INPUT rows (all H of them)
template <typename T, int POOL>
__global__ __launch_bounds__(256) void pool_relu_bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ shift,
                                                                     const float* __restrict__ ca,
                                                                     const float* __restrict__ cb,
                                                                     const float* __restrict__ cc, T* __restrict__ dz,
                                                                     RowGeom g) {
    const int G = g.G, Cp = g.Cp, W = g.W;
    const int Ho = g.H / POOL, Wo = W / POOL;
    float sc[8], sh[8], a8[8], b8[8], c8[8];
    if (g.fixed) {
        const int cg = threadIdx.x % G;
        load_coef8(scale, cg, sc); load_coef8(shift, cg, sh);
        load_coef8(ca, cg, a8); load_coef8(cb, cg, b8); load_coef8(cc, cg, c8);
    }
    for (int row = blockIdx.x; row < g.B * g.H; row += gridDim.x) {
        const int b = row / g.H, h = row - b * g.H;
        const T* __restrict__ zin = z + (size_t)row * W * Cp;
        T* __restrict__ dout = dz + (size_t)row * W * Cp;
        const bool hin = h < Ho * POOL;
        const T* __restrict__ din = dy + ((size_t)b * Ho + (hin ? h / POOL : 0)) * Wo * Cp;
        for (int it = threadIdx.x; it < g.items; it += 256) {
            const int w = it / G, cg = it - w * G;
            if (!g.fixed) {
                load_coef8(scale, cg, sc); load_coef8(shift, cg, sh);
                load_coef8(ca, cg, a8); load_coef8(cb, cg, b8); load_coef8(cc, cg, c8);
            }
            float zv[8], dv[8], o[8];
            load8<T>(zin + (size_t)it * 8, zv);
            const bool inside = hin && (w < Wo * POOL);
            if (inside) load8<T>(din + (size_t)(w / POOL) * Cp + cg * 8, dv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float gg = (inside && fmaf(zv[e], sc[e], sh[e]) > 0.f) ? dv[e] * (1.0f / (POOL * POOL)) : 0.f;
                o[e] = fmaf(a8[e], gg, fmaf(b8[e], zv[e], c8[e]));
            }
            store8<T>(dout + (size_t)it * 8, o);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ z,
                                                           const float* __restrict__ ca, const float* __restrict__ cb,
                                                           const float* __restrict__ cc, T* __restrict__ dz, size_t npix,
                                                           int Cp) {
    const int G = Cp >> 3;
    const size_t total = npix * G;
    const bool fixed = (256 % G == 0);     // then (gridDim*256) % G == 0 too: the channel group is per thread
    float a8[8], b8[8], c8[8];
    if (fixed) {
        const int cg = threadIdx.x % G;
        load_coef8(ca, cg, a8); load_coef8(cb, cg, b8); load_coef8(cc, cg, c8);
    }
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        if (!fixed) {
            const int cg = (int)(idx % G);
            load_coef8(ca, cg, a8); load_coef8(cb, cg, b8); load_coef8(cc, cg, c8);
        }
        float gv[8], zv[8], o[8];
        load8<T>(g + idx * 8, gv);
        load8<T>(z + idx * 8, zv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(a8[e], gv[e], fmaf(b8[e], zv[e], c8[e]));
        store8<T>(dz + idx * 8, o);
    }
}

// ---------------------------------------------------------------------------------------------
// head: mean over mel -> Linear -> logits (spectogram_models.py:193-197)
// one workgroup (128 threads) per (b, t)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(128) void head_fwd_kernel(const T* __restrict__ feat, const float* __restrict__ fc_w,
                                                       const float* __restrict__ fc_b, float* __restrict__ m_out,
                                                       float* __restrict__ pre, int Wf, int C, int Cp, int K) {
    __shared__ float ms[2048];
    __shared__ float wred[2];
    const size_t row = blockIdx.x;
    const float inv = 1.0f / (float)Wf;
    for (int c = threadIdx.x; c < Cp; c += 128) {
        float s = 0.f;
        for (int w = 0; w < Wf; ++w) s += to_f(feat[(row * Wf + w) * Cp + c]);
        s *= inv;
        ms[c] = s;
        m_out[row * Cp + c] = s;
    }
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        float s = 0.f;
        for (int c = threadIdx.x; c < C; c += 128) s = fmaf(ms[c], fc_w[(size_t)k * C + c], s);
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) pre[row * K + k] = wred[0] + wred[1] + fc_b[k];
        __syncthreads();
    }
}

// Cp = 128: 16 rows per 256-thread workgroup -- thread (row slot, 8-channel group) reads the row's Wf pixels with 16 / 32-byte loads,
// the K x 16 dot products read the means and the (padded: conflict-free) weights from LDS; one barrier instead of 2 K + 1 per row
// (0.023 -> 0.012 ms on the bench head: the 49 MB read is what is left)
#define HF_KMAX 32
template <typename T>
__global__ __launch_bounds__(256) void head_fwd16_kernel(const T* __restrict__ feat, const float* __restrict__ fc_w,
                                                         const float* __restrict__ fc_b, float* __restrict__ m_out,
                                                         float* __restrict__ pre, int rows, int Wf, int C, int K) {
    constexpr int Cp = 128, WPITCH = Cp + 1;
    __shared__ float ms[16 * Cp];
    __shared__ float wsm[HF_KMAX * WPITCH];
    const int tid = threadIdx.x, cg = tid & 15, slot = tid >> 4;
    for (int i = tid; i < K * Cp; i += 256) {
        const int k = i >> 7, c = i & 127;
        wsm[k * WPITCH + c] = c < C ? fc_w[(size_t)k * C + c] : 0.f;
    }
    const size_t row = (size_t)blockIdx.x * 16 + slot;
    if (row < (size_t)rows) {
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
        const T* src = feat + (row * Wf) * Cp + cg * 8;
        for (int w = 0; w < Wf; ++w) {
            float v[8];
            load8<T>(src + (size_t)w * Cp, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += v[e];
        }
        const float inv = 1.0f / (float)Wf;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] *= inv; ms[slot * Cp + cg * 8 + e] = s[e]; }
        store8<float>(m_out + row * Cp + cg * 8, s);
    }
    __syncthreads();
    for (int o = tid; o < 16 * K; o += 256) {
        const int r = o / K, k = o - r * K;
        const size_t rw = (size_t)blockIdx.x * 16 + r;
        if (rw >= (size_t)rows) continue;
        float a = 0.f;
        for (int c = 0; c < Cp; ++c) a = fmaf(ms[r * Cp + c], wsm[k * WPITCH + c], a);
        pre[rw * K + k] = a + fc_b[k];
    }
}

__global__ void interpolate_kernel(const float* __restrict__ pre, float* __restrict__ out, int t, int K, int ratio,
                                   size_t total) {
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int k = idx % K;
        const size_t f = idx / K;              // b*(t*ratio) + frame
        const size_t b = f / ((size_t)t * ratio);
        const int frame = f % ((size_t)t * ratio);
        out[idx] = pre[(b * t + frame / ratio) * K + k];
    }
}

// ---------------------------------------------------------------------------------------------
// WeightedBCE on the virtually-interpolated logits (utils/common.py:16-30)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }

__global__ __launch_bounds__(256) void bce_fwd_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ target,
                                                          float* __restrict__ dpre, float* __restrict__ loss_partial,
                                                          int B, int t, int K, int ratio, int Tt, int N, float wpos,
                                                          float inv_numel, float grad_scale) {
    __shared__ float wred[4];
    const size_t total = (size_t)B * t * K;
    const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    float lsum = 0.f;
    if (idx < total) {
        const int k = idx % K;
        const size_t bt = idx / K;
        const int tt = bt % t;
        const size_t b = bt / t;
        const float x = pre[idx];
        const float lsp = log_sigmoid(x), lsn = log_sigmoid(-x);
        const float sg = 1.0f / (1.0f + expf(-x));
        float gsum = 0.f;
        for (int j = 0; j < ratio; ++j) {
            const int f = tt * ratio + j;
            if (f < N) {
                const float y = target[(b * Tt + f) * K + k];
                lsum -= wpos * y * lsp + (1.f - y) * lsn;
                gsum += sg * (1.f + (wpos - 1.f) * y) - wpos * y;
            }
        }
        if (dpre) dpre[idx] = gsum * inv_numel * grad_scale;
    }
    lsum = wave_sum(lsum);
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) loss_partial[blockIdx.x] = (wred[0] + wred[1]) + (wred[2] + wred[3]);
}

__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ partial, int n, float inv_numel,
                                                            float* __restrict__ loss) {
    __shared__ double sm[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    s = block_sum_256(s, sm);
    if (threadIdx.x == 0) loss[0] = (float)(s * (double)inv_numel);
}

// ---------------------------------------------------------------------------------------------
// head backward
// ---------------------------------------------------------------------------------------------
#define HEAD_BWD_ROWS 8     // rows per workgroup (64 left the 375-workgroup launch latency-bound: 63 us)
// dlog is [rows*ratio][K]: the gradient w.r.t. the interpolated logits (ratio = 1: w.r.t. pre);
// the x`ratio` repeat backward (sum of `ratio` consecutive frames) is folded in here.
__device__ __forceinline__ float dpre_at(const float* __restrict__ dlog, size_t rr, int k, int K, int ratio) {
    float a = 0.f;
    for (int j = 0; j < ratio; ++j) a += dlog[(rr * ratio + j) * K + k];
    return a;
}
template <typename T>
__global__ __launch_bounds__(128) void head_bwd_kernel(const float* __restrict__ dlog, const float* __restrict__ m,
                                                       const float* __restrict__ fc_w, float* __restrict__ ws,
                                                       T* __restrict__ dfeat, size_t rows, int Wf, int C, int Cp,
                                                       int K, int ratio) {
    const size_t r0 = (size_t)blockIdx.x * HEAD_BWD_ROWS;
    const size_t r1 = r0 + HEAD_BWD_ROWS < rows ? r0 + HEAD_BWD_ROWS : rows;
    float* wsb = ws + (size_t)blockIdx.x * ((size_t)K * C + K);
    for (int pair = threadIdx.x; pair < K * C; pair += 128) {
        const int k = pair / C, c = pair % C;
        float a = 0.f;
        for (size_t rr = r0; rr < r1; ++rr) a = fmaf(dpre_at(dlog, rr, k, K, ratio), m[rr * Cp + c], a);
        wsb[pair] = a;
    }
    for (int k = threadIdx.x; k < K; k += 128) {
        float a = 0.f;
        for (size_t rr = r0; rr < r1; ++rr) a += dpre_at(dlog, rr, k, K, ratio);
        wsb[(size_t)K * C + k] = a;
    }
    const float inv = 1.0f / (float)Wf;
    for (size_t rr = r0; rr < r1; ++rr) {
        for (int c = threadIdx.x; c < Cp; c += 128) {
            float v = 0.f;
            if (c < C)
                for (int k = 0; k < K; ++k) v = fmaf(dpre_at(dlog, rr, k, K, ratio), fc_w[(size_t)k * C + c], v);
            const T o = from_f<T>(v * inv);
            for (int w = 0; w < Wf; ++w) dfeat[(rr * Wf + w) * Cp + c] = o;
        }
    }
}

// 64 rows per 256-thread workgroup (375 instead of 3000 partial rows at the bench size): the x`ratio` repeat backward of the
// block's rows and the FC weights are staged in LDS once, the weight-gradient pairs walk the rows from there, the feature
// gradient goes out as 16 / 32-byte stores (one per mel bin of the row)
#define HB_ROWS 64
#define HB_KMAX 32
#define HB_WCAP 8192
template <typename T>
__global__ __launch_bounds__(256) void head_bwd64_kernel(const float* __restrict__ dlog, const float* __restrict__ m,
                                                         const float* __restrict__ fc_w, float* __restrict__ ws,
                                                         T* __restrict__ dfeat, size_t rows, int Wf, int C, int Cp, int K, int ratio) {
    __shared__ float sd[HB_ROWS * HB_KMAX];
    __shared__ float wsm[HB_WCAP];
    const int tid = threadIdx.x;
    const size_t r0 = (size_t)blockIdx.x * HB_ROWS;
    const int nr = (int)(r0 + HB_ROWS < rows ? HB_ROWS : rows - r0);
    for (int i = tid; i < nr * K; i += 256) {
        const int r = i / K, k = i - r * K;
        sd[r * K + k] = dpre_at(dlog, r0 + r, k, K, ratio);
    }
    for (int i = tid; i < K * C; i += 256) wsm[i] = fc_w[i];
    __syncthreads();
    float* wsb = ws + (size_t)blockIdx.x * ((size_t)K * C + K);
    for (int pair = tid; pair < K * C; pair += 256) {
        const int k = pair / C, c = pair - k * C;
        float a = 0.f;
        for (int r = 0; r < nr; ++r) a = fmaf(sd[r * K + k], m[(r0 + r) * Cp + c], a);
        wsb[pair] = a;
    }
    if (tid < K) {
        float a = 0.f;
        for (int r = 0; r < nr; ++r) a += sd[r * K + tid];
        wsb[(size_t)K * C + tid] = a;
    }
    const int ncg = Cp >> 3, nslot = 256 / ncg;             // (Cp / 8 divides 256: checked by the launcher)
    const int cg = tid % ncg, slot = tid / ncg;
    const float inv = 1.0f / (float)Wf;
    for (int r = slot; r < nr; r += nslot) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cg * 8 + e;
            float a = 0.f;
            if (c < C)
                for (int k = 0; k < K; ++k) a = fmaf(sd[r * K + k], wsm[k * C + c], a);
            v[e] = a * inv;
        }
        T* dst = dfeat + ((r0 + r) * Wf) * Cp + cg * 8;
        for (int w = 0; w < Wf; ++w) store8<T>(dst + (size_t)w * Cp, v);
    }
}

// the two outputs of the head's weight-gradient partials (dW: n1 values, db: n2 values, consecutive in a partial row) in one launch
__global__ __launch_bounds__(256) void sum_partials2_kernel(const float* __restrict__ ws, float* __restrict__ out1, size_t n1,
                                                            float* __restrict__ out2, size_t n2, int nparts, size_t stride) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t i = wave; i < n1 + n2; i += nwaves) {
        double t = 0.0;
        for (int s = lane; s < nparts; s += 64) t += (double)ws[(size_t)s * stride + i];
#pragma unroll
        for (int mm = 32; mm >= 1; mm >>= 1) t += __shfl_xor(t, mm, 64);
        if (lane == 0) {
            if (i < n1) out1[i] = (float)t;
            else out2[i - n1] = (float)t;
        }
    }
}

// out[i] = sum_s ws[s*stride + i]; one wave per output walks the partials 64 at a time (fixed order)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                           int nparts, size_t n, size_t stride) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t i = wave; i < n; i += nwaves) {
        // fp64 accumulation, fixed order (lane-strided rows, then a butterfly over the lanes): one rounding to fp32 at the end,
        // so a row reduced here and finalized later (SyncBN) differs from the fp64 finalize of all rows by that rounding only
        double t = 0.0;
        for (int s = lane; s < nparts; s += 64) t += (double)ws[(size_t)s * stride + i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
        if (lane == 0) out[i] = (float)t;
    }
}

// ---------------------------------------------------------------------------------------------
// Adam-amsgrad (train.py:85): single-tensor torch semantics on a flat buffer
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_amsgrad_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                           float* __restrict__ m, float* __restrict__ v,
                                                           float* __restrict__ vmax, size_t n, float one_minus_b1,
                                                           float b2, float one_minus_b2, float eps, float step_size,
                                                           float inv_sqrt_bc2, float grad_scale) {
    const size_t n4 = n >> 2;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
        f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
        f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
        f32x4 xv = reinterpret_cast<f32x4*>(vmax)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gv[e] * grad_scale;
            mv[e] = mv[e] + one_minus_b1 * (gr - mv[e]);               // exp_avg.lerp_(grad, 1-beta1)
            vv[e] = vv[e] * b2 + (one_minus_b2 * gr) * gr;             // mul_(beta2).addcmul_(g, g, 1-beta2)
            xv[e] = fmaxf(xv[e], vv[e]);                               // max on the un-corrected v
            const float denom = sqrtf(xv[e]) * inv_sqrt_bc2 + eps;
            pv[e] = pv[e] - step_size * (mv[e] / denom);
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
        reinterpret_cast<f32x4*>(vmax)[i] = xv;
    }
    // tail
    const size_t tail0 = n4 << 2;
    const size_t gt = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (gt < n - tail0) {
        const size_t i = tail0 + gt;
        const float gr = g[i] * grad_scale;
        const float mm = m[i] + one_minus_b1 * (gr - m[i]);
        const float vv = v[i] * b2 + (one_minus_b2 * gr) * gr;
        const float xx = fmaxf(vmax[i], vv);
        const float denom = sqrtf(xx) * inv_sqrt_bc2 + eps;
        p[i] = p[i] - step_size * (mm / denom);
        m[i] = mm; v[i] = vv; vmax[i] = xx;
    }
}

// ---------------------------------------------------------------------------------------------
// casts / layout
// ---------------------------------------------------------------------------------------------
template <typename TD, typename TS>
__global__ void cast_kernel(TD* __restrict__ dst, const TS* __restrict__ src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = from_f<TD>(to_f(src[i]));
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int C, int H, int W,
                                    int Cp) {
    const size_t total = (size_t)B * H * W * Cp;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int c = idx % Cp;
        size_t t = idx / Cp;
        const int w = t % W; t /= W;
        const int h = t % H;
        const int b = t / H;
        dst[idx] = from_f<T>(c < C ? src[(((size_t)b * C + c) * H + h) * W + w] : 0.f);
    }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, int C, int H, int W,
                                    int Cp) {
    const size_t total = (size_t)B * C * H * W;
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int w = idx % W;
        size_t t = idx / W;
        const int h = t % H; t /= H;
        const int c = t % C;
        const int b = t / C;
        dst[idx] = to_f(src[(((size_t)b * H + h) * W + w) * Cp + c]);
    }
}

// =================================================================================================
// C ABI
// =================================================================================================
static RowGeom row_geom(int B, int H, int W, int Cp) {
    RowGeom g;
    g.B = B; g.H = H; g.W = W; g.Cp = Cp; g.G = Cp / 8; g.items = W * g.G; g.fixed = (256 % g.G == 0) ? 1 : 0;
    return g;
}
static inline int row_grid(long long rows) { return (int)(rows < 1 ? 1 : (rows > 8192 ? 8192 : rows)); }

static inline int ew_grid(size_t items) {
    const size_t g = (items + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

extern "C" int sed_bn_train_finalize(const float* partial, int nparts, double count, const float* gamma,
                                     const float* beta, float* running_mean, float* running_var, float momentum,
                                     float eps, float* scale, float* shift, float* mean, float* invstd, int C, int Cp,
                                     void* stream) {
    SED_REQUIRE(nparts > 0 && count > 0 && C <= Cp, "bad sizes");
    SED_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "running stats must both be given or both NULL");
    bn_train_finalize_kernel<<<Cp, 256, 0, (hipStream_t)stream>>>(partial, nparts, count, gamma, beta, running_mean,
                                                                  running_var, momentum, eps, scale, shift, mean, invstd,
                                                                  C, Cp);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* scale, float* shift, int C, int Cp,
                                  void* stream) {
    bn_eval_coeffs_kernel<<<cdiv(Cp, 256), 256, 0, (hipStream_t)stream>>>(gamma, beta, running_mean, running_var, eps,
                                                                          scale, shift, C, Cp);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_bn_bwd_finalize(const float* partial, int nparts, double count, const float* gamma,
                                   const float* mean, const float* invstd, float* dgamma, float* dbeta, float* ca,
                                   float* cb, float* cc, int C, int Cp, void* stream) {
    SED_REQUIRE(nparts > 0 && count > 0 && C <= Cp, "bad sizes");
    bn_bwd_finalize_kernel<<<Cp, 256, 0, (hipStream_t)stream>>>(partial, nparts, count, gamma, mean, invstd, dgamma,
                                                                dbeta, ca, cb, cc, C, Cp);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_bn_relu_pool_fwd(int dtype, const void* z, const float* scale, const float* shift, void* y, int B,
                                    int H, int W, int Cp, int pool, void* stream) {
    SED_REQUIRE(Cp % 32 == 0, "Cp must be a multiple of 32");
    SED_REQUIRE(pool == 1 || (H >= 2 && W >= 2), "input smaller than the pooling window");
    hipStream_t st = (hipStream_t)stream;
    const RowGeom geo = row_geom(B, H, W, Cp);
    const int grid = row_grid((long long)B * (H / pool));
#define ARGS <<<grid, 256, 0, st>>>((const T_*)z, scale, shift, (T_*)y, geo)
    if (dtype == SED_BF16 && pool == 1) { typedef bf16_t T_; bn_relu_pool_fwd_kernel<T_, 1> ARGS; }
    else if (dtype == SED_BF16 && pool == 2) { typedef bf16_t T_; bn_relu_pool_fwd_kernel<T_, 2> ARGS; }
    else if (dtype == SED_F32 && pool == 1) { typedef float T_; bn_relu_pool_fwd_kernel<T_, 1> ARGS; }
    else if (dtype == SED_F32 && pool == 2) { typedef float T_; bn_relu_pool_fwd_kernel<T_, 2> ARGS; }
    else SED_REQUIRE(false, "dtype must be SED_F32/SED_BF16 and pool 1 or 2");
#undef ARGS
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_bn_relu_pool_cnt_fwd(int dtype, const void* z, const float* scale, const float* shift, void* y, void* cnt,
                                        int B, int H, int W, int Cp, void* stream) {
    SED_REQUIRE(Cp % 32 == 0, "Cp must be a multiple of 32");
    SED_REQUIRE(H >= 2 && W >= 2 && cnt != nullptr, "2x2 pooling window, count buffer");
    hipStream_t st = (hipStream_t)stream;
    const RowGeom geo = row_geom(B, H, W, Cp);
    const int grid = row_grid((long long)B * (H / 2));
    {   // SED_POOL_PAIR=1: the pair-lane kernel (A/B).  Measured neutral (profiles/r05_h_ab_pool_pair_lanes.txt: block 0 0.1955 vs 0.1989 ms, block 1
        // 0.0960 vs 0.0951 ms): the generic kernel already moves its 1.08 GB at 5.5 TB/s -- the half-used lines of a load cost nothing
        const char* e = sed_getenv("SED_POOL_PAIR");
        const bool pair = e && e[0] == '1';
        if (pair && dtype == SED_BF16 && W * (Cp / 8) == 256 && (Cp == 32 || Cp == 64) && (W & 1) == 0) {
            const int g2 = row_grid(((long long)B * (H / 2) + 1) / 2);
            if (Cp == 32) bn_relu_pool2_cnt_pair_kernel<4><<<g2, 256, 0, st>>>((const bf16_t*)z, scale, shift, (bf16_t*)y, (unsigned char*)cnt, B, H, W);
            else bn_relu_pool2_cnt_pair_kernel<8><<<g2, 256, 0, st>>>((const bf16_t*)z, scale, shift, (bf16_t*)y, (unsigned char*)cnt, B, H, W);
            SED_LAUNCH_CHECK();
            return 0;
        }
    }
    if (dtype == SED_BF16) bn_relu_pool_fwd_kernel<bf16_t, 2, true><<<grid, 256, 0, st>>>((const bf16_t*)z, scale, shift, (bf16_t*)y, geo, (unsigned char*)cnt);
    else if (dtype == SED_F32) bn_relu_pool_fwd_kernel<float, 2, true><<<grid, 256, 0, st>>>((const float*)z, scale, shift, (float*)y, geo, (unsigned char*)cnt);
    else SED_REQUIRE(false, "dtype must be SED_F32/SED_BF16");
    SED_LAUNCH_CHECK();
    return 0;
}

static void stats_geometry(int Cp, int* G, int* PPB) {
    *G = Cp / 8;
    *PPB = 256 / *G;
    if (*PPB < 1) *PPB = 1;
}

extern "C" int sed_pool_bwd_nparts(int B, int H, int W, int Cp) {
    int G, PPB;
    stats_geometry(Cp, &G, &PPB);
    (void)W; (void)PPB;
    const long long rows = (long long)B * H;
    return (int)(rows < 1024 ? (rows < 1 ? 1 : rows) : 1024);
}

extern "C" int sed_pool_relu_bwd_stats(int dtype, const void* dy, const void* z, const float* scale,
                                       const float* shift, const float* mean, const float* invstd, float* partial,
                                       int B, int H, int W, int Cp, int pool, void* stream) {
    SED_REQUIRE(Cp % 32 == 0 && Cp <= 2048, "Cp must be a multiple of 32, <= 2048");
    int G, PPB;
    stats_geometry(Cp, &G, &PPB);
    const int grid = sed_pool_bwd_nparts(B, H, W, Cp);
    const size_t lds = (size_t)PPB * 2 * Cp * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const RowGeom geo = row_geom(B, H, W, Cp);
#define ARGS <<<grid, 256, lds, st>>>((const T_*)dy, (const T_*)z, scale, shift, mean, invstd, partial, geo)
    if (dtype == SED_BF16 && pool == 1) { typedef bf16_t T_; pool_relu_bwd_stats_kernel<T_, 1> ARGS; }
    else if (dtype == SED_BF16 && pool == 2) { typedef bf16_t T_; pool_relu_bwd_stats_kernel<T_, 2> ARGS; }
    else if (dtype == SED_F32 && pool == 1) { typedef float T_; pool_relu_bwd_stats_kernel<T_, 1> ARGS; }
    else if (dtype == SED_F32 && pool == 2) { typedef float T_; pool_relu_bwd_stats_kernel<T_, 2> ARGS; }
    else SED_REQUIRE(false, "dtype must be SED_F32/SED_BF16 and pool 1 or 2");
#undef ARGS
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_pool_relu_bwd_stats_if(const int* flag, int dtype, const void* dy, const void* z, const float* scale,
                                          const float* shift, const float* mean, const float* invstd, float* partial,
                                          int nparts, int B, int H, int W, int Cp, int pool, void* stream) {
    SED_REQUIRE(Cp % 32 == 0 && Cp <= 2048, "Cp must be a multiple of 32, <= 2048");
    SED_REQUIRE(flag != nullptr, "flag");
    int G, PPB;
    stats_geometry(Cp, &G, &PPB);
    SED_REQUIRE(nparts >= 1, "partial rows");
    const int own = sed_pool_bwd_nparts(B, H, W, Cp);
    const int grid = own < nparts ? own : nparts;          // (the kernel walks its rows grid-stride: any grid covers them)
    const size_t lds = (size_t)PPB * 2 * Cp * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const RowGeom geo = row_geom(B, H, W, Cp);
#define ARGS <<<grid, 256, lds, st>>>((const T_*)dy, (const T_*)z, scale, shift, mean, invstd, partial, geo, flag, nparts)
    if (dtype == SED_BF16 && pool == 1) { typedef bf16_t T_; pool_relu_bwd_stats_kernel<T_, 1> ARGS; }
    else if (dtype == SED_BF16 && pool == 2) { typedef bf16_t T_; pool_relu_bwd_stats_kernel<T_, 2> ARGS; }
    else if (dtype == SED_F32 && pool == 1) { typedef float T_; pool_relu_bwd_stats_kernel<T_, 1> ARGS; }
    else if (dtype == SED_F32 && pool == 2) { typedef float T_; pool_relu_bwd_stats_kernel<T_, 2> ARGS; }
    else SED_REQUIRE(false, "dtype must be SED_F32/SED_BF16 and pool 1 or 2");
#undef ARGS
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_pool_relu_bn_bwd_apply(int dtype, const void* dy, const void* z, const float* scale,
                                          const float* shift, const float* ca, const float* cb, const float* cc,
                                          void* dz, int B, int H, int W, int Cp, int pool, void* stream) {
    SED_REQUIRE(Cp % 32 == 0, "Cp must be a multiple of 32");
    hipStream_t st = (hipStream_t)stream;
    const RowGeom geo = row_geom(B, H, W, Cp);
    const int grid = row_grid((long long)B * H);
#define ARGS <<<grid, 256, 0, st>>>((const T_*)dy, (const T_*)z, scale, shift, ca, cb, cc, (T_*)dz, geo)
    if (dtype == SED_BF16 && pool == 1) { typedef bf16_t T_; pool_relu_bn_bwd_apply_kernel<T_, 1> ARGS; }
    else if (dtype == SED_BF16 && pool == 2) { typedef bf16_t T_; pool_relu_bn_bwd_apply_kernel<T_, 2> ARGS; }
    else if (dtype == SED_F32 && pool == 1) { typedef float T_; pool_relu_bn_bwd_apply_kernel<T_, 1> ARGS; }
    else if (dtype == SED_F32 && pool == 2) { typedef float T_; pool_relu_bn_bwd_apply_kernel<T_, 2> ARGS; }
    else SED_REQUIRE(false, "dtype must be SED_F32/SED_BF16 and pool 1 or 2");
#undef ARGS
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_bn_bwd_apply(int dtype, const void* g, const void* z, const float* ca, const float* cb,
                                const float* cc, void* dz, size_t npix, int Cp, void* stream) {
    SED_REQUIRE(Cp % 32 == 0, "Cp must be a multiple of 32");
    hipStream_t st = (hipStream_t)stream;
    const int grid = ew_grid(npix * (Cp / 8));
    if (dtype == SED_BF16)
        bn_bwd_apply_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)g, (const bf16_t*)z, ca, cb, cc, (bf16_t*)dz, npix, Cp);
    else if (dtype == SED_F32)
        bn_bwd_apply_kernel<float><<<grid, 256, 0, st>>>((const float*)g, (const float*)z, ca, cb, cc, (float*)dz, npix, Cp);
    else
        SED_REQUIRE(false, "bad dtype");
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_head_fwd(int dtype, const void* feat, const float* fc_w, const float* fc_b, float* m_out,
                            float* pre, int B, int t, int Wf, int C, int Cp, int K, void* stream) {
    SED_REQUIRE(Cp <= 2048 && C <= Cp && K >= 1, "bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int rows = B * t;
    const bool rows16 = Cp == 128 && K <= HF_KMAX && C <= Cp;          // 16 rows per workgroup (the CNN's 128-channel head)
    if (dtype == SED_BF16) {
        if (rows16) head_fwd16_kernel<bf16_t><<<(rows + 15) / 16, 256, 0, st>>>((const bf16_t*)feat, fc_w, fc_b, m_out, pre, rows, Wf, C, K);
        else head_fwd_kernel<bf16_t><<<rows, 128, 0, st>>>((const bf16_t*)feat, fc_w, fc_b, m_out, pre, Wf, C, Cp, K);
    } else if (dtype == SED_F32) {
        if (rows16) head_fwd16_kernel<float><<<(rows + 15) / 16, 256, 0, st>>>((const float*)feat, fc_w, fc_b, m_out, pre, rows, Wf, C, K);
        else head_fwd_kernel<float><<<rows, 128, 0, st>>>((const float*)feat, fc_w, fc_b, m_out, pre, Wf, C, Cp, K);
    } else {
        SED_REQUIRE(false, "bad dtype");
    }
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_interpolate(const float* pre, float* out, int B, int t, int K, int ratio, void* stream) {
    const size_t total = (size_t)B * t * ratio * K;
    interpolate_kernel<<<ew_grid(total), 256, 0, (hipStream_t)stream>>>(pre, out, t, K, ratio, total);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_bce_fwd_bwd(const float* pre, const float* target, float* loss, float* dpre, float* loss_partial,
                               int B, int t, int K, int ratio, int Tt, float recall_factor, float grad_scale,
                               void* stream) {
    SED_REQUIRE(B > 0 && t > 0 && K > 0 && ratio > 0 && Tt > 0, "bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const int N = t * ratio < Tt ? t * ratio : Tt;
    const float inv_numel = (float)(1.0 / ((double)B * N * K));
    const size_t total = (size_t)B * t * K;
    const int nblk = (int)((total + 255) / 256);
    bce_fwd_bwd_kernel<<<nblk, 256, 0, st>>>(pre, target, dpre, loss_partial, B, t, K, ratio, Tt, N, recall_factor,
                                             inv_numel, grad_scale);
    SED_LAUNCH_CHECK();
    loss_finalize_kernel<<<1, 256, 0, st>>>(loss_partial, nblk, inv_numel, loss);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t sed_head_bwd_ws_floats(int B, int t, int C, int K) {
    const size_t nblk = cdivz((size_t)B * t, HEAD_BWD_ROWS);       // (sized for the 8-row form; the 64-row form needs an eighth)
    return nblk * ((size_t)K * C + K);
}

extern "C" int sed_head_bwd(int dtype, const float* dpre, const float* m, const float* fc_w, float* dfc_w,
                            float* dfc_b, void* dfeat, float* workspace, int B, int t, int Wf, int C, int Cp, int K,
                            int ratio, void* stream) {
    SED_REQUIRE(ratio >= 1, "ratio must be >= 1");
    hipStream_t st = (hipStream_t)stream;
    const size_t rows = (size_t)B * t;
    SED_REQUIRE(dtype == SED_BF16 || dtype == SED_F32, "bad dtype");
    const bool big = K <= HB_KMAX && (size_t)K * C <= HB_WCAP && Cp % 8 == 0 && Cp <= 2048 && 256 % (Cp / 8) == 0 && C <= Cp;
    const int nblk = (int)cdivz(rows, big ? HB_ROWS : HEAD_BWD_ROWS);
    if (big) {
        if (dtype == SED_BF16) head_bwd64_kernel<bf16_t><<<nblk, 256, 0, st>>>(dpre, m, fc_w, workspace, (bf16_t*)dfeat, rows, Wf, C, Cp, K, ratio);
        else head_bwd64_kernel<float><<<nblk, 256, 0, st>>>(dpre, m, fc_w, workspace, (float*)dfeat, rows, Wf, C, Cp, K, ratio);
    } else {
        if (dtype == SED_BF16) head_bwd_kernel<bf16_t><<<nblk, 128, 0, st>>>(dpre, m, fc_w, workspace, (bf16_t*)dfeat, rows, Wf, C, Cp, K, ratio);
        else head_bwd_kernel<float><<<nblk, 128, 0, st>>>(dpre, m, fc_w, workspace, (float*)dfeat, rows, Wf, C, Cp, K, ratio);
    }
    SED_LAUNCH_CHECK();
    const size_t stride = (size_t)K * C + K;
    sum_partials2_kernel<<<ew_grid(stride * 64), 256, 0, st>>>(workspace, dfc_w, (size_t)K * C, dfc_b, (size_t)K, nblk, stride);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_sum_partials(const float* partial, int nparts, size_t n, float* out, void* stream) {
    sum_partials_kernel<<<ew_grid(n * 64), 256, 0, (hipStream_t)stream>>>(partial, out, nparts, n, n);
    SED_LAUNCH_CHECK();
    return 0;
}

// ---- the same step with its scalars on the DEVICE, so that a captured HIP graph of the whole train step can be replayed:
// hyper[0] = learning rate, hyper[1] = lr / (1 - beta1^t), hyper[2] = 1 / sqrt(1 - beta2^t); *step = t.  The update kernel
// advances t, refreshes the two bias-correction terms and applies train.py:108-110's decay (lr *= lr_decay after every
// decay_every-th step, effective from the next step) -- one thread, fp64 like the host path.
__global__ void adam_hyper_kernel(float* __restrict__ hyper, int* __restrict__ step, float beta1, float beta2, float lr_decay,
                                  int decay_every) {
    const int s = *step + 1;
    *step = s;
    const double lr = (double)hyper[0];
    const double bc1 = 1.0 - pow((double)beta1, (double)s), bc2 = 1.0 - pow((double)beta2, (double)s);
    hyper[1] = (float)(lr / bc1);
    hyper[2] = (float)(1.0 / sqrt(bc2));
    if (decay_every > 0 && s % decay_every == 0) hyper[0] = (float)(lr * (double)lr_decay);
}

__global__ __launch_bounds__(256) void adam_amsgrad_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                               float* __restrict__ m, float* __restrict__ v,
                                                               float* __restrict__ vmax, size_t n, float one_minus_b1, float b2,
                                                               float one_minus_b2, float eps, const float* __restrict__ hyper,
                                                               float grad_scale) {
    const float step_size = hyper[1], inv_sqrt_bc2 = hyper[2];
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gr = g[i] * grad_scale;
        const float mv = m[i] + one_minus_b1 * (gr - m[i]);
        const float vv = v[i] * b2 + (one_minus_b2 * gr) * gr;
        const float xv = fmaxf(vmax[i], vv);
        const float denom = sqrtf(xv) * inv_sqrt_bc2 + eps;
        p[i] = p[i] - step_size * (mv / denom);
        m[i] = mv; v[i] = vv; vmax[i] = xv;
    }
}

extern "C" int sed_adam_amsgrad_step_dev(float* p, const float* g, float* m, float* v, float* vmax, size_t n, float* hyper,
                                         int* step, float beta1, float beta2, float eps, float grad_scale, float lr_decay,
                                         int decay_every, void* stream) {
    SED_REQUIRE(p && g && m && v && vmax && hyper && step, "null argument");
    hipStream_t st = (hipStream_t)stream;
    adam_hyper_kernel<<<1, 1, 0, st>>>(hyper, step, beta1, beta2, lr_decay, decay_every);
    const float omb1 = (float)(1.0 - (double)beta1), omb2 = (float)(1.0 - (double)beta2);
    adam_amsgrad_dev_kernel<<<ew_grid(n), 256, 0, st>>>(p, g, m, v, vmax, n, omb1, beta2, omb2, eps, hyper, grad_scale);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_adam_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax, size_t n, float lr,
                                     float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
    SED_REQUIRE(step >= 1, "step is 1-based");
    SED_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)vmax) & 15) == 0,
                "flat buffers must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const float omb1 = (float)(1.0 - (double)beta1), omb2 = (float)(1.0 - (double)beta2);
    adam_amsgrad_kernel<<<ew_grid((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(p, g, m, v, vmax, n, omb1, beta2, omb2,
                                                                               eps, step_size, inv_sqrt_bc2, grad_scale);
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_cast(int dtype_dst, void* dst, int dtype_src, const void* src, size_t n, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int grid = ew_grid(n);
    if (dtype_dst == SED_BF16 && dtype_src == SED_F32)
        cast_kernel<bf16_t, float><<<grid, 256, 0, st>>>((bf16_t*)dst, (const float*)src, n);
    else if (dtype_dst == SED_F32 && dtype_src == SED_BF16)
        cast_kernel<float, bf16_t><<<grid, 256, 0, st>>>((float*)dst, (const bf16_t*)src, n);
    else if (dtype_dst == SED_F32 && dtype_src == SED_F32)
        cast_kernel<float, float><<<grid, 256, 0, st>>>((float*)dst, (const float*)src, n);
    else if (dtype_dst == SED_BF16 && dtype_src == SED_BF16)
        cast_kernel<bf16_t, bf16_t><<<grid, 256, 0, st>>>((bf16_t*)dst, (const bf16_t*)src, n);
    else
        SED_REQUIRE(false, "bad dtype");
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_nchw_to_nhwc(int dtype, const float* src, void* dst, int B, int C, int H, int W, int Cp,
                                void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int grid = ew_grid((size_t)B * H * W * Cp);
    if (dtype == SED_BF16) nchw_to_nhwc_kernel<bf16_t><<<grid, 256, 0, st>>>(src, (bf16_t*)dst, B, C, H, W, Cp);
    else if (dtype == SED_F32) nchw_to_nhwc_kernel<float><<<grid, 256, 0, st>>>(src, (float*)dst, B, C, H, W, Cp);
    else SED_REQUIRE(false, "bad dtype");
    SED_LAUNCH_CHECK();
    return 0;
}

extern "C" int sed_nhwc_to_nchw(int dtype, const void* src, float* dst, int B, int C, int H, int W, int Cp,
                                void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int grid = ew_grid((size_t)B * C * H * W);
    if (dtype == SED_BF16) nhwc_to_nchw_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)src, dst, B, C, H, W, Cp);
    else if (dtype == SED_F32) nhwc_to_nchw_kernel<float><<<grid, 256, 0, st>>>((const float*)src, dst, B, C, H, W, Cp);
    else SED_REQUIRE(false, "bad dtype");
    SED_LAUNCH_CHECK();
    return 0;
}
