// Evaluation-side kernels (gfx950): sigmoid + the 21-threshold recall/precision counting of
// calculate_metrics / compute_recall_precision (reference utils/metric_utils.py:4-37), done on the
// device so a whole-recording eval never copies frame probabilities to the host (train.py:12-74).
//
// Integer work: the counts are exact and order independent (integer LDS/global adds only).
#include "common.h"

#define METRIC_MAX_TH 64
#define METRIC_BLOCKS 256

struct ThresholdList {
    double th[METRIC_MAX_TH];
    int n;
};

// Thresholds are ascending, so "O > th_j" holds for exactly the first idx(p) thresholds; every element
// therefore lands in ONE histogram bin idx = #{j : p > th_j} in [0, nth]:
//   positives_j = #elements with idx > j
//   TP_j        = #{(2T-1)==1, idx > j} + #{(2T-0)==1, idx <= j}        (the reference's ((2T-O)==1) test)
// hist layout per block: [3][nth+1] = all, t_one (2t-1==1), t_half (2t==1)
__global__ __launch_bounds__(256) void metric_hist_kernel(const float* __restrict__ logits,
                                                          const float* __restrict__ target,
                                                          float* __restrict__ prob_out, ThresholdList tl, int raw_logits,
                                                          size_t total, unsigned int* __restrict__ hist_part,
                                                          double* __restrict__ gt_part) {
    __shared__ unsigned int h[3 * (METRIC_MAX_TH + 1)];
    __shared__ double sm[256];
    const int nb = tl.n + 1;
    for (int i = threadIdx.x; i < 3 * nb; i += 256) h[i] = 0u;
    __syncthreads();
    double gt = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float x = logits[i];
        const float p = raw_logits ? 1.0f / (1.0f + expf(-x)) : x;
        if (prob_out) prob_out[i] = p;
        const float t = target[i];
        gt += (double)t;
        const double pd = (double)p;        // numpy compares the fp32 probability with the fp64 threshold
        int idx = 0;
        for (int j = 0; j < tl.n; ++j) idx += (pd > tl.th[j]) ? 1 : 0;
        atomicAdd(&h[idx], 1u);
        if (2.0f * t - 1.0f == 1.0f) atomicAdd(&h[nb + idx], 1u);
        if (2.0f * t == 1.0f) atomicAdd(&h[2 * nb + idx], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * nb; i += 256) hist_part[(size_t)blockIdx.x * 3 * nb + i] = h[i];
    // fixed-shape tree: deterministic for a given (total, grid)
    sm[threadIdx.x] = gt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) gt_part[blockIdx.x] = sm[0];
}

__global__ __launch_bounds__(256) void metric_finalize_kernel(const unsigned int* __restrict__ hist_part,
                                                              const double* __restrict__ gt_part, int nblk, int nth,
                                                              unsigned long long* __restrict__ counts,
                                                              double* __restrict__ gt_sum) {
    __shared__ unsigned long long h[3 * (METRIC_MAX_TH + 1)];
    const int nb = nth + 1;
    for (int i = threadIdx.x; i < 3 * nb; i += 256) {
        unsigned long long s = 0;
        for (int b = 0; b < nblk; ++b) s += hist_part[(size_t)b * 3 * nb + i];
        h[i] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < nth) {
        const int j = threadIdx.x;
        unsigned long long pos = 0, tp = 0;
        for (int i = 0; i < nb; ++i) {
            if (i > j) { pos += h[i]; tp += h[nb + i]; }
            else tp += h[2 * nb + i];
        }
        counts[2 * j + 0] = tp;
        counts[2 * j + 1] = pos;
    }
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < nblk; ++b) s += gt_part[b];
        gt_sum[0] = s;
    }
}

extern "C" size_t sed_metric_counts_ws_bytes(int nth) {
    if (nth < 1 || nth > METRIC_MAX_TH) return 0;
    return (size_t)METRIC_BLOCKS * (3 * (size_t)(nth + 1) * sizeof(unsigned int) + sizeof(double));
}

extern "C" int sed_metric_counts(const float* output, const float* target, float* prob_out, const double* thresholds,
                                 int nth, int raw_logits, unsigned long long* counts, double* gt_sum, void* workspace,
                                 size_t n_out, size_t n_tgt, int K, void* stream) {
    SED_REQUIRE(nth >= 1 && nth <= METRIC_MAX_TH, "1..64 thresholds");
    SED_REQUIRE(K >= 1 && thresholds && counts && gt_sum && workspace, "bad arguments");
    ThresholdList tl;
    tl.n = nth;
    for (int j = 0; j < nth; ++j) {
        tl.th[j] = thresholds[j];
        SED_REQUIRE(j == 0 || thresholds[j] >= thresholds[j - 1], "thresholds must be ascending");
    }
    for (int j = nth; j < METRIC_MAX_TH; ++j) tl.th[j] = 0.0;
    const size_t N = n_out < n_tgt ? n_out : n_tgt;      // calculate_metrics: N = min(frames) (metric_utils.py:6)
    const size_t total = N * (size_t)K;
    hipStream_t st = (hipStream_t)stream;
    int nblk = (int)((total + 255) / 256);
    if (nblk > METRIC_BLOCKS) nblk = METRIC_BLOCKS;
    if (nblk < 1) nblk = 1;
    double* gt_part = reinterpret_cast<double*>(workspace);
    unsigned int* hist_part = reinterpret_cast<unsigned int*>(gt_part + METRIC_BLOCKS);
    metric_hist_kernel<<<nblk, 256, 0, st>>>(output, target, prob_out, tl, raw_logits, total, hist_part, gt_part);
    SED_LAUNCH_CHECK();
    metric_finalize_kernel<<<1, 256, 0, st>>>(hist_part, gt_part, nblk, nth, counts, gt_sum);
    SED_LAUNCH_CHECK();
    return 0;
}
