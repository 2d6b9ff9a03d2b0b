/*
 * sed_hip.h -- C ABI of libsed_hip.so: the MI355X (gfx950) kernels of the sound-event-detection
 * training hot path.
 *
 * The reference (ariel415el/SoundEventDetection-Pytorch) has no FFI: its hot path is Python calling
 * torch/ATen/librosa.  Each entry point below therefore names the reference *Python* call it
 * replaces (file:line under the reference root); the host-side mirror of the reference API
 * (soundeventdetection-pytorch_amd/) binds these with ctypes -- see INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless named h_*;
 *   - no ownership transfer: every buffer (workspaces included) is allocated by the caller;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises,
 *     so every call is hipGraph-capturable;
 *   - return value 0 = ok, non-zero = error (sed_last_error() gives the text; the Python side
 *     raises RuntimeError);
 *   - activations are NHWC = [B][H = time frames][W = mel bins][C], C padded to a multiple of 32
 *     (padded channels are exactly 0), element type `dtype` (SED_F32 or SED_BF16); statistics,
 *     parameters, gradients and optimizer state are always fp32;
 *   - "wpack" is a conv weight re-laid for the MFMA A-operand by sed_pack_conv_weight().
 */
#ifndef SED_HIP_H
#define SED_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SED_ABI_VERSION 1

/* SED_F32X3 / SED_F32H3 (round 6, "bf16x3" / "f16x3"): fp32 tensors in memory like SED_F32, but the GEMM-shaped kernels
 * (sed_conv3x3_fwd, sed_conv3x3_wgrad*, with operators packed by sed_pack_conv_weight(s_batch) under the same dtype: two 16-bit
 * images, hi then lo, in the wpack buffer of 9*Cinp*Coutp fp32 words) split every operand a = hi + lo/LS and run three 16-bit MFMAs
 * per product with fp32 accumulation (csrc/sed_conv_x3.hip).  X3: bf16 pieces, LS = 1, ~1e-5 relative per product.  H3: fp16 pieces,
 * LS = 2^11, ~5e-7 per product; because fp16 has five exponent bits, a call whose streamed operand is a GRADIENT (the data gradient's
 * x = dz, the weight gradient's dz) carries a signed power-of-two exponent e in bits 8..15 of the dtype argument
 * (dtype = SED_F32H3 | ((e & 0xff) << 8)): the operand is multiplied by 2^e before the split (and clamped to +-60000), the result by
 * 2^-e; e ~ log2(B*H*W) - 4 brings per-pixel loss gradients to O(1).  Every other entry point takes SED_F32 for such tensors.        */
enum { SED_F32 = 0, SED_BF16 = 1, SED_F32X3 = 2, SED_F32H3 = 3 };

/* prologue applied to the conv input while it is staged into LDS */
enum {
    SED_PRO_NONE = 0,   /* x as stored                                                          */
    SED_PRO_BNRELU = 1  /* relu(scale[c]*x + shift[c]): F.relu_(bn(.)) spectogram_models.py:155  */
};
/* epilogue applied to the conv accumulator */
enum {
    SED_EPI_STORE = 0,    /* store only                                                         */
    SED_EPI_STATS = 1,    /* store + per-channel sum / sum-of-squares partials (BatchNorm stats) */
    SED_EPI_RELUBWD = 2,  /* g = acc * (scale[c]*zref+shift[c] > 0); store g; partials of
                             sum(g), sum(g*xhat), xhat = (zref-mean[c])*invstd[c]               */
    SED_EPI_POOLSTATS = 4 /* sed_conv3x3_bwd_fused only: store + the pooled-tensor statistics of
                             sed_conv3x3_dgrad_poolstats (zref = pooled activation, cnt)         */
};

int sed_abi_version(void);
const char* sed_last_error(void);
/* SED_* tuning / A-B knobs are read from the environment once per name and cached inside the library; this drops the
 * cache (test hook: lets one process flip a knob between two calls).  No reference counterpart.                      */
void sed_config_reload(void);
/* build flags of this library: bit 0 = EXPERIMENTS (opt-in resident-weight kernels, SED_CONV_KERNEL=r/4, extra loader-wave
 * variants), bit 1 = DEBUG_SWITCHES (SED_DBG run-time ablation switches), bit 2 = STAMPS.  The product build returns 0. */
int sed_build_flags(void);
/* number of compute units of the current device (grid sizing on the host side) */
int sed_device_cu_count(void);

/* ---- weights ------------------------------------------------------------------------------
 * torch layout W[Cout][Cin][3][3] fp32 (nn.Conv2d weight, spectogram_models.py:132-140) ->
 * wpack[Cinp/32][9][32/KR][Coutp][KR], KR = 8 (bf16) or 1 (f32), zero padded to Cinp/Coutp.
 * transpose_flip != 0 packs the data-gradient operator W'[c][o][i][j] = W[o][c][2-i][2-j]
 * (then the packed "Cin" is the conv's Cout and vice versa).                                  */
int sed_pack_conv_weight(int dtype, const float* w, void* wpack, int Cout, int Cin, int Coutp,
                         int Cinp, int transpose_flip, void* stream);
/* inverse for gradients: dwpack fp32 [9][Cinp][Coutp] -> dW[Cout][Cin][3][3] fp32              */
int sed_unpack_conv_wgrad(const float* dwpack, float* dw, int Cout, int Cin, int Coutp, int Cinp,
                          void* stream);

/* ---- conv 3x3 s1 p1, no bias (nn.Conv2d forward, spectogram_models.py:155-156) -------------
 * First layer, Cin = 1: x fp32 [B][H][W] (optionally z-scored on load with mean/std[W],
 * SpectogramDataset.transform spectograms_dataset.py:104-108; pass NULL to skip), w fp32
 * [Cout][1][3][3] torch layout, z [B][H][W][Coutp].  stats_partial fp32 [nparts][2][Coutp]
 * receives per-workgroup (sum, sumsq); *nparts is fixed by sed_conv_c1_nparts().               */
int sed_conv_c1_nparts(int B, int H, int W);
int sed_conv3x3_c1_fwd(int dtype, const float* x, const float* mean, const float* std,
                       const float* w, void* z, float* stats_partial, int B, int H, int W,
                       int Cout, int Coutp, void* stream);
/* dW[Cout][1][3][3] for the first layer: dz [B][H][W][Coutp]; dw_partial fp32 [nparts][9][Coutp] */
int sed_conv3x3_c1_wgrad(int dtype, const float* x, const float* mean, const float* std,
                         const void* dz, float* dw_partial, int B, int H, int W, int Coutp,
                         void* stream);

/* Same with the layer's dz produced on load: dz = ca*g + cb*zsrc + cc (BatchNorm backward of the first
 * layer folded in; g = data-gradient epilogue output, zsrc = the layer's pre-BN output).  Block 0 has
 * no data gradient, so its dz never has to be written to memory.                                */
int sed_conv3x3_c1_wgrad_fused(int dtype, const float* x, const float* mean, const float* std,
                               const void* g, const void* zsrc, const float* ca, const float* cb,
                               const float* cc, float* dw_partial, int B, int H, int W, int Coutp,
                               void* stream);

/* Generic layer (Cinp, Coutp multiples of 32; W a power of two, 4..64): implicit GEMM on MFMA.
 * Serves forward (wpack of W) and data-gradient (wpack of W', transpose_flip).
 *   x [B][H][W][Cinp]; pro_scale/pro_shift fp32 [Cinp] (SED_PRO_BNRELU);
 *   z [B][H][W][Coutp];
 *   SED_EPI_STATS: partial fp32 [nparts][2][Coutp] = (sum z, sum z^2) from the fp32 accumulator;
 *   SED_EPI_RELUBWD: zref [B][H][W][Coutp], epi_scale/epi_shift/epi_mean/epi_invstd fp32 [Coutp];
 *                    partial = (sum g, sum g*xhat).
 * nparts = sed_conv_nparts(B, H, W).                                                           */
int sed_conv_nparts(int B, int H, int W);
int sed_conv3x3_fwd(int dtype, int pro, int epi, const void* x, const float* pro_scale,
                    const float* pro_shift, const void* wpack, void* z, const void* zref,
                    const float* epi_scale, const float* epi_shift, const float* epi_mean,
                    const float* epi_invstd, float* partial, int B, int H, int W, int Cinp,
                    int Coutp, void* stream);

/* The same call when the 3x3 weights have ZERO SIDE COLUMNS (taps 0,2,3,5,6,8): a k = 3 Conv1d over frames interleaved on
 * the W axis (the M5 layout, sed_m5_*).  The producer/consumer kernel then contracts taps 1, 4, 7 only (W = 8, bf16); every
 * other path computes all nine taps -- same result.                                                                */
int sed_conv3x3_fwd_col(int dtype, int pro, int epi, const void* x, const float* pro_scale,
                        const float* pro_shift, const void* wpack, void* z, const void* zref,
                        const float* epi_scale, const float* epi_shift, const float* epi_mean,
                        const float* epi_invstd, float* partial, int B, int H, int W, int Cinp,
                        int Coutp, void* stream);

/* Weight gradient of the generic layer: dwpack fp32 [9][Cinp][Coutp] (overwritten) =
 * sum_{b,h,w} a[b][h+i-1][w+j-1][c] * dz[b][h][w][o], a = pro(x).  workspace fp32 of
 * sed_conv_wgrad_ws_floats() floats.                                                          */
size_t sed_conv_wgrad_ws_floats(int B, int H, int W, int Cinp, int Coutp);
int sed_conv3x3_wgrad(int dtype, int pro, const void* x, const float* pro_scale,
                      const float* pro_shift, const void* dz, float* dwpack, float* workspace,
                      int B, int H, int W, int Cinp, int Coutp, void* stream);

/* Same weight gradient with the layer's dz PRODUCED on load (fused BatchNorm/ReLU/avg-pool backward,
 * i.e. the autograd nodes between two convolutions of ConvBlock.forward, spectogram_models.py:155-158):
 *   SED_DZ_POOL: dz = ca*g + cb*z + cc with g = up(gsrc)/pool^2 * [scale*z + shift > 0];
 *                gsrc = gradient w.r.t. the pooled block output [B][H/pool][W/pool][Coutp], zsrc = z2
 *   SED_DZ_BN  : dz = ca*gsrc + cb*zsrc + cc; gsrc = data-gradient epilogue output g, zsrc = z1
 * (ca, cb, cc from sed_bn_bwd_finalize).  If dz_out != NULL the produced dz [B][H][W][Coutp] is
 * also written there (consumed by the data-gradient call); it must not alias gsrc/zsrc.           */
enum { SED_DZ_POOL = 1, SED_DZ_BN = 2 };
int sed_conv3x3_wgrad_fused(int dtype, int pro, const void* x, const float* pro_scale,
                            const float* pro_shift, int dzmode, const void* gsrc, const void* zsrc,
                            const float* scale, const float* shift, const float* ca, const float* cb,
                            const float* cc, int pool, void* dz_out, float* dwpack, float* workspace,
                            int B, int H, int W, int Cinp, int Coutp, void* stream);

/* ---- BatchNorm2d (spectogram_models.py:142-143,155-156; eps 1e-5, momentum 0.1) ------------
 * Training statistics from the conv epilogue partials [nparts][2][Cp]: batch mean, biased var ->
 * scale = gamma*invstd, shift = beta - mean*scale, saved mean/invstd, and the running-stat
 * update (unbiased var, n/(n-1)).  gamma/beta/running_* have C (unpadded) entries; outputs Cp
 * entries with padded channels forced to scale = shift = 0.  count = B*H*W.                    */
int sed_bn_train_finalize(const float* partial, int nparts, double count, const float* gamma,
                          const float* beta, float* running_mean, float* running_var,
                          float momentum, float eps, float* scale, float* shift, float* mean,
                          float* invstd, int C, int Cp, void* stream);
/* Eval mode: scale/shift from the running statistics.                                         */
int sed_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, int C,
                       int Cp, void* stream);
/* Backward reduction finalize: partial [nparts][2][Cp] = (sum g, sum g*xhat) ->
 * dgamma[C], dbeta[C] and the coefficients of dz = ca*g + cb*z + cc (fp32 [Cp] each):
 *   ca = gamma*invstd, cb = -gamma*invstd^2*mgx, cc = -gamma*invstd*(mg - mean*invstd*mgx).   */
int sed_bn_bwd_finalize(const float* partial, int nparts, double count, const float* gamma,
                        const float* mean, const float* invstd, float* dgamma, float* dbeta,
                        float* ca, float* cb, float* cc, int C, int Cp, void* stream);

/* ---- fused elementwise stages --------------------------------------------------------------
 * y = avg_pool2d(relu(scale*z+shift), pool) (spectogram_models.py:156-158), pool in {1,2};
 * z [B][H][W][Cp] -> y [B][H/pool][W/pool][Cp].                                                */
int sed_bn_relu_pool_fwd(int dtype, const void* z, const float* scale, const float* shift, void* y,
                         int B, int H, int W, int Cp, int pool, void* stream);
/* backward, pass 1: g = up(dy)/pool^2 * (scale*z+shift > 0); partial [nparts][2][Cp] =
 * (sum g, sum g*xhat); nparts = sed_pool_bwd_nparts().  dy [B][H/pool][W/pool][Cp].           */
int sed_pool_bwd_nparts(int B, int H, int W, int Cp);
int sed_pool_relu_bwd_stats(int dtype, const void* dy, const void* z, const float* scale,
                            const float* shift, const float* mean, const float* invstd,
                            float* partial, int B, int H, int W, int Cp, int pool, void* stream);
/* The same statistics WITHOUT reading z (pool = 2, bf16): per pooled pixel P the forward also stores
 * cnt(P) = number of its four pixels with scale*z+shift > 0 (uint8 [B][H/2][W/2][Cp]); then
 *   sum g      = sum_P dy(P) * cnt(P) / 4
 *   sum g*xhat = sum_P dy(P) * (y(P) - beta*cnt(P)/4) / gamma      (y = the pooled activation the forward stored,
 *                                                                   gamma*xhat + beta = scale*z + shift)
 * are functions of pooled tensors only, and the data-gradient kernel that PRODUCES dy accumulates them in its epilogue
 * (sed_conv3x3_dgrad_poolstats: no separate pass, no second read of the full-resolution z).  A channel with gamma = 0
 * cannot be recovered that way (0/0): the kernel then raises *flag and sed_pool_relu_bwd_stats_if recomputes the
 * partials from z (it returns at once while *flag is 0).  partial [nparts][2][Cp], nparts >= sed_conv_nparts(B, H, W) of
 * the data-gradient launch; the conditional pass uses min(nparts, its own row count) workgroups and zero-fills the rest.  */
int sed_bn_relu_pool_cnt_fwd(int dtype, const void* z, const float* scale, const float* shift, void* y,
                             void* cnt, int B, int H, int W, int Cp, void* stream);
int sed_dgrad_poolstats_supported(int dtype, int W, int Cinp, int Coutp);
int sed_conv3x3_dgrad_poolstats(int dtype, const void* dz, const void* wpack_t, void* dy, const void* y_pooled,
                                const void* cnt, const float* scale, const float* shift, const float* mean,
                                const float* invstd, float* partial, int nparts, int* flag, int B, int H, int W,
                                int Cinp, int Coutp, void* stream);
int sed_pool_relu_bwd_stats_if(const int* flag, int dtype, const void* dy, const void* z, const float* scale,
                               const float* shift, const float* mean, const float* invstd, float* partial,
                               int nparts, int B, int H, int W, int Cp, int pool, void* stream);
/* backward, pass 2: dz = ca*g + cb*z + cc with g recomputed as in pass 1.                      */
int sed_pool_relu_bn_bwd_apply(int dtype, const void* dy, const void* z, const float* scale,
                               const float* shift, const float* ca, const float* cb,
                               const float* cc, void* dz, int B, int H, int W, int Cp, int pool,
                               void* stream);
/* dz = ca*g + cb*z + cc for a materialised g (data-gradient epilogue output); in place allowed. */
int sed_bn_bwd_apply(int dtype, const void* g, const void* z, const float* ca, const float* cb,
                     const float* cc, void* dz, size_t npix, int Cp, void* stream);

/* ---- head + loss ---------------------------------------------------------------------------
 * Cnn_AvgPooling.forward tail (spectogram_models.py:193-200): mean over mel (W) -> Linear(C,K)
 * -> raw logits; the x`ratio` interpolate() is NOT materialised here: pre [B][t][K] fp32.
 * feat [B][t][Wf][Cp] (post BN/ReLU/pool activations); fc_w [K][C], fc_b [K] fp32;
 * m_out [B][t][Cp] fp32 keeps the mel-mean for the backward pass.                              */
int sed_head_fwd(int dtype, const void* feat, const float* fc_w, const float* fc_b, float* m_out,
                 float* pre, int B, int t, int Wf, int C, int Cp, int K, void* stream);
/* interpolate(): out[b][i][k] = pre[b][i/ratio][k] (spectogram_models.py:9-22)                 */
int sed_interpolate(const float* pre, float* out, int B, int t, int K, int ratio, void* stream);
/* WeightedBCE.__call__ (utils/common.py:16-30) on the interpolated logits without materialising
 * them: N = min(t*ratio, Tt) frames; loss[0] = mean over B*N*K of
 * -(w*y*logsigmoid(x) + (1-y)*logsigmoid(-x)); dpre [B][t][K] = d loss / d pre (the x`ratio`
 * repeat backward already summed). target [B][Tt][K] fp32. loss_partial: fp32 scratch of at
 * least ceil(B*t*K/256) floats. dpre may be NULL (loss only).                                   */
int sed_bce_fwd_bwd(const float* pre, const float* target, float* loss, float* dpre,
                    float* loss_partial, int B, int t, int K, int ratio, int Tt, float recall_factor,
                    float grad_scale, void* stream);
/* Head backward: dfc_w [K][C], dfc_b [K] (overwritten) and dfeat [B][t][Wf][Cp] =
 * (dpre @ fc_w)/Wf broadcast over mel.  `dpre` is [B][t*ratio][K]: with ratio > 1 it is the
 * gradient w.r.t. the interpolate()d logits and the repeat-backward sum is folded in.           */
size_t sed_head_bwd_ws_floats(int B, int t, int C, int K);
int sed_head_bwd(int dtype, const float* dpre, const float* m, const float* fc_w, float* dfc_w,
                 float* dfc_b, void* dfeat, float* workspace, int B, int t, int Wf, int C, int Cp, int K,
                 int ratio, void* stream);

/* ---- optimizer -----------------------------------------------------------------------------
 * torch.optim.Adam(amsgrad=True, weight_decay=0) step (train.py:85,103) on flat fp32 buffers of
 * n elements; `step` is the 1-based step count; grad_scale multiplies g first (1/world_size).   */
int sed_adam_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax, size_t n,
                          float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                          void* stream);
/* The same update with its per-step scalars in device memory (for replaying a captured HIP graph of the whole train step):
 * hyper fp32 [3] = {lr, lr/(1-beta1^t), 1/sqrt(1-beta2^t)} (set hyper[0] = lr once), *step = t (starts at 0).  Each call
 * advances t on the device, refreshes the bias corrections and applies lr *= lr_decay after every decay_every-th step
 * (train.py:108-110; decay_every = 0 disables).                                                                   */
int sed_adam_amsgrad_step_dev(float* p, const float* g, float* m, float* v, float* vmax, size_t n,
                              float* hyper, int* step, float beta1, float beta2, float eps,
                              float grad_scale, float lr_decay, int decay_every, void* stream);

/* ---- log-mel front-end ---------------------------------------------------------------------
 * multichannel_stft + multichannel_complex_to_log_mel (+ transform) for one channel batch
 * (dataset/spectogram/preprocess.py:21-45, spectograms_dataset.py:104-108):
 * wave fp32 [B][samples] -> out fp32 [B][T][n_mels], T = 1 + samples/hop; window fp32 [nfft]
 * (already zero-padded/centred); melT fp32 [n_mels][nfft/2+1] (MEL_FILTER_BANK_MATRIX^T);
 * mel_lo/mel_hi int32 [n_mels] = first/last+1 non-zero bin of each filter; mean/std [n_mels] or
 * NULL.  nfft a power of two in [64, 32768].  workspace: sed_logmel_ws_bytes() -- always required, used for the
 * FFT twiddle table only while the stream is being captured or beyond four transform sizes per device; otherwise
 * the library keeps one table per (device, nfft) of its own: the FIRST call for a size allocates it, builds it
 * on `stream` and synchronises that stream once (round 5: no per-call table launch).                             */
size_t sed_logmel_ws_bytes(int B, int samples, int nfft, int hop);
int sed_logmel_fwd(const float* wave, const float* window, const float* melT, const int* mel_lo,
                   const int* mel_hi, const float* mean, const float* std, float* out, void* workspace,
                   int B, int samples, int nfft, int hop, int n_mels, void* stream);
/* multichannel_stft only: spec float2 [B][T][nfft/2+1] (complex64)                             */
int sed_stft_fwd(const float* wave, const float* window, void* spec, void* workspace, int B,
                 int samples, int nfft, int hop, void* stream);

/* multichannel_complex_to_log_mel on an existing spectrogram (preprocess.py:39-45; the in-loop
 * "Complex" mode of spectograms_dataset.py:104-110): spec float2 [nframes][bins] ->
 * out fp32 [nframes][n_mels] = 10*log10(max(1e-10, |spec|^2 . mel)) (optionally z-scored).       */
int sed_complex_to_logmel(const void* spec, const float* melT, const int* mel_lo, const int* mel_hi,
                          const float* mean, const float* std, float* out, size_t nframes, int bins,
                          int n_mels, void* stream);

/* In-loop "Complex" mode of SpectogramDataset (spectograms_dataset.py:58-78, 104-135) on a
 * spectrogram bank resident in HBM: bank float2 [bank_frames][bins] = all training STFTs
 * concatenated on the frame axis (:163).  For sample b: nmix[b] in 1..4 crops starting at frames
 * starts[b][0..nmix[b]-1] are averaged (augment_mix_samples :120-135; the label max is host
 * logic), real Gaussian noise of std noise_std[b] is added when > 0 (augment_add_noise :112-118),
 * the result is z-scored with the COMPLEX mean / real std per bin (transform :105; NULL = skip)
 * and converted with multichannel_complex_to_log_mel (preprocess.py:39-45).
 * noise (nullable) [B][crop][bins] standard-normal draws; NULL = generated in the kernel from
 * (seed, element index) with the counter-based generator restated in oracle/dataset_oracle.py.
 * starts_host/nmix_host are HOST copies of the device tables starts [B][4] / nmix [B]: they are
 * validated against bank_frames before anything is launched.  out fp32 [B][crop][n_mels].        */
int sed_complex_augment_logmel(const void* bank, size_t bank_frames, const int* starts_host,
                               const int* nmix_host, const int* starts, const int* nmix,
                               const float* noise_std, const float* noise, unsigned long long seed,
                               const void* cmean, const float* cstd, const float* melT,
                               const int* mel_lo, const int* mel_hi, float* out, int B, int crop,
                               int bins, int n_mels, void* stream);

/* "logMel" mode of SpectogramDataset.__getitem__ + transform (spectograms_dataset.py:66-69,
 * 104-108) for a batch: out[b][t][m] = (bank[starts[b]+t][m] - mean[m]) / std[m]; bank fp32
 * [bank_frames][n_mels] resident in HBM.  starts_host = host copy of starts (validated first).  */
int sed_logmel_crops(const float* bank, size_t bank_frames, const int* starts_host,
                     const int* starts, const float* mean, const float* std, float* out, int B,
                     int crop, int n_mels, void* stream);

/* ---- CRNN head: mean over mel -> bidirectional GRU -> Linear (BASELINE.json configs[3]) -------
 * Not in the reference repository; semantics are torch.nn.GRU's
 * (batch_first, bidirectional, gate order r,z,n; h' = (1-z) n + z h).  The FC on the GRU output
 * reuses sed_head_fwd / sed_head_bwd with dtype SED_F32, Wf = 1, C = Cp = 2*Hd.
 *
 * mean over the mel axis (spectogram_models.py:193): feat [rows][Wf][Cp] (dtype) -> m [rows][C]
 * fp32, and its backward dm -> dfeat (padding channels written as 0).                           */
int sed_mel_mean_fwd(int dtype, const void* feat, float* m, size_t rows, int Wf, int C, int Cp,
                     void* stream);
int sed_mel_mean_bwd(int dtype, const float* dm, void* dfeat, size_t rows, int Wf, int C, int Cp,
                     void* stream);
/* C[M][N] = A[M][K] . B[N][K]^T (+ bias[N]); row-major fp32 in memory, MFMA compute in
 * compute_dtype (SED_BF16: bf16 operands, fp32 accumulate; SED_F32: fp32 MFMA).  ksplit > 1 splits
 * K over workgroups (fixed-order reduction through `workspace`, sed_gemm_nt_ws_floats floats; no
 * bias then).  lda/ldb multiples of 4, A/B 16-byte aligned.                                      */
size_t sed_gemm_nt_ws_floats(int M, int N, int ksplit);
int sed_gemm_nt(int compute_dtype, const float* A, int lda, const float* B, int ldb,
                const float* bias, float* C, int ldc, int M, int N, int K, int ksplit,
                float* workspace, void* stream);
/* dst[c][r] = src[r - shift][c] for rows grouped in sequences of `seq` rows (0 where r - shift
 * leaves the sequence); shift in {-1, 0, +1}.  shift = +1 / -1 builds the "previous hidden state"
 * matrix of a forward / reverse recurrence, already transposed for the weight-gradient GEMM.     */
int sed_transpose_shift(const float* src, int ld_src, float* dst, int ld_dst, int R, int C, int seq,
                        int shift, void* stream);
/* Round 6: C[M][N] = sum_k A[k][m] . B[k - shift][n] -- both operands row-major fp32 with the reduction index k on the ROWS (A [K][lda],
 * B [K][ldb]): the weight-gradient products of the recurrence without transposing anything.  B's rows are shifted by `shift` in
 * {-1, 0, +1} inside sequences of `seq` consecutive rows (K % seq == 0; a row that leaves its sequence contributes zero): shift = +1 /
 * -1 pairs a gate gradient with the PREVIOUS hidden state of a forward / reverse recurrence.  colsum (nullable) [M] = sum_k A[k][m] (the
 * bias gradients).  ksplit > 1 splits K over workgroups, fixed-order reduction through `workspace` (sed_gemm_tn_ws_floats floats).
 * lda/ldb multiples of 4, A/B 16-byte aligned.  Replaces sed_transpose_shift x 5 + sed_gemm_nt x 4 + sed_row_sums x 4 of the BPTT tail.  */
size_t sed_gemm_tn_ws_floats(int M, int N, int ksplit);
int sed_gemm_tn(int compute_dtype, const float* A, int lda, const float* B, int ldb, float* C, int ldc, float* colsum, int M, int N,
                int K, int seq, int shift, int ksplit, float* workspace, void* stream);
/* The same product for up to 8 independent problems in ONE launch plus ONE reduction launch (each problem with its OWN workspace): the
 * four weight-gradient products of a bidirectional layer's BPTT tail (dW_ih, dW_hh and their biases, two directions) -- 12 launches of
 * 2-3 GFLOP each as four sed_gemm_tn calls.  Fields as sed_gemm_tn's arguments.                                                       */
typedef struct sed_gemm_tn_desc {
    const float* A; const float* B; float* C; float* colsum; float* workspace;
    int lda, ldb, ldc, M, N, K, seq, shift, ksplit;
} sed_gemm_tn_desc;
int sed_gemm_tn_batch(int compute_dtype, const sed_gemm_tn_desc* problems, int n, void* stream);
/* out[r] = sum_c src[r][c] (bias gradients from the transposed gate gradients)                   */
int sed_row_sums(const float* src, int ld, float* out, int R, int C, void* stream);
/* Recurrent weights weight_hh_l0 / weight_hh_l0_reverse ([3Hd][Hd] fp32) -> MFMA-fragment order in
 * `dtype`, for the forward recurrence (pack_fwd) and for its transpose product in BPTT (pack_bwd);
 * each buffer holds sed_gru_pack_elems(Hd) elements.  Hd: multiple of 32, <= 256.                */
size_t sed_gru_pack_elems(int Hd);
int sed_gru_pack_weights(int dtype, const float* whh_fwd, const float* whh_rev, void* pack_fwd,
                         void* pack_bwd, int Hd, void* stream);
/* Forward recurrence of both directions.  gi [B*t][2][3Hd] = x.W_ih^T + b_ih (sed_gemm_nt);
 * bhh [2][3Hd]; hseq [B*t][2][Hd] (= nn.GRU output (B, t, 2Hd)); saved (nullable for inference)
 * [B*t][2][4][Hd] = r, z, n, W_hn h + b_hn for the backward pass.  h0 = 0.                       */
int sed_gru_seq_fwd(int dtype, const float* gi, const float* bhh, const void* pack_fwd, float* hseq,
                    float* saved, int B, int t, int Hd, void* stream);
/* BPTT: dhseq [B*t][2][Hd] -> dgi, dgh [B*t][2][3Hd] (gradients w.r.t. the input-side and the
 * hidden-side gate pre-activations).  The weight/bias/input gradients follow as plain GEMMs.     */
int sed_gru_seq_bwd(int dtype, const float* dhseq, const float* hseq, const float* saved,
                    const void* pack_bwd, float* dgi, float* dgh, int B, int t, int Hd, void* stream);

/* ---- evaluation ----------------------------------------------------------------------------
 * calculate_metrics / compute_recall_precision (utils/metric_utils.py:4-37) without leaving the
 * device: output [n_out][K], target [n_tgt][K] fp32, N = min(n_out, n_tgt) frames are scored.
 * raw_logits != 0: output holds raw logits and torch.sigmoid (train.py:43) is applied first;
 * prob_out (nullable) [N][K] receives the probabilities.  thresholds: HOST array of nth (<= 64)
 * ascending fp64 values (np.arange(0, 1.05, 0.05) in the reference); the decision is the
 * reference's strict  (double)p > th.  counts (device) [nth][2] uint64 = {TP, positives} with
 * TP = #((2T - O) == 1); gt_sum (device) [1] fp64 = T.sum().  workspace: device scratch of
 * sed_metric_counts_ws_bytes(nth) bytes, 8-byte aligned.                                        */
size_t sed_metric_counts_ws_bytes(int nth);
int sed_metric_counts(const float* output, const float* target, float* prob_out,
                      const double* thresholds, int nth, int raw_logits,
                      unsigned long long* counts, double* gt_sum, void* workspace, size_t n_out,
                      size_t n_tgt, int K, void* stream);

/* First-layer weight gradient WITHOUT the layer's pre-BN output (z1 is never read):
 *   dW1[c][k] = ca[c]*A[c][k] + cb[c]*sum_j w1[c][j]*G[j][k] + cc[c]*sx[k],
 * A = plain sed_conv3x3_c1_wgrad of g (summed partials, [9][Coutp]); G / sx = Gram matrix and sums of the
 * 3x3 input patches: sed_conv3x3_c1_gram -> gram_partial fp32 [sed_conv_c1_gram_nparts][54] (45 upper-triangle
 * products then 9 sums); the combine reduces them in fp64 and writes dwpack [9][Coutp].                */
/* Launch-count diet of the train step (same arithmetic, fewer dispatches):
 *  - sed_pack_conv_weights_batch packs every conv layer of a step in ONE launch: desc = device array of n descriptors of
 *    eight 64-bit words {w, wpack, Cout, Cin, POp, PIp, transpose_flip, first_block} (POp/PIp = packed-out / packed-in
 *    padded channels as in sed_pack_conv_weight; descriptor i owns blocks [first_block_i, first_block_{i+1}) of 1024
 *    elements each; total_blocks = the sum);
 *  - the *_u variants additionally store the weight gradient in torch's [Cout][Cin][3][3] layout from the reduction
 *    kernel itself (what a following sed_unpack_conv_wgrad call would write).                                        */
int sed_pack_conv_weights_batch(int dtype, const void* desc, int n, int total_blocks, void* stream);
/* ---- weight gradient AND data gradient of a layer in one launch: dz never written (csrc/sed_bwd_fused.hip) -----------------
 * autograd through ConvBlock, spectogram_models.py:155-158 under train.py:102.  Replaces the pair
 *   sed_conv3x3_wgrad_fused_u(..., dz_out)  +  sed_conv3x3_fwd(dz_out, wpack_t, SED_EPI_RELUBWD)          (conv2 of a block)
 *   sed_conv3x3_wgrad_fused_u(..., dz_out)  +  sed_conv3x3_dgrad_poolstats(dz_out, wpack_t, ...)           (conv1 of a block)
 * with identical operands and results: dz = the BatchNorm / ReLU / avg-pool backward of (gsrc, zsrc) as in
 * sed_conv3x3_wgrad_fused is produced into an LDS row ring, the weight gradient (dwpack + torch-layout dw) and the data
 * gradient dx [B][H][W][Cinp] (with the epilogue `epi`: SED_EPI_STORE, SED_EPI_RELUBWD with zref / epi_* as in
 * sed_conv3x3_fwd, SED_EPI_POOLSTATS with zref = pooled activation, cnt, flag as in sed_conv3x3_dgrad_poolstats) are
 * contracted from that one image.  Covered (sed_conv3x3_bwd_fused_supported): bf16, W = 32, 32 -> 64 channels with
 * SED_DZ_BN / SED_PRO_NONE / STORE or POOLSTATS, and 64 -> 64 with SED_DZ_POOL / SED_PRO_BNRELU / RELUBWD -- the two layers
 * of the main network's second block, whose two-kernel backward sits at the HBM floor of its dataflow.
 * With an epilogue, zref must be x itself (true for both layers: conv2's ReLU / BN1 reference is the z tensor its prologue
 * reads, conv1's pooled activation is its input) -- the kernel takes the reference from the tile it already holds.
 * workspace: sed_conv_wgrad_ws_floats(B, H, W, Cinp, Coutp) floats; partial [nparts][2][Cinp].                              */
/* Block 0 in C1 mode: sed_conv3x3_wgrad_fused_c1_u and sed_conv3x3_dgrad_c1_stats in ONE launch (csrc/sed_bwd_fused_c1.hip): same
 * operands, same results (dwpack / dw = conv2's weight gradient, a_partial [sed_conv_dgrad_c1_nparts()][10][32] = per-workgroup
 * partials of [A (9 taps); sum g]); dz2 is produced into an LDS row ring and never written.  workspace:
 * sed_conv_wgrad_ws_floats(B, H, 64, 32, 32) floats.  Covered: bf16, W = 64, 32 -> 32, pool 2 (..._supported).
 * relu_mask may be NULL (round 5, what the engine passes): the kernel then takes conv1's ReLU decisions from the activation tile it
 * rebuilds for the weight gradient (a1 > 0 -- the same MFMA, the same bits as the forward's mask), and the forward call
 * sed_conv3x3_fwd_c1 can be given relu_mask = NULL as well: no mask tensor exists.                                            */
int sed_conv3x3_bwd_fused_c1_supported(int dtype, int W, int Coutp, int pool);
int sed_conv3x3_bwd_fused_c1(int dtype, const float* x1, const float* fmean, const float* fstd, const float* w1,
                             const float* pro_scale, const float* pro_shift, const void* gsrc, const void* zsrc,
                             const float* scale, const float* shift, const float* ca, const float* cb, const float* cc, int pool,
                             const void* wpack_t, const void* relu_mask, float* a_partial, float* dwpack, float* workspace, int B,
                             int H, int W, int Coutp, float* dw, int Cout, int Cin, void* stream);
int sed_conv3x3_bwd_fused_supported(int dtype, int W, int Cinp, int Coutp, int dzmode, int pro, int epi);
/* Round 4: the data gradient PRODUCES dz (csrc/sed_conv_pc.hip).  For the 128-output-channel layers (blocks 2-3 of the main
 * network, /root/reference/main.py:35) the weight-gradient kernel's loader waves were its critical path: besides staging two
 * operands per MFMA they computed dz = ca*g + cb*z + cc (BatchNorm / ReLU / avg-pool backward, as sed_conv3x3_wgrad_fused) and
 * wrote it out.  The data-gradient kernel has the spare loader cycles (weights stream from L2 into registers), so the pair
 *     sed_conv3x3_wgrad_fused_u(..., dz_out) ; sed_conv3x3_fwd(dz_out, wpack_t, ...)  /  sed_conv3x3_dgrad_poolstats(dz_out, ...)
 * becomes
 *     sed_conv3x3_dgrad_dz(gsrc, zsrc, ..., dz_out, dx, epi ...) ; sed_conv3x3_wgrad_u(x, dz_out, ...)
 * with identical operands and bit-identical results (dz is rounded to bf16 once, where it is produced).  C = channels of dz (the
 * layer's outputs), Cx = channels of dx (its inputs); dzmode / gsrc / zsrc / scale / shift / ca / cb / cc / pool as in
 * sed_conv3x3_wgrad_fused; epi / zref / cnt / epi_* / partial / nparts / flag as in sed_conv3x3_fwd (SED_EPI_RELUBWD, SED_EPI_STORE) and
 * sed_conv3x3_dgrad_poolstats (SED_EPI_POOLSTATS).  Covered (..._supported): bf16, W = 16 / 8, libraries built with
 * `make EXPERIMENTS=1` only -- measured bit-identical and 11 % slower than the round-3 order (tools/ab_dgrad_dz.py), so the engine
 * does not use it; the default library answers 0.                                                                              */
int sed_conv3x3_dgrad_dz_supported(int dtype, int W, int C, int Cx, int dzmode, int epi, int pool);
int sed_conv3x3_dgrad_dz(int dtype, int dzmode, const void* gsrc, const void* zsrc, const float* scale, const float* shift,
                         const float* ca, const float* cb, const float* cc, int pool, const void* wpack_t, void* dz_out, void* dx,
                         int epi, const void* zref, const void* cnt, const float* epi_scale, const float* epi_shift,
                         const float* epi_mean, const float* epi_invstd, float* partial, int nparts, int* flag, int B, int H, int W,
                         int C, int Cx, void* stream);
/* Weight gradient with dz given, gradient also in torch's [Cout][Cin][3][3] layout (dw), like sed_conv3x3_wgrad_fused_u.        */
int sed_conv3x3_wgrad_u(int dtype, int pro, const void* x, const float* pro_scale, const float* pro_shift, const void* dz,
                        float* dwpack, float* workspace, int B, int H, int W, int Cinp, int Coutp, float* dw, int Cout, int Cin,
                        void* stream);
/* The same question with the block's pooling size (what the SED_DZ_POOL form divides by): W = 32 covers pool 2 only, the
 * 128-output-channel layers at W = 16 / 8 (csrc/sed_bwd_fused_cs.hip: 64 -> 128 and 128 -> 128, the workgroups of a pixel strip
 * sliced by input channels) cover pool 1 and 2.  sed_conv3x3_bwd_fused_supported() answers for pool 2.                          */
int sed_conv3x3_bwd_fused_supported_pool(int dtype, int W, int Cinp, int Coutp, int dzmode, int pro, int epi, int pool);
int sed_conv3x3_bwd_fused(int dtype, int pro, const void* x, const float* pro_scale, const float* pro_shift, int dzmode,
                          const void* gsrc, const void* zsrc, const float* scale, const float* shift, const float* ca,
                          const float* cb, const float* cc, int pool, const void* wpack_t, void* dx, int epi, const void* zref,
                          const void* cnt, const float* epi_scale, const float* epi_shift, const float* epi_mean,
                          const float* epi_invstd, float* partial, int nparts, int* flag, float* dwpack, float* workspace,
                          int B, int H, int W, int Cinp, int Coutp, float* dw, int Cout, int Cin, void* stream);

int sed_conv3x3_wgrad_fused_u(int dtype, int pro, const void* x, const float* pro_scale,
                              const float* pro_shift, int dzmode, const void* gsrc, const void* zsrc,
                              const float* scale, const float* shift, const float* ca, const float* cb,
                              const float* cc, int pool, void* dz_out, float* dwpack, float* workspace,
                              int B, int H, int W, int Cinp, int Coutp, float* dw, int Cout, int Cin,
                              void* stream);
int sed_conv3x3_wgrad_fused_c1_u(int dtype, const float* x1, const float* fmean, const float* fstd,
                                 const float* w1, const float* pro_scale, const float* pro_shift,
                                 const void* gsrc, const void* zsrc, const float* scale,
                                 const float* shift, const float* ca, const float* cb, const float* cc,
                                 int pool, void* dz_out, float* dwpack, float* workspace, int B, int H,
                                 int W, int Coutp, float* dw, int Cout, int Cin, void* stream);
int sed_conv3x3_c1_wgrad_combine_u(const float* a_sum, const float* gram_partial, int nparts,
                                   const float* w, const float* ca, const float* cb, const float* cc,
                                   float* dwpack, int Cout, int Coutp, float* dw, void* stream);
int sed_conv_c1_gram_nparts(int B, int H, int W);
int sed_conv3x3_c1_gram(const float* x, const float* mean, const float* stdv, float* gram_partial,
                        int B, int H, int W, void* stream);
int sed_conv3x3_c1_wgrad_combine(const float* a_sum, const float* gram_partial, int nparts,
                                 const float* w, const float* ca, const float* cb, const float* cc,
                                 float* dwpack, int Cout, int Coutp, void* stream);

/* ---- "C1 mode": the first ConvBlock (spectogram_models.py:153-160) without conv1's output in memory --------
 * z1 = conv3x3(x_norm, w1) has one input channel, so the kernels that need it -- conv2 forward (as
 * relu(bn1(z1))), conv2's weight gradient, and the ReLU/BN1-backward epilogue of conv2's data gradient --
 * rebuild it in their loader waves (one MFMA per 32 pixels) from the fp32 input x1 [B][H][W] (z-scored per
 * column with fmean/fstd, nullable) and w1 [32][9]; BN1's batch statistics come from the Gram statistics of the input patches
 * (sed_conv3x3_c1_gram).  Covered: bf16, W = 64, 32 conv1 channels, conv2 32 -> 32 (sed_c1_mode_supported).  */
int sed_c1_mode_supported(int dtype, int W, int C1, int Cout2);
int sed_bn_train_finalize_c1(const float* gram_partial, int nparts, double count, const float* w1,
                             const float* gamma, const float* beta, float* running_mean,
                             float* running_var, float momentum, float eps, float* scale, float* shift,
                             float* mean, float* invstd, int C, int Cp, void* stream);
/* Round 5: sed_bn_train_finalize_c1 that also hands out the REDUCED Gram statistics gram_sum (54 doubles: the 45 upper-triangle
 * entries of the 9 x 9 patch Gram matrix, then the 9 patch sums), and the backward tail of block 0's conv1 in one launch:
 * sed_c1_bwd_tail = sed_sum_partials(a_partial [a_nparts][10][32] -> a_sum [10][32]) ; sed_bn_bwd_finalize_c1(partial = a_sum row 9,
 * nparts = 1, a_sum) ; sed_conv3x3_c1_wgrad_combine_u(a_sum, Gram, ...) with the Gram statistics taken from gram_sum instead of a
 * second reduction of the forward's partial rows.  Same formulas and rounding points; three dependent launches less per step.
 * Not for SyncBN (the all-reduce of a_sum sits between the first two steps there, and dW1 needs the LOCAL Gram).               */
int sed_bn_train_finalize_c1_g(const float* gram_partial, int nparts, double count, const float* w1,
                               const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float momentum, float eps, float* scale, float* shift,
                               float* mean, float* invstd, int C, int Cp, double* gram_sum, void* stream);
int sed_c1_bwd_tail(const float* a_partial, int a_nparts, const double* gram_sum, double count, const float* w1,
                    const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                    float* ca, float* cb, float* cc, float* a_sum, float* dwpack, int Cout, int Coutp, float* dw,
                    void* stream);
int sed_conv3x3_fwd_c1(int dtype, int epi, const float* x1, const float* fmean, const float* fstd,
                       const float* w1, const float* pro_scale, const float* pro_shift,
                       const void* wpack, void* z, float* partial, void* relu_mask, int B, int H, int W,
                       int Coutp, void* stream);
/* relu_mask: uint16 [B][H][W][2], conv1's ReLU decisions (half g, bit i <-> channel (i&3)+8*(i>>2)+4*g), written
 * by sed_conv3x3_fwd_c1 with SED_EPI_STATS and read by the data gradient below, whose epilogue then yields
 * partial[.][0][c] = sum g only; sed_bn_bwd_finalize_c1 completes BN1's backward with sum g*z1 = w1 . A,
 * A = summed sed_conv3x3_c1_wgrad partials of g ([9][Coutp]).                                               */
int sed_conv3x3_dgrad_c1(int dtype, const void* dz, const void* wpack_t, void* g, const void* relu_mask,
                         float* partial, int B, int H, int W, int Cinp, void* stream);
/* The same data gradient FUSED with its only consumers (csrc/sed_dgrad_c1.hip): g is never written.  The kernel gates
 * conv2^T(dz) with relu_mask in the accumulator registers and contracts it over the pixels on the matrix pipe:
 * a_partial fp32 [sed_conv_dgrad_c1_nparts()][10][32], rows 0..8 = A[tap][c] = sum_px g[px][c] * xz[px + tap]
 * (what sed_conv3x3_c1_wgrad computes from g), row 9 = sum_px g[px][c] (what partial[.][0][c] holds above).
 * Summed over the partial rows (sed_sum_partials) it feeds sed_bn_bwd_finalize_c1 (partial = row 9, nparts = 1,
 * a_sum = rows 0..8) and sed_conv3x3_c1_wgrad_combine.  Covered: bf16, W = 64, 32 -> 32 channels.              */
int sed_conv_dgrad_c1_nparts(void);
int sed_conv3x3_dgrad_c1_stats(int dtype, const void* dz, const void* wpack_t, const float* x1,
                               const float* fmean, const float* fstd, const void* relu_mask,
                               float* a_partial, int B, int H, int W, void* stream);
/* test hook: the same launch additionally stores the gated g [B][H][64][32] bf16 (what sed_conv3x3_dgrad_c1 writes) */
int sed_conv3x3_dgrad_c1_stats_g(int dtype, const void* dz, const void* wpack_t, const float* x1,
                                 const float* fmean, const float* fstd, const void* relu_mask,
                                 float* a_partial, void* g_out, int B, int H, int W, void* stream);
int sed_bn_bwd_finalize_c1(const float* partial, int nparts, double count, const float* a_sum,
                           const float* w1, const float* gamma, const float* mean, const float* invstd,
                           float* dgamma, float* dbeta, float* ca, float* cb, float* cc, int C, int Cp,
                           void* stream);
int sed_conv3x3_wgrad_fused_c1(int dtype, const float* x1, const float* fmean, const float* fstd,
                               const float* w1, const float* pro_scale, const float* pro_shift,
                               const void* gsrc, const void* zsrc, const float* scale, const float* shift,
                               const float* ca, const float* cb, const float* cc, int pool, void* dz_out,
                               float* dwpack, float* workspace, int B, int H, int W, int Coutp,
                               void* stream);

/* ---- raw-waveform M5 path (models/waveform_models.py:13-71) ------------------------------------
 * Activations use the conv3x3 layout with W = 8: eight frames interleaved on the W axis,
 * [N = B/8][L][8][Cp]; B must be a multiple of 8.  The k=3 Conv1d layers run through sed_conv3x3_*
 * with their weights expanded to 3x3 (zero side columns); these entry points add what only M5 has.
 * conv_block1.0 = Conv1d(1, 64, 79, stride 4, pad 39): x fp32 [B][L], w fp32 [64][79],
 * z [B/8][L1][8][64] with L1 = sed_m5_conv1_len(L); the bias is NOT applied (BatchNorm1d removes it).
 * stats_partial fp32 [sed_m5_conv1_nparts][2][64] (sum z, sum z^2; nullable).
 * wgrad: dw_partial fp32 [sed_m5_conv1_nparts][80][64] (tap-major, tap 79 is padding).            */
int sed_m5_conv1_len(int L);
int sed_m5_conv1_nparts(int B, int L);
int sed_m5_conv1_fwd(int dtype, const float* x, const float* w, void* z, float* stats_partial, int B,
                     int L, void* stream);
int sed_m5_conv1_wgrad(int dtype, const float* x, const void* dz, float* dw_partial, int B, int L,
                       void* stream);
/* bf16: the same weight gradient on the matrix pipe with the BatchNorm1d backward produced on load,
 * dz = ca*g + cb*zsrc + cc (what sed_bn_bwd_apply would have written): dz is never materialised.                */
int sed_m5_conv1_wgrad_fused(int dtype, const float* x, const void* g, const void* zsrc, const float* ca,
                             const float* cb, const float* cc, float* dw_partial, int B, int L,
                             void* stream);
/* Round 4, "z-free" first block (bf16): conv_block1's output z1 (8x its input: 2.9 GB at 2880 frames) is never stored.  With one input
 * channel a 128-step tile is 10 MFMAs per wave away from the staged input window, so the three consumers of z1 recompute it
 * (bit-identical to the stored tensor: the same MFMA sequence, rounded to bf16 where sed_m5_conv1_fwd stored it):
 *   sed_m5_conv1_stats             the BatchNorm1d statistics of sed_m5_conv1_fwd (stats_partial as there), no z
 *   sed_m5_conv1_bn_relu_pool_fwd  sed_m5_conv1_fwd + sed_bn_relu_maxpool4_fwd: y [B/8][L1/4][8][64]
 *   sed_m5_conv1_pool_bwd_stats    sed_maxpool4_relu_bwd with g = NULL: partial [sed_m5_conv1_nparts][2][64] = (sum g, sum g*xhat)
 *   sed_m5_conv1_wgrad_fused_pool_x  sed_m5_conv1_wgrad_fused_pool without zsrc (w = conv1's weights [64][79])
 * /root/reference/models/waveform_models.py:15-24 (conv_block1) forward and backward.  sed_m5_zfree_supported(): bf16 with the
 * matrix-pipe first layer on (SED_M5_MFMA != 0) and SED_M5_ZFREE=1 -- opt-in: the same values as the stored-z path (block 1's own
 * gradients to fp32 rounding) and measured slower as built (step 9.73 against 8.80 ms at 2880 frames: the recomputing backward kernels lose more than the forward gains). */
int sed_m5_zfree_supported(int dtype);
/* The algebraic backward of conv_block1 (round 4; opt-in, SED_M5_ALG=1: parity-green, measured slower -- the Gram kernel costs more than
 * the merged pass saves).  g = MaxPool / ReLU
 * backward of dy; BatchNorm1d backward dz = ca*g + cb*z + cc; with ONE input channel
 *     dW1[c][k] = sum_t dz[c][t] x[4t + k - 39] = ca[c]*G1[k][c] + cb[c] * (w1 . Gram)[c][k] + cc[c]*Sp[k]
 * where G1 = sum_t g (x) patch comes from the ONE pass that also yields the statistics (sum g, sum g*xhat) -- z is read once instead of
 * twice -- and Gram[k'][k] = sum_t x[4t+k'-39] x[4t+k-39], Sp[k] = sum_t x[4t+k-39] (bf16-rounded x, steps t < L1) depend on the input
 * only.  dz is never formed (and not rounded to bf16: the result is the exact contraction of the coefficients' form).
 *   sed_m5_conv1_gram            gram_partial fp32 [sed_m5_conv1_nparts][sed_m5_conv1_gram_floats()]  (sum the rows: sed_sum_partials)
 *   sed_m5_conv1_bwd_stats_g1    stats_partial [nparts][2][64] (as sed_maxpool4_relu_bwd), g1_partial [nparts][80][64] (as dw_partial)
 *   sed_m5_conv1_wgrad_combine   g1 [80][64], gram [gram_floats], w [64][79] -> dw [64][79]
 * /root/reference/models/waveform_models.py:15-24 backward.                                                                       */
int sed_m5_alg_supported(int dtype);
size_t sed_m5_conv1_gram_floats(void);
int sed_m5_conv1_gram(const float* x, float* gram_partial, int B, int L, void* stream);
int sed_m5_conv1_bwd_stats_g1(int dtype, const float* x, const void* dy, const void* zsrc, const float* scale, const float* shift,
                              const float* mean, const float* invstd, float* stats_partial, float* g1_partial, int B, int L,
                              void* stream);
int sed_m5_conv1_wgrad_combine(const float* g1, const float* gram, const float* w, const float* ca, const float* cb, const float* cc,
                               float* dw, void* stream);
int sed_m5_conv1_stats(int dtype, const float* x, const float* w, float* stats_partial, int B, int L, void* stream);
int sed_m5_conv1_bn_relu_pool_fwd(int dtype, const float* x, const float* w, const float* scale, const float* shift, void* y,
                                  void* z_out, int B, int L, void* stream);
/* z_out (nullable): also store z [B/8][L1][8][64] for a backward that reads it -- the default bf16 forward of conv_block1 (round 4,
 * sed_m5_fwd2_supported): sed_m5_conv1_stats, the finalize, then this launch; sed_bn_relu_maxpool4_fwd's pass over z is gone.     */
int sed_m5_fwd2_supported(int dtype);
int sed_m5_conv1_pool_bwd_stats(int dtype, const float* x, const float* w, const void* dy, const float* scale, const float* shift,
                                const float* mean, const float* invstd, float* partial, int B, int L, void* stream);
int sed_m5_conv1_wgrad_fused_pool_x(int dtype, const float* x, const float* w, const void* dy, const float* scale, const float* shift,
                                    const float* ca, const float* cb, const float* cc, float* dw_partial, int B, int L, void* stream);
/* ... and with g rebuilt on load too: the MaxPool1d(4) + ReLU backward of the pooled gradient dy [B/8][L1/4][8][64]
 * (dy goes to the first arg-max of relu(scale*zsrc + shift) in each window of 4 when that maximum is > 0).  Pairs with
 * sed_maxpool4_relu_bwd(..., g = NULL, ...), which then only produces the BatchNorm-backward statistics.             */
int sed_m5_conv1_wgrad_fused_pool(int dtype, const float* x, const void* dy, const void* zsrc,
                                  const float* scale, const float* shift, const float* ca,
                                  const float* cb, const float* cc, float* dw_partial, int B, int L,
                                  void* stream);
/* BatchNorm1d -> ReLU -> MaxPool1d(4,4) over H (floor): y [N][H/4][W][Cp] = max relu(scale*z+shift) */
int sed_bn_relu_maxpool4_fwd(int dtype, const void* z, const float* scale, const float* shift,
                             void* y, int N, int H, int W, int Cp, void* stream);
/* its backward: g [N][H][W][Cp] = dy at the FIRST arg-max of each window when that maximum is > 0,
 * else 0 (rows dropped by the floor: 0); partial fp32 [sed_maxpool4_bwd_nparts][2][Cp] =
 * (sum g, sum g*(z-mean)*invstd) for sed_bn_bwd_finalize.                                         */
int sed_maxpool4_bwd_nparts(int N, int H, int W, int Cp);
int sed_maxpool4_relu_bwd(int dtype, const void* dy, const void* z, const float* scale,
                          const float* shift, const float* mean, const float* invstd, void* g,
                          float* partial, int N, int H, int W, int Cp, void* stream);
/* Round 4: the same statistics from POOLED tensors -- g is dy at a window's arg-max where the pooled activation y is positive, and there
 * (z - mean)*invstd = (y - beta)/gamma: partial [sed_maxpool4_bwd_nparts][2][Cp] from one pass over dy and y [N][H/4][W][Cp] (the block
 * output sed_bn_relu_maxpool4_fwd / sed_m5_conv1_bn_relu_pool_fwd stored), a quarter of z's rows each.  y's bf16 rounding is amplified by
 * |beta/gamma|: a channel with |beta| > 8 |gamma| (or gamma = 0) and any active window sets *flag, and sed_maxpool4_relu_bwd_if (a no-op
 * while *flag == 0) then recomputes every partial from z as sed_maxpool4_relu_bwd(..., g = NULL) does AND resets *flag to 0 on the stream
 * afterwards: one fixed flag word (zeroed once by the caller) serves every step, also under graph replay of the pair with fixed pointers.
 * *flag_clear (nullable; kept for callers that alternate two words) is reset to 0 by sed_maxpool4_pooled_stats itself; it must not be the
 * word passed as `flag` in the same call.  Replaces the autograd backward of MaxPool1d + ReLU in front of BatchNorm1d's,
 * /root/reference/models/waveform_models.py:18-24, for a layer whose weight gradient rebuilds g itself (conv_block1).                */
int sed_maxpool4_pooled_stats(int dtype, const void* dy, const void* y, const float* scale, const float* shift, const float* mean,
                              const float* invstd, float* partial, int* flag, int* flag_clear, int N, int H, int W, int Cp, void* stream);
int sed_maxpool4_relu_bwd_if(const int* flag, int dtype, const void* dy, const void* z, const float* scale, const float* shift,
                             const float* mean, const float* invstd, float* partial, int N, int H, int W, int Cp, void* stream);
/* head: m fp32 [B][C] = mean over H of feat [B/8][H][8][Cp]; pre fp32 [B][K] = m.W^T + b; backward:
 * dfeat [B/8][H][8][Cp], dfc_w [K][C], dfc_b [K] from dpre [B][K].                                 */
int sed_m5_head_fwd(int dtype, const void* feat, const float* fc_w, const float* fc_b, float* m,
                    float* pre, int B, int H, int C, int Cp, int K, void* stream);
int sed_m5_head_bwd(int dtype, const float* dpre, const float* m, const float* fc_w, float* dfc_w,
                    float* dfc_b, void* dfeat, int B, int H, int C, int Cp, int K, void* stream);

/* ---- utilities -----------------------------------------------------------------------------*/
/* out[i] = sum_{s<nparts} partial[s][i], i < n (fixed order: deterministic)                     */
int sed_sum_partials(const float* partial, int nparts, size_t n, float* out, void* stream);
/* fp32 [n] <-> dtype [n] casts; NCHW fp32 (B,C,H,W) <-> NHWC dtype (B,H,W,Cp) re-layouts        */
int sed_cast(int dtype_dst, void* dst, int dtype_src, const void* src, size_t n, void* stream);
int sed_nchw_to_nhwc(int dtype, const float* src, void* dst, int B, int C, int H, int W, int Cp,
                     void* stream);
int sed_nhwc_to_nchw(int dtype, const void* src, float* dst, int B, int C, int H, int W, int Cp,
                     void* stream);

/* ---- deferred weight-gradient reduction (round 6) ------------------------------------------------------------------------------
 * Every weight-gradient entry point (sed_conv3x3_wgrad*, sed_conv3x3_bwd_fused, sed_conv3x3_bwd_fused_c1) ends with a launch that sums its
 * per-workgroup slabs workspace[slab][9][Cinp][Coutp] into dwpack (and dw, torch layout).  Called with dwpack == NULL it launches its
 * main kernel only and leaves the slabs in `workspace`; sed_wgrad_last_slabs() then returns their count (per calling thread, like
 * sed_last_error).  The caller reduces later: one layer with sed_wgrad_reduce, or several layers in ONE launch with
 * sed_wgrad_reduce_batch -- desc = device array of n descriptors of ten 64-bit words {workspace, dwpack (nullable), dw (nullable),
 * slabs, 9*Cinp*Coutp, Cout, Cin, Cinp, Coutp, first_block}; descriptor i owns blocks [first_block_i, first_block_{i+1}) of 64 outputs
 * each; total_blocks = the sum.  Same arithmetic and summation order as the inline reduction (bit-identical).  The weight gradients
 * feed only the optimizer / the gradient all-reduce (train.py:101-103), so a train step's seven dependent 10 us launches become one.  */
int sed_wgrad_last_slabs(void);
int sed_wgrad_reduce(const float* workspace, int nslabs, float* dwpack, float* dw, int Cout, int Cin, int Cinp, int Coutp, void* stream);
int sed_wgrad_reduce_batch(const void* desc, int n, int total_blocks, void* stream);

/* ---- box-measured peaks (bench.py: roofline.peak_measured) ------------------------------------------------------------------
 * SURVEY.md 8(d) asks for box-measured peaks beside the spec ones.  Both are plain launches on `stream`; the caller times them.
 * sed_peak_mfma_bf16: every SIMD of every CU issues `iters` x 64 register-fed v_mfma_f32_32x32x16_bf16 on pseudo-random operands
 *   (four waves per SIMD, four independent accumulators each; no LDS, no global traffic); *flops_out (host, nullable) = the FLOPs
 *   the launch performs.  sink: one device float (never written; anchors the loop).
 * sed_peak_stream_copy: dst[0..bytes) = src[0..bytes) with 16-byte grid-stride accesses, four loads in flight per thread: moves
 *   2 x bytes through HBM when the buffers exceed the 256 MB Infinity Cache.                                                    */
int sed_peak_mfma_bf16(int iters, float* sink, double* flops_out, void* stream);
int sed_peak_stream_copy(const void* src, void* dst, size_t bytes, void* stream);
/* sed_peak_stream_read: reads src[0..bytes) with the same access pattern and writes nothing (sink: one device float, never written):
 * the ceiling of a read-dominated kernel.                                                                                        */
int sed_peak_stream_read(const void* src, size_t bytes, float* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SED_HIP_H */
