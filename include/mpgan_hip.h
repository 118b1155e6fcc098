/*
 * mpgan_hip.h -- C ABI of libmpgan_hip.so, the MI355X (gfx950) hot path of the
 * T1->T2 GAN training step.
 *
 * The reference (mbrzus/Cross-Modality-Minipig-Gan) has no FFI of its own: the
 * hot path is reached through torch.nn modules.  Each entry point below names
 * the reference call site whose ATen/cuDNN work it replaces
 * (paths relative to the reference repo; "MONAI" = monai==0.4.0, un-vendored,
 * called from code/GAN/GAN_final.py:106-114).
 *
 * Conventions
 *   - Activations are channels-last (N, D, H, W, C) fp32 with an explicit
 *     channel pitch `ld*` (elements between consecutive pixels), so a tensor
 *     may be a channel slice of a wider buffer (concat elision).  2-D tensors
 *     use D = 1 and kernel/stride/pad 1/1/0 in the depth dimension.
 *   - Conv weights are consumed in packed "OTI" order [Cout][tap][Cin]
 *     (tap = (kz*Ky+ky)*Kx+kx), produced by mpgan_pack_weights from the
 *     torch-layout parameters.
 *   - The caller (PyTorch) owns every buffer.  The library never allocates or
 *     frees device memory and keeps no pointer after return.  All calls are
 *     asynchronous on `stream` (a hipStream_t passed as void*).
 *   - Return value: 0 on success, negative on error; mpgan_last_error() returns
 *     a thread-local message.  Unsupported shapes are errors, never fallbacks.
 */
#ifndef MPGAN_HIP_H
#define MPGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPGAN_OK 0
#define MPGAN_ERR_INVALID -1
#define MPGAN_ERR_UNSUPPORTED -2
#define MPGAN_ERR_HIP -3

#define MPGAN_ACT_NONE 0
#define MPGAN_ACT_LEAKY 1 /* x>0 ? x : slope*x ; PReLU (one shared alpha) and LeakyReLU(0.2) */

const char* mpgan_last_error(void);
int mpgan_abi_version(void);

/* Geometry of one convolution as the reference's nn.ConvNd / nn.ConvTransposeNd
 * describe it.  For a transposed conv, `in` is the (small) input and `out` the
 * up-sampled output: out = (in-1)*stride - 2*pad + k + out_pad. */
typedef struct {
  int32_t n;                 /* batch */
  int32_t in_dhw[3];         /* input  spatial (D,H,W) */
  int32_t out_dhw[3];        /* output spatial (D,H,W) */
  int32_t cin, cout;
  int32_t k[3];              /* kernel (kz,ky,kx) */
  int32_t stride[3];         /* per-dimension stride (1 in an unused depth dim) */
  int32_t pad[3];
  int32_t transposed;        /* 0: ConvNd, 1: ConvTransposeNd */
  int32_t flags;             /* MPGAN_CONV_* bits below; 0 = the reference's fp32 arithmetic */
  int32_t min_blocks;        /* launch size (in blocks) from which the big-tile forms serve this conv (the DMA-staged fp32
                              * form, the wide bf16 forms); 0 = the library's default (1024).  It travels WITH the
                              * geometry, so a sizing query (mpgan_conv_stats_rows*, mpgan_conv_variant*) and the launch
                              * it sizes cannot disagree; the library keeps no mutable dispatch state. */
} mpgan_conv_geom;

/* flags: matrix operands of this conv are rounded to bf16 on their way into LDS and contracted on the bf16 matrix
 * cores with fp32 accumulation (16x the fp32 matrix rate); tensors in HBM, the prologue's arithmetic, bias,
 * residual and statistics stay fp32.  Honoured by the MFMA kernels (K-stepped and 3-D patch forms, forward /
 * backward-data / backward-weight); the thin VALU kernels of 1-channel layers ignore it.  Statistics-row counts and
 * workspace sizes do not depend on it.  Config C5's generator (BASELINE.json: "3D 128^3 ... bf16"). */
#define MPGAN_CONV_MM_BF16 1

/* Optional "normalise + activate on load" prologue applied to the operand that
 * is a raw (pre-norm) conv output: a = act(z*scale[c] + shift[c]).
 * scale/shift may be per channel (n_stride = 0, BatchNorm) or per (n, c)
 * (n_stride = C, InstanceNorm).  slope_ptr (device, 1 float) overrides slope
 * when non-null (PReLU's learnable alpha). */
typedef struct {
  const float* scale;        /* null => no prologue */
  const float* shift;
  int32_t n_stride;
  int32_t act;
  float slope;
  const float* slope_ptr;
} mpgan_prologue;

/* ---- convolution blocks ------------------------------------------------- */

/* y = conv(prologue(x)) [+ bias] [+ resid] [tanh]   (forward of ConvNd, or of
 * ConvTransposeNd when g->transposed).
 * Replaces: conv3d/conv_transpose3d dispatched by MONAI Convolution /
 * ResidualUnit (GAN_final.py:106-114) and by Discriminator.model_conv
 * (GAN_final.py:167-189, test_runs/GAN.py:142-173).
 * stats_partials (nullable): fused BatchNorm statistics -- the kernel also leaves
 * per-tile partial sums [rows][2][Cout] of (y, y^2), rows = mpgan_conv_stats_rows(),
 * which mpgan_norm_finalize(n=1, chunks=rows, pixels=N*D*H*W, instance=0) consumes;
 * rows == 0 means this geometry has no fused statistics (use mpgan_channel_stats). */
int32_t mpgan_conv_stats_rows(const mpgan_conv_geom* g, int32_t has_prologue);
int mpgan_conv_forward(const mpgan_conv_geom* g, const float* x, int32_t ldx,
                       const float* w_packed, const float* bias,
                       const mpgan_prologue* pro,
                       const float* resid, int32_t ldr, int32_t tanh_out,
                       float* stats_partials,
                       float* y, int32_t ldy, void* stream);

/* The same forward with the K axis split over blocks, for output grids too small to fill
 * the chip (variant B's Linear(512*8^3 -> 64) expressed as a conv whose kernel spans the
 * 8^3 grid, test_runs/GAN.py:176-181).  workspace >= mpgan_conv_splitk_workspace() bytes
 * (0 = no split needed: plain forward is used). */
int64_t mpgan_conv_splitk_workspace(const mpgan_conv_geom* g);
int mpgan_conv_forward_splitk(const mpgan_conv_geom* g, const float* x, int32_t ldx,
                              const float* w_packed, const float* bias, const mpgan_prologue* pro,
                              void* workspace, int64_t workspace_bytes, float* y, int32_t ldy, void* stream);

/* dx = conv_backward_data(dy) [+ resid]: gradient w.r.t. the conv input
 * (for g->transposed: gradient w.r.t. the transposed conv's input).
 * `w_packed_bwd` is the packed weight for the backward direction
 * (mpgan_pack_weights with for_dgrad=1).
 * Replaces: the autograd conv backward-data kernels behind loss.backward()
 * (GAN_final.py:273,296 via Lightning's loop). */
int mpgan_conv_backward_data(const mpgan_conv_geom* g, const float* dy, int32_t lddy,
                             const float* w_packed_bwd,
                             const float* resid, int32_t ldr,
                             float* dx, int32_t lddx, void* stream);

/* Backward-data AND the reduce pass of the norm layer in front of the conv's input, in one launch: dx is the
 * gradient w.r.t. a = act(scale*z + shift); partials[rows][3][Cin] receive, per tile, the sums
 * mpgan_norm_bwd_reduce would form by re-reading dx and z (sum gy, sum gy*zhat, sum g*min(y,0)); feed them to
 * mpgan_norm_bwd_finalize with n = 1, chunks = rows.  BatchNorm only (scale/shift/mean/invstd: Cin values).
 * mpgan_conv_bwd_stats_rows: rows the launch leaves; 0 = this geometry is served by a thin or patch kernel
 * without fused sums.
 * Replaces: the autograd backward of `Conv -> BatchNorm -> LeakyReLU` chains of the discriminator
 * (GAN_final.py:167-189 under loss.backward(), :273,296). */
int32_t mpgan_conv_bwd_stats_rows(const mpgan_conv_geom* g);
int mpgan_conv_backward_data_stats(const mpgan_conv_geom* g, const float* dy, int32_t lddy,
                                   const float* w_packed_bwd, float* dx, int32_t lddx,
                                   const float* z, int32_t ldz, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, int32_t act, float slope,
                                   float* partials, void* stream);

/* Which kernel serves this geometry (for profiling labels): 1 = thin Cin==1 VALU
 * stencil, 2 = thin Cout==1 VALU stencil, 16 = fp32-MFMA patch kernel (2-D, <= 32
 * output channels, input patch + weights staged once in LDS; 17 = its merged form, one block per
 * tile walking every phase of a strided backward-data / transposed conv), 18 = the 3-D patch kernel
 * (16 -> 16 channels, 3x3x3, stride 1: 2x8x8 output tiles, 16x16x4 MFMA), 32/64/128 = fp32-MFMA
 * K-stepped implicit GEMM with that output-channel tile.
 * 1128 = the 128 tile's mask-free instance (pad-free forward conv, Cout % 128 == 0, has_prologue 3).
 * 2032 / 2064 = the 32 / 64 tile with the K axis split over two 4-wave groups inside the block (output grids
 * of about one block per CU: the U-Net's 32 x 32 levels).
 * has_prologue: 0 none, 1 per-channel scale/shift, 2 per-(sample, channel), 3 per-channel +
 * LeakyReLU with a host-known slope in [0, 1]. */
int32_t mpgan_conv_variant(const mpgan_conv_geom* g, int32_t backward_data, int32_t has_prologue);

/* Weight gradient: dW (torch layout, (Cout,Cin,k..) or (Cin,Cout,k..) for a
 * transposed conv) = beta*dW + sum over pixels.  x is the conv's input (with
 * optional prologue), dy the gradient of its raw output.  `workspace` holds
 * the split-K partial slabs (size from mpgan_conv_wgrad_workspace). */
int64_t mpgan_conv_wgrad_workspace(const mpgan_conv_geom* g);
int mpgan_conv_backward_weight(const mpgan_conv_geom* g, const float* x, int32_t ldx,
                               const mpgan_prologue* pro,
                               const float* dy, int32_t lddy,
                               float* dw, float* dbias /* nullable; ConvNd only: dbias = beta*dbias + colsum(dy), fused */,
                               float beta,
                               void* workspace, int64_t workspace_bytes, void* stream);

/* Repack every conv / linear weight of a network in ONE launch.
 * table: device int64 [n_entries][8] = {src_off, dst_off, cout, cin, taps,
 * transposed, layout, 0}; offsets in floats into `flat_params` / `packed`;
 * layout 0 = [cout][tap][cin] (forward), 1 = [cin][tap][cout] (backward-data),
 * 2 = [tap][cin][cout] (backward-data of a Linear expressed as a whole-grid conv). */
int mpgan_pack_weights(const float* flat_params, float* packed, const int64_t* table,
                       int32_t n_entries, int64_t max_elems, void* stream);

/* Perceptual-loss taps (variant B, test_runs/GAN.py:183-198,288-298): the discriminator
 * returns clones of every intermediate; the G loss adds sum_k mean|real_k - fake_k| / numel_k.
 * For a conv -> BatchNorm -> LeakyReLU layer the three taps (conv out z, norm out y,
 * activation a) are never materialised: their L1 terms and gradients are evaluated from the
 * two passes' raw conv outputs.  coef = device float[3]: d(loss)/d|tap| for (z, y, a). */
typedef struct {
  const float* z_peer;      /* the other pass's raw conv output, same shape */
  int32_t ld_peer;
  const float* scale_peer;  /* its norm scale / shift (per channel) */
  const float* shift_peer;
  const float* coef;        /* null => no peer */
} mpgan_peer_taps;

/* out3 = (mean|z_a-z_b|, mean|y_a-y_b|, mean|a_a-a_b|); partials >= mpgan_tap_l1_partials() floats */
int32_t mpgan_tap_l1_partials(void);
int mpgan_tap_l1(const float* za, int32_t lda, const mpgan_prologue* pa,
                 const float* zb, int32_t ldb, const mpgan_prologue* pb,
                 int64_t rows, int32_t c, float* partials, float* out3, void* stream);

/* ---- normalisation (BatchNorm training mode / InstanceNorm) -------------- */

/* Per-channel partial sums of z and z^2 over pixel chunks.
 * partials: [n][chunks][2][C]; returns chunks via mpgan_stats_chunks().
 * Replaces: the statistics half of batch_norm (every adn.N in MONAI UNet;
 * GAN_final.py:170,176,182,188). */
int32_t mpgan_stats_chunks(int64_t pixels_per_sample, int32_t c);
int mpgan_channel_stats(const float* z, int32_t ldz, int32_t n, int64_t pixels_per_sample,
                        int32_t c, float* partials, void* stream);

/* Finalise statistics: mean / biased var -> scale = gamma*invstd,
 * shift = beta - mean*scale; saves mean and invstd; updates running stats
 * (momentum, UNBIASED variance) when running_mean != null.
 * instance = 0: one set per channel (stats over n and pixels);
 * instance = 1: one set per (n, c).
 * With instance = 0 and more than 4096 partial rows the rows are folded first, into scratch
 * that must follow them in the same buffer: capacity >= (n*chunks + 32) * 2 * c floats. */
int mpgan_norm_finalize(const float* partials, int32_t n, int32_t chunks, int32_t c,
                        int64_t pixels_per_sample, int32_t instance,
                        const float* gamma, const float* beta, float eps, float momentum,
                        float* running_mean, float* running_var, int64_t* num_batches_tracked,
                        float* scale, float* shift, float* mean, float* invstd, void* stream);

/* The same over partial rows that are `cstride` >= c channels wide ([rows][2][cstride]): statistics of the FIRST c
 * channels of a conv that produced more (a ResidualUnit's first conv and its residual conv run as ONE conv
 * over their concatenated output channels -- same input, same geometry; only the first half feeds a norm). */
int mpgan_norm_finalize_strided(const float* partials, int32_t n, int32_t chunks, int32_t c, int32_t cstride,
                                int64_t pixels_per_sample, int32_t instance,
                                const float* gamma, const float* beta, float eps, float momentum,
                                float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                float* scale, float* shift, float* mean, float* invstd, void* stream);

/* Eval-mode BatchNorm (module.eval(), code/GAN/inferrence.py:97-110,169-170): scale/shift
 * from the running statistics; nothing is updated. */
int mpgan_norm_from_running(const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float eps, int32_t c,
                            float* scale, float* shift, float* mean, float* invstd, void* stream);

/* The same for every norm layer of a network in one launch.  table: device int64[n_layers][10] =
 * {gamma, beta, running_mean, running_var, scale, shift, mean, invstd (device addresses; gamma/beta
 * may be 0), C, eps (the float's bit pattern in the low 32 bits)}. */
int mpgan_norm_from_running_multi(const int64_t* table, int32_t n_layers, void* stream);

/* Eval-mode inference (code/GAN/inferrence.py:97-110,169-170: generator.eval(); generator(x)): with running-statistics
 * BatchNorm a layer's affine is known before its conv runs, so the conv's epilogue applies it, the PReLU and the residual
 * add, and the ACTIVATED tensor is what reaches memory -- no norm_act_add launch, no prologue in the consumer:
 *     y = prelu(conv(x) * scale[c] + shift[c], slope[c]) (+ resid) (tanh)
 * The conv's own bias is folded into `shift` by the caller (mpgan_epi_vectors_multi does it); slope[c] = 1 leaves a
 * channel linear.  scale / shift / slope: [cout] floats, 16-byte aligned.  Not combinable with fused statistics. */
int mpgan_conv_forward_act(const mpgan_conv_geom* g, const float* x, int32_t ldx, const float* w_packed,
                           const float* scale, const float* shift, const float* slope, const float* resid,
                           int32_t ldr, int32_t tanh_out, float* y, int32_t ldy, void* stream);

/* The three vectors of every conv of an eval-mode plan in one launch.  table: device int64[n_layers][12] = {gamma, beta,
 * running_mean, running_var, conv bias, PReLU weight (one element) (device addresses; gamma / beta / bias / PReLU may be
 * 0), scale, shift, slope (outputs), c_norm, c_total, eps (the float's bit pattern in the low 32 bits)}: channels
 * < c_norm get scale = gamma / sqrt(var + eps), shift = beta + (bias - mean) * scale, slope = the PReLU weight (1 without
 * one); channels c_norm .. c_total - 1 (no norm layer: the residual half of a fused conv) get 1, bias, 1. */
int mpgan_epi_vectors_multi(const int64_t* table, int32_t n_layers, void* stream);

/* out = act(z*scale+shift) [+ r]   where r is either a plain tensor or itself
 * act(zr*scale_r+shift_r); optional tanh on the sum.  (ResidualUnit.forward's
 * "cx + res" and the generator's final Tanh, GAN_final.py:117.) */
int mpgan_norm_act_add(const float* z, int32_t ldz, const mpgan_prologue* pz,
                       const float* r, int32_t ldr, const mpgan_prologue* pr,
                       int32_t n, int64_t pixels_per_sample, int32_t c, int32_t tanh_out,
                       float* out, int32_t ldo, void* stream);

/* Backward of a = act(z*scale+shift), given g = dL/da:
 *   reduce  : partials [n][chunks][3][C] of (sum gy, sum gy*zhat, sum g*min(y,0)), followed by one float per
 *             row (n*chunks of them): that row's third sums added over the channels (the PReLU-slope gradient's
 *             terms) -- the buffer holds n*chunks*(3*C + 1) floats, chunks = mpgan_stats_chunks(P, C)
 *   finalize: dgamma += , dbeta += , dslope += (dslope != null needs the per-row scalars a reduce pass left;
 *             rows written by mpgan_conv_backward_data_stats carry none); coef c1 = sum gy / M, c2 = sum gy*zhat / M
 *   apply   : dz = scale*(gy - c1 - zhat*c2)
 * g may carry a fused pointwise factor: g_eff = g * (1 - t^2) (tanh backward)
 * when tanh_y != null. */
int mpgan_norm_bwd_reduce(const float* g, int32_t ldg, const float* z, int32_t ldz,
                          const mpgan_prologue* p, const float* mean, const float* invstd,
                          const mpgan_peer_taps* peer /* nullable */,
                          int32_t n, int64_t pixels_per_sample, int32_t c,
                          float* partials, void* stream);
int mpgan_norm_bwd_finalize(const float* partials, int32_t n, int32_t chunks, int32_t c,
                            int64_t pixels_per_sample, int32_t instance,
                            float* dgamma, float* dbeta, float* dslope,
                            float* c1, float* c2, void* stream);
int mpgan_norm_bwd_apply(const float* g, int32_t ldg, const float* z, int32_t ldz,
                         const mpgan_prologue* p, const float* mean, const float* invstd,
                         const float* c1, const float* c2,
                         const mpgan_peer_taps* peer /* nullable */,
                         int32_t n, int64_t pixels_per_sample, int32_t c,
                         float* dz, int32_t lddz, void* stream);

/* out[c] = beta*out[c] + sum_p partials[p][c]  (deterministic order). */
int mpgan_reduce_partials(const float* partials, int32_t rows, int32_t row_stride, int32_t c,
                          float* out, float beta, void* stream);

/* ---- pointwise helpers --------------------------------------------------- */
/* y = a + b (optionally y = tanh(a+b));  dx = g*(1-y^2). */
int mpgan_add_tanh(const float* a, const float* b, int64_t numel, int32_t apply_tanh,
                   float* y, void* stream);
int mpgan_tanh_backward(const float* g, const float* y, int64_t numel, float* dx, void* stream);
int mpgan_axpby(const float* a, float alpha, const float* b, float beta, int64_t numel,
                float* y, void* stream);
/* strided channel-slice copy: dst[p*ldd + c] = src[p*lds + c] (accumulate: +=) */
int mpgan_copy_slice(const float* src, int32_t lds, float* dst, int32_t ldd, int64_t pixels,
                     int32_t c, int32_t accumulate, void* stream);

/* ---- discriminator head: Flatten + Linear(F,1) + Sigmoid + BCE ------------ */
/* logit[n] = bias + sum_k act(z[n][k]*scale+shift) * w_perm[k]  where z is the
 * last conv's raw output in channels-last order and w_perm the Linear weight
 * permuted to that order by mpgan_pack_weights.  (GAN_final.py:198-204.) */
int32_t mpgan_linear1_partials(int32_t n); /* floats of `partials` scratch */
int mpgan_linear1_forward(const float* z, const mpgan_prologue* p, int32_t n,
                          int64_t pixels_per_sample, int32_t c, const float* w_perm,
                          const float* bias, float* partials, float* logit, float* prob /* sigmoid(logit), nullable */,
                          void* stream);
/* g_a[n][k] = dlogit[n]*w_perm[k];  dW (torch order, C-major) += sum_n dlogit[n]*a[n][k];
 * dbias += sum_n dlogit[n]. */
int mpgan_linear1_backward(const float* z, const mpgan_prologue* p, int32_t n,
                           int64_t pixels_per_sample, int32_t c, const float* w_perm,
                           const float* dlogit, float* g_a, float* dw, float* dbias,
                           float beta, void* stream);

/* prob = sigmoid(logit); loss = mean BCE(prob, target) with log clamped at
 * -100 (F.binary_cross_entropy, GAN_final.py:244-245); dlogit = loss_scale *
 * dL/dlogit computed THROUGH the clamped log and the sigmoid exactly as
 * autograd does (BCE backward: (p-t)/max((1-p)p, 1e-12)/n, then p(1-p)). */
int mpgan_sigmoid_bce(const float* logit, int32_t n, float target, float loss_scale,
                      float* prob, float* loss, float* dlogit, void* stream);

/* The same loss as separate autograd-shaped pieces (the module API returns the
 * probability, as the reference's Discriminator.forward does, and
 * GAN.adversarial_loss consumes it):
 *   bce_forward : loss = mean_i -(t_i*max(log p_i,-100) + (1-t_i)*max(log(1-p_i),-100))
 *   bce_backward: dprob_i = gout * (p_i - t_i) / max((1-p_i)*p_i, 1e-12) / n   (gout: device scalar)
 *   sigmoid_backward: dlogit_i = dprob_i * (1-p_i) * p_i */
int mpgan_bce_forward(const float* prob, const float* target, int32_t n, float* loss, void* stream);
int mpgan_bce_backward(const float* prob, const float* target, int32_t n, const float* gout,
                       float* dprob, void* stream);
int mpgan_sigmoid_forward(const float* logit, int32_t n, float* prob, void* stream);
int mpgan_sigmoid_backward(const float* dprob, const float* prob, int32_t n, float* dlogit, void* stream);
/* y = x * (*scalar)  (scalar on the device: an upstream autograd gradient) */
int mpgan_scale_by_device_scalar(const float* x, const float* scalar, int64_t numel, float* y, void* stream);
/* out[0] = sum_i v[i]*w[i], i = 0..n-1 in index order (n <= 4096 device scalars).  Replaces the running sum of
 * `F.l1_loss(...) / numel` over the 16 perceptual taps (test_runs/GAN.py:288-298): v holds the taps' L1 means, w
 * their 1/numel weights. */
int mpgan_weighted_sum(const float* v, const float* w, int32_t n, float* out, void* stream);

/* loss = mean |a-b| (F.l1_loss, GAN_final.py:247-248); grad_a = scale*sign(a-b)/numel
 * (written when grad_a != null).  partials: >= mpgan_l1_partials() floats. */
int32_t mpgan_l1_partials(void);
int mpgan_l1_loss(const float* a, const float* b, int64_t numel, float grad_scale,
                  float* partials, float* loss, float* grad_a, void* stream);

/* ---- evaluation metrics (next-row N2: inferrence.py:188-204, metrics.py:213-223,
 *      psnr_ssim_metric.py:88-106) ---------------------------------------------------- */
/* y = clip(round((x-min)/(max-min)*(b_max-b_min)+b_min)): ScaleIntensityRangePercentiles(0,100,0,255)
 * + np.round as the inference script applies before writing / scoring; minmax2 receives (min,max).
 * partials >= mpgan_metric_partials() floats. */
int32_t mpgan_metric_partials(void);
int mpgan_rescale_minmax(const float* x, int64_t numel, float b_min, float b_max, int32_t do_round,
                         float* partials, float* minmax2, float* y, void* stream);
/* out3 = (MAE, MSE, PSNR = 10 log10(data_range^2 / MSE)) between two tensors. */
int mpgan_image_errors(const float* a, const float* b, int64_t numel, float data_range,
                       float* partials, float* out3, void* stream);

/* ---- pre-processing, array half of next-row N3 (GAN_final.py:386-394:
 *      ScaleIntensityRangePercentilesd(lower=1, upper=99, b_min=-1, b_max=1, clip=True)) ---------- */
/* out[j] = np.percentile(x, q[j]) (linear interpolation) for nq in {1, 2} percentiles: exact order
 * statistics by a 3-pass radix select; q_host is read on the host at call time.
 * workspace >= mpgan_percentile_workspace() bytes, 8-byte aligned. */
int64_t mpgan_percentile_workspace(void);
int mpgan_percentiles(const float* x, int64_t numel, const double* q_host, int32_t nq,
                      void* workspace, int64_t workspace_bytes, float* out, void* stream);
/* y = (x - a_min)/(a_max - a_min)*(b_max - b_min) + b_min, clipped to [b_min, b_max] when clip != 0
 * (MONAI ScaleIntensityRange; a_minmax = device float[2], e.g. the output of mpgan_percentiles). */
int mpgan_scale_intensity_range(const float* x, int64_t numel, const float* a_minmax, float b_min,
                                float b_max, int32_t clip, float* y, void* stream);

/* ResampleT1T2d (code/GAN/transforms.py:79-213) on arrays: linear-interpolation resampling of `vol` (contiguous
 * (D,H,W) fp32 with ITK geometry origin / spacing in (x,y,z) order and a row-major 3x3 direction matrix) onto
 * the identity-direction reference grid the transform builds: out_dhw voxels, origin = -size/2, spacing =
 * extent_mm / size (extent_mm = 256 in the reference), identity transform, default pixel 0.  Restates ITK 5's
 * ResampleImageFilter + LinearInterpolateImageFunction (ITK is not installable here: parity unpinned). */
int mpgan_resample_to_identity_grid(const float* vol, const int32_t* in_dhw, const double* origin_xyz,
                                    const double* spacing_xyz, const double* direction_3x3, const int32_t* out_dhw,
                                    double extent_mm, float* out, void* stream);

/* Mean structural similarity of two slices (dhw[0] == 1: 7x7 window) or volumes (dhw[0] >= 7: 7x7x7),
 * the algorithm skimage.metrics.structural_similarity runs with the arguments psnr_ssim_metric.py:91-92
 * passes (data_range only): uniform window, K1 = 0.01, K2 = 0.03, sample covariance, mean over the
 * interior that drops 3 border samples per windowed axis.  a, b: contiguous (D,H,W) fp32.
 * workspace >= mpgan_ssim_workspace(dhw) bytes (per-tile partial sums, double). */
int64_t mpgan_ssim_workspace(const int32_t* dhw);
int mpgan_ssim(const float* a, const float* b, const int32_t* dhw, float data_range,
               void* workspace, int64_t workspace_bytes, float* out1, void* stream);

/* ---- optimiser ------------------------------------------------------------ */
/* torch.optim.Adam.step over one flat buffer (GAN_final.py:306-307):
 * m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t)+eps).
 * grad_scale multiplies g first (1/world_size after a sum all-reduce).  Hyper-parameters are
 * doubles (python floats), rounded to fp32 where torch rounds them. */
int mpgan_adam_step(float* p, const float* g, float* m, float* v, int64_t numel,
                    double lr, double b1, double b2, double eps, int32_t step, float grad_scale,
                    void* stream);

/* ---- patch gather / scatter (variant B, test_runs/GAN.py:263-272,313-337) -- */
/* patches[(b*S+s)][roi^dims] = vol[b][corner+...]; bit-exact copy.
 * corners: device int32 [B*S][3] (z,y,x). */
int mpgan_patch_gather(const float* vol, int32_t b, const int32_t dhw[3],
                       const int32_t* corners, int32_t samples, const int32_t roi[3],
                       float* patches, void* stream);
/* dvol[b][corner+...] += dpatches (atomic-free: one thread per volume voxel
 * walks the corners that cover it, in sample order => deterministic). */
int mpgan_patch_scatter_add(const float* dpatches, int32_t b, const int32_t dhw[3],
                            const int32_t* corners, int32_t samples, const int32_t roi[3],
                            float* dvol, void* stream);

/* ---- BatchNorm statistics without a finalize launch (the generator's forward chain) ----------------------
 * A conv leaves per-channel fixed-point sums (sum y, sum y^2) of its raw output in integer accumulators
 * `acc` = int64 [replicas][MPGAN_ACC_WORDS][cstride] (zeroed by the caller before the producing launch;
 * order-independent, hence reproducible); the FIRST consumer of those statistics folds them at block start
 * (mpgan_norm_fold), publishes scale / shift / mean / invstd and advances the running statistics -- what
 * mpgan_norm_finalize does, without its launch.  See csrc/norm_fold.h. */
#define MPGAN_ACC_WORDS 4
typedef struct {
  const int64_t* acc;        /* null => no fold: the prologue's own scale / shift are used */
  int32_t replicas;
  int32_t cstride;           /* channels per accumulator row (>= the c channels the norm covers) */
  int64_t count;             /* elements per channel (N*D*H*W) */
  const float* gamma;
  const float* beta;
  float eps, momentum;
  float* running_mean;       /* nullable */
  float* running_var;
  int64_t* num_batches_tracked;
  float* scale;              /* outputs, written by one block of the consuming launch */
  float* shift;
  float* mean;
  float* invstd;
} mpgan_norm_fold;

/* Zero `bytes` bytes at ptr on the stream (the accumulators of all norm layers of a plan: one call per forward). */
int mpgan_zero_bytes(void* ptr, int64_t bytes, void* stream);
/* 1 if the kernel serving this conv can leave accumulators (stats_acc) / can fold them on load (fold). */
int32_t mpgan_conv_acc_supported(const mpgan_conv_geom* g, int32_t has_prologue);
int32_t mpgan_conv_fold_supported(const mpgan_conv_geom* g);
/* mpgan_conv_forward with either form of statistics on either side: `fold` (nullable) replaces pro->scale/shift
 * (pro still carries act / slope / slope_ptr); stats_acc (nullable, exclusive with stats_partials) receives this
 * conv's sums with `acc_replicas` replicas of Cout-wide rows. */
int mpgan_conv_forward_fold(const mpgan_conv_geom* g, const float* x, int32_t ldx, const float* w_packed,
                            const float* bias, const mpgan_prologue* pro, const mpgan_norm_fold* fold,
                            const float* resid, int32_t ldr, int32_t tanh_out, float* stats_partials,
                            int64_t* stats_acc, int32_t acc_replicas, float* y, int32_t ldy, void* stream);
/* mpgan_norm_act_add whose z-side scale / shift come from accumulators (fold_z; pz carries act / slope). */
int mpgan_norm_act_add_fold(const float* z, int32_t ldz, const mpgan_prologue* pz, const mpgan_norm_fold* fold_z,
                            const float* r, int32_t ldr, const mpgan_prologue* pr, int32_t n,
                            int64_t pixels_per_sample, int32_t c, int32_t tanh_out, float* out, int32_t ldo,
                            void* stream);

/* ---- bf16 storage path (BASELINE config C5: the reference's own 3-D graph, GAN_final.py:106-114,167-189) ----
 * Activations, activation gradients and packed weights are bf16 in HBM (void* below); products accumulate in fp32
 * on v_mfma_f32_32x32x16_bf16; BatchNorm statistics, parameters, weight gradients and Adam stay fp32.
 * There is no normalise-on-load prologue in this path: a layer's BatchNorm + LeakyReLU output is materialised
 * once by mpgan_norm_act_bf16 (see conv_bf16.hip for why) and convolutions read plain activations. */

/* y (bf16) = conv(x (bf16)) + bias; gathered channels % 64 == 0, output channels % 8 == 0, <= 32 taps per phase.
 * stats_partials (nullable): per-tile partial sums [rows][2][Cout] of (y, y^2) in fp32 taken before rounding,
 * rows = mpgan_conv_stats_rows_bf16(); consumed by mpgan_norm_finalize(n=1, chunks=rows, pixels=N*D*H*W). */
int32_t mpgan_conv_stats_rows_bf16(const mpgan_conv_geom* g);
int mpgan_conv_forward_bf16(const mpgan_conv_geom* g, const void* x, int32_t ldx, const void* w_packed,
                            const float* bias, float* stats_partials, void* y, int32_t ldy, void* stream);
/* Which bf16 kernel serves this geometry (profiling labels only): 0 = the K-stepped gather kernel,
 * 1 = the patch form for stride-1 3x3x3 gathers (D.conv2 forward / backward-data at config C5),
 * 2 / 3 / 4 = the wide K-stepped form: 256 x 256 tiles, 512 x 128 tiles, 256 x 256 tiles over pairs of phases of a
 * strided backward-data gather (D.conv3 / D.conv4 at config C5). */
int32_t mpgan_conv_variant_bf16(const mpgan_conv_geom* g, int32_t backward_data);

/* dx (bf16) = conv_backward_data(dy (bf16)); w_packed_bwd: bf16, layout 1 of mpgan_pack_weights_bf16. */
int mpgan_conv_backward_data_bf16(const mpgan_conv_geom* g, const void* dy, int32_t lddy, const void* w_packed_bwd,
                                  void* dx, int32_t lddx, void* stream);
/* kernel label of the bf16 weight gradient of this layer: 0 = 128 x 256 tiles, 1 = 256 x 256 (profiling only) */
int32_t mpgan_conv_wgrad_variant_bf16(const mpgan_conv_geom* g);
/* dW (fp32, torch layout) = beta*dW + sum over pixels of dy (bf16) x gathered x (bf16): pad-free ConvNd. */
int64_t mpgan_conv_wgrad_workspace_bf16(const mpgan_conv_geom* g);
int mpgan_conv_backward_weight_bf16(const mpgan_conv_geom* g, const void* x, int32_t ldx, const void* dy, int32_t lddy,
                                    float* dw, float beta, void* workspace, int64_t workspace_bytes, void* stream);
/* The 1-input-channel first layer (Discriminator.model_conv[0]): fp32 image in / bf16 raw output out, its
 * backward-data (bf16 dy -> fp32 dx) and its weight gradient with a bf16 dy (HBM-bound VALU kernels; packed
 * weights fp32 as in the fp32 path). */
int mpgan_conv_forward_f32_to_bf16(const mpgan_conv_geom* g, const float* x, int32_t ldx, const float* w_packed,
                                   const float* bias, float* stats_partials, void* y, int32_t ldy, void* stream);
int mpgan_conv_backward_data_bf16_to_f32(const mpgan_conv_geom* g, const void* dy, int32_t lddy,
                                         const float* w_packed_bwd, float* dx, int32_t lddx, void* stream);
int64_t mpgan_conv_wgrad_workspace_bf16dy(const mpgan_conv_geom* g);
int mpgan_conv_backward_weight_bf16dy(const mpgan_conv_geom* g, const float* x, int32_t ldx, const void* dy,
                                      int32_t lddy, float* dw, float* dbias, float beta, void* workspace,
                                      int64_t workspace_bytes, void* stream);
/* mpgan_pack_weights with a bf16 destination (same table; layouts 0 and 1; dst offsets in elements). */
int mpgan_pack_weights_bf16(const float* flat_params, void* packed, const int64_t* table, int32_t n_entries,
                            int64_t max_elems, void* stream);
/* out = LeakyReLU_slope(z*scale[c] + shift[c]) over [rows][C] bf16 z; out is bf16, or fp32 when out_f32 != 0
 * (the tensor the fp32 Linear head reads).  (nn.BatchNorm3d + nn.LeakyReLU(0.2), GAN_final.py:170-171.) */
int mpgan_norm_act_bf16(const void* z, int32_t ldz, const float* scale, const float* shift, float slope, int64_t rows,
                        int32_t c, void* out, int32_t ldo, int32_t out_f32, void* stream);
/* BatchNorm + LeakyReLU backward on bf16 z, given g = dL/da (bf16, or fp32 when g_f32 != 0):
 *   reduce: partial rows [mpgan_norm_bwd_rows_bf16()][3][C] for mpgan_norm_bwd_finalize(n=1, chunks=rows, P=rows_total);
 *   apply : dz (bf16) = scale*(gy - c1 - zhat*c2); bias_partials (nullable) receives per-block column sums
 *           [mpgan_norm_bwd_rows_bf16()][C] of the stored dz (the conv's bias gradient, reduce with mpgan_reduce_partials). */
/* Backward-data of the bf16 path + the reduce pass of the BatchNorm + LeakyReLU(slope) in front of the conv's input in one
 * launch (as mpgan_conv_backward_data_stats does for fp32): dx (bf16) is the gradient w.r.t. a = act(scale * z + shift);
 * partials[mpgan_conv_bwd_stats_rows_bf16(g)][3][cin] receive what mpgan_norm_bwd_reduce_bf16 would form from the STORED
 * dx and z.  rows == 0: this geometry runs on the narrow K-stepped kernel, which has no fused sums.
 * (nn.BatchNorm3d + nn.LeakyReLU(0.2) backward, GAN_final.py:170-171 under autograd.) */
int32_t mpgan_conv_bwd_stats_rows_bf16(const mpgan_conv_geom* g);
int mpgan_conv_backward_data_stats_bf16(const mpgan_conv_geom* g, const void* dy, int32_t lddy, const void* w_packed_bwd,
                                        void* dx, int32_t lddx, const void* z, int32_t ldz, const float* scale,
                                        const float* shift, const float* mean, const float* invstd, float slope,
                                        float* partials, void* stream);
int32_t mpgan_norm_bwd_rows_bf16(int64_t rows, int32_t c);
int mpgan_norm_bwd_reduce_bf16(const void* g, int32_t g_f32, int32_t ldg, const void* z, int32_t ldz, const float* scale,
                               const float* shift, const float* mean, const float* invstd, float slope, int64_t rows,
                               int32_t c, float* partials, void* stream);
int mpgan_norm_bwd_apply_bf16(const void* g, int32_t g_f32, int32_t ldg, const void* z, int32_t ldz, const float* scale,
                              const float* shift, const float* mean, const float* invstd, const float* c1,
                              const float* c2, float slope, int64_t rows, int32_t c, void* dz, int32_t lddz,
                              float* bias_partials, void* stream);

/* ---- development aids (no counterpart in the reference) ----------------------------------------------------
 * In-kernel phase stamps: a library built with `make STAMPS=1` records, for each of the next `launches`
 * gather-conv launches, the 100 MHz device clock at every block's phase boundaries into
 * buf[launch][blocks_per_launch][12] (uint64; blocks beyond the cap do not stamp).  The product build returns
 * MPGAN_ERR_UNSUPPORTED.  mpgan_debug_stamps(NULL, 0, 0) switches stamping off. */
int mpgan_debug_stamps(void* buf, int64_t launches, int64_t blocks_per_launch);
int64_t mpgan_debug_stamps_used(void);
int32_t mpgan_debug_clock_khz(void);
/* (Rounds 2-3 had two process-wide setters here for the launch size from which the big-tile forms are used;
 *  since ABI version 2 that threshold is the geometry's own `min_blocks` field: no mutable dispatch state.) */

#ifdef __cplusplus
}
#endif
#endif /* MPGAN_HIP_H */
