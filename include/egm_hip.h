/*
 * egm_hip.h — C ABI of libegm_hip.so, the MI355X (gfx950) implementation of the
 * EGM-UNet segmentation training hot path.
 *
 * The reference (feiyeha/EGM-Unet) is pure PyTorch Python and has no FFI of its
 * own: the drop-in boundary is the nn.Module forward()/state_dict() surface
 * (src/unet.py:61-96, src/EGM-UNet.py:1503-1541) and the train_utils entry
 * points (train_utils/train_and_eval.py:7-100).  The host-side mirror of that
 * surface lives in egm_unet_amd/ (Python, like the reference); every arithmetic
 * step it performs goes through the entry points declared here.  Each entry
 * point cites the reference computation it replaces.
 *
 * Conventions
 *  - plain C: caller-owned DEVICE pointers, sizes as int / long long, an explicit
 *    hipStream_t (passed as void*).  No allocation, no synchronisation inside.
 *  - return 0 on success, negative egm_status otherwise; egm_last_error() gives
 *    the message (thread-local).
 *  - activations are NHWC ("pixel-major"): element (n,y,x,c) of a tensor with
 *    pixel stride `ld` (in elements, >= C, multiple of 8) lives at
 *    base[((n*H + y)*W + x)*ld + c].  A channel slice of a wider buffer is the
 *    same thing with an offset base pointer, so concatenations are never copied.
 *  - `dtype` selects the activation storage type: EGM_F32 (parity path) or
 *    EGM_BF16 (throughput path).  All accumulation is fp32.  Parameters,
 *    statistics, gradients of parameters and optimizer state are always fp32.
 *  - channel counts of activation tensors are multiples of 8 (callers pad with
 *    zero channels; packers zero the matching weights).
 */
#ifndef EGM_HIP_H
#define EGM_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef void* egm_stream_t; /* hipStream_t */

enum egm_status { EGM_OK = 0, EGM_ERR_ARG = -1, EGM_ERR_LAUNCH = -2, EGM_ERR_UNSUPPORTED = -3 };
enum egm_dtype { EGM_F32 = 0, EGM_BF16 = 1 };
enum egm_act { EGM_ACT_NONE = 0, EGM_ACT_RELU = 1, EGM_ACT_SIGMOID = 2 };

int egm_version(void);
const char* egm_last_error(void);
/* 1 when the current HIP device is gfx950; 0 otherwise (message set). */
int egm_device_ok(void);

/* ---- layout conversion at the module boundary ------------------------------------------------
 * image  [N,C,H,W] fp32 -> NHWC dtype with ld channels (channels C..ld-1 zero-filled);
 * logits NHWC dtype -> [N,C,H,W] fp32.  (model(image)["out"], train_and_eval.py:58) */
int egm_nchw_to_nhwc(int dtype, const void* src_f32, void* dst, int ld, int N, int C, int H, int W, egm_stream_t s);
int egm_nhwc_to_nchw(int dtype, const void* src, int ld, void* dst_f32, int N, int C, int H, int W, egm_stream_t s);

/* ---- convolution (nn.Conv2d stride 1, 'same' padding = dil*(k-1)/2; src/EGM-UNet.py:49,893,964,1210-1218) ---- */
/* Pack fp32 OIHW weights [Cout][Cin/groups][KH][KW] into the two dense operand layouts the kernels read:
 *   wf [KH*KW][CoutP][CinP]  (forward;  CoutP/CinP = Cout/Cin rounded up to 8, block-diagonal for groups>1)
 *   wd [KH*KW][CinP][CoutP]  (data gradient: taps flipped, in/out swapped).  Either output may be NULL. */
int egm_conv_pack(int dtype, const void* w_oihw_f32, void* wf, void* wd, int Cout, int Cin, int KH, int KW, int groups,
                  egm_stream_t s);
/* y = conv(x, wf) (+bias).  Cin/Cout are the PADDED counts of wf.  bias (fp32, bias_n <= Cout valid entries; the rest count as 0) may be NULL.
 * stats, when non-NULL, receives per-pixel-tile partial sums [ntiles][2][Cout] of y and y*y
 * (consumed by egm_bn_finalize); egm_conv_stats_tiles() gives ntiles.
 * The data gradient is the same call on dy with wd (Cin/Cout swapped). */
int egm_conv_fwd(int dtype, const void* x, int ldx, const void* wf, const void* bias_f32, int bias_n, void* y, int ldy,
                 float* stats, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, egm_stream_t s);
int egm_conv_stats_tiles(int N, int H, int W);
/* Weight gradient: dw_oihw_f32 [CoutR][CinR/groups][KH][KW] (+)= sum_pixels dy (x) x.
 * Cin/Cout are padded counts of the activation buffers, CinR/CoutR the real (unpadded) ones.
 * workspace: egm_conv_wgrad_workspace() bytes.  accumulate != 0 adds to dw. */
long long egm_conv_wgrad_workspace(int N, int H, int W, int Cin, int Cout, int KH, int KW);
int egm_conv_wgrad(int dtype, const void* x, int ldx, const void* dy, int lddy, float* dw_oihw_f32, void* workspace,
                   int N, int H, int W, int Cin, int Cout, int CinR, int CoutR, int KH, int KW, int dil, int groups,
                   int accumulate, egm_stream_t s);
/* Depthwise 3x3 (RecursiveGatedAttention.dwconv, src/EGM-UNet.py:507-509): y = (dw3x3(x, w) + b) * scale.
 * w fp32 [C][1][3][3], b fp32 [C], scale fp32 [1] (device). */
int egm_dwconv3_fwd(int dtype, const void* x, int ldx, const float* w, const float* b, const float* scale, void* y, int ldy,
                    int N, int H, int W, int C, egm_stream_t s);
/* dx, dw, db, dscale of the above.  partial: fp32 scratch >= egm_dwconv3_bwd_workspace() bytes. */
long long egm_dwconv3_bwd_workspace(int N, int H, int W, int C);
int egm_dwconv3_bwd(int dtype, const void* x, int ldx, const void* dy, int lddy, const float* w, const float* b,
                    const float* scale, void* dx, int lddx, float* dw, float* db, float* dscale, void* workspace,
                    int N, int H, int W, int C, egm_stream_t s);

/* ---- per-channel reductions ------------------------------------------------------------------
 * Two-stage, deterministic: stage 1 writes per-block partial "tiles" [nblk][2][C] (sum, sum of squares) with
 * nblk = egm_channel_partials_blocks(npix, C); egm_reduce_tiles() (or egm_bn_finalize) sums tiles in fixed order
 * in double.  Used for bias gradients and for BN statistics of tensors not produced by egm_conv_fwd. */
int egm_channel_partials_blocks(long long npix, int C);
int egm_channel_sums(int dtype, const void* x, int ld, long long npix, int C, float* partials, egm_stream_t s);
int egm_reduce_tiles(const float* tiles, int ntiles, int C, float* out_2xC, egm_stream_t s);

/* ---- BatchNorm2d (train-mode batch statistics / eval-mode running statistics) + activation -----
 * (nn.BatchNorm2d + ReLU/Sigmoid: src/EGM-UNet.py:50-51,878-879,966-973) */
/* stats tiles [ntiles][2][C] -> mean/var; writes scale=gamma*rstd, shift=beta-mean*scale, save_mean, save_rstd (fp32 [C]);
 * when running_mean != NULL updates running stats with `momentum` (unbiased variance, as torch).
 * gamma/beta/running_* hold C_real entries; channels C_real..C-1 (zero padding) get scale = shift = 0. */
int egm_bn_finalize(const float* stats, int ntiles, long long count, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                    float* save_mean, float* save_rstd, int C, int C_real, egm_stream_t s);
/* eval mode: scale/shift (and save_mean/save_rstd when non-NULL) from running statistics. */
int egm_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, float* scale, float* shift, float* save_mean, float* save_rstd, int C, int C_real,
                       egm_stream_t s);
/* z = act(y*scale + shift) */
int egm_bn_act_fwd(int dtype, const void* y, int ldy, const float* scale, const float* shift, int act, void* z, int ldz,
                   long long npix, int C, egm_stream_t s);
/* Backward of z = act(BN(y)), dzp = dz*act'(y*scale+shift), xhat = (y-mean)*rstd:
 *   reduce: partial tiles [nblk][2][C] of (sum dzp, sum dzp*xhat)  -> egm_reduce_tiles -> sums [2][C]
 *           (sums[0] is dbeta, sums[1] is dgamma)
 *   apply : train != 0: dy = scale*(dzp - sums[0]/npix - xhat*sums[1]/npix);  train == 0: dy = scale*dzp */
int egm_bn_act_bwd_reduce(int dtype, const void* dz, int lddz, const void* y, int ldy, const float* scale,
                          const float* shift, const float* save_mean, const float* save_rstd, int act, float* partials,
                          long long npix, int C, egm_stream_t s);
int egm_bn_act_bwd_apply(int dtype, const void* dz, int lddz, const void* y, int ldy, const float* scale,
                         const float* shift, const float* save_mean, const float* save_rstd, int act, int train,
                         const float* sums, void* dy, int lddy, long long npix, int C, egm_stream_t s);

/* ---- pooling / resampling --------------------------------------------------------------------- */
/* nn.MaxPool2d(2,2) (src/EGM-UNet.py:908); H, W are the INPUT sizes (even). */
int egm_maxpool2_fwd(int dtype, const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, egm_stream_t s);
int egm_maxpool2_bwd(int dtype, const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, int N, int H, int W,
                     int C, egm_stream_t s);
/* Up.forward front half (src/EGM-UNet.py:937-947): out = cat([skip, pad(bilinear_x2_align_corners(low))], C).
 * Writes BOTH halves of `out` (ld = ldo >= Cs + Cl): skip [N,Hs,Ws,Cs], low [N,Hl,Wl,Cl]. */
int egm_upcat_fwd(int dtype, const void* skip, int lds, const void* low, int ldl, void* out, int ldo, int N, int Hs, int Ws,
                  int Cs, int Hl, int Wl, int Cl, egm_stream_t s);
/* dlow = transposed bilinear of dout[..., Cs:Cs+Cl]; (dskip is dout[..., :Cs], a view). */
int egm_upcat_bwd_low(int dtype, const void* dout, int ldo, void* dlow, int ldl, int N, int Hs, int Ws, int Cs, int Hl,
                      int Wl, int Cl, egm_stream_t s);

/* ---- generic fused elementwise helpers -------------------------------------------------------- */
/* out = alpha*a + beta*b (b may be NULL) */
int egm_axpby(int dtype, const void* a, int lda, float alpha, const void* b, int ldb, float beta, void* out, int ldo,
              long long npix, int C, egm_stream_t s);
/* fp32 vector add: y[i] += x[i] (parameter-gradient accumulation) */
int egm_vec_add_f32(float* y, const float* x, long long n, egm_stream_t s);
int egm_fill_f32(float* y, float v, long long n, egm_stream_t s);

/* ---- criterion, metrics, optimizer --------------------------------------------------------------
 * criterion(): train_utils/train_and_eval.py:7-19 + dice_coefficient_loss.py:7-108 (five terms; reference quirks kept:
 * stencils on raw logit channel 0 against the label map of sample 0).  logits/dlogits fp32 NCHW, target int64 [N,H,W].
 * loss6 = {total, ce, dice, laplace, lap, sobel} (device).  workspace: egm_loss_workspace() bytes, kept for the
 * backward; signs: N*H*W bytes (stencil sign codes), needed when dice != 0. */
long long egm_loss_workspace(int N, int C);
int egm_loss_fwd(const float* logits, const long long* target, const float* class_weight, int N, int C, int H, int W,
                 long long ignore_index, int dice, float* loss6, float* workspace, unsigned char* signs, egm_stream_t s);
int egm_loss_bwd(const float* logits, const long long* target, const float* class_weight, int N, int C, int H, int W,
                 long long ignore_index, int dice, const float* workspace, const unsigned char* signs, const float* grad_out,
                 float* dlogits, egm_stream_t s);
/* ConfusionMatrix.update + DiceCoefficient.update (train_utils/distributed_utils.py:81-91,135-144):
 * hist[C*C] += bincount(C*t + argmax); counts[N][C][3] += (inter, pred, tgt) one-hot counts over t != dice_ignore_index;
 * pred (int64 [N,H,W]) optional.  hist/counts are zeroed by the caller. */
int egm_argmax_hist(const float* logits, const long long* target, int N, int C, int H, int W, long long dice_ignore_index,
                    unsigned long long* hist, unsigned long long* counts, long long* pred, egm_stream_t s);
/* out[2+2C] = {dice (classes 1.., mean over images), acc_global, acc[C], iu[C]} (distributed_utils.py:97-105,147-151) */
int egm_metrics_finalize(const unsigned long long* hist, const unsigned long long* counts, int N, int C, float* out,
                         egm_stream_t s);
/* torch.optim.SGD(momentum, weight_decay) (train.py:115-118) over a device table of {float* p; const float* g; float* buf;
 * long long n;} entries: g' = g*grad_scale + wd*p; v = first_step ? g' : mu*v + g'; p -= lr*v.
 * lr_dev (device scalar) overrides lr when non-NULL. */
int egm_sgd_multi(const void* table_dev, int ntensors, const float* lr_dev, float lr, float momentum, float weight_decay,
                  float grad_scale, int first_step, egm_stream_t s);
/* table of {float* dst; const float* src; long long n;}: gradient bucket gather/scatter for the RCCL all-reduce. */
int egm_copy_multi(const void* table_dev, int ntensors, egm_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* EGM_HIP_H */
