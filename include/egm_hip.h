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
enum egm_act { EGM_ACT_NONE = 0, EGM_ACT_RELU = 1, EGM_ACT_SIGMOID = 2, EGM_ACT_SILU = 3 /* x*sigmoid(x): Conv, src/EGM-UNet.py:25-43 */ };

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
 *   wd [KH*KW][CinP][CoutP]  (data gradient: taps flipped, in/out swapped).  Either output may be NULL.
 * bf16 3x3 images whose CoutP and CinP are multiples of 16 are stored chunk-major instead, [tap][CinP/16][CoutP][16] (wd:
 * [tap][CoutP/16][CinP][16]): same size, the 16-channel weight slab of a cout tile is contiguous per tap, which is what the LDS-DMA
 * staging of the 3x3 kernel reads.  The layout is a function of (dtype, KH, KW, CinP, CoutP) alone; egm_conv_fwd derives it from its
 * own arguments, callers never see it. */
int egm_conv_pack(int dtype, const void* w_oihw_f32, void* wf, void* wd, int Cout, int Cin, int KH, int KW, int groups,
                  egm_stream_t s);
/* Every conv weight of a model in one launch.  table_dev: device array of 56-byte entries
 * {const float* w; void* wf; void* wd; int Cout, Cin, CoutP, CinP, KH, KW, groups, chunk0;}; an entry takes
 * ceil(KH*KW*CoutP*CinP / egm_conv_pack_chunk()) workgroups ("chunks"), chunk0 = the sum of the chunk counts of the entries before
 * it (the kernels of all *_multi entry points find their entry by binary search on chunk0), total_chunks = the sum over all. */
int egm_conv_pack_chunk(void);
int egm_conv_pack_multi(int dtype, const void* table_dev, int n, long long total_chunks, egm_stream_t s);
/* Launch groups: between egm_group_begin() and egm_group_end(s) the egm_conv_fwd* calls of THIS host thread are recorded instead of
 * launched; egm_group_end launches them, merging those that run the same kernel instantiation into one launch (up to 4 members; the
 * parallel branches of EdgeEnhancedGRFB, src/EGM-UNet.py:1256-1278).  The recorded convolutions must be independent of each other
 * and their outputs must not be used before egm_group_end.  Convolutions taking a kernel without a merged form are launched at once.
 * egm_group_abort() drops an open group (error paths). */
int egm_group_begin(void);
int egm_group_end(egm_stream_t s);
int egm_group_abort(void);
/* 3x3 kernel selection, a bit mask (default 5; env EGM_CONV_TILE): 1 = the 8-wave LDS-DMA tile kernel wherever its >= 64-cout tiles
 * fill the chip, 2 = its 32-cout tiles too (measured slower than the 4-wave kernel: parity tests and A/B runs only), 4 = the
 * weights-in-registers kernel for the 32 -> 32 layers; 0 = the 4-wave register-staged kernel everywhere; -1 = query only.
 * Returns the previous mode (A/B timing and parity tests). */
int egm_conv_tile_mode(int mode);
/* Diagnostics only (tools/conv_tile_diag.py): phase-elimination switches of the tile kernel, OR of 1 = no LDS-DMA, 2 = no MFMA phase,
 * 4 = no epilogue; outputs are wrong while any is set.  -1 = query.  Returns the previous value; 0 in production. */
int egm_conv_tile_debug(int dbg);
/* 7x7 kernel selection for 16 -> 16 channels (FusionConv's merged 3x3+5x5+7x7 conv at the 64-channel level, src/EGM-UNet.py:1210-1228):
 * 1 = the weights-in-registers v_mfma_f32_16x16x32_bf16 kernel (csrc/conv7x7_c16.hip; default, env EGM_CONV_C7), 0 = the generic
 * pipelined kernel, -1 = query only.  Returns the previous mode (A/B timing and parity tests). */
int egm_conv_c7_mode(int mode);
/* egm_conv_fwd with the output channels written to TWO tensors: couts [0, csplit) to y (pixel stride ldy), [csplit, Cout) to y2
 * (pixel stride ldy2).  The data gradient of the conv behind a channel concatenation (Up, src/EGM-UNet.py:947-949) written as the
 * gradients of the two concatenated tensors, each dense, instead of one interleaved tensor whose halves the consumers read as half
 * cache lines.  No bias, no statistics.  Supported where egm_conv_split_ok() says so (the 8-wave 3x3 kernel, csplit % 8 == 0). */
int egm_conv_split_ok(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, int csplit);
int egm_conv_fwd_split(int dtype, const void* x, int ldx, const void* wf, void* y, int ldy, void* y2, int ldy2, int csplit, int N,
                       int H, int W, int Cin, int Cout, int KH, int KW, int dil, egm_stream_t s);
/* Name of the kernel egm_conv_fwd launches for a shape, spelled like the rows of a rocprofv3 kernel trace (e.g.
 * "conv_igemm_pipe_kernel<2, 3, 3, 2>"); returns its length, copies at most buflen-1 characters into buf (may be NULL). */
int egm_conv_kernel_name(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, char* buf, int buflen);
/* y = conv(x, wf) (+bias).  Cin/Cout are the PADDED counts of wf.  bias (fp32, bias_n <= Cout valid entries; the rest count as 0) may be NULL.
 * stats, when non-NULL, receives per-pixel-tile partial sums [ntiles][2][Cout] of y and y*y
 * (consumed by egm_bn_finalize); egm_conv_stats_tiles() gives ntiles for the same dtype/shape/kernel.
 * The data gradient is the same call on dy with wd (Cin/Cout swapped). */
int egm_conv_fwd(int dtype, const void* x, int ldx, const void* wf, const void* bias_f32, int bias_n, void* y, int ldy,
                 float* stats, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, egm_stream_t s);
int egm_conv_stats_tiles(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil);
/* Weight gradient: dw_oihw_f32 [CoutR][CinR/groups][KH][KW] (+)= sum_pixels dy (x) x.
 * Cin/Cout are padded counts of the activation buffers, CinR/CoutR the real (unpadded) ones.
 * workspace: egm_conv_wgrad_workspace() bytes.  accumulate != 0 adds to dw. */
long long egm_conv_wgrad_workspace(int N, int H, int W, int Cin, int Cout, int KH, int KW);
int egm_conv_wgrad(int dtype, const void* x, int ldx, const void* dy, int lddy, float* dw_oihw_f32, void* workspace,
                   int N, int H, int W, int Cin, int Cout, int CinR, int CoutR, int KH, int KW, int dil, int groups,
                   int accumulate, egm_stream_t s);
/* Deferred form: egm_conv_wgrad with dw == NULL writes only the partial slabs (egm_conv_wgrad_slabs() of them) into the
 * workspace; egm_wgrad_reduce_multi() then finishes MANY convolutions in one launch.  table_dev: device array of 56-byte
 * entries {const float* slab; float* dw; int nslab, taps, CoutP, CinP, CoutR, CinR, groups, accumulate, chunk0, pad;};
 * chunks per entry = ceil(taps*CoutP*CinP / egm_wgrad_reduce_chunk()), chunk0 / total_chunks as for egm_conv_pack_multi. */
int egm_conv_wgrad_kernel_name(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil, char* buf, int buflen);   /* as egm_conv_kernel_name */
/* egm_conv_wgrad's slab-only form (dw == NULL) for 1..EGM_WGRAD_MULTI_MAX independent convolutions in one call: members that take the
 * same kernel instantiation (egm_conv_wgrad_kernel_name) run as ONE launch.  Arguments per member as egm_conv_wgrad's; slabs =
 * its workspace.  (The reference has no counterpart: torch.autograd launches cudnn's weight-gradient kernel per conv, train_utils'
 * loss.backward(); here the slabs of a backward pass have no reader before egm_wgrad_reduce_multi, so their launches can wait.) */
#define EGM_WGRAD_MULTI_MAX 4
typedef struct egm_conv_wgrad_desc {
    const void* x; const void* dy; void* slabs;
    int ldx, lddy, N, H, W, Cin, Cout, Cin_real, Cout_real, KH, KW, dil, groups, pad_;
} egm_conv_wgrad_desc;
int egm_conv_wgrad_multi(int dtype, const egm_conv_wgrad_desc* descs, int n, egm_stream_t stream);
int egm_conv_wgrad_slabs(int dtype, int N, int H, int W, int Cin, int Cout, int KH, int KW, int dil);
/* The reduction of ONE convolution's slabs, at once (what egm_conv_wgrad with dw != NULL runs behind its slab kernel). */
int egm_wgrad_reduce(const float* slabs, float* dw_oihw_f32, int nslab, int taps, int CoutP, int CinP, int CoutR, int CinR, int groups,
                     int accumulate, egm_stream_t s);
int egm_wgrad_reduce_chunk(void);   /* packed elements reduced by one workgroup of egm_wgrad_reduce_multi */
int egm_wgrad_reduce_multi(const void* table_dev, int n, long long total_chunks, egm_stream_t s);
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
/* batch independent reductions: tiles [batch][ntiles][2][C] -> out [batch][2][C] */
int egm_reduce_tiles_batched(const float* tiles, int batch, int ntiles, int C, float* out, egm_stream_t s);
/* The bias gradients (db[c] = sum over pixels of dy[., c]) of up to a whole backward pass in two launches; bit-identical, per tensor, to
 * egm_channel_sums + egm_reduce_tiles (same blocks, same order).  Replaces the per-layer bias reduction autograd performs for the
 * nn.Conv2d(bias=True) layers of src/EGM-UNet.py:1256-1313, 1362-1390, 1499 (torch: grad_bias = dy.sum((0, 2, 3))).
 * table_dev: device array of 2n egm_bsum_entry; [0, n): chunk0 = index of the tensor's first partial-sum block (nblk =
 * egm_channel_partials_blocks(npix, C) blocks each, blocks1 in all); [n, 2n): the same tensors with chunk0 = index of the first
 * reduction block (C / 8 each, blocks2 in all).  part = nblk * 2 * C floats of workspace, out = Cout floats (c < Cout <= C written). */
typedef struct { const void* x; float* part; float* out; long long npix; int ld, C, Cout, nblk, chunk0, pad; } egm_bsum_entry;
int egm_bias_grad_multi(int dtype, const void* table_dev, int n, long long blocks1, long long blocks2, egm_stream_t s);

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
/* BatchNorm backward with dz computed on the fly from its producer (csrc/bn_dz_fused.hip), the two passes of egm_bn_act_bwd_reduce /
 * egm_bn_act_bwd_apply (same partials geometry: egm_channel_partials_blocks(npix, C) blocks; C/8 must divide 256):
 *   _cls_: dz[p][c] = sum_{k < nc} dlogits[p][k]*w_cls[k*ldw + c] -- the data gradient of the 1x1 classifier behind the last
 *          BatchNorm+ReLU (OutConv, src/EGM-UNet.py:952-956): its conv launch and the tensor it wrote do not exist.  nc <= 8; w_cls is the
 *          fp32 OIHW weight [nc][ldw] (rounded to `dtype` inside, like the conv's operand pack); dlogits has >= 8 channels per pixel.
 *   _mca_: dz = dxo*(g_h + g_w + g_c)*inv + (A + B*x) -- the MCALayer's last backward step (egm_mca_bwd_dx) behind DoubleConv1's first
 *          BatchNorm+ReLU (src/EGM-UNet.py:893-896), x being this BatchNorm's own output, recomputed from y. */
/* Forward twin of _cls_: z = act(y*scale + shift) written to z AND the classifier applied in the same pass,
 * logits_nchw[n][k][h][w] (fp32, the module's output layout) = sum_c z[c]*w_cls[k*ldw + c] + bias[k], rounded to `dtype` like the conv
 * kernel's output (egm_bn_act_fwd + the 1x1 conv launch + egm_nhwc_to_nchw as one pass).  nc <= 8, C/8 a power of two <= 64. */
int egm_bn_act_cls_fwd(int dtype, const void* y, int ldy, const float* scale, const float* shift, int act, void* z, int ldz,
                       const float* w_cls, int nc, int ldw, const float* bias, float* logits_nchw, int N, int H, int W, int C,
                       egm_stream_t s);
int egm_bn_cls_bwd_reduce(int dtype, const void* dlogits, int lddl, const float* w_cls, int nc, int ldw, const void* y, int ldy,
                          const float* scale, const float* shift, const float* save_mean, const float* save_rstd, int act,
                          float* partials, long long npix, int C, egm_stream_t s);
int egm_bn_cls_bwd_apply(int dtype, const void* dlogits, int lddl, const float* w_cls, int nc, int ldw, const void* y, int ldy,
                         const float* scale, const float* shift, const float* save_mean, const float* save_rstd, int act, int train,
                         const float* sums, void* dy, int lddy, long long npix, int C, egm_stream_t s);
int egm_bn_mca_bwd_reduce(int dtype, const void* dxo, int ldd, const float* gates, const float* coef, int no_spatial, const void* y,
                          int ldy, const float* scale, const float* shift, const float* save_mean, const float* save_rstd, int act,
                          float* partials, int N, int H, int W, int C, egm_stream_t s);
int egm_bn_mca_bwd_apply(int dtype, const void* dxo, int ldd, const float* gates, const float* coef, int no_spatial, const void* y,
                         int ldy, const float* scale, const float* shift, const float* save_mean, const float* save_rstd, int act,
                         int train, const float* sums, void* dy, int lddy, int N, int H, int W, int C, egm_stream_t s);
/* Second stage of the BatchNorm backward: partial tiles of egm_bn_act_bwd_reduce -> sums [2][C] (dbeta | dgamma) and
 * cf [4][C] = scale | shift | cb | cc with dy = scale*dzp + cb + cc*y (what the fused apply passes read). */
int egm_bn_bwd_coefs(const float* partials, int ntiles, long long count, const float* scale, const float* shift,
                     const float* save_mean, const float* save_rstd, int train, float* sums_2xC, float* cf_4xC, int C,
                     egm_stream_t s);

/* Multi-tensor forms: the BatchNorm passes of up to EGM_BN_MULTI_MAX INDEPENDENT layers in one launch each (the parallel branches of
 * EdgeEnhancedGRFB, src/EGM-UNet.py:1256-1278, work on 8-32 channel tensors whose passes are launch-latency bound).  `descs` is a HOST
 * array (it is copied into the kernel argument); every pass reads the fields it needs and ignores the others; coef = [4][C] rows
 * scale | shift | save_mean | save_rstd.  Same arithmetic and block decomposition per tensor as the single-tensor entry points. */
#define EGM_BN_MULTI_MAX 4
enum egm_bn_multi_pass { EGM_BN_MULTI_FINALIZE = 0, EGM_BN_MULTI_FWD = 1, EGM_BN_MULTI_BWD_REDUCE = 2, EGM_BN_MULTI_BWD_COEFS = 3,
                         EGM_BN_MULTI_BWD_APPLY = 4 };
typedef struct egm_bn_desc {
    const void* y;  void* z;  const void* dz;  void* dy;        /* BatchNorm input, output, gradient of the output, of the input */
    float* coef;                                                  /* [4][C] */
    const float* stats;                                           /* finalize: conv partial tiles [ntiles][2][C] */
    const float* gamma;  const float* beta;  float* running_mean;  float* running_var;
    float* partials;  float* sums;  float* cf4;                   /* backward: [nblocks][2][C], [2][C], [4][C] */
    long long npix;
    int ldy, ldz, lddz, lddy, ntiles, nblocks, C, C_real, act, train;
    float eps, momentum;
} egm_bn_desc;
int egm_bn_multi(int dtype, int pass, const egm_bn_desc* descs, int n, egm_stream_t s);

/* BatchNorm(+act) fused with the element-wise op behind it (z and dz never reach memory; same rounding points as the unfused
 * chain, so results agree bit for bit):
 *   EGM_EW_GATE  out = p*(1 + z)          EdgeAwareFeatureEnhancer, src/EGM-UNet.py:884-886 (z = sigmoid(BN(y)))
 *   EGM_EW_SAR   out = relu(alpha*p + z)  EdgeEnhancedGRFB residual tail, :1315-1317 (z = BN(y) of the shortcut, p = fusion output)
 * fwd: reads y, p, writes out.  bwd_reduce: partial tiles [egm_channel_partials_blocks][2][C] of the BatchNorm backward sums, from
 * g (= dL/dout), q (GATE: p; SAR: out, for the ReLU mask) and y.  bwd_apply (cf from egm_bn_bwd_coefs): writes dy (gradient of the
 * BatchNorm input) and dp. */
enum egm_ew_mode { EGM_EW_GATE = 0, EGM_EW_SAR = 1 };
int egm_bn_ew_fwd(int dtype, int mode, const void* y, int ldy, const float* scale, const float* shift, int act, const void* p,
                  int ldp, float alpha, void* out, int ldo, long long npix, int C, egm_stream_t s);
int egm_bn_ew_bwd_reduce(int dtype, int mode, const void* g, int ldg, const void* q, int ldq, const void* y, int ldy,
                         const float* scale, const float* shift, const float* save_mean, const float* save_rstd, int act,
                         float alpha, float* partials, long long npix, int C, egm_stream_t s);
int egm_bn_ew_bwd_apply(int dtype, int mode, const void* g, int ldg, const void* q, int ldq, const void* y, int ldy,
                        const float* cf_4xC, int act, float alpha, void* dy, int lddy, void* dp, int lddp, long long npix, int C,
                        egm_stream_t s);

/* ---- 1x1 conv -> BatchNorm -> activation (-> element-wise consumer) from the INPUT's moments (csrc/pw_bn.hip) -------------------
 * BasicConv(k = 1) / EdgeAwareFeatureEnhancer / the GRFB shortcut (src/EGM-UNet.py:872-886, 958-975, 1256-1278, 1296-1317).  The batch
 * statistics of y = W x follow from S = sum x and G = sum x x^T, so the conv output never reaches memory:
 *   egm_pw_moments   x -> per-wave partial Gram blocks and channel sums           (once per input; heads on the same x share it)
 *   egm_pw_fwd_coefs partials -> covariance / mean (double) -> coef rows [4][CoutP] scale | shift | mean | rstd of y' = W x per head
 *                    (conv bias folded in), running statistics updated as nn.BatchNorm2d does; train = 0: from the running statistics
 *   egm_pw_fwd       out = F(p, act(scale * (W x) + shift)),  F: mode 0 identity, 1 GATE p*(1+z), 2 SAR relu(alpha*p + z)
 *   egm_pw_bwd_reduce  from x, g = dL/dout, q (GATE: p, SAR: out): partials of s0 = sum dzp, s1 = sum dzp*xhat, M = sum dzp x^T
 *   egm_pw_bwd_coefs   -> sums [2][CoutP] (dbeta | dgamma), cf4 [4][CoutP] (scale | shift | cb | cc), dw [Cout][Cin_real] =
 *                      sc*M + cb*S^T + cc*(W G) (the weight gradient in closed form), dbias (zero under batch statistics)
 *   egm_pw_bwd_apply   dy = sc*dzp + cb + cc*y' per element; dx = sum over the heads on x of W^T dy; dp of the element-wise consumer
 * One egm_pw_head per conv; heads with the same x pointer form one problem (<= 2 heads, stacked CoutP <= 128, Cin <= 128) and
 * share mom_partials / cov / mu / bwd_partials / dx, read from the first of them; <= 4 problems per call (one launch per kernel).
 * w = egm_conv_pack's wf [CoutP][Cin], wd its wd [Cin][CoutP] (1x1, groups 1), both in `dtype`.  All pointers device pointers. */
typedef struct egm_pw_head {
    const void* x;  const void* w;  const void* wd;
    const float* bias;  const float* gamma;  const float* beta;  float* running_mean;  float* running_var;
    float* coef;                                   /* [4][CoutP] */
    const void* p;  void* out;                     /* forward: element-wise partner (mode != 0), output */
    const void* g;  const void* q;  void* dp;      /* backward: dL/dout, GATE: p / SAR: out, gradient of p (may be NULL) */
    float* sums;  float* cf4;  float* dw;  float* dbias;
    float* mom_partials;  double* cov;  double* mu; /* egm_pw_moments_floats floats; [Cin][Cin]; [Cin] */
    float* bwd_partials;  void* dx;                /* egm_pw_bwd_floats floats; gradient of x (may be NULL) */
    long long npix;
    int ldx, lddx, Cin, Cin_real, Cout, CoutP, act, mode, ldp, ldo, ldg, ldq, lddp, train;
    float alpha, eps, momentum;
} egm_pw_head;
int egm_pw_supported(int dtype, int Cin, int CoutTot, int heads_on_input);
int egm_pw_moments_parts(long long npix);
long long egm_pw_moments_floats(long long npix, int Cin);
int egm_pw_bwd_parts(int dtype, long long npix, int Cin, int CoutTot);
long long egm_pw_bwd_floats(int dtype, long long npix, int Cin, int CoutTot);
int egm_pw_moments(int dtype, const egm_pw_head* heads, int n, egm_stream_t s);
int egm_pw_fwd_coefs(int dtype, const egm_pw_head* heads, int n, egm_stream_t s);
int egm_pw_fwd(int dtype, const egm_pw_head* heads, int n, egm_stream_t s);
int egm_pw_bwd_reduce(int dtype, const egm_pw_head* heads, int n, egm_stream_t s);
int egm_pw_bwd_coefs(int dtype, const egm_pw_head* heads, int n, egm_stream_t s);
int egm_pw_bwd_apply(int dtype, const egm_pw_head* heads, int n, egm_stream_t s);

/* Backward of a 1x1 convolution (groups 1) in ONE pass over dy and x (csrc/pw_bn.hip): dx = dy . W (data gradient, may be NULL) and
 * the weight-gradient slabs [egm_conv1x1_bwd_slabs(npix)][CoutP][CinP] fp32, laid out like egm_conv_wgrad's workspace (taps = 1) so that
 * egm_wgrad_reduce_multi / egm_conv_wgrad's own reduction sum them.  wd = egm_conv_pack's wd [CinP][CoutP] in `dtype`; Cin, Cout are the
 * PADDED channel counts (<= 128).  Replaces the egm_conv_fwd(dy, wd) + egm_conv_wgrad pair of BasicConv(k = 1) and friends
 * (src/EGM-UNet.py:958-975, 1210-1218).  Up to 4 convolutions per call (one launch). */
typedef struct egm_conv1x1_bwd_desc {
    const void* x;  const void* dy;  const void* wd;  void* dx;  float* slabs;
    long long npix;
    int ldx, lddy, lddx, Cin, Cout, pad_;
} egm_conv1x1_bwd_desc;
int egm_conv1x1_bwd_supported(int dtype, int Cin, int Cout);
int egm_conv1x1_bwd_slabs(long long npix);
int egm_conv1x1_bwd(int dtype, const egm_conv1x1_bwd_desc* descs, int n, egm_stream_t s);

/* ---- pooling / resampling --------------------------------------------------------------------- */
/* nn.MaxPool2d(2,2) (src/EGM-UNet.py:908); H, W are the INPUT sizes (even). */
int egm_maxpool2_fwd(int dtype, const void* x, int ldx, void* y, int ldy, int N, int H, int W, int C, egm_stream_t s);
int egm_maxpool2_bwd(int dtype, const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, int N, int H, int W,
                     int C, egm_stream_t s);
/* egm_maxpool2_bwd with a second gradient of x summed in the same pass: dx = scatter(dy) + add (even H, W). */
int egm_maxpool2_bwd_add(int dtype, const void* x, int ldx, const void* dy, int lddy, const void* add, int ldadd, void* dx, int lddx,
                         int N, int H, int W, int C, egm_stream_t s);
/* The max pool at an encoder skip connection fused into its neighbours (csrc/pool_fused.hip; src/EGM-UNet.py:908 behind :44-55 at the
 * top level and behind the EdgeEnhancedGRFB target gate :1319-1321 below it).  H, W = the full-resolution sizes, both even.
 *   egm_bn_act_fwd_pool   : z = act(y*scale + shift) AND pooled = maxpool2(z)                      (egm_bn_act_fwd + egm_maxpool2_fwd)
 *   egm_bn_pool_bwd_reduce: egm_bn_act_bwd_reduce with dz = gskip + scatter(gpool) computed on the fly (egm_maxpool2_bwd_add never runs);
 *                           partials [egm_bn_pool_bwd_blocks()][2][C]
 *   egm_bn_pool_bwd_apply : egm_bn_act_bwd_apply with the same on-the-fly dz
 *   egm_gate3_fwd_pool    : out = x*(1 + mean_k sigmoid(t[k])) AND pooled = maxpool2(out)           (egm_gate3_fwd + egm_maxpool2_fwd)
 *   egm_gate3_pool_bwd    : egm_gate3_bwd with g = gskip + scatter(gpool) computed on the fly */
int egm_bn_act_fwd_pool(int dtype, const void* y, int ldy, const float* scale, const float* shift, int act, void* z, int ldz,
                        void* pooled, int ldp, int N, int H, int W, int C, egm_stream_t s);
int egm_bn_pool_bwd_blocks(int N, int H, int W, int C);
int egm_bn_pool_bwd_reduce(int dtype, const void* gskip, int ldgs, const void* gpool, int ldgp, const void* y, int ldy,
                           const float* scale, const float* shift, const float* save_mean, const float* save_rstd, int act,
                           float* partials, int N, int H, int W, int C, egm_stream_t s);
int egm_bn_pool_bwd_apply(int dtype, const void* gskip, int ldgs, const void* gpool, int ldgp, const void* y, int ldy,
                          const float* scale, const float* shift, const float* save_mean, const float* save_rstd, int act, int train,
                          const float* sums, void* dy, int lddy, int N, int H, int W, int C, egm_stream_t s);
int egm_gate3_fwd_pool(int dtype, const void* x, int ldx, const void* t, int ldt, void* out, int ldo, void* pooled, int ldp, int N,
                       int H, int W, int C, egm_stream_t s);
int egm_gate3_pool_bwd(int dtype, const void* gskip, int ldgs, const void* gpool, int ldgp, const void* x, int ldx, const void* t,
                       int ldt, void* dx, int lddx, void* dt, int lddt, int N, int H, int W, int C, egm_stream_t s);
/* Up.forward front half (src/EGM-UNet.py:937-947): out = cat([skip, pad(bilinear_x2_align_corners(low))], C).
 * Writes BOTH halves of `out` (ld = ldo >= Cs + Cl): skip [N,Hs,Ws,Cs], low [N,Hl,Wl,Cl].
 * skip == NULL: the skip channels were produced straight into out[..., :Cs] by their own kernel; only the Cl upsampled channels are written. */
int egm_upcat_fwd(int dtype, const void* skip, int lds, const void* low, int ldl, void* out, int ldo, int N, int Hs, int Ws,
                  int Cs, int Hl, int Wl, int Cl, egm_stream_t s);
/* dlow = transposed bilinear of dout[..., Cs:Cs+Cl]; (dskip is dout[..., :Cs], a view). */
int egm_upcat_bwd_low(int dtype, const void* dout, int ldo, void* dlow, int ldl, int N, int Hs, int Ws, int Cs, int Hl,
                      int Wl, int Cl, egm_stream_t s);

/* ---- generic fused elementwise helpers -------------------------------------------------------- */
/* out = alpha*a + beta*b (b may be NULL) */
int egm_axpby(int dtype, const void* a, int lda, float alpha, const void* b, int ldb, float beta, void* out, int ldo,
              long long npix, int C, egm_stream_t s);
/* out = a + b + c (+ d when non-NULL): gradient fan-in of a tensor with 3-4 consumers in one pass */
int egm_sum4(int dtype, const void* a, int lda, const void* b, int ldb, const void* c, int ldc, const void* d, int ldd, void* out,
             int ldo, long long npix, int C, egm_stream_t s);
/* out = highpass3(a) + b (+ c) (+ d) in one pass (c, d may be NULL): gradient fan-in when one consumer of the tensor was egm_highpass3
 * (the stencil is self-adjoint); highpass3(a) is rounded to the storage type before the sum, like the separate pass. */
int egm_sum4_hp(int dtype, const void* a, int lda, const void* b, int ldb, const void* c, int ldc, const void* d, int ldd, void* out,
                int ldo, int N, int H, int W, int C, egm_stream_t s);
/* fp32 vector add: y[i] += x[i] (parameter-gradient accumulation) */
int egm_vec_add_f32(float* y, const float* x, long long n, egm_stream_t s);
int egm_fill_f32(float* y, float v, long long n, egm_stream_t s);
/* fp32 matrix column block copy: dst[r][col0_dst + c] = src[r][col0_src + c], c < ncols (weight re-layout for inputs padded per tensor) */
int egm_copy_cols_f32(const float* src, int ld_src, int col0_src, float* dst, int ld_dst, int col0_dst, int rows, int ncols,
                      egm_stream_t s);

/* ---- EdgeAwareFeatureEnhancer pieces (src/EGM-UNet.py:872-886) -------------------------------------------------- */
/* out = x - avgpool3x3(x) (zero pad, divisor 9).  Self-adjoint: the same call is its own backward. */
int egm_highpass3(int dtype, const void* x, int ldx, void* out, int ldo, int N, int H, int W, int C, egm_stream_t s);
/* out = x*(1+w);  backward: dx = g*(1+w), dw = g*x */
int egm_gate_mul_fwd(int dtype, const void* x, int ldx, const void* w, int ldw, void* out, int ldo, long long npix, int C,
                     egm_stream_t s);
int egm_gate_mul_bwd(int dtype, const void* g, int ldg, const void* x, int ldx, const void* w, int ldw, void* dx, int lddx,
                     void* dw, int lddw, long long npix, int C, egm_stream_t s);

/* ---- EdgeEnhancedGRFB tail (src/EGM-UNet.py:1315-1321) ----------------------------------------------------------- */
/* out = relu(alpha*a + b);  backward (mask = out > 0): da = alpha*g*mask, db = g*mask */
int egm_scale_add_relu_fwd(int dtype, const void* a, int lda, float alpha, const void* b, int ldb, void* out, int ldo,
                           long long npix, int C, egm_stream_t s);
int egm_scale_add_relu_bwd(int dtype, const void* g, int ldg, const void* out, int ldo, float alpha, void* da, int ldda,
                           void* db, int lddb, long long npix, int C, egm_stream_t s);
/* out = x*(1 + mean_k sigmoid(t[..,k])), k<3 (t: 8-channel map, 3 real);  backward writes dx and dt (8 channels) */
int egm_gate3_fwd(int dtype, const void* x, int ldx, const void* t, int ldt, void* out, int ldo, long long npix, int C,
                  egm_stream_t s);
int egm_gate3_bwd(int dtype, const void* g, int ldg, const void* x, int ldx, const void* t, int ldt, void* dx, int lddx,
                  void* dt, int lddt, long long npix, int C, egm_stream_t s);

/* ---- RecursiveGatedAttention pieces (src/EGM-UNet.py:531-544) ---------------------------------------------------- */
/* out = a * sigmoid(gl[..,0]) (gl: 8-channel map, 1 real);  backward writes da and dgl (8 channels) */
int egm_bcast_gate_fwd(int dtype, const void* a, int lda, const void* gl, int ldgl, void* out, int ldo, long long npix, int C,
                       egm_stream_t s);
int egm_bcast_gate_bwd(int dtype, const void* g, int ldg, const void* a, int lda, const void* gl, int ldgl, void* da, int ldda,
                       void* dgl, int lddgl, long long npix, int C, egm_stream_t s);
/* nn.GELU (erf form) */
int egm_gelu_fwd(int dtype, const void* x, int ldx, void* out, int ldo, long long npix, int C, egm_stream_t s);
int egm_gelu_bwd(int dtype, const void* g, int ldg, const void* x, int ldx, void* dx, int lddx, long long npix, int C,
                 egm_stream_t s);

/* ---- FusionConv pieces (src/EGM-UNet.py:1171-1236) ---------------------------------------------------------------- */
/* SpatialAttentionModule input: out[..,0] = mean_c x, out[..,1] = max_c x over the C_real real channels (8-channel map) */
int egm_chan_meanmax_fwd(int dtype, const void* x, int ldx, void* out, int ldo, long long npix, int C, int C_real,
                         egm_stream_t s);
int egm_chan_meanmax_bwd(int dtype, const void* g, int ldg, const void* x, int ldx, void* dx, int lddx, long long npix, int C,
                         int C_real, egm_stream_t s);
/* SpatialAttentionModule.conv1 (7x7, 2 -> 1, no bias, w fp32 [1][2][7][7]) as a direct stencil on the 8-channel (mean, max)
 * map; out/dx are 8-channel maps (ch0 / ch0-1 real).  bwd: dx, dw; workspace egm_sa_conv7_bwd_workspace() bytes. */
int egm_sa_conv7_fwd(int dtype, const void* x, int ldx, const float* w, void* y, int ldy, int N, int H, int W, egm_stream_t s);
long long egm_sa_conv7_bwd_workspace(int N, int H, int W);
int egm_sa_conv7_bwd(int dtype, const void* x, int ldx, const void* dy, int lddy, const float* w, void* dx, int lddx, float* dw,
                     void* workspace, int N, int H, int W, egm_stream_t s);
/* ChannelAttentionModule pools: out [2N][C] (rows 0..N-1 = global average, N..2N-1 = global max), argidx int32 [N][C] =
 * first position of the maximum; workspace egm_global_pool_workspace() bytes. */
long long egm_global_pool_workspace(int N, long long HW, int C);
int egm_global_avgmax_fwd(int dtype, const void* x, int ldx, void* out, int* argidx, void* workspace, int N, long long HW,
                          int C, egm_stream_t s);
int egm_global_avgmax_bwd(int dtype, const void* gout, const int* argidx, void* dx, int lddx, int N, long long HW, int C,
                          egm_stream_t s);
/* dst[n][c] = dst[N+n][c] = red[n][0][c] (red fp32 [N][2][C]): the channel-attention gradient of egm_fusion_combine_bwd, reduced by
 * egm_reduce_tiles_batched, as the [2N][C] rows the attention's backward takes. */
int egm_rows_dup(int dtype, const float* red, void* dst, int ldd, int N, int C, egm_stream_t s);
/* ChannelAttentionModule.fc (src/EGM-UNet.py:1171-1190) on the R = 2N pooled rows: logits = W2 . relu(W0 . pooled).  w0 fp32 [Cr][C],
 * w2 fp32 [C][Cr] (the 1x1 conv weights as stored), h fp32 [R][Cr] = the hidden activation kept for backward.  One workgroup each;
 * the rows, the hidden activations and both weight matrices must fit 64 KB of LDS (C = 128, Cr = 32, R = 16: 46 KB backward).  Backward OVERWRITES dw0 / dw2. */
int egm_ca_mlp_fwd(int dtype, const void* pooled, int ldp, const float* w0, const float* w2, float* h, void* logits, int ldo, int R,
                   int C, int Cr, egm_stream_t s);
int egm_ca_mlp_bwd(int dtype, const void* dlogits, int ldd, const void* pooled, int ldp, const float* h, const float* w0,
                   const float* w2, float* dw0, float* dw2, void* dpooled, int lddp, int R, int C, int Cr, egm_stream_t s);
/* out = f + s*sigmoid(sa[..,0])*sigmoid(ca[n][c] + ca[N+n][c])   (x_out = up(res + x_fused_s * x_fused_c), :1232-1235).
 * backward: ds, dsa (8-channel map) and per-image partial tiles [N][nblk][2][C] of dca (row 0; reduce with
 * egm_reduce_tiles per image); nblk = egm_fusion_combine_blocks(HW, C).  df is g itself. */
int egm_fusion_combine_fwd(int dtype, const void* f, int ldf, const void* sv, int lds, const void* sa, int ldsa, const void* ca,
                           void* out, int ldo, int N, long long HW, int C, egm_stream_t s);
int egm_fusion_combine_blocks(long long HW, int C);
int egm_fusion_combine_bwd(int dtype, const void* g, int ldg, const void* sv, int lds, const void* sa, int ldsa, const void* ca,
                           void* ds, int ldds, void* dsa, int lddsa, float* partials, int N, long long HW, int C, egm_stream_t s);
/* Algebraic folds of FusionConv parameters (fp32, tiny): down(cat[x,x]) == conv with W[:, :K] + W[:, K:];
 * conv3(f)+conv5(f)+conv7(f) == one 7x7 conv with the zero-padded kernels (and biases) summed. */
int egm_fold2_fwd(const float* w, float* out, int rows, int K, egm_stream_t s);
int egm_fold2_bwd(const float* g, float* dw, int rows, int K, egm_stream_t s);
int egm_merge357_fwd(const float* w3, const float* w5, const float* w7, const float* b3, const float* b5, const float* b7,
                     float* w, float* b, int Co, int Ci, egm_stream_t s);
int egm_merge357_bwd(const float* gw, float* d3, float* d5, float* d7, const float* gb, float* gb3, int Co, int Ci, egm_stream_t s);
/* egm_fold2_fwd / egm_merge357_fwd that ALSO write the operand packs (egm_conv_pack layouts, groups = 1) of the derived weight:
 * wf [taps][CoutP][CinP], wd [taps flipped][CinP][CoutP] in `dtype`, CoutP/CinP = the counts rounded up to 8. */
int egm_fold2_pack(int dtype, const float* w, float* out, void* wf, void* wd, int rows, int K, egm_stream_t s);
int egm_merge357_pack(int dtype, const float* w3, const float* w5, const float* w7, const float* b3, const float* b5, const float* b7,
                      float* w, float* b, void* wf, void* wd, int Co, int Ci, egm_stream_t s);

/* ---- MCALayer (src/EGM-UNet.py:686-791, MCAGate :836-869, StdPool :827-834) ---------------------------------------- */
/* Three-axis sums in one pass: sums fp32 [N][H+W+C][2] laid out per image as [H rows | W columns | C channels];
 * mode 0: (sum a, sum a^2); mode 1: (sum a*b, -).  workspace egm_mca_reduce_workspace() bytes. C: power of two <= 512. */
long long egm_mca_reduce_workspace(int N, int H, int W, int C);
int egm_mca_reduce(int dtype, int mode, const void* a, int lda, const void* b, int ldb, float* sums, void* workspace, int N,
                   int H, int W, int C, egm_stream_t s);
/* egm_mca_reduce mode 0 over z = act(y*scale + shift), the BatchNorm(+ReLU) output of the conv in front of the MCALayer
 * (src/EGM-UNet.py:893-896), which is WRITTEN to z on the way: egm_bn_act_fwd and the statistics pass over its result as one pass. */
int egm_mca_reduce_bn(int dtype, const void* y, int ldy, const float* scale, const float* shift, int act, void* z, int ldz,
                      float* sums, void* workspace, int N, int H, int W, int C, egm_stream_t s);
/* sums -> stats [N][L][2] (mean, unbiased std), o [N][L] (pre-conv gate input), gates [N][L] (sigmoid outputs); L=H+W+C.
 * w_*: MCAGate.weight (2 floats); k_*: the 1 x ks conv kernel of each gate (h_cw, w_hc, c_hw).
 * ks_c == 0 is MCALayer(no_spatial=True) (src/EGM-UNet.py:700-703,766-771): there is no c_hw gate, w_c / k_c may be NULL, the channel
 * rows of `gates` are written as zeros (so x*(g_h+g_w+g_c) is x*(g_h+g_w)) and, in the backward, their coef / dwts / dks rows are zeros. */
int egm_mca_gates_fwd(const float* sums, const float* w_h, const float* k_h, int ks_h, const float* w_w, const float* k_w,
                      int ks_w, const float* w_c, const float* k_c, int ks_c, float* stats, float* o, float* gates, int N, int H,
                      int W, int C, egm_stream_t s);
/* dG [N][L][2] (slot 0 = sum over each slice of dx_out*x) -> coef [N][L][2] (A, B with dx += A + B*x along each axis),
 * dwts [3][2] (gradients of the three MCAGate.weight), dks [3][8] (gradients of the conv kernels). */
int egm_mca_gates_bwd(const float* dG, const float* stats, const float* o, const float* gates, const float* w_h, const float* k_h,
                      int ks_h, const float* w_w, const float* k_w, int ks_w, const float* w_c, const float* k_c, int ks_c,
                      float* dz_scratch, float* coef, float* dwts, float* dks, int N, int H, int W, int C, egm_stream_t s);
/* x_out = x*(g_h+g_w+g_c)/3; no_spatial != 0: x*(g_h+g_w)/2 (gates from egm_mca_gates_fwd with ks_c == 0) */
int egm_mca_xout(int dtype, const void* x, int ldx, const float* gates, void* xo, int ldo, int N, int H, int W, int C,
                 int no_spatial, egm_stream_t s);
/* r1 = 0.51*xo + 0.2*(max3-min3)(xo) + 0.1*shuffle4(xo); u2 = (xo - avg3 xo)^2; codes (optional, N*H*W*C bytes) = window
 * positions of the first max / first min, for the backward.  Then out = r1 + 0.2*avg3(u2) via egm_add_avg3. */
int egm_mca_stencil1(int dtype, const void* xo, int ld, void* r1, int ldr, void* u2, int ldu, unsigned char* codes, int N, int H,
                     int W, int C, egm_stream_t s);
int egm_add_avg3(int dtype, const void* a, int lda, const void* b, int ldb, float scale, void* out, int ldo, int N, int H, int W,
                 int C, egm_stream_t s);
/* The three calls above in ONE pass (x read once with a 2-pixel halo; out / codes / optionally x_out written once): 16 x 16 pixel tiles
 * of 32-channel chunks staged through LDS.  xo may be NULL (the backward recomputes x_out = x * gate).  Same arithmetic, rounding points
 * and summation order as egm_mca_xout + egm_mca_stencil1 + egm_add_avg3. */
int egm_mca_fused_fwd(int dtype, const void* x, int ldx, const float* gates, void* xo, int ldxo, void* out, int ldo, unsigned char* codes,
                      int N, int H, int W, int C, int no_spatial, egm_stream_t s);
/* backward chain: du = 0.4*(xo - avg3 xo)*avg3(g); dxo = 0.51 g + 0.1 unshuffle(g) + du - avg3(du) + 0.2*range_bwd(codes, g);
 * dx = dxo*(g_h+g_w+g_c)/3 + sum_axes(A + B*x)   (no_spatial != 0: /2, as in the forward) */
int egm_mca_bwd_du(int dtype, const void* xo, int ld, const void* g, int ldg, void* du, int ldd, int N, int H, int W, int C,
                   egm_stream_t s);
int egm_mca_bwd_dxo(int dtype, const unsigned char* codes, const void* g, int ldg, const void* du, int ldd, void* dxo, int ldo,
                    int N, int H, int W, int C, egm_stream_t s);
/* egm_mca_bwd_du + egm_mca_bwd_dxo as ONE tiled pass, bf16 only (x_out, g and the codes staged with their halo in LDS, du kept there):
 * same operand order in every sum as the pair.  Replaces the du / dxo steps of MCALayer's autograd backward (src/EGM-UNet.py:729-772). */
int egm_mca_bwd_dudxo(int dtype, const unsigned char* codes, const void* xo, int ldxo, const void* g, int ldg, void* dxo, int ldo,
                      int N, int H, int W, int C, egm_stream_t s);
int egm_mca_bwd_dx(int dtype, const void* dxo, int ldd, const void* x, int ldx, const float* gates, const float* coef, void* dx,
                   int ldo, int N, int H, int W, int C, int no_spatial, egm_stream_t s);

/* ---- CLIP ViT / CLIPSeg inference path (clip/model.py:159-206,487-501; models/clipseg.py:79-133,188-256,436-496) --------
 * Activations are row-major [rows, D] token matrices (batch-first), bf16 or fp32; scores and statistics are fp32. */
/* Batched GEMM with fused epilogue: C[b] = act(alpha * A[b](M x K) * op(B[b]) + bias) + R[b].
 * transB != 0: B[b] is [N][K] (nn.Linear weight, K of q k^T); else [K][N] (V of P V, `x @ proj`).  Two batch levels
 * (nb1 x nb2, e.g. images x heads) with element strides s?1 / s?2.  act: 0 none, 1 ReLU, 2 QuickGELU.
 * c_is_f32 != 0 stores C as fp32 (attention scores).  bias fp32 [N] or NULL; R (dtype, ldr) or NULL. */
int egm_gemm(int dtype, const void* A, int lda, const void* B, int ldb, int transB, void* C, int ldc, int c_is_f32,
             const float* bias, int act, const void* R, int ldr, float alpha, int M, int N, int K, int nb1, int nb2,
             long long sA1, long long sA2, long long sB1, long long sB2, long long sC1, long long sC2, long long sR1,
             long long sR2, egm_stream_t s);
/* Large bf16 A * B^T products (M >= 512, N >= 256, K a multiple of 64, one batch, >= 128 tiles of 256 x 256) run on the 8-wave LDS-DMA
 * kernel (csrc/gemm_dma.hip), bit-identical to the register-staged one.  mode 1 / 0: on / off, 2: its 4-wave form (128-row wave tiles),
 * 3: on without the 256 x 128 tile rule, 4: 256-wide tiles whatever N (both for A/B runs, profiles/r04_ab_runs.md), -1: query; returns the
 * previous mode (default: env EGM_GEMM_DMA, else 1). */
int egm_gemm_dma_mode(int mode);
/* P[r][:] (dtype) = softmax(S[r][:L]) (S fp32); causal != 0 keeps columns j <= r % L (text encoder mask,
 * clip/model.py:462-468); accumulate != 0 adds to P (CSA: softmax(q q^T) + softmax(k k^T), models/clipseg.py:96-102).
 * Columns L..ldp-1 of P are written as 0. */
/* Fused multi-head self attention on packed qkv [B][L][3*H*64] (bf16, head dimension 64): out [B][L][H*64].
 * mode 0: softmax(q k^T / 8) v; 1: causal (clip/model.py:462-468); 2: CSA (softmax(q q^T / 8) + softmax(k k^T / 8)) v
 * (models/clipseg.py:96-102).  Online softmax, fp32 statistics; scores and probabilities never reach HBM. */
int egm_attention_fused(int dtype, const void* qkv, int ld, int B, int L, int H, int head_dim, int mode, void* out, int ldo,
                        egm_stream_t s);
int egm_softmax_rows(int dtype, const float* S, int lds, void* P, int ldp, long long rows, int L, int causal, int accumulate,
                     egm_stream_t s);
/* LayerNorm over the last dimension with fp32 statistics (clip/model.py:159-165). */
int egm_layernorm(int dtype, const void* x, int ldx, const float* gamma, const float* beta, float eps, void* y, int ldy,
                  long long rows, int D, egm_stream_t s);
/* VisionTransformer.conv1 front half: fp32 NCHW image -> patch rows [B*(H/P)*(W/P)][C*P*P] (then egm_gemm with conv1.weight). */
int egm_patchify(int dtype, const float* img_nchw, void* out, int B, int C, int H, int W, int P, egm_stream_t s);
/* x[b][0] = class_embedding + pos[0]; x[b][1+t] = tok[b][t] + pos[1+t]  (models/clipseg.py:205-213) */
int egm_vit_assemble(int dtype, const void* tok, const float* cls, const float* pos, void* x, int B, int Ltok, int D, egm_stream_t s);
/* x[n][t] = token_embedding[tokens[n][t]] + (t < split ? pos[t] : pos_res[t])  (clip/model.py:428-431,488-490; split = 20) */
int egm_text_embed(int dtype, const int* tokens, const float* emb, const float* pos, const float* pos_res, int split, void* x,
                   int n, int L, int D, egm_stream_t s);
/* FiLM: a[b][t][:] = a[b][t][:] * mul[b][:] + add[b][:]  (models/clipseg.py:467-471) */
int egm_film(int dtype, void* a, const void* mul, const void* add, int B, int L, int D, egm_stream_t s);
/* out[n][:] = x[n][idx[n]][:]  (EOT token / class token selection) */
int egm_gather_rows(int dtype, const void* x, const int* idx, void* out, int n, int L, int D, egm_stream_t s);
/* ConvTranspose2d(D -> 1, kernel P, stride P) back half: y[b*Ltot + tok_off + ty*g + tx][i*P + j] (+ bias) ->
 * out fp32 [B][1][g*P][g*P]  (models/clipseg.py:478-484) */
int egm_pixel_shuffle(int dtype, const void* y, int ldy, int tok_off, int Ltot, const float* bias, float* out, int B, int g,
                      int P, egm_stream_t s);
/* fp32 -> dtype copy of a weight matrix */
int egm_cast_f32(int dtype, const float* src, void* dst, long long n, egm_stream_t s);

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
/* CLIPSeg (+) UNet ensemble tail (predict_CLIPseg.py:501-525): fused = bilinear(clip_logits -> HxW, align_corners=False) +
 * alpha*unet_logits; pred (int64 [N,H,W], optional) = argmax_c fused; fused (fp32 [N,C,H,W], optional). */
int egm_ensemble_fuse(const float* clip_logits, const float* unet_logits, float alpha, int N, int C, int hc, int wc, int H, int W,
                      long long* pred, float* fused, egm_stream_t s);
/* Alpha grid search (eval_CLIPseg.py:656-723): one pass accumulates a confusion matrix for EVERY alpha of the grid
 * (hist: zeroed uint64 [na][C][C], accumulates across calls); egm_ensemble_miou gives mean IoU per alpha. na<=128, C<=4. */
int egm_ensemble_alpha_hist(const float* clip_logits, const float* unet_logits, const long long* target, const float* alphas, int na,
                            int N, int C, int hc, int wc, int H, int W, unsigned long long* hist, egm_stream_t s);
int egm_ensemble_miou(const unsigned long long* hist, int na, int C, float* miou, egm_stream_t s);
/* torch.optim.SGD(momentum, weight_decay) (train.py:115-118) over a device table of 40-byte {float* p; const float* g; float* buf;
 * long long n; long long chunk0;} entries (chunk0 as for egm_conv_pack_multi): g' = g*grad_scale + wd*p;
 * v = first_step ? g' : mu*v + g'; p -= lr*v.
 * lr_dev (device scalar) overrides lr when non-NULL. */
int egm_sgd_chunk(void);   /* elements per workgroup; total_chunks = sum over tensors of ceil(n / egm_sgd_chunk()) */
int egm_sgd_multi(const void* table_dev, int ntensors, long long total_chunks, const float* lr_dev, float lr, float momentum,
                  float weight_decay, float grad_scale, int first_step, egm_stream_t s);
/* table of {float* dst; const float* src; long long n;}: gradient bucket gather/scatter for the RCCL all-reduce. */
int egm_copy_multi(const void* table_dev, int ntensors, egm_stream_t s);

/* ---- Up(bilinear=False): nn.ConvTranspose2d(Cin, Cout, 2, stride 2) (src/unet.py:36, src/EGM-UNet.py:941) ------------------
 * Run as a 1x1 convolution with 4*CoutP output channels ordered (i, j, co) -- egm_convT2x2_pack rearranges the [Cin][Cout][2][2]
 * parameter into that OIHW matrix (to_packed = 1) or scatters a gradient back (to_packed = 0) -- followed by the 2x2 pixel shuffle,
 * which adds the bias and writes at offset (oy, ox) inside a zero-filled [Ho][Wo] frame (F.pad of Up.forward). */
int egm_convT2x2_pack(float* w_iohw, float* w4, int Cin, int Cout, int CoutP, int to_packed, egm_stream_t s);
int egm_shuffle2x2_fwd(int dtype, const void* y4, int ld4, const float* bias, int bias_n, void* out, int ldo, int N, int H, int W, int C,
                       int Ho, int Wo, int oy, int ox, egm_stream_t s);
int egm_shuffle2x2_bwd(int dtype, const void* g, int ldg, void* d4, int ld4, int N, int H, int W, int C, int Ho, int Wo, int oy, int ox,
                       egm_stream_t s);

/* ---- ELA, Efficient Local Attention (src/EGM-UNet.py:56-79; an unused ablation block of the reference) -----------
 * strip means mh [N][H][C] (over W) and mw [N][W][C] (over H), fp32 -> shared depthwise Conv1d(ks, no bias, conv_w [C][ks]) ->
 * GroupNorm(groups, C) -> sigmoid gates gh, gw -> out = x * gh[n,h,c] * gw[n,w,c].  yh/yw (conv outputs) and stats
 * [N][2][groups][2] (mean, rstd) are kept for egm_ela_bwd, which returns dx and the parameter gradients. */
int egm_ela_strip_means(int dtype, const void* x, int ldx, float* mh, float* mw, int N, int H, int W, int C, egm_stream_t s);
int egm_ela_gates_fwd(const float* mh, const float* mw, const float* conv_w, int ks, const float* gamma, const float* beta, float eps,
                      float* yh, float* yw, float* gh, float* gw, float* stats, int N, int H, int W, int C, int groups, egm_stream_t s);
int egm_ela_apply(int dtype, const void* x, int ldx, const float* gh, const float* gw, void* out, int ldo, int N, int H, int W, int C,
                  egm_stream_t s);
long long egm_ela_bwd_workspace(int N, int H, int W, int C, int ks);
int egm_ela_bwd(int dtype, const void* g, int ldg, const void* x, int ldx, const float* mh, const float* mw, const float* yh, const float* yw,
                const float* gh, const float* gw, const float* stats, const float* conv_w, int ks, const float* gamma, void* dx, int lddx,
                float* dconv_w, float* dgamma, float* dbeta, float* workspace, int N, int H, int W, int C, int groups, egm_stream_t s);

/* ---- HEGDC (src/EGM-UNet.py:210-340; an unused ablation block of the reference) ---------------------------------------
 * egm_hegdc_edge_features: the block's no_grad branch: channel mean -> fixed Scharr/16 + Sobel/4 stencils -> magnitudes ->
 * min-max over the WHOLE batch -> sqrt (gamma 0.5) -> blend a*scharr + (1-a)*sobel, a = sigmoid(mean difference); feats
 * [N][H][W][8] in the activation dtype holds (scharr_x, scharr_y, sobel_x, sobel_y, blend, 0, 0, 0).
 * egm_scale_sigmoid_*: W * sigmoid(den) ("density" scaling of conv1's weight) and its gradients.
 * egm_mul2_scalar_*: a * b * alpha with a learnable device scalar alpha and its gradients (db may be NULL). */
long long egm_hegdc_edge_workspace(int N, int H, int W);
int egm_hegdc_edge_features(int dtype, const void* x, int ldx, int C_real, void* feats, float* workspace, int N, int H, int W,
                            egm_stream_t s);
int egm_scale_sigmoid_fwd(const float* w, const float* den, float* out, long long n, egm_stream_t s);
int egm_scale_sigmoid_bwd(const float* g, const float* w, const float* den, float* dw, float* dden, float* partials, long long n,
                          egm_stream_t s);
int egm_mul2_scalar_fwd(int dtype, const void* a, int lda, const void* b, int ldb, const float* alpha, void* out, int ldo, long long npix,
                        int C, egm_stream_t s);
int egm_mul2_scalar_bwd(int dtype, const void* g, int ldg, const void* a, int lda, const void* b, int ldb, const float* alpha, void* da,
                        int ldda, void* db, int lddb, float* dalpha, float* partials, long long npix, int C, egm_stream_t s);

/* ---- device-side data path (transforms.py, my_dataset.py:118-132) ----------------------------------------------
 * Decoded uint8 images [H][W][C] already in device memory.
 * egm_resample_u8: one separable pass of Pillow's antialiased resize (F.resize -> Image.resize(BILINEAR), transforms.py:39):
 *   axis 1: dst [H][out_size][C], axis 0: dst [out_size][W][C];  out = clip8((2^21 + sum_j src[b0+j] * coefs[o][j]) >> 22)
 *   with bounds [out_size][2] = (b0, n) and coefs [out_size][ksize] int32 (22-bit fixed point) computed by the caller.
 * egm_gather_u8: dst[y][x][c] = src[yidx[y]][xidx[x]][c]  (NEAREST resize of the mask, transforms.py:40).
 * egm_augment_u8: hflip/vflip -> pad_if_smaller with zeros -> crop(top, left, crop_h, crop_w) -> to_tensor (/255) ->
 *   normalize(mean, std) (transforms.py:46-107), written into an [out_h][out_w] collate slot: outside the crop the image is
 *   0.0 and the target 255 (collate_fn).  out_img_chw fp32 [3][out_h][out_w], out_target int64 [out_h][out_w] (may be NULL with
 *   mask_hw NULL).  mean3/std3 are HOST pointers (3 floats each). */
int egm_resample_u8(const void* src, int H, int W, int C, void* dst, int axis, int out_size, const int* bounds, const int* coefs,
                    int ksize, egm_stream_t s);
int egm_gather_u8(const void* src, int H, int W, int C, void* dst, int Ho, int Wo, const int* yidx, const int* xidx, egm_stream_t s);
int egm_augment_u8(const void* img_hwc3, const void* mask_hw, int H, int W, int hflip, int vflip, int top, int left, int crop_h,
                   int crop_w, const float* mean3_host, const float* std3_host, float* out_img_chw, long long* out_target,
                   int out_h, int out_w, egm_stream_t s);

/* ---- CLIPSeg decoder training (models/clipseg.py:380-420,452-496; experiments/phrasecut.yaml:1-47) ------------------
 * The backward matrix products run on egm_gemm over transposed copies (egm_transpose: dst[b][c][r] = src[b][r][c]).
 * egm_relu_bwd: dst = g where out > 0.  egm_softmax_bwd_rows: dS = P*(dP - sum_j dP_j P_j)*alpha per row (P storage type,
 * dP fp32).  egm_layernorm_bwd: dx and per-block partials [egm_layernorm_bwd_blocks(rows)][2][D] of (dbeta, dgamma) for
 * egm_reduce_tiles.  egm_film_fwd/bwd: out = mul[b,:]*a + add[b,:] and its gradients.  egm_pixel_unshuffle: gradient of
 * egm_pixel_shuffle (rows of the tok_off leading tokens are zero).  egm_bce_logits_*: nn.BCEWithLogitsLoss(mean).
 * egm_adamw_multi: torch.optim.AdamW step over a device table of 48-byte {p, g, m, v, n, chunk0} entries. */
int egm_transpose(int dtype, const void* src, int rows, int cols, int ld_src, long long batch_stride_src, void* dst, int ld_dst,
                  long long batch_stride_dst, int batch, egm_stream_t s);
int egm_relu_bwd(int dtype, const void* g, const void* out, void* dst, long long n, egm_stream_t s);
int egm_softmax_bwd_rows(int dtype, const void* P, int ldp, const float* dP, int lddp, void* dS, int ldds, long long rows, int L,
                         float alpha, egm_stream_t s);
int egm_layernorm_bwd_blocks(long long rows);
int egm_layernorm_bwd(int dtype, const void* x, int ldx, const void* g, int ldg, const float* gamma, float eps, void* dx, int lddx,
                      float* partials, long long rows, int D, egm_stream_t s);
int egm_film_fwd(int dtype, const void* a, const void* mul, const void* add, void* out, int B, int L, int D, egm_stream_t s);
int egm_film_bwd(int dtype, const void* g, const void* a, const void* mul, void* da, void* dmul, void* dadd, int B, int L, int D,
                 egm_stream_t s);
int egm_pixel_unshuffle(int dtype, const float* dout, void* dy, int B, int g, int P, int tok_off, int Ltot, egm_stream_t s);
int egm_sum_f32(const float* x, long long n, float scale, float* partials, float* out, egm_stream_t s);
int egm_bce_logits_fwd(const float* x, const float* t, long long n, float* partials, float* loss, egm_stream_t s);
int egm_bce_logits_bwd(const float* x, const float* t, const float* grad_out, long long n, float* dx, egm_stream_t s);
int egm_adamw_chunk(void);
int egm_adamw_multi(const void* table_dev, int ntensors, long long total_chunks, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int step, egm_stream_t s);

/* ---- CLIPSeg decoder training: dropout, visual-prompt mask ---------------------------------------------------------------------
 * nn.TransformerEncoderLayer(dropout=0.1) in train mode (models/clipseg.py:421-422): out = residual + keep*x/(1-p), keep regenerated
 * from (seed, element index) by a counter-based hash, so the backward is the same call on the gradient (residual = NULL).
 * x/out may be fp32 (dtype EGM_F32) or bf16. */
int egm_dropout(int dtype, const void* x, const void* residual, void* out, long long n, float p, unsigned long long seed,
                egm_stream_t s);
/* CLIPDensePredTMasked (models/clipseg.py:500-525; forward_multihead_attention :111-117): the class token's attention row of every
 * (batch, head) is multiplied by a [nmask][ntok] mask, head bh taking mask row bh % nmask (the reference's repeat() pairing). */
int egm_attn_mask_cls(int dtype, void* probs, int ldp, long long head_stride, const float* mask, int nmask, int nbh, int ntok,
                      egm_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* EGM_HIP_H */
