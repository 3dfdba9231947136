/*
 * michigan_hip.h -- C ABI of libmichigan_hip.so (gfx950 / MI355X only).
 *
 * This is the drop-in boundary of the MichiGAN generator / discriminator / VGG
 * hot path.  The reference has NO native code and NO FFI: its operator layer is
 * Python calling ATen.  Every entry point below therefore
 * cites the reference *call site* whose ATen work it replaces; the Python
 * classes in michigan_amd/networks mirror the reference classes and reach the
 * GPU only through these symbols (ctypes).
 *
 * Conventions
 *   - every tensor is a raw device pointer; activations are NHWC
 *     ([N][H][W][C], C innermost); dtype is MG_F32 or MG_BF16 (accumulation is
 *     always fp32; statistics are always fp32)
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream)
 *   - return value: 0 on success, non-zero error code otherwise;
 *     mg_last_error() returns a thread-local message for the last failure
 *   - no entry point allocates, frees or synchronises
 */
#ifndef MICHIGAN_HIP_H
#define MICHIGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_ABI_VERSION 6

enum { MG_F32 = 0, MG_BF16 = 1 };
enum { MG_ACT_NONE = 0, MG_ACT_RELU = 1, MG_ACT_LRELU = 2, MG_ACT_TANH = 3 };
enum { MG_EPI_PLAIN = 0, MG_EPI_SPADE = 1 };
enum { MG_OK = 0, MG_ERR_ARG = 1, MG_ERR_LAUNCH = 2, MG_ERR_UNSUPPORTED = 3 };

#define MG_MAX_TAPS 64

/* ---------------------------------------------------------------------------
 * mg_conv_taps -- "tap list" implicit-GEMM convolution on the matrix cores.
 *
 *   out[n, jy*osy+ooy, jx*osx+oox, co] = epilogue( sum_t sum_ci
 *        W[t][co][ci] * in[n, jy*isy+tap_dy[t], jx*isx+tap_dx[t], ci] )
 *   for (jy,jx) in [0,Hj)x[0,Wj); input taps outside [0,Hin)x[0,Win) read 0.
 *
 * One kernel covers
 *   forward  nn.Conv2d (3x3 / 4x4 / 7x7 / 1x1, stride 1|2, zero pad):
 *            normalization.py:94-99,111-113 (SPADE mlp convs),
 *            architecture.py:31-35,70-71,79 (conv_0/conv_1/conv_s),
 *            generator.py:72,227 (conv_img), discriminator.py:84-96,
 *            architecture.py:163-178 (VGG19 slices), encoder.py:172-181,
 *            MaskGAN_networks.py:162-168 (ConvBlock)
 *   dgrad    of the same convs (autograd convolution_backward, data part):
 *            stride 1 = same kernel with flipped/transposed weights,
 *            stride 2 = one launch per output-parity class with its tap subset
 *
 * Weights are pre-packed [ntaps][CoutP][Cin] in `dtype`, CoutP = Cout_gemm
 * rounded up to a multiple of 128 with zero rows (packing is host-side glue).
 * Cin must be a multiple of 8 (host pads small channel counts with zeros).
 *
 * Epilogues
 *   MG_EPI_PLAIN: v = acc + bias[co] (+ resid[n,oy,ox,co]) ; out = act(v)   (bias has Cout_gemm entries);
 *                 if x != NULL: out = (x[n,oy,ox,co] > 0) ? out : mask_slope * out  -- a data gradient masked by the ReLU
 *                 (mask_slope 0) / LeakyReLU (its slope) whose output x the forward conv consumed (saves the separate
 *                 activation-backward pass)
 *   MG_EPI_SPADE: GEMM rows come in blocks of 64 = [32 gamma rows | 32 beta rows]
 *                 of the same 32 output channels (mlp_gamma/mlp_beta fused,
 *                 normalization.py:112-116).  For output channel c:
 *                   xhat = (x - mean[c]) * rstd[c];  g1 = 1 + gamma;
 *                   out  = act(xhat * g1 + beta);  gamma_out (optional) = g1
 *                 gamma/beta never reach HBM.  Cout = C (stored channels),
 *                 Cout_gemm = 2 * roundup(C, 32).
 * ------------------------------------------------------------------------- */
typedef struct mg_conv_desc {
    const void* in;        /* [N][Hin][Win][Cin]                              */
    const void* wt;        /* [ntaps][CoutP][Cin] packed, dtype               */
    void*       out;       /* [N][Hout][Wout][Cout]                           */
    const float* bias;     /* [Cout_gemm] (GEMM row order) or NULL            */
    const void* resid;     /* PLAIN: same shape/dtype as out, or NULL         */
    const void* x;         /* SPADE: un-normalised activations; PLAIN: optional ReLU-output mask; shape of out */
    const float* mean;     /* SPADE: [Cout]                                   */
    const float* rstd;     /* SPADE: [Cout]                                   */
    void*       gamma_out; /* SPADE: optional (1+gamma), shape/dtype of out   */
    int32_t dtype;
    int32_t N, Hin, Win, Cin;
    int32_t Hout, Wout, Cout;
    int32_t Cout_gemm, CoutP;
    int32_t Hj, Wj;
    int32_t isy, isx;
    int32_t osy, osx, ooy, oox;
    int32_t ntaps;
    int32_t epilogue;
    int32_t act;
    float   slope;         /* LeakyReLU negative slope                        */
    int32_t x_up;          /* SPADE: 1 = `x` is the [N][Hout/2][Wout/2][Cout] SOURCE of a nearest 2x upsample (generator.py:74,166-207):
                              pixel (y, x) reads source pixel (y >> 1, x >> 1); the upsampled tensor is never materialised */
    int8_t  tap_dy[MG_MAX_TAPS];
    int8_t  tap_dx[MG_MAX_TAPS];
    float   mask_slope;    /* PLAIN with x != NULL: out = (x > 0) ? out : mask_slope * out.  0 = the ReLU mask; s = the backward of a
                              LeakyReLU(s) whose OUTPUT x is (only legal when this data gradient is the ONLY gradient of x's producer:
                              unlike the ReLU mask the leaky one is not idempotent) */
} mg_conv_desc;

int mg_conv_taps(const mg_conv_desc* d, void* stream);

/* ---------------------------------------------------------------------------
 * mg_conv_wgrad -- weight gradient of a forward conv (autograd
 * convolution_backward, weight part), split-K over output pixels:
 *   dw[t][co][ci] += sum_{n,jy,jx} dy[n,jy,jx,co] * x[n, jy*isy+tap_dy[t], jx*isx+tap_dx[t], ci]
 * dw is fp32 and is ACCUMULATED INTO with hardware float atomics (caller
 * zeroes it).  dy has Cg channels per pixel in GEMM row order (for the fused
 * SPADE conv that is the [gamma|beta] block order).  Cg and Cin multiples of 8.
 * If dbias != NULL the bias gradient (column sums of dy) is accumulated into it by the same launch.
 * ------------------------------------------------------------------------- */
typedef struct mg_wgrad_desc {
    const void* x;         /* [N][Hin][Win][Cin]                              */
    const void* dy;        /* [N][Hj][Wj][Cg]                                 */
    float*      dw;        /* [ntaps][Cg][Cin] fp32                           */
    float*      dbias;     /* optional [Cg] fp32: += sum over pixels of dy    */
    int32_t dtype;
    int32_t N, Hin, Win, Cin;
    int32_t Hj, Wj, Cg;
    int32_t isy, isx;
    int32_t ntaps;
    int32_t splitk;        /* 0 = choose automatically                        */
    int32_t flags;         /* bit0: bf16 operands via ds_read_b64_tr_b16; bit1: this launch runs beside another stream's kernels -- the
                              kernel-row 3x3 kernel then keeps to ONE workgroup per CU (results unchanged) */
    int8_t  tap_dy[MG_MAX_TAPS];
    int8_t  tap_dx[MG_MAX_TAPS];
    void*   det_ws;        /* optional: deterministic split-K workspace (see below), NULL = fp32 atomics */
    int64_t det_ws_bytes;
} mg_wgrad_desc;

/* Split-K partial sums reach dw / dbias through fp32 atomics by default (their order, hence the last bits, vary from run to run).
 * With det_ws != NULL every split stores its partial tile into its own slab of the workspace instead and a finishing launch adds the
 * slabs to dw / dbias in a fixed order: bit-reproducible, one extra pass over splits x |dw|.  mg_wgrad_det_workspace returns the bytes
 * the launch mg_conv_wgrad(d) would need (it depends on the split count the launcher picks). */
int     mg_conv_wgrad(const mg_wgrad_desc* d, void* stream);
int64_t mg_wgrad_det_workspace(const mg_wgrad_desc* d);

/* ---------------------------------------------------------------------------
 * Per-channel statistics (sync-BN / instance-norm reduce).
 *   x is [G][P][C]; sums[g][0][c] = sum_p x, sums[g][1][c] = sum_p x*x  (fp64).  With `shift` the per-chunk fp32 partial
 *   sums are taken of x - x[g][0][c] and un-shifted in fp64 (sum x = s + n k, sum x^2 = ss + 2 k s + n k^2), so that
 *   var = E[x^2] - E[x]^2 in mg_norm_finalize carries no fp32 cancellation; the cross-rank all-reduce adds fp64 sums.
 *   shift = 0 is for column sums of zero-mean data (bias gradients).  mg_channel_stats_finalize always shifts.
 *   G = 1, P = N*H*W for batch norm (sync_batchnorm/batchnorm.py:63-68,128-145:
 *   F.batch_norm on one device, sum/ssum reduce on several); G = N, P = H*W for
 *   nn.InstanceNorm2d (normalization.py:47-48, encoder.py:173-181).
 *   `partial` is caller workspace of mg_stats_workspace(G,P,C) bytes; the
 *   reduction is two-stage and deterministic (no atomics).
 * ------------------------------------------------------------------------- */
int64_t mg_stats_workspace(int32_t G, int64_t P, int32_t C);
int mg_channel_stats(const void* x, int32_t dtype, int32_t G, int64_t P, int32_t C, int32_t shift,
                     double* sums /* [G][2][C] */, void* partial, void* stream);
/* mg_channel_stats followed by mg_norm_finalize in TWO launches instead of three (stage 2 finalizes): for statistics that need no
 * cross-rank reduction in between (instance norm; batch norm on one GPU).  sums are multiplied by sum_scale before use (4 for the
 * statistics of a nearest 2x upsample taken from its source, `count` then counts the upsampled elements).  Bit-identical to
 * mg_channel_stats + (sums *= sum_scale) + mg_norm_finalize. */
int mg_channel_stats_finalize(const void* x, int32_t dtype, int32_t G, int64_t P, int32_t C, float sum_scale, double count,
                              float eps, float momentum, float* running_mean, float* running_var,
                              double* sums, float* mean, float* rstd, void* partial, void* stream);

/* sums[G][2][C] + element count -> mean[G][C], rstd[G][C] = 1/sqrt(biased_var + eps) (fp64 inside); when
 * running_mean/var are given (G == 1) they are updated with momentum and the UNBIASED variance, as
 * F.batch_norm does (sync_batchnorm/batchnorm.py:65-68,136-143).  `count` is the global element count
 * (after the cross-rank all-reduce of `sums`). */
int mg_norm_finalize(const double* sums, int32_t G, int32_t C, double count, float eps, float momentum,
                     float* running_mean, float* running_var, float* mean, float* rstd, void* stream);

/* y = act((x - mean[g][c]) * rstd[g][c]) [+ resid]   (InstanceNorm2d + LeakyReLU, discriminator.py:88-93, encoder.py:188-197;
 * resid != NULL: the skip connection of the in-painting net's residual blocks, generator.py:463 `x + self.conv_block(x)`,
 * same shape and dtype as y, added after the activation) */
int mg_norm_act_fwd(const void* x, void* y, int32_t dtype, int32_t G, int64_t P, int32_t C,
                    const float* mean, const float* rstd, int32_t act, float slope, const void* resid, void* stream);

/* ---------------------------------------------------------------------------
 * Backward of  h = act(xhat * g1 + beta),  xhat = (x - mean) * rstd
 * (SPADE: g1 = 1 + gamma; plain norm+act: g1 == NULL means 1, no beta).
 *
 * mg_norm_bwd_reduce: dpre = dh * act'(h);  dxhat = dpre * g1
 *     sums[g][0][c] = sum_p dxhat ; sums[g][1][c] = sum_p dxhat * xhat
 *     if dgb != NULL (SPADE): writes d[gamma|beta] in GEMM row order
 *     ([P][2*roundup(C,32)]): dgamma = dpre * xhat, dbeta = dpre.
 * mg_norm_bwd_apply:  dx = rstd * (dxhat - S1[g][c] - xhat * S2[g][c]),  S1 = s1[g * sum_gstride + c] * sum_scale (S2 alike):
 *     the raw sums of mg_norm_bwd_reduce (s1 = sums, s2 = sums + C, sum_gstride = 2C, sum_scale = 1 / n) after the
 *     cross-rank all-reduce for sync-BN, or pre-divided per-group vectors (sum_gstride = C, sum_scale = 1).
 * Replaces autograd of normalization.py:105-116 + F.batch_norm / InstanceNorm.  h (the activation's output) is
 * read only for its sign and may be NULL when act == MG_ACT_NONE.
 * ------------------------------------------------------------------------- */
int mg_norm_bwd_reduce(const void* dh, const void* h, const void* x, const void* g1,
                       int32_t dtype, int32_t G, int64_t P, int32_t C,
                       const float* mean, const float* rstd, int32_t act, float slope,
                       void* dgb, float* sums /* [G][2][C] */, void* partial, void* stream);
int mg_norm_bwd_apply(const void* dh, const void* h, const void* x, const void* g1,
                      int32_t dtype, int32_t G, int64_t P, int32_t C,
                      const float* mean, const float* rstd, const float* s1, const float* s2,
                      int32_t sum_gstride, float sum_scale, int32_t act, float slope, void* dx, void* stream);

/* dpre = dy * act'(y) for the activations fused in conv epilogues
 * (ReLU: architecture.py:163-178, MaskGAN_networks.py:145; LeakyReLU:
 * discriminator.py:85; tanh: generator.py:228). */
int mg_act_bwd(const void* dy, const void* y, void* dpre, int32_t dtype, int64_t numel,
               int32_t act, float slope, void* stream);

/* nearest 2x upsample (generator.py:74 nn.Upsample(scale_factor=2)) and its
 * adjoint (sum of the 2x2 children). */
int mg_upsample2x_fwd(const void* x, void* y, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int mg_upsample2x_bwd(const void* dy, void* dx, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);

/* nn.ReflectionPad2d(P) on NHWC (MaskGAN_networks.py:114-121 ConvBlock(pad_type='reflect') as used by the
 * background encoder, :96-103: pad 3 before the 7x7 conv, pad 1 before each 4x4 stride-2 conv) and its adjoint.  y is [N, H+2P, W+2P, C]. */
int mg_reflect_pad_fwd(const void* x, void* y, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t P, void* stream);
int mg_reflect_pad_bwd(const void* dy, void* dx, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, int32_t P, void* stream);

/* F.avg_pool2d(k=3, s=2, p=1, count_include_pad=False) (discriminator.py:46-49) */
int mg_avgpool3s2_fwd(const void* x, void* y, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int mg_avgpool3s2_bwd(const void* dy, void* dx, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);

/* nn.MaxPool2d(2, 2) of the VGG tower (architecture.py:163-178); backward
 * routes dy to the first maximal element of each 2x2 window (ATen order). */
int mg_maxpool2_fwd(const void* x, void* y, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int mg_maxpool2_bwd(const void* dy, const void* x, void* dx, int32_t dtype, int32_t N, int32_t H, int32_t W, int32_t C,
                    int32_t relu_input, void* stream);   /* relu_input = 1: x is a ReLU's output; the routed gradient is also multiplied by (x > 0) */
/* out [N][HW][8] (dtype) = per pixel [ planar[n][0..cp)[p] (fp32 NCHW) | nhwc[p][0..cf) (dtype, pixel pitch cs) | zeros ]:
 * an 8-channel NHWC network input assembled from the reference's planar maps and an NHWC image in one pass */
int mg_assemble_nhwc8(const float* planar, int32_t cp, const void* nhwc, int32_t cs, int32_t cf, void* out, int32_t dtype,
                      int32_t N, int64_t HW, void* stream);
/* out = (g1 + g2) * act'(y) (g2 may be NULL): gradient of an activation output with two consumers, in one pass */
int mg_grad_sum_act(const void* g1, const void* g2, const void* y, void* out, int32_t dtype, int64_t numel, int32_t act, float slope, void* stream);

/* background blend  y = act(bg * (1 - hair[p]) + x * (1 - back[p]))
 * (generator.py:186,197,208,219; act = LeakyReLU after the last block, generator.py:227); hair/back are
 * fp32 [P] single-channel masks.  bwd: dpre = dy * act'(y); dbg = dpre * (1 - hair), dx = dpre * (1 - back). */
int mg_blend_fwd(const void* bg, const void* x, const float* hair, const float* back, void* y,
                 int32_t dtype, int64_t P, int32_t C, int32_t act, float slope, void* stream);
int mg_blend_bwd(const void* dy, const void* y, const float* hair, const float* back, void* dbg, void* dx,
                 int32_t dtype, int64_t P, int32_t C, int32_t act, float slope, void* stream);

/* Parameter re-layout (host-side glue of every conv call, as ONE launch each):
 * mg_pack_weight : reference fp32 [cout][cin][taps] (one tensor, or gamma+beta for the fused SPADE
 *     conv: GEMM row of channel co = 64*(co/32) + co%32, +32 for beta) -> dst[taps][rows_p][cols_p] in
 *     `dtype`, zero padded.  mode 0: rows = GEMM output channels, cols = cin (forward / wgrad image);
 *     mode 1: rows = cin, cols = GEMM output channels (dgrad image, transposed per tap).
 * mg_unpack_wgrad: fp32 dW in GEMM order [taps][rows][cols] -> reference layout (d1 != NULL: beta). */
int mg_pack_weight(const float* w0, const float* w1, void* dst, int32_t dtype, int32_t cout, int32_t cin,
                   int32_t taps, int32_t rows_p, int32_t cols_p, int32_t mode, void* stream);
int mg_unpack_wgrad(const float* dw, float* d0, float* d1, int32_t cout, int32_t cin, int32_t taps,
                    int32_t rows, int32_t cols, void* stream);

/* Fused L1 loss  mean|a - b|  over two same-shaped activation tensors (feature matching
 * loss.py:163-175 and the VGG taps loss.py:199-207): forward in one pass (partial = >= 1024 floats of
 * workspace, out = 1 float); backward da = sign(a - b) * gscale[0] / numel (b is a constant). */
int mg_l1_mean_fwd(const void* a, const void* b, int32_t dtype, int64_t numel, float* out, float* partial, void* stream);
int mg_l1_mean_bwd(const void* a, const void* b, const float* gscale, int32_t dtype, int64_t numel, void* da, void* stream);

/* Batch-norm input gradient for one or two consumers of the same x, optionally through a nearest 2x upsample of x
 * (SPADE norm_0 + norm_s of a residual block with a learned shortcut, architecture.py:68-79; generator.py:166-207):
 *   dx[p'] = sum_b sum_{q in quad(p')} rstd * (dh_b[q] * act_b'(h_b[q]) * g1_b[q] - s1_b - xhat[p'] * s2_b),   s1_b, s2_b = sums_b[0 / 1] * inv_count
 * dh / h / g1: [P][C] at full resolution (h NULL when act is NONE, g1 NULL = 1); sums_b: the [2][C] raw (all-reduced) sums of
 * mg_norm_bwd_reduce[_up].  up = 0: x, dx are [P][C] and quad(p') = {p'}; up = 1: x, dx are the [N][H/2][W/2][C] source /
 * its gradient, P = N*H*W.  Geometry must satisfy mg_norm_apply2_supported(dtype, C).
 * mg_norm_bwd_reduce_up = mg_norm_bwd_reduce (G = 1, P = N*H*W) with x given at half resolution. */
typedef struct mg_norm_apply2_desc {
    const void* dh[2]; const void* h[2]; const void* g1[2]; const float* sums[2];
    const void* x; const float* mean; const float* rstd; void* dx;
    int64_t P;
    int32_t dtype, C, up, H, W;
    int32_t act[2]; float slope[2]; float inv_count;
} mg_norm_apply2_desc;
int mg_norm_bwd_apply2(const mg_norm_apply2_desc* d, void* stream);
int mg_norm_apply2_supported(int32_t dtype, int32_t C);
int mg_norm_bwd_reduce_up(const void* dh, const void* h, const void* x, const void* g1, int32_t dtype, int32_t N, int32_t H, int32_t W,
                          int32_t C, const float* mean, const float* rstd, int32_t act, float slope, void* dgb, float* sums,
                          void* partial, void* stream);

/* Batched weight preparation (one launch for any number of images; tables live in device memory).
 * mg_pack_job: mode 0 / 1 = mg_pack_weight's forward / data-gradient GEMM image of (w0[, w1]) in `dtype`, every element divided
 *   by sigma[0] first when sigma != NULL (spectral norm); mode 2 = plain fp32 copy dst[i] = w0[i] / sigma[0] in the reference
 *   layout (the W_sn tensor itself).  first_block = running sum of mg_pack_job_blocks(...) over the preceding
 *   jobs; block_job[b] = job that owns workgroup b. */
typedef struct mg_pack_job {
    const float* w0; const float* w1; void* dst; const float* sigma;
    int32_t dtype, cout, cin, taps, rows_p, cols_p, mode, pad_;
    int64_t first_block;
} mg_pack_job;
int     mg_pack_weights(const mg_pack_job* jobs_dev, int32_t njobs, const int32_t* block_job_dev, int32_t nblocks, void* stream);
int64_t mg_pack_job_blocks(int32_t mode, int32_t cout, int32_t cin, int32_t taps, int32_t rows_p, int32_t cols_p);   /* workgroups of one job; -1 = not batchable (taps > 49) */

/* torch.nn.utils.spectral_norm's power iteration (dim 0, one iteration, spectral_norm.py: v <- normalize(W^T u, eps),
 * u <- normalize(W v, eps), sigma = u . (W v)) for all layers of the table in four launches; do_power_iteration = 0 only evaluates
 * sigma = u . (W v) with the stored vectors (eval mode).  w: [rows][cols] fp32; u [rows], v [cols] are updated in place (and
 * copied to u_copy / v_copy when given: the values this forward's backward needs); scratch per layer: t1 [cols], t2 [rows],
 * partial [mg_sn_layer_blocks(rows, cols, 2)][cols].  first_block_k1 / _k3 = running sums of mg_sn_layer_blocks(rows, cols, 0 / 1). */
typedef struct mg_sn_layer {
    const float* w; float* u; float* v; float* u_copy; float* v_copy; float* sigma;
    float* t1; float* t2; float* partial;
    int32_t rows, cols, first_block_k1, first_block_k3;
} mg_sn_layer;
int     mg_sn_power_iteration(const mg_sn_layer* layers_dev, int32_t nlayers, const int32_t* block_layer_k1, int32_t nblocks_k1,
                              const int32_t* block_layer_k3, int32_t nblocks_k3, int32_t do_power_iteration, float eps, void* stream);
int64_t mg_sn_layer_blocks(int32_t rows, int32_t cols, int32_t which);

/* Batched drain of GEMM-order weight gradients into reference-layout gradients (one call per optimiser step instead of a
 * fill + unpack + spectral-norm backward + accumulate chain per convolution).  mg_conv_wgrad ACCUMULATES (fp32 atomics) into
 * `gemm` / `dbias_gemm`, which therefore may live in a persistent arena; this call adds every slot's gradient to its
 * destination tensor(s) and leaves the drained GEMM memory zeroed:
 *   dst0[co][ci][t] (+ dst1 for the beta tensor of a fused SPADE gamma|beta pair, rows interleaved in blocks of 32)
 *       += gemm[t][row(co)][ci]                                   plain
 *       += (gemm[..] - s * u[co] * v[ci*taps + t]) / sigma[0]     spectral norm (w_sn != NULL), s = sum(gemm * w_sn)
 *   dbias0/1[co] += dbias_gemm[row(co)]
 * `swapped`: the GEMM image is [t][ci][row] (weight gradient computed with the operand roles exchanged).
 * `s` = one double per spectral-normed slot (written here: per-workgroup partials summed in a fixed order -> deterministic).  The table lives in device memory; `first_block` = running sum
 * of mg_grad_slot_blocks over the preceding slots and block_slot[b] = slot that owns workgroup b (host-built). */
typedef struct mg_grad_slot {
    float* gemm; float* dbias_gemm;
    float* dst0; float* dst1; float* dbias0; float* dbias1;
    const float* w_sn; const float* u; const float* v; const float* sigma; double* s;
    int32_t cout, cin, taps, rows, cols, swapped;
    int64_t first_block;
} mg_grad_slot;
int     mg_grad_drain(const mg_grad_slot* table_dev, int32_t nslots, const int32_t* block_slot_dev, int32_t nblocks,
                      double* partial, void* stream);    /* partial: nblocks doubles of scratch, or NULL when no slot has w_sn */
int64_t mg_grad_slot_blocks(int32_t cout, int32_t cin, int32_t ntens);        /* workgroups slot needs (host helper) */

/* Hinge GAN loss on a patch discriminator's 1-channel logit map with the wide-edge weight mask (loss.py:60-140).
 * mg_wide_edge_weight: label fp32 [N][Hl][Wl] in {0,1} -> weight fp32 [N][h][w] = e * wide + (1 - e), e = get_wide_edges of
 *   the label resized (nearest) to h x w with window k (= max(1, int(h * 0.06)), padding k / 2), resized back (nearest) from
 *   the pooled (h + 2(k/2) - k + 1)^2 map; depends only on the label: build once per batch and logit resolution.
 * mg_hinge_fwd: out[0] = -mean(f(x) * weight), x = n logits in `dtype`, weight fp32 or NULL (= 1);
 *   mode 0: f = x (generator), 1: f = min(x - 1, 0) (discriminator, real), 2: f = min(-x - 1, 0) (discriminator, fake).
 * mg_hinge_bwd: dx = -g[0] / n * f'(x) * weight in `dtype`. */
int mg_wide_edge_weight(const float* label, int32_t N, int32_t Hl, int32_t Wl, int32_t h, int32_t w, int32_t k, float wide,
                        float* out, void* stream);
int mg_hinge_fwd(const void* x, const float* weight, int32_t dtype, int64_t n, int32_t mode, float* out, void* stream);
int mg_hinge_bwd(const void* x, const float* weight, const float* g, int32_t dtype, int64_t n, int32_t mode, void* dx, void* stream);

/* Orientation-loss filter bank (loss.py:274-313: 32 oriented 17x17 Gabor filters on the gray image, clamp at 0,
 * max / arg-max over the 32 responses).  img is NHWC with C >= 3 (RGB in [-1,1] in channels 0..2); bank is
 * fp32 [32][17][17]; conf[N][H][W] = max_k max(resp_k, 0), idx[N][H][W] = first arg-max (u8).  The contraction
 * runs on the matrix cores in exact fp32.  bwd: dimg = adjoint of the winning-filter response w.r.t. the image
 * (dconf must already be zero where conf == 0, i.e. where the clamp was active). */
int mg_gabor_argmax_fwd(const void* img, const float* bank, float* conf, uint8_t* idx, int32_t dtype,
                        int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int mg_gabor_argmax_bwd(const float* dconf, const uint8_t* idx, const float* bank, void* dimg, int32_t dtype,
                        int32_t N, int32_t H, int32_t W, int32_t C, void* stream);

/* ---------------------------------------------------------------------------
 * Self-attention of the frozen orientation in-painting net (generator.py:467-485, SelfAttention.forward):
 *     out[n][i][:] = sum_j softmax_j( q[n][i][:] . k[n][j][:] ) * v[n][j][:]        i, j = 0 .. L-1 (L = H*W positions),
 * no 1/sqrt(d) scale (the reference has none).  q, k: [N][L][d_qk], v: [N][L][d_v], out: [N][L][d_v], all position-major
 * (= NHWC with H*W flattened) in `dtype`, rows ld* ELEMENTS apart (ld >= width: q / k / v may be column slices of one fused
 * projection, out the second half of the [x | out] concatenation the reference returns).  Flash-style: the [L, L] score
 * matrix is never materialised (the reference writes it: torch.bmm -> softmax -> torch.bmm); bf16: bf16 MFMA with fp32
 * accumulation, the probabilities enter the second product as hi + lo bf16 pairs (16 mantissa bits); fp32: exact-fp32 MFMA throughout.
 * Built for d_qk = 64, d_v = 256 (SelfAttention(256, downsample 4)); any L, any N <= 65535.
 * ------------------------------------------------------------------------- */
int mg_self_attention(const void* q, const void* k, const void* v, void* out, int32_t dtype, int32_t N, int32_t L,
                      int32_t d_qk, int32_t d_v, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, void* stream);

/* Fused Adam over one flat fp32 parameter buffer (torch.optim.Adam semantics,
 * pix2pix_model.py:137-145: eps 1e-8, no weight decay, bias correction).
 * step is the 1-based step count. */
int mg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                 int64_t numel, float lr, float beta1, float beta2, float eps, int32_t step,
                 float grad_scale, void* stream);

/* torch.nn.utils.spectral_norm (dim 0, one power iteration) as applied to every G / D convolution
 * (architecture.py:39-42, normalization.py:28-29), around the two gemv calls the host issues:
 *   mg_sn_normalize: dst (and dst2, optional) = t / max(||t||_2, eps); sigma (optional) = dst . t
 *   mg_sn_scale:     out = w / sigma[0]                                   (numel a multiple of 4)
 *   mg_sn_bwd:       out[r][c] = (g[r][c] - s[0] u[r] v[c]) / sigma[0]    (gradient through W / sigma, u and v constant,
 *                    s = sum(g * W_sn)); all fp32, device scalars. */
int mg_sn_normalize(const float* t, int32_t n, float eps, float* dst, float* dst2, float* sigma, void* stream);
int mg_sn_scale(const float* w, const float* sigma, float* out, int64_t numel, void* stream);
int mg_sn_bwd(const float* g, const float* u, const float* v, const float* s, const float* sigma, float* out,
              int32_t rows, int32_t cols, void* stream);

/* ---------------------------------------------------------------------------
 * Input pipeline on the device: the per-sample work the reference does with
 * PIL / numpy / cv2 in loader worker processes, from decoded u8 maps to the float tensors of the `data` dict.
 * Byte / index outputs are bit-exact restatements; all maps are dense, u8 sources are HWC as PIL decodes them.
 *
 * mg_input_crop_u8: resize(NEAREST, optional) + random crop + horizontal flip + ToTensor of get_transform
 *   (data/base_dataset.py:419-456) for a batch: dst[n][c][y][x] = conv(src[n][ytab[y0+y]][xtab[x0 + (flip ? W-1-x : x)]][c])
 *   with crop[n] = {x0, y0, flip} (device int32, get_params base_dataset.py:398-416), src u8 [N][Hs][Ws][C], dst fp32
 *   NCHW [N][C][H][W]; ytab / xtab (device int32 [load_h] / [load_w], NULL = identity) hold Pillow's nearest-neighbour
 *   source index per load-size coordinate (mg_nearest_table).  mode 0 (images): (v/255 - 0.5)/0.5 = ToTensor +
 *   Normalize(0.5, 0.5) -- images come at load size (mg_resize_bicubic_u8 brings them there); mode 1 (label,
 *   orientation and hole maps, pix2pix_dataset.py:72-73,117,146-147): (v/255)*255 and, when unknown_label >= 0, 255 ->
 *   unknown_label; mode 2: v/255 (the RGB orientation image, :127).  mul (optional, fp32 [N][1][H][W]) multiplies every
 *   channel ("* label_tensor", :127,133).  The float operations are single IEEE operations in the reference's order.
 * mg_onehot_labels: preprocess_input's FloatTensor(bs, nc, h, w).zero_().scatter_(1, label.long(), 1.0)
 *   (models/pix2pix_model.py:231-246); label fp32 [N][1][HW], out fp32 [N][nc][HW]; indices outside [0, nc) set nothing.
 * mg_orient_to_rgb_u8: trans_orient_to_rgb (base_dataset.py:363-385): out u8 [npix][3] = np.uint8(((cos 2t + 1)/2,
 *   (sin 2t + 1)/2, 0.5) * label * 255), t = orient/255*pi; `table` = device copy of mg_orient_rgb_table's 256x3 doubles.
 * mg_generate_hole_u8: generate_hole (base_dataset.py:335-361), one sample per workgroup: th[n] = the
 *   random.uniform(0.5, 1.2) draw, u[n] in [0,1) selects the centre among the non-zero pixels of orient_mask
 *   (center_idx = min(int(u * nums), nums - 1), row-major order as np.where lists them); hole u8 [N][H][W];
 *   info (optional, device int32 [N][4]) receives {nums, centre row, centre column, rr}.
 * mg_noise_octaves: generate_noise (base_dataset.py:387-396): out fp32 NCHW [N][3][S][S] = (sum over octaves o of
 *   cv2.resize(field_o, (S, S), INTER_LINEAR)) / n_octaves, fields = float64 Gaussian draws, per sample the octaves
 *   back to back, octave o = [S>>o][S>>o][3] while S>>o >= 8 (mg_noise_field_len(S) doubles per sample).
 * Host helpers (no GPU): mg_nearest_table fills table[dst] with Pillow's source index walk; mg_orient_rgb_table fills
 *   256x3 doubles; mg_noise_field_len returns the per-sample field length. */
int mg_input_crop_u8(const uint8_t* src, float* dst, const int32_t* crop, const int32_t* ytab, const int32_t* xtab,
                     const float* mul, int32_t N, int32_t Hs, int32_t Ws, int32_t C, int32_t H, int32_t W,
                     int32_t mode, int32_t unknown_label, void* stream);
int mg_onehot_labels(const float* label, float* out, int32_t N, int64_t HW, int32_t nc, void* stream);
int mg_orient_to_rgb_u8(const uint8_t* orient, const uint8_t* label, const double* table, uint8_t* out, int64_t npix, void* stream);
int mg_generate_hole_u8(const uint8_t* mask, const uint8_t* orient_mask, const double* th, const double* u,
                        uint8_t* hole, int32_t* info, int32_t N, int32_t H, int32_t W, void* stream);
int mg_noise_octaves(const double* fields, float* out, int32_t N, int32_t S, void* stream);
/* mg_resize_bicubic_u8: transforms.Resize(osize, Image.BICUBIC) on u8 images (base_dataset.py:421-424; arithmetic =
 *   Pillow Resample.c, 8 bits per channel): horizontal pass src [N][Hs][Ws][C] -> tmp [N][Hs][Wd][C], vertical pass ->
 *   dst [N][Hd][Wd][C]; x/y bounds (device int32 [out][2] = first source index, count) and 22-bit fixed-point
 *   coefficients (device int32 [out][k]) are copies of mg_bicubic_table's host output, k = mg_bicubic_ksize(in, out). */
int mg_resize_bicubic_u8(const uint8_t* src, uint8_t* tmp, uint8_t* dst, const int32_t* xbounds, const int32_t* xcoef, int32_t kx,
                         const int32_t* ybounds, const int32_t* ycoef, int32_t ky, int32_t N, int32_t Hs, int32_t Ws,
                         int32_t Hd, int32_t Wd, int32_t C, void* stream);
int     mg_inputs_set_option(int32_t key, int32_t value);   /* key 0: 0 = per-pixel noise kernel (default), 1 = LDS-tiled form (A/B); bit-identical results */
int     mg_bicubic_ksize(int32_t in_size, int32_t out_size);
int     mg_bicubic_table(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* coef);
int     mg_nearest_table(int32_t src, int32_t dst, int32_t* table);
int     mg_orient_rgb_table(double* table);
int64_t mg_noise_field_len(int32_t S);

/* ---- per-pixel glue around the convolutions (mg_glue.hip): each entry replaces a chain of eager element-wise ops with one launch ---- */

/* F.interpolate(mode='nearest') of planar fp32 maps (plane c of sample n starts at plane[c] + n * nstride[c], H x W) to `nlev`
 * resolutions at once, written as NHWC tensors out[l] = [N, h[l], w[l], cout] in `dtype` with channels >= nplanes zeroed: the SPADE
 * conditioning pyramid (normalization.py:109) and the hair / background mask pyramids (generator.py:186-219, encoder.py:331-341).
 * Source index = min(floor(dst * (float)in / out), in - 1), as ATen's nearest kernel computes it. */
typedef struct mg_pyramid_desc {
    const float* plane[8];
    int64_t nstride[8];
    int32_t nplanes, N, H, W, nlev, cout, dtype, pad_;
    int32_t h[8], w[8];
    void* out[8];
} mg_pyramid_desc;
int mg_nearest_pyramid(const mg_pyramid_desc* d, void* stream);

/* Mask half of PartialConv2d with a single-channel mask (partialconv2d.py:55-75): window sum S of mask_in [N,H,W] over k x k,
 * stride s, zero padding p -> upd = clamp(S, 0, 1), scale = k*k / (S + 1e-8) * upd, both fp32 [N,h,w]. */
int mg_pconv_mask(const float* mask_in, int32_t N, int32_t H, int32_t W, int32_t k, int32_t s, int32_t p,
                  float* scale, float* upd, void* stream);

/* y[p, c] = x[p, c] * a[p] (bias == NULL) or bias[c] * b[p] + x[p, c] * a[p] over P pixels x C channels (NHWC, C % 4 == 0): the
 * partial convolution's input masking x * m and output renormalisation ((raw - b) * ratio + b) * m' = raw * (ratio m') + b * m'.
 * bf16: a, b and bias are rounded to bf16 and bias * b is rounded once before the fused multiply-add, like the eager bf16 ops. */
int mg_pixel_affine(const void* x, const float* a, const float* bias, const float* b, int32_t dtype, int64_t P, int32_t C,
                    void* y, void* stream);

/* Input of the background encoder (encoder.py:288-320): mode 0: back = 1 - maxpool_kxk(hair) (k odd, stride 1, padding k/2);
 * mode 1: back = hair as given.  inp[n, y, x, 0:3] = image * back + noise * (1 - back) (noise == NULL: image * back; image == NULL:
 * noise), channels 3..7 zero, NHWC8 in `dtype`; image / noise are NCHW fp32 with 3 channels, hair is plane [H, W] of sample n at
 * hair + n * hair_nstride.  back: fp32 [N, H, W]. */
int mg_bg_compose(const float* image, const float* noise, const float* hair, int64_t hair_nstride, int32_t dtype,
                  int32_t N, int32_t H, int32_t W, int32_t k, int32_t mode, void* inp, float* back, void* stream);

/* ImageEncoder3's tail (encoder.py:211-220) and its adjoint: out[n, q, c] = w_out[n, q] * sum_p(x[n, p, c] * w_in[n, p]) / max(sum_p w_norm[n, p], 1),
 * x [N, P, C] in `dtype` (C % 4 == 0), the three weights fp32 [N, P], out fp32 [N, P, C].  Forward: w_in = w_norm = reference hair mask,
 * w_out = target hair mask; backward (x = the incoming gradient): w_in = target mask, w_out = w_norm = reference mask. */
int mg_masked_mean_fill(const void* x, const float* w_in, const float* w_out, const float* w_norm, int32_t dtype, int32_t N, int32_t P,
                        int32_t C, float* out, void* stream);

/* Tail of the Gabor orientation loss behind mg_gabor_argmax_fwd (loss.py:352-385).  conf_raw fp32 [N,HW] (max clamped response),
 * idx u8 [N,HW] (winning filter), label: label_ch == 2: planes (sin 2t, cos 2t) of sample n at label + n * label_nstride (+ HW for
 * the second); label_ch == 1: the loader's 0..255 angle map.  confidence = (tanh(conf_raw) + 1) / 2, fake = (sin 2a, cos 2a) *
 * confidence with a = idx * pi / 32.  out[0] = mean |fake * hair - label * hair| over N*2*HW, out[1] = -sum(log(clamp(confidence,
 * 1e-3, 1)) * hair) / sum(hair), out[2] = sum(hair); ws: >= 3072 floats.  bwd: dconf = d(g_orient[0] out[0] + g_conf[0] out[1]) / d conf_raw
 * (either gradient pointer may be NULL = 0); fwd_out = the forward's out. */
int mg_orient_loss_fwd(const float* conf_raw, const uint8_t* idx, const float* label, int32_t label_ch, int64_t label_nstride,
                       const float* hair, int64_t hair_nstride, int32_t N, int64_t HW, float* out, float* ws, void* stream);
int mg_orient_loss_bwd(const float* conf_raw, const uint8_t* idx, const float* label, int32_t label_ch, int64_t label_nstride,
                       const float* hair, int64_t hair_nstride, const float* g_orient, const float* g_conf, const float* fwd_out,
                       int32_t N, int64_t HW, float* dconf, void* stream);

/* ---------------------------------------------------------------------------
 * (iv) Collectives over RCCL / xGMI -- SURVEY.md section 8b's last export group.  One communicator per process (= per GPU); what the
 * reference does with Python threads inside ONE process is replaced by these in a one-process-per-GPU job:
 *   mg_allreduce_stats   sync_batchnorm/batchnorm.py:105-126 (_data_parallel_master: the master sums every replica's [sum x | sum x^2]
 *                        -- ReduceAddCoalesced, :117 -- and broadcasts mean / inv_std back, :120) and comm.py:49-53,102-133 (the
 *                        SyncMaster / SlavePipe rendez-vous): ONE in-place sum all-reduce of the [2C] vector, fp64 in forward
 *                        (`is_f64` = 1), fp32 for the backward's [sum dxhat | sum dxhat * xhat] (`is_f64` = 0); every rank then runs the
 *                        same finalize kernel (mg_channel_stats_finalize / mg_norm_finalize) on the reduced sums.
 *   mg_allreduce_grads   the backward of nn.DataParallel's replicate / gather (pix2pix_trainer.py:21-24: gradients reduce-added onto
 *                        GPU 0) as an in-place fp32 sum all-reduce of one contiguous gradient bucket (the 1/world average is folded
 *                        into mg_adam_step's `grad_scale`).
 * Both enqueue on `stream` and return; nothing synchronises.  RCCL is bound at run time (dlopen of librccl.so.1 -- inside a PyTorch-ROCm
 * process that is the copy torch already loaded, so both share one runtime); a process without RCCL gets MG_ERR_UNSUPPORTED from
 * mg_comm_unique_id / mg_comm_init and never needs the library otherwise.
 *   rendez-vous: rank 0 calls mg_comm_unique_id and hands the 128 bytes to the other ranks out of band (the launcher's store /
 *   torch.distributed broadcast / a file); every rank then calls mg_comm_init(id, rank, world) with ITS device current.
 * ------------------------------------------------------------------------- */
#define MG_COMM_ID_BYTES 128
int mg_comm_unique_id(void* id_out /* host, MG_COMM_ID_BYTES */);
int mg_comm_init(const void* id /* host, MG_COMM_ID_BYTES */, int32_t rank, int32_t world, int64_t* comm_out);
int mg_comm_destroy(int64_t comm);
int mg_comm_world(int64_t comm, int32_t* rank_out, int32_t* world_out);
int mg_allreduce_stats(int64_t comm, void* sums /* device, [n] fp64 or fp32, in place */, int64_t n, int32_t is_f64, void* stream);
int mg_allreduce_grads(int64_t comm, float* bucket /* device, [n] fp32, in place */, int64_t n, void* stream);

/* Hardware probes used by the test-suite (MFMA / ds_read_tr fragment maps). */
int mg_probe_mfma_layout(float* out /* [3][64][16] */, void* stream);
int mg_probe_tr16(const uint16_t* in /* [64][4] elements via LDS */, uint16_t* out /* [64][4] */, void* stream);

/* Tuning switches for A/B measurements (value 0 / 1, all default 1): key 0 = conv pipeline (0 register-staged
 * double buffer, 1 LDS-DMA ring); 1 = allow 256x256 tiles; 2 = 3x3 halo-tile kernel; 3 = kernel-row 3x3 weight-
 * gradient kernel; 4 = 128-channel x 16x16-pixel halo tiles; 5 = split-K for low-resolution long-K convolutions;
 * 6 = kernels for convolutions over an 8-channel input, forward and weight gradient (0 none, 1 only the register-weight 3x3 / stride-1
 *     ones, 2 default: also the LDS-weight kernels for any window of at most 7x7 taps at stride 1 or 2);
 * 7 = bf16 conv epilogues exchange channel quads between the two half-waves (v_permlane32_swap) and store 16 bytes per lane;
 * 8 = wave-per-pixel dot-product kernel for convolutions with <= 4 output channels over >= 128 input channels (the discriminators' heads).
 * 9 = weight-slab ring depth of the big halo tile (3 default, or 4); 15 = 1: the SPADE halo kernel loads x in its epilogue instead of
 * ahead of its K loop (default 0); 17 = 1: two K slices also for 161..320-workgroup launches with >= 256 K steps (default 0);
 * 18 = stages a split of the generic weight-gradient kernel keeps at least (default 32); 22 = the register-resident-weights kernel for 3x3 convolutions
 * over exactly 64 input channels (mg_conv_halo64.hip; bitwise the results of the kernel it replaces); 24 = pixel width of the column stripes the
 * kernel-row 3x3 weight-gradient kernel walks inside an image (a multiple of 32; default 64; 0 = whole image rows in raster order).
 * Results agree within accumulation-order rounding whatever the setting (each setting is bit-reproducible except
 * the weight gradients, which use fp32 atomics).
 * MEASUREMENT builds (wrong or no results, timing only; tools/probe_halo.py, tools/probe_wgrad3x3.py): key 10 = 1..6 variants of the big
 * halo tile (K loop only / no weight stream / no barrier / per-tap stamps / per-phase stamps / stores off), 12 = 1 stamped build of the
 * 3x3 weight-gradient kernel, 13 and 14 = low and high half of the device address the stamps go to, 21 = bytes of extra dynamic LDS per
 * halo-conv workgroup (81920 leaves ONE resident per CU = one wave per SIMD: tools/gate1_lone_wave.py).  All default 0. */
int         mg_set_option(int32_t key, int32_t value);

/* sizeof(mg_conv_desc) (which=0) / sizeof(mg_wgrad_desc) (which=1): lets a
 * foreign-language binding check its struct mirror without a GPU. */
int         mg_sizeof_desc(int32_t which);
int         mg_abi_version(void);
const char* mg_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MICHIGAN_HIP_H */
