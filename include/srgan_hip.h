/* srgan_hip.h -- C ABI of libsrgan_hip.so: the MI355X (gfx950) compute layer of the SRGAN/SGAN training step.
 *
 * The reference (golmschenk/sr-gan) is pure Python on PyTorch and has NO FFI of its own: its
 * arithmetic is reached through torch.nn / torch.nn.functional / torch.optim call sites.  Each entry point
 * below therefore cites the reference call site(s) whose arithmetic it replaces.  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns int: 0 = ok, < 0 = argument / shape error (-1 invalid, -2 unsupported,
 *     -3 out of range), > 0 = hipError_t.  srgan_last_error() returns a thread-local message.
 *   - pointers are DEVICE pointers to contiguous fp32 (int32 for arg-max), borrowed for the call only;
 *     the library never allocates or frees device memory.  The one block it keeps between calls is the split-K
 *     workspace the CALLER registers per (device, stream) with srgan_set_workspace (below); a contraction that
 *     needs it on a stream without one fails with -1 and a message.  Activations are NCHW, conv weights
 *     [K, C, R, S], transposed-conv weights [Cin, Cout, R, S] (torch layouts).
 *   - thread safety: calls may come from several host threads (the registry and the profile state are locked,
 *     srgan_last_error is thread-local); two threads must not launch on the SAME stream with the same workspace
 *     concurrently, as with any stream-ordered resource.
 *   - collectives: thin RCCL entry points (srgan_comm_*, srgan_all_reduce_sum, srgan_reduce_scatter_sum,
 *     srgan_all_gather, srgan_broadcast, at the end of this file) carry the data-parallel exchange for a caller without
 *     torch.distributed; they are the Python host's default device transport as well (sr-gan_amd/parallel.py: torch.distributed
 *     keeps the rendezvous and the host-side control messages; `SRGAN_ABI_COLLECTIVES=0` puts the device collectives back on
 *     torch.distributed's "nccl" backend = the same RCCL over xGMI), see INTEGRATION.md.
 *   - `stream` is a hipStream_t (NULL = default stream); every call is asynchronous on it.
 *   - `accumulate` != 0 adds into the existing output (gradient accumulation across the four
 *     discriminator backward passes, reference srgan.py:280-295) instead of overwriting it.
 *   - tensors are limited to < 2^31 elements.
 *   - `force_kernel`: 0 = automatic choice, 1 = direct VALU form, 2 = MFMA form (used by tests to
 *     cross-check the two implementations).
 */
#ifndef SRGAN_HIP_H
#define SRGAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int srgan_version(void);               /* 110 = ABI 1.1 (see "ABI history" in INTEGRATION.md; 100 = ABI 1.0) */
/* 16 hex digits: sha256 over the kernel sources this library was compiled from (sr-gan_amd/_build.py source_id()).
 * The Python loader refuses a library whose id differs from the sources next to it: a stale prebuilt .so must not
 * silently run old kernels under new parity tests. */
const char* srgan_build_id(void);
const char* srgan_last_error(void);

/* ---- capabilities, caller-owned workspace -------------------------------------------------------------------
 * srgan_capabilities fills the struct below (pass sizeof(srgan_capabilities_t)); SRGAN_DTYPE_* / SRGAN_FEATURE_*
 * are bit masks.  Reference analogue: none (torch picks kernels internally); SURVEY.md 8b asks for the query. */
#define SRGAN_DTYPE_F32 0x1u            /* fp32 in / fp32 accumulate (v_mfma_f32_32x32x2_f32): the parity path */
#define SRGAN_DTYPE_BF16 0x2u           /* bf16 MFMA operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16) */
#define SRGAN_DTYPE_F16 0x4u            /* fp16 MFMA operands, fp32 accumulate (v_mfma_f32_32x32x16_f16) */
#define SRGAN_FEATURE_FUSED_BNRELU 0x1u /* norm -> relu -> conv evaluated inside the convolution kernels */
#define SRGAN_FEATURE_SPLITK_WORKSPACE 0x2u
#define SRGAN_FEATURE_LIVE_PROFILE 0x4u
#define SRGAN_FEATURE_BLOCKED16 0x8u    /* ABI 1.1: the srgan_h_* entry points (16-bit storage in the blocked layout) */
typedef struct srgan_capabilities_t {
  int32_t abi_version;          /* = srgan_version() */
  int32_t struct_bytes;         /* sizeof(srgan_capabilities_t) as the library was built */
  char arch[16];                /* "gfx950" */
  uint32_t dtypes;              /* SRGAN_DTYPE_* */
  uint32_t features;            /* SRGAN_FEATURE_* */
  int64_t workspace_bytes;      /* = srgan_workspace_bytes() */
  int64_t max_tensor_elements;  /* 2^31 - 1 */
} srgan_capabilities_t;
int srgan_capabilities(srgan_capabilities_t* out, int32_t out_bytes);
/* Every sum that several workgroups of one launch share -- the K slices of a split contraction, the workers of a weight
 * gradient, per-channel parameter sums, per-example loss sums -- goes through a workspace: the workgroups leave their
 * partial tiles there and the last one (or a second launch) adds them in a fixed order (round 5: no fp32 atomics on data,
 * bit-reproducible results; round 2 used it only for tiny split-K outputs).  The caller owns that memory: register one
 * block of >= srgan_workspace_bytes() bytes (256 MiB; 16-byte aligned, device memory) per (current device, stream) before
 * the first launch on that stream; NULL unregisters.  Launches on one stream are ordered, so one block per stream is
 * enough.  A stream without a workspace still computes: the sums then meet through fp32 atomics. */
int64_t srgan_workspace_bytes(void);
int srgan_set_workspace(void* workspace, int64_t bytes, void* stream);
/* Round 5: a contraction that splits K over several workgroups (few output tiles: the small planes) finishes in a FIXED
 * order on a stream that has a workspace: every slice leaves its partial tile there, the tile's last workgroup adds the
 * slices in slice order and stores -- one launch, no zero-filled output, bit-identical from run to run (the reference's CPU
 * path is repeatable; fp32 atomics are not).  Returns 1 when launches on `stream` do so, 0 when they fall back to fp32
 * atomics into a zero-filled output (no workspace registered, or SRGAN_ATOMIC_SPLIT=1). */
int srgan_split_is_ordered(void* stream);

/* ---- convolution ------------------------------------------------------------------------------------------
 * Geometry of y = conv2d(x, w): x [N,C,H,W], w [K,C,R,S], y [N,K,OH,OW].  Batch strides (elements) allow a
 * channel-slice view of a wider buffer; 0 means dense. */
typedef struct srgan_conv_desc {
  int32_t N, C, H, W;
  int32_t K, R, S;
  int32_t stride_h, stride_w, pad_h, pad_w;
  int32_t OH, OW;
  int64_t x_batch_stride, y_batch_stride;
  int32_t compute_dtype;   /* SRGAN_COMPUTE_*: the MFMA operand type of this pass */
} srgan_conv_desc;

/* Mixed precision (BASELINE.json configs 2 and 5: "bf16", "fp16 with fp32 GP"; the reference itself is fp32-only).
 * Tensors stay fp32 in HBM -- activations, weights (the master copy), gradients, Adam moments -- and accumulation is
 * fp32; with SRGAN_COMPUTE_BF16 / _F16 the two operands of every contraction are rounded to bf16 / fp16 when the MFMA
 * fragments are formed (v_mfma_f32_32x32x16_bf16 / _f16, 16x the fp32 matrix rate).  The choice is per call, so a
 * caller keeps e.g. the gradient-penalty chain in SRGAN_COMPUTE_F32 while the rest of the step runs in fp16.  The
 * specialised fp32 kernels (fused batch-norm forms included: srgan_conv2d_bnrelu_supported then returns 0) are not
 * used in these modes; fp32 remains the parity path (1e-3), the mixed modes are tested against it with their own,
 * stated tolerances (tests/test_mixed_precision_gpu.py). */
#define SRGAN_COMPUTE_F32 0
#define SRGAN_COMPUTE_BF16 1
#define SRGAN_COMPUTE_F16 2

/* y = conv2d(x, w) + bias[k] (bias may be NULL).
 * Replaces torch.nn.Conv2d.forward at reference crowd/models.py:340-345,369,770-775,1072,1134-1135;
 * age/models.py:24,61-65; age/vgg.py:78. */
int srgan_conv2d_fwd(const srgan_conv_desc* desc, const float* x, const float* w, const float* bias, float* y,
                     int force_kernel, void* stream);

/* gx = d(conv2d)/dx applied to gy, + bias[c] (bias may be NULL).  Also IS the forward of
 * torch.nn.ConvTranspose2d (weight [Cin=K, Cout=C, R, S], input = gy, output = gx): reference
 * age/models.py:16-21,37-41, crowd/models.py:132-136,768-769.  As a backward pass it replaces autograd's
 * conv backward-to-input reached from reference srgan.py:280-295,304 and the create_graph pass of
 * srgan.py:368-370. */
int srgan_conv2d_bwd_data(const srgan_conv_desc* desc, const float* gy, const float* w, const float* bias, float* gx,
                          int accumulate, int force_kernel, void* stream);

/* gw = d(conv2d)/dw applied to gy (reduction over batch and output pixels). */
int srgan_conv2d_bwd_weight(const srgan_conv_desc* desc, const float* x, const float* gy, float* gw, int accumulate,
                            int force_kernel, void* stream);

/* Convolutions whose input is relu(batch_norm_eval(x)) evaluated ON THE FLY inside the kernel (the DenseNet
 * norm -> relu -> conv triples, reference crowd/models.py:338-345): the normalised tensor is never written to HBM.
 * Zero padding applies to the activated tensor, as in the reference sequence.  Supported geometries: 1x1 / stride 1 /
 * unpadded on images of a multiple of 32 pixels, and 3x3 / stride 1 / pad 1 with W >= 16 (weight gradient: W % 4 ==
 * 0, C >= 32; forward: C <= 512); srgan_conv2d_bnrelu_supported(desc, pass) (pass 0 forward, 1 data gradient,
 * 2 weight gradient) returns 1 when the fused form exists, otherwise the caller materialises the activation
 * (srgan_chan_affine_act) and uses the plain entry points.  Results are identical to that two-step form up to
 * summation order.
 * srgan_conv2d_bwd_data_bnrelu is the backward of the triple w.r.t. x in one kernel: gx (=, +=)
 * conv_bwd_data(gy, w) * [batch_norm(x) > 0] * inv_std * gamma, with x and gx sharing desc->x_batch_stride (gx may be
 * a channel-slice view that is accumulated into), plus -- when g_gamma / g_beta are given (both or neither) -- the
 * batch-norm parameter gradients accumulated into them.  Equals srgan_conv2d_bwd_data followed by srgan_bn_act_bwd. */
typedef struct srgan_bn_relu { const float* mean; const float* inv_std; const float* gamma; const float* beta; } srgan_bn_relu;
int srgan_conv2d_bnrelu_supported(const srgan_conv_desc* desc, int pass);
int srgan_conv2d_fwd_bnrelu(const srgan_conv_desc* desc, const float* x, const srgan_bn_relu* bn, const float* w,
                            const float* bias, float* y, void* stream);
/* Small problems split K over the grid and add their partial results with fp32 atomics into a zeroed y: one extra
 * zero-fill launch per convolution.  srgan_conv2d_fwd_bnrelu_splits(desc) tells how many K splits the forward of this
 * geometry uses (1: plain stores; -1: no fused form); a caller that hands over y ALREADY ZERO (one fill for the outputs
 * of a whole dense block) calls srgan_conv2d_fwd_bnrelu_into_zeros, which skips the per-call zero-fill. */
int srgan_conv2d_fwd_bnrelu_splits(const srgan_conv_desc* desc);
int srgan_conv2d_fwd_bnrelu_into_zeros(const srgan_conv_desc* desc, const float* x, const srgan_bn_relu* bn, const float* w,
                                       const float* bias, float* y, void* stream);
int srgan_conv2d_bwd_data_bnrelu(const srgan_conv_desc* desc, const float* gy, const float* w, const srgan_bn_relu* bn,
                                 const float* x, float* gx, float* g_gamma, float* g_beta, int accumulate, void* stream);
int srgan_conv2d_bwd_weight_bnrelu(const srgan_conv_desc* desc, const float* x, const srgan_bn_relu* bn, const float* gy,
                                   float* gw, int accumulate, void* stream);
/* Deferred parameter gradients of srgan_conv2d_bwd_data_bnrelu: the per-workgroup sums behind g_gamma / g_beta are left in
 * a caller-owned region of 2 * srgan_conv2d_bwd_data_bnrelu_tiles(desc) * C floats instead of being reduced by a small
 * kernel after EVERY convolution (597 launches per training step); one srgan_bn_partial_reduce_batched call then
 * reduces the regions of many convolutions (a DenseNet block's backward pass: two per layer).  `jobs_device` is a table in
 * DEVICE memory; partial_offset is in floats from `scratch`.  Offsets and pointers are stable from step to step, so the
 * caller builds the table once per block. */
typedef struct srgan_bn_reduce_job {
  int64_t partial_offset;
  int32_t tiles, channels;
  const float* inv_std;
  float* g_gamma;
  float* g_beta;
} srgan_bn_reduce_job;
int64_t srgan_conv2d_bwd_data_bnrelu_tiles(const srgan_conv_desc* desc);
int srgan_conv2d_bwd_data_bnrelu_partials(const srgan_conv_desc* desc, const float* gy, const float* w, const srgan_bn_relu* bn,
                                          const float* x, float* gx, float* partials, int accumulate, void* stream);
int srgan_bn_partial_reduce_batched(const srgan_bn_reduce_job* jobs_device, int32_t count, int32_t max_channels, int32_t max_tiles,
                                    const float* scratch, void* stream);

/* Grouped form of srgan_conv2d_bwd_weight_bnrelu: MANY independent weight gradients (all the 1x1, or all the 3x3,
 * convolutions of a dense block's backward: reference crowd/models.py:335-353 through loss.backward()) in ONE launch.
 * srgan_wgrad_group_plan fills one 128-byte table slot per problem on the host -- x / gy are ELEMENT OFFSETS from two base
 * pointers given at launch time, gw and the batch-norm vectors are absolute, the gradient is ACCUMULATED into gw -- and
 * returns the grid extent the problem needs and, in *ragged, the kernel variant the slot was planned for as a bit (3x3: 0 / 1 =
 * the ragged variant; 1x1: 1 = the LDS-staged 128 x 128 tiles, 2 = the register-streamed 64 x 64 tiles, 4 = their ragged
 * variant); the caller uploads the table once (it does not change between steps when the offsets are relative to per-step
 * buffers) and launches with the maxima of the grid extents and the OR of the variants over the group (a group whose slots
 * were planned for both 1x1 forms is served by one launch each).  All problems of a group share the plane size; group_size
 * (the number of problems that will be launched together) lets the plan give each problem fewer workgroups of its own, and
 * group_weights (the sum of CO x CI x taps over the group, 0 = not known) lets it share them out by work instead of in equal
 * parts (the staged 1x1 form: the same K range per worker whatever the problem's width).  bn == NULL plans the plain weight gradient
 * (x used as is; srgan_wgrad_group_run with fused_bn = 0): the double backward's gradients w.r.t. the scaled weights.
 * gw == NULL: the gradient goes to gw_base + gw_offset of the launch (a per-step buffer) instead of a fixed address.  sum_co_ci_taps / pixels / operand_elements only feed
 * the profile (logical FLOPs = 2 * sum_co_ci_taps * pixels; elements of x and gy read once).
 * Round 5, the ORDERED form: the K-slice workers of a problem no longer meet in gw through fp32 atomics -- each keeps its
 * partial tile in the stream's workspace and a second launch adds them in slice order (bit-identical from run to run).  The
 * caller lays the problems' partial regions out back to back: partial_offset = the sum of the *partial_floats the earlier
 * problems of the group returned, and srgan_wgrad_group_run gets the group's total (0, a total beyond the workspace, or no
 * workspace on the stream: fp32 atomics as before). */
int srgan_wgrad_group_plan(const srgan_conv_desc* desc, const srgan_bn_relu* bn, int64_t x_offset, int64_t gy_offset, float* gw,
                           int64_t gw_offset, int32_t group_size, int64_t group_weights, int64_t partial_offset, void* job,
                           int32_t* grid_x, int32_t* grid_y, int32_t* ragged, int64_t* partial_floats);
int srgan_wgrad_group_run(const void* jobs, int32_t count, int32_t kernel_size, int32_t grid_x, int32_t grid_y, int32_t ragged,
                          int32_t fused_bn, const float* x_base, const float* gy_base, float* gw_base, int64_t sum_co_ci_taps,
                          int64_t pixels, int64_t operand_elements, int64_t partial_floats, void* stream);

/* ---- strided GEMM  C[i*sci + j*scj] (=,+=) sum_k A[i*sai + k*sak] * B[k*sbk + j*sbj] + bias ----------------
 * C must be a dense M x N matrix (row- or column-major).  bias is indexed by row i, or by column j when
 * bias_on_columns != 0.  Replaces torch.nn.Linear forward/backward at reference coefficient/models.py:17-27,
 * 36-49,80-92 and age/vgg.py:33-41. */
int srgan_gemm_f32(int32_t M, int32_t N, int32_t K, const float* A, int64_t sai, int64_t sak, const float* B,
                   int64_t sbk, int64_t sbj, float* C, int64_t sci, int64_t scj, const float* bias,
                   int32_t bias_on_columns, int accumulate, int force_kernel, void* stream);
/* The same with the MFMA operand type chosen per call (compute_dtype = SRGAN_COMPUTE_*; srgan_gemm_f32 = fp32). */
int srgan_gemm(int32_t M, int32_t N, int32_t K, const float* A, int64_t sai, int64_t sak, const float* B,
               int64_t sbk, int64_t sbj, float* C, int64_t sci, int64_t scj, const float* bias,
               int32_t bias_on_columns, int accumulate, int force_kernel, int compute_dtype, void* stream);

/* ---- elementwise ------------------------------------------------------------------------------------------
 * Unary op codes: 0 copy, 1 neg, 2 abs, 3 sign, 4 sqrt, 5 exp, 6 log, 7 log1p, 8 square, 9 reciprocal, 10 tanh,
 * 11 relu, 12 step (x > 0), 13 affine (p0*x + p1), 14 pow (x^p0), 15 leaky_relu (slope p0), 16 sigmoid,
 * 17 softplus, 18 one_minus_square, 19 rsqrt.
 * Binary op codes: 0 add, 1 sub, 2 mul, 3 div, 4 div_safe (0 where b == 0), 5 max,
 * 6 leaky_mask_mul (a where b > 0 else p0*a: the backward of leaky_relu / relu), 7 axpy (a + p0*b).
 * Replace torch elementwise call sites: leaky_relu/relu/tanh (reference coefficient/models.py:24-26;
 * age/models.py:47-51,70-73; crowd/models.py:339,343,780-784,1150,1154), the distance functions
 * (utility.py:201-243), logsumexp / BCE / CE pieces (utility.py:161-182, sgan.py:14-15). */
int srgan_ew_unary(int op, const float* x, float* y, int64_t n, float p0, float p1, void* stream);
int srgan_ew_binary(int op, const float* a, const float* b, float* y, int64_t n, float p0, void* stream);
int srgan_fill(float* y, int64_t n, float value, void* stream);

/* y[n,c,i] = ((x ? x[n,c,i] : 1) - mean[c]) * scale_a[c] * scale_b[c] + shift[c], i < HW; every vector optional.
 * Frozen (eval-mode) batch-norm in one pass (mean = running_mean, scale_a = 1/sqrt(running_var + eps),
 * scale_b = gamma, shift = beta: reference srgan.py:538-542 + crowd/models.py:338,342,367,1073,1091) and every
 * broadcast: with N = 1 a per-row scale (gradient-penalty norms), with HW = 1 a per-column one. */
int srgan_chan_affine(const float* x, const float* mean, const float* scale_a, const float* scale_b, const float* shift,
                      float* y, int32_t N, int32_t C, int64_t HW, void* stream);

/* As srgan_chan_affine, then optionally y = max(y, 0) (relu != 0) and y = 0 where mask[n,c,i] <= 0 (mask may be
 * NULL): frozen batch-norm + ReLU forward in one pass, and its input gradient g * [y > 0] * gamma / sigma
 * (reference crowd/models.py:338-339,342-343,367-368,1073-1074,1149-1150). */
int srgan_chan_affine_act(const float* x, const float* mean, const float* scale_a, const float* scale_b,
                          const float* shift, const float* mask, int relu, float* y, int32_t N, int32_t C, int64_t HW,
                          void* stream);
/* srgan_chan_affine_act on channel-slice views: image n of x / mask / y starts at n * its batch stride (0 = dense), and
 * accumulate != 0 adds into y.  Used by the concat-free dense block: batch-norm reads a slice of the block
 * buffer; its input gradient is accumulated into a slice of the block's gradient buffer. */
int srgan_chan_affine_act_strided(const float* x, const float* mean, const float* scale_a, const float* scale_b,
                                  const float* shift, const float* mask, int relu, float* y, int32_t N, int32_t C,
                                  int64_t HW, int64_t x_batch_stride, int64_t mask_batch_stride, int64_t y_batch_stride,
                                  int accumulate, void* stream);
/* Per-weight-element part of the DOUBLE backward of norm -> relu -> conv (gradient penalty, reference
 * srgan.py:360-375 through crowd/models.py:338-345), w and q shaped [CO][CI][taps], a = inv_std * gamma per input
 * channel: w_scaled = w * a (optional); when q (the weight gradient taken with the masked, unscaled tangent) is
 * given: w_grad += q * a and gamma_grad[ci] += inv_std[ci] * sum_{co,tap} w * q. */
int srgan_bn_conv_tangent_weights(const float* w, const float* q, const float* inv_std, const float* gamma,
                                  float* w_scaled, float* w_grad, float* gamma_grad, int32_t CO, int32_t CI, int32_t taps,
                                  void* stream);

/* The same for many convolutions in ONE launch (every layer of a dense block's double backward).  The `_job` call fills
 * one 64-byte table slot on the host: the parameters where they live, and `offset` (elements) of this convolution's
 * scaled weights / weight gradient q inside two per-step buffers; the caller uploads the table once.  `_grouped` with
 * scaled_base != NULL writes every w_scaled (start of the double backward), with q_base != NULL it applies the gradient
 * part (its end); max_inner = max CI * taps, max_co = max CO over the table. */
int srgan_bn_conv_tangent_weights_job(const float* w, const float* inv_std, const float* gamma, float* w_grad,
                                      float* gamma_grad, int64_t offset, int32_t CO, int32_t CI, int32_t taps, void* job);
int srgan_bn_conv_tangent_weights_grouped(const void* jobs, int32_t count, int32_t max_inner, int32_t max_co, float* scaled_base,
                                          const float* q_base, void* stream);

/* Whole backward of frozen batch-norm (+ReLU when relu != 0) in one pass over (g, x): the activation mask is
 * recomputed from x (y = fma(x, a, b), a = inv_std*gamma, b = beta - mean*a, exactly the forward's arithmetic);
 * gx (=,+=) g*[y>0]*a (gx may be NULL; g / x / gx may be channel-slice views, batch stride 0 = dense; unscaled != 0
 * drops the factor a: gx = g*[y>0], the masked tangent used by the double backward of the gradient penalty);
 * g_gamma[c] += inv_std[c] * sum g*[y>0]*(x - mean[c]) and g_beta[c] += sum g*[y>0] with fp32 atomics (both NULL =
 * frozen parameters).  Replaces the autograd backward of reference crowd/models.py:338-343 (norm+relu pairs). */
int srgan_bn_act_bwd(const float* g, const float* x, const float* mean, const float* inv_std, const float* gamma,
                     const float* beta, int relu, float* gx, float* g_gamma, float* g_beta, int32_t N, int32_t C,
                     int64_t HW, int64_t g_batch_stride, int64_t x_batch_stride, int64_t gx_batch_stride,
                     int accumulate_gx, int unscaled, void* stream);

/* out[c] (=,+=) scale[c] * sum_{n,i} a[n,c,i] * ((b ? b[n,c,i] : 1) - mean[c])  (b, mean, scale optional).
 * Bias / batch-norm parameter gradients; with N = 1, C = batch it is the per-example dot product over C*H*W of
 * the gradient penalty (reference srgan.py:371,381); with HW = 1 the batch sum behind feature means
 * (srgan.py:442-443); with N = C = 1 a full sum. */
int srgan_chan_reduce(const float* a, const float* b, const float* mean, const float* scale, float* out, int32_t N,
                      int32_t C, int64_t HW, int accumulate, void* stream);
/* out[b] = max_f x[b,f] (logsumexp stabiliser, reference utility.py:179) and the one-hot of the nearest bin
 * (reference utility.py:141-144). */
int srgan_row_max(const float* x, float* out, int32_t B, int32_t F, void* stream);
int srgan_nearest_bin_onehot(const float* y, const float* bins, float* onehot, int32_t B, int32_t K, void* stream);

/* dst[n, dst_first + c, :] (=,+=) src[n, src_first + c, :], c < count: channel concat / slice
 * (reference crowd/models.py:353,1159-1165). */
int srgan_copy_channels(const float* src, int32_t src_channels, int32_t src_first, float* dst, int32_t dst_channels,
                        int32_t dst_first, int32_t count, int32_t N, int64_t HW, int accumulate, void* stream);

/* ---- pooling (planes = N*C) --------------------------------------------------------------------------------
 * reference crowd/models.py:371,1075,1151; age/vgg.py:76. */
/* The DenseNet stem's norm0 -> relu0 -> pool0 (reference crowd/models.py:1072-1076) as ONE pass each way: forward
 * y = maxpool(relu(batch_norm_eval(x))) with the arg-max of srgan_maxpool2d_fwd; backward gx = S * [bn(x) > 0] * inv_std *
 * gamma with S the pooled gradient gathered per input pixel, g_gamma / g_beta (both or neither) ADDED to.  The backward
 * exists for 3 / 2 / 1 windows on rows of whole float4s (`_supported`). */
int srgan_bn_relu_maxpool_fwd(const float* x, const float* mean, const float* inv_std, const float* gamma, const float* beta,
                              float* y, int32_t* argmax, int32_t N, int32_t C, int32_t H, int32_t W, int32_t k, int32_t s, int32_t p,
                              int32_t OH, int32_t OW, void* stream);
int srgan_bn_relu_maxpool_bwd_supported(int32_t N, int32_t C, int32_t H, int32_t W, int32_t k, int32_t s, int32_t p);
int srgan_bn_relu_maxpool_bwd(const float* gy, const int32_t* argmax, const float* x, const float* mean, const float* inv_std,
                              const float* gamma, const float* beta, float* gx, float* g_gamma, float* g_beta, int32_t N,
                              int32_t C, int32_t H, int32_t W, int32_t k, int32_t s, int32_t p, int32_t OH, int32_t OW,
                              void* stream);
/* avg_pool2d(relu(batch_norm_eval(x)), 2, 2) as ONE pass each way, for the DenseNet transitions evaluated as norm -> relu ->
 * pool -> conv (reference crowd/models.py:364-371 has conv -> pool; the 1x1 convolution and the average pooling commute, so
 * the convolution and its gradients run on a quarter of the pixels).  Forward: y[N, C, H/2, W/2]; backward: gx = 0.25 *
 * gy[h/2, w/2] * [bn(x) > 0] * inv_std * gamma, g_gamma / g_beta (both or neither) ADDED to.  Even H, W % 4 == 0
 * (`_supported`). */
int srgan_bn_relu_avgpool2_supported(int32_t N, int32_t C, int32_t H, int32_t W);
int srgan_bn_relu_avgpool2_fwd(const float* x, const float* mean, const float* inv_std, const float* gamma, const float* beta,
                               float* y, int32_t N, int32_t C, int32_t H, int32_t W, void* stream);
int srgan_bn_relu_avgpool2_bwd(const float* gy, const float* x, const float* mean, const float* inv_std, const float* gamma,
                               const float* beta, float* gx, float* g_gamma, float* g_beta, int32_t N, int32_t C, int32_t H,
                               int32_t W, void* stream);
int srgan_maxpool2d_fwd(const float* x, float* y, int32_t* argmax, int32_t planes, int32_t H, int32_t W, int32_t k,
                        int32_t s, int32_t p, int32_t OH, int32_t OW, void* stream);
int srgan_maxpool2d_bwd(const float* g, const int32_t* argmax, float* gx, int32_t planes, int32_t H, int32_t W, int32_t k,
                        int32_t s, int32_t p, int32_t OH, int32_t OW, void* stream);   /* max-pool backward, gather form */
int srgan_pool_scatter(const float* g, const int32_t* argmax, float* out, int32_t planes, int64_t in_plane,
                       int64_t out_plane, void* stream);      /* out[argmax] += g (out is zero-filled first): backward of
                                                                 srgan_pool_gather */
int srgan_pool_gather(const float* src, const int32_t* argmax, float* out, int32_t planes, int64_t in_plane,
                      int64_t out_plane, void* stream);       /* max-pool double-backward */
int srgan_avgpool2d_fwd(const float* x, float* y, int32_t planes, int32_t H, int32_t W, int32_t k, int32_t s,
                        int32_t OH, int32_t OW, void* stream);
int srgan_avgpool2d_bwd(const float* g, float* gx, int32_t planes, int32_t H, int32_t W, int32_t k, int32_t s,
                        int32_t OH, int32_t OW, void* stream);

/* ---- fused pieces of the step ------------------------------------------------------------------------------
 * out = alpha[b]*u + (1 - alpha[b])*fake (reference srgan.py:365-366). */
int srgan_gp_interpolate(const float* unlabeled, const float* fake, const float* alpha, float* out, int32_t B,
                         int64_t F, void* stream);
/* rows[b] = sum_{hw} mean_c |maps[b,c,hw] - target[b,hw]| and its backward (reference crowd/srgan.py:252). */
int srgan_crowd_map_l1_fwd(const float* maps, const float* target, float* rows, int32_t B, int32_t Cm, int64_t HW,
                           void* stream);
int srgan_crowd_map_l1_bwd(const float* maps, const float* target, const float* g_rows, float* g_maps, int32_t B,
                           int32_t Cm, int64_t HW, void* stream);
/* Training-batch assembly of the crowd application on the device (replaces the reference's 4-worker NumPy patch
 * extractor, crowd/shanghai_tech_data.py:76-104 with crowd/data.py:41-128,370-453): example b = the P x P patch centred
 * on (ys[b], xs[b]) of scene b (images_u8[b]: uint8 RGB [H, W, 3]; labels[b] / maps[b]: float [H, W]; all device
 * pointers, the pointer tables and the int32 arrays live on the device too), mirrored left-right when flips[b] != 0,
 * image normalised to [-1, 1] and planar: out_images [B, 3, P, P], out_labels / out_maps [B, P, P] (labels / maps and
 * their outputs may be NULL).  Pixels outside the scene are 0 before normalisation (-1 after), 0 in label / map. */
int srgan_crowd_extract_patches(const void* const* images_u8, const float* const* labels, const float* const* maps,
                                const int32_t* heights, const int32_t* widths, const int32_t* ys, const int32_t* xs,
                                const int32_t* flips, int32_t B, int32_t P, float* out_images, float* out_labels,
                                float* out_maps, void* stream);

/* Offline ikNN label of a crowd scene (reference crowd/database_preprocessor.py:93-101,266-290: generate_knn_map +
 * 1 / (map + epsilon)): out[y, x] = 1 / (mean over the k nearest heads of the Euclidean distance from (y, x), each
 * clipped at upper_bound when upper_bound > 0, + epsilon).  heads_yx: M (y, x) pairs, float; k <= 8 and k = min(k, M) as in the reference. */
int srgan_crowd_iknn_map(const float* heads_yx, int32_t M, int32_t H, int32_t W, int32_t k, float epsilon, float upper_bound,
                         float* out, void* stream);
/* Gaussian density label before its final rescaling (reference generate_density_label + make_gaussian,
 * crowd/database_preprocessor.py:110-249), every variant: out[y, x] = sum over the heads' (and bodies') windowed Gaussians,
 * each normalised by body_parts x its unclipped window sum.
 *   perspective == NULL: sigma_h = beta * mean distance of head h to its <= 11 nearest heads including itself (the
 *     "density{beta}" labels of :82-91);
 *   perspective = device map [H][W]: sigma_h = 0.2 m * perspective[y_h, x_h]; flag 2 (ignore_tiny) drops heads with a
 *     perspective < 3.1 (they do not count); flag 1 (include_body) adds a second Gaussian 0.875 m below the head with sigma
 *     (0.2 m, 0.5 m) * perspective and halves both normalisers;
 *   flag 4: perspective_resizing = False (sigma = 8 pixels); flag 8: positions are (x, y) pairs (yx_order = False).
 * Window half-sizes int(2 sigma).  workspace: 64 * M bytes of device memory; after the call float 7 of record 2 * h is 1
 * when head h counts.  The caller multiplies by counted heads / sum(out) (force_full_image_count_normalize). */
int srgan_crowd_density_label(const float* heads, int32_t M, int32_t H, int32_t W, float beta, const float* perspective,
                              int32_t flags, void* workspace, float* out, void* stream);

/* Adam on a flat arena, torch.optim.Adam defaults and operation order (reference srgan.py:131-138,266,297,305);
 * `step` is the 1-based update count. */
int srgan_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int32_t step, void* stream);
/* The same update with the count kept on the device: `state` = 3 x 4 bytes {int32 updates so far, float, float} owned
 * by the caller (initialise the first word; the other two are scratch for the bias corrections).  Every call advances
 * the count by one ON THE STREAM, so a HIP graph that captured the call replays as the next update each time. */
int srgan_adam_step_counted(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int32_t* state, void* stream);

/* Gradient buckets on the wire in bf16 (data-parallel exchange of the comm-sensitive configurations; the
 * reference has no collectives: its gradients complete at srgan.py:264,295,304 and stay on one device): fp32 -> bf16
 * (round to nearest even) and back; both buffers 16-byte aligned.  The master gradients stay fp32. */
int srgan_pack_bf16(const float* src, uint16_t* dst, int64_t n, void* stream);
int srgan_unpack_bf16(const uint16_t* src, float* dst, int64_t n, void* stream);

/* ---- collectives (RCCL over xGMI) -------------------------------------------------------------------------------
 * The reference has no collectives (one device: its batch means are `features.mean(0)`, srgan.py:442-443, and its
 * gradients complete on that device at srgan.py:264,295,304); the data-parallel path needs the SUM of the
 * ranks' feature sums in the forward pass (F floats: 80 for crowd) and the sum of the flat gradient arenas.  These are thin
 * wrappers: RCCL is resolved with dlopen("librccl.so.1") at first use -- the copy already mapped by the process if there
 * is one -- so the library has no link-time dependency on it.  A communicator belongs to the device that was current
 * when it was created; collectives are asynchronous on `stream`, in issue order; `dtype`: 0 = fp32, 1 = bf16 (the wire
 * format of srgan_pack_bf16).  A failing RCCL call returns 10000 + ncclResult_t with RCCL's message in srgan_last_error();
 * -2 when RCCL cannot be loaded. */
int srgan_comm_available(void);                               /* 1 when librccl could be resolved, else 0 (no error) */
int srgan_comm_unique_id(void* id128);                        /* ncclGetUniqueId: 128 bytes, made on rank 0 and handed to every rank by the caller */
int srgan_comm_init(void** comm, int32_t world_size, int32_t rank, const void* id128);   /* ncclCommInitRank on the current device */
int srgan_comm_world_size(void* comm, int32_t* world_size);
int srgan_comm_destroy(void* comm);
/* recv[i] = sum over ranks of send[i], i < count (send == recv: in place): gradient buckets and the tiny feature sums */
int srgan_all_reduce_sum(void* comm, const void* send, void* recv, int64_t count, int32_t dtype, void* stream);
/* the same sum as two halves over all links at once (point-to-point xGMI): rank r receives elements [r, r + 1) * recv_count
 * of the sum of send[0 .. world * recv_count); all_gather puts rank r's send_count elements at [r, r + 1) * send_count */
int srgan_reduce_scatter_sum(void* comm, const void* send, void* recv, int64_t recv_count, int32_t dtype, void* stream);
int srgan_all_gather(void* comm, const void* send, void* recv, int64_t send_count, int32_t dtype, void* stream);
/* ABI 1.1: buffer[0 .. count) of rank `root` to every rank, in place (ncclBroadcast): the initial weights, so that a caller whose
 * control plane is a host channel (TCP store, gloo) needs no second device transport */
int srgan_broadcast(void* comm, void* buffer, int64_t count, int32_t dtype, int32_t root, void* stream);

/* ---- 16-bit data path (ABI 1.1; BASELINE.json configs[1] "bf16" and configs[4] "fp16") ---------------------------------
 * "Blocked" tensors: logical [N, C, H, W] stored as [N][ceil(C / 8)][H][W][8] elements of bf16 (`dtype` 1) or fp16 (2); the 8
 * channels of a group at one pixel are one 16-byte slot = one MFMA operand fragment (sr-gan_amd/csrc/blocked16.h); channels
 * beyond C inside the last group hold zeros; an [N, F] matrix is the H = W = 1 case (row-major, row pitch ceil(F / 8) slots).
 * Activations, their gradients and a per-layer shadow of the weights live in this form; master weights, weight / bias
 * gradients (accumulated into), Adam and the losses stay fp32.  The reference has no analogue (it is fp32-only);
 * the call sites served are VGG-16's `conv3x3 -> ReLU`, `MaxPool2d(2, 2)` and `Linear -> ReLU` (age/vgg.py:33-41,70-84) and
 * the DCGAN stacks' `leaky_relu(conv)` (age/models.py:48-50,70-73).
 * epi (epilogue of a contraction): 0 = store; 1 = + bias (may be NULL), then leaky_relu with `slope` (0 = relu, 1 = identity);
 * 2 = multiply by the derivative of the activation that produced `ref` (1 where ref > 0, `slope` elsewhere; ref has the
 * output's shape): the gradient with respect to the PRE-activation tensor leaves the kernel, the un-masked one never exists. */
int srgan_h_pack(const float* x_nchw, void* out, const void* mask_ref, float slope, int32_t N, int32_t C, int64_t HW, int dtype,
                 void* stream);                                /* fp32 NCHW -> blocked (times mask(ref) when mask_ref != NULL) */
int srgan_h_unpack(const void* x, float* out_nchw, int32_t N, int32_t C, int64_t HW, int dtype, void* stream);
int srgan_h_add(const void* a, const void* b, void* out, int64_t slots, int dtype, void* stream);
int srgan_h_channel_sums(const void* x, float* out, int32_t N, int32_t C, int64_t HW, int dtype, void* stream);   /* out[c] += ... (bias gradient) */
/* mode 0: out = max_pool2d(x, 2, 2); mode 2: out = g (shape of x) gathered at the arg-max of x.  planes = N * groups. */
int srgan_h_maxpool2(const void* x, const void* g, void* out, int64_t planes, int32_t H, int32_t W, int mode, int dtype, void* stream);
/* gx (shape of x) = gp placed at the arg-max of x (first maximum in scan order, torch's rule), times mask(x, slope) if masked */
int srgan_h_maxpool2_bwd(const void* x, const void* gp, void* gx, int64_t planes, int32_t H, int32_t W, int masked, float slope,
                         int dtype, void* stream);
/* The 16-bit shadow of conv2d weights w[K][C][R][S] in operand slots; transposed = 1: the data-gradient operand (rows = C,
 * reduced over K, taps mirrored).  srgan_h_conv_weight_slots(rows, reduced, R, S) 16-byte slots. */
int64_t srgan_h_conv_weight_slots(int32_t rows, int32_t reduced, int32_t R, int32_t S);
int srgan_h_pack_conv_weights(const float* w, void* packed, int32_t K, int32_t C, int32_t R, int32_t S, int transposed, int dtype,
                              void* stream);
/* out[N, C_out, H, W] = epi(conv2d 3x3 / stride 1 / pad 1 of x[N, C_in, H, W]) with a packed operand of `rows` rows. */
int srgan_h_conv3x3(const void* x, const void* packed, const float* bias, const void* ref, float slope, int epi, void* out,
                    int32_t N, int32_t C_in, int32_t C_out, int32_t rows, int32_t H, int32_t W, int dtype, void* stream);
/* gw[C_out][C_in][3][3] (fp32) += weight gradient from x[N, C_in, H, W] and gy[N, C_out, H, W] (fixed summation order) */
int srgan_h_conv3x3_wgrad(const void* x, const void* gy, float* gw, int32_t N, int32_t C_in, int32_t C_out, int32_t H, int32_t W,
                          int dtype, void* stream);
/* out16[rows][cols] = src[map(r, row_plane) * row_stride + map(c, col_plane) * col_stride] (0 outside rows_real x cols_real);
 * map(i, P) = NCHW-flattened index of element i of a flattened blocked tensor with P pixels per channel (P <= 1: i). */
int srgan_h_pack_matrix(const float* src, void* out, int64_t rows, int64_t cols, int64_t rows_real, int64_t cols_real,
                        int64_t row_stride, int64_t col_stride, int32_t row_plane, int32_t col_plane, int dtype, void* stream);
/* out[N][out_cols] = epi(B[N][K] * A[M][K]^T): torch.nn.functional.linear (age/vgg.py:33-41) and its data gradient */
int srgan_h_gemm(const void* a, const void* b, const float* bias, const void* ref, float slope, int epi, void* out, int32_t M,
                 int32_t N, int32_t K, int32_t out_cols, int32_t bias_entries, int dtype, void* stream);
/* gw element (m, k) at gw[map(m, row_plane) * ldw_m + map(k, col_plane) * ldw_k] += sum_n S[n][m] * X[n][k] */
int srgan_h_linear_wgrad(const void* s, const void* x, float* gw, int32_t N, int32_t M, int32_t K, int64_t M_real, int64_t K_real,
                         int64_t ldw_m, int64_t ldw_k, int32_t row_plane, int32_t col_plane, int dtype, void* stream);

/* 4x4 / stride 2 / pad 1 pair (the DCGAN stacks, reference age/models.py:37-51,61-73).  A weight tensor [A][B][4][4] in torch's
 * layout -- conv2d weights [K][C][4][4]: A = K, B = C; conv_transpose2d weights [Cin][Cout][4][4]: A = Cin, B = Cout -- always
 * has A = the channels of the tensor on the SMALL (H/2 x W/2) plane.  direction 0 "down" (big -> small plane, rows = A): conv2d's
 * forward, conv_transpose2d's data gradient; direction 1 "up" (small -> big plane, rows = B, the four output parity classes):
 * conv_transpose2d's forward, conv2d's data gradient.  epi as srgan_h_conv3x3.  This family (and srgan_h_pack / _unpack / _add /
 * _channel_sums) also takes `dtype` 0: fp32 tensors in the same blocked layout with FOUR channels per 16-byte slot
 * ([N][ceil(C / 4)][H][W][4]), v_mfma_f32_32x32x2_f32 -- the exact form (the fp32 gradient-penalty chain of configs[4], the
 * crowd generator, reference crowd/models.py:132-146). */
int64_t srgan_h_k4s2_weight_slots(int32_t A, int32_t B, int direction, int dtype);
int srgan_h_pack_k4s2_weights(const float* w, void* packed, int32_t A, int32_t B, int direction, int dtype, void* stream);
/* The batched form of the two weight packers above: every convolution shadow of a network re-rounded by ONE launch behind its
 * optimizer update.  The caller keeps a device array of jobs, srgan_h_pack_job_bytes() each; a job function takes the arguments
 * of its single-layer call, writes the job(s) at `jobs` (HOST memory; the "up" direction of the 4x4 family writes four) with
 * workgroups from `first_block` on and returns the number of workgroups they take (< 0: error); srgan_h_pack_batched launches
 * `count` jobs of `blocks` workgroups in total from the device copy of the array. */
int32_t srgan_h_pack_job_bytes(void);
int64_t srgan_h_pack_job_conv_weights(void* jobs, int64_t first_block, const float* w, void* packed, int32_t K, int32_t C, int32_t R,
                                      int32_t S, int transposed, int dtype);
int64_t srgan_h_pack_job_k4s2_weights(void* jobs, int64_t first_block, const float* w, void* packed, int32_t A, int32_t B,
                                      int direction, int dtype, int32_t* jobs_written);
int srgan_h_pack_batched(const void* jobs_device, int32_t count, int64_t blocks, void* stream);
int srgan_h_conv4x4s2(const void* x, const void* packed, const float* bias, const void* ref, float slope, int epi, void* out,
                      int32_t N, int32_t C_in, int32_t rows, int32_t H, int32_t W, int dtype, void* stream);
int srgan_h_conv_transpose4x4s2(const void* x, const void* packed, const float* bias, const void* ref, float slope, int epi,
                                void* out, int32_t N, int32_t C_in, int32_t rows, int32_t h, int32_t w, int dtype, void* stream);
/* gw (fp32 [A][B][4][4]) += weight gradient from `small` [N, A, H/2, W/2] and `big` [N, B, H, W] (small_is_rows must be 1) */
int srgan_h_k4s2_wgrad(const void* big, const void* small, float* gw, int32_t N, int32_t C_big, int32_t C_small, int32_t H, int32_t W,
                       int small_is_rows, int dtype, void* stream);

/* ---- measurement ---------------------------------------------------------------------------------------------
 * Between begin and end every contraction launch (conv / gemm passes) is bracketed by a pair of HIP events on
 * its launch stream; end() synchronises and returns the summed kernel time, the summed logical 2*M*N*K, the
 * part of it executed by the MFMA kernel, and the launch count (bench.py's roofline leg). */
int srgan_profile_begin(void);
int srgan_profile_end(double* kernel_ms, double* flops, double* mfma_flops, int64_t* launches);
/* Algorithmic HBM bytes of the bracketed launches: every operand / result element once, 4 B each (a 3x3 tap or a
 * stride class does not count an input element twice): the figure bench.py's roofline.algorithmic_bytes_per_step
 * reports next to the PMC traffic. */
int srgan_profile_bytes(double* algorithmic_bytes_total);
/* The part of the bracketed launches that ran with bf16 / fp16 MFMA operands: logical FLOPs and summed kernel time (the
 * mixed-precision bench lines price it against the 2.5 PFLOP/s dense bf16 / fp16 peak, the rest against 157.3 TFLOP/s). */
int srgan_profile_mixed(double* flops, double* kernel_ms);
/* Per-shape text report of the last profiled region ("M N K kind bm bn split akf bkf count ms bytes" per line; kind:
 * 0 gg_direct, 1 gg_mfma, 2 conv3x3_lds, 3 pointwise, 4 conv3x3_wgrad, 5 gg_rows, 6 pointwise_wgrad,
 * 8 pointwise_ksplit, 9 gg_dot, 10 stem7x7_fwd, 11 stem7x7_wgrad); returns the bytes needed. */
int64_t srgan_profile_report(char* buffer, int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* SRGAN_HIP_H */
