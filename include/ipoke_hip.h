/*
 * libipoke_hip -- C ABI of the MI355X (gfx950) implementation of the iPOKE hot path.
 *
 * The reference (CompVis/ipoke) is pure Python on PyTorch and has no FFI of its own; every entry
 * point below replaces the PyTorch op sequence of one reference function (cited per group).  The
 * reference-side binding a maintainer would add is the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; all buffers are device memory owned by the caller;
 *   - every function returns 0 (IPOKE_OK) or a negative ipoke_status, never throws; the message of
 *     the last failure on the calling thread is available from ipoke_last_error();
 *   - `stream` is a hipStream_t passed as void*; nothing synchronises the host, so every call is
 *     capturable in a hipGraph;
 *   - `dtype` selects the arithmetic type of the matrix-core contractions:
 *         IPOKE_F32  exact-f32 MFMA (v_mfma_f32_16x16x4_f32), parity mode
 *         IPOKE_BF16 bf16 MFMA inputs, fp32 accumulate (v_mfma_f32_16x16x32_bf16), throughput mode
 *     Flow state, affine transforms, log-determinants, normalisation statistics, gradients of
 *     parameters and optimizer state are always fp32.
 *   - flow state tensors are "positions-major" [B, 64, ld] fp32 (NHWC of the 8x8 latent) inside the
 *     library; ipoke_nchw_to_state / ipoke_state_to_nchw convert at the boundary.
 */
#ifndef IPOKE_HIP_H
#define IPOKE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  IPOKE_OK = 0,
  IPOKE_ERR_INVALID = -1,   /* bad argument / unsupported shape */
  IPOKE_ERR_HIP = -2,       /* a HIP runtime call failed        */
  IPOKE_ERR_STATE = -3      /* call sequence error (e.g. backward without forward) */
} ipoke_status;

enum { IPOKE_F32 = 0, IPOKE_BF16 = 1 };
enum { IPOKE_ACT_NONE = 0, IPOKE_ACT_ELU = 1, IPOKE_ACT_RELU = 2, IPOKE_ACT_LRELU02 = 3, IPOKE_ACT_TANH = 4,
       IPOKE_ACT_SIGMOID = 5 };

const char* ipoke_last_error(void);
int ipoke_version(void);
/* size in bytes of one element of `dtype` */
int ipoke_dtype_size(int dtype);

/* ---------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on the matrix cores.
 * Replaces F.conv2d / nn.Conv2d / nn.Conv3d / nn.ConvTranspose2d call sites of the reference:
 *   models/modules/INN/macow_utils.py:270-281 (NICE nets), :492-499 (shifted conv),
 *   models/modules/motion_models/motion_encoder.py:161-241 (Conv3d ResNet),
 *   models/modules/motion_models/rnn.py:16-18 (ConvGRU gates),
 *   models/modules/autoencoders/util.py:52-55,252-255 (Conv2d / ConvTranspose2d blocks).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  /* geometry: rows of the GEMM are output positions (n, od, oh, ow); Do, Ho, Wo must be powers of two */
  int32_t NB, Di, Hi, Wi, Do, Ho, Wo;
  int32_t kd, kh, kw, sd, sh, sw, pd, ph, pw;
  int32_t transposed;      /* 0: in = out*s - p + k (conv);  1: in = (out + p - k)/s when divisible
                              (ConvTranspose forward, or data-gradient of a strided conv)          */
  /* A operand (activations) */
  const void* A;           /* a_f32 ? float : dtype */
  int32_t a_f32;           /* 1: A is fp32 and is converted on load (flow state, input images)     */
  int64_t a_sn, a_sd, a_sh, a_sw, a_sc;  /* element strides of n, d, h, w, channel                 */
  int32_t a_coff;          /* element offset of the first channel used; channel c is at a_coff + c*a_sc */
  int32_t Kc_real;         /* channels per tap actually present                                    */
  int32_t Kc;              /* channels per tap in the K index (>= Kc_real, multiple of 16B/elt)    */
  /* B operand (weights), dtype, [Nout][ldw] with k = tap*Kc + c, zero padded                      */
  const void* W;
  int32_t ldw;
  int32_t Nout;
  /* epilogue */
  const float* bias;       /* [Nout] or NULL                                                       */
  int32_t act;             /* IPOKE_ACT_* applied after bias                                        */
  const void* dact;        /* optional [M][ld_dact] dtype: multiply by act'(saved output) (dgrad)  */
  int32_t ld_dact, dact_act;
  void* C;                 /* output                                                                */
  int32_t c_f32;           /* 1: C is fp32, 0: dtype                                                */
  int32_t c_accumulate;    /* 1: C += result (fp32 only)                                            */
  int64_t ldc;             /* row stride of C in elements                                           */
  int32_t c_coff, c_cstride; /* column n is stored at c_coff + n*c_cstride                          */
  int32_t splitk;          /* >1: fp32 partials C[z][M][ldc], epilogue skipped (bias/act by consumer);
                              with c_accumulate: the K slices are added into C instead (deterministically
                              through acc_scratch below, with atomics without it)                       */
  /* optional output scatter (c_scatter = 1; dtype or fp32 outputs, splitk = 1, no dact): output position (n, od = 0, oh, ow) is
   * stored at row c_row0 + n*c_sn + oh*c_sh + ow*c_sw of C instead of row m -- the four sub-pixel phases of a stride-2
   * ConvTranspose2d (util.py:52-55: 3 x 3, padding 1, output_padding 1) run as four stride-1 convolutions with 1 / 2 / 2 / 4 taps
   * that write every second row and column of the up-sampled map, instead of one launch that multiplies 9 taps of which 2.25
   * fall on input pixels. */
  int32_t c_scatter;
  int64_t c_sn, c_sh, c_sw, c_row0;
  /* optional per-image-group scale of the accumulators, applied before the bias (splitk = 1): the outputs of image n are multiplied by
   * row_scale[(n / rs_images) * rs_stride].  First-stage training decodes the T - 1 generated frames of a batch of clips as ONE batch
   * ordered (frame, clip); torch's spectral_norm runs one power iteration per decoder call, i.e. frame t uses W / sigma_t (util.py:52,
   * 252) -- conv(x, W / sigma_t) = conv(x, W) / sigma_t, so one operand of W_orig serves every frame and 1 / sigma_t is this scale
   * (rs_images = clips per batch, row_scale = the {sigma, 1/sigma} table of ipoke_spectral_sigma_multi + 1, rs_stride = 2). */
  const float* row_scale;
  int32_t rs_images, rs_stride;
  /* 1: W is K-MAJOR -- the row-major [Ktot][ldw] matrix, element (k, n) at W[k * ldw + n] (bf16, 1 x 1 kernels over dense rows, Kc a
   * multiple of 64).  The data gradient of a 1 x 1 convolution then reads the weight's own straight copy W[out][in] (it reduces over
   * `out`, the row index), so that weight needs no transposed operand (the coupling nets' conv2, macow_utils.py:270-281). */
  int32_t w_kmajor;
  /* depth stride of the output scatter (c_scatter with Do > 1: position (n, od, oh, ow) goes to row c_row0 + n*c_sn + od*c_sd + oh*c_sh +
   * ow*c_sw): the data gradient of a stride-2 Conv3d (motion_encoder.py:33-36, 80-91 layer transitions) run as one stride-1 convolution
   * per output parity class -- (1 or 2) taps per strided dimension instead of all 3 of which half fall between the gradient's samples */
  int64_t c_sd;
  /* optional scratch of the DETERMINISTIC split-K accumulation (c_accumulate with splitk > 1): ipoke_conv_acc_scratch_bytes(M, Nout,
   * splitk) bytes, 16-byte aligned, prepared once with ipoke_conv_acc_scratch_init and left in that state by every launch; one
   * launch at a time.  The K slices park their tiles in the scratch, the workgroup that arrives last at a tile sums them in a fixed
   * order and adds the sum to C (no atomics: bit-reproducible).  NULL: the slices are added to C with fp32 atomics, whose rounding
   * depends on their arrival order. */
  void* acc_scratch;
  int64_t acc_scratch_bytes;
} ipoke_conv_desc;

int ipoke_conv_forward(const ipoke_conv_desc* d, int dtype, void* stream);
/* size / one-time preparation of ipoke_conv_desc.acc_scratch for accumulating launches of at most M output rows, Nout columns and
 * splitk K slices (the conv1 data gradient of NICEConvBlock, macow_utils.py:270, added into the gradient of the conditioning
 * channels; the reference runs with deterministic=True, experiments/experiment.py:33, 86) */
int64_t ipoke_conv_acc_scratch_bytes(int M, int Nout, int splitk);
int ipoke_conv_acc_scratch_init(void* scratch, void* stream);

/* Split count the library wants for the skinny 3x3 convolutions of the coupling nets (conv3 forward: split-K partial
 * slabs; conv1 data gradient: accumulation into the gradient state, ipoke_conv_desc.acc_scratch) at M = 64*B output rows and Kc input channels -- callers size their
 * partial-sum slabs with it and pass it as ipoke_conv_desc.splitk.  0: no preference (the stationary-input kernel does
 * not apply).  Reference call sites: conv3 / conv1 of NICEConvBlock (models/modules/INN/macow_utils.py:270-281,
 * 3x3, padding 1, on the 8x8 latent). */
int ipoke_conv3x3_skinny_splitk(int M, int Kc, int dtype);

/* Weight gradient: dW[n][tap*Kc + c] (+)= sum_m dY[m][n] * A[src(m,tap)][c].
 * dY is dtype [M][ldy]; A as in ipoke_conv_desc (fp32 or dtype).  Output fp32, written through a
 * (n, c, tap) stride triple so PyTorch's [out][in][k...] layout is produced directly. */
/* Optional Adam-amsgrad in the epilogue of a BATCHED weight-gradient launch (ipoke_wgrad_desc.adam): the gradient tile is consumed
 * in registers instead of being written to w_base -- parameters and moments of the tile are read, updated with exactly the arithmetic
 * of ipoke_adam_amsgrad_step (torch.optim.Adam(amsgrad=True, weight_decay), second_stage_video.py:648-650) and written back, and the
 * updated parameters' cast goes to `operand` (the bf16 matrix-core copy the next forward / data-gradient pass reads).  params, m, v,
 * vmax: bases of buffers in the layout of w_base (problem z uses element offset entries[z].w_off in all four), operand: base of
 * the operand copies (problem z at element offset entries[z].sh_off, same [out][in] order).  Dense 1 x 1 problems in whole
 * 128 x 128 tiles (Nout, Kc multiples of 128; w_sn = Kc, w_sc = 1), bf16, one reduction split.  -8 of the 42 bytes per parameter
 * that weight gradient + optimizer move (the gradient is neither written nor read back). */
typedef struct {
  float* params; float* m; float* v; float* vmax; void* operand;
  float lr, beta1, beta2, eps, weight_decay, grad_scale;
  int32_t step;                          /* 1-based optimizer step (bias corrections) */
  int32_t keep_grad;                     /* 1: the gradient is written to w_base as well (costs the 4 bytes the fusion saves) */
} ipoke_wgrad_adam;

typedef struct {
  int32_t NB, Di, Hi, Wi, Do, Ho, Wo, kd, kh, kw, sd, sh, sw, pd, ph, pw, transposed;
  const void* A; int32_t a_f32; int64_t a_sn, a_sd, a_sh, a_sw, a_sc; int32_t a_coff, Kc_real, Kc;
  const void* dY; int32_t ldy; int32_t y_coff; int32_t Nout;
  float* dW; int64_t w_sn, w_sc, w_st;   /* strides of out-channel n, in-channel c, tap */
  int32_t accumulate;                    /* 1: dW += */
  int32_t splitm;                        /* >1: reduction split over gridDim.y with fp32 atomics (dW must be pre-zeroed or accumulate) */
  int32_t Kc_store;                      /* channels per tap written to dW (0: Kc_real); < Kc_real when A carries zero padding */
  int64_t split_stride;                  /* with splitm > 1: != 0 -> slice z of the reduction is *stored* to dW + z*split_stride
                                            (deterministic; the caller sums the splitm slabs, e.g. ipoke_reduce_rows), 0 -> atomics */
  int32_t max_workgroups;                /* > 0: issue the output tiles in launches of at most this many workgroups, so that a
                                            weight gradient running on a side stream never holds every CU of the chip */
  const ipoke_wgrad_adam* adam;          /* batched launches only: see ipoke_wgrad_adam; NULL = plain weight gradient */
} ipoke_wgrad_desc;

int ipoke_conv_wgrad(const ipoke_wgrad_desc* d, int dtype, void* stream);
/* Reduction splits (slabs: splitm with split_stride) the kernel that ipoke_conv_wgrad dispatches THIS problem to prefers when the caller aims
 * at target_workgroups workgroups; 0: no preference (keep the caller's rule).  > 0 for stride-1 3x3 / 3x3x3 "same" convolutions on maps
 * whose height and width are multiples of 16 (BasicBlock of the 3-D encoder, motion_encoder.py:45-74; the SPADE decoder's 3x3 layers,
 * autoencoders/util.py:106-192): they run in a halo-staged kernel that tiles dW by 64 x 64 x 9 taps per depth tap. */
int ipoke_conv_wgrad_splitm(const ipoke_wgrad_desc* d, int dtype, int target_workgroups);
/* nbatch same-shape problems in one launch; entries_dev[i] = {int64 a_off bytes, int64 y_off bytes, int64 w_off floats,
 * int32 kh, kw, ph, pw, int64 sh_off elements (used with ipoke_wgrad_desc.adam only)} relative to a_base / y_base / w_base (the
 * descriptor's A, dY, dW and kh/kw/ph/pw are ignored) */
int ipoke_wgrad_batch_entry_size(void);
int ipoke_conv_wgrad_batched(const ipoke_wgrad_desc* d, const void* entries_dev, int nbatch, const void* a_base,
                             const void* y_base, float* w_base, int dtype, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Flow-state element-wise layers (fp32).  State = [B*P][ld], P = 64 positions of the 8x8 latent.
 * ------------------------------------------------------------------------------------------- */
int ipoke_nchw_to_state(const float* x_nchw, float* state, int B, int C, int P, int ld, void* stream);
int ipoke_state_to_nchw(const float* state, float* x_nchw, int B, int C, int P, int ld, void* stream);
/* act(cond) in dtype, channels-last: shared input of all MCF 1x1 convs (macow_utils.py:429-431) */
/* dense, zero-padded dtype copy of `C` state channels starting at `off` with stride `stride` (NICE split, macow2.py:364-375) */
int ipoke_extract_cols(const float* state, int ld, int off, int stride, int C, void* out, int ldo, int64_t M, int dtype,
                       void* stream);
int ipoke_cond_prepare(const float* cond_nchw, void* out, int B, int Cc, int P, int act, int dtype, void* stream);
/* dst[m][0:C] = src[m][0:C] for M rows of `dtype` elements with row pitches lds / ldd (elements); widths, pitches and bases multiples
 * of 16 bytes.  condition_nice: torch.cat([conv2 out, h]) of NICEConvBlock.forward (macow_utils.py:328-332) is this copy of the
 * activated conditioning map into the last h_channels columns of the hidden tile conv2 writes */
int ipoke_copy_cols(const void* src, int lds, void* dst, int ldd, int C, int64_t M, int dtype, void* stream);

/* ActNorm2dFlow (macow2.py:476-540) optionally fused with the Shuffle that follows it
 * (flow_blocks.py:314-326): out[:, c0+j] = in[:, c0+idx[j]] * exp(ls[idx[j]]) + bias[idx[j]].
 * log_scale == bias == NULL gives the bare permutation; idx == NULL the bare ActNorm.
 * The batch-independent log-det H*W*sum(log_scale) is accounted for by ipoke_actnorm_logdet. */
int ipoke_actnorm_fwd(const float* in, float* out, int M, int ld, int c0, int C, const float* log_scale,
                      const float* bias, const int32_t* idx, void* stream);
/* inverse: undo the permutation with inv_idx (= backward_shuffle_idx), then (x - bias)/(exp(ls)+1e-8) */
int ipoke_actnorm_inv(const float* in, float* out, int M, int ld, int c0, int C, const float* log_scale,
                      const float* bias, const int32_t* inv_idx, void* stream);
/* The same with a second output for the sampling direction: the channels e_off + j*e_stride (j < e_C) of `out` once more as a dense,
 * zero-padded [M][ext_ld] operand of `dtype` -- the conditioning input of the coupling that is inverted next (NICE2d.backward,
 * macow2.py:449-470, reads its net's input from the state this ActNorm inverse produces), which then needs no ipoke_extract_cols
 * launch.  ext == NULL: ipoke_actnorm_inv.  ext_ld - e_C <= ld. */
int ipoke_actnorm_inv_ext(const float* in, float* out, int M, int ld, int c0, int C, const float* log_scale,
                          const float* bias, const int32_t* inv_idx, void* ext, int ext_ld, int e_off, int e_stride, int e_C,
                          int dtype, void* stream);
/* per-sample partial sums part[b] = [d_log_scale(C) | d_bias(C)]; reduce over b with ipoke_reduce_rows */
int ipoke_actnorm_bwd(const float* dy, const float* x, float* dx, int M, int ld, int c0, int C,
                      const float* log_scale, const int32_t* idx, const float* dld, int B, int P,
                      float* part, void* stream);
/* data-dependent init (macow2.py:526-539): overwrites log_scale / bias in place */
int ipoke_actnorm_init(const float* x, int M, int ld, int c0, int C, float* log_scale, float* bias, void* stream);

/* Affine coupling transform (macow_utils.py:42-66) on the channels t_off + i*t_stride, i < Cp.
 * raw = [mu | s] is given as nsplit fp32 partial sums (the split-K output of the coupling net's
 * last conv) plus an optional bias; scale = tanh(s/2) + 1. */
typedef struct {
  const float* raw; int32_t nsplit; int64_t split_stride; int32_t ldraw;
  const float* bias;
  int32_t Cp, t_off, t_stride;
  int32_t P, ld;
} ipoke_affine_desc;
int ipoke_affine_fwd(const ipoke_affine_desc* d, const float* in, float* out, float* scale_out /* [M][Cp] or NULL */,
                     float* logdet_slot /* [B*slot_stride] or NULL */, int slot_stride, int B, void* stream);
int ipoke_affine_inv(const ipoke_affine_desc* d, const float* in, float* out, int B, void* stream);
/* The same with one more output: the channels the coupling just wrote, again as a dense zero-padded [M][ext_ld] operand of `dtype` --
 * the conditioning input of the next coupling's first convolution when that coupling conditions on exactly these channels
 * (coupling*_up followed by coupling*_dn, macow2.py:364-375, 397-448), which then needs no extract_cols launch.  ext may be NULL. */
int ipoke_affine_fwd_ext(const ipoke_affine_desc* d, const float* in, float* out, float* scale_out, float* logdet_slot, int slot_stride,
                         int B, void* ext, int ext_ld, int dtype, void* stream);
int ipoke_affine_inv_ext(const ipoke_affine_desc* d, const float* in, float* out, int B, void* ext, int ext_ld, int dtype, void* stream);
int ipoke_affine_bwd(int Cp, int t_off, int t_stride, int P, int ld, const float* dy, const float* x,
                     const float* scale, const float* dld, float* dx, void* dparams /* dtype [M][ldp] */, int ldp,
                     float* dbias_part /* [B][2Cp] or NULL */, int B, int dtype, void* stream);
/* A coupling followed by an ActNorm2dFlow (+ Shuffle) -- MaCowStep's coupling1_dn -> actnorm2, coupling2_dn -> the next step's actnorm1,
 * MultiScalePrior's coupling -> actnorm (macow2.py:1066-1117, 569-593) -- as ONE launch per direction.  Forward: ipoke_affine_fwd followed by
 * ipoke_actnorm_fwd on its output; `out` (the coupling's output, the ActNorm's saved input; NULL when nothing is saved) and `out2` are both
 * written.  Backward: ipoke_actnorm_bwd (dy2, x1 = out, part) followed by ipoke_affine_bwd on the gradient it passes on. */
int ipoke_affine_actnorm_fwd(const ipoke_affine_desc* d, const float* in, float* out, float* out2, float* scale_out, float* logdet_slot,
                             int slot_stride, int B, int c0, int C, const float* log_scale, const float* bias, const int32_t* idx, void* stream);
int ipoke_actnorm_affine_bwd(int c0, int C, const float* log_scale, const int32_t* idx, const float* dy2, const float* x1, float* part,
                             int Cp, int t_off, int t_stride, int P, int ld, const float* x0, const float* scale, const float* dld, float* dx,
                             void* dparams, int ldp, float* dbias_part, int B, int dtype, void* stream);
/* conv3 of a coupling net AND the coupling transform it feeds in one launch (NICEConvBlock's last convolution, macow_utils.py:270-281,
 * followed by the affine transform, macow_utils.py:42-66, optionally with the ActNorm2dFlow (+ Shuffle) behind it): the K splits of a
 * 128-row tile exchange their partial sums inside the launch (mode / results exactly those of ipoke_conv_forward with
 * splitk = ipoke_conv3x3_coupling_splitk(...) followed by ipoke_affine_fwd_ext / ipoke_affine_actnorm_fwd / ipoke_affine_inv_ext:
 * bit-identical states, scales and log-det slots; no partial-sum slabs are written).
 *   conv: the 3x3 / pad 1 convolution on 8x8 maps as for ipoke_conv_forward (bf16, dense input, Nout = 2*Cp <= 64, no bias /
 *         activation; C and splitk are ignored);  aff: bias, Cp, t_off, t_stride, P = 64, ld (raw / nsplit / strides ignored);
 *   mode 0: out = coupling(in) (+ ext, scale_out, logdet_slot with slot_stride >= 4)   [ipoke_affine_fwd_ext]
 *   mode 1: out (may be NULL) as mode 0, out2 = ActNorm(+ shuffle)(out)                 [ipoke_affine_actnorm_fwd]
 *   mode 2: out = coupling^-1(in) (+ ext)                                               [ipoke_affine_inv_ext]
 * xchg: ipoke_conv3x3_coupling_xchg_bytes() bytes of device scratch, initialised ONCE by ipoke_conv3x3_coupling_xchg_init and left in
 * that state by every launch; one launch at a time per scratch.  Word 0 counts hand-offs that timed out (0 on a healthy device). */
typedef struct {
  int32_t mode;
  const float* in; float* out; float* out2; float* scale_out; float* logdet_slot; int32_t slot_stride;
  int32_t an_c0, an_C; const float* an_log_scale; const float* an_bias; const int32_t* an_idx;
  void* ext; int32_t ext_ld;
  void* xchg;
} ipoke_coupling_epi;
/* K splits of the fused launch at M = 64*B rows and Kc input channels: 4, 8, 16 or 32; 0 = the fused kernel does not apply */
int ipoke_conv3x3_coupling_splitk(int M, int Kc, int dtype);
int64_t ipoke_conv3x3_coupling_xchg_bytes(void);
int ipoke_conv3x3_coupling_xchg_init(void* xchg, void* stream);
int ipoke_conv3x3_coupling(const ipoke_conv_desc* conv, const ipoke_affine_desc* aff, const ipoke_coupling_epi* epi, int B, int dtype,
                           void* stream);
int ipoke_reduce_rows(const float* src, float* dst, int R, int ncols, void* stream);
/* multi-tensor form: entries_dev[i] = {int64 src, int64 dst (float offsets), int32 ld, int32 ncols, int32 rmul, int32 pad};
 * entry i sums R * max(rmul, 1) rows */
int ipoke_reduce_entry_size(void);
int ipoke_reduce_rows_multi(const float* src, float* dst, const void* entries_dev, int nentries, int R, void* stream);

int ipoke_logdet_finalize(const float* slots, int nslots, int B, int slot_w, float const_term, const float* const_dev,
                          float* logdet, void* stream);
int ipoke_actnorm_logdet(const float* params, const void* refs_dev, int n, int P, float* out_scalar, void* stream);
/* FlowLoss (loss.py:6-31,75-79): scalars3 = (loss, nll, nlogdet); optional gradients in state layout */
int ipoke_flow_nll(const float* z_state, const float* logdet, int B, int P, int C, int ld, float logdet_weight,
                   float* scalars3, float* d_out_state, float* dld, void* stream);
/* torch.optim.Adam(amsgrad=True) step over a flat fp32 buffer (second_stage_video.py:648-650) */
int ipoke_adam_amsgrad_step(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
/* the same with a cap on the persistent grid (0: default), for an update issued underneath other work */
int ipoke_adam_amsgrad_step_grid(float* p, const float* g, float* m, float* v, float* vmax, int64_t n, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int step, float grad_scale, int max_blocks,
                                 void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused masked convolutional flow (macow2.py:25-288, macow_utils.py:407-499).
 * Weight operands are the shadows produced by ipoke_flow_prepare_weights (dims: ipoke_mcf_shadow_dims): matrices
 * [rows][K] (W1: k = tap*Cp + c; W2: k over [hidden | cond]; W1T / W2T: their transposes), zero padded, stored
 * FRAGMENT-TILED: tiles of 16 rows x 64 bytes of K, tile (rb, ks) at ((rb * K/Kt + ks) * 1024) bytes (Kt = 64 / element size),
 * inside a tile the 16-byte chunk q of row r at (16 q + r) * 16 bytes -- one wave-wide fragment load is one contiguous KB.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const float* x; float* y;       /* state in / out (distinct buffers), fwd: x->y, inv: y_in -> x_out */
  int32_t ld, C, B;
  const void* cond; int32_t Cc;   /* act(cond), dtype [B*64][Cc] */
  const void* W1; const void* W2; const float* bias2;
  int32_t order;                  /* 0..3 = A..D */
  int32_t rows_per_block;         /* fwd only: 16/32/64, 0 = default */
  void* a2_save; float* scale_save; float* logdet_slot;   /* fwd saves (may be NULL) */
  /* backward */
  const void* W2T; const void* W1T;
  const float* dy; const float* dld; float* dx;
  void* dparams_save; void* dc_save; float* dbias_part;
  /* optional fused ActNorm2dFlow on the same C channels right after the coupling (MaCowUnit: MCF -> MCF -> ActNorm,
   * macow2.py:957-995): fwd writes y = (scale*x + mu)*exp(post_log_scale) + post_bias; bwd takes dy with respect to that
   * output, needs it saved (y_post) and writes the ActNorm's per-sample parameter-gradient partials
   * post_part[b] = [d_log_scale(C) | d_bias(C)] in the layout of ipoke_actnorm_bwd.  Requires C % 4 == 0, ld % 4 == 0. */
  const float* post_log_scale; const float* post_bias;
  const float* y_post; float* post_part;
  /* ipoke_macow_unit_bwd only, optional: dtype [B*64][round_up(C, 32)] copy of this layer's input state (columns >= C zero) --
   * the A operand of the shifted-conv weight gradient in the matrix cores' own dtype, so that gradient runs on the LDS-DMA GEMM */
  void* x_op_save;
  /* ipoke_macow_unit_fwd only, optional (layer 3 of the unit): the conditioning operand of the NICE coupling that follows the unit
   * (NICE2d.forward, macow2.py:397-448: the net reads the channels zc_off + k * zc_stride, k < zc_cin, of the unit's output) as a
   * dtype [B*64][zc_ld] matrix, columns zc_cin .. zc_ld - 1 zero -- what ipoke_extract_cols would produce in a launch of its own */
  void* zc_out; int32_t zc_off, zc_stride, zc_cin, zc_ld;
  /* ipoke_macow_unit_fwd / _bwd only, read from d4[0]: split = 2 or 4 runs a sample's 8x8 latent on that many workgroups (grid rows
   * dealt in order; the workgroups exchange the 1-2 halo rows a masked convolution reads across the cut inside the launch).
   * xchg = scratch of ipoke_macow_unit_xchg_bytes(B, split) bytes, zero-initialised ONCE by the caller and used by one launch at a
   * time (launches on one stream); the kernels leave it all-zero.  Its first 32-bit word counts hand-off time-outs (0 = healthy).
   * Outputs as with split = 0 / 1 (bit-identical states, saves and data gradients) except: logdet_slot[b * w + s] holds part s of
   * the layer's log-det (w >= split), and dbias_part / post_part have split rows per sample (row b * split + s). */
  int32_t split; void* xchg;
  /* ipoke_macow_unit_bwd with split >= 2 only, read from d4[0], optional: the ActNorm2dFlow (+ Shuffle) and the NICE coupling IN FRONT of
   * the unit (forward order coupling -> ActNorm -> unit: MaCowStep, macow2.py:1066-1117) differentiated by the same launch -- see
   * ipoke_unit_pair_desc below.  The launch then writes pair->dx instead of d4[0].dx. */
  const struct ipoke_unit_pair_desc* pair;
} ipoke_mcf_desc;
/* The work of ipoke_actnorm_affine_bwd on the gradient a unit's backward launch produces, done by that launch row slice by row slice
 * (both layers are row-wise maps): an ActNorm on the unit's channels [0, C) (log_scale NULL: pure shuffle; an_x = its saved input, the
 * coupling's output; an_part [B * split][2C] partial sums, row b * split + s) and the affine coupling in front of it (x0 = its saved
 * input, scale = its saved scales [M][Cp]; outputs dx [M][ld], dparams dtype [M][ldp] = [d mu | d s | 0 pad], dbias_part
 * [B * split][2 Cp]; dx must be d4[0].dx: columns >= C pass through).  Results equal ipoke_actnorm_affine_bwd's (dx, dparams bit
 * for bit; the partial sums in `split` parts). */
typedef struct ipoke_unit_pair_desc {
  const float* an_log_scale; const int32_t* an_idx; const float* an_x; float* an_part;
  int32_t Cp, t_off, t_stride; const float* x0; const float* scale; void* dparams; int32_t ldp; float* dbias_part;
  float* dx;
} ipoke_unit_pair_desc;
int ipoke_mcf_shadow_dims(int C, int Cc, int dtype, int32_t* dims8);
int ipoke_mcf_fwd(const ipoke_mcf_desc* d, int dtype, void* stream);
int ipoke_mcf_inv(const ipoke_mcf_desc* d, int dtype, void* stream);
int ipoke_mcf_bwd(const ipoke_mcf_desc* d, int dtype, void* stream);

/* Fused MaCowUnit (macow2.py:957-995: conv1(A) -> conv2(B) -> actnorm1 -> conv3(C) -> conv4(D) -> actnorm2) in ONE launch per
 * direction, one workgroup per sample; bf16 only (ipoke_macow_unit_supported).  d4 = the four masked-conv descriptors in forward
 * order, filled as for the per-layer calls with these differences:
 *   forward : d4[0].x is the unit's input; d4[k].x (k > 0) is ignored (the state stays on chip); d4[k].y may be NULL for k < 3
 *             (intermediate states not stored; the backward pass needs them); d4[3].y also receives the pass-through channels;
 *             d4[k].logdet_slot[b * w] receives the layer's whole log-det of sample b, w = 64 / d4[0].rows_per_block (the slot
 *             width of the per-layer call with that setting; 1 when rows_per_block == 0); post_log_scale / post_bias on d4[1]
 *             and d4[3] are the two ActNorms.
 *   backward: d4[k].x = saved input state of layer k; d4[3].dy = incoming gradient, d4[0].dx = gradient passed on, d4[0].dld;
 *             per layer a2_save / scale_save (from forward), dparams_save / dc_save / dbias_part (outputs), and for the layers
 *             followed by an ActNorm y_post (its saved output) / post_part. */
int ipoke_macow_unit_supported(int C, int Cc, int dtype);
int64_t ipoke_macow_unit_xchg_bytes(int B, int split);
int ipoke_macow_unit_fwd(const ipoke_mcf_desc* d4, int dtype, void* stream);
int ipoke_macow_unit_bwd(const ipoke_mcf_desc* d4, int dtype, void* stream);
/*   inverse : d4[3].x = the unit's OUTPUT state, d4[0].y = the reconstructed input (distinct buffers); W1 / W2 / bias2 /
 *             post_* per layer as for the forward call; two samples per workgroup, the 4 x 8 strips of macow2.py:174-288 */
int ipoke_macow_unit_inv(const ipoke_mcf_desc* d4, int dtype, void* stream);

/* multi-tensor weight preparation (job tables are built by the flow engine) */
int ipoke_relayout_job_size(void);
int ipoke_wn_job_size(void);
/* block_job_dev: optional int32 [total_blocks] map block -> job (NULL: every block searches the job table) */
int ipoke_relayout_multi(const float* params, void* shadow, const float* wn_scale, const void* jobs_dev, int njobs,
                         int total_blocks, const int32_t* block_job_dev, int dtype, void* stream);
int ipoke_relayout_multi_range(const float* params, void* shadow, const float* wn_scale, const void* jobs_dev, int njobs,
                               int block_begin, int nblocks, const int32_t* block_job_dev, int dtype, void* stream);
int ipoke_wn_scale_multi_range(const float* params, float* scale, float* inv_norm, const void* jobs_dev, int job_begin, int njobs,
                               int row_begin, int nrows, void* stream);
int ipoke_wn_scale_multi(const float* params, float* scale, float* inv_norm, const void* jobs_dev, int njobs,
                         int total_rows, void* stream);
int ipoke_wn_bwd_multi(const float* params, float* grads, const float* inv_norm, const void* jobs_dev, int njobs,
                       int total_rows, void* stream);
int ipoke_wn_bwd_multi_range(const float* params, float* grads, const float* inv_norm, const void* jobs_dev, int job_begin, int njobs,
                             int row_begin, int nrows, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Flow engine: the whole SupervisedMacowTransformer (INN.py:446-481; MultiScaleInternal macow2.py:821-920,
 * MaCowStep :999-1117, MaCowUnit :925-995, MultiScalePrior :543-593) as one native layer program.
 *
 * The engine defines the flat fp32 parameter layout (state-dict order of the reference, every tensor
 * starting on a 16-byte boundary) and reports the reference's state-dict names so the host can expose
 * named views.  Caller-owned buffers:
 *     params  float [param_count]          master weights (also: grads, Adam state of the same layout)
 *     perm    int32 [index_count]          forward/backward shuffle indices
 *     shadow  bytes [shadow_bytes]         matrix-core weight operands, refreshed by prepare_weights
 *     workspace bytes [workspace_bytes(B, training)]
 * ------------------------------------------------------------------------------------------- */

/* LU-parametrised invertible 1x1 convolution (macow2.py:596-649), used by the flow engine when use1x1 is set: prepare builds
 * [W | W^-1 | wl | wu] (C*C floats each) per job in the workspace; apply: out[:, :C] = in[:, :C] mat^T (or mat), rest copied,
 * over 64 * B rows (the 1x1 convolution is row-wise: one workgroup per block of 64 rows = one sample of the 8x8 latent; B counts
 * those blocks, so any row count that is a multiple of 64 is valid);
 * wgrad writes dl, du, dlog_s of one layer into the flat gradient buffer (B samples of P8 positions each). */
int ipoke_lu_job_size(void);
int ipoke_lu_prepare(const float* params, const float* fbuf, float* workspace, const void* jobs_dev, int njobs, void* stream);
/* out[m][:C] = mat (or mat^T) in[m][:C] on M state rows of pitch ld (M = B * positions per sample); columns >= C are copied */
int ipoke_lu_apply(const float* in, float* out, int64_t M, int ld, int C, const float* mat, int transposed, void* stream);
int ipoke_lu_wgrad(const float* dy, const float* x, int B, int P8, int ld, const float* params, const float* fbuf,
                   const float* workspace, const void* job_dev, const float* dld, float* grads, void* stream);

typedef struct ipoke_flow ipoke_flow;
typedef struct {
  int32_t z_channels;       /* flow_in_channels                               */
  int32_t hidden;           /* flow_mid_channels                              */
  int32_t cond_channels;    /* h_channels                                     */
  int32_t factor;
  int32_t n_levels;
  int32_t num_steps[32];
  int32_t kernel_h, kernel_w;
  int32_t dtype;            /* IPOKE_F32 / IPOKE_BF16                         */
  int32_t max_batch;
  int32_t use1x1;           /* LU-parametrised invertible 1x1 convs as the per-level shuffle layers (macow2.py:862) */
  int32_t condition_nice;   /* every NICE coupling net (steps and priors) sees the conditioning map: conv3 takes hidden + cond_channels
                             * inputs, the last cond_channels being ELU(h) (macow2.py:1024-1060, 553; macow_utils.py:275-283, 328-332) */
} ipoke_flow_config;

int ipoke_flow_create(const ipoke_flow_config* cfg, ipoke_flow** out);
void ipoke_flow_destroy(ipoke_flow* f);
int64_t ipoke_flow_param_count(const ipoke_flow* f);
int64_t ipoke_flow_index_count(const ipoke_flow* f);
int32_t ipoke_flow_tensor_count(const ipoke_flow* f);
int32_t ipoke_flow_op_count(const ipoke_flow* f);
/* kind: 0 float parameter (offset in floats), 1 forward_shuffle_idx, 2 backward_shuffle_idx (offset in
 * int32 entries of perm), 3 uint8 `initialized` flag (host-side state only, offset -1), 4 float buffer (offset in floats of
 * the float-buffer table: InvertibleConvLU1d's permutated / sign_s / lmask / umask / eye) */
int ipoke_flow_tensor_info(const ipoke_flow* f, int i, char* name, int name_cap, int64_t* offset, int32_t* ndim,
                           int64_t* shape4, int32_t* kind);
/* float buffers of the state dict (use1x1 only): element count of the table, and the caller-owned device copy the layer
 * program reads (must be set before forward / reverse / backward when the count is non-zero) */
/* hipGraph mode: the layer program of forward / reverse / one-shot backward is captured on the second call with a given
 * set of pointer arguments and replayed afterwards (IPOKE_GRAPH=1 sets the initial state; default off: on ROCm 7.2 the replay
 * of the ~1 000-node programs is 1-2 % slower than the eager launches, whose host cost is hidden behind the GPU anyway) */
int ipoke_flow_set_graph(ipoke_flow* f, int enable);
int64_t ipoke_flow_float_buffer_count(const ipoke_flow* f);
int ipoke_flow_set_float_buffers(ipoke_flow* f, const float* fbuf_dev);
int64_t ipoke_flow_shadow_bytes(const ipoke_flow* f);
/* introspection: fields of op i = {type, C, c0, Cn, p_ls, p_bias, idx_fwd, idx_bwd, order, p_w1, p_b, p_g, p_v, sh_w1,
 * sh_w1t, sh_w2, sh_w2t, wn_off, cin, cout, z_off, z_stride, t_off, t_stride, p_c1, p_c2, sh_c1, sh_c1t, sh_c2, sh_c2t,
 * sh_c3, sh_c3t}; shadow offsets are in dtype elements past ipoke_flow_shadow_base() bytes */
int ipoke_flow_op_info(const ipoke_flow* f, int i, int64_t* out32);
int64_t ipoke_flow_shadow_base(const ipoke_flow* f);
int64_t ipoke_flow_workspace_bytes(ipoke_flow* f, int B, int training);
/* fold weight norm and lay the weights out for the matrix cores; call after every parameter update */
int ipoke_flow_prepare_weights(ipoke_flow* f, const float* params, void* shadow, void* stream);
/* out, logdet = flow(x, cond)   (INN.py:469-473) */
int ipoke_flow_forward(ipoke_flow* f, const float* params, const int32_t* perm, const void* shadow, const float* x_nchw,
                       const float* cond_nchw, int B, float* out_nchw, float* logdet, void* workspace,
                       int save_for_backward, void* stream);
/* first forward with initialized == 0 everywhere: data-dependent ActNorm init, zero-init couplings
 * (macow2.py:503-505,526-539; macow_utils.py:231-250).  Overwrites ActNorm/weight-norm parameters. */
int ipoke_flow_init_forward(ipoke_flow* f, float* params, const int32_t* perm, const float* x_nchw, int B,
                            float* out_nchw, float* logdet, void* workspace, void* stream);
/* x = flow(z, cond, reverse=True)   (INN.py:475-476, macow2.py:901-920) */
int ipoke_flow_reverse(ipoke_flow* f, const float* params, const int32_t* perm, const void* shadow, const float* z_nchw,
                       const float* cond_nchw, int B, float* x_nchw, void* workspace, void* stream);
/* refresh only the shadows derived from the parameters in params[begin, end) (a range announced by
 * ipoke_flow_backward_pieces, i.e. whole levels): lets the weight preparation follow the per-group optimizer update */
int ipoke_flow_prepare_weights_range(ipoke_flow* f, const float* params, void* shadow, int64_t begin, int64_t end, void* stream);
/* torch.optim.Adam(amsgrad=True) over params[begin, end) (second_stage_video.py:648-650) fused with the refresh of the weight shadows
 * derived from that range: the plain 1x1 weights (conv2 of every coupling net, macow_utils.py:274) are updated tile by tile and their
 * two operands written from the registers that hold the new values; the rest is updated linearly and laid out as by
 * ipoke_flow_prepare_weights_range.  Equivalent to ipoke_adam_amsgrad_step_grid on the range + ipoke_flow_prepare_weights_range.
 * m, v, vmax: optimizer state with the layout of params.  [begin, end) covers whole tensors. */
int ipoke_flow_adam_range(ipoke_flow* f, float* params, const float* grads, float* m, float* v, float* vmax, void* shadow, int64_t begin,
                          int64_t end, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                          int max_blocks, void* stream);
/* its two kernels on their own (job / segment tables as the engine builds them; sizes for layout checks) */
int ipoke_adam_tile_job_size(void);
int ipoke_adam_seg_size(void);
int ipoke_adam_amsgrad_shadow_tiles(float* p, const float* g, float* m, float* v, float* vmax, void* shadow, const void* jobs_dev, int njobs,
                                    int tile_begin, int ntiles, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                    float grad_scale, int max_blocks, int dtype, void* stream);
/* the tile kernel's counterpart for tensors with ONE operand, their own cast (a flow whose conv2 data gradient reads the straight copy
 * K-major, ipoke_conv_desc.w_kmajor): a linear stream over the same job table, chunks of 4096 elements */
int ipoke_adam_amsgrad_cast_tiles(float* p, const float* g, float* m, float* v, float* vmax, void* shadow, const void* jobs_dev, int njobs,
                                  int tile_begin, int ntiles, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                  float grad_scale, int max_blocks, int dtype, void* stream);
int ipoke_adam_amsgrad_segments(float* p, const float* g, float* m, float* v, float* vmax, const void* segs_dev, int seg_begin, int nsegs,
                                int64_t begin, int64_t end, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                float grad_scale, int blocks_per_segment, void* stream);
/* parameter gradients (written, not accumulated) and optionally d/dx, given d/d_out and d/d_logdet */
int ipoke_flow_backward(ipoke_flow* f, const float* params, const int32_t* perm, const void* shadow,
                        const float* d_out_nchw, const float* d_logdet, int B, float* grads, float* dx_nchw,
                        void* workspace, void* stream);
/* The same backward issued in `npieces` groups of consecutive MaCowSteps / priors (equal parameter counts), last level first, for overlapping the data-parallel gradient
 * exchange with the rest of the backward pass (DDP bucket hooks in the reference's Lightning run,
 * experiments/second_stage_video.py:46-65).  After a group's kernels are queued, `ready_stream` is made to wait for them
 * (without blocking the backward chain on `stream`) and `ready(user, piece, begin, end)` is called on the calling thread
 * once per contiguous range grads[begin, end) that is final at that point of `ready_stream`; a collective launched there
 * on (a stream ordered after) `ready_stream` overlaps the remaining pieces.  On return `stream` is ordered after
 * `ready_stream`.  npieces = 1, ready_stream = stream, ready = NULL is ipoke_flow_backward. */
typedef void (*ipoke_grad_ready_fn)(void* user, int piece, int64_t begin, int64_t end);
/* host only: the (piece, begin, end) triples that ipoke_flow_backward_pieces(npieces) will announce, in callback order; returns their
 * number (at most max_ranges triples are written into ranges[3 * max_ranges]) */
int ipoke_flow_piece_ranges(const ipoke_flow* f, int npieces, int64_t* ranges, int max_ranges);
/* Single-process training: let the engine apply torch.optim.Adam(amsgrad=True) (second_stage_video.py:648-650) itself.  While set
 * (m != NULL; m, v, vmax: optimizer state in the layout of params), ipoke_flow_backward_pieces queues -- on `ready_stream`, right
 * behind a piece's last gradient kernel -- ipoke_adam_amsgrad_step_grid over the piece's parameter ranges and the refresh of their
 * weight shadows, i.e. what a `ready` callback would launch from the host, without the host round trip (params and shadow are
 * written: pass buffers the caller owns mutably).  `step` is the 1-based optimizer step the bias corrections use. */
int ipoke_flow_set_native_adam(ipoke_flow* f, float* m, float* v, float* vmax, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int step, float grad_scale, int max_blocks);
int ipoke_flow_backward_pieces(ipoke_flow* f, const float* params, const int32_t* perm, const void* shadow,
                               const float* d_out_nchw, const float* d_logdet, int B, float* grads, float* dx_nchw,
                               void* workspace, int npieces, void* ready_stream, ipoke_grad_ready_fn ready, void* user,
                               void* stream);
/* The row-split MaCowUnit launches and the fused conv3 + coupling launches hand partial results between workgroups through two
 * exchange scratches the flow owns; their spins are bounded and a spin that gives up lets the launch finish on garbage.  The engine
 * polls the two time-out counters at the end of every eager pass (a one-thread kernel into pinned host memory + an event) and the NEXT
 * entry point of the same flow fails with IPOKE_ERR_STATE (scratches re-initialised) once it sees a non-zero count; a pass issued on
 * another stream than the previous one is ordered behind it (the scratches serve one launch at a time).  This call is the
 * synchronising form for trainers / tests / benchmarks: it waits for the device and writes out[0] = time-outs of the unit launches,
 * out[1] = of the coupling launches since the scratches were last (re-)initialised.  (The reference has no counterpart: a single
 * PyTorch stream orders macow2.py:925-995 / macow_utils.py:270-281 by construction.) */
int ipoke_flow_handoff_timeouts(ipoke_flow* f, uint32_t* out);
/* the hipStream_t (as void*) of the flow's weight-gradient side stream, NULL on failure: lets the host verify that the streams IT
 * keeps busy beside the backward pass do not share a hardware queue with it (the reference has one stream; experiments/experiment.py
 * runs DDP's bucket all-reduce on NCCL's own) */
void* ipoke_flow_side_stream(ipoke_flow* f);


/* ---------------------------------------------------------------------------------------------
 * First-stage VAE helpers on channels-last activations [N][S][ld] of dtype.
 * ------------------------------------------------------------------------------------------- */
/* GroupNorm (InstanceNorm: G = C, no affine) fused with affine / SPADE modulation / residual / activation:
 *   y = act( xhat * gamma + beta  [ * (1 + mod_gamma) + mod_beta ]  [ + res ] )      (res_post: act( ... ) + res)
 * Replaces nn.GroupNorm / nn.InstanceNorm2d call sites: motion_encoder.py:49-72, autoencoders/util.py:26-36,
 * 223-233 and Spade.forward util.py:494-500. */
typedef struct {
  const void* x; int32_t ldx; void* y; int32_t ldy; int32_t y_f32;
  int32_t N, S, C, G; float eps;
  const float* gamma; const float* beta;                 /* [C] or NULL */
  const void* mod_gamma; const void* mod_beta; int32_t ld_mod;
  const void* res; int32_t ld_res;
  int32_t act;
  float* workspace;                                      /* ipoke_groupnorm_workspace_floats(N, S, G) floats */
  int32_t mod_samples;                                   /* > 0: mod_gamma / mod_beta hold mod_samples samples and sample n reads those of
                                                            sample n % mod_samples (all generated frames of a clip share the SPADE maps of
                                                            its start frame: the frames are decoded as ONE batch ordered (frame, clip)) */
  int32_t res_post;                                      /* 1: y = act( ... ) + res -- the residual joins BEHIND the activation: ResBlock's
                                                            `conv2(conv1(x)) + res_conv(x)` (util.py:106-192) with the norm + activation of
                                                            res_conv as this call and conv2's output as `res`, one pass instead of two */
  /* statistics handed from one norm to the next (the SPADE norm behind such a ResBlock): the producer call also writes the chunk
   * statistics (count, mean, M2 per (sample, chunk of 256 positions, group of next_G groups)) of its OUTPUT to next_part --
   * ipoke_groupnorm_workspace_floats(N, S, next_G) floats, which the consumer call then passes as ITS workspace with
   * part_chunks = ceil(S / 256): its own statistics pass over the tensor is skipped */
  float* next_part; int32_t next_G;
  int32_t part_chunks;
} ipoke_norm_desc;
int64_t ipoke_groupnorm_workspace_floats(int N, int S, int G);
int ipoke_groupnorm(const ipoke_norm_desc* d, int dtype, void* stream);
int ipoke_add_act(const void* a, int lda, const void* b, int ldb, void* y, int ldy, int64_t M, int C, int act, int dtype,
                  void* stream);
/* ConvGRU cell element-wise stages (rnn.py:48-56) */
int ipoke_gru_gates(const void* ur_pre, const void* h, int ldh, void* hr_out, int ld_hr, void* u_out, int64_t M, int Ch,
                    int dtype, void* stream);
int ipoke_gru_update(const void* o_pre, const void* u, const void* h, int ldh, void* h_new, int ld_new, int64_t M, int Ch,
                     int dtype, void* stream);
/* z = mu + eps*exp(logvar/2) (motion_encoder.py:218-222); mulv = [mu | logvar] per row */
int ipoke_reparameterize(const void* mulv, int ld, const float* eps, float* z, float* mu, float* logvar, int64_t M, int Z,
                         int dtype, void* stream);
/* F.interpolate(mode='bilinear', align_corners=True) (util.py:495): NCHW fp32 -> channels-last fp32 */
int ipoke_bilinear_cl(const float* x_nchw, float* y_cl, int N, int C, int Hi, int Wi, int Ho, int Wo, void* stream);
int ipoke_cl_to_nchw(const void* x_cl, int ld, float* y, int N, int C, int S, int dtype, void* stream);
int ipoke_nchw_to_cl(const float* x, void* y_cl, int ld, int N, int C, int S, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * First-stage VAE training: backward of the helpers above.  These replace the
 * autograd of SpadeCondMotionModel.forward (first_stage_motion_model.py:469-522) and of the L1 + KL part of
 * MotionModel.training_step (:263-276); convolutions use ipoke_conv_forward (data gradient = the transposed /
 * direct convolution with the same weights) and ipoke_conv_wgrad.
 * ------------------------------------------------------------------------------------------- */
/* statistics only: fills the workspace exactly as ipoke_groupnorm does (mean, rstd per (n, g)) */
int ipoke_groupnorm_stats(const void* x, int ldx, int N, int S, int C, int G, float eps, float* workspace, int dtype, void* stream);
/* backward of  y = act( xhat*gamma + beta [*(1 + mod_gamma) + mod_beta] [+ res] ):
 *   dx, dres (= d/d res), dmod_gamma, dmod_beta (dtype, per position) and dgamma, dbeta (fp32 [C], written). */
typedef struct {
  const void* x; int32_t ldx;            /* saved input                                              */
  const void* y; int32_t ldy;            /* saved output (needed when act != NONE)                   */
  const void* dy; int32_t lddy;
  void* dx; int32_t lddx;
  void* dres; int32_t lddres;            /* NULL when the forward had no residual                    */
  void* dmod_gamma; void* dmod_beta; int32_t ld_dmod;   /* NULL without SPADE modulation              */
  int32_t N, S, C, G; float eps;
  const float* gamma; const float* beta; /* [C] or NULL                                              */
  const void* mod_gamma; int32_t ld_mod; /* saved modulation gamma                                   */
  float* dgamma; float* dbeta;           /* [C] fp32 or NULL                                         */
  int32_t act;
  float* workspace;                      /* ipoke_groupnorm_bwd_workspace_floats(N, S, C, G) floats  */
  const float* stats;                    /* optional [N][G][2] (mean, rstd) kept from the forward pass (the tail of the forward
                                            workspace, see ipoke_groupnorm_stats_offset); NULL: recomputed from x */
  int32_t mod_samples;                   /* as ipoke_norm_desc.mod_samples: sample n read the modulation of sample n % mod_samples;
                                            dmod_gamma / dmod_beta are still written per sample (sum the frames: ipoke_sum_frames) */
  /* optional (rs_scale != NULL; no SPADE modulation): x is the output of a frame-batched spectral-norm convolution WITHOUT activation,
   * y_conv = conv(.) / sigma_t + b, that only this norm reads.  The pass over (dy, x) that ipoke_rowscale_bwd would make on the norm's dx
   * is folded into the norm's own last pass: `dx` receives gs = round(dx) / sigma_t (the rows of the convolution's two gradient GEMMs),
   * rs_dots[t] = sum over the rows of frame t of round(dx) * (x - b), rs_dbias[c] = column sums of round(dx) -- the same values from the
   * same rounded dx, summed over blocks of 512 positions in a fixed order.  The samples are ordered (frame, clip). */
  const float* rs_scale; int32_t rs_scale_stride;   /* 1 / sigma_t of frame t at rs_scale[t * rs_scale_stride]                     */
  int64_t rs_rows_per_group;             /* rows of one frame = clips * S; N * S must be a multiple                               */
  const float* rs_bias;                  /* the convolution's bias [C] (fp32) or NULL                                              */
  float* rs_dots;                        /* [N * S / rs_rows_per_group]                                                            */
  float* rs_dbias;                       /* [C] or NULL                                                                            */
  float* rs_workspace;                   /* ipoke_groupnorm_bwd_rs_workspace_floats(N, S, C) floats                                */
  int32_t dmod_summed;                   /* 1 (with 0 < mod_samples < N): dmod_gamma / dmod_beta hold mod_samples * S rows -- the SUM over the
                                            frames that share a modulation row, accumulated in fp32 in frame order and rounded once (no
                                            per-sample maps, no ipoke_sum_frames pass)                                              */
  int32_t act_from_pre;                  /* 1: the forward pass was y = act(pre) + res (ipoke_norm_desc.res_post): act' is evaluated from the
                                            pre-activation value recomputed from x and the statistics (y is not read); the residual's
                                            gradient is dy itself (dres must be NULL); no SPADE modulation                            */
} ipoke_norm_bwd_desc;
/* float offset of the (mean, rstd) table inside the workspace ipoke_groupnorm / ipoke_groupnorm_stats just filled */
int64_t ipoke_groupnorm_stats_offset(int N, int S, int G);
int64_t ipoke_groupnorm_bwd_workspace_floats(int N, int S, int C, int G);
int64_t ipoke_groupnorm_bwd_rs_workspace_floats(int N, int S, int C);
int ipoke_groupnorm_bwd(const ipoke_norm_bwd_desc* d, int dtype, void* stream);
/* out[:, :C] = dy * act'(y), out[:, C:Cpad] = 0   (backward of an activation fused into a conv / add epilogue) */
int ipoke_act_bwd(const void* dy, int lddy, const void* y, int ldy, void* out, int ldo, int64_t M, int C, int Cpad, int act,
                  int dtype, void* stream);
/* bias gradients: out[c] (+)= sum_m src[m][c] */
int64_t ipoke_colsum_workspace_floats(int64_t M, int C);
int ipoke_colsum(const void* src, int ld, int64_t M, int C, int src_f32, float* out, int accumulate, float* workspace, int dtype,
                 void* stream);
/* ConvGRU cell (rnn.py:48-56) backward of the two element-wise stages */
int ipoke_gru_update_bwd(const void* o_pre, const void* u, const void* h, int ldh, const void* d_hnew, int ld_dhnew, void* d_o_pre,
                         void* d_u, void* d_h, int ld_dh, int64_t M, int Ch, int dtype, void* stream);
int ipoke_gru_gates_bwd(const void* ur_pre, const void* h, int ldh, const void* d_hr, int ld_dhr, const void* d_u, void* d_ur_pre,
                        void* d_h, int ld_dh, int64_t M, int Ch, int dtype, void* stream);
/* ---- first-stage training helpers (vae_train.hip) ------------------------------------------------------------------
 * Conv weight in PyTorch layout (fp32 [cout][cin][taps], or ConvTranspose storage [cin][cout][taps] with transposed = 1) ->
 * the [cout][taps*kc] operand of the compute dtype that ipoke_conv_forward reads, scaled by *inv_scale (device, may be NULL:
 * 1/sigma of spectral norm).  Replaces the reshape/permute/pad/cast chain of nn.Conv2d's weight on every call. */
int ipoke_conv_weight_operand(const float* w, int cout, int cin, int taps, int transposed, const float* inv_scale, void* out, int kc,
                              int dtype, void* stream);
/* the same for `count` weights in one launch (no inv_scale): w / out are HOST arrays of device pointers, dims5 a HOST array of
 * {cout, cin, taps, transposed, kc} per weight -- the refresh of every cached operand behind an optimizer step */
int ipoke_conv_weight_operand_multi(const float* const* w, void* const* out, const int32_t* dims5, int count, int dtype, void* stream);
/* ---- native unroll of the ConvGRU (csrc/gru.hip) ---------------------------------------------------------------------------------
 * T steps x L stacked ConvGRU cells on the [B][H][W] latent (reference models/modules/motion_models/rnn.py:4-133 as
 * SpadeCondMotionModel.forward drives it, models/first_stage_motion_model.py:503-514: every cell starts from the same hidden state, cell 0
 * sees a constant input), forward and backward, issued back to back by host code of this library.  Replaces the Python loop over
 * ConvGRUCell.forward and its autograd graph.  Activations are dtype rows [M = B*H*W][ld]; weights fp32 in PyTorch layout, 4 pointers per
 * cell: w_ur [2Ch][Cx+Ch][3][3] (update gate's rows, then reset gate's), b_ur [2Ch], w_o [Ch][Cx+Ch][3][3], b_o [Ch]. */
typedef struct { int32_t B, T, L, Cx, Ch, H, W; } ipoke_gru_desc;
int64_t ipoke_gru_workspace_bytes(const ipoke_gru_desc* d, int dtype);
/* out [T][M][ldo]: the last cell's hidden state after every step.  The workspace keeps every operand for ipoke_gru_unroll_backward. */
/* The forward unroll runs as ONE launch where it applies (a workgroup per sample runs the T x L recurrence on LDS-resident operands;
 * bf16, 8 x 8 map, Cx = Ch in {32, 64}), otherwise as four launches per cell and step; both forms fill the same workspace for
 * ipoke_gru_unroll_backward (test hook to force either: ipoke_gru_set_fused, ipoke_hip_dev.h). */
int ipoke_gru_unroll_forward(const ipoke_gru_desc* d, const void* x0, int ldx, const void* h0, int ldh, const float* const* weights,
                             void* workspace, void* out, int ldo, int dtype, void* stream);
/* d_out [T][M][ldo] -> dweights (the layouts of `weights`, written), d_x0 [M][Cx] and d_h0 [M][Ch] (fp32; d_h0 summed over the cells) */
int ipoke_gru_unroll_backward(const ipoke_gru_desc* d, const void* d_out, int ldo, void* workspace, float* const* dweights, float* d_x0,
                              float* d_h0, int dtype, void* stream);
/* ---- frames of a batch of clips decoded as ONE batch (first-stage training, ipoke_conv_desc.row_scale) ----------------------------
 * Backward pass over (dy, y) of a layer  y = act(conv(x, W) * scale[group] + bias)  whose rows are grouped by frame (group of row m =
 * m / rows_per_group; scale[group * scale_stride] = 1 / sigma_t of torch's spectral_norm, util.py:52, 252):
 *   gs[m][c]    = dy * act'(y) * scale[group]      (dtype, columns C .. Cpad zero): rows of the data- and weight-gradient GEMMs
 *   dots[group] = sum over the group's rows and channels of dy * act'(y) * (pre(y) - bias[c])  = <dW_eff_t, W / sigma_t>
 *   dbias[c]    = column sums of dy * act'(y)      (optional)
 * pre(y) is the pre-activation recovered from the saved output (act: NONE, RELU, ELU, LRELU02).  Deterministic (per-block partial
 * sums reduced in a fixed order).  Replaces, for all T - 1 decoder calls of a training pass at once, autograd's backward of
 * `weight_orig / sigma` in torch.nn.utils.spectral_norm's compute_weight. */
typedef struct {
  const void* dy; int32_t lddy;
  const void* y; int32_t ldy;            /* saved layer output (post-activation)                     */
  int64_t M; int32_t C, Cpad; int32_t act;
  const float* bias;                     /* [C] or NULL                                              */
  const float* scale; int32_t scale_stride; int64_t rows_per_group;
  void* gs; int32_t ldgs;
  float* dots;                           /* [M / rows_per_group]                                     */
  float* dbias;                          /* [C] or NULL                                              */
  float* workspace;                      /* ipoke_rowscale_bwd_workspace_floats(M, C, rows_per_group) floats */
} ipoke_rowscale_bwd_desc;
int64_t ipoke_rowscale_bwd_workspace_floats(int64_t M, int C, int64_t rows_per_group);
int ipoke_rowscale_bwd(const ipoke_rowscale_bwd_desc* d, int dtype, void* stream);
/* In place: grad (gradient of W_orig in PyTorch weight layout, holding sum_t dW_eff_t / sigma_t: the weight gradient of the rows
 * scaled by 1 / sigma_t) -= sum_t dots[t] / sigma_t * u_t v_t^T -- the part that reaches W_orig through sigma_t = u_t^T W v_t.
 * snapshots[t * snap_stride ..] = u_t | v_t and sig[t * sig_stride ..] = {sigma_t, 1 / sigma_t} as ipoke_spectral_sigma_multi wrote them. */
int ipoke_spectral_bwd_frames(const float* w, int cout, int cin, int taps, int transposed, float* grad, const float* snapshots,
                              int64_t snap_stride, const float* sig, int64_t sig_stride, const float* dots, int frames, void* stream);
/* dst[i] = sum_{f < frames} src[f * n + i] (dtype in and out, fp32 accumulation; n a multiple of 16 bytes): gradients of maps shared by
 * the frames of a clip -- the SPADE modulation of util.py:494-500, which autograd sums over the reference's T - 1 decoder calls */
int ipoke_sum_frames(const void* src, void* dst, int frames, int64_t n, int dtype, void* stream);
/* torch.nn.utils.spectral_norm (reference: models/modules/autoencoders/util.py:52,252): one power iteration
 * v = normalize(W^T u), u = normalize(W v) (iterate != 0; u, v updated in place, eps as in F.normalize) and
 * sigma = u^T W v with W = weight.reshape(cout, -1) (weight.transpose(0,1).reshape(cout, -1) when transposed).
 * out = {sigma, 1/sigma}; snapshot (optional, cout + cin*taps floats) receives the u | v sigma was computed with.
 * workspace: ipoke_spectral_workspace_floats() floats, zeroed once by the caller. */
long ipoke_spectral_workspace_floats(int cout, int cin, int taps);
int ipoke_spectral_sigma(const float* w, int cout, int cin, int taps, int transposed, float* u, float* v, int iterate, float eps,
                         float* out, float* snapshot, float* workspace, void* stream);
/* Several power iterations of several weights ahead of time: torch's spectral_norm iterates once per forward call of the wrapped
 * module (first_stage_motion_model.py decodes T - 1 frames per pass through the same decoder convolutions); the iterations depend on
 * the weight and its own u / v only, so the T - 1 {sigma, 1/sigma} pairs and u | v snapshots of every decoder weight are produced by
 * 3 launches per iteration instead of 3 per call.  Same arithmetic as ipoke_spectral_sigma(iterate = 1), call by call. */
typedef struct {
  const float* w; int32_t cout, cin, taps, transposed;
  float* u; float* v;                    /* updated in place, as after `iterations` calls                                  */
  float* out; int64_t out_stride;        /* iteration k: {sigma, 1/sigma} at out + k * out_stride                            */
  float* snap; int64_t snap_stride;      /* iteration k: u | v snapshot (cout + cin*taps floats) at snap + k * snap_stride    */
  float* workspace;                      /* ipoke_spectral_workspace_floats floats, zeroed once (shared with ipoke_spectral_sigma) */
} ipoke_sn_job;
int ipoke_sn_job_size(void);             /* bytes per job of the device-side table the caller provides */
/* host jobs -> device table (synchronises; once per model: the pointers stay valid across steps) */
int ipoke_sn_jobs_upload(const ipoke_sn_job* jobs, int njobs, void* jobs_dev, void* stream);
/* `iterations` power iterations of every job, 3 launches per iteration; max_rows / max_cols: largest cout / cin*taps of the table */
int ipoke_spectral_sigma_multi(const void* jobs_dev, int njobs, int max_rows, int max_cols, int iterations, float eps, void* stream);
/* in place: gradient w.r.t. w_orig / sigma -> gradient w.r.t. w_orig:  (G - <G, W/sigma> u v^T) / sigma.
 * workspace: ipoke_spectral_bwd_workspace_floats() floats (per-workgroup partial sums, summed in a fixed order; no initialisation). */
long ipoke_spectral_bwd_workspace_floats(void);
int ipoke_spectral_bwd(const float* w, int cout, int cin, int taps, int transposed, float* grad, const float* snapshot,
                       const float* sig, float* workspace, void* stream);
/* torch.optim.Adam (amsgrad off, coupled weight decay; reference first_stage_motion_model.py:283-300) over `count` tensors;
 * p/g/m/v/n are HOST arrays of device pointers / element counts (passed to the kernel by value, 48 tensors per launch). */
int ipoke_adam_multi(float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n, int count, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
/* KL(q || N(0, I)) as in utils/losses.py:47-48: loss[0] += -0.5 * mean over positions of sum_c (1 + lv - mu^2 - exp(lv));
 * dmu / dlv receive its gradient.  fp32 [positions * Z], any common element order. */
int ipoke_kl_loss(const float* mu, const float* lv, int64_t positions, int Z, float* loss, float* dmu, float* dlv, void* stream);

/* Pooling of the first-stage temporal discriminator on channels-last rows (reference patchgan_3d.py:202 MaxPool3d(3, (1,2,2), 1)
 * and :209 AvgPool3d((1, s, s))).  dims = {N, C, Di, Hi, Wi, Do, Ho, Wo, kd, kh, kw, sd, sh, sw, pd, ph, pw}; idx int32
 * [out rows][C] keeps the chosen input row (first maximum in d, h, w scan order, as torch) for the backward pass, which is
 * a gather (no atomics).  avgpool_rows: mean over the S consecutive rows of each of G groups. */
int ipoke_maxpool3d_fwd(const int* dims, const void* x, int ldx, void* y, int ldy, int* idx, int dtype, void* stream);
int ipoke_maxpool3d_bwd(const int* dims, const void* dy, int ldy, const int* idx, void* dx, int ldx, int dtype, void* stream);
int ipoke_avgpool_rows(const void* x, int ldx, void* y, int ldy, int64_t G, int S, int C, int dtype, void* stream);
int ipoke_avgpool_rows_bwd(const void* dy, int ldy, void* dx, int ldx, int64_t G, int S, int C, int dtype, void* stream);

/* GroupNorm tangent for the discriminators' gradient penalty (patchgan_3d.py:285-294), evaluated forward-over-reverse (see
 * vae_train.hip): ydot = act'(y) (gamma r (xdot - <xdot> - xhat <xhat xdot>) + resdot) per (sample, group); the backward
 * returns the gradients on xdot, on the PRIMAL input x (the second-order term, through the statistics), on resdot and on
 * gamma (fp32 [C], atomically accumulated).  x, xdot, y, q, ... are channels-last rows of the compute dtype. */
int ipoke_groupnorm_jvp(const void* x, int ldx, const void* xdot, int ldxd, const void* y, int ldy, const void* resdot, int ldres,
                        void* ydot, int ldyd, const float* gamma, int N, int S, int C, int G, int act, float eps, float* workspace, int dtype,
                        void* stream);
int ipoke_groupnorm_jvp_bwd(const void* x, int ldx, const void* xdot, int ldxd, const void* y, int ldy, const void* q, int ldq,
                            void* dxdot, int lddxd, void* dx, int lddx, void* dresdot, int lddres, float* dgamma, const float* gamma,
                            int N, int S, int C, int G, int act, float eps, float* workspace, int dtype, void* stream);
long ipoke_groupnorm_jvp_workspace_floats(int N, int G);     /* workspace of the two calls above */
/* y[o][c] = x[idx[o][c]][c]: the tangent of MaxPool3d under the primal pass's selection (idx of ipoke_maxpool3d_fwd). */
int ipoke_gather_rows(const void* x, int ldx, const int* idx, void* y, int ldy, int64_t Mo, int C, int dtype, void* stream);

/* Feature-matching term of the discriminators (patchgan_3d.py:297-304, patchgan.py:449-457): loss[0] += scale * sum |a - b| over
 * two channels-last maps of the compute dtype; grad [M][ldg] = scale * sign(a - b), the gradient w.r.t. a. */
int ipoke_l1_pair(const void* a, int lda, const void* b, int ldb, int64_t M, int C, float scale, float* loss, void* grad, int ldg, int dtype,
                  void* stream);

/* reparameterize backward: dmulv = [dz + dmu | dz*eps*exp(lv/2)/2 + dlv]  (any of dz/dmu/dlv may be NULL) */
int ipoke_reparam_bwd(const void* mulv, int ld, const float* eps, const float* dz, const float* dmu, const float* dlv, void* dmulv,
                      int ldo, int64_t M, int Z, int dtype, void* stream);
/* *loss_accum += scale * sum |yhat - x|, grad = scale * sign(yhat - x); yhat channels-last fp32, x fp32 [N][C][S]
 * with sample stride x_sn (first_stage_motion_model.py:265: L1 between the generated and the true frames).
 * partials: ipoke_l1_loss_partials() floats of scratch -> the value is summed in a fixed order (reproducible run to run);
 * NULL: the workgroups add their sums with float atomics */
long ipoke_l1_loss_partials(void);
int ipoke_l1_loss(const float* yhat_cl, int ldy, const float* x_nchw, int N, int C, int S, int64_t x_sn, float scale,
                  float* loss_accum, float* grad_cl, int ldg, float* partials, void* stream);

/* ---------------------------------------------------------------------------------------------
 * FVD evaluation (reference utils/metrics.py; host side in ipoke_amd/fvd.py).  The I3D convolutions are ipoke_conv_forward
 * with the eval-mode BatchNorm folded into weight and bias; these are the element-wise / window kernels around them.
 * ------------------------------------------------------------------------------------------- */
/* motion_encoder.py:161 (ResNetMotionEncoder.conv1 = nn.Conv3d(3, 64, (3, 7, 7), 2, (1, 3, 3)) on the fp32 clip): the clip
 * src[n][c][t][y][x] (element strides s_*, 3 channels) as channels-last pixels of FOUR channels of the compute dtype (the fourth 0),
 * rows dst[((n*T + t)*H + y)*(pad_l + W + pad_r) + pad_l + x][4] with zero columns left and right -- a 7-tap window along x is
 * then one 16-byte aligned run of 8 pixels and the stem runs as 21 taps of 32 channels through ipoke_conv_forward. */
int ipoke_clip_to_cl4(const float* src, int64_t s_n, int64_t s_c, int64_t s_t, int64_t s_h, int64_t s_w, int N, int T, int H, int W,
                      int pad_l, int pad_r, void* dst, int dtype, void* stream);
/* metrics.py:787-792 (F.interpolate(..., mode='bilinear', size=(224, 224), align_corners=True) of every frame): frame f of clip n
 * of a strided fp32 tensor (element strides s_n, s_f, s_c, s_h, s_w) -> channels-last fp32 rows dst[(frame*Ho + y)*Wp + pad_l + x][C],
 * Wp = pad_l + Wo + pad_r, border columns zero (the stem convolution's TF-SAME padding along W is stored so that it can read a
 * 7-pixel window as one tap).  minval (may be NULL) receives the running minimum of the produced values (start it with
 * ipoke_min_reset); dst == NULL takes the minimum only.  With Ho == Hi, Wo == Wi the copy is exact. */
int ipoke_video_to_cl(const float* src, int64_t s_n, int64_t s_f, int64_t s_c, int64_t s_h, int64_t s_w, int N, int T, int C, int Hi,
                      int Wi, float* dst, int Ho, int Wo, int pad_l, int pad_r, float* minval, void* stream);
int ipoke_min_reset(float* minval, void* stream);
/* metrics.py:794-798: x = (x + 1) / 2 on the interior columns of `rows` padded rows when *minval < 0 (decided on the device) */
int ipoke_denorm_if_negative(float* x, int64_t rows, int Wo, int pad_l, int pad_r, int C, const float* minval, void* stream);
/* metrics.py:450-481 (SSIM_custom / PSNR_custom = pytorch_lightning.metrics.functional.ssim / psnr with their defaults; logged per
 * validation batch, second_stage_video.py:511-512): out[0] = 10 log10((max(target) - min(target))^2 / mse), out[1] = mean SSIM map
 * (11 x 11 Gaussian window, sigma 1.5, k1 = 0.01, k2 = 0.03, data range = the larger of the two tensors' ranges) over the positions
 * whose windows lie inside the image, of `planes` fp32 planes [planes][H][W] (N * C planes of NCHW tensors); H, W >= 11.
 * workspace: ipoke_image_metrics_workspace_bytes(planes, H, W) bytes. */
int64_t ipoke_image_metrics_workspace_bytes(int64_t planes, int H, int W);
int ipoke_psnr_ssim(const float* preds, const float* target, int64_t planes, int H, int W, void* workspace, float* out, void* stream);
/* metrics.py:939-960 MaxPool3dTFPadding: ZERO padding, then MaxPool3d(ceil_mode=True), on channels-last rows of the compute dtype.
 * dims = {N, C, Di, Hi, Wi, Do, Ho, Wo, kd, kh, kw, sd, sh, sw, pd, ph, pw, ed, eh, ew}: p* = front padding, e* = input extent +
 * back padding (window positions in the zero border count as 0, positions beyond it are ignored). */
int ipoke_pool3d_same(const int* dims, const void* x, int ldx, void* y, int ldy, int dtype, void* stream);
/* y[g][c] = sum_k w[k] x[g*S + k][c] (w: device fp32 [S]): metrics.py:1085, 1091 -- AvgPool3d((2, 7, 7), 1) and the mean over the
 * remaining time steps as one weighted mean over a clip's rows, taken before the (linear) logits convolution. */
int ipoke_pool_rows_weighted(const void* x, int ldx, void* y, int ldy, int64_t G, int S, int C, const float* w, int dtype, void* stream);
/* metrics.py:733-771 (calculate_moments / calculate_activation_statistics): rows of act fp32 [n][D] without any non-NaN entry are
 * dropped, mu = mean, sigma = np.cov(rowvar=False), both float64.  workspace: (n + 1) int32. */
int ipoke_activation_moments(const float* act, int n, int D, double* mu, double* sigma, int* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Input pipeline on the device (reference data/base_dataset.py; host side in ipoke_amd/data.py).
 * ------------------------------------------------------------------------------------------- */
/* _get_flow (:651-693): dst [B][C][Ho][Wo] = bilinear, align_corners=True, of src [B][C][Hi][Wi] / divide_by (the division is applied
 * to the source values first, as the reference does for scale_poke_to_res: divide_by = Hi / Ho; pass 1 otherwise). */
int ipoke_flow_resize(const float* src, float* dst, int B, int C, int Hi, int Wi, int Ho, int Wo, float divide_by, void* stream);
/* _get_poke (:507-648) for a batch of flows [B][2][H][W]: candidate positions from the normalised flow amplitude on the window
 * [poke_size, size - poke_size)^2 (amplitude > mean + 2 std, fallbacks > mean + std, > mean; zero-poke samples: centres below the
 * 5th percentile, values from positions > mean + std / > mean), the number of pokes and the picks from the caller's uniforms
 * u [B][1 + 2 n_pokes] in [0, 1): u[0] -> count = 1 + floor(u min(n_pokes, #candidates)) unless fix_n_pokes, u[1 .. n_pokes] -> value
 * sources (zero-poke samples), u[1 + n_pokes ..] -> centres, each as floor(u * #set) in row-major order of the set.
 * Outputs: poke [B][2][H][W] (later pokes overwrite earlier ones), centers int64 [B][n_pokes][2] = (row, col), -1 padded,
 * flow_out (optional) = flow, zeroed for zero-poke samples (:680-681), status [B] (1: no candidate -- the reference raises FlowError
 * and resamples).  zero_poke: int32 [B] or NULL.  workspace: ipoke_poke_workspace_bytes. */
int64_t ipoke_poke_workspace_bytes(int B, int H, int W, int poke_size, int n_pokes);
int ipoke_poke_simulate(const float* flow, int B, int H, int W, int poke_size, int n_pokes, int fix_n_pokes, int equal_poke_val,
                        const int* zero_poke, const float* u, float* poke, int64_t* centers, float* flow_out, int* status, void* workspace,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IPOKE_HIP_H */
