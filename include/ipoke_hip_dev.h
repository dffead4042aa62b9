/*
 * libipoke_hip -- developer / test hooks of the library (NOT part of the drop-in boundary of include/ipoke_hip.h): kernel-dispatch
 * overrides and dispatch introspection for the parity tests, in-situ event timing for bench.py's roofline objects, repeated launches
 * and a spin kernel for the probe scripts.  Same conventions as ipoke_hip.h (status codes, ipoke_last_error, `stream` = hipStream_t).
 */
#ifndef IPOKE_HIP_DEV_H
#define IPOKE_HIP_DEV_H

#include "ipoke_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* n back-to-back native launches of the same convolution (kernel timing without host round trips) */
int ipoke_conv_forward_repeat(const ipoke_conv_desc* d, int dtype, int n, void* stream);
/* Test hook: kernel-dispatch switch `name` ("c64": conv3x3_c64, "halo16": conv3x3_halo16) <- value (0 off, 1 the measured default
 * rule, 2 wherever the kernel can run; < 0: back to the environment default IPOKE_C64 / IPOKE_HALO16).  The switches are read from the
 * environment once per process -- no getenv on the launch path. */
int ipoke_set_dispatch_override(const char* name, int value);
/* Test hook: the kernel family the calling thread's last ipoke_conv_forward was dispatched to */
enum { IPOKE_KERNEL_NONE = 0, IPOKE_KERNEL_IGEMM = 1, IPOKE_KERNEL_S8 = 2, IPOKE_KERNEL_HALO = 3, IPOKE_KERNEL_HALO16 = 4, IPOKE_KERNEL_C64 = 5, IPOKE_KERNEL_K8 = 6 };
int ipoke_last_conv_kernel(void);

/* developer probe (IPOKE_SIDE_DELAY_US): one wave spinning for about `us` microseconds on `stream` */
int ipoke_spin_delay(int us, void* stream);

/* In-situ timing for the benchmark's roofline objects: between ipoke_timing_start() and ipoke_timing_stop() every launch of
 * a tagged kernel family is bracketed by HIP events on the stream it is launched on (the rest of the step runs as usual).
 * Tags: 1 = ipoke_conv_forward with a 1x1 kernel and Nout = K >= 1024 (the NICE conv2 GEMM; its data gradient too unless that is
 *       read from the K-major weight: tag 6),
 *       2 = ipoke_conv_wgrad / _batched of the same shape (a batched launch counts once per problem and its time is divided
 *       by the problem count), 3 = ipoke_macow_unit_fwd, 4 = ipoke_macow_unit_bwd. */
#define IPOKE_TAG_NT_SQUARE 1
#define IPOKE_TAG_TN_SQUARE 2
#define IPOKE_TAG_UNIT_FWD 3
#define IPOKE_TAG_UNIT_BWD 4
#define IPOKE_TAG_UNIT_INV 5
#define IPOKE_TAG_NN_SQUARE 6     /* the same square GEMM with the weight read K-major (ipoke_conv_desc.w_kmajor): the conv2 data gradient */
/* every other ipoke_conv_forward launch is tagged by the kernel family the dispatcher chose (IPOKE_TAG_CONV_BASE + IPOKE_KERNEL_*), every
 * other weight gradient IPOKE_TAG_WGRAD; both carry their algorithmic work: FLOPs = 2 * rows * Nout * taps * channels (transposed
 * strided forms: divided by the stride product -- the taps that meet an input pixel), bytes = input + weights + output, each once. */
#define IPOKE_TAG_CONV_BASE 16
#define IPOKE_TAG_WGRAD 32
int ipoke_timing_start(void);
int ipoke_timing_start_all(void);   /* also the IPOKE_TAG_CONV_* / IPOKE_TAG_WGRAD families */
int ipoke_timing_stop(const int* tags, int ntags, int* counts, double* mean_us);
/* per tag: launches, SUM of durations (us), SUM of algorithmic FLOPs and bytes -- the per-configuration rooflines of bench.py */
int ipoke_timing_stop_ex(const int* tags, int ntags, int* counts, double* total_us, double* flops, double* bytes);

/* Test hook: sizeof of the descriptor structs of ipoke_hip.h in the order conv, wgrad, affine, coupling_epi, mcf, unit_pair, flow_config,
 * norm, norm_bwd, rowscale_bwd, sn_job, wgrad_adam (out: at least 12 entries; returns the count) -- the binding's own structs must match. */
int ipoke_desc_sizes(int32_t* out, int n);

/* Test hook: make the flow's next polled pass report a hand-off time-out (which = 0: the row-split unit scratch, 1: the fused
 * conv3 + coupling scratch) -- exercises the engine's IPOKE_ERR_STATE + scratch re-initialisation path (ipoke_flow_handoff_timeouts) */
int ipoke_flow_test_inject_timeout(ipoke_flow* f, int which, void* stream);

/* Test hook: forward unroll of the ConvGRU as one launch (1), as launches per phase (0), or the IPOKE_GRU_FUSED environment default (< 0) */
int ipoke_gru_set_fused(int mode);

#ifdef __cplusplus
}
#endif
#endif
