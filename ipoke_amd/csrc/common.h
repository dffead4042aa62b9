// Shared device/host helpers for libipoke_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/ipoke_hip.h"
#include "../../include/ipoke_hip_dev.h"

namespace ipoke {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// ---- error plumbing (never throws across the C ABI) -----------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define IPK_HIP(expr)                                                                      \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return ::ipoke::fail(IPOKE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

#define IPK_REQUIRE(cond, msg)                                              \
  do {                                                                      \
    if (!(cond)) return ::ipoke::fail(IPOKE_ERR_INVALID, std::string(msg) + " [" #cond "]"); \
  } while (0)

#define IPK_LAUNCH_CHECK() IPK_HIP(hipGetLastError())

// ---- in-situ timing of tagged launches (common.cpp; bench.py's roofline objects) ---------------------------------
bool timing_active(int tag = 1);      // tags >= IPOKE_TAG_CONV_BASE are recorded only at level 2 (ipoke_timing_start_all)
int timing_begin(int tag, hipStream_t s, int units = 1);
void timing_end(int slot, hipStream_t s);
void timing_annotate(int slot, int tag, double flops, double bytes);
struct TimedScope {       // records an event pair around the launches issued while it is alive (no-op unless timing is on)
  int slot; hipStream_t s;
  TimedScope(int tag, hipStream_t st, int units = 1) : slot(tag != 0 && timing_active(tag) ? timing_begin(tag, st, units) : -1), s(st) {}
  ~TimedScope() { timing_end(slot, s); }
  // set once the dispatcher has chosen the kernel: the family tag and the launch's algorithmic work (ipoke_timing_stop_ex)
  void annotate(int tag, double flops, double bytes) { if (slot >= 0) timing_annotate(slot, tag, flops, bytes); }
};

// ---- element traits --------------------------------------------------------
template <typename T> struct ET;
template <> struct ET<bf16_t> {
  static constexpr int E16 = 8;           // elements per 16-byte chunk
  typedef bf16x8 frag;
  static __device__ __forceinline__ bf16_t from_f32(float x) { return (bf16_t)x; }
  static __device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
};
template <> struct ET<float> {
  static constexpr int E16 = 4;
  typedef f32x4 frag;
  static __device__ __forceinline__ float from_f32(float x) { return x; }
  static __device__ __forceinline__ float to_f32(float x) { return x; }
};

// One 64-byte K "super-step" of a 16x16 output fragment.
//   bf16: one v_mfma_f32_16x16x32_bf16 (lane group g = lane>>4 holds k = 8g..8g+7)
//   f32 : four v_mfma_f32_16x16x4_f32; lane group g holds the 4 contiguous floats k = 4g..4g+3 and
//         instruction c uses component c -- a K permutation applied identically to both operands.
// Operand order (b, a) yields D[n][m]: the lane owning column m = lane&15 holds 4 consecutive n
// (n = 4*(lane>>4)+r), i.e. row-major C gets 4 contiguous outputs per lane.
__device__ __forceinline__ void mma64(const bf16x8& a, const bf16x8& b, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma64(const f32x4& a, const f32x4& b, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[0], a[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[1], a[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[2], a[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[3], a[3], acc, 0, 0, 0);
}

// ---- activations -----------------------------------------------------------
__device__ __forceinline__ float act_apply(int act, float x) {
  switch (act) {
    case IPOKE_ACT_ELU: return x > 0.f ? x : expm1f(x);
    case IPOKE_ACT_RELU: return x > 0.f ? x : 0.f;
    case IPOKE_ACT_LRELU02: return x > 0.f ? x : 0.2f * x;
    case IPOKE_ACT_TANH: return tanhf(x);
    case IPOKE_ACT_SIGMOID: return 1.f / (1.f + expf(-x));
    default: return x;
  }
}
// derivative expressed through the activation OUTPUT y
__device__ __forceinline__ float act_grad_from_out(int act, float y) {
  switch (act) {
    case IPOKE_ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
    case IPOKE_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case IPOKE_ACT_LRELU02: return y > 0.f ? 1.f : 0.2f;
    case IPOKE_ACT_TANH: return 1.f - y * y;
    case IPOKE_ACT_SIGMOID: return y * (1.f - y);
    default: return 1.f;
  }
}
// derivative expressed through the pre-activation value f (the same function: act_grad_from_out(act, act_apply(act, f)))
__device__ __forceinline__ float act_grad_from_pre(int act, float f) {
  switch (act) {
    case IPOKE_ACT_ELU: return f > 0.f ? 1.f : expm1f(f) + 1.f;
    case IPOKE_ACT_RELU: return f > 0.f ? 1.f : 0.f;
    case IPOKE_ACT_LRELU02: return f > 0.f ? 1.f : 0.2f;
    case IPOKE_ACT_NONE: return 1.f;
    default: return act_grad_from_out(act, act_apply(act, f));
  }
}

// ---- kernel-argument prefetch ------------------------------------------------
#ifndef IPK_KERNARG_PREFETCH
#define IPK_KERNARG_PREFETCH 1      // (0: developer A/B build)
#endif
// hipcc fetches kernel arguments lazily (an s_load next to each first use, each behind its own s_waitcnt).  The argument block of a launch
// is cold -- the command processor has just written it -- so every first touch of a 64-byte line is a memory round trip (~0.3 us) and a
// kernel with a few hundred bytes of arguments starts with a CHAIN of them.  One dword of every line of the first BYTES bytes requested
// back to back at entry: one round trip; the compiler's own loads then hit the scalar cache.  Up to 8 lines (512 bytes); BYTES must not
// exceed the size of the kernel's explicit arguments.
template <int BYTES>
__device__ __forceinline__ void kernarg_prefetch() {
  static_assert(BYTES >= 4, "at least one dword");
  constexpr int last = ((BYTES - 4) / 64) * 64;
  constexpr int o1 = 64 < last ? 64 : last, o2 = 128 < last ? 128 : last, o3 = 192 < last ? 192 : last, o4 = 256 < last ? 256 : last,
                o5 = 320 < last ? 320 : last, o6 = 384 < last ? 384 : last, o7 = 448 < last ? 448 : last;
  const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
  unsigned d0, d1, d2, d3, d4, d5, d6, d7;
  asm volatile(
      "s_load_dword %0, %8, 0\n\ts_load_dword %1, %8, %9\n\ts_load_dword %2, %8, %10\n\ts_load_dword %3, %8, %11\n\t"
      "s_load_dword %4, %8, %12\n\ts_load_dword %5, %8, %13\n\ts_load_dword %6, %8, %14\n\ts_load_dword %7, %8, %15\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7)
      : "s"(ka), "n"(o1), "n"(o2), "n"(o3), "n"(o4), "n"(o5), "n"(o6), "n"(o7)
      : "memory");
}

// ---- division by a kernel argument --------------------------------------------------------------------------------------------
// x / d and x % d for a launch-uniform divisor: a shift / mask when d is a power of two (the state widths, channel counts and strides
// of the shipped flows are), the compiler's ~25-instruction division sequence otherwise.  The test on `sh` is scalar; the element-wise
// kernels of the chain spent 1-3 us per launch on index arithmetic before this.  Dividends are non-negative.
#ifndef IPK_FDIV
#define IPK_FDIV 1      // 0: always divide (developer A/B)
#endif
struct FDiv {
  int d, sh;
  __device__ __forceinline__ explicit FDiv(int d_) : d(d_), sh(IPK_FDIV && d_ > 0 && (d_ & (d_ - 1)) == 0 ? __builtin_ctz((unsigned)d_) : -1) {}
  __device__ __forceinline__ int div(int x) const { return sh >= 0 ? x >> sh : x / d; }
  __device__ __forceinline__ int mod(int x) const { return sh >= 0 ? x & (d - 1) : x % d; }
  __device__ __forceinline__ long div(long x) const { return sh >= 0 ? x >> sh : x / d; }
  __device__ __forceinline__ int mod(long x) const { return sh >= 0 ? (int)(x & (long)(d - 1)) : (int)(x % d); }
  // rel >= 0, a multiple of d, and rel / d < n  (the "is column rel one of the n strided channels" test)
  __device__ __forceinline__ bool strided_hit(int rel, int n) const { return rel >= 0 && mod(rel) == 0 && div(rel) < n; }
};

// ---- reductions ------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// block-wide sum; `red` is >= (blockDim.x/64) floats of LDS. All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// ---- Adam with amsgrad (torch.optim.Adam semantics), one element; shared by every kernel that applies the update
struct AdamHyper { float lr_bc1, beta1, beta2, eps, wd, bc2_sqrt, grad_scale; };     // lr_bc1 = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t)
static inline AdamHyper adam_make_hyper(float lr, float beta1, float beta2, float eps, float wd, int step, float grad_scale) {
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  return AdamHyper{lr / (float)bc1, beta1, beta2, eps, wd, (float)sqrt(bc2), grad_scale};
}
// Every product-sum is an explicit fmaf and contraction is off inside the function, so that each kernel that applies the update
// (linear, per segment, tile-wise with the shadow writes) produces bit-identical parameters whatever the compiler would have fused.
__device__ __forceinline__ void adam_amsgrad_update(float& p, float g, float& m, float& v, float& vx, const AdamHyper& h) {
#pragma clang fp contract(off)
  const float gr = fmaf(h.wd, p, g * h.grad_scale);
  m = fmaf(h.beta1, m, (1.f - h.beta1) * gr);
  v = fmaf(h.beta2, v, ((1.f - h.beta2) * gr) * gr);
  vx = fmaxf(vx, v);
  const float denom = sqrtf(vx) / h.bc2_sqrt + h.eps;
  p = fmaf(-h.lr_bc1, m / denom, p);
}
__device__ __forceinline__ void adam_amsgrad_update4(f32x4& p, const f32x4& g, f32x4& m, f32x4& v, f32x4& vx, const AdamHyper& h) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float pk = p[k], mk = m[k], vk = v[k], xk = vx[k];
    adam_amsgrad_update(pk, g[k], mk, vk, xk, h);
    p[k] = pk; m[k] = mk; v[k] = vk; vx[k] = xk;
  }
}

struct McfGeom { int kh, kw, oy, ox; };
__host__ __device__ inline McfGeom mcf_geom(int order) {
  // input(y + ky + oy, x + kx + ox) feeds output (y, x): strictly above / below / left / right
  switch (order) {
    case 0: return {2, 3, -2, -1};   // A
    case 1: return {2, 3, +1, -1};   // B
    case 2: return {3, 2, -1, -2};   // C
    default: return {3, 2, -1, +1};  // D
  }
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
static inline int ilog2_exact(int v) {   // -1 when v is not a power of two
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

}  // namespace ipoke
