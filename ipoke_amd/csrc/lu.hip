// LU-parametrised invertible 1x1 convolution (reference models/modules/INN/macow2.py:596-649; selected by `use1x1` for the
// per-level shuffle layers, :862):
//     W = P (L * lmask + I) (U * umask + diag(sign_s exp(log_s)))        y[:, p] = W x[:, p]
//     log-det = H W sum(log_s)      inverse: W^-1 = wu^-1 wl^-1 P^-1 (three torch.inverse calls in the reference)
// C <= 64 channels on an 8x8 latent: this is a few MFLOP per layer and 15 layers per flow -- plain fp32 FMA kernels (exact
// fp32 like the reference; the matrix cores would add nothing but bf16 rounding), one workgroup per matrix / per sample.
// HBM-bound by construction: the state is read and written once per application.
#include "common.h"

namespace ipoke {

struct LuJob {
  long p_l, p_u, p_logs;                               // floats into the parameter buffer
  long b_perm, b_sign, b_lmask, b_umask, b_eye;        // floats into the float-buffer table
  long w_off;                                          // floats into the workspace: [W | Winv | wl | wu], C*C each
  int C, pad;
};

static constexpr int kLuMax = 64;
static constexpr int kLuP = kLuMax + 1;                // LDS row pitch

// one workgroup per matrix: builds wl, wu, W = P wl wu and W^-1 = wu^-1 wl^-1 P^T
__global__ __launch_bounds__(256) void lu_prepare_kernel(const float* __restrict__ params, const float* __restrict__ fbuf,
                                                         float* __restrict__ ws, const LuJob* __restrict__ jobs) {
  const LuJob j0 = jobs[blockIdx.x];
  const LuJob& j = j0;
  __shared__ float A[kLuMax * kLuP], Bm[kLuMax * kLuP], T[kLuMax * kLuP];          // 3 x 16.6 KB (static LDS limit 64 KB)
  const float* Pm = fbuf + j0.b_perm;                                               // P is read in place
  const int C = j.C, tid = threadIdx.x;
  float* out = ws + j.w_off;
  for (int e = tid; e < C * C; e += 256) {
    const int r = e / C, c = e - r * C;
    const float wl = params[j.p_l + e] * fbuf[j.b_lmask + e] + fbuf[j.b_eye + e];
    float wu = params[j.p_u + e] * fbuf[j.b_umask + e];
    if (r == c) wu += fbuf[j.b_sign + r] * expf(params[j.p_logs + r]);
    A[r * kLuP + c] = wl; Bm[r * kLuP + c] = wu;
    out[2 * C * C + e] = wl; out[3 * C * C + e] = wu;
  }
  __syncthreads();
  for (int e = tid; e < C * C; e += 256) {            // T = wl wu
    const int r = e / C, c = e - r * C;
    float s = 0.f;
    for (int k = 0; k < C; ++k) s = fmaf(A[r * kLuP + k], Bm[k * kLuP + c], s);
    T[r * kLuP + c] = s;
  }
  __syncthreads();
  for (int e = tid; e < C * C; e += 256) {            // W = P T
    const int r = e / C, c = e - r * C;
    float s = 0.f;
    for (int k = 0; k < C; ++k) s = fmaf(Pm[r * C + k], T[k * kLuP + c], s);
    out[e] = s;
  }
  __syncthreads();
  // triangular inverses, one column per thread: wl (unit lower) by forward, wu (upper) by backward substitution; results
  // overwrite T (wl^-1); wu^-1 goes to A after wl has been consumed
  if (tid < C) {
    const int col = tid;
    for (int i = 0; i < C; ++i) {
      float s = i == col ? 1.f : 0.f;
      for (int k = 0; k < i; ++k) s = fmaf(-A[i * kLuP + k], T[k * kLuP + col], s);
      T[i * kLuP + col] = s;                           // A's diagonal is 1
    }
  }
  __syncthreads();
  if (tid < C) {
    const int col = tid;
    for (int i = C - 1; i >= 0; --i) {
      float s = i == col ? 1.f : 0.f;
      for (int k = i + 1; k < C; ++k) s = fmaf(-Bm[i * kLuP + k], A[k * kLuP + col], s);
      A[i * kLuP + col] = s / Bm[i * kLuP + i];        // A now holds wu^-1 (rows > i of this column are final)
    }
  }
  __syncthreads();
  for (int e = tid; e < C * C; e += 256) {            // Bm = wu^-1 wl^-1
    const int r = e / C, c = e - r * C;
    float s = 0.f;
    for (int k = 0; k < C; ++k) s = fmaf(A[r * kLuP + k], T[k * kLuP + c], s);
    Bm[r * kLuP + c] = s;
  }
  __syncthreads();
  for (int e = tid; e < C * C; e += 256) {            // W^-1 = (wu^-1 wl^-1) P^T   (P is a permutation matrix)
    const int r = e / C, c = e - r * C;
    float s = 0.f;
    for (int k = 0; k < C; ++k) s = fmaf(Bm[r * kLuP + k], Pm[c * C + k], s);
    out[C * C + e] = s;
  }
}

// out[m][i] = sum_j mat[i][j] in[m][j]  (transposed: mat[j][i]) on the first C columns of one sample; the rest is copied
__global__ __launch_bounds__(256) void lu_apply_kernel(const float* __restrict__ in, float* __restrict__ out, long M, int ld, int C,
                                                       const float* __restrict__ mat, int transposed) {
  __shared__ float Ws[kLuMax * kLuP], X[64 * kLuP];
  const int tid = threadIdx.x;
  const long row0 = (long)blockIdx.x * 64;
  for (int e = tid; e < C * C; e += 256) {
    const int r = e / C, c = e - r * C;
    Ws[(transposed ? c : r) * kLuP + (transposed ? r : c)] = mat[e];
  }
  const int rows = (int)(M - row0 < 64 ? M - row0 : 64);        // the last workgroup of a row count that is not a multiple of 64
  for (int e = tid; e < rows * ld; e += 256) {
    const int m = e / ld, c = e - m * ld;
    const float v = in[(row0 + m) * ld + c];
    if (c < C) X[m * kLuP + c] = v; else out[(row0 + m) * ld + c] = v;
  }
  __syncthreads();
  for (int e = tid; e < rows * C; e += 256) {
    const int m = e / C, i = e - m * C;
    float s = 0.f;
    for (int k = 0; k < C; ++k) s = fmaf(Ws[i * kLuP + k], X[m * kLuP + k], s);
    out[(row0 + m) * ld + i] = s;
  }
}

// parameter gradients of one layer: dW = dy^T x over all M rows, then
//   dl = (P^T dW wu^T) * lmask,  du = (wl^T P^T dW) * umask,  dlog_s[i] = (wl^T P^T dW)[i][i] sign_s[i] exp(log_s[i]) + P8 sum_b dld[b]
__global__ __launch_bounds__(256) void lu_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int M, int ld,
                                                       const float* __restrict__ params, const float* __restrict__ fbuf,
                                                       const float* __restrict__ ws, const LuJob* __restrict__ job,
                                                       const float* __restrict__ dld, int B, int P8, float* __restrict__ grads) {
  __shared__ float G[kLuMax * kLuP], Ta[64 * kLuP], Tb[64 * kLuP];               // 3 x 16.6 KB
  const LuJob j = *job;
  const int C = j.C, tid = threadIdx.x;
  const int per = (C * C + 255) / 256;                 // outputs per thread (<= 16)
  float acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  for (int m0 = 0; m0 < M; m0 += 64) {
    __syncthreads();
    for (int e = tid; e < 64 * C; e += 256) {
      const int m = e / C, c = e - m * C;
      const bool ok = m0 + m < M;
      Ta[m * kLuP + c] = ok ? dy[(long)(m0 + m) * ld + c] : 0.f;
      Tb[m * kLuP + c] = ok ? x[(long)(m0 + m) * ld + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = tid + q * 256;
      if (q < per && e < C * C) {
        const int i = e / C, k = e - i * C;
        float s = acc[q];
        for (int m = 0; m < 64; ++m) s = fmaf(Ta[m * kLuP + i], Tb[m * kLuP + k], s);
        acc[q] = s;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int e = tid + q * 256;
    if (q < per && e < C * C) G[(e / C) * kLuP + e % C] = acc[q];          // dW
  }
  __syncthreads();
  const float* Pm = fbuf + j.b_perm;
  float* Q = Ta;                                        // the row tiles are dead
  for (int e = tid; e < C * C; e += 256) {             // Q = P^T dW
    const int r = e / C, c = e - r * C;
    float s = 0.f;
    for (int k = 0; k < C; ++k) s = fmaf(Pm[k * C + r], G[k * kLuP + c], s);
    Q[r * kLuP + c] = s;
  }
  __syncthreads();
  const float* wl = ws + j.w_off + 2 * C * C;
  const float* wu = ws + j.w_off + 3 * C * C;
  float* WL = Tb; float* WU = G;                        // dW is consumed
  for (int e = tid; e < C * C; e += 256) { WL[(e / C) * kLuP + e % C] = wl[e]; WU[(e / C) * kLuP + e % C] = wu[e]; }
  __syncthreads();
  float dsum = 0.f;
  for (int b = 0; b < B; ++b) dsum += dld[b];
  for (int e = tid; e < C * C; e += 256) {
    const int r = e / C, c = e - r * C;
    float sl = 0.f, su = 0.f;
    for (int k = 0; k < C; ++k) {
      sl = fmaf(Q[r * kLuP + k], WU[c * kLuP + k], sl);        // (Q wu^T)[r][c]
      su = fmaf(WL[k * kLuP + r], Q[k * kLuP + c], su);        // (wl^T Q)[r][c]
    }
    grads[j.p_l + e] = sl * fbuf[j.b_lmask + e];
    grads[j.p_u + e] = su * fbuf[j.b_umask + e];
    if (r == c) grads[j.p_logs + r] = su * fbuf[j.b_sign + r] * expf(params[j.p_logs + r]) + (float)P8 * dsum;
  }
}

}  // namespace ipoke

using namespace ipoke;

extern "C" int ipoke_lu_job_size(void) { return (int)sizeof(LuJob); }

extern "C" int ipoke_lu_prepare(const float* params, const float* fbuf, float* workspace, const void* jobs_dev, int njobs,
                                void* stream) {
  IPK_REQUIRE(params && fbuf && workspace && jobs_dev && njobs >= 1, "bad arguments");
  hipLaunchKernelGGL(lu_prepare_kernel, dim3(njobs), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), params, fbuf, workspace,
                     reinterpret_cast<const LuJob*>(jobs_dev));
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

extern "C" int ipoke_lu_apply(const float* in, float* out, int64_t M, int ld, int C, const float* mat, int transposed, void* stream) {
  IPK_REQUIRE(in && out && mat && M >= 1 && M < (1L << 37) && C >= 1 && C <= kLuMax && ld >= C && in != out, "bad arguments");
  hipLaunchKernelGGL(lu_apply_kernel, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, out, (long)M, ld, C,
                     mat, transposed);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

extern "C" int ipoke_lu_wgrad(const float* dy, const float* x, int B, int P8, int ld, const float* params, const float* fbuf,
                              const float* workspace, const void* job_dev, const float* dld, float* grads, void* stream) {
  IPK_REQUIRE(dy && x && params && fbuf && workspace && job_dev && dld && grads && B >= 1, "bad arguments");
  hipLaunchKernelGGL(lu_wgrad_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, x, B * P8, ld, params, fbuf,
                     workspace, reinterpret_cast<const LuJob*>(job_dev), dld, B, P8, grads);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
