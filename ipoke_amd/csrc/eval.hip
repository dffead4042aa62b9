// Evaluation-side kernels: the FVD metric's I3D network around the implicit-GEMM convolutions (reference utils/metrics.py:
// preprocess :787-800, MaxPool3dTFPadding :939-960, I3D head :1085-1096, activation moments :733-771).
// All of them stream channels-last rows once: HBM-bound element-wise / window work, 16-byte accesses along the channels.
#include "common.h"

#include <cmath>
#include <cstring>

using namespace ipoke;
#define STREAM(s) reinterpret_cast<hipStream_t>(s)
static int grid1(long n, int cap = 4096) { long g = (n + 255) / 256; if (g < 1) g = 1; if (g > cap) g = cap; return (int)g; }

// ---------------------------------------------------------------------------------------------- clip -> padded channels-last rows
// float minimum through integer atomics: non-negative floats order like ints, negative ones reversed as unsigned
__device__ __forceinline__ void atomic_min_f32(float* addr, float v) {
  if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
  else if (v < 0.f) atomicMax(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}

// Bilinear resize with align_corners=True of every frame of a strided fp32 clip tensor (frame f of clip n at
// n*s_n + f*s_f, channel c at c*s_c, pixel (y, x) at y*s_h + x*s_w) to Ho x Wo, written as channels-last rows
// dst[(frame*Ho + y)*Wp + pad_l + x][C] with zero columns left and right (the I3D stem reads a 7-pixel window of a row as
// one 21-channel tap, so the TF-SAME zero padding along W is stored).  dst == nullptr: only the minimum is taken.
__global__ void video_to_cl_kernel(const float* __restrict__ src, long s_n, long s_f, long s_c, long s_h, long s_w, int T, int C, int Hi, int Wi,
                                   float* __restrict__ dst, long frames, int Ho, int Wo, int pad_l, int Wp, float* __restrict__ minval) {
  const long total = frames * Ho * Wp;
  const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  float lo = INFINITY;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xp = (int)(i % Wp); long t = i / Wp;
    const int oy = (int)(t % Ho); const long fr = t / Ho;
    const int ox = xp - pad_l;
    if (ox < 0 || ox >= Wo) {
      if (dst) for (int c = 0; c < C; ++c) dst[i * C + c] = 0.f;
      continue;
    }
    const float fy = oy * sy, fx = ox * sx;
    const int y0 = min((int)fy, Hi - 1), x0 = min((int)fx, Wi - 1);
    const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const float* p = src + (fr / T) * s_n + (fr % T) * s_f;
    for (int c = 0; c < C; ++c) {
      const float* q = p + c * s_c;
      const float v = (1.f - wy) * ((1.f - wx) * q[y0 * s_h + x0 * s_w] + wx * q[y0 * s_h + x1 * s_w]) +
                      wy * ((1.f - wx) * q[y1 * s_h + x0 * s_w] + wx * q[y1 * s_h + x1 * s_w]);
      if (dst) dst[i * C + c] = v;
      lo = fminf(lo, v);
    }
  }
  if (minval) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lo = fminf(lo, __shfl_xor(lo, o, 64));
    if ((threadIdx.x & 63) == 0) atomic_min_f32(minval, lo);
  }
}
// (x + 1) / 2 on the interior columns when the minimum over the whole data set was negative (metrics.py:794-798)
__global__ void denorm_if_negative_kernel(float* __restrict__ x, long rows, int Wo, int pad_l, int Wp, int C, const float* __restrict__ minval) {
  if (!(*minval < 0.f)) return;
  const long total = rows * Wo * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long t = i / C;
    const int ox = (int)(t % Wo); const long r = t / Wo;
    float* p = x + (r * Wp + pad_l + ox) * C + c;
    *p = (*p + 1.0f) / 2.0f;
  }
}

// ---------------------------------------------------------------------------------------------- max pooling, TF "SAME"
// The reference pads with ZEROS and then pools with ceil_mode: a window position inside the zero border contributes 0, one
// beyond it (ceil_mode overhang) nothing.  e* = input extent + back padding.  One thread = one 16-byte channel chunk of an
// output position.
struct PoolSame { int N, C, Di, Hi, Wi, Do, Ho, Wo, kd, kh, kw, sd, sh, sw, pd, ph, pw, ed, eh, ew; };
template <typename T>
__global__ void pool3d_same_kernel(PoolSame g, const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy) {
  constexpr int E = ET<T>::E16;
  typedef typename ET<T>::frag frag;
  const int chunks = g.C / E;
  const long total = (long)g.N * g.Do * g.Ho * g.Wo * chunks;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % chunks); const long r = i / chunks;
    const int ow = (int)(r % g.Wo); long t = r / g.Wo;
    const int oh = (int)(t % g.Ho); t /= g.Ho;
    const int od = (int)(t % g.Do); const int n = (int)(t / g.Do);
    float best[E];
#pragma unroll
    for (int e = 0; e < E; ++e) best[e] = -INFINITY;
    for (int a = 0; a < g.kd; ++a) {
      const int d = od * g.sd - g.pd + a; if (d >= g.ed) continue;
      for (int b = 0; b < g.kh; ++b) {
        const int h = oh * g.sh - g.ph + b; if (h >= g.eh) continue;
        for (int c = 0; c < g.kw; ++c) {
          const int w = ow * g.sw - g.pw + c; if (w >= g.ew) continue;
          if ((unsigned)d < (unsigned)g.Di && (unsigned)h < (unsigned)g.Hi && (unsigned)w < (unsigned)g.Wi) {
            const long row = ((long)(n * g.Di + d) * g.Hi + h) * g.Wi + w;
            const frag v = *reinterpret_cast<const frag*>(x + row * ldx + ch * E);
#pragma unroll
            for (int e = 0; e < E; ++e) best[e] = fmaxf(best[e], (float)v[e]);
          } else {
#pragma unroll
            for (int e = 0; e < E; ++e) best[e] = fmaxf(best[e], 0.f);
          }
        }
      }
    }
    frag o;
#pragma unroll
    for (int e = 0; e < E; ++e) o[e] = ET<T>::from_f32(best[e]);
    *reinterpret_cast<frag*>(y + r * ldy + ch * E) = o;
  }
}

// y[g][c] = sum_k w[k] * x[g*S + k][c]: AvgPool3d((2,7,7), 1) followed by the mean over the remaining time steps is one
// weighted mean over a clip's rows (the 1x1 logits convolution is linear and commutes with it)
template <typename T>
__global__ void pool_rows_weighted_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, long G, int S, int C, const float* __restrict__ w) {
  const long total = G * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); const long gi = i / C;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += w[k] * ET<T>::to_f32(x[(gi * S + k) * ldx + c]);
    y[gi * ldy + c] = ET<T>::from_f32(s);
  }
}

// ---------------------------------------------------------------------------------------------- activation moments (float64)
// rows without a single non-NaN entry are dropped (metrics.py:765-767); mean, then the unbiased covariance as np.cov
__global__ void moments_valid_kernel(const float* __restrict__ a, int n, int D, int* __restrict__ valid, int* __restrict__ count) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int any = 0;
  for (int c = 0; c < D; ++c) any |= !(a[(long)r * D + c] != a[(long)r * D + c]);
  valid[r] = any;
  if (any) atomicAdd(count, 1);
}
__global__ void moments_mean_kernel(const float* __restrict__ a, int n, int D, const int* __restrict__ valid, const int* __restrict__ count, double* __restrict__ mu) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  double s = 0.0;
  for (int r = 0; r < n; ++r) if (valid[r]) s += (double)a[(long)r * D + c];
  mu[c] = s / (double)(*count);
}
__global__ void moments_cov_kernel(const float* __restrict__ a, int n, int D, const int* __restrict__ valid, const int* __restrict__ count,
                                   const double* __restrict__ mu, double* __restrict__ sigma) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= D) return;
  const double mi = mu[i], mj = mu[j];
  double s = 0.0;
  for (int r = 0; r < n; ++r) if (valid[r]) s += ((double)a[(long)r * D + i] - mi) * ((double)a[(long)r * D + j] - mj);
  sigma[(long)i * D + j] = s / (double)(*count - 1);
}
// ------------------------------------------------------------------------------------------------ image metrics
// pytorch_lightning.metrics.functional.psnr / ssim as the reference's validation logging calls them (second_stage_video.py:511-512,
// metrics.py:450-481: defaults -- 11 x 11 Gaussian window of sigma 1.5, k1 = 0.01, k2 = 0.03, data ranges taken from the tensors).
// mm[0..3] = {-min(a), max(a), -min(b), max(b)} (one atomic flavour); sums in double.
// Chosen by the SIGN BIT, not by `v >= 0`: -0.0f (the negated minimum of an image whose darkest pixel is exactly 0) compares >= 0 but its
// bit pattern is INT_MIN, which as a signed integer never beats the 0xffffffff initial value.
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {
  if (__float_as_int(v) >= 0) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__global__ void pair_stats_kernel(const float* __restrict__ a, const float* __restrict__ b, long n, float* __restrict__ mm, double* __restrict__ sse) {
  float amin = INFINITY, amax = -INFINITY, bmin = INFINITY, bmax = -INFINITY;
  double acc = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = a[i], y = b[i], d = x - y;
    amin = fminf(amin, x); amax = fmaxf(amax, x); bmin = fminf(bmin, y); bmax = fmaxf(bmax, y);
    acc += (double)d * (double)d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    amin = fminf(amin, __shfl_xor(amin, o, 64)); amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    bmin = fminf(bmin, __shfl_xor(bmin, o, 64)); bmax = fmaxf(bmax, __shfl_xor(bmax, o, 64));
    acc += __shfl_xor(acc, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomic_max_f32(mm + 0, -amin); atomic_max_f32(mm + 1, amax); atomic_max_f32(mm + 2, -bmin); atomic_max_f32(mm + 3, bmax);
    atomicAdd(sse, acc);
  }
}
__global__ void psnr_final_kernel(const float* __restrict__ mm, const double* __restrict__ sse, long n, float* __restrict__ out) {
  // psnr = 10 log10(range^2 / mse), range = max(target) - min(target)   (functional/psnr.py: data_range=None, base 10)
  const double range = (double)mm[3] + (double)mm[2];
  const double mse = sse[0] / (double)n;
  out[0] = (float)(10.0 * (2.0 * log(range) - log(mse)) / log(10.0));
}
// SSIM map of one plane tile: 32 x 32 output positions whose 11 x 11 windows lie inside the image (the reference pads by reflection and
// crops the padded border away again, functional/ssim.py), separable Gaussian of the five moments, per-block partial sum in double.
__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, int H, int W, const float* __restrict__ mm,
                                                   double* __restrict__ part) {
  constexpr int TS = 32, K = 11, IN = TS + K - 1;           // 42 x 42 inputs per tile
  __shared__ float sa[IN][IN + 1], sb[IN][IN + 1];
  __shared__ float hz[5][IN][TS + 1];                       // horizontally filtered moments
  __shared__ float gw[K];
  __shared__ double red[4];
  const int Ho = H - (K - 1), Wo = W - (K - 1);
  const int tiles_x = (Wo + TS - 1) / TS;
  const int plane = blockIdx.y, ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * TS, x0 = tx * TS;
  const float* pa = a + (long)plane * H * W;
  const float* pb = b + (long)plane * H * W;
  if (threadIdx.x < K) {
    float sum = 0.f, mine = 0.f;
    for (int i = 0; i < K; ++i) {
      const float d = (float)(i - K / 2) / 1.5f, gi = expf(-d * d / 2.f);
      sum += gi;
      if (i == (int)threadIdx.x) mine = gi;
    }
    gw[threadIdx.x] = mine / sum;
  }
  for (int i = threadIdx.x; i < IN * IN; i += 256) {
    const int r = i / IN, c = i - r * IN, y = y0 + r, x = x0 + c;
    const bool in = y < H && x < W;
    sa[r][c] = in ? pa[(long)y * W + x] : 0.f;
    sb[r][c] = in ? pb[(long)y * W + x] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < IN * TS; i += 256) {
    const int r = i / TS, c = i - r * TS;
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float w = gw[k], x = sa[r][c + k], y = sb[r][c + k];
      m0 += w * x; m1 += w * y; m2 += w * x * x; m3 += w * y * y; m4 += w * x * y;
    }
    hz[0][r][c] = m0; hz[1][r][c] = m1; hz[2][r][c] = m2; hz[3][r][c] = m3; hz[4][r][c] = m4;
  }
  __syncthreads();
  const float range = fmaxf(mm[1] + mm[0], mm[3] + mm[2]);   // max(preds.max() - preds.min(), target.max() - target.min())
  const float c1 = (0.01f * range) * (0.01f * range), c2 = (0.03f * range) * (0.03f * range);
  double acc = 0.0;
  for (int i = threadIdx.x; i < TS * TS; i += 256) {
    const int r = i / TS, c = i - r * TS;
    if (y0 + r >= Ho || x0 + c >= Wo) continue;
    float mu_a = 0.f, mu_b = 0.f, e_aa = 0.f, e_bb = 0.f, e_ab = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float w = gw[k];
      mu_a += w * hz[0][r + k][c]; mu_b += w * hz[1][r + k][c]; e_aa += w * hz[2][r + k][c]; e_bb += w * hz[3][r + k][c]; e_ab += w * hz[4][r + k][c];
    }
    const float maa = mu_a * mu_a, mbb = mu_b * mu_b, mab = mu_a * mu_b;
    const float upper = 2.f * (e_ab - mab) + c2, lower = (e_aa - maa) + (e_bb - mbb) + c2;
    acc += (double)(((2.f * mab + c1) * upper) / ((maa + mbb + c1) * lower));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[(long)plane * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void ssim_final_kernel(const double* __restrict__ part, long nparts, double count, float* __restrict__ out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (long i = threadIdx.x; i < nparts; i += 256) acc += part[i];      // fixed order: reproducible
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = (float)(red[0] / count);
}



// ==============================================================================================
extern "C" int ipoke_video_to_cl(const float* src, int64_t s_n, int64_t s_f, int64_t s_c, int64_t s_h, int64_t s_w, int N, int T, int C, int Hi,
                                 int Wi, float* dst, int Ho, int Wo, int pad_l, int pad_r, float* minval, void* stream) {
  IPK_REQUIRE(src && (dst || minval) && N >= 1 && T >= 1 && C >= 1 && Hi >= 1 && Wi >= 1 && Ho >= 1 && Wo >= 1 && pad_l >= 0 && pad_r >= 0, "bad arguments");
  const long frames = (long)N * T;
  const int Wp = pad_l + Wo + pad_r;
  hipLaunchKernelGGL(video_to_cl_kernel, dim3(grid1(frames * Ho * Wp)), dim3(256), 0, STREAM(stream), src, (long)s_n, (long)s_f, (long)s_c, (long)s_h,
                     (long)s_w, T, C, Hi, Wi, dst, frames, Ho, Wo, pad_l, Wp, minval);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
/* *minval = a huge float: the start value of the running minimum of ipoke_video_to_cl */
extern "C" int ipoke_min_reset(float* minval, void* stream) {
  IPK_REQUIRE(minval, "null argument");
  IPK_HIP(hipMemsetAsync(minval, 0x7f, sizeof(float), STREAM(stream)));
  return IPOKE_OK;
}
extern "C" int ipoke_denorm_if_negative(float* x, int64_t rows, int Wo, int pad_l, int pad_r, int C, const float* minval, void* stream) {
  IPK_REQUIRE(x && minval && rows >= 1 && Wo >= 1 && C >= 1, "bad arguments");
  hipLaunchKernelGGL(denorm_if_negative_kernel, dim3(grid1(rows * Wo * C)), dim3(256), 0, STREAM(stream), x, (long)rows, Wo, pad_l, pad_l + Wo + pad_r, C, minval);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

extern "C" int ipoke_pool3d_same(const int* dims, const void* x, int ldx, void* y, int ldy, int dtype, void* stream) {
  IPK_REQUIRE(dims && x && y, "null argument");
  PoolSame g;
  std::memcpy(&g, dims, sizeof(g));
  const int e16 = dtype == IPOKE_BF16 ? 8 : 4;
  IPK_REQUIRE(dtype == IPOKE_BF16 || dtype == IPOKE_F32, "bad dtype");
  IPK_REQUIRE(g.N >= 1 && g.C >= 1 && g.Do >= 1 && g.Ho >= 1 && g.Wo >= 1 && g.kd >= 1 && g.kh >= 1 && g.kw >= 1, "bad pool geometry");
  IPK_REQUIRE(g.C % e16 == 0 && ldx % e16 == 0 && ldy % e16 == 0 && ldx >= g.C && ldy >= g.C, "channels and pitches: multiples of 16 bytes");
  IPK_REQUIRE(g.ed >= g.Di && g.eh >= g.Hi && g.ew >= g.Wi && g.pd >= 0 && g.ph >= 0 && g.pw >= 0, "padded extents must cover the input");
  // every window must see at least its first position inside the padded extent (ceil_mode never starts a window in the overhang)
  IPK_REQUIRE((g.Do - 1) * g.sd - g.pd < g.ed && (g.Ho - 1) * g.sh - g.ph < g.eh && (g.Wo - 1) * g.sw - g.pw < g.ew, "window starts beyond the padded extent");
  const long total = (long)g.N * g.Do * g.Ho * g.Wo * (g.C / e16);
  if (dtype == IPOKE_BF16)
    hipLaunchKernelGGL(pool3d_same_kernel<bf16_t>, dim3(grid1(total, 8192)), dim3(256), 0, STREAM(stream), g, (const bf16_t*)x, ldx, (bf16_t*)y, ldy);
  else
    hipLaunchKernelGGL(pool3d_same_kernel<float>, dim3(grid1(total, 8192)), dim3(256), 0, STREAM(stream), g, (const float*)x, ldx, (float*)y, ldy);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

extern "C" int ipoke_pool_rows_weighted(const void* x, int ldx, void* y, int ldy, int64_t G, int S, int C, const float* w, int dtype, void* stream) {
  IPK_REQUIRE(x && y && w && G >= 1 && S >= 1 && C >= 1 && ldx >= C && ldy >= C, "bad arguments");
  if (dtype == IPOKE_BF16)
    hipLaunchKernelGGL(pool_rows_weighted_kernel<bf16_t>, dim3(grid1(G * C)), dim3(256), 0, STREAM(stream), (const bf16_t*)x, ldx, (bf16_t*)y, ldy, (long)G, S, C, w);
  else
    hipLaunchKernelGGL(pool_rows_weighted_kernel<float>, dim3(grid1(G * C)), dim3(256), 0, STREAM(stream), (const float*)x, ldx, (float*)y, ldy, (long)G, S, C, w);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

/* workspace: (n + 1) int32 */
extern "C" int ipoke_activation_moments(const float* act, int n, int D, double* mu, double* sigma, int* workspace, void* stream) {
  IPK_REQUIRE(act && mu && sigma && workspace && n >= 2 && D >= 1, "bad arguments");
  int* count = workspace; int* valid = workspace + 1;
  IPK_HIP(hipMemsetAsync(count, 0, sizeof(int), STREAM(stream)));
  hipLaunchKernelGGL(moments_valid_kernel, dim3((n + 255) / 256), dim3(256), 0, STREAM(stream), act, n, D, valid, count);
  hipLaunchKernelGGL(moments_mean_kernel, dim3((D + 63) / 64), dim3(64), 0, STREAM(stream), act, n, D, valid, count, mu);
  hipLaunchKernelGGL(moments_cov_kernel, dim3((D + 63) / 64, D), dim3(64), 0, STREAM(stream), act, n, D, valid, count, mu, sigma);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

/* Workspace of ipoke_psnr_ssim in bytes (planes of H x W fp32 pixels). */
extern "C" int64_t ipoke_image_metrics_workspace_bytes(int64_t planes, int H, int W) {
  const int64_t Ho = H > 10 ? H - 10 : 0, Wo = W > 10 ? W - 10 : 0;
  return 64 + 8 * (planes * ((Ho + 31) / 32) * ((Wo + 31) / 32) + 1);
}
/* out[0] = psnr(preds, target), out[1] = ssim(preds, target) of `planes` fp32 image planes [planes][H][W] (the N * C planes of NCHW
 * tensors): pytorch_lightning.metrics.functional.psnr / ssim with their defaults, as SSIM_custom / PSNR_custom call them
 * (metrics.py:450-481). */
extern "C" int ipoke_psnr_ssim(const float* preds, const float* target, int64_t planes, int H, int W, void* workspace, float* out, void* stream) {
  IPK_REQUIRE(preds && target && workspace && out && planes >= 1 && H >= 11 && W >= 11, "bad arguments (images must be at least 11 x 11)");
  hipStream_t s = STREAM(stream);
  float* mm = reinterpret_cast<float*>(workspace);
  double* sse = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(workspace) + 32);
  double* part = sse + 4;
  const long n = (long)planes * H * W;
  IPK_HIP(hipMemsetAsync(mm, 0xff, 4 * sizeof(float), s));       // 0xffffffff: below every float in the order of atomic_max_f32
  IPK_HIP(hipMemsetAsync(sse, 0, sizeof(double), s));
  hipLaunchKernelGGL(pair_stats_kernel, dim3(grid1(n, 2048)), dim3(256), 0, s, preds, target, n, mm, sse);
  IPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(psnr_final_kernel, dim3(1), dim3(1), 0, s, mm, sse, n, out);
  IPK_LAUNCH_CHECK();
  const int Ho = H - 10, Wo = W - 10, tiles = ((Ho + 31) / 32) * ((Wo + 31) / 32);
  hipLaunchKernelGGL(ssim_kernel, dim3(tiles, (unsigned)planes), dim3(256), 0, s, preds, target, H, W, mm, part);
  IPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(ssim_final_kernel, dim3(1), dim3(256), 0, s, part, (long)planes * tiles, (double)planes * Ho * Wo, out + 1);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
