// First-stage (video VAE) training helpers that used to be PyTorch glue around the convolution kernels:
//   * conv weight (PyTorch layout, fp32) -> matrix-core operand [Cout][taps*Kc] of the compute dtype, optionally scaled by a
//     device-resident 1/sigma (spectral norm), in one launch;
//   * spectral normalisation: one power iteration + sigma = u^T W v (torch.nn.utils.spectral_norm semantics, used by every
//     decoder convolution: reference models/modules/autoencoders/util.py:52,252 via first_stage_motion_model.py:263-276) and
//     the backward of w_eff = w_orig / sigma;
//   * multi-tensor Adam (torch.optim.Adam semantics, coupled weight decay; first_stage_motion_model.py:283-300);
//   * KL(q || N(0,1)) value and gradient in one pass (utils/losses.py:47-48).
// All of it is HBM-bound streaming work: every weight element is read once per pass (twice per power iteration).
#include "common.h"

#include <vector>
#include <cstring>

namespace ipoke {

// ---------------------------------------------------------------------------------------------- weight operand
// w: [cout][cin][taps] (transposed = 0) or [cin][cout][taps] (transposed = 1: ConvTranspose storage, read as a conv weight)
template <typename T>
__global__ void conv_weight_operand_kernel(const float* __restrict__ w, int cout, int cin, int taps, int transposed,
                                           const float* __restrict__ inv_scale, T* __restrict__ out, int kc) {
  const long total = (long)cout * taps * kc;
  const float s = inv_scale ? *inv_scale : 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % kc); const long r = i / kc;
    const int t = (int)(r % taps), o = (int)(r / taps);
    float v = 0.f;
    if (c < cin) v = s * (transposed ? w[((long)c * cout + o) * taps + t] : w[((long)o * cin + c) * taps + t]);
    out[i] = ET<T>::from_f32(v);
  }
}

// the same for up to kWopJobs weights per launch (blockIdx.y = job, jobs by value): the operands of every cached weight are refreshed
// right behind the optimizer step in ONE launch instead of lazily, one 8 us launch per weight on the chain (c4: 101 per step)
constexpr int kWopJobs = 64;
struct WopJob { const float* w; void* out; int cout, cin, taps, transposed, kc, pad; };
struct WopJobs { WopJob j[kWopJobs]; };
template <typename T>
__global__ __launch_bounds__(256) void conv_weight_operand_multi_kernel(const WopJobs J) {
  const WopJob& q = J.j[blockIdx.y];
  const long total = (long)q.cout * q.taps * q.kc;
  T* out = reinterpret_cast<T*>(q.out);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % q.kc); const long r = i / q.kc;
    const int t = (int)(r % q.taps), o = (int)(r / q.taps);
    float v = 0.f;
    if (c < q.cin) v = q.transposed ? q.w[((long)c * q.cout + o) * q.taps + t] : q.w[((long)o * q.cin + c) * q.taps + t];
    out[i] = ET<T>::from_f32(v);
  }
}

// ---------------------------------------------------------------------------------------------- spectral norm
// The weight matrix W [R][C] is a strided view of the conv weight: element (r, c = c1 * n2 + c2) at r*s_r + c1*s_1 + c2.
struct SnView { const float* w; int R, C, n2; long s_r, s_1; };
__device__ __forceinline__ float sn_at(const SnView& V, int r, int c) {
  const int c1 = c / V.n2, c2 = c - c1 * V.n2;
  return V.w[(long)r * V.s_r + (long)c1 * V.s_1 + c2];
}
// tv[c] = sum_r W[r][c] u[r]: a workgroup owns kSnCols columns, its 256 threads are kSnCols columns x kSnRowGroups row groups (rows
// r = group, group + kSnRowGroups, ...); the groups meet in LDS and are summed in a FIXED order -- no float atomics, so that the
// power iteration (and with it sigma, the decoder's output and every gradient of a training step) is reproducible run to run.
// (Rounds 2-4 split the rows over workgroups and added the chunks with atomicAdd: the order of those additions changed sigma in its
//  last bits from run to run, which bf16 roundings downstream amplify into visibly different gradients.)
constexpr int kSnCols = 32, kSnRowGroups = 8;
__device__ __forceinline__ void sn_cols_body(const SnView& V, const float* __restrict__ u, float* __restrict__ tv, float (*part)[kSnCols]) {
  const int cl = threadIdx.x % kSnCols, rg = threadIdx.x / kSnCols;
  const int c = blockIdx.x * kSnCols + cl;
  float s = 0.f;
  if (c < V.C) {
    const int c1 = c / V.n2, c2 = c - c1 * V.n2;
    const float* col = V.w + (long)c1 * V.s_1 + c2;
#pragma unroll 4
    for (int r = rg; r < V.R; r += kSnRowGroups) s = fmaf(col[(long)r * V.s_r], u[r], s);
  }
  part[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < V.C) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < kSnRowGroups; ++q) t += part[q][cl];
    tv[c] = t;
  }
}
__global__ __launch_bounds__(256) void sn_cols_kernel(SnView V, const float* __restrict__ u, float* __restrict__ tv) {
  __shared__ float part[kSnRowGroups][kSnCols];
  sn_cols_body(V, u, tv, part);
}
// one wave per row: tu[r] = sum_c W[r][c] vhat[c], vhat = iterate ? tv / max(||tv||, eps) : v
// (iterate: every block derives ||tv|| itself; block 0 also stores vhat into v; ||tu|| is taken by the final kernel)
__global__ __launch_bounds__(256) void sn_rows_kernel(SnView V, const float* __restrict__ tv, float* __restrict__ v, float* __restrict__ tu,
                                                      float* __restrict__ acc, int iterate, float eps) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float inv = 1.f;
  if (iterate) {
    float sq = 0.f;
    for (int c = threadIdx.x; c < V.C; c += 256) sq = fmaf(tv[c], tv[c], sq);
    inv = 1.f / fmaxf(sqrtf(block_sum(sq, red)), eps);
  }
  const float* src = iterate ? tv : v;
  const int r = blockIdx.x * 4 + wave;
  if (r < V.R) {
    float s = 0.f;
    for (int c = lane; c < V.C; c += 64) s = fmaf(sn_at(V, r, c), src[c] * inv, s);
    s = wave_sum(s);
    if (lane == 0) tu[r] = s;
  }
  if (iterate && blockIdx.x == 0)
    for (int c = threadIdx.x; c < V.C; c += 256) v[c] = tv[c] * inv;
}
// u = iterate ? tu / max(||tu||, eps) : u;  sigma = sum_r u[r] tu[r];  out = {sigma, 1/sigma};  snapshot = [u | v];  acc reset
__global__ __launch_bounds__(256) void sn_final_kernel(int R, int C, float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ tu,
                                                       float* __restrict__ acc, float* __restrict__ out, float* __restrict__ snap, int iterate,
                                                       float eps, float* __restrict__ tv) {
  __shared__ float red[4];
  float inv = 1.f;
  if (iterate) {
    float sq = 0.f;
    for (int r = threadIdx.x; r < R; r += 256) sq = fmaf(tu[r], tu[r], sq);
    inv = 1.f / fmaxf(sqrtf(block_sum(sq, red)), eps);
    __syncthreads();                                      // `red` is reused below
  }
  float s = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    const float ur = iterate ? tu[r] * inv : u[r];
    if (iterate) u[r] = ur;
    if (snap) snap[r] = ur;
    s = fmaf(ur, tu[r], s);
  }
  if (snap) for (int c = threadIdx.x; c < C; c += 256) snap[R + c] = v[c];
  s = block_sum(s, red);
  if (threadIdx.x == 0) { out[0] = s; out[1] = 1.f / s; }
}
// ---- the same three kernels over MANY weights per launch (blockIdx.z = job) and a given iteration index: a decoder weight is
// normalised once per generated frame, i.e. T - 1 power iterations per training pass whose inputs are the weight and its own
// u / v only -- they are run ahead of the time loop, one launch triple per iteration for all weights instead of one per call.
struct SnJob { SnView V; float* u; float* v; float* out; float* snap; float* ws; long out_stride, snap_stride; };
__global__ __launch_bounds__(256) void sn_cols_multi_kernel(const SnJob* __restrict__ jobs) {
  const SnJob J = jobs[blockIdx.z];
  const SnView& V = J.V;
  __shared__ float part[kSnRowGroups][kSnCols];
  if ((int)blockIdx.x * kSnCols >= V.C) return;          // (uniform per workgroup)
  sn_cols_body(V, J.u, J.ws + 4 + V.R, part);
}
__global__ __launch_bounds__(256) void sn_rows_multi_kernel(const SnJob* __restrict__ jobs, float eps) {
  __shared__ float red[4];
  const SnJob J = jobs[blockIdx.z];
  const SnView& V = J.V;
  if ((int)blockIdx.x * 4 >= V.R) return;
  float* acc = J.ws; float* tu = J.ws + 4; float* tv = tu + V.R;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float sq = 0.f;
  for (int c = threadIdx.x; c < V.C; c += 256) sq = fmaf(tv[c], tv[c], sq);
  const float inv = 1.f / fmaxf(sqrtf(block_sum(sq, red)), eps);
  const int r = blockIdx.x * 4 + wave;
  if (r < V.R) {
    float s = 0.f;
    for (int c = lane; c < V.C; c += 64) s = fmaf(sn_at(V, r, c), tv[c] * inv, s);
    s = wave_sum(s);
    if (lane == 0) tu[r] = s;
  }
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < V.C; c += 256) J.v[c] = tv[c] * inv;
}
__global__ __launch_bounds__(256) void sn_final_multi_kernel(const SnJob* __restrict__ jobs, int it, float eps) {
  __shared__ float red[4];
  const SnJob J = jobs[blockIdx.z];
  const int R = J.V.R, C = J.V.C;
  float* acc = J.ws; float* tu = J.ws + 4; float* tv = tu + R;
  float* out = J.out + (long)it * J.out_stride; float* snap = J.snap + (long)it * J.snap_stride;
  float sq = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) sq = fmaf(tu[r], tu[r], sq);
  const float inv = 1.f / fmaxf(sqrtf(block_sum(sq, red)), eps);
  __syncthreads();                                        // `red` is reused below
  float s = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    const float ur = tu[r] * inv;
    J.u[r] = ur; snap[r] = ur;
    s = fmaf(ur, tu[r], s);
  }
  for (int c = threadIdx.x; c < C; c += 256) snap[R + c] = J.v[c];
  s = block_sum(s, red);
  if (threadIdx.x == 0) { out[0] = s; out[1] = 1.f / s; }
}

// backward of w_eff = w / sigma with sigma = u^T W v (u, v constants):  dW = (G - <G, W_eff> u v^T) / sigma
__global__ __launch_bounds__(256) void sn_bwd_dot_kernel(SnView V, const float* __restrict__ g, float* __restrict__ acc) {
  __shared__ float red[4];
  const long total = (long)V.R * V.C;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i / V.C), c = (int)(i - (long)r * V.C);
    const int c1 = c / V.n2, c2 = c - c1 * V.n2;
    const long off = (long)r * V.s_r + (long)c1 * V.s_1 + c2;
    s = fmaf(g[off], V.w[off], s);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) acc[blockIdx.x] = s;              // one partial per workgroup, summed in a fixed order by the apply kernel
}
__global__ __launch_bounds__(256) void sn_bwd_apply_kernel(SnView V, float* __restrict__ g, const float* __restrict__ snap, const float* __restrict__ sig,
                                                           const float* __restrict__ acc) {
  const long total = (long)V.R * V.C;
  // the dot kernel's per-workgroup partials, summed ONCE per workgroup in their fixed order (every thread of every workgroup used to walk
  // all of them: up to 256 dependent loads per thread)
  __shared__ float tot_s;
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int b = 0; b < (int)gridDim.x; ++b) t += acc[b];
    tot_s = t;
  }
  __syncthreads();
  const float tot = tot_s;
  const float inv = sig[1], dot = tot * inv;             // <G, W/sigma>
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i / V.C), c = (int)(i - (long)r * V.C);
    const int c1 = c / V.n2, c2 = c - c1 * V.n2;
    const long off = (long)r * V.s_r + (long)c1 * V.s_1 + c2;
    g[off] = (g[off] - dot * snap[r] * snap[V.R + c]) * inv;
  }
}

// ---------------------------------------------------------------------------------------------- frames decoded as one batch
// First-stage training decodes the T - 1 generated frames of a batch of clips as ONE batch ordered (frame, clip) with one operand of
// W_orig and 1 / sigma_t (torch's spectral_norm iterates once per decoder call, util.py:52, 252) applied per image group in the GEMM
// epilogue (ipoke_conv_desc.row_scale):  y = act(conv(x, W) / sigma_t + b).  Backward of that layer needs, in one pass over (dy, y):
//   gs    = dy * act'(y) / sigma_t                    the gradient of conv(x, W): rows of the data- and weight-gradient GEMMs
//   dot_t = sum_{rows of frame t} dy * act'(y) * (pre(y) - b)  = <dW_eff_t, W / sigma_t>     (sigma_t's own gradient, below)
//   dbias = column sums of dy * act'(y)
// pre(y): the pre-activation recovered from the saved output (where it cannot be recovered -- ReLU's zeros -- act'(y) is zero).
// Grid (row blocks of a group, groups); per-block partials, reduced in a fixed order by rowscale_final_kernel (deterministic).
struct RowScaleBwd {
  const void* dy; int lddy; const void* y; int ldy; long M; int C, Cpad, act;
  const float* bias; const float* scale; int scale_stride; long rows_per_group;
  void* gs; int ldgs; float* dots; float* dbias; float* dot_part; float* col_part; int nbx, ngroups, rows_per_block;
};
__device__ __forceinline__ float pre_from_out(int act, float y) {
  switch (act) {
    case IPOKE_ACT_ELU: return y > 0.f ? y : log1pf(fmaxf(y, -0.99999994f));
    case IPOKE_ACT_LRELU02: return y > 0.f ? y : 5.f * y;
    default: return y;          // NONE; RELU (the gradient is zero where the output was clipped)
  }
}
template <typename T>
__global__ __launch_bounds__(256) void rowscale_bwd_kernel(const RowScaleBwd a) {
  constexpr int E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  __shared__ float sm[256 * E16];
  __shared__ float red[4];
  const int grp = blockIdx.y;
  const long r0 = (long)grp * a.rows_per_group + (long)blockIdx.x * a.rows_per_block;
  const long r1 = min((long)(grp + 1) * a.rows_per_group, min(a.M, r0 + a.rows_per_block));
  const float sc = a.scale[(long)grp * a.scale_stride];
  const int cvec = a.Cpad / E16;
  float dot = 0.f;
  float* cpart = a.col_part ? a.col_part + ((long)grp * a.nbx + blockIdx.x) * a.C : nullptr;
  for (int g0 = 0; g0 < cvec; g0 += 256) {
    const int groups = cvec - g0 < 256 ? cvec - g0 : 256;
    const int rp = 256 / groups;
    const int cg = g0 + threadIdx.x % groups, rr = threadIdx.x / groups;
    float cs[E16], bb[E16];
#pragma unroll
    for (int e = 0; e < E16; ++e) { cs[e] = 0.f; const int c = cg * E16 + e; bb[e] = (a.bias && c < a.C) ? a.bias[c] : 0.f; }
    if (rr < rp) {
      for (long m = r0 + rr; m < r1; m += rp) {
        const frag_t gy = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.dy) + m * a.lddy + cg * E16);
        const frag_t yv = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.y) + m * a.ldy + cg * E16);
        frag_t o;
#pragma unroll
        for (int e = 0; e < E16; ++e) {
          const int c = cg * E16 + e;
          float g = 0.f;
          if (c < a.C) {
            const float yy = ET<T>::to_f32(yv[e]);
            g = ET<T>::to_f32(gy[e]) * act_grad_from_out(a.act, yy);
            if (g != 0.f) dot = fmaf(g, pre_from_out(a.act, yy) - bb[e], dot);
            cs[e] += g;
          }
          o[e] = ET<T>::from_f32(g * sc);
        }
        *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(a.gs) + m * a.ldgs + cg * E16) = o;
      }
    }
    if (cpart) {
#pragma unroll
      for (int e = 0; e < E16; ++e) sm[threadIdx.x * E16 + e] = rr < rp ? cs[e] : 0.f;
      __syncthreads();
      for (int i = threadIdx.x; i < groups * E16; i += 256) {
        const int cgi = i / E16, e = i - cgi * E16, c = (g0 + cgi) * E16 + e;
        float t = 0.f;
        for (int k = 0; k < rp; ++k) t += sm[(k * groups + cgi) * E16 + e];
        if (c < a.C) cpart[c] = t;
      }
      __syncthreads();
    }
  }
  dot = block_sum(dot, red);
  if (threadIdx.x == 0) a.dot_part[(long)grp * a.nbx + blockIdx.x] = dot;
}
// blocks 0 .. ngroups-1: dots[group]; blocks ngroups ..: 16 channels of dbias each (16 row lanes, fixed order)
__global__ __launch_bounds__(256) void rowscale_final_kernel(const RowScaleBwd a) {
  __shared__ float red[256];
  if ((int)blockIdx.x < a.ngroups) {
    float t = 0.f;
    for (int k = threadIdx.x; k < a.nbx; k += 256) t += a.dot_part[(long)blockIdx.x * a.nbx + k];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) a.dots[blockIdx.x] = red[0];
    return;
  }
  const int cl = threadIdx.x % 16, rl = threadIdx.x / 16, c = ((int)blockIdx.x - a.ngroups) * 16 + cl;
  const long nblk = (long)a.ngroups * a.nbx;
  float t = 0.f;
  if (c < a.C) {
    // eight partial rows in flight per lane (fixed order): at the 128 x 128 layers a lane walks 600 partial rows, and one load at a
    // time made this 4-block kernel 61 us long
    float u[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    long k = rl;
    for (; k + 7 * 16 < nblk; k += 8 * 16) {
#pragma unroll
      for (int q = 0; q < 8; ++q) u[q] += a.col_part[(k + q * 16) * a.C + c];
    }
    for (; k < nblk; k += 16) u[0] += a.col_part[k * a.C + c];
    t = ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
  }
  red[threadIdx.x] = t;
  __syncthreads();
  if (rl == 0 && c < a.C) {
    float u = 0.f;
    for (int k = 0; k < 16; ++k) u += red[k * 16 + cl];
    a.dbias[c] = u;
  }
}
// grad[r][c] -= sum_t dots[t] / sigma_t * u_t[r] v_t[c]: the gradient that reaches W_orig through sigma_t = u_t^T W v_t, summed over
// the frames (grad already holds sum_t dW_eff_t / sigma_t -- the weight gradient of the rows scaled by 1 / sigma_t)
__global__ __launch_bounds__(256) void sn_bwd_frames_kernel(SnView V, float* __restrict__ g, const float* __restrict__ snap, long snap_stride,
                                                            const float* __restrict__ sig, long sig_stride, const float* __restrict__ dots, int frames) {
  const long total = (long)V.R * V.C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i / V.C), c = (int)(i - (long)r * V.C);
    const int c1 = c / V.n2, c2 = c - c1 * V.n2;
    const long off = (long)r * V.s_r + (long)c1 * V.s_1 + c2;
    float t = 0.f;
    for (int f = 0; f < frames; ++f) {
      const float* sp = snap + (long)f * snap_stride;
      t = fmaf(dots[f] * sig[(long)f * sig_stride + 1], sp[r] * sp[V.R + c], t);
    }
    g[off] -= t;
  }
}
// dst[i] = sum_f src[f][i] (fp32 accumulation): per-position gradients of maps shared by the frames of a clip (the SPADE modulation)
template <typename T>
__global__ __launch_bounds__(256) void sum_frames_kernel(const T* __restrict__ src, T* __restrict__ dst, int frames, long n) {
  constexpr int E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  const long nv = n / E16;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    float acc[E16];
#pragma unroll
    for (int e = 0; e < E16; ++e) acc[e] = 0.f;
    for (int f = 0; f < frames; ++f) {
      const frag_t v = *reinterpret_cast<const frag_t*>(src + (long)f * n + i * E16);
#pragma unroll
      for (int e = 0; e < E16; ++e) acc[e] += ET<T>::to_f32(v[e]);
    }
    frag_t o;
#pragma unroll
    for (int e = 0; e < E16; ++e) o[e] = ET<T>::from_f32(acc[e]);
    *reinterpret_cast<frag_t*>(dst + i * E16) = o;
  }
}

// ---------------------------------------------------------------------------------------------- multi-tensor Adam
constexpr int kAdamMax = 48;
struct AdamPack { float* p[kAdamMax]; const float* g[kAdamMax]; float* m[kAdamMax]; float* v[kAdamMax]; long n[kAdamMax]; };
__global__ __launch_bounds__(256) void adam_multi_kernel(AdamPack P, float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                         float bc2_sqrt, float grad_scale) {
  const int t = blockIdx.y;
  float* __restrict__ p = P.p[t]; const float* __restrict__ g = P.g[t]; float* __restrict__ m = P.m[t]; float* __restrict__ v = P.v[t];
  const long n = P.n[t];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gr = g[i] * grad_scale + wd * p[i];
    const float mm = beta1 * m[i] + (1.f - beta1) * gr;
    const float vv = beta2 * v[i] + (1.f - beta2) * gr * gr;
    m[i] = mm; v[i] = vv;
    p[i] -= (lr / bc1) * (mm / (sqrtf(vv) / bc2_sqrt + eps));
  }
}

// ---------------------------------------------------------------------------------------------- KL
// kl = -0.5/Mpos * sum_{m,c} (1 + lv - mu^2 - exp(lv));  dmu = mu / Mpos, dlv = 0.5 (exp(lv) - 1) / Mpos   (times `scale`)
__global__ __launch_bounds__(256) void kl_loss_kernel(const float* __restrict__ mu, const float* __restrict__ lv, long n, float inv_m,
                                                      float* __restrict__ loss, float* __restrict__ dmu, float* __restrict__ dlv) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float a = mu[i], b = lv[i], e = expf(b);
    s += 1.f + b - a * a - e;
    dmu[i] = a * inv_m;
    dlv[i] = 0.5f * (e - 1.f) * inv_m;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(loss, -0.5f * inv_m * s);      // (ONE workgroup up to 2^20 elements: a deterministic value)
}

// ---------------------------------------------------------------------------------------------- pooling (discriminators)
// MaxPool3d on channels-last rows (reference patchgan_3d.py:202: kernel 3, stride (1,2,2), padding 1).  The first maximum in
// (d, h, w) scan order wins, as in torch; its input row is kept for the backward pass.
struct PoolGeom { int N, C, Di, Hi, Wi, Do, Ho, Wo, kd, kh, kw, sd, sh, sw, pd, ph, pw; };
template <typename T>
__global__ void maxpool3d_fwd_kernel(PoolGeom g, const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int* __restrict__ idx) {
  const long total = (long)g.N * g.Do * g.Ho * g.Wo * g.C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % g.C); long r = i / g.C;
    const int ow = (int)(r % g.Wo); long t = r / g.Wo;
    const int oh = (int)(t % g.Ho); t /= g.Ho;
    const int od = (int)(t % g.Do); const int n = (int)(t / g.Do);
    float best = -INFINITY; int arg = -1;
    for (int a = 0; a < g.kd; ++a) {
      const int d = od * g.sd - g.pd + a; if ((unsigned)d >= (unsigned)g.Di) continue;
      for (int b = 0; b < g.kh; ++b) {
        const int h = oh * g.sh - g.ph + b; if ((unsigned)h >= (unsigned)g.Hi) continue;
        for (int e = 0; e < g.kw; ++e) {
          const int w = ow * g.sw - g.pw + e; if ((unsigned)w >= (unsigned)g.Wi) continue;
          const int row = ((n * g.Di + d) * g.Hi + h) * g.Wi + w;
          const float v = ET<T>::to_f32(x[(long)row * ldx + c]);
          if (v > best || arg < 0) { best = v; arg = row; }
        }
      }
    }
    y[r * ldy + c] = ET<T>::from_f32(best);
    idx[r * g.C + c] = arg;
  }
}
// gather form (no atomics): an input position collects dy of every window that chose it
template <typename T>
__global__ void maxpool3d_bwd_kernel(PoolGeom g, const T* __restrict__ dy, int ldy, const int* __restrict__ idx, T* __restrict__ dx, int ldx) {
  const long total = (long)g.N * g.Di * g.Hi * g.Wi * ldx;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % ldx); const long row = i / ldx;
    float s = 0.f;
    if (c < g.C) {
      const int w = (int)(row % g.Wi); long t = row / g.Wi;
      const int h = (int)(t % g.Hi); t /= g.Hi;
      const int d = (int)(t % g.Di); const int n = (int)(t / g.Di);
      for (int od = max(0, (d + g.pd - g.kd + g.sd) / g.sd); od <= min(g.Do - 1, (d + g.pd) / g.sd); ++od)
        for (int oh = max(0, (h + g.ph - g.kh + g.sh) / g.sh); oh <= min(g.Ho - 1, (h + g.ph) / g.sh); ++oh)
          for (int ow = max(0, (w + g.pw - g.kw + g.sw) / g.sw); ow <= min(g.Wo - 1, (w + g.pw) / g.sw); ++ow) {
            const long o = ((long)(n * g.Do + od) * g.Ho + oh) * g.Wo + ow;
            if (idx[o * g.C + c] == (int)row) s += ET<T>::to_f32(dy[o * ldy + c]);
          }
    }
    dx[i] = ET<T>::from_f32(s);
  }
}
// the same gather, 8 bf16 channels (16 bytes of dy, 32 bytes of idx) per thread: the scalar form issued twelve 4-byte idx loads and
// as many 2-byte dy loads per element (0.95 ms at the discriminator's 12 x 64 x 64 x 64 input)
__global__ void maxpool3d_bwd_vec8_kernel(PoolGeom g, const bf16_t* __restrict__ dy, int ldy, const int* __restrict__ idx, bf16_t* __restrict__ dx, int ldx) {
  const int cv = ldx >> 3;
  const long total = (long)g.N * g.Di * g.Hi * g.Wi * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cv) * 8; const long row = i / cv;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (c0 < g.C) {                        // C is a multiple of 8 (host-checked): a chunk is all real channels or all padding
      const int w = (int)(row % g.Wi); long t = row / g.Wi;
      const int h = (int)(t % g.Hi); t /= g.Hi;
      const int d = (int)(t % g.Di); const int n = (int)(t / g.Di);
      for (int od = max(0, (d + g.pd - g.kd + g.sd) / g.sd); od <= min(g.Do - 1, (d + g.pd) / g.sd); ++od)
        for (int oh = max(0, (h + g.ph - g.kh + g.sh) / g.sh); oh <= min(g.Ho - 1, (h + g.ph) / g.sh); ++oh)
          for (int ow = max(0, (w + g.pw - g.kw + g.sw) / g.sw); ow <= min(g.Wo - 1, (w + g.pw) / g.sw); ++ow) {
            const long o = ((long)(n * g.Do + od) * g.Ho + oh) * g.Wo + ow;
            const int4 i0 = *reinterpret_cast<const int4*>(idx + o * g.C + c0), i1 = *reinterpret_cast<const int4*>(idx + o * g.C + c0 + 4);
            const int sel[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
            bool any = false;
#pragma unroll
            for (int e = 0; e < 8; ++e) any |= sel[e] == (int)row;
            if (!any) continue;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(dy + o * ldy + c0);
#pragma unroll
            for (int e = 0; e < 8; ++e) if (sel[e] == (int)row) s[e] += (float)v[e];
          }
    }
    bf16x8 o8;
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = (bf16_t)s[e];
    *reinterpret_cast<bf16x8*>(dx + row * ldx + c0) = o8;
  }
}
// mean over the S = H*W positions of each (sample, frame): x [G*S][ldx] -> y [G][ldy];  backward broadcasts dy / S
template <typename T>
__global__ void avgpool_rows_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, long G, int S, int C) {
  const long total = G * ldy;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % ldy); const long gi = i / ldy;
    float s = 0.f;
    if (c < C) for (int k = 0; k < S; ++k) s += ET<T>::to_f32(x[(gi * S + k) * ldx + c]);
    y[i] = ET<T>::from_f32(s / (float)S);
  }
}
template <typename T>
__global__ void avgpool_rows_bwd_kernel(const T* __restrict__ dy, int ldy, T* __restrict__ dx, int ldx, long G, int S, int C) {
  const long total = G * S * ldx;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % ldx); const long row = i / ldx;
    dx[i] = ET<T>::from_f32(c < C ? ET<T>::to_f32(dy[(row / S) * ldy + c]) / (float)S : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------- GroupNorm tangent (gradient penalty)
// The gradient penalty of the discriminators (patchgan_3d.py:285-294: mean_b |d sum(pred) / d x|^2, differentiated w.r.t. the
// parameters) is evaluated forward-over-reverse: with g = d sum(pred)/dx from an ordinary backward pass and v = (2/B) g held
// constant, d reg / d theta = d/d theta [ D_v sum(pred) ], the derivative of the network along v.  D_v is propagated by a
// tangent forward pass next to the primal one: convolutions are linear (the same kernels run on the tangent), ReLU and
// max-pooling apply the primal pass's masks / selections, and GroupNorm contributes the only second-order term:
//   y = gamma xhat + beta, xhat = (x - mu) r:    ydot = gamma r (u - mean(u) - xhat mean(xhat u)),  u = xdot
// (the same projection as GroupNorm's backward), then act'(y) and the residual tangent.  Its backward, for an upstream
// gradient q on ydot, with w = gamma act'(y) q and per-group means a = <w>, b = <w xhat>, c = <w u>, ub = <u>, m = <xhat u>:
//   d/d xdot = r (w - a - xhat b)
//   d/d x    = -r^2 [ xhat (c - a ub - 3 b m) + m (w - a) + b (u - ub) ]          (through mu, r and xhat)
//   d/d gamma_c = sum act'(y) q r (u - ub - xhat m),   d/d resdot = act'(y) q
// One workgroup per (sample, group); element e of the group is position e / cpg, channel g*cpg + e % cpg.
struct GnJvp {
  const void* x; int ldx; const void* xd; int ldxd; const void* y; int ldy; const void* resd; int ldres;
  void* yd; int ldyd;                 // forward output
  const void* q; int ldq; void* dxd; int lddxd; void* dx; int lddx; void* dresd; int lddresd; float* dgamma;   // backward
  const float* gamma; int N, S, C, G, act; float eps;
};
// Two launches per direction, both on a (position chunk, sample) grid with 16-byte loads (thread (rr, cg) owns the E16 channels
// of column group cg over rows rr, rr + rp, ...): (1) per-channel sums over the chunk -> per-group sums, added atomically
// into gsum[n][g][.]: sum x, x^2, u, x u (+ w, w x, w u in the backward); (2) the element-wise pass with the group scalars
// derived from those sums (xhat sums follow from raw ones: <xhat u> = r (<x u> - mu <u>)).
constexpr int kGjPos = 128;          // positions per block
constexpr int kGjMaxG = 512;         // groups per sample the block-level sums hold (host-checked)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void gn_jvp_sums_kernel(const GnJvp a, float* __restrict__ gsum) {
  constexpr int E16 = ET<T>::E16, NK = BWD ? 7 : 4;
  typedef typename ET<T>::frag frag_t;
  const int n = blockIdx.y, p0 = blockIdx.x * kGjPos, p1 = min(a.S, p0 + kGjPos);
  const int cpg = a.C / a.G, cvec = a.C / E16;
  // block-level sums in LDS first, ONE global atomic per (group, quantity) and block afterwards: with every thread adding its
  // partial sums straight into gsum[n][g][.] (2 560 addresses for 14 M atomics at the discriminator's 12 x 64 x 64 maps) a launch
  // took 1.0 ms for 0.1 ms of memory traffic
  __shared__ float bsum[NK][kGjMaxG];
  for (int i = threadIdx.x; i < NK * kGjMaxG; i += 256) (&bsum[0][0])[i] = 0.f;
  __syncthreads();
  for (int g0 = 0; g0 < cvec; g0 += 256) {
    const int groups = cvec - g0 < 256 ? cvec - g0 : 256;
    const int rp = 256 / groups;
    const int cg = g0 + threadIdx.x % groups, rr = threadIdx.x / groups;
    float acc[NK][E16];
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int e = 0; e < E16; ++e) acc[k][e] = 0.f;
    if (rr < rp) {
      for (int p = p0 + rr; p < p1; p += rp) {
        const long m = (long)n * a.S + p;
        const frag_t xv = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.x) + m * a.ldx + cg * E16);
        const frag_t uv = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.xd) + m * a.ldxd + cg * E16);
        frag_t qv, yv;
        if (BWD) {
          qv = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.q) + m * a.ldq + cg * E16);
          if (a.act != IPOKE_ACT_NONE) yv = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.y) + m * a.ldy + cg * E16);
        }
#pragma unroll
        for (int e = 0; e < E16; ++e) {
          const float x = ET<T>::to_f32(xv[e]), u = ET<T>::to_f32(uv[e]);
          acc[0][e] += x; acc[1][e] += x * x; acc[2][e] += u; acc[3][e] += x * u;
          if constexpr (BWD) {
            float w = ET<T>::to_f32(qv[e]) * (a.gamma ? a.gamma[cg * E16 + e] : 1.f);
            if (a.act != IPOKE_ACT_NONE) w *= act_grad_from_out(a.act, ET<T>::to_f32(yv[e]));
            acc[4][e] += w; acc[5][e] += w * x; acc[6][e] += w * u;
          }
        }
      }
    }
    // per-thread sums -> per-group sums: a thread's E16 channels lie in at most two groups when cpg < E16, in one otherwise
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      if (rr < rp) {
        if (cpg >= E16) {
          float t = 0.f;
#pragma unroll
          for (int e = 0; e < E16; ++e) t += acc[k][e];
          atomicAdd(&bsum[k][(cg * E16) / cpg], t);
        } else {
          for (int e0 = 0; e0 < E16; e0 += cpg) {
            float t = 0.f;
            for (int e = e0; e < e0 + cpg; ++e) t += acc[k][e];
            atomicAdd(&bsum[k][(cg * E16 + e0) / cpg], t);
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NK * a.G; i += 256) {
    const int k = i / a.G, g = i - k * a.G;
    atomicAdd(gsum + ((long)n * a.G + g) * 8 + k, bsum[k][g]);
  }
}
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void gn_jvp_apply_kernel(const GnJvp a, const float* __restrict__ gsum) {
  constexpr int E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  extern __shared__ float lds[];            // [8][G]: mu, r, ub, mm, aa, bb, k0, -  per group of this sample; then [C] dgamma partials
  const int n = blockIdx.y, p0 = blockIdx.x * kGjPos, p1 = min(a.S, p0 + kGjPos);
  const int cpg = a.C / a.G, cvec = a.C / E16, G = a.G;
  const float inv = 1.f / ((float)a.S * cpg);
  for (int g = threadIdx.x; g < G; g += 256) {
    const float* s = gsum + ((long)n * G + g) * 8;
    const float mu = s[0] * inv, var = fmaxf(s[1] * inv - mu * mu, 0.f), r = rsqrtf(var + a.eps);
    const float ub = s[2] * inv, mm = r * (s[3] * inv - mu * ub);
    lds[g] = mu; lds[G + g] = r; lds[2 * G + g] = ub; lds[3 * G + g] = mm;
    if (BWD) {
      const float aa = s[4] * inv, bb = r * (s[5] * inv - mu * aa), cc = s[6] * inv;
      lds[4 * G + g] = aa; lds[5 * G + g] = bb; lds[6 * G + g] = cc - aa * ub - 3.f * bb * mm;
    }
  }
  float* dg = lds + 8 * G;
  if (BWD && a.dgamma) for (int c = threadIdx.x; c < a.C; c += 256) dg[c] = 0.f;
  __syncthreads();
  for (int g0 = 0; g0 < cvec; g0 += 256) {
    const int groups = cvec - g0 < 256 ? cvec - g0 : 256;
    const int rp = 256 / groups;
    const int cg = g0 + threadIdx.x % groups, rr = threadIdx.x / groups;
    float dgl[E16];
#pragma unroll
    for (int e = 0; e < E16; ++e) dgl[e] = 0.f;
    if (rr < rp) {
      for (int p = p0 + rr; p < p1; p += rp) {
        const long m = (long)n * a.S + p;
        const frag_t xv = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.x) + m * a.ldx + cg * E16);
        const frag_t uv = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.xd) + m * a.ldxd + cg * E16);
        frag_t yv, rv, qv, o1, o2, o3;
        if (a.act != IPOKE_ACT_NONE) yv = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.y) + m * a.ldy + cg * E16);
        if (!BWD && a.resd) rv = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.resd) + m * a.ldres + cg * E16);
        if (BWD) qv = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.q) + m * a.ldq + cg * E16);
#pragma unroll
        for (int e = 0; e < E16; ++e) {
          const int c = cg * E16 + e, g = c / cpg;
          const float mu = lds[g], r = lds[G + g], ub = lds[2 * G + g], mm = lds[3 * G + g];
          const float xh = (ET<T>::to_f32(xv[e]) - mu) * r, u = ET<T>::to_f32(uv[e]);
          const float gm = a.gamma ? a.gamma[c] : 1.f;
          const float da = a.act != IPOKE_ACT_NONE ? act_grad_from_out(a.act, ET<T>::to_f32(yv[e])) : 1.f;
          const float proj = r * (u - ub - xh * mm);
          if (!BWD) {
            float yd = gm * proj;
            if (a.resd) yd += ET<T>::to_f32(rv[e]);
            o1[e] = ET<T>::from_f32(yd * da);
          } else {
            const float qd = ET<T>::to_f32(qv[e]) * da, w = qd * gm;
            const float aa = lds[4 * G + g], bb = lds[5 * G + g], k0 = lds[6 * G + g];
            o1[e] = ET<T>::from_f32(r * (w - aa - xh * bb));
            o2[e] = ET<T>::from_f32(-r * r * (xh * k0 + mm * (w - aa) + bb * (u - ub)));
            o3[e] = ET<T>::from_f32(qd);
            dgl[e] += qd * proj;
          }
        }
        if (!BWD) {
          *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(a.yd) + m * a.ldyd + cg * E16) = o1;
        } else {
          *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(a.dxd) + m * a.lddxd + cg * E16) = o1;
          *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(a.dx) + m * a.lddx + cg * E16) = o2;
          if (a.dresd) *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(a.dresd) + m * a.lddresd + cg * E16) = o3;
        }
      }
      if (BWD && a.dgamma) {
#pragma unroll
        for (int e = 0; e < E16; ++e) atomicAdd(dg + cg * E16 + e, dgl[e]);
      }
    }
  }
  if (BWD && a.dgamma) {
    __syncthreads();
    for (int c = threadIdx.x; c < a.C; c += 256) atomicAdd(a.dgamma + c, dg[c]);
  }
}
// y[o][c] = x[idx[o][c]][c]: the tangent of max-pooling (the primal pass's selection); its backward is ipoke_maxpool3d_bwd
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ x, int ldx, const int* __restrict__ idx, T* __restrict__ y, int ldy, long Mo, int C) {
  const long total = Mo * ldy;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % ldy); const long o = i / ldy;
    y[i] = c < C ? x[(long)idx[o * C + c] * ldx + c] : ET<T>::from_f32(0.f);
  }
}

// ---------------------------------------------------------------------------------------------- feature matching
// loss += scale * sum |a - b|,  grad = scale * sign(a - b)   (fmap_loss of the discriminators, patchgan_3d.py:297-304) on two
// channels-last maps of the compute dtype
template <typename T>
__global__ __launch_bounds__(256) void l1_pair_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb, long M, int C, float scale,
                                                      float* __restrict__ loss, T* __restrict__ grad, int ldg) {
  __shared__ float red[4];
  const long total = M * ldg;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % ldg); const long m = i / ldg;
    float g = 0.f;
    if (c < C) {
      const float d = ET<T>::to_f32(a[m * lda + c]) - ET<T>::to_f32(b[m * ldb + c]);
      s += fabsf(d);
      g = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
    }
    grad[i] = ET<T>::from_f32(g);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) atomicAdd(loss, scale * s);
}

static int grid1(long n, int cap = 2048) { long g = (n + 255) / 256; if (g < 1) g = 1; if (g > cap) g = cap; return (int)g; }

}  // namespace ipoke

using namespace ipoke;
#define STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" int ipoke_conv_weight_operand(const float* w, int cout, int cin, int taps, int transposed, const float* inv_scale, void* out,
                                         int kc, int dtype, void* stream) {
  IPK_REQUIRE(w && out && cout >= 1 && cin >= 1 && taps >= 1 && kc >= cin, "bad arguments");
  IPK_REQUIRE(dtype == IPOKE_F32 || dtype == IPOKE_BF16, "bad dtype");
  const long total = (long)cout * taps * kc;
  if (dtype == IPOKE_BF16)
    hipLaunchKernelGGL(conv_weight_operand_kernel<bf16_t>, dim3(grid1(total)), dim3(256), 0, STREAM(stream), w, cout, cin, taps, transposed,
                       inv_scale, reinterpret_cast<bf16_t*>(out), kc);
  else
    hipLaunchKernelGGL(conv_weight_operand_kernel<float>, dim3(grid1(total)), dim3(256), 0, STREAM(stream), w, cout, cin, taps, transposed,
                       inv_scale, reinterpret_cast<float*>(out), kc);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

/* `count` weights at once: w / out: HOST arrays of device pointers, dims5: HOST array of {cout, cin, taps, transposed, kc} per weight.
 * Same values as `count` ipoke_conv_weight_operand calls without inv_scale. */
extern "C" int ipoke_conv_weight_operand_multi(const float* const* w, void* const* out, const int32_t* dims5, int count, int dtype, void* stream) {
  IPK_REQUIRE(w && out && dims5 && count >= 1, "bad arguments");
  IPK_REQUIRE(dtype == IPOKE_F32 || dtype == IPOKE_BF16, "bad dtype");
  for (int i0 = 0; i0 < count; i0 += kWopJobs) {
    WopJobs J; std::memset(&J, 0, sizeof(J));
    const int n = count - i0 < kWopJobs ? count - i0 : kWopJobs;
    long most = 1;
    for (int i = 0; i < n; ++i) {
      const int32_t* d = dims5 + 5 * (i0 + i);
      IPK_REQUIRE(w[i0 + i] && out[i0 + i] && d[0] >= 1 && d[1] >= 1 && d[2] >= 1 && d[4] >= d[1], "bad weight-operand job");
      J.j[i] = {w[i0 + i], out[i0 + i], d[0], d[1], d[2], d[3], d[4], 0};
      const long total = (long)d[0] * d[2] * d[4];
      if (total > most) most = total;
    }
    const dim3 grid(grid1((most + 3) / 4, 256), n);          // ~4 elements per thread of the largest job; smaller jobs loop less
    if (dtype == IPOKE_BF16) hipLaunchKernelGGL(conv_weight_operand_multi_kernel<bf16_t>, grid, dim3(256), 0, STREAM(stream), J);
    else hipLaunchKernelGGL(conv_weight_operand_multi_kernel<float>, grid, dim3(256), 0, STREAM(stream), J);
    IPK_LAUNCH_CHECK();
  }
  return IPOKE_OK;
}

static SnView sn_view(const float* w, int cout, int cin, int taps, int transposed) {
  // W = weight.reshape(cout, -1), or weight.transpose(0, 1).reshape(cout, -1) for ConvTranspose storage [cin][cout][taps]
  SnView V; V.w = w; V.R = cout; V.C = cin * taps; V.n2 = taps;
  if (!transposed) { V.s_r = (long)cin * taps; V.s_1 = taps; }
  else { V.s_r = taps; V.s_1 = (long)cout * taps; }
  return V;
}

extern "C" long ipoke_spectral_workspace_floats(int cout, int cin, int taps) { return (long)cout + (long)cin * taps + 4; }

/* One power iteration (iterate != 0; u, v updated in place) and sigma = u^T W v; out = {sigma, 1/sigma}; snapshot (optional,
 * cout + cin*taps floats) receives the u | v that sigma was computed with (what the backward needs).  workspace:
 * ipoke_spectral_workspace_floats floats, zero-initialised once by the caller (the kernels leave its accumulators at zero). */
extern "C" int ipoke_spectral_sigma(const float* w, int cout, int cin, int taps, int transposed, float* u, float* v, int iterate, float eps,
                                    float* out, float* snapshot, float* workspace, void* stream) {
  IPK_REQUIRE(w && u && v && out && workspace && cout >= 1 && cin >= 1 && taps >= 1, "bad arguments");
  const SnView V = sn_view(w, cout, cin, taps, transposed);
  float* acc = workspace; float* tu = workspace + 4; float* tv = tu + cout;
  if (iterate) {
    hipLaunchKernelGGL(sn_cols_kernel, dim3((V.C + kSnCols - 1) / kSnCols), dim3(256), 0, STREAM(stream), V, u, tv);
    IPK_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(sn_rows_kernel, dim3((V.R + 3) / 4), dim3(256), 0, STREAM(stream), V, tv, v, tu, acc, iterate, eps);
  IPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(sn_final_kernel, dim3(1), dim3(256), 0, STREAM(stream), V.R, V.C, u, v, tu, acc, out, snapshot, iterate, eps, tv);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

/* Device-side job table of ipoke_spectral_sigma_multi: jobs (HOST array) -> jobs_dev (njobs * ipoke_sn_job_size() bytes owned by the
 * caller).  Synchronises the stream (done once per model, the table holds pointers that stay valid across steps). */
extern "C" int ipoke_sn_job_size(void) { return (int)sizeof(SnJob); }
extern "C" int ipoke_sn_jobs_upload(const ipoke_sn_job* jobs, int njobs, void* jobs_dev, void* stream) {
  IPK_REQUIRE(jobs && jobs_dev && njobs >= 1, "bad arguments");
  std::vector<SnJob> h(njobs);
  for (int i = 0; i < njobs; ++i) {
    const ipoke_sn_job& j = jobs[i];
    IPK_REQUIRE(j.w && j.u && j.v && j.out && j.snap && j.workspace && j.cout >= 1 && j.cin >= 1 && j.taps >= 1, "bad spectral-norm job");
    h[i].V = sn_view(j.w, j.cout, j.cin, j.taps, j.transposed);
    h[i].u = j.u; h[i].v = j.v; h[i].out = j.out; h[i].snap = j.snap; h[i].ws = j.workspace;
    h[i].out_stride = j.out_stride; h[i].snap_stride = j.snap_stride;
  }
  hipStream_t s = STREAM(stream);
  IPK_HIP(hipMemcpyAsync(jobs_dev, h.data(), (size_t)njobs * sizeof(SnJob), hipMemcpyHostToDevice, s));
  IPK_HIP(hipStreamSynchronize(s));
  return IPOKE_OK;
}
/* `iterations` successive power iterations of the `njobs` weights of an uploaded table, three launches per iteration for all of them
 * (max_rows / max_cols: the largest cout / cin*taps among the jobs); iteration k writes {sigma, 1/sigma} to out + k*out_stride and the
 * u | v snapshot to snap + k*snap_stride; u, v end as after the last iteration.  No host synchronisation. */
extern "C" int ipoke_spectral_sigma_multi(const void* jobs_dev, int njobs, int max_rows, int max_cols, int iterations, float eps, void* stream) {
  IPK_REQUIRE(jobs_dev && njobs >= 1 && iterations >= 1 && max_rows >= 1 && max_cols >= 1, "bad arguments");
  hipStream_t s = STREAM(stream);
  const SnJob* jd = reinterpret_cast<const SnJob*>(jobs_dev);
  for (int it = 0; it < iterations; ++it) {
    hipLaunchKernelGGL(sn_cols_multi_kernel, dim3((max_cols + kSnCols - 1) / kSnCols, 1, njobs), dim3(256), 0, s, jd);
    hipLaunchKernelGGL(sn_rows_multi_kernel, dim3((max_rows + 3) / 4, 1, njobs), dim3(256), 0, s, jd, eps);
    hipLaunchKernelGGL(sn_final_multi_kernel, dim3(1, 1, njobs), dim3(256), 0, s, jd, it, eps);
  }
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

/* In place: grad (gradient w.r.t. w_orig / sigma, PyTorch weight layout) -> gradient w.r.t. w_orig.  snapshot / sig: as written by
 * ipoke_spectral_sigma for the forward call; workspace: ipoke_spectral_bwd_workspace_floats() floats (per-workgroup partial sums of
 * <G, W>, no initialisation needed). */
static constexpr int kSnBwdBlocks = 256;
extern "C" long ipoke_spectral_bwd_workspace_floats(void) { return kSnBwdBlocks; }
extern "C" int ipoke_spectral_bwd(const float* w, int cout, int cin, int taps, int transposed, float* grad, const float* snapshot,
                                  const float* sig, float* workspace, void* stream) {
  IPK_REQUIRE(w && grad && snapshot && sig && workspace, "bad arguments");
  const SnView V = sn_view(w, cout, cin, taps, transposed);
  const long total = (long)V.R * V.C;
  const int g = grid1(total, kSnBwdBlocks);
  hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(g), dim3(256), 0, STREAM(stream), V, grad, workspace);
  IPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(g), dim3(256), 0, STREAM(stream), V, grad, snapshot, sig, workspace);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

// rows per block (IPOKE_ROWSCALE_ROWS, developer A/B on c4: 512 / 1024 / 2048 -> 50.3 / 50.7 / 52.1 ms: the final column sums get shorter with larger
// blocks, the pass itself slower by more)
static const int kRowScaleRows = getenv("IPOKE_ROWSCALE_ROWS") ? atoi(getenv("IPOKE_ROWSCALE_ROWS")) : 512;
extern "C" int64_t ipoke_rowscale_bwd_workspace_floats(int64_t M, int C, int64_t rows_per_group) {
  if (rows_per_group < 1 || M < 1) return 0;
  const int64_t ngroups = (M + rows_per_group - 1) / rows_per_group, nbx = (rows_per_group + kRowScaleRows - 1) / kRowScaleRows;
  return ngroups * nbx * (int64_t)(C + 1);
}
/* see ipoke_rowscale_bwd_desc (include/ipoke_hip.h) */
extern "C" int ipoke_rowscale_bwd(const ipoke_rowscale_bwd_desc* d, int dtype, void* stream) {
  IPK_REQUIRE(d && d->dy && d->y && d->gs && d->scale && d->dots && d->workspace, "null tensor");
  IPK_REQUIRE(dtype == IPOKE_F32 || dtype == IPOKE_BF16, "bad dtype");
  const int e16 = dtype == IPOKE_BF16 ? 8 : 4;
  IPK_REQUIRE(d->M >= 1 && d->C >= 1 && d->Cpad >= d->C && d->Cpad % e16 == 0 && d->lddy % e16 == 0 && d->ldy % e16 == 0 && d->ldgs % e16 == 0 &&
              d->lddy >= d->Cpad && d->ldy >= d->Cpad && d->ldgs >= d->Cpad && d->Cpad <= 4096, "pitches must cover the padded channel count in 16-byte steps");
  IPK_REQUIRE(d->rows_per_group >= 1 && d->M % d->rows_per_group == 0 && d->M / d->rows_per_group <= 65535, "rows must divide into the image groups");
  IPK_REQUIRE(d->act == IPOKE_ACT_NONE || d->act == IPOKE_ACT_RELU || d->act == IPOKE_ACT_ELU || d->act == IPOKE_ACT_LRELU02,
              "the pre-activation must be recoverable from the saved output");
  RowScaleBwd a;
  a.dy = d->dy; a.lddy = d->lddy; a.y = d->y; a.ldy = d->ldy; a.M = d->M; a.C = d->C; a.Cpad = d->Cpad; a.act = d->act;
  a.bias = d->bias; a.scale = d->scale; a.scale_stride = d->scale_stride < 1 ? 1 : d->scale_stride; a.rows_per_group = d->rows_per_group;
  a.gs = d->gs; a.ldgs = d->ldgs; a.dots = d->dots; a.dbias = d->dbias;
  a.ngroups = (int)(d->M / d->rows_per_group); a.rows_per_block = kRowScaleRows;
  a.nbx = (int)((d->rows_per_group + kRowScaleRows - 1) / kRowScaleRows);
  a.dot_part = d->workspace; a.col_part = d->dbias ? d->workspace + (int64_t)a.ngroups * a.nbx : nullptr;
  hipStream_t s = STREAM(stream);
  if (dtype == IPOKE_BF16) hipLaunchKernelGGL(rowscale_bwd_kernel<bf16_t>, dim3(a.nbx, a.ngroups), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(rowscale_bwd_kernel<float>, dim3(a.nbx, a.ngroups), dim3(256), 0, s, a);
  IPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(rowscale_final_kernel, dim3(a.ngroups + (d->dbias ? (d->C + 15) / 16 : 0)), dim3(256), 0, s, a);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

// the fixed-order sums of rowscale_final_kernel for partials another kernel wrote (the norm backward that folds this pass in, vae_bwd.hip):
// dot_part[group][nbx], col_part[group * nbx + k][C] (or null)
namespace ipoke {
int rowscale_finalize(float* dot_part, float* col_part, int ngroups, int nbx, int C, float* dots, float* dbias, hipStream_t s) {
  RowScaleBwd a{};
  a.C = C; a.ngroups = ngroups; a.nbx = nbx; a.dot_part = dot_part; a.col_part = col_part; a.dots = dots; a.dbias = dbias;
  hipLaunchKernelGGL(rowscale_final_kernel, dim3(ngroups + (dbias ? (C + 15) / 16 : 0)), dim3(256), 0, s, a);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
}  // namespace ipoke

/* In place: grad (PyTorch weight layout; holds sum_t dW_eff_t / sigma_t) -= sum_t dots[t] / sigma_t * u_t v_t^T.  snapshots[t] = u_t | v_t and
 * sig[t] = {sigma_t, 1 / sigma_t} as written by ipoke_spectral_sigma_multi (strides in floats); dots from ipoke_rowscale_bwd. */
extern "C" int ipoke_spectral_bwd_frames(const float* w, int cout, int cin, int taps, int transposed, float* grad, const float* snapshots,
                                         int64_t snap_stride, const float* sig, int64_t sig_stride, const float* dots, int frames, void* stream) {
  IPK_REQUIRE(w && grad && snapshots && sig && dots && frames >= 1 && sig_stride >= 2, "bad arguments");
  const SnView V = sn_view(w, cout, cin, taps, transposed);
  const long total = (long)V.R * V.C;
  hipLaunchKernelGGL(sn_bwd_frames_kernel, dim3(grid1(total, 2048)), dim3(256), 0, STREAM(stream), V, grad, snapshots, (long)snap_stride, sig,
                     (long)sig_stride, dots, frames);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

/* dst[i] = sum_{f < frames} src[f * n + i], dtype in and out, fp32 accumulation; n a multiple of 16 bytes */
extern "C" int ipoke_sum_frames(const void* src, void* dst, int frames, int64_t n, int dtype, void* stream) {
  IPK_REQUIRE(src && dst && frames >= 1 && n >= 1, "bad arguments");
  const int e16 = dtype == IPOKE_BF16 ? 8 : 4;
  IPK_REQUIRE(n % e16 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "16-byte aligned rows");
  hipStream_t s = STREAM(stream);
  const int g = grid1(n / e16, 4096);
  if (dtype == IPOKE_BF16) hipLaunchKernelGGL(sum_frames_kernel<bf16_t>, dim3(g), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, frames, (long)n);
  else hipLaunchKernelGGL(sum_frames_kernel<float>, dim3(g), dim3(256), 0, s, (const float*)src, (float*)dst, frames, (long)n);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

/* torch.optim.Adam (no amsgrad, coupled weight decay) over `count` tensors given as host arrays of device pointers. */
extern "C" int ipoke_adam_multi(float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n, int count,
                                float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
  IPK_REQUIRE(p && g && m && v && n && count >= 1 && step >= 1, "bad arguments");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
  for (int t0 = 0; t0 < count; t0 += kAdamMax) {
    const int k = count - t0 < kAdamMax ? count - t0 : kAdamMax;
    AdamPack P;
    long nmax = 1;
    for (int i = 0; i < kAdamMax; ++i) {
      const int j = i < k ? t0 + i : t0;
      P.p[i] = p[j]; P.g[i] = g[j]; P.m[i] = m[j]; P.v[i] = v[j]; P.n[i] = i < k ? (long)n[j] : 0;
      IPK_REQUIRE(i >= k || (p[j] && g[j] && m[j] && v[j]), "null tensor");
      if (i < k && n[j] > nmax) nmax = (long)n[j];
    }
    hipLaunchKernelGGL(adam_multi_kernel, dim3(grid1(nmax, 256), k), dim3(256), 0, STREAM(stream), P, lr, beta1, beta2, eps, weight_decay, bc1,
                       bc2s, grad_scale);
    IPK_LAUNCH_CHECK();
  }
  return IPOKE_OK;
}

/* loss[0] += -0.5 * mean_over_positions(sum_c(1 + lv - mu^2 - exp(lv))); dmu / dlv receive d kl / d mu, d kl / d lv.
 * mu, lv, dmu, dlv: fp32 [positions * Z] in any (identical) element order. */
extern "C" int ipoke_kl_loss(const float* mu, const float* lv, int64_t positions, int Z, float* loss, float* dmu, float* dlv, void* stream) {
  IPK_REQUIRE(mu && lv && loss && dmu && dlv && positions >= 1 && Z >= 1, "bad arguments");
  const long n = (long)positions * Z;
  // one workgroup for the usual latents (B * 64 positions x z channels): the value does not depend on the order of float atomics
  hipLaunchKernelGGL(kl_loss_kernel, dim3(n <= (1L << 20) ? 1 : grid1(n, 256)), dim3(256), 0, STREAM(stream), mu, lv, n, 1.f / (float)positions, loss,
                     dmu, dlv);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

static int pool_geom(PoolGeom& g, const int* dims) {
  g.N = dims[0]; g.C = dims[1]; g.Di = dims[2]; g.Hi = dims[3]; g.Wi = dims[4]; g.Do = dims[5]; g.Ho = dims[6]; g.Wo = dims[7];
  g.kd = dims[8]; g.kh = dims[9]; g.kw = dims[10]; g.sd = dims[11]; g.sh = dims[12]; g.sw = dims[13]; g.pd = dims[14]; g.ph = dims[15]; g.pw = dims[16];
  IPK_REQUIRE(g.N >= 1 && g.C >= 1 && g.kd >= 1 && g.kh >= 1 && g.kw >= 1 && g.sd >= 1 && g.sh >= 1 && g.sw >= 1, "bad pooling geometry");
  IPK_REQUIRE((long)g.N * g.Di * g.Hi * g.Wi < (1L << 31), "input rows exceed int32");
  return IPOKE_OK;
}

/* MaxPool3d on channels-last rows.  dims = {N, C, Di, Hi, Wi, Do, Ho, Wo, kd, kh, kw, sd, sh, sw, pd, ph, pw};
 * x [N*Di*Hi*Wi][ldx], y [N*Do*Ho*Wo][ldy] of the compute dtype; idx int32 [N*Do*Ho*Wo][C] = chosen input row. */
extern "C" int ipoke_maxpool3d_fwd(const int* dims, const void* x, int ldx, void* y, int ldy, int* idx, int dtype, void* stream) {
  IPK_REQUIRE(dims && x && y && idx, "null argument");
  PoolGeom g; int rc = pool_geom(g, dims); if (rc) return rc;
  const long total = (long)g.N * g.Do * g.Ho * g.Wo * g.C;
  if (dtype == IPOKE_BF16)
    hipLaunchKernelGGL(maxpool3d_fwd_kernel<bf16_t>, dim3(grid1(total, 4096)), dim3(256), 0, STREAM(stream), g, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, idx);
  else
    hipLaunchKernelGGL(maxpool3d_fwd_kernel<float>, dim3(grid1(total, 4096)), dim3(256), 0, STREAM(stream), g, (const float*)x, ldx, (float*)y, ldy, idx);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
extern "C" int ipoke_maxpool3d_bwd(const int* dims, const void* dy, int ldy, const int* idx, void* dx, int ldx, int dtype, void* stream) {
  IPK_REQUIRE(dims && dy && dx && idx, "null argument");
  PoolGeom g; int rc = pool_geom(g, dims); if (rc) return rc;
  const long total = (long)g.N * g.Di * g.Hi * g.Wi * ldx;
  if (dtype == IPOKE_BF16 && g.C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && (((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)idx) & 15) == 0)
    hipLaunchKernelGGL(maxpool3d_bwd_vec8_kernel, dim3(grid1(total / 8, 8192)), dim3(256), 0, STREAM(stream), g, (const bf16_t*)dy, ldy, idx, (bf16_t*)dx, ldx);
  else if (dtype == IPOKE_BF16)
    hipLaunchKernelGGL(maxpool3d_bwd_kernel<bf16_t>, dim3(grid1(total, 4096)), dim3(256), 0, STREAM(stream), g, (const bf16_t*)dy, ldy, idx, (bf16_t*)dx, ldx);
  else
    hipLaunchKernelGGL(maxpool3d_bwd_kernel<float>, dim3(grid1(total, 4096)), dim3(256), 0, STREAM(stream), g, (const float*)dy, ldy, idx, (float*)dx, ldx);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
/* Mean over the S consecutive rows of each of G groups (AvgPool3d((1, H, W)) on channels-last rows) and its backward. */
extern "C" int ipoke_avgpool_rows(const void* x, int ldx, void* y, int ldy, int64_t G, int S, int C, int dtype, void* stream) {
  IPK_REQUIRE(x && y && G >= 1 && S >= 1 && C >= 1 && ldx >= C && ldy >= C, "bad arguments");
  if (dtype == IPOKE_BF16)
    hipLaunchKernelGGL(avgpool_rows_kernel<bf16_t>, dim3(grid1(G * ldy)), dim3(256), 0, STREAM(stream), (const bf16_t*)x, ldx, (bf16_t*)y, ldy, (long)G, S, C);
  else
    hipLaunchKernelGGL(avgpool_rows_kernel<float>, dim3(grid1(G * ldy)), dim3(256), 0, STREAM(stream), (const float*)x, ldx, (float*)y, ldy, (long)G, S, C);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
extern "C" int ipoke_avgpool_rows_bwd(const void* dy, int ldy, void* dx, int ldx, int64_t G, int S, int C, int dtype, void* stream) {
  IPK_REQUIRE(dy && dx && G >= 1 && S >= 1 && C >= 1 && ldx >= C && ldy >= C, "bad arguments");
  if (dtype == IPOKE_BF16)
    hipLaunchKernelGGL(avgpool_rows_bwd_kernel<bf16_t>, dim3(grid1(G * S * ldx)), dim3(256), 0, STREAM(stream), (const bf16_t*)dy, ldy, (bf16_t*)dx, ldx, (long)G, S, C);
  else
    hipLaunchKernelGGL(avgpool_rows_bwd_kernel<float>, dim3(grid1(G * S * ldx)), dim3(256), 0, STREAM(stream), (const float*)dy, ldy, (float*)dx, ldx, (long)G, S, C);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

/* GroupNorm tangent (see the comment above gn_jvp_sums_kernel). */
static int gn_jvp_fill(GnJvp& a, const void* x, int ldx, const void* xd, int ldxd, const void* y, int ldy, const float* gamma, int N, int S,
                       int C, int G, int act, float eps) {
  IPK_REQUIRE(x && xd && N >= 1 && S >= 1 && C >= 1 && G >= 1 && C % G == 0, "bad GroupNorm tangent arguments");
  IPK_REQUIRE(act == IPOKE_ACT_NONE || y, "the activation mask needs the primal output");
  std::memset(&a, 0, sizeof(a));
  a.x = x; a.ldx = ldx; a.xd = xd; a.ldxd = ldxd; a.y = y; a.ldy = ldy; a.gamma = gamma; a.N = N; a.S = S; a.C = C; a.G = G; a.act = act; a.eps = eps;
  return IPOKE_OK;
}

template <bool BWD>
static int gn_jvp_launch(const GnJvp& a, float* gsum, int dtype, void* stream) {
  const int e16 = dtype == IPOKE_BF16 ? 8 : 4, cpg = a.C / a.G;
  IPK_REQUIRE(a.C % e16 == 0 && (cpg % e16 == 0 || e16 % cpg == 0) && a.ldx % e16 == 0 && a.ldxd % e16 == 0, "channels / pitches: multiples of 16 bytes");
  IPK_REQUIRE(a.G <= kGjMaxG, "GroupNorm tangent: at most 512 groups");
  const int nch = (a.S + kGjPos - 1) / kGjPos;
  const size_t lds = ((size_t)8 * a.G + a.C) * sizeof(float);
  IPK_HIP(hipMemsetAsync(gsum, 0, (size_t)a.N * a.G * 8 * sizeof(float), STREAM(stream)));
  if (dtype == IPOKE_BF16) {
    hipLaunchKernelGGL((gn_jvp_sums_kernel<bf16_t, BWD>), dim3(nch, a.N), dim3(256), 0, STREAM(stream), a, gsum);
    hipLaunchKernelGGL((gn_jvp_apply_kernel<bf16_t, BWD>), dim3(nch, a.N), dim3(256), lds, STREAM(stream), a, gsum);
  } else {
    hipLaunchKernelGGL((gn_jvp_sums_kernel<float, BWD>), dim3(nch, a.N), dim3(256), 0, STREAM(stream), a, gsum);
    hipLaunchKernelGGL((gn_jvp_apply_kernel<float, BWD>), dim3(nch, a.N), dim3(256), lds, STREAM(stream), a, gsum);
  }
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
/* workspace of ipoke_groupnorm_jvp / _bwd: N * G * 8 floats */
extern "C" long ipoke_groupnorm_jvp_workspace_floats(int N, int G) { return (long)N * G * 8; }
extern "C" int ipoke_groupnorm_jvp(const void* x, int ldx, const void* xdot, int ldxd, const void* y, int ldy, const void* resdot, int ldres,
                                   void* ydot, int ldyd, const float* gamma, int N, int S, int C, int G, int act, float eps, float* workspace,
                                   int dtype, void* stream) {
  GnJvp a; int rc = gn_jvp_fill(a, x, ldx, xdot, ldxd, y, ldy, gamma, N, S, C, G, act, eps); if (rc) return rc;
  IPK_REQUIRE(ydot, "null output");
  a.resd = resdot; a.ldres = ldres; a.yd = ydot; a.ldyd = ldyd;
  IPK_REQUIRE(workspace, "null workspace");
  rc = gn_jvp_launch<false>(a, workspace, dtype, stream); if (rc) return rc;
  return IPOKE_OK;
}
/* q: gradient on ydot.  Outputs: dxdot, dx (gradient on the PRIMAL input, through the statistics), dresdot (optional), dgamma
 * (optional, fp32 [C], accumulated atomically: zero it first). */
extern "C" int ipoke_groupnorm_jvp_bwd(const void* x, int ldx, const void* xdot, int ldxd, const void* y, int ldy, const void* q, int ldq,
                                       void* dxdot, int lddxd, void* dx, int lddx, void* dresdot, int lddres, float* dgamma,
                                       const float* gamma, int N, int S, int C, int G, int act, float eps, float* workspace, int dtype,
                                       void* stream) {
  GnJvp a; int rc = gn_jvp_fill(a, x, ldx, xdot, ldxd, y, ldy, gamma, N, S, C, G, act, eps); if (rc) return rc;
  IPK_REQUIRE(q && dxdot && dx, "null tensor");
  a.q = q; a.ldq = ldq; a.dxd = dxdot; a.lddxd = lddxd; a.dx = dx; a.lddx = lddx; a.dresd = dresdot; a.lddresd = lddres; a.dgamma = dgamma;
  IPK_REQUIRE(workspace, "null workspace");
  rc = gn_jvp_launch<true>(a, workspace, dtype, stream); if (rc) return rc;
  return IPOKE_OK;
}
/* y[o][c] = x[idx[o][c]][c] for c < C (zero beyond): the tangent of MaxPool3d given the primal pass's selection. */
extern "C" int ipoke_gather_rows(const void* x, int ldx, const int* idx, void* y, int ldy, int64_t Mo, int C, int dtype, void* stream) {
  IPK_REQUIRE(x && idx && y && Mo >= 1 && C >= 1 && ldy >= C, "bad arguments");
  if (dtype == IPOKE_BF16)
    hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, dim3(grid1(Mo * ldy, 4096)), dim3(256), 0, STREAM(stream), (const bf16_t*)x, ldx, idx, (bf16_t*)y, ldy, (long)Mo, C);
  else
    hipLaunchKernelGGL(gather_rows_kernel<float>, dim3(grid1(Mo * ldy, 4096)), dim3(256), 0, STREAM(stream), (const float*)x, ldx, idx, (float*)y, ldy, (long)Mo, C);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

/* loss[0] += scale * sum_{m, c < C} |a - b|; grad [M][ldg] = scale * sign(a - b) (zero beyond C). */
extern "C" int ipoke_l1_pair(const void* a, int lda, const void* b, int ldb, int64_t M, int C, float scale, float* loss, void* grad, int ldg,
                             int dtype, void* stream) {
  IPK_REQUIRE(a && b && loss && grad && M >= 1 && C >= 1 && lda >= C && ldb >= C && ldg >= C, "bad arguments");
  if (dtype == IPOKE_BF16)
    hipLaunchKernelGGL(l1_pair_kernel<bf16_t>, dim3(grid1(M * ldg, 1024)), dim3(256), 0, STREAM(stream), (const bf16_t*)a, lda, (const bf16_t*)b, ldb,
                       (long)M, C, scale, loss, (bf16_t*)grad, ldg);
  else
    hipLaunchKernelGGL(l1_pair_kernel<float>, dim3(grid1(M * ldg, 1024)), dim3(256), 0, STREAM(stream), (const float*)a, lda, (const float*)b, ldb,
                       (long)M, C, scale, loss, (float*)grad, ldg);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
