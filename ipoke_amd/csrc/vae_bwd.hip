// Backward kernels of the first-stage VAE (training row a18 / config c4): GroupNorm / InstanceNorm / SPADE
// backward, activation backward, bias-gradient column sums, ConvGRU gate backward, reparameterisation backward
// and the L1 reconstruction loss with its gradient.  Reference call sites whose autograd these replace:
// motion_encoder.py:45-74 (GroupNorm+ReLU+residual), autoencoders/util.py:26-36, 223-233, 473-500 (norms, Spade),
// motion_models/rnn.py:48-56 (ConvGRU), motion_encoder.py:218-222 (reparameterize),
// first_stage_motion_model.py:263-272 (L1 term of the loss).
//
// Activations are channels-last [N*S][ld] of the compute dtype T; every reduction is fp32.
#include "common.h"

namespace ipoke {

static inline int grid1d_b(long n, int block = 256, int cap = 4096) {
  long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// out = dy * act'(y)   (y = saved activation output)
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy, T* __restrict__ out, int ldo,
                               long M, int C, int Cpad, int act) {
  const long total = M * Cpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / Cpad; const int c = (int)(i - m * Cpad);
    float v = 0.f;
    if (c < C) v = ET<T>::to_f32(dy[m * lddy + c]) * act_grad_from_out(act, ET<T>::to_f32(y[m * ldy + c]));
    out[m * ldo + c] = ET<T>::from_f32(v);
  }
}

// column sums of a T (or fp32) matrix: part[blk][C] per block of rows, then a second pass over blocks.
// 16-byte loads: thread (rr, cg) owns the E16 columns of group cg over rows rr, rr + rows_par, ...
template <typename T>
__global__ __launch_bounds__(256) void colsum_part_kernel(const T* __restrict__ src, int ld, long M, int C, int rows_per_block,
                                                          float* __restrict__ part) {
  constexpr int E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  __shared__ float sm[256 * 8];
  const int ngroups = (C + E16 - 1) / E16;                       // ld >= ngroups*E16 (host-checked)
  for (int g0 = 0; g0 < ngroups; g0 += blockDim.x) {
    const int groups = ngroups - g0 < (int)blockDim.x ? ngroups - g0 : (int)blockDim.x;
    const int rp = blockDim.x / groups;
    const int cg = threadIdx.x % groups, rr = threadIdx.x / groups;
    float acc[E16];
#pragma unroll
    for (int e = 0; e < E16; ++e) acc[e] = 0.f;
    if (rr < rp) {
      long r = r0 + rr;
      for (; r + 3 * rp < r1; r += 4 * rp) {                     // four rows per trip: all four loads in flight
        const frag_t a = *reinterpret_cast<const frag_t*>(src + r * ld + (g0 + cg) * E16);
        const frag_t b = *reinterpret_cast<const frag_t*>(src + (r + rp) * ld + (g0 + cg) * E16);
        const frag_t c = *reinterpret_cast<const frag_t*>(src + (r + 2 * rp) * ld + (g0 + cg) * E16);
        const frag_t d = *reinterpret_cast<const frag_t*>(src + (r + 3 * rp) * ld + (g0 + cg) * E16);
#pragma unroll
        for (int e = 0; e < E16; ++e) acc[e] += (ET<T>::to_f32(a[e]) + ET<T>::to_f32(b[e])) + (ET<T>::to_f32(c[e]) + ET<T>::to_f32(d[e]));
      }
      for (; r < r1; r += rp) {
        const frag_t a = *reinterpret_cast<const frag_t*>(src + r * ld + (g0 + cg) * E16);
#pragma unroll
        for (int e = 0; e < E16; ++e) acc[e] += ET<T>::to_f32(a[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < E16; ++e) sm[threadIdx.x * E16 + e] = rr < rp ? acc[e] : 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < groups * E16; i += blockDim.x) {
      const int cgi = i / E16, e = i - cgi * E16;
      float t = 0.f;
      for (int k = 0; k < rp; ++k) t += sm[(k * groups + cgi) * E16 + e];
      const int c = (g0 + cgi) * E16 + e;
      if (c < C) part[(long)blockIdx.x * C + c] = t;
    }
    __syncthreads();
  }
}
// 16 columns x 16 row lanes per block: a lane sums every 16th partial row, then a fixed-order LDS reduction (one thread per
// column walking all partial rows serially took 11 us at 1 280 rows; this is a latency kernel)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ out, int accumulate) {
  __shared__ float red[256];
  const int cl = threadIdx.x % 16, rl = threadIdx.x / 16;
  const int c = blockIdx.x * 16 + cl;
  float t0 = 0.f, t1 = 0.f;
  if (c < C) {
    int b = rl;
    for (; b + 16 < nblk; b += 32) { t0 += part[(long)b * C + c]; t1 += part[(long)(b + 16) * C + c]; }
    if (b < nblk) t0 += part[(long)b * C + c];
  }
  red[threadIdx.x] = t0 + t1;
  __syncthreads();
  if (rl == 0 && c < C) {
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += red[k * 16 + cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}

// ---------------------------------------------------------------------------------------------- GroupNorm backward
struct NormBwd {
  const void* x; int ldx; const void* y; int ldy; const void* dy; int lddy;
  void* dx; int lddx; void* dres; int lddres; void* dmg; void* dmb; int ld_dmod;
  int N, S, C, G;
  const float* stats;                  // [N][G][2] mean, rstd (recomputed)
  const float* gamma; const float* beta;
  const void* mod_gamma; int ld_mod; int mod_N;     // mod_N > 0: sample n reads the modulation rows of sample n % mod_N
  int act;
  int act_from_pre;                    // 1: act' from the recomputed pre-activation value (forward: y = act(pre) + res, ipoke_norm_desc.res_post)
  float* part;                         // [N][nchunks][2][C]  (sum du, sum du*xhat)
  float* gsum;                         // [N][G][2]           (S1 = sum dxhat, S2 = sum dxhat*xhat)
  int nchunks, pos_per_block;
  // folded ipoke_rowscale_bwd (ipoke_norm_bwd_desc.rs_*): frame of sample n = n / rs_clips
  const float* rs_scale; int rs_scale_stride, rs_clips, rs_pos_per_block, rs_nchunks;
  const float* rs_bias; float* rs_dot_part; float* rs_col_part;
};
int rowscale_finalize(float* dot_part, float* col_part, int ngroups, int nbx, int C, float* dots, float* dbias, hipStream_t s);   // vae_train.hip
// du for one element: dw = dy*act'(y); du = dw*(1+mg)
template <typename T>
__device__ __forceinline__ float norm_dw(const NormBwd& a, long m, int c) {
  const float g = ET<T>::to_f32(reinterpret_cast<const T*>(a.dy)[m * a.lddy + c]);
  if (a.act == IPOKE_ACT_NONE) return g;
  return g * act_grad_from_out(a.act, ET<T>::to_f32(reinterpret_cast<const T*>(a.y)[m * a.ldy + c]));
}
// pass 1: per (n, chunk) and channel: sum_p du, sum_p du*xhat.  Thread (rr, cg) owns the E16 channels of column group cg
// over rows rr, rr + rows_par, ... (16-byte loads, register partial sums, one fixed-order LDS reduction).
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const NormBwd a) {
  if (IPK_KERNARG_PREFETCH) kernarg_prefetch<(int)sizeof(NormBwd)>();
  constexpr int E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  __shared__ float sm[2][256 * E16];
  const int n = blockIdx.y, chunk = blockIdx.x;
  const long mod_row0 = (long)(a.mod_N > 0 ? n % a.mod_N : n) * a.S;
  const int p0 = chunk * a.pos_per_block, p1 = min(a.S, p0 + a.pos_per_block);
  const int cpg = a.C / a.G;
  const int cvec = a.C / E16;
  for (int g0 = 0; g0 < cvec; g0 += 256) {
    const int groups = cvec - g0 < 256 ? cvec - g0 : 256;
    const int rp = 256 / groups;
    const int cg = g0 + threadIdx.x % groups, rr = threadIdx.x / groups;
    float s1[E16], s2[E16], mean[E16], rstd[E16], fsc[E16], fsh[E16];
#pragma unroll
    for (int e = 0; e < E16; ++e) {
      s1[e] = 0.f; s2[e] = 0.f;
      const int g = (cg * E16 + e) / cpg;
      mean[e] = a.stats[((long)n * a.G + g) * 2]; rstd[e] = a.stats[((long)n * a.G + g) * 2 + 1];
      fsc[e] = 0.f; fsh[e] = 0.f;
      if (a.act_from_pre) {                 // the forward pass's own scale / shift (norm.hip gn_apply_kernel): pre = x * fsc + fsh
        const float gm = a.gamma ? a.gamma[cg * E16 + e] : 1.f, bt = a.beta ? a.beta[cg * E16 + e] : 0.f;
        fsc[e] = rstd[e] * gm; fsh[e] = bt - mean[e] * rstd[e] * gm;
      }
    }
    const bool need_y = a.act != IPOKE_ACT_NONE && !a.act_from_pre;
    if (rr < rp) {
      for (int p = p0 + rr; p < p1; p += rp) {
        const long m = (long)n * a.S + p;
        const frag_t gy = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.dy) + m * a.lddy + cg * E16);
        const frag_t xv = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.x) + m * a.ldx + cg * E16);
        frag_t yv, mg;
        if (need_y) yv = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.y) + m * a.ldy + cg * E16);
        if (a.mod_gamma) mg = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.mod_gamma) + (mod_row0 + p) * a.ld_mod + cg * E16);
#pragma unroll
        for (int e = 0; e < E16; ++e) {
          float du = ET<T>::to_f32(gy[e]);
          if (need_y) du *= act_grad_from_out(a.act, ET<T>::to_f32(yv[e]));
          else if (a.act_from_pre) du *= act_grad_from_pre(a.act, ET<T>::to_f32(xv[e]) * fsc[e] + fsh[e]);
          if (a.mod_gamma) du *= 1.f + ET<T>::to_f32(mg[e]);
          const float xh = (ET<T>::to_f32(xv[e]) - mean[e]) * rstd[e];
          s1[e] += du; s2[e] += du * xh;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < E16; ++e) {
      sm[0][threadIdx.x * E16 + e] = rr < rp ? s1[e] : 0.f;
      sm[1][threadIdx.x * E16 + e] = rr < rp ? s2[e] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < groups * E16; i += 256) {
      const int cgi = i / E16, e = i - cgi * E16;
      float t1 = 0.f, t2 = 0.f;
      for (int k = 0; k < rp; ++k) { t1 += sm[0][(k * groups + cgi) * E16 + e]; t2 += sm[1][(k * groups + cgi) * E16 + e]; }
      float* o = a.part + (((long)n * a.nchunks + chunk) * 2) * a.C;
      o[(g0 + cgi) * E16 + e] = t1; o[a.C + (g0 + cgi) * E16 + e] = t2;
    }
    __syncthreads();
  }
}
// pass 2a: per (n, g): S1 = sum_c gamma_c * sum du, S2 = sum_c gamma_c * sum du*xhat.  One block per sample, 16 threads per group.
__device__ __forceinline__ void gn_bwd_groupsum_body(const NormBwd& a, int n, float* r1, float* r2) {
  const int cpg = a.C / a.G;
  constexpr int PAR = 16;
  const int gl = threadIdx.x / PAR, pr = threadIdx.x % PAR;
  for (int g0 = 0; g0 < a.G; g0 += 256 / PAR) {
    const int g = g0 + gl;
    float S1 = 0.f, S2 = 0.f;
    if (g < a.G) {
      for (int k = pr; k < a.nchunks * cpg; k += PAR) {
        const int ch = k / cpg, c = g * cpg + (k - ch * cpg);
        const float* o = a.part + (((long)n * a.nchunks + ch) * 2) * a.C;
        const float gm = a.gamma ? a.gamma[c] : 1.f;
        S1 += gm * o[c]; S2 += gm * o[a.C + c];
      }
    }
    r1[threadIdx.x] = S1; r2[threadIdx.x] = S2;
    __syncthreads();
    if (pr == 0 && g < a.G) {
      float t1 = 0.f, t2 = 0.f;
      for (int k = 0; k < PAR; ++k) { t1 += r1[gl * PAR + k]; t2 += r2[gl * PAR + k]; }
      a.gsum[((long)n * a.G + g) * 2] = t1; a.gsum[((long)n * a.G + g) * 2 + 1] = t2;
    }
    __syncthreads();
  }
}
// pass 2b: dgamma / dbeta per channel over all samples and chunks; 16 channels x 16 row lanes per block
__device__ __forceinline__ void gn_bwd_affine_body(const NormBwd& a, int blk, float* __restrict__ dgamma, float* __restrict__ dbeta, float* r1,
                                                   float* r2) {
  const int cl = threadIdx.x % 16, rl = threadIdx.x / 16;
  const int c = blk * 16 + cl;
  float t1 = 0.f, t2 = 0.f;
  if (c < a.C) {
    for (int k = rl; k < a.N * a.nchunks; k += 16) {
      const float* o = a.part + ((long)k * 2) * a.C;
      t1 += o[c]; t2 += o[a.C + c];
    }
  }
  r1[threadIdx.x] = t1; r2[threadIdx.x] = t2;
  __syncthreads();
  if (rl == 0 && c < a.C) {
    float u1 = 0.f, u2 = 0.f;
    for (int k = 0; k < 16; ++k) { u1 += r1[k * 16 + cl]; u2 += r2[k * 16 + cl]; }
    dbeta[c] = u1; dgamma[c] = u2;
  }
}
// passes 2a and 2b as ONE launch (round 5): workgroups [0, N) sum the groups of a sample, the rest the affine gradients of 16 channels --
// two independent reductions over the same partials that used to cost two dependent launches per norm on the chain's queue
__global__ __launch_bounds__(256) void gn_bwd_sums_kernel(const NormBwd a, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  if (IPK_KERNARG_PREFETCH) kernarg_prefetch<(int)sizeof(NormBwd)>();
  __shared__ float r1[256], r2[256];
  if ((int)blockIdx.x < a.N) gn_bwd_groupsum_body(a, blockIdx.x, r1, r2);
  else gn_bwd_affine_body(a, blockIdx.x - a.N, dgamma, dbeta, r1, r2);
}
// pass 3: dx = rstd * (dxhat - (S1 + xhat*S2)/cnt), plus the residual / modulation gradients.  Grid (chunks, N): the
// per-channel constants of the sample live in LDS, the inner loop is 16-byte loads / stores without divisions.
// RS (round 6): x is the un-activated output of a frame-batched spectral-norm convolution that only this norm reads -- the pass
// ipoke_rowscale_bwd would make over (dx, x) right behind this kernel (5 of the 15 such passes of a first-stage step, the ones at a
// block's output resolution) happens on the registers here: dx leaves as round(dx) / sigma_t, the frame's <dx, x - b> and the column sums
// go to per-block partials (blocks of rs_pos_per_block positions; fixed-order sums in rowscale_final_kernel).
template <typename T, bool RS>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const NormBwd a) {
  if (IPK_KERNARG_PREFETCH) kernarg_prefetch<(int)sizeof(NormBwd)>();
  constexpr int E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  extern __shared__ float lds[];            // [8][C]: mean, rstd, gamma, beta, S1/cnt, S2/cnt, forward scale, shift  (RS: + [256 * E16] column sums + [4])
  const int n = blockIdx.y, cpg = a.C / a.G, C = a.C;
  const long mod_row0 = (long)(a.mod_N > 0 ? n % a.mod_N : n) * a.S;
  const float inv_cnt = 1.f / ((float)a.S * cpg);
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    lds[c] = a.stats[((long)n * a.G + g) * 2]; lds[C + c] = a.stats[((long)n * a.G + g) * 2 + 1];
    lds[2 * C + c] = a.gamma ? a.gamma[c] : 1.f; lds[3 * C + c] = a.beta ? a.beta[c] : 0.f;
    lds[4 * C + c] = a.gsum[((long)n * a.G + g) * 2] * inv_cnt; lds[5 * C + c] = a.gsum[((long)n * a.G + g) * 2 + 1] * inv_cnt;
    if (a.act_from_pre) {                   // the forward pass's own scale / shift (norm.hip gn_apply_kernel): pre = x * fsc + fsh
      lds[6 * C + c] = lds[C + c] * lds[2 * C + c]; lds[7 * C + c] = lds[3 * C + c] - lds[c] * lds[C + c] * lds[2 * C + c];
    }
  }
  __syncthreads();
  const int cvec = C / E16;
  const int ppb = RS ? a.rs_pos_per_block : a.pos_per_block;
  const int p0 = blockIdx.x * ppb, p1 = min(a.S, p0 + ppb);
  float* sm = lds + (a.act_from_pre ? 8 : 6) * C;
  float rs_sc = 0.f, dot = 0.f;
  float* cpart = nullptr;
  if (RS) {
    rs_sc = a.rs_scale[(long)(n / a.rs_clips) * a.rs_scale_stride];
    if (a.rs_col_part) cpart = a.rs_col_part + ((long)n * a.rs_nchunks + blockIdx.x) * C;
  }
  for (int g0 = 0; g0 < cvec; g0 += 256) {
    const int groups = cvec - g0 < 256 ? cvec - g0 : 256;
    const int rp = 256 / groups;
    const int cg = g0 + threadIdx.x % groups, rr = threadIdx.x / groups;
    float cs[E16], bb[E16];
    if (RS) {
#pragma unroll
      for (int e = 0; e < E16; ++e) { cs[e] = 0.f; bb[e] = a.rs_bias ? a.rs_bias[cg * E16 + e] : 0.f; }
    }
    if (!RS && rr >= rp) continue;
    if (rr < rp)
    for (int p = p0 + rr; p < p1; p += rp) {
      const long m = (long)n * a.S + p;
      const frag_t gy = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.dy) + m * a.lddy + cg * E16);
      const frag_t xv = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.x) + m * a.ldx + cg * E16);
      frag_t yv, mg;
      const bool need_y = a.act != IPOKE_ACT_NONE && !a.act_from_pre;
      if (need_y) yv = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.y) + m * a.ldy + cg * E16);
      if (a.mod_gamma) mg = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.mod_gamma) + (mod_row0 + p) * a.ld_mod + cg * E16);
      frag_t odx, odw, odmg;
#pragma unroll
      for (int e = 0; e < E16; ++e) {
        const int c = cg * E16 + e;
        float dw = ET<T>::to_f32(gy[e]);
        if (need_y) dw *= act_grad_from_out(a.act, ET<T>::to_f32(yv[e]));
        else if (a.act_from_pre) dw *= act_grad_from_pre(a.act, ET<T>::to_f32(xv[e]) * lds[6 * C + c] + lds[7 * C + c]);
        const float xh = (ET<T>::to_f32(xv[e]) - lds[c]) * lds[C + c];
        float du = dw;
        if (a.mod_gamma) {
          du = dw * (1.f + ET<T>::to_f32(mg[e]));
          odmg[e] = ET<T>::from_f32(dw * (xh * lds[2 * C + c] + lds[3 * C + c]));
        }
        odw[e] = ET<T>::from_f32(dw);
        const float dxv = lds[C + c] * (du * lds[2 * C + c] - (lds[4 * C + c] + xh * lds[5 * C + c]));
        if (RS) {
          const float g = ET<T>::to_f32(ET<T>::from_f32(dxv));        // the value the un-fused pass reads back
          if (g != 0.f) dot = fmaf(g, ET<T>::to_f32(xv[e]) - bb[e], dot);
          cs[e] += g;
          odx[e] = ET<T>::from_f32(g * rs_sc);
        } else {
          odx[e] = ET<T>::from_f32(dxv);
        }
      }
      *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(a.dx) + m * a.lddx + cg * E16) = odx;
      if (a.dres) *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(a.dres) + m * a.lddres + cg * E16) = odw;
      if (a.mod_gamma) {
        *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(a.dmg) + m * a.ld_dmod + cg * E16) = odmg;
        *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(a.dmb) + m * a.ld_dmod + cg * E16) = odw;
      }
    }
    if (RS && cpart) {
#pragma unroll
      for (int e = 0; e < E16; ++e) sm[threadIdx.x * E16 + e] = rr < rp ? cs[e] : 0.f;
      __syncthreads();
      for (int i = threadIdx.x; i < groups * E16; i += 256) {
        const int cgi = i / E16, e = i - cgi * E16;
        float t = 0.f;
        for (int k = 0; k < rp; ++k) t += sm[(k * groups + cgi) * E16 + e];
        cpart[(g0 + cgi) * E16 + e] = t;
      }
      __syncthreads();
    }
  }
  if (RS) {
    dot = block_sum(dot, sm + 256 * E16);
    if (threadIdx.x == 0) a.rs_dot_part[(long)n * a.rs_nchunks + blockIdx.x] = dot;
  }
}

// The same pass for SPADE norms whose modulation maps are shared by the frames of a clip (samples ordered (frame, clip), mod_N = clips;
// ipoke_norm_bwd_desc.dmod_summed): a workgroup owns a block of positions of ONE clip and walks the frames, so that the gradients of the
// shared maps accumulate in registers (fp32, frame order) and leave once.  The per-sample form wrote two maps per SAMPLE and
// ipoke_sum_frames read them all back: at the 128 x 128 x 64 level of a first-stage step 2 x 629 MB written + read per step for 2 x 42 MB
// of result.  Thread (rr, cg) owns E16 channels of ROWS positions (16 accumulated channels-positions x 2 maps per thread: more costs the
// occupancy that hides the per-frame constant reload); a block is ROWS * (256 / (C / E16)) positions.
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_frames_kernel(const NormBwd a) {
  if (IPK_KERNARG_PREFETCH) kernarg_prefetch<(int)sizeof(NormBwd)>();
  constexpr int E16 = ET<T>::E16, ROWS = 16 / E16;
  typedef typename ET<T>::frag frag_t;
  extern __shared__ float lds[];            // [6][C]: mean, rstd, gamma, beta, S1/cnt, S2/cnt
  const int clip = blockIdx.y, clips = a.mod_N, frames = a.N / clips, cpg = a.C / a.G, C = a.C;
  const float inv_cnt = 1.f / ((float)a.S * cpg);
  const int cvec = C / E16, rp = 256 / cvec;
  const int cg = threadIdx.x % cvec, rr = threadIdx.x / cvec;
  const int p0 = blockIdx.x * (ROWS * rp);
  for (int c = threadIdx.x; c < C; c += 256) { lds[2 * C + c] = a.gamma ? a.gamma[c] : 1.f; lds[3 * C + c] = a.beta ? a.beta[c] : 0.f; }
  float adg[ROWS][E16], adb[ROWS][E16];
  frag_t mg[ROWS];
  bool live[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int p = p0 + rr + i * rp;
    live[i] = rr < rp && p < a.S;
    if (live[i]) mg[i] = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.mod_gamma) + ((long)clip * a.S + p) * a.ld_mod + cg * E16);
#pragma unroll
    for (int e = 0; e < E16; ++e) { adg[i][e] = 0.f; adb[i][e] = 0.f; }
  }
#pragma unroll 1
  for (int t = 0; t < frames; ++t) {
    const int n = t * clips + clip;
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      const int g = c / cpg;
      lds[c] = a.stats[((long)n * a.G + g) * 2]; lds[C + c] = a.stats[((long)n * a.G + g) * 2 + 1];
      lds[4 * C + c] = a.gsum[((long)n * a.G + g) * 2] * inv_cnt; lds[5 * C + c] = a.gsum[((long)n * a.G + g) * 2 + 1] * inv_cnt;
    }
    __syncthreads();
    frag_t gy[ROWS], xv[ROWS], yv[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      if (!live[i]) continue;
      const long m = (long)n * a.S + p0 + rr + i * rp;
      gy[i] = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.dy) + m * a.lddy + cg * E16);
      xv[i] = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.x) + m * a.ldx + cg * E16);
      if (a.act != IPOKE_ACT_NONE) yv[i] = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.y) + m * a.ldy + cg * E16);
    }
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      if (!live[i]) continue;
      const long m = (long)n * a.S + p0 + rr + i * rp;
      frag_t odx, odw;
#pragma unroll
      for (int e = 0; e < E16; ++e) {
        const int c = cg * E16 + e;
        float dw = ET<T>::to_f32(gy[i][e]);
        if (a.act != IPOKE_ACT_NONE) dw *= act_grad_from_out(a.act, ET<T>::to_f32(yv[i][e]));
        const float xh = (ET<T>::to_f32(xv[i][e]) - lds[c]) * lds[C + c];
        const float du = dw * (1.f + ET<T>::to_f32(mg[i][e]));
        adg[i][e] += dw * (xh * lds[2 * C + c] + lds[3 * C + c]);
        adb[i][e] += dw;
        odw[e] = ET<T>::from_f32(dw);
        odx[e] = ET<T>::from_f32(lds[C + c] * (du * lds[2 * C + c] - (lds[4 * C + c] + xh * lds[5 * C + c])));
      }
      *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(a.dx) + m * a.lddx + cg * E16) = odx;
      if (a.dres) *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(a.dres) + m * a.lddres + cg * E16) = odw;
    }
  }
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    if (!live[i]) continue;
    const long m = (long)clip * a.S + p0 + rr + i * rp;
    frag_t og, ob;
#pragma unroll
    for (int e = 0; e < E16; ++e) { og[e] = ET<T>::from_f32(adg[i][e]); ob[e] = ET<T>::from_f32(adb[i][e]); }
    *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(a.dmg) + m * a.ld_dmod + cg * E16) = og;
    *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(a.dmb) + m * a.ld_dmod + cg * E16) = ob;
  }
}

// ---------------------------------------------------------------------------------------------- ConvGRU backward
// update: h' = h*(1-u) + tanh(o_pre)*u
template <typename T>
__global__ void gru_update_bwd_kernel(const T* __restrict__ o_pre, const T* __restrict__ u, const T* __restrict__ h, int ldh,
                                      const T* __restrict__ dhn, int lddhn, T* __restrict__ do_pre, T* __restrict__ du,
                                      T* __restrict__ dh, int lddh, long M, int Ch) {
  const long total = M * Ch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / Ch; const int c = (int)(i - m * Ch);
    const float g = ET<T>::to_f32(dhn[m * lddhn + c]);
    const float uu = ET<T>::to_f32(u[m * Ch + c]);
    const float o = tanhf(ET<T>::to_f32(o_pre[m * Ch + c]));
    const float hv = ET<T>::to_f32(h[m * ldh + c]);
    do_pre[m * Ch + c] = ET<T>::from_f32(g * uu * (1.f - o * o));
    du[m * Ch + c] = ET<T>::from_f32(g * (o - hv));
    dh[m * lddh + c] = ET<T>::from_f32(g * (1.f - uu));
  }
}
// gates: u = sigmoid(u_pre), r = sigmoid(r_pre), hr = h*r.  Inputs d_hr, d_u; outputs d_ur_pre [M][2Ch], dh
template <typename T>
__global__ void gru_gates_bwd_kernel(const T* __restrict__ ur_pre, const T* __restrict__ h, int ldh, const T* __restrict__ d_hr,
                                     int ld_dhr, const T* __restrict__ d_u, T* __restrict__ d_ur_pre, T* __restrict__ dh, int lddh,
                                     long M, int Ch) {
  const long total = M * Ch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / Ch; const int c = (int)(i - m * Ch);
    const float u = act_apply(IPOKE_ACT_SIGMOID, ET<T>::to_f32(ur_pre[m * 2 * Ch + c]));
    const float r = act_apply(IPOKE_ACT_SIGMOID, ET<T>::to_f32(ur_pre[m * 2 * Ch + Ch + c]));
    const float hv = ET<T>::to_f32(h[m * ldh + c]);
    const float ghr = ET<T>::to_f32(d_hr[m * ld_dhr + c]);
    const float gu = d_u ? ET<T>::to_f32(d_u[m * Ch + c]) : 0.f;
    d_ur_pre[m * 2 * Ch + c] = ET<T>::from_f32(gu * u * (1.f - u));
    d_ur_pre[m * 2 * Ch + Ch + c] = ET<T>::from_f32(ghr * hv * r * (1.f - r));
    dh[m * lddh + c] = ET<T>::from_f32(ghr * r);
  }
}
// z = mu + eps*exp(lv/2):  dmulv = [dz + dmu | dz*eps*0.5*exp(lv/2) + dlv]
template <typename T>
__global__ void reparam_bwd_kernel(const T* __restrict__ mulv, int ld, const float* __restrict__ eps, const float* __restrict__ dz,
                                   const float* __restrict__ dmu, const float* __restrict__ dlv, T* __restrict__ dmulv, int ldo,
                                   long M, int Z) {
  const long total = M * ldo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / ldo; const int c = (int)(i - m * ldo);
    float v = 0.f;
    if (c < Z) {
      v = (dz ? dz[m * Z + c] : 0.f) + (dmu ? dmu[m * Z + c] : 0.f);
    } else if (c < 2 * Z) {
      const int k = c - Z;
      const float lv = ET<T>::to_f32(mulv[m * ld + c]);
      v = (dlv ? dlv[m * Z + k] : 0.f);
      if (dz && eps) v += dz[m * Z + k] * eps[m * Z + k] * 0.5f * expf(0.5f * lv);
    }
    dmulv[i] = ET<T>::from_f32(v);
  }
}
// L1: loss += scale * sum |yhat - x| ; grad = scale * sign(yhat - x).  yhat channels-last fp32 [N*S][ldy], x fp32 [N][C][S]
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ yhat, int ldy, const float* __restrict__ x, int N, int C,
                                                      int S, long x_sn, float scale, float* __restrict__ loss, float* __restrict__ grad,
                                                      int ldg, float* __restrict__ partials) {
  __shared__ float red[8];
  const long total = (long)N * S * C;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); const long m = i / C;
    const int n = (int)(m / S), s = (int)(m % S);
    const float d = yhat[m * ldy + c] - x[(long)n * x_sn + (long)c * S + s];
    acc += fabsf(d);
    if (grad) grad[m * ldg + c] = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) {
    if (partials) partials[blockIdx.x] = tot * scale;     // summed in a fixed order by l1_loss_final_kernel
    else atomicAdd(loss, tot * scale);
  }
}
__global__ __launch_bounds__(256) void l1_loss_final_kernel(const float* __restrict__ partials, int n, float* __restrict__ loss) {
  __shared__ float red[8];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) loss[0] += s;
}

}  // namespace ipoke

using namespace ipoke;
#define STREAM(s) reinterpret_cast<hipStream_t>(s)
#define DISPATCH_T(dtype, CALL_BF, CALL_F32) do { if ((dtype) == IPOKE_BF16) { CALL_BF; } else { CALL_F32; } } while (0)

extern "C" int ipoke_act_bwd(const void* dy, int lddy, const void* y, int ldy, void* out, int ldo, int64_t M, int C, int Cpad,
                             int act, int dtype, void* stream) {
  IPK_REQUIRE(dy && y && out && Cpad >= C && ldo >= Cpad, "bad arguments");
  const long total = (long)M * Cpad;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(grid1d_b(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)dy, lddy, (const bf16_t*)y, ldy, (bf16_t*)out, ldo, (long)M, C, Cpad, act),
    hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(grid1d_b(total)), dim3(256), 0, STREAM(stream), (const float*)dy, lddy, (const float*)y, ldy, (float*)out, ldo, (long)M, C, Cpad, act));
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

static const int kColsumRows = 512;      // 640 blocks at the 128x128 maps of B = 20 (2048 rows left 160 blocks of 32-trip latency chains: 24 us)
extern "C" int64_t ipoke_colsum_workspace_floats(int64_t M, int C) { return ((M + kColsumRows - 1) / kColsumRows) * (int64_t)C; }
/* out[c] (+)= sum_m src[m][c]; src of dtype (src_f32 = 1: fp32) */
extern "C" int ipoke_colsum(const void* src, int ld, int64_t M, int C, int src_f32, float* out, int accumulate, float* workspace,
                            int dtype, void* stream) {
  IPK_REQUIRE(src && out && workspace && M >= 1 && C >= 1, "bad arguments");
  {
    const int e16 = (src_f32 || dtype == IPOKE_F32) ? 4 : 8;
    IPK_REQUIRE(ld % e16 == 0 && ld >= (C + e16 - 1) / e16 * e16, "row pitch must cover the channel count rounded up to 16 bytes");
  }
  const int nblk = (int)((M + kColsumRows - 1) / kColsumRows);
  hipStream_t s = STREAM(stream);
  if (src_f32 || dtype == IPOKE_F32)
    hipLaunchKernelGGL(colsum_part_kernel<float>, dim3(nblk), dim3(256), 0, s, (const float*)src, ld, (long)M, C, kColsumRows, workspace);
  else
    hipLaunchKernelGGL(colsum_part_kernel<bf16_t>, dim3(nblk), dim3(256), 0, s, (const bf16_t*)src, ld, (long)M, C, kColsumRows, workspace);
  IPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(256), 0, s, workspace, nblk, C, out, accumulate);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

// positions per workgroup of the two passes (IPOKE_NORM_BWD_POS: developer A/B)
static const int kNormBwdPos = getenv("IPOKE_NORM_BWD_POS") ? atoi(getenv("IPOKE_NORM_BWD_POS")) : 128;
static const int kNormRsPos = 512;          // positions per block of the pass that folds ipoke_rowscale_bwd in (that kernel's own block size)
extern "C" int64_t ipoke_groupnorm_bwd_rs_workspace_floats(int N, int S, int C) {
  return (int64_t)N * ((S + kNormRsPos - 1) / kNormRsPos) * (C + 1);
}
extern "C" int64_t ipoke_groupnorm_bwd_workspace_floats(int N, int S, int C, int G) {
  const int nchunks = (S + kNormBwdPos - 1) / kNormBwdPos;
  return ipoke_groupnorm_workspace_floats(N, S, G) + (int64_t)N * nchunks * 2 * C + (int64_t)N * G * 2;
}

// forward-statistics kernels of norm.hip (same translation-unit-independent entry point)
extern "C" int ipoke_groupnorm_stats(const void* x, int ldx, int N, int S, int C, int G, float eps, float* workspace, int dtype,
                                     void* stream);

extern "C" int ipoke_groupnorm_bwd(const ipoke_norm_bwd_desc* d, int dtype, void* stream) {
  IPK_REQUIRE(d && d->x && d->dy && d->dx && d->workspace, "null tensor");
  IPK_REQUIRE(d->act == IPOKE_ACT_NONE || d->y, "the activation gradient needs the saved output");
  IPK_REQUIRE(d->C % d->G == 0, "channels must be a multiple of the group count");
  IPK_REQUIRE((d->mod_gamma == nullptr) == (d->dmod_gamma == nullptr) && (d->dmod_gamma == nullptr) == (d->dmod_beta == nullptr),
              "modulation gradients come as a pair, with the saved modulation");
  IPK_REQUIRE((d->dgamma == nullptr) == (d->dbeta == nullptr) && (!d->dgamma || d->gamma), "affine gradient pair");
  hipStream_t s = STREAM(stream);
  const int e16 = dtype == IPOKE_BF16 ? 8 : 4;
  IPK_REQUIRE(d->C % e16 == 0 && d->ldx % e16 == 0 && d->lddy % e16 == 0 && d->lddx % e16 == 0 && (!d->y || d->ldy % e16 == 0) &&
              (!d->dres || d->lddres % e16 == 0) && (!d->mod_gamma || (d->ld_mod % e16 == 0 && d->ld_dmod % e16 == 0)) && d->C <= 4096,
              "channels and pitches must be multiples of 16 bytes");
  const int ppb_f = 128, nch_f = (d->S + ppb_f - 1) / ppb_f;
  NormBwd a{};
  if (d->stats) {
    a.stats = d->stats;                    // (mean, rstd) saved by the forward pass
  } else {                                 // recomputed from the saved input, exactly as in the forward pass
    int rc = ipoke_groupnorm_stats(d->x, d->ldx, d->N, d->S, d->C, d->G, d->eps, d->workspace, dtype, stream);
    if (rc) return rc;
    a.stats = d->workspace + (int64_t)d->N * nch_f * d->G * 3;
  }
  a.x = d->x; a.ldx = d->ldx; a.y = d->y; a.ldy = d->ldy; a.dy = d->dy; a.lddy = d->lddy;
  a.dx = d->dx; a.lddx = d->lddx; a.dres = d->dres; a.lddres = d->lddres; a.dmg = d->dmod_gamma; a.dmb = d->dmod_beta;
  a.ld_dmod = d->ld_dmod; a.N = d->N; a.S = d->S; a.C = d->C; a.G = d->G;
  a.gamma = d->gamma; a.beta = d->beta; a.mod_gamma = d->mod_gamma; a.ld_mod = d->ld_mod; a.act = d->act;
  a.act_from_pre = d->act_from_pre ? 1 : 0;
  IPK_REQUIRE(!d->act_from_pre || (!d->mod_gamma && !d->dres), "act' from the pre-activation: un-modulated norms whose residual joined behind the activation (its gradient is dy)");
  a.mod_N = d->mod_samples > 0 && d->mod_samples < d->N ? d->mod_samples : 0;
  a.nchunks = (d->S + kNormBwdPos - 1) / kNormBwdPos; a.pos_per_block = kNormBwdPos;
  a.part = d->workspace + ipoke_groupnorm_workspace_floats(d->N, d->S, d->G);
  a.gsum = a.part + (int64_t)d->N * a.nchunks * 2 * d->C;
  IPK_REQUIRE(!a.gamma || a.beta || !a.mod_gamma, "SPADE with an affine norm needs beta");
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<bf16_t>, dim3(a.nchunks, d->N), dim3(256), 0, s, a),
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<float>, dim3(a.nchunks, d->N), dim3(256), 0, s, a));
  IPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_bwd_sums_kernel, dim3(d->N + (d->dgamma ? (d->C + 15) / 16 : 0)), dim3(256), 0, s, a, d->dgamma, d->dbeta);
  IPK_LAUNCH_CHECK();
  const size_t lds = (size_t)(d->act_from_pre ? 8 : 6) * d->C * sizeof(float);
  if (d->dmod_summed && a.mod_N > 0) {
    IPK_REQUIRE(d->mod_gamma && !d->rs_scale && d->N % a.mod_N == 0 && d->C / e16 <= 256, "summed modulation gradients: whole frames of modulated samples");
    const int ppb = (16 / e16) * (256 / (d->C / e16));
    DISPATCH_T(dtype,
      hipLaunchKernelGGL(gn_bwd_apply_frames_kernel<bf16_t>, dim3((d->S + ppb - 1) / ppb, a.mod_N), dim3(256), lds, s, a),
      hipLaunchKernelGGL(gn_bwd_apply_frames_kernel<float>, dim3((d->S + ppb - 1) / ppb, a.mod_N), dim3(256), lds, s, a));
    IPK_LAUNCH_CHECK();
    return IPOKE_OK;
  }
  if (d->rs_scale) {
    IPK_REQUIRE(!d->mod_gamma && d->rs_dots && d->rs_workspace && d->rs_rows_per_group >= d->S && d->rs_rows_per_group % d->S == 0 &&
                ((int64_t)d->N * d->S) % d->rs_rows_per_group == 0 && d->C <= 2048, "folded row-scale pass: whole frames of un-modulated samples");
    a.rs_scale = d->rs_scale; a.rs_scale_stride = d->rs_scale_stride < 1 ? 1 : d->rs_scale_stride;
    a.rs_clips = (int)(d->rs_rows_per_group / d->S); a.rs_pos_per_block = kNormRsPos; a.rs_nchunks = (d->S + kNormRsPos - 1) / kNormRsPos;
    a.rs_bias = d->rs_bias; a.rs_dot_part = d->rs_workspace;
    a.rs_col_part = d->rs_dbias ? d->rs_workspace + (int64_t)d->N * a.rs_nchunks : nullptr;
    const size_t lds_rs = lds + (size_t)(256 * e16 + 4) * sizeof(float);
    DISPATCH_T(dtype,
      hipLaunchKernelGGL((gn_bwd_apply_kernel<bf16_t, true>), dim3(a.rs_nchunks, d->N), dim3(256), lds_rs, s, a),
      hipLaunchKernelGGL((gn_bwd_apply_kernel<float, true>), dim3(a.rs_nchunks, d->N), dim3(256), lds_rs, s, a));
    IPK_LAUNCH_CHECK();
    return rowscale_finalize(a.rs_dot_part, a.rs_col_part, d->N / a.rs_clips, a.rs_clips * a.rs_nchunks, d->C, d->rs_dots, d->rs_dbias, s);
  }
  DISPATCH_T(dtype,
    hipLaunchKernelGGL((gn_bwd_apply_kernel<bf16_t, false>), dim3(a.nchunks, d->N), dim3(256), lds, s, a),
    hipLaunchKernelGGL((gn_bwd_apply_kernel<float, false>), dim3(a.nchunks, d->N), dim3(256), lds, s, a));
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

extern "C" int ipoke_gru_update_bwd(const void* o_pre, const void* u, const void* h, int ldh, const void* d_hnew, int ld_dhnew,
                                    void* d_o_pre, void* d_u, void* d_h, int ld_dh, int64_t M, int Ch, int dtype, void* stream) {
  IPK_REQUIRE(o_pre && u && h && d_hnew && d_o_pre && d_u && d_h, "null tensor");
  const long total = (long)M * Ch;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(gru_update_bwd_kernel<bf16_t>, dim3(grid1d_b(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)o_pre, (const bf16_t*)u, (const bf16_t*)h, ldh, (const bf16_t*)d_hnew, ld_dhnew, (bf16_t*)d_o_pre, (bf16_t*)d_u, (bf16_t*)d_h, ld_dh, (long)M, Ch),
    hipLaunchKernelGGL(gru_update_bwd_kernel<float>, dim3(grid1d_b(total)), dim3(256), 0, STREAM(stream), (const float*)o_pre, (const float*)u, (const float*)h, ldh, (const float*)d_hnew, ld_dhnew, (float*)d_o_pre, (float*)d_u, (float*)d_h, ld_dh, (long)M, Ch));
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
extern "C" int ipoke_gru_gates_bwd(const void* ur_pre, const void* h, int ldh, const void* d_hr, int ld_dhr, const void* d_u,
                                   void* d_ur_pre, void* d_h, int ld_dh, int64_t M, int Ch, int dtype, void* stream) {
  IPK_REQUIRE(ur_pre && h && d_hr && d_ur_pre && d_h, "null tensor");
  const long total = (long)M * Ch;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(gru_gates_bwd_kernel<bf16_t>, dim3(grid1d_b(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)ur_pre, (const bf16_t*)h, ldh, (const bf16_t*)d_hr, ld_dhr, (const bf16_t*)d_u, (bf16_t*)d_ur_pre, (bf16_t*)d_h, ld_dh, (long)M, Ch),
    hipLaunchKernelGGL(gru_gates_bwd_kernel<float>, dim3(grid1d_b(total)), dim3(256), 0, STREAM(stream), (const float*)ur_pre, (const float*)h, ldh, (const float*)d_hr, ld_dhr, (const float*)d_u, (float*)d_ur_pre, (float*)d_h, ld_dh, (long)M, Ch));
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
extern "C" int ipoke_reparam_bwd(const void* mulv, int ld, const float* eps, const float* dz, const float* dmu, const float* dlv,
                                 void* dmulv, int ldo, int64_t M, int Z, int dtype, void* stream) {
  IPK_REQUIRE(mulv && dmulv && ldo >= 2 * Z, "bad arguments");
  const long total = (long)M * ldo;
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(reparam_bwd_kernel<bf16_t>, dim3(grid1d_b(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)mulv, ld, eps, dz, dmu, dlv, (bf16_t*)dmulv, ldo, (long)M, Z),
    hipLaunchKernelGGL(reparam_bwd_kernel<float>, dim3(grid1d_b(total)), dim3(256), 0, STREAM(stream), (const float*)mulv, ld, eps, dz, dmu, dlv, (float*)dmulv, ldo, (long)M, Z));
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
static constexpr int kL1Blocks = 1024;
extern "C" long ipoke_l1_loss_partials(void) { return kL1Blocks; }
extern "C" int ipoke_l1_loss(const float* yhat_cl, int ldy, const float* x_nchw, int N, int C, int S, int64_t x_sn, float scale,
                             float* loss_accum, float* grad_cl, int ldg, float* partials, void* stream) {
  IPK_REQUIRE(yhat_cl && x_nchw && loss_accum && ldy >= C && (!grad_cl || ldg >= C), "bad arguments");
  const long total = (long)N * S * C;
  const int g = grid1d_b(total, 256, kL1Blocks);
  hipLaunchKernelGGL(l1_loss_kernel, dim3(g), dim3(256), 0, STREAM(stream), yhat_cl, ldy, x_nchw, N, C, S, (long)x_sn, scale, loss_accum,
                     grad_cl, ldg, partials);
  IPK_LAUNCH_CHECK();
  if (partials) {
    hipLaunchKernelGGL(l1_loss_final_kernel, dim3(1), dim3(256), 0, STREAM(stream), partials, g, loss_accum);
    IPK_LAUNCH_CHECK();
  }
  return IPOKE_OK;
}
