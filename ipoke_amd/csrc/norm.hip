// GroupNorm / InstanceNorm / SPADE modulation on channels-last activations, plus small element-wise
// helpers of the first-stage VAE (ConvGRU gate math, reparameterisation, bilinear resize).
// Reference call sites: motion_encoder.py:45-74 (GroupNorm(16)+ReLU+residual), autoencoders/util.py:26-36,
// 223-233 (GroupNorm / InstanceNorm in conv blocks), :473-500 (Spade), motion_models/rnn.py:32-56 (ConvGRU).
//
// Statistics are fp32 and numerically robust: every block reduces a chunk of positions to
// (count, mean, M2) per group and a finalize kernel merges chunks with Chan's parallel update.
#include "common.h"
#include "mcf_dev.h"

namespace ipoke {

// x: [N][S][ldx] of T, channels [0, C) normalised in G groups of cpg = C/G consecutive channels.
// part: [N][nchunks][G][3] = (count, mean, M2).
// Thread (rr, cg) owns the E16 channels of column group cg over rows rr, rr + rows_par, ... of the chunk: partial sums stay
// in registers, meet in LDS once, and are reduced in a fixed order (no float atomics: they serialise and are not
// reproducible).  C <= 64*E16 per pass of 256 threads; wider tensors take several passes.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, int S, int ldx, int C, int G, int pos_per_block,
                                                       float* __restrict__ part) {
  constexpr int E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  __shared__ float sm[2][256 * E16];      // per-thread partial sums, then per-channel totals in sm[.][0:C]
  __shared__ float tot[2][4096];          // per-channel sum / sum of squares of the chunk
  const int n = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
  const int p0 = chunk * pos_per_block, p1 = min(S, p0 + pos_per_block);
  const int cvec = C / E16;                          // host guarantees C % E16 == 0 and C <= 4096
  const T* xb = x + ((long)n * S) * ldx;
  for (int g0 = 0; g0 < cvec; g0 += 256) {
    const int groups = cvec - g0 < 256 ? cvec - g0 : 256;
    const int rp = 256 / groups;
    const int cg = threadIdx.x % groups, rr = threadIdx.x / groups;
    float s[E16], q[E16];
#pragma unroll
    for (int e = 0; e < E16; ++e) { s[e] = 0.f; q[e] = 0.f; }
    if (rr < rp) {
      int p = p0 + rr;
      for (; p + 3 * rp < p1; p += 4 * rp) {        // four rows per trip (round 6; same sums in the same order as two trips of two)
        const frag_t a = *reinterpret_cast<const frag_t*>(xb + (long)p * ldx + (g0 + cg) * E16);
        const frag_t b = *reinterpret_cast<const frag_t*>(xb + (long)(p + rp) * ldx + (g0 + cg) * E16);
        const frag_t c = *reinterpret_cast<const frag_t*>(xb + (long)(p + 2 * rp) * ldx + (g0 + cg) * E16);
        const frag_t d = *reinterpret_cast<const frag_t*>(xb + (long)(p + 3 * rp) * ldx + (g0 + cg) * E16);
#pragma unroll
        for (int e = 0; e < E16; ++e) {
          const float fa = ET<T>::to_f32(a[e]), fb = ET<T>::to_f32(b[e]), fc = ET<T>::to_f32(c[e]), fd = ET<T>::to_f32(d[e]);
          s[e] += fa + fb; q[e] += fa * fa + fb * fb;
          s[e] += fc + fd; q[e] += fc * fc + fd * fd;
        }
      }
      for (; p + rp < p1; p += 2 * rp) {            // two rows per trip: both loads in flight
        const frag_t a = *reinterpret_cast<const frag_t*>(xb + (long)p * ldx + (g0 + cg) * E16);
        const frag_t b = *reinterpret_cast<const frag_t*>(xb + (long)(p + rp) * ldx + (g0 + cg) * E16);
#pragma unroll
        for (int e = 0; e < E16; ++e) {
          const float fa = ET<T>::to_f32(a[e]), fb = ET<T>::to_f32(b[e]);
          s[e] += fa + fb; q[e] += fa * fa + fb * fb;
        }
      }
      if (p < p1) {
        const frag_t a = *reinterpret_cast<const frag_t*>(xb + (long)p * ldx + (g0 + cg) * E16);
#pragma unroll
        for (int e = 0; e < E16; ++e) { const float fa = ET<T>::to_f32(a[e]); s[e] += fa; q[e] += fa * fa; }
      }
    }
#pragma unroll
    for (int e = 0; e < E16; ++e) {
      sm[0][threadIdx.x * E16 + e] = rr < rp ? s[e] : 0.f;
      sm[1][threadIdx.x * E16 + e] = rr < rp ? q[e] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < groups * E16; i += 256) {
      const int cgi = i / E16, e = i - cgi * E16;
      float ts = 0.f, tq = 0.f;
      for (int k = 0; k < rp; ++k) { ts += sm[0][(k * groups + cgi) * E16 + e]; tq += sm[1][(k * groups + cgi) * E16 + e]; }
      tot[0][(g0 + cgi) * E16 + e] = ts; tot[1][(g0 + cgi) * E16 + e] = tq;
    }
    __syncthreads();
  }
  const int cpg = C / G;
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float s = 0.f, q = 0.f;
    for (int c = 0; c < cpg; ++c) { s += tot[0][g * cpg + c]; q += tot[1][g * cpg + c]; }
    const float cnt = (float)(p1 - p0) * cpg;
    const float mean = cnt > 0 ? s / cnt : 0.f;
    float m2 = q - s * mean;
    if (m2 < 0.f) m2 = 0.f;
    float* o = part + (((long)n * nchunks + chunk) * G + g) * 3;
    o[0] = cnt; o[1] = mean; o[2] = m2;
  }
}
// stats[n][g] = (mean, rstd): Chan's parallel merge of the chunk statistics; 16 threads per group, then a fixed-order merge
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, int nchunks, int G, float eps,
                                                          float* __restrict__ stats) {
  __shared__ float sc[256], sme[256], sm2[256];
  const int n = blockIdx.x;
  constexpr int PAR = 16;
  const int gl = threadIdx.x / PAR, pr = threadIdx.x % PAR;          // 16 groups per pass
  // grid (N, ceil(G / 16)): a block owns 16 groups (InstanceNorm has G = C up to 256 groups: sixteen serial passes of one block
  // took 21 us)
  {
    const int g0 = blockIdx.y * (256 / PAR);
    const int g = g0 + gl;
    float cnt = 0.f, mean = 0.f, m2 = 0.f;
    if (g < G) {
      // the partials of up to 8 chunks are requested before the (serially dependent) merge touches any of them: one memory
      // latency per batch instead of one per chunk (this kernel is pure latency: 19.5 -> ~5 us at 128 chunks)
      constexpr int PF = 8;
      for (int c0 = pr; c0 < nchunks; c0 += PAR * PF) {
        float pc[PF], pm[PF], pq[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
          const int c = c0 + k * PAR;
          pc[k] = 0.f; pm[k] = 0.f; pq[k] = 0.f;
          if (c < nchunks) {
            const float* o = part + (((long)n * nchunks + c) * G + g) * 3;
            pc[k] = o[0]; pm[k] = o[1]; pq[k] = o[2];
          }
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) {
          const float cb = pc[k];
          if (cb <= 0.f) continue;
          const float delta = pm[k] - mean, tot = cnt + cb;
          mean += delta * cb / tot;
          m2 += pq[k] + delta * delta * cnt * cb / tot;
          cnt = tot;
        }
      }
    }
    sc[threadIdx.x] = cnt; sme[threadIdx.x] = mean; sm2[threadIdx.x] = m2;
    __syncthreads();
    if (pr == 0 && g < G) {
      float C0 = 0.f, M0 = 0.f, Q0 = 0.f;
      for (int k = 0; k < PAR; ++k) {
        const float cb = sc[gl * PAR + k];
        if (cb <= 0.f) continue;
        const float delta = sme[gl * PAR + k] - M0, tot = C0 + cb;
        M0 += delta * cb / tot;
        Q0 += sm2[gl * PAR + k] + delta * delta * C0 * cb / tot;
        C0 = tot;
      }
      stats[((long)n * G + g) * 2] = M0;
      stats[((long)n * G + g) * 2 + 1] = rsqrtf(Q0 / C0 + eps);     // biased variance, as torch group_norm
    }
    __syncthreads();
  }
}
struct NormApply {
  const void* x; int ldx; void* y; int ldy; int y_f32;
  int N, S, C, G;
  const float* stats;              // [N][G][2]
  const float* gamma; const float* beta;      // [C] affine or NULL
  const void* mod_gamma; const void* mod_beta; int ld_mod;   // SPADE: T [N*S][ld_mod]; y = xhat*(1+mg)+mb
  int mod_N;                       // > 0: the modulation maps hold mod_N samples, sample n reads those of n % mod_N
  const void* res; int ld_res;     // optional residual added before the activation (res_post: behind it)
  int act;
  int res_post = 0;
  float* next_part = nullptr; int next_G = 0;      // gn_apply_kernel<T, true>: chunk statistics (count, mean, M2) of the OUTPUT for a following norm
  int pos_per_block;
  float* stats_out = nullptr;      // gn_fused_kernel: [N][G][2] (mean, rstd), kept for the backward pass
};
// grid (chunks, N): a block normalises pos_per_block positions of ONE sample, so the per-channel scale / shift
// (rstd*gamma, beta - mean*rstd*gamma) are built once in LDS and the inner loop is one FMA per element, no divisions.
// NEXT (round 6): the pass also leaves the chunk statistics of its (rounded) output in gn_stats_kernel's layout -- the norm that reads
// this output next (the SPADE norm behind a ResBlock whose sum this pass carries, res_post) starts at gn_finalize: one read of the
// tensor less.  Chunk = this pass's block of positions; one pass of 256 threads over the channels (C <= 256 * E16).
template <typename T, bool NEXT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const NormApply a) {
  if (IPK_KERNARG_PREFETCH) kernarg_prefetch<(int)sizeof(NormApply)>();
  constexpr int E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  __shared__ float sscale[NEXT ? 2048 : 4096], sshift[NEXT ? 2048 : 4096];
  __shared__ float nsm[NEXT ? 2 : 1][NEXT ? 256 * E16 : 1];
  __shared__ float ntot[NEXT ? 2 : 1][NEXT ? 2048 : 1];
  const int n = blockIdx.y;
  const int cpg = a.C / a.G;
  for (int c = threadIdx.x; c < a.C; c += 256) {
    const int g = c / cpg;
    const float mean = a.stats[((long)n * a.G + g) * 2], rstd = a.stats[((long)n * a.G + g) * 2 + 1];
    const float gm = a.gamma ? a.gamma[c] : 1.f, bt = a.beta ? a.beta[c] : 0.f;
    sscale[c] = rstd * gm; sshift[c] = bt - mean * rstd * gm;
  }
  __syncthreads();
  const int cvec = a.C / E16;
  const int p0 = blockIdx.x * a.pos_per_block, p1 = min(a.S, p0 + a.pos_per_block);
  for (int g0 = 0; g0 < cvec; g0 += 256) {
    const int groups = cvec - g0 < 256 ? cvec - g0 : 256;
    const int rp = 256 / groups;
    const int cc = g0 + threadIdx.x % groups, rr = threadIdx.x / groups;
    if (!NEXT && rr >= rp) continue;
    float sc[E16], sh[E16], ns[E16], nq[E16];
#pragma unroll
    for (int e = 0; e < E16; ++e) { sc[e] = sscale[cc * E16 + e]; sh[e] = sshift[cc * E16 + e]; ns[e] = 0.f; nq[e] = 0.f; }
    if (rr < rp)
    for (int p = p0 + rr; p < p1; p += rp) {
      const long m = (long)n * a.S + p;
      const frag_t v = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.x) + m * a.ldx + cc * E16);
      frag_t r, mg, mb;
      if (a.res) r = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.res) + m * a.ld_res + cc * E16);
      if (a.mod_gamma) {
        const long mm = a.mod_N > 0 ? (long)(n % a.mod_N) * a.S + p : m;
        mg = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.mod_gamma) + mm * a.ld_mod + cc * E16);
        mb = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.mod_beta) + mm * a.ld_mod + cc * E16);
      }
      float o[E16];
#pragma unroll
      for (int e = 0; e < E16; ++e) {
        float f = ET<T>::to_f32(v[e]) * sc[e] + sh[e];
        if (a.mod_gamma) f = f * (1.f + ET<T>::to_f32(mg[e])) + ET<T>::to_f32(mb[e]);
        if (a.res && !a.res_post) f += ET<T>::to_f32(r[e]);
        o[e] = act_apply(a.act, f);
        if (a.res && a.res_post) o[e] += ET<T>::to_f32(r[e]);
      }
      if (a.y_f32) {
        float* yp = reinterpret_cast<float*>(a.y) + m * a.ldy + cc * E16;
#pragma unroll
        for (int e = 0; e < E16; ++e) yp[e] = o[e];
      } else {
        frag_t w;
#pragma unroll
        for (int e = 0; e < E16; ++e) {
          w[e] = ET<T>::from_f32(o[e]);
          if (NEXT) { const float fv = ET<T>::to_f32(w[e]); ns[e] += fv; nq[e] += fv * fv; }
        }
        *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(a.y) + m * a.ldy + cc * E16) = w;
      }
    }
    if (NEXT) {
#pragma unroll
      for (int e = 0; e < E16; ++e) {
        nsm[0][threadIdx.x * E16 + e] = rr < rp ? ns[e] : 0.f;
        nsm[1][threadIdx.x * E16 + e] = rr < rp ? nq[e] : 0.f;
      }
      __syncthreads();
      for (int i = threadIdx.x; i < groups * E16; i += 256) {
        const int cgi = i / E16, e = i - cgi * E16;
        float ts = 0.f, tq = 0.f;
        for (int k = 0; k < rp; ++k) { ts += nsm[0][(k * groups + cgi) * E16 + e]; tq += nsm[1][(k * groups + cgi) * E16 + e]; }
        ntot[0][cgi * E16 + e] = ts; ntot[1][cgi * E16 + e] = tq;
      }
      __syncthreads();
    }
  }
  if (NEXT) {
    const int cpgn = a.C / a.next_G;
    for (int g = threadIdx.x; g < a.next_G; g += 256) {
      float s_ = 0.f, q_ = 0.f;
      for (int c = 0; c < cpgn; ++c) { s_ += ntot[0][g * cpgn + c]; q_ += ntot[1][g * cpgn + c]; }
      const float cnt = (float)(p1 - p0) * cpgn;
      const float mean = cnt > 0 ? s_ / cnt : 0.f;
      float m2 = q_ - s_ * mean;
      if (m2 < 0.f) m2 = 0.f;
      float* o = a.next_part + (((long)n * gridDim.x + blockIdx.x) * a.next_G + g) * 3;
      o[0] = cnt; o[1] = mean; o[2] = m2;
    }
  }
}

// Small maps (the 16 x 16 ... 8 x 8 stages of the encoders): statistics, normalisation, affine, residual and activation of one
// (sample, channel slab) in ONE launch instead of stats -> finalize -> apply.  Those three launches had 20 ... 160 workgroups
// to work with (6 + 5 + 30 us for 256 positions x 256 channels x 20 samples); a slab of CS channels (whole groups) gives
// N * C / CS workgroups, is read from memory once into LDS, and the exact two-pass variance (mean first, then the squared
// deviations) is taken from there.  grid (C / CS, N), dynamic LDS = S * CS * sizeof(T).
template <typename T>
__global__ __launch_bounds__(256) void gn_fused_kernel(const NormApply a, int CS, float eps) {
  if (IPK_KERNARG_PREFETCH) kernarg_prefetch<(int)sizeof(NormApply)>();
  constexpr int E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  __shared__ float red[256 * E16];
  __shared__ float chan[2][2048];            // per channel of the slab: totals, then (scale, shift)
  T* tile = reinterpret_cast<T*>(gsm);
  const int n = blockIdx.y, c0 = blockIdx.x * CS;
  const int cvec = CS / E16, rp = 256 / cvec;
  const int cc = threadIdx.x % cvec, rr = threadIdx.x / cvec;
  const bool on = rr < rp;
  const int cpg = a.C / a.G, ngl = CS / cpg;
  const T* xb = reinterpret_cast<const T*>(a.x) + ((long)n * a.S) * a.ldx + c0 + cc * E16;
  // ---- pass 1: memory -> LDS, per-channel sums
  float s[E16];
#pragma unroll
  for (int e = 0; e < E16; ++e) s[e] = 0.f;
  if (on) {
    // four rows per trip, all loads issued before the first use: a slab is streamed by ONE workgroup, so its memory-level
    // parallelism is what the loop keeps in flight (S = 4096 x 16 channels: 58 us with one load per trip)
    for (int p = rr; p < a.S; p += 4 * rp) {
      frag_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p + u * rp < a.S) v[u] = *reinterpret_cast<const frag_t*>(xb + (long)(p + u * rp) * a.ldx);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (p + u * rp < a.S) {
          *reinterpret_cast<frag_t*>(tile + (long)(p + u * rp) * CS + cc * E16) = v[u];
#pragma unroll
          for (int e = 0; e < E16; ++e) s[e] += ET<T>::to_f32(v[u][e]);
        }
      }
    }
  }
  auto channel_totals = [&](const float (&v)[E16]) {      // chan[0][c] = sum over the row lanes, fixed order
#pragma unroll
    for (int e = 0; e < E16; ++e) red[threadIdx.x * E16 + e] = on ? v[e] : 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < CS; i += 256) {
      const int cgi = i / E16, e = i - cgi * E16;
      float t = 0.f;
      for (int k = 0; k < rp; ++k) t += red[(k * cvec + cgi) * E16 + e];
      chan[0][i] = t;
    }
    __syncthreads();
  };
  channel_totals(s);
  const float inv_cnt = 1.f / ((float)a.S * (float)cpg);
  for (int g = threadIdx.x; g < ngl; g += 256) {
    float t = 0.f;
    for (int c = 0; c < cpg; ++c) t += chan[0][g * cpg + c];
    const float mean = t * inv_cnt;
    for (int c = 0; c < cpg; ++c) chan[1][g * cpg + c] = mean;
    if (a.stats_out) a.stats_out[((long)n * a.G + c0 / cpg + g) * 2] = mean;
  }
  __syncthreads();
  // ---- pass 2 (LDS): squared deviations from the group mean
  float mu[E16], q[E16];
#pragma unroll
  for (int e = 0; e < E16; ++e) { mu[e] = chan[1][cc * E16 + e]; q[e] = 0.f; }
  if (on) {
    for (int p = rr; p < a.S; p += rp) {
      const frag_t v = *reinterpret_cast<const frag_t*>(tile + (long)p * CS + cc * E16);
#pragma unroll
      for (int e = 0; e < E16; ++e) { const float d = ET<T>::to_f32(v[e]) - mu[e]; q[e] += d * d; }
    }
  }
  channel_totals(q);
  for (int g = threadIdx.x; g < ngl; g += 256) {
    float t = 0.f;
    for (int c = 0; c < cpg; ++c) t += chan[0][g * cpg + c];
    const float rstd = rsqrtf(t * inv_cnt + eps);                     // biased variance, as torch group_norm
    for (int c = 0; c < cpg; ++c) chan[0][g * cpg + c] = rstd;
    if (a.stats_out) a.stats_out[((long)n * a.G + c0 / cpg + g) * 2 + 1] = rstd;
  }
  __syncthreads();
  // scale -> chan[0], shift -> chan[1] (a thread rewrites only the channels it has just read)
  for (int i = threadIdx.x; i < CS; i += 256) {
    const float mean = chan[1][i], rstd = chan[0][i];
    const float gm = a.gamma ? a.gamma[c0 + i] : 1.f, bt = a.beta ? a.beta[c0 + i] : 0.f;
    chan[0][i] = rstd * gm; chan[1][i] = bt - mean * rstd * gm;
  }
  __syncthreads();
  // ---- pass 3 (LDS -> memory)
  if (!on) return;
  float sc[E16], sh[E16];
#pragma unroll
  for (int e = 0; e < E16; ++e) { sc[e] = chan[0][cc * E16 + e]; sh[e] = chan[1][cc * E16 + e]; }
  for (int p = rr; p < a.S; p += 4 * rp) {
    frag_t r[4];
    if (a.res) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p + u * rp < a.S) r[u] = *reinterpret_cast<const frag_t*>(reinterpret_cast<const T*>(a.res) + ((long)n * a.S + p + u * rp) * a.ld_res + c0 + cc * E16);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (p + u * rp >= a.S) continue;
      const long m = (long)n * a.S + p + u * rp;
      const frag_t v = *reinterpret_cast<const frag_t*>(tile + (long)(p + u * rp) * CS + cc * E16);
      frag_t w;
#pragma unroll
      for (int e = 0; e < E16; ++e) {
        float f = ET<T>::to_f32(v[e]) * sc[e] + sh[e];
        if (a.res && !a.res_post) f += ET<T>::to_f32(r[u][e]);
        f = act_apply(a.act, f);
        if (a.res && a.res_post) f += ET<T>::to_f32(r[u][e]);
        w[e] = ET<T>::from_f32(f);
      }
      *reinterpret_cast<frag_t*>(reinterpret_cast<T*>(a.y) + m * a.ldy + c0 + cc * E16) = w;
    }
  }
}

// generic element-wise: y = act(a (+ b)) on dense T rows of C channels with pitches
// 16-byte chunks along the channels (C and all pitches multiples of the chunk): no per-element 64-bit division
template <typename T>
__global__ void ew_add_act_vec_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb, T* __restrict__ y, int ldy,
                                      long M, int C, int act) {
  constexpr int E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  const int cv = C / E16;
  const long total = M * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / cv; const int c = (int)(i - m * cv) * E16;
    const frag_t va = *reinterpret_cast<const frag_t*>(a + m * lda + c);
    frag_t vb, o;
    if (b) vb = *reinterpret_cast<const frag_t*>(b + m * ldb + c);
#pragma unroll
    for (int e = 0; e < E16; ++e) {
      float f = ET<T>::to_f32(va[e]);
      if (b) f += ET<T>::to_f32(vb[e]);
      o[e] = ET<T>::from_f32(act_apply(act, f));
    }
    *reinterpret_cast<frag_t*>(y + m * ldy + c) = o;
  }
}
template <typename T>
__global__ void ew_add_act_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb, T* __restrict__ y, int ldy,
                                  long M, int C, int act) {
  const long total = M * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / C; const int c = (int)(i - m * C);
    float f = ET<T>::to_f32(a[m * lda + c]);
    if (b) f += ET<T>::to_f32(b[m * ldb + c]);
    y[m * ldy + c] = ET<T>::from_f32(act_apply(act, f));
  }
}

// ConvGRU (rnn.py:48-56).  Phase 1: xh_r[:, Cx:Cx+Ch] = h * sigmoid(r_pre);  u = sigmoid(u_pre) (in place)
template <typename T>
__global__ void gru_gates_kernel(const T* __restrict__ ur_pre /* [M][2Ch]: u | r */, const T* __restrict__ h, int ldh,
                                 T* __restrict__ hr_out, int ld_hr, T* __restrict__ u_out, long M, int Ch) {
  const long total = M * Ch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / Ch; const int c = (int)(i - m * Ch);
    const float u = act_apply(IPOKE_ACT_SIGMOID, ET<T>::to_f32(ur_pre[m * 2 * Ch + c]));
    const float r = act_apply(IPOKE_ACT_SIGMOID, ET<T>::to_f32(ur_pre[m * 2 * Ch + Ch + c]));
    hr_out[m * ld_hr + c] = ET<T>::from_f32(ET<T>::to_f32(h[m * ldh + c]) * r);
    u_out[m * Ch + c] = ET<T>::from_f32(u);
  }
}
// Phase 2: h' = h*(1-u) + tanh(o_pre)*u
template <typename T>
__global__ void gru_update_kernel(const T* __restrict__ o_pre, const T* __restrict__ u, const T* __restrict__ h, int ldh,
                                  T* __restrict__ h_new, int ld_new, long M, int Ch) {
  const long total = M * Ch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / Ch; const int c = (int)(i - m * Ch);
    const float uu = ET<T>::to_f32(u[m * Ch + c]);
    const float o = tanhf(ET<T>::to_f32(o_pre[m * Ch + c]));
    h_new[m * ld_new + c] = ET<T>::from_f32(ET<T>::to_f32(h[m * ldh + c]) * (1.f - uu) + o * uu);
  }
}
// z = mu + eps * exp(0.5*logvar)   (motion_encoder.py:218-222); mulv: T [M][2z] = mu | logvar; out fp32 NCHW-free [M][z]
template <typename T>
__global__ void reparam_kernel(const T* __restrict__ mulv, int ld, const float* __restrict__ eps, float* __restrict__ z,
                               float* __restrict__ mu_out, float* __restrict__ lv_out, long M, int Z) {
  const long total = M * Z;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / Z; const int c = (int)(i - m * Z);
    const float mu = ET<T>::to_f32(mulv[m * ld + c]), lv = ET<T>::to_f32(mulv[m * ld + Z + c]);
    z[i] = eps ? mu + eps[i] * expf(0.5f * lv) : mu;
    if (mu_out) mu_out[i] = mu;
    if (lv_out) lv_out[i] = lv;
  }
}
// bilinear resize, align_corners=True (util.py:495), NCHW fp32 [N][C][Hi][Wi] -> channels-last fp32 [N][Ho][Wo][C]
__global__ void bilinear_cl_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int Hi, int Wi, int Ho, int Wo) {
  const long total = (long)N * Ho * Wo * C;
  const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long t = i / C;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho); const int n = (int)(t / Ho);
    const float fy = oy * sy, fx = ox * sx;
    const int y0 = min((int)fy, Hi - 1), x0 = min((int)fx, Wi - 1);
    const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const float* p = x + ((long)n * C + c) * Hi * Wi;
    const float v = (1.f - wy) * ((1.f - wx) * p[y0 * Wi + x0] + wx * p[y0 * Wi + x1]) +
                    wy * ((1.f - wx) * p[y1 * Wi + x0] + wx * p[y1 * Wi + x1]);
    y[i] = v;
  }
}
// channels-last T [M][ld] (first C channels) -> fp32 NCHW-like [N][C][S]
template <typename T>
__global__ void cl_to_nchw_kernel(const T* __restrict__ x, int ld, float* __restrict__ y, int N, int C, int S) {
  const long total = (long)N * C * S;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int s = (int)(i % S); long t = i / S;
    const int c = (int)(t % C); const int n = (int)(t / C);
    y[i] = ET<T>::to_f32(x[((long)n * S + s) * ld + c]);
  }
}
// fp32 NCHW-like [N][C][S] -> channels-last T [N*S][ld] (zero padded to ld)
template <typename T>
__global__ void nchw_to_cl_kernel(const float* __restrict__ x, T* __restrict__ y, int ld, int N, int C, int S) {
  const long total = (long)N * S * ld;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % ld); long t = i / ld;
    const int s = (int)(t % S); const int n = (int)(t / S);
    y[i] = ET<T>::from_f32(c < C ? x[((long)n * C + c) * S + s] : 0.f);
  }
}

// fp32 clip src[n][c][t][y][x] (element strides s_*; 3 channels) -> channels-last pixels of FOUR channels of T (the fourth is 0),
// rows dst[((n*Tn + t)*H + y)*Wp + pad_l + x][4], zero columns left and right: with 8- / 16-byte pixels a 7-tap window along x is
// one contiguous, 16-byte aligned run of 8 pixels, so the encoder's (3, 7, 7) stem runs as 21 taps of 32 channels on the LDS-DMA
// GEMM (motion_encoder.py:161; the reference reads the same pixels through nn.Conv3d(3, 64, (3, 7, 7), 2, (1, 3, 3))).
template <typename T>
__global__ void clip_to_cl4_kernel(const float* __restrict__ src, long s_n, long s_c, long s_t, long s_h, long s_w, int Tn, int H, int W,
                                   int pad_l, int Wp, long rows, T* __restrict__ dst) {
  typedef typename Pack4<T>::type pack_t;
  const long total = rows * Wp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xp = (int)(i % Wp); long r = i / Wp;
    const int y = (int)(r % H); r /= H;
    const int t = (int)(r % Tn); const long n = r / Tn;
    const int x = xp - pad_l;
    pack_t o;
    if (x >= 0 && x < W) {
      const float* q = src + n * s_n + t * s_t + y * s_h + x * s_w;
      o[0] = ET<T>::from_f32(q[0]); o[1] = ET<T>::from_f32(q[s_c]); o[2] = ET<T>::from_f32(q[2 * s_c]);
    } else {
      o[0] = ET<T>::from_f32(0.f); o[1] = o[0]; o[2] = o[0];
    }
    o[3] = ET<T>::from_f32(0.f);
    *reinterpret_cast<pack_t*>(dst + i * 4) = o;
  }
}

static inline int grid1d(long n, int block = 256, int cap = 4096) {
  long g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace ipoke

using namespace ipoke;
#define STREAM(s) reinterpret_cast<hipStream_t>(s)
#define DISPATCH_T(dtype, CALL_BF, CALL_F32) do { if ((dtype) == IPOKE_BF16) { CALL_BF; } else { CALL_F32; } } while (0)

extern "C" int64_t ipoke_groupnorm_workspace_floats(int N, int S, int G) {
  const int ppb = 128;
  const int nchunks = (S + ppb - 1) / ppb;
  return (int64_t)N * nchunks * G * 3 + (int64_t)N * G * 2;
}

extern "C" int64_t ipoke_groupnorm_stats_offset(int N, int S, int G) {
  const int ppb = 128;
  return (int64_t)N * ((S + ppb - 1) / ppb) * G * 3;
}

extern "C" int ipoke_groupnorm_stats(const void* x, int ldx, int N, int S, int C, int G, float eps, float* workspace, int dtype,
                                     void* stream) {
  IPK_REQUIRE(x && workspace && C % G == 0, "bad arguments");
  const int e16 = dtype == IPOKE_BF16 ? 8 : 4;
  IPK_REQUIRE(C % e16 == 0 && ldx % e16 == 0 && C <= 4096, "channels must be a multiple of 16 bytes");
  const int ppb = 128;
  const int nchunks = (S + ppb - 1) / ppb;
  float* part = workspace;
  float* stats = part + (int64_t)N * nchunks * G * 3;
  hipStream_t s = STREAM(stream);
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, dim3(nchunks, N), dim3(256), 0, s, (const bf16_t*)x, S, ldx, C, G, ppb, part),
    hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(nchunks, N), dim3(256), 0, s, (const float*)x, S, ldx, C, G, ppb, part));
  IPK_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(N, (G + 15) / 16), dim3(256), 0, s, part, nchunks, G, eps, stats);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

extern "C" int ipoke_groupnorm(const ipoke_norm_desc* d, int dtype, void* stream) {
  IPK_REQUIRE(d && d->x && d->y && d->workspace, "null tensor");
  const int e16 = dtype == IPOKE_BF16 ? 8 : 4;
  IPK_REQUIRE(d->C % e16 == 0 && d->C % d->G == 0 && d->C <= 4096, "channels must be a multiple of 16 bytes and of the group count");
  IPK_REQUIRE(d->ldx % e16 == 0 && d->ldy % (d->y_f32 ? 1 : e16) == 0, "pitches must keep 16-byte alignment");
  IPK_REQUIRE((d->gamma == nullptr) == (d->beta == nullptr) && (d->mod_gamma == nullptr) == (d->mod_beta == nullptr), "affine pairs");
  IPK_REQUIRE(!d->next_part || (d->next_G >= 1 && d->C % d->next_G == 0 && !d->y_f32 && d->C <= 2048 && d->C / e16 <= 256),
              "statistics for a following norm: a dtype output of at most 2048 channels, whole groups");
  IPK_REQUIRE(d->part_chunks >= 0 && (d->part_chunks == 0 || d->part_chunks <= (d->S + 127) / 128), "chunk statistics must fit the workspace");
  const int ppb = 128;
  const int nchunks = (d->S + ppb - 1) / ppb;
  float* part = d->workspace;
  float* stats = part + (int64_t)d->N * nchunks * d->G * 3;
  hipStream_t s = STREAM(stream);
  {
    // small maps: one launch per norm (gn_fused_kernel).  Slab = the smallest run of whole groups that is a multiple of 16 bytes
    // and at least 32 channels wide (64-byte rows), as long as the S x CS tile fits in LDS.  IPOKE_GN_FUSED=0: developer A/B.
    static const bool fused_on = !(getenv("IPOKE_GN_FUSED") && atoi(getenv("IPOKE_GN_FUSED")) == 0);
    const int cpg = d->C / d->G, esz = dtype == IPOKE_BF16 ? 2 : 4;
    int unit = cpg; while (unit % e16) unit += cpg;                    // lcm(cpg, e16)
    // Round 6: a LARGE tensor (>= 32 MB: the decoder's InstanceNorms over all frames of a batch) only takes this path with whole 128-byte
    // row segments and a tile small enough for four workgroups per CU.  The rule above gave the 64 x 64 x 128 InstanceNorm slabs of 16
    // channels -- 32 of every 128 bytes fetched used, one 128 KB workgroup per CU: 662 us for 503 MB at N = 480, where the statistics and
    // apply passes take 90 + 190 us (scripts/r6/probe_norm.py); 280 vs 45 + 100 us at 32 x 32 x 256.
    const bool large = (int64_t)d->N * d->S * d->C * esz >= ((int64_t)32 << 20);
    const int want_ch = large ? 128 / esz : 32;
    int CS = unit; while (CS < want_ch && d->C % (CS + unit) == 0 && CS + unit <= d->C) CS += unit;
    while (d->C % CS) CS += unit;
    const size_t tile = (size_t)d->S * CS * esz;
    const bool shape_ok = large ? (CS * esz >= 128 && tile <= 32 * 1024) : tile <= 128 * 1024;
    if (fused_on && !d->y_f32 && !d->mod_gamma && CS <= 2048 && CS / e16 <= 256 && shape_ok && d->ldy % e16 == 0 &&
        (!d->res || d->ld_res % e16 == 0)) {
      NormApply a;
      a.x = d->x; a.ldx = d->ldx; a.y = d->y; a.ldy = d->ldy; a.y_f32 = 0; a.N = d->N; a.S = d->S; a.C = d->C; a.G = d->G;
      a.stats = nullptr; a.gamma = d->gamma; a.beta = d->beta; a.mod_gamma = nullptr; a.mod_beta = nullptr; a.ld_mod = 0; a.mod_N = 0;
      a.res = d->res; a.ld_res = d->ld_res; a.act = d->act; a.pos_per_block = 0; a.stats_out = stats; a.res_post = d->res_post;
      if (dtype == IPOKE_BF16) {
        static bool attr_bf = false;
        if (!attr_bf) { IPK_HIP(hipFuncSetAttribute((const void*)gn_fused_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)); attr_bf = true; }
        hipLaunchKernelGGL(gn_fused_kernel<bf16_t>, dim3(d->C / CS, d->N), dim3(256), tile, s, a, CS, d->eps);
      } else {
        static bool attr_f = false;
        if (!attr_f) { IPK_HIP(hipFuncSetAttribute((const void*)gn_fused_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)); attr_f = true; }
        hipLaunchKernelGGL(gn_fused_kernel<float>, dim3(d->C / CS, d->N), dim3(256), tile, s, a, CS, d->eps);
      }
      IPK_LAUNCH_CHECK();
      if (d->next_part) {                 // the one-launch kernel owns channel slabs, not position chunks: the output's statistics as a pass
        const int nppb = 256, nnch = (d->S + nppb - 1) / nppb;
        DISPATCH_T(dtype,
          hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, dim3(nnch, d->N), dim3(256), 0, s, (const bf16_t*)d->y, d->S, d->ldy, d->C, d->next_G, nppb, d->next_part),
          hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(nnch, d->N), dim3(256), 0, s, (const float*)d->y, d->S, d->ldy, d->C, d->next_G, nppb, d->next_part));
        IPK_LAUNCH_CHECK();
      }
      return IPOKE_OK;
    }
  }
  if (d->part_chunks > 0) {               // the producer of x left its chunk statistics in the workspace (next_part of that call)
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(d->N, (d->G + 15) / 16), dim3(256), 0, s, part, d->part_chunks, d->G, d->eps, stats);
  } else {
    DISPATCH_T(dtype,
      hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, dim3(nchunks, d->N), dim3(256), 0, s, (const bf16_t*)d->x, d->S, d->ldx, d->C, d->G, ppb, part),
      hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(nchunks, d->N), dim3(256), 0, s, (const float*)d->x, d->S, d->ldx, d->C, d->G, ppb, part));
    IPK_LAUNCH_CHECK();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(d->N, (d->G + 15) / 16), dim3(256), 0, s, part, nchunks, d->G, d->eps, stats);
  }
  IPK_LAUNCH_CHECK();
  NormApply a;
  a.x = d->x; a.ldx = d->ldx; a.y = d->y; a.ldy = d->ldy; a.y_f32 = d->y_f32; a.N = d->N; a.S = d->S; a.C = d->C; a.G = d->G;
  a.stats = stats; a.gamma = d->gamma; a.beta = d->beta; a.mod_gamma = d->mod_gamma; a.mod_beta = d->mod_beta; a.ld_mod = d->ld_mod;
  a.res = d->res; a.ld_res = d->ld_res; a.act = d->act; a.res_post = d->res_post;
  a.mod_N = d->mod_samples > 0 && d->mod_samples < d->N ? d->mod_samples : 0;
  a.pos_per_block = 256;
  const int achunks = (d->S + a.pos_per_block - 1) / a.pos_per_block;
  if (d->next_part) {
    a.next_part = d->next_part; a.next_G = d->next_G;
    DISPATCH_T(dtype,
      hipLaunchKernelGGL((gn_apply_kernel<bf16_t, true>), dim3(achunks, d->N), dim3(256), 0, s, a),
      hipLaunchKernelGGL((gn_apply_kernel<float, true>), dim3(achunks, d->N), dim3(256), 0, s, a));
    IPK_LAUNCH_CHECK();
    return IPOKE_OK;
  }
  DISPATCH_T(dtype,
    hipLaunchKernelGGL((gn_apply_kernel<bf16_t, false>), dim3(achunks, d->N), dim3(256), 0, s, a),
    hipLaunchKernelGGL((gn_apply_kernel<float, false>), dim3(achunks, d->N), dim3(256), 0, s, a));
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

extern "C" int ipoke_add_act(const void* a, int lda, const void* b, int ldb, void* y, int ldy, int64_t M, int C, int act,
                             int dtype, void* stream) {
  IPK_REQUIRE(a && y, "null tensor");
  const int e16 = dtype == IPOKE_BF16 ? 8 : 4;
  if (C % e16 == 0 && lda % e16 == 0 && ldy % e16 == 0 && (!b || ldb % e16 == 0) &&
      (((uintptr_t)a | (uintptr_t)y | (uintptr_t)(b ? b : a)) & 15) == 0) {
    const long nv = (long)M * (C / e16);
    DISPATCH_T(dtype,
      hipLaunchKernelGGL(ew_add_act_vec_kernel<bf16_t>, dim3(grid1d(nv)), dim3(256), 0, STREAM(stream), (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)y, ldy, (long)M, C, act),
      hipLaunchKernelGGL(ew_add_act_vec_kernel<float>, dim3(grid1d(nv)), dim3(256), 0, STREAM(stream), (const float*)a, lda, (const float*)b, ldb, (float*)y, ldy, (long)M, C, act));
    IPK_LAUNCH_CHECK();
    return IPOKE_OK;
  }
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(ew_add_act_kernel<bf16_t>, dim3(grid1d(M * C)), dim3(256), 0, STREAM(stream), (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)y, ldy, (long)M, C, act),
    hipLaunchKernelGGL(ew_add_act_kernel<float>, dim3(grid1d(M * C)), dim3(256), 0, STREAM(stream), (const float*)a, lda, (const float*)b, ldb, (float*)y, ldy, (long)M, C, act));
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
extern "C" int ipoke_gru_gates(const void* ur_pre, const void* h, int ldh, void* hr_out, int ld_hr, void* u_out, int64_t M, int Ch,
                               int dtype, void* stream) {
  IPK_REQUIRE(ur_pre && h && hr_out && u_out, "null tensor");
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(gru_gates_kernel<bf16_t>, dim3(grid1d(M * Ch)), dim3(256), 0, STREAM(stream), (const bf16_t*)ur_pre, (const bf16_t*)h, ldh, (bf16_t*)hr_out, ld_hr, (bf16_t*)u_out, (long)M, Ch),
    hipLaunchKernelGGL(gru_gates_kernel<float>, dim3(grid1d(M * Ch)), dim3(256), 0, STREAM(stream), (const float*)ur_pre, (const float*)h, ldh, (float*)hr_out, ld_hr, (float*)u_out, (long)M, Ch));
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
extern "C" int ipoke_gru_update(const void* o_pre, const void* u, const void* h, int ldh, void* h_new, int ld_new, int64_t M, int Ch,
                                int dtype, void* stream) {
  IPK_REQUIRE(o_pre && u && h && h_new, "null tensor");
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(gru_update_kernel<bf16_t>, dim3(grid1d(M * Ch)), dim3(256), 0, STREAM(stream), (const bf16_t*)o_pre, (const bf16_t*)u, (const bf16_t*)h, ldh, (bf16_t*)h_new, ld_new, (long)M, Ch),
    hipLaunchKernelGGL(gru_update_kernel<float>, dim3(grid1d(M * Ch)), dim3(256), 0, STREAM(stream), (const float*)o_pre, (const float*)u, (const float*)h, ldh, (float*)h_new, ld_new, (long)M, Ch));
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
extern "C" int ipoke_reparameterize(const void* mulv, int ld, const float* eps, float* z, float* mu, float* logvar, int64_t M, int Z,
                                    int dtype, void* stream) {
  IPK_REQUIRE(mulv && z, "null tensor");
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(reparam_kernel<bf16_t>, dim3(grid1d(M * Z)), dim3(256), 0, STREAM(stream), (const bf16_t*)mulv, ld, eps, z, mu, logvar, (long)M, Z),
    hipLaunchKernelGGL(reparam_kernel<float>, dim3(grid1d(M * Z)), dim3(256), 0, STREAM(stream), (const float*)mulv, ld, eps, z, mu, logvar, (long)M, Z));
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
extern "C" int ipoke_bilinear_cl(const float* x_nchw, float* y_cl, int N, int C, int Hi, int Wi, int Ho, int Wo, void* stream) {
  IPK_REQUIRE(x_nchw && y_cl, "null tensor");
  hipLaunchKernelGGL(bilinear_cl_kernel, dim3(grid1d((long)N * Ho * Wo * C)), dim3(256), 0, STREAM(stream), x_nchw, y_cl, N, C, Hi, Wi, Ho, Wo);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
extern "C" int ipoke_clip_to_cl4(const float* src, int64_t s_n, int64_t s_c, int64_t s_t, int64_t s_h, int64_t s_w, int N, int T, int H,
                                 int W, int pad_l, int pad_r, void* dst, int dtype, void* stream) {
  IPK_REQUIRE(src && dst && N >= 1 && T >= 1 && H >= 1 && W >= 1 && pad_l >= 0 && pad_r >= 0, "bad arguments");
  IPK_REQUIRE(dtype == IPOKE_BF16 || dtype == IPOKE_F32, "bad dtype");
  const int Wp = pad_l + W + pad_r;
  const long rows = (long)N * T * H;
  if (dtype == IPOKE_BF16)
    hipLaunchKernelGGL(clip_to_cl4_kernel<bf16_t>, dim3(grid1d(rows * Wp, 256, 16384)), dim3(256), 0, STREAM(stream), src, (long)s_n, (long)s_c,
                       (long)s_t, (long)s_h, (long)s_w, T, H, W, pad_l, Wp, rows, (bf16_t*)dst);
  else
    hipLaunchKernelGGL(clip_to_cl4_kernel<float>, dim3(grid1d(rows * Wp, 256, 16384)), dim3(256), 0, STREAM(stream), src, (long)s_n, (long)s_c,
                       (long)s_t, (long)s_h, (long)s_w, T, H, W, pad_l, Wp, rows, (float*)dst);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

extern "C" int ipoke_cl_to_nchw(const void* x_cl, int ld, float* y, int N, int C, int S, int dtype, void* stream) {
  IPK_REQUIRE(x_cl && y, "null tensor");
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(cl_to_nchw_kernel<bf16_t>, dim3(grid1d((long)N * C * S)), dim3(256), 0, STREAM(stream), (const bf16_t*)x_cl, ld, y, N, C, S),
    hipLaunchKernelGGL(cl_to_nchw_kernel<float>, dim3(grid1d((long)N * C * S)), dim3(256), 0, STREAM(stream), (const float*)x_cl, ld, y, N, C, S));
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
extern "C" int ipoke_nchw_to_cl(const float* x, void* y_cl, int ld, int N, int C, int S, int dtype, void* stream) {
  IPK_REQUIRE(x && y_cl && ld >= C, "bad arguments");
  DISPATCH_T(dtype,
    hipLaunchKernelGGL(nchw_to_cl_kernel<bf16_t>, dim3(grid1d((long)N * S * ld)), dim3(256), 0, STREAM(stream), x, (bf16_t*)y_cl, ld, N, C, S),
    hipLaunchKernelGGL(nchw_to_cl_kernel<float>, dim3(grid1d((long)N * S * ld)), dim3(256), 0, STREAM(stream), x, (float*)y_cl, ld, N, C, S));
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
