// Fused Masked-Convolutional-Flow kernels (reference models/modules/INN/macow2.py:25-288 and
// macow_utils.py:407-499).  One MCF layer is
//     c = shiftconv_{2x3|3x2}(x)  ->  ELU(cat[c, h])  ->  weight-normed 1x1  ->  (mu, s)
//     y = (tanh(s/2)+1) * x + mu ,  logdet = sum log scale
// and all of it runs inside one workgroup per sample (or per slice of a sample): the 8x8xC latent,
// the hidden activations and the coupling parameters never leave LDS; only weights stream in (from L2)
// as matrix-core B fragments.  The analytic inverse walks the 8 rows/columns inside a single launch.
//
// Matrix-core tiling: 4 wave64 split the output channels; each wave owns MF x NFW 16x16 fragments.
// A fragments come from LDS (gathered with the autoregressive tap offsets), B fragments are 16-byte
// global loads of the K-contiguous weight shadows prepared by ipoke_flow_prepare_weights.
#include "mcf_dev.h"

namespace ipoke {

// ------------------------------------------------------------------------------------------ forward
template <typename T, int MF, bool FAST>
__global__ __launch_bounds__(kMcfThreads) void mcf_fwd_kernel(const McfParams P) {
  constexpr int MT = MF * 16, RS = 64 / MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float red[8];
  McfW<T> wr;
  if constexpr (FAST) mcf_preload<T>(P, wr);
  const int b = blockIdx.x / RS, rs = blockIdx.x % RS;
  const int pos0 = rs * MT;
  const McfGeom g = mcf_geom(P.order);
  const int xs_pitch = P.Cp * (int)sizeof(T) + 16;
  const int a2_pitch = P.K2p * (int)sizeof(T) + 16;
  const int N2 = 2 * P.C;
  unsigned char* xs = smem;                                    // 64 rows + one all-zero row
  unsigned char* a2 = xs + 65 * xs_pitch;
  float* prm = reinterpret_cast<float*>(a2 + MT * a2_pitch);

  const float* xb = P.x + (long)b * 64 * P.ld;
  const long row0 = (long)b * 64 + pos0;
  // everything that only depends on the input is requested up front: the pass-through channels, the fp32 x of this
  // slice for the affine epilogue, the latent tile and the conditioning rows
  const bool vec = ((P.C | P.ld) & 3) == 0;
  const int G4 = P.C >> 2;
  constexpr int NE = (MF + 1) / 2;                 // epilogue items (row, 4-channel group) per thread
  f32x4 xe[NE];
  if (vec) {
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int e = threadIdx.x + k * kMcfThreads;
      if (e < MT * G4) { const int p = e / G4, c = (e - p * G4) * 4; xe[k] = *reinterpret_cast<const f32x4*>(P.x + (row0 + p) * P.ld + c); }
    }
  }
  copy_rest(P.x, P.y, row0, MT, P.C, P.ld);
  stage_x<T>(xb, P.ld, P.C, P.Cp, xs, xs_pitch);
  for (int i = threadIdx.x; i < xs_pitch / 4; i += blockDim.x) reinterpret_cast<unsigned*>(xs + 64 * xs_pitch)[i] = 0u;
  fill_cond<T>(P, MT, [&](int row) { return (long)b * 64 + pos0 + row; }, a2, a2_pitch);
  __syncthreads();
  auto rowfn = [&](int row, const unsigned char*& tile, int& pos) { tile = xs; pos = pos0 + row; };
  if constexpr (FAST) mcf_gemm1_pre<T, MF>(P, g, rowfn, xs + 64 * xs_pitch, a2, a2_pitch, wr);
  else mcf_gemm1<T, MF>(P, g, rowfn, xs + 64 * xs_pitch, a2, a2_pitch);
  __syncthreads();
  if (P.a2_save) {
    constexpr int E16 = ET<T>::E16;
    const int chunks = P.K2p / E16;
    T* dst = reinterpret_cast<T*>(P.a2_save) + ((long)b * 64 + pos0) * P.K2p;
    for (int i = threadIdx.x; i < MT * chunks; i += blockDim.x) {
      const int row = i / chunks, ch = i - row * chunks;
      *reinterpret_cast<u32x4*>(dst + (long)row * P.K2p + ch * E16) =
          *reinterpret_cast<const u32x4*>(a2 + row * a2_pitch + ch * 16);
    }
  }
  if constexpr (FAST) mcf_gemm2_pre<T, MF>(P, a2, a2_pitch, prm, N2, wr);
  else mcf_gemm2<T, MF>(P, a2, a2_pitch, prm, N2);
  __syncthreads();
  float ld_acc = 0.f;
  if (vec) {
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int e = threadIdx.x + k * kMcfThreads;
      if (e < MT * G4) {
        const int p = e / G4, c = (e - p * G4) * 4;
        const f32x4 mu = *reinterpret_cast<const f32x4*>(prm + p * N2 + c);
        const f32x4 sv = *reinterpret_cast<const f32x4*>(prm + p * N2 + P.C + c);
        f32x4 sc, yv;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          sc[q] = tanhf(0.5f * sv[q]) + 1.f;
          yv[q] = sc[q] * xe[k][q] + mu[q];
          ld_acc += logf(sc[q]);
        }
        if (P.post_ls) {                      // fused ActNorm (its log-det is a constant handled by actnorm_logdet)
          const f32x4 pl = *reinterpret_cast<const f32x4*>(P.post_ls + c), pb = *reinterpret_cast<const f32x4*>(P.post_bias + c);
#pragma unroll
          for (int q = 0; q < 4; ++q) yv[q] = yv[q] * expf(pl[q]) + pb[q];
        }
        *reinterpret_cast<f32x4*>(P.y + (row0 + p) * P.ld + c) = yv;
        if (P.scale_save) *reinterpret_cast<f32x4*>(P.scale_save + (row0 + p) * P.C + c) = sc;
      }
    }
  } else {
    for (int e = threadIdx.x; e < MT * P.C; e += blockDim.x) {
      const int p = e / P.C, c = e - p * P.C;
      const float mu = prm[p * N2 + c], sv = prm[p * N2 + P.C + c];
      const float sc = tanhf(0.5f * sv) + 1.f;
      const long off = (row0 + p) * P.ld + c;
      P.y[off] = sc * P.x[off] + mu;
      if (P.scale_save) P.scale_save[(row0 + p) * P.C + c] = sc;
      ld_acc += logf(sc);
    }
  }
  const float tot = block_sum(ld_acc, red);
  if (threadIdx.x == 0 && P.ld_slot) P.ld_slot[(long)b * RS + rs] = tot;
}

// ------------------------------------------------------------------------------------------ inverse
// Two samples per workgroup: a strip of 8 positions per sample fills one 16-row matrix-core tile.
template <typename T, bool FAST>
__global__ __launch_bounds__(kMcfThreads) void mcf_inv_kernel(const McfParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  McfW<T> wr;                                 // the 8 strips reuse the same weights: fetched once
  if constexpr (FAST) mcf_preload<T>(P, wr);
  const int b0 = blockIdx.x * 2;
  const int nb = min(2, P.B - b0);
  const McfGeom g = mcf_geom(P.order);
  const int xs_pitch = P.Cp * (int)sizeof(T) + 16;
  const int a2_pitch = P.K2p * (int)sizeof(T) + 16;
  const int N2 = 2 * P.C;
  unsigned char* xs = smem;                                   // [2][64] rows of T + one all-zero row
  unsigned char* a2 = xs + 129 * xs_pitch;                    // [16] rows
  float* prm = reinterpret_cast<float*>(a2 + 16 * a2_pitch);  // [16][2C]
  float* xf = prm + 16 * N2;                                  // [2][64][C] exact reconstruction

  for (int i = threadIdx.x; i < 129 * xs_pitch / 4; i += blockDim.x) reinterpret_cast<unsigned*>(xs)[i] = 0u;
  __syncthreads();
  const bool rows_first = P.order < 2, backwards = (P.order & 1);
  for (int step = 0; step < 8; ++step) {
    const int i = backwards ? 7 - step : step;
    auto strip_pos = [&](int j) { return rows_first ? i * 8 + j : j * 8 + i; };
    fill_cond<T>(P, 16, [&](int row) -> long {
      const int s = row >> 3;
      return s < nb ? (long)(b0 + s) * 64 + strip_pos(row & 7) : -1L;
    }, a2, a2_pitch);
    auto rowfn = [&](int row, const unsigned char*& tile, int& pos) {
      tile = xs + (row >> 3) * 64 * xs_pitch; pos = strip_pos(row & 7);
    };
    if constexpr (FAST) mcf_gemm1_pre<T, 1>(P, g, rowfn, xs + 128 * xs_pitch, a2, a2_pitch, wr);
    else mcf_gemm1<T, 1>(P, g, rowfn, xs + 128 * xs_pitch, a2, a2_pitch);
    __syncthreads();
    if constexpr (FAST) mcf_gemm2_pre<T, 1>(P, a2, a2_pitch, prm, N2, wr);
    else mcf_gemm2<T, 1>(P, a2, a2_pitch, prm, N2);
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * P.C; e += blockDim.x) {
      const int row = e / P.C, c = e - row * P.C;
      const int s = row >> 3;
      if (s < nb) {
        const int p = strip_pos(row & 7);
        const float mu = prm[row * N2 + c], sv = prm[row * N2 + P.C + c];
        const float sc = tanhf(0.5f * sv) + 1.f;
        const float yv = P.x[((long)(b0 + s) * 64 + p) * P.ld + c];
        const float xv = (yv - mu) / (sc + 1e-12f);
        xf[(s * 64 + p) * P.C + c] = xv;
        *reinterpret_cast<T*>(xs + (s * 64 + p) * xs_pitch + c * (int)sizeof(T)) = ET<T>::from_f32(xv);
      }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < nb * 64 * P.ld; e += blockDim.x) {
    const int c = e % P.ld, sp = e / P.ld;        // sp = s*64 + p
    const long off = ((long)b0 * 64 + sp) * P.ld + c;
    P.y[off] = c < P.C ? xf[sp * P.C + c] : P.x[off];
  }
}

// ------------------------------------------------------------------------------------------ backward (data path)
template <typename T, bool FAST>
__global__ __launch_bounds__(kMcfThreads) void mcf_bwd_kernel(const McfParams P) {
  constexpr int KS = K64<T>::value, E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  // FAST: every weight fragment of the layer and the saved ELU outputs are requested before anything else (see McfW).
  // Phase (c) splits K over the two wave groups: wave w owns channel fragment w & 3 and taps 3*(w >> 2) .. +2.
  typedef typename Pack4<T>::type pack_t;
  frag_t w2t[4][kJ16];
  frag_t w1t[3][8];
  pack_t cact[4][kJ16];
  if constexpr (FAST) {
    const int lane0 = threadIdx.x & 63, wave0 = threadIdx.x >> 6, r0 = lane0 & 15, gq0 = lane0 >> 4;
    const int NFh = (P.H + 15) >> 4, n3 = P.K3p / KS, hs = P.Hq / KS;
    const T* W2T = reinterpret_cast<const T*>(P.W2T);
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int j = 0; j < kJ16; ++j)
        if (st < n3 && wave0 + kMcfWaves * j < NFh)
          w2t[st][j] = load_wfrag<T>(W2T, P.K3p, (wave0 + kMcfWaves * j) * 16 + r0, st * KS + E16 * gq0);
    const T* a2s0 = reinterpret_cast<const T*>(P.a2_save);
#pragma unroll
    for (int j = 0; j < kJ16; ++j) {
      const int n = (wave0 + kMcfWaves * j) * 16 + 4 * gq0;
      if (n < P.H) {
#pragma unroll
        for (int i = 0; i < 4; ++i) cact[i][j] = *reinterpret_cast<const pack_t*>(a2s0 + ((long)b * 64 + i * 16 + r0) * P.K2p + n);
      }
    }
    const int nfrag0 = wave0 & 3, kh0 = wave0 >> 2;
    if (nfrag0 < ((P.C + 15) >> 4)) {
      const T* W1T = reinterpret_cast<const T*>(P.W1T);
      const int Ktot = 6 * P.Hq;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int st = 0; st < 8; ++st)
          if (st < hs) w1t[t][st] = load_wfrag<T>(W1T, Ktot, nfrag0 * 16 + r0, (kh0 * 3 + t) * P.Hq + st * KS + E16 * gq0);
    }
  }
  const McfGeom g = mcf_geom(P.order);
  const int N2 = 2 * P.C;
  const int dp_pitch = P.K3p * (int)sizeof(T) + 16;
  const int dc_pitch = P.Hq * (int)sizeof(T) + 16;
  unsigned char* dp = smem;                                       // [64][K3p] T
  unsigned char* dc = dp + 64 * dp_pitch;                         // [64][Hq]  T + one all-zero row
  float* dxd = reinterpret_cast<float*>(dc + 65 * dc_pitch);      // [64][C]   dy*scale
  float* colsum = dxd + 64 * P.C;                                 // [2C]   (scalar fallback path)
  float* psum = colsum + N2;                                      // [<=64 rows][2C] per-thread partial column sums
  float* red2 = psum + 2 * 4096;                                  // [2][Q][2C]  (second halves: fused ActNorm sums)
  const long row0 = (long)b * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, gq = lane >> 4;

  copy_rest(P.dy, P.dx, row0, 64, P.C, P.ld);      // pass-through channels: independent of everything below
  for (int i = threadIdx.x; i < N2; i += blockDim.x) colsum[i] = 0.f;
  for (int i = threadIdx.x; i < 65 * dc_pitch / 16; i += blockDim.x) reinterpret_cast<u32x4*>(dc)[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  // (a) gradients of the coupling parameters.  Vector path: every thread owns 4-channel groups of one row, so dy / x /
  // scale arrive as independent 16-byte loads and the column sums need one LDS atomic per channel and thread.
  const float g_ld = P.dld[b];
  T* dps = reinterpret_cast<T*>(P.dparams_save);
  const bool vec_a = (P.C & 3) == 0 && (P.ld & 3) == 0;
  if (vec_a) {
    // thread (r0, c4) owns the 4-channel group c4 of rows r0 and r0 + rows_par: both rows' loads are in flight together
    // and the column sums start as per-thread partials (deterministic; LDS float atomics cost ~10 us here)
    const int G4 = P.C >> 2;                                   // 4-channel groups per row (<= 16)
    const int rows_par = kMcfThreads / G4;                     // >= 32
    const int rows_used = rows_par < 64 ? rows_par : 64;
    const int c4 = threadIdx.x % G4, r0 = threadIdx.x / G4, c = c4 * 4;
    if (r0 < rows_used) {
      const int nit = r0 + rows_par < 64 ? 2 : 1;
      int pp[2];
      pp[0] = r0; pp[1] = nit == 2 ? r0 + rows_par : r0;
      f32x4 gyv[2], xvv[2], scv[2], ypv[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        gyv[k] = *reinterpret_cast<const f32x4*>(P.dy + (row0 + pp[k]) * P.ld + c);
        xvv[k] = *reinterpret_cast<const f32x4*>(P.x + (row0 + pp[k]) * P.ld + c);
        scv[k] = *reinterpret_cast<const f32x4*>(P.scale_save + (row0 + pp[k]) * P.C + c);
        if (P.post_ls) ypv[k] = *reinterpret_cast<const f32x4*>(P.y_post + (row0 + pp[k]) * P.ld + c);
      }
      f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sd = sg, s_ls = sg, s_b = sg;
      if (P.post_ls) {
        // backward of the fused ActNorm: dls = sum dy*(y_post - bias), dbias = sum dy, gradient passed on = dy*exp(ls)
        const f32x4 pl = *reinterpret_cast<const f32x4*>(P.post_ls + c), pb = *reinterpret_cast<const f32x4*>(P.post_bias + c);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (k < nit) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float g0 = gyv[k][q];
              s_ls[q] += g0 * (ypv[k][q] - pb[q]);
              s_b[q] += g0;
              gyv[k][q] = g0 * expf(pl[q]);
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (k < nit) {
          const int p = pp[k];
          const f32x4 gy = gyv[k], xv = xvv[k], sc = scv[k];
          f32x4 ds, dxv;
          pack_t tm, ts;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float t = sc[q] - 1.f;
            ds[q] = (gy[q] * xv[q] + g_ld / sc[q]) * 0.5f * (1.f - t * t);
            dxv[q] = gy[q] * sc[q];
            tm[q] = ET<T>::from_f32(gy[q]); ts[q] = ET<T>::from_f32(ds[q]);
            sg[q] += gy[q]; sd[q] += ds[q];
          }
          *reinterpret_cast<f32x4*>(dxd + p * P.C + c) = dxv;
          *reinterpret_cast<pack_t*>(dp + p * dp_pitch + c * (int)sizeof(T)) = tm;
          *reinterpret_cast<pack_t*>(dp + p * dp_pitch + (P.C + c) * (int)sizeof(T)) = ts;
          if (dps) {
            *reinterpret_cast<pack_t*>(dps + (row0 + p) * P.K3p + c) = tm;
            *reinterpret_cast<pack_t*>(dps + (row0 + p) * P.K3p + P.C + c) = ts;
          }
        }
      }
      *reinterpret_cast<f32x4*>(psum + r0 * N2 + c) = sg;
      *reinterpret_cast<f32x4*>(psum + r0 * N2 + P.C + c) = sd;
      if (P.post_ls) {
        *reinterpret_cast<f32x4*>(psum + 4096 + r0 * N2 + c) = s_ls;
        *reinterpret_cast<f32x4*>(psum + 4096 + r0 * N2 + P.C + c) = s_b;
      }
    }
    const int padc = P.K3p - N2;                               // zero the K padding (LDS tile and saved tensor)
    for (int e = threadIdx.x; e < 64 * padc; e += blockDim.x) {
      const int p = e / padc, j = N2 + e - p * padc;
      *reinterpret_cast<T*>(dp + p * dp_pitch + j * (int)sizeof(T)) = (T)0.f;
      if (dps) dps[(row0 + p) * P.K3p + j] = (T)0.f;
    }
  } else {
    for (int e = threadIdx.x; e < 64 * P.K3p; e += blockDim.x) {
      const int p = e / P.K3p, j = e - p * P.K3p;
      float v = 0.f;
      if (j < N2) {
        const int c = j < P.C ? j : j - P.C;
        const float gy = P.dy[(row0 + p) * P.ld + c];
        const float sc = P.scale_save[(row0 + p) * P.C + c];
        if (j < P.C) {
          v = gy;
          dxd[p * P.C + c] = gy * sc;
        } else {
          const float t = sc - 1.f;
          v = (gy * P.x[(row0 + p) * P.ld + c] + g_ld / sc) * 0.5f * (1.f - t * t);
        }
        psum[p * N2 + j] = v;            // (column sums below, rows in order: no LDS float atomics, run-to-run reproducible)
      }
      const T tv = ET<T>::from_f32(v);
      *reinterpret_cast<T*>(dp + p * dp_pitch + j * (int)sizeof(T)) = tv;
      if (dps) dps[(row0 + p) * P.K3p + j] = tv;
    }
  }
  __syncthreads();
  if (vec_a) {
    // column sums of the per-thread partials: Q threads per column, then one thread per column
    const int rows_par = kMcfThreads / (P.C >> 2);
    const int rows_used = rows_par < 64 ? rows_par : 64;
    const int Q = kMcfThreads / N2;                            // >= 4
    const int col = threadIdx.x % N2, part = threadIdx.x / N2;
    if (part < Q) {
      float t = 0.f, t2 = 0.f;
      for (int rr = part; rr < rows_used; rr += Q) t += psum[rr * N2 + col];
      red2[part * N2 + col] = t;
      if (P.post_ls) {
        for (int rr = part; rr < rows_used; rr += Q) t2 += psum[4096 + rr * N2 + col];
        red2[512 + part * N2 + col] = t2;
      }
    }
    __syncthreads();
    if (threadIdx.x < N2 && P.dbias_part) {
      float t = 0.f;
      for (int q = 0; q < Q; ++q) t += red2[q * N2 + threadIdx.x];
      P.dbias_part[(long)b * N2 + threadIdx.x] = t;
    }
    if (threadIdx.x < N2 && P.post_ls && P.post_part) {
      float t = 0.f;
      for (int q = 0; q < Q; ++q) t += red2[512 + q * N2 + threadIdx.x];
      // [d_log_scale | d_bias]; the log-det term of the ActNorm adds P * dld[b] to every d_log_scale
      P.post_part[(long)b * N2 + threadIdx.x] = threadIdx.x < P.C ? t + 64.f * g_ld : t;
    }
  } else if (P.dbias_part) {
    for (int i = threadIdx.x; i < N2; i += blockDim.x) {
      float t = 0.f;
      for (int p = 0; p < 64; ++p) t += psum[p * N2 + i];
      P.dbias_part[(long)b * N2 + i] = t;
    }
  }

  // (b) dA2[:, :H] = dparams x W2[:, :H]  , times ELU'(c) -> dc
  {
    const int NF = (P.H + 15) >> 4;
    f32x4 acc[4][kJ16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < kJ16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const T* a2s = reinterpret_cast<const T*>(P.a2_save);
    const int nsteps = P.K3p / KS;
    if constexpr (FAST) {
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        if (st < nsteps) {
          frag_t fa[4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            fa[i] = *reinterpret_cast<const frag_t*>(dp + (i * 16 + r) * dp_pitch + (st * KS + E16 * gq) * (int)sizeof(T));
#pragma unroll
          for (int j = 0; j < kJ16; ++j)
            if (wave + kMcfWaves * j < NF) {
#pragma unroll
              for (int i = 0; i < 4; ++i) mma64(fa[i], w2t[st][j], acc[i][j]);
            }
        }
      }
    } else {
      const T* W2T = reinterpret_cast<const T*>(P.W2T);
      constexpr int PF = 4;
      frag_t ring[PF][kJ16];
      auto load_b = [&](int st, frag_t* bb) {
#pragma unroll
        for (int j = 0; j < kJ16; ++j)
          if (wave + kMcfWaves * j < NF) bb[j] = load_wfrag<T>(W2T, P.K3p, (wave + kMcfWaves * j) * 16 + r, st * KS + E16 * gq);
      };
#pragma unroll
      for (int d = 0; d < PF; ++d) if (d < nsteps) load_b(d, ring[d]);
      for (int k0 = 0; k0 < nsteps; k0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
          const int st = k0 + d;
          if (st < nsteps) {
            frag_t fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
              fa[i] = *reinterpret_cast<const frag_t*>(dp + (i * 16 + r) * dp_pitch + (st * KS + E16 * gq) * (int)sizeof(T));
#pragma unroll
            for (int j = 0; j < kJ16; ++j)
              if (wave + kMcfWaves * j < NF) {
#pragma unroll
                for (int i = 0; i < 4; ++i) mma64(fa[i], ring[d][j], acc[i][j]);
              }
            if (st + PF < nsteps) load_b(st + PF, ring[d]);
          }
        }
      }
    }
    T* dcs = reinterpret_cast<T*>(P.dc_save);
#pragma unroll
    for (int j = 0; j < kJ16; ++j) {
      const int n = (wave + kMcfWaves * j) * 16 + 4 * gq;
      if (n < P.H) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p = i * 16 + r;
          pack_t ca;
          if constexpr (FAST) ca = cact[i][j];
          else ca = *reinterpret_cast<const pack_t*>(a2s + (row0 + p) * P.K2p + n);    // K2p, n multiples of 4
          pack_t tv;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            tv[q] = ET<T>::from_f32(acc[i][j][q] * act_grad_from_out(IPOKE_ACT_ELU, ET<T>::to_f32(ca[q])));
          *reinterpret_cast<pack_t*>(dc + p * dc_pitch + n * (int)sizeof(T)) = tv;
          if (dcs) *reinterpret_cast<pack_t*>(dcs + (row0 + p) * P.Hq + n) = tv;
        }
      }
    }
    if (dcs) {   // zero the K padding of the saved tensor (read by the weight-gradient GEMM)
      const int padc = P.Hq - P.H;
      for (int e = threadIdx.x; e < 64 * padc; e += blockDim.x) {
        const int p = e / padc, c = P.H + e - p * padc;
        dcs[(row0 + p) * P.Hq + c] = (T)0.f;
      }
    }
  }
  __syncthreads();

  // (c) dx = dy*scale + sum_tap dc[p - off(tap)] x W1[:, tap, :]
  if constexpr (FAST) {
    // wave w: channel fragment w & 3, taps 3*(w >> 2) .. +2, all four row fragments; the two tap halves meet in LDS
    const int NF = (P.C + 15) >> 4;
    const int nfrag = wave & 3, kh = wave >> 2, hs = P.Hq / KS;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* zrow = dc + 64 * dc_pitch;
    float* part = reinterpret_cast<float*>(dp);      // [64][C] fp32: the dparams tile is dead after (b)
    const int n = nfrag * 16 + 4 * gq;
    if (nfrag < NF) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const unsigned char* src[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) src[i] = tap_src_adj(dc, zrow, dc_pitch, g, i * 16 + r, kh * 3 + t) + E16 * gq * (int)sizeof(T);
#pragma unroll
        for (int st = 0; st < 8; ++st) {
          if (st < hs) {
            frag_t fa[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const frag_t*>(src[i] + st * KS * (int)sizeof(T));
#pragma unroll
            for (int i = 0; i < 4; ++i) mma64(fa[i], w1t[t][st], acc[i]);
          }
        }
      }
      if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (n + q < P.C) part[(i * 16 + r) * P.C + n + q] = acc[i][q];
      }
    }
    __syncthreads();
    if (nfrag < NF && kh == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = i * 16 + r;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (n + q < P.C) P.dx[(row0 + p) * P.ld + n + q] = dxd[p * P.C + n + q] + acc[i][q] + part[p * P.C + n + q];
      }
    }
  } else {
    //     16 (row-fragment, channel-fragment) pairs over the 8 waves: wave w owns channel fragment w & 3 and the two
    //     row fragments 2*(w >> 2), 2*(w >> 2) + 1.
    const int NF = (P.C + 15) >> 4;       // <= 4 channel fragments
    const int nfrag = wave & 3, mbase = (wave >> 2) * 2;
    f32x4 acc[2];
    acc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0];
    const int ntaps = g.kh * g.kw;
    const int Ktot = ntaps * P.Hq;
    const T* W1T = reinterpret_cast<const T*>(P.W1T);
    if (nfrag < NF) {
      // Hq is a multiple of the K step: taps unrolled statically, source rows (or the zero row) resolved once per tap
      const unsigned char* zrow = dc + 64 * dc_pitch;
#pragma unroll
      for (int tap = 0; tap < 6; ++tap) {
        const unsigned char* src[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) src[i] = tap_src_adj(dc, zrow, dc_pitch, g, (mbase + i) * 16 + r, tap) + E16 * gq * (int)sizeof(T);
        for (int c = 0; c < P.Hq; c += KS) {
          const frag_t fb = load_wfrag<T>(W1T, Ktot, nfrag * 16 + r, tap * P.Hq + c + E16 * gq);
          frag_t fa[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const frag_t*>(src[i] + c * (int)sizeof(T));
#pragma unroll
          for (int i = 0; i < 2; ++i) mma64(fa[i], fb, acc[i]);
        }
      }
      const int n = nfrag * 16 + 4 * gq;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int p = (mbase + i) * 16 + r;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (n + q < P.C) P.dx[(row0 + p) * P.ld + n + q] = dxd[p * P.C + n + q] + acc[i][q];
      }
    }
  }
}

static int fill_params(McfParams& P, const ipoke_mcf_desc* d, int dtype) {
  const int esz = dtype == IPOKE_BF16 ? 2 : 4, e16 = 16 / esz, ks = 64 / esz;
  IPK_REQUIRE(d->C >= 2 && d->C <= 64 && d->C % 2 == 0, "MCF supports even channel counts up to 64");
  IPK_REQUIRE(d->order >= 0 && d->order <= 3, "order is 0..3 (A..D)");
  IPK_REQUIRE(d->ld >= d->C, "state pitch smaller than channel count");
  P.x = d->x; P.y = d->y; P.ld = d->ld; P.C = d->C; P.B = d->B;
  P.cond = d->cond; P.Cc = d->Cc;
  P.H = 4 * d->C;
  P.Cp = round_up(d->C, ks);          // channels per tap padded to the 64-byte K step
  P.K1p = 6 * P.Cp;
  P.K2p = round_up(P.H + d->Cc, ks);
  P.K3p = round_up(2 * d->C, ks);
  P.Hq = round_up(P.H, ks);
  IPK_REQUIRE(d->Cc % e16 == 0 && (P.H % e16) == 0, "hidden/cond widths must be multiples of 16 bytes");
  P.W1 = d->W1; P.W2 = d->W2; P.bias2 = d->bias2; P.order = d->order;
  P.a2_save = d->a2_save; P.scale_save = d->scale_save; P.ld_slot = d->logdet_slot;
  P.W2T = d->W2T; P.W1T = d->W1T; P.dy = d->dy; P.dld = d->dld; P.dx = d->dx;
  P.dparams_save = d->dparams_save; P.dc_save = d->dc_save; P.dbias_part = d->dbias_part;
  P.post_ls = d->post_log_scale; P.post_bias = d->post_bias; P.y_post = d->y_post; P.post_part = d->post_part;
  IPK_REQUIRE((P.post_ls == nullptr) == (P.post_bias == nullptr), "fused ActNorm needs log_scale and bias");
  IPK_REQUIRE(!P.post_ls || (((d->C | d->ld) & 3) == 0), "fused ActNorm needs C % 4 == 0 and ld % 4 == 0");
  return IPOKE_OK;
}

// the register-resident weight path covers the shipped widths (C <= 64, 4C + Cc <= 384) in bf16
static bool fast_ok(const McfParams& P) {
  static const bool off = getenv("IPOKE_MCF_NOFAST") != nullptr;
  return !off && P.Cp <= kW1Steps * 32 && P.K2p <= kW2Steps * 32 && P.Hq <= 8 * 32 && P.K3p <= 4 * 32;
}

}  // namespace ipoke

using namespace ipoke;

extern "C" int ipoke_mcf_shadow_dims(int C, int Cc, int dtype, int32_t* dims8) {
  IPK_REQUIRE(dims8 && (dtype == IPOKE_BF16 || dtype == IPOKE_F32), "bad arguments");
  const int esz = dtype == IPOKE_BF16 ? 2 : 4, e16 = 16 / esz, ks = 64 / esz;
  const int H = 4 * C, Cp = round_up(C, ks);
  dims8[0] = Cp;                       // channels per tap in K1 (padded to the 64-byte K step)
  dims8[1] = 6 * Cp;                   // K1p
  dims8[2] = round_up(H + Cc, ks);     // K2p
  dims8[3] = round_up(2 * C, ks);      // K3p
  dims8[4] = round_up(H, ks);          // Hq
  dims8[5] = round_up(H, 16);          // rows of W1 / W2T
  dims8[6] = round_up(2 * C, 16);      // rows of W2
  dims8[7] = round_up(C, 16);          // rows of W1T
  return IPOKE_OK;
}

extern "C" int ipoke_mcf_fwd(const ipoke_mcf_desc* d, int dtype, void* stream) {
  IPK_REQUIRE(d && d->x && d->y && d->cond && d->W1 && d->W2, "null tensor");
  McfParams P;
  int rc = fill_params(P, d, dtype); if (rc) return rc;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int esz = dtype == IPOKE_BF16 ? 2 : 4;
  // rows per workgroup: a full sample when LDS allows it, else half
  auto lds_bytes = [&](int MT) { return (size_t)65 * (P.Cp * esz + 16) + (size_t)MT * (P.K2p * esz + 16) + (size_t)MT * 2 * P.C * 4; };
  int MT = d->rows_per_block > 0 ? d->rows_per_block : (dtype == IPOKE_BF16 ? 32 : 16);
  IPK_REQUIRE(MT == 16 || MT == 32 || MT == 64, "rows_per_block must be 16, 32 or 64");
  const size_t lds = lds_bytes(MT);
  IPK_REQUIRE(lds <= 158 * 1024, "MCF tile does not fit LDS");
  const int RS = 64 / MT;
#define LAUNCH_FWD(TT, MF, FAST)                                                                \
  do {                                                                                          \
    rc = ensure_lds<mcf_fwd_kernel<TT, MF, FAST>>(lds); if (rc) return rc;                        \
    hipLaunchKernelGGL((mcf_fwd_kernel<TT, MF, FAST>), dim3(d->B * RS), dim3(kMcfThreads), lds, s, P);  \
  } while (0)
  if (dtype == IPOKE_BF16 && fast_ok(P)) {
    if (MT == 64) LAUNCH_FWD(bf16_t, 4, true); else if (MT == 32) LAUNCH_FWD(bf16_t, 2, true); else LAUNCH_FWD(bf16_t, 1, true);
  } else if (dtype == IPOKE_BF16) {
    if (MT == 64) LAUNCH_FWD(bf16_t, 4, false); else if (MT == 32) LAUNCH_FWD(bf16_t, 2, false); else LAUNCH_FWD(bf16_t, 1, false);
  } else {
    if (MT == 64) LAUNCH_FWD(float, 4, false); else if (MT == 32) LAUNCH_FWD(float, 2, false); else LAUNCH_FWD(float, 1, false);
  }
#undef LAUNCH_FWD
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

extern "C" int ipoke_mcf_inv(const ipoke_mcf_desc* d, int dtype, void* stream) {
  IPK_REQUIRE(d && d->x && d->y && d->cond && d->W1 && d->W2, "null tensor");
  McfParams P;
  int rc = fill_params(P, d, dtype); if (rc) return rc;
  P.a2_save = nullptr; P.scale_save = nullptr; P.ld_slot = nullptr;
  const int esz = dtype == IPOKE_BF16 ? 2 : 4;
  const size_t lds = (size_t)129 * (P.Cp * esz + 16) + (size_t)16 * (P.K2p * esz + 16) + (size_t)16 * 2 * P.C * 4 +
                     (size_t)128 * P.C * 4;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == IPOKE_BF16 && fast_ok(P)) {
    rc = ensure_lds<mcf_inv_kernel<bf16_t, true>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((mcf_inv_kernel<bf16_t, true>), dim3((d->B + 1) / 2), dim3(kMcfThreads), lds, s, P);
  } else if (dtype == IPOKE_BF16) {
    rc = ensure_lds<mcf_inv_kernel<bf16_t, false>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((mcf_inv_kernel<bf16_t, false>), dim3((d->B + 1) / 2), dim3(kMcfThreads), lds, s, P);
  } else {
    rc = ensure_lds<mcf_inv_kernel<float, false>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((mcf_inv_kernel<float, false>), dim3((d->B + 1) / 2), dim3(kMcfThreads), lds, s, P);
  }
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

extern "C" int ipoke_mcf_bwd(const ipoke_mcf_desc* d, int dtype, void* stream) {
  IPK_REQUIRE(d && d->x && d->dy && d->dx && d->dld && d->W2T && d->W1T && d->a2_save && d->scale_save, "null tensor");
  IPK_REQUIRE(!d->post_log_scale || d->y_post, "the fused ActNorm backward needs the saved output");
  McfParams P;
  int rc = fill_params(P, d, dtype); if (rc) return rc;
  const int esz = dtype == IPOKE_BF16 ? 2 : 4;
  const size_t lds = (size_t)64 * (P.K3p * esz + 16) + (size_t)65 * (P.Hq * esz + 16) + (size_t)64 * P.C * 4 + 2 * P.C * 4 + (2 * 4096 + 2 * 512) * 4;
  IPK_REQUIRE(lds <= 158 * 1024, "MCF backward tile does not fit LDS");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == IPOKE_BF16 && fast_ok(P)) {
    rc = ensure_lds<mcf_bwd_kernel<bf16_t, true>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((mcf_bwd_kernel<bf16_t, true>), dim3(d->B), dim3(kMcfThreads), lds, s, P);
  } else if (dtype == IPOKE_BF16) {
    rc = ensure_lds<mcf_bwd_kernel<bf16_t, false>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((mcf_bwd_kernel<bf16_t, false>), dim3(d->B), dim3(kMcfThreads), lds, s, P);
  } else {
    rc = ensure_lds<mcf_bwd_kernel<float, false>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((mcf_bwd_kernel<float, false>), dim3(d->B), dim3(kMcfThreads), lds, s, P);
  }
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
