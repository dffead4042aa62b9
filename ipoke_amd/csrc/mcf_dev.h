// Device-side building blocks shared by the per-layer MCF kernels (mcf.hip) and the fused MaCowUnit kernels
// (mcf_unit.hip): LDS staging of the 8x8xC latent, the two matrix-core contractions of a masked-conv flow with
// register-resident weight fragments, autoregressive tap addressing.
#pragma once
#include "common.h"

namespace ipoke {

template <typename T> struct Pack4;     // 4 consecutive values of the compute dtype
template <> struct Pack4<bf16_t> { typedef __attribute__((ext_vector_type(4))) __bf16 type; };
template <> struct Pack4<float> { typedef f32x4 type; };

// MCF workgroups run 8 wave64 (two per SIMD): the kernels are chains of short dependent phases, a second wave per SIMD
// overlaps one wave's LDS/global latency with the other's matrix-core work.
static constexpr int kMcfWaves = 8;
static constexpr int kMcfThreads = kMcfWaves * 64;
static constexpr int kJ16 = 16 / kMcfWaves;      // fragment columns per wave when N <= 256
static constexpr int kJ8 = (8 + kMcfWaves - 1) / kMcfWaves;   // ... when N <= 128

template <typename T> struct K64 { static constexpr int value = 64 / (int)sizeof(T); };   // K per super-step

// 16-byte chunk of the (virtual) im2col row of position p: channels [c, c+E16) of tap `tap`
template <typename T>
__device__ __forceinline__ typename ET<T>::frag gather_fwd(const unsigned char* tile, int pitch, const McfGeom& g, int p,
                                                           int tap, int c, int ntaps) {
  typedef typename ET<T>::frag frag_t;
  frag_t z;
#pragma unroll
  for (int e = 0; e < ET<T>::E16; ++e) z[e] = (T)0.f;
  if (tap >= ntaps) return z;
  const int ky = g.kw == 3 ? (tap >= 3) : (tap >> 1);
  const int kx = tap - ky * g.kw;
  const int yy = (p >> 3) + ky + g.oy, xx = (p & 7) + kx + g.ox;
  if ((unsigned)yy >= 8u || (unsigned)xx >= 8u) return z;
  return *reinterpret_cast<const frag_t*>(tile + (yy * 8 + xx) * pitch + c * (int)sizeof(T));
}
// adjoint gather: positions q whose receptive field contains p through tap `tap`
template <typename T>
__device__ __forceinline__ typename ET<T>::frag gather_adj(const unsigned char* tile, int pitch, const McfGeom& g, int p,
                                                           int tap, int c) {
  typedef typename ET<T>::frag frag_t;
  frag_t z;
#pragma unroll
  for (int e = 0; e < ET<T>::E16; ++e) z[e] = (T)0.f;
  const int ky = g.kw == 3 ? (tap >= 3) : (tap >> 1);
  const int kx = tap - ky * g.kw;
  const int yy = (p >> 3) - ky - g.oy, xx = (p & 7) - kx - g.ox;
  if ((unsigned)yy >= 8u || (unsigned)xx >= 8u) return z;
  return *reinterpret_cast<const frag_t*>(tile + (yy * 8 + xx) * pitch + c * (int)sizeof(T));
}

// LDS address of the row feeding position p through tap (compile-time) `tap`, or the all-zero row when out of range
__device__ __forceinline__ const unsigned char* tap_src_fwd(const unsigned char* tile, const unsigned char* zrow, int pitch,
                                                            const McfGeom& g, int p, int tap) {
  const int ky = g.kw == 3 ? (tap >= 3) : (tap >> 1);
  const int kx = tap - ky * g.kw;
  const int yy = (p >> 3) + ky + g.oy, xx = (p & 7) + kx + g.ox;
  return ((unsigned)yy < 8u && (unsigned)xx < 8u) ? tile + (yy * 8 + xx) * pitch : zrow;
}
__device__ __forceinline__ const unsigned char* tap_src_adj(const unsigned char* tile, const unsigned char* zrow, int pitch,
                                                            const McfGeom& g, int p, int tap) {
  const int ky = g.kw == 3 ? (tap >= 3) : (tap >> 1);
  const int kx = tap - ky * g.kw;
  const int yy = (p >> 3) - ky - g.oy, xx = (p & 7) - kx - g.ox;
  return ((unsigned)yy < 8u && (unsigned)xx < 8u) ? tile + (yy * 8 + xx) * pitch : zrow;
}

// 16-byte chunk [k, k + E16) of row `row` of a weight operand stored in the fragment-tiled order (prep.hip: tiled_offset):
// tiles of 16 rows x 64 bytes, lane-linear inside, so that a wave's B-fragment load is one contiguous KB.
template <typename T>
__device__ __forceinline__ typename ET<T>::frag load_wfrag(const T* W, int ldw, int row, int k) {
  constexpr int KS = 64 / (int)sizeof(T), E16 = ET<T>::E16;
  const int ks = k / KS, cq = k - ks * KS;
  const long off = ((long)(row >> 4) * (ldw / KS) + ks) * (16 * KS) + (cq / E16) * (16 * E16) + (row & 15) * E16;
  return *reinterpret_cast<const typename ET<T>::frag*>(W + off);
}

struct McfParams {
  const float* x; float* y; int ld; int C; int B;
  const void* cond; int Cc;          // T [B][64][Cc] = act(cond)
  const void* W1; int K1p; int H;    // [round16(H)][K1p], k = tap*Cp + c
  const void* W2; int K2p;           // [round16(2C)][K2p], k over [hidden | cond]
  const float* bias2;                // [2C]
  int Cp, order;
  void* a2_save;                     // T [M][K2p] or NULL
  float* scale_save;                 // [M][C] or NULL
  float* ld_slot;                    // [B][RS] or NULL
  // backward only
  const void* W2T; int K3p;          // [round16(H)][K3p] : W2T[n][j] = W2[j][n]
  const void* W1T; int Hq;           // [round16(C)][6*Hq]: W1T[c][tap*Hq + n] = W1[n][tap*Cp + c]
  const float* dy; const float* dld; float* dx;
  void* dparams_save;                // T [M][K3p]
  void* dc_save;                     // T [M][Hq]
  float* dbias_part;                 // [B][2C]
  const float* post_ls; const float* post_bias;   // fused ActNorm after the coupling (or NULL)
  const float* y_post; float* post_part;          // backward of the fused ActNorm
};

// ------------------------------------------------------------------------------------------
// stage the sample's latent (fp32, channels-last with pitch ld) into LDS as T [64][Cp] (zero padded)
template <typename T>
__device__ __forceinline__ void stage_x(const float* xb, int ld, int C, int Cp, unsigned char* xs, int pitch) {
  typedef typename Pack4<T>::type pack_t;
  if (((C | ld | Cp) & 3) == 0) {
    // 16-byte loads, two per thread in flight (a sample is at most 64 x 64 floats = 1024 groups)
    const int G = Cp >> 2, n = 64 * G;
    for (int i0 = threadIdx.x; i0 < n; i0 += 2 * blockDim.x) {
      const int i1 = i0 + blockDim.x;
      const bool has1 = i1 < n;
      const int j1 = has1 ? i1 : i0;
      const int p0 = i0 / G, c0 = (i0 - p0 * G) * 4, p1 = j1 / G, c1 = (j1 - p1 * G) * 4;
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
      if (c0 < C) v0 = *reinterpret_cast<const f32x4*>(xb + (long)p0 * ld + c0);
      if (c1 < C) v1 = *reinterpret_cast<const f32x4*>(xb + (long)p1 * ld + c1);
      pack_t t0, t1;
#pragma unroll
      for (int q = 0; q < 4; ++q) { t0[q] = ET<T>::from_f32(v0[q]); t1[q] = ET<T>::from_f32(v1[q]); }
      *reinterpret_cast<pack_t*>(xs + p0 * pitch + c0 * (int)sizeof(T)) = t0;
      if (has1) *reinterpret_cast<pack_t*>(xs + p1 * pitch + c1 * (int)sizeof(T)) = t1;
    }
    return;
  }
  for (int i = threadIdx.x; i < 64 * Cp; i += blockDim.x) {
    const int p = i / Cp, c = i - p * Cp;
    const float v = c < C ? xb[(long)p * ld + c] : 0.f;
    *reinterpret_cast<T*>(xs + p * pitch + c * (int)sizeof(T)) = ET<T>::from_f32(v);
  }
}

// y[:, C:ld] = x[:, C:ld] for `rows` positions: independent of the coupling, issued first
__device__ __forceinline__ void copy_rest(const float* x, float* y, long row0, int rows, int C, int ld) {
  const int rest = ld - C;
  if (rest <= 0) return;
  if (((C | ld) & 3) == 0) {
    const int R4 = rest >> 2;
    for (int e = threadIdx.x; e < rows * R4; e += blockDim.x) {
      const int p = e / R4, c = C + (e - p * R4) * 4;
      *reinterpret_cast<f32x4*>(y + (row0 + p) * ld + c) = *reinterpret_cast<const f32x4*>(x + (row0 + p) * ld + c);
    }
    return;
  }
  for (int e = threadIdx.x; e < rows * rest; e += blockDim.x) {
    const int p = e / rest, c = C + e - p * rest;
    y[(row0 + p) * ld + c] = x[(row0 + p) * ld + c];
  }
}

// hidden = ELU(A1 x W1^T) for MF*16 rows starting at local row 0 (global position pos0 + row) -> a2[row][0:H]
// rowpos(r) maps a local row to (tile index, position) -- supplied by the caller through lambdas.
template <typename T, int MF, typename RowFn>
__device__ __forceinline__ void mcf_gemm1(const McfParams& P, const McfGeom& g, RowFn rowfn, const unsigned char* zrow,
                                          unsigned char* a2, int a2_pitch) {
  constexpr int KS = K64<T>::value, E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, gq = lane >> 4;
  const int NF1 = (P.H + 15) >> 4;
  f32x4 acc[MF][kJ16];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < kJ16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned char* tile[MF]; int pos[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i) rowfn(i * 16 + r, tile[i], pos[i]);
  const T* W1 = reinterpret_cast<const T*>(P.W1);
  const int xpitch = P.Cp * (int)sizeof(T) + 16;
  // K index = tap*Cp + c with Cp a multiple of the 64-byte K step: a step never straddles taps, so the six taps are
  // unrolled statically (per-tap source rows resolved once: valid neighbour or the all-zero row) and only the
  // channel loop is dynamic.  Keeps the per-step instruction count -- the real limiter with one wave per SIMD -- small.
#pragma unroll
  for (int tap = 0; tap < 6; ++tap) {
    const unsigned char* src[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) src[i] = tap_src_fwd(tile[i], zrow, xpitch, g, pos[i], tap) + E16 * gq * (int)sizeof(T);
    for (int c = 0; c < P.Cp; c += KS) {
      frag_t fa[MF], fb[kJ16];
#pragma unroll
      for (int j = 0; j < kJ16; ++j)
        if (wave + kMcfWaves * j < NF1) fb[j] = load_wfrag<T>(W1, P.K1p, (wave + kMcfWaves * j) * 16 + r, tap * P.Cp + c + E16 * gq);
#pragma unroll
      for (int i = 0; i < MF; ++i) fa[i] = *reinterpret_cast<const frag_t*>(src[i] + c * (int)sizeof(T));
#pragma unroll
      for (int j = 0; j < kJ16; ++j) {
        if (wave + kMcfWaves * j < NF1) {
#pragma unroll
          for (int i = 0; i < MF; ++i) mma64(fa[i], fb[j], acc[i][j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kJ16; ++j) {
    const int n = (wave + kMcfWaves * j) * 16 + 4 * gq;
    if (n < P.H) {     // H is a multiple of 4: the 4 columns of a lane are all valid or all invalid
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        T* dst = reinterpret_cast<T*>(a2 + (i * 16 + r) * a2_pitch) + n;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = ET<T>::from_f32(act_apply(IPOKE_ACT_ELU, acc[i][j][q]));
      }
    }
  }
}

// params[row][0:2C] = A2 x W2^T + bias2  (rows MF*16)
template <typename T, int MF>
__device__ __forceinline__ void mcf_gemm2(const McfParams& P, const unsigned char* a2, int a2_pitch, float* prm, int prm_ld) {
  constexpr int KS = K64<T>::value, E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, gq = lane >> 4;
  const int N2 = 2 * P.C, NF2 = (N2 + 15) >> 4;
  f32x4 acc[MF][kJ8];
#pragma unroll
  for (int i = 0; i < MF; ++i) {
#pragma unroll
    for (int j = 0; j < kJ8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const T* W2 = reinterpret_cast<const T*>(P.W2);
  constexpr int PF = 4;
  const int nsteps = P.K2p / KS;
  frag_t ring[PF][kJ8];
  auto load_b = [&](int st, frag_t* b) {
#pragma unroll
    for (int j = 0; j < kJ8; ++j)
      if (wave + kMcfWaves * j < NF2) b[j] = load_wfrag<T>(W2, P.K2p, (wave + kMcfWaves * j) * 16 + r, st * KS + E16 * gq);
  };
#pragma unroll
  for (int d = 0; d < PF; ++d) if (d < nsteps) load_b(d, ring[d]);
  for (int k0 = 0; k0 < nsteps; k0 += PF) {
#pragma unroll
    for (int d = 0; d < PF; ++d) {
      const int st = k0 + d;
      if (st < nsteps) {
        frag_t fa[MF];
#pragma unroll
        for (int i = 0; i < MF; ++i)
          fa[i] = *reinterpret_cast<const frag_t*>(a2 + (i * 16 + r) * a2_pitch + (st * KS + E16 * gq) * (int)sizeof(T));
#pragma unroll
        for (int j = 0; j < kJ8; ++j) {
          if (wave + kMcfWaves * j < NF2) {
#pragma unroll
            for (int i = 0; i < MF; ++i) mma64(fa[i], ring[d][j], acc[i][j]);
          }
        }
        if (st + PF < nsteps) load_b(st + PF, ring[d]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kJ8; ++j) {
    const int n = (wave + kMcfWaves * j) * 16 + 4 * gq;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (n + q < N2) {
        const float b = P.bias2 ? P.bias2[n + q] : 0.f;
#pragma unroll
        for (int i = 0; i < MF; ++i) prm[(i * 16 + r) * prm_ld + n + q] = acc[i][j][q] + b;
      }
    }
  }
}

// ---- register-resident weights (bf16 fast path) ---------------------------------------------------
// A layer's weights are used once per launch and come from HBM/L2 with ~1 us latency.  Fetching them fragment by
// fragment inside the K loops serialises 12 (forward) to 48 (backward) such latencies per workgroup, which is what
// these kernels used to spend their time on.  With two waves per SIMD a wave owns 256 VGPRs: enough to issue *every*
// B-fragment load of the layer up front (forward: 144 VGPRs) and let the matrix-core loops consume them as they land.
static constexpr int kW1Steps = 2;      // Cp / KS   (C <= 64, bf16)
static constexpr int kW2Steps = 12;     // K2p / KS  (4C + Cc <= 384, bf16)
template <typename T> struct McfW {
  typename ET<T>::frag w1[6][kW1Steps][kJ16];
  typename ET<T>::frag w2[kW2Steps][kJ8];
};
template <typename T>
__device__ __forceinline__ void mcf_preload_w1(const McfParams& P, McfW<T>& w) {
  constexpr int KS = K64<T>::value, E16 = ET<T>::E16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, gq = lane >> 4;
  const int NF1 = (P.H + 15) >> 4, cs = P.Cp / KS;
  const T* W1 = reinterpret_cast<const T*>(P.W1);
#pragma unroll
  for (int tap = 0; tap < 6; ++tap)
#pragma unroll
    for (int st = 0; st < kW1Steps; ++st)
#pragma unroll
      for (int j = 0; j < kJ16; ++j)
        if (st < cs && wave + kMcfWaves * j < NF1)
          w.w1[tap][st][j] = load_wfrag<T>(W1, P.K1p, (wave + kMcfWaves * j) * 16 + r, tap * P.Cp + st * KS + E16 * gq);
}
template <typename T>
__device__ __forceinline__ void mcf_preload_w2(const McfParams& P, McfW<T>& w) {
  constexpr int KS = K64<T>::value, E16 = ET<T>::E16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, gq = lane >> 4;
  const int NF2 = (2 * P.C + 15) >> 4, n2 = P.K2p / KS;
  const T* W2 = reinterpret_cast<const T*>(P.W2);
#pragma unroll
  for (int st = 0; st < kW2Steps; ++st)
#pragma unroll
    for (int j = 0; j < kJ8; ++j)
      if (st < n2 && wave + kMcfWaves * j < NF2)
        w.w2[st][j] = load_wfrag<T>(W2, P.K2p, (wave + kMcfWaves * j) * 16 + r, st * KS + E16 * gq);
}
template <typename T>
__device__ __forceinline__ void mcf_preload(const McfParams& P, McfW<T>& w) {
  mcf_preload_w1<T>(P, w);
  mcf_preload_w2<T>(P, w);
}

template <typename T, int MF, typename RowFn>
__device__ __forceinline__ void mcf_gemm1_pre(const McfParams& P, const McfGeom& g, RowFn rowfn, const unsigned char* zrow,
                                              unsigned char* a2, int a2_pitch, const McfW<T>& w) {
  constexpr int KS = K64<T>::value, E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, gq = lane >> 4;
  const int NF1 = (P.H + 15) >> 4, cs = P.Cp / KS;
  f32x4 acc[MF][kJ16];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < kJ16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned char* tile[MF]; int pos[MF];
#pragma unroll
  for (int i = 0; i < MF; ++i) rowfn(i * 16 + r, tile[i], pos[i]);
  const int xpitch = P.Cp * (int)sizeof(T) + 16;
#pragma unroll
  for (int tap = 0; tap < 6; ++tap) {
    const unsigned char* src[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) src[i] = tap_src_fwd(tile[i], zrow, xpitch, g, pos[i], tap) + E16 * gq * (int)sizeof(T);
#pragma unroll
    for (int st = 0; st < kW1Steps; ++st) {
      if (st < cs) {
        frag_t fa[MF];
#pragma unroll
        for (int i = 0; i < MF; ++i) fa[i] = *reinterpret_cast<const frag_t*>(src[i] + st * KS * (int)sizeof(T));
#pragma unroll
        for (int j = 0; j < kJ16; ++j) {
          if (wave + kMcfWaves * j < NF1) {
#pragma unroll
            for (int i = 0; i < MF; ++i) mma64(fa[i], w.w1[tap][st][j], acc[i][j]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kJ16; ++j) {
    const int n = (wave + kMcfWaves * j) * 16 + 4 * gq;
    if (n < P.H) {
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        typename Pack4<T>::type tv;
#pragma unroll
        for (int q = 0; q < 4; ++q) tv[q] = ET<T>::from_f32(act_apply(IPOKE_ACT_ELU, acc[i][j][q]));
        *reinterpret_cast<typename Pack4<T>::type*>(a2 + (i * 16 + r) * a2_pitch + n * (int)sizeof(T)) = tv;
      }
    }
  }
}

template <typename T, int MF>
__device__ __forceinline__ void mcf_gemm2_pre(const McfParams& P, const unsigned char* a2, int a2_pitch, float* prm, int prm_ld,
                                              const McfW<T>& w) {
  constexpr int KS = K64<T>::value, E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, gq = lane >> 4;
  const int N2 = 2 * P.C, NF2 = (N2 + 15) >> 4, n2 = P.K2p / KS;
  f32x4 acc[MF][kJ8];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < kJ8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int st = 0; st < kW2Steps; ++st) {
    if (st < n2) {
      frag_t fa[MF];
#pragma unroll
      for (int i = 0; i < MF; ++i)
        fa[i] = *reinterpret_cast<const frag_t*>(a2 + (i * 16 + r) * a2_pitch + (st * KS + E16 * gq) * (int)sizeof(T));
#pragma unroll
      for (int j = 0; j < kJ8; ++j) {
        if (wave + kMcfWaves * j < NF2) {
#pragma unroll
          for (int i = 0; i < MF; ++i) mma64(fa[i], w.w2[st][j], acc[i][j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kJ8; ++j) {
    const int n = (wave + kMcfWaves * j) * 16 + 4 * gq;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (n + q < N2) {
        const float b = P.bias2 ? P.bias2[n + q] : 0.f;
#pragma unroll
        for (int i = 0; i < MF; ++i) prm[(i * 16 + r) * prm_ld + n + q] = acc[i][j][q] + b;
      }
    }
  }
}

// copy act(cond) rows and zero the K padding of the 1x1 conv's input tile
template <typename T, typename RowFn>
__device__ __forceinline__ void fill_cond(const McfParams& P, int rows, RowFn grow /* local row -> global row or -1 */,
                                          unsigned char* a2, int a2_pitch) {
  constexpr int E16 = ET<T>::E16;
  const int chunks = (P.K2p - P.H) / E16;              // cond columns + zero padding, in 16-byte chunks
  const T* cond = reinterpret_cast<const T*>(P.cond);
  for (int i = threadIdx.x; i < rows * chunks; i += blockDim.x) {
    const int row = i / chunks, ch = i - row * chunks;
    const long gr = grow(row);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (gr >= 0 && ch * E16 < P.Cc) v = *reinterpret_cast<const u32x4*>(cond + gr * P.Cc + ch * E16);
    *reinterpret_cast<u32x4*>(a2 + row * a2_pitch + (P.H + ch * E16) * (int)sizeof(T)) = v;
  }
}

// host: raise the dynamic-LDS limit of a kernel instantiation (grows monotonically)
template <auto Kern>
static int ensure_lds(size_t bytes) {
  static size_t granted = 0;          // per kernel instantiation; grows monotonically
  if (bytes > granted) {
    IPK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    granted = bytes;
  }
  return IPOKE_OK;
}

}  // namespace ipoke
