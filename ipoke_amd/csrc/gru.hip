// Native unroll of the first stage's ConvGRU (reference models/modules/motion_models/rnn.py:4-133 as SpadeCondMotionModel.forward drives
// it, models/first_stage_motion_model.py:503-514): T steps x L cells on the 8 x 8 latent, forward and backward, issued back to back on the
// caller's stream by host code of this library -- no Python between the cells.  The cells are tiny (B * 64 rows, 64-128 input channels):
// a Python / autograd loop over them queues ~10 torch-level operations per cell and step and left the GPU idle for ~18 ms of a 56 ms
// first-stage training step (profiles/r04_c4_trace_gaps.txt); here a cell is four launches per direction --
//
//   forward   ur = conv3x3(cat[x, h]) ; u = sigmoid(ur[:Ch]), hr = h * sigmoid(ur[Ch:]) ; o = conv3x3(cat[x, hr]) ; h' = h (1 - u) + tanh(o) u
//   backward  (d h', o, u, h) -> d o, d u, d h ; conv_o data gradient ; (d hr, d u, ur, h) -> d ur, d h ; conv_ur data gradient
//
// -- the concatenations never exist as operations: the update kernel writes h' straight into the operand buffers of its consumers (the
// h-half of the same cell's next step, the x-halves of the next cell's two convolutions, the output sequence), and the gate kernel writes
// h * r into the second convolution's operand.  Every operand of every (cell, step) is kept (T * L * ~0.8 MB at B = 20): the weight and
// bias gradients of a cell are then ONE weight-gradient GEMM and one column sum over the rows of all T steps (the weights are shared by
// the steps), issued after the serial data-gradient chain.
#include <atomic>
#include <cstdlib>
#include <iterator>
#include <mutex>
#include <unordered_map>
#include <cstring>
#include <type_traits>
#include <vector>

#include "common.h"

namespace ipoke {

template <typename T> struct GVec;     // 16 bytes
template <> struct GVec<bf16_t> { typedef bf16x8 type; };
template <> struct GVec<float> { typedef f32x4 type; };

// broadcast the constant cell-0 input into the x-halves of all T steps' operands and the initial hidden state into the h-halves of step 0
template <typename T>
__global__ void gru_fill_kernel(const T* __restrict__ x0, int ldx, const T* __restrict__ h0, int ldh, T* __restrict__ XH, T* __restrict__ XHR,
                                long M, int Cx, int Ch, int Kc, int Tn, int L) {
  const long nx = (long)Tn * M * Cx, nh = (long)L * M * Ch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nx + nh; i += (long)gridDim.x * blockDim.x) {
    if (i < nx) {
      const int c = (int)(i % Cx); const long r = i / Cx; const long m = r % M; const long t = r / M;
      const T v = x0[m * ldx + c];
      XH[(t * M + m) * Kc + c] = v; XHR[(t * M + m) * Kc + c] = v;          // cell 0: slot t
    } else {
      const long k = i - nx;
      const int c = (int)(k % Ch); const long r = k / Ch; const long m = r % M; const long l = r / M;
      XH[((l * Tn) * M + m) * Kc + Cx + c] = h0[m * ldh + c];              // cell l, step 0
    }
  }
}
// u = sigmoid(ur[:, :Ch]) ; hr = h * sigmoid(ur[:, Ch:2Ch]) written into the second operand's h-half
template <typename T>
__global__ void gru_gates_fwd_kernel(const T* __restrict__ ur, int ldur, const T* __restrict__ h, int ldh, T* __restrict__ hr, int ldhr,
                                     T* __restrict__ u, long M, int Ch) {
  const long total = M * Ch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / Ch; const int c = (int)(i - m * Ch);
    const float uu = act_apply(IPOKE_ACT_SIGMOID, ET<T>::to_f32(ur[m * ldur + c]));
    const float r = act_apply(IPOKE_ACT_SIGMOID, ET<T>::to_f32(ur[m * ldur + Ch + c]));
    hr[m * ldhr + c] = ET<T>::from_f32(ET<T>::to_f32(h[m * ldh + c]) * r);
    u[m * Ch + c] = ET<T>::from_f32(uu);
  }
}
// h' = h (1 - u) + tanh(o) u, stored to up to four consumers
struct GruDst { void* p[4]; int ld[4]; };
template <typename T>
__global__ void gru_update_fwd_kernel(const T* __restrict__ o, int ldo, const T* __restrict__ u, const T* __restrict__ h, int ldh, GruDst d,
                                      long M, int Ch) {
  const long total = M * Ch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / Ch; const int c = (int)(i - m * Ch);
    const float uu = ET<T>::to_f32(u[m * Ch + c]);
    const T v = ET<T>::from_f32(ET<T>::to_f32(h[m * ldh + c]) * (1.f - uu) + tanhf(ET<T>::to_f32(o[m * ldo + c])) * uu);
#pragma unroll
    for (int k = 0; k < 4; ++k) if (d.p[k]) reinterpret_cast<T*>(d.p[k])[m * d.ld[k] + c] = v;
  }
}
// d h' = sum of up to five sources (the same cell's next step: its two h-path terms; the next cell of this step: its two x-path terms;
// the loss gradient of the output sequence) ; d o_pre = d h' u (1 - tanh(o)^2) ; d u = d h' (tanh(o) - h) ; d h (direct) = d h' (1 - u)
struct GruSrc { const void* p[5]; int ld[5]; };
template <typename T>
__global__ void gru_update_bwd_kernel(GruSrc s, const T* __restrict__ o, int ldo, const T* __restrict__ u, const T* __restrict__ h, int ldh,
                                      T* __restrict__ d_o, int lddo, T* __restrict__ d_u, T* __restrict__ d_h1, long M, int Ch) {
  const long total = M * lddo;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / lddo; const int c = (int)(i - m * lddo);
    if (c >= Ch) { d_o[m * lddo + c] = ET<T>::from_f32(0.f); continue; }        // zero K padding of the data-gradient GEMM
    float g = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) if (s.p[k]) g += ET<T>::to_f32(reinterpret_cast<const T*>(s.p[k])[m * s.ld[k] + c]);
    const float uu = ET<T>::to_f32(u[m * Ch + c]), th = tanhf(ET<T>::to_f32(o[m * ldo + c])), hv = ET<T>::to_f32(h[m * ldh + c]);
    d_o[m * lddo + c] = ET<T>::from_f32(g * uu * (1.f - th * th));
    d_u[m * Ch + c] = ET<T>::from_f32(g * (th - hv));
    d_h1[m * Ch + c] = ET<T>::from_f32(g * (1.f - uu));
  }
}
// d ur_pre = [d u * u (1 - u) | d hr * h * r (1 - r)] ; d h (so far) = d h1 + d hr * r
template <typename T>
__global__ void gru_gates_bwd_kernel(const T* __restrict__ ur, int ldur, const T* __restrict__ h, int ldh, const T* __restrict__ d_hr, int lddhr,
                                     const T* __restrict__ d_u, const T* __restrict__ d_h1, T* __restrict__ d_ur, int lddur,
                                     T* __restrict__ d_h2, long M, int Ch) {
  const long total = M * lddur;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / lddur; const int c = (int)(i - m * lddur);
    if (c >= 2 * Ch) { d_ur[m * lddur + c] = ET<T>::from_f32(0.f); continue; }
    if (c < Ch) {
      const float uu = act_apply(IPOKE_ACT_SIGMOID, ET<T>::to_f32(ur[m * ldur + c]));
      d_ur[m * lddur + c] = ET<T>::from_f32(ET<T>::to_f32(d_u[m * Ch + c]) * uu * (1.f - uu));
    } else {
      const int k = c - Ch;
      const float r = act_apply(IPOKE_ACT_SIGMOID, ET<T>::to_f32(ur[m * ldur + c]));
      const float ghr = ET<T>::to_f32(d_hr[m * lddhr + k]);
      d_ur[m * lddur + c] = ET<T>::from_f32(ghr * ET<T>::to_f32(h[m * ldh + k]) * r * (1.f - r));
      d_h2[m * Ch + k] = ET<T>::from_f32(ET<T>::to_f32(d_h1[m * Ch + k]) + ghr * r);
    }
  }
}
// acc[m][c] (+)= a[m][c] + b[m][c]   (fp32 accumulator: the gradient of the constant cell-0 input over the steps, and d h0 per cell)
template <typename T>
__global__ void gru_add2_kernel(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb, float* __restrict__ acc, int ldacc, long M, int C,
                                int accumulate) {
  const long total = M * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / C; const int c = (int)(i - m * C);
    const float v = ET<T>::to_f32(a[m * lda + c]) + ET<T>::to_f32(b[m * ldb + c]);
    acc[m * ldacc + c] = accumulate ? acc[m * ldacc + c] + v : v;
  }
}

// ------------------------------------------------------------------------------------------------ fused forward unroll
// The whole T x L recurrence of ONE sample in one workgroup (the cells couple positions of a sample, never samples): 4 launches of
// 4-8 us per (cell, step) -- two implicit GEMMs of 0.05-0.3 GFLOP at launch latency and two element-wise kernels -- become two matrix-
// core phases on LDS-resident operands.  8 x 8 map, bf16, Cx = Ch = CH in {32, 64}.  Per (cell l, step t), tile set l & 1:
//   P0  xh = [x | h_l(t-1)] staged (x of cell 0 is the constant input; of cell l > 0 it was written by cell l - 1 of this step), the
//       operand rows saved for the backward pass (XH, x-half of XHR)
//   P1  ur = conv3x3(xh) + b  ->  u = sigmoid(ur[:CH]), hr = h * sigmoid(ur[CH:]) into the xhr tile; UR, U, hr-half of XHR saved
//   P2  o = conv3x3(xhr) + b  ->  h' = h (1 - u) + tanh(o) u  ->  the cell's state, the x-halves of cell l + 1's tiles, `out` for the last
// Every value is rounded to bf16 where the unfused path stores it (ur, u, hr, o, h'), so that both paths compute the same function of
// the same rounded operands (the fp32 sums differ in their order only).  Weight fragments: one 16-column fragment of the convolution
// per wave and all 9 * Kc / 32 K steps in registers, requested for the next phase as soon as the current phase's MFMAs are done.
// fp32 conv weight [N][Kc][3][3] -> the fused kernel's operand: fragment (column block nf, K step q) is ONE contiguous KB, lane l of it
// the 8 consecutive k = 32 q + 8 (l >> 4) .. of output column 16 nf + (l & 15), k = tap * Kc + channel.  (Read from the row-major
// [N][9 Kc] operand a wave's fragment load touched 16 rows of 64 bytes each: 16 cache lines per instruction instead of 8 full ones.)
__global__ void gru_tile_operand_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int N, int Kc) {
  const long total = (long)N * 9 * Kc;
  const int nfr = 9 * Kc / 32;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63); const long fq = i >> 9;
    const int q = (int)(fq % nfr), nf = (int)(fq / nfr);
    const int o = nf * 16 + (lane & 15), kk = q * 32 + (lane >> 4) * 8 + e;
    const int t = kk / Kc, c = kk - t * Kc;
    out[i] = (bf16_t)w[((long)o * Kc + c) * 9 + t];
  }
}

struct GruFusedPtrs { const bf16_t* w_ur[16]; const bf16_t* w_o[16]; const float* b_ur[16]; const float* b_o[16]; };
template <typename T> struct GPack4;
template <> struct GPack4<bf16_t> { typedef __attribute__((ext_vector_type(4))) __bf16 type; };
template <int CH>
__global__ __launch_bounds__(512) void gru_fused_fwd_kernel(const bf16_t* __restrict__ x0, int ldx, const bf16_t* __restrict__ h0, int ldh,
                                                            const GruFusedPtrs P,
                                                            bf16_t* __restrict__ XH, bf16_t* __restrict__ XHR, bf16_t* __restrict__ UR,
                                                            bf16_t* __restrict__ U, bf16_t* __restrict__ O, bf16_t* __restrict__ out, int ldo,
                                                            long M, int Tn, int L) {
  typedef bf16_t T;
  typedef typename ET<T>::frag frag_t;
  typedef typename GPack4<T>::type pack_t;
  constexpr int KC = 2 * CH, KS = KC / 32, NFR = 9 * KS;            // K steps per tap, weight fragments per wave and convolution
  constexpr int NA = NFR / 2, NB = NFR - NA;                        // the two K halves of a convolution's fragments
  constexpr int PK = KC * 2 + 32;                                   // tile row pitch in bytes: 2 (mod 4) 16-byte units (mcf_unit.hip: kTilePad)
  constexpr int NF1 = 2 * CH / 16, RG1 = 8 / NF1, RT1 = 4 / RG1;    // conv_ur: column fragments, row groups, row tiles per wave
  constexpr int NF2 = CH / 16, RG2 = 8 / NF2, RT2 = 4 / RG2;        // conv_o
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* tiles = smem;                                      // [2 sets][xh, xhr][65 rows][PK]
  T* hst = reinterpret_cast<T*>(tiles + 4 * 65 * PK);               // [L][64][CH] state of the cells
  T* x0s = hst + (long)L * 64 * CH;                                 // [64][CH]
  T* us = x0s + 64 * CH;                                            // [64][CH] update gate of the current cell
  float* bs = reinterpret_cast<float*>(us + 64 * CH);               // [L][3 CH] biases: b_ur (2 CH), b_o (CH)
  const int tid = threadIdx.x;
  const long row0 = (long)blockIdx.x * 64;

  // one-time staging: zero tiles (zero rows stay zero), the constant input, the initial state of every cell, the biases
  for (int i = tid; i < 4 * 65 * PK / 16; i += 512) reinterpret_cast<u32x4*>(tiles)[i] = u32x4{0u, 0u, 0u, 0u};
  constexpr int CV = CH / 8;                                         // 16-byte chunks per half row
  for (int i = tid; i < 64 * CV; i += 512) {
    const int p = i / CV, c = (i - p * CV) * 8;
    const u32x4 xv = *reinterpret_cast<const u32x4*>(x0 + (row0 + p) * ldx + c);
    const u32x4 hv = *reinterpret_cast<const u32x4*>(h0 + (row0 + p) * ldh + c);
    *reinterpret_cast<u32x4*>(x0s + p * CH + c) = xv;
    for (int l = 0; l < L; ++l) *reinterpret_cast<u32x4*>(hst + ((long)l * 64 + p) * CH + c) = hv;
  }
  for (int i = tid; i < L * 3 * CH; i += 512) {
    const int l = i / (3 * CH), c = i - l * 3 * CH;
    bs[i] = c < 2 * CH ? P.b_ur[l][c] : P.b_o[l][c - 2 * CH];
  }
  // Weight fragments: a wave owns ONE 16-column fragment of a convolution and all NFR K steps; they stream through two register sets of
  // half a convolution each -- while the MFMAs of one half run, the other set is in flight (the next half of this convolution, or the first
  // half of the next one), so that the 147-295 KB a convolution reads from L2 per (cell, step) are requested a whole half ahead.
  frag_t wa[NA], wb[NB];
  auto wbase = [&](const T* wop, int nfrag_col, int lane) { return wop + (long)nfrag_col * NFR * 512 + lane * 8; };   // gru_tile_operand_kernel
  auto load_a = [&](const T* base) {
#pragma unroll
    for (int q = 0; q < NA; ++q) wa[q] = *reinterpret_cast<const frag_t*>(base + q * 512);
  };
  auto load_b = [&](const T* base) {
#pragma unroll
    for (int q = 0; q < NB; ++q) wb[q] = *reinterpret_cast<const frag_t*>(base + (NA + q) * 512);
  };
  // address of the tile row feeding position p through tap (ty, tx) of the padded 3 x 3 window
  auto tap_row = [&](const unsigned char* tile, int p, int tap) {
    const int ty = tap / 3, tx = tap - 3 * ty;
    const int yy = (p >> 3) + ty - 1, xx = (p & 7) + tx - 1;
    return ((unsigned)yy < 8u && (unsigned)xx < 8u) ? tile + (yy * 8 + xx) * PK : tile + 64 * PK;
  };
  load_a(wbase(P.w_ur[0], (tid >> 6) % NF1, tid & 63));
  __syncthreads();
#pragma unroll 1
  for (int t = 0; t < Tn; ++t) {
#pragma unroll 1
    for (int l = 0; l < L; ++l) {
      // lane indices from an opaque copy of the thread id: keeps the phases' addresses out of the loops' live-in set (they would sit next
      // to the weight registers for the whole kernel -- mcf_unit.hip)
      int tl = tid;
      asm volatile("" : "+v"(tl));
      const int lane = tl & 63, wave = tl >> 6, r = lane & 15, gq = lane >> 4;
      const int wn1 = wave % NF1, rg1 = wave / NF1, wn2 = wave % NF2, rg2 = wave / NF2;
      unsigned char* xh = tiles + (l & 1) * 2 * 65 * PK;
      unsigned char* xhr = xh + 65 * PK;
      unsigned char* nxh = tiles + ((l + 1) & 1) * 2 * 65 * PK;       // tiles of cell l + 1
      unsigned char* nxhr = nxh + 65 * PK;
      T* hl = hst + (long)l * 64 * CH;
      const long cell = ((long)l * Tn + t) * M + row0;                // first row of this (cell, step, sample) in the per-cell buffers
      const T* w_ur = wbase(P.w_ur[l], wn1, lane);
      const T* w_o = wbase(P.w_o[l], wn2, lane);
      const int ln = l + 1 < L ? l + 1 : 0;
      const bool more = l + 1 < L || t + 1 < Tn;
      const T* w_next = wbase(P.w_ur[ln], wn1, lane);
      // one convolution: K half A (already requested) while half B is requested, then half B while `next_a` (the following convolution's
      // half A) is requested
      auto conv = [&](const unsigned char* tile, auto rt_c, int rowt0, const T* wthis, const T* next_a, bool have_next, f32x4* acc) {
        constexpr int RT = decltype(rt_c)::value;
#pragma unroll
        for (int i = 0; i < RT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        load_b(wthis);
#pragma unroll
        for (int q = 0; q < NFR; ++q) {
          const int tap = q / KS, ks = q - tap * KS;
          if (q == NA) {
            __builtin_amdgcn_sched_barrier(0);
            if (have_next) load_a(next_a);
          }
          frag_t fa[RT];
#pragma unroll
          for (int i = 0; i < RT; ++i)
            fa[i] = *reinterpret_cast<const frag_t*>(tap_row(tile, (rowt0 + i) * 16 + r, tap) + gq * 16 + ks * 64);
#pragma unroll
          for (int i = 0; i < RT; ++i) mma64(fa[i], q < NA ? wa[q < NA ? q : 0] : wb[q < NA ? 0 : q - NA], acc[i]);
        }
      };
      // ---- P0
      for (int i = tl; i < 64 * CV; i += 512) {
        const int p = i / CV, c = (i - p * CV) * 8;
        u32x4 xv;
        if (l == 0) {
          xv = *reinterpret_cast<const u32x4*>(x0s + p * CH + c);
          *reinterpret_cast<u32x4*>(xh + p * PK + c * 2) = xv;
          *reinterpret_cast<u32x4*>(xhr + p * PK + c * 2) = xv;
        } else {
          xv = *reinterpret_cast<const u32x4*>(xh + p * PK + c * 2);
        }
        const u32x4 hv = *reinterpret_cast<const u32x4*>(hl + p * CH + c);
        *reinterpret_cast<u32x4*>(xh + p * PK + (CH + c) * 2) = hv;
        *reinterpret_cast<u32x4*>(XH + (cell + p) * KC + c) = xv;
        *reinterpret_cast<u32x4*>(XH + (cell + p) * KC + CH + c) = hv;
        *reinterpret_cast<u32x4*>(XHR + (cell + p) * KC + c) = xv;
      }
      __syncthreads();
      // ---- P1: ur = conv3x3([x | h]) + b
      {
        f32x4 acc[RT1];
        conv(xh, std::integral_constant<int, RT1>(), rg1 * RT1, w_ur, w_o, true, acc);
        const int n = wn1 * 16 + 4 * gq;                                // columns n .. n + 3 of ur
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bs + l * 3 * CH + n);
#pragma unroll
        for (int i = 0; i < RT1; ++i) {
          const int p = (rg1 * RT1 + i) * 16 + r;
          pack_t urb;
#pragma unroll
          for (int q = 0; q < 4; ++q) urb[q] = ET<T>::from_f32(acc[i][q] + b4[q]);
          *reinterpret_cast<pack_t*>(UR + (cell + p) * KC + n) = urb;
          if (n < CH) {
            pack_t ub;
#pragma unroll
            for (int q = 0; q < 4; ++q) ub[q] = ET<T>::from_f32(act_apply(IPOKE_ACT_SIGMOID, ET<T>::to_f32(urb[q])));
            *reinterpret_cast<pack_t*>(us + p * CH + n) = ub;
            *reinterpret_cast<pack_t*>(U + (cell + p) * CH + n) = ub;
          } else {
            const int c = n - CH;
            const pack_t hb = *reinterpret_cast<const pack_t*>(hl + p * CH + c);
            pack_t hrb;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              hrb[q] = ET<T>::from_f32(ET<T>::to_f32(hb[q]) * act_apply(IPOKE_ACT_SIGMOID, ET<T>::to_f32(urb[q])));
            *reinterpret_cast<pack_t*>(xhr + p * PK + (CH + c) * 2) = hrb;
            *reinterpret_cast<pack_t*>(XHR + (cell + p) * KC + CH + c) = hrb;
          }
        }
      }
      __syncthreads();
      // ---- P2: o = conv3x3([x | h r]) + b ; h' = h (1 - u) + tanh(o) u
      {
        f32x4 acc[RT2];
        conv(xhr, std::integral_constant<int, RT2>(), rg2 * RT2, w_o, w_next, more, acc);
        const int n = wn2 * 16 + 4 * gq;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bs + l * 3 * CH + 2 * CH + n);
#pragma unroll
        for (int i = 0; i < RT2; ++i) {
          const int p = (rg2 * RT2 + i) * 16 + r;
          pack_t ob;
#pragma unroll
          for (int q = 0; q < 4; ++q) ob[q] = ET<T>::from_f32(acc[i][q] + b4[q]);
          *reinterpret_cast<pack_t*>(O + (cell + p) * CH + n) = ob;
          const pack_t hb = *reinterpret_cast<const pack_t*>(hl + p * CH + n);
          const pack_t ub = *reinterpret_cast<const pack_t*>(us + p * CH + n);
          pack_t hn;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float uu = ET<T>::to_f32(ub[q]);
            hn[q] = ET<T>::from_f32(ET<T>::to_f32(hb[q]) * (1.f - uu) + tanhf(ET<T>::to_f32(ob[q])) * uu);
          }
          *reinterpret_cast<pack_t*>(hl + p * CH + n) = hn;
          if (l + 1 < L) {
            *reinterpret_cast<pack_t*>(nxh + p * PK + n * 2) = hn;
            *reinterpret_cast<pack_t*>(nxhr + p * PK + n * 2) = hn;
          } else {
            *reinterpret_cast<pack_t*>(out + ((long)t * M + row0 + p) * ldo + n) = hn;
          }
        }
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------ fused backward unroll
// Operand of a DATA-GRADIENT convolution for the fused backward kernel, from the fp32 weight w [Nw][Kc][3][3]: column block nf / K step q
// tiled as in gru_tile_operand_kernel, output column j = 16 nf + (l & 15) an INPUT channel of the convolution, k = tap * Nw + c with the
// tap mirrored (the gradient of position p collects d[p + delta(tap)] through the weight of tap 8 - tap).
__global__ void gru_tile_operand_t_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int Nw, int Kc) {
  const long total = (long)Kc * 9 * Nw;
  const int nfr = 9 * Nw / 32;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63); const long fq = i >> 9;
    const int q = (int)(fq % nfr), nf = (int)(fq / nfr);
    const int j = nf * 16 + (lane & 15), kk = q * 32 + (lane >> 4) * 8 + e;
    const int t = kk / Nw, c = kk - t * Nw;
    out[i] = (bf16_t)w[((long)c * Kc + j) * 9 + (8 - t)];
  }
}

// The reverse recurrence of ONE sample in one workgroup, on the workspace the forward pass filled (gru_fused_fwd_kernel or the
// launch-per-phase form: the same buffers).  Per (cell l, step t), in reverse order -- the four launches of the unfused form as phases:
//   A  g = sum of the gradients reaching h'(l, t) (same cell's next step: two terms; cell l + 1 of this step: two terms, or the loss
//      gradient of the output sequence) ; d o = g u (1 - tanh(o)^2) ; d u = g (tanh(o) - h) ; d h1 = g (1 - u)
//   B  [d x | d hr] = conv_o data gradient of d o
//   C  d ur = [d u u (1 - u) | d hr h r (1 - r)] ; d h2 = d h1 + d hr r
//   D  [d x | d h] = conv_ur data gradient of d ur
// d o and d ur of every (cell, step) go to the workspace (operands of the weight-gradient GEMMs over all steps); the gradient of the
// constant cell-0 input and of the common initial state are accumulated in fp32 in the order of the unfused form, and every
// intermediate is rounded to bf16 where that form stores it.
template <int CH>
__global__ __launch_bounds__(512) void gru_fused_bwd_kernel(const bf16_t* __restrict__ d_out, int ldo, const GruFusedPtrs P,
                                                            const bf16_t* __restrict__ XH, const bf16_t* __restrict__ UR,
                                                            const bf16_t* __restrict__ U, const bf16_t* __restrict__ O,
                                                            bf16_t* __restrict__ DO, bf16_t* __restrict__ DUR, float* __restrict__ d_x0,
                                                            float* __restrict__ d_h0, long M, int Tn, int L) {
  typedef bf16_t T;
  typedef typename ET<T>::frag frag_t;
  typedef typename GPack4<T>::type pack_t;
  constexpr int KC = 2 * CH;
  constexpr int KS_O = CH / 32, NFR_O = 9 * KS_O, KS_U = KC / 32, NFR_U = 9 * KS_U;
  constexpr int NH = NFR_U / 2;                                     // register half-set (the larger convolution's half)
  constexpr int PO = CH * 2 + 32, PU = KC * 2 + 32;                 // tile pitches in bytes (2 mod 4 16-byte units)
  constexpr int NFc = KC / 16, RG = 8 / NFc, RT = 4 / RG;           // both data gradients produce KC columns
  constexpr int CV = CH / 8;                                        // 16-byte chunks per CH-wide row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* do_t = smem;                                       // [65][PO]   d o     (A operand of phase B)
  unsigned char* dur_t = do_t + 65 * PO;                            // [65][PU]   d ur    (A operand of phase D)
  T* dxhr = reinterpret_cast<T*>(dur_t + 65 * PU);                  // [64][KC]   result of phase B: [d x | d hr]
  T* dxhx = dxhr + 64 * KC;                                         // [64][CH]   x-half of phase D's result
  T* dhn = dxhx + 64 * CH;                                          // [L][2][64][CH]  d h2 and the h-half of phase D's result, per cell
  float* dx0 = reinterpret_cast<float*>(dhn + (long)L * 2 * 64 * CH);   // [64][CH] gradient of the constant input
  const int tid = threadIdx.x;
  const long row0 = (long)blockIdx.x * 64;
  for (int i = tid; i < (65 * PO + 65 * PU) / 16; i += 512) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
  for (int i = tid; i < 64 * CH; i += 512) dx0[i] = 0.f;

  frag_t wa[NH], wb[NH];
  auto load_set = [&](frag_t* w, const T* base, int first, int n) {
#pragma unroll
    for (int q = 0; q < NH; ++q)
      if (q < n) w[q] = *reinterpret_cast<const frag_t*>(base + (long)(first + q) * 512);
  };
  auto tap_row = [&](const unsigned char* tile, int pitch, int p, int tap) {
    const int ty = tap / 3, tx = tap - 3 * ty;
    const int yy = (p >> 3) + ty - 1, xx = (p & 7) + tx - 1;
    return ((unsigned)yy < 8u && (unsigned)xx < 8u) ? tile + (yy * 8 + xx) * pitch : tile + 64 * pitch;
  };
  {
    const int lane = tid & 63, wave = tid >> 6;
    load_set(wa, P.w_o[L - 1] + (long)(wave % NFc) * NFR_O * 512 + lane * 8, 0, NFR_O / 2);
  }
  __syncthreads();
  bool pend_x0 = false;
#pragma unroll 1
  for (int t = Tn - 1; t >= 0; --t) {
#pragma unroll 1
    for (int l = L - 1; l >= 0; --l) {
      int tl = tid;
      asm volatile("" : "+v"(tl));
      const int lane = tl & 63, wave = tl >> 6, r = lane & 15, gq = lane >> 4;
      const int wn = wave % NFc, rg = wave / NFc;
      const long cell = ((long)l * Tn + t) * M + row0;
      const T* w_o = P.w_o[l] + (long)wn * NFR_O * 512 + lane * 8;
      const T* w_u = P.w_ur[l] + (long)wn * NFR_U * 512 + lane * 8;
      const int ln = l > 0 ? l - 1 : L - 1;
      const bool more = l > 0 || t > 0;
      const T* w_next = P.w_o[ln] + (long)wn * NFR_O * 512 + lane * 8;
      T* dh2 = dhn + ((long)l * 2 + 0) * 64 * CH;
      T* dxhh = dhn + ((long)l * 2 + 1) * 64 * CH;
      // one data-gradient convolution: K half A (requested earlier) under the request of half B, half B under the request of the next
      // convolution's half A
      auto conv = [&](const unsigned char* tile, auto pitch_c, auto ks_c, const T* wthis, const T* next_a, auto nnext_c, bool have_next, f32x4* acc) {
        constexpr int PT = decltype(pitch_c)::value, KS = decltype(ks_c)::value, NFR = 9 * KS, NA = NFR / 2, NNA = decltype(nnext_c)::value;
#pragma unroll
        for (int i = 0; i < RT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        load_set(wb, wthis, NA, NFR - NA);
#pragma unroll
        for (int q = 0; q < NFR; ++q) {
          const int tap = q / KS, ks = q - tap * KS;
          if (q == NA) {
            __builtin_amdgcn_sched_barrier(0);
            if (have_next) load_set(wa, next_a, 0, NNA);
          }
          frag_t fa[RT];
#pragma unroll
          for (int i = 0; i < RT; ++i)
            fa[i] = *reinterpret_cast<const frag_t*>(tap_row(tile, PT, (rg * RT + i) * 16 + r, tap) + gq * 16 + ks * 64);
#pragma unroll
          for (int i = 0; i < RT; ++i) mma64(fa[i], q < NA ? wa[q < NA ? q : 0] : wb[q < NA ? 0 : q - NA], acc[i]);
        }
      };
      // ---- A
      const bool act = tl < 64 * CV;
      const int pe = tl / CV, ce = (tl - pe * CV) * 8;                 // this thread's row and first channel in the element-wise phases
      bf16x8 hv, du8, dh18;
      if (act) {
        if (pend_x0) {                                                // cell 0 of the step just left: its two x-path gradients
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(dxhx + pe * CH + ce), b = *reinterpret_cast<const bf16x8*>(dxhr + pe * KC + ce);
#pragma unroll
          for (int q = 0; q < 8; ++q) dx0[pe * CH + ce + q] += ET<T>::to_f32(a[q]) + ET<T>::to_f32(b[q]);
        }
        float g[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) g[q] = 0.f;
        if (t + 1 < Tn) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(dh2 + pe * CH + ce), b = *reinterpret_cast<const bf16x8*>(dxhh + pe * CH + ce);
#pragma unroll
          for (int q = 0; q < 8; ++q) { g[q] += ET<T>::to_f32(a[q]); g[q] += ET<T>::to_f32(b[q]); }
        }
        if (l + 1 < L) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(dxhx + pe * CH + ce), b = *reinterpret_cast<const bf16x8*>(dxhr + pe * KC + ce);
#pragma unroll
          for (int q = 0; q < 8; ++q) { g[q] += ET<T>::to_f32(a[q]); g[q] += ET<T>::to_f32(b[q]); }
        } else {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(d_out + ((long)t * M + row0 + pe) * ldo + ce);
#pragma unroll
          for (int q = 0; q < 8; ++q) g[q] += ET<T>::to_f32(a[q]);
        }
        const bf16x8 ov = *reinterpret_cast<const bf16x8*>(O + (cell + pe) * CH + ce);
        const bf16x8 uv = *reinterpret_cast<const bf16x8*>(U + (cell + pe) * CH + ce);
        hv = *reinterpret_cast<const bf16x8*>(XH + (cell + pe) * KC + CH + ce);
        bf16x8 do8;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float uu = ET<T>::to_f32(uv[q]), th = tanhf(ET<T>::to_f32(ov[q])), hh = ET<T>::to_f32(hv[q]);
          do8[q] = ET<T>::from_f32(g[q] * uu * (1.f - th * th));
          du8[q] = ET<T>::from_f32(g[q] * (th - hh));
          dh18[q] = ET<T>::from_f32(g[q] * (1.f - uu));
        }
        *reinterpret_cast<bf16x8*>(do_t + pe * PO + ce * 2) = do8;
        *reinterpret_cast<bf16x8*>(DO + (cell + pe) * CH + ce) = do8;
      }
      pend_x0 = false;
      __syncthreads();
      // ---- B: [d x | d hr] = conv_o data gradient
      {
        f32x4 acc[RT];
        conv(do_t, std::integral_constant<int, PO>(), std::integral_constant<int, KS_O>(), w_o, w_u, std::integral_constant<int, NFR_U / 2>(), true, acc);
        const int n = wn * 16 + 4 * gq;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const int p = (rg * RT + i) * 16 + r;
          pack_t v;
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = ET<T>::from_f32(acc[i][q]);
          *reinterpret_cast<pack_t*>(dxhr + p * KC + n) = v;
        }
      }
      __syncthreads();
      // ---- C
      if (act) {
        const bf16x8 uru = *reinterpret_cast<const bf16x8*>(UR + (cell + pe) * KC + ce);
        const bf16x8 urr = *reinterpret_cast<const bf16x8*>(UR + (cell + pe) * KC + CH + ce);
        const bf16x8 ghr8 = *reinterpret_cast<const bf16x8*>(dxhr + pe * KC + CH + ce);
        bf16x8 a8, b8, h28;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float uu = act_apply(IPOKE_ACT_SIGMOID, ET<T>::to_f32(uru[q]));
          a8[q] = ET<T>::from_f32(ET<T>::to_f32(du8[q]) * uu * (1.f - uu));
          const float rr = act_apply(IPOKE_ACT_SIGMOID, ET<T>::to_f32(urr[q])), ghr = ET<T>::to_f32(ghr8[q]);
          b8[q] = ET<T>::from_f32(ghr * ET<T>::to_f32(hv[q]) * rr * (1.f - rr));
          h28[q] = ET<T>::from_f32(ET<T>::to_f32(dh18[q]) + ghr * rr);
        }
        *reinterpret_cast<bf16x8*>(dur_t + pe * PU + ce * 2) = a8;
        *reinterpret_cast<bf16x8*>(dur_t + pe * PU + (CH + ce) * 2) = b8;
        *reinterpret_cast<bf16x8*>(DUR + (cell + pe) * KC + ce) = a8;
        *reinterpret_cast<bf16x8*>(DUR + (cell + pe) * KC + CH + ce) = b8;
        *reinterpret_cast<bf16x8*>(dh2 + pe * CH + ce) = h28;
      }
      __syncthreads();
      // ---- D: [d x | d h] = conv_ur data gradient
      {
        f32x4 acc[RT];
        conv(dur_t, std::integral_constant<int, PU>(), std::integral_constant<int, KS_U>(), w_u, w_next, std::integral_constant<int, NFR_O / 2>(), more, acc);
        const int n = wn * 16 + 4 * gq;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          const int p = (rg * RT + i) * 16 + r;
          pack_t v;
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = ET<T>::from_f32(acc[i][q]);
          if (n < CH) *reinterpret_cast<pack_t*>(dxhx + p * CH + n) = v;
          else *reinterpret_cast<pack_t*>(dxhh + p * CH + n - CH) = v;
        }
      }
      if (l == 0) pend_x0 = true;
      __syncthreads();
    }
  }
  // the last (cell 0, step 0) contribution to d x0, then both results
  for (int i = tid; i < 64 * CH; i += 512) {
    const int p = i / CH, c = i - p * CH;
    const float v = dx0[i] + (ET<T>::to_f32(dxhx[i]) + ET<T>::to_f32(dxhr[p * KC + c]));
    d_x0[(row0 + p) * CH + c] = v;
    float h = 0.f;
    for (int l = 0; l < L; ++l) {
      const float s2 = ET<T>::to_f32(dhn[((long)l * 2 + 0) * 64 * CH + i]) + ET<T>::to_f32(dhn[((long)l * 2 + 1) * 64 * CH + i]);
      h = l > 0 ? h + s2 : s2;
    }
    d_h0[(row0 + p) * CH + c] = h;
  }
}

}  // namespace ipoke

using namespace ipoke;

namespace {

struct GruPlan {
  int B, T, L, Cx, Ch, H, W, S, Kc, N2p, Chp, esz, e16;
  long M;
  // byte offsets into the workspace
  long XH, XHR, UR, U, O, DO, DUR, DXH, DXHR, DH2, DU, DH1, WOP, SLAB, CSW, bytes;
  long wop_ur, wop_urT, wop_o, wop_oT, wop_layer;      // operand bytes
};

long take(long& cur, long bytes) { const long o = cur; cur = (cur + bytes + 255) / 256 * 256; return o; }

int gru_plan(GruPlan& p, const ipoke_gru_desc* d, int dtype) {
  IPK_REQUIRE(d && d->B >= 1 && d->T >= 1 && d->L >= 1 && d->L <= 16 && d->H >= 1 && d->W >= 1, "bad geometry");
  IPK_REQUIRE(dtype == IPOKE_F32 || dtype == IPOKE_BF16, "bad dtype");
  p.esz = dtype == IPOKE_BF16 ? 2 : 4; p.e16 = 16 / p.esz;
  IPK_REQUIRE(d->Cx % p.e16 == 0 && d->Ch % p.e16 == 0 && d->Cx >= p.e16 && d->Ch >= p.e16, "channel counts must be multiples of 16 bytes");
  IPK_REQUIRE(d->L == 1 || d->Cx == d->Ch, "stacked cells take the hidden state of the cell below: Cx == Ch");
  IPK_REQUIRE(ilog2_exact(d->H) >= 0 && ilog2_exact(d->W) >= 0, "map extents must be powers of two");
  p.B = d->B; p.T = d->T; p.L = d->L; p.Cx = d->Cx; p.Ch = d->Ch; p.H = d->H; p.W = d->W; p.S = d->H * d->W;
  p.M = (long)d->B * p.S;
  p.Kc = d->Cx + d->Ch; p.N2p = round_up(2 * d->Ch, p.e16); p.Chp = round_up(d->Ch, p.e16);
  const long cells = (long)p.L * p.T, E = p.esz;
  long cur = 0;
  p.XH = take(cur, cells * p.M * p.Kc * E); p.XHR = take(cur, cells * p.M * p.Kc * E);
  p.UR = take(cur, cells * p.M * p.N2p * E); p.U = take(cur, cells * p.M * p.Ch * E); p.O = take(cur, cells * p.M * p.Chp * E);
  p.DO = take(cur, cells * p.M * p.Chp * E); p.DUR = take(cur, cells * p.M * p.N2p * E);
  p.DXH = take(cur, (long)p.L * p.M * p.Kc * E); p.DXHR = take(cur, (long)p.L * p.M * p.Kc * E); p.DH2 = take(cur, (long)p.L * p.M * p.Ch * E);
  p.DU = take(cur, p.M * p.Ch * E); p.DH1 = take(cur, p.M * p.Ch * E);
  p.wop_ur = (long)2 * p.Ch * 9 * p.Kc * E; p.wop_urT = (long)p.Kc * 9 * p.N2p * E;
  p.wop_o = (long)p.Ch * 9 * p.Kc * E; p.wop_oT = (long)p.Kc * 9 * p.Chp * E;
  p.wop_layer = ((p.wop_ur + 255) / 256 + (p.wop_urT + 255) / 256 + (p.wop_o + 255) / 256 + (p.wop_oT + 255) / 256) * 256;
  p.WOP = take(cur, p.wop_layer * p.L);
  p.SLAB = take(cur, (long)64 * 2 * p.Ch * p.Kc * 9 * 4);           // split-M slabs of one weight gradient (<= 64 splits)
  p.CSW = take(cur, (ipoke_colsum_workspace_floats(p.M * p.T, 2 * p.Ch) + 16) * 4);
  p.bytes = cur;
  return IPOKE_OK;
}

template <typename K, typename... A>
int launch1d(K kern, long total, hipStream_t s, A... a) {
  long g = (total + 255) / 256; if (g < 1) g = 1; if (g > 4096) g = 4096;
  hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(256), 0, s, a...);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

struct GruCtx {
  GruPlan p; int dtype; unsigned char* ws; hipStream_t s;
  unsigned char* cell(long base, int l, int t, long row_elems) const { return ws + base + ((long)l * p.T + t) * p.M * row_elems * p.esz; }
  unsigned char* lay(long base, int l, long row_elems) const { return ws + base + (long)l * p.M * row_elems * p.esz; }
  unsigned char* wop(int l, int which) const {        // 0 ur, 1 ur^T, 2 o, 3 o^T
    long off = 0;
    if (which > 0) off += (p.wop_ur + 255) / 256 * 256;
    if (which > 1) off += (p.wop_urT + 255) / 256 * 256;
    if (which > 2) off += (p.wop_o + 255) / 256 * 256;
    return ws + p.WOP + (long)l * p.wop_layer + off;
  }
};

// 3 x 3 / padding 1 convolution on the [B][H][W] map: rows of A are [M][lda] dense, `transposed`: the data gradient (operand given transposed)
int gru_conv(const GruCtx& c, const void* A, int lda, int Kc, const void* Wop, int Nout, const float* bias, void* C, int ldc, int transposed) {
  ipoke_conv_desc d; std::memset(&d, 0, sizeof(d));
  const GruPlan& p = c.p;
  d.NB = p.B; d.Di = 1; d.Hi = p.H; d.Wi = p.W; d.Do = 1; d.Ho = p.H; d.Wo = p.W;
  d.kd = 1; d.kh = 3; d.kw = 3; d.sd = d.sh = d.sw = 1; d.pd = 0; d.ph = 1; d.pw = 1; d.transposed = transposed;
  d.A = A; d.a_f32 = 0; d.a_sn = (long)p.S * lda; d.a_sd = 0; d.a_sh = (long)p.W * lda; d.a_sw = lda; d.a_sc = 1; d.Kc_real = Kc; d.Kc = Kc;
  d.W = Wop; d.ldw = 9 * Kc; d.Nout = Nout; d.bias = bias; d.act = IPOKE_ACT_NONE;
  d.C = C; d.ldc = ldc; d.c_cstride = 1; d.splitk = 1;
  return ipoke_conv_forward(&d, c.dtype, reinterpret_cast<void*>(c.s));
}

}  // namespace

static std::atomic<int> g_gru_fused{-1};
// which operand layout the last forward pass left in a workspace (host-side: the backward pass must not read device memory to find out)
static std::mutex g_gru_ws_mutex;
static std::unordered_map<const void*, std::pair<int, long>> g_gru_ws_layout;   // workspace -> (1: fragment-tiled operands / 0: row-major, sequence number)
static long g_gru_ws_seq = 0;
static void gru_note_layout(const void* ws, int tiled) {
  std::lock_guard<std::mutex> g(g_gru_ws_mutex);
  g_gru_ws_layout[ws] = {tiled, ++g_gru_ws_seq};
  if (g_gru_ws_layout.size() > 4096) {         // workspaces come and go with the caller's allocator: forget the oldest half
    for (auto it = g_gru_ws_layout.begin(); it != g_gru_ws_layout.end();)
      it = it->second.second + 2048 < g_gru_ws_seq ? g_gru_ws_layout.erase(it) : std::next(it);
  }
}
static int gru_layout_of(const void* ws) {       // -1: no forward pass on record for this workspace
  std::lock_guard<std::mutex> g(g_gru_ws_mutex);
  auto it = g_gru_ws_layout.find(ws);
  return it == g_gru_ws_layout.end() ? -1 : it->second.first;
}
// dynamic-LDS limit of a kernel instantiation, raised once (grows monotonically)
template <typename K> static int gru_ensure_lds(K kern, size_t bytes, size_t& granted) {
  if (bytes > granted) {
    IPK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    granted = bytes;
  }
  return IPOKE_OK;
}
/* Test hook: 0 = the launch-per-phase forward unroll, 1 = the fused kernel where it applies, < 0 = re-read IPOKE_GRU_FUSED at the next call. */
extern "C" int ipoke_gru_set_fused(int mode) { g_gru_fused.store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_relaxed); return IPOKE_OK; }

extern "C" int64_t ipoke_gru_workspace_bytes(const ipoke_gru_desc* d, int dtype) {
  GruPlan p;
  if (gru_plan(p, d, dtype) != IPOKE_OK) return -1;
  return p.bytes;
}

/* weights: 4 device pointers per cell -- w_ur [2Ch][Cx+Ch][3][3] (update gate rows first, as the reference's two gate convolutions stacked),
 * b_ur [2Ch], w_o [Ch][Cx+Ch][3][3], b_o [Ch], fp32 in PyTorch layout.  x0 [M][ldx] is the constant input of cell 0, h0 [M][ldh] the initial
 * hidden state of EVERY cell (first_stage_motion_model.py:503-506).  out [T][M][ldo]: the last cell's hidden state after every step. */
extern "C" int ipoke_gru_unroll_forward(const ipoke_gru_desc* d, const void* x0, int ldx, const void* h0, int ldh, const float* const* weights,
                                        void* workspace, void* out, int ldo, int dtype, void* stream) {
  GruPlan p; int rc = gru_plan(p, d, dtype); if (rc) return rc;
  IPK_REQUIRE(x0 && h0 && weights && workspace && out && ldx >= p.Cx && ldh >= p.Ch && ldo >= p.Ch, "bad arguments");
  GruCtx c{p, dtype, reinterpret_cast<unsigned char*>(workspace), reinterpret_cast<hipStream_t>(stream)};
  // one workgroup per sample runs the whole recurrence (gru_fused_fwd_kernel) where it applies; IPOKE_GRU_FUSED=0: the launch-per-phase form
  int fm = g_gru_fused.load(std::memory_order_relaxed);
  if (fm < 0) { fm = (getenv("IPOKE_GRU_FUSED") && atoi(getenv("IPOKE_GRU_FUSED")) == 0) ? 0 : 1; g_gru_fused.store(fm, std::memory_order_relaxed); }
  const bool fused = fm != 0 && dtype == IPOKE_BF16 && p.H == 8 && p.W == 8 && p.Cx == p.Ch && (p.Ch == 32 || p.Ch == 64) && p.L <= 16 &&
                     ldx % 8 == 0 && ldh % 8 == 0 && ldo % 4 == 0 && ((reinterpret_cast<uintptr_t>(x0) | reinterpret_cast<uintptr_t>(h0)) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(out) & 7) == 0;
  bool fused_ok = fused;
  for (int l = 0; fused_ok && l < p.L; ++l)
    fused_ok = weights[4 * l] && weights[4 * l + 1] && weights[4 * l + 2] && weights[4 * l + 3] &&
               ((reinterpret_cast<uintptr_t>(weights[4 * l + 1]) | reinterpret_cast<uintptr_t>(weights[4 * l + 3])) & 3) == 0;
  // matrix-core operands of the four convolutions of every cell (forward and data-gradient forms), once per pass; the fused kernel
  // takes its two forward operands fragment-tiled (gru_tile_operand_kernel) in the same workspace slots
  for (int l = 0; l < p.L; ++l) {
    const float* w_ur = weights[4 * l], *w_o = weights[4 * l + 2];
    IPK_REQUIRE(w_ur && weights[4 * l + 1] && w_o && weights[4 * l + 3], "null weight");
    if (fused_ok) {
      rc = launch1d(gru_tile_operand_kernel, (long)2 * p.Ch * 9 * p.Kc, c.s, w_ur, reinterpret_cast<bf16_t*>(c.wop(l, 0)), 2 * p.Ch, p.Kc); if (rc) return rc;
      rc = launch1d(gru_tile_operand_kernel, (long)p.Ch * 9 * p.Kc, c.s, w_o, reinterpret_cast<bf16_t*>(c.wop(l, 2)), p.Ch, p.Kc); if (rc) return rc;
    } else {
      rc = ipoke_conv_weight_operand(w_ur, 2 * p.Ch, p.Kc, 9, 0, nullptr, c.wop(l, 0), p.Kc, dtype, stream); if (rc) return rc;
      rc = ipoke_conv_weight_operand(w_o, p.Ch, p.Kc, 9, 0, nullptr, c.wop(l, 2), p.Kc, dtype, stream); if (rc) return rc;
    }
    if (fused_ok) {
      rc = launch1d(gru_tile_operand_t_kernel, (long)p.Kc * 9 * 2 * p.Ch, c.s, w_ur, reinterpret_cast<bf16_t*>(c.wop(l, 1)), 2 * p.Ch, p.Kc); if (rc) return rc;
      rc = launch1d(gru_tile_operand_t_kernel, (long)p.Kc * 9 * p.Ch, c.s, w_o, reinterpret_cast<bf16_t*>(c.wop(l, 3)), p.Ch, p.Kc); if (rc) return rc;
    } else {
      rc = ipoke_conv_weight_operand(w_ur, p.Kc, 2 * p.Ch, 9, 1, nullptr, c.wop(l, 1), p.N2p, dtype, stream); if (rc) return rc;
      rc = ipoke_conv_weight_operand(w_o, p.Kc, p.Ch, 9, 1, nullptr, c.wop(l, 3), p.Chp, dtype, stream); if (rc) return rc;
    }
  }
  gru_note_layout(workspace, fused_ok ? 1 : 0);
  if (fused_ok) {
    GruFusedPtrs P; std::memset(&P, 0, sizeof(P));
    for (int l = 0; l < p.L; ++l) {
      P.w_ur[l] = reinterpret_cast<const bf16_t*>(c.wop(l, 0)); P.w_o[l] = reinterpret_cast<const bf16_t*>(c.wop(l, 2));
      P.b_ur[l] = weights[4 * l + 1]; P.b_o[l] = weights[4 * l + 3];
    }
    {
      const int PK = 2 * p.Kc + 32;
      const size_t lds = (size_t)4 * 65 * PK + ((size_t)p.L * 64 * p.Ch + 2 * 64 * p.Ch) * 2 + (size_t)p.L * 3 * p.Ch * 4;
      bf16_t* XH = reinterpret_cast<bf16_t*>(c.ws + p.XH); bf16_t* XHR = reinterpret_cast<bf16_t*>(c.ws + p.XHR);
      bf16_t* URp = reinterpret_cast<bf16_t*>(c.ws + p.UR); bf16_t* Up = reinterpret_cast<bf16_t*>(c.ws + p.U); bf16_t* Op = reinterpret_cast<bf16_t*>(c.ws + p.O);
      if (p.Ch == 64) {
        static size_t granted64 = 0; rc = gru_ensure_lds(gru_fused_fwd_kernel<64>, lds, granted64); if (rc) return rc;
        hipLaunchKernelGGL(gru_fused_fwd_kernel<64>, dim3(p.B), dim3(512), lds, c.s, (const bf16_t*)x0, ldx, (const bf16_t*)h0, ldh, P, XH, XHR, URp, Up, Op,
                           (bf16_t*)out, ldo, p.M, p.T, p.L);
      } else {
        static size_t granted32 = 0; rc = gru_ensure_lds(gru_fused_fwd_kernel<32>, lds, granted32); if (rc) return rc;
        hipLaunchKernelGGL(gru_fused_fwd_kernel<32>, dim3(p.B), dim3(512), lds, c.s, (const bf16_t*)x0, ldx, (const bf16_t*)h0, ldh, P, XH, XHR, URp, Up, Op,
                           (bf16_t*)out, ldo, p.M, p.T, p.L);
      }
      IPK_LAUNCH_CHECK();
      return IPOKE_OK;
    }
  }
  const long fill = (long)p.T * p.M * p.Cx + (long)p.L * p.M * p.Ch;
  if (dtype == IPOKE_BF16) rc = launch1d(gru_fill_kernel<bf16_t>, fill, c.s, (const bf16_t*)x0, ldx, (const bf16_t*)h0, ldh, (bf16_t*)(c.ws + p.XH), (bf16_t*)(c.ws + p.XHR), p.M, p.Cx, p.Ch, p.Kc, p.T, p.L);
  else rc = launch1d(gru_fill_kernel<float>, fill, c.s, (const float*)x0, ldx, (const float*)h0, ldh, (float*)(c.ws + p.XH), (float*)(c.ws + p.XHR), p.M, p.Cx, p.Ch, p.Kc, p.T, p.L);
  if (rc) return rc;
  const long E = p.esz;
  for (int t = 0; t < p.T; ++t) {
    for (int l = 0; l < p.L; ++l) {
      unsigned char* xh = c.cell(p.XH, l, t, p.Kc); unsigned char* xhr = c.cell(p.XHR, l, t, p.Kc);
      unsigned char* ur = c.cell(p.UR, l, t, p.N2p); unsigned char* u = c.cell(p.U, l, t, p.Ch); unsigned char* o = c.cell(p.O, l, t, p.Chp);
      const unsigned char* h = xh + (long)p.Cx * E;
      rc = gru_conv(c, xh, p.Kc, p.Kc, c.wop(l, 0), 2 * p.Ch, weights[4 * l + 1], ur, p.N2p, 0); if (rc) return rc;
      GruDst dst; std::memset(&dst, 0, sizeof(dst));
      int nd = 0;
      if (t + 1 < p.T) { dst.p[nd] = c.cell(p.XH, l, t + 1, p.Kc) + (long)p.Cx * E; dst.ld[nd++] = p.Kc; }       // this cell, next step: h-half
      if (l + 1 < p.L) {                                                                                         // next cell, this step: both x-halves
        dst.p[nd] = c.cell(p.XH, l + 1, t, p.Kc); dst.ld[nd++] = p.Kc;
        dst.p[nd] = c.cell(p.XHR, l + 1, t, p.Kc); dst.ld[nd++] = p.Kc;
      } else {
        dst.p[nd] = reinterpret_cast<unsigned char*>(out) + (long)t * p.M * ldo * E; dst.ld[nd++] = ldo;
      }
      if (dtype == IPOKE_BF16) {
        rc = launch1d(gru_gates_fwd_kernel<bf16_t>, p.M * p.Ch, c.s, (const bf16_t*)ur, p.N2p, (const bf16_t*)h, p.Kc, (bf16_t*)(xhr + (long)p.Cx * E), p.Kc, (bf16_t*)u, p.M, p.Ch);
        if (rc) return rc;
        rc = gru_conv(c, xhr, p.Kc, p.Kc, c.wop(l, 2), p.Ch, weights[4 * l + 3], o, p.Chp, 0); if (rc) return rc;
        rc = launch1d(gru_update_fwd_kernel<bf16_t>, p.M * p.Ch, c.s, (const bf16_t*)o, p.Chp, (const bf16_t*)u, (const bf16_t*)h, p.Kc, dst, p.M, p.Ch);
      } else {
        rc = launch1d(gru_gates_fwd_kernel<float>, p.M * p.Ch, c.s, (const float*)ur, p.N2p, (const float*)h, p.Kc, (float*)(xhr + (long)p.Cx * E), p.Kc, (float*)u, p.M, p.Ch);
        if (rc) return rc;
        rc = gru_conv(c, xhr, p.Kc, p.Kc, c.wop(l, 2), p.Ch, weights[4 * l + 3], o, p.Chp, 0); if (rc) return rc;
        rc = launch1d(gru_update_fwd_kernel<float>, p.M * p.Ch, c.s, (const float*)o, p.Chp, (const float*)u, (const float*)h, p.Kc, dst, p.M, p.Ch);
      }
      if (rc) return rc;
    }
  }
  return IPOKE_OK;
}

/* Backward of ipoke_gru_unroll_forward on the SAME workspace (every operand of the forward pass is still there).  d_out [T][M][ldo]: the
 * gradient of the output sequence.  Written: dweights[4 l ..] (fp32, the layouts of `weights`), d_x0 [M][Cx] and d_h0 [M][Ch] (fp32; d_h0
 * is the SUM over the L cells, which all start from the same state). */
extern "C" int ipoke_gru_unroll_backward(const ipoke_gru_desc* d, const void* d_out, int ldo, void* workspace, float* const* dweights,
                                         float* d_x0, float* d_h0, int dtype, void* stream) {
  GruPlan p; int rc = gru_plan(p, d, dtype); if (rc) return rc;
  IPK_REQUIRE(d_out && workspace && dweights && d_x0 && d_h0 && ldo >= p.Ch, "bad arguments");
  GruCtx c{p, dtype, reinterpret_cast<unsigned char*>(workspace), reinterpret_cast<hipStream_t>(stream)};
  const long E = p.esz;
  // a workspace the library has no forward pass on record for (never run, or evicted after > 2048 later forward passes) must not be
  // read with a guessed operand layout: tiled operands read as row-major give wrong gradients without any error
  const int ws_layout = gru_layout_of(workspace);
  IPK_REQUIRE(ws_layout >= 0, "ipoke_gru_unroll_backward: no forward pass on record for this workspace (run ipoke_gru_unroll_forward on it first; "
                              "the record of a workspace is kept for the 2048 most recent forward passes)");
  const bool fused_bwd = ws_layout == 1 && dtype == IPOKE_BF16 && ldo % 8 == 0 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0;
  if (ws_layout == 1) IPK_REQUIRE(fused_bwd, "the forward pass left fragment-tiled operands: d_out must be 16-byte aligned rows of a multiple of 8 elements");
  if (fused_bwd) {
    GruFusedPtrs P; std::memset(&P, 0, sizeof(P));
    for (int l = 0; l < p.L; ++l) { P.w_ur[l] = reinterpret_cast<const bf16_t*>(c.wop(l, 1)); P.w_o[l] = reinterpret_cast<const bf16_t*>(c.wop(l, 3)); }
    const int PO = 2 * p.Ch + 32, PU = 2 * p.Kc + 32;
    const size_t lds = (size_t)65 * PO + (size_t)65 * PU + ((size_t)64 * p.Kc + 64 * p.Ch + (size_t)p.L * 2 * 64 * p.Ch) * 2 + (size_t)64 * p.Ch * 4;
    const bf16_t* XH = reinterpret_cast<const bf16_t*>(c.ws + p.XH); const bf16_t* URp = reinterpret_cast<const bf16_t*>(c.ws + p.UR);
    const bf16_t* Up = reinterpret_cast<const bf16_t*>(c.ws + p.U); const bf16_t* Op = reinterpret_cast<const bf16_t*>(c.ws + p.O);
    bf16_t* DOp = reinterpret_cast<bf16_t*>(c.ws + p.DO); bf16_t* DURp = reinterpret_cast<bf16_t*>(c.ws + p.DUR);
    if (p.Ch == 64) {
      static size_t granted64 = 0; rc = gru_ensure_lds(gru_fused_bwd_kernel<64>, lds, granted64); if (rc) return rc;
      hipLaunchKernelGGL(gru_fused_bwd_kernel<64>, dim3(p.B), dim3(512), lds, c.s, (const bf16_t*)d_out, ldo, P, XH, URp, Up, Op, DOp, DURp, d_x0, d_h0, p.M, p.T, p.L);
    } else {
      static size_t granted32 = 0; rc = gru_ensure_lds(gru_fused_bwd_kernel<32>, lds, granted32); if (rc) return rc;
      hipLaunchKernelGGL(gru_fused_bwd_kernel<32>, dim3(p.B), dim3(512), lds, c.s, (const bf16_t*)d_out, ldo, P, XH, URp, Up, Op, DOp, DURp, d_x0, d_h0, p.M, p.T, p.L);
    }
    IPK_LAUNCH_CHECK();
  }
  unsigned char* du = c.ws + p.DU; unsigned char* dh1 = c.ws + p.DH1;
  for (int t = fused_bwd ? -1 : p.T - 1; t >= 0; --t) {
    for (int l = p.L - 1; l >= 0; --l) {
      unsigned char* xh = c.cell(p.XH, l, t, p.Kc);
      unsigned char* ur = c.cell(p.UR, l, t, p.N2p); unsigned char* u = c.cell(p.U, l, t, p.Ch); unsigned char* o = c.cell(p.O, l, t, p.Chp);
      unsigned char* d_o = c.cell(p.DO, l, t, p.Chp); unsigned char* d_ur = c.cell(p.DUR, l, t, p.N2p);
      unsigned char* dxh = c.lay(p.DXH, l, p.Kc); unsigned char* dxhr = c.lay(p.DXHR, l, p.Kc); unsigned char* dh2 = c.lay(p.DH2, l, p.Ch);
      const unsigned char* h = xh + (long)p.Cx * E;
      GruSrc src; std::memset(&src, 0, sizeof(src));
      int ns = 0;
      if (t + 1 < p.T) {                           // the same cell's next step (its gradient buffers still hold step t + 1)
        src.p[ns] = dh2; src.ld[ns++] = p.Ch;
        src.p[ns] = dxh + (long)p.Cx * E; src.ld[ns++] = p.Kc;
      }
      if (l + 1 < p.L) {                           // the next cell of this step: x-halves of its two data gradients
        src.p[ns] = c.lay(p.DXH, l + 1, p.Kc); src.ld[ns++] = p.Kc;
        src.p[ns] = c.lay(p.DXHR, l + 1, p.Kc); src.ld[ns++] = p.Kc;
      } else {
        src.p[ns] = reinterpret_cast<const unsigned char*>(d_out) + (long)t * p.M * ldo * E; src.ld[ns++] = ldo;
      }
      if (dtype == IPOKE_BF16)
        rc = launch1d(gru_update_bwd_kernel<bf16_t>, p.M * p.Chp, c.s, src, (const bf16_t*)o, p.Chp, (const bf16_t*)u, (const bf16_t*)h, p.Kc, (bf16_t*)d_o, p.Chp, (bf16_t*)du, (bf16_t*)dh1, p.M, p.Ch);
      else
        rc = launch1d(gru_update_bwd_kernel<float>, p.M * p.Chp, c.s, src, (const float*)o, p.Chp, (const float*)u, (const float*)h, p.Kc, (float*)d_o, p.Chp, (float*)du, (float*)dh1, p.M, p.Ch);
      if (rc) return rc;
      rc = gru_conv(c, d_o, p.Chp, p.Chp, c.wop(l, 3), p.Kc, nullptr, dxhr, p.Kc, 1); if (rc) return rc;
      if (dtype == IPOKE_BF16)
        rc = launch1d(gru_gates_bwd_kernel<bf16_t>, p.M * p.N2p, c.s, (const bf16_t*)ur, p.N2p, (const bf16_t*)h, p.Kc, (const bf16_t*)(dxhr + (long)p.Cx * E), p.Kc, (const bf16_t*)du, (const bf16_t*)dh1, (bf16_t*)d_ur, p.N2p, (bf16_t*)dh2, p.M, p.Ch);
      else
        rc = launch1d(gru_gates_bwd_kernel<float>, p.M * p.N2p, c.s, (const float*)ur, p.N2p, (const float*)h, p.Kc, (const float*)(dxhr + (long)p.Cx * E), p.Kc, (const float*)du, (const float*)dh1, (float*)d_ur, p.N2p, (float*)dh2, p.M, p.Ch);
      if (rc) return rc;
      rc = gru_conv(c, d_ur, p.N2p, p.N2p, c.wop(l, 1), p.Kc, nullptr, dxh, p.Kc, 1); if (rc) return rc;
      if (l == 0) {                                // the constant input of cell 0 collects the x-path gradients of every step
        if (dtype == IPOKE_BF16) rc = launch1d(gru_add2_kernel<bf16_t>, p.M * p.Cx, c.s, (const bf16_t*)dxh, p.Kc, (const bf16_t*)dxhr, p.Kc, d_x0, p.Cx, p.M, p.Cx, t + 1 < p.T ? 1 : 0);
        else rc = launch1d(gru_add2_kernel<float>, p.M * p.Cx, c.s, (const float*)dxh, p.Kc, (const float*)dxhr, p.Kc, d_x0, p.Cx, p.M, p.Cx, t + 1 < p.T ? 1 : 0);
        if (rc) return rc;
      }
    }
  }
  // d h0 = sum over the cells of (d h2 + h-half of the first convolution's data gradient) after step 0
  for (int l = 0; l < (fused_bwd ? 0 : p.L); ++l) {
    unsigned char* dxh = c.lay(p.DXH, l, p.Kc); unsigned char* dh2 = c.lay(p.DH2, l, p.Ch);
    if (dtype == IPOKE_BF16) rc = launch1d(gru_add2_kernel<bf16_t>, p.M * p.Ch, c.s, (const bf16_t*)dh2, p.Ch, (const bf16_t*)(dxh + (long)p.Cx * E), p.Kc, d_h0, p.Ch, p.M, p.Ch, l > 0 ? 1 : 0);
    else rc = launch1d(gru_add2_kernel<float>, p.M * p.Ch, c.s, (const float*)dh2, p.Ch, (const float*)(dxh + (long)p.Cx * E), p.Kc, d_h0, p.Ch, p.M, p.Ch, l > 0 ? 1 : 0);
    if (rc) return rc;
  }
  // weight and bias gradients: the steps share the weights -- one GEMM / column sum per convolution over the rows of all T steps
  float* slabs = reinterpret_cast<float*>(c.ws + p.SLAB);
  float* csw = reinterpret_cast<float*>(c.ws + p.CSW);
  for (int l = 0; l < p.L; ++l) {
    for (int which = 0; which < 2; ++which) {      // 0: update | reset gates (operand cat[x, h]); 1: candidate (operand cat[x, h * r])
      const int Nout = which == 0 ? 2 * p.Ch : p.Ch, ldy = which == 0 ? p.N2p : p.Chp;
      float* dW = dweights[4 * l + 2 * which]; float* dB = dweights[4 * l + 2 * which + 1];
      IPK_REQUIRE(dW && dB, "null gradient tensor");
      ipoke_wgrad_desc w; std::memset(&w, 0, sizeof(w));
      w.NB = p.T * p.B; w.Di = 1; w.Hi = p.H; w.Wi = p.W; w.Do = 1; w.Ho = p.H; w.Wo = p.W; w.kd = 1; w.kh = w.kw = 3; w.sd = w.sh = w.sw = 1; w.ph = w.pw = 1;
      w.A = c.cell(which == 0 ? p.XH : p.XHR, l, 0, p.Kc); w.a_f32 = 0;
      w.a_sn = (long)p.S * p.Kc; w.a_sd = 0; w.a_sh = (long)p.W * p.Kc; w.a_sw = p.Kc; w.a_sc = 1; w.Kc_real = p.Kc; w.Kc = p.Kc; w.Kc_store = p.Kc;
      w.dY = c.cell(which == 0 ? p.DUR : p.DO, l, 0, ldy); w.ldy = ldy; w.Nout = Nout;
      w.w_sn = (long)p.Kc * 9; w.w_sc = 9; w.w_st = 1;
      const long rows = (long)p.T * p.M, wsize = (long)Nout * p.Kc * 9;
      const int tiles = ceil_div(Nout, 128) * ceil_div(9 * p.Kc, 128);
      long splitm = rows / (8 * 16 * p.e16); if (splitm > 512 / tiles) splitm = 512 / tiles; if (splitm > 64) splitm = 64; if (splitm < 1) splitm = 1;
      if (splitm > 1) {
        w.splitm = (int)splitm; w.split_stride = wsize; w.dW = slabs;
        rc = ipoke_conv_wgrad(&w, dtype, stream); if (rc) return rc;
        rc = ipoke_reduce_rows(slabs, dW, (int)splitm, (int)wsize, stream); if (rc) return rc;
      } else {
        w.dW = dW;
        rc = ipoke_conv_wgrad(&w, dtype, stream); if (rc) return rc;
      }
      rc = ipoke_colsum(w.dY, ldy, rows, Nout, 0, dB, 0, csw, dtype, stream); if (rc) return rc;
    }
  }
  return IPOKE_OK;
}
