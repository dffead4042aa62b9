// Fused MaCowUnit kernels with a sample's 8x8 latent split by ROWS over S = 2 or 4 workgroups (reference
// models/modules/INN/macow2.py:925-995; masks / kernel offsets of the four masked convolutions: macow_utils.py:407-499).
//
// Why: the one-workgroup-per-sample kernels of mcf_unit.hip put a c2 batch on 20 of 256 CUs and run as a chain of per-workgroup
// latencies (4 layers x [shifted conv -> barrier -> 1x1 conv -> barrier -> coupling]).  Everything in a layer is local to a
// position EXCEPT the shifted convolution, which reads the two rows above (A), the two rows below (B) or one row on either side
// (C, D).  With the grid rows dealt to S workgroups, a workgroup needs per layer the 1-2 neighbouring rows of the layer's input
// from the workgroup above / below: <= 2 KB of bf16 state in the forward pass, <= 8 KB of bf16 hidden gradients in the backward
// pass.  (Splitting the hidden width instead would leave each workgroup a quarter of the weight stream but need two exchanges of
// a [64 x 2C] fp32 partial sum and of the whole state per layer.)  Every workgroup streams ALL weights of the unit (L2-resident
// after the first touch), each owns 64 / S rows of every contraction.
//
// Hand-off (MI355X_MICROARCH.md "handoff-1to1", cdna_hip_programming.md Guideline 16, form R2): the data IS the flag.  A halo is
// published as 8-byte granules {tag = 1, value = two bf16} with relaxed agent-scope (sc1, write-through) stores straight from the
// registers of the coupling epilogue -- no drain, no release fence, no flag -- and swept by the consumer with relaxed agent-scope
// loads (L1 bypassed) until every tag is set; the consumer then clears its granules, so that the scratch is all-zero again when
// the launch ends (no per-launch memset node; correct under hipGraph replay).  Every granule is written once and read by exactly
// one workgroup per launch; a layer's granules are distinct from every other layer's.  Placement-independent: nothing depends on
// which XCD a workgroup runs on or in which order the workgroups start -- a workgroup only ever waits for its two neighbours,
// whose block ids are adjacent, so a partly resident grid cannot deadlock as long as one sample's S workgroups fit on the chip.
// Spins are bounded (kSpinMax polls): a time-out is counted in word 0 of the scratch and the launch finishes with garbage
// instead of hanging the device.
//
// Results: identical arithmetic per row (the matrix-core tiles are the same 16 rows, K order unchanged), so outputs, saved
// activations and data gradients are BIT-IDENTICAL to the one-workgroup kernels; log-det and the bias / ActNorm parameter-gradient
// partials are summed per part (slot s of the log-det slot, row b * S + s of the partial-sum matrices).
#include "mcf_unit_dev.h"

namespace ipoke {

typedef __attribute__((address_space(1))) unsigned long long gu64;
static constexpr unsigned kSpinMax = 1u << 21;        // ~ seconds; a healthy hand-off takes a few polls
static constexpr int kXchgHeader = 32;                // granules (256 B): word 0 counts spin time-outs

// rows of its INPUT a masked convolution of geometry g reads above / below the output row (forward direction)
__device__ __forceinline__ int rows_above(const McfGeom& g) { return g.oy < 0 ? -g.oy : 0; }                       // A 2, B 0, C / D 1
__device__ __forceinline__ int rows_below(const McfGeom& g) { const int m = g.kh - 1 + g.oy; return m > 0 ? m : 0; }  // A 0, B 2, C / D 1

// granules of (sample b, layer k, consumer part s): [4 halo rows: 0, 1 above / 2, 3 below the consumer][8 columns][pw pairs]
__device__ __forceinline__ gu64* xg_region(const UnitParams& U, int b, int k, int S, int s) {
  return reinterpret_cast<gu64*>(reinterpret_cast<unsigned long long>(U.xchg)) + kXchgHeader + (((long)b * 4 + k) * S + s) * (long)U.xchg_stride;
}
__device__ __forceinline__ void xg_store(gu64* g, unsigned value) {
  __hip_atomic_store(g, (1ull << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sweep of the halo rows this part needs (nu rows above, nd rows below) into the LDS tile at their GLOBAL positions:
// value -> tile + (y * 8 + x) * pitch + 4 * cp.  NI = granules per thread (compile-time bound), pw = pairs per position.
// Two halves: halo_begin requests every granule once (a round trip to the memory side: ~1 500 cycles), the caller then computes
// whatever does not need the halo; halo_end re-requests what had not been published yet until every tag is set.
template <int NI> struct Halo { gu64* gp[NI]; unsigned* dst[NI]; unsigned long long v[NI]; };
template <int S, int NI>
__device__ __forceinline__ void halo_begin(Halo<NI>& h, gu64* reg, int s, int nu, int nd, int pw, int pws, unsigned char* tile, int pitch, int tl) {
  constexpr int RY = 8 / S;
  const int y0 = s * RY, per_row = 8 * pw, items = (nu + nd) * per_row;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = tl + i * kMcfThreads;
    const int hi = e / per_row, rem = e - hi * per_row, x = rem / pw, cp = rem - x * pw;
    const bool up = hi < nu;
    const int hrow = up ? 2 - nu + hi : 2 + hi - nu;
    const int y = up ? y0 - nu + hi : y0 + RY + hi - nu;
    h.gp[i] = e < items ? reg + (hrow * 8 + x) * pws + cp : nullptr;
    h.dst[i] = reinterpret_cast<unsigned*>(tile + (y * 8 + x) * pitch + cp * 4);
    h.v[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i)
    if (h.gp[i]) h.v[i] = __hip_atomic_load(h.gp[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int NI>
__device__ __forceinline__ void halo_end(const UnitParams& U, Halo<NI>& h) {
  unsigned spins = 0;
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < NI; ++i) ok = ok && (!h.gp[i] || (h.v[i] >> 32) == 1ull);
    if (ok) break;
    if (++spins >= kSpinMax) {
      atomicAdd(reinterpret_cast<unsigned*>(U.xchg), 1u);
      break;
    }
    __builtin_amdgcn_s_sleep(2);
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (h.gp[i] && (h.v[i] >> 32) != 1ull) h.v[i] = __hip_atomic_load(h.gp[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int i = 0; i < NI; ++i)
    if (h.gp[i]) {
      *h.dst[i] = (unsigned)h.v[i];
      __hip_atomic_store(h.gp[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // the scratch is all-zero again when the launch ends
    }
}

// ------------------------------------------------------------------------------------------------ forward
// hidden = ELU(A1 x W1^T) for the 16 positions [ptile, ptile + 16) -> a2t[r][0:H]; RELOAD: the weight fragments of a tap are
// re-requested for the NEXT layer (buffer rs_next; soff_bias = kOob behind the last layer: zeros, no traffic) as soon as the tap's
// matrix-core instructions have been issued -- the 36 KB per wave trickle out underneath the remaining taps instead of stalling
// all eight waves in a burst behind the contraction.
#ifndef IPOKE_UNIT_SPREAD
#define IPOKE_UNIT_SPREAD 1      // 1: the next layer's weight requests in batches of 4-8 per wave between the phases of this layer (0: in
#endif                           //    one burst inside the first contraction's last tile -- developer A/B)
// re-request the fragments of taps [LO, HI) of the shifted-conv weights / of K steps [LO, HI) of the 1x1 weights from another layer's operand
template <typename T, bool WIDE, int LO, int HI>
__device__ __forceinline__ void reload_w1(McfW<T>& w, rsrc_t rs, int lane, int wave, int nks, int soff_bias) {
  constexpr int J1 = UC<WIDE>::J1, CS = UC<WIDE>::CS;
#pragma unroll
  for (int tap = LO; tap < HI; ++tap)
#pragma unroll
    for (int st = 0; st < CS; ++st)
#pragma unroll
      for (int j = 0; j < J1; ++j)
        w.w1[tap][st][j] = buf_frag<T>(rs, (wave + kMcfWaves * j) * nks * 1024 + lane * 16, (tap * CS + st) * 1024 + soff_bias);
}
template <typename T, bool WIDE, int LO, int HI>
__device__ __forceinline__ void reload_w2(McfW<T>& w, rsrc_t rs, int lane, int wave, int n2, int soff_bias) {
#pragma unroll
  for (int st = LO; st < HI; ++st)
    if (st < UC<WIDE>::N2S) w.w2[st][0] = buf_frag<T>(rs, wave * n2 * 1024 + lane * 16, (st < n2 ? st * 1024 : kOob) + soff_bias);
}
template <typename T, bool WIDE, bool RELOAD>
__device__ __forceinline__ void split_gemm1_tile(const unsigned char* xs, int xs_pitch, const McfGeom& g, unsigned char* a2t, int a2_pitch, int H,
                                                 McfW<T>& w, int ptile, int lane, int wave, rsrc_t rs_next, int nks, int soff_bias, rsrc_t rs_w2,
                                                 int n2) {
  constexpr int KS = K64<T>::value, E16 = ET<T>::E16, J1 = UC<WIDE>::J1, CS = UC<WIDE>::CS;
  typedef typename ET<T>::frag frag_t;
  const int r = lane & 15, gq = lane >> 4;
  const unsigned char* zrow = xs + 64 * xs_pitch;
  if (RELOAD && !IPOKE_UNIT_SPREAD) {
    // THIS layer's 1x1 weights (their registers are free since the previous layer's second contraction): requested here -- behind
    // the halo sweep, so that the sweep's loads and the previous coupling's hand-off stores do not queue behind them in the CU's
    // one vector-memory pipeline -- they land underneath this tile and the barrier behind it
    const int voff2 = wave * n2 * 1024 + lane * 16;
#pragma unroll
    for (int st = 0; st < UC<WIDE>::N2S; ++st) w.w2[st][0] = buf_frag<T>(rs_w2, voff2, st < n2 ? st * 1024 : kOob);
  }
  f32x4 acc[J1];
#pragma unroll
  for (int j = 0; j < J1; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int voff[J1];
#pragma unroll
  for (int j = 0; j < J1; ++j) voff[j] = (wave + kMcfWaves * j) * nks * 1024 + lane * 16;
#pragma unroll
  for (int tap = 0; tap < 6; ++tap) {
    const unsigned char* src = tap_src_fwd(xs, zrow, xs_pitch, g, ptile + r, tap) + E16 * gq * (int)sizeof(T);
#pragma unroll
    for (int st = 0; st < CS; ++st) {
      const frag_t fa = *reinterpret_cast<const frag_t*>(src + st * KS * (int)sizeof(T));
#pragma unroll
      for (int j = 0; j < J1; ++j) mma64(fa, w.w1[tap][st][j], acc[j]);
    }
    if (RELOAD && (!IPOKE_UNIT_SPREAD || tap >= 4)) {        // spread: taps 0, 1 behind the last two taps; the rest between the later phases
#pragma unroll
      for (int st = 0; st < CS; ++st)
#pragma unroll
        for (int j = 0; j < J1; ++j) {
          const int rt = IPOKE_UNIT_SPREAD ? tap - 4 : tap;
          w.w1[rt][st][j] = buf_frag<T>(rs_next, voff[j], (rt * CS + st) * 1024 + soff_bias);
        }
    }
  }
#pragma unroll
  for (int j = 0; j < J1; ++j) {
    const int n = (wave + kMcfWaves * j) * 16 + 4 * gq;
    if (n < H) {
      typename Pack4<T>::type tv;
#pragma unroll
      for (int q = 0; q < 4; ++q) tv[q] = ET<T>::from_f32(fast_elu(acc[j][q]));
      *reinterpret_cast<typename Pack4<T>::type*>(a2t + r * a2_pitch + n * (int)sizeof(T)) = tv;
    }
  }
}
// raw (mu, s)[row][0:2C] = A2 x W2^T for the MT tiles of this part
template <typename T, bool WIDE, int MT>
__device__ __forceinline__ void split_gemm2(const unsigned char* a2, int a2_pitch, float* prm, int N2, int prm_ld, const McfW<T>& w, int lane, int wave) {
  constexpr int KS = K64<T>::value, E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  const int r = lane & 15, gq = lane >> 4;
  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int st = 0; st < UC<WIDE>::N2S; ++st) {
    frag_t fa[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
      fa[i] = *reinterpret_cast<const frag_t*>(a2 + (i * 16 + r) * a2_pitch + (st * KS + E16 * gq) * (int)sizeof(T));
#pragma unroll
    for (int i = 0; i < MT; ++i) mma64(fa[i], w.w2[st][0], acc[i]);
  }
  const int n = wave * 16 + 4 * gq;
  if (n < N2) {
#pragma unroll
    for (int i = 0; i < MT; ++i) *reinterpret_cast<f32x4*>(prm + (i * 16 + r) * prm_ld + n) = acc[i];
  }
}

template <typename T, bool WIDE, int S>
__global__ __launch_bounds__(kMcfThreads) void macow_unit_fwd_split_kernel(const UnitParams U) {
  constexpr int RY = 8 / S, R = 64 / S, MT = 4 / S;
  constexpr int KS = K64<T>::value;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unit_kernarg_prefetch();
  const int b = blockIdx.x / S, s = blockIdx.x - b * S, tid = threadIdx.x;
  const int y0 = s * RY, p0 = y0 * 8;
  McfW<T> wr;
  UNIT_STAMP_S(0);
  const int C = U.C, N2 = 2 * C, ld = U.ld;
  constexpr int K2c = UC<WIDE>::N2S * 32;
  const int xs_pitch = U.Cp * (int)sizeof(T) + kTilePad;
  constexpr int a2_pitch = K2c * (int)sizeof(T) + kTilePad;
  unsigned char* xs = smem;                                        // T [64 + zero row][Cp], GLOBAL position index: own rows + halo rows
  unsigned char* a2 = xs + 65 * xs_pitch;                          // T [R][K2c]: [ELU(c) | ELU(cond) | 0], local rows
  const int prm_ld = N2 + 4;
  float* prm = reinterpret_cast<float*>(a2 + R * a2_pitch);        // [R][2C (+4)] raw (mu, s)
  float* xf = prm + R * prm_ld;                                    // [R][C] fp32 state of the own rows
  float* bias_s = xf + R * C;                                      // [4][2C]
  float* post_s = bias_s + 4 * N2;                                 // [4][2][C]
  float* red = post_s + 8 * C;                                     // [8 waves][4 layers]
  const long row0 = (long)b * 64;
  const int G2 = C >> 1;
  const float inv_g2 = 1.f / (float)G2;
  const int nks = U.K1p / KS, n2 = U.K2p / KS;

  // prologue: the own rows plus the two rows on either side (layer 0 reads its halo from the unit's input in global memory)
  constexpr int E16c = ET<T>::E16;
  const int cchunks = U.Cc / E16c;
  const int ylo = y0 >= 2 ? y0 - 2 : 0, yhi = y0 + RY + 2 <= 8 ? y0 + RY + 2 : 8;
  const int nposx = (yhi - ylo) * 8;
  constexpr int NX = S == 1 ? 4 : ((RY + 4) * 8 * 32 + kMcfThreads - 1) / kMcfThreads;
  constexpr int NCI = (R * 16 + kMcfThreads - 1) / kMcfThreads;
  constexpr int NIT = (R * 32 + kMcfThreads - 1) / kMcfThreads;
  const T* condp = reinterpret_cast<const T*>(U.cond) + (row0 + p0) * U.Cc;
  f32x2 xin[NX]; u32x4 cin[NCI];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int e = tid + i * kMcfThreads;
    const int pp = (int)(((float)e + 0.5f) * inv_g2), c = (e - pp * G2) * 2;
    if (e < nposx * G2) xin[i] = *reinterpret_cast<const f32x2*>(U.x + (row0 + ylo * 8 + pp) * ld + c);
  }
#pragma unroll
  for (int i = 0; i < NCI; ++i) {
    const int e = tid + i * kMcfThreads;
    if (e < R * cchunks) cin[i] = *reinterpret_cast<const u32x4*>(condp + (long)(e / cchunks) * U.Cc + (e % cchunks) * E16c);
  }
  // (static layer index: the pointers are uniform kernel arguments -- indexed per lane they become vector loads of the argument
  //  block followed by s_waitcnt vmcnt(0), i.e. dependent memory round trips in front of the weight requests)
  float bias_v = 0.f, post_e = 0.f, post_b = 0.f;
  bool post_on = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (tid >= q * N2 && tid < (q + 1) * N2) bias_v = U.L[q].bias2[tid - q * N2];
    if (U.L[q].post_ls && tid >= q * C && tid < (q + 1) * C) {
      post_e = U.L[q].post_ls[tid - q * C]; post_b = U.L[q].post_bias[tid - q * C]; post_on = true;
    }
  }
  unit_load_w1<T, WIDE>(wr, U.L[0].W1, U);
  if (IPOKE_UNIT_SPREAD) unit_load_w2<T, WIDE>(wr, U.L[0].W2, U);
  for (int i = tid; i < (65 * xs_pitch + R * a2_pitch) / 16; i += kMcfThreads) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
  if (U.L[3].y && ld > C) {
    const int R2 = (ld - C) >> 1;
    for (int e = tid; e < R * R2; e += kMcfThreads) {
      const int p = p0 + e / R2, c = C + (e % R2) * 2;
      *reinterpret_cast<f32x2*>(U.L[3].y + (row0 + p) * ld + c) = *reinterpret_cast<const f32x2*>(U.x + (row0 + p) * ld + c);
    }
  }
  __syncthreads();
  if (tid < 4 * N2) bias_s[tid] = bias_v;
  if (tid < 4 * C) {
    post_s[(tid / C) * 2 * C + tid % C] = post_on ? __expf(post_e) : 1.f;
    post_s[(tid / C) * 2 * C + C + tid % C] = post_b;
  }
#pragma unroll
  for (int i = 0; i < NCI; ++i) {
    const int e = tid + i * kMcfThreads;
    if (e < R * cchunks)
      *reinterpret_cast<u32x4*>(a2 + (e / cchunks) * a2_pitch + (U.H + (e % cchunks) * E16c) * (int)sizeof(T)) = cin[i];
  }
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int e = tid + i * kMcfThreads;
    const int pp = (int)(((float)e + 0.5f) * inv_g2), c = (e - pp * G2) * 2;
    if (e < nposx * G2) {
      const int p = ylo * 8 + pp;
      if (p >= p0 && p < p0 + R) *reinterpret_cast<f32x2*>(xf + (p - p0) * C + c) = xin[i];
      bf16x2 tv; tv[0] = ET<T>::from_f32(xin[i][0]); tv[1] = ET<T>::from_f32(xin[i][1]);
      *reinterpret_cast<bf16x2*>(xs + p * xs_pitch + c * (int)sizeof(T)) = tv;
    }
  }
  __syncthreads();
  UNIT_STAMP_S(1);
  float ld_k[4] = {0.f, 0.f, 0.f, 0.f};        // per-thread log-det partials of the four layers, summed over the workgroup at the end
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const UnitLayer& Lk = U.L[k];
    const McfGeom g = mcf_geom(Lk.order);
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int lane = tl & 63, wave = tl >> 6;
    const int kn = k < 3 ? k + 1 : 3, soff_bias = k < 3 ? 0 : kOob;
    const rsrc_t rs1n = make_rsrc(U.L[kn].W1, ((U.H + 15) & ~15) * U.K1p * (int)sizeof(T));
    const rsrc_t rs2 = make_rsrc(Lk.W2, ((2 * C + 15) & ~15) * U.K2p * (int)sizeof(T));
    const rsrc_t rs2n = make_rsrc(U.L[kn].W2, ((2 * C + 15) & ~15) * U.K2p * (int)sizeof(T));
    // halo rows of this layer's input that live in the neighbouring parts (layer 0: staged from global memory above)
    const int nu = (k > 0 && s > 0) ? rows_above(g) : 0, nd = (k > 0 && s < S - 1) ? rows_below(g) : 0;
    if constexpr (MT == 2) {
      // the tile next to the part boundary waits for the halo, the other one runs first (S = 2: one neighbour)
      const int t_dep = s == 0 ? 1 : 0;
      const bool wait = (nu + nd) > 0;
      const int t0 = wait ? 1 - t_dep : 0;
      Halo<1> hl;
      if (wait) halo_begin<S, 1>(hl, xg_region(U, b, k, S, s), s, nu, nd, G2, 32, xs, xs_pitch, tl);
      split_gemm1_tile<T, WIDE, false>(xs, xs_pitch, g, a2 + t0 * 16 * a2_pitch, a2_pitch, U.H, wr, p0 + t0 * 16, lane, wave, rs1n, nks, soff_bias, rs2, n2);
      __builtin_amdgcn_sched_barrier(0);
      UNIT_STAMP_S(2 + 8 * k);
      if (wait) {
        halo_end<1>(U, hl);
        __syncthreads();
      }
      __builtin_amdgcn_sched_barrier(0);
      UNIT_STAMP_S(3 + 8 * k);
      if (IPOKE_UNIT_SPREAD && k > 0) reload_w2<T, WIDE, 8, 12>(wr, rs2, lane, wave, n2, 0);
      split_gemm1_tile<T, WIDE, true>(xs, xs_pitch, g, a2 + (1 - t0) * 16 * a2_pitch, a2_pitch, U.H, wr, p0 + (1 - t0) * 16, lane, wave, rs1n, nks,
                                      soff_bias, rs2, n2);
    } else if constexpr (MT == 1) {
      UNIT_STAMP_S(2 + 8 * k);
      if (nu + nd > 0) {
        Halo<1> hl;
        halo_begin<S, 1>(hl, xg_region(U, b, k, S, s), s, nu, nd, G2, 32, xs, xs_pitch, tl);
        halo_end<1>(U, hl);
        __syncthreads();
      }
      UNIT_STAMP_S(3 + 8 * k);
      if (IPOKE_UNIT_SPREAD && k > 0) reload_w2<T, WIDE, 8, 12>(wr, rs2, lane, wave, n2, 0);
      split_gemm1_tile<T, WIDE, true>(xs, xs_pitch, g, a2, a2_pitch, U.H, wr, p0, lane, wave, rs1n, nks, soff_bias, rs2, n2);
    } else {
#pragma unroll
      for (int t = 0; t < MT - 1; ++t)
        split_gemm1_tile<T, WIDE, false>(xs, xs_pitch, g, a2 + t * 16 * a2_pitch, a2_pitch, U.H, wr, p0 + t * 16, lane, wave, rs1n, nks, soff_bias, rs2, n2);
      split_gemm1_tile<T, WIDE, true>(xs, xs_pitch, g, a2 + (MT - 1) * 16 * a2_pitch, a2_pitch, U.H, wr, p0 + (MT - 1) * 16, lane, wave, rs1n, nks,
                                      soff_bias, rs2, n2);
    }
    __builtin_amdgcn_sched_barrier(0);
    UNIT_STAMP_S(4 + 8 * k);
    __syncthreads();
    UNIT_STAMP_S(5 + 8 * k);
    if (IPOKE_UNIT_SPREAD) reload_w1<T, WIDE, 2, 3>(wr, rs1n, lane, wave, nks, soff_bias);
    if (Lk.a2_save) {
      constexpr int E16 = ET<T>::E16;
      const int chunks = U.K2p / E16;
      T* dst = reinterpret_cast<T*>(Lk.a2_save) + (row0 + p0) * U.K2p;
      const float inv_ch = 1.f / (float)chunks;                        // i / chunks: exact for i < 2048, chunks <= 48
      for (int i = tid; i < R * chunks; i += kMcfThreads) {
        const int row = (int)(((float)i + 0.5f) * inv_ch), ch = i - row * chunks;
        *reinterpret_cast<u32x4*>(dst + (long)row * U.K2p + ch * E16) = *reinterpret_cast<const u32x4*>(a2 + row * a2_pitch + ch * 16);
      }
    }
    UNIT_STAMP_S(6 + 8 * k);
    if (IPOKE_UNIT_SPREAD) reload_w1<T, WIDE, 3, 4>(wr, rs1n, lane, wave, nks, soff_bias);
    split_gemm2<T, WIDE, MT>(a2, a2_pitch, prm, N2, prm_ld, wr, lane, wave);
    __builtin_amdgcn_sched_barrier(0);
    if (IPOKE_UNIT_SPREAD) reload_w1<T, WIDE, 4, 5>(wr, rs1n, lane, wave, nks, soff_bias);
    UNIT_STAMP_S(7 + 8 * k);
    __syncthreads();
    UNIT_STAMP_S(8 + 8 * k);
    // affine coupling (+ ActNorm) on the own rows; rows a neighbour needs for the next layer leave as granules from here
    float ld_acc = 0.f;
    const float* bk = bias_s + k * N2;
    const float* pe = post_s + k * 2 * C;
    const McfGeom gn = mcf_geom(U.L[kn].order);
    const int to_up = (k < 3 && s > 0) ? rows_below(gn) : 0;          // the part above reads my top rows
    const int to_dn = (k < 3 && s < S - 1) ? rows_above(gn) : 0;      // the part below reads my bottom rows
    gu64* reg_up = xg_region(U, b, kn, S, s > 0 ? s - 1 : 0);
    gu64* reg_dn = xg_region(U, b, kn, S, s < S - 1 ? s + 1 : 0);
    {
      f32x2 mu[NIT], sv[NIT], xv[NIT];
      int pi[NIT], ci[NIT];
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int e = tl + i * kMcfThreads;
        pi[i] = (int)(((float)e + 0.5f) * inv_g2);
        ci[i] = (e - pi[i] * G2) * 2;
        if (e < R * G2) {
          mu[i] = *reinterpret_cast<const f32x2*>(prm + pi[i] * prm_ld + ci[i]);
          sv[i] = *reinterpret_cast<const f32x2*>(prm + pi[i] * prm_ld + C + ci[i]);
          xv[i] = *reinterpret_cast<const f32x2*>(xf + pi[i] * C + ci[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int e = tl + i * kMcfThreads;
        if (e < R * G2) {
          const int pl = pi[i], c = ci[i], p = p0 + pl;
          f32x2 sc, yv;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            sc[q] = fast_scale(sv[i][q] + bk[C + c + q]);
            yv[q] = sc[q] * xv[i][q] + (mu[i][q] + bk[c + q]);
          }
          ld_acc += __logf(sc[0] * sc[1]);
          if (Lk.scale_save) *reinterpret_cast<f32x2*>(Lk.scale_save + (row0 + p) * C + c) = sc;
          if (Lk.post_ls) {
#pragma unroll
            for (int q = 0; q < 2; ++q) yv[q] = yv[q] * pe[c + q] + pe[C + c + q];
          }
          *reinterpret_cast<f32x2*>(xf + pl * C + c) = yv;
          bf16x2 tv; tv[0] = ET<T>::from_f32(yv[0]); tv[1] = ET<T>::from_f32(yv[1]);
          const unsigned bits = __builtin_bit_cast(unsigned, tv);
          const int ry = pl >> 3, x = pl & 7;
          if (ry < to_up) xg_store(reg_up + ((2 + ry) * 8 + x) * 32 + (c >> 1), bits);
          if (RY - 1 - ry < to_dn) xg_store(reg_dn + ((2 - (RY - ry)) * 8 + x) * 32 + (c >> 1), bits);
          *reinterpret_cast<unsigned*>(xs + p * xs_pitch + c * (int)sizeof(T)) = bits;
          if (Lk.y) *reinterpret_cast<f32x2*>(Lk.y + (row0 + p) * ld + c) = yv;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) ld_k[q] += q == k ? ld_acc : 0.f;
    if (IPOKE_UNIT_SPREAD) {                      // (behind the coupling's stores and hand-off granules)
      reload_w1<T, WIDE, 5, 6>(wr, rs1n, lane, wave, nks, soff_bias);
      reload_w2<T, WIDE, 0, 4>(wr, rs2n, lane, wave, n2, soff_bias);
    }
    __syncthreads();                              // the state update above is complete
    UNIT_STAMP_S(9 + 8 * k);
    if (IPOKE_UNIT_SPREAD) reload_w2<T, WIDE, 4, 8>(wr, rs2n, lane, wave, n2, soff_bias);
    if (Lk.zc) {
      const int half = Lk.zc_ld >> 1;
      T* zb = reinterpret_cast<T*>(Lk.zc) + (row0 + p0) * Lk.zc_ld;
      const T z0 = ET<T>::from_f32(0.f);
      const float inv_half = 1.f / (float)half;
      for (int i = tid; i < R * half; i += kMcfThreads) {
        const int r = (int)(((float)i + 0.5f) * inv_half), k2 = (i - r * half) * 2;
        const unsigned char* xr = xs + (p0 + r) * xs_pitch;
        bf16x2 v;
        v[0] = k2 < Lk.zc_cin ? *reinterpret_cast<const T*>(xr + (Lk.zc_off + k2 * Lk.zc_stride) * (int)sizeof(T)) : z0;
        v[1] = k2 + 1 < Lk.zc_cin ? *reinterpret_cast<const T*>(xr + (Lk.zc_off + (k2 + 1) * Lk.zc_stride) * (int)sizeof(T)) : z0;
        *reinterpret_cast<bf16x2*>(zb + (long)r * Lk.zc_ld + k2) = v;
      }
    }
  }
  // log-dets: one reduction for the four layers (off every layer's critical path)
#pragma unroll
  for (int q = 0; q < 4; ++q) ld_k[q] = wave_sum(ld_k[q]);
  if ((tid & 63) == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) red[(tid >> 6) * 4 + q] = ld_k[q];
  }
  __syncthreads();
  if (tid < 4) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < kMcfWaves; ++w) t += red[w * 4 + tid];
    float* slot = tid == 0 ? U.L[0].ld_slot : tid == 1 ? U.L[1].ld_slot : tid == 2 ? U.L[2].ld_slot : U.L[3].ld_slot;
    if (slot) slot[(long)b * U.slot_w + s] = t;
  }
}

template <int S>
static int fwd_split_launch_s(const UnitParams& U, hipStream_t s) {
  const bool wide = U.Cp > 32;
  const int R = 64 / S;
  const size_t lds = (size_t)65 * (U.Cp * 2 + kTilePad) + (size_t)R * ((wide ? 384 : 256) * 2 + kTilePad) + (size_t)R * (2 * U.C + 4) * 4 +
                     (size_t)R * U.C * 4 + (size_t)(8 * U.C + 8 * U.C + 32) * 4;
  int rc;
  if (wide) {
    rc = ensure_lds<macow_unit_fwd_split_kernel<bf16_t, true, S>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((macow_unit_fwd_split_kernel<bf16_t, true, S>), dim3(U.B * S), dim3(kMcfThreads), lds, s, U);
  } else {
    rc = ensure_lds<macow_unit_fwd_split_kernel<bf16_t, false, S>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((macow_unit_fwd_split_kernel<bf16_t, false, S>), dim3(U.B * S), dim3(kMcfThreads), lds, s, U);
  }
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

int unit_fwd_split_launch(const UnitParams& U, int S, hipStream_t s) {
  IPK_REQUIRE(U.xchg != nullptr && U.xchg_stride >= 4 * 8 * 32, "row-split unit launch without exchange scratch");
  IPK_REQUIRE(U.slot_w >= S, "log-det slot narrower than the split");
  if (S == 2) return fwd_split_launch_s<2>(U, s);
  if (S == 4) return fwd_split_launch_s<4>(U, s);
  IPK_REQUIRE(false, "split must be 2 or 4");
  return IPOKE_OK;
}

// ------------------------------------------------------------------------------------------------ backward
// As macow_unit_bwd_kernel (mcf_unit.hip) on the R = 64 / S rows of this part.  Per layer: (a) coupling-parameter gradients and
// (b) the gradient of the hidden activations dc are local to a position; (c), the adjoint of the shifted convolution, reads dc of the
// 1-2 rows across the cut (the mirror image of the forward halo: layer A needs the two rows BELOW, B the two rows above).  The dc
// rows a neighbour needs leave as granules from the epilogue of (b) (a lane's four hidden channels = two granules = one 16-byte
// write-through store); (c) runs the tile that does not touch the cut first, sweeps the halo, then the tile at the cut.
// Parameter-gradient partials: one row per part (row b * S + s of dbias_part / post_part).
template <typename T, bool WIDE, int S>
__global__ __launch_bounds__(kMcfThreads) void macow_unit_bwd_split_kernel(const UnitParams U) {
  constexpr int RY = 8 / S, R = 64 / S, MT = 4 / S;
  constexpr int J1 = UC<WIDE>::J1, N3S = UC<WIDE>::N3S, HS = UC<WIDE>::HS;
  constexpr int KS = K64<T>::value, E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  typedef typename Pack4<T>::type pack_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unit_kernarg_prefetch();
  const int b = blockIdx.x / S, s = blockIdx.x - b * S, tid = threadIdx.x;
  const int y0 = s * RY, p0 = y0 * 8;
  const int lane = tid & 63, wave = tid >> 6, r = lane & 15, gq = lane >> 4;
  const int C = U.C, N2 = 2 * C, ld = U.ld, H = U.H;
  const int n3 = U.K3p / KS, hs = U.Hq / KS;
  const int nfrag = wave & 3, kh = wave >> 2;
  const long row0 = (long)b * 64, rowp = row0 + p0;
  frag_t w2t[N3S][J1];
  frag_t w1t[3][HS];
  pack_t cact[MT][J1];
  const int Hr = (H + 15) & ~15;
  int vo_w2t[J1], vo_ca[J1];
#pragma unroll
  for (int j = 0; j < J1; ++j) {
    vo_w2t[j] = (wave + kMcfWaves * j) * n3 * 1024 + lane * 16;
    vo_ca[j] = (r * U.K2p + (wave + kMcfWaves * j) * 16 + 4 * gq) * (int)sizeof(T);
  }
  const int vo_w1t = (nfrag * 6 * hs + kh * 3 * hs) * 1024 + lane * 16;
  auto load_w2t = [&](const UnitLayer& Lk) {
    const rsrc_t rs = make_rsrc(Lk.W2T, Hr * U.K3p * (int)sizeof(T));
#pragma unroll
    for (int st = 0; st < N3S; ++st) {
      const int soff = st < n3 ? st * 1024 : kOob;
#pragma unroll
      for (int j = 0; j < J1; ++j) w2t[st][j] = buf_frag<T>(rs, vo_w2t[j], soff);
    }
  };
  auto load_cact = [&](const UnitLayer& Lk) {     // saved ELU outputs of the own rows, the 4 columns a lane owns in phase (b)
    const rsrc_t rs = make_rsrc(reinterpret_cast<const T*>(Lk.a2_save) + rowp * U.K2p, R * U.K2p * (int)sizeof(T));
#pragma unroll
    for (int j = 0; j < J1; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, vo_ca[j], i * 16 * U.K2p * (int)sizeof(T), 0);
        cact[i][j] = __builtin_bit_cast(pack_t, v);
      }
  };
  auto load_w1t = [&](const UnitLayer& Lk) {
    const rsrc_t rs = make_rsrc(Lk.W1T, ((C + 15) & ~15) * 6 * U.Hq * (int)sizeof(T));
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int st = 0; st < HS; ++st)
        w1t[t][st] = buf_frag<T>(rs, vo_w1t, st < hs ? (t * hs + st) * 1024 : kOob);
  };
  UNIT_STAMP_S(0);

  constexpr int dp_pitch = N3S * 32 * (int)sizeof(T) + kTilePad;
  constexpr int dc_pitch = HS * 32 * (int)sizeof(T) + kTilePad;
  unsigned char* dp = smem;                                       // T [R][K3p], local rows (later: fp32 [R][CP] tap-half partials)
  unsigned char* dc = dp + R * dp_pitch;                          // T [64 + zero row][Hq], GLOBAL position index: own rows + halo rows
  float* gb = reinterpret_cast<float*>(dc + 65 * dc_pitch);       // [R][CP] running gradient / dy*scale, local rows
  const int CP = unit_gb_pitch(C);
  float* psum = gb + R * CP;                                      // [2][rows_par <= 64][2C]
  float* red2 = psum + 2 * 4096;                                  // [2][Q][2C]
  const int G2 = C >> 1;
  const float inv_g2 = 1.f / (float)G2;
  constexpr int NIT = (R * 32 + kMcfThreads - 1) / kMcfThreads;
  f32x2 gin[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int e = tid + i * kMcfThreads;
    const int p = (int)(((float)e + 0.5f) * inv_g2), c = (e - p * G2) * 2;
    if (e < R * G2) gin[i] = *reinterpret_cast<const f32x2*>(U.dy + (rowp + p) * ld + c);
  }
  const float g_ld = U.dld[b];
  load_w2t(U.L[3]); load_cact(U.L[3]); load_w1t(U.L[3]);
  for (int i = tid; i < (R * dp_pitch + 65 * dc_pitch) / 16; i += kMcfThreads) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int e = tid + i * kMcfThreads;
    const int p = (int)(((float)e + 0.5f) * inv_g2), c = (e - p * G2) * 2;
    if (e < R * G2) *reinterpret_cast<f32x2*>(gb + p * CP + c) = gin[i];
  }
  if (ld > C) {
    const int R2 = (ld - C) >> 1;
    for (int e = tid; e < R * R2; e += kMcfThreads) {
      const int p = e / R2, c = C + (e - p * R2) * 2;
      *reinterpret_cast<f32x2*>(U.dx + (rowp + p) * ld + c) = *reinterpret_cast<const f32x2*>(U.dy + (rowp + p) * ld + c);
    }
  }
  const int rows_par = kMcfThreads / G2;                          // >= 16 (C <= 64)
  const int rows_used = rows_par < R ? rows_par : R;
  const rsrc_t rs_x = make_rsrc(U.xchg, 0x7ffffff0);              // granule stores (16 bytes = two granules, write-through)
  __syncthreads();
  UNIT_STAMP_S(1);
#pragma unroll 1
  for (int k = 3; k >= 0; --k) {
    const UnitLayer& Lk = U.L[k];
    const McfGeom g = mcf_geom(Lk.order);
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int lane = tl & 63, wave = tl >> 6, r = lane & 15, gq = lane >> 4, nfrag = wave & 3, kh = wave >> 2;
    const int c2 = (tl % G2) * 2, r0 = tl / G2;
    // halo of the adjoint: output row y reads dc of the rows y - ky - oy
    const int nu = s > 0 ? rows_below(g) : 0, nd = s < S - 1 ? rows_above(g) : 0;          // rows I need from above / below
    const int to_up = s > 0 ? rows_above(g) : 0, to_dn = s < S - 1 ? rows_below(g) : 0;    // my top / bottom rows a neighbour needs
    // (a) thread (r0, c2) owns the channel pair c2 of the local rows r0, r0 + rows_par, ...
    if (r0 < rows_used) {
      f32x2 xv[MT], scv[MT], ypv[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int p = r0 + i * rows_par;
        if (p < R) {
          xv[i] = *reinterpret_cast<const f32x2*>(Lk.x + (rowp + p) * ld + c2);
          scv[i] = *reinterpret_cast<const f32x2*>(Lk.scale_save + (rowp + p) * C + c2);
          if (Lk.post_ls) ypv[i] = *reinterpret_cast<const f32x2*>(Lk.y_post + (rowp + p) * ld + c2);
        }
      }
      f32x2 sg = {0.f, 0.f}, sd = sg, s_ls = sg, s_b = sg;
      f32x2 pl = {0.f, 0.f}, pb = pl;
      if (Lk.post_ls) {
        pl = *reinterpret_cast<const f32x2*>(Lk.post_ls + c2); pb = *reinterpret_cast<const f32x2*>(Lk.post_bias + c2);
        pl[0] = __expf(pl[0]); pl[1] = __expf(pl[1]);
      }
      T* dps = reinterpret_cast<T*>(Lk.dparams_save);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int p = r0 + i * rows_par;
        if (p < R) {
          f32x2 gy = *reinterpret_cast<const f32x2*>(gb + p * CP + c2);
          if (Lk.post_ls) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              s_ls[q] += gy[q] * (ypv[i][q] - pb[q]);
              s_b[q] += gy[q];
              gy[q] *= pl[q];
            }
          }
          f32x2 ds, dxv;
          bf16x2 tm, ts;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const float sc = scv[i][q], t = sc - 1.f;
            ds[q] = (gy[q] * xv[i][q] + __fdividef(g_ld, sc)) * 0.5f * (1.f - t * t);
            dxv[q] = gy[q] * sc;
            tm[q] = ET<T>::from_f32(gy[q]); ts[q] = ET<T>::from_f32(ds[q]);
            sg[q] += gy[q]; sd[q] += ds[q];
          }
          *reinterpret_cast<f32x2*>(gb + p * CP + c2) = dxv;
          *reinterpret_cast<bf16x2*>(dp + p * dp_pitch + c2 * (int)sizeof(T)) = tm;
          *reinterpret_cast<bf16x2*>(dp + p * dp_pitch + (C + c2) * (int)sizeof(T)) = ts;
          *reinterpret_cast<bf16x2*>(dps + (rowp + p) * U.K3p + c2) = tm;
          *reinterpret_cast<bf16x2*>(dps + (rowp + p) * U.K3p + C + c2) = ts;
          if (Lk.x_op) {
            bf16x2 xo;
            xo[0] = ET<T>::from_f32(xv[i][0]); xo[1] = ET<T>::from_f32(xv[i][1]);
            *reinterpret_cast<bf16x2*>(reinterpret_cast<T*>(Lk.x_op) + (rowp + p) * U.Cp + c2) = xo;
          }
        }
      }
      *reinterpret_cast<f32x2*>(psum + r0 * N2 + c2) = sg;
      *reinterpret_cast<f32x2*>(psum + r0 * N2 + C + c2) = sd;
      if (Lk.post_ls) {
        *reinterpret_cast<f32x2*>(psum + 4096 + r0 * N2 + c2) = s_ls;
        *reinterpret_cast<f32x2*>(psum + 4096 + r0 * N2 + C + c2) = s_b;
      }
    }
    {   // zero the K padding of the saved coupling-parameter gradients / of the operand copy of x (read by the weight-gradient GEMMs)
      T* dps = reinterpret_cast<T*>(Lk.dparams_save);
      const int padc = U.K3p - N2;
      for (int e = tid; e < R * padc; e += kMcfThreads) {
        const int p = e / padc, j = N2 + e - p * padc;
        dps[(rowp + p) * U.K3p + j] = (T)0.f;
      }
      if (Lk.x_op) {
        T* xo = reinterpret_cast<T*>(Lk.x_op);
        const int padx = U.Cp - C;
        for (int e = tid; e < R * padx; e += kMcfThreads) {
          const int p = e / padx, j = C + e - p * padx;
          xo[(rowp + p) * U.Cp + j] = (T)0.f;
        }
      }
    }
    UNIT_STAMP_S(2 + 8 * (3 - k));
    __syncthreads();
    {   // column sums of the per-thread partials
      const int Q = kMcfThreads / N2;
      const int col = tid % N2, part = tid / N2;
      if (part < Q) {
        float t = 0.f, t2 = 0.f;
        for (int rr = part; rr < rows_used; rr += Q) t += psum[rr * N2 + col];
        red2[part * N2 + col] = t;
        if (Lk.post_ls) {
          for (int rr = part; rr < rows_used; rr += Q) t2 += psum[4096 + rr * N2 + col];
          red2[512 + part * N2 + col] = t2;
        }
      }
    }
    // (b) dA2[:, :H] = dparams x W2[:, :H], times ELU'(c) -> dc (LDS, global position) + dc_save + the neighbours' granules
    {
      T* dcs = reinterpret_cast<T*>(Lk.dc_save);
      const int reg_up = (int)((reinterpret_cast<unsigned long long>(xg_region(U, b, k, S, s > 0 ? s - 1 : 0)) - reinterpret_cast<unsigned long long>(U.xchg)));
      const int reg_dn = (int)((reinterpret_cast<unsigned long long>(xg_region(U, b, k, S, s < S - 1 ? s + 1 : 0)) - reinterpret_cast<unsigned long long>(U.xchg)));
      f32x4 acc[MT][J1];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < J1; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < N3S; ++st) {
        frag_t fa[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
          fa[i] = *reinterpret_cast<const frag_t*>(dp + (i * 16 + r) * dp_pitch + (st * KS + E16 * gq) * (int)sizeof(T));
#pragma unroll
        for (int j = 0; j < J1; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) mma64(fa[i], w2t[st][j], acc[i][j]);
      }
#pragma unroll
      for (int j = 0; j < J1; ++j) {
        const int n = (wave + kMcfWaves * j) * 16 + 4 * gq;
        if (n < H) {
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            const int pl = i * 16 + r;
            const pack_t ca = cact[i][j];
            pack_t tv;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              tv[q] = ET<T>::from_f32(acc[i][j][q] * act_grad_from_out(IPOKE_ACT_ELU, ET<T>::to_f32(ca[q])));
            *reinterpret_cast<pack_t*>(dc + (p0 + pl) * dc_pitch + n * (int)sizeof(T)) = tv;
            *reinterpret_cast<pack_t*>(dcs + (rowp + pl) * U.Hq + n) = tv;
            const u32x2 bits = __builtin_bit_cast(u32x2, tv);
            const u32x4 gran = {bits[0], 1u, bits[1], 1u};
            const int ry = pl >> 3, x = pl & 7;
            if (ry < to_up) __builtin_amdgcn_raw_buffer_store_b128(gran, rs_x, reg_up + (((2 + ry) * 8 + x) * 128 + (n >> 1)) * 8, 0, 16);
            if (RY - 1 - ry < to_dn) __builtin_amdgcn_raw_buffer_store_b128(gran, rs_x, reg_dn + (((2 - (RY - ry)) * 8 + x) * 128 + (n >> 1)) * 8, 0, 16);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      const int padc = U.Hq - H;
      for (int e = tid; e < R * padc; e += kMcfThreads) {
        const int p = e / padc, c = H + e - p * padc;
        dcs[(rowp + p) * U.Hq + c] = (T)0.f;
      }
    }
    UNIT_STAMP_S(3 + 8 * (3 - k));
    __syncthreads();
    UNIT_STAMP_S(4 + 8 * (3 - k));
    if (tid < N2 && Lk.dbias_part) {
      const int Q = kMcfThreads / N2;
      float t = 0.f;
      for (int q = 0; q < Q; ++q) t += red2[q * N2 + tid];
      Lk.dbias_part[((long)b * S + s) * N2 + tid] = t;
    }
    if (tid < N2 && Lk.post_ls && Lk.post_part) {
      const int Q = kMcfThreads / N2;
      float t = 0.f;
      for (int q = 0; q < Q; ++q) t += red2[512 + q * N2 + tid];
      // [d_log_scale | d_bias]; the log-det term of the ActNorm adds (positions of this part) * dld[b] to every d_log_scale
      Lk.post_part[((long)b * S + s) * N2 + tid] = tid < C ? t + (float)R * g_ld : t;
    }
    // (c) g = dy*scale + sum_tap dc[p - off(tap)] x W1[:, tap, :] : wave w owns channel fragment w & 3 and taps 3*(w >> 2) .. +2
    {
      const unsigned char* zrow = dc + 64 * dc_pitch;
      float* part = reinterpret_cast<float*>(dp);               // fp32 [R][CP]: the dparams tile is dead after (b)
      const int n = nfrag * 16 + 4 * gq;
      f32x4 acc_c[MT];
      auto pass_c = [&](int t, f32x4& acc) {
        constexpr int NS = 3 * HS, PD = 3;
        acc = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* src[3];
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) src[tt] = tap_src_adj(dc, zrow, dc_pitch, g, p0 + t * 16 + r, kh * 3 + tt) + E16 * gq * (int)sizeof(T);
        frag_t fa[PD + 1];
#pragma unroll
        for (int s0 = 0; s0 < PD; ++s0) fa[s0] = *reinterpret_cast<const frag_t*>(src[s0 / HS] + (s0 % HS) * KS * (int)sizeof(T));
#pragma unroll
        for (int s0 = 0; s0 < NS; ++s0) {
          if (s0 + PD < NS) fa[(s0 + PD) % (PD + 1)] = *reinterpret_cast<const frag_t*>(src[(s0 + PD) / HS] + ((s0 + PD) % HS) * KS * (int)sizeof(T));
          mma64(fa[s0 % (PD + 1)], w1t[s0 / HS][s0 % HS], acc);
        }
      };
      const bool wait = (nu + nd) > 0;
      Halo<4> hl;
      if (wait) halo_begin<S, 4>(hl, xg_region(U, b, k, S, s), s, nu, nd, H >> 1, 128, dc, dc_pitch, tl);
      // the next layer's (b) operands are requested BEHIND the halo sweep: the CU's vector-memory pipeline is a FIFO, and in the train
      // step (in-situ stamps, scripts/exp/insitu_unit_stamps.py) the sweep's loads queued behind these 72 KB
      if constexpr (MT == 2) {
        const int t_dep = s == 0 ? 1 : 0, t0 = wait ? 1 - t_dep : 0;
        if (t0 == 0) pass_c(0, acc_c[0]); else pass_c(1, acc_c[1]);
        __builtin_amdgcn_sched_barrier(0);
        UNIT_STAMP_S(5 + 8 * (3 - k));
        if (wait) { halo_end<4>(U, hl); __syncthreads(); }
        UNIT_STAMP_S(6 + 8 * (3 - k));
        if (k > 0) { load_w2t(U.L[k - 1]); load_cact(U.L[k - 1]); }
        if (t0 == 0) pass_c(1, acc_c[1]); else pass_c(0, acc_c[0]);
      } else {
        UNIT_STAMP_S(5 + 8 * (3 - k));
        if (wait) { halo_end<4>(U, hl); __syncthreads(); }
        UNIT_STAMP_S(6 + 8 * (3 - k));
        if (k > 0) { load_w2t(U.L[k - 1]); load_cact(U.L[k - 1]); }
#pragma unroll
        for (int t = 0; t < MT; ++t) pass_c(t, acc_c[t]);
      }
      __builtin_amdgcn_sched_barrier(0);
      UNIT_STAMP_S(7 + 8 * (3 - k));
      if (k > 0) load_w1t(U.L[k - 1]);
      if (kh == 1 && n < C) {
#pragma unroll
        for (int i = 0; i < MT; ++i) *reinterpret_cast<f32x4*>(part + (i * 16 + r) * CP + n) = acc_c[i];
      }
      __syncthreads();
      if (kh == 0 && n < C) {
        const bool vec_out = k == 0 && n + 3 < C && (ld & 3) == 0;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int p = i * 16 + r;
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(gb + p * CP + n), a1 = *reinterpret_cast<const f32x4*>(part + p * CP + n);
          const f32x4 a2v = acc_c[i];
          f32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = a0[q] + a2v[q] + a1[q];
          *reinterpret_cast<f32x4*>(gb + p * CP + n) = v;
          if (k == 0 && !U.pair.on) {
            if (vec_out) *reinterpret_cast<f32x4*>(U.dx + (rowp + p) * ld + n) = v;
            else {
#pragma unroll
              for (int q = 0; q < 4; ++q) if (n + q < C) U.dx[(rowp + p) * ld + n + q] = v[q];
            }
          }
        }
      }
      __syncthreads();
      UNIT_STAMP_S(8 + 8 * (3 - k));
      if (k > 0) {   // `part` aliased the dparams tile: restore the zero K padding phase (a) does not rewrite
        const int padc = N3S * 32 - N2;
        for (int e = tid; e < R * padc; e += kMcfThreads) {
          const int p = e / padc, j = N2 + e - p * padc;
          *reinterpret_cast<T*>(dp + p * dp_pitch + j * (int)sizeof(T)) = (T)0.f;
        }
      }
    }
  }
  // ---- the ActNorm (+ shuffle) and the coupling in front of the unit: both are row-wise maps, and this workgroup holds its R rows of
  // the gradient they receive in `gb`.  One launch (ipoke_actnorm_affine_bwd: 13.7 us in the step, 100 times per step) and one pass of
  // that gradient through memory less.  Phase A = actnorm_bwd_kernel, phase B = affine_bwd_kernel (elementwise.hip) on R rows; the
  // parameter-gradient partials go to row b * S + s of their matrices like the unit's own.  The ActNorm's window is the unit's [0, C);
  // columns >= C pass through all three layers and were copied by the prologue (U.dx == pair.dx, host-checked).
  if (U.pair.on) {
    const UnitPair& Q = U.pair;
    float* g1 = psum;                              // [R][CP]: gradient w.r.t. the coupling's output
    float* ps = psum + R * CP;                     // partial column sums
    const int tl = tid;
    {   // phase A
      const int rows_par = kMcfThreads / C;        // >= 8
      const int j = tl % C, r0 = tl / C;
      float a_ls = 0.f, a_b = 0.f;
      if (r0 < rows_par) {
        const int src = Q.an_idx ? Q.an_idx[j] : j;
        const float e = Q.an_ls ? expf(Q.an_ls[src]) : 1.f;
        for (int m = r0; m < R; m += rows_par) {
          const float g = gb[m * CP + j];
          const float xv = Q.an_ls ? Q.an_x[(rowp + m) * ld + src] : 0.f;
          g1[m * CP + src] = g * e;
          a_ls += g * xv * e;
          a_b += g;
        }
        if (Q.an_ls) { ps[r0 * C + j] = a_ls; ps[(rows_par + r0) * C + j] = a_b; }
      }
      __syncthreads();
      if (Q.an_ls && tl < C) {
        const int s_ = Q.an_idx ? Q.an_idx[tl] : tl;       // thread tl accumulated channel s_
        const int used = rows_par < R ? rows_par : R;
        float t_ls = 0.f, t_b = 0.f;
        for (int rr = 0; rr < used; ++rr) { t_ls += ps[rr * C + tl]; t_b += ps[(rows_par + rr) * C + tl]; }
        Q.an_part[((long)b * S + s) * 2 * C + s_] = t_ls + (float)R * g_ld;
        Q.an_part[((long)b * S + s) * 2 * C + C + s_] = t_b;
      }
      __syncthreads();                              // ps is reused below
    }
    {   // phase B
      const int Cq = Q.Cp, rows_par = kMcfThreads / Cq;
      const int i = tl % Cq, r0 = tl / Cq;
      T* dprm = reinterpret_cast<T*>(Q.dparams);
      float a_mu = 0.f, a_s = 0.f;
      const bool act = r0 < rows_par;
      if (act) {
        const int col = Q.t_off + i * Q.t_stride;
        for (int m = r0; m < R; m += rows_par) {
          const float g = g1[m * CP + col];
          const float xv = Q.x0[(rowp + m) * ld + col];
          const float sc = Q.scale[(rowp + m) * Cq + i];
          const float t = sc - 1.f;                                           // tanh(s/2)
          const float ds = (g * xv + g_ld / sc) * 0.5f * (1.f - t * t);
          Q.dx[(rowp + m) * ld + col] = g * sc;
          dprm[(rowp + m) * Q.ldp + i] = ET<T>::from_f32(g);                  // d mu
          dprm[(rowp + m) * Q.ldp + Cq + i] = ET<T>::from_f32(ds);            // d s
          a_mu += g; a_s += ds;
        }
        ps[r0 * 2 * Cq + i] = a_mu; ps[r0 * 2 * Cq + Cq + i] = a_s;
      }
      for (int e = tl; e < R * ld; e += kMcfThreads) {      // untouched channels
        const int m = e / ld, col = e - m * ld;
        const int rel = col - Q.t_off;
        const bool transformed = rel >= 0 && rel % Q.t_stride == 0 && rel / Q.t_stride < Cq;
        if (col < C && !transformed) Q.dx[(rowp + m) * ld + col] = g1[m * CP + col];
      }
      {
        const int pad = Q.ldp - 2 * Cq;
        for (int e = tl; e < R * pad; e += kMcfThreads) {
          const int m = e / pad, j = 2 * Cq + (e - m * pad);
          dprm[(rowp + m) * Q.ldp + j] = ET<T>::from_f32(0.f);
        }
      }
      __syncthreads();
      if (Q.dbias_part && tl < 2 * Cq) {
        const int used = rows_par < R ? rows_par : R;
        float t = 0.f;
        for (int rr = 0; rr < used; ++rr) t += ps[rr * 2 * Cq + tl];
        Q.dbias_part[((long)b * S + s) * 2 * Cq + tl] = t;
      }
    }
  }
}

template <int S>
static int bwd_split_launch_s(const UnitParams& U, hipStream_t s) {
  const bool wide = U.Cp > 32;
  const int R = 64 / S;
  const size_t dp_bytes = (size_t)R * ((wide ? 128 : 64) * 2 + kTilePad);
  const size_t CP = unit_gb_pitch(U.C);
  const size_t lds = dp_bytes + (size_t)65 * ((wide ? 256 : 128) * 2 + kTilePad) + (size_t)R * CP * 4 + (2 * 4096 + 2 * 512) * 4;
  IPK_REQUIRE((size_t)R * CP * 4 <= dp_bytes, "tap-half partials must fit the dparams tile");
  int rc;
  if (wide) {
    rc = ensure_lds<macow_unit_bwd_split_kernel<bf16_t, true, S>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((macow_unit_bwd_split_kernel<bf16_t, true, S>), dim3(U.B * S), dim3(kMcfThreads), lds, s, U);
  } else {
    rc = ensure_lds<macow_unit_bwd_split_kernel<bf16_t, false, S>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((macow_unit_bwd_split_kernel<bf16_t, false, S>), dim3(U.B * S), dim3(kMcfThreads), lds, s, U);
  }
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

int unit_bwd_split_launch(const UnitParams& U, int S, hipStream_t s) {
  IPK_REQUIRE(U.xchg != nullptr && U.xchg_stride >= 4 * 8 * 128, "row-split unit launch without exchange scratch");
  if (S == 2) return bwd_split_launch_s<2>(U, s);
  if (S == 4) return bwd_split_launch_s<4>(U, s);
  IPK_REQUIRE(false, "split must be 2 or 4");
  return IPOKE_OK;
}

}  // namespace ipoke
