// Fused MaCowUnit kernels (reference models/modules/INN/macow2.py:957-995):
//     MCF(A) -> MCF(B) -> ActNorm -> MCF(C) -> MCF(D) -> ActNorm
// in ONE launch per direction, one 8-wave workgroup per sample.  The per-layer kernels of mcf.hip are each bound by
// their own launch + weight-fetch latency (7..25 us for 19 MFLOP per sample); here the 8x8xC latent stays in LDS (fp32
// master copy + matrix-core operand copy) across the four masked convolutions, the conditioning rows are staged once,
// and the weight fragments of layer k+1 are requested into the registers layer k has just finished with, so that only
// the first layer of a unit pays a memory latency.  Everything a backward pass needs (per-layer input states, ELU
// outputs, coupling scales) is written on the way, exactly as the per-layer kernels write it.
//
// bf16 matrix-core inputs only (the register-resident weight path); f32 parity mode runs the per-layer kernels.
#include <type_traits>
#include "mcf_unit_dev.h"

namespace ipoke {


// ------------------------------------------------------------------------------------------------ forward
// hidden = ELU(A1 x W1^T) for rows [32 h, 32 h + 32) -> a2[row][0:H]
template <typename T, bool WIDE>
__device__ __forceinline__ void unit_gemm1(const unsigned char* xs, int xs_pitch, const McfGeom& g, unsigned char* a2, int a2_pitch,
                                           int H, const McfW<T>& w, int h, int lane, int wave) {
  constexpr int KS = K64<T>::value, E16 = ET<T>::E16, J1 = UC<WIDE>::J1;
  typedef typename ET<T>::frag frag_t;
  const int r = lane & 15, gq = lane >> 4;
  const unsigned char* zrow = xs + 64 * xs_pitch;
  f32x4 acc[2][J1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < J1; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < 6; ++tap) {
    const unsigned char* src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) src[i] = tap_src_fwd(xs, zrow, xs_pitch, g, (2 * h + i) * 16 + r, tap) + E16 * gq * (int)sizeof(T);
#pragma unroll
    for (int st = 0; st < UC<WIDE>::CS; ++st) {
      frag_t fa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const frag_t*>(src[i] + st * KS * (int)sizeof(T));
#pragma unroll
      for (int j = 0; j < J1; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) mma64(fa[i], w.w1[tap][st][j], acc[i][j]);
    }
  }
#pragma unroll
  for (int j = 0; j < J1; ++j) {
    const int n = (wave + kMcfWaves * j) * 16 + 4 * gq;
    if (n < H) {     // H is a multiple of 8: the 4 columns of a lane are all valid or all invalid
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        typename Pack4<T>::type tv;
#pragma unroll
        for (int q = 0; q < 4; ++q) tv[q] = ET<T>::from_f32(fast_elu(acc[i][j][q]));
        *reinterpret_cast<typename Pack4<T>::type*>(a2 + ((2 * h + i) * 16 + r) * a2_pitch + n * (int)sizeof(T)) = tv;
      }
    }
  }
}
// raw (mu, s)[row][0:2C] = A2 x W2^T for all 64 rows (bias added by the epilogue)
template <typename T, bool WIDE>
__device__ __forceinline__ void unit_gemm2(const unsigned char* a2, int a2_pitch, float* prm, int N2, int prm_ld, const McfW<T>& w, int lane,
                                           int wave) {
  constexpr int KS = K64<T>::value, E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  const int r = lane & 15, gq = lane >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int st = 0; st < UC<WIDE>::N2S; ++st) {
    frag_t fa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      fa[i] = *reinterpret_cast<const frag_t*>(a2 + (i * 16 + r) * a2_pitch + (st * KS + E16 * gq) * (int)sizeof(T));
#pragma unroll
    for (int i = 0; i < 4; ++i) mma64(fa[i], w.w2[st][0], acc[i]);
  }
  const int n = wave * 16 + 4 * gq;
  if (n < N2) {      // 2C is a multiple of 4
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(prm + (i * 16 + r) * prm_ld + n) = acc[i];
  }
}

template <typename T, bool WIDE>
__global__ __launch_bounds__(kMcfThreads) void macow_unit_fwd_kernel(const UnitParams U) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unit_kernarg_prefetch();
  const int b = blockIdx.x, tid = threadIdx.x;
  McfW<T> wr;
  UNIT_STAMP(0);
  const int C = U.C, N2 = 2 * C, ld = U.ld;
  constexpr int K2c = UC<WIDE>::N2S * 32;                          // tile width of the class (zero beyond K2p)
  const int xs_pitch = U.Cp * (int)sizeof(T) + kTilePad;
  constexpr int a2_pitch = K2c * (int)sizeof(T) + kTilePad;
  unsigned char* xs = smem;                                        // T [64 + zero row][Cp]
  unsigned char* a2 = xs + 65 * xs_pitch;                          // T [64][K2c]: [ELU(c) | ELU(cond) | 0]
  const int prm_ld = N2 + 4;                                       // +4 floats: the fragment-shaped 16-byte stores of 16 rows hit 64 distinct banks
  float* prm = reinterpret_cast<float*>(a2 + 64 * a2_pitch);       // [64][2C (+4)] raw (mu, s)
  float* xf = prm + 64 * prm_ld;                                   // [64][C] fp32 state
  float* bias_s = xf + 64 * C;                                     // [4][2C]
  float* post_s = bias_s + 4 * N2;                                 // [4][2][C]: exp(log_scale), bias of the ActNorms
  float* red = post_s + 8 * C;                                     // [8]
  const long row0 = (long)b * 64;
  const int G2 = C >> 1;
  const float inv_g2 = 1.f / (float)G2;

  // Everything the prologue reads from global memory is requested first (the sample's state, its conditioning rows, the
  // biases / ActNorm parameters of the four layers), then the LDS tiles are zeroed while those loads are in flight.
  constexpr int E16c = ET<T>::E16;
  const int cchunks = U.Cc / E16c;                               // 16-byte chunks per conditioning row (<= 16)
  const T* condp = reinterpret_cast<const T*>(U.cond) + row0 * U.Cc;
  f32x2 xin[4]; u32x4 cin[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + i * kMcfThreads;
    const int p = (int)(((float)e + 0.5f) * inv_g2), c = (e - p * G2) * 2;
    if (e < 64 * G2) xin[i] = *reinterpret_cast<const f32x2*>(U.x + (row0 + p) * ld + c);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + i * kMcfThreads;
    if (e < 64 * cchunks) cin[i] = *reinterpret_cast<const u32x4*>(condp + (long)(e / cchunks) * U.Cc + (e % cchunks) * E16c);
  }
  // (static layer index: the pointers are uniform kernel arguments -- indexed per lane they become vector loads of the argument
  //  block followed by s_waitcnt vmcnt(0); the one in the staging code below waited for the whole weight stream of layer A)
  float bias_v = 0.f, post_e = 0.f, post_b = 0.f;
  bool post_on = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (tid >= q * N2 && tid < (q + 1) * N2) bias_v = U.L[q].bias2[tid - q * N2];      // 4 * 2C <= 512
    if (U.L[q].post_ls && tid >= q * C && tid < (q + 1) * C) {
      post_e = U.L[q].post_ls[tid - q * C]; post_b = U.L[q].post_bias[tid - q * C]; post_on = true;
    }
  }
  // ... and behind them every weight fragment of layer A (vector-memory results return in issue order: requested first,
  // the 295 KB of weights would hold the few KB of the prologue back)
  unit_load_w1<T, WIDE>(wr, U.L[0].W1, U);
  unit_load_w2<T, WIDE>(wr, U.L[0].W2, U);
  for (int i = tid; i < (65 * xs_pitch + 64 * a2_pitch) / 16; i += kMcfThreads) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
  if (U.L[3].y && ld > C) {                         // pass-through channels go straight to the unit's output state
    const int R2 = (ld - C) >> 1;
    for (int e = tid; e < 64 * R2; e += kMcfThreads) {
      const int p = e / R2, c = C + (e - p * R2) * 2;
      *reinterpret_cast<f32x2*>(U.L[3].y + (row0 + p) * ld + c) = *reinterpret_cast<const f32x2*>(U.x + (row0 + p) * ld + c);
    }
  }
  __syncthreads();                                  // the zero fill above precedes the staging below
  if (tid < 4 * N2) bias_s[tid] = bias_v;
  if (tid < 4 * C) {
    post_s[(tid / C) * 2 * C + tid % C] = post_on ? __expf(post_e) : 1.f;
    post_s[(tid / C) * 2 * C + C + tid % C] = post_b;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {      // ELU(cond) rows of this sample behind the hidden columns (the same for the four layers)
    const int e = tid + i * kMcfThreads;
    if (e < 64 * cchunks)
      *reinterpret_cast<u32x4*>(a2 + (e / cchunks) * a2_pitch + (U.H + (e % cchunks) * E16c) * (int)sizeof(T)) = cin[i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + i * kMcfThreads;
    const int p = (int)(((float)e + 0.5f) * inv_g2), c = (e - p * G2) * 2;
    if (e < 64 * G2) {
      *reinterpret_cast<f32x2*>(xf + p * C + c) = xin[i];
      bf16x2 tv; tv[0] = ET<T>::from_f32(xin[i][0]); tv[1] = ET<T>::from_f32(xin[i][1]);
      *reinterpret_cast<bf16x2*>(xs + p * xs_pitch + c * (int)sizeof(T)) = tv;
    }
  }
  __syncthreads();
  UNIT_STAMP(1);
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const UnitLayer& Lk = U.L[k];
    const McfGeom g = mcf_geom(Lk.order);
    // lane indices from an opaque copy of the thread id: keeps the addresses of the phases out of the layer loop's
    // live-in set (they would sit next to the 144 weight registers for the whole kernel)
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int lane = tl & 63, wave = tl >> 6;
    // two passes of 32 rows: halves the accumulator / A-fragment registers next to the weight registers
    unit_gemm1<T, WIDE>(xs, xs_pitch, g, a2, a2_pitch, U.H, wr, 0, lane, wave);
    unit_gemm1<T, WIDE>(xs, xs_pitch, g, a2, a2_pitch, U.H, wr, 1, lane, wave);
    __builtin_amdgcn_sched_barrier(0);
    UNIT_STAMP(2 + 6 * k);
    if (k < 3) unit_load_w1<T, WIDE>(wr, U.L[k + 1].W1, U);      // next layer's shifted-conv weights: their registers are free
    __syncthreads();
    UNIT_STAMP(3 + 6 * k);
    if (Lk.a2_save) {
      constexpr int E16 = ET<T>::E16;
      const int chunks = U.K2p / E16;
      T* dst = reinterpret_cast<T*>(Lk.a2_save) + row0 * U.K2p;
      for (int i = tid; i < 64 * chunks; i += kMcfThreads) {
        const int row = i / chunks, ch = i - row * chunks;
        *reinterpret_cast<u32x4*>(dst + (long)row * U.K2p + ch * E16) = *reinterpret_cast<const u32x4*>(a2 + row * a2_pitch + ch * 16);
      }
    }
    UNIT_STAMP(4 + 6 * k);
    unit_gemm2<T, WIDE>(a2, a2_pitch, prm, N2, prm_ld, wr, lane, wave);
    UNIT_STAMP(5 + 6 * k);
    __syncthreads();
    UNIT_STAMP(6 + 6 * k);
    // affine coupling (+ ActNorm): y = (tanh(s/2) + 1) x + mu ; the fp32 state and its operand copy are updated in place
    float ld_acc = 0.f;
    const float* bk = bias_s + k * N2;
    const float* pe = post_s + k * 2 * C;
    {
      // up to four (row, channel pair) items per thread, all LDS operands in flight before the transcendental math
      f32x2 mu[4], sv[4], xv[4];
      int pi[4], ci[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = tl + i * kMcfThreads;
        pi[i] = (int)(((float)e + 0.5f) * inv_g2);              // e / G2, exact for e < 2048, G2 <= 32
        ci[i] = (e - pi[i] * G2) * 2;
        if (e < 64 * G2) {
          mu[i] = *reinterpret_cast<const f32x2*>(prm + pi[i] * prm_ld + ci[i]);
          sv[i] = *reinterpret_cast<const f32x2*>(prm + pi[i] * prm_ld + C + ci[i]);
          xv[i] = *reinterpret_cast<const f32x2*>(xf + pi[i] * C + ci[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = tl + i * kMcfThreads;
        if (e < 64 * G2) {
          const int p = pi[i], c = ci[i];
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
          *reinterpret_cast<f32x2*>(xf + p * C + c) = yv;
          bf16x2 tv; tv[0] = ET<T>::from_f32(yv[0]); tv[1] = ET<T>::from_f32(yv[1]);
          *reinterpret_cast<bf16x2*>(xs + p * xs_pitch + c * (int)sizeof(T)) = tv;
          if (Lk.y) *reinterpret_cast<f32x2*>(Lk.y + (row0 + p) * ld + c) = yv;
        }
      }
    }
    const float tot = block_sum(ld_acc, red);       // two barriers: the state update above is complete behind them
    UNIT_STAMP(7 + 6 * k);
    if (tid == 0 && Lk.ld_slot) Lk.ld_slot[(long)b * U.slot_w] = tot;
    if (Lk.zc) {
      // conditioning operand of the coupling behind the unit: the selected columns of the (rounded) state tile in LDS, two per
      // 4-byte store, zero beyond zc_cin (scattered 2-byte stores from the update loop above cost 1.5 us per launch)
      const int half = Lk.zc_ld >> 1;
      T* zb = reinterpret_cast<T*>(Lk.zc) + row0 * Lk.zc_ld;
      const T z0 = ET<T>::from_f32(0.f);
      for (int i = tid; i < 64 * half; i += kMcfThreads) {
        const int r = i / half, k2 = (i - r * half) * 2;
        const unsigned char* xr = xs + r * xs_pitch;
        bf16x2 v;
        v[0] = k2 < Lk.zc_cin ? *reinterpret_cast<const T*>(xr + (Lk.zc_off + k2 * Lk.zc_stride) * (int)sizeof(T)) : z0;
        v[1] = k2 + 1 < Lk.zc_cin ? *reinterpret_cast<const T*>(xr + (Lk.zc_off + (k2 + 1) * Lk.zc_stride) * (int)sizeof(T)) : z0;
        *reinterpret_cast<bf16x2*>(zb + (long)r * Lk.zc_ld + k2) = v;
      }
    }
    // the 1x1 weights of the next layer are not needed before its second contraction: requested here, they land
    // underneath its first one (and do not add to the register pressure of the epilogue above)
    if (k < 3) unit_load_w2<T, WIDE>(wr, U.L[k + 1].W2, U);
  }
}

// ------------------------------------------------------------------------------------------------ backward
// Gradient of the whole unit: D, C (through the second ActNorm), then B, A (through the first).  The running gradient
// stays in LDS as fp32 [64][C]; per layer the three phases of mcf_bwd_kernel: (a) gradients of the coupling parameters,
// (b) back through the weight-normed 1x1 conv and ELU, (c) adjoint of the shifted convolution.  Weight fragments of the
// next (= previous in forward order) layer are requested as soon as a phase has consumed the current ones.
template <typename T, bool WIDE>
__global__ __launch_bounds__(kMcfThreads) void macow_unit_bwd_kernel(const UnitParams U) {
  constexpr int J1 = UC<WIDE>::J1, N3S = UC<WIDE>::N3S, HS = UC<WIDE>::HS;
  constexpr int KS = K64<T>::value, E16 = ET<T>::E16;
  typedef typename ET<T>::frag frag_t;
  typedef typename Pack4<T>::type pack_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unit_kernarg_prefetch();
  const int b = blockIdx.x, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, r = lane & 15, gq = lane >> 4;
  const int C = U.C, N2 = 2 * C, ld = U.ld, H = U.H;
  const int n3 = U.K3p / KS, hs = U.Hq / KS;
  const int nfrag = wave & 3, kh = wave >> 2;
  const long row0 = (long)b * 64;
  frag_t w2t[N3S][J1];
  frag_t w1t[3][HS];
  pack_t cact[4][J1];
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
  auto load_cact = [&](const UnitLayer& Lk) {     // saved ELU outputs of this sample, the 4 columns a lane owns in phase (b)
    const rsrc_t rs = make_rsrc(reinterpret_cast<const T*>(Lk.a2_save) + row0 * U.K2p, 64 * U.K2p * (int)sizeof(T));
#pragma unroll
    for (int j = 0; j < J1; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
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
  UNIT_STAMP(0);

  constexpr int dp_pitch = N3S * 32 * (int)sizeof(T) + kTilePad;        // class widths: columns beyond K3p / Hq stay zero
  constexpr int dc_pitch = HS * 32 * (int)sizeof(T) + kTilePad;
  unsigned char* dp = smem;                                       // T [64][K3p]  (later: fp32 [64][C] tap-half partials)
  unsigned char* dc = dp + 64 * dp_pitch;                         // T [64 + zero row][Hq]
  float* gb = reinterpret_cast<float*>(dc + 65 * dc_pitch);       // [64][C] running gradient / dy*scale
  const int CP = unit_gb_pitch(C);                                // row pitch of gb / part: 2 (mod 4) 16-byte units (see kTilePad)
  float* psum = gb + 64 * CP;                                     // [2][rows_par <= 64][2C] per-thread partial column sums
  float* red2 = psum + 2 * 4096;                                  // [2][Q][2C]
  // one-time: zero tiles (K padding and the zero row stay zero), incoming gradient, pass-through channels
  // (the incoming gradient is requested before the weights: vector-memory results return in issue order)
  const int G2 = C >> 1;
  const float inv_g2 = 1.f / (float)G2;
  f32x2 gin[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + i * kMcfThreads;
    const int p = (int)(((float)e + 0.5f) * inv_g2), c = (e - p * G2) * 2;     // e / G2, exact for e < 2048
    if (e < 64 * G2) gin[i] = *reinterpret_cast<const f32x2*>(U.dy + (row0 + p) * ld + c);
  }
  const float g_ld = U.dld[b];
  load_w2t(U.L[3]); load_cact(U.L[3]); load_w1t(U.L[3]);
  for (int i = tid; i < (64 * dp_pitch + 65 * dc_pitch) / 16; i += kMcfThreads) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + i * kMcfThreads;
    const int p = (int)(((float)e + 0.5f) * inv_g2), c = (e - p * G2) * 2;
    if (e < 64 * G2) *reinterpret_cast<f32x2*>(gb + p * CP + c) = gin[i];
  }
  if (ld > C) {
    const int R2 = (ld - C) >> 1;
    for (int e = tid; e < 64 * R2; e += kMcfThreads) {
      const int p = e / R2, c = C + (e - p * R2) * 2;
      *reinterpret_cast<f32x2*>(U.dx + (row0 + p) * ld + c) = *reinterpret_cast<const f32x2*>(U.dy + (row0 + p) * ld + c);
    }
  }
  const int rows_par = kMcfThreads / G2;                          // >= 16 (C <= 64)
  const int rows_used = rows_par < 64 ? rows_par : 64;
  const int c2 = (tid % G2) * 2, r0 = tid / G2;
  __syncthreads();
#pragma unroll 1
  for (int k = 3; k >= 0; --k) {
    const UnitLayer& Lk = U.L[k];
    const McfGeom g = mcf_geom(Lk.order);
    // Lane-derived indices are recomputed per layer from an opaque copy of the thread id: otherwise every LDS / global
    // address of the three phases is hoisted out of the layer loop and kept live next to the 144 weight registers.
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int lane = tl & 63, wave = tl >> 6, r = lane & 15, gq = lane >> 4, nfrag = wave & 3, kh = wave >> 2;
    const int c2 = (tl % G2) * 2, r0 = tl / G2;
    UNIT_STAMP(1 + 6 * (3 - k));
    // (a) thread (r0, c2) owns the channel pair c2 of rows r0, r0 + rows_par, ... (at most 4)
    if (r0 < rows_used) {
      f32x2 xv[4], scv[4], ypv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int p = r0 + i * rows_par;
        if (p < 64) {
          xv[i] = *reinterpret_cast<const f32x2*>(Lk.x + (row0 + p) * ld + c2);
          scv[i] = *reinterpret_cast<const f32x2*>(Lk.scale_save + (row0 + p) * C + c2);
          if (Lk.post_ls) ypv[i] = *reinterpret_cast<const f32x2*>(Lk.y_post + (row0 + p) * ld + c2);
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
      for (int i = 0; i < 4; ++i) {
        const int p = r0 + i * rows_par;
        if (p < 64) {
          f32x2 gy = *reinterpret_cast<const f32x2*>(gb + p * CP + c2);
          if (Lk.post_ls) {
            // ActNorm behind this layer: dls = sum dy (y_post - bias), dbias = sum dy, gradient passed on = dy exp(ls)
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
          *reinterpret_cast<bf16x2*>(dps + (row0 + p) * U.K3p + c2) = tm;
          *reinterpret_cast<bf16x2*>(dps + (row0 + p) * U.K3p + C + c2) = ts;
          if (Lk.x_op) {        // the layer input in the matrix cores' dtype: A operand of the shifted-conv weight gradient
            bf16x2 xo;
            xo[0] = ET<T>::from_f32(xv[i][0]); xo[1] = ET<T>::from_f32(xv[i][1]);
            *reinterpret_cast<bf16x2*>(reinterpret_cast<T*>(Lk.x_op) + (row0 + p) * U.Cp + c2) = xo;
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
    {   // zero the K padding of the saved coupling-parameter gradients (read by the weight-gradient GEMM)
      T* dps = reinterpret_cast<T*>(Lk.dparams_save);
      const int padc = U.K3p - N2;
      for (int e = tid; e < 64 * padc; e += kMcfThreads) {
        const int p = e / padc, j = N2 + e - p * padc;
        dps[(row0 + p) * U.K3p + j] = (T)0.f;
      }
      if (Lk.x_op) {
        T* xo = reinterpret_cast<T*>(Lk.x_op);
        const int padx = U.Cp - C;
        for (int e = tid; e < 64 * padx; e += kMcfThreads) {
          const int p = e / padx, j = C + e - p * padx;
          xo[(row0 + p) * U.Cp + j] = (T)0.f;
        }
      }
    }
    UNIT_STAMP(2 + 6 * (3 - k));
    __syncthreads();
    UNIT_STAMP(3 + 6 * (3 - k));
    {   // column sums of the per-thread partials: Q threads per column, then one thread per column
      const int Q = kMcfThreads / N2;                            // >= 4
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
    // (b) dA2[:, :H] = dparams x W2[:, :H], times ELU'(c) -> dc ; two passes of 32 rows (accumulator registers)
    {
      T* dcs = reinterpret_cast<T*>(Lk.dc_save);
      auto pass_b = [&](auto half_c) {
        constexpr int h = decltype(half_c)::value;
        f32x4 acc[2][J1];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < J1; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < N3S; ++st) {
          frag_t fa[2];
#pragma unroll
          for (int i = 0; i < 2; ++i)
            fa[i] = *reinterpret_cast<const frag_t*>(dp + ((2 * h + i) * 16 + r) * dp_pitch + (st * KS + E16 * gq) * (int)sizeof(T));
#pragma unroll
          for (int j = 0; j < J1; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) mma64(fa[i], w2t[st][j], acc[i][j]);
        }
#pragma unroll
        for (int j = 0; j < J1; ++j) {
          const int n = (wave + kMcfWaves * j) * 16 + 4 * gq;
          if (n < H) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const int p = (2 * h + i) * 16 + r;
              const pack_t ca = cact[2 * h + i][j];
              pack_t tv;
#pragma unroll
              for (int q = 0; q < 4; ++q)
                tv[q] = ET<T>::from_f32(acc[i][j][q] * act_grad_from_out(IPOKE_ACT_ELU, ET<T>::to_f32(ca[q])));
              *reinterpret_cast<pack_t*>(dc + p * dc_pitch + n * (int)sizeof(T)) = tv;
              *reinterpret_cast<pack_t*>(dcs + (row0 + p) * U.Hq + n) = tv;
            }
          }
        }
      };
      pass_b(std::integral_constant<int, 0>());
      __builtin_amdgcn_sched_barrier(0);
      pass_b(std::integral_constant<int, 1>());
      __builtin_amdgcn_sched_barrier(0);
      if (k > 0) { load_w2t(U.L[k - 1]); load_cact(U.L[k - 1]); }
      const int padc = U.Hq - H;
      for (int e = tid; e < 64 * padc; e += kMcfThreads) {
        const int p = e / padc, c = H + e - p * padc;
        dcs[(row0 + p) * U.Hq + c] = (T)0.f;
      }
    }
    UNIT_STAMP(4 + 6 * (3 - k));
    __syncthreads();
    UNIT_STAMP(5 + 6 * (3 - k));
    if (tid < N2 && Lk.dbias_part) {
      const int Q = kMcfThreads / N2;
      float t = 0.f;
      for (int q = 0; q < Q; ++q) t += red2[q * N2 + tid];
      Lk.dbias_part[(long)b * N2 + tid] = t;
    }
    if (tid < N2 && Lk.post_ls && Lk.post_part) {
      const int Q = kMcfThreads / N2;
      float t = 0.f;
      for (int q = 0; q < Q; ++q) t += red2[512 + q * N2 + tid];
      // [d_log_scale | d_bias]; the log-det term of the ActNorm adds P * dld[b] to every d_log_scale
      Lk.post_part[(long)b * N2 + tid] = tid < C ? t + 64.f * g_ld : t;
    }
    // (c) g = dy*scale + sum_tap dc[p - off(tap)] x W1[:, tap, :] : wave w owns channel fragment w & 3 and taps 3*(w >> 2) .. +2;
    //     two passes of 32 rows; the two tap halves meet in LDS
    {
      const unsigned char* zrow = dc + 64 * dc_pitch;
      float* part = reinterpret_cast<float*>(dp);               // fp32 [64][C]: the dparams tile is dead after (b)
      const int n = nfrag * 16 + 4 * gq;
      f32x4 acc_lo[2], acc_hi[2];
      auto pass_c = [&](auto half_c, f32x4* acc) {
        constexpr int h = decltype(half_c)::value;
        constexpr int NS = 3 * HS, PD = 3;           // K steps of this wave's tap half; A fragments requested PD steps ahead
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* src[3][2];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            src[t][i] = tap_src_adj(dc, zrow, dc_pitch, g, (2 * h + i) * 16 + r, kh * 3 + t) + E16 * gq * (int)sizeof(T);
        // every MFMA needs its own A fragment here (one channel fragment per wave): without the explicit look-ahead the
        // compiler issues read -> wait -> MFMA one at a time and the phase runs at LDS latency (19 k cycles per layer measured)
        frag_t fa[PD + 1][2];
#pragma unroll
        for (int s0 = 0; s0 < PD; ++s0)
#pragma unroll
          for (int i = 0; i < 2; ++i) fa[s0][i] = *reinterpret_cast<const frag_t*>(src[s0 / HS][i] + (s0 % HS) * KS * (int)sizeof(T));
#pragma unroll
        for (int s0 = 0; s0 < NS; ++s0) {
          if (s0 + PD < NS) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
              fa[(s0 + PD) % (PD + 1)][i] = *reinterpret_cast<const frag_t*>(src[(s0 + PD) / HS][i] + ((s0 + PD) % HS) * KS * (int)sizeof(T));
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) mma64(fa[s0 % (PD + 1)][i], w1t[s0 / HS][s0 % HS], acc[i]);
        }
      };
      pass_c(std::integral_constant<int, 0>(), acc_lo);
      __builtin_amdgcn_sched_barrier(0);
      UNIT_STAMP(32 + 4 * (3 - k));
      pass_c(std::integral_constant<int, 1>(), acc_hi);
      __builtin_amdgcn_sched_barrier(0);
      UNIT_STAMP(33 + 4 * (3 - k));
      if (k > 0) load_w1t(U.L[k - 1]);
      UNIT_STAMP(34 + 4 * (3 - k));
      // the two tap halves meet in LDS: 16-byte accesses, a lane's 4 channels are contiguous (columns >= C are padding)
      if (kh == 1 && n < C) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(part + (i * 16 + r) * CP + n) = i < 2 ? acc_lo[i & 1] : acc_hi[i & 1];
      }
      __syncthreads();
      if (kh == 0 && n < C) {
        const bool vec_out = k == 0 && n + 3 < C && (ld & 3) == 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p = i * 16 + r;
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(gb + p * CP + n), a1 = *reinterpret_cast<const f32x4*>(part + p * CP + n);
          const f32x4 a2v = i < 2 ? acc_lo[i & 1] : acc_hi[i & 1];
          f32x4 v;
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = a0[q] + a2v[q] + a1[q];
          *reinterpret_cast<f32x4*>(gb + p * CP + n) = v;
          if (k == 0) {
            if (vec_out) *reinterpret_cast<f32x4*>(U.dx + (row0 + p) * ld + n) = v;
            else {
#pragma unroll
              for (int q = 0; q < 4; ++q) if (n + q < C) U.dx[(row0 + p) * ld + n + q] = v[q];
            }
          }
        }
      }
      __syncthreads();
      UNIT_STAMP(35 + 4 * (3 - k));
      UNIT_STAMP(6 + 6 * (3 - k));
      if (k > 0) {   // `part` aliased the dparams tile: restore the zero K padding phase (a) does not rewrite
        const int padc = N3S * 32 - N2;
        for (int e = tid; e < 64 * padc; e += kMcfThreads) {
          const int p = e / padc, j = N2 + e - p * padc;
          *reinterpret_cast<T*>(dp + p * dp_pitch + j * (int)sizeof(T)) = (T)0.f;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ inverse (sampling)
// x = unit^-1(y): ActNorm2^-1, D^-1, C^-1, ActNorm1^-1, B^-1, A^-1 in one launch, two samples per workgroup (a strip of 8
// positions per sample fills one 16-row matrix-core tile, as in mcf_inv_kernel).  A masked conv flow is inverted strip by
// strip (8 rows / columns, macow2.py:174-288): the conditioner of a strip only sees strips already reconstructed.  The
// state stays in LDS for the 4 x 8 strips; the weight fragments of the next layer are requested while the last strip of the
// current one is being processed.
template <typename T, bool WIDE>
__global__ __launch_bounds__(kMcfThreads) void macow_unit_inv_kernel(const UnitParams U) {
  constexpr int KS = K64<T>::value, E16 = ET<T>::E16, J1 = UC<WIDE>::J1;
  typedef typename ET<T>::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unit_kernarg_prefetch();
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * 2, nb = min(2, U.B - b0);
  const int C = U.C, N2 = 2 * C, ld = U.ld, H = U.H;
  constexpr int K2c = UC<WIDE>::N2S * 32;
  const int xs_pitch = U.Cp * (int)sizeof(T) + kTilePad;
  constexpr int a2_pitch = K2c * (int)sizeof(T) + kTilePad;
  unsigned char* xs = smem;                                        // T [2][64 + zero row][Cp]: reconstructed inputs (operand copy)
  unsigned char* a2 = xs + 2 * 65 * xs_pitch;                      // T [2][16][K2c]: the strip's [hidden | cond] operand, double-buffered
  unsigned char* cs = a2 + 2 * 16 * a2_pitch;                      // T [2][64][Cc]: ELU(cond) of the two samples
  float* yf = reinterpret_cast<float*>(cs + 2 * 64 * U.Cc * (int)sizeof(T));      // [2][64][C] fp32 state: y on entry of a layer, x on exit
  float* bias_s = yf + 2 * 64 * C;                                 // [4][2C]
  float* post_s = bias_s + 4 * N2;                                 // [4][2][C]: exp(log_scale) + 1e-8, bias of the ActNorms
  const long row0 = (long)b0 * 64;
  const int G2 = C >> 1;
  const int rows = nb * 64;
  const float inv_g2 = 1.f / (float)G2, inv_cch = 1.f / (float)(U.Cc / E16);     // (index arithmetic of the strip loop: no integer-division sequences)

  // prologue: small loads first (results return in issue order), then the weights of layer D
  f32x2 yin[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = tid + i * kMcfThreads;
    const int p = e / G2, c = (e - p * G2) * 2;
    if (e < rows * G2) yin[i] = *reinterpret_cast<const f32x2*>(U.L[3].x + (row0 + p) * ld + c);
  }
  float bias_v = 0.f, post_e = 0.f, post_b = 0.f;
  bool post_on = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) {      // static layer index: uniform pointers (see the forward kernel)
    if (tid >= q * N2 && tid < (q + 1) * N2) bias_v = U.L[q].bias2[tid - q * N2];
    if (U.L[q].post_ls && tid >= q * C && tid < (q + 1) * C) {
      post_e = U.L[q].post_ls[tid - q * C]; post_b = U.L[q].post_bias[tid - q * C]; post_on = true;
    }
  }
  McfW<T> wr;
  // The 1x1 conv's output columns are taken in the order (mu_c, s_c, mu_c+1, s_c+1, ...): wave w computes the raw shift AND scale of
  // channels 8w .. 8w+7, so that a lane's four accumulator values are (mu, s) of two channels of its row and the inversion of the
  // coupling runs straight from the registers -- no [16][2C] staging tile, no barrier between the contraction and the coupling
  // (the strip loop is a chain of barrier-separated phases: 128 strips x 3 barriers per launch).  Rows of the fragment-tiled operand
  // are picked per lane: fragment row n of wave w = operand row 8w + n/2 (+ C for odd n); rows of channels >= C read as zeros.
  const int w2_n2 = U.K2p / KS;
  int w2_voff;
  {
    const int lane_ = tid & 63, wave_ = tid >> 6, nloc = lane_ & 15, q = lane_ >> 4;
    const int ch = 8 * wave_ + (nloc >> 1), R_ = (nloc & 1) ? C + ch : ch;
    w2_voff = ch < C ? ((R_ >> 4) * w2_n2) * 1024 + (16 * q + (R_ & 15)) * 16 : kOob;
  }
  auto load_w2_pairs = [&](const void* W2) {
    const rsrc_t rs = make_rsrc(W2, ((2 * C + 15) & ~15) * U.K2p * (int)sizeof(T));
#pragma unroll
    for (int st = 0; st < UC<WIDE>::N2S; ++st) wr.w2[st][0] = buf_frag<T>(rs, w2_voff, st < w2_n2 ? st * 1024 : kOob);
  };
  unit_load_w1<T, WIDE>(wr, U.L[3].W1, U);
  load_w2_pairs(U.L[3].W2);
  for (int i = tid; i < (2 * 65 * xs_pitch + 2 * 16 * a2_pitch) / 16; i += kMcfThreads) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
  {   // ELU(cond) rows of both samples
    const int cchunks = U.Cc / E16;
    const T* condp = reinterpret_cast<const T*>(U.cond) + row0 * U.Cc;
    for (int e = tid; e < rows * cchunks; e += kMcfThreads)
      *reinterpret_cast<u32x4*>(cs + (long)e * 16) = *reinterpret_cast<const u32x4*>(condp + (long)e * E16);
  }
  if (ld > C) {                                     // pass-through channels
    const int R2 = (ld - C) >> 1;
    for (int e = tid; e < rows * R2; e += kMcfThreads) {
      const int p = e / R2, c = C + (e - p * R2) * 2;
      *reinterpret_cast<f32x2*>(U.L[0].y + (row0 + p) * ld + c) = *reinterpret_cast<const f32x2*>(U.L[3].x + (row0 + p) * ld + c);
    }
  }
  if (tid < 4 * N2) bias_s[tid] = bias_v;
  if (tid < 4 * C) {
    post_s[(tid / C) * 2 * C + tid % C] = post_on ? __expf(post_e) + 1e-8f : 1.f;      // macow2.py:520
    post_s[(tid / C) * 2 * C + C + tid % C] = post_b;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = tid + i * kMcfThreads;
    const int p = e / G2, c = (e - p * G2) * 2;
    if (e < rows * G2) *reinterpret_cast<f32x2*>(yf + p * C + c) = yin[i];
  }
  __syncthreads();
  UNIT_STAMP(0);
#pragma unroll 1
  for (int k = 3; k >= 0; --k) {
    const UnitLayer& Lk = U.L[k];
    const McfGeom g = mcf_geom(Lk.order);
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int lane = tl & 63, wave = tl >> 6, r = lane & 15, gq = lane >> 4;
    if (Lk.post_ls) {                               // the ActNorm behind this layer, inverted first
      const float* pe = post_s + k * 2 * C;
      for (int e = tl; e < rows * G2; e += kMcfThreads) {
        const int p = e / G2, c = (e - p * G2) * 2;
        f32x2 v = *reinterpret_cast<const f32x2*>(yf + p * C + c);
        v[0] = __fdividef(v[0] - pe[C + c], pe[c]); v[1] = __fdividef(v[1] - pe[C + c + 1], pe[c + 1]);
        *reinterpret_cast<f32x2*>(yf + p * C + c) = v;
      }
      __syncthreads();
    }
    const bool rows_first = Lk.order < 2, backwards = (Lk.order & 1);
    const float* bk = bias_s + k * N2;
    // conditioning rows of strip `st` behind the hidden columns of operand buffer st & 1: for the first strip here, for every
    // later one underneath the coupling phase of the strip before it (they used to head every strip: 550 of its ~5 400 cycles)
    auto stage_cond = [&](int st) {
      const int si_ = backwards ? 7 - st : st;
      unsigned char* dst = a2 + (st & 1) * 16 * a2_pitch;
      const int cchunks = U.Cc / E16;
      for (int e = tl; e < 16 * cchunks; e += kMcfThreads) {
        const int row = (int)(((float)e + 0.5f) * inv_cch), ch = e - row * cchunks;      // e / cchunks, exact for e < 2048, cchunks <= 16
        const int sidx = row >> 3, j = row & 7;
        const int pos = rows_first ? si_ * 8 + j : j * 8 + si_;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (sidx < nb) v = *reinterpret_cast<const u32x4*>(cs + ((sidx * 64 + pos) * cchunks + ch) * 16);
        *reinterpret_cast<u32x4*>(dst + row * a2_pitch + (H + ch * E16) * (int)sizeof(T)) = v;
      }
    };
    stage_cond(0);
    // One strip.  LAST (the layer's eighth strip, compile time): the next layer's weight fragments are re-requested piecewise -- a tap of the
    // shifted conv as soon as its matrix-core instructions have been issued, a K step of the 1x1 conv likewise -- instead of in two bursts
    // of 196 + 98 KB per workgroup behind the contractions, in whose ISSUE the eight waves sat for ~5 500 cycles per layer (the CU's
    // vector-memory pipeline is a FIFO).
    const bool reload = k > 0;
    const UnitLayer& Ln = U.L[k > 0 ? k - 1 : 0];
    const rsrc_t rs1n = make_rsrc(Ln.W1, ((U.H + 15) & ~15) * U.K1p * (int)sizeof(T));
    const rsrc_t rs2n = make_rsrc(Ln.W2, ((2 * C + 15) & ~15) * U.K2p * (int)sizeof(T));
    const int nks1 = U.K1p / KS;
    auto strip = [&](const int step, auto last_tag) {
      constexpr bool LAST = decltype(last_tag)::value;
      const int si = backwards ? 7 - step : step;
      unsigned char* a2b = a2 + (step & 1) * 16 * a2_pitch;
      UNIT_STAMP(1 + 6 * step + 0 + (k == 3 ? 0 : 1000));
      UNIT_STAMP(1 + 6 * step + 1 + (k == 3 ? 0 : 1000));
      {   // hidden = ELU(shifted conv of the strips reconstructed so far)
        const int sidx = r >> 3, j = r & 7;
        const int pos = rows_first ? si * 8 + j : j * 8 + si;
        const unsigned char* tile = xs + sidx * 65 * xs_pitch;
        const unsigned char* zrow = tile + 64 * xs_pitch;
        f32x4 acc[J1];
#pragma unroll
        for (int jj = 0; jj < J1; ++jj) acc[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 6; ++tap) {
          const unsigned char* src = tap_src_fwd(tile, zrow, xs_pitch, g, pos, tap) + E16 * gq * (int)sizeof(T);
#pragma unroll
          for (int st = 0; st < UC<WIDE>::CS; ++st) {
            const frag_t fa = *reinterpret_cast<const frag_t*>(src + st * KS * (int)sizeof(T));
#pragma unroll
            for (int jj = 0; jj < J1; ++jj) mma64(fa, wr.w1[tap][st][jj], acc[jj]);
          }
          if constexpr (LAST) {
            if (reload) {
#pragma unroll
              for (int st = 0; st < UC<WIDE>::CS; ++st)
#pragma unroll
                for (int jj = 0; jj < J1; ++jj)
                  wr.w1[tap][st][jj] = buf_frag<T>(rs1n, (wave + kMcfWaves * jj) * nks1 * 1024 + lane * 16, (tap * UC<WIDE>::CS + st) * 1024);
            }
          }
        }
#pragma unroll
        for (int jj = 0; jj < J1; ++jj) {
          const int n = (wave + kMcfWaves * jj) * 16 + 4 * gq;
          if (n < H) {
            typename Pack4<T>::type tv;
#pragma unroll
            for (int q = 0; q < 4; ++q) tv[q] = ET<T>::from_f32(fast_elu(acc[jj][q]));
            *reinterpret_cast<typename Pack4<T>::type*>(a2b + r * a2_pitch + n * (int)sizeof(T)) = tv;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      UNIT_STAMP(1 + 6 * step + 2 + (k == 3 ? 0 : 1000));
      __syncthreads();
      UNIT_STAMP(1 + 6 * step + 3 + (k == 3 ? 0 : 1000));
      {   // raw (mu, s) of the 16 rows, two channels per lane, and x = (y - mu) / (scale + 1e-12) (macow_utils.py:61-66) straight from
          // the accumulator; the strip joins the operand tile
        const int c = 8 * wave + 2 * gq;                       // this lane: row r, channels c, c + 1
        const int sidx = r >> 3, j = r & 7;
        const int pos = rows_first ? si * 8 + j : j * 8 + si;
        const bool live = c < C && sidx < nb;
        f32x2 yv = {0.f, 0.f}, bm = {0.f, 0.f}, bs = {0.f, 0.f};
        if (live) {                                            // (requested ahead of the contraction)
          yv = *reinterpret_cast<const f32x2*>(yf + (sidx * 64 + pos) * C + c);
          bm = *reinterpret_cast<const f32x2*>(bk + c); bs = *reinterpret_cast<const f32x2*>(bk + C + c);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < UC<WIDE>::N2S; ++st) {
          const frag_t fa = *reinterpret_cast<const frag_t*>(a2b + r * a2_pitch + (st * KS + E16 * gq) * (int)sizeof(T));
          mma64(fa, wr.w2[st][0], acc);
          if constexpr (LAST) {
            if (reload) wr.w2[st][0] = buf_frag<T>(rs2n, w2_voff, st < w2_n2 ? st * 1024 : kOob);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        UNIT_STAMP(1 + 6 * step + 4 + (k == 3 ? 0 : 1000));
        UNIT_STAMP(1 + 6 * step + 5 + (k == 3 ? 0 : 1000));
        if (live) {
          f32x2 xv;
          xv[0] = __fdividef(yv[0] - (acc[0] + bm[0]), fast_scale(acc[1] + bs[0]) + 1e-12f);
          xv[1] = __fdividef(yv[1] - (acc[2] + bm[1]), fast_scale(acc[3] + bs[1]) + 1e-12f);
          *reinterpret_cast<f32x2*>(yf + (sidx * 64 + pos) * C + c) = xv;
          bf16x2 tv; tv[0] = ET<T>::from_f32(xv[0]); tv[1] = ET<T>::from_f32(xv[1]);
          *reinterpret_cast<bf16x2*>(xs + (sidx * 65 + pos) * xs_pitch + c * (int)sizeof(T)) = tv;
        }
      }
      if constexpr (!LAST) stage_cond(step + 1);  // (buffer (step + 1) & 1: last read two barriers ago)
      __syncthreads();
    };
#pragma unroll 1
    for (int step = 0; step < 7; ++step) strip(step, std::false_type{});
    strip(7, std::true_type{});
    UNIT_STAMP(49 + (3 - k));
  }
  for (int e = tid; e < rows * G2; e += kMcfThreads) {
    const int p = e / G2, c = (e - p * G2) * 2;
    *reinterpret_cast<f32x2*>(U.L[0].y + (row0 + p) * ld + c) = *reinterpret_cast<const f32x2*>(yf + p * C + c);
  }
}

static int unit_params(UnitParams& U, const ipoke_mcf_desc* d, int dtype, bool bwd) {
  IPK_REQUIRE(d, "null descriptor array");
  IPK_REQUIRE(dtype == IPOKE_BF16, "the fused MaCowUnit kernels take bf16 matrix-core inputs; f32 runs the per-layer kernels");
  const int C = d[0].C, ld = d[0].ld, Cc = d[0].Cc, B = d[0].B;
  IPK_REQUIRE(C >= 2 && C <= 64 && C % 2 == 0 && ld >= C && ld % 2 == 0, "even channel counts up to 64, even state pitch");
  IPK_REQUIRE(Cc % 8 == 0 && Cc <= 128 && 4 * C + Cc <= kW2Steps * 32, "conditioning width: multiple of 8, <= 128, 4C + Cc <= 384");
  IPK_REQUIRE(B >= 1 && d[0].cond, "bad batch / null cond");
  std::memset(&U, 0, sizeof(U));
  U.ld = ld; U.C = C; U.B = B; U.Cc = Cc; U.cond = d[0].cond;
  U.H = 4 * C; U.Cp = round_up(C, 32); U.K1p = 6 * U.Cp; U.K2p = round_up(U.H + Cc, 32); U.K3p = round_up(2 * C, 32);
  U.Hq = round_up(U.H, 32);
  for (int k = 0; k < 4; ++k) {
    const ipoke_mcf_desc& s = d[k];
    IPK_REQUIRE(s.C == C && s.ld == ld && s.Cc == Cc && s.B == B && s.cond == d[0].cond, "the four layers of a unit share C, ld, cond, B");
    IPK_REQUIRE(s.order >= 0 && s.order <= 3, "order is 0..3 (A..D)");
    IPK_REQUIRE((s.post_log_scale == nullptr) == (s.post_bias == nullptr), "fused ActNorm needs log_scale and bias");
    UnitLayer& L = U.L[k];
    L.W1 = s.W1; L.W2 = s.W2; L.bias2 = s.bias2; L.W1T = s.W1T; L.W2T = s.W2T; L.y = s.y; L.a2_save = s.a2_save;
    L.scale_save = s.scale_save; L.ld_slot = s.logdet_slot; L.post_ls = s.post_log_scale; L.post_bias = s.post_bias;
    L.x = s.x; L.y_post = s.y_post; L.post_part = s.post_part; L.dparams_save = s.dparams_save; L.dc_save = s.dc_save;
    L.dbias_part = s.dbias_part; L.order = s.order; L.x_op = bwd ? s.x_op_save : nullptr;
    L.zc = bwd ? nullptr : s.zc_out; L.zc_off = s.zc_off; L.zc_stride = s.zc_stride; L.zc_cin = s.zc_cin; L.zc_ld = s.zc_ld;
    if (L.zc) IPK_REQUIRE(k == 3 && (s.zc_stride == 1 || s.zc_stride == 2) && s.zc_cin >= 1 && s.zc_ld >= s.zc_cin && s.zc_ld % 2 == 0 && s.zc_off >= 0 &&
                          s.zc_off + (s.zc_cin - 1) * s.zc_stride < C, "conditioning operand: layer 3 only, stride 1 or 2, columns inside the state");
    if (!bwd) IPK_REQUIRE(s.W1 && s.W2 && s.bias2, "null forward operand");
    else {
      IPK_REQUIRE(s.W1T && s.W2T && s.x && s.a2_save && s.scale_save && s.dparams_save && s.dc_save, "null backward operand");
      IPK_REQUIRE(!s.post_log_scale || s.y_post, "the fused ActNorm backward needs the saved output");
    }
  }
  U.x = d[0].x; U.dy = d[3].dy; U.dld = d[0].dld; U.dx = d[0].dx;
  U.slot_w = d[0].rows_per_block > 0 ? 64 / d[0].rows_per_block : 1;
  if (d[0].split > 1) {
    IPK_REQUIRE(d[0].split == 2 || d[0].split == 4, "split: 1 (one workgroup per sample), 2 or 4");
    IPK_REQUIRE(d[0].xchg != nullptr, "row-split unit launch needs the exchange scratch (ipoke_macow_unit_xchg_bytes, zero-initialised)");
    U.xchg = reinterpret_cast<unsigned long long*>(d[0].xchg); U.xchg_stride = kUnitXchgStride;
  }
  return IPOKE_OK;
}

}  // namespace ipoke

using namespace ipoke;

#ifdef IPOKE_UNIT_STAMPS
static unsigned long long* g_stamps = nullptr;
extern "C" void ipoke_macow_unit_set_stamps(void* p) { g_stamps = reinterpret_cast<unsigned long long*>(p); }
#endif

extern "C" int ipoke_macow_unit_supported(int C, int Cc, int dtype) {
  return dtype == IPOKE_BF16 && C >= 2 && C <= 64 && C % 2 == 0 && Cc % 8 == 0 && Cc <= 128 && 4 * C + Cc <= kW2Steps * 32;
}

extern "C" int64_t ipoke_macow_unit_xchg_bytes(int B, int split) {
  if (split <= 1 || B < 1) return 0;
  return 256 + (int64_t)B * 4 * split * kUnitXchgStride * 8;
}

extern "C" int ipoke_macow_unit_fwd(const ipoke_mcf_desc* d4, int dtype, void* stream) {
  UnitParams U;
  int rc = unit_params(U, d4, dtype, false); if (rc) return rc;
  IPK_REQUIRE(U.x && U.L[3].y, "null input / output state");
#ifdef IPOKE_UNIT_STAMPS
  U.stamps = g_stamps;
#endif
  const bool wide = U.Cp > 32;
  const size_t lds = (size_t)65 * (U.Cp * 2 + kTilePad) + (size_t)64 * ((wide ? 384 : 256) * 2 + kTilePad) + (size_t)64 * (2 * U.C + 4) * 4 +
                     (size_t)64 * U.C * 4 + (size_t)(8 * U.C + 8 * U.C + 8) * 4;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  TimedScope ts(IPOKE_TAG_UNIT_FWD, s);
  if (d4[0].split > 1) return unit_fwd_split_launch(U, d4[0].split, s);
  if (wide) {
    rc = ensure_lds<macow_unit_fwd_kernel<bf16_t, true>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((macow_unit_fwd_kernel<bf16_t, true>), dim3(U.B), dim3(kMcfThreads), lds, s, U);
  } else {
    rc = ensure_lds<macow_unit_fwd_kernel<bf16_t, false>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((macow_unit_fwd_kernel<bf16_t, false>), dim3(U.B), dim3(kMcfThreads), lds, s, U);
  }
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

extern "C" int ipoke_macow_unit_bwd(const ipoke_mcf_desc* d4, int dtype, void* stream) {
  UnitParams U;
  int rc = unit_params(U, d4, dtype, true); if (rc) return rc;
  IPK_REQUIRE(U.dy && U.dld && U.dx, "null gradient tensor");
  std::memset(&U.pair, 0, sizeof(U.pair));
  if (d4[0].pair) {
    const ipoke_unit_pair_desc& q = *d4[0].pair;
    IPK_REQUIRE(d4[0].split > 1, "the ActNorm + coupling pair runs in the row-split launches only");
    IPK_REQUIRE(q.Cp >= 1 && q.Cp <= 32 && q.t_stride >= 1 && q.t_off >= 0 && q.t_off + (q.Cp - 1) * q.t_stride < U.C && q.ldp >= 2 * q.Cp,
                "pair: the coupling must transform channels of the unit's window");
    IPK_REQUIRE(q.x0 && q.scale && q.dparams && q.dx && q.dx != U.dy, "pair: null tensor");
    IPK_REQUIRE(q.dx == U.dx, "pair: the pass-through columns (>= C) reach the pair's dx through the launch's own dx: pass the same buffer");
    IPK_REQUIRE(!q.an_log_scale || (q.an_x && q.an_part), "pair: the ActNorm's saved input / partials are missing");
    U.pair.an_ls = q.an_log_scale; U.pair.an_idx = q.an_idx; U.pair.an_x = q.an_x; U.pair.an_part = q.an_part;
    U.pair.x0 = q.x0; U.pair.scale = q.scale; U.pair.dparams = q.dparams; U.pair.dbias_part = q.dbias_part; U.pair.dx = q.dx;
    U.pair.Cp = q.Cp; U.pair.t_off = q.t_off; U.pair.t_stride = q.t_stride; U.pair.ldp = q.ldp; U.pair.on = 1;
  }
#ifdef IPOKE_UNIT_STAMPS
  U.stamps = g_stamps;
#endif
  const bool wide = U.Cp > 32;
  const size_t dp_bytes = (size_t)64 * ((wide ? 128 : 64) * 2 + kTilePad);
  const size_t CP = unit_gb_pitch(U.C);
  const size_t lds = dp_bytes + (size_t)65 * ((wide ? 256 : 128) * 2 + kTilePad) + (size_t)64 * CP * 4 + (2 * 4096 + 2 * 512) * 4;
  IPK_REQUIRE((size_t)64 * CP * 4 <= dp_bytes, "tap-half partials must fit the dparams tile");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  TimedScope ts(IPOKE_TAG_UNIT_BWD, s);
  if (d4[0].split > 1) return unit_bwd_split_launch(U, d4[0].split, s);
  if (wide) {
    rc = ensure_lds<macow_unit_bwd_kernel<bf16_t, true>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((macow_unit_bwd_kernel<bf16_t, true>), dim3(U.B), dim3(kMcfThreads), lds, s, U);
  } else {
    rc = ensure_lds<macow_unit_bwd_kernel<bf16_t, false>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((macow_unit_bwd_kernel<bf16_t, false>), dim3(U.B), dim3(kMcfThreads), lds, s, U);
  }
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

/* d4 in forward order; d4[3].x = the unit's OUTPUT state (input of the inverse), d4[0].y = the reconstructed unit input. */
extern "C" int ipoke_macow_unit_inv(const ipoke_mcf_desc* d4, int dtype, void* stream) {
  UnitParams U;
  int rc = unit_params(U, d4, dtype, false); if (rc) return rc;
  IPK_REQUIRE(U.L[3].x && U.L[0].y && U.L[3].x != U.L[0].y, "null / aliased state");
#ifdef IPOKE_UNIT_STAMPS
  U.stamps = g_stamps;
#endif
  const bool wide = U.Cp > 32;
  const size_t lds = (size_t)2 * 65 * (U.Cp * 2 + kTilePad) + (size_t)2 * 16 * ((wide ? 384 : 256) * 2 + kTilePad) + (size_t)2 * 64 * U.Cc * 2 +
                     (size_t)2 * 64 * U.C * 4 + (size_t)(8 * U.C + 8 * U.C) * 4;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int grid = (U.B + 1) / 2;
  TimedScope ts(IPOKE_TAG_UNIT_INV, s);
  // four masked-conv flows per sample: 64 positions x (6-tap shifted conv C -> 4C, 1x1 conv (4C + Cc) -> 2C); weights once, state in + out
  ts.annotate(0, 4.0 * U.B * 64.0 * (64.0 * U.C * U.C + 4.0 * U.Cc * U.C), 4.0 * (32.0 * U.C * U.C + 2.0 * U.Cc * U.C) * 2.0 + 2.0 * U.B * 64.0 * U.C * 4.0);
  if (wide) {
    rc = ensure_lds<macow_unit_inv_kernel<bf16_t, true>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((macow_unit_inv_kernel<bf16_t, true>), dim3(grid), dim3(kMcfThreads), lds, s, U);
  } else {
    rc = ensure_lds<macow_unit_inv_kernel<bf16_t, false>>(lds); if (rc) return rc;
    hipLaunchKernelGGL((macow_unit_inv_kernel<bf16_t, false>), dim3(grid), dim3(kMcfThreads), lds, s, U);
  }
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
