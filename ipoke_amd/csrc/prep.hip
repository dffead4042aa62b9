// Multi-tensor weight preparation: one launch converts every fp32 master weight of the flow (PyTorch
// [out][in][kh][kw] layout, as in the reference's state dict) into the K-contiguous, zero padded
// matrix-core operand layouts ("shadows") used by the GEMM kernels, folding the weight-norm
// reparameterisation w = g * v / ||v||  (reference macow_utils.py:225-229, nn.utils.weight_norm).
// Also: weight-norm row statistics and the weight-norm backward (dW_eff -> dg, dv), multi-tensor.
#include "common.h"

namespace ipoke {

// One job = one weight tensor W[n][k][tap] of the state dict (tap stride 1; n = output channel, k = input channel)
// and BOTH operand layouts derived from it, so the fp32 source is read once, coalesced, through an LDS tile:
//   A[n][tap*A_inner_pad + k]   (rows padded to A_rows_pad)        -- forward operand, K contiguous
//   B[k][tap*B_inner_pad + n]   (rows padded to B_rows_pad)        -- data-gradient operand (transposed)
// value = W * scale[n] (weight norm) ; zero outside n_real x k_real and for B rows >= B_rows_real.
struct RelayoutJob {
  long src_off;        // floats, into params
  long s_n, s_k;       // source strides of n and k (floats)
  long dstA, dstB;     // elements, into the shadow buffer
  long scale_off;      // floats into the wn-scale buffer (indexed by n), or -1
  int taps, n_real, k_real;
  int A_rows_pad, A_inner_pad, B_rows_pad, B_inner_pad, B_rows_real;
  int tile, tiles_k;   // tile edge (64 for 1x1, 32 otherwise), tiles along k
  int block_start;     // first block of this job
  int frag_tiled;      // 1: both operands in the fragment-tiled order of the MCF kernels (tiled_offset below)
};

// Fragment-tiled order of the masked-conv-flow weight operands: the matrix [rows][ld] is cut into tiles of 16 rows x 64 bytes
// of K (one matrix-core B fragment of one wave: lane l = 16 * (K chunk) + row holds 16 bytes), tile (rb, ks) at
// (rb * ld/KS + ks) KB, lane-linear inside.  One wave-wide 16-byte load then reads ONE contiguous KB instead of sixteen
// 64-byte row segments (half-used cache lines: 24 B/clk/CU measured; the weight stream of a layer bounds the MCF kernels).
template <typename T>
__device__ __forceinline__ long tiled_offset(int row, int col, int ld) {
  constexpr int KS = 64 / (int)sizeof(T), E16 = 16 / (int)sizeof(T);
  const int rb = row >> 4, r = row & 15, ks = col / KS, cq = col - ks * KS;
  return ((long)rb * (ld / KS) + ks) * (16 * KS) + (cq / E16) * (16 * E16) + r * E16 + (cq % E16);
}

template <typename T> struct RPack4;
template <> struct RPack4<bf16_t> { typedef __attribute__((ext_vector_type(4))) __bf16 type; };
template <> struct RPack4<float> { typedef f32x4 type; };

// one tile of one job; TAPS and the tile edge are compile-time so that the index arithmetic is shifts and constant divisions
template <typename T, int TAPS, int TE>
__device__ __forceinline__ void relayout_tile(const RelayoutJob& j, const float* __restrict__ params, T* __restrict__ shadow,
                                              const float* __restrict__ wn_scale, int n0, int k0, float* tile) {
  constexpr int TP = TE + 1;
  constexpr int total = TE * TE * TAPS;
  typedef typename RPack4<T>::type pack_t;
  const bool k_mid = j.s_k < j.s_n;
  if (TAPS == 1 && j.s_k == 1 && ((j.s_n | j.src_off) & 3) == 0) {
    // 1x1 weights, k contiguous: 16-byte loads
    for (int e = threadIdx.x; e < total / 4; e += 256) {
      const int kl = (e % (TE / 4)) * 4, nl = e / (TE / 4);
      const int n = n0 + nl, k = k0 + kl;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n < j.n_real && k < j.k_real) {
        if (k + 3 < j.k_real) {
          v = *reinterpret_cast<const f32x4*>(params + j.src_off + (long)n * j.s_n + k);
        } else {
          for (int q = 0; q < 4; ++q) if (k + q < j.k_real) v[q] = params[j.src_off + (long)n * j.s_n + k + q];
        }
        if (j.scale_off >= 0) { const float sc = wn_scale[j.scale_off + n]; v[0] *= sc; v[1] *= sc; v[2] *= sc; v[3] *= sc; }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) tile[nl * TP + kl + q] = v[q];
    }
  } else {
    // read in source order: tap fastest, then whichever of (n, k) has the smaller stride
    for (int e = threadIdx.x; e < total; e += 256) {
      const int t = e % TAPS, rest = e / TAPS;
      const int m = rest % TE, sl = rest / TE;
      const int nl = k_mid ? sl : m, kl = k_mid ? m : sl;
      const int n = n0 + nl, k = k0 + kl;
      float v = 0.f;
      if (n < j.n_real && k < j.k_real) {
        v = params[j.src_off + (long)n * j.s_n + (long)k * j.s_k + t];
        if (j.scale_off >= 0) v *= wn_scale[j.scale_off + n];
      }
      tile[(nl * TAPS + t) * TP + kl] = v;
    }
  }
  __syncthreads();
  const int ldA = TAPS * j.A_inner_pad, ldB = TAPS * j.B_inner_pad;
  // A: rows n, columns (tap, k); four consecutive k per thread (all pads are multiples of 4)
  for (int e = threadIdx.x; e < total / 4; e += 256) {
    const int kl = (e % (TE / 4)) * 4, rest = e / (TE / 4);
    const int t = rest % TAPS, nl = rest / TAPS;
    const int n = n0 + nl, k = k0 + kl;
    if (n < j.A_rows_pad && k < j.A_inner_pad) {
      pack_t o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = ET<T>::from_f32(tile[(nl * TAPS + t) * TP + kl + q]);
      const long off = j.frag_tiled ? tiled_offset<T>(n, t * j.A_inner_pad + k, ldA) : (long)n * ldA + t * j.A_inner_pad + k;
      *reinterpret_cast<pack_t*>(shadow + j.dstA + off) = o;
    }
  }
  // B: rows k, columns (tap, n); four consecutive n per thread (dstB < 0: the tensor has no transposed operand)
  for (int e = threadIdx.x; j.dstB >= 0 && e < total / 4; e += 256) {
    const int nl = (e % (TE / 4)) * 4, rest = e / (TE / 4);
    const int t = rest % TAPS, kl = rest / TAPS;
    const int n = n0 + nl, k = k0 + kl;
    if (k < j.B_rows_pad && n < j.B_inner_pad) {
      pack_t o;
      const bool live = k < j.B_rows_real;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = ET<T>::from_f32(live ? tile[((nl + q) * TAPS + t) * TP + kl] : 0.f);
      const long off = j.frag_tiled ? tiled_offset<T>(k, t * j.B_inner_pad + n, ldB) : (long)k * ldB + t * j.B_inner_pad + n;
      *reinterpret_cast<pack_t*>(shadow + j.dstB + off) = o;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void relayout_kernel(const float* __restrict__ params, T* __restrict__ shadow,
                                                       const float* __restrict__ wn_scale, const RelayoutJob* __restrict__ jobs,
                                                       int njobs, const int* __restrict__ block_job, int block_base, int nblocks) {
  __shared__ __attribute__((aligned(16))) float tile[32 * 33 * 9];   // [n_l][tap][k_l], k pitch TE+1 (>= 64*65)
  // Grid-stride over the tiles [block_base, block_base + nblocks) so that the grid can be capped (IPOKE_RELAYOUT_BLOCKS).  Measured,
  // round 3: the refresh of a slice is the worst neighbour a chain kernel can have -- a conv2 GEMM that overlaps it takes 72-170 us
  // instead of 24, a fused MaCowUnit backward 193 instead of 59 (profiles/r03_overlap_before.txt: one workgroup per tile, four per CU
  // by LDS, keeps every CU full for the ~300 us of a slice) -- and yet capping it LOSES: 62.3 ms per step uncapped against 62.8 / 63.7 /
  // 65.3 / 65.0 / 76.9 at 1024 / 512 / 256 / 128 / 64 workgroups.  Short and brutal beats long and mild; the default stays uncapped.
  for (int bid = (int)blockIdx.x + block_base; bid < block_base + nblocks; bid += (int)gridDim.x) {
    int lo = 0;
    if (block_job) {
      lo = block_job[bid];
    } else {                     // locate the job of this block (block_start is ascending)
      int hi = njobs - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block_start <= bid) lo = mid; else hi = mid - 1;
      }
    }
    const RelayoutJob j = jobs[lo];
    const int t_id = bid - j.block_start;
    const int n0 = (t_id / j.tiles_k) * j.tile, k0 = (t_id % j.tiles_k) * j.tile;
    if (j.taps == 1) relayout_tile<T, 1, 64>(j, params, shadow, wn_scale, n0, k0, tile);
    else if (j.taps == 9) relayout_tile<T, 9, 32>(j, params, shadow, wn_scale, n0, k0, tile);
    else relayout_tile<T, 6, 32>(j, params, shadow, wn_scale, n0, k0, tile);
    __syncthreads();             // the tile is reused by the next iteration
  }
}

// weight-norm rows: scale[n] = g[n] / ||v[n]||, inv_norm[n] = 1/||v[n]||.  One wave per row.
struct WnJob { long v_off, g_off, out_off; int rows, K; int row_start; };

__global__ void wn_scale_kernel(const float* __restrict__ params, float* __restrict__ scale, float* __restrict__ inv_norm,
                                const WnJob* __restrict__ jobs, int njobs, int row_base, int total_rows) {
  const int grow = row_base + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (grow >= row_base + total_rows) return;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].row_start <= grow) lo = mid; else hi = mid - 1;
  }
  const WnJob j = jobs[lo];
  const int row = grow - j.row_start, lane = threadIdx.x & 63;
  const float* v = params + j.v_off + (long)row * j.K;
  float s = 0.f;
  if ((j.K & 3) == 0 && (((j.v_off + (long)row * j.K)) & 3) == 0) {       // 16-byte loads, two in flight per lane
    const f32x4* v4 = reinterpret_cast<const f32x4*>(v);
    const int n4 = j.K >> 2;
    int k = lane;
    for (; k + 64 < n4; k += 128) {
      const f32x4 a = v4[k], b = v4[k + 64];
      s += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3] + b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
    }
    if (k < n4) { const f32x4 a = v4[k]; s += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3]; }
  } else {
    for (int k = lane; k < j.K; k += 64) s += v[k] * v[k];
  }
  s = wave_sum(s);
  if (lane == 0) {
    const float nrm = sqrtf(s);
    scale[j.out_off + row] = params[j.g_off + row] / nrm;
    inv_norm[j.out_off + row] = 1.f / nrm;
  }
}

// in place on the gradient buffer: grads[v_off..] holds dW_eff on entry and dv on exit; dg is written.
__global__ void wn_bwd_kernel(const float* __restrict__ params, float* __restrict__ grads, const float* __restrict__ inv_norm,
                              const WnJob* __restrict__ jobs, int njobs, int row_base, int total_rows) {
  const int grow = row_base + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (grow >= row_base + total_rows) return;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].row_start <= grow) lo = mid; else hi = mid - 1;
  }
  const WnJob j = jobs[lo];
  const int row = grow - j.row_start, lane = threadIdx.x & 63;
  const float* v = params + j.v_off + (long)row * j.K;
  float* dw = grads + j.v_off + (long)row * j.K;
  const bool vec = (j.K & 3) == 0 && (((j.v_off + (long)row * j.K)) & 3) == 0;
  float dot = 0.f;
  if (vec && (j.K >> 2) <= 192) {
    // short rows (3x3 kernels on <= 64 channels, the masked convolutions: most rows of a piece): the row of v and of dW_eff stays in
    // registers between the dot product and the update instead of being read twice (8 of this kernel's 20 bytes per parameter; it
    // runs on the optimizer's queue, which bounds the backward pass).  Same products; the per-lane partial sums associate differently
    // from the long-row loop's (one expression per 8 / 4 products here), so the two paths agree to fp32 round-off, not bit for bit --
    // a given row always takes the same path, which is what run-to-run reproducibility needs.
    const f32x4* v4 = reinterpret_cast<const f32x4*>(v);
    f32x4* d4 = reinterpret_cast<f32x4*>(dw);
    const int n4 = j.K >> 2;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const bool h0 = lane < n4, h1 = lane + 64 < n4, h2 = lane + 128 < n4;
    const f32x4 a0 = h0 ? v4[lane] : z4, b0 = h0 ? d4[lane] : z4, a1 = h1 ? v4[lane + 64] : z4, b1 = h1 ? d4[lane + 64] : z4;
    const f32x4 a2 = h2 ? v4[lane + 128] : z4, b2 = h2 ? d4[lane + 128] : z4;
    if (h1) dot += a0[0] * b0[0] + a0[1] * b0[1] + a0[2] * b0[2] + a0[3] * b0[3] + a1[0] * b1[0] + a1[1] * b1[1] + a1[2] * b1[2] + a1[3] * b1[3];
    else if (h0) dot += a0[0] * b0[0] + a0[1] * b0[1] + a0[2] * b0[2] + a0[3] * b0[3];
    if (h2) dot += a2[0] * b2[0] + a2[1] * b2[1] + a2[2] * b2[2] + a2[3] * b2[3];
    dot = wave_sum(dot);
    const float inv = inv_norm[j.out_off + row], g = params[j.g_off + row];
    const float a = g * inv, bcoef = g * dot * inv * inv * inv;
    f32x4 o0, o1, o2;
#pragma unroll
    for (int q = 0; q < 4; ++q) { o0[q] = a * b0[q] - bcoef * a0[q]; o1[q] = a * b1[q] - bcoef * a1[q]; o2[q] = a * b2[q] - bcoef * a2[q]; }
    if (h0) d4[lane] = o0;
    if (h1) d4[lane + 64] = o1;
    if (h2) d4[lane + 128] = o2;
    if (lane == 0) grads[j.g_off + row] = dot * inv;
    return;
  }
  if (vec) {
    const f32x4* v4 = reinterpret_cast<const f32x4*>(v);
    const f32x4* d4 = reinterpret_cast<const f32x4*>(dw);
    const int n4 = j.K >> 2;
    int k = lane;
    for (; k + 64 < n4; k += 128) {
      const f32x4 a0 = v4[k], b0 = d4[k], a1 = v4[k + 64], b1 = d4[k + 64];
      dot += a0[0] * b0[0] + a0[1] * b0[1] + a0[2] * b0[2] + a0[3] * b0[3] + a1[0] * b1[0] + a1[1] * b1[1] + a1[2] * b1[2] + a1[3] * b1[3];
    }
    if (k < n4) { const f32x4 a0 = v4[k], b0 = d4[k]; dot += a0[0] * b0[0] + a0[1] * b0[1] + a0[2] * b0[2] + a0[3] * b0[3]; }
  } else {
    for (int k = lane; k < j.K; k += 64) dot += dw[k] * v[k];
  }
  dot = wave_sum(dot);
  const float inv = inv_norm[j.out_off + row], g = params[j.g_off + row];
  const float a = g * inv, bcoef = g * dot * inv * inv * inv;
  if (vec) {
    const f32x4* v4 = reinterpret_cast<const f32x4*>(v);
    f32x4* d4 = reinterpret_cast<f32x4*>(dw);
    const int n4 = j.K >> 2;
    int k = lane;
    for (; k + 64 < n4; k += 128) {
      const f32x4 a0 = v4[k], a1 = v4[k + 64];
      f32x4 b0 = d4[k], b1 = d4[k + 64];
#pragma unroll
      for (int q = 0; q < 4; ++q) { b0[q] = a * b0[q] - bcoef * a0[q]; b1[q] = a * b1[q] - bcoef * a1[q]; }
      d4[k] = b0; d4[k + 64] = b1;
    }
    if (k < n4) {
      const f32x4 a0 = v4[k]; f32x4 b0 = d4[k];
#pragma unroll
      for (int q = 0; q < 4; ++q) b0[q] = a * b0[q] - bcoef * a0[q];
      d4[k] = b0;
    }
  } else {
    for (int k = lane; k < j.K; k += 64) dw[k] = a * dw[k] - bcoef * v[k];
  }
  if (lane == 0) grads[j.g_off + row] = dot * inv;
}


// ---------------------------------------------------------------------------------------------------------------------
// Adam-amsgrad fused with the shadow refresh of plain 1x1 weights (conv2 of every coupling net: [hidden][hidden], 73 % of the
// flow's parameters).  These tensors need no weight norm, their forward operand is the bf16 cast of the tensor in place and
// their data-gradient operand its transpose -- so the optimizer, which holds the updated p in registers, writes both and
// `relayout` never re-reads them (per step at z = 64: 3.6 GB less HBM read, 215 relayout jobs less).  One 64 x 64 tile per
// iteration of a persistent workgroup: a thread owns 8 consecutive k of two rows (two 16-byte loads per array, all 20 issued
// before the arithmetic), stores p / m / v / v_max and the straight operand row segment, and the tile meets in LDS once for
// the transposed operand.  Same arithmetic as adam_amsgrad_kernel (adam_amsgrad_update).
struct AdamTileJob { long src_off, dstA, dstB; int N, K, tile_start, pad; };

template <typename T> struct Out8;
template <> struct Out8<bf16_t> {
  static __device__ __forceinline__ void store(bf16_t* dst, const float* x) {
    bf16x8 o;
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = (bf16_t)x[q];
    *reinterpret_cast<bf16x8*>(dst) = o;
  }
};
template <> struct Out8<float> {
  static __device__ __forceinline__ void store(float* dst, const float* x) {
    *reinterpret_cast<f32x4*>(dst) = f32x4{x[0], x[1], x[2], x[3]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{x[4], x[5], x[6], x[7]};
  }
};

// Optimizer state is a pure stream (every byte read once and written once per step, 36 bytes per parameter): the accesses carry the
// non-temporal hint so that they do not displace the chain's weights / activations from L2 and the memory-side cache while the update
// runs underneath the backward pass (c2: 57.4 -> 56.6 ms, 58.3 -> 57.15 on a slower box; the hint on the bf16 operand stores or on the
// weight-gradient GEMM's fp32 stores gave nothing -- both are read again soon enough to be served from the cache).
template <bool NT> __device__ __forceinline__ f32x4 ld_stream(const float* q) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q));
  return *reinterpret_cast<const f32x4*>(q);
}
template <bool NT> __device__ __forceinline__ void st_stream(float* q, f32x4 v) {
  if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(q));
  else *reinterpret_cast<f32x4*>(q) = v;
}
#ifdef IPOKE_ADAM_TEMPORAL          // developer A/B build
static constexpr bool kNtCast = false, kNtSeg = false;
#else
static constexpr bool kNtCast = true, kNtSeg = true;
#endif

template <typename T>
__global__ __launch_bounds__(256) void adam_shadow_tile_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                               float* __restrict__ v, float* __restrict__ vmax, T* __restrict__ shadow,
                                                               const AdamTileJob* __restrict__ jobs, int njobs, int tile_begin,
                                                               int ntiles, AdamHyper h) {
  constexpr int TP = 65;
  __shared__ float tile[64 * TP];
  const int tid = threadIdx.x;
  const int G = (int)gridDim.x, end = tile_begin + ntiles;
  struct Regs { f32x4 pp[4], gg[4], mm[4], vv[4], vx[4]; long off[2]; long dstA, dstB; int N, K, n0, k0; };
  // the 20 sixteen-byte loads of a tile; issued one tile AHEAD of the arithmetic (two register sets), so that a workgroup always
  // has a tile's worth of HBM requests in flight while it updates, stores and transposes the previous one -- without this the
  // persistent grid streamed at ~3.2 TB/s against the linear kernel's 5.7
  auto load = [&](int t, Regs& R) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].tile_start <= t) lo = mid; else hi = mid - 1;
    }
    const AdamTileJob j = jobs[lo];
    const int lt = t - j.tile_start, tiles_k = j.K >> 6;
    R.n0 = (lt / tiles_k) << 6; R.k0 = (lt % tiles_k) << 6; R.N = j.N; R.K = j.K; R.dstA = j.dstA; R.dstB = j.dstB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, r = u >> 3, c = (u & 7) * 8;
      R.off[i] = j.src_off + (long)(R.n0 + r) * j.K + R.k0 + c;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        R.pp[2 * i + q] = ld_stream<kNtCast>(p + R.off[i] + 4 * q);
        R.gg[2 * i + q] = ld_stream<kNtCast>(g + R.off[i] + 4 * q);
        R.mm[2 * i + q] = ld_stream<kNtCast>(m + R.off[i] + 4 * q);
        R.vv[2 * i + q] = ld_stream<kNtCast>(v + R.off[i] + 4 * q);
        R.vx[2 * i + q] = ld_stream<kNtCast>(vmax + R.off[i] + 4 * q);
      }
    }
  };
  auto process = [&](Regs& R) {
#pragma unroll
    for (int e = 0; e < 4; ++e) adam_amsgrad_update4(R.pp[e], R.gg[e], R.mm[e], R.vv[e], R.vx[e], h);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, r = u >> 3, c = (u & 7) * 8;
      float x[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        st_stream<kNtCast>(p + R.off[i] + 4 * q, R.pp[2 * i + q]);
        st_stream<kNtCast>(m + R.off[i] + 4 * q, R.mm[2 * i + q]);
        st_stream<kNtCast>(v + R.off[i] + 4 * q, R.vv[2 * i + q]);
        st_stream<kNtCast>(vmax + R.off[i] + 4 * q, R.vx[2 * i + q]);
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[4 * q + k] = R.pp[2 * i + q][k]; tile[r * TP + c + 4 * q + k] = R.pp[2 * i + q][k]; }
      }
      Out8<T>::store(shadow + R.dstA + (long)(R.n0 + r) * R.K + R.k0 + c, x);      // forward operand: [n][k], the tensor's own order
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int u = tid + 256 * i, kl = u >> 3, nl = (u & 7) * 8;
      float x[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = tile[(nl + q) * TP + kl];
      Out8<T>::store(shadow + R.dstB + (long)(R.k0 + kl) * R.N + R.n0 + nl, x);     // data-gradient operand: [k][n]
    }
    __syncthreads();
  };
  int t = tile_begin + (int)blockIdx.x;
  if (t >= end) return;
  Regs RA, RB;
  load(t, RA);
  while (true) {
    const int t1 = t + G;
    if (t1 < end) load(t1, RB);
    process(RA);
    if (t1 >= end) break;
    const int t2 = t1 + G;
    if (t2 < end) load(t2, RA);
    process(RB);
    if (t2 >= end) break;
    t = t2;
  }
}

// The same tensors when their data gradient reads the straight copy K-major (flow engine: c2_straight): the ONE operand is the tensor's
// own cast, so the update is a linear stream -- chunks of 4096 consecutive elements (= the 64 x 64 tiles of the job table, used here only
// as a prefix sum over the jobs), 16 per thread in four 16-byte groups, all 20 loads issued before the arithmetic; p / m / v / v_max and
// the bf16 operand leave the registers fully coalesced.  Same arithmetic as adam_amsgrad_kernel (adam_amsgrad_update).
template <typename T>
__global__ __launch_bounds__(256) void adam_cast_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ vmax, T* __restrict__ shadow,
                                                        const AdamTileJob* __restrict__ jobs, int njobs, int chunk_begin, int nchunks, AdamHyper h) {
  typedef typename RPack4<T>::type pack_t;
  for (int c = chunk_begin + (int)blockIdx.x; c < chunk_begin + nchunks; c += (int)gridDim.x) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].tile_start <= c) lo = mid; else hi = mid - 1;
    }
    const AdamTileJob j = jobs[lo];
    const long rel = (long)(c - j.tile_start) * 4096 + threadIdx.x * 4;
    f32x4 pp[4], gg[4], mm[4], vv[4], vx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long i = j.src_off + rel + q * 1024;
      pp[q] = ld_stream<kNtCast>(p + i); gg[q] = ld_stream<kNtCast>(g + i);
      mm[q] = ld_stream<kNtCast>(m + i); vv[q] = ld_stream<kNtCast>(v + i);
      vx[q] = ld_stream<kNtCast>(vmax + i);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      adam_amsgrad_update4(pp[q], gg[q], mm[q], vv[q], vx[q], h);
      const long i = j.src_off + rel + q * 1024;
      st_stream<kNtCast>(p + i, pp[q]); st_stream<kNtCast>(m + i, mm[q]);
      st_stream<kNtCast>(v + i, vv[q]); st_stream<kNtCast>(vmax + i, vx[q]);
      pack_t o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = ET<T>::from_f32(pp[q][k]);
      *reinterpret_cast<pack_t*>(shadow + j.dstA + rel + q * 1024) = o;
    }
  }
}

// the same update over the gaps between those tensors: segment s0 + blockIdx.y of a table, clipped to [begin, end)
struct AdamSeg { long off, len; };
__global__ void adam_amsgrad_seg_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                        float* __restrict__ vmax, const AdamSeg* __restrict__ segs, int s0, long begin, long end,
                                        AdamHyper h) {
  const AdamSeg sg = segs[s0 + blockIdx.y];
  const long lo = sg.off > begin ? sg.off : begin, hi = sg.off + sg.len < end ? sg.off + sg.len : end;
  const long stride = (long)gridDim.x * blockDim.x * 4;
  // four 16-byte groups per thread and iteration, all 20 loads issued before the arithmetic: with one group in flight the few blocks a
  // segment gets underneath the backward pass streamed at 1.8 TB/s (profiles/r04_bench_steady.txt: 310 us for the 27 % of a piece)
  for (long i0 = lo + ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i0 < hi; i0 += 4 * stride) {
    if (i0 + 3 * stride + 3 < hi) {
      f32x4 pp[4], gg[4], mm[4], vv[4], vx[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long i = i0 + q * stride;
        pp[q] = ld_stream<kNtSeg>(p + i); gg[q] = ld_stream<kNtSeg>(g + i);
        mm[q] = ld_stream<kNtSeg>(m + i); vv[q] = ld_stream<kNtSeg>(v + i); vx[q] = ld_stream<kNtSeg>(vmax + i);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const long i = i0 + q * stride;
        adam_amsgrad_update4(pp[q], gg[q], mm[q], vv[q], vx[q], h);
        st_stream<kNtSeg>(p + i, pp[q]); st_stream<kNtSeg>(m + i, mm[q]);
        st_stream<kNtSeg>(v + i, vv[q]); st_stream<kNtSeg>(vmax + i, vx[q]);
      }
      continue;
    }
    for (int q = 0; q < 4; ++q) {
      const long i = i0 + q * stride;
      if (i + 3 < hi) {
        f32x4 pp = *reinterpret_cast<f32x4*>(p + i), gg = *reinterpret_cast<const f32x4*>(g + i);
        f32x4 mm = *reinterpret_cast<f32x4*>(m + i), vv = *reinterpret_cast<f32x4*>(v + i);
        f32x4 vx = *reinterpret_cast<f32x4*>(vmax + i);
        adam_amsgrad_update4(pp, gg, mm, vv, vx, h);
        *reinterpret_cast<f32x4*>(p + i) = pp; *reinterpret_cast<f32x4*>(m + i) = mm;
        *reinterpret_cast<f32x4*>(v + i) = vv; *reinterpret_cast<f32x4*>(vmax + i) = vx;
      } else {
        for (long k = i; k < hi; ++k) adam_amsgrad_update(p[k], g[k], m[k], v[k], vmax[k], h);
      }
    }
  }
}

}  // namespace ipoke

using namespace ipoke;

extern "C" int ipoke_adam_tile_job_size(void) { return (int)sizeof(AdamTileJob); }
extern "C" int ipoke_adam_seg_size(void) { return (int)sizeof(AdamSeg); }

static AdamHyper make_hyper(float lr, float beta1, float beta2, float eps, float wd, int step, float grad_scale) {
  return adam_make_hyper(lr, beta1, beta2, eps, wd, step, grad_scale);
}

extern "C" int ipoke_adam_amsgrad_shadow_tiles(float* p, const float* g, float* m, float* v, float* vmax, void* shadow, const void* jobs_dev,
                                               int njobs, int tile_begin, int ntiles, float lr, float beta1, float beta2, float eps,
                                               float weight_decay, int step, float grad_scale, int max_blocks, int dtype, void* stream) {
  IPK_REQUIRE(p && g && m && v && vmax && shadow && jobs_dev && njobs >= 1 && ntiles >= 0 && step >= 1, "bad arguments");
  IPK_REQUIRE(dtype == IPOKE_BF16 || dtype == IPOKE_F32, "bad dtype");
  if (ntiles == 0) return IPOKE_OK;
  const AdamHyper h = make_hyper(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
  const int grid = max_blocks > 0 && max_blocks < ntiles ? max_blocks : ntiles;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == IPOKE_BF16)
    hipLaunchKernelGGL(adam_shadow_tile_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, p, g, m, v, vmax, (bf16_t*)shadow,
                       (const AdamTileJob*)jobs_dev, njobs, tile_begin, ntiles, h);
  else
    hipLaunchKernelGGL(adam_shadow_tile_kernel<float>, dim3(grid), dim3(256), 0, s, p, g, m, v, vmax, (float*)shadow,
                       (const AdamTileJob*)jobs_dev, njobs, tile_begin, ntiles, h);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

/* as ipoke_adam_amsgrad_shadow_tiles for tensors with ONE operand (their own cast in the tensor's order; AdamTileJob.dstB is ignored):
 * the tile range [tile_begin, tile_begin + ntiles) of the job table is walked as chunks of 4096 consecutive elements */
extern "C" int ipoke_adam_amsgrad_cast_tiles(float* p, const float* g, float* m, float* v, float* vmax, void* shadow, const void* jobs_dev,
                                             int njobs, int tile_begin, int ntiles, float lr, float beta1, float beta2, float eps,
                                             float weight_decay, int step, float grad_scale, int max_blocks, int dtype, void* stream) {
  IPK_REQUIRE(p && g && m && v && vmax && shadow && jobs_dev && njobs >= 1 && ntiles >= 0 && step >= 1, "bad arguments");
  IPK_REQUIRE(dtype == IPOKE_BF16 || dtype == IPOKE_F32, "bad dtype");
  if (ntiles == 0) return IPOKE_OK;
  const AdamHyper h = make_hyper(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
  const int cap = max_blocks > 0 ? max_blocks : 4096;
  const int grid = cap < ntiles ? cap : ntiles;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == IPOKE_BF16)
    hipLaunchKernelGGL(adam_cast_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, p, g, m, v, vmax, (bf16_t*)shadow, (const AdamTileJob*)jobs_dev, njobs,
                       tile_begin, ntiles, h);
  else
    hipLaunchKernelGGL(adam_cast_kernel<float>, dim3(grid), dim3(256), 0, s, p, g, m, v, vmax, (float*)shadow, (const AdamTileJob*)jobs_dev, njobs,
                       tile_begin, ntiles, h);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

extern "C" int ipoke_adam_amsgrad_segments(float* p, const float* g, float* m, float* v, float* vmax, const void* segs_dev, int seg_begin,
                                           int nsegs, int64_t begin, int64_t end, float lr, float beta1, float beta2, float eps,
                                           float weight_decay, int step, float grad_scale, int blocks_per_segment, void* stream) {
  IPK_REQUIRE(p && g && m && v && vmax && segs_dev && nsegs >= 0 && seg_begin >= 0 && end >= begin && step >= 1, "bad arguments");
  if (nsegs == 0) return IPOKE_OK;
  const AdamHyper h = make_hyper(lr, beta1, beta2, eps, weight_decay, step, grad_scale);
  hipLaunchKernelGGL(adam_amsgrad_seg_kernel, dim3(blocks_per_segment > 0 ? blocks_per_segment : 8, nsegs), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), p, g, m, v, vmax, (const AdamSeg*)segs_dev, seg_begin, (long)begin, (long)end, h);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

extern "C" int ipoke_relayout_job_size(void) { return (int)sizeof(RelayoutJob); }
extern "C" int ipoke_wn_job_size(void) { return (int)sizeof(WnJob); }

extern "C" int ipoke_relayout_multi(const float* params, void* shadow, const float* wn_scale, const void* jobs_dev, int njobs,
                                    int total_blocks, const int32_t* block_job_dev, int dtype, void* stream) {
  return ipoke_relayout_multi_range(params, shadow, wn_scale, jobs_dev, njobs, 0, total_blocks, block_job_dev, dtype, stream);
}
/* blocks [block_begin, block_begin + nblocks) of the job table only (the jobs of a contiguous range of tensors) */
extern "C" int ipoke_relayout_multi_range(const float* params, void* shadow, const float* wn_scale, const void* jobs_dev, int njobs,
                                          int block_begin, int nblocks, const int32_t* block_job_dev, int dtype, void* stream) {
  IPK_REQUIRE(params && shadow && jobs_dev && njobs > 0 && nblocks >= 0 && block_begin >= 0, "bad arguments");
  if (nblocks == 0) return IPOKE_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // optional grid cap (developer A/B, see relayout_kernel); one workgroup per tile otherwise
  static const int cap_env = getenv("IPOKE_RELAYOUT_BLOCKS") ? atoi(getenv("IPOKE_RELAYOUT_BLOCKS")) : 0;
  const int cap = block_begin == 0 && nblocks > 100000 ? 65536 : (cap_env > 0 ? cap_env : nblocks);
  const int grid = nblocks < cap ? nblocks : cap;
  if (dtype == IPOKE_BF16)
    hipLaunchKernelGGL(relayout_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, params, (bf16_t*)shadow, wn_scale,
                       (const RelayoutJob*)jobs_dev, njobs, block_job_dev, block_begin, nblocks);
  else
    hipLaunchKernelGGL(relayout_kernel<float>, dim3(grid), dim3(256), 0, s, params, (float*)shadow, wn_scale,
                       (const RelayoutJob*)jobs_dev, njobs, block_job_dev, block_begin, nblocks);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
extern "C" int ipoke_wn_scale_multi(const float* params, float* scale, float* inv_norm, const void* jobs_dev, int njobs,
                                    int total_rows, void* stream) {
  return ipoke_wn_scale_multi_range(params, scale, inv_norm, jobs_dev, 0, njobs, 0, total_rows, stream);
}
extern "C" int ipoke_wn_scale_multi_range(const float* params, float* scale, float* inv_norm, const void* jobs_dev, int job_begin,
                                          int njobs, int row_begin, int nrows, void* stream) {
  IPK_REQUIRE(params && scale && inv_norm && jobs_dev && njobs >= 0 && nrows >= 0, "bad arguments");
  if (njobs == 0 || nrows == 0) return IPOKE_OK;
  hipLaunchKernelGGL(wn_scale_kernel, dim3(ceil_div(nrows, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     params, scale, inv_norm, (const WnJob*)jobs_dev + job_begin, njobs, row_begin, nrows);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
extern "C" int ipoke_wn_bwd_multi(const float* params, float* grads, const float* inv_norm, const void* jobs_dev, int njobs,
                                  int total_rows, void* stream) {
  IPK_REQUIRE(params && grads && inv_norm && jobs_dev && njobs > 0, "bad arguments");
  hipLaunchKernelGGL(wn_bwd_kernel, dim3(ceil_div(total_rows, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     params, grads, inv_norm, (const WnJob*)jobs_dev, njobs, 0, total_rows);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
/* the same for a contiguous sub-range of jobs covering global rows [row_begin, row_begin + nrows) */
extern "C" int ipoke_wn_bwd_multi_range(const float* params, float* grads, const float* inv_norm, const void* jobs_dev, int job_begin,
                                        int njobs, int row_begin, int nrows, void* stream) {
  IPK_REQUIRE(params && grads && inv_norm && jobs_dev && njobs >= 0 && nrows >= 0, "bad arguments");
  if (njobs == 0 || nrows == 0) return IPOKE_OK;
  hipLaunchKernelGGL(wn_bwd_kernel, dim3(ceil_div(nrows, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     params, grads, inv_norm, (const WnJob*)jobs_dev + job_begin, njobs, row_begin, nrows);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
