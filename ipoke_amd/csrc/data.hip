// Input pipeline on the device (reference data/base_dataset.py): optical-flow resize (_get_flow :651-693) and the poke
// simulation from a flow field (_get_poke :507-648), batched -- one workgroup per sample for the statistics / selection,
// a pixel-parallel fill for the poke tensors.  Byte/index work plus a few reductions: nothing here touches the matrix cores.
#include "common.h"

#include <cmath>
#include <cstring>

using namespace ipoke;
#define STREAM(s) reinterpret_cast<hipStream_t>(s)
static int grid1(long n, int cap = 4096) { long g = (n + 255) / 256; if (g < 1) g = 1; if (g > cap) g = cap; return (int)g; }

// ---------------------------------------------------------------------------------------------- _get_flow
// dst[b][c][y][x] = bilinear(src[b][c] / div) with align_corners=True; the division comes first, as in the reference
__global__ void flow_resize_kernel(const float* __restrict__ src, float* __restrict__ dst, int BC, int Hi, int Wi, int Ho, int Wo, float div) {
  const long total = (long)BC * Ho * Wo;
  const float sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo); long t = i / Wo;
    const int oy = (int)(t % Ho); const long bc = t / Ho;
    const float fy = oy * sy, fx = ox * sx;
    const int y0 = min((int)fy, Hi - 1), x0 = min((int)fx, Wi - 1);
    const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const float* p = src + bc * Hi * Wi;
    const float a = __fdiv_rn(p[y0 * Wi + x0], div), b = __fdiv_rn(p[y0 * Wi + x1], div);
    const float c = __fdiv_rn(p[y1 * Wi + x0], div), d = __fdiv_rn(p[y1 * Wi + x1], div);
    dst[i] = (1.f - wy) * ((1.f - wx) * a + wx * b) + wy * ((1.f - wx) * c + wx * d);
  }
}

// ---------------------------------------------------------------------------------------------- _get_poke
constexpr int kPokeThreads = 1024;
constexpr int kMaxPokes = 16;
struct PokeArgs {
  const float* flow; int B, H, W, poke_size, n_pokes, fix_n_pokes;
  const int* zero; const float* u; float* amp; int* sel; long long* centers; int* status;
};

__device__ __forceinline__ double block_sum_d(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_minmax(float v, bool want_max, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const float w = __shfl_xor(v, o, 64); v = want_max ? fmaxf(v, w) : fminf(v, w); }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t = want_max ? fmaxf(t, red[i]) : fminf(t, red[i]);
  return t;
}
// exclusive prefix of one int per thread; returns the prefix, *total = block sum
__device__ __forceinline__ int block_scan(int v, int* sh, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
  __syncthreads();
  if (lane == 63) sh[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { if (i < w) base += sh[i]; tot += sh[i]; }
  *total = tot;
  return base + inc - v;
}
// k-th smallest (0-based) of n non-negative floats by a 4 x 8-bit radix select on the bit patterns
__device__ float kth_smallest(const float* a, int n, int k, int* hist) {
  unsigned prefix = 0, mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const unsigned b = __float_as_uint(a[i]);
      if ((b & mask) == prefix) atomicAdd(&hist[(b >> shift) & 255], 1);
    }
    __syncthreads();
    int acc = 0, d = 0;
    for (; d < 256; ++d) { if (acc + hist[d] > k) break; acc += hist[d]; }     // every thread walks the same 256 counters
    k -= acc;
    prefix |= (unsigned)d << shift; mask |= 255u << shift;
  }
  return __uint_as_float(prefix);
}

enum { SEL_GT2 = 0, SEL_GT1 = 1, SEL_GT0 = 2, SEL_LT = 3 };
__device__ __forceinline__ bool sel_test(int mode, float v, float t) { return mode == SEL_LT ? v < t : v > t; }

// positions (row-major order inside the window) of the picks[j]-th element passing the test, j < npick
__device__ void pick_positions(const float* a, int n, int mode, float thr, const int* picks, int npick, int* out, int* sh) {
  const int per = (n + blockDim.x - 1) / blockDim.x, lo = threadIdx.x * per, hi = min(n, lo + per);
  int cnt = 0;
  for (int i = lo; i < hi; ++i) cnt += sel_test(mode, a[i], thr);
  int total;
  const int off = block_scan(cnt, sh, &total);
  for (int j = 0; j < npick; ++j) {
    const int k = picks[j];
    if (k >= off && k < off + cnt) {
      int seen = off;
      for (int i = lo; i < hi; ++i)
        if (sel_test(mode, a[i], thr)) { if (seen == k) { out[j] = i; break; } ++seen; }
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(kPokeThreads) void poke_select_kernel(PokeArgs p) {
  __shared__ double red_d[16];
  __shared__ float red_f[16];
  __shared__ int sh_i[256];
  __shared__ int picks[kMaxPokes], pos[kMaxPokes], pos_src[kMaxPokes];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int h0 = p.poke_size, w0 = p.poke_size, wh = p.H - 2 * p.poke_size, ww = p.W - 2 * p.poke_size, n = wh * ww;
  const float* fx = p.flow + (long)b * 2 * p.H * p.W;
  const float* fy = fx + (long)p.H * p.W;
  float* a = p.amp + (long)b * n;
  const bool zero = p.zero && p.zero[b];
  const float* u = p.u + (long)b * (1 + 2 * p.n_pokes);
  // amplitude = torch.norm(flow, 2, dim=0) in fp32 without contraction, shifted to min 0, scaled to max 1
  float lo = INFINITY;
  for (int i = tid; i < n; i += blockDim.x) {
    const int y = h0 + i / ww, x = w0 + i % ww;
    const float vx = fx[y * p.W + x], vy = fy[y * p.W + x];
    const float v = __fsqrt_rn(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)));
    a[i] = v; lo = fminf(lo, v);
  }
  lo = block_minmax(lo, false, red_f);
  float hi = -INFINITY;
  for (int i = tid; i < n; i += blockDim.x) { const float v = __fsub_rn(a[i], lo); a[i] = v; hi = fmaxf(hi, v); }
  hi = block_minmax(hi, true, red_f);
  double s = 0.0;
  for (int i = tid; i < n; i += blockDim.x) { const float v = __fdiv_rn(a[i], hi); a[i] = v; s += (double)v; }
  const double mean_d = block_sum_d(s, red_d) / (double)n;
  double q = 0.0;
  for (int i = tid; i < n; i += blockDim.x) { const double d = (double)a[i] - mean_d; q += d * d; }
  const float mean = (float)mean_d, stdv = (float)sqrt(block_sum_d(q, red_d) / (double)(n - 1));
  const float t2 = __fadd_rn(mean, __fmul_rn(stdv, 2.0f)), t1 = __fadd_rn(mean, stdv);

  // candidate counts of every rule in one pass
  int c2 = 0, c1 = 0, c0 = 0;
  for (int i = tid; i < n; i += blockDim.x) { const float v = a[i]; c2 += v > t2; c1 += v > t1; c0 += v > mean; }
  int n2, n1, n0;
  block_scan(c2, sh_i, &n2); block_scan(c1, sh_i, &n1); block_scan(c0, sh_i, &n0);
  int cand_mode, cand_cnt; float cand_thr;
  int src_cnt = 0; float src_thr = t1;
  if (zero) {
    // np.percentile(amplitude, 5), linear interpolation between the two neighbouring order statistics
    const double vi = 0.05 * (double)(n - 1);
    const int k = (int)floor(vi);
    const double g = vi - (double)k;
    const float ak = kth_smallest(a, n, k, sh_i), ak1 = kth_smallest(a, n, min(k + 1, n - 1), sh_i);
    const double diff = (double)ak1 - (double)ak;
    const double pv = g >= 0.5 ? (double)ak1 - diff * (1.0 - g) : (double)ak + diff * g;
    cand_thr = (float)pv; cand_mode = SEL_LT;
    int cl = 0;
    for (int i = tid; i < n; i += blockDim.x) cl += a[i] < cand_thr;
    block_scan(cl, sh_i, &cand_cnt);
    if (n1 > 0) { src_thr = t1; src_cnt = n1; } else { src_thr = mean; src_cnt = n0; }
  } else if (n2 > 0) { cand_mode = SEL_GT2; cand_thr = t2; cand_cnt = n2; }
  else if (n1 > 0) { cand_mode = SEL_GT1; cand_thr = t1; cand_cnt = n1; }
  else { cand_mode = SEL_GT0; cand_thr = mean; cand_cnt = n0; }

  long long* cen = p.centers + (long)b * p.n_pokes * 2;
  int* sel = p.sel + (long)b * (1 + 4 * p.n_pokes);
  if (cand_cnt == 0 || (zero && src_cnt == 0)) {                      // the reference raises FlowError and resamples
    if (tid == 0) { p.status[b] = 1; sel[0] = 0; }
    for (int i = tid; i < 2 * p.n_pokes; i += blockDim.x) cen[i] = -1;
    return;
  }
  const int np = p.fix_n_pokes ? p.n_pokes : 1 + (int)floor((double)u[0] * (double)min(p.n_pokes, cand_cnt));
  if (tid < np) picks[tid] = (int)floor((double)u[1 + p.n_pokes + tid] * (double)cand_cnt);
  __syncthreads();
  pick_positions(a, n, cand_mode == SEL_LT ? SEL_LT : SEL_GT0, cand_thr, picks, np, pos, sh_i);
  if (zero) {
    if (tid < np) picks[tid] = (int)floor((double)u[1 + tid] * (double)src_cnt);
    __syncthreads();
    pick_positions(a, n, SEL_GT0, src_thr, picks, np, pos_src, sh_i);
  }
  if (tid == 0) { p.status[b] = 0; sel[0] = np; }
  if (tid < p.n_pokes) {
    const bool on = tid < np;
    const int r = on ? h0 + pos[tid] / ww : -1, c = on ? w0 + pos[tid] % ww : -1;
    cen[2 * tid] = r; cen[2 * tid + 1] = c;
    int* e = sel + 1 + 4 * tid;
    e[0] = r; e[1] = c;
    e[2] = on ? (zero ? h0 + pos_src[tid] / ww : r) : -1;
    e[3] = on ? (zero ? w0 + pos_src[tid] % ww : c) : -1;
  }
}

// poke[b][ch][y][x] = value of the LAST poke whose window covers (y, x) (later pokes overwrite earlier ones), else 0;
// flow_out (optional): the sample's flow, zeroed for zero-poke samples (_get_flow :680-681)
__global__ void poke_fill_kernel(const float* __restrict__ flow, const int* __restrict__ sel_all, const int* __restrict__ zero, float* __restrict__ poke,
                                 float* __restrict__ flow_out, int B, int H, int W, int half, int n_pokes, int equal_val) {
  const long total = (long)B * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W); long t = i / W;
    const int y = (int)(t % H); const int b = (int)(t / H);
    const int* sel = sel_all + (long)b * (1 + 4 * n_pokes);
    const float* f = flow + (long)b * 2 * H * W;
    float v0 = 0.f, v1 = 0.f;
    for (int j = sel[0] - 1; j >= 0; --j) {
      const int r = sel[1 + 4 * j], c = sel[2 + 4 * j];
      if (abs(y - r) <= half && abs(x - c) <= half) {
        const int sr = sel[3 + 4 * j] + (equal_val ? 0 : y - r), sc = sel[4 + 4 * j] + (equal_val ? 0 : x - c);
        v0 = f[sr * W + sc]; v1 = f[(long)H * W + sr * W + sc];
        break;
      }
    }
    float* o = poke + (long)b * 2 * H * W;
    o[y * W + x] = v0; o[(long)H * W + y * W + x] = v1;
    if (flow_out) {
      const bool z = zero && zero[b];
      float* fo = flow_out + (long)b * 2 * H * W;
      fo[y * W + x] = z ? 0.f : f[y * W + x]; fo[(long)H * W + y * W + x] = z ? 0.f : f[(long)H * W + y * W + x];
    }
  }
}

// ==============================================================================================
extern "C" int ipoke_flow_resize(const float* src, float* dst, int B, int C, int Hi, int Wi, int Ho, int Wo, float divide_by, void* stream) {
  IPK_REQUIRE(src && dst && B >= 1 && C >= 1 && Hi >= 1 && Wi >= 1 && Ho >= 1 && Wo >= 1 && divide_by != 0.f, "bad arguments");
  hipLaunchKernelGGL(flow_resize_kernel, dim3(grid1((long)B * C * Ho * Wo)), dim3(256), 0, STREAM(stream), src, dst, B * C, Hi, Wi, Ho, Wo, divide_by);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}

extern "C" int64_t ipoke_poke_workspace_bytes(int B, int H, int W, int poke_size, int n_pokes) {
  const long n = (long)(H - 2 * poke_size) * (W - 2 * poke_size);
  return (int64_t)B * (n * (long)sizeof(float) + (1 + 4L * n_pokes) * (long)sizeof(int));
}

extern "C" int ipoke_poke_simulate(const float* flow, int B, int H, int W, int poke_size, int n_pokes, int fix_n_pokes, int equal_poke_val,
                                   const int* zero_poke, const float* u, float* poke, int64_t* centers, float* flow_out, int* status, void* workspace,
                                   void* stream) {
  IPK_REQUIRE(flow && u && poke && centers && status && workspace, "null argument");
  IPK_REQUIRE(B >= 1 && poke_size >= 1 && n_pokes >= 1 && n_pokes <= kMaxPokes, "bad poke configuration");
  IPK_REQUIRE(H > 2 * poke_size + 1 && W > 2 * poke_size + 1, "the candidate window [poke_size, size - poke_size) is empty");
  const long n = (long)(H - 2 * poke_size) * (W - 2 * poke_size);
  PokeArgs a;
  a.flow = flow; a.B = B; a.H = H; a.W = W; a.poke_size = poke_size; a.n_pokes = n_pokes; a.fix_n_pokes = fix_n_pokes;
  a.zero = zero_poke; a.u = u; a.amp = reinterpret_cast<float*>(workspace);
  a.sel = reinterpret_cast<int*>(a.amp + (long)B * n); a.centers = reinterpret_cast<long long*>(centers); a.status = status;
  hipLaunchKernelGGL(poke_select_kernel, dim3(B), dim3(kPokeThreads), 0, STREAM(stream), a);
  hipLaunchKernelGGL(poke_fill_kernel, dim3(grid1((long)B * H * W)), dim3(256), 0, STREAM(stream), flow, a.sel, zero_poke, poke, flow_out, B, H, W,
                     poke_size / 2, n_pokes, equal_poke_val);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
