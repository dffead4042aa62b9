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
#include <cstring>
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
  // matrix-core operands of the four convolutions of every cell (forward and data-gradient forms), once per pass
  for (int l = 0; l < p.L; ++l) {
    const float* w_ur = weights[4 * l], *w_o = weights[4 * l + 2];
    IPK_REQUIRE(w_ur && weights[4 * l + 1] && w_o && weights[4 * l + 3], "null weight");
    rc = ipoke_conv_weight_operand(w_ur, 2 * p.Ch, p.Kc, 9, 0, nullptr, c.wop(l, 0), p.Kc, dtype, stream); if (rc) return rc;
    rc = ipoke_conv_weight_operand(w_ur, p.Kc, 2 * p.Ch, 9, 1, nullptr, c.wop(l, 1), p.N2p, dtype, stream); if (rc) return rc;
    rc = ipoke_conv_weight_operand(w_o, p.Ch, p.Kc, 9, 0, nullptr, c.wop(l, 2), p.Kc, dtype, stream); if (rc) return rc;
    rc = ipoke_conv_weight_operand(w_o, p.Kc, p.Ch, 9, 1, nullptr, c.wop(l, 3), p.Chp, dtype, stream); if (rc) return rc;
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
  unsigned char* du = c.ws + p.DU; unsigned char* dh1 = c.ws + p.DH1;
  for (int t = p.T - 1; t >= 0; --t) {
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
  for (int l = 0; l < p.L; ++l) {
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
