// Multi-tensor weight preparation: one launch converts every fp32 master weight of the flow (PyTorch
// [out][in][kh][kw] layout, as in the reference's state dict) into the K-contiguous, zero padded
// matrix-core operand layouts ("shadows") used by the GEMM kernels, folding the weight-norm
// reparameterisation w = g * v / ||v||  (reference macow_utils.py:225-229, nn.utils.weight_norm).
// Also: weight-norm row statistics and the weight-norm backward (dW_eff -> dg, dv), multi-tensor.
#include "common.h"

namespace ipoke {

// dst[row][k], k = tap*inner_pad + i :  value = src[row*s_row + i*s_inner + tap*s_tap] * scale
struct RelayoutJob {
  long src_off;        // floats, into params
  long dst_off;        // elements, into the shadow buffer
  long scale_off;      // floats into the wn-scale buffer, or -1
  int rows_pad, rows_real;
  int taps, inner_pad, inner_real;
  int ld;              // dst row pitch (elements) >= taps*inner_pad
  long s_row, s_inner, s_tap;
  int scale_on_row;    // 1: scale[row], 0: scale[i]
  int block_start;     // first block of this job
};

template <typename T>
__global__ void relayout_kernel(const float* __restrict__ params, T* __restrict__ shadow, const float* __restrict__ wn_scale,
                                const RelayoutJob* __restrict__ jobs, int njobs) {
  // locate the job of this block (block_start is ascending)
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const RelayoutJob j = jobs[lo];
  const long total = (long)j.rows_pad * j.ld;
  const long base = ((long)blockIdx.x - j.block_start) * blockDim.x * 8;
  for (int u = 0; u < 8; ++u) {
    const long e = base + (long)u * blockDim.x + threadIdx.x;
    if (e >= total) break;
    const int row = (int)(e / j.ld), k = (int)(e - (long)row * j.ld);
    const int tap = k / j.inner_pad, i = k - tap * j.inner_pad;
    float v = 0.f;
    if (row < j.rows_real && tap < j.taps && i < j.inner_real) {
      v = params[j.src_off + row * j.s_row + i * j.s_inner + tap * j.s_tap];
      if (j.scale_off >= 0) v *= wn_scale[j.scale_off + (j.scale_on_row ? row : i)];
    }
    shadow[j.dst_off + e] = ET<T>::from_f32(v);
  }
}

// weight-norm rows: scale[n] = g[n] / ||v[n]||, inv_norm[n] = 1/||v[n]||.  One wave per row.
struct WnJob { long v_off, g_off, out_off; int rows, K; int row_start; };

__global__ void wn_scale_kernel(const float* __restrict__ params, float* __restrict__ scale, float* __restrict__ inv_norm,
                                const WnJob* __restrict__ jobs, int njobs, int total_rows) {
  const int grow = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (grow >= total_rows) return;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].row_start <= grow) lo = mid; else hi = mid - 1;
  }
  const WnJob j = jobs[lo];
  const int row = grow - j.row_start, lane = threadIdx.x & 63;
  const float* v = params + j.v_off + (long)row * j.K;
  float s = 0.f;
  for (int k = lane; k < j.K; k += 64) s += v[k] * v[k];
  s = wave_sum(s);
  if (lane == 0) {
    const float nrm = sqrtf(s);
    scale[j.out_off + row] = params[j.g_off + row] / nrm;
    inv_norm[j.out_off + row] = 1.f / nrm;
  }
}

// in place on the gradient buffer: grads[v_off..] holds dW_eff on entry and dv on exit; dg is written.
__global__ void wn_bwd_kernel(const float* __restrict__ params, float* __restrict__ grads, const float* __restrict__ inv_norm,
                              const WnJob* __restrict__ jobs, int njobs, int total_rows) {
  const int grow = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (grow >= total_rows) return;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].row_start <= grow) lo = mid; else hi = mid - 1;
  }
  const WnJob j = jobs[lo];
  const int row = grow - j.row_start, lane = threadIdx.x & 63;
  const float* v = params + j.v_off + (long)row * j.K;
  float* dw = grads + j.v_off + (long)row * j.K;
  float dot = 0.f;
  for (int k = lane; k < j.K; k += 64) dot += dw[k] * v[k];
  dot = wave_sum(dot);
  const float inv = inv_norm[j.out_off + row], g = params[j.g_off + row];
  const float a = g * inv, bcoef = g * dot * inv * inv * inv;
  for (int k = lane; k < j.K; k += 64) dw[k] = a * dw[k] - bcoef * v[k];
  if (lane == 0) grads[j.g_off + row] = dot * inv;
}

}  // namespace ipoke

using namespace ipoke;

extern "C" int ipoke_relayout_job_size(void) { return (int)sizeof(RelayoutJob); }
extern "C" int ipoke_wn_job_size(void) { return (int)sizeof(WnJob); }

extern "C" int ipoke_relayout_multi(const float* params, void* shadow, const float* wn_scale, const void* jobs_dev, int njobs,
                                    int total_blocks, int dtype, void* stream) {
  IPK_REQUIRE(params && shadow && jobs_dev && njobs > 0 && total_blocks > 0, "bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == IPOKE_BF16)
    hipLaunchKernelGGL(relayout_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, s, params, (bf16_t*)shadow, wn_scale,
                       (const RelayoutJob*)jobs_dev, njobs);
  else
    hipLaunchKernelGGL(relayout_kernel<float>, dim3(total_blocks), dim3(256), 0, s, params, (float*)shadow, wn_scale,
                       (const RelayoutJob*)jobs_dev, njobs);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
extern "C" int ipoke_wn_scale_multi(const float* params, float* scale, float* inv_norm, const void* jobs_dev, int njobs,
                                    int total_rows, void* stream) {
  IPK_REQUIRE(params && scale && inv_norm && jobs_dev && njobs > 0, "bad arguments");
  hipLaunchKernelGGL(wn_scale_kernel, dim3(ceil_div(total_rows, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     params, scale, inv_norm, (const WnJob*)jobs_dev, njobs, total_rows);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
extern "C" int ipoke_wn_bwd_multi(const float* params, float* grads, const float* inv_norm, const void* jobs_dev, int njobs,
                                  int total_rows, void* stream) {
  IPK_REQUIRE(params && grads && inv_norm && jobs_dev && njobs > 0, "bad arguments");
  hipLaunchKernelGGL(wn_bwd_kernel, dim3(ceil_div(total_rows, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     params, grads, inv_norm, (const WnJob*)jobs_dev, njobs, total_rows);
  IPK_LAUNCH_CHECK();
  return IPOKE_OK;
}
