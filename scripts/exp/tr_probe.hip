// Developer experiment: semantics of ds_read_b64_tr_b16 (gfx950).  LDS[i] = i (fp16); lane l passes address A(l).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void probe(int mode, float* out) {
  __shared__ __fp16 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (__fp16)(float)(i % 2048);
  __syncthreads();
  const int l = threadIdx.x;
  int elem;
  if (mode == 0) elem = l * 4;                                   // lane l -> its own 4 consecutive elements
  else if (mode == 1) elem = (l & 15) * 64 + (l >> 4) * 4;        // 16 rows of pitch 64 elements, group g -> column block g
  else elem = (l & 15) * 16 + (l >> 4) * 256;                     // rows of pitch 16
  h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(lds + elem));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)v[j];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4);
  float h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("  lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %5.0f", h[l * 4 + j]); if (l % 4 == 3) printf("\n"); }
  }
  return 0;
}
