// Developer probe: what one CU's load path delivers from L2-resident data -- LDS-DMA (global_load_lds, 16 B per lane), plain 16-byte loads
// into registers, and both at once.  One 512-thread workgroup per CU re-reads its own 64 KB region.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) void glb_void;
constexpr int REGION = 64 * 1024, NTHR = 512, PER_PASS = REGION / (NTHR * 16);   // 8 instructions per thread and pass

template <int MODE> __global__ __launch_bounds__(NTHR) void probe(const v4u* src, unsigned* out, int passes, int active_waves) {
  extern __shared__ unsigned char smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  if (wave >= active_waves) return;
  const v4u* base = src + (size_t)blockIdx.x * (REGION / 16);
  v4u acc = {0, 0, 0, 0};
  const bool dma = MODE == 0 || (MODE == 2 && (wave & 1) == 0);
  if (dma) {
    for (int p = 0; p < passes; ++p) {
#pragma unroll
      for (int i = 0; i < PER_PASS; ++i)
        __builtin_amdgcn_global_load_lds((glb_void*)(base + i * NTHR + tid), (lds_void*)(smem + ((p & 1) * REGION) + (i * NTHR + wave * 64) * 16), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    for (int p = 0; p < passes; p += 2) {
      v4u v[2 * PER_PASS];
#pragma unroll
      for (int i = 0; i < 2 * PER_PASS; ++i) { const v4u* q = base + (i % PER_PASS) * NTHR + tid; asm volatile("" : "+v"(q)); v[i] = *q; }   // opaque: not hoisted
#pragma unroll
      for (int i = 0; i < 2 * PER_PASS; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x] = acc.x;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <typename F> static void run(const char* name, F kern, int lds, const v4u* src, unsigned* out, int passes, int waves, int grid) {
  if (lds) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), lds, 0, src, out, passes, waves);
  CK(hipEventRecord(a, 0));
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), lds, 0, src, out, passes, waves);
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  const double bytes = (double)passes * REGION * waves / 8.0;
  printf("%-34s waves %d grid %3d: %8.1f us, %7.1f GB/s per CU, %6.2f TB/s chip\n", name, waves, grid, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes * grid / (ms * 1e-3) / 1e12);
}
int main() {
  const int grid = 256, passes = 256;
  v4u* src; unsigned* out;
  CK(hipMalloc(&src, (size_t)grid * REGION)); CK(hipMemset(src, 1, (size_t)grid * REGION)); CK(hipMalloc(&out, 4096));
  for (int waves : {8, 4, 2}) {
    run("LDS-DMA 16 B/lane", probe<0>, 2 * REGION, src, out, passes, waves, grid);
    run("plain 16 B loads", probe<1>, 0, src, out, passes, waves, grid);
    run("even waves DMA, odd waves plain", probe<2>, 2 * REGION, src, out, passes, waves, grid);
  }
  run("LDS-DMA, 32 workgroups", probe<0>, 2 * REGION, src, out, passes, 8, 32);
  run("plain, 32 workgroups", probe<1>, 0, src, out, passes, 8, 32);
  return 0;
}
