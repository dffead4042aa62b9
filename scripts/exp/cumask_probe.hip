// Developer probe: which physical CUs does a CU-masked stream use?  hipcc --offload-arch=gfx950 cumask_probe.hip -o cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
__global__ void where(unsigned* out) {
  if (threadIdx.x == 0) {
    unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID bits [3:0]
    unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));       // HW_REG_HW_ID
    out[blockIdx.x * 2] = xcc; out[blockIdx.x * 2 + 1] = hw;
  }
  // spin a little so that blocks spread over all CUs
  long t0 = wall_clock64(); while (wall_clock64() - t0 < 2000) {}
}
static void run(const char* name, std::vector<uint32_t> mask) {
  hipStream_t s;
  if (mask.empty()) hipStreamCreate(&s); else if (hipExtStreamCreateWithCUMask(&s, mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
  const int nb = 4096;
  unsigned* d; hipMalloc(&d, nb * 8);
  where<<<nb, 64, 0, s>>>(d);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(nb * 2); hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
  std::set<unsigned> cus; int per_xcc[16] = {0};
  std::set<unsigned> per[16];
  for (int i = 0; i < nb; ++i) { unsigned xcc = h[2 * i] & 15, hw = h[2 * i + 1]; unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7; unsigned id = (se << 5) | (sh << 4) | cu; per[xcc].insert(id); cus.insert((xcc << 8) | id); }
  printf("%-28s distinct CUs %3zu  per XCC:", name, cus.size());
  for (int x = 0; x < 8; ++x) printf(" %zu", per[x].size());
  printf("\n");
  hipFree(d); hipStreamDestroy(s);
}
int main() {
  run("no mask", {});
  run("all 256 bits", std::vector<uint32_t>(8, 0xffffffffu));
  run("first 32 bits", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
  run("first 128 bits", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0});
  run("bits 0..7 cleared", {0xffffff00u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
  run("every 8th bit cleared", std::vector<uint32_t>(8, 0xfefefefeu));
  run("one word (32 bits) only arg", {0xffffffffu});
  run("one word 0x00ffffff", {0x00ffffffu});
  return 0;
}
