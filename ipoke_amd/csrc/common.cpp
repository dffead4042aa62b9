#include "common.h"
namespace ipoke {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) { g_last_error = msg; return code; }
}  // namespace ipoke
extern "C" const char* ipoke_last_error(void) { return ipoke::g_last_error.c_str(); }
extern "C" int ipoke_version(void) { return 100; }
extern "C" int ipoke_dtype_size(int dtype) { return dtype == IPOKE_BF16 ? 2 : dtype == IPOKE_F32 ? 4 : -1; }

// ---- in-situ kernel timing (bench.py roofline): HIP events around tagged launches, recorded on the stream the kernel is
// launched on, while the whole step runs as usual.  Off unless ipoke_timing_start() was called; never used in a timed step.
#include <vector>
namespace ipoke {
struct TimedLaunch { int tag, units; hipEvent_t e0, e1; double flops = 0.0, bytes = 0.0; };
static int g_timing = 0;        // 0 off, 1 the flow's tagged families (tags < IPOKE_TAG_CONV_BASE), 2 every convolution / weight gradient as well
static std::vector<TimedLaunch> g_timed;
static std::vector<hipEvent_t> g_pool;
static size_t g_pool_next = 0;
static hipEvent_t pooled_event() {
  if (g_pool_next == g_pool.size()) { hipEvent_t e = nullptr; (void)hipEventCreate(&e); g_pool.push_back(e); }
  return g_pool[g_pool_next++];
}
bool timing_active(int tag) { return g_timing >= (tag >= IPOKE_TAG_CONV_BASE ? 2 : 1); }
int timing_begin(int tag, hipStream_t s, int units) {
  if (!g_timing) return -1;
  TimedLaunch t{tag, units < 1 ? 1 : units, pooled_event(), pooled_event(), 0.0, 0.0};
  (void)hipEventRecord(t.e0, s);
  g_timed.push_back(t);
  return (int)g_timed.size() - 1;
}
void timing_end(int slot, hipStream_t s) {
  if (slot >= 0) (void)hipEventRecord(g_timed[slot].e1, s);
}
void timing_annotate(int slot, int tag, double flops, double bytes) {
  if (slot < 0 || slot >= (int)g_timed.size()) return;
  if (tag) g_timed[slot].tag = tag;
  g_timed[slot].flops = flops; g_timed[slot].bytes = bytes;
}
}  // namespace ipoke
extern "C" int ipoke_timing_start(void) {
  ipoke::g_timed.clear(); ipoke::g_pool_next = 0; ipoke::g_timing = 1;
  return IPOKE_OK;
}
extern "C" int ipoke_timing_start_all(void) {
  ipoke::g_timed.clear(); ipoke::g_pool_next = 0; ipoke::g_timing = 2;
  return IPOKE_OK;
}
/* stops recording; for every tag in tags[0..ntags) writes the number of recorded problems (a batched launch counts one per
 * problem) and the mean duration per problem (us).  Synchronises the device. */
extern "C" int ipoke_timing_stop(const int* tags, int ntags, int* counts, double* mean_us) {
  ipoke::g_timing = 0;
  IPK_HIP(hipDeviceSynchronize());
  for (int k = 0; k < ntags; ++k) { counts[k] = 0; mean_us[k] = 0.0; }
  for (const auto& t : ipoke::g_timed) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.e0, t.e1) != hipSuccess) continue;
    for (int k = 0; k < ntags; ++k) if (tags[k] == t.tag) { counts[k] += t.units; mean_us[k] += 1e3 * ms; }
  }
  for (int k = 0; k < ntags; ++k) if (counts[k]) mean_us[k] /= counts[k];
  ipoke::g_timed.clear();
  return IPOKE_OK;
}
/* as ipoke_timing_stop, per tag additionally the SUM of the recorded durations (us) and of the launches' algorithmic FLOPs and bytes
 * (convolutions and weight gradients annotate themselves: include/ipoke_hip.h IPOKE_TAG_CONV_* / IPOKE_TAG_WGRAD) */
extern "C" int ipoke_timing_stop_ex(const int* tags, int ntags, int* counts, double* total_us, double* flops, double* bytes) {
  ipoke::g_timing = 0;
  IPK_HIP(hipDeviceSynchronize());
  for (int k = 0; k < ntags; ++k) { counts[k] = 0; total_us[k] = 0.0; flops[k] = 0.0; bytes[k] = 0.0; }
  for (const auto& t : ipoke::g_timed) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.e0, t.e1) != hipSuccess) continue;
    for (int k = 0; k < ntags; ++k)
      if (tags[k] == t.tag) { counts[k] += t.units; total_us[k] += 1e3 * ms; flops[k] += t.flops; bytes[k] += t.bytes; }
  }
  ipoke::g_timed.clear();
  return IPOKE_OK;
}

extern "C" int ipoke_desc_sizes(int32_t* out, int n) {
  const int32_t sz[] = {(int32_t)sizeof(ipoke_conv_desc), (int32_t)sizeof(ipoke_wgrad_desc), (int32_t)sizeof(ipoke_affine_desc),
                        (int32_t)sizeof(ipoke_coupling_epi), (int32_t)sizeof(ipoke_mcf_desc), (int32_t)sizeof(ipoke_unit_pair_desc),
                        (int32_t)sizeof(ipoke_flow_config), (int32_t)sizeof(ipoke_norm_desc), (int32_t)sizeof(ipoke_norm_bwd_desc),
                        (int32_t)sizeof(ipoke_rowscale_bwd_desc), (int32_t)sizeof(ipoke_sn_job), (int32_t)sizeof(ipoke_wgrad_adam)};
  const int cnt = (int)(sizeof(sz) / sizeof(sz[0]));
  for (int i = 0; i < cnt && i < n; ++i) out[i] = sz[i];
  return cnt;
}
