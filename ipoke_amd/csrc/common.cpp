#include "common.h"
namespace ipoke {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) { g_last_error = msg; return code; }
}  // namespace ipoke
extern "C" const char* ipoke_last_error(void) { return ipoke::g_last_error.c_str(); }
extern "C" int ipoke_version(void) { return 100; }
extern "C" int ipoke_dtype_size(int dtype) { return dtype == IPOKE_BF16 ? 2 : dtype == IPOKE_F32 ? 4 : -1; }
