// ABI version + thread-local error string.
#include "common.h"

namespace mdf {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}
static thread_local const char* g_last_launch = "";
void note_launch(const char* kernel) { g_last_launch = kernel; }
}  // namespace mdf

extern "C" int mdf_abi_version(void) { return MDF_ABI_VERSION; }
extern "C" const char* mdf_last_error(void) { return mdf::g_err; }
extern "C" const char* mdf_last_launch(void) { return mdf::g_last_launch; }
