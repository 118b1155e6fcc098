// Error plumbing shared by every entry point of libmpgan_hip.so.
#include "mpgan_common.h"
#include <string.h>

namespace mpgan {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mpgan

extern "C" const char* mpgan_last_error(void) { return mpgan::g_err; }
extern "C" int mpgan_abi_version(void) { return 1; }
