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

// Zero a device buffer on `stream` (the statistics accumulators of a plan, once per forward).
extern "C" int mpgan_zero_bytes(void* ptr, int64_t bytes, void* stream) {
  if (!ptr || bytes <= 0) {
    mpgan::set_error("zero_bytes: bad argument");
    return MPGAN_ERR_INVALID;
  }
  hipError_t e = hipMemsetAsync(ptr, 0, (size_t)bytes, (hipStream_t)stream);
  if (e != hipSuccess) {
    mpgan::set_error("zero_bytes: %s", hipGetErrorString(e));
    return MPGAN_ERR_HIP;
  }
  return MPGAN_OK;
}
