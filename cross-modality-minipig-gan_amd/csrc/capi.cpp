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

// Development aid (include/mpgan_hip.h): in a `make STAMPS=1` build every following gather-conv launch records
// its blocks' phase time stamps into `buf`; the product build has no stamps and says so.
#ifdef MPGAN_STAMPS
namespace mpgan {
StampCtx& stamp_ctx() {
  static StampCtx c{nullptr, 0, 0, 0};
  return c;
}
}  // namespace mpgan
#endif
extern "C" int mpgan_debug_stamps(void* buf, int64_t launches, int64_t blocks_per_launch) {
#ifdef MPGAN_STAMPS
  mpgan::StampCtx& c = mpgan::stamp_ctx();
  c.base = static_cast<unsigned long long*>(buf);
  c.launches = buf ? launches : 0;
  c.blocks = blocks_per_launch;
  c.next = 0;
  return MPGAN_OK;
#else
  (void)buf; (void)launches; (void)blocks_per_launch;
  mpgan::set_error("debug_stamps: this library was built without -DMPGAN_STAMPS (make STAMPS=1)");
  return MPGAN_ERR_UNSUPPORTED;
#endif
}
// Launches stamped since the last mpgan_debug_stamps call (-1 in the product build).
extern "C" int64_t mpgan_debug_stamps_used(void) {
#ifdef MPGAN_STAMPS
  return mpgan::stamp_ctx().next;
#else
  return -1;
#endif
}
// Rate of the stamps' clock in kHz (hipDeviceAttributeWallClockRate of the current device).
extern "C" int32_t mpgan_debug_clock_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return -1;
  return khz;
}
