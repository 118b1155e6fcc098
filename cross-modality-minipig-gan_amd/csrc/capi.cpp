// Error plumbing shared by every entry point of libmpgan_hip.so.
#include "mpgan_common.h"
#include <string.h>
#include <stdlib.h>

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
extern "C" int mpgan_abi_version(void) { return 2; }   // 2: mpgan_conv_geom carries flags + min_blocks

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

#ifdef MPGAN_DRYRUN
// Host dry-run build (mpgan_common.h): what a launch may ask of a gfx950 CU, checked instead of launching.
namespace mpgan {
static long g_dry_launches = 0, g_dry_bad = 0;
static hipError_t g_dry_err = hipSuccess;
int dry_note_launch(const char* kernel, const char* where, dim3 grid, dim3 block, size_t lds) {
  ++g_dry_launches;
  static const bool trace = getenv("MPGAN_DRY_TRACE") != nullptr;      // one line per launch: which kernel, what grid
  if (trace)       // (a launcher that holds its instance in a variable: the launcher's own template arguments name it)
    fprintf(stderr, "[dry-run] launch %s grid %u block %u lds %zu\n", strcmp(kernel, "kern") ? kernel : where,
            grid.x * grid.y * grid.z, block.x, lds);
  const unsigned long threads = (unsigned long)block.x * block.y * block.z;
  const bool ok = grid.x >= 1 && grid.y >= 1 && grid.z >= 1 && grid.y <= 65535 && grid.z <= 65535 &&
                  (unsigned long)grid.x * grid.y * grid.z < (1ul << 32) && threads >= 1 && threads <= 1024 &&
                  threads % 64 == 0 && lds <= 160 * 1024;
  if (!ok) {
    ++g_dry_bad;
    g_dry_err = hipErrorInvalidConfiguration;
    fprintf(stderr, "[dry-run] BAD launch %s: grid (%u,%u,%u) block (%u,%u,%u) lds %zu\n", kernel, grid.x, grid.y, grid.z,
            block.x, block.y, block.z, lds);
  }
  return ok ? 0 : 1;
}
hipError_t dry_last_error() {
  const hipError_t e = g_dry_err;
  g_dry_err = hipSuccess;
  return e;
}
}  // namespace mpgan
extern "C" long mpgan_dry_launches(void) { return mpgan::g_dry_launches; }
extern "C" long mpgan_dry_bad_launches(void) { return mpgan::g_dry_bad; }
#endif
