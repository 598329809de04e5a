// Error reporting and version for the C-ABI (include/idee_hip.h).
#include <stdarg.h>

#include "common.h"

namespace idee {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace idee

extern "C" const char* idee_version(void) { return "idee_hip 0.1.0 (gfx950)"; }
extern "C" const char* idee_last_error(void) { return idee::err_buf(); }

// Refuses to run anywhere but on a gfx950 device; returns IDEE_OK when device `dev` qualifies.
extern "C" int idee_check_device(int dev) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return idee::fail(IDEE_ERR_LAUNCH, "hipGetDeviceProperties(%d): %s", dev, hipGetErrorString(e));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return idee::fail(IDEE_ERR_UNSUPPORTED, "device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName);
  return IDEE_OK;
}
