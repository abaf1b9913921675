// core.hip — library identification, error text, device probe.
#include <string.h>

#include "pzn_common.h"

PZN_EXPORT int pzn_version(void) { return 100; /* 0.1.0 */ }

PZN_EXPORT const char* pzn_strerror(int status) {
  switch (status) {
    case PZN_OK: return "ok";
    case PZN_EINVAL: return "invalid argument (shape, null pointer or alignment)";
    case PZN_ELAUNCH: return "HIP launch failed (hipGetLastError != hipSuccess)";
    case PZN_EUNSUPPORTED: return "unsupported size for this build";
    case PZN_ENODEVICE: return "no usable gfx950 device";
    default: return "unknown pzn status";
  }
}

PZN_EXPORT int pzn_device_check(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return PZN_ENODEVICE;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return PZN_ENODEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return PZN_ENODEVICE;
  return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? PZN_OK : PZN_ENODEVICE;
}
