// C-ABI plumbing shared by every entry point of libfar3d_hip.so: error string, version, device probe.
#include "common.hpp"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void far3d_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* far3d_last_error(void) { return g_err; }

extern "C" int far3d_abi_version(void) { return 1; }

// Returns the number of visible HIP devices (0 on a CPU-only box), or a negative error code.
extern "C" int far3d_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    far3d_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return 0;
  }
  return n;
}

// gcnArchName of device `dev` copied into `buf` (NUL-terminated). 0 on success.
extern "C" int far3d_device_arch(int dev, char* buf, int buflen) {
  FAR3D_CHECK_ARG(buf && buflen > 0, "far3d_device_arch: bad buffer");
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) {
    far3d_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e));
    return FAR3D_ERR_LAUNCH;
  }
  strncpy(buf, prop.gcnArchName, buflen - 1);
  buf[buflen - 1] = 0;
  return FAR3D_OK;
}
