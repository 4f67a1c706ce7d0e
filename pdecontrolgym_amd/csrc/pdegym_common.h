// pdegym_common.h -- error slot and launch check shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

namespace pdegym {

// thread-local message returned by pdegym_last_error(); defined in pdegym_abi.hip
char* error_slot();
int fail(int code, const char* msg);
int check_launch(const char* what);

// Per-device caches: the library keeps no state that is bound to "the first device that called" (a process may drive
// several GPUs, one engine per device, each call made with that engine's device current).
constexpr int kMaxDevices = 64;
inline int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  return dev;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel call site, device); `slots` is that site's static array
inline bool raise_dynamic_lds_limit(const void* kernel, int bytes, signed char (&slots)[kMaxDevices]) {
  const int dev = current_device();
  if (slots[dev] == 0)
    slots[dev] = (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) ? 1 : -1;
  return slots[dev] > 0;
}

}  // namespace pdegym
