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

// SIMDs of the current device (compute units x 4), cached per device: launch heuristics that mean "at most one wave per SIMD" ask
// here instead of assuming the 256 CUs of an unpartitioned MI355X (a CPX / DPX partition or another SKU has fewer).
inline int simd_count() {
  static int cached[kMaxDevices] = {};
  const int dev = current_device();
  if (cached[dev] == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cached[dev] = cus * 4;
  }
  return cached[dev];
}

}  // namespace pdegym
