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
// The thread's current device and its cache slot: devices beyond kMaxDevices have no slot (-1) and are queried every time --
// they are never mistaken for device 0.
inline int current_device(int* slot = nullptr) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  if (slot) *slot = dev < kMaxDevices ? dev : -1;
  return dev;
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel call site, device); `slots` is that site's static array
inline bool raise_dynamic_lds_limit(const void* kernel, int bytes, signed char (&slots)[kMaxDevices]) {
  int slot;
  current_device(&slot);
  if (slot < 0) return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
  if (slots[slot] == 0)
    slots[slot] = (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess) ? 1 : -1;
  return slots[slot] > 0;
}

// SIMDs of the current device (compute units x 4), cached per device: launch heuristics that mean "at most one wave per SIMD" ask
// here instead of assuming the 256 CUs of an unpartitioned MI355X (a CPX / DPX partition or another SKU has fewer).
inline int simd_count() {
  static int cached[kMaxDevices] = {};
  int slot;
  const int dev = current_device(&slot);
  if (slot >= 0 && cached[slot] != 0) return cached[slot];
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  if (slot >= 0) cached[slot] = cus * 4;
  return cus * 4;
}

}  // namespace pdegym
