// pdegym_abi.hip -- ABI version and error reporting of libpdegym_hip.so (see include/pdegym.h).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "pdegym.h"
#include "pdegym_common.h"

namespace pdegym {

char* error_slot() {
  static thread_local char slot[512] = {0};
  return slot;
}

int fail(int code, const char* msg) {
  std::snprintf(error_slot(), 512, "%s", msg);
  return code;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    std::snprintf(error_slot(), 512, "%s: %s", what, hipGetErrorString(e));
    return -100;
  }
  return 0;
}

}  // namespace pdegym

extern "C" {

int pdegym_abi_version(void) { return PDEGYM_ABI_VERSION; }

const char* pdegym_last_error(void) { return pdegym::error_slot(); }

}  // extern "C"
