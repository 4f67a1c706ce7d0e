// pdegym_common.h -- error slot and launch check shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

namespace pdegym {

// thread-local message returned by pdegym_last_error(); defined in pdegym_abi.hip
char* error_slot();
int fail(int code, const char* msg);
int check_launch(const char* what);

}  // namespace pdegym
