// Shared host-side helpers for the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/muzero_amd.h"

namespace mz {

char* error_buffer();  // thread-local, 512 bytes

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MZ_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return MZ_OK;
}

inline hipStream_t as_stream(mz_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace mz

#define MZ_REQUIRE(cond, ...) \
  do {                        \
    if (!(cond)) return mz::fail(MZ_E_ARG, __VA_ARGS__); \
  } while (0)
