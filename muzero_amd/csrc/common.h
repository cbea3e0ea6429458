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

constexpr int kMaxDevices = 64;

inline int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  return dev;
}

// Raise a kernel's dynamic-LDS limit once per device (needed above 64 KiB).  `done` is a
// per-kernel array of kMaxDevices flags.
inline int ensure_dynamic_lds(const void* kernel, int bytes, bool* done, const char* what) {
  const int dev = current_device();
  if (done[dev]) return MZ_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return fail(MZ_E_LAUNCH, "%s: LDS attribute: %s", what, hipGetErrorString(e));
  done[dev] = true;
  return MZ_OK;
}

inline int compute_units() {
  static int cached[kMaxDevices] = {0};
  const int dev = current_device();
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

}  // namespace mz

#define MZ_REQUIRE(cond, ...) \
  do {                        \
    if (!(cond)) return mz::fail(MZ_E_ARG, __VA_ARGS__); \
  } while (0)
