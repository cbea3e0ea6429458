// Library-wide pieces of the C ABI: version and per-thread error text.
#include "common.h"

namespace mz {
char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace mz

extern "C" {
int mz_version(void) { return 1; }
const char* mz_last_error(void) { return mz::error_buffer(); }
}
