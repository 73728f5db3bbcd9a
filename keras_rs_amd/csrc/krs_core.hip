// Library-wide entry points of libkrs_hip.so.
#include "krs_common.h"

namespace krs {
char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace krs

extern "C" int krs_version(void) { return KRS_VERSION; }
extern "C" const char* krs_last_error(void) { return krs::error_buffer(); }
