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

static_assert(sizeof(krs_table) == 32, "krs_table layout (mirrored in keras_rs_amd/_lib.py)");
static_assert(sizeof(krs_feature) == 24, "krs_feature layout");
static_assert(sizeof(krs_gemm_epilogue) == 80, "krs_gemm_epilogue layout");
