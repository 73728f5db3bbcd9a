// Entry points declared in include/krs.h whose kernels are not written yet:
// they fail loudly (KRS_ERR_UNSUPPORTED).  Each one moves to its own .hip.
#include "krs_common.h"
#define KRS_TODO(name) return krs::fail(KRS_ERR_UNSUPPORTED, name ": not implemented yet")
extern "C" {
int krs_gemm(const void*, int64_t, int, const void*, int64_t, int, void*, int64_t, int64_t, int64_t, int64_t, int, int,
             const krs_gemm_epilogue*, void*, size_t, void*) { KRS_TODO("krs_gemm"); }
size_t krs_gemm_workspace_bytes(int64_t, int64_t, int64_t, int) { return 0; }
int krs_cross_epilogue_fwd(const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, float, int,
                           void*) { KRS_TODO("krs_cross_epilogue_fwd"); }
int krs_cross_epilogue_bwd(const void*, const void*, const void*, const void*, void*, void*, int, void*, float*,
                           int64_t, int64_t, int64_t, float, int, void*) { KRS_TODO("krs_cross_epilogue_bwd"); }
int krs_colsum(const void*, int64_t, int64_t, int64_t, int, float*, void*) { KRS_TODO("krs_colsum"); }
int krs_dot_interaction_fwd(const void* const*, const int64_t*, int, int64_t, int, int, int, int, void*, int64_t,
                            void*) { KRS_TODO("krs_dot_interaction_fwd"); }
int krs_dot_interaction_bwd(const void* const*, const int64_t*, int, int64_t, int, int, int, int, const void*,
                            int64_t, void* const*, const int64_t*, void*) { KRS_TODO("krs_dot_interaction_bwd"); }
size_t krs_mod_bucketize_workspace_bytes(int64_t, int) { return 0; }
int krs_mod_bucketize(const void*, int, int64_t, int, void*, int32_t*, int64_t*, void*, size_t, void*) { KRS_TODO("krs_mod_bucketize"); }
}
