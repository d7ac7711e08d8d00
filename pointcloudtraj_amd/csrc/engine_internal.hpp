// engine_internal.hpp -- what the other translation units of libpct_engine.so (voxel.hip, traj.hip) share with
// engine.hip: the library's stream, lazy initialisation and the error string.  Not part of the ABI (hidden symbols).
#pragma once
#include <hip/hip_runtime.h>

namespace pct_internal {
__attribute__((visibility("hidden"))) hipStream_t stream();     // the library-owned stream (valid after require_init)
__attribute__((visibility("hidden"))) int require_init();       // pct_init(0) on first use
__attribute__((visibility("hidden"))) int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace pct_internal
