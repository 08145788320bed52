// host_api.h -- helpers shared by the translation units that implement the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/objnerf_hip.h"

namespace objnerf {
// records `msg` as the calling thread's last error and returns `code`
int set_error(int code, const char* msg);
// hipGetLastError() after a launch -> 0 or a recorded negative error
int check_launch(const char* what);

// MLP kernel launchers, one translation unit per input form (compile time)
int launch_mlp_fused(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s);
int launch_mlp_memory(const objnerf_mlp_args& a, long ntiles, unsigned grid, hipStream_t s);
}  // namespace objnerf
