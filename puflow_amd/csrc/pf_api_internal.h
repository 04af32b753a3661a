// Internal glue for the C-ABI entry points (error codes mirror include/puflow_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/puflow_hip.h"

static inline int pf_last_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PF_OK : PF_ERR_LAUNCH;
}
