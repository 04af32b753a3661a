// Version / error strings of the C ABI (include/puflow_hip.h).
#include "pf_api_internal.h"

extern "C" int pf_version(void) { return 100; }   // 0.1.0

extern "C" const char* pf_error_string(int code) {
    switch (code) {
        case PF_OK: return "ok";
        case PF_ERR_NULL: return "required pointer is NULL";
        case PF_ERR_SHAPE: return "shape precondition violated";
        case PF_ERR_UNSUPPORTED: return "configuration not built";
        case PF_ERR_LAUNCH: return "kernel launch failed";
        case PF_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}
