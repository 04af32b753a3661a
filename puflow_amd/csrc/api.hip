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

// ---- the weight-gradient stream (pf_api_internal.h) ---------------------------------------------------------------------
namespace {
thread_local void* g_dw_stream = nullptr;
struct EventRing {                                        // events are only ever recorded and waited for back to back: a small ring
    hipEvent_t ev[64] = {};
    int dev[64];
    unsigned next = 0;
};
thread_local EventRing g_ring;
}  // namespace

extern "C" int pf_train_set_dw_stream(void* stream) {
    g_dw_stream = stream;
    return PF_OK;
}
void* pf_dw_stream_get() { return g_dw_stream; }

hipStream_t pf_dw_fork(hipStream_t s) {
    hipStream_t s2 = (hipStream_t)g_dw_stream;
    if (!s2 || s2 == s) return s;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned k = g_ring.next++ & 63u;
    if (g_ring.ev[k] && g_ring.dev[k] != dev) { (void)hipEventDestroy(g_ring.ev[k]); g_ring.ev[k] = nullptr; }
    if (!g_ring.ev[k]) {
        if (hipEventCreateWithFlags(&g_ring.ev[k], hipEventDisableTiming) != hipSuccess) return s;      // no side stream then
        g_ring.dev[k] = dev;
    }
    if (hipEventRecord(g_ring.ev[k], s) != hipSuccess || hipStreamWaitEvent(s2, g_ring.ev[k], 0) != hipSuccess) return s;
    return s2;
}
