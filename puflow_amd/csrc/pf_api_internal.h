// Internal glue for the C-ABI entry points (error codes mirror include/puflow_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/puflow_hip.h"

static inline int pf_last_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PF_OK : PF_ERR_LAUNCH;
}

// Dynamic LDS above 64 KiB needs hipFuncAttributeMaxDynamicSharedMemorySize on the kernel.  Set ONCE per (device, kernel) to the
// largest size any launcher of this library asks for (150 KiB: every launcher checks its own request against it), behind a
// mutex: a per-launch set with the launch's own size let two host threads interleave set(small) between another thread's
// set(large) and its launch, cost a driver call on every launch of the 1 ms paths and ran inside graph captures.
#include <mutex>
#include <utility>
#include <vector>
constexpr size_t PF_LDS_CAP = 150 * 1024;
static inline void pf_allow_lds(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return;
    static std::mutex mu;
    static std::vector<std::pair<int, const void*>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    for (const auto& d : done)
        if (d.first == dev && d.second == kernel) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes > PF_LDS_CAP ? bytes : PF_LDS_CAP));
    done.emplace_back(dev, kernel);
}

// ---- weight gradients beside the backward chain (pf_train_set_dw_stream, csrc/api.hip) ------------------------------------
// A backward entry point produces two kinds of results: the gradient wrt its INPUT, which the layer before it waits for, and
// the gradients wrt its WEIGHTS, which nobody reads before the optimizer.  With a weight-gradient stream set (per host thread)
// the split-K weight-gradient kernels and their reductions are enqueued there, after an event that orders them behind what
// the calling stream holds at that point; the caller joins the stream once, before the optimizer (puflow_amd/train_ops.py does
// it at the end of the autograd pass).  Inside a hipGraph capture the event becomes an edge: a parallel branch of the graph.
void* pf_dw_stream_get();
// the stream weight-gradient work of a call should go to: s itself when none is set (or it IS s), else the weight-gradient stream,
// made to wait for everything enqueued on s so far
hipStream_t pf_dw_fork(hipStream_t s);
// pf_gemm_ex with an addend: C = A B + bias + addend (addend [M, ldc] laid out like C, nullable, may alias C) - csrc/train_ops.hip
// slabs_left: nullable; when given and the product is split over K, the slabs stay in ws ([n][M, N], *slabs_left = n, no reduction
// launch: the caller sums them where it reads the result) - else *slabs_left = 0 and C holds the product
int pf_gemm_addend(int arith, const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn,
                   float* C, long long ldc, const float* bias, const float* addend, int M, int N, int K, float* ws,
                   long long ws_floats, void* stream, int* slabs_left);
