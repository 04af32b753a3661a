// Brute-force kNN (replaces pytorch3d.ops.knn_points, reference call sites
// modules/discrete/interpflow.py:104,:328) and K=1 nearest neighbour for Chamfer
// (pytorch3d.loss.chamfer_distance, metric/loss.py:42; arithmetic as the plain-C `nnsearch`
// in evaluation/tf_ops/nn_distance/tf_nndistance.cpp:21-43).
//
// Semantics (defined by the build): squared L2 in UNFUSED fp32
// ((dx*dx)+(dy*dy))+(dz*dz), result ordered by (distance asc, index asc), self included.
// Bit-exact against oracle/ref_cpu.py::knn_canonical.
//
// v1 mapping: one lane per query, reference points read with wave-uniform (scalar) loads,
// per-lane sorted top-K kept in registers.  Candidates are visited in increasing index, so a
// strict `<` keeps equal distances in index order.
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"

namespace {

__device__ __forceinline__ float sqdist(float qx, float qy, float qz, float rx, float ry, float rz) {
    const float dx = __fsub_rn(qx, rx), dy = __fsub_rn(qy, ry), dz = __fsub_rn(qz, rz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

template <int K>
__global__ __launch_bounds__(64) void knn_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                 int N, int M, int* __restrict__ idx_out,
                                                 float* __restrict__ dist_out) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 64 + threadIdx.x;
    const bool live = n < N;
    const float* q = p1 + ((size_t)b * N + (live ? n : N - 1)) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float* __restrict__ r = p2 + (size_t)b * M * 3;

    float bd[K];
    int bi[K];
#pragma unroll
    for (int i = 0; i < K; ++i) { bd[i] = __builtin_inff(); bi[i] = -1; }

    for (int j = 0; j < M; ++j) {
        float d = sqdist(qx, qy, qz, r[j * 3 + 0], r[j * 3 + 1], r[j * 3 + 2]);
        if (d < bd[K - 1]) {             // wave-divergent guard; body is a branch-free shifting insert
            // lt[i] = d < bd[i] is monotone in i (bd ascending).  Slot i takes its left neighbour when
            // the new element lands left of it, the new element when it lands exactly here, else keeps.
            // Strict `<` puts the new (larger-index) element AFTER existing equal distances and never
            // reorders existing entries.
#pragma unroll
            for (int i = K - 1; i >= 1; --i) {
                const bool ltl = d < bd[i - 1], lti = d < bd[i];
                bi[i] = ltl ? bi[i - 1] : (lti ? j : bi[i]);
                bd[i] = ltl ? bd[i - 1] : (lti ? d : bd[i]);
            }
            if (d < bd[0]) { bd[0] = d; bi[0] = j; }
        }
    }
    if (live) {
        int* o = idx_out + ((size_t)b * N + n) * K;
#pragma unroll
        for (int i = 0; i < K; ++i) o[i] = bi[i];
        if (dist_out) {
            float* od = dist_out + ((size_t)b * N + n) * K;
#pragma unroll
            for (int i = 0; i < K; ++i) od[i] = bd[i];
        }
    }
}

// K = 1: nearest neighbour distance + index (first minimum wins ties).
__global__ __launch_bounds__(64) void nn1_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                 int N, int M, float* __restrict__ dist_out,
                                                 int* __restrict__ idx_out) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 64 + threadIdx.x;
    const bool live = n < N;
    const float* q = p1 + ((size_t)b * N + (live ? n : N - 1)) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float* __restrict__ r = p2 + (size_t)b * M * 3;
    float best = __builtin_inff();
    int besti = 0;
    for (int j = 0; j < M; ++j) {
        const float d = sqdist(qx, qy, qz, r[j * 3 + 0], r[j * 3 + 1], r[j * 3 + 2]);
        const bool c = d < best;
        best = c ? d : best;
        besti = c ? j : besti;
    }
    if (live) {
        dist_out[(size_t)b * N + n] = best;
        if (idx_out) idx_out[(size_t)b * N + n] = besti;
    }
}

}  // namespace

extern "C" int pf_knn(const float* p1, const float* p2, int B, int N, int M, int K, int* idx_out,
                      float* dist_out, void* stream) {
    if (!p1 || !p2 || !idx_out) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || M <= 0 || K <= 0 || K > M || B > 65535) return PF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((N + 63) / 64, B), block(64);
    switch (K) {
        case 8:  hipLaunchKernelGGL(knn_kernel<8>, grid, block, 0, s, p1, p2, N, M, idx_out, dist_out); break;
        case 16: hipLaunchKernelGGL(knn_kernel<16>, grid, block, 0, s, p1, p2, N, M, idx_out, dist_out); break;
        case 4:  hipLaunchKernelGGL(knn_kernel<4>, grid, block, 0, s, p1, p2, N, M, idx_out, dist_out); break;
        case 32: hipLaunchKernelGGL(knn_kernel<32>, grid, block, 0, s, p1, p2, N, M, idx_out, dist_out); break;
        default: return PF_ERR_UNSUPPORTED;
    }
    return pf_last_launch_status();
}

extern "C" int pf_nn1(const float* p1, const float* p2, int B, int N, int M, float* dist_out, int* idx_out,
                      void* stream) {
    if (!p1 || !p2 || !dist_out) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || M <= 0 || B > 65535) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(nn1_kernel, dim3((N + 63) / 64, B), dim3(64), 0, (hipStream_t)stream, p1, p2, N, M, dist_out,
                       idx_out);
    return pf_last_launch_status();
}
