// Patch pipeline operators around the network: farthest point sampling and the
// large-K patch kNN.  Replace pointnet2_ops.furthest_point_sample (modules/utils/patch.py:102,156)
// and knn_cuda.KNN(k=256) (patch.py:33,107).  Both third-party packages are un-vendored and
// unpinned in the reference (docker/Dockerfile:47-49), so the semantics are defined here and in
// oracle/patch_ref.py:  FPS starts at index 0, squared distances are unfused fp32
// ((dx*dx)+(dy*dy))+(dz*dz), the farthest point is the FIRST maximum (smallest index);
// kNN is ordered by (distance, index) like pf_knn.
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"

namespace {

__device__ __forceinline__ float sqd(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)          // unfused, whatever -ffp-contract says
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// ---- FPS: one 1024-thread workgroup per cloud; running min-distance in `mind` (global, L2-resident);
// sequential in npoint by nature: per step = one strided sweep + one workgroup arg-max.
constexpr int FPS_T = 1024;

__global__ __launch_bounds__(FPS_T) void fps_kernel(const float* __restrict__ xyz, int N, int npoint,
                                                    float* __restrict__ mind, int* __restrict__ out) {
    __shared__ float sv[FPS_T / 64];
    __shared__ int si[FPS_T / 64];
    __shared__ int cur;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = xyz + (size_t)b * N * 3;
    float* md = mind + (size_t)b * N;
    int* o = out + (size_t)b * npoint;
    for (int k = tid; k < N; k += FPS_T) md[k] = 1e10f;
    if (tid == 0) { cur = 0; o[0] = 0; }
    __syncthreads();
    for (int j = 1; j < npoint; ++j) {
        const int last = cur;
        const float lx = p[last * 3 + 0], ly = p[last * 3 + 1], lz = p[last * 3 + 2];
        float best = -1.f;
        int besti = 0;
        for (int k = tid; k < N; k += FPS_T) {
            const float d = fminf(md[k], sqd(p[k * 3 + 0], p[k * 3 + 1], p[k * 3 + 2], lx, ly, lz));
            md[k] = d;
            if (d > best) { best = d; besti = k; }          // increasing k: first maximum kept
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const float ov = __shfl_xor(best, m);
            const int oi = __shfl_xor(besti, m);
            if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
        }
        if (lane == 0) { sv[wave] = best; si[wave] = besti; }
        __syncthreads();
        if (tid == 0) {
            float bv = sv[0];
            int bi = si[0];
            for (int w = 1; w < FPS_T / 64; ++w)
                if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
            cur = bi;
            o[j] = bi;
        }
        __syncthreads();
    }
}

// ---- large-K kNN: one workgroup per query; all N keys (dist bits << 32 | index) bitonic-sorted in LDS
constexpr int KS_T = 1024;
constexpr int KS_NMAX = 16384;          // 128 KiB of 64-bit keys

__global__ __launch_bounds__(KS_T) void knn_sort_kernel(const float* __restrict__ ref, const float* __restrict__ query,
                                                       int N, int M, int K, int NP /*pow2 >= N*/, int* __restrict__ idx_out,
                                                       float* __restrict__ dist_out) {
    extern __shared__ unsigned long long keys[];
    const int b = blockIdx.y, q = blockIdx.x, tid = threadIdx.x;
    const float* r = ref + (size_t)b * N * 3;
    const float* qq = query + ((size_t)b * M + q) * 3;
    const float qx = qq[0], qy = qq[1], qz = qq[2];
    for (int i = tid; i < NP; i += KS_T) {
        unsigned long long key = ~0ull;
        if (i < N) key = ((unsigned long long)__float_as_uint(sqd(qx, qy, qz, r[i * 3 + 0], r[i * 3 + 1], r[i * 3 + 2])) << 32) | (unsigned)i;
        keys[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= NP; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < NP; i += KS_T) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = keys[i], c = keys[l];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { keys[i] = c; keys[l] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < K; i += KS_T) {
        const unsigned long long key = keys[i];
        idx_out[((size_t)b * M + q) * K + i] = (int)(key & 0xffffffffu);
        if (dist_out) dist_out[((size_t)b * M + q) * K + i] = __uint_as_float((unsigned)(key >> 32));
    }
}

}  // namespace

// xyz [B,N,3] -> idx [B,npoint] int32; mind: [B,N] float scratch
extern "C" int pf_fps(const float* xyz, int B, int N, int npoint, float* mind, int* idx_out, void* stream) {
    if (!xyz || !mind || !idx_out) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || npoint <= 0 || npoint > N) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(fps_kernel, dim3(B), dim3(FPS_T), 0, (hipStream_t)stream, xyz, N, npoint, mind, idx_out);
    return pf_last_launch_status();
}

// K nearest references of every query, K <= N <= 16384: idx [B,M,K] int32, dist [B,M,K] squared L2 (nullable)
extern "C" int pf_knn_large(const float* ref, const float* query, int B, int N, int M, int K, int* idx_out,
                            float* dist_out, void* stream) {
    if (!ref || !query || !idx_out) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || M <= 0 || K <= 0 || K > N || B > 65535) return PF_ERR_SHAPE;
    if (N > KS_NMAX) return PF_ERR_UNSUPPORTED;
    int np = 1;
    while (np < N) np <<= 1;
    const size_t lds = (size_t)np * 8;
    if (lds > 64 * 1024)        // idempotent opt-in to > 64 KiB of dynamic LDS (no state kept on our side)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(knn_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            KS_NMAX * 8);
    hipLaunchKernelGGL(knn_sort_kernel, dim3(M, B), dim3(KS_T), lds, (hipStream_t)stream, ref, query, N, M, K, np, idx_out,
                       dist_out);
    return pf_last_launch_status();
}
