// Small fused pieces of the TRAINING step that were chains of one-element torch launches.
//
// At 32 x 256 points a step is a few hundred launches on one dependency chain; every torch scalar op (a mean, a multiply by a
// loss weight, an index expression, a zero fill) is one of them.  Here:
//
//   pf_interp_wsum_fwd / _bwd   interpolation of the latent (modules/discrete/interpflow.py:153-186, 312-318): softmax over the 8
//                               neighbours of the R weight channels and the weighted sum of the GATHERED latent rows, written as
//                               the [T R, 3] rows flow g reads - replaces advanced indexing (arange + cast + index + reshape),
//                               pf_softmax_wsum_fwd and a transposing copy; backward likewise, with the scatter-add of the
//                               neighbours' gradients (float atomics, as pf_scatter_rows)
//   pf_emd_init                 inputs of the auction: price 0, assignment / inverse -1 (metric/emd/emd_module.py:45-56)
//   pf_pugan_loss_fwd / _bwd    train_pugan.py:52-67: loss = w_logp logp + w_emd sum_b sum_n dist[b, n] / radius[b] + w_cd mean_b cd[b]
//                               from its pieces; backward: the per-point gradient seeds of the EMD and Chamfer backward kernels,
//                               the zero-filled gradient buffers they accumulate into, and d logp
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"

namespace {

constexpr int IW_K = 8, IW_RMAX = 8;

// one thread per original point t: a[t,k,r] = softmax_k w[t,k,r];  u[t R + r, c] = sum_k a[t,k,r] z[b(t) N + idx[t,k], c]
__global__ __launch_bounds__(256) void interp_wsum_fwd_kernel(const float* __restrict__ w, int ldw, const float* __restrict__ z,
                                                             const int* __restrict__ idx, int N, int R, long long T,
                                                             float* __restrict__ a, float* __restrict__ u) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const long long base = (t / N) * N;
    float zz[IW_K][3];
#pragma unroll
    for (int k = 0; k < IW_K; ++k) {
        const float* zp = z + (base + idx[t * IW_K + k]) * 3;
        zz[k][0] = zp[0]; zz[k][1] = zp[1]; zz[k][2] = zp[2];
    }
    for (int r = 0; r < R; ++r) {
        float m = -__builtin_inff();
#pragma unroll
        for (int k = 0; k < IW_K; ++k) m = fmaxf(m, w[(t * IW_K + k) * ldw + r]);
        float e[IW_K], s = 0.f;
#pragma unroll
        for (int k = 0; k < IW_K; ++k) { e[k] = expf(w[(t * IW_K + k) * ldw + r] - m); s += e[k]; }
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
        for (int k = 0; k < IW_K; ++k) {
            const float ak = e[k] / s;
            a[(t * IW_K + k) * R + r] = ak;
            o0 += ak * zz[k][0]; o1 += ak * zz[k][1]; o2 += ak * zz[k][2];
        }
        float* up = u + (t * R + r) * 3;
        up[0] = o0; up[1] = o1; up[2] = o2;
    }
}
// dw[t,k,r] = a (da - sum_k a da), da[t,k,r] = sum_c du[t R + r, c] z_k[c] (dw beyond R = 0);  dz[neighbour] += sum_r a du
__global__ __launch_bounds__(256) void interp_wsum_bwd_kernel(const float* __restrict__ a, const float* __restrict__ z,
                                                             const int* __restrict__ idx, const float* __restrict__ du, int N,
                                                             int R, int ldw, long long T, float* __restrict__ dw,
                                                             float* __restrict__ dz) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const long long base = (t / N) * N;
    float zz[IW_K][3], g[IW_K][3];
    int jj[IW_K];
#pragma unroll
    for (int k = 0; k < IW_K; ++k) {
        jj[k] = idx[t * IW_K + k];
        const float* zp = z + (base + jj[k]) * 3;
        zz[k][0] = zp[0]; zz[k][1] = zp[1]; zz[k][2] = zp[2];
        g[k][0] = g[k][1] = g[k][2] = 0.f;
    }
    for (int r = 0; r < R; ++r) {
        const float* gp = du + (t * R + r) * 3;
        const float g0 = gp[0], g1 = gp[1], g2 = gp[2];
        float da[IW_K], ak[IW_K], dot = 0.f;
#pragma unroll
        for (int k = 0; k < IW_K; ++k) {
            ak[k] = a[(t * IW_K + k) * R + r];
            da[k] = g0 * zz[k][0] + g1 * zz[k][1] + g2 * zz[k][2];
            dot += ak[k] * da[k];
            g[k][0] += ak[k] * g0; g[k][1] += ak[k] * g1; g[k][2] += ak[k] * g2;
        }
#pragma unroll
        for (int k = 0; k < IW_K; ++k) dw[(t * IW_K + k) * ldw + r] = ak[k] * (da[k] - dot);
    }
    for (int r = R; r < ldw; ++r)
#pragma unroll
        for (int k = 0; k < IW_K; ++k) dw[(t * IW_K + k) * ldw + r] = 0.f;
#pragma unroll
    for (int k = 0; k < IW_K; ++k) {
        float* dp = dz + (base + jj[k]) * 3;
        atomicAdd(dp, g[k][0]); atomicAdd(dp + 1, g[k][1]); atomicAdd(dp + 2, g[k][2]);
    }
}
__global__ __launch_bounds__(256) void glue_zero_kernel(float* p, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = 0.f;
}

__global__ __launch_bounds__(256) void emd_init_kernel(float* price, int* assign2, long long Bn) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < 2 * Bn; i += (long long)gridDim.x * 256) {
        if (i < Bn) price[i] = 0.f;
        assign2[i] = -1;
    }
}

// one workgroup: out[0] = loss, out[1] = w_emd emd, out[2] = w_logp logp, out[3] = w_cd cd
__global__ __launch_bounds__(256) void pugan_loss_fwd_kernel(const float* __restrict__ logp, const float* __restrict__ dist,
                                                            const float* __restrict__ radius, const float* __restrict__ per,
                                                            int B, int n, float w_logp, float w_emd, float w_cd,
                                                            float* __restrict__ out) {
    __shared__ float sh[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float emd = 0.f;
    for (int b0 = 0; b0 < B; b0 += 4) {                  // a wave per sample, samples in order: fixed summation order
        const int b = b0 + wave;
        float s = 0.f;
        if (b < B)
            for (int i = lane; i < n; i += 64) s += dist[(size_t)b * n + i];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
        if (lane == 0) sh[wave] = b < B ? s / (radius ? radius[b] : 1.f) : 0.f;
        __syncthreads();
        if (threadIdx.x == 0) emd += (sh[0] + sh[1]) + (sh[2] + sh[3]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float cd = 0.f;
        for (int b = 0; b < B; ++b) cd += per[b];
        cd /= (float)B;
        const float l = logp[0];
        out[1] = w_emd * emd; out[2] = w_logp * l; out[3] = w_cd * cd;
        out[0] = (w_logp * l + w_emd * emd) + w_cd * cd;
    }
}
// seeds of the loss pieces' backward kernels from g = d loss
__global__ __launch_bounds__(256) void pugan_loss_bwd_kernel(const float* __restrict__ g, const float* __restrict__ radius, int B,
                                                            int N, int M, float w_logp, float w_emd, float w_cd,
                                                            float* __restrict__ graddist, float* __restrict__ g1,
                                                            float* __restrict__ g2, float* __restrict__ dlogp,
                                                            float* __restrict__ gx, float* __restrict__ gy) {
    const float gv = g[0];
    const long long nx = (long long)B * N, ny = (long long)B * M;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < 3 * (nx > ny ? nx : ny); i += (long long)gridDim.x * 256) {
        if (i < nx) {
            graddist[i] = gv * w_emd / (radius ? radius[i / N] : 1.f);
            g1[i] = gv * w_cd / ((float)B * (float)N);
        }
        if (i < ny) g2[i] = gv * w_cd / ((float)B * (float)M);
        if (i < 3 * nx) gx[i] = 0.f;
        if (i < 3 * ny) gy[i] = 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) dlogp[0] = gv * w_logp;
}

inline unsigned glue_grid(long long n) {
    long long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

// w [T, 8, ldw] logits (first R channels), z [B N, 3], idx [T, 8] batch-local neighbours (T = B N) -> a [T, 8, R], u [T R, 3]
extern "C" int pf_interp_wsum_fwd(const float* w, int ldw, const float* z, const int* idx, int N, int K, int R, long long T,
                                  float* a, float* u, void* stream) {
    if (!w || !z || !idx || !a || !u) return PF_ERR_NULL;
    if (K != IW_K || R <= 0 || R > IW_RMAX || R > ldw || T <= 0 || N <= 0 || T % N != 0) return PF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(interp_wsum_fwd_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ldw, z, idx,
                       N, R, T, a, u);
    return pf_last_launch_status();
}
// du [T R, 3] -> dw [T, 8, ldw], dz [B N, 3] (zero-filled here, then accumulated)
extern "C" int pf_interp_wsum_bwd(const float* a, const float* z, const int* idx, const float* du, int N, int K, int R, int ldw,
                                  long long T, float* dw, float* dz, void* stream) {
    if (!a || !z || !idx || !du || !dw || !dz) return PF_ERR_NULL;
    if (K != IW_K || R <= 0 || R > IW_RMAX || R > ldw || T <= 0 || N <= 0 || T % N != 0) return PF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(glue_zero_kernel, dim3(glue_grid(T * 3)), dim3(256), 0, (hipStream_t)stream, dz, T * 3);
    hipLaunchKernelGGL(interp_wsum_bwd_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, z, idx, du,
                       N, R, ldw, T, dw, dz);
    return pf_last_launch_status();
}

// price [B n] = 0; assign2 [2][B n] = -1 (assignment, assignment_inv)
extern "C" int pf_emd_init(float* price, int* assign2, long long Bn, void* stream) {
    if (!price || !assign2) return PF_ERR_NULL;
    if (Bn <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(emd_init_kernel, dim3(glue_grid(2 * Bn)), dim3(256), 0, (hipStream_t)stream, price, assign2, Bn);
    return pf_last_launch_status();
}

// logp [1], dist [B, n] (EMD auction), radius [B] (nullable), per [B] (per-sample Chamfer) -> out [4]
extern "C" int pf_pugan_loss_fwd(const float* logp, const float* dist, const float* radius, const float* per, int B, int n,
                                 float w_logp, float w_emd, float w_cd, float* out, void* stream) {
    if (!logp || !dist || !per || !out) return PF_ERR_NULL;
    if (B <= 0 || n <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(pugan_loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logp, dist, radius, per, B, n, w_logp,
                       w_emd, w_cd, out);
    return pf_last_launch_status();
}
// g [1] -> graddist [B, N], g1 [B, N], g2 [B, M], dlogp [1]; gx [B, N, 3], gy [B, M, 3] zero-filled
extern "C" int pf_pugan_loss_bwd(const float* g, const float* radius, int B, int N, int M, float w_logp, float w_emd, float w_cd,
                                 float* graddist, float* g1, float* g2, float* dlogp, float* gx, float* gy, void* stream) {
    if (!g || !graddist || !g1 || !g2 || !dlogp || !gx || !gy) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || M <= 0) return PF_ERR_SHAPE;
    const long long n = 3LL * B * (N > M ? N : M);
    hipLaunchKernelGGL(pugan_loss_bwd_kernel, dim3(glue_grid(n)), dim3(256), 0, (hipStream_t)stream, g, radius, B, N, M, w_logp,
                       w_emd, w_cd, graddist, g1, g2, dlogp, gx, gy);
    return pf_last_launch_status();
}
