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

// one thread per (point t, replica r): a[t,k,r] = softmax_k w[t,k,r];  u[t R + r, c] = sum_k a[t,k,r] z[b(t) N + idx[t,k], c]
__global__ __launch_bounds__(256) void interp_wsum_fwd_kernel(const float* __restrict__ w, int ldw, const float* __restrict__ z,
                                                             const int* __restrict__ idx, int N, int R, long long T,
                                                             float* __restrict__ a, float* __restrict__ u) {
    const long long tr = (long long)blockIdx.x * 256 + threadIdx.x;
    if (tr >= T * R) return;
    const long long t = tr / R;
    const int r = (int)(tr - t * R);
    const long long base = (t / N) * N;
    float lg[IW_K], zz[IW_K][3];
#pragma unroll
    for (int k = 0; k < IW_K; ++k) {
        lg[k] = w[(t * IW_K + k) * ldw + r];
        const float* zp = z + (base + idx[t * IW_K + k]) * 3;
        zz[k][0] = zp[0]; zz[k][1] = zp[1]; zz[k][2] = zp[2];
    }
    float m = lg[0];
#pragma unroll
    for (int k = 1; k < IW_K; ++k) m = fmaxf(m, lg[k]);
    float e[IW_K], s = 0.f;
#pragma unroll
    for (int k = 0; k < IW_K; ++k) { e[k] = expf(lg[k] - m); s += e[k]; }
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
    for (int k = 0; k < IW_K; ++k) {
        const float ak = e[k] / s;
        a[(t * IW_K + k) * R + r] = ak;
        o0 += ak * zz[k][0]; o1 += ak * zz[k][1]; o2 += ak * zz[k][2];
    }
    float* up = u + tr * 3;
    up[0] = o0; up[1] = o1; up[2] = o2;
}
// one thread per (point t, neighbour k), the 8 lanes of a point side by side:
// dw[t,k,r] = a (da - sum_k a da), da[t,k,r] = sum_c du[t R + r, c] z_k[c] (dw beyond R = 0);  dz[neighbour] += sum_r a du
__global__ __launch_bounds__(256) void interp_wsum_bwd_kernel(const float* __restrict__ a, const float* __restrict__ z,
                                                             const int* __restrict__ idx, const float* __restrict__ du, int N,
                                                             int R, int ldw, long long T, float* __restrict__ dw,
                                                             float* __restrict__ dz) {
    const long long tk = (long long)blockIdx.x * 256 + threadIdx.x;      // T * 8 is a multiple of 8: a point's lanes are all in or all out
    if (tk >= T * IW_K) return;
    const long long t = tk / IW_K;
    const long long base = (t / N) * N;
    const int j = idx[tk];
    const float* zp = z + (base + j) * 3;
    const float z0 = zp[0], z1 = zp[1], z2 = zp[2];
    float g0s = 0.f, g1s = 0.f, g2s = 0.f;
    for (int r = 0; r < R; ++r) {
        const float* gp = du + (t * R + r) * 3;
        const float g0 = gp[0], g1 = gp[1], g2 = gp[2];
        const float ak = a[tk * R + r];
        const float da = g0 * z0 + g1 * z1 + g2 * z2;
        float dot = ak * da;
        dot += __shfl_xor(dot, 1); dot += __shfl_xor(dot, 2); dot += __shfl_xor(dot, 4);
        dw[tk * ldw + r] = ak * (da - dot);
        g0s += ak * g0; g1s += ak * g1; g2s += ak * g2;
    }
    for (int r = R; r < ldw; ++r) dw[tk * ldw + r] = 0.f;
    if (!dz) return;                                             // deterministic form: interp_dz_gather_kernel below
    float* dp = dz + (base + j) * 3;
    atomicAdd(dp, g0s); atomicAdd(dp + 1, g1s); atomicAdd(dp + 2, g2s);
}
// dz without float atomics: one thread per point j walks the (sorted) list of the (point t, neighbour k) pairs that point AT j
// (pf_knn_csr of the interpolation's neighbour lists: edge id = t 8 + k) and adds sum_r a du in list order - the same sum, one
// order, run after run
__global__ __launch_bounds__(256) void interp_dz_gather_kernel(const float* __restrict__ a, const float* __restrict__ du,
                                                              const int* __restrict__ off, const int* __restrict__ edge, int R,
                                                              long long T, float* __restrict__ dz) {
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= T) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int q = off[j]; q < off[j + 1]; ++q) {
        const long long tk = edge[q], t = tk / IW_K;
        float g0s = 0.f, g1s = 0.f, g2s = 0.f;
        for (int r = 0; r < R; ++r) {
            const float* gp = du + (t * R + r) * 3;
            const float ak = a[tk * R + r];
            g0s += ak * gp[0]; g1s += ak * gp[1]; g2s += ak * gp[2];
        }
        s0 += g0s; s1 += g1s; s2 += g2s;
    }
    dz[j * 3 + 0] = s0; dz[j * 3 + 1] = s1; dz[j * 3 + 2] = s2;
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

// one workgroup of 16 waves, a wave per sample (strided): out[0] = loss, out[1] = w_emd emd, out[2] = w_logp logp, out[3] = w_cd cd
__global__ __launch_bounds__(1024) void pugan_loss_fwd_kernel(const float* __restrict__ logp, const float* __restrict__ dist,
                                                             const float* __restrict__ radius, const float* __restrict__ per,
                                                             int B, int n, float w_logp, float w_emd, float w_cd,
                                                             float* __restrict__ out) {
    __shared__ float part[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;                                     // this wave's samples, in order
    for (int b = wave; b < B; b += 16) {
        const float* dp = dist + (size_t)b * n;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int i = lane;
        for (; i + 192 < n; i += 256) { s0 += dp[i]; s1 += dp[i + 64]; s2 += dp[i + 128]; s3 += dp[i + 192]; }
        for (; i < n; i += 64) s0 += dp[i];
        float sv = (s0 + s1) + (s2 + s3);
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) sv += __shfl_xor(sv, m);
        acc += sv / (radius ? radius[b] : 1.f);
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float emd = 0.f;
        for (int k = 0; k < 16; ++k) emd += part[k];
        float cd = 0.f;
        if (per) {
            for (int b = 0; b < B; ++b) cd += per[b];
            cd /= (float)B;
        }
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

// The prediction's gradient of the whole loss head in two launches (it was four: seeds + zero fill, two Chamfer directions, EMD):
//   own terms, plain stores:   gx[b,i] = 2 gd_b (x_i - y_assign(i)) + 2 g1 (x_i - y_idx1(i)),   gd_b = g w_emd / radius_b, g1 = g w_cd / (B N)
//   (emd_cuda.cu:284-300; the first Chamfer direction, d dist1 / d x)
// then the second Chamfer direction scattered onto it with float atomics:  gx[b, idx2(j)] -= 2 g2 (y_j - x_idx2(j)),  g2 = g w_cd / (B M).
// For the default path only: the ground truth gets no gradient here and the sums come in arrival order (PuganLossFn keeps the
// four-launch form for a ground truth that wants its gradient and for the bit-reproducible mode).
__global__ __launch_bounds__(256) void pugan_grad_own_kernel(const float* __restrict__ g, const float* __restrict__ radius,
                                                            const float* __restrict__ x, const float* __restrict__ y,
                                                            const int* __restrict__ assign, const int* __restrict__ idx1, int B, int n,
                                                            float w_logp, float w_emd, float w_cd, float* __restrict__ gx,
                                                            float* __restrict__ dlogp) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const float gv = g[0];
    if (t == 0) dlogp[0] = gv * w_logp;
    if (t >= (long long)B * n) return;
    const long long b = t / n;
    const float ge = 2.f * (gv * w_emd / (radius ? radius[b] : 1.f));
    const int j = assign[t];
    float o[3];
    if ((unsigned)j >= (unsigned)n) {                             // an assignment that is not an index: loud, never out of bounds
        o[0] = o[1] = o[2] = __builtin_nanf("");
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = ge * (x[t * 3 + c] - y[(b * n + j) * 3 + c]);
    }
    if (idx1) {
        const float w = 2.f * (gv * w_cd / ((float)B * (float)n));
        const int j1 = idx1[t];
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] += w * (x[t * 3 + c] - y[(b * n + j1) * 3 + c]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) gx[t * 3 + c] = o[c];
}
__global__ __launch_bounds__(256) void pugan_grad_other_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                              const float* __restrict__ y, const int* __restrict__ idx2, int B, int n,
                                                              int m, float w_cd, float* __restrict__ gx) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)B * m) return;
    const long long b = t / m;
    const float w = 2.f * (g[0] * w_cd / ((float)B * (float)m));
    const int j = idx2[t];
#pragma unroll
    for (int c = 0; c < 3; ++c) atomicAdd(&gx[(b * n + j) * 3 + c], -(w * (y[t * 3 + c] - x[(b * n + j) * 3 + c])));
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
    hipLaunchKernelGGL(interp_wsum_fwd_kernel, dim3((unsigned)((T * R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ldw, z,
                       idx, N, R, T, a, u);
    return pf_last_launch_status();
}
// du [T R, 3] -> dw [T, 8, ldw], dz [B N, 3] (zero-filled here, then accumulated)
extern "C" int pf_interp_wsum_bwd(const float* a, const float* z, const int* idx, const float* du, int N, int K, int R, int ldw,
                                  long long T, float* dw, float* dz, void* stream) {
    return pf_interp_wsum_bwd_det(a, z, idx, du, N, K, R, ldw, T, dw, dz, nullptr, nullptr, stream);
}
// csr_off / csr_edge (pf_knn_csr of idx, both or neither): dz as a gather in a fixed order instead of float atomics
extern "C" int pf_interp_wsum_bwd_det(const float* a, const float* z, const int* idx, const float* du, int N, int K, int R, int ldw,
                                      long long T, float* dw, float* dz, const int* csr_off, const int* csr_edge, void* stream) {
    if (!a || !z || !idx || !du || !dw || !dz) return PF_ERR_NULL;
    if (K != IW_K || R <= 0 || R > IW_RMAX || R > ldw || T <= 0 || N <= 0 || T % N != 0) return PF_ERR_UNSUPPORTED;
    if ((csr_off == nullptr) != (csr_edge == nullptr)) return PF_ERR_NULL;
    const bool det = csr_off != nullptr;
    if (!det) hipLaunchKernelGGL(glue_zero_kernel, dim3(glue_grid(T * 3)), dim3(256), 0, (hipStream_t)stream, dz, T * 3);
    hipLaunchKernelGGL(interp_wsum_bwd_kernel, dim3((unsigned)((T * IW_K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, z,
                       idx, du, N, R, ldw, T, dw, det ? (float*)nullptr : dz);
    if (det)
        hipLaunchKernelGGL(interp_dz_gather_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, du, csr_off,
                           csr_edge, R, T, dz);
    return pf_last_launch_status();
}

// price [B n] = 0; assign2 [2][B n] = -1 (assignment, assignment_inv)
extern "C" int pf_emd_init(float* price, int* assign2, long long Bn, void* stream) {
    if (!price || !assign2) return PF_ERR_NULL;
    if (Bn <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(emd_init_kernel, dim3(glue_grid(2 * Bn)), dim3(256), 0, (hipStream_t)stream, price, assign2, Bn);
    return pf_last_launch_status();
}

// g [1] = d loss; pred x [B, n, 3], ground truth y [B, m, 3] (m = n for the EMD), assign [B, n] (the auction's assignment), idx1
// [B, n] / idx2 [B, m] (Chamfer's nearest neighbours; both NULL: no Chamfer term), radius [B] (nullable) -> gx [B, n, 3], dlogp [1]
extern "C" int pf_pugan_grad(const float* g, const float* radius, const float* x, const float* y, const int* assign, const int* idx1,
                             const int* idx2, int B, int n, int m, float w_logp, float w_emd, float w_cd, float* gx, float* dlogp,
                             void* stream) {
    if (!g || !x || !y || !assign || !gx || !dlogp || (idx1 == nullptr) != (idx2 == nullptr)) return PF_ERR_NULL;
    if (B <= 0 || n <= 0 || m != n) return PF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const long long t1 = (long long)B * n, t2 = (long long)B * m;
    hipLaunchKernelGGL(pugan_grad_own_kernel, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, s, g, radius, x, y, assign, idx1, B, n,
                       w_logp, w_emd, w_cd, gx, dlogp);
    if (idx2)
        hipLaunchKernelGGL(pugan_grad_other_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, s, g, x, y, idx2, B, n, m, w_cd, gx);
    return pf_last_launch_status();
}

// logp [1], dist [B, n] (EMD auction), radius [B] (nullable), per [B] (per-sample Chamfer; nullable: no such term) -> out [4]
extern "C" int pf_pugan_loss_fwd(const float* logp, const float* dist, const float* radius, const float* per, int B, int n,
                                 float w_logp, float w_emd, float w_cd, float* out, void* stream) {
    if (!logp || !dist || !out) return PF_ERR_NULL;          // per nullable: no Chamfer term
    if (B <= 0 || n <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(pugan_loss_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logp, dist, radius, per, B, n, w_logp,
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

// ---- the weight unit's first conv folded into its two producers' last linear layers (train_ops.py interp_weights) ------------
// WeightEstimationUnit.mlp[0] (interpflow.py:144) sees cat[d, feat] where d = DistanceEncoder's last conv (W6, b6: linear,
// interpflow.py:98) and feat = the EdgeConv's conv_out (Wout, bout: linear, :219-221) - no nonlinearity in between.  So
//   W0 [d; feat] + b0 = (W0a W6) a2 + (W0b Wout) e + (W0a b6 + W0b bout + b0),   W0 = [W0a | W0b]  ([o, 2 o]).
// Forward: the four product tensors; backward: the chain rule back to W0, b0, W6, b6, Wout, bout.  A few hundred thousand MACs:
// one thread per output element, fixed summation order (bit-reproducible, capturable), two launches per step.
namespace {
struct FoldWuArgs {
    const float* W0; const float* b0; const float* W6; const float* b6; const float* Wout; const float* bout;
    int o, k6, ko;                       // W0 [o, 2 o], W6 [o, k6], Wout [o, ko]
    float* W6f; float* b6f; float* Wof; float* bof;                       // forward outputs [o, k6], [o], [o, ko], [o]
    const float* dW6f; const float* db6f; const float* dWof; const float* dbof;
    float* dW0; float* db0; float* dW6; float* db6; float* dWout; float* dbout;
};
// sum_k x[k sx] y[k sy], k < n: four interleaved partial sums (k mod 4) added as (s0 + s1) + (s2 + s3) - a fixed order - with eight
// products' loads in flight (one dependent fma per L2 round trip made the naive loop 46 us for 60 k outputs)
__device__ __forceinline__ float fold_dot(const float* __restrict__ x, int sx, const float* __restrict__ y, int sy, int n) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 7 < n; k += 8) {
        float xv[8], yv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { xv[u] = x[(size_t)(k + u) * sx]; yv[u] = y[(size_t)(k + u) * sy]; }
        s0 = fmaf(xv[0], yv[0], s0); s1 = fmaf(xv[1], yv[1], s1); s2 = fmaf(xv[2], yv[2], s2); s3 = fmaf(xv[3], yv[3], s3);
        s0 = fmaf(xv[4], yv[4], s0); s1 = fmaf(xv[5], yv[5], s1); s2 = fmaf(xv[6], yv[6], s2); s3 = fmaf(xv[7], yv[7], s3);
    }
    for (; k < n; ++k) {
        const float p = x[(size_t)k * sx] * y[(size_t)k * sy];
        if ((k & 3) == 0) s0 += p; else if ((k & 3) == 1) s1 += p; else if ((k & 3) == 2) s2 += p; else s3 += p;
    }
    return (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void fold_wu_fwd_kernel(FoldWuArgs a) {
    const int o = a.o, n6 = o * a.k6, no = o * a.ko;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n6 + no + 2 * o; i += gridDim.x * 256) {
        if (i < n6) {                                                       // W6f[r, c] = sum_m W0a[r, m] W6[m, c]
            const int r = i / a.k6, c = i % a.k6;
            a.W6f[i] = fold_dot(a.W0 + (size_t)r * 2 * o, 1, a.W6 + c, a.k6, o);
        } else if (i < n6 + no) {                                           // Wof[r, c] = sum_m W0b[r, m] Wout[m, c]
            const int j = i - n6, r = j / a.ko, c = j % a.ko;
            a.Wof[j] = fold_dot(a.W0 + (size_t)r * 2 * o + o, 1, a.Wout + c, a.ko, o);
        } else if (i < n6 + no + o) {                                       // b6f = W0a b6 + b0
            const int r = i - n6 - no;
            a.b6f[r] = fold_dot(a.W0 + (size_t)r * 2 * o, 1, a.b6, 1, o) + a.b0[r];
        } else {                                                            // bof = W0b bout
            const int r = i - n6 - no - o;
            a.bof[r] = fold_dot(a.W0 + (size_t)r * 2 * o + o, 1, a.bout, 1, o);
        }
    }
}
__global__ __launch_bounds__(256) void fold_wu_bwd_kernel(FoldWuArgs a) {
    const int o = a.o, k6 = a.k6, ko = a.ko;
    const int n0 = o * 2 * o, n6 = o * k6, no = o * ko;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n0 + n6 + no + 3 * o; i += gridDim.x * 256) {
        if (i < n0) {                                                       // dW0[r, m]
            const int r = i / (2 * o), m = i % (2 * o);
            if (m < o)                                                      // dW0a = dW6f W6^T + db6f (x) b6
                a.dW0[i] = fmaf(a.db6f[r], a.b6[m], fold_dot(a.dW6f + (size_t)r * k6, 1, a.W6 + (size_t)m * k6, 1, k6));
            else                                                            // dW0b = dWof Wout^T + dbof (x) bout
                a.dW0[i] = fmaf(a.dbof[r], a.bout[m - o], fold_dot(a.dWof + (size_t)r * ko, 1, a.Wout + (size_t)(m - o) * ko, 1, ko));
        } else if (i < n0 + n6) {                                           // dW6[m, c] = sum_r W0a[r, m] dW6f[r, c]
            const int j = i - n0, m = j / k6, c = j % k6;
            a.dW6[j] = fold_dot(a.W0 + m, 2 * o, a.dW6f + c, k6, o);
        } else if (i < n0 + n6 + no) {                                      // dWout[m, c] = sum_r W0b[r, m] dWof[r, c]
            const int j = i - n0 - n6, m = j / ko, c = j % ko;
            a.dWout[j] = fold_dot(a.W0 + o + m, 2 * o, a.dWof + c, ko, o);
        } else {
            const int j = i - n0 - n6 - no, which = j / o, m = j % o;
            if (which == 0) a.db0[m] = a.db6f[m];
            else if (which == 1) a.db6[m] = fold_dot(a.W0 + m, 2 * o, a.db6f, 1, o);          // db6 = W0a^T db6f
            else a.dbout[m] = fold_dot(a.W0 + o + m, 2 * o, a.dbof, 1, o);                      // dbout = W0b^T dbof
        }
    }
}
}  // namespace

extern "C" int pf_fold_wu_fwd(const float* W0, const float* b0, const float* W6, const float* b6, const float* Wout,
                              const float* bout, int o, int k6, int ko, float* W6f, float* b6f, float* Wof, float* bof,
                              void* stream) {
    if (!W0 || !b0 || !W6 || !b6 || !Wout || !bout || !W6f || !b6f || !Wof || !bof) return PF_ERR_NULL;
    if (o < 1 || o > 1024 || k6 < 1 || ko < 1 || k6 > 4096 || ko > 4096) return PF_ERR_SHAPE;
    FoldWuArgs a{};
    a.W0 = W0; a.b0 = b0; a.W6 = W6; a.b6 = b6; a.Wout = Wout; a.bout = bout; a.o = o; a.k6 = k6; a.ko = ko;
    a.W6f = W6f; a.b6f = b6f; a.Wof = Wof; a.bof = bof;
    const int n = o * (k6 + ko + 2);
    hipLaunchKernelGGL(fold_wu_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return pf_last_launch_status();
}

extern "C" int pf_fold_wu_bwd(const float* W0, const float* W6, const float* b6, const float* Wout, const float* bout, int o,
                              int k6, int ko, const float* dW6f, const float* db6f, const float* dWof, const float* dbof,
                              float* dW0, float* db0, float* dW6, float* db6, float* dWout, float* dbout, void* stream) {
    if (!W0 || !W6 || !b6 || !Wout || !bout || !dW6f || !db6f || !dWof || !dbof || !dW0 || !db0 || !dW6 || !db6 || !dWout || !dbout)
        return PF_ERR_NULL;
    if (o < 1 || o > 1024 || k6 < 1 || ko < 1 || k6 > 4096 || ko > 4096) return PF_ERR_SHAPE;
    FoldWuArgs a{};
    a.W0 = W0; a.W6 = W6; a.b6 = b6; a.Wout = Wout; a.bout = bout; a.o = o; a.k6 = k6; a.ko = ko;
    a.dW6f = dW6f; a.db6f = db6f; a.dWof = dWof; a.dbof = dbof;
    a.dW0 = dW0; a.db0 = db0; a.dW6 = dW6; a.db6 = db6; a.dWout = dWout; a.dbout = dbout;
    const int n = o * (2 * o + k6 + ko + 3);
    hipLaunchKernelGGL(fold_wu_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return pf_last_launch_status();
}

// ------------------------------------------------------------------------------------------------ gradient fan-in
// out = ((p0 + p1) + p2) + ... over n <= 8 tensors of the same size, 16-byte loads, ONE launch: a tensor with several consumers
// (the flattened conditioning features: both flow chains and the two families of injector nets) gets its gradient from one
// pass over the operands instead of autograd's n - 1 pairwise adds (each a full read-read-write of the running sum).
namespace {
typedef float sum_f4 __attribute__((ext_vector_type(4)));
struct SumNArgs { const sum_f4* p[8]; sum_f4* out; long long n4; };
template <int N>
__global__ __launch_bounds__(256) void sum_n_kernel(SumNArgs a) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += (long long)gridDim.x * 256) {
        sum_f4 v[N];
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = __builtin_nontemporal_load(a.p[j] + i);
        sum_f4 s = v[0];
#pragma unroll
        for (int j = 1; j < N; ++j) s += v[j];
        a.out[i] = s;
    }
}
}  // namespace

extern "C" int pf_sum_n(const float* const* ptrs, int n_terms, float* out, long long n, void* stream) {
    if (!ptrs || !out) return PF_ERR_NULL;
    if (n_terms < 2 || n_terms > 8 || n <= 0 || n % 4 != 0) return PF_ERR_SHAPE;
    SumNArgs a{};
    for (int j = 0; j < n_terms; ++j) {
        if (!ptrs[j]) return PF_ERR_NULL;
        if (((uintptr_t)ptrs[j]) & 15) return PF_ERR_SHAPE;
        a.p[j] = reinterpret_cast<const sum_f4*>(ptrs[j]);
    }
    if (((uintptr_t)out) & 15) return PF_ERR_SHAPE;
    a.out = reinterpret_cast<sum_f4*>(out); a.n4 = n / 4;
    const long long want = (a.n4 + 255) / 256;
    const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
    hipStream_t s = (hipStream_t)stream;
    switch (n_terms) {
        case 2: hipLaunchKernelGGL(sum_n_kernel<2>, dim3(grid), dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(sum_n_kernel<3>, dim3(grid), dim3(256), 0, s, a); break;
        case 4: hipLaunchKernelGGL(sum_n_kernel<4>, dim3(grid), dim3(256), 0, s, a); break;
        case 5: hipLaunchKernelGGL(sum_n_kernel<5>, dim3(grid), dim3(256), 0, s, a); break;
        case 6: hipLaunchKernelGGL(sum_n_kernel<6>, dim3(grid), dim3(256), 0, s, a); break;
        case 7: hipLaunchKernelGGL(sum_n_kernel<7>, dim3(grid), dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(sum_n_kernel<8>, dim3(grid), dim3(256), 0, s, a); break;
    }
    return pf_last_launch_status();
}

// ------------------------------------------------------------------------------------------------ batch hand-over
// n <= 8 contiguous regions of 32-bit words copied by ONE launch (blockIdx.y = region): a captured training step takes its batch
// (points, ground truth, radii, ...) into the tensors the graph reads - one copy launch per tensor before, each with its own
// submission in front of the replay.
namespace {
struct CopyNArgs { const unsigned* src[8]; unsigned* dst[8]; long long words[8]; };
__global__ __launch_bounds__(256) void copy_n_kernel(CopyNArgs a) {
    const int r = blockIdx.y;
    const unsigned* s = a.src[r];
    unsigned* d = a.dst[r];
    const long long n = a.words[r];
    const bool vec = ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0;
    const long long n4 = vec ? n / 4 : 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256)
        reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
    for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) d[i] = s[i];
}
}  // namespace

extern "C" int pf_copy_n(const void* const* src, void* const* dst, const long long* words, int n, void* stream) {
    if (!src || !dst || !words) return PF_ERR_NULL;
    if (n < 1 || n > 8) return PF_ERR_SHAPE;
    CopyNArgs a{};
    long long most = 0;
    for (int j = 0; j < n; ++j) {
        if (!src[j] || !dst[j]) return PF_ERR_NULL;
        if (words[j] <= 0 || ((((uintptr_t)src[j]) | ((uintptr_t)dst[j])) & 3)) return PF_ERR_SHAPE;
        a.src[j] = static_cast<const unsigned*>(src[j]); a.dst[j] = static_cast<unsigned*>(dst[j]); a.words[j] = words[j];
        most = words[j] > most ? words[j] : most;
    }
    const long long want = (most / 4 + 255) / 256;
    const unsigned gx = (unsigned)(want < 1 ? 1 : (want > 512 ? 512 : want));
    hipLaunchKernelGGL(copy_n_kernel, dim3(gx, n), dim3(256), 0, (hipStream_t)stream, a);
    return pf_last_launch_status();
}
