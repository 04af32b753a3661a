// The element-wise half of a flow block in the TRAINING step, fused per direction.
//
// Reference: modules/discrete/interpflow.py:46-82 (FlowBlock: ActNorm -> invertible 3x3 linear -> affine coupling ->
// reverse permutation -> conditional affine injector, forward and exact inverse), modules/utils/normalize.py:28-54,
// permutate.py:77-124, coupling.py:55-137.  The conditioner MLPs are csrc/train_mlp.hip; what remains per block and
// direction is a chain of [rows, 3] element-wise steps and a few 3-wide reductions for the parameter gradients.  The
// un-fused path ran it as ~14 launches per block and direction plus ~25 tiny torch launches for the parameter-only
// scalars (3x3 inverse, log-determinant); at these sizes every one of them is pure launch latency.  Here:
//
//   pf_flow_params     W, logs -> W^-1, (sum(logs) + log|det W|) N                 one thread; backward likewise
//   pf_flow_affine     f: y = W (x e^logs + bias)            g: y = (W^-1 [x_head, x_tail + o] - bias) e^-logs
//                      backward: dx (, do) and the 15 parameter-gradient sums (logs, bias, matrix) reduced in the kernel
//   pf_couple_inject   out = (reverse([y_head, y_tail - o]) - t) e^-s  and  sum(s)   (the log-det term)
//   pf_inject_inv      v = reverse(u e^s + t) with s, t of the ORIGINAL point (row / R): the replicated tensors of the
//                      reference (repeat_interleave, interpflow.py:319) are never built; backward sums ds, dt over the R rows
//
// Reductions: per-workgroup partial sums written with device-scope atomic stores, the workgroup that arrives last (atomic
// counter) adds them up in a fixed order - deterministic, no second launch and no device-wide fence (a release fence
// writes the whole L2 back on this multi-die part: tens of microseconds per launch, see csrc/train_fused.hip).
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"

namespace {

constexpr int FL_MAXWG = 256;

// sums of NV per-thread values over the whole grid -> fin(lane-local array) on the last workgroup; partial: [gridDim.x][NV]
template <int NV, typename FIN>
__device__ __forceinline__ void grid_sums(float (&v)[NV], float* partial, unsigned* counter, FIN fin) {
    __shared__ float red[4][NV];
    __shared__ int last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float s = v[i];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        const float s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        __hip_atomic_store(partial + (size_t)blockIdx.x * NV + threadIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(counter, 1u) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!last) return;
    // fixed-order sum of the gridDim.x partials: 16 threads per value (one strided subset each), then a tree over the 16
    __shared__ float fs[16][16];
    {
        const int vi = threadIdx.x & 15, sub = threadIdx.x >> 4;                 // NV <= 16
        float s = 0.f;
        if (vi < NV)
            for (unsigned w = sub; w < gridDim.x; w += 16)
                s += __hip_atomic_load(partial + (size_t)w * NV + vi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fs[sub][vi] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += fs[k][threadIdx.x];
        fin(threadIdx.x, s);
    }
    if (threadIdx.x == 0) *counter = 0u;
}

inline unsigned flow_grid(long long R) {
    long long g = (R + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > FL_MAXWG ? FL_MAXWG : g));
}

// ---------------------------------------------------------------------------------------------- parameters
__global__ void flow_params_fwd_kernel(const float* W, const float* logs, float n, float* Winv, float* ld) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float a = W[0], b = W[1], c = W[2], d = W[3], e = W[4], f = W[5], g = W[6], h = W[7], i = W[8];
    const float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;       // cofactors of row 0
    const float det = a * c00 + b * c01 + c * c02;
    const float r = 1.f / det;
    Winv[0] = c00 * r; Winv[1] = (c * h - b * i) * r; Winv[2] = (b * f - c * e) * r;
    Winv[3] = c01 * r; Winv[4] = (a * i - c * g) * r; Winv[5] = (c * d - a * f) * r;
    Winv[6] = c02 * r; Winv[7] = (b * g - a * h) * r; Winv[8] = (a * e - b * d) * r;
    ld[0] = (logs[0] + logs[1] + logs[2] + logf(fabsf(det))) * n;
}
// dW = dld n W^-T - W^-T dWinv W^-T ;  dlogs = dld n
__global__ void flow_params_bwd_kernel(const float* Winv, const float* dWinv, const float* dld, float n, float* dW, float* dlogs) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float gl = dld ? dld[0] * n : 0.f;
    float t[9];                                                     // t = Winv^T dWinv
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0.f;
            if (dWinv)
                for (int k = 0; k < 3; ++k) s += Winv[k * 3 + i] * dWinv[k * 3 + j];
            t[i * 3 + j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0.f;
            for (int k = 0; k < 3; ++k) s += t[i * 3 + k] * Winv[j * 3 + k];       // (t Winv^T)[i][j]
            dW[i * 3 + j] = gl * Winv[j * 3 + i] - s;
        }
    dlogs[0] = dlogs[1] = dlogs[2] = gl;
}

// ---------------------------------------------------------------------------------------------- ActNorm + 3x3 linear
struct AffArgs {
    const float* x; const float* o; int td;
    const float* logs; const float* bias; const float* M;
    int inv; long long R;
    float* y;
    const float* dy; float* dx; float* dobuf;
    float* dlogs; float* dbias; float* dM;
    float* partial; unsigned* counter;
};
__global__ __launch_bounds__(256) void flow_affine_fwd_kernel(AffArgs a) {
    float M[9], el[3], b[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = a.M[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) { el[i] = expf(a.inv ? -a.logs[i] : a.logs[i]); b[i] = a.bias[i]; }
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < a.R; r += (long long)gridDim.x * 256) {
        float x[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] = a.x[r * 3 + c];
        if (a.o)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (c >= a.td) x[c] += a.o[r * (3 - a.td) + (c - a.td)];
        float y[3];
        if (!a.inv) {
            float t[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) t[c] = fmaf(x[c], el[c], b[c]);
#pragma unroll
            for (int i = 0; i < 3; ++i) y[i] = M[i * 3] * t[0] + M[i * 3 + 1] * t[1] + M[i * 3 + 2] * t[2];
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) y[i] = (M[i * 3] * x[0] + M[i * 3 + 1] * x[1] + M[i * 3 + 2] * x[2] - b[i]) * el[i];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) a.y[r * 3 + c] = y[c];
    }
}
__global__ __launch_bounds__(256) void flow_affine_bwd_kernel(AffArgs a) {
    float M[9], el[3], b[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = a.M[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) { el[i] = expf(a.inv ? -a.logs[i] : a.logs[i]); b[i] = a.bias[i]; }
    float acc[15];                                                  // dlogs[3], dbias[3], dM[9]
#pragma unroll
    for (int i = 0; i < 15; ++i) acc[i] = 0.f;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < a.R; r += (long long)gridDim.x * 256) {
        float x[3], g[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { x[c] = a.x[r * 3 + c]; g[c] = a.dy[r * 3 + c]; }
        if (a.o)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (c >= a.td) x[c] += a.o[r * (3 - a.td) + (c - a.td)];
        float dx[3];
        if (!a.inv) {
            float t[3], dt[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) t[c] = fmaf(x[c], el[c], b[c]);
#pragma unroll
            for (int j = 0; j < 3; ++j) dt[j] = M[j] * g[0] + M[3 + j] * g[1] + M[6 + j] * g[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dx[c] = dt[c] * el[c];
                acc[c] += dt[c] * x[c] * el[c];
                acc[3 + c] += dt[c];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[6 + i * 3 + j] += g[i] * t[j];
        } else {
            float m[3], dm[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                m[i] = M[i * 3] * x[0] + M[i * 3 + 1] * x[1] + M[i * 3 + 2] * x[2] - b[i];
                dm[i] = g[i] * el[i];
                acc[i] -= dm[i] * m[i];
                acc[3 + i] -= dm[i];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[6 + i * 3 + j] += dm[i] * x[j];
#pragma unroll
            for (int j = 0; j < 3; ++j) dx[j] = M[j] * dm[0] + M[3 + j] * dm[1] + M[6 + j] * dm[2];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a.dx[r * 3 + c] = dx[c];
            if (a.dobuf && c >= a.td) a.dobuf[r * (3 - a.td) + (c - a.td)] = dx[c];
        }
    }
    grid_sums<15>(acc, a.partial, a.counter, [&](int i, float s) {
        if (i < 3) a.dlogs[i] = s;
        else if (i < 6) a.dbias[i - 3] = s;
        else a.dM[i - 6] = s;
    });
}

// ---------------------------------------------------------------------------------------------- coupling + injector, forward direction
struct CiArgs {
    const float* y; const float* o; const float* s; const float* t; int td; long long R;
    float* out; float* ssum;
    const float* dout; const float* dssum;
    float* dy; float* dobuf; float* ds; float* dt;
    float* partial; unsigned* counter;
};
__global__ __launch_bounds__(256) void couple_inject2_fwd_kernel(CiArgs a) {
    float acc[1] = {0.f};
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < a.R; r += (long long)gridDim.x * 256) {
        float h[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) h[c] = a.y[r * 3 + c] - (c >= a.td ? a.o[r * (3 - a.td) + (c - a.td)] : 0.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float sv = a.s[r * 3 + c];
            a.out[r * 3 + c] = (h[2 - c] - a.t[r * 3 + c]) * expf(-sv);
            acc[0] += sv;
        }
    }
    grid_sums<1>(acc, a.partial, a.counter, [&](int, float s) { a.ssum[0] = s; });
}
__global__ __launch_bounds__(256) void couple_inject2_bwd_kernel(CiArgs a) {
    const float gs = a.dssum ? a.dssum[0] : 0.f;
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < a.R; r += (long long)gridDim.x * 256) {
        float dv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = a.dout[r * 3 + c];
            dv[c] = g * expf(-a.s[r * 3 + c]);
            a.ds[r * 3 + c] = gs - g * a.out[r * 3 + c];
            a.dt[r * 3 + c] = -dv[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = dv[2 - c];
            a.dy[r * 3 + c] = g;
            if (c >= a.td) a.dobuf[r * (3 - a.td) + (c - a.td)] = -g;
        }
    }
}

// ---------------------------------------------------------------------------------------------- injector, inverse direction
// v[row] = reverse(u[row] e^{s[row / Rr]} + t[row / Rr]);  backward: du, and ds / dt summed over the Rr rows of a point
__global__ __launch_bounds__(256) void inject_inv2_fwd_kernel(const float* u, const float* s, const float* t, int Rr, long long R,
                                                              float* v) {
    for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < R; r += (long long)gridDim.x * 256) {
        const long long p = r / Rr;
#pragma unroll
        for (int c = 0; c < 3; ++c) v[r * 3 + (2 - c)] = fmaf(u[r * 3 + c], expf(s[p * 3 + c]), t[p * 3 + c]);
    }
}
__global__ __launch_bounds__(256) void inject_inv2_bwd_kernel(const float* u, const float* s, const float* dv, int Rr, long long T,
                                                              float* du, float* ds, float* dt) {
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < T; p += (long long)gridDim.x * 256) {
        float e[3], as[3] = {0.f, 0.f, 0.f}, at[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c) e[c] = expf(s[p * 3 + c]);
        for (int k = 0; k < Rr; ++k) {
            const long long r = p * Rr + k;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float g = dv[r * 3 + (2 - c)];
                du[r * 3 + c] = g * e[c];
                as[c] += g * u[r * 3 + c] * e[c];
                at[c] += g;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { ds[p * 3 + c] = as[c]; dt[p * 3 + c] = at[c]; }
    }
}

}  // namespace

// W [3,3], logs [3] -> Winv [3,3], ld [1] = (sum(logs) + log|det W|) * n       (normalize.py:34-36, permutate.py:118-119)
extern "C" int pf_flow_params_fwd(const float* W, const float* logs, float n, float* Winv, float* ld, void* stream) {
    if (!W || !logs || !Winv || !ld) return PF_ERR_NULL;
    hipLaunchKernelGGL(flow_params_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, W, logs, n, Winv, ld);
    return pf_last_launch_status();
}
// dWinv, dld nullable (no gradient from that output)
extern "C" int pf_flow_params_bwd(const float* Winv, const float* dWinv, const float* dld, float n, float* dW, float* dlogs,
                                  void* stream) {
    if (!Winv || !dW || !dlogs) return PF_ERR_NULL;
    hipLaunchKernelGGL(flow_params_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, Winv, dWinv, dld, n, dW, dlogs);
    return pf_last_launch_status();
}

// inv = 0: y = M (x e^logs + bias)                      (ActNorm then the invertible 3x3 linear, M = W)
// inv = 1: y = (M [x_head, x_tail + o] - bias) e^-logs   (their inverses, M = W^-1; o nullable: [R, 3 - td] coupling shift)
extern "C" int pf_flow_affine_fwd(const float* x, const float* o, int td, const float* logs, const float* bias, const float* M,
                                  int inv, long long R, float* y, void* stream) {
    if (!x || !logs || !bias || !M || !y) return PF_ERR_NULL;
    if (R <= 0 || td < 0 || td > 3) return PF_ERR_SHAPE;
    AffArgs a{};
    a.x = x; a.o = o; a.td = td; a.logs = logs; a.bias = bias; a.M = M; a.inv = inv; a.R = R; a.y = y;
    hipLaunchKernelGGL(flow_affine_fwd_kernel, dim3(flow_grid(R)), dim3(256), 0, (hipStream_t)stream, a);
    return pf_last_launch_status();
}
// dx [R,3], dobuf [R, 3-td] (nullable), dlogs [3], dbias [3], dM [9]; partial: >= 256*15 floats; counter: one zeroed word
extern "C" int pf_flow_affine_bwd(const float* x, const float* o, int td, const float* logs, const float* bias, const float* M,
                                  int inv, long long R, const float* dy, float* dx, float* dobuf, float* dlogs, float* dbias,
                                  float* dM, float* partial, unsigned* counter, void* stream) {
    if (!x || !logs || !bias || !M || !dy || !dx || !dlogs || !dbias || !dM || !partial || !counter) return PF_ERR_NULL;
    if (R <= 0 || td < 0 || td > 3) return PF_ERR_SHAPE;
    AffArgs a{};
    a.x = x; a.o = o; a.td = td; a.logs = logs; a.bias = bias; a.M = M; a.inv = inv; a.R = R;
    a.dy = dy; a.dx = dx; a.dobuf = dobuf; a.dlogs = dlogs; a.dbias = dbias; a.dM = dM; a.partial = partial; a.counter = counter;
    hipLaunchKernelGGL(flow_affine_bwd_kernel, dim3(flow_grid(R)), dim3(256), 0, (hipStream_t)stream, a);
    return pf_last_launch_status();
}

// out = (reverse([y_head, y_tail - o]) - t) e^-s, ssum [1] = sum(s)    (coupling.py:55-58,114-118,132-137; permutate.py:77)
extern "C" int pf_couple_inject2_fwd(const float* y, const float* o, const float* s, const float* t, int td, long long R,
                                     float* out, float* ssum, float* partial, unsigned* counter, void* stream) {
    if (!y || !o || !s || !t || !out || !ssum || !partial || !counter) return PF_ERR_NULL;
    if (R <= 0 || td < 1 || td > 2) return PF_ERR_SHAPE;
    CiArgs a{};
    a.y = y; a.o = o; a.s = s; a.t = t; a.td = td; a.R = R; a.out = out; a.ssum = ssum; a.partial = partial; a.counter = counter;
    hipLaunchKernelGGL(couple_inject2_fwd_kernel, dim3(flow_grid(R)), dim3(256), 0, (hipStream_t)stream, a);
    return pf_last_launch_status();
}
extern "C" int pf_couple_inject2_bwd(const float* out, const float* dout, const float* dssum, const float* s, int td, long long R,
                                     float* dy, float* dobuf, float* ds, float* dt, void* stream) {
    if (!out || !dout || !s || !dy || !dobuf || !ds || !dt) return PF_ERR_NULL;
    if (R <= 0 || td < 1 || td > 2) return PF_ERR_SHAPE;
    CiArgs a{};
    a.out = const_cast<float*>(out); a.dout = dout; a.dssum = dssum; a.s = s; a.td = td; a.R = R;
    a.dy = dy; a.dobuf = dobuf; a.ds = ds; a.dt = dt;
    hipLaunchKernelGGL(couple_inject2_bwd_kernel, dim3(flow_grid(R)), dim3(256), 0, (hipStream_t)stream, a);
    return pf_last_launch_status();
}

// v [R,3] from u [R,3] and the per-point s, t [R / Rr, 3]     (interpflow.py:319-321 + coupling.py:120-127 + permutate.py:80)
extern "C" int pf_inject_inv2_fwd(const float* u, const float* s, const float* t, int Rr, long long R, float* v, void* stream) {
    if (!u || !s || !t || !v) return PF_ERR_NULL;
    if (R <= 0 || Rr < 1 || R % Rr != 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(inject_inv2_fwd_kernel, dim3(flow_grid(R)), dim3(256), 0, (hipStream_t)stream, u, s, t, Rr, R, v);
    return pf_last_launch_status();
}
extern "C" int pf_inject_inv2_bwd(const float* u, const float* s, const float* dv, int Rr, long long R, float* du, float* ds,
                                  float* dt, void* stream) {
    if (!u || !s || !dv || !du || !ds || !dt) return PF_ERR_NULL;
    if (R <= 0 || Rr < 1 || R % Rr != 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(inject_inv2_bwd_kernel, dim3(flow_grid(R / Rr)), dim3(256), 0, (hipStream_t)stream, u, s, dv, Rr, R / Rr, du,
                       ds, dt);
    return pf_last_launch_status();
}
