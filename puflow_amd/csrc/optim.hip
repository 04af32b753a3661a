// Gradient clipping + Adam for the whole model in two launches.
//
// Reference: Lightning's `gradient_clip_val=1e-2` (train_pu1k.py:149 -> torch.nn.utils.clip_grad_norm_, L2 norm over all
// parameters) followed by torch.optim.Adam (train_pu1k.py:46).  In a captured training step PyTorch's capturable Adam fell
// back to per-tensor element-wise kernels: ~550 launches and 2.4 ms per step for 806 103 parameters in ~230 tensors.  Here
// the gradients are one flat buffer (the all-reduce bucket), the moments are flat buffers of the same layout, the parameters
// stay where they are (a table of their addresses), and the update is
//   norm    : sum of squares per chunk -> last workgroup: total norm, clip coefficient min(1, max_norm / (norm + 1e-6)),
//             step counter += 1
//   update  : g *= coef;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;
//             p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)           (torch/optim/adam.py, single-tensor form)
// Chunks never cross a tensor boundary: chunk table rows are (tensor id, offset inside the tensor, length, offset in the flat
// buffers).
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"

namespace {

struct AdamArgs {
    float* g; float* m; float* v;
    float* const* params;            // device array of parameter addresses
    const int* chunks;               // [nchunks][4]
    int nchunks;
    const float* lr; float* step;    // device scalars
    float beta1, beta2, eps, max_norm;
    double* partial; unsigned* counter; float* coef;
    float* const* gtab;              // nullable: device array of the gradient tensors' addresses (parameter order) - g is not read then
};

__global__ __launch_bounds__(256) void adam_norm_kernel(AdamArgs a) {
    __shared__ double red[4];
    __shared__ int last;
    const int c = blockIdx.x;
    const int len = a.chunks[c * 4 + 2], fo = a.chunks[c * 4 + 3];
    const float* gp = a.gtab ? a.gtab[a.chunks[c * 4 + 0]] + a.chunks[c * 4 + 1] : a.g + fo;
    double s = 0.0;
    for (int i = threadIdx.x; i < len; i += 256) { const float x = gp[i]; s += (double)x * (double)x; }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(a.partial + c, (red[0] + red[1]) + (red[2] + red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last = atomicAdd(a.counter, 1u) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!last) return;
    double t = 0.0;
    for (int k = threadIdx.x; k < a.nchunks; k += 256)
        t += __hip_atomic_load(a.partial + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) t += __shfl_xor(t, m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
        const float cf = a.max_norm / (norm + 1e-6f);
        // a non-finite gradient norm (a NaN / inf loss: e.g. the NaN distances a timed-out EMD grid barrier leaves) must not
        // reach the parameters or the moments - one such update makes the whole run unrecoverable, and inside a captured step
        // the host reads the status words only every few replays.  The update is SKIPPED on the device (no step count, moments and
        // parameters untouched) and counted in coef[3]; the late host check only reports it.
        const bool bad = !(norm <= 3.0e38f);
        a.coef[0] = bad ? 0.f : (cf < 1.f ? cf : 1.f);
        a.coef[1] = norm;
        a.coef[2] = bad ? 1.f : 0.f;
        if (bad) a.coef[3] += 1.f;
        else a.step[0] += 1.f;
        *a.counter = 0u;
    }
}

__global__ __launch_bounds__(256) void adam_update_kernel(AdamArgs a) {
    const int c = blockIdx.x;
    const int tid = a.chunks[c * 4 + 0], lo = a.chunks[c * 4 + 1], len = a.chunks[c * 4 + 2], fo = a.chunks[c * 4 + 3];
    float* p = a.params[tid] + lo;
    float* gp = a.gtab ? a.gtab[tid] + lo : a.g + fo;
    if (a.coef[2] != 0.f) return;                         // non-finite gradient norm: this update is skipped (adam_norm_kernel)
    const float cf = a.coef[0], lr = a.lr[0], t = a.step[0];
    const float bc1 = 1.f - powf(a.beta1, t), bc2 = 1.f - powf(a.beta2, t);
    const float step_size = lr / bc1, bc2s = sqrtf(bc2);
    for (int i = threadIdx.x; i < len; i += 256) {
        const float g = gp[i] * cf;
        const float m = a.beta1 * a.m[fo + i] + (1.f - a.beta1) * g;
        const float v = a.beta2 * a.v[fo + i] + (1.f - a.beta2) * g * g;
        gp[i] = g;
        a.m[fo + i] = m;
        a.v[fo + i] = v;
        p[i] -= step_size * (m / (sqrtf(v) / bc2s + a.eps));
    }
}

}  // namespace

// flat_g / m / v: [numel] fp32 in the chunk table's flat layout; params: device array of the parameter tensors' addresses;
// chunks: device int32 [nchunks][4] = (tensor id, offset in the tensor, length, offset in the flat buffers);
// lr, step: device scalars (step counts completed updates: incremented here, before use); partial: >= nchunks doubles;
// counter: one zeroed 32-bit word (left zero); coef: 4 floats - [0] clip coefficient, [1] gradient norm before clipping,
// [2] 1 when this update was skipped because the norm is not finite (nothing written, step not counted), [3] += 1 per skipped
// update (zeroed by the caller).
extern "C" int pf_clip_adam(float* flat_g, float* m, float* v, float* const* params, const int* chunks, int nchunks,
                            const float* lr, float* step, float beta1, float beta2, float eps, float max_norm, double* partial,
                            unsigned* counter, float* coef, void* stream) {
    if (!flat_g || !m || !v || !params || !chunks || !lr || !step || !partial || !counter || !coef) return PF_ERR_NULL;
    if (nchunks <= 0) return PF_ERR_SHAPE;
    AdamArgs a{flat_g, m, v, params, chunks, nchunks, lr, step, beta1, beta2, eps, max_norm, partial, counter, coef, nullptr};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_norm_kernel, dim3(nchunks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(adam_update_kernel, dim3(nchunks), dim3(256), 0, s, a);
    return pf_last_launch_status();
}

// The same update with the gradients where autograd left them: grads = device array of the gradient tensors' addresses, one per
// parameter in the chunk table's order (each contiguous, as long as its parameter; a parameter without a gradient points at
// zeros).  No concatenation into a flat buffer first - inside a captured training step the addresses are the same at every replay,
// so the table is written once.  The clipped gradients are written back into those tensors.
extern "C" int pf_clip_adam_ptrs(float* const* grads, float* m, float* v, float* const* params, const int* chunks, int nchunks,
                                 const float* lr, float* step, float beta1, float beta2, float eps, float max_norm, double* partial,
                                 unsigned* counter, float* coef, void* stream) {
    if (!grads || !m || !v || !params || !chunks || !lr || !step || !partial || !counter || !coef) return PF_ERR_NULL;
    if (nchunks <= 0) return PF_ERR_SHAPE;
    AdamArgs a{nullptr, m, v, params, chunks, nchunks, lr, step, beta1, beta2, eps, max_norm, partial, counter, coef, grads};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_norm_kernel, dim3(nchunks), dim3(256), 0, s, a);
    hipLaunchKernelGGL(adam_update_kernel, dim3(nchunks), dim3(256), 0, s, a);
    return pf_last_launch_status();
}
