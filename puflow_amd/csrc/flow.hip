// Flow blocks: f (forward + log-det pieces) and g (exact inverse on the R-times replicated
// conditioning), all 6 blocks per launch.  Replaces FlowBlock.forward / .inverse
// (modules/discrete/interpflow.py:66-82) with ActNorm (modules/flows/normalize.py:30-43),
// InvertibleConv1x1_1D (modules/flows/permutate.py:117-126), the additive spatial coupling
// (modules/flows/coupling.py:55-58,82-85,114-118), channel reverse (permutate.py:75-80) and the
// conditional affine injector (coupling.py:127-151); drivers PointInterpFlow.f / .g
// (interpflow.py:302-321).
//
// Host-side folds (puflow_amd/packing.py): actnorm o inv1x1 = one 3x3 affine (A, a0) and its
// inverse (Ai, ai0); injector (s, t) and the c-part of coupling1's first layer (cp) come
// precomputed per ORIGINAL point from pf_post.  What is left per row is the coupling MLP
// 64 -> 64 -> (1|2), run on the f32 MFMA with 16 rows per column tile.
//
// Per-block weight record (floats, 16-byte aligned pieces), stride FLOW_REC:
//   [0,4096)  W2 fragments (4 ob x 4 cb)                                  (f32 image; unused by the split-bf16 path)
//   [4096,5120) W4 fragments (1 ob x 4 cb; rows replicated into every 4-row q group)
//   [5120,5184) b2      [5184,5200) b4 (replicated likewise)      [5200,5328) W0h [64][2]
//   [5328,5360) A(9) a0(3) Ai(9) ai0(3) pad
//   [5360,13040) split-bf16 image (csrc/pf_mfma.h) of W2 (4 ob x 2 pairs) then W4 (1 ob x 2 pairs): 10 fragments x 768 floats.
// The 64 -> 64 -> (1|2) coupling net runs on the bf16 pipe with fp32-class accuracy: 60 x 16 instead of 80 x 32 MFMA cycles
// per 16 rows and block.
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"
#include "pf_mfma.h"

// tuned on MI355X with tools/tune_variants.py (P = column tiles per wave, NW = waves per workgroup)
#ifndef PF_FLOW_P
#define PF_FLOW_P 1
#endif
#ifndef PF_FLOW_NW
#define PF_FLOW_NW 4
#endif

namespace {

constexpr int FLOW_REC = 13040;
constexpr float LOG2PI_F = 1.8378770664093453f;

struct FlowArgs {
    const float* in;     // fwd: xyz [T,3]      inv: u [T*R,3]
    const float* cp;     // [6][T][64]
    const float* st;     // [6][T][8]
    const float* w;      // 6 x FLOW_REC
    float* out;          // fwd: z [T,3]        inv: x [T*R,3]
    float* ld_pt;        // fwd only: [T]  -sum_blocks sum_ch s
    int T;               // original points
    int R;               // replicas (1 for fwd)
    int rows;            // T*R
    int ntiles;
};

// coupling MLP for one column tile: returns o[0..1] (bias_net output, 3-td values), identical in all q lanes
template <int TD, int P, class WS>
__device__ __forceinline__ void coupling_net(const WS& ws, const float* __restrict__ rec, const float* __restrict__ cpu,
                                             const int (&pt)[P], const float (&h1)[P][2], int q, float (&o)[P][2]) {
    f4 hid[P][4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const f4 wa = *reinterpret_cast<const f4*>(rec + 5200 + (cb * 16 + 4 * q) * 2);       // W0h rows ch, ch+1
        const f4 wb = *reinterpret_cast<const f4*>(rec + 5200 + (cb * 16 + 4 * q) * 2 + 4);   // rows ch+2, ch+3
#pragma unroll
        for (int p = 0; p < P; ++p) {
            f4 v = *reinterpret_cast<const f4*>(cpu + (size_t)pt[p] * 64 + cb * 16 + 4 * q);
            v.x = fmaf(wa.x, h1[p][0], v.x); v.y = fmaf(wa.z, h1[p][0], v.y);
            v.z = fmaf(wb.x, h1[p][0], v.z); v.w = fmaf(wb.z, h1[p][0], v.w);
            if (TD == 2) {
                v.x = fmaf(wa.y, h1[p][1], v.x); v.y = fmaf(wa.w, h1[p][1], v.y);
                v.z = fmaf(wb.y, h1[p][1], v.z); v.w = fmaf(wb.w, h1[p][1], v.w);
            }
            hid[p][cb] = pf_lrelu(v, 0.01f);
        }
    }
    f4 h2[P][4];
    PfPair hp[P][2];
#pragma unroll
    for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) h2[p][ob] = pf_bias(rec + 5120, ob, q);
        hp[p][0] = pf_pair(hid[p][0], hid[p][1]);
        hp[p][1] = pf_pair(hid[p][2], hid[p][3]);
    }
    pf_mm3<4, 2, 2>(ws, 0, hp, 0, h2, 0);
    f4 acc[P][1];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        acc[p][0] = *reinterpret_cast<const f4*>(rec + 5184 + 4 * q);
        hp[p][0] = pf_pair(pf_lrelu(h2[p][0], 0.01f), pf_lrelu(h2[p][1], 0.01f));
        hp[p][1] = pf_pair(pf_lrelu(h2[p][2], 0.01f), pf_lrelu(h2[p][3], 0.01f));
    }
    pf_mm3<1, 2, 2>(ws, 8, hp, 0, acc, 0);
#pragma unroll
    for (int p = 0; p < P; ++p) { o[p][0] = acc[p][0].x; o[p][1] = acc[p][0].y; }
}

template <bool INV, int P, int NW>
__global__ __launch_bounds__(NW * 64) void flow_kernel(FlowArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int r0 = (tile * NW + wave) * P * 16;
        int row[P], pt[P];
        bool ok[P];
        float v[P][3], ld[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int g = r0 + p * 16 + col;
            ok[p] = g < a.rows;
            row[p] = ok[p] ? g : a.rows - 1;
            pt[p] = row[p] / a.R;
            ld[p] = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[p][c] = a.in[(size_t)row[p] * 3 + c];
        }

        pf_static_for<0, 6>([&](auto uc) {
            constexpr int u = INV ? 5 - decltype(uc)::value : decltype(uc)::value;
            constexpr int TD = (u % 2 == 0) ? 1 : 2;
            const float* rec = a.w + u * FLOW_REC;
            const float* cpu = a.cp + (size_t)u * a.T * 64;
            const float* stu = a.st + (size_t)u * a.T * 8;
            const PfW3Buf ws(rec + 5360, lane);
            const float* fc = rec + 5328;
            float s[P][3], t[P][3];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const f4 s0 = *reinterpret_cast<const f4*>(stu + (size_t)pt[p] * 8);
                const f4 s1 = *reinterpret_cast<const f4*>(stu + (size_t)pt[p] * 8 + 4);
                s[p][0] = s0.x; s[p][1] = s0.y; s[p][2] = s0.z; t[p][0] = s0.w; t[p][1] = s1.x; t[p][2] = s1.y;
            }
            if constexpr (!INV) {
                float h1[P][2], o[P][2];
#pragma unroll
                for (int p = 0; p < P; ++p) {                 // actnorm o inv1x1:  v = A v + a0
                    const float x0 = v[p][0], x1 = v[p][1], x2 = v[p][2];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        v[p][i] = fmaf(fc[3 * i + 2], x2, fmaf(fc[3 * i + 1], x1, fmaf(fc[3 * i], x0, fc[9 + i])));
                    h1[p][0] = v[p][0]; h1[p][1] = v[p][1];
                }
                coupling_net<TD>(ws, rec, cpu, pt, h1, q, o);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    if (TD == 1) { v[p][1] -= o[p][0]; v[p][2] -= o[p][1]; } else { v[p][2] -= o[p][0]; }
                    const float y0 = v[p][2], y1 = v[p][1], y2 = v[p][0];      // reverse channels
                    v[p][0] = (y0 - t[p][0]) * expf(-s[p][0]);
                    v[p][1] = (y1 - t[p][1]) * expf(-s[p][1]);
                    v[p][2] = (y2 - t[p][2]) * expf(-s[p][2]);
                    ld[p] -= (s[p][0] + s[p][1]) + s[p][2];
                }
            } else {
                float h1[P][2], o[P][2];
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const float y0 = fmaf(v[p][0], expf(s[p][0]), t[p][0]);
                    const float y1 = fmaf(v[p][1], expf(s[p][1]), t[p][1]);
                    const float y2 = fmaf(v[p][2], expf(s[p][2]), t[p][2]);
                    v[p][0] = y2; v[p][1] = y1; v[p][2] = y0;                  // reverse^-1 (self-inverse)
                    h1[p][0] = v[p][0]; h1[p][1] = v[p][1];
                }
                coupling_net<TD>(ws, rec, cpu, pt, h1, q, o);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    if (TD == 1) { v[p][1] += o[p][0]; v[p][2] += o[p][1]; } else { v[p][2] += o[p][0]; }
                    const float x0 = v[p][0], x1 = v[p][1], x2 = v[p][2];
#pragma unroll
                    for (int i = 0; i < 3; ++i)                 // (inv1x1 o actnorm)^-1:  v = Ai v + ai0
                        v[p][i] = fmaf(fc[12 + 3 * i + 2], x2, fmaf(fc[12 + 3 * i + 1], x1, fmaf(fc[12 + 3 * i], x0, fc[21 + i])));
                }
            }
        });

#pragma unroll
        for (int p = 0; p < P; ++p) {
            if (ok[p] && q == 0) {
                a.out[(size_t)row[p] * 3 + 0] = v[p][0];
                a.out[(size_t)row[p] * 3 + 1] = v[p][1];
                a.out[(size_t)row[p] * 3 + 2] = v[p][2];
                if (!INV) a.ld_pt[row[p]] = ld[p];
            }
        }
    }
}

template <bool INV>
int launch(FlowArgs a, hipStream_t s) {
    constexpr int P = PF_FLOW_P, NW = PF_FLOW_NW;
    a.ntiles = (a.rows + NW * P * 16 - 1) / (NW * P * 16);
    const int grid = a.ntiles < 2048 ? a.ntiles : 2048;
    hipLaunchKernelGGL((flow_kernel<INV, P, NW>), dim3(grid), dim3(NW * 64), 0, s, a);
    return pf_last_launch_status();
}

// one workgroup per batch item: deterministic tree sums
__global__ __launch_bounds__(256) void logp_batch_kernel(const float* __restrict__ z, const float* __restrict__ ld_pt,
                                                        float ld_const, int N, float* __restrict__ ldj,
                                                        float* __restrict__ lpsum) {
    __shared__ float sa[256], sb[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    float accl = 0.f, accz = 0.f;
    for (int n = tid; n < N; n += 256) {
        accl += ld_pt[(size_t)b * N + n];
        const float* zz = z + ((size_t)b * N + n) * 3;
        accz += -0.5f * (zz[0] * zz[0] + LOG2PI_F) + -0.5f * (zz[1] * zz[1] + LOG2PI_F) + -0.5f * (zz[2] * zz[2] + LOG2PI_F);
    }
    sa[tid] = accl; sb[tid] = accz;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { sa[tid] += sa[tid + s]; sb[tid] += sb[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        const float l = sa[0] + ld_const * (float)N;
        ldj[b] = l;
        lpsum[b] = sb[0] + l;
    }
}

__global__ void logp_final_kernel(const float* __restrict__ lpsum, int B, float* __restrict__ logp) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += lpsum[b];
        *logp = -s / (float)B;
    }
}

}  // namespace

extern "C" int pf_flow_fwd(const float* xyz, const float* cp, const float* st, const float* w, float* z, float* ld_pt,
                           int T, void* stream) {
    if (!xyz || !cp || !st || !w || !z || !ld_pt) return PF_ERR_NULL;
    if (T <= 0) return PF_ERR_SHAPE;
    FlowArgs a{};
    a.in = xyz; a.cp = cp; a.st = st; a.w = w; a.out = z; a.ld_pt = ld_pt; a.T = T; a.R = 1; a.rows = T;
    return launch<false>(a, (hipStream_t)stream);
}

extern "C" int pf_flow_inv(const float* u, const float* cp, const float* st, const float* w, float* x, int T, int R,
                           void* stream) {
    if (!u || !cp || !st || !w || !x) return PF_ERR_NULL;
    if (T <= 0 || R <= 0 || (long long)T * R > (1ll << 30)) return PF_ERR_SHAPE;
    FlowArgs a{};
    a.in = u; a.cp = cp; a.st = st; a.w = w; a.out = x; a.ld_pt = nullptr; a.T = T; a.R = R; a.rows = T * R;
    return launch<true>(a, (hipStream_t)stream);
}

extern "C" int pf_logp(const float* z, const float* ld_pt, float ld_const, int B, int N, float* ldj, float* lpsum,
                       float* logp, void* stream) {
    if (!z || !ld_pt || !ldj || !lpsum || !logp) return PF_ERR_NULL;
    if (B <= 0 || N <= 0) return PF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(logp_batch_kernel, dim3(B), dim3(256), 0, s, z, ld_pt, ld_const, N, ldj, lpsum);
    hipLaunchKernelGGL(logp_final_kernel, dim3(1), dim3(64), 0, s, lpsum, B, logp);
    return pf_last_launch_status();
}
