// One FeatureExtractUnit (EdgeConv dense block) of the TRAINING step as a handful of launches.
//
// Reference: modules/discrete/interpflow.py:190-248 (FeatureExtractUnit.forward in train() mode: edge feature ->
// [Conv2d 1x1 + BatchNorm2d (batch statistics) + LeakyReLU(0.05), dense concatenation] x nconv -> conv_out -> max over the K
// neighbours) and the autograd backward PyTorch derives from it.  The un-fused path (train_ops.hip + train_ops.py) runs
// this as ~170 launches per unit and step (GEMM, two-pass statistics, apply, concatenations, gradient adds); at
// 32 x 256 points every one of those kernels is shorter than the gap between two launches, so the step was bound by the
// NUMBER of launches.  Here a unit is 7 launches forward and ~12 backward:
//
//   forward   fold        Wp = W1 - W3, Wq = W2 + W3 of all convs -> Wpq [2S, C]   (the edge feature [x_i; x_j; x_j - x_i]
//                         enters every conv only through P = Wp x_i (+ bias) and Q = Wq x_j: packing.fold_edgeconv)
//             gemm        PQ [T, 2S] = x Wpq^T + bias                              (train_ops.hip: pf_gemm)
//             layer t     Y[:, g t : g (t+1)] = P_t[i] + Q_t[j] + lrelu(bn(Y[:, :g t])) Wg_t^T   - the BatchNorm of the
//                         EARLIER layers is applied on load (scale / shift per channel), this layer's pre-activation
//                         output is stored and its column sums / sums of squares leave in the epilogue
//                         (the workgroup that finishes last turns the sums into scale / shift / running statistics)
//             out         conv_out on lrelu(bn(Y)) + P_out[i] + Q_out[j], max over the 16 edges of a point in the MFMA
//                         accumulator layout (the [E, odim] tensor is never written), argmax kept for the backward
//   backward  out         dA [E, GT] = dYout Wg_out, dYout generated from (dh, argmax) on load; epilogue: BatchNorm-backward
//                         sums of the last growth layer
//             layer t     (t = nconv-1 .. 1)  dy_t = BN-backward of dA[:, slice t] formed on load and stored in place;
//                         dA[:, :g t] += dy_t Wg_t; epilogue: sums for layer t-1
//             layer 0     dA[:, :g] -> dy_0 in place
//             pq          dP[i] = sum_k dy, dQ[j] += dy (atomics), conv_out part from (dh, argmax)
//             dw          all growth-weight gradients of the unit in ONE split-K launch: [S, GT] = dY^T lrelu(bn(Y))
//             gemm x2     dx = dPQ Wpq, dWpq = dPQ^T x
//             assemble    conv weight gradients [*, 3C + g t] from dWpq (un-folding) and the dw partial sums
//
// All matrix products are v_mfma_f32_16x16x4_f32 (exact fp32 fma chains).  Rows of every per-edge tensor are edges in
// point-major order (e = i K + k), so a 16-row MFMA tile is one point's 16 neighbours (K = 16) or two points (K = 8).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "pf_api_internal.h"
#include "pf_mfma.h"

extern "C" int pf_gemm(const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn, float* C,
                       long long ldc, const float* bias, int M, int N, int K, float* ws, long long ws_floats, void* stream);
extern "C" long long pf_gemm_ws_floats(int M, int N, int K);

namespace {

#ifndef PF_EC_GRID
#define PF_EC_GRID 512
#endif
constexpr int EC_GRID = PF_EC_GRID;   // persistent workgroups of the per-edge kernels (2 per CU)

__device__ __forceinline__ float lrelu1(float v, float s) { return fmaxf(v, v * s); }
__device__ __forceinline__ f4 lrelu4(f4 z, float s) {
    f4 r;
    r.x = fmaxf(z.x, z.x * s); r.y = fmaxf(z.y, z.y * s); r.z = fmaxf(z.z, z.z * s); r.w = fmaxf(z.w, z.w * s);
    return r;
}
__device__ __forceinline__ f4 mfma4(f4 a, f4 b, f4 c) {
    c = pf_mfma(a.x, b.x, c); c = pf_mfma(a.y, b.y, c); c = pf_mfma(a.z, b.z, c); c = pf_mfma(a.w, b.w, c);
    return c;
}
// B operand that makes mfma4(a, ident, c) add the A-layout tile `a` (lane (row, q) holds channels 4q..4q+3 of its row) to the
// accumulator-layout tile c: the matrix pipe as a transposer for per-row gathered addends
__device__ __forceinline__ f4 ident_b(int row, int q) {
    f4 r;
    r.x = 4 * q + 0 == row ? 1.f : 0.f; r.y = 4 * q + 1 == row ? 1.f : 0.f;
    r.z = 4 * q + 2 == row ? 1.f : 0.f; r.w = 4 * q + 3 == row ? 1.f : 0.f;
    return r;
}

// ---- column statistics without a second launch: every workgroup adds its column sums to 64 double accumulators, the
// workgroup that arrives last turns them into the layer's constants and clears them for the next user.
//   mode 1 (BatchNorm forward): sums of y, y^2 -> scale, shift, mean, 1/std (aff rows 0..3), running statistics
//   mode 2 (BatchNorm backward): sums of dz, dz xhat -> their means (coef rows 0, 1), dbeta, dgamma
struct StatFin {
    double* acc;                      // [STAT_COPIES][2][STAT_W] + a counter word behind them; all zero between uses
    int mode, g, col0, ld;
    float* aff; const float* gamma; const float* beta; float* run_mean; float* run_var; float eps, momentum;
    float* coef; float* dgamma; float* dbeta;
    double R;
    double* defer;                    // SyncBN: non-null = the last workgroup does NOT finish the layer; it leaves the LOCAL sums in
                                      // defer[0 .. ncol) / defer[STAT_W ..] and the local row count in defer[2 STAT_W] (mode 2: dbeta /
                                      // dgamma are written from the local sums, as torch.nn.SyncBatchNorm does); the host all-reduces
                                      // the 2 STAT_W + 1 doubles over the ranks and stat_finalize_kernel finishes with the global sums
    int det;                          // PF_TRAIN_DETERMINISTIC: the accumulators hold 64-bit FIXED-POINT sums (quantum 2^-28) added with
                                      // integer atomics - exact, so independent of the order in which the workgroups arrive; the
                                      // default (double atomics) rounds in arrival order once a sum needs more than 53 bits
};

constexpr int STAT_COPIES = 16;       // workgroups spread their atomics over this many accumulator sets (same-address atomics serialise)
constexpr int STAT_W = 128;           // statistics columns per launch (EdgeConv layers use <= 32, the BatchNorm MLPs up to 128)
constexpr int STAT_DOUBLES = STAT_COPIES * 2 * STAT_W + 1;
// deterministic accumulation (StatFin::det): a workgroup's float partial as a multiple of 2^-28 in a 64-bit integer (|sum| < 3.4e10);
// the same 8-byte accumulator words, zero in either reading
#define PF_DET(p) (((p)->flags & PF_TRAIN_DETERMINISTIC) ? 1 : 0)
constexpr double STAT_FIX = 268435456.0, STAT_FIX_INV = 1.0 / 268435456.0;
__device__ __forceinline__ void stat_add(double* acc, float v, int det) {
    if (det) atomicAdd(reinterpret_cast<unsigned long long*>(acc), (unsigned long long)(long long)__double2ll_rn((double)v * STAT_FIX));
    else unsafeAtomicAdd(acc, (double)v);
}
__device__ __forceinline__ double stat_load(const double* acc, int det) {
    if (det)
        return (double)(long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) *
               STAT_FIX_INV;
    return __hip_atomic_load(acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sums of one column -> the layer's constants.  R: rows the sums run over (the GLOBAL count under SyncBN); param_grads: mode 2
// also writes dbeta / dgamma from these sums (not under SyncBN: there they are the LOCAL sums, written by stat_flush)
__device__ __forceinline__ void stat_finish_col(const StatFin& f, int c, double a0, double a1, double R, bool param_grads) {
    if (f.mode == 1) {
        // a0, a1 are sums of (y - pivot), (y - pivot)^2 with pivot = the running mean the kernels started from (read here
        // before it is updated below; 0 without running statistics)
        const double pv = f.run_mean ? (double)f.run_mean[c] : 0.0;
        const double dm = a0 / R;
        const double mean = pv + dm;
        double var = a1 / R - dm * dm;
        if (var < 0.0) var = 0.0;
        const float rstd = 1.0f / sqrtf((float)var + f.eps);
        const float sc = f.gamma[c] * rstd;
        f.aff[f.col0 + c] = sc;
        f.aff[f.ld + f.col0 + c] = f.beta[c] - (float)mean * sc;
        f.aff[2 * f.ld + f.col0 + c] = (float)mean;
        f.aff[3 * f.ld + f.col0 + c] = rstd;
        if (f.run_mean) {
            f.run_mean[c] = (1.f - f.momentum) * f.run_mean[c] + f.momentum * (float)mean;
            f.run_var[c] = (1.f - f.momentum) * f.run_var[c] + f.momentum * (float)(var * (R / (R - 1.0)));
        }
    } else {
        f.coef[f.col0 + c] = (float)(a0 / R);
        f.coef[f.ld + f.col0 + c] = (float)(a1 / R);
        if (param_grads) { f.dbeta[c] = (float)a0; f.dgamma[c] = (float)a1; }
    }
}

// s0 / s1: this lane's sums for column (lane & 15) of each 16-column tile; `first`: the column that maps to statistics
// column 0; ncol <= STAT_W.  red: 4 * 2 * STAT_W floats of LDS.
template <int NT>
__device__ __forceinline__ void stat_flush(float (&s0)[NT], float (&s1)[NT], int first, int ncol, const StatFin& f, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s0[nt] += __shfl_xor(s0[nt], 16); s0[nt] += __shfl_xor(s0[nt], 32);
        s1[nt] += __shfl_xor(s1[nt], 16); s1[nt] += __shfl_xor(s1[nt], 32);
    }
    if (lane < 16) {                                                  // red[wave][2][STAT_W]
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c = nt * 16 + lane - first;
            if (c >= 0 && c < ncol) { red[wave * 2 * STAT_W + c] = s0[nt]; red[wave * 2 * STAT_W + STAT_W + c] = s1[nt]; }
        }
    }
    __syncthreads();
    if ((threadIdx.x & (STAT_W - 1)) < ncol) {                        // 256 threads = 2 x STAT_W sums
        const int t = threadIdx.x;
        const float v = (red[t] + red[2 * STAT_W + t]) + (red[4 * STAT_W + t] + red[6 * STAT_W + t]);
        stat_add(f.acc + (blockIdx.x % STAT_COPIES) * 2 * STAT_W + t, v, f.det);
    }
    // order the accumulator atomics before the arrival count WITHOUT a release fence: a device-scope fence writes the whole
    // L2 back on this multi-die part (tens of microseconds per launch); the atomics themselves are performed at the coherent
    // level, so waiting for their acknowledgement is enough
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned* counter = reinterpret_cast<unsigned*>(f.acc + STAT_COPIES * 2 * STAT_W);
    if (threadIdx.x == 0) red[0] = atomicAdd(counter, 1u) == gridDim.x - 1 ? 1.f : 0.f;
    __syncthreads();
    if (red[0] == 0.f) return;
    const int c = threadIdx.x;
    if (c < ncol) {
        double a0 = 0.0, a1 = 0.0;
        for (int k = 0; k < STAT_COPIES; ++k) {
            a0 += stat_load(f.acc + k * 2 * STAT_W + c, f.det);          // (16 multiples of 2^-28: exact in double, any order)
            a1 += stat_load(f.acc + k * 2 * STAT_W + STAT_W + c, f.det);
            f.acc[k * 2 * STAT_W + c] = 0.0; f.acc[k * 2 * STAT_W + STAT_W + c] = 0.0;
        }
        if (f.defer) {                                                // SyncBN: local sums out, the layer is finished after the all-reduce
            f.defer[c] = a0;
            f.defer[STAT_W + c] = a1;
            if (f.mode == 2) { f.dbeta[c] = (float)a0; f.dgamma[c] = (float)a1; }
        } else
            stat_finish_col(f, c, a0, a1, f.R, true);
    }
    if (threadIdx.x == 0) {
        if (f.defer) f.defer[2 * STAT_W] = f.R;
        *counter = 0u;
    }
}

// SyncBN: the layer's constants from the all-reduced sums (defer[] as stat_flush left it, summed over the ranks by the host)
__global__ __launch_bounds__(STAT_W) void stat_finalize_kernel(StatFin f, int ncol) {
    const int c = threadIdx.x;
    if (c < ncol) stat_finish_col(f, c, f.defer[c], f.defer[STAT_W + c], f.defer[2 * STAT_W], false);
}

// ------------------------------------------------------------------------------------------------ forward, one conv
// growth layer t (OUT = false): Y[:, col0 : col0 + g] = P_t[i] + Q_t[j] + lrelu(bn(Y[:, :kin])) W^T, statistics of the result
// conv_out (OUT = true): the same product on all GT growth channels, then max over the 16 edges of a point (POOL) or the
// per-edge rows
struct EcFwdArgs {
    float* Y; int ldy;               // [E, ldy] pre-BN outputs of the growth layers
    const float* aff;                // [4][ldy]: scale, shift, mean, rstd of the finished layers
    const float* W; int ldw;         // growth columns of this conv: W[c * ldw + u], c < nout, u < kin
    const float* pq; int ldpq;       // [T, ldpq] = P (+ bias) | Q
    int poff, qoff;                  // columns of this conv's P and Q
    const int* idx;                  // [E] batch-local neighbour index
    int N, K;
    int kin, col0, nout;
    int ntiles;                      // E / 16
    float slope;
    float* out; unsigned char* arg;  // conv_out only
    StatFin fin;                     // growth layers only
};

template <int NT, bool OUT, bool POOL>
__global__ __launch_bounds__(256) void ec_fwd_kernel(EcFwdArgs a) {
    extern __shared__ float lds[];
    __shared__ float red[8 * STAT_W];
    const int kin16 = (a.kin + 15) & ~15, kp = kin16 + 4, KS = kin16 / 16;
    float* Wl = lds;
    float* al = lds + NT * 16 * kp;
    float* bl = al + kin16;
    // 16 lanes along a weight row (coalesced, no division); a thread's <= 8 elements of a row are loaded together, then stored:
    // a load -> store loop pays a full memory latency per element
    for (int c = threadIdx.x >> 4; c < NT * 16; c += 16) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int u = (threadIdx.x & 15) + 16 * k;
            v[k] = (u < kin16 && c < a.nout && u < a.kin) ? a.W[(size_t)c * a.ldw + u] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int u = (threadIdx.x & 15) + 16 * k;
            if (u < kin16) Wl[c * kp + u] = v[k];
        }
    }
    for (int i = threadIdx.x; i < kin16; i += 256) {
        al[i] = i < a.kin ? a.aff[i] : 0.f;
        bl[i] = i < a.kin ? a.aff[a.ldy + i] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, q = lane >> 4;
    const f4 ident = ident_b(row, q);
    // column statistics are accumulated CENTRED on a pivot (the layer's running mean as every workgroup reads it at its
    // start - the last workgroup updates it only after all have arrived): sum (y - p), sum (y - p)^2.  E[y^2] - E[y]^2 on raw
    // fp32 partial sums loses |mean|^2 / var digits; the running mean tracks the batch mean, so the centred form does not
    float s0[NT], s1[NT], piv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s0[nt] = s1[nt] = 0.f;
        const int col = nt * 16 + row;
        piv[nt] = (!OUT && a.fin.run_mean && col < a.nout) ? a.fin.run_mean[col] : 0.f;
    }
    const int tile0 = blockIdx.x * 4 + wave;
    int jnext = tile0 < a.ntiles ? a.idx[(long long)tile0 * 16 + row] : 0;
    for (int tile = tile0; tile < a.ntiles; tile += gridDim.x * 4) {
        const long long e0 = (long long)tile * 16;
        // all loads of the tile first: the growth-feature row of this lane's edge and its P[i] + Q[j] addend (the neighbour
        // index was fetched during the previous tile: one dependent memory latency less per tile)
        const int er = (int)e0 + row;
        const int ir = er / a.K;
        const long long jr = (long long)(ir / a.N) * a.N + jnext;
        {
            const int tn = tile + gridDim.x * 4;
            if (tn < a.ntiles) jnext = a.idx[(long long)tn * 16 + row];
        }
        const float* yrow = a.Y + (size_t)er * a.ldy;
        f4 yv[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            yv[ks] = pf_splat(0.f);
            if (ks < KS && ks * 16 + 4 * q < a.kin) yv[ks] = *reinterpret_cast<const f4*>(yrow + ks * 16 + 4 * q);
        }
        f4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c4 = nt * 16 + 4 * q;
            f4 ex = pf_splat(0.f);
            if (c4 < a.nout)
                ex = *reinterpret_cast<const f4*>(a.pq + (size_t)ir * a.ldpq + a.poff + c4) +
                     *reinterpret_cast<const f4*>(a.pq + (size_t)jr * a.ldpq + a.qoff + c4);
            acc[nt] = mfma4(ex, ident, pf_splat(0.f));
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks < KS) {
                const int u = ks * 16 + 4 * q;
                f4 av = pf_splat(0.f);
                if (u < a.kin)
                    av = lrelu4(yv[ks] * *reinterpret_cast<const f4*>(al + u) + *reinterpret_cast<const f4*>(bl + u), a.slope);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = mfma4(av, *reinterpret_cast<const f4*>(Wl + (nt * 16 + row) * kp + u), acc[nt]);
            }
        }
        if (!OUT) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 16 + row;
                if (col < a.nout) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[nt][r];
                        a.Y[(e0 + 4 * q + r) * a.ldy + a.col0 + col] = v;
                        const float vc = v - piv[nt];
                        s0[nt] += vc; s1[nt] = fmaf(vc, vc, s1[nt]);
                    }
                }
            }
        } else if (!POOL) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 16 + row;
                if (col < a.nout)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a.out[(e0 + 4 * q + r) * a.nout + col] = acc[nt][r];
            }
        } else {                                                       // K = 16: the tile is point `tile`
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float best = acc[nt][0];
                int bk = 4 * q;
#pragma unroll
                for (int r = 1; r < 4; ++r)
                    if (acc[nt][r] > best) { best = acc[nt][r]; bk = 4 * q + r; }
#pragma unroll
                for (int m = 16; m < 64; m <<= 1) {
                    const float ov = __shfl_xor(best, m);
                    const int ok = __shfl_xor(bk, m);
                    if (ov > best || (ov == best && ok < bk)) { best = ov; bk = ok; }
                }
                const int col = nt * 16 + row;
                if (q == 0 && col < a.nout) {
                    a.out[(long long)tile * a.nout + col] = best;
                    a.arg[(long long)tile * a.nout + col] = (unsigned char)bk;
                }
            }
        }
    }
    if (!OUT) stat_flush<NT>(s0, s1, 0, a.nout, a.fin, red);
}

// ------------------------------------------------------------------------------------------------ forward, conv_out on the fp16 pipe
// conv_out is the one forward kernel whose matrix work is not negligible (E x GT x odim products: 38 % pipe-busy on f32 MFMAs).
// Here its products run as split-fp16 (csrc/pf_mfma.h "f16x2": x = hi + lo' 2^-11, hi.hi in the main accumulator, hi.lo' + lo'.hi
// in a second one folded in as acc + accx 2^-11 - 22+ significant bits per operand; activations after BatchNorm + LeakyReLU and
// weights are far inside the fp16 range): three v_mfma_f32_16x16x32_f16 per 32 channels instead of eight f32 MFMAs.  Lane
// (row = edge, kg = l >> 4) holds the 8 channels 8 kg .. 8 kg + 7 of a 32-channel chunk of its edge's feature row; the weights
// are converted once per workgroup into ready fragments [tile][chunk][hi | lo'][lane][8 x f16]; the per-edge addend P[i] + Q[j]
// enters through an identity B operand, split the same way.  Accumulator layout, pooling and outputs as in ec_fwd_kernel.
template <int NT, bool POOL>
__global__ __launch_bounds__(256) void ec_fwd16_kernel(EcFwdArgs a) {
    extern __shared__ float lds[];
    const int nch = (a.kin + 31) / 32;
    uint4* Wf = reinterpret_cast<uint4*>(lds);                    // ((nt * nch + chunk) * 2 + hi|lo) * 64 + lane
    float* al = lds + (size_t)NT * nch * 2 * 64 * 4;
    float* bl = al + nch * 32;
    for (int unit = threadIdx.x; unit < NT * nch * 64; unit += 256) {
        const int frag = unit >> 6, ln = unit & 63, u = ln & 15, kg = ln >> 4, nt = frag / nch, ch = frag % nch;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = ch * 32 + 8 * kg + j, o = nt * 16 + u;
            v[j] = (c < a.kin && o < a.nout) ? a.W[(size_t)o * a.ldw + c] : 0.f;
        }
        const f4 v0 = {v[0], v[1], v[2], v[3]}, v1 = {v[4], v[5], v[6], v[7]};
        const PfPair2 f = pf_pair2(v0, v1);
        Wf[(frag * 2 + 0) * 64 + ln] = __builtin_bit_cast(uint4, f.h);
        Wf[(frag * 2 + 1) * 64 + ln] = __builtin_bit_cast(uint4, f.l);
    }
    for (int i = threadIdx.x; i < nch * 32; i += 256) {
        al[i] = i < a.kin ? a.aff[i] : 0.f;
        bl[i] = i < a.kin ? a.aff[a.ldy + i] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, q = lane >> 4;
    h8 identh;                                            // B[k][j] = (k == j) for k < 16, this lane: j = row, k = 8 q + jj
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) identh[jj] = (8 * q + jj == row) ? (_Float16)1.f : (_Float16)0.f;
    const int tile0 = blockIdx.x * 4 + wave;
    int jnext = tile0 < a.ntiles ? a.idx[(long long)tile0 * 16 + row] : 0;
    for (int tile = tile0; tile < a.ntiles; tile += gridDim.x * 4) {
        const long long e0 = (long long)tile * 16;
        const int er = (int)e0 + row;
        const int ir = er / a.K;
        const long long jr = (long long)(ir / a.N) * a.N + jnext;
        {
            const int tn = tile + gridDim.x * 4;
            if (tn < a.ntiles) jnext = a.idx[(long long)tn * 16 + row];
        }
        const float* yrow = a.Y + (size_t)er * a.ldy;
        f4 y0[4], y1[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            y0[ch] = y1[ch] = pf_splat(0.f);
            const int c = ch * 32 + 8 * q;
            if (ch < nch && c < a.kin) {                             // kin is a multiple of 8
                y0[ch] = *reinterpret_cast<const f4*>(yrow + c);
                y1[ch] = *reinterpret_cast<const f4*>(yrow + c + 4);
            }
        }
        f4 acc[NT], accx[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int c8 = nt * 16 + 8 * q;                          // the addend's channels: k = 8 q + jj < 16 of this tile
            f4 e0v = pf_splat(0.f), e1v = pf_splat(0.f);
            if (q < 2 && c8 < a.nout) {
                const float* pp = a.pq + (size_t)ir * a.ldpq + a.poff + c8;
                const float* qp = a.pq + (size_t)jr * a.ldpq + a.qoff + c8;
                e0v = *reinterpret_cast<const f4*>(pp) + *reinterpret_cast<const f4*>(qp);
                e1v = *reinterpret_cast<const f4*>(pp + 4) + *reinterpret_cast<const f4*>(qp + 4);
            }
            const PfPair2 E = pf_pair2(e0v, e1v);
            acc[nt] = pf_mfma_f16(E.h, identh, pf_splat(0.f));
            accx[nt] = pf_mfma_f16(E.l, identh, pf_splat(0.f));
        }
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            if (ch < nch) {
                const int c = ch * 32 + 8 * q;
                const f4 a0 = lrelu4(y0[ch] * *reinterpret_cast<const f4*>(al + c) + *reinterpret_cast<const f4*>(bl + c), a.slope);
                const f4 a1 = lrelu4(y1[ch] * *reinterpret_cast<const f4*>(al + c + 4) + *reinterpret_cast<const f4*>(bl + c + 4), a.slope);
                const PfPair2 A = pf_pair2(c < a.kin ? a0 : pf_splat(0.f), c < a.kin ? a1 : pf_splat(0.f));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const h8 bh = __builtin_bit_cast(h8, Wf[((nt * nch + ch) * 2 + 0) * 64 + lane]);
                    const h8 blo = __builtin_bit_cast(h8, Wf[((nt * nch + ch) * 2 + 1) * 64 + lane]);
                    acc[nt] = pf_mfma_f16(A.h, bh, acc[nt]);
                    accx[nt] = pf_mfma_f16(A.h, blo, accx[nt]);
                    accx[nt] = pf_mfma_f16(A.l, bh, accx[nt]);
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = acc[nt] + accx[nt] * PF_LO_INV;
        if (!POOL) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 16 + row;
                if (col < a.nout)
#pragma unroll
                    for (int r = 0; r < 4; ++r) a.out[(e0 + 4 * q + r) * a.nout + col] = acc[nt][r];
            }
        } else {                                                       // K = 16: the tile is point `tile`
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float best = acc[nt][0];
                int bk = 4 * q;
#pragma unroll
                for (int r = 1; r < 4; ++r)
                    if (acc[nt][r] > best) { best = acc[nt][r]; bk = 4 * q + r; }
#pragma unroll
                for (int m = 16; m < 64; m <<= 1) {
                    const float ov = __shfl_xor(best, m);
                    const int ok = __shfl_xor(bk, m);
                    if (ov > best || (ov == best && ok < bk)) { best = ov; bk = ok; }
                }
                const int col = nt * 16 + row;
                if (q == 0 && col < a.nout) {
                    a.out[(long long)tile * a.nout + col] = best;
                    a.arg[(long long)tile * a.nout + col] = (unsigned char)bk;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward, the whole unit in ONE launch
// The per-layer kernels above pay, per unit, five launch ramps / drains and re-read every earlier layer's pre-BatchNorm output
// from memory (0 + 1 + 2 + 3 growth blocks for the growth layers, all four again for conv_out: ~170 MB per 128-channel unit next
// to the 67 MB it has to write for the backward).  BatchNorm's batch statistics are the only thing that couples two edges, so a
// PERSISTENT grid can keep an edge tile's features in registers through the whole dense block and meet at a grid barrier once
// per BatchNorm layer:
//   * one workgroup per CU (512 threads = 8 waves, 2 per SIMD, <= 256 VGPRs), a wave owns up to ECP_TPW tiles of 16 edges (= one
//     point and its K = 16 neighbours) for the whole launch;
//   * channel-major chain (pf_mfma.h): output channels on MFMA rows, the 16 edges on the columns - a layer's accumulator tile
//     IS the next layer's B operand, no transposition, no LDS round trip;
//   * layer t: y_t = P_t[i] + Q_t[j] + W_t f_{<t} (v_mfma_f32_16x16x4_f32, the k order of ec_fwd_kernel: the stored Y is bit
//     for bit the per-layer kernels'), Y stored once for the backward, column sums (centred on the running mean) -> 16 spread
//     double accumulators -> arrival counter; the workgroup that arrives last turns the sums into scale / shift / running
//     statistics (the StatFin arithmetic) and publishes the barrier's generation word; everyone applies BatchNorm + LeakyReLU to
//     the tile it still holds;
//   * conv_out on split-fp16 products (the f16x2 arithmetic of ec_fwd16_kernel, weights converted once per workgroup into A
//     fragments in LDS), max over the 16 edges = the 16 lanes of a DPP row, argmax = smallest k among the maxima.
// Barrier: agent-scope relaxed atomics only (arrive: s_waitcnt vmcnt(0) + atomic add; release: the last arriver's atomic stores
// of aff, s_waitcnt, then the generation word) - no release fence (a device-scope fence writes the L2 back: tens of us).
// Co-residency is the HOST's job (pf_ec_train_fwd: occupancy x CU count >= grid, and the caller's PF_EC_PERSISTENT flag says no
// other barrier kernel of this process can be in flight); the spin is bounded anyway: on timeout the status word sync[3] is
// set, the unit's output becomes NaN and the grid drains.
constexpr int ECP_WAVES = 8, ECP_T = 64 * ECP_WAVES, ECP_TPW = 4;
// Tiles per wave of the narrow units (growth 8 / 16) as a build parameter.  They wait 82 - 84 % of their cycles (PMC) with two
// waves per SIMD; at 2 tiles per wave they need half the registers (86 - 120 VGPRs) and run as 512 workgroups, two per CU.  Measured
// (round 5, -DPF_ECP_TPW_SMALL=2): the unit's forward alone 78 -> 102 us (twice the arrivals per barrier, twice the weight staging),
// and inside the training step - where the side stream's kernels hold wave slots and a grid of exactly 2 x 256 workgroups has no
// slack - barrier time-outs.  Not used: 4, like the 128-channel units (which at 2 tiles per wave would need 143 - 176 registers: two
// workgroups do not fit a CU).
#ifndef PF_ECP_TPW_SMALL
#define PF_ECP_TPW_SMALL 4
#endif
__host__ __device__ constexpr int ecp_tpw(int G) { return G <= 16 ? PF_ECP_TPW_SMALL : ECP_TPW; }
constexpr int ECP_SPIN = 1 << 22;
// timing-only ablations of ec_fwdp_kernel (tools/time_ecunit.py with -DPF_ECP_DBG=mask builds; results are WRONG with any bit set):
// 1 no conv_out, 2 barriers pass at once, 4 no weight staging, 8 no Y stores, 16 no statistics atomics
#ifndef PF_ECP_DBG
#define PF_ECP_DBG 0
#endif

struct EcFwdPArgs {
    float* Y; int ldy;               // [E, GT]
    float* aff;                      // [4][GT]
    const float* Wg[8]; int ldwg[8]; // growth columns of conv t (pointer past the 3C edge-feature columns)
    const float* Wout; int ldwout;
    const float* pq; int ldpq; int S;
    const int* idx; int N;
    int ntiles;                      // = points (K = 16)
    float slope;
    float* out; unsigned char* arg;
    const float* gamma[8]; const float* beta[8]; float* run_mean[8]; float* run_var[8];
    float eps, momentum; double R;
    double* acc;                     // [STAT_COPIES][2][STAT_W] accumulators (zero between uses)
    unsigned* sync;                  // [0] arrivals [1] generation [2] exits [3] status (sticky: 1 = a barrier timed out)
};

__device__ __forceinline__ float ecp_ald(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ecp_ast(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int CTRL>
__device__ __forceinline__ float ecp_dppf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float ecp_rowsum16(float v) {           // sum over the 16 lanes of a DPP row, in every lane
    v += ecp_dppf<0x128>(v); v += ecp_dppf<0x124>(v); v += ecp_dppf<0x122>(v); v += ecp_dppf<0x121>(v);
    return v;
}
__device__ __forceinline__ float ecp_rowmax16(float v) {
    v = fmaxf(v, ecp_dppf<0x128>(v)); v = fmaxf(v, ecp_dppf<0x124>(v)); v = fmaxf(v, ecp_dppf<0x122>(v)); v = fmaxf(v, ecp_dppf<0x121>(v));
    return v;
}
__device__ __forceinline__ int ecp_rowmin16(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false)); v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false)); v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false));
    return v;
}

// grid barrier number `gen` (1, 2, ...) of this launch: pure arrival counting - the workgroup that arrives last publishes
// the generation word at once (what follows a barrier is done by every workgroup for itself).  Returns false when the spin gave
// up (uniform over the workgroup).
__device__ __forceinline__ bool ecp_barrier(unsigned* sync, unsigned gen, int* flag) {
    if (PF_ECP_DBG & 2) { __syncthreads(); return true; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        int fl = 0;
        if (atomicAdd(sync, 1u) == gen * gridDim.x - 1)
            __hip_atomic_store(sync + 1, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else {
            int budget = ECP_SPIN;
            while (__hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen && --budget > 0) __builtin_amdgcn_s_sleep(1);
            if (budget <= 0) { fl = 2; __hip_atomic_store(sync + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        *flag = fl;
    }
    __syncthreads();
    return *flag != 2;
}

template <int G, int NC, int ODIM>
__global__ __launch_bounds__(ECP_T) void ec_fwdp_kernel(EcFwdPArgs a) {
    constexpr int TPW = ecp_tpw(G);                   // tiles of 16 edges a wave owns for the whole launch
    constexpr int GT = G * NC, NB = GT / 16, NTG = (G + 15) / 16, NTO = ODIM / 16, NCP = NB / 2;
#ifndef PF_ECP_OCH
#define PF_ECP_OCH 1
#endif
    constexpr int OCH = NB >= 8 ? PF_ECP_OCH : 2;                           // conv_out blocks per accumulator chunk: 2 x 4 x ECP_TPW x OCH accumulator
                                                               // registers beside the wave's ECP_TPW x NB x 4 feature registers
    constexpr bool OWN = G % 16 == 0;                          // a layer's 16-channel blocks are its own (G = 8: two layers share one)
    static_assert(GT % 32 == 0 && ODIM % 16 == 0 && NTO % OCH == 0 && 32 * NC <= STAT_W && G <= 32, "shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[ECP_WAVES * 2 * 32];
    __shared__ float scsh[2 * 32];
    __shared__ int flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;

    // ---- this wave's tiles
    int tl[TPW], jr[TPW];
    bool ok[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
        tl[s] = blockIdx.x * ECP_WAVES + wave + s * gridDim.x * ECP_WAVES;
        ok[s] = tl[s] < a.ntiles;
        const int tt = ok[s] ? tl[s] : 0;
        jr[s] = (tt / a.N) * a.N + a.idx[(size_t)tt * 16 + col];
        tl[s] = tt;
    }
    f4 f[TPW][NB];
#pragma unroll
    for (int s = 0; s < TPW; ++s)
#pragma unroll
        for (int b = 0; b < NB; ++b) f[s][b] = pf_splat(0.f);
    // the addends P_t[i] + Q_t[j] of layer t for all the wave's tiles (independent gathers, all in flight together).  OWN layers
    // accumulate in the feature slots they are about to fill (free until then), so layer t + 1's addends can be fetched BEFORE
    // the barrier of layer t and their latency disappears behind it
    f4 accs[OWN ? 1 : TPW][NTG];
    auto addends = [&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int col0 = G * t, b0 = col0 / 16;
#pragma unroll
        for (int s = 0; s < TPW; ++s)
#pragma unroll
            for (int nt = 0; nt < NTG; ++nt) {
                const int c4 = 16 * (b0 + nt) + 4 * q;
                f4 v = pf_splat(0.f);
                if (c4 >= col0 && c4 < col0 + G)
                    v = *reinterpret_cast<const f4*>(a.pq + (size_t)tl[s] * a.ldpq + c4) +
                        *reinterpret_cast<const f4*>(a.pq + (size_t)jr[s] * a.ldpq + a.S + c4);
                if constexpr (OWN) f[s][b0 + nt] = v;
                else accs[s][nt] = v;
            }
    };
    addends(std::integral_constant<int, 0>{});                // layer 0 needs no weights: its gathers fly while the weights are staged

    // ---- LDS images: growth weights of layers 1 .. NC-1 (fp32, row = channel inside the layer's first block, padded rows /
    // columns zero), then conv_out as split-fp16 A fragments [ob][cp][hi | lo'][lane]; first read after barrier 1
    int woff[NC];
    {
        int o = 0;
#pragma unroll
        for (int t = 1; t < NC; ++t) { woff[t] = o; o += NTG * 16 * (((G * t + 15) & ~15) + 4); }
        woff[0] = o;                                           // [0]: start of the conv_out fragments (a multiple of 64 floats)
    }
    // (all of a thread's loads of a matrix are in flight before its first LDS store: a load -> store loop pays one memory
    // latency per element)
    pf_static_for<1, NC>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if (PF_ECP_DBG & 4) return;
        constexpr int kin = G * t, kin16 = (kin + 15) & ~15, kp = kin16 + 4, ro = (G * t) % 16, NE = NTG * 16 * kin16;
        constexpr int IT = (NE + ECP_T - 1) / ECP_T;
        float* Wl = lds + woff[t];
        float v[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = threadIdx.x + k * ECP_T, rw = i / kin16, u = i % kin16, c = rw - ro;      // tile row -> row of the conv
            v[k] = (i < NE && c >= 0 && c < G && u < kin) ? a.Wg[t][(size_t)c * a.ldwg[t] + u] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int i = threadIdx.x + k * ECP_T;
            if (i < NE) Wl[(i / kin16) * kp + i % kin16] = v[k];
        }
    });
    uint4* Wf = reinterpret_cast<uint4*>(lds + woff[0]);
    if (!(PF_ECP_DBG & 4)) {
        constexpr int NU = NTO * NCP * 64, IT = (NU + ECP_T - 1) / ECP_T;
        f4 w0[IT], w1[IT];
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int unit = threadIdx.x + k * ECP_T, uu = unit < NU ? unit : 0;
            const int frag = uu >> 6, ln = uu & 63, o = (frag / NCP) * 16 + (ln & 15), cp = frag % NCP, kq = ln >> 4;
            const float* wr = a.Wout + (size_t)o * a.ldwout + 32 * cp + 4 * kq;
            w0[k] = (f4){wr[0], wr[1], wr[2], wr[3]};
            w1[k] = (f4){wr[16], wr[17], wr[18], wr[19]};
        }
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int unit = threadIdx.x + k * ECP_T;
            if (unit < NU) {
                const PfPair2 fr = pf_pair2(w0[k], w1[k]);
                Wf[((unit >> 6) * 2 + 0) * 64 + (unit & 63)] = __builtin_bit_cast(uint4, fr.h);
                Wf[((unit >> 6) * 2 + 1) * 64 + (unit & 63)] = __builtin_bit_cast(uint4, fr.l);
            }
        }
    }

    bool alive = true;
    pf_static_for<0, NC>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int col0 = G * t, b0 = col0 / 16, KS = (G * t + 15) / 16, kp = ((G * t + 15) & ~15) + 4;
        if (!alive) return;
        const float* Wl = lds + woff[t];
        auto A = [&](int s, int nt) -> f4& {
            if constexpr (OWN) return f[s][b0 + nt];
            else return accs[s][nt];
        };
        if constexpr (!OWN && t > 0) addends(tc);
        // this lane's channels of the layer: c4 = 16 (b0 + nt) + 4 q .. + 3
        bool cv[NTG];
        f4 piv[NTG], s0[NTG], s1[NTG], ycur[OWN ? 1 : TPW][NTG];
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) {
            const int c4 = 16 * (b0 + nt) + 4 * q;
            cv[nt] = c4 >= col0 && c4 < col0 + G;
            piv[nt] = pf_splat(0.f);
            if (cv[nt] && a.run_mean[t]) piv[nt] = *reinterpret_cast<const f4*>(a.run_mean[t] + (c4 - col0));
            s0[nt] = s1[nt] = pf_splat(0.f);
        }
        if constexpr (t > 0) {                                  // the tiles' MFMA chains, interleaved (independent accumulators)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int nt = 0; nt < NTG; ++nt) {
                    const f4 w = *reinterpret_cast<const f4*>(Wl + (nt * 16 + col) * kp + ks * 16 + 4 * q);
#pragma unroll
                    for (int s = 0; s < TPW; ++s) A(s, nt) = mfma4(w, f[s][ks], A(s, nt));
                }
        }
#pragma unroll
        for (int s = 0; s < TPW; ++s)
#pragma unroll
            for (int nt = 0; nt < NTG; ++nt) {
                const f4 v = A(s, nt);
                if constexpr (!OWN) ycur[s][nt] = v;
                if (cv[nt] && ok[s]) {
                    const f4 vc = v - piv[nt];
                    s0[nt] = s0[nt] + vc;
                    s1[nt] = s1[nt] + vc * vc;
                }
            }
        // ---- column sums: DPP row -> LDS over the waves -> spread double accumulators (columns 32 t ..: a layer has its own,
        // nothing has to be cleared between two barriers of a launch)
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a0 = ecp_rowsum16(s0[nt][r]), a1 = ecp_rowsum16(s1[nt][r]);
                const int c = 16 * (b0 + nt) + 4 * q + r - col0;
                if (col == 0 && c >= 0 && c < G) { red[wave * 64 + c] = a0; red[wave * 64 + 32 + c] = a1; }
            }
        __syncthreads();
        if (!(PF_ECP_DBG & 16) && threadIdx.x < 64 && (threadIdx.x & 31) < G) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < ECP_WAVES; ++w) v += red[w * 64 + threadIdx.x];
            unsafeAtomicAdd(a.acc + (blockIdx.x % STAT_COPIES) * 2 * STAT_W + (threadIdx.x >> 5) * STAT_W + 32 * t + (threadIdx.x & 31), (double)v);
        }
        if constexpr (OWN && t + 1 < NC) addends(std::integral_constant<int, t + 1>{});      // next layer's gathers fly during the barrier
        // the pivot of the column this thread finalises below, read BEFORE the barrier: workgroup 0 updates the running mean right
        // after it (every workgroup has arrived, i.e. has read its pivots, by then)
        const int fc = threadIdx.x >> 3;
        const float fpv = (a.run_mean[t] && threadIdx.x < 256 && fc < G) ? a.run_mean[t][fc] : 0.f;
        alive = ecp_barrier(a.sync, (unsigned)(t + 1), &flag);
        if (!alive) return;
        // ---- every workgroup turns the sums into the layer's constants for itself (the StatFin mode-1 arithmetic); workgroup 0
        // also leaves them in `aff` for the backward and updates the running statistics
        // 256 threads: thread (part = tid & 3, stat = (tid >> 2) & 1, column = tid >> 3) fetches 4 of the 16 copies of one sum
        // (all loads of the workgroup in flight at once, 8 registers each), the 4 parts meet through lane shuffles
        double part = 0.0;
        if (threadIdx.x < 256) {
            const int pt = threadIdx.x & 3, stt = (threadIdx.x >> 2) & 1, c = threadIdx.x >> 3;
            double v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] = __hip_atomic_load(a.acc + (4 * pt + k) * 2 * STAT_W + stt * STAT_W + 32 * t + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            part = (v[0] + v[1]) + (v[2] + v[3]);
            part += __shfl_xor(part, 1);
            part += __shfl_xor(part, 2);                       // lanes pt = 0..3 now hold the sum over all 16 copies
        }
        const double other = __shfl_xor(part, 4);              // the other statistic of the same column
        if (threadIdx.x < 256 && (threadIdx.x & 7) == 0 && (threadIdx.x >> 3) < G) {
            const int c = threadIdx.x >> 3;
            const double a0 = part, a1 = other;
            const double pv = (double)fpv;
            const double dm = a0 / a.R;
            const double mean = pv + dm;
            double var = a1 / a.R - dm * dm;
            if (var < 0.0) var = 0.0;
            const float rstd = 1.0f / sqrtf((float)var + a.eps);
            const float sc = a.gamma[t][c] * rstd, sh = a.beta[t][c] - (float)mean * sc;
            scsh[c] = sc;
            scsh[32 + c] = sh;
            if (blockIdx.x == 0) {
                a.aff[col0 + c] = sc;
                a.aff[a.ldy + col0 + c] = sh;
                a.aff[2 * a.ldy + col0 + c] = (float)mean;
                a.aff[3 * a.ldy + col0 + c] = rstd;
                if (a.run_mean[t]) {            // every workgroup read its pivot before it arrived at the barrier above
                    a.run_mean[t][c] = (1.f - a.momentum) * a.run_mean[t][c] + a.momentum * (float)mean;
                    a.run_var[t][c] = (1.f - a.momentum) * a.run_var[t][c] + a.momentum * (float)(var * (a.R / (a.R - 1.0)));
                }
            }
        }
        __syncthreads();
        // ---- BatchNorm + LeakyReLU on the tiles this wave still holds: they become f[.][b0 ..] (rows of other layers that share
        // the block stay as they are: scale = shift = 0 outside the layer gives lrelu(0) = 0)
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) {
            f4 sc = pf_splat(0.f), sh = pf_splat(0.f);
            if (cv[nt]) {
                const int cl = 16 * (b0 + nt) + 4 * q - col0;
                sc = *reinterpret_cast<const f4*>(scsh + cl);
                sh = *reinterpret_cast<const f4*>(scsh + 32 + cl);
            }
            // the raw tile goes to memory only now (the backward reads it): its stores are in flight during the next layer
            // instead of in front of this layer's barrier, whose s_waitcnt would have waited for them
#pragma unroll
            for (int s = 0; s < TPW; ++s) {
                f4 raw;
                if constexpr (OWN) raw = f[s][b0 + nt];
                else raw = ycur[s][nt];
                if (cv[nt] && ok[s] && !(PF_ECP_DBG & 8))
                    *reinterpret_cast<f4*>(a.Y + ((size_t)tl[s] * 16 + col) * a.ldy + 16 * (b0 + nt) + 4 * q) = raw;
                if constexpr (OWN) f[s][b0 + nt] = lrelu4(raw * sc + sh, a.slope);
                else f[s][b0 + nt] = f[s][b0 + nt] + lrelu4(raw * sc + sh, a.slope);
            }
        }
    });

    // ---- conv_out + max over the 16 edges of the point
    // Operands swapped against the growth layers: the feature pair registers are bit for bit also the A operand with the EDGES on
    // the MFMA rows, the weight fragments the B operand with the channels on the columns, so D[edge][channel] puts 4 edges of ONE
    // channel into a lane - the max over the 16 edges is 3 in-lane comparisons + 2 exchanges across the lane rows instead of a
    // 16-lane reduction per value.  The addend comes in the same layout: P_out[i][c] once, Q_out[j_k][c] for the lane's four
    // edges k = 4 q + r (4-byte gathers, 64 B per 16 lanes).
    // What bounded this phase (58 of 128 us, and the 60 us of ec_fwd16_kernel) is the LDS weight stream: every tile read all
    // 64 KiB of fragments.  Here the features are converted ONCE into split operand pairs - in place of the fp32 registers they
    // replace, same count - and every fragment read serves all of the wave's tiles: a quarter of the LDS bytes.
    if (alive && !(PF_ECP_DBG & 1)) {
        PfPair2 fp[TPW][NCP];
#pragma unroll
        for (int s = 0; s < TPW; ++s)
#pragma unroll
            for (int cp = 0; cp < NCP; ++cp) fp[s][cp] = pf_pair2(f[s][2 * cp], f[s][2 * cp + 1]);
        int jq[TPW][4];
#pragma unroll
        for (int s = 0; s < TPW; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) jq[s][r] = __shfl(jr[s], 4 * q + r);
#pragma unroll
        for (int oc = 0; oc < NTO; oc += OCH) {
            f4 acc[TPW][OCH], accx[TPW][OCH];
#pragma unroll
            for (int s = 0; s < TPW; ++s)
#pragma unroll
                for (int o = 0; o < OCH; ++o) {
                    const int c = GT + 16 * (oc + o) + col;
                    const float pv = a.pq[(size_t)tl[s] * a.ldpq + c];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[s][o][r] = pv + a.pq[(size_t)jq[s][r] * a.ldpq + a.S + c];
                    accx[s][o] = pf_splat(0.f);
                }
#pragma unroll
            for (int cp = 0; cp < NCP; ++cp)
#pragma unroll
                for (int o = 0; o < OCH; ++o) {
                    const h8 wh = __builtin_bit_cast(h8, Wf[(((oc + o) * NCP + cp) * 2 + 0) * 64 + lane]);
                    const h8 wl = __builtin_bit_cast(h8, Wf[(((oc + o) * NCP + cp) * 2 + 1) * 64 + lane]);
#pragma unroll
                    for (int s = 0; s < TPW; ++s) {
                        acc[s][o] = pf_mfma_f16(fp[s][cp].h, wh, acc[s][o]);
                        accx[s][o] = pf_mfma_f16(fp[s][cp].l, wh, accx[s][o]);
                        accx[s][o] = pf_mfma_f16(fp[s][cp].h, wl, accx[s][o]);
                    }
                }
#pragma unroll
            for (int s = 0; s < TPW; ++s)
#pragma unroll
                for (int o = 0; o < OCH; ++o) {
                    const f4 v = acc[s][o] + accx[s][o] * PF_LO_INV;
                    float best = v[0];
                    int bk = 4 * q;
#pragma unroll
                    for (int r = 1; r < 4; ++r)
                        if (v[r] > best) { best = v[r]; bk = 4 * q + r; }
#pragma unroll
                    for (int m = 16; m < 64; m <<= 1) {
                        const float ov = __shfl_xor(best, m);
                        const int okk = __shfl_xor(bk, m);
                        if (ov > best || (ov == best && okk < bk)) { best = ov; bk = okk; }
                    }
                    if (q == 0 && ok[s]) {
                        const size_t o0 = (size_t)tl[s] * ODIM + 16 * (oc + o) + col;
                        a.out[o0] = best;
                        a.arg[o0] = (unsigned char)bk;
                    }
                }
        }
    } else {
#pragma unroll
        for (int s = 0; s < TPW; ++s)
            if (ok[s] && col == 0)
                for (int c = 4 * q; c < ODIM; c += 16)
                    *reinterpret_cast<f4*>(a.out + (size_t)tl[s] * ODIM + c) = pf_splat(__builtin_nanf(""));
    }
    // ---- the workgroup that leaves last clears the accumulator columns the layers used and puts the barrier words back to zero
    // (every workgroup is past every barrier and has read every sum by then)
    __syncthreads();
    if (threadIdx.x == 0) flag = atomicAdd(a.sync + 2, 1u) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (flag == 1) {
        for (int i = threadIdx.x; i < STAT_COPIES * 2 * STAT_W; i += ECP_T)
            if ((i % STAT_W) < 32 * NC) __hip_atomic_store(a.acc + i, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) {
            __hip_atomic_store(a.sync + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.sync + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward through one conv
// SRC 0: the conv is conv_out of a pooled unit: dYout[e, c] = dh[i, c] if argmax[i, c] == k else 0, formed on load
// SRC 1: conv_out without pooling: dYout [E, kin] dense
// SRC 2: growth conv t: dy = BatchNorm+LeakyReLU backward of dA[:, c0 : c0 + kin], formed on load AND stored back in place
// result: dA[:, 0 : nout]  =  (SRC 2: +=)  dYsrc W      (W[c * ldw + u], c < kin, u < nout)
// epilogue: BatchNorm-backward sums (sum dz, sum dz xhat) of the layer in columns [sc0, sc0 + sg), whose gradient is now final
struct EcBwdArgs {
    const float* dh; const unsigned char* arg;
    const float* dyout;
    float* dA; const float* Y; int ld;
    const float* aff;                // [4][ld]
    const float* coef;               // [2][ld]
    int c0, kin;
    const float* W; int ldw;
    int nout;
    int sc0, sg;
    int ntiles;
    float slope;
    StatFin fin;
};

template <int NT, int SRC>
__global__ __launch_bounds__(256) void ec_bwd_kernel(EcBwdArgs a) {
    extern __shared__ float lds[];
    __shared__ float red[8 * STAT_W];
    const int kin16 = (a.kin + 15) & ~15, kp = kin16 + 4, KS = kin16 / 16;
    float* Wt = lds;                                   // Wt[u][c]
    float* cf = lds + NT * 16 * kp;                    // SRC 2: [6][kin16] scale, shift, mean, rstd, m1, m2 of the source layer
    for (int c = threadIdx.x >> 4; c < kin16; c += 16) {                       // 16 lanes along a weight row (u): coalesced
        float v[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int u = (threadIdx.x & 15) + 16 * k;
            v[k] = (c < a.kin && u < a.nout) ? a.W[(size_t)c * a.ldw + u] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NT; ++k) Wt[((threadIdx.x & 15) + 16 * k) * kp + c] = v[k];
    }
    if (SRC == 2) {
        for (int i = threadIdx.x; i < kin16; i += 256) {
            const bool ok = i < a.kin;
#pragma unroll
            for (int w = 0; w < 4; ++w) cf[w * kin16 + i] = ok ? a.aff[w * a.ld + a.c0 + i] : 0.f;
            cf[4 * kin16 + i] = ok ? a.coef[a.c0 + i] : 0.f;
            cf[5 * kin16 + i] = ok ? a.coef[a.ld + a.c0 + i] : 0.f;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, q = lane >> 4;
    float s0[NT], s1[NT], ssc[NT], ssh[NT], smu[NT], srs[NT];
    bool scol[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s0[nt] = s1[nt] = 0.f;
        const int col = nt * 16 + row;
        scol[nt] = col >= a.sc0 && col < a.sc0 + a.sg;
        ssc[nt] = scol[nt] ? a.aff[col] : 0.f;
        ssh[nt] = scol[nt] ? a.aff[a.ld + col] : 0.f;
        smu[nt] = scol[nt] ? a.aff[2 * a.ld + col] : 0.f;
        srs[nt] = scol[nt] ? a.aff[3 * a.ld + col] : 0.f;
    }
    for (int tile = blockIdx.x * 4 + wave; tile < a.ntiles; tile += gridDim.x * 4) {
        const long long e0 = (long long)tile * 16;
        // the tile's loads first
        f4 src[SRC == 2 ? 2 : 8];
        f4 ysrc[SRC == 2 ? 2 : 1];
        unsigned ag[SRC == 0 ? 8 : 1];
#pragma unroll
        for (int ks = 0; ks < (SRC == 2 ? 2 : 8); ++ks) {
            const int c = ks * 16 + 4 * q;
            src[ks] = pf_splat(0.f);
            if (ks < KS && c < a.kin) {
                if (SRC == 0) {
                    src[ks] = *reinterpret_cast<const f4*>(a.dh + (long long)tile * a.kin + c);
                    ag[ks] = *reinterpret_cast<const unsigned*>(a.arg + (long long)tile * a.kin + c);
                } else if (SRC == 1) {
                    src[ks] = *reinterpret_cast<const f4*>(a.dyout + (e0 + row) * a.kin + c);
                } else {
                    src[ks] = *reinterpret_cast<const f4*>(a.dA + (e0 + row) * a.ld + a.c0 + c);
                    ysrc[ks] = *reinterpret_cast<const f4*>(a.Y + (e0 + row) * a.ld + a.c0 + c);
                }
            }
        }
        float old[NT][4];
        if (SRC == 2) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = nt * 16 + row;
                    old[nt][r] = col < a.nout ? a.dA[(e0 + 4 * q + r) * a.ld + col] : 0.f;
                }
        }
        f4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = pf_splat(0.f);
#pragma unroll
        for (int ks = 0; ks < (SRC == 2 ? 2 : 8); ++ks) {
            if (ks < KS) {
                const int c = ks * 16 + 4 * q;
                f4 av = pf_splat(0.f);
                if (c < a.kin) {
                    if (SRC == 0) {
                        const f4 dv = src[ks];
                        const unsigned g4 = ag[ks];
                        av.x = (int)(g4 & 255u) == row ? dv.x : 0.f; av.y = (int)((g4 >> 8) & 255u) == row ? dv.y : 0.f;
                        av.z = (int)((g4 >> 16) & 255u) == row ? dv.z : 0.f; av.w = (int)(g4 >> 24) == row ? dv.w : 0.f;
                    } else if (SRC == 1) {
                        av = src[ks];
                    } else {
                        const f4 d = src[ks], y = ysrc[ks];
                        const f4 sc = *reinterpret_cast<const f4*>(cf + c), sh = *reinterpret_cast<const f4*>(cf + kin16 + c);
                        const f4 mu = *reinterpret_cast<const f4*>(cf + 2 * kin16 + c), rs = *reinterpret_cast<const f4*>(cf + 3 * kin16 + c);
                        const f4 m1 = *reinterpret_cast<const f4*>(cf + 4 * kin16 + c), m2 = *reinterpret_cast<const f4*>(cf + 5 * kin16 + c);
                        const f4 z = y * sc + sh;
                        const f4 xh = (y - mu) * rs;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const float dz = d[w] * (z[w] > 0.f ? 1.f : a.slope);
                            av[w] = sc[w] * (dz - m1[w] - xh[w] * m2[w]);
                        }
                        *reinterpret_cast<f4*>(a.dA + (e0 + row) * a.ld + a.c0 + c) = av;
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = mfma4(av, *reinterpret_cast<const f4*>(Wt + (nt * 16 + row) * kp + c), acc[nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = nt * 16 + row;
            if (col < a.nout) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long e = e0 + 4 * q + r;
                    float v = acc[nt][r];
                    if (SRC == 2) v += old[nt][r];
                    a.dA[e * a.ld + col] = v;
                    if (scol[nt]) {
                        const float y = a.Y[e * a.ld + col];
                        const float dz = v * (fmaf(y, ssc[nt], ssh[nt]) > 0.f ? 1.f : a.slope);
                        s0[nt] += dz;
                        s1[nt] = fmaf(dz, (y - smu[nt]) * srs[nt], s1[nt]);
                    }
                }
            }
        }
    }
    stat_flush<NT>(s0, s1, a.sc0, a.sg, a.fin, red);
}

// ------------------------------------------------------------------------------------------------ backward, gather form
// The scatter form above walks the convs from last to first and ADDS each one's contribution into the gradient columns of all
// earlier layers: the [E, GT] gradient tensor is read and rewritten once per layer (452 MB per 128-channel unit, 2.5 - 2.8 TB/s
// in every one of those kernels).  Here layer s GATHERS its own g columns from everything that consumes them, once:
//     G_s = dYout Wout[:, cols_s] + sum_{t > s} dy_t W_t[:, cols_s]
// with dy_t = BatchNorm + LeakyReLU backward of G_t (formed on load for t = s + 1, whose sums the previous launch finalised, and
// stored back; already in place for t > s + 1).  Same products, 4 + 4 + ... launches replaced by one per layer, every gradient
// column written once raw and once transformed: ~230 MB per unit.  Epilogue: the BatchNorm-backward sums of layer s.
struct EcBwdgArgs {
    const float* dh; const unsigned char* arg; const float* dyout; int odim;
    float* dA; const float* Y; int ld;
    const float* aff; const float* coef;
    const float* Wout; int ldwout;       // Wout[c * ldwout + col]: conv_out row c, growth column col (pointer offset by 3C)
    const float* Wg[8]; int ldwg[8];     // growth conv t: Wg[t][c * ldwg[t] + col]
    int s, nc, g, ntiles;
    float slope;
    StatFin fin;
};

template <int NTG, int SRC>
__global__ __launch_bounds__(256) void ec_bwdg_kernel(EcBwdgArgs a) {
    extern __shared__ float lds[];
    __shared__ float red[8 * STAT_W];
    const int g = a.g, g16 = (g + 15) & ~15, od16 = (a.odim + 15) & ~15;
    const int kpo = od16 + 4, kpg = g16 + 4, KSo = od16 / 16, KSg = g16 / 16;
    const int c0 = g * a.s, nsrc = a.nc - 1 - a.s;
    float* Wo = lds;                                   // Wo[u][c] = Wout[c][c0 + u]
    float* Wgl = Wo + NTG * 16 * kpo;                  // per later layer t: [NTG * 16][kpg], Wg_t[u][c] = Wg[t][c][c0 + u]
    float* cf = Wgl + nsrc * NTG * 16 * kpg;           // [6][g16]: scale, shift, mean, rstd, m1, m2 of layer s + 1
    for (int c = threadIdx.x >> 4; c < od16; c += 16) {
        float v[NTG];
#pragma unroll
        for (int k = 0; k < NTG; ++k) {
            const int u = (threadIdx.x & 15) + 16 * k;
            v[k] = (c < a.odim && u < g) ? a.Wout[(size_t)c * a.ldwout + c0 + u] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NTG; ++k) Wo[((threadIdx.x & 15) + 16 * k) * kpo + c] = v[k];
    }
    for (int n = 0; n < nsrc; ++n) {
        const int t = a.s + 1 + n;
        const float* W = a.Wg[t];
        const int ldw = a.ldwg[t];
        float* dst = Wgl + n * NTG * 16 * kpg;
        for (int c = threadIdx.x >> 4; c < g16; c += 16) {
#pragma unroll
            for (int k = 0; k < NTG; ++k) {
                const int u = (threadIdx.x & 15) + 16 * k;
                dst[u * kpg + c] = (c < g && u < g) ? W[(size_t)c * ldw + c0 + u] : 0.f;
            }
        }
    }
    if (nsrc > 0) {
        const int c1 = c0 + g;
        for (int i = threadIdx.x; i < g16; i += 256) {
            const bool ok = i < g;
#pragma unroll
            for (int w = 0; w < 4; ++w) cf[w * g16 + i] = ok ? a.aff[w * a.ld + c1 + i] : 0.f;
            cf[4 * g16 + i] = ok ? a.coef[c1 + i] : 0.f;
            cf[5 * g16 + i] = ok ? a.coef[a.ld + c1 + i] : 0.f;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, q = lane >> 4;
    float s0[NTG], s1[NTG], ssc[NTG], ssh[NTG], smu[NTG], srs[NTG];
#pragma unroll
    for (int nt = 0; nt < NTG; ++nt) {
        s0[nt] = s1[nt] = 0.f;
        const int cl = nt * 16 + row;
        const bool ok = cl < g;
        ssc[nt] = ok ? a.aff[c0 + cl] : 0.f;
        ssh[nt] = ok ? a.aff[a.ld + c0 + cl] : 0.f;
        smu[nt] = ok ? a.aff[2 * a.ld + c0 + cl] : 0.f;
        srs[nt] = ok ? a.aff[3 * a.ld + c0 + cl] : 0.f;
    }
    for (int tile = blockIdx.x * 4 + wave; tile < a.ntiles; tile += gridDim.x * 4) {
        const long long e0 = (long long)tile * 16;
        // the tile's loads first: conv_out's gradient rows, then the later layers' gradient columns of this lane's edge
        f4 src[8];
        unsigned ag[SRC == 0 ? 8 : 1];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int c = ks * 16 + 4 * q;
            src[ks] = pf_splat(0.f);
            if (ks < KSo && c < a.odim) {
                if (SRC == 0) {
                    src[ks] = *reinterpret_cast<const f4*>(a.dh + (long long)tile * a.odim + c);
                    ag[ks] = *reinterpret_cast<const unsigned*>(a.arg + (long long)tile * a.odim + c);
                } else src[ks] = *reinterpret_cast<const f4*>(a.dyout + (e0 + row) * a.odim + c);
            }
        }
        float* drow = a.dA + (e0 + row) * a.ld;
        f4 gsrc[7][2];
        f4 ysrc[2];
#pragma unroll
        for (int n = 0; n < 7; ++n)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                gsrc[n][ks] = pf_splat(0.f);
                const int c = ks * 16 + 4 * q;
                if (n < nsrc && ks < KSg && c < g) gsrc[n][ks] = *reinterpret_cast<const f4*>(drow + c0 + g * (n + 1) + c);
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            ysrc[ks] = pf_splat(0.f);
            const int c = ks * 16 + 4 * q;
            if (nsrc > 0 && ks < KSg && c < g) ysrc[ks] = *reinterpret_cast<const f4*>(a.Y + (e0 + row) * a.ld + c0 + g + c);
        }
        f4 acc[NTG];
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) acc[nt] = pf_splat(0.f);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks < KSo) {
                const int c = ks * 16 + 4 * q;
                f4 av = src[ks];
                if (SRC == 0) {
                    const f4 dv = src[ks];
                    const unsigned g4 = ag[ks];
                    av.x = (int)(g4 & 255u) == row ? dv.x : 0.f; av.y = (int)((g4 >> 8) & 255u) == row ? dv.y : 0.f;
                    av.z = (int)((g4 >> 16) & 255u) == row ? dv.z : 0.f; av.w = (int)(g4 >> 24) == row ? dv.w : 0.f;
                    if (c >= a.odim) av = pf_splat(0.f);
                }
#pragma unroll
                for (int nt = 0; nt < NTG; ++nt)
                    acc[nt] = mfma4(av, *reinterpret_cast<const f4*>(Wo + (nt * 16 + row) * kpo + c), acc[nt]);
            }
        }
#pragma unroll
        for (int n = 0; n < 7; ++n) {
            if (n < nsrc) {
                const float* Wt = Wgl + n * NTG * 16 * kpg;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (ks < KSg) {
                        const int c = ks * 16 + 4 * q;
                        f4 av = gsrc[n][ks];
                        if (n == 0 && c < g) {                      // layer s + 1: raw gradient -> dy, stored back
                            const f4 d = gsrc[0][ks], y = ysrc[ks];
                            const f4 sc = *reinterpret_cast<const f4*>(cf + c), sh = *reinterpret_cast<const f4*>(cf + g16 + c);
                            const f4 mu = *reinterpret_cast<const f4*>(cf + 2 * g16 + c), rs = *reinterpret_cast<const f4*>(cf + 3 * g16 + c);
                            const f4 m1 = *reinterpret_cast<const f4*>(cf + 4 * g16 + c), m2 = *reinterpret_cast<const f4*>(cf + 5 * g16 + c);
                            const f4 z = y * sc + sh;
                            const f4 xh = (y - mu) * rs;
#pragma unroll
                            for (int w = 0; w < 4; ++w) {
                                const float dz = d[w] * (z[w] > 0.f ? 1.f : a.slope);
                                av[w] = sc[w] * (dz - m1[w] - xh[w] * m2[w]);
                            }
                            *reinterpret_cast<f4*>(drow + c0 + g + c) = av;
                        }
#pragma unroll
                        for (int nt = 0; nt < NTG; ++nt)
                            acc[nt] = mfma4(av, *reinterpret_cast<const f4*>(Wt + (nt * 16 + row) * kpg + c), acc[nt]);
                    }
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) {
            const int cl = nt * 16 + row;
            if (cl < g) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long e = e0 + 4 * q + r;
                    const float v = acc[nt][r];
                    a.dA[e * a.ld + c0 + cl] = v;
                    const float y = a.Y[e * a.ld + c0 + cl];
                    const float dz = v * (fmaf(y, ssc[nt], ssh[nt]) > 0.f ? 1.f : a.slope);
                    s0[nt] += dz;
                    s1[nt] = fmaf(dz, (y - smu[nt]) * srs[nt], s1[nt]);
                }
            }
        }
    }
    stat_flush<NTG>(s0, s1, 0, g, a.fin, red);
}

// The same gather on the bf16 matrix pipe (both operands split as in ec_dw3_kernel: x = hi + mid, fp32 exponent range, three
// v_mfma_f32_16x16x32_bf16 per 32 channels instead of eight f32 MFMAs): lane (row = edge, kg = l >> 4) holds the 8 channels
// 8 kg .. 8 kg + 7 of a 32-channel chunk of its edge's gradient row (two 16-byte loads), the weights sit in LDS as ready
// fragments [tile][chunk][hi | mid][lane][8 x bf16] written once per workgroup.  Accumulators, epilogue and statistics are those of
// the f32 kernel (the 16x16 accumulator layout does not depend on the input type).
typedef __bf16 gbf8 __attribute__((ext_vector_type(8)));
struct GBf2 { gbf8 hi, mid; };
__device__ __forceinline__ GBf2 g_split(const float (&x)[8]) {
    unsigned hw[4], mw[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned b0 = __float_as_uint(x[2 * p]), b1 = __float_as_uint(x[2 * p + 1]);
        const unsigned h0 = b0 & 0xffff0000u, h1 = b1 & 0xffff0000u;
        const unsigned m0 = __float_as_uint(x[2 * p] - __uint_as_float(h0)), m1 = __float_as_uint(x[2 * p + 1] - __uint_as_float(h1));
        hw[p] = (h0 >> 16) | h1;
        mw[p] = (m0 >> 16) | (m1 & 0xffff0000u);
    }
    GBf2 r;
    r.hi = __builtin_bit_cast(gbf8, *reinterpret_cast<const uint4*>(hw));
    r.mid = __builtin_bit_cast(gbf8, *reinterpret_cast<const uint4*>(mw));
    return r;
}

template <int NTG, int SRC>
__global__ __launch_bounds__(256) void ec_bwdg16_kernel(EcBwdgArgs a) {
    extern __shared__ float lds[];
    __shared__ float red[8 * STAT_W];
    const int g = a.g, g16 = (g + 15) & ~15;
    const int nco = (a.odim + 31) / 32, ncg = (g + 31) / 32;       // 32-channel chunks of conv_out / of a growth layer
    const int c0 = g * a.s, nsrc = a.nc - 1 - a.s;
    uint4* Wf = reinterpret_cast<uint4*>(lds);                    // fragments: ((frag * 2 + hi|mid) * 64 + lane), 16 bytes each
    const int nfrag = NTG * (nco + nsrc * ncg);                   // frag = nt * nco + chunk | NTG * nco + (n * NTG + nt) * ncg + chunk
    float* cf = lds + (size_t)nfrag * 2 * 64 * 4;                 // [6][g16]: scale, shift, mean, rstd, m1, m2 of layer s + 1
    for (int unit = threadIdx.x; unit < nfrag * 64; unit += 256) {
        const int frag = unit >> 6, ln = unit & 63, u = ln & 15, kg = ln >> 4;
        const float* W;
        int ldw, kin, nt, chunk;
        if (frag < NTG * nco) { nt = frag / nco; chunk = frag % nco; W = a.Wout; ldw = a.ldwout; kin = a.odim; }
        else {
            const int f2 = frag - NTG * nco, n = f2 / (NTG * ncg), r2 = f2 % (NTG * ncg);
            nt = r2 / ncg; chunk = r2 % ncg; W = a.Wg[a.s + 1 + n]; ldw = a.ldwg[a.s + 1 + n]; kin = g;
        }
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = chunk * 32 + 8 * kg + j, uu = nt * 16 + u;
            v[j] = (c < kin && uu < g) ? W[(size_t)c * ldw + c0 + uu] : 0.f;
        }
        const GBf2 f = g_split(v);
        Wf[(frag * 2 + 0) * 64 + ln] = __builtin_bit_cast(uint4, f.hi);
        Wf[(frag * 2 + 1) * 64 + ln] = __builtin_bit_cast(uint4, f.mid);
    }
    if (nsrc > 0) {
        const int c1 = c0 + g;
        for (int i = threadIdx.x; i < g16; i += 256) {
            const bool ok = i < g;
#pragma unroll
            for (int w = 0; w < 4; ++w) cf[w * g16 + i] = ok ? a.aff[w * a.ld + c1 + i] : 0.f;
            cf[4 * g16 + i] = ok ? a.coef[c1 + i] : 0.f;
            cf[5 * g16 + i] = ok ? a.coef[a.ld + c1 + i] : 0.f;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, q = lane >> 4;
    float s0[NTG], s1[NTG], ssc[NTG], ssh[NTG], smu[NTG], srs[NTG];
#pragma unroll
    for (int nt = 0; nt < NTG; ++nt) {
        s0[nt] = s1[nt] = 0.f;
        const int cl = nt * 16 + row;
        const bool ok = cl < g;
        ssc[nt] = ok ? a.aff[c0 + cl] : 0.f;
        ssh[nt] = ok ? a.aff[a.ld + c0 + cl] : 0.f;
        smu[nt] = ok ? a.aff[2 * a.ld + c0 + cl] : 0.f;
        srs[nt] = ok ? a.aff[3 * a.ld + c0 + cl] : 0.f;
    }
    auto mma = [&](const GBf2& A, int frag, f4& acc) {
        const gbf8 bh = __builtin_bit_cast(gbf8, Wf[(frag * 2 + 0) * 64 + lane]);
        const gbf8 bm = __builtin_bit_cast(gbf8, Wf[(frag * 2 + 1) * 64 + lane]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.mid, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.hi, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A.hi, bh, acc, 0, 0, 0);
    };
    for (int tile = blockIdx.x * 4 + wave; tile < a.ntiles; tile += gridDim.x * 4) {
        const long long e0 = (long long)tile * 16;
        // the tile's loads first: 8 channels per 32-channel chunk of conv_out's gradient row and of the later layers' columns
        float csrc[4][8];
        unsigned ag[SRC == 0 ? 4 : 1][2];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const int c = ch * 32 + 8 * q;
#pragma unroll
            for (int j = 0; j < 8; ++j) csrc[ch][j] = 0.f;
            if (ch < nco && c < a.odim) {                            // odim is a multiple of 16: an 8-channel group is all in or all out
                const float* sp = SRC == 0 ? a.dh + (long long)tile * a.odim + c : a.dyout + (e0 + row) * a.odim + c;
                const f4 v0 = *reinterpret_cast<const f4*>(sp), v1 = *reinterpret_cast<const f4*>(sp + 4);
                csrc[ch][0] = v0.x; csrc[ch][1] = v0.y; csrc[ch][2] = v0.z; csrc[ch][3] = v0.w;
                csrc[ch][4] = v1.x; csrc[ch][5] = v1.y; csrc[ch][6] = v1.z; csrc[ch][7] = v1.w;
                if (SRC == 0) {
                    const unsigned* apw = reinterpret_cast<const unsigned*>(a.arg + (long long)tile * a.odim + c);
                    ag[ch][0] = apw[0]; ag[ch][1] = apw[1];
                }
            }
        }
        float* drow = a.dA + (e0 + row) * a.ld;
        float gsrc[7][8], ysrc[8];
        const int cq = 8 * q;                                      // g <= 32: one chunk per growth layer
#pragma unroll
        for (int n = 0; n < 7; ++n) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gsrc[n][j] = 0.f;
            if (n < nsrc && cq < g) {                              // g is a multiple of 8
                const float* sp = drow + c0 + g * (n + 1) + cq;
                const f4 v0 = *reinterpret_cast<const f4*>(sp), v1 = *reinterpret_cast<const f4*>(sp + 4);
                gsrc[n][0] = v0.x; gsrc[n][1] = v0.y; gsrc[n][2] = v0.z; gsrc[n][3] = v0.w;
                gsrc[n][4] = v1.x; gsrc[n][5] = v1.y; gsrc[n][6] = v1.z; gsrc[n][7] = v1.w;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) ysrc[j] = 0.f;
        if (nsrc > 0 && cq < g) {
            const float* sp = a.Y + (e0 + row) * a.ld + c0 + g + cq;
            const f4 v0 = *reinterpret_cast<const f4*>(sp), v1 = *reinterpret_cast<const f4*>(sp + 4);
            ysrc[0] = v0.x; ysrc[1] = v0.y; ysrc[2] = v0.z; ysrc[3] = v0.w; ysrc[4] = v1.x; ysrc[5] = v1.y; ysrc[6] = v1.z; ysrc[7] = v1.w;
        }
        f4 acc[NTG];
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) acc[nt] = pf_splat(0.f);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            if (ch < nco) {
                float av[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    av[j] = csrc[ch][j];
                    if (SRC == 0) {
                        const unsigned wv = ag[ch][j >> 2];
                        av[j] = (int)((wv >> (8 * (j & 3))) & 255u) == row ? csrc[ch][j] : 0.f;
                    }
                }
                const GBf2 A = g_split(av);
#pragma unroll
                for (int nt = 0; nt < NTG; ++nt) mma(A, nt * nco + ch, acc[nt]);
            }
        }
#pragma unroll
        for (int n = 0; n < 7; ++n) {
            if (n < nsrc) {
                float av[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) av[j] = gsrc[n][j];
                if (n == 0 && cq < g) {                            // layer s + 1: raw gradient -> dy, stored back
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int c = cq + j;
                        const float sc = cf[c], sh = cf[g16 + c], mu = cf[2 * g16 + c], rs = cf[3 * g16 + c];
                        const float m1 = cf[4 * g16 + c], m2 = cf[5 * g16 + c];
                        const float y = ysrc[j], z = fmaf(y, sc, sh), xh = (y - mu) * rs;
                        const float dz = gsrc[0][j] * (z > 0.f ? 1.f : a.slope);
                        av[j] = sc * (dz - m1 - xh * m2);
                    }
                    float* dp = drow + c0 + g + cq;
                    f4 o0 = {av[0], av[1], av[2], av[3]}, o1 = {av[4], av[5], av[6], av[7]};
                    *reinterpret_cast<f4*>(dp) = o0;
                    *reinterpret_cast<f4*>(dp + 4) = o1;
                }
                const GBf2 A = g_split(av);
#pragma unroll
                for (int nt = 0; nt < NTG; ++nt) mma(A, NTG * nco + (n * NTG + nt) * ncg, acc[nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) {
            const int cl = nt * 16 + row;
            if (cl < g) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long e = e0 + 4 * q + r;
                    const float v = acc[nt][r];
                    a.dA[e * a.ld + c0 + cl] = v;
                    const float y = a.Y[e * a.ld + c0 + cl];
                    const float dz = v * (fmaf(y, ssc[nt], ssh[nt]) > 0.f ? 1.f : a.slope);
                    s0[nt] += dz;
                    s1[nt] = fmaf(dz, (y - smu[nt]) * srs[nt], s1[nt]);
                }
            }
        }
    }
    stat_flush<NTG>(s0, s1, 0, g, a.fin, red);
}

// ------------------------------------------------------------------------------------------------ backward of the dense block in ONE launch
// The gather-form backward above is one launch per growth layer (+ ec_bwd0_kernel): layer s reads conv_out's gradient and the dy
// of EVERY later layer from memory again, and each launch ends in the BatchNorm-backward sums of its layer.  Same construction
// as ec_fwdp_kernel: a persistent grid (one workgroup per CU, 8 waves, a wave owns up to ECP_TPW tiles for the whole launch) keeps
// the tile's dy of the layers done so far in registers and meets at a grid barrier once per layer:
//   layer s = NC-1 .. 0:  G_s = dYout Wout[:, cols_s] + sum_{t > s} dy_t W_t[:, cols_s]   (channel-major: the columns of layer s on
//                          the MFMA rows, edges on the columns - a dy tile is the B operand of every earlier layer as it stands;
//                          split-bf16 products like ec_bwdg16_kernel: x = hi + mid, three v_mfma_f32_16x16x32_bf16 per 32
//                          channels; the weights are split once per workgroup into A fragments in LDS; dYout is formed from
//                          (dh, argmax) on load)
//                        dz = G_s lrelu'(bn(y_s));  sums of dz and dz xhat -> barrier -> dy_s = scale (dz - m1 - xhat m2), kept in
//                          registers for the layers below and stored ONCE to dA (the weight-gradient and dPQ kernels read it).
// dA ends up exactly as ec_bwd0_kernel leaves it; coef / dgamma / dbeta are written by workgroup 0.
struct EcBwdPArgs {
    const float* dh; const unsigned char* arg;       // [T, ODIM]
    float* dA; const float* Y; int ld;               // [E, GT]
    const float* aff; float* coef;                   // [4][GT], [2][GT]
    const float* Wout; int ldwout;
    const float* Wg[8]; int ldwg[8];
    float* dgamma[8]; float* dbeta[8];
    int ntiles;
    float slope; double R;
    double* acc; unsigned* sync;
    float* dP; int ldp;                              // nullable: dPQ [T, ldp] - the P half's growth columns = sum of dA over a point's 16 edges
};

// -DPF_EC_BWDP_DP=0: the P half's growth columns are summed by ec_pq_bwd_csr_kernel from dA again (the A/B reference)
#ifndef PF_EC_BWDP_DP
#define PF_EC_BWDP_DP 1
#endif

template <int G, int NC, int ODIM>
__global__ __launch_bounds__(ECP_T) void ec_bwdp_kernel(EcBwdPArgs a) {
    constexpr int TPW = ecp_tpw(G);
    constexpr int GT = G * NC, NB = GT / 16, NTG = (G + 15) / 16, NCO = ODIM / 32, NCP = GT / 32;
    constexpr bool OWN = G % 16 == 0;
    static_assert(GT % 32 == 0 && ODIM % 32 == 0 && 32 * NC <= STAT_W && G <= 32, "shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float red[ECP_WAVES * 2 * 32];
    __shared__ float m12[2 * 32];
    __shared__ float bnc2[2][4 * 32];     // double-buffered by layer parity: a wave may enter the next layer while another still reads
    __shared__ int flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
    // ---- A fragments (split-bf16, [hi | mid][lane] x 16 B): first conv_out's, one per (16-column block b of the growth
    // channels, 32-channel chunk co of odim) - Wout[c][16 b + row] - then per layer s its dy chunks cp >= cpmin(s) (32 growth
    // channels each: rows of the LATER layers' convs, zero for channels of layers <= s): frag fbase(s) + nt * ndy(s) + (cp - cpmin(s))
    auto cpmin = [](int s) { return (G * (s + 1)) / 32; };
    auto ndy = [&](int s) { return NCP - cpmin(s); };
    int fbase[NC + 1];
    fbase[0] = NB * NCO;
#pragma unroll
    for (int s = 0; s < NC; ++s) fbase[s + 1] = fbase[s] + NTG * ndy(s);
#ifdef PF_EC_BWDG_F32
    // A/B build (bench.py grad_parity): the SAME kernel on plain f32 products - fp32 images [row][k] with row = growth column,
    // k = source channel: Wo [GT][ODIM + 4] (conv_out), Ms [NC][NTG * 16][GT + 4] (the later layers' rows, zero for layers <= s)
    constexpr int KPO = ODIM + 4, KPG = GT + 4;
    float* Wo = lds;
    float* Ms = lds + GT * KPO;
    for (int i = threadIdx.x; i < GT * ODIM; i += ECP_T) Wo[(i % GT) * KPO + i / GT] = a.Wout[(size_t)(i / GT) * a.ldwout + i % GT];
    for (int i = threadIdx.x; i < NC * NTG * 16 * GT; i += ECP_T) {
        const int s = i / (NTG * 16 * GT), rw = (i / GT) % (NTG * 16), c = i % GT, t = c / G;
        const int ug = 16 * ((G * s) / 16) + rw;
        Ms[(s * NTG * 16 + rw) * KPG + c] = (ug >= G * s && ug < G * (s + 1) && t > s) ? a.Wg[t][(size_t)(c - G * t) * a.ldwg[t] + ug] : 0.f;
    }
#else
    uint4* Wf = reinterpret_cast<uint4*>(lds);
    {
        const int nunit = fbase[NC] * 64;
        constexpr int MAXU = ((NB * NCO + NTG * NC * NCP) * 64 + ECP_T - 1) / ECP_T;
        float v[MAXU][8];
#pragma unroll
        for (int k = 0; k < MAXU; ++k) {
            const int unit = threadIdx.x + k * ECP_T, uu = unit < nunit ? unit : 0;
            const int frag = uu >> 6, ln = uu & 63, row = ln & 15, kq = ln >> 4;
            if (frag < NB * NCO) {
                const int b = frag / NCO, co = frag % NCO, ug = 16 * b + row;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[k][j] = a.Wout[(size_t)(32 * co + (j < 4 ? 0 : 16) + 4 * kq + (j & 3)) * a.ldwout + ug];
            } else {
                int s = 0;
#pragma unroll
                for (int t = 1; t < NC; ++t) s = frag >= fbase[t] ? t : s;
                const int fr = frag - fbase[s], nd = ndy(s) > 0 ? ndy(s) : 1, nt = fr / nd, cp = cpmin(s) + fr % nd;
                const int b0 = (G * s) / 16, ug = 16 * (b0 + nt) + row;             // growth column of this output row
                const bool rowok = ug >= G * s && ug < G * (s + 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = 32 * cp + (j < 4 ? 0 : 16) + 4 * kq + (j & 3), t = c / G;
                    v[k][j] = (rowok && t > s) ? a.Wg[t][(size_t)(c - G * t) * a.ldwg[t] + ug] : 0.f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < MAXU; ++k) {
            const int unit = threadIdx.x + k * ECP_T;
            if (unit < nunit) {
                const GBf2 f2 = g_split(v[k]);
                Wf[((unit >> 6) * 2 + 0) * 64 + (unit & 63)] = __builtin_bit_cast(uint4, f2.hi);
                Wf[((unit >> 6) * 2 + 1) * 64 + (unit & 63)] = __builtin_bit_cast(uint4, f2.mid);
            }
        }
    }
#endif
    // ---- this wave's tiles
    int tl[TPW];
    bool ok[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
        tl[s] = blockIdx.x * ECP_WAVES + wave + s * gridDim.x * ECP_WAVES;
        ok[s] = tl[s] < a.ntiles;
        tl[s] = ok[s] ? tl[s] : 0;
    }
    f4 dy[TPW][NB];
#pragma unroll
    for (int s = 0; s < TPW; ++s)
#pragma unroll
        for (int b = 0; b < NB; ++b) dy[s][b] = pf_splat(0.f);
    __syncthreads();
#ifndef PF_EC_BWDG_F32
    auto mma = [&](int frag, const GBf2& B, f4& acc) {
        const gbf8 wh = __builtin_bit_cast(gbf8, Wf[(frag * 2 + 0) * 64 + lane]);
        const gbf8 wm = __builtin_bit_cast(gbf8, Wf[(frag * 2 + 1) * 64 + lane]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, B.hi, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, B.mid, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, B.hi, acc, 0, 0, 0);
    };
#endif

    // ---- conv_out's contribution to EVERY layer in one pass: dy[.][b] = Wout[:, 16 b ..]^T dYout, dYout formed from (dh, argmax)
    // and split once per (tile, chunk) - this lane's 8 channels of a 32-channel chunk, edge = col.  Layer s adds the later
    // layers' part into its own slots below.
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const float* dp = a.dh + (size_t)tl[t] * ODIM + 32 * co + 4 * q;
            const unsigned char* ap = a.arg + (size_t)tl[t] * ODIM + 32 * co + 4 * q;
            const f4 d0 = *reinterpret_cast<const f4*>(dp), d1 = *reinterpret_cast<const f4*>(dp + 16);
            const unsigned g0 = *reinterpret_cast<const unsigned*>(ap), g1 = *reinterpret_cast<const unsigned*>(ap + 16);
            float x[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                x[j] = (int)((g0 >> (8 * j)) & 255u) == col ? d0[j] : 0.f;
                x[4 + j] = (int)((g1 >> (8 * j)) & 255u) == col ? d1[j] : 0.f;
            }
#ifdef PF_EC_BWDG_F32
            const f4 x0 = {x[0], x[1], x[2], x[3]}, x1 = {x[4], x[5], x[6], x[7]};
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                dy[t][b] = mfma4(*reinterpret_cast<const f4*>(Wo + (16 * b + col) * KPO + 32 * co + 4 * q), x0, dy[t][b]);
                dy[t][b] = mfma4(*reinterpret_cast<const f4*>(Wo + (16 * b + col) * KPO + 32 * co + 16 + 4 * q), x1, dy[t][b]);
            }
#else
            const GBf2 B = g_split(x);
#pragma unroll
            for (int b = 0; b < NB; ++b) mma(b * NCO + co, B, dy[t][b]);
#endif
        }
    }

    bool alive = true;
    pf_static_for<0, NC>([&](auto sc_) {
        constexpr int s = NC - 1 - decltype(sc_)::value;             // layers last to first
        constexpr int col0 = G * s, b0 = col0 / 16, CP0 = (G * (s + 1)) / 32, NDY = NCP - CP0;
        if (!alive) return;
        const int fb = fbase[s];
        // the layer's BatchNorm constants (scale, shift, mean, 1/std) as f4 rows in LDS: read when needed, not held in registers
        float* bnc = bnc2[s & 1];
        if (threadIdx.x < 4 * 32) {
            const int w = threadIdx.x >> 5, c = threadIdx.x & 31;
            bnc[w * 32 + c] = c < G ? a.aff[w * a.ld + col0 + c] : 0.f;
        }
        bool cv[NTG];
        f4 yv[TPW][NTG], accs[OWN ? 1 : TPW][NTG];
        auto A = [&](int t, int nt) -> f4& {
            if constexpr (OWN) return dy[t][b0 + nt];                // the layer's own slots hold conv_out's part already
            else return accs[t][nt];
        };
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) {
            const int c4 = 16 * (b0 + nt) + 4 * q;
            cv[nt] = c4 >= col0 && c4 < col0 + G;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                yv[t][nt] = pf_splat(0.f);
                if (cv[nt] && ok[t]) yv[t][nt] = *reinterpret_cast<const f4*>(a.Y + ((size_t)tl[t] * 16 + col) * a.ld + c4);
                if constexpr (!OWN) accs[t][nt] = cv[nt] ? dy[t][b0 + nt] : pf_splat(0.f);   // rows of the layer that shares the block: not ours
            }
        }
        // ---- the later layers' dy (registers)
#pragma unroll
        for (int cp = CP0; cp < NCP; ++cp) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
#ifdef PF_EC_BWDG_F32
#pragma unroll
                for (int nt = 0; nt < NTG; ++nt) {
                    const float* mrow = Ms + ((size_t)s * NTG * 16 + nt * 16 + col) * KPG + 32 * cp + 4 * q;
                    A(t, nt) = mfma4(*reinterpret_cast<const f4*>(mrow), dy[t][2 * cp], A(t, nt));
                    A(t, nt) = mfma4(*reinterpret_cast<const f4*>(mrow + 16), dy[t][2 * cp + 1], A(t, nt));
                }
#else
                float x[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) { x[j] = dy[t][2 * cp][j]; x[4 + j] = dy[t][2 * cp + 1][j]; }
                const GBf2 B = g_split(x);
#pragma unroll
                for (int nt = 0; nt < NTG; ++nt) mma(fb + nt * NDY + cp - CP0, B, A(t, nt));
#endif
            }
        }
        __syncthreads();                                              // bnc
        // ---- dz (it replaces the raw gradient), its sums; xhat stays for the transform behind the barrier
        f4 s0[NTG], s1[NTG], xh[TPW][NTG];
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) {
            s0[nt] = s1[nt] = pf_splat(0.f);
            const int cl = cv[nt] ? 16 * (b0 + nt) + 4 * q - col0 : 0;
            const f4 bsc = *reinterpret_cast<const f4*>(bnc + cl), bsh = *reinterpret_cast<const f4*>(bnc + 32 + cl);
            const f4 bmu = *reinterpret_cast<const f4*>(bnc + 64 + cl), brs = *reinterpret_cast<const f4*>(bnc + 96 + cl);
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const f4 y = yv[t][nt], z = y * bsc + bsh;
                xh[t][nt] = (y - bmu) * brs;
                f4 dz;
#pragma unroll
                for (int r = 0; r < 4; ++r) dz[r] = A(t, nt)[r] * (z[r] > 0.f ? 1.f : a.slope);
                if (cv[nt]) {
                    A(t, nt) = dz;
                    if (ok[t]) { s0[nt] = s0[nt] + dz; s1[nt] = s1[nt] + dz * xh[t][nt]; }
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a0 = ecp_rowsum16(s0[nt][r]), a1 = ecp_rowsum16(s1[nt][r]);
                const int c = 16 * (b0 + nt) + 4 * q + r - col0;
                if (col == 0 && c >= 0 && c < G) { red[wave * 64 + c] = a0; red[wave * 64 + 32 + c] = a1; }
            }
        __syncthreads();
        if (threadIdx.x < 64 && (threadIdx.x & 31) < G) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < ECP_WAVES; ++w) v += red[w * 64 + threadIdx.x];
            unsafeAtomicAdd(a.acc + (blockIdx.x % STAT_COPIES) * 2 * STAT_W + (threadIdx.x >> 5) * STAT_W + 32 * s + (threadIdx.x & 31), (double)v);
        }
        alive = ecp_barrier(a.sync, (unsigned)(NC - s), &flag);
        if (!alive) return;
        double part = 0.0;
        if (threadIdx.x < 256) {
            const int pt = threadIdx.x & 3, stt = (threadIdx.x >> 2) & 1, c = threadIdx.x >> 3;
            double v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] = __hip_atomic_load(a.acc + (4 * pt + k) * 2 * STAT_W + stt * STAT_W + 32 * s + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            part = (v[0] + v[1]) + (v[2] + v[3]);
            part += __shfl_xor(part, 1);
            part += __shfl_xor(part, 2);
        }
        const double other = __shfl_xor(part, 4);
        if (threadIdx.x < 256 && (threadIdx.x & 7) == 0 && (threadIdx.x >> 3) < G) {
            const int c = threadIdx.x >> 3;
            m12[c] = (float)(part / a.R);
            m12[32 + c] = (float)(other / a.R);
            if (blockIdx.x == 0) {                                    // the StatFin mode-2 outputs
                a.coef[col0 + c] = (float)(part / a.R);
                a.coef[a.ld + col0 + c] = (float)(other / a.R);
                a.dbeta[s][c] = (float)part;
                a.dgamma[s][c] = (float)other;
            }
        }
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NTG; ++nt) {
            if (!cv[nt]) continue;
            const int cl = 16 * (b0 + nt) + 4 * q - col0;
            const f4 m1 = *reinterpret_cast<const f4*>(m12 + cl), m2 = *reinterpret_cast<const f4*>(m12 + 32 + cl);
            const f4 bsc = *reinterpret_cast<const f4*>(bnc + cl);
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const f4 v = bsc * (A(t, nt) - m1 - xh[t][nt] * m2);
                if (ok[t]) *reinterpret_cast<f4*>(a.dA + ((size_t)tl[t] * 16 + col) * a.ld + 16 * (b0 + nt) + 4 * q) = v;
                dy[t][b0 + nt] = v;
                // a tile = the 16 edges of ONE point (K = 16), an edge = a lane of the 16-lane row: the sum over the row IS the point's
                // dP for these columns - the column sums ec_pq_bwd_csr_kernel otherwise reads all of dA again for (67 MB per 128-wide unit)
                if (a.dP) {
                    f4 ps;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ps[r] = ecp_rowsum16(v[r]);
                    if (ok[t] && col == 0) *reinterpret_cast<f4*>(a.dP + (size_t)tl[t] * a.ldp + 16 * (b0 + nt) + 4 * q) = ps;
                }
            }
        }
    });
    if (!alive) {                                                     // loud: the layers that were not finished
#pragma unroll
        for (int t = 0; t < TPW; ++t)
            if (ok[t])
                for (int c = 4 * q; c < GT; c += 16)
                    *reinterpret_cast<f4*>(a.dA + ((size_t)tl[t] * 16 + col) * a.ld + c) = pf_splat(__builtin_nanf(""));
    }
    __syncthreads();
    if (threadIdx.x == 0) flag = atomicAdd(a.sync + 2, 1u) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (flag == 1) {
        for (int i = threadIdx.x; i < STAT_COPIES * 2 * STAT_W; i += ECP_T)
            if ((i % STAT_W) < 32 * NC) __hip_atomic_store(a.acc + i, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) {
            __hip_atomic_store(a.sync + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.sync + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// growth layer 0 has no growth input: only dA[:, 0:g] -> dy in place
__global__ __launch_bounds__(256) void ec_bwd0_kernel(float* dA, const float* Y, int ld, const float* aff, const float* coef, int g,
                                                      long long E, float slope) {
    const int g4 = g / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < E * g4; i += (long long)gridDim.x * 256) {
        const long long e = i / g4;
        const int c = (int)(i % g4) * 4;
        float* dp = dA + e * ld + c;
        const f4 d = *reinterpret_cast<const f4*>(dp);
        const f4 y = *reinterpret_cast<const f4*>(Y + e * ld + c);
        f4 o;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc = aff[c + w], z = fmaf(y[w], sc, aff[ld + c + w]);
            const float xh = (y[w] - aff[2 * ld + c + w]) * aff[3 * ld + c + w];
            const float dz = d[w] * (z > 0.f ? 1.f : slope);
            o[w] = sc * (dz - coef[c + w] - xh * coef[ld + c + w]);
        }
        *reinterpret_cast<f4*>(dp) = o;
    }
}

// ------------------------------------------------------------------------------------------------ dPQ
// dPQ [T, 2S]: P half written (sum over the K edges of a point), Q half accumulated with atomics (zeroed by the caller).
// columns 0..GT-1: growth layers (dY = the in-place converted dA), GT..S-1: conv_out (pooled: from dh / argmax).
struct EcPqBwdArgs {
    const float* dY; int ld;         // [E, ld = GT]
    const float* dh; const unsigned char* arg; const float* dyout; int pooled;
    const int* idx;
    int N, K, GT, odim, S;
    long long T;
    float* dPQ;                      // [T, 2S]
};
__global__ __launch_bounds__(256) void ec_pq_bwd_kernel(EcPqBwdArgs a) {
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < a.T * a.S; t += (long long)gridDim.x * 256) {
        const long long i = t / a.S;
        const int c = (int)(t % a.S);
        const long long base = (i / a.N) * a.N;
        float* dq = a.dPQ + a.S + c;
        float sum = 0.f;
        if (c < a.GT) {
            for (int k = 0; k < a.K; ++k) {
                const long long e = i * a.K + k;
                const float v = a.dY[e * a.ld + c];
                sum += v;
                atomicAdd(dq + (base + a.idx[e]) * 2 * a.S, v);
            }
        } else if (a.pooled) {
            const int co = c - a.GT;
            sum = a.dh[i * a.odim + co];
            const long long e = i * a.K + a.arg[i * a.odim + co];
            atomicAdd(dq + (base + a.idx[e]) * 2 * a.S, sum);
        } else {
            const int co = c - a.GT;
            for (int k = 0; k < a.K; ++k) {
                const long long e = i * a.K + k;
                const float v = a.dyout[e * a.odim + co];
                sum += v;
                atomicAdd(dq + (base + a.idx[e]) * 2 * a.S, v);
            }
        }
        a.dPQ[i * 2 * a.S + c] = sum;
    }
}

// The same without atomics, given the transposed neighbour lists (built once per step, pf_knn_csr): in-edges of point j are
// csr_edge[csr_off[j] .. csr_off[j+1]).  dQ[j] = sum over them (plain loads, coalesced over the channels), dP as above; the
// caller need not clear dPQ.
struct EcPqCsrArgs {
    EcPqBwdArgs b;
    const int* off; const int* edge;
    int p_done;                                      // the P half's growth columns were written by ec_bwdp_kernel
};
__global__ __launch_bounds__(256) void ec_pq_bwd_csr_kernel(EcPqCsrArgs a) {
    // one thread per (point, 4 channels): float4 loads, four edges in flight per accumulation step (a scalar thread per channel
    // with one load per dependent add ran at 2.2 TB/s with 86 % of its wave cycles waiting)
    const EcPqBwdArgs& b = a.b;
    const int S4 = b.S / 4;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < b.T * S4; t += (long long)gridDim.x * 256) {
        const int i = (int)(t / S4);
        const int c = (int)(t % S4) * 4;
        const int lo = a.off[i], hi = a.off[i + 1];
        f4 sum = pf_splat(0.f), q = pf_splat(0.f);
        if (c < b.GT || !b.pooled) {
            const float* src = c < b.GT ? b.dY + c : b.dyout + (c - b.GT);
            const size_t ld = c < b.GT ? (size_t)b.ld : (size_t)b.odim;
            const float* own = src + (size_t)i * b.K * ld;
            f4 s1 = pf_splat(0.f), s2 = pf_splat(0.f), s3 = pf_splat(0.f);
            const bool have_p = a.p_done && c < b.GT;
            int k = have_p ? b.K : 0;
            for (; k + 3 < b.K; k += 4) {
                const f4 v0 = *reinterpret_cast<const f4*>(own + (size_t)k * ld), v1 = *reinterpret_cast<const f4*>(own + (size_t)(k + 1) * ld);
                const f4 v2 = *reinterpret_cast<const f4*>(own + (size_t)(k + 2) * ld), v3 = *reinterpret_cast<const f4*>(own + (size_t)(k + 3) * ld);
                sum += v0; s1 += v1; s2 += v2; s3 += v3;
            }
            for (; k < b.K; ++k) sum += *reinterpret_cast<const f4*>(own + (size_t)k * ld);
            sum = (sum + s1) + (s2 + s3);
            f4 q1 = pf_splat(0.f), q2 = pf_splat(0.f), q3 = pf_splat(0.f);
            int n = lo;
            for (; n + 3 < hi; n += 4) {
                const int e0 = a.edge[n], e1 = a.edge[n + 1], e2 = a.edge[n + 2], e3 = a.edge[n + 3];
                const f4 v0 = *reinterpret_cast<const f4*>(src + (size_t)e0 * ld), v1 = *reinterpret_cast<const f4*>(src + (size_t)e1 * ld);
                const f4 v2 = *reinterpret_cast<const f4*>(src + (size_t)e2 * ld), v3 = *reinterpret_cast<const f4*>(src + (size_t)e3 * ld);
                q += v0; q1 += v1; q2 += v2; q3 += v3;
            }
            for (; n < hi; ++n) q += *reinterpret_cast<const f4*>(src + (size_t)a.edge[n] * ld);
            q = (q + q1) + (q2 + q3);
        } else {
            const int co = c - b.GT;
            sum = *reinterpret_cast<const f4*>(b.dh + (size_t)i * b.odim + co);
            // the pooled gradient reaches this point through the edges whose argmax it is: four edges in flight (edge id -> its
            // point's argmax word and gradient row: two dependent loads per edge - one edge at a time was the kernel's longest chain)
            auto take = [&](int e, unsigned g4, const f4& dv) {
                const int k = e - (e / b.K) * b.K;
                if ((int)(g4 & 255u) == k) q.x += dv.x;
                if ((int)((g4 >> 8) & 255u) == k) q.y += dv.y;
                if ((int)((g4 >> 16) & 255u) == k) q.z += dv.z;
                if ((int)(g4 >> 24) == k) q.w += dv.w;
            };
            int n = lo;
            for (; n + 3 < hi; n += 4) {
                int e[4];
                unsigned g4[4];
                f4 dv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) e[u] = a.edge[n + u];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const size_t ii = (size_t)(e[u] / b.K);
                    g4[u] = *reinterpret_cast<const unsigned*>(b.arg + ii * b.odim + co);
                    dv[u] = *reinterpret_cast<const f4*>(b.dh + ii * b.odim + co);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) take(e[u], g4[u], dv[u]);                  // (in list order: the same sums as before)
            }
            for (; n < hi; ++n) {
                const int e = a.edge[n];
                const size_t ii = (size_t)(e / b.K);
                take(e, *reinterpret_cast<const unsigned*>(b.arg + ii * b.odim + co), *reinterpret_cast<const f4*>(b.dh + ii * b.odim + co));
            }
        }
        if (!(a.p_done && c < b.GT)) *reinterpret_cast<f4*>(b.dPQ + (size_t)i * 2 * b.S + c) = sum;
        *reinterpret_cast<f4*>(b.dPQ + (size_t)i * 2 * b.S + b.S + c) = q;
    }
}

// transposed neighbour lists of idx [T, K] (batch-local indices, N points per sample): count -> scan -> fill
__global__ __launch_bounds__(256) void csr_count_kernel(const int* idx, int N, int K, long long E, int* cnt) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < E; e += (long long)gridDim.x * 256) {
        const long long i = e / K;
        atomicAdd(cnt + (i / N) * N + idx[e], 1);
    }
}
// exclusive scan of cnt[T] -> off[T+1] by ONE workgroup of 1024 threads (T <= a few 100 k); cnt is left as the running fill cursor
__global__ __launch_bounds__(1024) void csr_scan_kernel(int* cnt, int T, int* off) {
    __shared__ int part[1024];
    const int per = (T + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(T, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += cnt[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (int i = lo; i < hi; ++i) { const int c = cnt[i]; off[i] = run; cnt[i] = run; run += c; }
    if (threadIdx.x == 1023) off[T] = part[1023];
}
__global__ __launch_bounds__(256) void csr_fill_kernel(const int* idx, int N, int K, long long E, int* cursor, int* edge) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < E; e += (long long)gridDim.x * 256) {
        const long long i = e / K;
        edge[atomicAdd(cursor + (i / N) * N + idx[e], 1)] = (int)e;
    }
}
// Two lists from one pass over idx [T, K]: all K columns (cnt / edge ids i K + k) and the first K2 columns (cnt2 / edge ids
// i K2 + k: what pf_knn_csr gives for idx[:, :K2] stored contiguously) - the training step needs both (K = 16: feature units,
// K2 = 8: the interpolation unit), and each launch here is a few microseconds of work behind a launch of its own.
__global__ __launch_bounds__(256) void csr_count2_kernel(const int* idx, int N, int K, int K2, long long E, int* cnt, int* cnt2) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < E; e += (long long)gridDim.x * 256) {
        const long long i = e / K;
        const long long j = (i / N) * N + idx[e];
        atomicAdd(cnt + j, 1);
        if ((int)(e - i * K) < K2) atomicAdd(cnt2 + j, 1);
    }
}
__global__ __launch_bounds__(1024) void csr_scan2_kernel(int* cnt, int T, int Tpad, int* off, int* off2) {
    __shared__ int part[1024];
    if (blockIdx.x) { cnt += Tpad; off = off2; }
    const int per = (T + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = min(T, lo + per);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += cnt[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (int i = lo; i < hi; ++i) { const int c = cnt[i]; off[i] = run; cnt[i] = run; run += c; }
    if (threadIdx.x == 1023) off[T] = part[1023];
}
__global__ __launch_bounds__(256) void csr_fill2_kernel(const int* idx, int N, int K, int K2, long long E, int* cursor, int* cursor2,
                                                        int* edge, int* edge2) {
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < E; e += (long long)gridDim.x * 256) {
        const long long i = e / K;
        const long long j = (i / N) * N + idx[e];
        const int k = (int)(e - i * K);
        edge[atomicAdd(cursor + j, 1)] = (int)e;
        if (k < K2) edge2[atomicAdd(cursor2 + j, 1)] = (int)(i * K2 + k);
    }
}
// the fill above hands out a list's slots in arrival order: sort every list (edge ids ascending) so that whatever is summed over it
// - the dQ gather of the EdgeConv backward, the latent's gradient, the Chamfer gradient - adds in ONE order, run after run
__global__ __launch_bounds__(256) void csr_sort_kernel(const int* __restrict__ off, int* __restrict__ edge, int T) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= T) return;
    const int lo = off[j], hi = off[j + 1];
    for (int a = lo + 1; a < hi; ++a) {
        const int v = edge[a];
        int b = a - 1;
        while (b >= lo && edge[b] > v) { edge[b + 1] = edge[b]; --b; }
        edge[b + 1] = v;
    }
}

// ------------------------------------------------------------------------------------------------ growth-weight gradients
// part[chunk][c][u] = sum over the chunk's edges of dYfull[e, c] * lrelu(bn(Y[e, u])), c < S (growth layers then conv_out),
// u < GT.  K dimension = edges: both operands are channel-fast in memory, so a block of 32 edges is staged through LDS
// (coalesced float4 loads, the activation / the pooled gradient formed once on the way) and the MFMA operands are read
// from there edge-major.  blockIdx.y = 0: the conv_out rows; 1: the growth rows, whose output is block lower triangular
// (layer t sees columns u < g t only) - structurally empty 16 x 16 tiles are skipped.
struct EcDwArgs {
    const float* dY; const float* Y; int ld;
    const float* aff;
    const float* dh; const unsigned char* arg; const float* dyout; int pooled;
    int g, GT, odim, S, K;
    long long E; int chunk;
    float slope;
    float* part;
    float* bpart;                    // [nchunk][S]: column sums of dYfull over the chunk (the conv bias gradients)
};
#ifndef PF_DW_EB
#define PF_DW_EB 32
#endif
constexpr int DW_EB = PF_DW_EB;                     // edges per staged block

// NWV waves per workgroup share one staged block: 8 waves (4 per SIMD with two workgroups per CU) keep the matrix pipe fed while
// other waves sit in the load -> LDS -> barrier phase (PMC at 4 waves: MFMA busy 24 %, 57 % of the wave cycles waiting)
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void ec_dw_kernel(EcDwArgs a) {
    constexpr int NTH = 64 * NWV, SL = 32 / NWV;         // row-tile slots per wave: 8 at 4 waves, 4 at 8
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, q = lane >> 4;
    const bool outrows = blockIdx.y == 0;
    const int RA = outrows ? a.odim : a.GT;
    const int lda = RA + 16, ldb = a.GT + 16;
    float* As = lds;
    float* Bs = lds + DW_EB * lda;
    // tile (rt, ct) of the output block -> wave: WC = min(4, NT) waves side by side along the columns, 4 / WC groups of
    // them along the rows; a wave owns column tiles ct_j = w % WC + WC j (j < 2) and row tiles rt_s = w / WC + (4 / WC) s
    // (s < 8): per K-step it reads <= 2 B values and <= 8 A values from LDS for <= 16 MFMAs
    const int NT = a.GT / 16, NRT = RA / 16;
    const int WC = NT < 4 ? NT : 4, rstep = NWV / WC, rbase = wave / WC;
    int ctj[2];
    bool cval[2];
#pragma unroll
    for (int jc = 0; jc < 2; ++jc) { ctj[jc] = wave % WC + WC * jc; cval[jc] = ctj[jc] < NT; }
    bool val[SL][2];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        const int rt = rbase + rstep * s;
        int ntn = NT;
        if (!outrows) {
            const int last = rt * 16 + 15;
            ntn = ((last / a.g) * a.g + 15) / 16;                 // column tiles u < g * (layer of the tile's last row)
        }
#pragma unroll
        for (int jc = 0; jc < 2; ++jc) val[s][jc] = rt < NRT && cval[jc] && ctj[jc] < ntn;
    }
    f4 acc[SL][2];
#pragma unroll
    for (int s = 0; s < SL; ++s) { acc[s][0] = pf_splat(0.f); acc[s][1] = pf_splat(0.f); }
    const int e_lo = blockIdx.x * a.chunk, e_hi = min((int)a.E, e_lo + a.chunk);      // E < 2^30: 32-bit edge indices
    const int ra4 = RA / 4, gt4 = a.GT / 4;
    float bsum = 0.f;
    // float4 staging units, thread t owns units t, t + 256, ... (fixed (edge, column) per unit); the next block's units are
    // fetched into registers while the current block is multiplied
    constexpr int UN = DW_EB * 32 / NTH;                 // DW_EB * 128 / 4 / threads
    int elA[UN], cA[UN], elB[UN], cB[UN];
#pragma unroll
    for (int n = 0; n < UN; ++n) {
        const int k = threadIdx.x + NTH * n;
        elA[n] = k / ra4; cA[n] = (k - elA[n] * ra4) * 4;
        elB[n] = k / gt4; cB[n] = (k - elB[n] * gt4) * 4;
    }
    f4 ra[UN], rbv[UN];
    auto fetch = [&](int eb) {
#pragma unroll
        for (int n = 0; n < UN; ++n) {
            f4 v = pf_splat(0.f);
            const int e = eb + elA[n], c = cA[n];
            if (elA[n] < DW_EB && e < e_hi) {
                if (!outrows) v = *reinterpret_cast<const f4*>(a.dY + (size_t)e * a.ld + c);
                else if (a.pooled) {
                    const int ii = e >> 4, k = e & 15;                          // pooled units have K = 16
                    const f4 dv = *reinterpret_cast<const f4*>(a.dh + (size_t)ii * a.odim + c);
                    const unsigned g4 = *reinterpret_cast<const unsigned*>(a.arg + (size_t)ii * a.odim + c);
                    v.x = (int)(g4 & 255u) == k ? dv.x : 0.f; v.y = (int)((g4 >> 8) & 255u) == k ? dv.y : 0.f;
                    v.z = (int)((g4 >> 16) & 255u) == k ? dv.z : 0.f; v.w = (int)(g4 >> 24) == k ? dv.w : 0.f;
                } else v = *reinterpret_cast<const f4*>(a.dyout + (size_t)e * a.odim + c);
            }
            ra[n] = v;
        }
#pragma unroll
        for (int n = 0; n < UN; ++n) {
            f4 v = pf_splat(0.f);
            const int e = eb + elB[n], c = cB[n];
            if (elB[n] < DW_EB && e < e_hi)
                v = lrelu4(*reinterpret_cast<const f4*>(a.Y + (size_t)e * a.ld + c) * *reinterpret_cast<const f4*>(a.aff + c) +
                           *reinterpret_cast<const f4*>(a.aff + a.ld + c), a.slope);
            rbv[n] = v;
        }
    };
    fetch(e_lo);
    for (int eb = e_lo; eb < e_hi; eb += DW_EB) {
        __syncthreads();
#pragma unroll
        for (int n = 0; n < UN; ++n) {
            if (elA[n] < DW_EB) *reinterpret_cast<f4*>(As + elA[n] * lda + cA[n]) = ra[n];
            if (elB[n] < DW_EB) *reinterpret_cast<f4*>(Bs + elB[n] * ldb + cB[n]) = rbv[n];
        }
        __syncthreads();
        if (eb + DW_EB < e_hi) fetch(eb + DW_EB);
        if (threadIdx.x < RA)
#pragma unroll 8
            for (int el = 0; el < DW_EB; ++el) bsum += As[el * lda + threadIdx.x];
#pragma unroll
        for (int ks = 0; ks < DW_EB / 4; ++ks) {
            const float* ar = As + (4 * ks + q) * lda + row + rbase * 16;
            const float* br = Bs + (4 * ks + q) * ldb + row;
            const float b0 = cval[0] ? br[ctj[0] * 16] : 0.f, b1 = cval[1] ? br[ctj[1] * 16] : 0.f;
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                if (val[s][0] || val[s][1]) {
                    const float av = ar[rstep * s * 16];
                    if (val[s][0]) acc[s][0] = pf_mfma(av, b0, acc[s][0]);
                    if (val[s][1]) acc[s][1] = pf_mfma(av, b1, acc[s][1]);
                }
            }
        }
    }
    if (threadIdx.x < RA) a.bpart[(size_t)blockIdx.x * a.S + (outrows ? a.GT : 0) + threadIdx.x] = bsum;
    float* out = a.part + ((size_t)blockIdx.x * a.S + (outrows ? a.GT : 0)) * a.GT;
#pragma unroll
    for (int s = 0; s < SL; ++s)
#pragma unroll
        for (int jc = 0; jc < 2; ++jc)
            if (val[s][jc])
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[(size_t)((rbase + rstep * s) * 16 + 4 * q + r) * a.GT + ctj[jc] * 16 + row] = acc[s][jc][r];
}

// ------------------------------------------------------------------------------------------------ growth-weight gradients, no LDS
// The same partial sums with the operands loaded straight into the 32x32x2 f32 MFMA layout: a K-step is two edges, lane l holds
// A[channel l & 31][edge l >> 5] and B[edge l >> 5][channel l & 31] - one 4-byte load each, 128 contiguous bytes per half wave.
// No LDS image, no barriers: every wave streams its own operands four K-steps ahead and owns a strip of <= 4 output tiles
// (64 accumulator registers).  Jobs (one wave each): conv_out row strips with all GT / 32 column tiles, growth row strips with
// the column tiles their layers can see.  The staged kernel above spent 54 % of its wave cycles waiting with the matrix pipe
// 30 % busy; it stays for shapes that are not multiples of 32.
struct EcDw2Job { int out, rt, nct; };
struct EcDw2Args {
    EcDwArgs d;
    EcDw2Job job[8];
    int njob;
};
typedef float f16v __attribute__((ext_vector_type(16)));
#ifndef PF_DW2_D
#define PF_DW2_D 8
#endif
constexpr int DW2_D = PF_DW2_D;                               // K-steps (of two edges) per software-pipeline stage

#ifdef PF_EC_DW_F32                                    // the f32-product form of the no-LDS kernel: A/B builds only
__global__ __launch_bounds__(512) void ec_dw2_kernel(EcDw2Args g2) {
    const EcDwArgs& a = g2.d;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
    if (wave >= g2.njob) return;
    const EcDw2Job jb = g2.job[wave];
    const int e_lo = blockIdx.x * a.chunk, e_hi = min((int)a.E, e_lo + a.chunk);      // multiples of 16
    const int crow = jb.rt * 32 + col;                  // this lane's A channel
    const bool outj = jb.out != 0;
    float sc[4], sh[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = t * 32 + col;
        sc[t] = t < jb.nct ? a.aff[c] : 0.f;
        sh[t] = t < jb.nct ? a.aff[a.ld + c] : 0.f;
    }
    f16v acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float asum = 0.f;
    // one stage = DW2_D K-steps = 8 edges (inside one point: K = 16 edges per point for pooled units)
    float an[DW2_D], bn[DW2_D][4];
    auto fetch = [&](int e0) {
        if (outj && a.pooled) {
            const int ii = e0 >> 4;
            const float dv = a.dh[(size_t)ii * a.odim + crow];
            const int kk = a.arg[(size_t)ii * a.odim + crow];
#pragma unroll
            for (int k = 0; k < DW2_D; ++k) an[k] = (((e0 + 2 * k + h) & 15) == kk) ? dv : 0.f;
        } else {
#pragma unroll
            for (int k = 0; k < DW2_D; ++k) {
                const size_t e = (size_t)(e0 + 2 * k + h);
                an[k] = outj ? a.dyout[e * a.odim + crow] : a.dY[e * a.ld + crow];
            }
        }
#pragma unroll
        for (int k = 0; k < DW2_D; ++k) {
            const float* yr = a.Y + (size_t)(e0 + 2 * k + h) * a.ld + col;
#pragma unroll
            for (int t = 0; t < 4; ++t) bn[k][t] = t < jb.nct ? yr[t * 32] : 0.f;
        }
    };
    if (e_lo < e_hi) fetch(e_lo);
    for (int e0 = e_lo; e0 < e_hi; e0 += 2 * DW2_D) {
        float ac[DW2_D], bc[DW2_D][4];
#pragma unroll
        for (int k = 0; k < DW2_D; ++k) {
            ac[k] = an[k];
#pragma unroll
            for (int t = 0; t < 4; ++t) bc[k][t] = bn[k][t];
        }
        if (e0 + 2 * DW2_D < e_hi) fetch(e0 + 2 * DW2_D);
#pragma unroll
        for (int k = 0; k < DW2_D; ++k) {
            asum += ac[k];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t < jb.nct) {
                    const float z = fmaf(bc[k][t], sc[t], sh[t]);
                    const float bv = fmaxf(z, z * a.slope);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[k], bv, acc[t], 0, 0, 0);
                }
            }
        }
    }
    const int rowbase = outj ? a.GT : 0;
    asum += __shfl_xor(asum, 32);
    if (h == 0) a.bpart[(size_t)blockIdx.x * a.S + rowbase + crow] = asum;
    float* out = a.part + ((size_t)blockIdx.x * a.S + rowbase + jb.rt * 32) * a.GT;
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (t < jb.nct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ri = (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)ri * a.GT + t * 32 + col] = acc[t][r];
            }
}
#endif

// The same jobs on the bf16 matrix pipe: x = hi + mid with hi = the top 16 bits of x and mid = bf16(x - hi) (16 mantissa bits
// together, fp32 exponent range - gradients of 1e-7 keep their digits, which fp16 halves would not), three
// v_mfma_f32_32x32x16_bf16 per tile and 16 edges (hi hi + hi mid + mid hi, fp32 accumulate) instead of eight f32 MFMAs: 96 against
// 512 matrix-pipe cycles.  Lane l holds its channel for the 8 edges 8 (l >> 5) + j of a 16-edge step.  In the no-LDS structure
// the f32 pipe WAS the limit (the busiest SIMD of a workgroup owns 7 of its 22 tiles: 448 cycles per two edges); in the staged
// kernel the same change bought nothing.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
struct Bf2 { bf8 hi, mid; };
__device__ __forceinline__ Bf2 dw3_split(const float (&x)[8]) {
    unsigned hw[4], mw[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned b0 = __float_as_uint(x[2 * p]), b1 = __float_as_uint(x[2 * p + 1]);
        const unsigned h0 = b0 & 0xffff0000u, h1 = b1 & 0xffff0000u;
        const unsigned m0 = __float_as_uint(x[2 * p] - __uint_as_float(h0)), m1 = __float_as_uint(x[2 * p + 1] - __uint_as_float(h1));
        hw[p] = (h0 >> 16) | h1;
        mw[p] = (m0 >> 16) | (m1 & 0xffff0000u);
    }
    Bf2 r;
    r.hi = __builtin_bit_cast(bf8, *reinterpret_cast<const uint4*>(hw));
    r.mid = __builtin_bit_cast(bf8, *reinterpret_cast<const uint4*>(mw));
    return r;
}

#ifndef PF_EC_DW3_DEPTH
#define PF_EC_DW3_DEPTH 1
#endif
__global__ __launch_bounds__(512) void ec_dw3_kernel(EcDw2Args g2) {
    const EcDwArgs& a = g2.d;
    // the wave index through readfirstlane: its job (row strip, column tile count) is then wave-uniform to the compiler - scalar
    // loads of the job, scalar branches around the per-tile MFMAs instead of exec masks
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
    if (wave >= g2.njob) return;
    const EcDw2Job jb = g2.job[wave];
    const int e_lo = blockIdx.x * a.chunk, e_hi = min((int)a.E, e_lo + a.chunk);      // multiples of 16
    const int crow = jb.rt * 32 + col;
    const bool outj = jb.out != 0;
    float sc[4], sh[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = t * 32 + col;
        sc[t] = t < jb.nct ? a.aff[c] : 0.f;
        sh[t] = t < jb.nct ? a.aff[a.ld + c] : 0.f;
    }
    f16v acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float asum = 0.f;
    // PF_EC_DW3_DEPTH 16-edge steps' operands in flight.  Measured (round 5, same box, whole step): depth 1 4.566 ms, 2 4.571,
    // 3 4.58 - 4.79: the kernel does not wait for its loads
    float an[PF_EC_DW3_DEPTH][8], bn[PF_EC_DW3_DEPTH][4][8];
    auto fetch = [&](int e0, float (&an_)[8], float (&bn_)[4][8]) {
        if (outj && a.pooled) {
            const int ii = e0 >> 4;
            const float dv = a.dh[(size_t)ii * a.odim + crow];
            const int kk = a.arg[(size_t)ii * a.odim + crow];
#pragma unroll
            for (int j = 0; j < 8; ++j) an_[j] = (8 * h + j == kk) ? dv : 0.f;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const size_t e = (size_t)(e0 + 8 * h + j);
                an_[j] = outj ? a.dyout[e * a.odim + crow] : a.dY[e * a.ld + crow];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* yr = a.Y + (size_t)(e0 + 8 * h + j) * a.ld + col;
#pragma unroll
            for (int t = 0; t < 4; ++t) bn_[t][j] = t < jb.nct ? yr[t * 32] : 0.f;
        }
    };
#pragma unroll
    for (int u = 0; u < PF_EC_DW3_DEPTH; ++u)
        if (e_lo + 16 * u < e_hi) fetch(e_lo + 16 * u, an[u], bn[u]);
    for (int e0 = e_lo; e0 < e_hi; e0 += 16 * PF_EC_DW3_DEPTH)
#pragma unroll
    for (int u = 0; u < PF_EC_DW3_DEPTH; ++u) {
        if (e0 + 16 * u >= e_hi) break;
        float ac[8], bc[4][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ac[j] = an[u][j];
            asum += an[u][j];
#pragma unroll
            for (int t = 0; t < 4; ++t) bc[t][j] = bn[u][t][j];
        }
        if (e0 + 16 * (u + PF_EC_DW3_DEPTH) < e_hi) fetch(e0 + 16 * (u + PF_EC_DW3_DEPTH), an[u], bn[u]);
        const Bf2 A = dw3_split(ac);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < jb.nct) {
                float bv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float z = fmaf(bc[t][j], sc[t], sh[t]);
                    bv[j] = fmaxf(z, z * a.slope);
                }
                const Bf2 B = dw3_split(bv);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.mid, B.hi, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.hi, B.mid, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.hi, B.hi, acc[t], 0, 0, 0);
            }
        }
    }
    const int rowbase = outj ? a.GT : 0;
    asum += __shfl_xor(asum, 32);
    if (h == 0) a.bpart[(size_t)blockIdx.x * a.S + rowbase + crow] = asum;
    float* out = a.part + ((size_t)blockIdx.x * a.S + rowbase + jb.rt * 32) * a.GT;
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (t < jb.nct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ri = (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)ri * a.GT + t * 32 + col] = acc[t][r];
            }
}

// Round 5: TWO row strips per wave.  ec_dw3_kernel ran at the VALU rate: every wave loads its own copy of the B operand
// act(Y) - BatchNorm + LeakyReLU + the bf16 split of 32 values per lane and 16-edge step - although it is the same for all row
// strips; staging it once per workgroup through LDS (ec_dw4_kernel below) lost to its own barriers.  Here a wave simply owns
// two strips (jobs 2 w and 2 w + 1; the launcher orders the jobs so that a pair's tile counts add up evenly): the B tile is
// converted once for both, the conversions per MFMA fall from 5/4 to 6/8, eight accumulator tiles live in the (unified)
// register file.  Same operands, same split, same product order per (strip, tile): the partials are bit for bit those of
// ec_dw3_kernel.
// MEASURED NEGATIVE (round 5, same box): the step 4.68 against 4.53 ms with ec_dw3_kernel - 343 registers leave one wave per
// SIMD, and the strip's dependent MFMA chains (3 products per tile into one accumulator) then have nothing to interleave with.
// Kept for the A/B only (-DPF_EC_DW5=1).
#ifndef PF_EC_DW5
#define PF_EC_DW5 0
#endif
__global__ __launch_bounds__(256) void ec_dw5_kernel(EcDw2Args g2) {
    const EcDwArgs& a = g2.d;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
    const int j0 = 2 * wave;
    if (j0 >= g2.njob) return;
    const bool two = j0 + 1 < g2.njob;
    const EcDw2Job jb[2] = {g2.job[j0], g2.job[two ? j0 + 1 : j0]};
    const int nct[2] = {jb[0].nct, two ? jb[1].nct : 0};
    const int nctm = nct[0] > nct[1] ? nct[0] : nct[1];
    const int e_lo = blockIdx.x * a.chunk, e_hi = min((int)a.E, e_lo + a.chunk);      // multiples of 16
    const int crow[2] = {jb[0].rt * 32 + col, jb[1].rt * 32 + col};
    const bool outj[2] = {jb[0].out != 0, jb[1].out != 0};
    float sc[4], sh[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int c = t * 32 + col;
        sc[t] = t < nctm ? a.aff[c] : 0.f;
        sh[t] = t < nctm ? a.aff[a.ld + c] : 0.f;
    }
    f16v acc[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][t][r] = 0.f;
    float asum[2] = {0.f, 0.f};
    float an[2][8], bn[4][8];                            // the next 16-edge step's operands, in flight
    auto fetch = [&](int e0) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 1 && !two) break;
            if (outj[s] && a.pooled) {
                const int ii = e0 >> 4;
                const float dv = a.dh[(size_t)ii * a.odim + crow[s]];
                const int kk = a.arg[(size_t)ii * a.odim + crow[s]];
#pragma unroll
                for (int j = 0; j < 8; ++j) an[s][j] = (8 * h + j == kk) ? dv : 0.f;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const size_t e = (size_t)(e0 + 8 * h + j);
                    an[s][j] = outj[s] ? a.dyout[e * a.odim + crow[s]] : a.dY[e * a.ld + crow[s]];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* yr = a.Y + (size_t)(e0 + 8 * h + j) * a.ld + col;
#pragma unroll
            for (int t = 0; t < 4; ++t) bn[t][j] = t < nctm ? yr[t * 32] : 0.f;
        }
    };
#pragma unroll
    for (int j = 0; j < 8; ++j) an[1][j] = 0.f;
    if (e_lo < e_hi) fetch(e_lo);
    for (int e0 = e_lo; e0 < e_hi; e0 += 16) {
        float ac[2][8], bc[4][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ac[0][j] = an[0][j]; ac[1][j] = an[1][j];
            asum[0] += an[0][j]; asum[1] += an[1][j];
#pragma unroll
            for (int t = 0; t < 4; ++t) bc[t][j] = bn[t][j];
        }
        if (e0 + 16 < e_hi) fetch(e0 + 16);
        const Bf2 A0 = dw3_split(ac[0]), A1 = dw3_split(ac[1]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < nctm) {
                float bv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float z = fmaf(bc[t][j], sc[t], sh[t]);
                    bv[j] = fmaxf(z, z * a.slope);
                }
                const Bf2 B = dw3_split(bv);
                if (t < nct[0]) {
                    acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0.mid, B.hi, acc[0][t], 0, 0, 0);
                    acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0.hi, B.mid, acc[0][t], 0, 0, 0);
                    acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0.hi, B.hi, acc[0][t], 0, 0, 0);
                }
                if (t < nct[1]) {
                    acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1.mid, B.hi, acc[1][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1.hi, B.mid, acc[1][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1.hi, B.hi, acc[1][t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (s == 1 && !two) break;
        const int rowbase = outj[s] ? a.GT : 0;
        float as = asum[s];
        as += __shfl_xor(as, 32);
        if (h == 0) a.bpart[(size_t)blockIdx.x * a.S + rowbase + crow[s]] = as;
        float* out = a.part + ((size_t)blockIdx.x * a.S + rowbase + jb[s].rt * 32) * a.GT;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (t < nct[s])
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ri = (r & 3) + 8 * (r >> 2) + 4 * h;
                    out[(size_t)ri * a.GT + t * 32 + col] = acc[s][t][r];
                }
    }
}

// Round 5: the same jobs with the B operand - act(Y), the SAME for all eight row strips - staged ONCE per workgroup.
// ec_dw3_kernel lets every wave load and convert its own copy: 32 of its 40 four-byte loads per 16-edge step and, worse, 224 of
// its 264 vector instructions per step (BatchNorm + LeakyReLU + the bf16 split of 32 values per lane) were identical work in
// eight waves - the kernel ran at the VALU rate (~30 us of its 57 per 128-channel unit), not at the matrix or memory rate.
// Here a stage = 64 edges: every thread fetches two (column, 8-edge) runs of Y, applies the activation, splits them and writes
// the two ready-made B fragments (hi | mid, 16 bytes each) into LDS in MFMA operand order; a wave then reads its nct tiles'
// fragments with ds_read_b128 and keeps only its own A operand (dy rows) private.  Two LDS buffers, one barrier per stage, the
// next stage's Y in registers during the MFMAs.  Same values, same split, same product order per tile: partials are bit for
// bit those of ec_dw3_kernel.
// MEASURED NEGATIVE (round 5, same box, rocprofv3 per-step sums over the 7 units): 513 us against ec_dw3_kernel's 400 us, the
// step 4.60 against 4.51 ms - the per-stage barrier and the LDS round trip cost more than the eight-fold conversion saves; the
// kernel is kept for the A/B only (-DPF_EC_DW4=1).
#ifndef PF_EC_DW4
#define PF_EC_DW4 0
#endif
constexpr int DW4_SE = 64;                                    // edges per stage
#if PF_EC_DW4
__global__ __launch_bounds__(512) void ec_dw4_kernel(EcDw2Args g2) {
    const EcDwArgs& a = g2.d;
    extern __shared__ __attribute__((aligned(16))) uint4 dw4_lds[];        // [2 buffers][4 steps][NCT tiles][hi | mid][64 lanes]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
    const int NCT = a.GT / 32;
    const bool has_job = wave < g2.njob;
    const EcDw2Job jb = g2.job[has_job ? wave : 0];
    const int nct = has_job ? jb.nct : 0;
    const int e_lo = blockIdx.x * a.chunk, e_hi = min((int)a.E, e_lo + a.chunk);      // multiples of 16
    const int crow = jb.rt * 32 + col;
    const bool outj = jb.out != 0;
    // ---- staging role: unit u = (edge group of 8, column); this thread's units u = tid, tid + 512
    const int nunit = 8 * a.GT;                               // per stage
    int ucol[2], ueg[2];
    bool uok[2];
    float usc[2], ush[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int u = threadIdx.x + 512 * i;
        uok[i] = u < nunit;
        ueg[i] = uok[i] ? u / a.GT : 0;
        ucol[i] = uok[i] ? u - ueg[i] * a.GT : 0;
        usc[i] = a.aff[ucol[i]];
        ush[i] = a.aff[a.ld + ucol[i]];
    }
    float yb[2][8];
    auto fetch_b = [&](int e0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = e0 + 8 * ueg[i] + j;
                const bool ok = uok[i] && e < e_hi;
                const float v = a.Y[(size_t)(ok ? e : e_lo) * a.ld + ucol[i]];
                yb[i][j] = ok ? v : 0.f;
            }
    };
    auto stash_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!uok[i]) continue;
            float bv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float z = fmaf(yb[i][j], usc[i], ush[i]);
                bv[j] = fmaxf(z, z * a.slope);
            }
            const Bf2 B = dw3_split(bv);
            const int step = ueg[i] >> 1, hh = ueg[i] & 1, t = ucol[i] >> 5, ln = (ucol[i] & 31) + 32 * hh;
            uint4* dst = dw4_lds + (((size_t)buf * 4 + step) * NCT + t) * 2 * 64 + ln;
            dst[0] = __builtin_bit_cast(uint4, B.hi);
            dst[64] = __builtin_bit_cast(uint4, B.mid);
        }
    };
    f16v acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float asum = 0.f;
    float an[8];
    auto fetch_a = [&](int e0) {
        if (outj && a.pooled) {
            const int ii = e0 >> 4;
            const float dv = a.dh[(size_t)ii * a.odim + crow];
            const int kk = a.arg[(size_t)ii * a.odim + crow];
#pragma unroll
            for (int j = 0; j < 8; ++j) an[j] = (8 * h + j == kk) ? dv : 0.f;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const size_t e = (size_t)(e0 + 8 * h + j);
                an[j] = outj ? a.dyout[e * a.odim + crow] : a.dY[e * a.ld + crow];
            }
        }
    };
    if (e_lo < e_hi) {
        fetch_b(e_lo);
        if (has_job) fetch_a(e_lo);
        stash_b(0);
    }
    __syncthreads();
    int buf = 0;
    for (int s0 = e_lo; s0 < e_hi; s0 += DW4_SE) {
        const bool more = s0 + DW4_SE < e_hi;
        if (more) fetch_b(s0 + DW4_SE);
        if (has_job) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int e0 = s0 + 16 * st;
                if (e0 >= e_hi) break;
                float ac[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { ac[j] = an[j]; asum += an[j]; }
                if (e0 + 16 < e_hi) fetch_a(e0 + 16);
                const Bf2 A = dw3_split(ac);
                const uint4* src = dw4_lds + (((size_t)buf * 4 + st) * NCT) * 2 * 64 + lane;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (t < nct) {
                        const bf8 bh = __builtin_bit_cast(bf8, src[(t * 2 + 0) * 64]), bm = __builtin_bit_cast(bf8, src[(t * 2 + 1) * 64]);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.mid, bh, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.hi, bm, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.hi, bh, acc[t], 0, 0, 0);
                    }
                }
            }
        }
        if (more) stash_b(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (!has_job) return;
    const int rowbase = outj ? a.GT : 0;
    asum += __shfl_xor(asum, 32);
    if (h == 0) a.bpart[(size_t)blockIdx.x * a.S + rowbase + crow] = asum;
    float* out = a.part + ((size_t)blockIdx.x * a.S + rowbase + jb.rt * 32) * a.GT;
#pragma unroll
    for (int t = 0; t < 4; ++t)
        if (t < nct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ri = (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)ri * a.GT + t * 32 + col] = acc[t][r];
            }
}
#endif

// dWpq [R, C] = dPQ^T x for a unit whose input has C <= 4 channels (the first unit: the coordinates): one thread per output row
// r, a chunk of points per workgroup, partial sums as split-K slabs for the reduction kernel of the point GEMMs.  The general
// GEMM takes its scalar staging path for these shapes (C is not a multiple of 4): 69 us for 2.4 M products.
__global__ __launch_bounds__(256) void ec_dwpq_small_kernel(const float* __restrict__ dPQ, const float* __restrict__ x, int R, int C,
                                                           int T, int chunk, float* __restrict__ slabs) {
    const int r = blockIdx.y * 256 + threadIdx.x;
    const int t_lo = blockIdx.x * chunk, t_hi = min(T, t_lo + chunk);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (r < R) {
        int t = t_lo;
        for (; t + 7 < t_hi; t += 8) {                                // eight loads in flight per thread
            float g[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = dPQ[(size_t)(t + k) * R + r];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float* xp = x + (size_t)(t + k) * C;            // wave-uniform: scalar loads
                a0 = fmaf(g[k], xp[0], a0);
                if (C > 1) a1 = fmaf(g[k], xp[1], a1);
                if (C > 2) a2 = fmaf(g[k], xp[2], a2);
                if (C > 3) a3 = fmaf(g[k], xp[3], a3);
            }
        }
        for (; t < t_hi; ++t) {
            const float g = dPQ[(size_t)t * R + r];
            const float* xp = x + (size_t)t * C;
            a0 = fmaf(g, xp[0], a0);
            if (C > 1) a1 = fmaf(g, xp[1], a1);
            if (C > 2) a2 = fmaf(g, xp[2], a2);
            if (C > 3) a3 = fmaf(g, xp[3], a3);
        }
    }
    if (r < R) {
        float* o = slabs + ((size_t)blockIdx.x * R + r) * C;
        o[0] = a0;
        if (C > 1) o[1] = a1;
        if (C > 2) o[2] = a2;
        if (C > 3) o[3] = a3;
    }
}

// ------------------------------------------------------------------------------------------------ weight folding / un-folding
struct EcConvs {
    const float* W[9]; const float* bias[9];
    float* dW[9]; float* dbias[9];
    int rows[9], rowoff[10], width[9];      // conv t: [rows, width = 3C + g t]; rowoff: first row in the S-row stacking
    int nconvs, C, S, GT;
};
// Wpq [2S, C], bpq [2S] = (bias | 0)
__global__ __launch_bounds__(256) void ec_fold_kernel(EcConvs cv, float* Wpq, float* bpq) {
    const int total = cv.S * cv.C;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int srow = i / cv.C, col = i % cv.C;
        int t = 0;
        while (t + 1 < cv.nconvs && srow >= cv.rowoff[t + 1]) ++t;
        const float* w = cv.W[t] + (size_t)(srow - cv.rowoff[t]) * cv.width[t];
        Wpq[(size_t)srow * cv.C + col] = w[col] - w[2 * cv.C + col];
        Wpq[(size_t)(cv.S + srow) * cv.C + col] = w[cv.C + col] + w[2 * cv.C + col];
        if (col == 0) { bpq[srow] = cv.bias[t][srow - cv.rowoff[t]]; bpq[cv.S + srow] = 0.f; }
    }
}
// The same fold for several units in ONE launch (pf_ec_train_fold_batch): Wpq / bpq depend on parameters only, so a training step
// folds all of its units before the first one runs instead of paying a 5 us launch at the head of every unit's forward.
constexpr int EC_FOLD_MAX = 8;
struct EcFoldOne {
    const float* W[9]; const float* bias[9];
    float* Wpq; float* bpq;
    int rowoff[10], width[9];
    int nconvs, C, S, pad;
};
struct EcFoldBatch { EcFoldOne u[EC_FOLD_MAX]; };
static_assert(sizeof(EcFoldBatch) <= 4032, "kernel argument block");
__global__ __launch_bounds__(256) void ec_fold_batch_kernel(EcFoldBatch fb) {
    const EcFoldOne& cv = fb.u[blockIdx.y];
    const int total = cv.S * cv.C;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int srow = i / cv.C, col = i % cv.C;
        int t = 0;
        while (t + 1 < cv.nconvs && srow >= cv.rowoff[t + 1]) ++t;
        const float* w = cv.W[t] + (size_t)(srow - cv.rowoff[t]) * cv.width[t];
        cv.Wpq[(size_t)srow * cv.C + col] = w[col] - w[2 * cv.C + col];
        cv.Wpq[(size_t)(cv.S + srow) * cv.C + col] = w[cv.C + col] + w[2 * cv.C + col];
        if (col == 0) { cv.bpq[srow] = cv.bias[t][srow - cv.rowoff[t]]; cv.bpq[cv.S + srow] = 0.f; }
    }
}
// dW_t[r, :] = [dWp | dWq | dWq - dWp | sum_chunks part[:, rowoff_t + r, :g t]],  dbias_t[r] = sum_chunks bpart[:, rowoff_t + r].
// 64 consecutive elements per workgroup, the chunk sum split four ways (threadIdx.y) and joined through LDS.
constexpr int ASM_G = 16;            // groups of 64 threads that share the chunk range of an output element
// dWpq arrives as `nslab` split-K slabs [nslab][2 S, C] of its point GEMM (nslab = 1: the finished product): their sum is taken here,
// by the same 16 lanes per element that add the growth partials - it was a launch of its own (gemm_reduce_kernel) in front of this
// one, seven per step, and a replayed step pays ~5 - 10 us per kernel boundary.
__global__ __launch_bounds__(64 * ASM_G) void ec_assemble_kernel(EcConvs cv, const float* dWpq, int nslab, const float* part, int nchunk,
                                                                const float* bpart, int total) {
    __shared__ double sh[ASM_G][64], shb[ASM_G][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    int t = 0, r = 0, col = 0, srow = 0;
    bool ok = i < total, grow = false;
    if (ok) {
        int rem = i;
        while (rem >= cv.rows[t] * cv.width[t]) { rem -= cv.rows[t] * cv.width[t]; ++t; }
        r = rem / cv.width[t]; col = rem % cv.width[t];
        srow = cv.rowoff[t] + r;
        grow = col >= 3 * cv.C;
    }
    double s = 0.0, sb = 0.0;
    if (ok && grow) {
        const int u = col - 3 * cv.C;
        int k = ty;
        for (; k + 3 * ASM_G < nchunk; k += 4 * ASM_G) {              // four loads in flight per thread
            const float v0 = part[((size_t)k * cv.S + srow) * cv.GT + u], v1 = part[((size_t)(k + ASM_G) * cv.S + srow) * cv.GT + u];
            const float v2 = part[((size_t)(k + 2 * ASM_G) * cv.S + srow) * cv.GT + u];
            const float v3 = part[((size_t)(k + 3 * ASM_G) * cv.S + srow) * cv.GT + u];
            s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
        }
        for (; k < nchunk; k += ASM_G) s += (double)part[((size_t)k * cv.S + srow) * cv.GT + u];
    }
    if (ok && !grow) {                                 // [dWp | dWq | dWq - dWp] from the slabs of dWpq = [dWp; dWq]
        const int kind = col / cv.C, cc = col - kind * cv.C;
        const size_t ip = (size_t)srow * cv.C + cc, iq = (size_t)(cv.S + srow) * cv.C + cc, stride = (size_t)2 * cv.S * cv.C;
        double sp = 0.0, sq = 0.0;
        for (int k = ty; k < nslab; k += ASM_G) {
            const float vp = dWpq[k * stride + (kind != 1 ? ip : iq)], vq = dWpq[k * stride + (kind != 0 ? iq : ip)];
            sp += (double)vp; sq += (double)vq;
        }
        s = kind == 0 ? sp : (kind == 1 ? sq : sq - sp);
    }
    if (ok && col == 0)
        for (int k = ty; k < nchunk; k += ASM_G) sb += (double)bpart[(size_t)k * cv.S + srow];
    sh[ty][tx] = s; shb[ty][tx] = sb;
    __syncthreads();
    if (ty != 0 || !ok) return;
    float v;
    {
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < ASM_G; ++k) a += sh[k][tx];
        v = (float)a;
    }
    cv.dW[t][(size_t)r * cv.width[t] + col] = v;
    if (col == 0) {
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < ASM_G; ++k) a += shb[k][tx];
        cv.dbias[t][r] = (float)a;
    }
}

__global__ __launch_bounds__(256) void ec_zero_kernel(f4* p, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) p[i] = pf_splat(0.f);
}

template <typename KERNEL>
void allow_lds(KERNEL k, size_t bytes) {
    pf_allow_lds(reinterpret_cast<const void*>(k), bytes);
}

// SyncBN on the fused kernels: after a launch whose StatFin defers (the local sums sit in fin.defer), the caller's callback
// all-reduces them over the ranks (stream-ordered, e.g. torch.distributed.all_reduce on the tensor behind the pointer) and
// one small launch finishes the layer with the global sums.
typedef int (*PfSyncFn)(void* user, double* sums, int n, void* stream);
int stat_sync(const StatFin& fin, int ncol, PfSyncFn cb, void* user, hipStream_t s) {
    if (!fin.defer) return PF_OK;
    if (!cb) return PF_ERR_NULL;
    if (cb(user, fin.defer, 2 * STAT_W + 1, (void*)s) != 0) return PF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(stat_finalize_kernel, dim3(1), dim3(STAT_W), 0, s, fin, ncol);
    return PF_OK;
}

struct Dims {
    int T, GT, S, nconvs;
    long long E;
    int ntiles, grid, grid_light, nchunk, chunk;
};
#ifndef PF_EC_DW_CHUNK
#define PF_EC_DW_CHUNK 512
#endif
constexpr int EC_DW_CHUNK = PF_EC_DW_CHUNK;
#ifndef PF_EC_DW_CHUNK_SCALE
#define PF_EC_DW_CHUNK_SCALE 1
#endif
#ifndef EC_DW_WAVES
#define EC_DW_WAVES 8
#endif

int ec_dims(const PfEcTrain* p, Dims& d) {
    if (!p) return PF_ERR_NULL;
    if (p->B <= 0 || p->N <= 0 || p->K <= 0 || p->C <= 0 || p->nconv < 1 || p->nconv > 8) return PF_ERR_SHAPE;
    if (p->growth != 8 && p->growth != 16 && p->growth != 32) return PF_ERR_UNSUPPORTED;
    if (p->odim % 16 != 0 || p->odim > 128 || p->odim < 16) return PF_ERR_UNSUPPORTED;
    if (p->pooling && p->K != 16) return PF_ERR_UNSUPPORTED;
    d.T = p->B * p->N;
    d.GT = p->growth * p->nconv;
    if (d.GT != 32 && d.GT != 64 && d.GT != 128) return PF_ERR_UNSUPPORTED;
    d.S = d.GT + p->odim;
    d.nconvs = p->nconv + 1;
    d.E = (long long)d.T * p->K;
    if (d.E % 16 != 0 || d.E > (1ll << 30)) return PF_ERR_SHAPE;
    d.ntiles = (int)(d.E / 16);
    d.grid = (d.ntiles + 3) / 4 < EC_GRID ? (d.ntiles + 3) / 4 : EC_GRID;
    // growth-layer kernels: measured (layer with 96 input channels, 131 072 edges) 44 / 30 / 29 / 41 us at 128 / 256 / 512 / 2048
    // workgroups: beyond 2 per CU the fixed cost per workgroup (weight staging, 64 statistics atomics) outweighs the latency hiding
    d.grid_light = d.grid;
    // edges per split-K chunk of the weight-gradient launch: one workgroup per chunk with (odim + GT) / 32 waves - the narrow units
    // (GT = 32 / 64: 2 / 4 waves) get shorter chunks so that they, too, put two waves on every SIMD
    d.chunk = PF_EC_DW_CHUNK_SCALE ? (d.GT <= 32 ? EC_DW_CHUNK / 4 : (d.GT <= 64 ? EC_DW_CHUNK / 2 : EC_DW_CHUNK)) : EC_DW_CHUNK;
    d.nchunk = (int)((d.E + d.chunk - 1) / d.chunk);
    return PF_OK;
}
EcConvs ec_convs(const PfEcTrain* p, const Dims& d) {
    EcConvs cv{};
    cv.nconvs = d.nconvs; cv.C = p->C; cv.S = d.S; cv.GT = d.GT;
    int off = 0;
    for (int t = 0; t < d.nconvs; ++t) {
        cv.W[t] = p->W[t]; cv.bias[t] = p->bias[t]; cv.dW[t] = p->dW[t]; cv.dbias[t] = p->dbias[t];
        cv.rows[t] = t < p->nconv ? p->growth : p->odim;
        cv.width[t] = 3 * p->C + p->growth * t;
        cv.rowoff[t] = off;
        off += cv.rows[t];
    }
    cv.rowoff[d.nconvs] = off;
    return cv;
}
long long gemm_ws_max(const PfEcTrain* p, const Dims& d) {
    long long g1 = pf_gemm_ws_floats(2 * d.S, p->C, d.T), g2 = pf_gemm_ws_floats(d.T, p->C, 2 * d.S),
              g3 = pf_gemm_ws_floats(d.T, 2 * d.S, p->C);
    long long gm = g1 > g2 ? g1 : g2;
    return gm > g3 ? gm : g3;
}

// ---- the unit's forward as ONE persistent launch (ec_fwdp_kernel): shapes it is instantiated for, and whether all of its
// workgroups can be resident at once on this device (grid barriers).  The caller's PF_EC_PERSISTENT flag is the promise that no
// OTHER barrier kernel of this process runs beside it (two barrier kernels can starve each other; ordinary kernels only delay it).
template <int G, int ODIM>
size_t ecp_lds_bytes() {
    size_t fl = 0;
    for (int t = 1; t < 4; ++t) fl += (size_t)((G + 15) / 16) * 16 * (((G * t + 15) & ~15) + 4);
    return fl * sizeof(float) + (size_t)(ODIM / 16) * (G * 4 / 32) * 2 * 64 * 16;
}
// (the occupancy answers are cached per device: the queries ran on every call of the 1 ms step and inside graph captures - ADVICE r4)
constexpr int ECP_MAXDEV = 16;
template <int G, int ODIM>
int ecp_capacity() {
    static int cache[ECP_MAXDEV];                   // 0 = not asked yet, -1 = does not fit, > 0 = workgroups that can be resident
    static std::mutex mu;
    int ncu = 0, dev = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < ECP_MAXDEV && cache[dev] != 0) return cache[dev] > 0 ? cache[dev] : 0;
    const size_t lds = ecp_lds_bytes<G, ODIM>();
    allow_lds(ec_fwdp_kernel<G, 4, ODIM>, lds);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    int cap = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ec_fwdp_kernel<G, 4, ODIM>, ECP_T, lds) != hipSuccess) (void)hipGetLastError();
    else {
        const int want = ECP_TPW / ecp_tpw(G);      // workgroups per CU the unit's grid needs resident (2 for the narrow units)
        cap = (per_cu < want ? per_cu : want) * ncu;
    }
    if (dev >= 0 && dev < ECP_MAXDEV) cache[dev] = cap > 0 ? cap : -1;
    return cap;
}
bool ec_persistent_ok(const PfEcTrain* p, const Dims& d) {
    if (!(p->flags & PF_EC_PERSISTENT) || (p->flags & PF_TRAIN_DETERMINISTIC) || !p->sync || !p->pooling || p->K != 16 || p->nconv != 4 ||
        p->sync_sums)
        return false;
    int cap = 0;
    if (p->growth == 8 && p->odim == 32) cap = ecp_capacity<8, 32>();
    else if (p->growth == 16 && p->odim == 64) cap = ecp_capacity<16, 64>();
    else if (p->growth == 32 && p->odim == 128) cap = ecp_capacity<32, 128>();
    return cap > 0 && (long long)d.ntiles <= (long long)cap * ECP_WAVES * ecp_tpw(p->growth);
}
template <int G, int ODIM>
size_t ecpb_lds_bytes() {
#ifdef PF_EC_BWDG_F32
    return sizeof(float) * ((size_t)G * 4 * (ODIM + 4) + (size_t)4 * ((G + 15) / 16) * 16 * (G * 4 + 4));
#else
    int nf = (G * 4 / 16) * (ODIM / 32);
    for (int s = 0; s < 4; ++s) nf += ((G + 15) / 16) * (G * 4 / 32 - (G * (s + 1)) / 32);
    return (size_t)nf * 2 * 64 * 16;
#endif
}
template <int G, int ODIM>
bool ecpb_fits() {
    static int cache[ECP_MAXDEV];                   // 0 = not asked yet, 1 = fits, -1 = does not
    static std::mutex mu;
    int dev = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < ECP_MAXDEV && cache[dev] != 0) return cache[dev] > 0;
    const size_t lds = ecpb_lds_bytes<G, ODIM>();
    allow_lds(ec_bwdp_kernel<G, 4, ODIM>, lds);
    bool fits = false;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ec_bwdp_kernel<G, 4, ODIM>, ECP_T, lds) != hipSuccess) (void)hipGetLastError();
    else fits = per_cu >= ECP_TPW / ecp_tpw(G);
    if (dev >= 0 && dev < ECP_MAXDEV) cache[dev] = fits ? 1 : -1;
    return fits;
}
bool ec_bwd_persistent_fits(const PfEcTrain* p) {
    return p->growth == 8 ? ecpb_fits<8, 32>() : (p->growth == 16 ? ecpb_fits<16, 64>() : ecpb_fits<32, 128>());
}
int ecp_grid(const Dims& d, int tpw) {
    const int wgs = (d.ntiles + ECP_WAVES * tpw - 1) / (ECP_WAVES * tpw);               // fewest workgroups that hold every tile ...
    int ncu = 0, dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const int spread = (d.ntiles + ECP_WAVES - 1) / ECP_WAVES;                         // ... spread over the CUs when there are fewer tiles
    return spread < ncu ? (spread > wgs ? spread : wgs) : (wgs > ncu ? wgs : ncu);
}
int ec_bwd_persistent(const PfEcTrain* p, const Dims& d, const EcConvs& cv, hipStream_t s) {
    EcBwdPArgs a{};
    a.dh = p->dout; a.arg = p->arg; a.dA = p->dA; a.Y = p->Y; a.ld = d.GT; a.aff = p->aff; a.coef = p->coef;
    a.Wout = p->W[p->nconv] + 3 * p->C; a.ldwout = cv.width[p->nconv];
    for (int t = 0; t < p->nconv; ++t) {
        a.Wg[t] = p->W[t] + 3 * p->C; a.ldwg[t] = cv.width[t];
        a.dgamma[t] = p->dgamma[t]; a.dbeta[t] = p->dbeta[t];
    }
    a.ntiles = d.ntiles; a.slope = p->slope; a.R = (double)d.E; a.acc = p->stat; a.sync = p->sync;
#if PF_EC_BWDP_DP
    if (p->K == 16 && p->csr_off && p->csr_edge) { a.dP = p->dPQ; a.ldp = 2 * d.S; }       // (the scatter form zeroes and accumulates dPQ itself)
#endif
    const int grid = ecp_grid(d, ecp_tpw(p->growth));
    const size_t l8 = ecpb_lds_bytes<8, 32>(), l16 = ecpb_lds_bytes<16, 64>(), l32 = ecpb_lds_bytes<32, 128>();
    if (p->growth == 8) hipLaunchKernelGGL((ec_bwdp_kernel<8, 4, 32>), dim3(grid), dim3(ECP_T), l8, s, a);
    else if (p->growth == 16) hipLaunchKernelGGL((ec_bwdp_kernel<16, 4, 64>), dim3(grid), dim3(ECP_T), l16, s, a);
    else hipLaunchKernelGGL((ec_bwdp_kernel<32, 4, 128>), dim3(grid), dim3(ECP_T), l32, s, a);
    return pf_last_launch_status();
}
int ec_fwd_persistent(const PfEcTrain* p, const Dims& d, const EcConvs& cv, hipStream_t s) {
    EcFwdPArgs a{};
    a.Y = p->Y; a.ldy = d.GT; a.aff = p->aff; a.pq = p->PQ; a.ldpq = 2 * d.S; a.S = d.S; a.idx = p->idx; a.N = p->N;
    a.ntiles = d.ntiles; a.slope = p->slope; a.out = p->out; a.arg = p->arg;
    for (int t = 0; t < p->nconv; ++t) {
        a.Wg[t] = p->W[t] + 3 * p->C; a.ldwg[t] = cv.width[t];
        a.gamma[t] = p->gamma[t]; a.beta[t] = p->beta[t]; a.run_mean[t] = p->run_mean[t]; a.run_var[t] = p->run_var[t];
    }
    a.Wout = p->W[p->nconv] + 3 * p->C; a.ldwout = cv.width[p->nconv];
    a.eps = p->eps; a.momentum = p->momentum; a.R = (double)d.E; a.acc = p->stat; a.sync = p->sync;
    const int grid = ecp_grid(d, ecp_tpw(p->growth));
    const size_t l8 = ecp_lds_bytes<8, 32>(), l16 = ecp_lds_bytes<16, 64>(), l32 = ecp_lds_bytes<32, 128>();
    if (p->growth == 8) hipLaunchKernelGGL((ec_fwdp_kernel<8, 4, 32>), dim3(grid), dim3(ECP_T), l8, s, a);
    else if (p->growth == 16) hipLaunchKernelGGL((ec_fwdp_kernel<16, 4, 64>), dim3(grid), dim3(ECP_T), l16, s, a);
    else hipLaunchKernelGGL((ec_fwdp_kernel<32, 4, 128>), dim3(grid), dim3(ECP_T), l32, s, a);
    return pf_last_launch_status();
}

}  // namespace

// floats of scratch for either direction: dw partials [nchunk][S][GT] + [nchunk][S] + split-K slabs of the point GEMMs
extern "C" long long pf_ec_train_ws_floats(const PfEcTrain* p) {
    Dims d;
    if (ec_dims(p, d) != PF_OK) return -1;
    return (long long)d.nchunk * d.S * (d.GT + 1) + gemm_ws_max(p, d);
}

// transposed neighbour lists: off [T+1], edge [T*K] (edge ids e = i K + k grouped by the point they point AT), cnt [T] scratch.
// Built once per step and shared by every unit that uses the same idx (pf_ec_train_bwd: csr_off / csr_edge).
extern "C" int pf_knn_csr(const int* idx, int B, int N, int K, int* off, int* edge, int* cnt, void* stream) {
    if (!idx || !off || !edge || !cnt) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || K <= 0 || (long long)B * N > (1ll << 26)) return PF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int T = B * N;
    const long long E = (long long)T * K;
    const unsigned g = (unsigned)((E + 255) / 256 > 2048 ? 2048 : (E + 255) / 256);
    hipLaunchKernelGGL(ec_zero_kernel, dim3(64), dim3(256), 0, s, reinterpret_cast<f4*>(cnt), (long long)(T + 3) / 4);
    hipLaunchKernelGGL(csr_count_kernel, dim3(g), dim3(256), 0, s, idx, N, K, E, cnt);
    hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, s, cnt, T, off);
    hipLaunchKernelGGL(csr_fill_kernel, dim3(g), dim3(256), 0, s, idx, N, K, E, cnt, edge);
    return pf_last_launch_status();
}

// pf_knn_csr for idx [B*N, K] AND for its first K2 columns (as if stored contiguously: edge ids i K2 + k) in the same four
// launches: off / edge as above, off2 [T+1], edge2 [T*K2]; cnt: 2 x ((T + 3) / 4 * 4) ints of scratch.
extern "C" int pf_knn_csr_pair(const int* idx, int B, int N, int K, int K2, int* off, int* edge, int* off2, int* edge2, int* cnt,
                               void* stream) {
    if (!idx || !off || !edge || !off2 || !edge2 || !cnt) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || K <= 0 || K2 <= 0 || K2 > K || (long long)B * N > (1ll << 26)) return PF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int T = B * N, Tpad = (T + 3) / 4 * 4;
    const long long E = (long long)T * K;
    const unsigned g = (unsigned)((E + 255) / 256 > 2048 ? 2048 : (E + 255) / 256);
    hipLaunchKernelGGL(ec_zero_kernel, dim3(64), dim3(256), 0, s, reinterpret_cast<f4*>(cnt), (long long)(2 * Tpad) / 4);
    hipLaunchKernelGGL(csr_count2_kernel, dim3(g), dim3(256), 0, s, idx, N, K, K2, E, cnt, cnt + Tpad);
    hipLaunchKernelGGL(csr_scan2_kernel, dim3(2), dim3(1024), 0, s, cnt, T, Tpad, off, off2);
    hipLaunchKernelGGL(csr_fill2_kernel, dim3(g), dim3(256), 0, s, idx, N, K, K2, E, cnt, cnt + Tpad, edge, edge2);
    return pf_last_launch_status();
}

// Sorts every list of pf_knn_csr (edge ids ascending).  The fill hands out a list's slots in arrival order; whatever is summed over
// a sorted list adds in ONE order, run after run (PF_TRAIN_DETERMINISTIC: the gather-form gradients).  Not needed otherwise.
extern "C" int pf_knn_csr_sort(const int* off, int* edge, int T, void* stream) {
    if (!off || !edge) return PF_ERR_NULL;
    if (T <= 0) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(csr_sort_kernel, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream, off, edge, T);
    return pf_last_launch_status();
}

extern "C" int pf_ec_train_fold_batch(const PfEcTrain* descs, int n, void* stream) {
    if (!descs) return PF_ERR_NULL;
    if (n < 1 || n > EC_FOLD_MAX) return PF_ERR_SHAPE;
    EcFoldBatch fb{};
    int most = 0;
    for (int k = 0; k < n; ++k) {
        const PfEcTrain* p = descs + k;
        Dims d;
        const int st = ec_dims(p, d);
        if (st) return st;
        if (!p->Wpq || !p->bpq) return PF_ERR_NULL;
        const EcConvs cv = ec_convs(p, d);
        EcFoldOne& u = fb.u[k];
        for (int t = 0; t < d.nconvs; ++t) {
            if (!p->W[t] || !p->bias[t]) return PF_ERR_NULL;
            u.W[t] = cv.W[t]; u.bias[t] = cv.bias[t]; u.width[t] = cv.width[t]; u.rowoff[t] = cv.rowoff[t];
        }
        u.rowoff[d.nconvs] = cv.rowoff[d.nconvs];
        u.Wpq = p->Wpq; u.bpq = p->bpq; u.nconvs = d.nconvs; u.C = p->C; u.S = d.S;
        most = d.S * p->C > most ? d.S * p->C : most;
    }
    hipLaunchKernelGGL(ec_fold_batch_kernel, dim3((most + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, fb);
    return pf_last_launch_status();
}

extern "C" int pf_ec_train_fwd(const PfEcTrain* p, void* stream) {
    Dims d;
    int st = ec_dims(p, d);
    if (st) return st;
    if (!p->x || !p->idx || !p->Wpq || !p->bpq || !p->PQ || !p->Y || !p->aff || !p->out || !p->ws || !p->stat) return PF_ERR_NULL;
    if (p->pooling && !p->arg) return PF_ERR_NULL;
    for (int t = 0; t < d.nconvs; ++t)
        if (!p->W[t] || !p->bias[t]) return PF_ERR_NULL;
    for (int t = 0; t < p->nconv; ++t)
        if (!p->gamma[t] || !p->beta[t]) return PF_ERR_NULL;
    if (p->ws_floats < pf_ec_train_ws_floats(p)) return PF_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const EcConvs cv = ec_convs(p, d);
    float* gws = p->ws + (long long)d.nchunk * d.S * (d.GT + 1);
    if (!(p->flags & PF_EC_PREFOLDED))
        hipLaunchKernelGGL(ec_fold_kernel, dim3((d.S * p->C + 255) / 256), dim3(256), 0, s, cv, p->Wpq, p->bpq);
    st = pf_gemm(p->x, p->C, 1, p->Wpq, 1, p->C, p->PQ, 2 * d.S, p->bpq, d.T, 2 * d.S, p->C, gws,
                 pf_gemm_ws_floats(d.T, 2 * d.S, p->C), stream);
    if (st) return st;
    const int g = p->growth;
    if (ec_persistent_ok(p, d)) return ec_fwd_persistent(p, d, cv, s);
    EcFwdArgs a{};
    a.Y = p->Y; a.ldy = d.GT; a.aff = p->aff; a.pq = p->PQ; a.ldpq = 2 * d.S; a.idx = p->idx; a.N = p->N; a.K = p->K;
    a.ntiles = d.ntiles; a.slope = p->slope;
    for (int t = 0; t < p->nconv; ++t) {
        a.W = p->W[t] + 3 * p->C; a.ldw = cv.width[t]; a.poff = g * t; a.qoff = d.S + g * t;
        a.kin = g * t; a.col0 = g * t; a.nout = g;
        a.fin = StatFin{p->stat, 1, g, g * t, d.GT, p->aff, p->gamma[t], p->beta[t], p->run_mean[t], p->run_var[t], p->eps,
                        p->momentum, nullptr, nullptr, nullptr, (double)d.E, p->sync_sums};
        a.fin.det = PF_DET(p);
        const int kin16 = (a.kin + 15) & ~15;
        const int nt = g > 16 ? 2 : 1;
        const size_t lds = sizeof(float) * ((size_t)nt * 16 * (kin16 + 4) + 2 * kin16);
        if (nt == 2) hipLaunchKernelGGL((ec_fwd_kernel<2, false, false>), dim3(d.grid_light), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((ec_fwd_kernel<1, false, false>), dim3(d.grid_light), dim3(256), lds, s, a);
        if ((st = stat_sync(a.fin, g, p->sync_cb, p->sync_user, s))) return st;     // SyncBN: global statistics before the next layer
    }
    a.W = p->W[p->nconv] + 3 * p->C; a.ldw = cv.width[p->nconv]; a.poff = d.GT; a.qoff = d.S + d.GT;
    a.kin = d.GT; a.col0 = 0; a.nout = p->odim; a.out = p->out; a.arg = p->arg; a.fin = StatFin{};
    const int nto = p->odim / 16;
#ifdef PF_EC_FWD_F32
    const size_t lds = sizeof(float) * ((size_t)(nto <= 2 ? 2 : (nto <= 4 ? 4 : 8)) * 16 * (d.GT + 4) + 2 * d.GT);
#define PF_ECO(NT)                                                                                                        \
    do {                                                                                                                  \
        if (p->pooling) { allow_lds(ec_fwd_kernel<NT, true, true>, lds);                                                  \
            hipLaunchKernelGGL((ec_fwd_kernel<NT, true, true>), dim3(d.grid), dim3(256), lds, s, a); }                    \
        else { allow_lds(ec_fwd_kernel<NT, true, false>, lds);                                                            \
            hipLaunchKernelGGL((ec_fwd_kernel<NT, true, false>), dim3(d.grid), dim3(256), lds, s, a); }                   \
    } while (0)
#else
    const int nchk = (d.GT + 31) / 32;
    const size_t lds = (size_t)(nto <= 2 ? 2 : (nto <= 4 ? 4 : 8)) * nchk * 2 * 64 * 16 + sizeof(float) * 2 * nchk * 32;
#define PF_ECO(NT)                                                                                                        \
    do {                                                                                                                  \
        if (p->pooling) { allow_lds(ec_fwd16_kernel<NT, true>, lds);                                                      \
            hipLaunchKernelGGL((ec_fwd16_kernel<NT, true>), dim3(d.grid), dim3(256), lds, s, a); }                        \
        else { allow_lds(ec_fwd16_kernel<NT, false>, lds);                                                                \
            hipLaunchKernelGGL((ec_fwd16_kernel<NT, false>), dim3(d.grid), dim3(256), lds, s, a); }                       \
    } while (0)
#endif
    if (nto <= 2) PF_ECO(2); else if (nto <= 4) PF_ECO(4); else PF_ECO(8);
#undef PF_ECO
    return pf_last_launch_status();
}

extern "C" int pf_ec_train_bwd(const PfEcTrain* p, void* stream) {
    Dims d;
    int st = ec_dims(p, d);
    if (st) return st;
    if (!p->x || !p->idx || !p->Wpq || !p->PQ || !p->Y || !p->aff || !p->dout || !p->dA || !p->dPQ || !p->coef || !p->dWpq ||
        !p->ws || !p->stat)
        return PF_ERR_NULL;
    if (p->pooling && !p->arg) return PF_ERR_NULL;
    for (int t = 0; t < d.nconvs; ++t)
        if (!p->W[t] || !p->dW[t] || !p->dbias[t]) return PF_ERR_NULL;
    for (int t = 0; t < p->nconv; ++t)
        if (!p->dgamma[t] || !p->dbeta[t]) return PF_ERR_NULL;
    if (p->ws_floats < pf_ec_train_ws_floats(p)) return PF_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const EcConvs cv = ec_convs(p, d);
    const int g = p->growth, nc = p->nconv;
    // weight gradients on their own stream (pf_train_set_dw_stream) take the second workspace: the partial slabs and the
    // scratch of the dWpq GEMM; the calling stream keeps p->ws for the dx GEMM
    const bool dw_side = pf_dw_stream_get() && pf_dw_stream_get() != stream && p->ws_dw && p->ws_dw_floats >= pf_ec_train_ws_floats(p);
    float* dwpart = dw_side ? p->ws_dw : p->ws;
    float* bpart = dwpart + (long long)d.nchunk * d.S * d.GT;
    float* gws_dw = bpart + (long long)d.nchunk * d.S;
    float* gws = p->ws + (long long)d.nchunk * d.S * (d.GT + 1);
    const bool csr = p->csr_off && p->csr_edge;
    // cleared by a kernel rather than hipMemsetAsync: see csrc/emd.hip (memset nodes inside a captured hipGraph)
    if (!csr)
        hipLaunchKernelGGL(ec_zero_kernel, dim3(512), dim3(256), 0, s, reinterpret_cast<f4*>(p->dPQ), (long long)d.T * 2 * d.S / 4);

#ifndef PF_EC_BWD_SCATTER
    const bool persistent = ec_persistent_ok(p, d) && ec_bwd_persistent_fits(p);
    if (persistent) {
        st = ec_bwd_persistent(p, d, cv, s);
        if (st) return st;
    }
    // ---- gather form: layer s = nc - 1 .. 0 collects its own gradient columns from conv_out and the later growth convs
    for (int sl = nc - 1; sl >= 0 && !persistent; --sl) {
        EcBwdgArgs a{};
        a.dh = p->dout; a.arg = p->arg; a.dyout = p->dout; a.odim = p->odim;
        a.dA = p->dA; a.Y = p->Y; a.ld = d.GT; a.aff = p->aff; a.coef = p->coef;
        a.Wout = p->W[nc] + 3 * p->C; a.ldwout = cv.width[nc];
        for (int t = 1; t < nc; ++t) { a.Wg[t] = p->W[t] + 3 * p->C; a.ldwg[t] = cv.width[t]; }
        a.s = sl; a.nc = nc; a.g = g; a.ntiles = d.ntiles; a.slope = p->slope;
        a.fin = StatFin{p->stat, 2, g, g * sl, d.GT, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, p->coef,
                        p->dgamma[sl], p->dbeta[sl], (double)d.E, p->sync_sums};
        a.fin.det = PF_DET(p);
        const int g16 = (g + 15) & ~15, od16 = (p->odim + 15) & ~15, ntg = g16 / 16;
#ifdef PF_EC_BWDG_F32
        const size_t lds = sizeof(float) * ((size_t)ntg * 16 * ((od16 + 4) + (size_t)(nc - 1 - sl) * (g16 + 4)) + 6 * g16);
#define PF_ECG(NTG, SRC)                                                                                                  \
    do { allow_lds(ec_bwdg_kernel<NTG, SRC>, lds);                                                                        \
         hipLaunchKernelGGL((ec_bwdg_kernel<NTG, SRC>), dim3(d.grid), dim3(256), lds, s, a); } while (0)
#else
        (void)od16;
        const size_t nfrag = (size_t)ntg * ((p->odim + 31) / 32 + (size_t)(nc - 1 - sl) * ((g + 31) / 32));
        const size_t lds = nfrag * 2 * 64 * 16 + sizeof(float) * 6 * g16;
#define PF_ECG(NTG, SRC)                                                                                                  \
    do { allow_lds(ec_bwdg16_kernel<NTG, SRC>, lds);                                                                      \
         hipLaunchKernelGGL((ec_bwdg16_kernel<NTG, SRC>), dim3(d.grid), dim3(256), lds, s, a); } while (0)
#endif
        if (p->pooling) { if (ntg == 1) PF_ECG(1, 0); else PF_ECG(2, 0); }
        else { if (ntg == 1) PF_ECG(1, 1); else PF_ECG(2, 1); }
#undef PF_ECG
        if ((st = stat_sync(a.fin, g, p->sync_cb, p->sync_user, s))) return st;     // SyncBN: global sums before the transform of this layer
    }
#else
    if (p->sync_sums) return PF_ERR_UNSUPPORTED;                     // SyncBN is wired into the gather form only
    // ---- conv_out: dA = dYout Wg_out (+ sums of the last growth layer)
    {
        EcBwdArgs a{};
        a.dh = p->dout; a.arg = p->arg; a.dyout = p->dout; a.dA = p->dA; a.Y = p->Y; a.ld = d.GT; a.aff = p->aff; a.coef = p->coef;
        a.c0 = 0; a.kin = p->odim; a.W = p->W[nc] + 3 * p->C; a.ldw = cv.width[nc]; a.nout = d.GT;
        a.sc0 = g * (nc - 1); a.sg = g; a.ntiles = d.ntiles; a.slope = p->slope;
        a.fin = StatFin{p->stat, 2, g, g * (nc - 1), d.GT, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, p->coef,
                        p->dgamma[nc - 1], p->dbeta[nc - 1], (double)d.E};
        a.fin.det = PF_DET(p);
        const int nt = d.GT / 16 <= 2 ? 2 : (d.GT / 16 <= 4 ? 4 : 8);
        const size_t lds = sizeof(float) * ((size_t)nt * 16 * (((p->odim + 15) & ~15) + 4));
#define PF_ECB(NT, SRC)                                                                                                   \
    do { allow_lds(ec_bwd_kernel<NT, SRC>, lds);                                                                          \
         hipLaunchKernelGGL((ec_bwd_kernel<NT, SRC>), dim3(SRC == 2 ? d.grid_light : d.grid), dim3(256), lds, s, a); } while (0)
        if (p->pooling) { if (nt == 2) PF_ECB(2, 0); else if (nt == 4) PF_ECB(4, 0); else PF_ECB(8, 0); }
        else { if (nt == 2) PF_ECB(2, 1); else if (nt == 4) PF_ECB(4, 1); else PF_ECB(8, 1); }
    }
    // ---- growth layers, last to first
    for (int t = nc - 1; t >= 1; --t) {
        EcBwdArgs a{};
        a.dA = p->dA; a.Y = p->Y; a.ld = d.GT; a.aff = p->aff; a.coef = p->coef;
        a.c0 = g * t; a.kin = g; a.W = p->W[t] + 3 * p->C; a.ldw = cv.width[t]; a.nout = g * t;
        a.sc0 = g * (t - 1); a.sg = g; a.ntiles = d.ntiles; a.slope = p->slope;
        a.fin = StatFin{p->stat, 2, g, g * (t - 1), d.GT, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, p->coef,
                        p->dgamma[t - 1], p->dbeta[t - 1], (double)d.E};
        a.fin.det = PF_DET(p);
        const int nt16 = (a.nout + 15) / 16;
        const int nt = nt16 <= 1 ? 1 : (nt16 <= 2 ? 2 : (nt16 <= 4 ? 4 : 8));
        const int kin16 = (g + 15) & ~15;
        const size_t lds = sizeof(float) * ((size_t)nt * 16 * (kin16 + 4) + 6 * kin16);
        if (nt == 1) PF_ECB(1, 2); else if (nt == 2) PF_ECB(2, 2); else if (nt == 4) PF_ECB(4, 2); else PF_ECB(8, 2);
#undef PF_ECB
    }
#endif
#ifndef PF_EC_BWD_SCATTER
    if (!persistent)
#endif
    {
        const long long n = d.E * (g / 4);
        hipLaunchKernelGGL(ec_bwd0_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, s,
                           p->dA, p->Y, d.GT, p->aff, p->coef, g, d.E, p->slope);
    }
    // ---- dPQ (+ bias column sums)
    {
        EcPqBwdArgs a{p->dA, d.GT, p->dout, p->arg, p->dout, p->pooling, p->idx, p->N, p->K, d.GT, p->odim, d.S, (long long)d.T,
                      p->dPQ};
        const long long n = (long long)d.T * d.S;
        const dim3 grid((unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256));
        const dim3 grid4((unsigned)((n / 4 + 255) / 256 > 8192 ? 8192 : (n / 4 + 255) / 256));
        const int p_done = (PF_EC_BWDP_DP && persistent && p->K == 16) ? 1 : 0;
        if (csr) hipLaunchKernelGGL(ec_pq_bwd_csr_kernel, grid4, dim3(256), 0, s, EcPqCsrArgs{a, p->csr_off, p->csr_edge, p_done});
        else hipLaunchKernelGGL(ec_pq_bwd_kernel, grid, dim3(256), 0, s, a);
    }
    // ---- dx first: it is what the unit before this one waits for
    if (p->dx) {
        st = pf_gemm_addend(0, p->dPQ, 2 * d.S, 1, p->Wpq, p->C, 1, p->dx, p->C, nullptr, p->dx_add, d.T, p->C, 2 * d.S, gws,
                            pf_gemm_ws_floats(d.T, p->C, 2 * d.S), stream, nullptr);
        if (st) return st;
    }
    // ---- growth-weight gradients (partials), dWpq, assembly: nobody reads them before the optimizer
    hipStream_t sm = s;                                   // (the kernels below are written with `s` and `stream`)
    if (dw_side) { s = pf_dw_fork(sm); stream = (void*)s; }
    {
        EcDwArgs a{p->dA, p->Y, d.GT, p->aff, p->dout, p->arg, p->dout, p->pooling, g, d.GT, p->odim, d.S, p->K, d.E, d.chunk,
                   p->slope, dwpart, bpart};
        // jobs of the no-LDS kernel: conv_out strips, then growth strips (a strip's column tiles: what its LAST row's layer sees)
        EcDw2Args a2{};
        a2.d = a;
        int nj = 0;
        bool direct = d.GT % 32 == 0 && p->odim % 32 == 0 && d.GT <= 128 && (!p->pooling || p->K == 16) && d.chunk % 16 == 0;
#ifdef PF_EC_DW_STAGED
        direct = false;
#endif
        if (direct) {
            for (int rt = 0; rt < p->odim / 32 && nj < 8; ++rt) a2.job[nj++] = EcDw2Job{1, rt, d.GT / 32};
            for (int rt = d.GT / 32 - 1; rt >= 0; --rt) {
                const int see = ((rt * 32 + 31) / g) * g;                   // growth columns u < g * layer(last row)
                const int nct = (see + 31) / 32;                            // 0: the strip still owes its bias sums
                if (nj >= 8) { direct = false; break; }
                a2.job[nj++] = EcDw2Job{0, rt, nct};
            }
            a2.njob = nj;
        }
        if (direct) {
#ifdef PF_EC_DW_F32
            hipLaunchKernelGGL(ec_dw2_kernel, dim3(d.nchunk), dim3(64 * nj), 0, s, a2);
#else
#if PF_EC_DW4
            if (d.chunk % DW4_SE == 0) {
                const size_t lds = (size_t)2 * 4 * (d.GT / 32) * 2 * 64 * 16;
                allow_lds(ec_dw4_kernel, lds);
                hipLaunchKernelGGL(ec_dw4_kernel, dim3(d.nchunk), dim3(512), lds, s, a2);
            } else
#endif
            {
#if PF_EC_DW5
                // two strips per wave: the conv_out strips pair up as they come (equal tile counts), the growth strips - tile
                // counts descending - as (first, last), (second, second to last), ... so that every wave has about the same work
                EcDw2Args a5 = a2;
                int no = 0;
                while (no < nj && a2.job[no].out) ++no;                 // conv_out jobs first in a2
                int k = 0;
                for (int i = 0; i < no; ++i) a5.job[k++] = a2.job[i];
                if (no & 1) { a5.job[k++] = a2.job[nj - 1]; }           // an odd conv_out strip takes the lightest growth strip
                const int g_lo = no, g_hi = (no & 1) ? nj - 1 : nj;     // growth jobs left: [g_lo, g_hi)
                for (int lo = g_lo, hi = g_hi - 1; lo <= hi; ++lo, --hi) {
                    a5.job[k++] = a2.job[lo];
                    if (hi != lo) a5.job[k++] = a2.job[hi];
                }
                hipLaunchKernelGGL(ec_dw5_kernel, dim3(d.nchunk), dim3(64 * ((nj + 1) / 2)), 0, s, a5);
#else
                hipLaunchKernelGGL(ec_dw3_kernel, dim3(d.nchunk), dim3(64 * nj), 0, s, a2);
#endif
            }
#endif
        } else {
            const int ramax = p->odim > d.GT ? p->odim : d.GT;
            const size_t lds = sizeof(float) * (size_t)DW_EB * ((ramax + 16) + (d.GT + 16));
            pf_allow_lds(reinterpret_cast<const void*>(ec_dw_kernel<EC_DW_WAVES>), lds);
            hipLaunchKernelGGL(ec_dw_kernel<EC_DW_WAVES>, dim3(d.nchunk, 2), dim3(64 * EC_DW_WAVES), lds, s, a);
        }
    }
    // slabs of the small-C kernel: as many as the GEMM scratch of this unit holds, at most 128
    const long long dw_cap = gemm_ws_max(p, d) / ((long long)2 * d.S * p->C);
    const int dw_want = (int)(dw_cap < 128 ? dw_cap : 128);
    const int dw_chunk = dw_want > 0 ? (d.T + dw_want - 1) / dw_want : d.T, dw_slabs = (d.T + dw_chunk - 1) / dw_chunk;
    const float* asm_src = p->dWpq;
    int asm_slabs = 1;
    if (p->C <= 4 && dw_want >= 16) {
        hipLaunchKernelGGL(ec_dwpq_small_kernel, dim3(dw_slabs, (2 * d.S + 255) / 256), dim3(256), 0, s, p->dPQ, p->x, 2 * d.S, p->C,
                           d.T, dw_chunk, gws_dw);
        st = pf_last_launch_status();
        asm_src = gws_dw; asm_slabs = dw_slabs;
    } else {
        int nsl = 0;                                    // split-K slabs left in gws_dw (0: the product went straight into dWpq)
        st = pf_gemm_addend(0, p->dPQ, 1, 2 * d.S, p->x, p->C, 1, p->dWpq, p->C, nullptr, nullptr, 2 * d.S, p->C, d.T, gws_dw,
                            pf_gemm_ws_floats(2 * d.S, p->C, d.T), stream, &nsl);
        if (nsl > 0) { asm_src = gws_dw; asm_slabs = nsl; }
    }
    if (st) return st;
    int total = 0;
    for (int t = 0; t < d.nconvs; ++t) total += cv.rows[t] * cv.width[t];
    hipLaunchKernelGGL(ec_assemble_kernel, dim3((total + 63) / 64), dim3(64 * ASM_G), 0, s, cv, asm_src, asm_slabs, dwpart, d.nchunk, bpart,
                       total);
    return pf_last_launch_status();
}

// =====================================================================================================================
// BatchNorm MLPs of the interpolation module in the training step: DistanceEncoder and WeightEstimationUnit
// (modules/discrete/interpflow.py:85-151: Conv2d 1x1 + BatchNorm2d + LeakyReLU(0.01), twice, then Conv2d 1x1) on the
// [B N K, C] edge rows.  Same construction as the EdgeConv unit above, minus the neighbour gather: a layer's kernel applies
// the PREVIOUS layer's BatchNorm + LeakyReLU on load, stores its own pre-BatchNorm output and leaves the column sums in the
// epilogue (finalised by the last workgroup); the backward forms BatchNorm-backward on load.  The weight unit's input
// cat[d, feat] (256 wide) is never built: the first layer runs as two K-passes over the two tensors.
// =====================================================================================================================
namespace {

struct BnlFwdArgs {
    const float* addA; const float* addB;      // both non-NULL: no product at all - out = addA + addB ([rows, nout] each), statistics as usual
    const float* X; int ldx, kin;              // input rows [rows, ldx], kin <= 128 columns used
    const float* sc; const float* sh;          // BatchNorm scale / shift of the producing layer (nullable: raw input)
    float slope;
    const float* W; int ldw;                   // W[c * ldw + u], c < nout, u < kin (already offset to this K-slice)
    const float* bias;                         // nullable
    float* out; int nout;                      // [rows, nout]
    int accum;                                 // out += (second K-pass)
    int rows, ntiles;
    int want_stats;
    StatFin fin;
};

#ifndef PF_BNL_PREFETCH
#define PF_BNL_PREFETCH 0
#endif
template <int NT, bool SUM2 = false>
__global__ __launch_bounds__(256) void bnl_fwd_kernel(BnlFwdArgs a) {
    extern __shared__ float lds[];
    __shared__ float red[8 * STAT_W];
    const int kin16 = (a.kin + 15) & ~15, kp = kin16 + 4, KS = kin16 / 16;
    float* Wl = lds;
    float* al = lds + NT * 16 * kp;
    float* bl = al + kin16;
    constexpr bool sum2 = SUM2;                                   // its own instantiation: out = addA + addB, no weights, no product
    for (int c = threadIdx.x >> 4; c < NT * 16 && !sum2; c += 16) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int u = (threadIdx.x & 15) + 16 * k;
            v[k] = (u < kin16 && c < a.nout && u < a.kin) ? a.W[(size_t)c * a.ldw + u] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int u = (threadIdx.x & 15) + 16 * k;
            if (u < kin16) Wl[c * kp + u] = v[k];
        }
    }
    for (int i = threadIdx.x; i < kin16; i += 256) {
        al[i] = i < a.kin ? (a.sc ? a.sc[i] : 1.f) : 0.f;
        bl[i] = (i < a.kin && a.sh) ? a.sh[i] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, q = lane >> 4;
    const bool vec = (a.kin & 3) == 0 && (a.ldx & 3) == 0;
    const float slope = a.sc ? a.slope : 1.f;                     // raw input: identity
    float s0[NT], s1[NT], bv[NT], piv[NT];                      // piv: centred statistics, see ec_fwd_kernel
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s0[nt] = s1[nt] = 0.f;
        const int col = nt * 16 + row;
        bv[nt] = (a.bias && col < a.nout) ? a.bias[col] : 0.f;
        piv[nt] = (a.want_stats && a.fin.run_mean && col < a.nout) ? a.fin.run_mean[col] : 0.f;
    }
    // -DPF_BNL_PREFETCH=1: the NEXT tile's rows in flight while this one is multiplied (a wave has ~2 tiles at the bench shape and
    // pays one memory latency for each).  MEASURED NEGATIVE (round 5, same box, whole step): 4.592 vs 4.557 ms - 32 more
    // registers take the 128-wide shape from 3 to 2 waves per SIMD, and the branch runs beside the main chain anyway
    auto loadx = [&](int tile, f4 (&xv_)[8]) {
        const int rr = min(tile * 16 + row, a.rows - 1);
        const float* xrow = a.X + (size_t)rr * a.ldx;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            xv_[ks] = pf_splat(0.f);
            const int u = ks * 16 + 4 * q;
            if (ks < KS && u < a.kin) {
                if (vec) xv_[ks] = *reinterpret_cast<const f4*>(xrow + u);
                else {
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        if (u + w < a.kin) xv_[ks][w] = xrow[u + w];
                }
            }
        }
    };
    const int tstep = gridDim.x * 4;
#if PF_BNL_PREFETCH
    f4 xn[8];
    if (!sum2 && (int)(blockIdx.x * 4 + wave) < a.ntiles) loadx(blockIdx.x * 4 + wave, xn);
#endif
    for (int tile = blockIdx.x * 4 + wave; tile < a.ntiles; tile += tstep) {
        const int r0 = tile * 16;
        f4 xv[8];
#if PF_BNL_PREFETCH
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xv[ks] = xn[ks];
        if (!sum2 && tile + tstep < a.ntiles) loadx(tile + tstep, xn);
#else
        if (!sum2) loadx(tile, xv);
#endif
        f4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = pf_splat(0.f);
        if (sum2) {                                               // the accumulator layout read straight from the two tensors
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 16 + row;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rw = r0 + 4 * q + r;
                    if (col < a.nout && rw < a.rows) acc[nt][r] = a.addA[(size_t)rw * a.nout + col] + a.addB[(size_t)rw * a.nout + col];
                }
            }
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks < KS && !sum2) {
                const int u = ks * 16 + 4 * q;
                const f4 av = lrelu4(xv[ks] * *reinterpret_cast<const f4*>(al + u) + *reinterpret_cast<const f4*>(bl + u), slope);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = mfma4(av, *reinterpret_cast<const f4*>(Wl + (nt * 16 + row) * kp + u), acc[nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = nt * 16 + row;
            if (col < a.nout) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rw = r0 + 4 * q + r;
                    if (rw < a.rows) {
                        float* op = a.out + (size_t)rw * a.nout + col;
                        float v = acc[nt][r] + bv[nt];
                        if (a.accum) v += *op;
                        *op = v;
                        const float vc = v - piv[nt];
                        s0[nt] += vc; s1[nt] = fmaf(vc, vc, s1[nt]);
                    }
                }
            }
        }
    }
    if (a.want_stats) stat_flush<NT>(s0, s1, 0, a.nout, a.fin, red);
}

// out = addA + addB ([rows, nout], nout = 16 NT) with the column statistics of the sum (PF_BNMLP_SUM_INPUTS): a streaming kernel -
// float4 per thread along the row, the column sums kept per thread over its rows, added over the workgroup's row groups through
// LDS and handed to stat_flush in the accumulator layout it expects (wave 0, the q = 0 lanes).  (The first version read the sum
// in the MFMA accumulator layout of bnl_fwd_kernel - 4-byte loads, 16 rows apart: 105 us for 100 MB.)
template <int NT>
__global__ __launch_bounds__(256) void bnl_sum_kernel(BnlFwdArgs a) {
    constexpr int C = 16 * NT, C4 = C / 4, RPP = 256 / C4;             // threads per row, rows per pass
    static_assert(256 % C4 == 0 && C4 <= 256, "shape");
    __shared__ float red[8 * STAT_W];
    __shared__ float part[2][RPP][C];
    const int c4 = threadIdx.x % C4, rr = threadIdx.x / C4;
    f4 piv = pf_splat(0.f);
    if (a.want_stats && a.fin.run_mean) piv = *reinterpret_cast<const f4*>(a.fin.run_mean + 4 * c4);
    f4 s0 = pf_splat(0.f), s1 = pf_splat(0.f);
    const long long step = (long long)gridDim.x * RPP;
    long long row = (long long)blockIdx.x * RPP + rr;
    for (; row + 3 * step < a.rows; row += 4 * step) {                     // eight loads in flight per thread
        f4 x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x[u] = *reinterpret_cast<const f4*>(a.addA + (row + u * step) * C + 4 * c4);
            y[u] = *reinterpret_cast<const f4*>(a.addB + (row + u * step) * C + 4 * c4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f4 v = x[u] + y[u];
            *reinterpret_cast<f4*>(a.out + (row + u * step) * C + 4 * c4) = v;
            const f4 vc = v - piv;
            s0 += vc; s1 += vc * vc;
        }
    }
    for (; row < a.rows; row += step) {
        const f4 v = *reinterpret_cast<const f4*>(a.addA + row * C + 4 * c4) + *reinterpret_cast<const f4*>(a.addB + row * C + 4 * c4);
        *reinterpret_cast<f4*>(a.out + row * C + 4 * c4) = v;
        const f4 vc = v - piv;
        s0 += vc; s1 += vc * vc;
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) { part[0][rr][4 * c4 + w] = s0[w]; part[1][rr][4 * c4 + w] = s1[w]; }
    __syncthreads();
    float t0[NT], t1[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        t0[nt] = t1[nt] = 0.f;
        if (threadIdx.x < 16) {                                           // wave 0, q = 0: column nt * 16 + lane
            const int c = nt * 16 + threadIdx.x;
            for (int g = 0; g < RPP; ++g) { t0[nt] += part[0][g][c]; t1[nt] += part[1][g][c]; }
        }
    }
    if (a.want_stats) stat_flush<NT>(t0, t1, 0, a.nout, a.fin, red);
}

// backward through one layer.  SRC 1: dy [rows, kin] dense (the last layer, or a later K-pass of an already converted buffer);
// SRC 2: dy = BatchNorm + LeakyReLU backward of dbuf (gradient wrt the layer's ACTIVATED output), formed on load and stored
// back in place.  dx [rows, nout] = dy W (nullable: conversion only); epilogue: BatchNorm-backward sums of the layer that
// produced this layer's input (pre-BN values xpre, constants aff_prev), when that layer has one.
struct BnlBwdArgs {
    const float* dy;
    float* dbuf; const float* ypre; const float* aff; const float* coef;
    int kin; float slope;
    const float* W; int ldw;                   // W[c * ldw + u], c < kin, u < nout
    float* dx; int nout;
    const float* xpre; const float* aff_prev; int want_stats;
    int rows, ntiles;
    StatFin fin;
};

template <int NT, int SRC>
__global__ __launch_bounds__(256) void bnl_bwd_kernel(BnlBwdArgs a) {
    extern __shared__ float lds[];
    __shared__ float red[8 * STAT_W];
    const int kin16 = (a.kin + 15) & ~15, kp = kin16 + 4, KS = kin16 / 16;
    float* Wt = lds;                                   // Wt[u][c]
    float* cf = lds + NT * 16 * kp;                    // SRC 2: [6][kin16]
    if (a.dx)
        for (int c = threadIdx.x >> 4; c < kin16; c += 16) {
            float v[NT];
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int u = (threadIdx.x & 15) + 16 * k;
                v[k] = (c < a.kin && u < a.nout) ? a.W[(size_t)c * a.ldw + u] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < NT; ++k) Wt[((threadIdx.x & 15) + 16 * k) * kp + c] = v[k];
        }
    if (SRC == 2) {
        for (int i = threadIdx.x; i < kin16; i += 256) {
            const bool ok = i < a.kin;
#pragma unroll
            for (int w = 0; w < 4; ++w) cf[w * kin16 + i] = ok ? a.aff[w * a.kin + i] : 0.f;
            cf[4 * kin16 + i] = ok ? a.coef[i] : 0.f;
            cf[5 * kin16 + i] = ok ? a.coef[a.kin + i] : 0.f;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane & 15, q = lane >> 4;
    float s0[NT], s1[NT], ssc[NT], ssh[NT], smu[NT], srs[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s0[nt] = s1[nt] = 0.f;
        const int col = nt * 16 + row;
        const bool ok = a.want_stats && col < a.nout;
        ssc[nt] = ok ? a.aff_prev[col] : 0.f;
        ssh[nt] = ok ? a.aff_prev[a.nout + col] : 0.f;
        smu[nt] = ok ? a.aff_prev[2 * a.nout + col] : 0.f;
        srs[nt] = ok ? a.aff_prev[3 * a.nout + col] : 0.f;
    }
    for (int tile = blockIdx.x * 4 + wave; tile < a.ntiles; tile += gridDim.x * 4) {
        const int r0 = tile * 16;
        const int rr = min(r0 + row, a.rows - 1);
        const bool rok = r0 + row < a.rows;
        f4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = pf_splat(0.f);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks < KS) {
                const int c = ks * 16 + 4 * q;
                f4 av = pf_splat(0.f);
                if (c < a.kin && rok) {
                    if (SRC == 1) av = *reinterpret_cast<const f4*>(a.dy + (size_t)rr * a.kin + c);
                    else {
                        float* dp = a.dbuf + (size_t)rr * a.kin + c;
                        const f4 d = *reinterpret_cast<const f4*>(dp);
                        const f4 y = *reinterpret_cast<const f4*>(a.ypre + (size_t)rr * a.kin + c);
                        const f4 sc = *reinterpret_cast<const f4*>(cf + c), sh = *reinterpret_cast<const f4*>(cf + kin16 + c);
                        const f4 mu = *reinterpret_cast<const f4*>(cf + 2 * kin16 + c), rs = *reinterpret_cast<const f4*>(cf + 3 * kin16 + c);
                        const f4 m1 = *reinterpret_cast<const f4*>(cf + 4 * kin16 + c), m2 = *reinterpret_cast<const f4*>(cf + 5 * kin16 + c);
                        const f4 z = y * sc + sh;
                        const f4 xh = (y - mu) * rs;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const float dz = d[w] * (z[w] > 0.f ? 1.f : a.slope);
                            av[w] = sc[w] * (dz - m1[w] - xh[w] * m2[w]);
                        }
                        *reinterpret_cast<f4*>(dp) = av;
                    }
                }
                if (a.dx)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = mfma4(av, *reinterpret_cast<const f4*>(Wt + (nt * 16 + row) * kp + c), acc[nt]);
            }
        }
        if (a.dx)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 16 + row;
                if (col < a.nout) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rw = r0 + 4 * q + r;
                        if (rw < a.rows) {
                            const float v = acc[nt][r];
                            a.dx[(size_t)rw * a.nout + col] = v;
                            if (a.want_stats) {
                                const float y = a.xpre[(size_t)rw * a.nout + col];
                                const float dz = v * (fmaf(y, ssc[nt], ssh[nt]) > 0.f ? 1.f : a.slope);
                                s0[nt] += dz;
                                s1[nt] = fmaf(dz, (y - smu[nt]) * srs[nt], s1[nt]);
                            }
                        }
                    }
                }
            }
    }
    if (a.want_stats) stat_flush<NT>(s0, s1, 0, a.nout, a.fin, red);
}

// part[chunk][c][u] = sum over the chunk's rows of dy[row, c] * act(X[row, u]) (c < RA, u < RB), bpart[chunk][c] = sum dy
constexpr int BNL_EB = 32;                          // rows per staged block of bnl_dw_kernel
struct BnlDwArgs {
    const float* dy; int RA;                   // [rows, RA]
    const float* X; int ldx, RB;               // [rows, ldx], RB columns used
    const float* sc; const float* sh; float slope;
    int rows, chunk;
    float* part; float* bpart;                 // [nchunk][RA16][RB16], [nchunk][RA16]
};
// one staged 32-row block for a wave that owns NS output tiles; SAME: consecutive row tiles of ONE column tile (one B read per
// k step).  Every operand read of a k step is issued before its MFMAs, through one LDS address per tile with the k step as an
// immediate offset (round 5: see mlp_dw_kernel in train_mlp.hip - with lane-dependent tile lists and run-time LDS strides hipcc
// kept an address register per (k step, tile) read and waited for one LDS read per MFMA)
constexpr int BNL_LD = 144;                         // LDS row stride of the staged blocks: >= 128 columns, = 16 (mod 32) floats
constexpr int BNL_DW_WAVES = 8, BNL_DW_T = 64 * BNL_DW_WAVES, BNL_DW_SLOTS = 8;     // <= 64 output tiles over 8 waves
template <int NS, bool SAME>
__device__ __forceinline__ void bnl_dw_block(const float* ar0, const float* br0, const int (&rts)[BNL_DW_SLOTS],
                                             const int (&cts)[BNL_DW_SLOTS], f4 (&acc)[BNL_DW_SLOTS]) {
    const float* ap[NS];
    const float* bp[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { ap[s] = ar0 + rts[s] * 16; bp[s] = br0 + cts[SAME ? 0 : s] * 16; }
#pragma unroll
    for (int ks = 0; ks < BNL_EB / 4; ++ks) {
        float av[NS], bv[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            av[s] = ap[s][4 * ks * BNL_LD];
            bv[s] = (SAME && s > 0) ? bv[0] : bp[s][4 * ks * BNL_LD];
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] = pf_mfma(av[s], bv[s], acc[s]);
    }
}
__global__ __launch_bounds__(BNL_DW_T) void bnl_dw_kernel(BnlDwArgs a) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), row = lane & 15, q = lane >> 4;
    const int RA = (a.RA + 15) & ~15, RB = (a.RB + 15) & ~15;
    constexpr int lda = BNL_LD, ldb = BNL_LD;
    float* As = lds;
    float* Bs = lds + BNL_EB * lda;
    float* Sc = Bs + BNL_EB * ldb;                       // [2][128]: BatchNorm scale / shift of the input columns (1 / 0 without, 0 / 0 beyond RB)
    const int NT = RB / 16, NRT = RA / 16;
    // a wave's tiles: consecutive ids in column-major order (id = column tile x NRT + row tile)
    const int nper = (NRT * NT + BNL_DW_WAVES - 1) / BNL_DW_WAVES;
    const int ns = min(nper, max(0, NRT * NT - wave * nper));
    int rts[BNL_DW_SLOTS], cts[BNL_DW_SLOTS];
    bool same = true;
#pragma unroll
    for (int s = 0; s < BNL_DW_SLOTS; ++s) {
        const int id = wave * nper + s;
        const bool v = s < ns;
        cts[s] = v ? id / NRT : 0; rts[s] = v ? id - cts[s] * NRT : 0;
        if (v && cts[s] != cts[0]) same = false;
    }
    f4 acc[BNL_DW_SLOTS];
#pragma unroll
    for (int s = 0; s < BNL_DW_SLOTS; ++s) acc[s] = pf_splat(0.f);
    const int r_lo = blockIdx.x * a.chunk, r_hi = min(a.rows, r_lo + a.chunk);
    const int ra4 = RA / 4, rb4 = RB / 4;
    constexpr int UN = (BNL_EB * 32 + BNL_DW_T - 1) / BNL_DW_T;
    int elA[UN], cA[UN], elB[UN], cB[UN];
#pragma unroll
    for (int n = 0; n < UN; ++n) {
        const int k = threadIdx.x + BNL_DW_T * n;
        elA[n] = k / ra4; cA[n] = (k - elA[n] * ra4) * 4;
        elB[n] = k / rb4; cB[n] = (k - elB[n] * rb4) * 4;
    }
    const bool veca = (a.RA & 3) == 0, vecb = (a.RB & 3) == 0 && (a.ldx & 3) == 0;
    const float slope = a.sc ? a.slope : 1.f;
    for (int i = threadIdx.x; i < BNL_EB * (lda + ldb); i += BNL_DW_T) As[i] = 0.f;      // padding columns: never written again
    if (threadIdx.x < 128) {
        const int c = threadIdx.x;
        Sc[c] = c < a.RB ? (a.sc ? a.sc[c] : 1.f) : 0.f;
        Sc[128 + c] = (c < a.RB && a.sc) ? a.sh[c] : 0.f;
    }
    // every load is issued unconditionally through an address that is valid even when the unit is not (then replaced by zero)
    f4 ra[UN], rbx[UN];
    auto fetch = [&](int rb) {
#pragma unroll
        for (int n = 0; n < UN; ++n) {
            const int r = rb + elA[n], c = cA[n];
            const bool ok = elA[n] < BNL_EB && r < r_hi && c < a.RA;
            const float* ptr = ok ? a.dy + (size_t)r * a.RA + c : a.dy;
            f4 v;
            if (veca) v = *reinterpret_cast<const f4*>(ptr);
            else {
#pragma unroll
                for (int w = 0; w < 4; ++w) { const bool okw = ok && c + w < a.RA; const float x = ptr[okw ? w : 0]; v[w] = okw ? x : 0.f; }
            }
            ra[n] = ok ? v : pf_splat(0.f);
        }
#pragma unroll
        for (int n = 0; n < UN; ++n) {
            const int r = rb + elB[n], c = cB[n];
            const bool ok = elB[n] < BNL_EB && r < r_hi && c < a.RB;
            const float* ptr = ok ? a.X + (size_t)r * a.ldx + c : a.X;
            f4 v;
            if (vecb) v = *reinterpret_cast<const f4*>(ptr);
            else {
#pragma unroll
                for (int w = 0; w < 4; ++w) { const bool okw = ok && c + w < a.RB; const float x = ptr[okw ? w : 0]; v[w] = okw ? x : 0.f; }
            }
            rbx[n] = ok ? v : pf_splat(0.f);
        }
    };
    const bool bpow = (RA & (RA - 1)) == 0;              // RA = 16 .. 128: thread (column, row group) sums its rows of every block
    const int bcol = threadIdx.x & (RA - 1), bgrp = threadIdx.x / RA, brows = bpow ? BNL_EB / (BNL_DW_T / RA) : 0;
    float bsum = 0.f;
    fetch(r_lo);
    for (int rb = r_lo; rb < r_hi; rb += BNL_EB) {
        __syncthreads();
#pragma unroll
        for (int n = 0; n < UN; ++n) {
            if (elA[n] < BNL_EB) *reinterpret_cast<f4*>(As + elA[n] * lda + cA[n]) = ra[n];
            if (elB[n] < BNL_EB) {
                const f4 s1 = *reinterpret_cast<const f4*>(Sc + cB[n]), s2 = *reinterpret_cast<const f4*>(Sc + 128 + cB[n]);
                *reinterpret_cast<f4*>(Bs + elB[n] * ldb + cB[n]) = lrelu4(rbx[n] * s1 + s2, slope);
            }
        }
        __syncthreads();
        if (rb + BNL_EB < r_hi) fetch(rb + BNL_EB);
        if (bpow) {
            for (int e = 0; e < brows; ++e) bsum += As[(bgrp * brows + e) * lda + bcol];
        } else if ((int)threadIdx.x < RA) {
#pragma unroll 8
            for (int el = 0; el < BNL_EB; ++el) bsum += As[el * lda + threadIdx.x];
        }
        const float* ar0 = As + q * lda + row;
        const float* br0 = Bs + q * ldb + row;
        switch (same ? ns : -ns) {
#define PF_BNLB(NS)                                                                 \
            case NS: bnl_dw_block<NS, true>(ar0, br0, rts, cts, acc); break;        \
            case -NS: bnl_dw_block<NS, false>(ar0, br0, rts, cts, acc); break;
            PF_BNLB(1) PF_BNLB(2) PF_BNLB(3) PF_BNLB(4) PF_BNLB(5) PF_BNLB(6) PF_BNLB(7) PF_BNLB(8)
#undef PF_BNLB
            default: break;
        }
    }
    if (bpow) {                                          // the row groups' bias sums, added in group order
        __syncthreads();
        if (brows > 0) As[bgrp * lda + bcol] = bsum;
        __syncthreads();
        bsum = 0.f;
        if ((int)threadIdx.x < RA)
            for (int gI = 0; gI < BNL_DW_T / RA; ++gI) bsum += As[gI * lda + threadIdx.x];
    }
    if ((int)threadIdx.x < RA) a.bpart[(size_t)blockIdx.x * RA + threadIdx.x] = bsum;
    float* out = a.part + (size_t)blockIdx.x * RA * RB;
#pragma unroll
    for (int s = 0; s < BNL_DW_SLOTS; ++s)
        if (s < ns)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(rts[s] * 16 + 4 * q + r) * RB + cts[s] * 16 + row] = acc[s][r];
}

// dW[c * ldw + coff + u] = sum_chunks part[k][c][u] (c < RA, u < RB); db[c] = sum_chunks bpart[k][c] (db nullable)
__global__ __launch_bounds__(256) void bnl_reduce_kernel(const float* part, const float* bpart, int nchunk, int RA, int RB, int RA16,
                                                         int RB16, float* dW, int ldw, int coff, float* db) {
    __shared__ double shr[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    const int total = RA * (RB + 1);
    const bool ok = i < total;
    const int c = ok ? i / (RB + 1) : 0, u = ok ? i % (RB + 1) : 0;
    double s = 0.0;
    if (ok) {
        if (u == RB) { for (int k = ty; k < nchunk; k += 4) s += (double)bpart[(size_t)k * RA16 + c]; }
        else {
            int k = ty;
            for (; k + 12 < nchunk; k += 16) {                        // four loads in flight per thread
                const float v0 = part[((size_t)k * RA16 + c) * RB16 + u], v1 = part[((size_t)(k + 4) * RA16 + c) * RB16 + u];
                const float v2 = part[((size_t)(k + 8) * RA16 + c) * RB16 + u], v3 = part[((size_t)(k + 12) * RA16 + c) * RB16 + u];
                s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
            }
            for (; k < nchunk; k += 4) s += (double)part[((size_t)k * RA16 + c) * RB16 + u];
        }
    }
    shr[ty][tx] = s;
    __syncthreads();
    if (ty != 0 || !ok) return;
    s = (shr[0][tx] + shr[1][tx]) + (shr[2][tx] + shr[3][tx]);
    if (u == RB) { if (db) db[c] = (float)s; }
    else dW[(size_t)c * ldw + coff + u] = (float)s;
}

#ifndef PF_BNL_CHUNK
#define PF_BNL_CHUNK 256
#endif
constexpr int BNL_CHUNK = PF_BNL_CHUNK;

int bnl_check(const PfBnMlpTrain* p) {
    if (!p) return PF_ERR_NULL;
    if (p->rows < 16 || (p->nl != 2 && p->nl != 3)) return PF_ERR_SHAPE;       // (rows >= 2 also keeps the unbiased-variance factor R / (R - 1) finite)
    if (p->kin0a < 1 || p->kin0a > 128 || p->kin0b < 0 || p->kin0b > 128) return PF_ERR_UNSUPPORTED;
    if (p->kin0b > 0 && (p->kin0a & 3)) return PF_ERR_UNSUPPORTED;
    for (int l = 0; l < p->nl; ++l)
        if (p->width[l] < 16 || p->width[l] > 128 || p->width[l] % 16 != 0) return PF_ERR_UNSUPPORTED;
    const bool sum_in = (p->flags & PF_BNMLP_SUM_INPUTS) != 0;
    if (sum_in && (p->nl < 2 || p->kin0a != p->width[0] || p->kin0b != p->width[0] || !p->xb)) return PF_ERR_SHAPE;
    for (int l = 0; l < p->nl; ++l)
        if ((!p->W[l] && !(sum_in && l == 0)) || !p->y[l]) return PF_ERR_NULL;
    for (int l = 0; l < p->nl - 1; ++l)
        if (!p->gamma[l] || !p->beta[l] || !p->aff[l]) return PF_ERR_NULL;
    if (!p->xa || (p->kin0b > 0 && !p->xb) || !p->stat) return PF_ERR_NULL;
    return PF_OK;
}
inline int bnl_nt(int w) { const int n = (w + 15) / 16; return n <= 1 ? 1 : (n <= 2 ? 2 : (n <= 4 ? 4 : 8)); }

template <int NT>
void bnl_fwd_launch(const BnlFwdArgs& a, int grid, hipStream_t s) {
    const int kin16 = (a.kin + 15) & ~15;
    const size_t lds = sizeof(float) * ((size_t)NT * 16 * (kin16 + 4) + 2 * kin16);
    if (a.addA) {
        if (a.nout == 16 * NT && NT >= 1) {                               // the streaming form (full 16-column blocks)
            hipLaunchKernelGGL(bnl_sum_kernel<NT>, dim3(grid), dim3(256), 0, s, a);
            return;
        }
        allow_lds((bnl_fwd_kernel<NT, true>), lds);
        hipLaunchKernelGGL((bnl_fwd_kernel<NT, true>), dim3(grid), dim3(256), lds, s, a);
        return;
    }
    allow_lds((bnl_fwd_kernel<NT, false>), lds);
    hipLaunchKernelGGL((bnl_fwd_kernel<NT, false>), dim3(grid), dim3(256), lds, s, a);
}
void bnl_fwd_dispatch(const BnlFwdArgs& a, int grid, hipStream_t s) {
    switch (bnl_nt(a.nout)) {
        case 1: bnl_fwd_launch<1>(a, grid, s); break;
        case 2: bnl_fwd_launch<2>(a, grid, s); break;
        case 4: bnl_fwd_launch<4>(a, grid, s); break;
        default: bnl_fwd_launch<8>(a, grid, s); break;
    }
}
template <int NT, int SRC>
void bnl_bwd_launch(const BnlBwdArgs& a, int grid, hipStream_t s) {
    const int kin16 = (a.kin + 15) & ~15;
    const size_t lds = sizeof(float) * ((size_t)NT * 16 * (kin16 + 4) + 6 * kin16);
    allow_lds(bnl_bwd_kernel<NT, SRC>, lds);
    hipLaunchKernelGGL((bnl_bwd_kernel<NT, SRC>), dim3(grid), dim3(256), lds, s, a);
}
void bnl_bwd_dispatch(const BnlBwdArgs& a, int src, int grid, hipStream_t s) {
    const int nt = a.dx ? bnl_nt(a.nout) : 1;
#define PF_BNLB(NT) do { if (src == 1) bnl_bwd_launch<NT, 1>(a, grid, s); else bnl_bwd_launch<NT, 2>(a, grid, s); } while (0)
    switch (nt) {
        case 1: PF_BNLB(1); break;
        case 2: PF_BNLB(2); break;
        case 4: PF_BNLB(4); break;
        default: PF_BNLB(8); break;
    }
#undef PF_BNLB
}

}  // namespace

extern "C" long long pf_bnmlp_train_ws_floats(const PfBnMlpTrain* p) {
    if (!p || p->rows < 16) return -1;
    const long long nchunk = (p->rows + BNL_CHUNK - 1) / BNL_CHUNK;
    return nchunk * (128ll * 128 + 128);
}

extern "C" int pf_bnmlp_train_fwd(const PfBnMlpTrain* p, void* stream) {
    int st = bnl_check(p);
    if (st) return st;
    hipStream_t s = (hipStream_t)stream;
    const int ntiles = (p->rows + 15) / 16;
    const int grid = (ntiles + 3) / 4 < EC_GRID ? (ntiles + 3) / 4 : EC_GRID;
    const int in0 = p->kin0a + p->kin0b;
    for (int l = 0; l < p->nl; ++l) {
        const bool bn = l < p->nl - 1;
        BnlFwdArgs a{};
        a.slope = p->slope; a.out = p->y[l]; a.nout = p->width[l]; a.rows = p->rows; a.ntiles = ntiles;
        if (bn) a.fin = StatFin{p->stat, 1, p->width[l], 0, p->width[l], p->aff[l], p->gamma[l], p->beta[l], p->run_mean[l],
                                p->run_var[l], p->eps, p->momentum, nullptr, nullptr, nullptr, (double)p->rows, p->sync_sums};
        if (bn) a.fin.det = PF_DET(p);
        if (l == 0 && (p->flags & PF_BNMLP_SUM_INPUTS)) {
            // layer 0 is NOT a product: its pre-BatchNorm output is the sum of the two inputs (their producers' last linear layers
            // carry this layer's weights folded in - train_ops.py interp_weights): y[0] = xa + xb, statistics as usual
            a.addA = p->xa; a.addB = p->xb; a.X = p->xa; a.ldx = p->kin0a; a.kin = 16; a.W = nullptr; a.ldw = 0; a.bias = nullptr;
            a.want_stats = bn;
            bnl_fwd_dispatch(a, grid, s);
        } else if (l == 0) {
            a.X = p->xa; a.ldx = p->kin0a; a.kin = p->kin0a; a.W = p->W[0]; a.ldw = in0; a.bias = p->b[0];
            a.want_stats = bn && p->kin0b == 0;
            bnl_fwd_dispatch(a, grid, s);
            if (p->kin0b > 0) {
                a.X = p->xb; a.ldx = p->kin0b; a.kin = p->kin0b; a.W = p->W[0] + p->kin0a; a.bias = nullptr; a.accum = 1;
                a.want_stats = bn;
                bnl_fwd_dispatch(a, grid, s);
            }
        } else {
            a.X = p->y[l - 1]; a.ldx = p->width[l - 1]; a.kin = p->width[l - 1];
            a.sc = p->aff[l - 1]; a.sh = p->aff[l - 1] + p->width[l - 1];
            a.W = p->W[l]; a.ldw = p->width[l - 1]; a.bias = p->b[l]; a.want_stats = bn;
            bnl_fwd_dispatch(a, grid, s);
        }
        if (bn && (st = stat_sync(a.fin, p->width[l], p->sync_cb, p->sync_user, s))) return st;   // SyncBN: global statistics
    }
    return pf_last_launch_status();
}

extern "C" int pf_bnmlp_train_bwd(const PfBnMlpTrain* p, void* stream) {
    int st = bnl_check(p);
    if (st) return st;
    if (!p->dout || !p->ws) return PF_ERR_NULL;
    for (int l = 0; l < p->nl; ++l)
        if (!p->dW[l] && !((p->flags & PF_BNMLP_SUM_INPUTS) && l == 0)) return PF_ERR_NULL;
    for (int l = 0; l < p->nl - 1; ++l)
        if (!p->d[l] || !p->coef[l] || !p->dgamma[l] || !p->dbeta[l]) return PF_ERR_NULL;
    if (p->ws_floats < pf_bnmlp_train_ws_floats(p)) return PF_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int ntiles = (p->rows + 15) / 16;
    const int grid = (ntiles + 3) / 4 < EC_GRID ? (ntiles + 3) / 4 : EC_GRID;
    const int nchunk = (p->rows + BNL_CHUNK - 1) / BNL_CHUNK;
    float* part = p->ws;
    float* bpart = p->ws + (size_t)nchunk * 128 * 128;
    const int in0 = p->kin0a + p->kin0b;
    auto dw = [&](const float* dy, int RA, const float* X, int ldx, int RB, const float* sc, const float* sh, float* dW, int ldw,
                  int coff, float* db) {
        const int RA16 = (RA + 15) & ~15, RB16 = (RB + 15) & ~15;
        BnlDwArgs a{dy, RA, X, ldx, RB, sc, sh, p->slope, p->rows, BNL_CHUNK, part, bpart};
        const size_t lds = sizeof(float) * ((size_t)BNL_EB * 2 * BNL_LD + 256);
        (void)RA16; (void)RB16;
        hipLaunchKernelGGL(bnl_dw_kernel, dim3(nchunk), dim3(BNL_DW_T), lds, s, a);
        const int total = RA * (RB + 1);
        hipLaunchKernelGGL(bnl_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, s, part, bpart, nchunk, RA, RB, RA16, RB16, dW,
                           ldw, coff, db);
    };
    for (int l = p->nl - 1; l >= 0; --l) {
        const bool bn = l < p->nl - 1;
        const float* dyl = bn ? p->d[l] : p->dout;          // after the kernel below: the gradient wrt this layer's pre-BN output
        BnlBwdArgs a{};
        a.kin = p->width[l]; a.slope = p->slope; a.rows = p->rows; a.ntiles = ntiles;
        if (bn) { a.dbuf = p->d[l]; a.ypre = p->y[l]; a.aff = p->aff[l]; a.coef = p->coef[l]; }
        else a.dy = p->dout;
        if (l > 0) {
            a.W = p->W[l]; a.ldw = p->width[l - 1]; a.dx = p->d[l - 1]; a.nout = p->width[l - 1];
            a.xpre = p->y[l - 1]; a.aff_prev = p->aff[l - 1]; a.want_stats = 1;
            a.fin = StatFin{p->stat, 2, p->width[l - 1], 0, p->width[l - 1], nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f,
                            p->coef[l - 1], p->dgamma[l - 1], p->dbeta[l - 1], (double)p->rows, p->sync_sums};
            a.fin.det = PF_DET(p);
            bnl_bwd_dispatch(a, bn ? 2 : 1, grid, s);
            if ((st = stat_sync(a.fin, p->width[l - 1], p->sync_cb, p->sync_user, s))) return st;   // SyncBN: global sums of layer l - 1
            dw(dyl, p->width[l], p->y[l - 1], p->width[l - 1], p->width[l - 1], p->aff[l - 1], p->aff[l - 1] + p->width[l - 1],
               p->dW[l], p->width[l - 1], 0, p->db[l]);
        } else if (p->flags & PF_BNMLP_SUM_INPUTS) {
            // y[0] = xa + xb: the gradient of both inputs is d[0] after its BatchNorm backward (converted in place); no weights
            a.W = nullptr; a.ldw = 0; a.dx = nullptr; a.nout = 0;
            if (bn) bnl_bwd_dispatch(a, 2, grid, s);
        } else {
            // first layer: one pass per input tensor; the first pass also converts d[0] in place
            a.W = p->W[0]; a.ldw = in0; a.dx = p->dxa; a.nout = p->kin0a;
            if (bn || p->dxa) bnl_bwd_dispatch(a, bn ? 2 : 1, grid, s);
            if (p->kin0b > 0 && p->dxb) {
                BnlBwdArgs b2 = a;
                b2.dy = dyl; b2.dbuf = nullptr; b2.W = p->W[0] + p->kin0a; b2.dx = p->dxb; b2.nout = p->kin0b;
                bnl_bwd_dispatch(b2, 1, grid, s);
            }
            dw(dyl, p->width[0], p->xa, p->kin0a, p->kin0a, nullptr, nullptr, p->dW[0], in0, 0, p->db[0]);
            if (p->kin0b > 0) dw(dyl, p->width[0], p->xb, p->kin0b, p->kin0b, nullptr, nullptr, p->dW[0], in0, p->kin0a, nullptr);
        }
    }
    return pf_last_launch_status();
}
