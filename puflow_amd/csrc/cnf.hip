// Continuous (CNF) variant of the flow blocks: the ODE right-hand side and the Runge-Kutta
// bookkeeping of an adaptive Dormand-Prince solver.  Replaces ODEfunc.forward + ODEnet + ConcatSquashLinear
// (modules/continuous/odefunc.py:60-148, diffeq_layers.py:72-86) and the arithmetic of torchdiffeq's dopri5
// stages (called from modules/continuous/cnf.py:97-113).  The step-size CONTROL (a handful of scalars per step)
// stays on the host: puflow_amd/cnf.py.
//
// State rows are [y0 y1 y2 logp]; one MFMA column tile = 16 rows.  A ConcatSquash layer is
//     out = (W x + b) * sigmoid(Wg [t; c] + bg) + Wb [t; c]
// and everything that depends on the context c only is precomputed per ORIGINAL point by a GEMM
// (ctx = Hc c + hb, 288 floats per point and block), so a right-hand-side evaluation is
//     3 -> 64 (VALU) -> tanh -> 64 -> 64 (split-fp16 MFMA) -> tanh -> 64 -> 3 (MFMA)
// plus the Hutchinson term  e^T (d f / d y) e  that the reference obtains with autograd (odefunc.py:9-31):
// here the vector-Jacobian product is written out (W3^T, tanh', W2^T as a second MFMA image, tanh', W1^T).
//
// Weight record of one block (floats; packing.pack_cnf_record), resident in LDS:
//   [0,4096)      f16x2 image of W2          [4096,8192)  f16x2 image of W2^T
//   [8192,9216)   f16x2 image of W3 (3 rows replicated into every 4-row q group)
//   [9216,9472)   [W1 | b1]  [64][4]         [9472,9728)  W3^T [64][4] (3 used, 4th zero)
//   [9728,9792) b1   [9792,9856) b2   [9856,9872) b3 (replicated)   [9872,10160) time coefficients, ctx layout
// ctx layout (per point, 288 floats): gate1[64] bias1[64] gate2[64] bias2[64] gate3[16, replicated] bias3[16, replicated]
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"
#include "pf_mfma.h"

namespace {

constexpr int CNF_REC = 10160;
constexpr int CNF_CTX = 288;
#ifndef PF_CNF_SPLIT_FWD
#define PF_CNF_SPLIT_FWD 1           // 0: A/B builds with the factored gates in the inverse pass only
#endif
#ifndef PF_CNF_NW
#define PF_CNF_NW 4
#endif
#ifndef PF_CNF_WPE
#define PF_CNF_WPE 2                      // waves per SIMD the step kernels are compiled for (register budget 512 / WPE)
#endif
constexpr int CNF_NW = PF_CNF_NW;         // waves per workgroup

constexpr int CTL_T = 0, CTL_DT = 1, CTL_T1 = 2, CTL_NTOT = 3, CTL_CUR = 4, CTL_DONE = 5, CTL_ACC = 6, CTL_REJ = 7, CTL_NFE = 8,
              CTL_STATUS = 9, CTL_REV = 10, CTL_H0 = 11, CTL_D1 = 12,    // device-side dopri5 state, see cnf_ctl_update
              CTL_ARRIVE = 13;                                           // (as a 32-bit word) workgroups of the running attempt that are done

struct CnfArgs {
    const float* y0;        // [rows,4] state at the start of the step
    const float* k;         // [7][rows][4] stage derivatives
    float coef[6];          // yi = y0 + h * sum_j coef[j] k[j], j < ncoef
    int ncoef;
    float h;
    float t;                // time the net sees
    float sgn;              // -1: reversed integration (returns -f)
    const float* ctx;       // [T, 288]
    const float* e;         // [T, 3]
    const float* rec;       // CNF_REC floats
    float* kout;            // [rows,4]
    float* yout;            // nullable: the stage state yi (the last stage's is the step's solution)
    int rows, R, ntiles;
    const double* ctl;      // nullable: take h = ctl[CTL_H0] and t = +-(ctl[CTL_T] + h) from the device (initial-step probe)
};

// sigmoid and tanh on the hardware exp / rcp (1 ulp each): absolute error ~2e-7, against ~25 instructions for tanhf and a
// full-precision division - the right-hand side is bound by these (32 tanh + 32 sigmoid per lane and evaluation), not by its
// 54 MFMAs.  tanh(x) = 1 - 2 / (e^{2x} + 1) saturates correctly (e -> inf: 1, e -> 0: -1).
// The arguments arrive PRESCALED by the host (packing.pack_cnf_block): gates carry -log2e x, the tanh layers' pre-activations
// 2 log2e x, so each function is v_exp_f32 + add + v_rcp_f32 (+ one fma): one multiply per gate and per tanh saved.
__device__ __forceinline__ float sigm(float xs) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(xs)); }          // xs = -log2e x
__device__ __forceinline__ float tanh_fast(float xs) { return fmaf(-2.f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(xs) + 1.f), 1.f); }   // xs = 2 log2e x

// One evaluation for this lane's row: k = sgn * (f(t, y), -e^T (df/dy) e).  All four q groups of a column return the same f4.
struct CnfW {
    const float* rec;          // LDS copy of the record
    PfW2Lds w2, w2t, w3;
};

// SPLIT (cnf_step_dev_kernel, context rows in LDS): the gates depend on (point, channel, stage time) only, and
// 2^(gt (t + alpha h) + gc) = 2^(gt t + gc) x 2^(gt alpha h).  The first factor is computed ONCE per point and tile into the
// context rows' gate slots (shared by the point's R rows and the step's six evaluations), the second once per launch into a
// [stage][channel] table `tvg` points at: a gate is fma + v_rcp_f32, without the v_exp_f32 (36 of an evaluation's 135
// quarter-rate transcendentals, on a kernel the PMC counters show VALU-bound).
// COMPACT (the forward pass's step kernel, R = 1: 64 points per tile, whose full context rows would not fit in LDS): cx / tvg hold
// only the 144 gate columns, layer after layer (gate 1 at 0, gate 2 at 64, gate 3 at 128); the bias columns come from cxb (the
// row in global memory, the full 288-column layout).
template <bool SPLIT = false, bool COMPACT = false>
__device__ __forceinline__ f4 cnf_eval(const CnfW& w, int q, f4 y, float t, float sgn, const float* __restrict__ cx,
                                       float e0, float e1, float e2, const float* __restrict__ tvg = nullptr,
                                       const float* __restrict__ cxb = nullptr) {
    const float* rec = w.rec;
    const float* tv = rec + 9872;
    if (!SPLIT) tvg = tv;
    if (!COMPACT) cxb = cx;
    constexpr int G2 = COMPACT ? 64 : 128, G3 = COMPACT ? 128 : 256;
    auto gatef = [](float gt, float tt, float gc) {
        return SPLIT ? __builtin_amdgcn_rcpf(fmaf(gc, gt, 1.f)) : sigm(fmaf(gt, tt, gc));
    };
    const int col = threadIdx.x & 15;
    // ---- layer 1 (3 -> 64): this lane's channels 16 cb + 4 q + r.  W1 y + b1 is ONE v_mfma_f32_16x16x4_f32 per 16 channels
    // ([W1 | b1] x [y; 1]: k = 3 carries the bias) instead of 16 x (an LDS row + 3 fmas) on a VALU that PMC shows ~90 % busy
    // (profiles/r4_cnf/: 50 M VALU wave-instructions per step launch on 1024 SIMDs)
    const float yb = q == 0 ? y.x : (q == 1 ? y.y : (q == 2 ? y.z : 1.f));
    f4 h1[1][4], g1[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const int ch = cb * 16 + 4 * q;
        const f4 gc = *reinterpret_cast<const f4*>(cx + ch), bc = *reinterpret_cast<const f4*>(cxb + 64 + ch);
        const f4 gt = *reinterpret_cast<const f4*>(tvg + ch), bt = *reinterpret_cast<const f4*>(tv + 64 + ch);
        const f4 lin4 = pf_mfma(rec[9216 + (cb * 16 + col) * 4 + q], yb, pf_splat(0.f));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lin = lin4[r];
            const float gate = gatef(gt[r], t, gc[r]);
            g1[cb][r] = gate;
            h1[0][cb][r] = tanh_fast(fmaf(lin, gate, fmaf(bt[r], t, bc[r])));
        }
    }
    // ---- layer 2 (64 -> 64)
    f4 h2[1][4], g2[4];
    {
        PfPair2 hp[1][2];
        hp[0][0] = pf_pair2(h1[0][0], h1[0][1]);
        hp[0][1] = pf_pair2(h1[0][2], h1[0][3]);
        f4 a2[1][4];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) a2[0][ob] = pf_bias(rec + 9792, ob, q);
        pf_mm2f<4, 2, 2>(w.w2, 0, hp, 0, a2, 0);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int ch = cb * 16 + 4 * q;
            const f4 gc = *reinterpret_cast<const f4*>(cx + G2 + ch), bc = *reinterpret_cast<const f4*>(cxb + 192 + ch);
            const f4 gt = *reinterpret_cast<const f4*>(tvg + G2 + ch), bt = *reinterpret_cast<const f4*>(tv + 192 + ch);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gate = gatef(gt[r], t, gc[r]);
                g2[cb][r] = gate;
                h2[0][cb][r] = tanh_fast(fmaf(a2[0][cb][r], gate, fmaf(bt[r], t, bc[r])));
            }
        }
    }
    // ---- layer 3 (64 -> 3; rows replicated: every q group holds channels 0..2 in .x .y .z)
    f4 dy, g3;
    {
        PfPair2 hp[1][2];
        hp[0][0] = pf_pair2(h2[0][0], h2[0][1]);
        hp[0][1] = pf_pair2(h2[0][2], h2[0][3]);
        f4 a3[1][1];
        a3[0][0] = *reinterpret_cast<const f4*>(rec + 9856 + 4 * q);
        pf_mm2f<1, 2, 2>(w.w3, 0, hp, 0, a3, 0);
        const f4 gc = *reinterpret_cast<const f4*>(cx + G3 + 4 * q), bc = *reinterpret_cast<const f4*>(cxb + 272 + 4 * q);
        const f4 gt = *reinterpret_cast<const f4*>(tvg + G3 + 4 * q), bt = *reinterpret_cast<const f4*>(tv + 272 + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            g3[r] = gatef(gt[r], t, gc[r]);
            dy[r] = fmaf(a3[0][0][r], g3[r], fmaf(bt[r], t, bc[r]));
        }
    }
    // ---- Hutchinson term e^T J e by the vector-Jacobian product of e through the three layers
    const float v0 = e0 * g3.x, v1 = e1 * g3.y, v2 = e2 * g3.z;
    const float vb = q == 0 ? v0 : (q == 1 ? v1 : (q == 2 ? v2 : 0.f));       // W3^T v as one f32 MFMA per 16 channels, like layer 1
    f4 w2v[1][4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const f4 u4 = pf_mfma(rec[9472 + (cb * 16 + col) * 4 + q], vb, pf_splat(0.f));               // W3[:, ch] (4th column zero)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float hh = h2[0][cb][r];
            w2v[0][cb][r] = u4[r] * (1.f - hh * hh) * g2[cb][r];
        }
    }
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    {
        PfPair2 wp[1][2];
        wp[0][0] = pf_pair2(w2v[0][0], w2v[0][1]);
        wp[0][1] = pf_pair2(w2v[0][2], w2v[0][3]);
        f4 u1[1][4];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) u1[0][ob] = pf_splat(0.f);
        pf_mm2f<4, 2, 2>(w.w2t, 0, wp, 0, u1, 0);
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float hh = h1[0][cb][r];
                const float w1v = u1[0][cb][r] * (1.f - hh * hh) * g1[cb][r];
                const f4 wr = *reinterpret_cast<const f4*>(rec + 9216 + (cb * 16 + 4 * q + r) * 4);  // W1[ch, :]
                r0 = fmaf(wr.x, w1v, r0); r1 = fmaf(wr.y, w1v, r1); r2 = fmaf(wr.z, w1v, r2);
            }
    }
    // sum over the 4 q groups of the column (lanes col, col+16, col+32, col+48)
    r0 += __shfl_xor(r0, 16); r1 += __shfl_xor(r1, 16); r2 += __shfl_xor(r2, 16);
    r0 += __shfl_xor(r0, 32); r1 += __shfl_xor(r1, 32); r2 += __shfl_xor(r2, 32);
    // (the W1 rows of the record carry the forward's 2 log2e: taken out of the three sums here)
    const float div = fmaf(r2, e2, fmaf(r1, e1, r0 * e0)) * 0.34657359027997264f;
    return (f4){sgn * dy.x, sgn * dy.y, sgn * dy.z, -sgn * div};
}

__device__ __forceinline__ CnfW cnf_stage_weights(f4* wl, const float* rec_g, int nthreads, int lane) {
    for (int i = threadIdx.x; i < CNF_REC / 4; i += nthreads) wl[i] = reinterpret_cast<const f4*>(rec_g)[i];
    __syncthreads();
    const float* rec = reinterpret_cast<const float*>(wl);
    return CnfW{rec, PfW2Lds{reinterpret_cast<const u4*>(rec), lane}, PfW2Lds{reinterpret_cast<const u4*>(rec + 4096), lane},
                PfW2Lds{reinterpret_cast<const u4*>(rec + 8192), lane}};
}

__global__ __launch_bounds__(CNF_NW * 64) void cnf_rhs_kernel(CnfArgs a) {
    __shared__ f4 wl[CNF_REC / 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    const CnfW w = cnf_stage_weights(wl, a.rec, CNF_NW * 64, lane);
    const size_t kstride = (size_t)a.rows * 4;
    float h = a.h, t = a.t;
    if (a.ctl) {
        const double hd = a.ctl[CTL_H0], td = a.ctl[CTL_T] + hd;
        h = (float)hd;
        t = (float)(a.ctl[CTL_REV] != 0.0 ? -td : td);
    }
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int g = (tile * CNF_NW + wave) * 16 + col;
        const bool ok = g < a.rows;
        const int row = ok ? g : a.rows - 1;
        const int pt = row / a.R;
        f4 y = *reinterpret_cast<const f4*>(a.y0 + (size_t)row * 4);
        for (int j = 0; j < a.ncoef; ++j) {
            const f4 kj = *reinterpret_cast<const f4*>(a.k + j * kstride + (size_t)row * 4);
            y += kj * (h * a.coef[j]);
        }
        if (a.yout && ok && q == 0) *reinterpret_cast<f4*>(a.yout + (size_t)row * 4) = y;
        const float* cx = a.ctx + (size_t)pt * CNF_CTX;
        const f4 k = cnf_eval(w, q, y, t, a.sgn, cx, a.e[(size_t)pt * 3 + 0], a.e[(size_t)pt * 3 + 1],
                              a.e[(size_t)pt * 3 + 2]);
        if (ok && q == 0) *reinterpret_cast<f4*>(a.kout + (size_t)row * 4) = k;
    }
}

// ---- one whole Dormand-Prince step per launch: the six stage evaluations run back to back on register-held stage
// derivatives (the context rows are re-read from L1/L2, not HBM); outputs the step's solution, its FSAL derivative,
// optionally the dense-output mid-point, and this workgroup's share of the scaled error sum.
struct CnfStepArgs {
    const float* y0;        // [rows,4]
    const float* f0;        // [rows,4]  derivative at the start (FSAL)
    const float* ctx;
    const float* e;
    const float* rec;
    float* y1;              // [rows,4]
    float* f1;              // [rows,4]
    float* ymid;            // nullable
    double* partial;        // [gridDim.x]
    float t, h, sgn, tsign; // net time of stage s: tsign * (t + alpha_s h)
    float rtol, atol;
    int rows, R, ntiles;
};

__global__ __launch_bounds__(CNF_NW * 64) void cnf_step_kernel(CnfStepArgs a) {
    __shared__ f4 wl[CNF_REC / 4];
    __shared__ double red[CNF_NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    const CnfW w = cnf_stage_weights(wl, a.rec, CNF_NW * 64, lane);
    constexpr float AL[6] = {1.f / 5, 3.f / 10, 4.f / 5, 8.f / 9, 1.f, 1.f};
    constexpr float BE[6][6] = {
        {1.f / 5, 0, 0, 0, 0, 0},
        {3.f / 40, 9.f / 40, 0, 0, 0, 0},
        {44.f / 45, -56.f / 15, 32.f / 9, 0, 0, 0},
        {19372.f / 6561, -25360.f / 2187, 64448.f / 6561, -212.f / 729, 0, 0},
        {9017.f / 3168, -355.f / 33, 46732.f / 5247, 49.f / 176, -5103.f / 18656, 0},
        {35.f / 384, 0, 500.f / 1113, 125.f / 192, -2187.f / 6784, 11.f / 84}};
    constexpr float CE[7] = {(float)(35. / 384 - 1951. / 21600), 0, (float)(500. / 1113 - 22642. / 50085),
                             (float)(125. / 192 - 451. / 720), (float)(-2187. / 6784 + 12231. / 42400),
                             (float)(11. / 84 - 649. / 6300), (float)(-1. / 60)};
    constexpr float CM[7] = {(float)(6025192743. / 30085553152 / 2), 0, (float)(51252292925. / 65400821598 / 2),
                             (float)(-2691868925. / 45128329728 / 2), (float)(187940372067. / 1594534317056 / 2),
                             (float)(-1776094331. / 19743644256 / 2), (float)(11237099. / 235043384 / 2)};
    double acc = 0.0;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int g = (tile * CNF_NW + wave) * 16 + col;
        const bool ok = g < a.rows;
        const int row = ok ? g : a.rows - 1;
        const int pt = row / a.R;
        const f4 y0 = *reinterpret_cast<const f4*>(a.y0 + (size_t)row * 4);
        const float* cx = a.ctx + (size_t)pt * CNF_CTX;
        const float e0 = a.e[(size_t)pt * 3 + 0], e1 = a.e[(size_t)pt * 3 + 1], e2 = a.e[(size_t)pt * 3 + 2];
        f4 k[7];
        k[0] = *reinterpret_cast<const f4*>(a.f0 + (size_t)row * 4);
        f4 yi = y0;
        pf_static_for<0, 6>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            f4 comb = k[0] * BE[s][0];
#pragma unroll
            for (int j = 1; j <= s; ++j) comb += k[j] * BE[s][j];
            yi = y0 + comb * a.h;
            k[s + 1] = cnf_eval(w, q, yi, a.tsign * (a.t + AL[s] * a.h), a.sgn, cx, e0, e1, e2);
        });
        f4 err = k[0] * CE[0];
#pragma unroll
        for (int j = 1; j < 7; ++j) err += k[j] * CE[j];
        err *= a.h;
        if (ok && q == 0) {
            *reinterpret_cast<f4*>(a.y1 + (size_t)row * 4) = yi;
            *reinterpret_cast<f4*>(a.f1 + (size_t)row * 4) = k[6];
            if (a.ymid) {
                f4 m = k[0] * CM[0];
#pragma unroll
                for (int j = 1; j < 7; ++j) m += k[j] * CM[j];
                *reinterpret_cast<f4*>(a.ymid + (size_t)row * 4) = y0 + m * a.h;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float r = err[c] / (a.atol + a.rtol * fmaxf(fabsf(y0[c]), fabsf(yi[c])));
                acc += (double)r * (double)r;
            }
        }
    }
    // workgroup sum (fixed order): lanes -> wave -> workgroup
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < CNF_NW; ++i) t += red[i];
        a.partial[blockIdx.x] = t;
    }
}

// ---- adaptive step control on the device ------------------------------------------------------------
// The host-side dopri5 loop read the error norm after every step attempt (one device->host sync per attempt, ~60 per
// forward).  Here the controller state lives in device memory and a one-thread kernel takes torchdiffeq's decisions
// (accept / reject, next step size, end-time coverage: the same formulas as oracle/cnf_ref.py::dopri5 and the host loop in
// puflow_amd/cnf.py); the step kernel reads (t, dt, which buffer is current) from that state and is a no-op once the
// integration is done, so the host can enqueue a batch of attempts and look at the state once per batch.
//   ctl (doubles): [0] t  [1] dt  [2] t1  [3] n_tot  [4] cur (0/1)  [5] done  [6] accepted  [7] rejected  [8] nfe
//                  [9] status (0 ok, 1 non-finite error norm, 2 dt underflow)  [10] reverse (0/1)
// The step that covers t1 writes the dense-output value at t1 (quartic through y0, y_mid, y1: torchdiffeq's interpolation)
// straight into `out`; if that attempt is rejected a later covering attempt overwrites it.

// The per-workgroup sums (<= 1024) added in a fixed order by a WHOLE workgroup: four independent loads per thread in flight at
// once, then wave shuffles and four LDS words.  (One wave walking the array in a dependent loop was 16 L2 round trips in a
// row: ~10 us at the tail of every step attempt, ~100 attempts per forward.)  Every thread returns the total.
__device__ __forceinline__ double cnf_partial_total(const double* partial, int nblocks, double* red4) {
    double v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = threadIdx.x + j * CNF_NW * 64;
        v[j] = i < nblocks ? __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
    static_assert(CNF_NW * 64 * 4 >= 1024, "a workgroup covers the largest grid in four loads per thread");
    double sum = ((v[0] + v[1]) + v[2]) + v[3];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) sum += __shfl_xor(sum, m);
    __syncthreads();                                                         // red4 may still be read from an earlier total
    if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = sum;
    __syncthreads();
    double t = red4[0];
#pragma unroll
    for (int w = 1; w < CNF_NW; ++w) t += red4[w];
    return t;
}

// torchdiffeq's `_adaptive_step` decisions on the error norm of the attempt just made (thread 0 decides).  Called by the whole
// workgroup of cnf_step_dev_kernel that finishes LAST (every partial sum is in place by then).
__device__ __forceinline__ void cnf_ctl_update(double* ctl, const double* partial, int nblocks, double* red4) {
    const double sum = cnf_partial_total(partial, nblocks, red4);
    if (threadIdx.x != 0) return;
    const double ratio = sqrt(sum / ctl[CTL_NTOT]);
    const double t = ctl[CTL_T], dt = ctl[CTL_DT], t1 = ctl[CTL_T1];
    ctl[CTL_NFE] += 6.0;
    if (!(ratio == ratio) || ratio > 1.7e308) { ctl[CTL_STATUS] = 1.0; ctl[CTL_DONE] = 1.0; return; }
    if (!(t + dt > t)) { ctl[CTL_STATUS] = 2.0; ctl[CTL_DONE] = 1.0; return; }
    double tn = t;
    if (ratio <= 1.0) {
        ctl[CTL_ACC] += 1.0;
        tn = t + dt;
        ctl[CTL_T] = tn;
        ctl[CTL_CUR] = 1.0 - ctl[CTL_CUR];                       // accepted state; FSAL: f1 becomes the next f0
    } else {
        ctl[CTL_REJ] += 1.0;
    }
    double ndt;
    if (ratio == 0.0) ndt = dt * 10.0;
    else {
        const double dfac = ratio < 1.0 ? 1.0 : 0.2;
        double fac = 0.9 / pow(ratio, 1.0 / 5.0);
        fac = fac > dfac ? fac : dfac;
        fac = fac < 10.0 ? fac : 10.0;
        ndt = dt * fac;
    }
    ctl[CTL_DT] = ndt;
    if (!(t1 > tn)) ctl[CTL_DONE] = 1.0;
}

struct CnfDevArgs {
    double* ctl;
    float* yb[2];
    float* fb[2];
    const float* ctx; const float* e; const float* rec;
    float* out;              // [rows,4] dense output at t1
    double* partial;
    float rtol, atol;
    int rows, R, ntiles;
};

__device__ __forceinline__ bool cnf_gate_row(int r) { return r < 64 || (r >= 128 && r < 192) || (r >= 256 && r < 272); }

// CTX_LDS (inverse direction, R >= 4 rows per original point): the context rows of the workgroup's 64 / R points are copied
// to LDS once per tile and all six stage evaluations read them there - each evaluation re-read 1 152 B per row from L2 before
// SPLIT (the caller's PF_CNF_SPLIT_GATES): see cnf_eval.  Without CTX_LDS (the forward pass, R = 1: 64 points per tile) only the
// 144 gate columns of a point go to LDS, in the compact layout (cnf_eval<.., COMPACT>); the bias columns stay in global memory.
constexpr int CNF_GATES = 144;
__device__ __forceinline__ int cnf_gate_col(int c) { return c < 64 ? c : (c < 128 ? c + 64 : c + 128); }     // compact -> ctx column
template <bool CTX_LDS, bool SPLIT>
__global__ __launch_bounds__(CNF_NW * 64, PF_CNF_WPE) void cnf_step_dev_kernel(CnfDevArgs a) {
    constexpr bool COMPACT = SPLIT && !CTX_LDS;
    constexpr int TW = COMPACT ? CNF_GATES : CNF_CTX;               // width of a stage's row of the factor table
    __shared__ f4 wl[CNF_REC / 4];
    __shared__ double red[CNF_NW];
    __shared__ f4 sctx[CTX_LDS ? CNF_NW * 4 * CNF_CTX / 4 : (COMPACT ? CNF_NW * 16 * CNF_GATES / 4 : 1)];
    __shared__ float stf[SPLIT ? 6 * TW : 1];                        // [stage][column]: 2^(gt tsign alpha_s h) of the gate columns
    if (a.ctl[CTL_DONE] != 0.0) return;                              // uniform over the grid
    const int cur = (int)a.ctl[CTL_CUR];
    const float t = (float)a.ctl[CTL_T], h = (float)a.ctl[CTL_DT];
    const bool reverse = a.ctl[CTL_REV] != 0.0;
    const float sgn = reverse ? -1.f : 1.f, tsign = sgn;
    const bool cover = a.ctl[CTL_T1] <= a.ctl[CTL_T] + a.ctl[CTL_DT];
    float wd[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (cover) {                                                     // weights of (y0, y1, y_mid, dt f0, dt f1) at x = (t1 - t) / dt
        const double x = (a.ctl[CTL_T1] - a.ctl[CTL_T]) / a.ctl[CTL_DT], x2 = x * x, x3 = x2 * x, x4 = x2 * x2, dt = a.ctl[CTL_DT];
        wd[0] = (float)(-8 * x4 + 18 * x3 - 11 * x2 + 1); wd[1] = (float)(-8 * x4 + 14 * x3 - 5 * x2);
        wd[2] = (float)(16 * x4 - 32 * x3 + 16 * x2);
        wd[3] = (float)(dt * (-2 * x4 + 5 * x3 - 4 * x2 + x)); wd[4] = (float)(dt * (2 * x4 - 3 * x3 + x2));
    }
    const float* y0p = a.yb[cur];
    const float* f0p = a.fb[cur];
    float* y1p = a.yb[1 - cur];
    float* f1p = a.fb[1 - cur];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    const CnfW w = cnf_stage_weights(wl, a.rec, CNF_NW * 64, lane);
    constexpr float AL[6] = {1.f / 5, 3.f / 10, 4.f / 5, 8.f / 9, 1.f, 1.f};
    constexpr float BE[6][6] = {
        {1.f / 5, 0, 0, 0, 0, 0},
        {3.f / 40, 9.f / 40, 0, 0, 0, 0},
        {44.f / 45, -56.f / 15, 32.f / 9, 0, 0, 0},
        {19372.f / 6561, -25360.f / 2187, 64448.f / 6561, -212.f / 729, 0, 0},
        {9017.f / 3168, -355.f / 33, 46732.f / 5247, 49.f / 176, -5103.f / 18656, 0},
        {35.f / 384, 0, 500.f / 1113, 125.f / 192, -2187.f / 6784, 11.f / 84}};
    constexpr float CE[7] = {(float)(35. / 384 - 1951. / 21600), 0, (float)(500. / 1113 - 22642. / 50085),
                             (float)(125. / 192 - 451. / 720), (float)(-2187. / 6784 + 12231. / 42400),
                             (float)(11. / 84 - 649. / 6300), (float)(-1. / 60)};
    constexpr float CM[7] = {(float)(6025192743. / 30085553152 / 2), 0, (float)(51252292925. / 65400821598 / 2),
                             (float)(-2691868925. / 45128329728 / 2), (float)(187940372067. / 1594534317056 / 2),
                             (float)(-1776094331. / 19743644256 / 2), (float)(11237099. / 235043384 / 2)};
    if (SPLIT) {
        const float* tv = w.rec + 9872;
        for (int i = threadIdx.x; i < 6 * TW; i += CNF_NW * 64) {
            const int s = i / TW, r = COMPACT ? cnf_gate_col(i % TW) : i % TW;
            stf[i] = cnf_gate_row(r) ? __builtin_amdgcn_exp2f(tv[r] * (tsign * AL[s] * h)) : 0.f;
        }                                                             // (visible after the tile loop's barriers)
    }
    double acc = 0.0;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int g = (tile * CNF_NW + wave) * 16 + col;
        const bool ok = g < a.rows;
        const int row = ok ? g : a.rows - 1;
        const int pt = row / a.R;
        const f4 y0 = *reinterpret_cast<const f4*>(y0p + (size_t)row * 4);
        const float* cx = a.ctx + (size_t)pt * CNF_CTX;
        const float* cxb = cx;
        if (COMPACT) {
            // the gate columns of the tile's points (rows [64 tile, 64 tile + 63] -> points first .. last) as 2^(gt tsign t + gc)
            const long long npts = ((long long)a.rows + a.R - 1) / a.R;
            const long long r0 = (long long)tile * CNF_NW * 16;
            const long long p0 = r0 / a.R;
            long long pe = (r0 + CNF_NW * 16 - 1) / a.R;
            pe = pe < npts ? pe : npts - 1;
            const int np = (int)(pe - p0 + 1);                     // <= 64
            __syncthreads();                                       // every wave is done with the previous tile's columns
            const float tt = tsign * t;
            for (int i = threadIdx.x; i < np * (CNF_GATES / 4); i += CNF_NW * 64) {
                const int pp = i / (CNF_GATES / 4), c4 = (i % (CNF_GATES / 4)) * 4;
                const int r = cnf_gate_col(c4);
                const f4 v = *reinterpret_cast<const f4*>(a.ctx + (size_t)(p0 + pp) * CNF_CTX + r);
                const f4 g4 = *reinterpret_cast<const f4*>(w.rec + 9872 + r);
                f4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = __builtin_amdgcn_exp2f(fmaf(g4[k], tt, v[k]));
                sctx[i] = o;
            }
            __syncthreads();
            const long long pl = pt - p0;
            cx = reinterpret_cast<const float*>(sctx) + (size_t)(pl < 0 ? 0 : (pl >= np ? np - 1 : pl)) * CNF_GATES;
        }
        if (CTX_LDS) {
            const int ppw = CNF_NW * 16 / a.R;                     // points of this workgroup's tile (<= 16)
            const long long p0 = (long long)tile * ppw;
            const long long npts = ((long long)a.rows + a.R - 1) / a.R;
            __syncthreads();                                       // every wave is done with the previous tile's rows
            for (int i = threadIdx.x; i < ppw * (CNF_CTX / 4); i += CNF_NW * 64) {
                const long long pp = p0 + i / (CNF_CTX / 4);
                f4 v = reinterpret_cast<const f4*>(a.ctx)[(pp < npts ? pp : npts - 1) * (CNF_CTX / 4) + i % (CNF_CTX / 4)];
                if (SPLIT && cnf_gate_row(4 * (i % (CNF_CTX / 4)))) {  // the gate rows become 2^(gt tsign t + gc), once per point
                    const f4 g4 = *reinterpret_cast<const f4*>(w.rec + 9872 + 4 * (i % (CNF_CTX / 4)));
                    const float tt = tsign * t;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_exp2f(fmaf(g4[r], tt, v[r]));
                }
                sctx[i] = v;
            }
            __syncthreads();
            const long long pl = pt - p0;                            // a clamped out-of-range lane may point past the tile
            cx = reinterpret_cast<const float*>(sctx) + (size_t)(pl < 0 ? 0 : (pl >= ppw ? ppw - 1 : pl)) * CNF_CTX;
        }
        const float e0 = a.e[(size_t)pt * 3 + 0], e1 = a.e[(size_t)pt * 3 + 1], e2 = a.e[(size_t)pt * 3 + 2];
        f4 k[7];
        k[0] = *reinterpret_cast<const f4*>(f0p + (size_t)row * 4);
        f4 yi = y0;
        pf_static_for<0, 6>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            f4 comb = k[0] * BE[s][0];
#pragma unroll
            for (int j = 1; j <= s; ++j) comb += k[j] * BE[s][j];
            yi = y0 + comb * h;
            k[s + 1] = cnf_eval<SPLIT, COMPACT>(w, q, yi, tsign * (t + AL[s] * h), sgn, cx, e0, e1, e2, stf + (SPLIT ? s * TW : 0), cxb);
        });
        f4 err = k[0] * CE[0];
#pragma unroll
        for (int j = 1; j < 7; ++j) err += k[j] * CE[j];
        err *= h;
        if (ok && q == 0) {
            *reinterpret_cast<f4*>(y1p + (size_t)row * 4) = yi;
            *reinterpret_cast<f4*>(f1p + (size_t)row * 4) = k[6];
            if (cover) {
                f4 m = k[0] * CM[0];
#pragma unroll
                for (int j = 1; j < 7; ++j) m += k[j] * CM[j];
                const f4 ymid = y0 + m * h;
                // the same term order as the host path's pf_lincomb: ((((w0 y0) + w1 y1) + w2 ymid) + w3 f0) + w4 f1, fused
                f4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    o[c] = fmaf(wd[4], k[6][c], fmaf(wd[3], k[0][c], fmaf(wd[2], ymid[c], fmaf(wd[1], yi[c], fmaf(wd[0], y0[c], 0.f)))));
                *reinterpret_cast<f4*>(a.out + (size_t)row * 4) = o;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float r = err[c] / (a.atol + a.rtol * fmaxf(fabsf(y0[c]), fabsf(yi[c])));
                acc += (double)r * (double)r;
            }
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    // ---- this workgroup's share of the error sum; the workgroup that finishes LAST takes the controller's decision (it was a
    // one-wave launch of its own behind every attempt: ~100 launches of 5 - 7 us per forward).  Every workgroup has read the
    // controller state at its start, before the last one can have arrived here.
    __shared__ int last;
    if (threadIdx.x == 0) {
        double tt = 0.0;
        for (int i = 0; i < CNF_NW; ++i) tt += red[i];
        __hip_atomic_store(a.partial + blockIdx.x, tt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned* counter = reinterpret_cast<unsigned*>(a.ctl + CTL_ARRIVE);
        last = atomicAdd(counter, 1u) == gridDim.x - 1 ? 1 : 0;
        if (last) *counter = 0u;
    }
    __syncthreads();
    if (last) cnf_ctl_update(a.ctl, a.partial, (int)gridDim.x, red);                     // `last` is uniform over the workgroup
}

// torchdiffeq's `_select_initial_step` (oracle/cnf_ref.py::dopri5, first lines) around the probe evaluation
// f(t0 + h0, y0 + h0 f0): d0 = rms(y0 / scale), d1 = rms(f0 / scale), d2 = rms((f1 - f0) / scale) / h0.
// The same start in TWO launches (it was 13: reset, f0, two norms of two launches each, init_a, the probe, its norm, init_b -
// ~5 us of timeline apiece, twelve integrations per forward).  Launch A builds the state rows (x, 0) from the caller's points
// (row stride 3 or 4: the previous block's state is taken as it lies), evaluates f0 and sums |y0 / scale|^2 and |f0 / scale|^2;
// launch B evaluates the probe f(t0 + h0, y0 + h0 f0) and sums |(f1 - f0) / scale|^2 without storing f1.  In both, the
// workgroup that finishes last adds the per-workgroup sums in a fixed order and takes the one-thread decision.
struct CnfInitArgs {
    double* ctl;
    const float* x; int xs;   // A: points, row stride in floats
    float* y;                 // [rows,4]  A: written, B: read
    float* f0;                // [rows,4]  A: written, B: read
    const float* ctx; const float* e; const float* rec;
    double t0, t1, n_tot, extra_scale, reverse;
    const double* extra_d0;   // nullable device word: what the state rows outside y add to |y0 / scale|^2 (x extra_scale)
    float sgn, rtol, atol;
    int rows, R, ntiles;
    double* partial;          // [3][1024]
};

template <bool PROBE>
__global__ __launch_bounds__(CNF_NW * 64) void cnf_init_kernel(CnfInitArgs a) {
    __shared__ f4 wl[CNF_REC / 4];
    __shared__ double red[2][CNF_NW];
    __shared__ int last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    const CnfW w = cnf_stage_weights(wl, a.rec, CNF_NW * 64, lane);
    float h = 0.f, t = (float)(a.reverse != 0.0 ? -a.t0 : a.t0);
    if (PROBE) {
        const double hd = a.ctl[CTL_H0], td = a.ctl[CTL_T] + hd;
        h = (float)hd;
        t = (float)(a.ctl[CTL_REV] != 0.0 ? -td : td);
    }
    double acc0 = 0.0, acc1 = 0.0;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int g = (tile * CNF_NW + wave) * 16 + col;
        const bool ok = g < a.rows;
        const int row = ok ? g : a.rows - 1;
        const int pt = row / a.R;
        f4 y0, f0 = pf_splat(0.f);
        if (PROBE) {
            y0 = *reinterpret_cast<const f4*>(a.y + (size_t)row * 4);
            f0 = *reinterpret_cast<const f4*>(a.f0 + (size_t)row * 4);
        } else {
            const float* xp = a.x + (size_t)row * a.xs;
            y0 = (f4){xp[0], xp[1], xp[2], 0.f};
            if (ok && q == 0) *reinterpret_cast<f4*>(a.y + (size_t)row * 4) = y0;
        }
        f4 y = y0;
        if (PROBE) y += f0 * (h * 1.f);
        const float* cx = a.ctx + (size_t)pt * CNF_CTX;
        const f4 k = cnf_eval(w, q, y, t, a.sgn, cx, a.e[(size_t)pt * 3 + 0], a.e[(size_t)pt * 3 + 1], a.e[(size_t)pt * 3 + 2]);
        if (ok && q == 0 && !PROBE) *reinterpret_cast<f4*>(a.f0 + (size_t)row * 4) = k;
        if (ok) {                                            // the four lanes of a column take one component each (same box:
                                                             // 12.96 -> 12.86 ms per forward against the q = 0 lanes doing all four)
            const float yc = q == 0 ? y0.x : (q == 1 ? y0.y : (q == 2 ? y0.z : y0.w));
            const float kc = q == 0 ? k.x : (q == 1 ? k.y : (q == 2 ? k.z : k.w));
            const float sc = a.atol + a.rtol * fabsf(yc);
            if (PROBE) {
                const float fc = q == 0 ? f0.x : (q == 1 ? f0.y : (q == 2 ? f0.z : f0.w));
                const float r = (kc - fc) / sc;
                acc0 += (double)r * (double)r;
            } else {
                const float r0 = yc / sc, r1 = kc / sc;
                acc0 += (double)r0 * (double)r0;
                acc1 += (double)r1 * (double)r1;
            }
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { acc0 += __shfl_xor(acc0, m); acc1 += __shfl_xor(acc1, m); }
    if (lane == 0) { red[0][wave] = acc0; red[1][wave] = acc1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s0 = 0.0, s1 = 0.0;
        for (int i = 0; i < CNF_NW; ++i) { s0 += red[0][i]; s1 += red[1][i]; }
        __hip_atomic_store(a.partial + (PROBE ? 2048 : 0) + blockIdx.x, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!PROBE) __hip_atomic_store(a.partial + 1024 + blockIdx.x, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned* counter = reinterpret_cast<unsigned*>(a.ctl + CTL_ARRIVE);
        last = atomicAdd(counter, 1u) == gridDim.x - 1 ? 1 : 0;
        if (last) *counter = 0u;
    }
    __syncthreads();
    if (!last) return;                                                                   // uniform over the workgroup
    double* ctl = a.ctl;
    if (PROBE) {
        const double r2 = cnf_partial_total(a.partial + 2048, (int)gridDim.x, red[0]);
        if (threadIdx.x != 0) return;
        const double h0 = ctl[CTL_H0], d1 = ctl[CTL_D1];
        const double d2 = sqrt(r2 / ctl[CTL_NTOT]) / h0;
        const double dm = d1 > d2 ? d1 : d2;
        const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? (1e-6 > h0 * 1e-3 ? 1e-6 : h0 * 1e-3) : pow(0.01 / dm, 1.0 / 5.0);
        ctl[CTL_DT] = 100.0 * h0 < h1 ? 100.0 * h0 : h1;
        ctl[CTL_NFE] = 2.0;
    } else {
        const double r0 = cnf_partial_total(a.partial, (int)gridDim.x, red[0]);
        const double r1 = cnf_partial_total(a.partial + 1024, (int)gridDim.x, red[0]);
        if (threadIdx.x != 0) return;
        for (int i = 0; i < 16; ++i)
            ctl[i] = i == CTL_T ? a.t0 : i == CTL_T1 ? a.t1 : i == CTL_NTOT ? a.n_tot : i == CTL_REV ? a.reverse : 0.0;
        const double extra = a.extra_d0 ? a.extra_d0[0] * a.extra_scale : 0.0;
        const double d0 = sqrt((r0 + extra) / a.n_tot), d1 = sqrt(r1 / a.n_tot);
        ctl[CTL_H0] = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        ctl[CTL_D1] = d1;
    }
}

// ---- Runge-Kutta bookkeeping -----------------------------------------------------------------------
struct LinArgs {
    const float* p[8];
    float w[8];
    int n_terms;
    float* out;
    long long n;
};

__global__ __launch_bounds__(256) void lincomb_kernel(LinArgs a) {                 // out = sum_j w[j] p[j]
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    float s = 0.f;
    for (int j = 0; j < a.n_terms; ++j) s = fmaf(a.w[j], a.p[j][i], s);
    a.out[i] = s;
}

// sum over elements of ((A - B) / (atol + rtol * max(|S0|, |S1|)))^2, two deterministic levels, double accumulation.
// A may be given as h * sum_j w[j] K_j (error estimate) by passing n_terms > 0.
struct SumsqArgs {
    const float* a;          // nullable when n_terms > 0
    const float* b;          // nullable
    const float* s0;         // scale state
    const float* s1;         // nullable (second state of the max)
    const float* k;          // [n_terms][n]
    float w[8];
    int n_terms;
    float h, rtol, atol;
    long long n;
    double* partial;         // [gridDim.x]
};

__global__ __launch_bounds__(256) void sumsq_kernel(SumsqArgs s) {
    __shared__ double sh[256];
    double acc = 0.0;
    // a - b against s0 (| s1) with 16-byte loads when the layout allows it (the context norms: 8 M elements per block, one
    // 4-byte load and one division per loop trip took 50 us = 0.66 TB/s); the per-thread order of the additions changes,
    // the double accumulation keeps the sum's low bits out of reach of the step-size heuristic it feeds
    const bool vec = s.n_terms == 0 && (s.n & 3) == 0 && ((reinterpret_cast<size_t>(s.a) | reinterpret_cast<size_t>(s.b) |
                     reinterpret_cast<size_t>(s.s0) | reinterpret_cast<size_t>(s.s1)) & 15) == 0;
    if (vec) {
        const long long n4 = s.n / 4;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            f4 v = reinterpret_cast<const f4*>(s.a)[i];
            if (s.b) v -= reinterpret_cast<const f4*>(s.b)[i];
            const f4 m0 = reinterpret_cast<const f4*>(s.s0)[i];
            f4 m = {fabsf(m0.x), fabsf(m0.y), fabsf(m0.z), fabsf(m0.w)};
            if (s.s1) {
                const f4 m1 = reinterpret_cast<const f4*>(s.s1)[i];
                m = (f4){fmaxf(m.x, fabsf(m1.x)), fmaxf(m.y, fabsf(m1.y)), fmaxf(m.z, fabsf(m1.z)), fmaxf(m.w, fabsf(m1.w))};
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float r = v[k] / (s.atol + s.rtol * m[k]);
                acc += (double)r * (double)r;
            }
        }
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < s.n && !vec; i += (long long)gridDim.x * 256) {
        float v;
        if (s.n_terms > 0) {
            v = 0.f;
            for (int j = 0; j < s.n_terms; ++j) v = fmaf(s.w[j], s.k[(long long)j * s.n + i], v);
            v *= s.h;
        } else {
            v = s.a[i] - (s.b ? s.b[i] : 0.f);
        }
        float m = fabsf(s.s0[i]);
        if (s.s1) m = fmaxf(m, fabsf(s.s1[i]));
        const float r = v / (s.atol + s.rtol * m);
        acc += (double)r * (double)r;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) s.partial[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const double* __restrict__ partial, int np, double* __restrict__ out) {
    __shared__ double sh[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) acc += partial[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

constexpr int SUMSQ_BLOCKS = 256;

}  // namespace

extern "C" int pf_cnf_rhs(const float* y0, const float* k, const float* coef, int ncoef, float h, float t, float sgn,
                          const float* ctx, const float* e, const float* rec, float* kout, float* yout, int rows, int R,
                          void* stream) {
    if (!y0 || !ctx || !e || !rec || !kout || (ncoef > 0 && (!k || !coef))) return PF_ERR_NULL;
    if (rows <= 0 || R <= 0 || ncoef < 0 || ncoef > 6) return PF_ERR_SHAPE;
    CnfArgs a{};
    a.y0 = y0; a.k = k; a.ncoef = ncoef; a.h = h; a.t = t; a.sgn = sgn; a.ctx = ctx; a.e = e; a.rec = rec;
    a.kout = kout; a.yout = yout; a.rows = rows; a.R = R;
    for (int j = 0; j < ncoef; ++j) a.coef[j] = coef[j];
    a.ntiles = (rows + CNF_NW * 16 - 1) / (CNF_NW * 16);
    const int grid = a.ntiles < 1024 ? a.ntiles : 1024;
    hipLaunchKernelGGL(cnf_rhs_kernel, dim3(grid), dim3(CNF_NW * 64), 0, (hipStream_t)stream, a);
    return pf_last_launch_status();
}

// out[i] = sum_j w[j] * p[j][i], j < n_terms <= 8 (host arrays of device pointers / weights)
extern "C" int pf_lincomb(const float* const* ptrs, const float* w, int n_terms, float* out, long long n, void* stream) {
    if (!ptrs || !w || !out) return PF_ERR_NULL;
    if (n_terms <= 0 || n_terms > 8 || n <= 0) return PF_ERR_SHAPE;
    LinArgs a{};
    for (int j = 0; j < n_terms; ++j) {
        if (!ptrs[j]) return PF_ERR_NULL;
        a.p[j] = ptrs[j]; a.w[j] = w[j];
    }
    a.n_terms = n_terms; a.out = out; a.n = n;
    hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    return pf_last_launch_status();
}

// out[0] (double) = sum_i (v_i / (atol + rtol * max(|s0_i|, |s1_i|)))^2 with v = a - b, or v = h * sum_j w[j] k[j] when
// n_terms > 0 (k: [n_terms][n]).  ws: >= 256 doubles of scratch.  Deterministic (fixed two-level tree).
extern "C" int pf_scaled_sumsq(const float* a, const float* b, const float* s0, const float* s1, const float* k,
                               const float* w, int n_terms, float h, float rtol, float atol, long long n, double* ws,
                               double* out, void* stream) {
    if (!s0 || !ws || !out || (n_terms == 0 && !a) || (n_terms > 0 && (!k || !w))) return PF_ERR_NULL;
    if (n <= 0 || n_terms < 0 || n_terms > 8) return PF_ERR_SHAPE;
    SumsqArgs s{};
    s.a = a; s.b = b; s.s0 = s0; s.s1 = s1; s.k = k; s.n_terms = n_terms; s.h = h; s.rtol = rtol; s.atol = atol;
    s.n = n; s.partial = ws;
    for (int j = 0; j < n_terms; ++j) s.w[j] = w[j];
    hipLaunchKernelGGL(sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, (hipStream_t)stream, s);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, SUMSQ_BLOCKS, out);
    return pf_last_launch_status();
}

// One Dormand-Prince 5(4) step attempt: y1 = y0 + h sum b_j k_j (k_1 = f0 given, FSAL), f1 = k_7, optional dense-output
// mid-point, and out[0] (double) = sum_i (err_i / (atol + rtol max(|y0_i|, |y1_i|)))^2 of the embedded error estimate.
// t: start of the step in solver time; reverse != 0 integrates the way torchdiffeq does for decreasing times
// (net time = -solver time, derivative negated).  ws: >= 1024 doubles.
extern "C" int pf_cnf_step(const float* y0, const float* f0, float t, float h, int reverse, const float* ctx, const float* e,
                           const float* rec, float* y1, float* f1, float* ymid, float rtol, float atol, int rows, int R,
                           double* ws, double* out, void* stream) {
    if (!y0 || !f0 || !ctx || !e || !rec || !y1 || !f1 || !ws || !out) return PF_ERR_NULL;
    if (rows <= 0 || R <= 0) return PF_ERR_SHAPE;
    CnfStepArgs a{};
    a.y0 = y0; a.f0 = f0; a.ctx = ctx; a.e = e; a.rec = rec; a.y1 = y1; a.f1 = f1; a.ymid = ymid; a.partial = ws;
    a.t = t; a.h = h; a.sgn = reverse ? -1.f : 1.f; a.tsign = reverse ? -1.f : 1.f; a.rtol = rtol; a.atol = atol;
    a.rows = rows; a.R = R;
    a.ntiles = (rows + CNF_NW * 16 - 1) / (CNF_NW * 16);
    const int grid = a.ntiles < 1024 ? a.ntiles : 1024;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cnf_step_kernel, dim3(grid), dim3(CNF_NW * 64), 0, s, a);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, s, ws, grid, out);
    return pf_last_launch_status();
}

// `n_attempts` step attempts of the adaptive integration whose state is in `ctl` (layout above), enqueued back to back; the
// controller's decision is taken inside each attempt's launch by the workgroup that finishes last: no host synchronisation.  ya / yb and fa / fb: the two state and derivative buffers
// ([rows,4]; ctl's `cur` says which one holds the current state), out: the solution at t1 once ctl's `done` is set with
// status 0.  ws: >= 1024 doubles.
// flags: PF_CNF_SPLIT_GATES - the caller vouches that log2(e) max|gate time weight| |t1 - t0| <= 100 for this record
// (packing.cnf_split_ok); the gates' 2^x then factor into a per-point and a per-stage part without overflow (cnf_eval).
extern "C" int pf_cnf_steps(double* ctl, float* ya, float* yb, float* fa, float* fb, const float* ctx, const float* e,
                            const float* rec, float* out, float rtol, float atol, int rows, int R, int n_attempts, double* ws,
                            int flags, void* stream) {
    if (!ctl || !ya || !yb || !fa || !fb || !ctx || !e || !rec || !out || !ws) return PF_ERR_NULL;
    if (rows <= 0 || R <= 0 || n_attempts <= 0) return PF_ERR_SHAPE;
    CnfDevArgs a{};
    a.ctl = ctl; a.yb[0] = ya; a.yb[1] = yb; a.fb[0] = fa; a.fb[1] = fb; a.ctx = ctx; a.e = e; a.rec = rec; a.out = out;
    a.partial = ws; a.rtol = rtol; a.atol = atol; a.rows = rows; a.R = R;
    a.ntiles = (rows + CNF_NW * 16 - 1) / (CNF_NW * 16);
    const int grid = a.ntiles < 1024 ? a.ntiles : 1024;
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < n_attempts; ++i) {
        // context rows through LDS when a tile's 64 rows belong to <= 16 whole points
        if (R >= 4 && (CNF_NW * 16) % R == 0) {
            if (flags & PF_CNF_SPLIT_GATES) hipLaunchKernelGGL((cnf_step_dev_kernel<true, true>), dim3(grid), dim3(CNF_NW * 64), 0, s, a);
            else hipLaunchKernelGGL((cnf_step_dev_kernel<true, false>), dim3(grid), dim3(CNF_NW * 64), 0, s, a);
        } else {
            if ((flags & PF_CNF_SPLIT_GATES) && PF_CNF_SPLIT_FWD) hipLaunchKernelGGL((cnf_step_dev_kernel<false, true>), dim3(grid), dim3(CNF_NW * 64), 0, s, a);
            else hipLaunchKernelGGL((cnf_step_dev_kernel<false, false>), dim3(grid), dim3(CNF_NW * 64), 0, s, a);
        }
    }
    return pf_last_launch_status();
}

// The start of an adaptive integration over [t0, t1] (solver time; the network sees -t when reverse), all on the device, two
// launches (cnf_init_kernel): the state rows y = (x, 0) from the points x (row stride x_stride = 3 or 4 floats), f0 = f(t0, y),
// the controller state reset, torchdiffeq's initial step size into ctl[1].  n_tot: elements of the RMS norms (rows*4 + the
// context rows torchdiffeq integrates alongside), extra_d0 (a DEVICE double, nullable) x extra_scale: what those extra rows
// add to the squared norm of y0 (pf_scaled_sumsq of the context with itself as the scale; the host never reads it).
// ws: >= 3072 doubles.  ctl: 16 doubles, ALL ZERO before its first use (the arrival word at [13] is left zero by every launch).
extern "C" int pf_cnf_init(double* ctl, const float* x, int x_stride, float* y, float* f0, const float* ctx, const float* e,
                           const float* rec, double t0, double t1, double n_tot, const double* extra_d0, double extra_scale,
                           int reverse, float rtol, float atol, int rows, int R, double* ws, void* stream) {
    if (!ctl || !x || !y || !f0 || !ctx || !e || !rec || !ws) return PF_ERR_NULL;
    if (rows <= 0 || R <= 0 || !(n_tot > 0.0) || (x_stride != 3 && x_stride != 4)) return PF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    CnfInitArgs a{};
    a.ctl = ctl; a.x = x; a.xs = x_stride; a.y = y; a.f0 = f0; a.ctx = ctx; a.e = e; a.rec = rec;
    a.t0 = t0; a.t1 = t1; a.n_tot = n_tot; a.extra_d0 = extra_d0; a.extra_scale = extra_scale; a.reverse = reverse ? 1.0 : 0.0;
    a.sgn = reverse ? -1.f : 1.f; a.rtol = rtol; a.atol = atol; a.rows = rows; a.R = R; a.partial = ws;
    a.ntiles = (rows + CNF_NW * 16 - 1) / (CNF_NW * 16);
    const int grid = a.ntiles < 1024 ? a.ntiles : 1024;
    hipLaunchKernelGGL(cnf_init_kernel<false>, dim3(grid), dim3(CNF_NW * 64), 0, s, a);
    hipLaunchKernelGGL(cnf_init_kernel<true>, dim3(grid), dim3(CNF_NW * 64), 0, s, a);
    return pf_last_launch_status();
}
