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
// 64 -> 64 -> (1|2): split-fp16 products with a natural-scale low half on the fp16 MFMA (pf_mfma.h "f16n", fp32-class
// accuracy; W2 / W4 arrive scaled by a power of two each, their biases pre-multiplied, the kernel multiplies by the
// inverses), 16 rows per column tile, 30 MFMAs per tile and block.
//
// The chain of 6 blocks is latency-bound, not throughput-bound (per tile: 6 x [gather cp/st -> 64 VALU ->
// 24 MFMA -> split -> 6 MFMA -> exp/affine]), so the kernel is built for occupancy and short latencies:
// all six weight records live in LDS for the whole persistent workgroup (126 KiB), 16 waves per workgroup
// at <= 128 VGPRs (4 waves per SIMD), and the next block's cp / st rows are fetched while the current block computes.
//
// Per-block weight record (floats), stride FLOW_REC = 5360:
//   [0,4096)    f16n image of 2^s2 W2 (4 ob x 2 pairs)      [4096,5120) f16n image of 2^s4 W4 (1 ob x 2 pairs; rows
//               replicated into every 4-row q group)
//   [5120,5184) 2^s2 b2      [5184,5200) 2^s4 b4 (replicated likewise)      [5200,5328) W0h [64][2]
//   [5328,5352) A(9) a0(3) Ai(9) ai0(3)   [5352] 2^-s2   [5353] 2^-s4   [5354,5360) pad
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"
#include "pf_mfma.h"

// waves per workgroup: both directions fit 128 VGPRs -> 16 waves (4 per SIMD) since the coupling nets run on the f16n
// helpers (no cross accumulator)
#ifndef PF_FLOW_NW
#define PF_FLOW_NW 16
#endif
#ifndef PF_FLOW_NW_FWD
#define PF_FLOW_NW_FWD 16
#endif

namespace {

constexpr int FLOW_REC = 5360;
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
    // fwd with the log-likelihood folded in (pf_flow_fwd_logp; N % 16 == 0): every wave leaves the sums of its 16 rows, the
    // workgroup that finishes last reduces them per batch item in a fixed order
    float* part;         // [2][rows / 16]: sum of ld_pt, sum of -0.5 (z^2 + log 2 pi) over a wave tile; nullable = no fold
    unsigned* counter;   // arrival counter behind the partials (zero between launches)
    float* ldj; float* lps; float* logp;      // [B], [B], [1]
    float ld_const;
    int B, N;
    // inv fed by the interpolation WEIGHTS instead of u (pf_flow_inv_interp; R <= 4): u[n R + r] = sum_k aw[n][k][r] z[j_k]
    const float* aw;     // [T][8][4]  (nullable = read `in`)
    const float* z;      // [T,3]
    const int* idx;      // [T,16] (first 8 used), index inside the batch item
};

struct FlowCond { f4 cp[4]; f4 s0, s1; };        // one block's conditioning of one row: cp [64] (this lane's 16), s|t [8]

__device__ __forceinline__ FlowCond flow_cond(const float* __restrict__ cp, const float* __restrict__ st, int T, int u,
                                              int pt, int q) {
    FlowCond c;
    const float* cr = cp + ((size_t)u * T + pt) * 64 + 4 * q;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) c.cp[cb] = *reinterpret_cast<const f4*>(cr + cb * 16);
    const float* sr = st + ((size_t)u * T + pt) * 8;
    c.s0 = *reinterpret_cast<const f4*>(sr);
    c.s1 = *reinterpret_cast<const f4*>(sr + 4);
    return c;
}

// coupling MLP of one column tile: o[0..1] = bias_net output (3 - TD values), identical in all q lanes
template <int TD>
__device__ __forceinline__ void coupling_net(const float* rec /*LDS*/, int lane, int q, const FlowCond& c,
                                             const float (&h1)[2], float (&o)[2]) {
    PfPairN hp[1][2];
    const float i2 = rec[5352], i4 = rec[5353];
    {
        f4 hid[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const f4 wa = *reinterpret_cast<const f4*>(rec + 5200 + (cb * 16 + 4 * q) * 2);       // W0h rows ch, ch+1
            const f4 wb = *reinterpret_cast<const f4*>(rec + 5200 + (cb * 16 + 4 * q) * 2 + 4);   // rows ch+2, ch+3
            f4 v = c.cp[cb];
            v.x = fmaf(wa.x, h1[0], v.x); v.y = fmaf(wa.z, h1[0], v.y);
            v.z = fmaf(wb.x, h1[0], v.z); v.w = fmaf(wb.z, h1[0], v.w);
            if (TD == 2) {
                v.x = fmaf(wa.y, h1[1], v.x); v.y = fmaf(wa.w, h1[1], v.y);
                v.z = fmaf(wb.y, h1[1], v.z); v.w = fmaf(wb.w, h1[1], v.w);
            }
            hid[cb] = pf_lrelu(v, 0.01f);
        }
        hp[0][0] = pf_pairn(hid[0], hid[1]);
        hp[0][1] = pf_pairn(hid[2], hid[3]);
    }
    const PfW2Lds ws{reinterpret_cast<const u4*>(rec), lane};
    f4 h2[1][4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) h2[0][ob] = pf_bias(rec + 5120, ob, q);
    pf_mmn<false, 4, 2, 2>(ws, 0, hp, h2);
    hp[0][0] = pf_pairn(pf_lrelu(h2[0][0] * i2, 0.01f), pf_lrelu(h2[0][1] * i2, 0.01f));
    hp[0][1] = pf_pairn(pf_lrelu(h2[0][2] * i2, 0.01f), pf_lrelu(h2[0][3] * i2, 0.01f));
    f4 acc[1][1];
    acc[0][0] = *reinterpret_cast<const f4*>(rec + 5184 + 4 * q);
    pf_mmn<false, 1, 2, 2>(ws, 8, hp, acc);
    o[0] = acc[0][0].x * i4; o[1] = acc[0][0].y * i4;
}

template <bool INV, int NW>
__global__ __launch_bounds__(NW * 64) void flow_kernel(FlowArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, q = lane >> 4;
    __shared__ f4 wl[6 * FLOW_REC / 4];
    pf_stage_lds(wl, reinterpret_cast<const f4*>(a.w), 6 * FLOW_REC / 4);
    __syncthreads();
    const float* wlf = reinterpret_cast<const float*>(wl);

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int g = (tile * NW + wave) * 16 + col;
        const bool ok = g < a.rows;
        const int row = ok ? g : a.rows - 1;
        const int pt = row / a.R;
        float v[3], ld = 0.f;
        if (INV && a.aw) {
            // the weighted latent sum of the interpolation module (interpflow.py:183-185) in the order of interp_kernel's
            // shuffle tree - ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)), products and sums unfused - so that the
            // split path (pf_interp_weights + this) returns the bits of pf_interp + pf_flow_inv
            const int r = row - pt * a.R, bN = (pt / a.N) * a.N;
            const int4 j0 = *reinterpret_cast<const int4*>(a.idx + (size_t)pt * 16), j1 = *reinterpret_cast<const int4*>(a.idx + (size_t)pt * 16 + 4);
            const int jj[8] = {j0.x, j0.y, j0.z, j0.w, j1.x, j1.y, j1.z, j1.w};
            float sk[8][3];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float wk = a.aw[((size_t)pt * 8 + k) * 4 + r];
                const float* zz = a.z + (size_t)(bN + jj[k]) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) sk[k][c] = __fmul_rn(wk, zz[c]);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)
                v[c] = __fadd_rn(__fadd_rn(__fadd_rn(sk[0][c], sk[1][c]), __fadd_rn(sk[2][c], sk[3][c])),
                                 __fadd_rn(__fadd_rn(sk[4][c], sk[5][c]), __fadd_rn(sk[6][c], sk[7][c])));
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = a.in[(size_t)row * 3 + c];
        }
        FlowCond cn = flow_cond(a.cp, a.st, a.T, INV ? 5 : 0, pt, q);

        pf_static_for<0, 6>([&](auto uc) {
            constexpr int i = decltype(uc)::value;
            constexpr int u = INV ? 5 - i : i;
            constexpr int TD = (u % 2 == 0) ? 1 : 2;
            const float* rec = wlf + u * FLOW_REC;
            const float* fc = rec + 5328;
            const FlowCond c = cn;
            if constexpr (i < 5) cn = flow_cond(a.cp, a.st, a.T, INV ? u - 1 : u + 1, pt, q);    // next block's rows
            __builtin_amdgcn_sched_barrier(0);
            const float s[3] = {c.s0.x, c.s0.y, c.s0.z}, t[3] = {c.s0.w, c.s1.x, c.s1.y};
            float h1[2], o[2];
            if constexpr (!INV) {
                const float x0 = v[0], x1 = v[1], x2 = v[2];                   // actnorm o inv1x1:  v = A v + a0
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    v[k] = fmaf(fc[3 * k + 2], x2, fmaf(fc[3 * k + 1], x1, fmaf(fc[3 * k], x0, fc[9 + k])));
                h1[0] = v[0]; h1[1] = v[1];
                coupling_net<TD>(rec, lane, q, c, h1, o);
                if (TD == 1) { v[1] -= o[0]; v[2] -= o[1]; } else { v[2] -= o[0]; }
                const float y0 = v[2], y1 = v[1], y2 = v[0];                   // reverse channels
                v[0] = (y0 - t[0]) * expf(-s[0]);
                v[1] = (y1 - t[1]) * expf(-s[1]);
                v[2] = (y2 - t[2]) * expf(-s[2]);
                ld -= (s[0] + s[1]) + s[2];
            } else {
                const float y0 = fmaf(v[0], expf(s[0]), t[0]);
                const float y1 = fmaf(v[1], expf(s[1]), t[1]);
                const float y2 = fmaf(v[2], expf(s[2]), t[2]);
                v[0] = y2; v[1] = y1; v[2] = y0;                               // reverse^-1 (self-inverse)
                h1[0] = v[0]; h1[1] = v[1];
                coupling_net<TD>(rec, lane, q, c, h1, o);
                if (TD == 1) { v[1] += o[0]; v[2] += o[1]; } else { v[2] += o[0]; }
                const float x0 = v[0], x1 = v[1], x2 = v[2];
#pragma unroll
                for (int k = 0; k < 3; ++k)                                    // (inv1x1 o actnorm)^-1:  v = Ai v + ai0
                    v[k] = fmaf(fc[12 + 3 * k + 2], x2, fmaf(fc[12 + 3 * k + 1], x1, fmaf(fc[12 + 3 * k], x0, fc[21 + k])));
            }
        });

        if (ok && q == 0) {
            a.out[(size_t)row * 3 + 0] = v[0];
            a.out[(size_t)row * 3 + 1] = v[1];
            a.out[(size_t)row * 3 + 2] = v[2];
            if (!INV) a.ld_pt[row] = ld;
        }
        if constexpr (!INV) {
            if (a.part) {
                // sums over the 16 rows of this wave tile (rows % 16 == 0: the tile is all valid or all padding, and it
                // lies inside one batch item), fixed butterfly order; written with an agent-scope store (the reader sits
                // on another XCD: its L2 is not coherent with ours)
                float sl = ld;
                float sz = -0.5f * (v[0] * v[0] + LOG2PI_F) + -0.5f * (v[1] * v[1] + LOG2PI_F) + -0.5f * (v[2] * v[2] + LOG2PI_F);
                sl += __shfl_xor(sl, 1); sz += __shfl_xor(sz, 1);
                sl += __shfl_xor(sl, 2); sz += __shfl_xor(sz, 2);
                sl += __shfl_xor(sl, 4); sz += __shfl_xor(sz, 4);
                sl += __shfl_xor(sl, 8); sz += __shfl_xor(sz, 8);
                if (ok && lane == 0) {
                    const int wt = g >> 4;
                    __hip_atomic_store(a.part + wt, sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.part + (a.rows >> 4) + wt, sz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    if constexpr (!INV) {
        if (!a.part) return;
        // ---- log-likelihood: the workgroup that arrives last reduces the wave-tile sums per batch item (probs.py:73-75,
        // interpflow.py:339-345), a fixed order whichever workgroup that is.  No release fence (it would write the whole L2
        // back, tens of microseconds): the partials are agent-scope stores, waiting for their acknowledgement orders them
        // before the arrival count (the same protocol as train_fused.hip stat_flush).
        __shared__ float sa[NW * 64];
        __shared__ int is_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) is_last = atomicAdd(a.counter, 1u) == gridDim.x - 1 ? 1 : 0;
        __syncthreads();
        if (!is_last) return;
        // one WAVE per batch item (items wave, wave + NW, ...): lane-strided sums of the item's N / 16 wave-tile sums, then a
        // fixed shuffle tree - no workgroup barrier per item
        const int tid = threadIdx.x;
        const int per = a.N >> 4, half = a.rows >> 4;
        for (int b = wave; b < a.B; b += NW) {
            float accl = 0.f, accz = 0.f;
            for (int i = lane; i < per; i += 64) {
                accl += __hip_atomic_load(a.part + b * per + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                accz += __hip_atomic_load(a.part + half + b * per + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) { accl += __shfl_xor(accl, m); accz += __shfl_xor(accz, m); }
            if (lane == 0) {
                const float l = accl + a.ld_const * (float)a.N;
                a.ldj[b] = l;
                a.lps[b] = accz + l;
                if (b < NW * 64) sa[b] = accz + l;
            }
        }
        __syncthreads();
        float tot = 0.f;
        if (tid == 0)
            for (int b = 0; b < a.B; ++b) tot += b < NW * 64 ? sa[b] : a.lps[b];       // batch order, whatever wave produced the term
        if (tid == 0) {
            *a.logp = -tot / (float)a.B;
            __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
        }
    }
}

// Waves per workgroup: 16 when that still gives every CU a tile; small batches take 8 or 4, so that the (latency-bound)
// chain of a tile runs on more CUs at once - the 126 KiB weight prologue per workgroup is the price, paid in parallel.
template <bool INV, int NW>
int launch_nw(FlowArgs a, hipStream_t s) {
    a.ntiles = (a.rows + NW * 16 - 1) / (NW * 16);
    const int grid = a.ntiles < 256 ? a.ntiles : 256;          // persistent: 126 KiB of LDS = one workgroup per CU
    hipLaunchKernelGGL((flow_kernel<INV, NW>), dim3(grid), dim3(NW * 64), 0, s, a);
    return pf_last_launch_status();
}

template <bool INV>
int launch(FlowArgs a, hipStream_t s) {
    constexpr int NWMAX = INV ? PF_FLOW_NW : PF_FLOW_NW_FWD;
    const int wt = (a.rows + 15) / 16;                          // wave tiles
    if (wt >= 200 * NWMAX) return launch_nw<INV, NWMAX>(a, s);
    if (wt >= 200 * 8) return launch_nw<INV, 8>(a, s);
    return launch_nw<INV, 4>(a, s);
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

// pf_flow_fwd + pf_logp in ONE launch: z, ld_pt as pf_flow_fwd; ldj [B], lpsum [B], logp [1] as pf_logp (the sums run in a
// different - still fixed - order than pf_logp's: equal to it within fp32 rounding, identical from run to run).
// ws: 2 * ceil(B N / 16) floats + 1 word, ZERO before the first call (the kernel leaves the counter word zero again).
// N % 16 != 0: the two-launch path (a wave tile would straddle batch items).
extern "C" long long pf_flow_fwd_logp_ws_floats(int B, int N) { return 2ll * (((long long)B * N + 15) / 16) + 4; }

extern "C" int pf_flow_fwd_logp(const float* xyz, const float* cp, const float* st, const float* w, float* z, float* ld_pt,
                                float ld_const, int B, int N, float* ldj, float* lpsum, float* logp, float* ws, void* stream) {
    if (!xyz || !cp || !st || !w || !z || !ld_pt || !ldj || !lpsum || !logp || !ws) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || (long long)B * N > (1ll << 30)) return PF_ERR_SHAPE;
    const int T = B * N;
    if (N % 16 != 0) {
        const int rc = pf_flow_fwd(xyz, cp, st, w, z, ld_pt, T, stream);
        return rc != PF_OK ? rc : pf_logp(z, ld_pt, ld_const, B, N, ldj, lpsum, logp, stream);
    }
    FlowArgs a{};
    a.in = xyz; a.cp = cp; a.st = st; a.w = w; a.out = z; a.ld_pt = ld_pt; a.T = T; a.R = 1; a.rows = T;
    a.part = ws; a.counter = reinterpret_cast<unsigned*>(ws + 2 * (T / 16));
    a.ldj = ldj; a.lps = lpsum; a.logp = logp; a.ld_const = ld_const; a.B = B; a.N = N;
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

// Flow g fed by the interpolation weights: x [B, N R, 3] = g(u) with u[n R + r] = sum_k aw[n][k][r] z[idx8[n][k]] formed inside
// the kernel (R <= 4) - bit-identical to pf_interp followed by pf_flow_inv.  aw: pf_interp_weights.
extern "C" int pf_flow_inv_interp(const float* aw, const float* z, const int* idx16, const float* cp, const float* st, const float* w,
                                  float* x, int B, int N, int R, void* stream) {
    if (!aw || !z || !idx16 || !cp || !st || !w || !x) return PF_ERR_NULL;
    if (B <= 0 || N < 8 || (long long)B * N * R > (1ll << 30)) return PF_ERR_SHAPE;
    if (R < 1 || R > 4) return PF_ERR_UNSUPPORTED;
    FlowArgs a{};
    a.in = nullptr; a.cp = cp; a.st = st; a.w = w; a.out = x; a.ld_pt = nullptr; a.T = B * N; a.R = R; a.rows = B * N * R;
    a.aw = aw; a.z = z; a.idx = idx16; a.N = N;
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
