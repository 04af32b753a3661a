// Auction-algorithm EMD (approximate earth mover's distance) forward + backward.
// Replaces the reference's in-tree CUDA extension metric/emd/emd_cuda.cu (forward :228-282,
// backward :284-316; pybind surface metric/emd/emd.cpp:14-31) - a fresh design, not a port:
//
//   * the reference runs 7 tiny kernels per auction iteration (351 launches for 50 iterations);
//     here ONE launch runs the whole auction: one 1024-thread workgroup per batch sample iterates
//     bid -> winner -> assign with workgroup barriers, ground-truth points and prices resident in LDS;
//   * bidding is wave-cooperative: a wave owns one unassigned point, its 64 lanes scan the objects
//     strided (coalesced LDS reads), then a 6-step butterfly merges (best, second-best, argbest);
//   * the reference's racy winner selection (float atomicMax by CAS + "within 1e-6, last writer
//     wins", cu:10-20,188-191) becomes two integer atomics: max of the increment's bit pattern
//     (increments are > 0, so the bits order like the floats), then max of the bidder index among
//     exact-maximum bidders.  Deterministic PER KERNEL: the same (x, y) give the same assignment, bit for bit (rules stated in
//     oracle/emd_ref.py).  The training STEP around it is bit-reproducible only under cfg.deterministic (PF_TRAIN_DETERMINISTIC:
//     BatchNorm statistics as exact sums, ordered gathers instead of float atomics) - in the default mode x itself varies by
//     ~1e-7 from run to run, and a discrete assignment turns that into different matchings.
//
// Arithmetic follows the reference text: value = float((3.0 - (double)sqrtf(d2)) - (double)price)
// (the literal 3.0 in cu:146 is a double), d2 unfused fp32.
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"

namespace {

constexpr int EMD_THREADS = 1024;
constexpr int EMD_NMAX_LDS = 4096;      // y + price resident in LDS up to this n (64 KiB)
constexpr int EMD_NMAX_ALL = 2048;      // the WHOLE auction state resident in LDS up to this n (11 n words = 88 KiB)

struct EmdArgs {
    const float* x;          // xyz1 [B,n,3] prediction
    const float* y;          // xyz2 [B,n,3] ground truth
    float* dist;             // [B,n]
    int* assignment;         // [B,n]   in/out (-1 = unassigned)
    int* assignment_inv;     // [B,n]   in/out
    float* price;            // [B,n]   in/out
    int* bid;                // [B,n]   scratch
    float* bid_inc;          // [B,n]   scratch
    unsigned* max_inc_bits;  // [B,n]   scratch (caller's max_increments buffer, zero = empty)
    int* max_idx;            // [B,n]   scratch
    int* unass_idx;          // [B,n]   scratch: compact list of unassigned points
    int n, iters;
    float eps;
};

// Reduction of the lanes' (best, second-best, argbest) triples over a wave, result wave-uniform: best = largest value, idx =
// smallest index among the lanes holding it, better = largest of everything else (the other lanes' best values, the winner
// lane's own second).  This is the fold of the pairwise merge
//     take (obest, oidx) if obest > best or (obest == best and oidx < idx);  better = max(min(best, obest), max(better, obetter))
// over the lanes (a total order: value descending, index ascending; lanes hold distinct indices), done
// with DPP row rotates + 4 readlanes per reduction instead of 18 dependent ds_bpermute shuffles.  (No measurable change of the 0.87 ms a
// far-off 32 x 1024 prediction takes, tools/time_emd.py: a bid is bound by its 16 objects per lane x ~26 operations.)
template <int CTRL>
__device__ __forceinline__ float emd_dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float emd_wave_max(float v) {              // values are never NaN here
    v = fmaxf(v, emd_dpp_f<0x128>(v)); v = fmaxf(v, emd_dpp_f<0x124>(v));
    v = fmaxf(v, emd_dpp_f<0x122>(v)); v = fmaxf(v, emd_dpp_f<0x121>(v));
    const int b = __builtin_bit_cast(int, v);
    return fmaxf(fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))),
                 fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48))));
}
__device__ __forceinline__ unsigned emd_wave_min_u(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false));
    return min(min((unsigned)__builtin_amdgcn_readlane((int)v, 0), (unsigned)__builtin_amdgcn_readlane((int)v, 16)),
               min((unsigned)__builtin_amdgcn_readlane((int)v, 32), (unsigned)__builtin_amdgcn_readlane((int)v, 48)));
}
__device__ __forceinline__ void tri_wave(float& best, float& better, int& idx) {
    const float vmax = emd_wave_max(best);
    const unsigned widx = emd_wave_min_u(best == vmax ? (unsigned)idx : 0xffffffffu);
    const bool winner = best == vmax && (unsigned)idx == widx;
    const float second = emd_wave_max(winner ? better : best);         // best >= better in every lane
    best = vmax; idx = (int)widx; better = second;
}

// MODE 0: state in global memory; 1: y + price in LDS; 2: the whole state in LDS (n <= EMD_NMAX_ALL) - every phase of every
// iteration then runs out of LDS: the 50 x 5 barrier-separated phases of a 1024-point auction were bound by the latency of
// their global loads / atomics (72 us per iteration), not by the arithmetic.
template <int MODE>
__global__ __launch_bounds__(EMD_THREADS) void emd_auction_kernel(EmdArgs a) {
    constexpr bool IN_LDS = MODE >= 1, ALL = MODE == 2;
    constexpr int NY = ALL ? EMD_NMAX_ALL : EMD_NMAX_LDS;
    __shared__ float sy[IN_LDS ? 3 * NY : 3];
    __shared__ float sprice[IN_LDS ? NY : 1];
    __shared__ int sstate[ALL ? 7 * EMD_NMAX_ALL : 1];
    __shared__ int ucount;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n;
    const size_t o0 = (size_t)b * n;
    const float* x = a.x + o0 * 3;
    const float* yg = a.y + o0 * 3;
    int* assignment_g = a.assignment + o0;
    int* assignment_inv_g = a.assignment_inv + o0;
    float* priceg = a.price + o0;
    int* assignment = ALL ? sstate : assignment_g;
    int* assignment_inv = ALL ? sstate + EMD_NMAX_ALL : assignment_inv_g;
    int* bid = ALL ? sstate + 2 * EMD_NMAX_ALL : a.bid + o0;
    float* bid_inc = ALL ? reinterpret_cast<float*>(sstate + 3 * EMD_NMAX_ALL) : a.bid_inc + o0;
    unsigned* maxb = ALL ? reinterpret_cast<unsigned*>(sstate + 4 * EMD_NMAX_ALL) : a.max_inc_bits + o0;
    int* max_idx = ALL ? sstate + 5 * EMD_NMAX_ALL : a.max_idx + o0;
    int* ulist = ALL ? sstate + 6 * EMD_NMAX_ALL : a.unass_idx + o0;
    if (ALL)
        for (int i = tid; i < n; i += EMD_THREADS) { assignment[i] = assignment_g[i]; assignment_inv[i] = assignment_inv_g[i]; }

    if (IN_LDS) {
        for (int i = tid; i < 3 * n; i += EMD_THREADS) sy[i] = yg[i];
        for (int i = tid; i < n; i += EMD_THREADS) sprice[i] = priceg[i];
    }
    for (int i = tid; i < n; i += EMD_THREADS) { maxb[i] = 0u; max_idx[i] = -1; }
    __syncthreads();
    const float* yy = IN_LDS ? sy : yg;
    float* price = IN_LDS ? sprice : priceg;

    for (int it = 0; it < a.iters; ++it) {
        const bool last = it == a.iters - 1;
        if (tid == 0) ucount = 0;
        __syncthreads();
        for (int i = tid; i < n; i += EMD_THREADS)
            if (assignment[i] == -1) ulist[atomicAdd(&ucount, 1)] = i;
        __syncthreads();
        const int U = ucount;
        if (U == 0) break;                                      // uniform: nothing left to assign

        // ---- bid: one wave per unassigned point
        for (int u = wave; u < U; u += EMD_THREADS / 64) {
            const int i = ulist[u];
            const float x1 = x[i * 3 + 0], y1 = x[i * 3 + 1], z1 = x[i * 3 + 2];
            float tbest = -1e9f, tbetter = -1e9f;
            int tidx = 0x7fffffff;
            for (int k = lane; k < n; k += 64) {
                const float dx = __fsub_rn(yy[k * 3 + 0], x1), dy = __fsub_rn(yy[k * 3 + 1], y1),
                            dz = __fsub_rn(yy[k * 3 + 2], z1);
                const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                const float v = (float)((3.0 - (double)sqrtf(d2)) - (double)price[k]);
                if (v > tbest) { tbetter = tbest; tbest = v; tidx = k; }
                else if (v > tbetter) tbetter = v;
            }
            tri_wave(tbest, tbetter, tidx);
            if (lane == 0) {
                // a NaN / inf prediction row makes every value NaN or -inf: no `v > tbest` ever fires and the index stays at
                // its sentinel.  Bid on a valid object (the point's own index) with the minimum increment instead of
                // indexing out of bounds: the NaN then reaches dist and the loss, where TrainerModule.training_step's NaN
                // guard (train_pu1k.py:71-73) handles it.  (The reference starts best_i at -1: a one-element underrun.)
                if ((unsigned)tidx >= (unsigned)n) { tidx = i; tbest = tbetter = 0.f; }
                const float inc = __fadd_rn(__fsub_rn(tbest, tbetter), a.eps);
                bid[i] = tidx;
                bid_inc[i] = inc;
                atomicMax(&maxb[tidx], __float_as_uint(inc));
            }
        }
        __syncthreads();
        // ---- winner of each object: largest index among the bidders holding the exact maximum increment
        for (int u = tid; u < U; u += EMD_THREADS) {
            const int i = ulist[u], o = bid[i];
            if (__float_as_uint(bid_inc[i]) == maxb[o]) atomicMax(&max_idx[o], i);
        }
        __syncthreads();
        // ---- assign (winner takes the object, previous owner is evicted; last iteration: everyone is forced)
        for (int u = tid; u < U; u += EMD_THREADS) {
            const int i = ulist[u], o = bid[i];
            if (last || max_idx[o] == i) {
                if (!last) {
                    const int prev = assignment_inv[o];
                    if (prev != -1) assignment[prev] = -1;
                }
                assignment_inv[o] = i;
                assignment[i] = o;
                atomicAdd(&price[o], bid_inc[i]);
            }
        }
        __syncthreads();
        for (int u = tid; u < U; u += EMD_THREADS) {
            const int o = bid[ulist[u]];
            maxb[o] = 0u;
            max_idx[o] = -1;
        }
        __syncthreads();
    }
    __syncthreads();
    // ---- squared distance to the assigned ground-truth point (cu:217-226) + state write-back
    for (int i = tid; i < n; i += EMD_THREADS) {
        const int k = assignment[i];
        const float dx = __fsub_rn(x[i * 3 + 0], yy[k * 3 + 0]), dy = __fsub_rn(x[i * 3 + 1], yy[k * 3 + 1]),
                    dz = __fsub_rn(x[i * 3 + 2], yy[k * 3 + 2]);
        a.dist[o0 + i] = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        if (IN_LDS) priceg[i] = sprice[i];
        if (ALL) { assignment_g[i] = k; assignment_inv_g[i] = assignment_inv[i]; }
    }
}

// ---- the same auction with G workgroups per sample ------------------------------------------------------------------
// One workgroup per sample leaves 224 of 256 CUs idle at the training batch (32 x 1024 points), and while the prediction is
// still far from the target most points stay unassigned for all 50 iterations: ~30 000 point-bids x 1024 objects per sample,
// 3 ms on one CU.  Here a sample's points are split over G workgroups (consecutive block indices); ground truth and prices
// are copied to LDS per iteration, bids of a workgroup's own points stay in its LDS, and only the per-object state that
// other workgroups need (maximum increment, winner, owner, price, assignment) is global, accessed with agent-scope atomics.
// Grid barriers per iteration: after bidding, [after the winner vote,] after assignment - an arrival counter per sample,
// thread 0 spins on it.  With 64-bit vote words (increment bits << 32 | bidder index, one atomic max) the vote phase and its
// barrier are gone: two barriers per iteration.  The per-object vote arrays are double-buffered by iteration parity, so an iteration's entries are
// cleared during the NEXT iteration (by the workgroup that wrote them) instead of behind a fourth barrier.
// Same arithmetic and tie rules as the single-workgroup kernel: identical assignment.
// Progress: the host launches this kernel only when the device can hold all B * G workgroups at once (pf_emd_forward_ex:
// occupancy of this kernel x CU count); the spin is bounded anyway - on timeout the sample's distances are written as NaN,
// the status word counts the sample and the grid drains.  Two PROCESSES sharing one GPU with full-size batches can still starve
// each other's barriers (the occupancy query knows nothing about the other process; seen in a 2-rank rehearsal on one
// device): such callers pass groups = 1 (loss.EarthMoverDistance(groups=1)), and a starved run is reported, not silent.
constexpr int EMDC_NMAX = 2048;
// the barrier words are cleared by a kernel, not by hipMemsetAsync: inside a captured hipGraph a memset node was observed to
// race with the kernel node that follows it (arrival counters cleared under the running auction -> missed barriers)
__global__ __launch_bounds__(256) void emd_zero_kernel(unsigned* p, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = 0u;
}
struct EmdCoopArgs {
    const float* x; const float* y;
    float* dist; int* assignment; int* assignment_inv; float* price;
    unsigned* mb0; unsigned* mb1; int* mi0; int* mi1;     // [B,n] each: maximum increment bits / winner index, per parity
    unsigned long long* k0; unsigned long long* k1;        // KEY64: [B,n] (increment bits << 32 | bidder) per parity, instead
    unsigned* sync;                                        // [B,n] zeroed by the host: words [0..1] = one 64-bit barrier word per sample
                                                           // (arrivals | running total of unassigned points << 32)
    unsigned* status;                                      // nullable: [0] += 1 for every WORKGROUP whose grid barrier timed out
    int n, iters, G;
    float eps;
};
template <typename T>
__device__ __forceinline__ T ald(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ void ast(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- a wave's bid for one point with a cheap FILTER in front of the exact values (round 5) -----------------------------------
// A bid needs the exact largest and second largest value over all n objects (and the smallest index holding the largest).  The
// exact value costs ~30 instructions per object - unfused distance, a correctly rounded square root, the reference's double
// subtraction - and the auction is bound by exactly that.  Pass 1 evaluates every object APPROXIMATELY
// (fused distance, v_sqrt_f32, float subtraction: 12 instructions with the running top two) and keeps the 16 values of a lane in
// registers; the wave's second largest approximate value minus twice the error bound is a threshold below which no object can be
// among the exact top two; pass 2 evaluates exactly only the objects at or above it (typically 2 - 4 of 1024).  |approx - exact|:
// three float operations on magnitudes <= 4 + |price| against one rounding of the double expression, plus 1 ulp of the square
// root: < 2^-22 (4 + |v|); the bound used is 2^-20 (4 + |v|).  The result is bit for bit that of the plain scan (same values,
// same strict-`>` order inside a lane, same wave merge); NaN / inf rows produce no candidate, as before.
#ifndef PF_EMD_FILTER
#define PF_EMD_FILTER 1
#endif
__device__ __forceinline__ void emd_bid_scan(const float4* y4, int n, float x1, float y1, float z1, int lane, float& tbest,
                                             float& tbetter, int& tidx) {
    tbest = -1e9f; tbetter = -1e9f; tidx = 0x7fffffff;
#if PF_EMD_FILTER
    for (int k0 = 0; k0 < n; k0 += 64 * 16) {
        float va[16];
        float m1 = -__builtin_inff(), m2 = -__builtin_inff();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = k0 + j * 64 + lane;
            float v = -__builtin_inff();
            if (k < n) {
                const float4 o = y4[k];
                const float dx = o.x - x1, dy = o.y - y1, dz = o.z - z1;
                v = (3.0f - __builtin_amdgcn_sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)))) - o.w;
            }
            va[j] = v;
            const float lo = fminf(m1, v);
            m1 = fmaxf(m1, v);
            m2 = fmaxf(m2, lo);
        }
        int li = lane;
        tri_wave(m1, m2, li);                                       // m2: the second largest approximate value of the chunk
        const float thr = m2 - 0x1p-19f * (4.f + fabsf(m2));        // 2 x the error bound
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = k0 + j * 64 + lane;
            if (va[j] >= thr && k < n) {
                const float4 o = y4[k];
                const float dx = __fsub_rn(o.x, x1), dy = __fsub_rn(o.y, y1), dz = __fsub_rn(o.z, z1);
                const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                const float v = (float)((3.0 - (double)sqrtf(d2)) - (double)o.w);
                if (v > tbest) { tbetter = tbest; tbest = v; tidx = k; }
                else if (v > tbetter) tbetter = v;
            }
        }
    }
#else
    for (int k = lane; k < n; k += 64) {
        const float4 o = y4[k];
        const float dx = __fsub_rn(o.x, x1), dy = __fsub_rn(o.y, y1), dz = __fsub_rn(o.z, z1);
        const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        const float v = (float)((3.0 - (double)sqrtf(d2)) - (double)o.w);
        if (v > tbest) { tbetter = tbest; tbest = v; tidx = k; }
        else if (v > tbetter) tbetter = v;
    }
#endif
    tri_wave(tbest, tbetter, tidx);
}

#ifndef PF_EMD_SOLO
#define PF_EMD_SOLO 16         // a sample with at most this many unassigned points is finished by one of its workgroups alone (0: never)
#endif
#ifndef PF_EMD_XCD
#define PF_EMD_XCD 1            // 0: the linear mapping (A/B: 860 -> 843, 871 -> 850 us on the far case, the near case unchanged)
#endif
template <bool KEY64>
__global__ __launch_bounds__(EMD_THREADS) void emd_coop_kernel(EmdCoopArgs a) {
    __shared__ float4 sy4[EMDC_NMAX];                            // (y, price) of every object: one 16-byte LDS read per evaluation
    __shared__ int ulist[EMDC_NMAX], sbid[EMDC_NMAX], pbid[EMDC_NMAX];
    __shared__ float sinc[EMDC_NMAX];
    __shared__ int ucount, dead;
    // the one-workgroup endgame (PF_EMD_SOLO, below): the sample's whole state in this workgroup's LDS
    __shared__ int sassign[EMDC_NMAX], sainv[EMDC_NMAX];
    __shared__ unsigned long long skey[EMDC_NMAX];
    // workgroup -> (sample, slice).  Consecutive workgroup ids go round-robin over the 8 XCDs, so the linear mapping spreads the
    // G workgroups of a sample over G different XCDs; with PF_EMD_XCD the G workgroups of a sample share an XCD (its L2) when
    // the batch divides by 8
#if PF_EMD_XCD
    const int G = a.G;
    const int nsamp = gridDim.x / G;
    int b = blockIdx.x / G, w = blockIdx.x % G;
    if ((nsamp & 7) == 0) {
        const int x = blockIdx.x & 7, r = blockIdx.x >> 3;
        b = x * (nsamp >> 3) + r / G;
        w = r % G;
    }
#else
    const int G = a.G, b = blockIdx.x / G, w = blockIdx.x % G;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n;
    const size_t o0 = (size_t)b * n;
    const float* x = a.x + o0 * 3;
    int* assignment = a.assignment + o0;
    int* ainv = a.assignment_inv + o0;
    float* price = a.price + o0;
    unsigned* cnt = a.sync + o0;
    const int i0 = (int)((long long)w * n / G), i1 = (int)((long long)(w + 1) * n / G);
    unsigned nb = 0;
    if (tid == 0) dead = 0;
    // The barrier word is 64 bits: arrivals in the low half, the running total of unassigned points in the high half - a
    // workgroup adds (its count << 32 | 1) in ONE atomic, and the poll that sees every arrival has the sample's total with it
    // (a separate counter cost one more dependent memory round trip per iteration; a round trip is ~1.2 us, an iteration ~17).
    unsigned long long* cnt64 = reinterpret_cast<unsigned long long*>(cnt);
    __shared__ unsigned utot;
    auto barrier = [&](unsigned add_u) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ++nb;
        if (tid == 0) {
            atomicAdd(cnt64, ((unsigned long long)add_u << 32) | 1ull);
            const unsigned target = nb * (unsigned)G;
            int budget = 1 << 18;                       // ~0.2 s: a barrier normally completes in microseconds
            unsigned long long w = ald(cnt64);
            while ((unsigned)w < target && --budget > 0) { __builtin_amdgcn_s_sleep(2); w = ald(cnt64); }
            if (budget <= 0) dead = 1;
            utot = (unsigned)(w >> 32);
        }
        __syncthreads();
    };
    for (int i = tid; i < n; i += EMD_THREADS)
        sy4[i] = make_float4(a.y[(o0 + i) * 3 + 0], a.y[(o0 + i) * 3 + 1], a.y[(o0 + i) * 3 + 2], 0.f);
    for (int i = i0 + tid; i < i1; i += EMD_THREADS) {
        if (KEY64) { ast(a.k0 + o0 + i, 0ull); ast(a.k1 + o0 + i, 0ull); }
        else { ast(a.mb0 + o0 + i, 0u); ast(a.mb1 + o0 + i, 0u); ast(a.mi0 + o0 + i, -1); ast(a.mi1 + o0 + i, -1); }
    }
    barrier(0u);
    int pU = 0, solo_from = -1;
    unsigned uprev = 0;
    for (int it = 0; it < a.iters && !dead; ++it) {
        const bool last = it == a.iters - 1;
        const int par = it & 1;
        unsigned* mb = KEY64 ? nullptr : (par ? a.mb1 : a.mb0) + o0;
        int* mi = KEY64 ? nullptr : (par ? a.mi1 : a.mi0) + o0;
        unsigned* mbo = KEY64 ? nullptr : (par ? a.mb0 : a.mb1) + o0;
        int* mio = KEY64 ? nullptr : (par ? a.mi0 : a.mi1) + o0;
        unsigned long long* key = KEY64 ? (par ? a.k1 : a.k0) + o0 : nullptr;
        unsigned long long* keyo = KEY64 ? (par ? a.k0 : a.k1) + o0 : nullptr;
        // the slice's assignments are fetched together with the prices (one memory round trip, not two)
        const int as0 = i0 + tid < i1 ? ald(assignment + i0 + tid) : 0;
        for (int i = tid; i < n; i += EMD_THREADS) sy4[i].w = ald(price + i);
        if (tid == 0) ucount = 0;
        __syncthreads();
        if (i0 + tid < i1 && as0 == -1) ulist[atomicAdd(&ucount, 1)] = i0 + tid;
        for (int i = i0 + tid + EMD_THREADS; i < i1; i += EMD_THREADS)
            if (ald(assignment + i) == -1) ulist[atomicAdd(&ucount, 1)] = i;
        __syncthreads();
        const int U = ucount;
        // ---- bid: one wave per unassigned point of this workgroup's slice
        for (int u = wave; u < U; u += EMD_THREADS / 64) {
            const int i = ulist[u];
            const float x1 = x[i * 3 + 0], y1 = x[i * 3 + 1], z1 = x[i * 3 + 2];
            float tbest, tbetter;
            int tidx;
            emd_bid_scan(sy4, n, x1, y1, z1, lane, tbest, tbetter, tidx);
            if (lane == 0) {
                if ((unsigned)tidx >= (unsigned)n) { tidx = i; tbest = tbetter = 0.f; }      // NaN / inf row: see the kernel above
                const float inc = __fadd_rn(__fsub_rn(tbest, tbetter), a.eps);
                sbid[u] = tidx;
                sinc[u] = inc;
                // KEY64: one 64-bit max decides the object's winner - largest increment (positive floats order like their
                // bits), then largest bidder index: the vote phase and its barrier disappear
                if (KEY64) atomicMax(key + tidx, ((unsigned long long)__float_as_uint(inc) << 32) | (unsigned)i);
                else atomicMax(mb + tidx, __float_as_uint(inc));
            }
        }
        barrier((unsigned)U);
        const unsigned ucur = utot;                              // every workgroup reads the same total: its poll saw all arrivals
        if (ucur == uprev) break;                                // nothing left to assign in the whole sample (uniform)
        const unsigned u_it = ucur - uprev;                      // the sample's unassigned points at the start of this iteration
        uprev = ucur;
        if (!KEY64) {
            // ---- winner of each object: largest index among the bidders holding the exact maximum increment
            for (int u = tid; u < U; u += EMD_THREADS) {
                const int o = sbid[u];
                if (__float_as_uint(sinc[u]) == ald(mb + o)) atomicMax(mi + o, ulist[u]);
            }
            barrier(0u);
        }
        // ---- assign; clear the OTHER parity's entries this workgroup wrote in the previous iteration
        for (int u = tid; u < U; u += EMD_THREADS) {
            const int i = ulist[u], o = sbid[u];
            const bool won = KEY64 ? (int)(unsigned)(ald(key + o) & 0xffffffffull) == i : ald(mi + o) == i;
            if (last || won) {
                if (!last) {
                    const int prev = ald(ainv + o);
                    if (prev != -1) ast(assignment + prev, -1);
                }
                ast(ainv + o, i);
                ast(assignment + i, o);
                atomicAdd(price + o, sinc[u]);
            }
        }
        for (int u = tid; u < pU; u += EMD_THREADS) {
            if (KEY64) ast(keyo + pbid[u], 0ull);
            else { ast(mbo + pbid[u], 0u); ast(mio + pbid[u], -1); }
        }
        for (int u = tid; u < U; u += EMD_THREADS) pbid[u] = sbid[u];
        pU = U;
        barrier(0u);
        // The unassigned count of a sample never grows (a winner takes one point off the list and evicts at most one), and once
        // it is small an iteration is nothing but its fixed cost: the price copy, the list build and TWO grid barriers - 9 us
        // for a handful of bids (a prediction near its target has <= 40 unassigned points from the fifth iteration on and
        // still runs all 50).  From here on ONE workgroup finishes the sample out of its own LDS with workgroup barriers only.
        if (KEY64 && PF_EMD_SOLO > 0 && !last && u_it <= (unsigned)PF_EMD_SOLO) { solo_from = it + 1; break; }
    }
    if (solo_from >= 0 && !dead) {                               // uniform over the sample's workgroups (u_it is the barrier's total)
        if (w != 0) return;                                      // every write of the iterations so far is behind the barrier above
        for (int i = tid; i < n; i += EMD_THREADS) {
            sassign[i] = ald(assignment + i); sainv[i] = ald(ainv + i); sy4[i].w = ald(price + i); skey[i] = 0ull;
        }
        __syncthreads();
        for (int it = solo_from; it < a.iters; ++it) {
            const bool last = it == a.iters - 1;
            if (tid == 0) ucount = 0;
            __syncthreads();
            for (int i = tid; i < n; i += EMD_THREADS)
                if (sassign[i] == -1) ulist[atomicAdd(&ucount, 1)] = i;
            __syncthreads();
            const int U = ucount;
            if (U == 0) break;
            for (int u = wave; u < U; u += EMD_THREADS / 64) {     // the same bid, vote word and tie rules as above
                const int i = ulist[u];
                const float x1 = x[i * 3 + 0], y1 = x[i * 3 + 1], z1 = x[i * 3 + 2];
                float tbest, tbetter;
                int tidx;
                emd_bid_scan(sy4, n, x1, y1, z1, lane, tbest, tbetter, tidx);
                if (lane == 0) {
                    if ((unsigned)tidx >= (unsigned)n) { tidx = i; tbest = tbetter = 0.f; }
                    const float inc = __fadd_rn(__fsub_rn(tbest, tbetter), a.eps);
                    sbid[u] = tidx;
                    sinc[u] = inc;
                    atomicMax(skey + tidx, ((unsigned long long)__float_as_uint(inc) << 32) | (unsigned)i);
                }
            }
            __syncthreads();
            for (int u = tid; u < U; u += EMD_THREADS) {
                const int i = ulist[u], o = sbid[u];
                const bool won = (int)(unsigned)(skey[o] & 0xffffffffull) == i;
                if (last || won) {
                    if (!last) {
                        const int prev = sainv[o];
                        if (prev != -1) sassign[prev] = -1;
                    }
                    sainv[o] = i;
                    sassign[i] = o;
                    atomicAdd(&sy4[o].w, sinc[u]);
                }
            }
            __syncthreads();
            for (int u = tid; u < U; u += EMD_THREADS) skey[sbid[u]] = 0ull;      // read again only behind the next iteration's barriers
        }
        __syncthreads();
        for (int i = tid; i < n; i += EMD_THREADS) {
            const int k = sassign[i];
            assignment[i] = k; ainv[i] = sainv[i]; price[i] = sy4[i].w;
            float d = __builtin_nanf("");
            if ((unsigned)k < (unsigned)n) {
                const float dx = __fsub_rn(x[i * 3 + 0], sy4[k].x), dy = __fsub_rn(x[i * 3 + 1], sy4[k].y),
                            dz = __fsub_rn(x[i * 3 + 2], sy4[k].z);
                d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            }
            a.dist[o0 + i] = d;
        }
        return;
    }
    // ---- squared distance to the assigned ground-truth point (cu:217-226)
    for (int i = i0 + tid; i < i1; i += EMD_THREADS) {
        const int k = ald(assignment + i);
        float d = __builtin_nanf("");
        if (!dead && (unsigned)k < (unsigned)n) {
            const float dx = __fsub_rn(x[i * 3 + 0], sy4[k].x), dy = __fsub_rn(x[i * 3 + 1], sy4[k].y),
                        dz = __fsub_rn(x[i * 3 + 2], sy4[k].z);
            d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        }
        a.dist[o0 + i] = d;
    }
    // a timed-out barrier is REPORTED by EVERY workgroup that saw it (workgroup 0 of the sample may have got through its last
    // barrier while a peer timed out: that peer's slice of dist is NaN): the host raises on any non-zero count at its next
    // synchronisation point; the NaN distances above only keep a consumer from using the partial assignment silently
    if (dead && tid == 0 && a.status) atomicAdd(a.status, 1u);
}

// ---- round 5: the multi-workgroup auction with REPLICATED state - one grid barrier per iteration instead of two ---------------
// emd_coop_kernel keeps a sample's assignment / owner / price arrays in global memory: every iteration its G workgroups reload
// the prices, bid, meet at a barrier, resolve winners with global atomics and assign through global memory, and meet again -
// ~9 us of round trips per iteration for what is, from the fifth iteration on, a handful of bids.  Here EVERY workgroup of the
// sample keeps the WHOLE state (assignment, owner, price, vote words) in its own LDS and applies ALL bids of the sample to it:
// the update is a deterministic function of the iteration's bid records, so the G copies stay identical.  Per iteration a
// workgroup bids for the unassigned points of its slice, publishes one 8-byte record per bid
// (increment bits << 32 | valid | object << 16 | bidder; slots of its own slice of a per-parity record array, stale slots
// cleared), meets the others at ONE barrier, reads the sample's n record slots (8 bytes per thread) and resolves winners and
// assignments in LDS.  No price reload, no global atomics, no second barrier.  Same bids, vote words and tie rules as the
// other kernels: identical assignment (tests/test_gpu_losses.py compares all three with the oracle).
#ifndef PF_EMD_REPL
#define PF_EMD_REPL 1
#endif
__global__ __launch_bounds__(EMD_THREADS) void emd_repl_kernel(EmdCoopArgs a) {
    __shared__ float4 sy4[EMDC_NMAX];
    __shared__ int ulist[EMDC_NMAX], sbid[EMDC_NMAX];
    __shared__ float sinc[EMDC_NMAX];
    __shared__ int sassign[EMDC_NMAX], sainv[EMDC_NMAX];
    __shared__ unsigned long long skey[EMDC_NMAX];
    __shared__ int ucount, dead;
    __shared__ unsigned utot;
    const int G = a.G;
    const int nsamp = gridDim.x / G;
    int b = blockIdx.x / G, w = blockIdx.x % G;
    if ((nsamp & 7) == 0) {                                      // a sample's workgroups on one XCD (speed only)
        const int xc = blockIdx.x & 7, r = blockIdx.x >> 3;
        b = xc * (nsamp >> 3) + r / G;
        w = r % G;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n;
    const size_t o0 = (size_t)b * n;
    const float* x = a.x + o0 * 3;
    int* assignment = a.assignment + o0;
    int* ainv = a.assignment_inv + o0;
    float* price = a.price + o0;
    unsigned long long* cnt64 = reinterpret_cast<unsigned long long*>(a.sync + o0);
    const int i0 = (int)((long long)w * n / G), i1 = (int)((long long)(w + 1) * n / G);
    unsigned nb = 0;
    if (tid == 0) dead = 0;
    auto barrier = [&](unsigned add_u) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ++nb;
        if (tid == 0) {
            atomicAdd(cnt64, ((unsigned long long)add_u << 32) | 1ull);
            const unsigned target = nb * (unsigned)G;
            int budget = 1 << 18;
            unsigned long long v = ald(cnt64);
            while ((unsigned)v < target && --budget > 0) { __builtin_amdgcn_s_sleep(1); v = ald(cnt64); }
            if (budget <= 0) dead = 1;
            utot = (unsigned)(v >> 32);
        }
        __syncthreads();
    };
    for (int i = tid; i < n; i += EMD_THREADS) {
        sy4[i] = make_float4(a.y[(o0 + i) * 3 + 0], a.y[(o0 + i) * 3 + 1], a.y[(o0 + i) * 3 + 2], price[i]);
        sassign[i] = assignment[i]; sainv[i] = ainv[i]; skey[i] = 0ull;
    }
    for (int i = i0 + tid; i < i1; i += EMD_THREADS) { ast(a.k0 + o0 + i, 0ull); ast(a.k1 + o0 + i, 0ull); }
    barrier(0u);
    int pU0 = 0, pU1 = 0, solo_from = -1;
    unsigned uprev = 0;
    bool done = false;
    for (int it = 0; it < a.iters && !dead; ++it) {
        const bool last = it == a.iters - 1;
        const int par = it & 1;
        unsigned long long* rec = (par ? a.k1 : a.k0) + o0;
        if (tid == 0) ucount = 0;
        __syncthreads();
        for (int i = i0 + tid; i < i1; i += EMD_THREADS)
            if (sassign[i] == -1) ulist[atomicAdd(&ucount, 1)] = i;
        __syncthreads();
        const int U = ucount;
        for (int u = wave; u < U; u += EMD_THREADS / 64) {
            const int i = ulist[u];
            float tbest, tbetter;
            int tidx;
            emd_bid_scan(sy4, n, x[i * 3 + 0], x[i * 3 + 1], x[i * 3 + 2], lane, tbest, tbetter, tidx);
            if (lane == 0) {
                if ((unsigned)tidx >= (unsigned)n) { tidx = i; tbest = tbetter = 0.f; }      // NaN / inf row: see emd_auction_kernel
                const float inc = __fadd_rn(__fsub_rn(tbest, tbetter), a.eps);
                ast(rec + i0 + u, ((unsigned long long)__float_as_uint(inc) << 32) | 0x80000000ull | ((unsigned long long)tidx << 16) |
                                      (unsigned long long)i);
            }
        }
        const int pUp = par ? pU1 : pU0;                          // slots this workgroup filled two iterations ago and does not refill
        for (int u = U + tid; u < pUp; u += EMD_THREADS) ast(rec + i0 + u, 0ull);
        if (par) pU1 = U; else pU0 = U;
        barrier((unsigned)U);
        const unsigned ucur = utot;
        if (ucur == uprev) { done = true; break; }               // nothing left to assign in the whole sample (uniform)
        const unsigned u_it = ucur - uprev;
        uprev = ucur;
        // ---- every workgroup applies every bid of the sample to its own copy of the state
        unsigned long long r[(EMDC_NMAX + EMD_THREADS - 1) / EMD_THREADS];
#pragma unroll
        for (int q = 0; q < (EMDC_NMAX + EMD_THREADS - 1) / EMD_THREADS; ++q) {
            const int t = tid + q * EMD_THREADS;
            r[q] = t < n ? ald(rec + t) : 0ull;
        }
#pragma unroll
        for (int q = 0; q < (EMDC_NMAX + EMD_THREADS - 1) / EMD_THREADS; ++q)
            if (r[q] & 0x80000000ull) atomicMax(skey + (int)((r[q] >> 16) & 0x7fffu), (r[q] & 0xffffffff00000000ull) | (r[q] & 0xffffull));
        __syncthreads();
#pragma unroll
        for (int q = 0; q < (EMDC_NMAX + EMD_THREADS - 1) / EMD_THREADS; ++q)
            if (r[q] & 0x80000000ull) {
                const int o = (int)((r[q] >> 16) & 0x7fffu), i = (int)(r[q] & 0xffffu);
                if (last || (int)(unsigned)(skey[o] & 0xffffffffull) == i) {
                    const float inc = __uint_as_float((unsigned)(r[q] >> 32));
                    if (!last) {
                        const int prev = sainv[o];
                        if (prev != -1) sassign[prev] = -1;
                        sy4[o].w = __fadd_rn(sy4[o].w, inc);           // one winner per object
                    } else
                        atomicAdd(&sy4[o].w, inc);                      // the forced last round: several bidders may take one object
                    sainv[o] = i;
                    sassign[i] = o;
                }
            }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < (EMDC_NMAX + EMD_THREADS - 1) / EMD_THREADS; ++q)
            if (r[q] & 0x80000000ull) skey[(int)((r[q] >> 16) & 0x7fffu)] = 0ull;     // read again behind the next iteration's barriers
        // few unassigned points left in the sample (their number never grows): workgroup 0 finishes alone, without grid barriers
        if (PF_EMD_SOLO > 0 && !last && u_it <= (unsigned)PF_EMD_SOLO) { solo_from = it + 1; break; }
    }
    (void)done;
    int w0 = i0, w1 = i1;                                        // the slice of the outputs this workgroup writes
    if (solo_from >= 0 && !dead) {
        if (w != 0) return;
        w0 = 0; w1 = n;
        for (int it = solo_from; it < a.iters; ++it) {
            const bool last = it == a.iters - 1;
            if (tid == 0) ucount = 0;
            __syncthreads();
            for (int i = tid; i < n; i += EMD_THREADS)
                if (sassign[i] == -1) ulist[atomicAdd(&ucount, 1)] = i;
            __syncthreads();
            const int U = ucount;
            if (U == 0) break;
            for (int u = wave; u < U; u += EMD_THREADS / 64) {
                const int i = ulist[u];
                float tbest, tbetter;
                int tidx;
                emd_bid_scan(sy4, n, x[i * 3 + 0], x[i * 3 + 1], x[i * 3 + 2], lane, tbest, tbetter, tidx);
                if (lane == 0) {
                    if ((unsigned)tidx >= (unsigned)n) { tidx = i; tbest = tbetter = 0.f; }
                    const float inc = __fadd_rn(__fsub_rn(tbest, tbetter), a.eps);
                    sbid[u] = tidx;
                    sinc[u] = inc;
                    atomicMax(skey + tidx, ((unsigned long long)__float_as_uint(inc) << 32) | (unsigned)i);
                }
            }
            __syncthreads();
            for (int u = tid; u < U; u += EMD_THREADS) {
                const int i = ulist[u], o = sbid[u];
                if (last || (int)(unsigned)(skey[o] & 0xffffffffull) == i) {
                    if (!last) {
                        const int prev = sainv[o];
                        if (prev != -1) sassign[prev] = -1;
                    }
                    sainv[o] = i;
                    sassign[i] = o;
                    atomicAdd(&sy4[o].w, sinc[u]);
                }
            }
            __syncthreads();
            for (int u = tid; u < U; u += EMD_THREADS) skey[sbid[u]] = 0ull;
        }
    }
    __syncthreads();
    // ---- outputs: the slice's state and its squared distances to the assigned ground-truth points (cu:217-226)
    for (int i = w0 + tid; i < w1; i += EMD_THREADS) {
        const int k = sassign[i];
        assignment[i] = k; ainv[i] = sainv[i]; price[i] = sy4[i].w;
        float d = __builtin_nanf("");
        if (!dead && (unsigned)k < (unsigned)n) {
            const float dx = __fsub_rn(x[i * 3 + 0], sy4[k].x), dy = __fsub_rn(x[i * 3 + 1], sy4[k].y),
                        dz = __fsub_rn(x[i * 3 + 2], sy4[k].z);
            d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        }
        a.dist[o0 + i] = d;
    }
    if (dead && tid == 0 && a.status) atomicAdd(a.status, 1u);
}

__global__ __launch_bounds__(256) void emd_grad_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                      const float* __restrict__ gdist, const int* __restrict__ idx,
                                                      float* __restrict__ gx, int n, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const long long b = t / n;
    const int j = idx[t];
    const float g = gdist[t] * 2.f;                               // cu:295
    if ((unsigned)j >= (unsigned)n) {                             // an assignment that is not an index: loud, never out of bounds
#pragma unroll
        for (int c = 0; c < 3; ++c) gx[t * 3 + c] = __builtin_nanf("");
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) gx[t * 3 + c] += g * (x[t * 3 + c] - y[(b * n + j) * 3 + c]);
}

}  // namespace

// groups: workgroups per sample of the multi-workgroup auction - 0 = choose from the device (below), 1 = the one-workgroup
// kernel (always safe: no inter-workgroup waits), g > 1 = at most g.  status: nullable device word; the multi-workgroup
// kernel adds 1 to it for every workgroup whose grid barrier timed out (that workgroup's slice of dist is NaN): any non-zero
// value is a failure.  The occupancy test below counts THIS kernel alone: kernels of the same process on other streams (a
// training side stream, a second captured graph in flight) take CUs too - the bounded spin + status word cover that case.
// The multi-workgroup kernel waits on grid barriers, so ALL its B * G workgroups must be resident at once.  That is decided
// here from what the device can hold for THIS kernel - hipOccupancyMaxActiveBlocksPerMultiprocessor (its LDS / VGPR / wave
// footprint) x the CU count - not assumed: G is the largest power of two with B * G <= resident capacity (and at most one
// workgroup per CU, and >= 64 points per workgroup); if not even G = 2 fits, the one-workgroup kernel runs.  A caller that
// knows the device is shared with other processes passes groups = 1 (or a smaller cap).
extern "C" int pf_emd_forward_ex(const float* xyz1, const float* xyz2, float* dist, int* assignment, float* price,
                                 int* assignment_inv, int* bid, float* bid_increments, float* max_increments,
                                 int* unass_idx, int* max_idx, float eps, int iters, int B, int n, int groups,
                                 unsigned* status, void* stream) {
    if (!xyz1 || !xyz2 || !dist || !assignment || !price || !assignment_inv || !bid || !bid_increments ||
        !max_increments || !unass_idx || !max_idx)
        return PF_ERR_NULL;
    if (B <= 0 || n <= 0 || iters <= 0 || n > (1 << 20) || groups < 0) return PF_ERR_SHAPE;
    EmdArgs a{xyz1, xyz2, dist, assignment, assignment_inv, price, bid, bid_increments,
              reinterpret_cast<unsigned*>(max_increments), max_idx, unass_idx, n, iters, eps};
    hipStream_t s = (hipStream_t)stream;
    // 64-bit vote words need two of the caller's [B,n] scratch arrays back to back (8-byte aligned): true when they are
    // slices of one allocation, as puflow_amd.loss.emdFunction passes them; otherwise the three-barrier protocol
    const size_t bn = (size_t)B * n;
    const bool pair0 = bid_increments == max_increments + bn && (reinterpret_cast<size_t>(max_increments) & 7) == 0;
    const bool pair1 = bid == max_idx + bn && (reinterpret_cast<size_t>(max_idx) & 7) == 0;
    const bool key64 = pair0 && pair1;
    int G = 1;
    if (groups != 1 && n <= EMDC_NMAX && (n & 1) == 0 && (reinterpret_cast<size_t>(unass_idx) & 7) == 0) {   // 64-bit barrier word per sample
        int ncu = 0, dev = 0, per_cu = 0;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
        const hipError_t oc = key64
            ? (PF_EMD_REPL ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, emd_repl_kernel, EMD_THREADS, 0)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, emd_coop_kernel<true>, EMD_THREADS, 0))
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, emd_coop_kernel<false>, EMD_THREADS, 0);
        if (oc != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
        const long long resident = per_cu >= 1 ? (long long)ncu : 0;        // one workgroup per CU at most: each wants a whole CU's issue slots
        const int cap = groups > 1 ? groups : 16;
        while (2 * G <= cap && (long long)B * (2 * G) <= resident && n / (2 * G) >= 64) G *= 2;
    }
    if (G >= 2) {
        hipLaunchKernelGGL(emd_zero_kernel, dim3(64), dim3(256), 0, s, reinterpret_cast<unsigned*>(unass_idx), (long long)B * n);
        EmdCoopArgs c{xyz1, xyz2, dist, assignment, assignment_inv, price, reinterpret_cast<unsigned*>(max_increments),
                      reinterpret_cast<unsigned*>(bid_increments), max_idx, bid, nullptr, nullptr,
                      reinterpret_cast<unsigned*>(unass_idx), status, n, iters, G, eps};
        if (key64) {
            c.k0 = reinterpret_cast<unsigned long long*>(max_increments);
            c.k1 = reinterpret_cast<unsigned long long*>(max_idx);
            // each array pair holds 2 B n 32-bit words = B n 64-bit words: sample b's keys at word offset b n
            if (PF_EMD_REPL) hipLaunchKernelGGL(emd_repl_kernel, dim3(B * G), dim3(EMD_THREADS), 0, s, c);
            else hipLaunchKernelGGL(emd_coop_kernel<true>, dim3(B * G), dim3(EMD_THREADS), 0, s, c);
        } else
            hipLaunchKernelGGL(emd_coop_kernel<false>, dim3(B * G), dim3(EMD_THREADS), 0, s, c);
        return pf_last_launch_status();
    }
    if (n <= EMD_NMAX_ALL) hipLaunchKernelGGL(emd_auction_kernel<2>, dim3(B), dim3(EMD_THREADS), 0, s, a);
    else if (n <= EMD_NMAX_LDS) hipLaunchKernelGGL(emd_auction_kernel<1>, dim3(B), dim3(EMD_THREADS), 0, s, a);
    else hipLaunchKernelGGL(emd_auction_kernel<0>, dim3(B), dim3(EMD_THREADS), 0, s, a);
    return pf_last_launch_status();
}

// the reference's emd.forward argument list (emd.cpp:14-31): workgroups per sample chosen from the device, no status word
extern "C" int pf_emd_forward(const float* xyz1, const float* xyz2, float* dist, int* assignment, float* price,
                              int* assignment_inv, int* bid, float* bid_increments, float* max_increments,
                              int* unass_idx, int* max_idx, float eps, int iters, int B, int n, void* stream) {
    return pf_emd_forward_ex(xyz1, xyz2, dist, assignment, price, assignment_inv, bid, bid_increments, max_increments,
                             unass_idx, max_idx, eps, iters, B, n, 0, nullptr, stream);
}

extern "C" int pf_emd_backward(const float* xyz1, const float* xyz2, float* gradxyz, const float* graddist,
                               const int* idx, int B, int n, void* stream) {
    if (!xyz1 || !xyz2 || !gradxyz || !graddist || !idx) return PF_ERR_NULL;
    if (B <= 0 || n <= 0) return PF_ERR_SHAPE;
    const long long total = (long long)B * n;
    hipLaunchKernelGGL(emd_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xyz1,
                       xyz2, graddist, idx, gradxyz, n, total);
    return pf_last_launch_status();
}
