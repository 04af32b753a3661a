// Patch pipeline operators around the network: farthest point sampling and the
// large-K patch kNN.  Replace pointnet2_ops.furthest_point_sample (modules/utils/patch.py:102,156)
// and knn_cuda.KNN(k=256) (patch.py:33,107).  Both third-party packages are un-vendored and
// unpinned in the reference (docker/Dockerfile:47-49), so the semantics are defined here and in
// oracle/patch_ref.py:  FPS starts at index 0, squared distances are unfused fp32
// ((dx*dx)+(dy*dy))+(dz*dz), the farthest point is the FIRST maximum (smallest index);
// kNN is ordered by (distance, index) like pf_knn.
#include <cstdlib>
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

// ---- cooperative FPS: G workgroups per cloud, every point and its running min-distance live in REGISTERS ----
// The single-workgroup kernel above re-reads 16 B per point from L2 on every one of the npoint sequential steps
// (99 840 points -> 1.6 MB per step through ONE CU's 64 B/clk texture path: 19 us per step, 388 ms for the CLI's
// 99 840 -> 20 024 merge, 99.7 % of its per-cloud GPU time).  Here a cloud is split over G <= 32 workgroups
// (<= 32 points per thread), a step is: local arg-max in registers -> one 64-bit candidate per WAVE
// (distance bits << 32 | ~index, so an integer max is "farthest, then smallest index") published with an
// agent-scope atomic store into a 4-deep ring of G x 4 slots -> wave 0 of every workgroup polls the slots of the
// step (agent-scope atomic loads) and reduces them.  The 64-bit word IS the whole message (coordinates are re-read
// from the read-only input), so relaxed ordering suffices: no L2 write-back / invalidate per step, which is what an
// acquire / release pair costs at agent scope on this chip.  No read-modify-write atomics, no counters: a slot is
// "filled" when it carries the current step's 2-bit tag (the same slot held step j-4's word before): see the kernel.
// Progress: blocks are dispatched in index order and a cloud's workgroups are contiguous, so the lowest
// unfinished cloud always has all its workgroups resident; a bounded spin + abort guarantees the grid drains
// even if that assumption were ever violated.  A cloud's status word is 0 only after all of its steps completed -
// pf_fps_scratch_layout tells the caller where that word is, puflow_amd.ops.furthest_point_sample checks it and raises.
#ifndef PF_FPSC_T
#define PF_FPSC_T 256
#endif
constexpr int FPSC_T = PF_FPSC_T;
constexpr int FPSC_GMAX = 32;
constexpr int FPSC_SLOTS = FPSC_GMAX * (FPSC_T / 64);    // one slot per wave of a cloud
constexpr int FPSC_RING = 4 * FPSC_SLOTS;                // 64-bit words per cloud, then the status word
static_assert(FPSC_SLOTS <= 128, "a lane polls two slots");
constexpr unsigned FPSC_SPIN_MAX = 1u << 24;
// status word of a cloud (ring[FPSC_RING]): 2 = not finished (set by fps_init_kernel), 1 = aborted, 0 = complete
constexpr unsigned long long FPSC_ST_DONE = 0ull, FPSC_ST_ABORT = 1ull, FPSC_ST_INIT = 2ull;

__device__ __forceinline__ unsigned long long fps_ld(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fps_st(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave-wide reductions without LDS traffic: rotate-reduce inside the 16-lane rows (DPP row_ror), then the four row results
// through SGPRs.  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int wave_max_i32(int v) {
    v = max(v, dpp_i<0x128>(v)); v = max(v, dpp_i<0x124>(v)); v = max(v, dpp_i<0x122>(v)); v = max(v, dpp_i<0x121>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)dpp_i<0x128>((int)v)); v = max(v, (unsigned)dpp_i<0x124>((int)v));
    v = max(v, (unsigned)dpp_i<0x122>((int)v)); v = max(v, (unsigned)dpp_i<0x121>((int)v));
    return max(max((unsigned)__builtin_amdgcn_readlane((int)v, 0), (unsigned)__builtin_amdgcn_readlane((int)v, 16)),
               max((unsigned)__builtin_amdgcn_readlane((int)v, 32), (unsigned)__builtin_amdgcn_readlane((int)v, 48)));
}

// A step: every WAVE reduces its own points in registers (DPP) and publishes one 64-bit word (distance bits << 32 | tag |
// ~index: an integer max is "farthest, then smallest index") into its slot of a 4-deep ring; wave 0 of every workgroup reads
// the G x 4 slots of the step (two per lane), reduces them and hands the winner's coordinates to the other waves through LDS
// (double-buffered: ONE barrier per step).  Measured on the CLI's merge (99 840 -> 20 024, 25 workgroups), per step:
//   one word per workgroup, shuffles through LDS, two barriers (round 2's first version)   2.86 us
//   one word per wave, every wave reads all slots itself (no barrier at all)               2.95-3.1 us  (100 readers of the
//                                                                          same lines: the reads get in each other's way)
//   one word per wave, wave 0 reads (this kernel)                                          2.2-2.3 us
//   the same with the workgroup's four words merged in LDS first (25 slots)                2.56 us
//   ... waiting for the OWN slots only (wrong results, timing experiment)                  1.65 us  = what the store -> load
//                                                                          round trip through the fabric leaves of a step
// Tried and dropped: all workgroups of a cloud on one XCD (8x oversubscribed grid, HW_REG_XCC_ID, claim counter) - agent-scope
// atomics bypass the XCD's L2 wherever the peers run (2.5 us), workgroup-scope loads are served by the CU's own L1 and never
// see the peers' stores, and an L1 invalidate (buffer_inv sc1) per read costs 6.3 us per step; two staggered reads in flight
// (2.37 us); s_sleep between reads (no change); touching the whole cloud once per XCD before the loop (-0.06 us).
template <int PPT>
__global__ __launch_bounds__(FPSC_T) void fps_coop_kernel(const float* __restrict__ xyz, int N, int npoint, int G,
                                                          unsigned long long* __restrict__ ringbuf, long long ring_stride,
                                                          int* __restrict__ out) {
    constexpr int NW = FPSC_T / 64;
    __shared__ float s_l[2][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const float* p = xyz + (size_t)b * N * 3;
    unsigned long long* ring = ringbuf + (size_t)b * ring_stride;       // [4][FPSC_SLOTS] + status word
    unsigned long long* abort_w = ring + FPSC_RING;
    int* o = out + (size_t)b * npoint;
    const int S = G * NW;                                               // slots = waves of the cloud (<= 128: two per lane)

    float px[PPT], py[PPT], pz[PPT], md[PPT];
    int pi[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int i = (g * PPT + k) * FPSC_T + tid;                     // increasing in k: first maximum = smallest index
        pi[k] = i;
        const bool in = i < N;
        const int ic = in ? i : N - 1;
        px[k] = p[ic * 3 + 0]; py[k] = p[ic * 3 + 1]; pz[k] = p[ic * 3 + 2];
        md[k] = in ? 1e10f : -1.f;                                      // padding can never be the farthest point
    }
    if (g == 0 && tid == 0) o[0] = 0;
    float lx = p[0], ly = p[1], lz = p[2];
    for (int j = 1; j < npoint; ++j) {
        float best = -1.f;
        int besti = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const float d = fminf(md[k], sqd(px[k], py[k], pz[k], lx, ly, lz));
            md[k] = d;
            if (d > best) { best = d; besti = pi[k]; }
        }
        // distances are >= 0 or the -1 of padding: their bit patterns order like signed integers
        const int vb = __float_as_int(best);
        const int vmax = wave_max_i32(vb);
        const unsigned imin = ~wave_max_u32(vb == vmax ? ~(unsigned)besti : 0u);      // smallest index among the maxima
        unsigned long long* slot = ring + (j & 3) * FPSC_SLOTS;
        // word = distance bits << 32 | step tag (2 bits) | valid bit | ~index (29 bits).  The tag (j / 4) & 3 tells a
        // word of THIS step from the one the same slot held four steps ago, so a consumer can never take a stale
        // candidate whatever order two relaxed stores to different addresses become visible in (no slot clearing,
        // no release / acquire); within a step every word carries the same tag, so the integer max is unchanged.
        const unsigned tag = ((unsigned)(((j >> 2) & 3) << 1) | 1u) << 29;
        if (lane == 0) {
            const unsigned long long key =
                vmax < 0 ? (unsigned long long)tag : (((unsigned long long)(unsigned)vmax << 32) | tag | ((~imin) & 0x1fffffffu));
            fps_st(slot + g * NW + wave, key);
        }
        if (wave == 0) {
            unsigned long long k0, k1;
            unsigned spins = 0;
            bool dead = false;
            for (;;) {
                k0 = lane < S ? fps_ld(slot + lane) : (unsigned long long)tag;
                k1 = lane + 64 < S ? fps_ld(slot + lane + 64) : (unsigned long long)tag;
                if (!__any(((unsigned)k0 & (7u << 29)) != tag || ((unsigned)k1 & (7u << 29)) != tag)) break;
                if (++spins > FPSC_SPIN_MAX || (spins % 1024 == 0 && fps_ld(abort_w) == FPSC_ST_ABORT)) { dead = true; break; }
            }
            if (dead) {                                                     // uniform over the wave
                if (lane == 0) { fps_st(abort_w, FPSC_ST_ABORT); s_l[j & 1][3] = -1.f; }
            } else {
                const unsigned long long km = k0 > k1 ? k0 : k1;
                // every lane fetches ITS candidate's coordinates while the maximum is being reduced: the winner's are then
                // already in registers (one dependent L2 round trip less per step)
                const unsigned ci = (~(unsigned)km) & 0x1fffffffu;
                const unsigned cic = ci < (unsigned)N ? ci : 0u;
                const float cx = p[cic * 3 + 0], cy = p[cic * 3 + 1], cz = p[cic * 3 + 2];
                const unsigned hi = (unsigned)(km >> 32), lo = (unsigned)km;
                const unsigned hmax = wave_max_u32(hi);
                const unsigned lmax = wave_max_u32(hi == hmax ? lo : 0u);
                const int wl = __builtin_ctzll(__ballot(hi == hmax && lo == lmax));       // keys are distinct: one lane matches
                if (lane == wl) { s_l[j & 1][0] = cx; s_l[j & 1][1] = cy; s_l[j & 1][2] = cz; s_l[j & 1][3] = 1.f; }
                if (g == 0 && lane == 0) o[j] = (int)((~lmax) & 0x1fffffffu);
            }
        }
        // s_l is double-buffered: wave 0 rewrites this step's half only after the NEXT step's barrier, which every wave
        // reaches after it has read the half here
        __syncthreads();
        lx = s_l[j & 1][0]; ly = s_l[j & 1][1]; lz = s_l[j & 1][2];
        if (s_l[j & 1][3] < 0.f) return;                                  // uniform over the workgroup: aborted
    }
    // every step of this cloud completed (a step completes only when all of its waves published): mark the row valid
    if (g == 0 && tid == 0) fps_st(abort_w, FPSC_ST_DONE);
}

// ---- two samples per exchange ------------------------------------------------------------------------------------------
// A step of fps_coop_kernel is ~0.4 us of arithmetic and ~1.8 us of waiting for the words to cross the fabric.  Here every wave
// publishes its TWO best points (keys K_a > K_b), so a round knows the global best c1 AND the best of the rest c2.  After c1
// has been added, every min-distance can only shrink, i.e. every key can only fall; points nobody published lie below their
// wave's K_b, which lies below K2 = key(c2).  So if c2 itself is not touched by c1 - d(c2, c1) >= md(c2), computed exactly as
// the update would - and md(c2) > 0, so that c2 also stays above c1's own new key (0, index of c1) - c2 is still the largest
// key after the update: it IS the next sample of the sequential algorithm (keys are distinct, so ties are decided exactly as
// there: farthest, then smallest index), and the round emits both.  Otherwise it
// emits c1 alone, as before.  Every workgroup takes the same decision from the same words.  Bit-identical output; on the CLI's
// merge (99 840 -> 20 024) 1.59 samples per round: 46.0 -> 30.4 ms per cloud.
constexpr int FPSC_SLOTS2 = 2 * FPSC_SLOTS;
constexpr int FPSC_RING2 = 4 * FPSC_SLOTS2;              // status word of the two-sample kernel's ring

__device__ __forceinline__ unsigned long long fps_key(int vbits, unsigned idx, unsigned tag) {
    return vbits < 0 ? (unsigned long long)tag : (((unsigned long long)(unsigned)vbits << 32) | tag | ((~idx) & 0x1fffffffu));
}
__device__ __forceinline__ unsigned long long u64max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned long long u64min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {       // two 32-bit passes: high word, then low
    const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
    const unsigned hmax = wave_max_u32(hi);
    const unsigned lmax = wave_max_u32(hi == hmax ? lo : 0u);
    return ((unsigned long long)hmax << 32) | lmax;
}

template <int PPT>
__global__ __launch_bounds__(FPSC_T) void fps_coop2_kernel(const float* __restrict__ xyz, int N, int npoint, int G,
                                                           unsigned long long* __restrict__ ringbuf, long long ring_stride,
                                                           int* __restrict__ out) {
    constexpr int NW = FPSC_T / 64;
    __shared__ float s_l[2][8];                                         // c1 xyz, c2 xyz, accepted-2 flag, alive flag
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const float* p = xyz + (size_t)b * N * 3;
    unsigned long long* ring = ringbuf + (size_t)b * ring_stride;       // [4][FPSC_SLOTS2] + status word
    unsigned long long* abort_w = ring + FPSC_RING2;
    int* o = out + (size_t)b * npoint;
    const int S2 = 2 * G * NW;                                          // words per round (<= 256: four per lane)

    float px[PPT], py[PPT], pz[PPT], md[PPT];
    int pi[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int i = (g * PPT + k) * FPSC_T + tid;                     // increasing in k: first maximum = smallest index
        pi[k] = i;
        const bool in = i < N;
        const int ic = in ? i : N - 1;
        px[k] = p[ic * 3 + 0]; py[k] = p[ic * 3 + 1]; pz[k] = p[ic * 3 + 2];
        md[k] = in ? 1e10f : -1.f;                                      // padding can never be the farthest point
    }
    if (g == 0 && tid == 0) o[0] = 0;
    float l1x = p[0], l1y = p[1], l1z = p[2], l2x = 0.f, l2y = 0.f, l2z = 0.f;
    bool have2 = false;
    int j = 1, rounds = 0;
    for (int r = 0; j < npoint; ++r) {
        // update with the sample(s) of the previous round; per-lane best two by (distance, then smaller index)
        float b1 = -1.f, b2 = -1.f;
        int i1 = 0x7fffffff, i2 = 0x7fffffff;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            float d = fminf(md[k], sqd(px[k], py[k], pz[k], l1x, l1y, l1z));
            if (have2) d = fminf(d, sqd(px[k], py[k], pz[k], l2x, l2y, l2z));
            md[k] = d;
            const bool gt1 = d > b1, gt2 = d > b2;                      // indices grow with k: strict comparisons keep the smaller
            b2 = gt1 ? b1 : (gt2 ? d : b2);
            i2 = gt1 ? i1 : (gt2 ? pi[k] : i2);
            b1 = gt1 ? d : b1;
            i1 = gt1 ? pi[k] : i1;
        }
        // the wave's best two (distances are >= 0 or the -1 of padding: their bit patterns order like signed integers)
        const int vb = __float_as_int(b1);
        const int vmax = wave_max_i32(vb);
        const unsigned imin = ~wave_max_u32(vb == vmax ? ~(unsigned)i1 : 0u);
        const bool mine = vb == vmax && (unsigned)i1 == imin;           // the lane that holds the wave's best
        const int cb = mine ? __float_as_int(b2) : vb;
        const unsigned ci = mine ? (unsigned)i2 : (unsigned)i1;
        const int v2 = wave_max_i32(cb);
        const unsigned i2m = ~wave_max_u32(cb == v2 ? ~ci : 0u);
        unsigned long long* slot = ring + (r & 3) * FPSC_SLOTS2;
        const unsigned tag = ((unsigned)(((r >> 2) & 3) << 1) | 1u) << 29;      // see fps_coop_kernel
        if (lane == 0) {
            fps_st(slot + 2 * (g * NW + wave), fps_key(vmax, imin, tag));
            fps_st(slot + 2 * (g * NW + wave) + 1, fps_key(v2, i2m, tag));
        }
        if (wave == 0) {
            unsigned long long k[4];
            unsigned spins = 0;
            bool dead = false;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    k[t] = lane + 64 * t < S2 ? fps_ld(slot + lane + 64 * t) : (unsigned long long)tag;
                    ok = ok && ((unsigned)k[t] & (7u << 29)) == tag;
                }
                if (!__any(!ok)) break;
                if (++spins > FPSC_SPIN_MAX || (spins % 1024 == 0 && fps_ld(abort_w) == FPSC_ST_ABORT)) { dead = true; break; }
            }
            if (dead) {                                                     // uniform over the wave
                if (lane == 0) { fps_st(abort_w, FPSC_ST_ABORT); s_l[r & 1][7] = -1.f; }
            } else {
                // this lane's best two of its four words, then the wave's best two
                const unsigned long long a0 = u64max(k[0], k[1]), a1 = u64min(k[0], k[1]);
                const unsigned long long c0 = u64max(k[2], k[3]), c1m = u64min(k[2], k[3]);
                const unsigned long long f1 = u64max(a0, c0);
                const unsigned long long f2 = u64max(u64min(a0, c0), a0 > c0 ? a1 : c1m);
                // both candidates' coordinates are fetched while the maxima are reduced
                const unsigned q1 = (~(unsigned)f1) & 0x1fffffffu, q2 = (~(unsigned)f2) & 0x1fffffffu;
                const unsigned q1c = q1 < (unsigned)N ? q1 : 0u, q2c = q2 < (unsigned)N ? q2 : 0u;
                const float ax = p[q1c * 3 + 0], ay = p[q1c * 3 + 1], az = p[q1c * 3 + 2];
                const float bx = p[q2c * 3 + 0], by = p[q2c * 3 + 1], bz = p[q2c * 3 + 2];
                const unsigned long long K1 = wave_max_u64(f1);
                const bool own1 = f1 == K1;                                  // keys are distinct: one lane
                const unsigned long long K2 = wave_max_u64(own1 ? f2 : f1);
                const int w1 = __builtin_ctzll(__ballot(own1));
                const bool own2 = (own1 ? f2 : f1) == K2;
                const int w2 = __builtin_ctzll(__ballot(own2));
                const bool k2_second = w2 == w1;                             // K2 is the second word of the lane that owns K1
                auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
                const float c1x = rl(ax, w1), c1y = rl(ay, w1), c1z = rl(az, w1);
                const float c2x = k2_second ? rl(bx, w2) : rl(ax, w2), c2y = k2_second ? rl(by, w2) : rl(ay, w2),
                            c2z = k2_second ? rl(bz, w2) : rl(az, w2);
                const unsigned idx1 = (~(unsigned)K1) & 0x1fffffffu, idx2 = (~(unsigned)K2) & 0x1fffffffu;
                // c2 follows c1 at once iff adding c1 leaves its min-distance (the high word of K2) untouched
                const float md2 = __uint_as_float((unsigned)(K2 >> 32));
                // (md2 > 0: once every min-distance is 0 - more samples than distinct points - c1 keeps the largest key
                // (0, smallest index) after it has been added, and the sequential algorithm picks it again, not c2)
                const bool two = j + 1 < npoint && idx2 < (unsigned)N && idx1 < (unsigned)N && md2 > 0.f &&
                                 sqd(c2x, c2y, c2z, c1x, c1y, c1z) >= md2;
                if (lane == 0) {
                    float* sl = s_l[r & 1];
                    sl[0] = c1x; sl[1] = c1y; sl[2] = c1z; sl[3] = c2x; sl[4] = c2y; sl[5] = c2z;
                    sl[6] = two ? 1.f : 0.f; sl[7] = 1.f;
                    if (g == 0) { o[j] = (int)idx1; if (two) o[j + 1] = (int)idx2; }
                }
            }
        }
        // s_l is double-buffered (see fps_coop_kernel): one barrier per round
        __syncthreads();
        const float* sl = s_l[r & 1];
        if (sl[7] < 0.f) return;                                          // uniform over the workgroup: aborted
        l1x = sl[0]; l1y = sl[1]; l1z = sl[2]; l2x = sl[3]; l2y = sl[4]; l2z = sl[5];
        have2 = sl[6] > 0.f;
        j += have2 ? 2 : 1;
        rounds = r + 1;
    }
    // the word after the status word: exchange rounds this cloud took (measurement only: bench.py --mode pugan reports
    // samples per round and the time per round next to the exchange floor of pf_fps_exchange_probe)
    if (g == 0 && tid == 0) { fps_st(abort_w + 1, (unsigned long long)rounds); fps_st(abort_w, FPSC_ST_DONE); }
}

// ---- many samples per exchange (the shipped kernel; -DPF_FPS_TWO_SAMPLE / -DPF_FPS_ONE_SAMPLE build the predecessors) -------
// A round of fps_coop2_kernel knows more than it uses.  Every wave publishes its KW best points; wave 0 of every workgroup
// holds ALL published candidates (keys and, after one gather, coordinates).  Let B = the largest of the waves' LAST published
// keys: every point nobody published lies below its own wave's last key, hence below B - and adding samples only lowers
// min-distances, i.e. keys.  So wave 0 can run the sequential algorithm on the published candidates alone: take the largest
// key c1 (always the true next sample); lower the candidates' keys by c1 exactly as the update will (min(md, sqd(cand, c1)),
// the same unfused arithmetic, same argument order); the largest updated key K' is the true next sample iff K' >= B (nothing
// unpublished can reach it) and its distance is > 0 (a zero distance would tie with the samples already taken, whose keys are
// (0, index): the sequential algorithm then picks by index among points this round does not know).  Repeat until the test
// fails or MS samples are out.  Unlike the two-sample rule a TOUCHED candidate may still be taken (its lowered key is exact),
// and so may the third, fourth, ... .  Every workgroup runs the same chain on the same words and reaches the same samples.
// CPU simulation of the CLI's merge shape (99 840 -> 20 024, 100 waves): KW = 2, MS = 8: 7.3 samples per round (cap reached
// in 76 % of the rounds); KW = 2, MS = 16: 11.3; KW = 4, MS = 16: 15.5 - against 1.6-1.7 of fps_coop2_kernel.
#ifndef PF_FPS_KW
#define PF_FPS_KW 4                                      // words (best points) a wave publishes per round: 2 or 4
#endif
#ifndef PF_FPS_MS
#define PF_FPS_MS 64                                     // samples a round may emit
#endif
constexpr int FPSM_KW = PF_FPS_KW, FPSM_MS = PF_FPS_MS;
#ifndef PF_FPS_NC
#define PF_FPS_NC 2                                      // candidates per lane the chain of wave 0 works on
#endif
static_assert(FPSM_KW >= 2 && 4 * FPSM_KW * FPSC_SLOTS + 2 <= 8192 / 2, "the ring must fit the scratch row of the smallest cooperative cloud (8192 points = 4096 words)");
constexpr int FPSM_NC = PF_FPS_NC;
constexpr int FPSC_SLOTSM = FPSM_KW * FPSC_SLOTS;
constexpr int FPSC_RINGM = 4 * FPSC_SLOTSM;              // status word of this kernel's ring
constexpr int FPSM_NT = FPSM_KW * 2;                     // words a lane of wave 0 polls (<= 128 waves per cloud)

template <int V> struct FpsInt { static constexpr int value = V; };
typedef float fps_f2 __attribute__((ext_vector_type(2)));
// two squared distances at once: the same unfused fp32 operations as sqd(), issued as packed v_pk_add / v_pk_mul
__device__ __forceinline__ fps_f2 sqd2(fps_f2 ax, fps_f2 ay, fps_f2 az, float bx, float by, float bz) {
#pragma clang fp contract(off)
    const fps_f2 dx = ax - bx, dy = ay - by, dz = az - bz;
    return (dx * dx + dy * dy) + dz * dz;
}

template <int PPT>
__global__ __launch_bounds__(FPSC_T, (PPT <= 16 ? 4 : (PPT <= 24 ? 3 : 2))) void fps_coopm_kernel(const float* __restrict__ xyz, int N,
                                                                                 int npoint, int G,
                                                                                 unsigned long long* __restrict__ ringbuf,
                                                                                 long long ring_stride, int* __restrict__ out) {
    constexpr int NW = FPSC_T / 64, KW = FPSM_KW, MS = FPSM_MS, NT = FPSM_NT, PH = (PPT + 1) / 2, NC = FPSM_NC, CAP = 64 * NC;
    __shared__ unsigned long long s_ck[CAP];                            // wave 0: the round's compacted candidates
    __shared__ float s_l[2][3 * MS + 4];                                // samples of a round, [3 MS] = count, [3 MS + 1] = alive
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x / G, g = blockIdx.x % G;
    const float* p = xyz + (size_t)b * N * 3;
    unsigned long long* ring = ringbuf + (size_t)b * ring_stride;       // [4][FPSC_SLOTSM] + status word
    unsigned long long* abort_w = ring + FPSC_RINGM;
    int* o = out + (size_t)b * npoint;
    const int SW = KW * G * NW;                                         // words per round

    // a thread holds PPT CONSECUTIVE points, a wave 64 PPT consecutive points: indices grow with k (first maximum = smallest
    // index), and when the input has any spatial order (the CLI's merge: 1024 consecutive candidates = one patch) a wave's
    // points share a small bounding box - see the culling test below
    const int base = ((g * NW + wave) * 64 + lane) * PPT;
    fps_f2 px[PH], py[PH], pz[PH], md[PH];                              // points 2h, 2h + 1 of this thread
    float lox = 3e38f, loy = 3e38f, loz = 3e38f, hix = -3e38f, hiy = -3e38f, hiz = -3e38f;
#pragma unroll
    for (int k = 0; k < 2 * PH; ++k) {
        const int i = base + k;
        const bool in = k < PPT && i < N;
        const int ic = in ? i : N - 1;
        const float x = p[ic * 3 + 0], y = p[ic * 3 + 1], z = p[ic * 3 + 2];
        px[k >> 1][k & 1] = x; py[k >> 1][k & 1] = y; pz[k >> 1][k & 1] = z;
        md[k >> 1][k & 1] = in ? 1e10f : -1.f;                          // padding can never be the farthest point
        if (in) { lox = fminf(lox, x); loy = fminf(loy, y); loz = fminf(loz, z); hix = fmaxf(hix, x); hiy = fmaxf(hiy, y); hiz = fmaxf(hiz, z); }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {                            // the wave's bounding box (once)
        lox = fminf(lox, __shfl_xor(lox, off)); loy = fminf(loy, __shfl_xor(loy, off)); loz = fminf(loz, __shfl_xor(loz, off));
        hix = fmaxf(hix, __shfl_xor(hix, off)); hiy = fmaxf(hiy, __shfl_xor(hiy, off)); hiz = fmaxf(hiz, __shfl_xor(hiz, off));
    }
    if (tid == 0) {
        if (g == 0) o[0] = 0;
        s_l[1][0] = p[0]; s_l[1][1] = p[1]; s_l[1][2] = p[2];           // "round -1" emitted point 0
        s_l[1][3 * MS] = 1.f; s_l[1][3 * MS + 1] = 1.f;
    }
    __syncthreads();
    float Dw = 1e10f;                                                   // >= every min-distance of this wave (its last published best)
    int pv[KW];                                                         // the wave's last published keys (distance bits, index)
    unsigned pidx[KW];
#pragma unroll
    for (int e = 0; e < KW; ++e) { pv[e] = -1; pidx[e] = 0u; }
    int j = 1, rounds = 0;
    for (int r = 0; j < npoint; ++r) {
        // update with the samples of the previous round - those that can reach this wave.  Lane s looks at sample s: the
        // nearest point q of the wave's box to it is at least as close as every point of the wave, in fp32 as computed by
        // sqd() too (subtraction, multiplication and addition round monotonically), so sqd(q, c) >= Dw >= md means
        // fminf(md, sqd(p, c)) = md for the whole wave: the sample is skipped EXACTLY.  Late in a merge a sample reaches a
        // few per cent of the waves; a wave no sample reached publishes its previous keys again.
        bool touched = false;
        {
            const float* sp = s_l[(r + 1) & 1];
            const int m = (int)sp[3 * MS];
            const int ls = lane < MS ? lane : MS - 1;
            const float cx = sp[3 * ls], cy = sp[3 * ls + 1], cz = sp[3 * ls + 2];
            const float bd = sqd(fminf(fmaxf(cx, lox), hix), fminf(fmaxf(cy, loy), hiy), fminf(fmaxf(cz, loz), hiz), cx, cy, cz);
            unsigned long long need = __ballot(lane < m && bd < Dw);
            touched = need != 0ull;
            while (need) {
                const int s = __builtin_ctzll(need);
                need &= need - 1ull;
                auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
                const float lx = rl(cx, s), ly = rl(cy, s), lz = rl(cz, s);
#pragma unroll
                for (int h = 0; h < PH; ++h) {
                    const fps_f2 d = sqd2(px[h], py[h], pz[h], lx, ly, lz);
                    md[h][0] = fminf(md[h][0], d[0]); md[h][1] = fminf(md[h][1], d[1]);
                }
            }
        }
        unsigned long long* slot = ring + (r & 3) * FPSC_SLOTSM;
        const unsigned tag = ((unsigned)(((r >> 2) & 3) << 1) | 1u) << 29;      // see fps_coop_kernel
        if (touched) {                                                  // uniform over the wave
            // per-lane best KW by (distance, then smaller index): indices grow with k, strict comparisons keep the smaller
            float bv[KW];
            int bk[KW];
#pragma unroll
            for (int q = 0; q < KW; ++q) { bv[q] = -1.f; bk[q] = 0; }
#pragma unroll
            for (int k = 0; k < 2 * PH; ++k) {
                const float d = md[k >> 1][k & 1];
                bool gt[KW];
#pragma unroll
                for (int q = 0; q < KW; ++q) gt[q] = d > bv[q];
#pragma unroll
                for (int q = KW - 1; q >= 0; --q) {
                    const bool up = q > 0 && gt[q > 0 ? q - 1 : 0];     // the element above moves down into q
                    bv[q] = gt[q] ? (up ? bv[q > 0 ? q - 1 : 0] : d) : bv[q];
                    bk[q] = gt[q] ? (up ? bk[q > 0 ? q - 1 : 0] : k) : bk[q];
                }
            }
            // the wave's best KW (distances are >= 0 or the -1 of padding: their bit patterns order like signed integers)
#pragma unroll
            for (int e = 0; e < KW; ++e) {
                const int vb = __float_as_int(bv[0]);
                const unsigned ib = (unsigned)(base + bk[0]);
                const int vmax = wave_max_i32(vb);
                const unsigned imin = ~wave_max_u32(vb == vmax ? ~ib : 0u);
                const bool mine = vb == vmax && ib == imin;             // the lane that holds this one pops it
                pv[e] = vmax; pidx[e] = imin;
#pragma unroll
                for (int q = 0; q + 1 < KW; ++q) { bv[q] = mine ? bv[q + 1] : bv[q]; bk[q] = mine ? bk[q + 1] : bk[q]; }
                bv[KW - 1] = mine ? -1.f : bv[KW - 1];
            }
            Dw = __int_as_float(pv[0]);
        }
        if (lane == 0) {
#pragma unroll
            for (int e = 0; e < KW; ++e) fps_st(slot + KW * (g * NW + wave) + e, fps_key(pv[e], pidx[e], tag));
        }
        if (wave == 0) {
            unsigned long long kk[NT];
            unsigned spins = 0;
            bool dead = false;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    kk[t] = lane + 64 * t < SW ? fps_ld(slot + lane + 64 * t) : (unsigned long long)tag;
                    ok = ok && ((unsigned)kk[t] & (7u << 29)) == tag;
                }
                if (!__any(!ok)) break;
                if (++spins > FPSC_SPIN_MAX || (spins % 1024 == 0 && fps_ld(abort_w) == FPSC_ST_ABORT)) { dead = true; break; }
            }
            float* sl = s_l[r & 1];
            if (dead) {                                                     // uniform over the wave
                if (lane == 0) { fps_st(abort_w, FPSC_ST_ABORT); sl[3 * MS + 1] = -1.f; }
            } else {
                // B: the largest LAST key of a wave (word w = KW * wave + e is a last key iff w mod KW == KW - 1)
                unsigned long long lastk = 0ull, f0 = 0ull;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f0 = u64max(f0, kk[t]);
                    if ((lane + 64 * t) % KW == KW - 1) lastk = u64max(lastk, kk[t]);
                }
                const unsigned long long Bk = wave_max_u64(lastk);
                const unsigned long long K1 = wave_max_u64(f0);            // the round's first sample: the largest key of all
                // Only candidates with a key >= B (and a distance > 0) can follow it in this round - keys only fall: compact
                // them, NC per lane, through LDS (this wave only; the LDS operations of a wave execute in order) and chain on
                // those alone.  On the CLI's merge 12 (KW = 2), 34 (3) or 65 (4) candidates qualify on average, 144 at most; a
                // round with more than 64 NC emits its first sample only (still exact; KW = 4: 0.7 % of the rounds).
                unsigned long long rm[NT];
                int cnt = 0;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    rm[t] = __ballot(kk[t] == K1 || (kk[t] >= Bk && (unsigned)(kk[t] >> 32) != 0u));
                    cnt += __builtin_popcountll(rm[t]);
                }
                if (cnt > CAP) {                                            // uniform
                    cnt = 1;
#pragma unroll
                    for (int t = 0; t < NT; ++t) rm[t] = __ballot(kk[t] == K1);
                }
                {
                    int before = 0;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int pos = before + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(rm[t] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)rm[t], 0u));
                        if ((rm[t] >> lane) & 1ull) s_ck[pos] = kk[t];
                        before += __builtin_popcountll(rm[t]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // the chain on NCC candidates per lane: one when the round's candidates fit the wave (the usual case on the
                // pipeline's clouds: ~20 qualify), NC otherwise
                int m = 0;
                int kx = 0, ky = 0, kz = 0, ki = 0;
                auto chain = [&](auto ncc_) {
                    constexpr int NCC = decltype(ncc_)::value;
                    unsigned long long ck[NCC];
                    float cx[NCC], cy[NCC], cz[NCC];
#pragma unroll
                    for (int c = 0; c < NCC; ++c) {
                        ck[c] = lane + 64 * c < cnt ? s_ck[lane + 64 * c] : 0ull;
                        const unsigned q = (~(unsigned)ck[c]) & 0x1fffffffu;
                        const unsigned qc = q < (unsigned)N ? q : 0u;           // the cloud is read-only: plain cached loads
                        cx[c] = p[qc * 3 + 0]; cy[c] = p[qc * 3 + 1]; cz[c] = p[qc * 3 + 2];
                    }
                    __builtin_amdgcn_wave_barrier();                            // s_ck is written again next round, after these reads
                    for (;;) {
                        // this lane's best, then the wave's
                        unsigned long long f = ck[0];
                        float fx = cx[0], fy = cy[0], fz = cz[0];
#pragma unroll
                        for (int c = 1; c < NCC; ++c) {
                            const bool gt = ck[c] > f;
                            f = gt ? ck[c] : f;
                            fx = gt ? cx[c] : fx; fy = gt ? cy[c] : fy; fz = gt ? cz[c] : fz;
                        }
                        const unsigned fh = (unsigned)(f >> 32), fl = (unsigned)f;
                        const unsigned hmax = wave_max_u32(fh);
                        unsigned long long own = __ballot(fh == hmax);
                        unsigned lmax;
                        if (__builtin_popcountll(own) == 1) {                  // the usual case: one lane holds the largest distance
                            lmax = (unsigned)__builtin_amdgcn_readlane((int)fl, __builtin_ctzll(own));
                        } else {
                            lmax = wave_max_u32(fh == hmax ? fl : 0u);
                            own = __ballot(fh == hmax && fl == lmax);
                        }
                        const unsigned long long K = ((unsigned long long)hmax << 32) | lmax;
                        if (m > 0 && !(K >= Bk && __uint_as_float(hmax) > 0.f)) break;
                        const int w = __builtin_ctzll(own);
                        auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
                        const float sx = rl(fx, w), sy = rl(fy, w), sz = rl(fz, w);
                        const unsigned idx = (~lmax) & 0x1fffffffu;
                        // lane m keeps sample m (a select on the wave-uniform values): ONE LDS write and one index store per
                        // lane after the chain instead of an exec-masked write + store per sample
                        {
                            const bool me = lane == m;
                            kx = me ? __float_as_int(sx) : kx; ky = me ? __float_as_int(sy) : ky;
                            kz = me ? __float_as_int(sz) : kz; ki = me ? (int)idx : ki;
                        }
                        ++m;
                        if (m == MS || j + m >= npoint || idx >= (unsigned)N) break;
                        // lower the candidates' keys exactly as the update will (the sample itself falls to distance 0)
#pragma unroll
                        for (int c = 0; c < NCC; ++c) {
                            const float d = sqd(cx[c], cy[c], cz[c], sx, sy, sz);
                            const unsigned kh = (unsigned)(ck[c] >> 32);
                            if (d < __uint_as_float(kh)) ck[c] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)ck[c];
                        }
                    }
                };
                if (cnt <= 64) chain(FpsInt<1>{}); else chain(FpsInt<NC>{});
                if (lane < m) {
                    sl[3 * lane] = __int_as_float(kx); sl[3 * lane + 1] = __int_as_float(ky); sl[3 * lane + 2] = __int_as_float(kz);
                    if (g == 0) o[j + lane] = ki;
                }
                if (lane == 0) { sl[3 * MS] = (float)m; sl[3 * MS + 1] = 1.f; }
            }
        }
        // s_l is double-buffered (see fps_coop_kernel): one barrier per round
        __syncthreads();
        const float* sl = s_l[r & 1];
        if (sl[3 * MS + 1] < 0.f) return;                                 // uniform over the workgroup: aborted
        j += (int)sl[3 * MS];
        rounds = r + 1;
    }
    // the word after the status word: exchange rounds this cloud took (measurement only: bench.py --mode pugan reports
    // samples per round and the time per round next to the exchange floor of pf_fps_exchange_probe)
    if (g == 0 && tid == 0) { fps_st(abort_w + 1, (unsigned long long)rounds); fps_st(abort_w, FPSC_ST_DONE); }
}

// ---- the exchange alone: what a round of fps_coop2_kernel costs with NO points to update ------------------------------
// Same protocol, same ring, same shapes (G workgroups of FPSC_T threads, two words per wave, wave 0 polls 4 words per lane,
// reduces them and hands a result to the other waves through the double-buffered LDS words, one barrier per round) - only the
// per-point min-distance update and the candidate's coordinate fetch are missing.  Timed with HIP events it is the floor of a
// round: the store -> load round trip through the fabric plus waiting for the slowest of the G peers.
__global__ __launch_bounds__(FPSC_T) void fps_exchange_probe_kernel(int G, int rounds, unsigned long long* __restrict__ ring) {
    constexpr int NW = FPSC_T / 64;
    __shared__ float s_l[2][2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x;
    unsigned long long* abort_w = ring + FPSC_RING2;
    const int S2 = 2 * G * NW;
    unsigned acc = (unsigned)(g * NW + wave) + 1u;
    for (int r = 0; r < rounds; ++r) {
        unsigned long long* slot = ring + (r & 3) * FPSC_SLOTS2;
        const unsigned tag = ((unsigned)(((r >> 2) & 3) << 1) | 1u) << 29;
        if (lane == 0) {
            fps_st(slot + 2 * (g * NW + wave), ((unsigned long long)acc << 32) | tag | 1u);
            fps_st(slot + 2 * (g * NW + wave) + 1, ((unsigned long long)(acc >> 1) << 32) | tag);
        }
        if (wave == 0) {
            unsigned long long k[4];
            unsigned spins = 0;
            bool dead = false;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    k[t] = lane + 64 * t < S2 ? fps_ld(slot + lane + 64 * t) : (unsigned long long)tag;
                    ok = ok && ((unsigned)k[t] & (7u << 29)) == tag;
                }
                if (!__any(!ok)) break;
                if (++spins > FPSC_SPIN_MAX || (spins % 1024 == 0 && fps_ld(abort_w) == FPSC_ST_ABORT)) { dead = true; break; }
            }
            if (dead) {
                if (lane == 0) { fps_st(abort_w, FPSC_ST_ABORT); s_l[r & 1][1] = -1.f; }
            } else {
                const unsigned long long K1 = wave_max_u64(u64max(u64max(k[0], k[1]), u64max(k[2], k[3])));
                if (lane == 0) { s_l[r & 1][0] = __uint_as_float((unsigned)(K1 >> 32) & 0xffffu); s_l[r & 1][1] = 1.f; }
            }
        }
        __syncthreads();
        if (s_l[r & 1][1] < 0.f) return;
        acc = acc * 1664525u + (unsigned)__float_as_uint(s_l[r & 1][0]) + 1013904223u;      // the next word depends on this round's result
        acc &= 0x7fffffffu;
    }
    if (g == 0 && tid == 0) fps_st(abort_w, FPSC_ST_DONE);
}

// ---- large-K kNN: one workgroup per query; keys (dist bits << 32 | index) bitonic-sorted in LDS.
// N <= 16384: all N keys in one sort.  Larger clouds (knn_cuda.KNN takes any N, patch.py:33,107): the references are
// streamed in chunks; LDS holds the running best KP >= K keys in front of the chunk's keys, every pass sorts the
// NP = 16384 keys and keeps the front.  Keys are distinct (the index is part of the key), so the result is the exact
// (distance, index)-ordered top K whatever the chunking.
constexpr int KS_T = 1024;
constexpr int KS_NMAX = 16384;          // 128 KiB of 64-bit keys
constexpr int KS_KMAX = 8192;           // chunked path: at least half of the LDS keys are fresh references per pass

__global__ __launch_bounds__(KS_T) void knn_sort_kernel(const float* __restrict__ ref, const float* __restrict__ query,
                                                       int N, int M, int K, int NP /*pow2: keys sorted per pass*/,
                                                       int KP /*0: single pass; else pow2 >= K kept between passes*/,
                                                       int* __restrict__ idx_out, float* __restrict__ dist_out) {
    extern __shared__ unsigned long long keys[];
    const int b = blockIdx.y, q = blockIdx.x, tid = threadIdx.x;
    const float* r = ref + (size_t)b * N * 3;
    const float* qq = query + ((size_t)b * M + q) * 3;
    const float qx = qq[0], qy = qq[1], qz = qq[2];
    const int C = NP - KP;                                              // fresh references per pass
    for (int i = tid; i < KP; i += KS_T) keys[i] = ~0ull;
    for (int c0 = 0; c0 < N; c0 += C) {
        for (int i = tid; i < C; i += KS_T) {
            const int j = c0 + i;
            unsigned long long key = ~0ull;
            if (j < N) key = ((unsigned long long)__float_as_uint(sqd(qx, qy, qz, r[j * 3 + 0], r[j * 3 + 1], r[j * 3 + 2])) << 32) | (unsigned)j;
            keys[KP + i] = key;
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
    }
    for (int i = tid; i < K; i += KS_T) {
        const unsigned long long key = keys[i];
        idx_out[((size_t)b * M + q) * K + i] = (int)(key & 0xffffffffu);
        if (dist_out) dist_out[((size_t)b * M + q) * K + i] = __uint_as_float((unsigned)(key >> 32));
    }
}

// ---- normalize_pc (modules/utils/patch.py:168-178): centroid = mean over the points, pc - centroid, divided by the largest
// norm.  One workgroup per cloud / patch with a FIXED summation order (thread t adds points t, t + 256, ... in order, then a
// binary tree over the 256 partial sums), so a cloud's result does not depend on how many clouds share the launch - torch's
// mean picks its reduction tree from the whole tensor shape.  All arithmetic unfused fp32, IEEE sqrt and division.
constexpr int NRM_T = 256;
__global__ __launch_bounds__(NRM_T) void normalize_pc_kernel(const float* __restrict__ x, int N, float* __restrict__ out,
                                                             float* __restrict__ centroid, float* __restrict__ fdist) {
#pragma clang fp contract(off)
    __shared__ float red[3][NRM_T];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* p = x + (size_t)b * N * 3;
    float* o = out + (size_t)b * N * 3;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int i = tid; i < N; i += NRM_T) {
        sx = __fadd_rn(sx, p[i * 3 + 0]); sy = __fadd_rn(sy, p[i * 3 + 1]); sz = __fadd_rn(sz, p[i * 3 + 2]);
    }
    red[0][tid] = sx; red[1][tid] = sy; red[2][tid] = sz;
    __syncthreads();
    for (int s = NRM_T / 2; s > 0; s >>= 1) {
        if (tid < s) {
            red[0][tid] = __fadd_rn(red[0][tid], red[0][tid + s]);
            red[1][tid] = __fadd_rn(red[1][tid], red[1][tid + s]);
            red[2][tid] = __fadd_rn(red[2][tid], red[2][tid + s]);
        }
        __syncthreads();
    }
    const float cx = __fdiv_rn(red[0][0], (float)N), cy = __fdiv_rn(red[1][0], (float)N), cz = __fdiv_rn(red[2][0], (float)N);
    __syncthreads();
    float m = 0.f;
    for (int i = tid; i < N; i += NRM_T) {
        const float dx = __fsub_rn(p[i * 3 + 0], cx), dy = __fsub_rn(p[i * 3 + 1], cy), dz = __fsub_rn(p[i * 3 + 2], cz);
        m = fmaxf(m, __fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz))));
    }
    red[0][tid] = m;
    __syncthreads();
    for (int s = NRM_T / 2; s > 0; s >>= 1) {
        if (tid < s) red[0][tid] = fmaxf(red[0][tid], red[0][tid + s]);
        __syncthreads();
    }
    const float fd = red[0][0];
    for (int i = tid; i < N; i += NRM_T) {
        o[i * 3 + 0] = __fdiv_rn(__fsub_rn(p[i * 3 + 0], cx), fd);
        o[i * 3 + 1] = __fdiv_rn(__fsub_rn(p[i * 3 + 1], cy), fd);
        o[i * 3 + 2] = __fdiv_rn(__fsub_rn(p[i * 3 + 2], cz), fd);
    }
    if (tid == 0) {
        centroid[b * 3 + 0] = cx; centroid[b * 3 + 1] = cy; centroid[b * 3 + 2] = cz;
        fdist[b] = fd;
    }
}

}  // namespace

namespace {
// clears the scratch rows and sets every cloud's status word to "not finished"
__global__ __launch_bounds__(256) void fps_init_kernel(float* p, long long n, int B, long long stride_words, int status_word) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float v = 0.f;
        const long long w = i >> 1;                                       // 64-bit word index
        if ((i & 1) == 0 && stride_words > 0 && w % stride_words == status_word && w / stride_words < B)
            v = __uint_as_float((unsigned)FPSC_ST_INIT);                  // low half of the status word
        p[i] = v;
    }
}
}  // namespace

// 1 when pf_fps runs the cooperative kernel for clouds of N points (its scratch row then holds the candidate ring):
// stride_words = 64-bit words between the rings of consecutive clouds, abort_word = index of the abort word in a ring
// two samples per exchange (fps_coop2_kernel, the shipped kernel); -DPF_FPS_ONE_SAMPLE builds the one-sample kernel instead
// (tools/time_fps.py A/B) - a compile-time choice: pf_fps and pf_fps_scratch_layout must agree
#if defined(PF_FPS_ONE_SAMPLE)
static constexpr int fps_mode() { return 1; }
#elif defined(PF_FPS_TWO_SAMPLE)
static constexpr int fps_mode() { return 2; }
#else
static constexpr int fps_mode() { return 3; }                             // fps_coopm_kernel
#endif
static constexpr int fps_status_word() { return fps_mode() == 1 ? FPSC_RING : fps_mode() == 2 ? FPSC_RING2 : FPSC_RINGM; }

extern "C" int pf_fps_scratch_layout(int N, long long* stride_words, long long* abort_word) {
    if (stride_words) *stride_words = ((long long)N / 2) & ~1ll;
    if (abort_word) *abort_word = fps_status_word();
    return N >= 8192 && N <= FPSC_GMAX * 1024 * 8 ? 1 : 0;
}

// xyz [B,N,3] -> idx [B,npoint] int32; mind: [B,N] float scratch
// points per thread of the cooperative kernel.  `group` > 0: the caller says that every `group` consecutive points are one
// spatial neighbourhood (the CLI's merge: 1280 candidates per patch); when a wave can hold exactly one group (64 x 12 / 20 / 24
// points) its bounding box is that neighbourhood's, which is what the culling of fps_coopm_kernel lives on.
static int fps_ppt(int N, int group) {
    if (fps_mode() == 3 && group > 0 && group % 64 == 0) {
        const int p = group / 64;
        if ((p == 12 || p == 20 || p == 24) && (N + FPSC_T * p - 1) / (FPSC_T * p) <= FPSC_GMAX && N >= FPSC_T * p * 2) return p;
    }
    int ppt = N >= 16 * FPSC_T ? 4 : 1;                                               // fewer, fuller workgroups
    while ((N + FPSC_T * ppt - 1) / (FPSC_T * ppt) > FPSC_GMAX) ppt *= 2;              // -> 1, 4, 8 (16, 32) points per thread
    return ppt;
}

extern "C" int pf_fps(const float* xyz, int B, int N, int npoint, float* mind, int* idx_out, void* stream) {
    return pf_fps_grouped(xyz, B, N, npoint, 0, mind, idx_out, stream);
}

extern "C" int pf_fps_grouped(const float* xyz, int B, int N, int npoint, int group, float* mind, int* idx_out, void* stream) {
    if (!xyz || !mind || !idx_out) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || npoint <= 0 || npoint > N || group < 0) return PF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    // cooperative kernel when the cloud is worth more than one CU; the candidate ring (4 x 32 + 1 64-bit words)
    // lives at the start of each cloud's N-float scratch row
    if (N >= 8192 && N <= FPSC_GMAX * 1024 * 8 && ((size_t)mind & 7) == 0) {
        const int ppt = fps_ppt(N, group);
        const int G = (N + FPSC_T * ppt - 1) / (FPSC_T * ppt);
        const long long stride = ((long long)N / 2) & ~1ll;                           // 64-bit words per cloud
        unsigned long long* ring = reinterpret_cast<unsigned long long*>(mind);
        // cleared by a kernel, not hipMemsetAsync: a memset node inside a captured hipGraph was observed to race with the
        // kernel node that follows it (csrc/emd.hip)
        hipLaunchKernelGGL(fps_init_kernel, dim3(256), dim3(256), 0, s, mind, (long long)B * N, B, stride,
                           fps_status_word());
        const dim3 grid(B * G), block(FPSC_T);
#define PF_FPS_LAUNCH(PPT)                                                                                                  \
        if (fps_mode() == 3) hipLaunchKernelGGL(fps_coopm_kernel<PPT>, grid, block, 0, s, xyz, N, npoint, G, ring, stride, idx_out); \
        else if (fps_mode() == 2) hipLaunchKernelGGL(fps_coop2_kernel<PPT>, grid, block, 0, s, xyz, N, npoint, G, ring, stride, idx_out); \
        else hipLaunchKernelGGL(fps_coop_kernel<PPT>, grid, block, 0, s, xyz, N, npoint, G, ring, stride, idx_out)
#define PF_FPS_LAUNCHM(PPT) hipLaunchKernelGGL(fps_coopm_kernel<PPT>, grid, block, 0, s, xyz, N, npoint, G, ring, stride, idx_out)
        switch (ppt) {
            case 1: PF_FPS_LAUNCH(1); break;
            case 4: PF_FPS_LAUNCH(4); break;
            case 8: PF_FPS_LAUNCH(8); break;
            case 12: PF_FPS_LAUNCHM(12); break;                                        // (group-aligned shapes: fps_mode() == 3 only)
            case 16: PF_FPS_LAUNCH(16); break;
            case 20: PF_FPS_LAUNCHM(20); break;
            case 24: PF_FPS_LAUNCHM(24); break;
            default: PF_FPS_LAUNCH(32); break;
        }
#undef PF_FPS_LAUNCHM
#undef PF_FPS_LAUNCH
        return pf_last_launch_status();
    }
    hipLaunchKernelGGL(fps_kernel, dim3(B), dim3(FPS_T), 0, s, xyz, N, npoint, mind, idx_out);
    return pf_last_launch_status();
}

// `rounds` exchange rounds of the cooperative FPS protocol between G workgroups (2 <= G <= 32) with no points to update: the
// floor of a round of fps_coop2_kernel.  ring: >= 1032 64-bit words of scratch (cleared here); its word 1024 is 0 afterwards
// when every round completed (1 = the bounded spin gave up).  Measurement aid of bench.py --mode pugan.
extern "C" int pf_fps_exchange_probe(int G, int rounds, unsigned long long* ring, void* stream) {
    if (!ring) return PF_ERR_NULL;
    if (G < 2 || G > FPSC_GMAX || rounds <= 0 || ((size_t)ring & 7) != 0) return PF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(fps_init_kernel, dim3(8), dim3(256), 0, s, reinterpret_cast<float*>(ring), 2ll * (FPSC_RING2 + 8), 1,
                       (long long)(FPSC_RING2 + 8), FPSC_RING2);
    hipLaunchKernelGGL(fps_exchange_probe_kernel, dim3(G), dim3(FPSC_T), 0, s, G, rounds, ring);
    return pf_last_launch_status();
}

// K nearest references of every query, K <= N (K <= 8192 when N > 16384): idx [B,M,K] int32, dist [B,M,K] squared L2 (nullable)
extern "C" int pf_knn_large(const float* ref, const float* query, int B, int N, int M, int K, int* idx_out,
                            float* dist_out, void* stream) {
    if (!ref || !query || !idx_out) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || M <= 0 || K <= 0 || K > N || B > 65535 || (long long)N * 3 > 0x7fffffffll) return PF_ERR_SHAPE;
    int np = 1, kp = 0;
    if (N <= KS_NMAX) {
        while (np < N) np <<= 1;
    } else {
        if (K > KS_KMAX) return PF_ERR_UNSUPPORTED;
        np = KS_NMAX;
        kp = 1;
        while (kp < K) kp <<= 1;
    }
    const size_t lds = (size_t)np * 8;
    pf_allow_lds(reinterpret_cast<const void*>(knn_sort_kernel), lds);
    hipLaunchKernelGGL(knn_sort_kernel, dim3(M, B), dim3(KS_T), lds, (hipStream_t)stream, ref, query, N, M, K, np, kp,
                       idx_out, dist_out);
    return pf_last_launch_status();
}

// normalize_pc (patch.py:168-178) of B clouds / patches: x [B,N,3] -> out [B,N,3] (may alias x), centroid [B,3], fdist [B]
extern "C" int pf_normalize_pc(const float* x, int B, int N, float* out, float* centroid, float* fdist, void* stream) {
    if (!x || !out || !centroid || !fdist) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || (long long)N * 3 > 0x7fffffffll) return PF_ERR_SHAPE;
    hipLaunchKernelGGL(normalize_pc_kernel, dim3(B), dim3(NRM_T), 0, (hipStream_t)stream, x, N, out, centroid, fdist);
    return pf_last_launch_status();
}
