// Brute-force kNN (replaces pytorch3d.ops.knn_points, reference call sites
// modules/discrete/interpflow.py:104,:328) and K=1 nearest neighbour for Chamfer
// (pytorch3d.loss.chamfer_distance, metric/loss.py:42; arithmetic as the plain-C `nnsearch`
// in evaluation/tf_ops/nn_distance/tf_nndistance.cpp:21-43).
//
// Semantics (defined by the build): squared L2 in UNFUSED fp32
// ((dx*dx)+(dy*dy))+(dz*dz), result ordered by (distance asc, index asc), self included.
// Bit-exact against oracle/ref_cpu.py::knn_canonical.
//
// Four kernels, same results bit for bit:
//  * knn_kernel (simple): one lane per query, references read with wave-uniform scalar loads,
//    per-lane sorted top-K in registers, branch-free shifting insert.  Its cost is the insert:
//    a lane inserts only ~K ln(M/K) times, but SOME lane of the 64 inserts at almost every
//    candidate, so the wave executes the 5K-op insert ~M times.
//  * knn2_kernel (default for K <= 16, M >= 256): two sweeps over the references remove
//    that divergence (references are broadcast from registers with v_readlane: no LDS traffic for the
//    candidates).  Sweep A keeps 32 strided group minima per query (1 op per candidate); the
//    K-th smallest group minimum tau bounds the K-th neighbour distance (K distinct candidates
//    are <= tau).  Sweep B appends every candidate with d <= tau to a per-lane LDS list (~22
//    entries for K = 16: coupon-collector count of hitting K of 32 groups); only those are
//    inserted.  A lane whose list overflows (heavy ties) makes its wave redo the exact simple scan.
//  * knn4_kernel: knn2 with the references split over the 4 / 8 / 16 waves of a workgroup (small and medium grids).
//  * knn5_kernel (grids that fill the chip, M <= 4096): the two sweeps run on a filter the f32 MFMA computes for 16 x 16
//    (query, reference) pairs per instruction; survivors are ranked by EXACT distances (see its header).
// Candidates are visited in increasing index in every kernel, so a strict `<` keeps equal
// distances in index order.
#include <hip/hip_runtime.h>
#include "pf_api_internal.h"
#include <type_traits>

#ifndef PF_KNN5
#define PF_KNN5 1                        // 0: A/B builds without the matrix-pipe sweeps (knn5_kernel)
#endif
#ifndef PF_KNN5_NWMAX
#define PF_KNN5_NWMAX 16              // waves per workgroup at most (A/B at 32 x 2048: 16 -> 55.9 us, 8 -> 65.3, 4 -> 72.4)
#endif
#ifndef PF_KNN5_MIN_M
#define PF_KNN5_MIN_M 256
#endif
#ifndef PF_KNN5_MIN_WGS
#define PF_KNN5_MIN_WGS 64               // 64-query tiles from which knn5_kernel is used: 4 x 2048 and up measured faster (28 vs 31 us; 16 x 2048: 35 vs 52); below: knn4_kernel's 16 reference slices
#endif
typedef float f4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sqdist(float qx, float qy, float qz, float rx, float ry, float rz) {
#pragma clang fp contract(off)          // ((dx*dx) + (dy*dy)) + (dz*dz), never an fma: bit-exact with the reference order
    const float dx = __fsub_rn(qx, rx), dy = __fsub_rn(qy, ry), dz = __fsub_rn(qz, rz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// hipcc folds `c ? bi[i - 1] : bi[i]` on the INT list into one load through a selected address, which keeps the list in scratch
// (48 bytes per lane, ~60 scratch instructions per query with a run-time offset in knn5_kernel<16, .>: round 5, found in the ISA;
// the float list next to it stays in registers).  An empty asm makes the two values opaque: the selects stay selects.
__device__ __forceinline__ int knn_opaque(int v) { asm("" : "+v"(v)); return v; }

// shifting insert of (d, j) into the ascending lists bd/bi (see header comment for the tie rule)
template <int K>
__device__ __forceinline__ void topk_insert(float (&bd)[K], int (&bi)[K], float d, int j) {
#pragma unroll
    for (int i = K - 1; i >= 1; --i) {
        const bool ltl = d < bd[i - 1], lti = d < bd[i];
        bi[i] = ltl ? knn_opaque(bi[i - 1]) : (lti ? j : knn_opaque(bi[i]));
        bd[i] = ltl ? bd[i - 1] : (lti ? d : bd[i]);
    }
    if (d < bd[0]) { bd[0] = d; bi[0] = j; }
}

// the same when only the first T entries of the lists are finite (the T-th insertion into an empty list): slots 0 .. T
template <int K, int T>
__device__ __forceinline__ void topk_insert_head(float (&bd)[K], int (&bi)[K], float d, int j) {
    constexpr int TOP = T < K - 1 ? T : K - 1;
#pragma unroll
    for (int i = TOP; i >= 1; --i) {
        const bool ltl = d < bd[i - 1], lti = d < bd[i];
        bi[i] = ltl ? knn_opaque(bi[i - 1]) : (lti ? j : knn_opaque(bi[i]));
        bd[i] = ltl ? bd[i - 1] : (lti ? d : bd[i]);
    }
    if (d < bd[0]) { bd[0] = d; bi[0] = j; }
}

template <int I, int N, class F>
__device__ __forceinline__ void pf_static_for_knn(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        pf_static_for_knn<I + 1, N>(f);
    }
}

// exact scan over all M references with wave-uniform (scalar) loads
template <int K>
__device__ __forceinline__ void exact_scan(float qx, float qy, float qz, const float* __restrict__ r, int M,
                                           float (&bd)[K], int (&bi)[K]) {
#pragma unroll
    for (int i = 0; i < K; ++i) { bd[i] = __builtin_inff(); bi[i] = -1; }
    for (int j = 0; j < M; ++j) {
        const float d = sqdist(qx, qy, qz, r[j * 3 + 0], r[j * 3 + 1], r[j * 3 + 2]);
        if (d < bd[K - 1]) topk_insert<K>(bd, bi, d, j);      // wave-divergent guard
    }
}

template <int K>
__device__ __forceinline__ void store_topk(const float (&bd)[K], const int (&bi)[K], size_t row, int* __restrict__ idx_out,
                                           float* __restrict__ dist_out) {
    if (idx_out) {                                          // (pf_nn1 may ask for the distances alone)
        int* o = idx_out + row * K;
#pragma unroll
        for (int i = 0; i < K; ++i) o[i] = bi[i];
    }
    if (dist_out) {
        float* od = dist_out + row * K;
#pragma unroll
        for (int i = 0; i < K; ++i) od[i] = bd[i];
    }
}

template <int K>
__global__ __launch_bounds__(64) void knn_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                 int N, int M, int* __restrict__ idx_out,
                                                 float* __restrict__ dist_out) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 64 + threadIdx.x;
    const bool live = n < N;
    const float* q = p1 + ((size_t)b * N + (live ? n : N - 1)) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    float bd[K];
    int bi[K];
    exact_scan<K>(qx, qy, qz, p2 + (size_t)b * M * 3, M, bd, bi);
    if (live) store_topk<K>(bd, bi, (size_t)b * N + n, idx_out, dist_out);
}

// ascending bitonic sort of 16 registers (80 compare-exchanges, all indices compile-time)
__device__ __forceinline__ void sort16(float (&v)[16]) {
#pragma unroll
    for (int k = 2; k <= 16; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const float a = v[i], c = v[l];
                    const bool up = (i & k) == 0;
                    v[i] = up ? fminf(a, c) : fmaxf(a, c);
                    v[l] = up ? fmaxf(a, c) : fminf(a, c);
                }
            }
}

constexpr int KNN2_T = 256;       // queries (threads) per workgroup
constexpr int KNN2_CAP = 64;      // per-lane candidate list capacity (u16 indices)

__device__ __forceinline__ float bcast(float v, int srclane) {      // wave-uniform broadcast through an SGPR
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), srclane));
}

// References are streamed 64 at a time: lane l loads reference j0 + l (coalesced, no LDS), then every reference is
// broadcast to the whole wave with v_readlane (an SGPR operand of the distance arithmetic).  LDS holds only the
// per-lane candidate lists.
template <int K>
__global__ __launch_bounds__(KNN2_T) void knn2_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                      int N, int M, int* __restrict__ idx_out,
                                                      float* __restrict__ dist_out) {
    static_assert(K <= 16, "threshold selection uses two 16-element sorted halves");
    __shared__ unsigned short lst[KNN2_CAP][KNN2_T];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    const int n = blockIdx.x * KNN2_T + tid;
    const bool live = n < N;
    const float* q = p1 + ((size_t)b * N + (live ? n : N - 1)) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float* __restrict__ r = p2 + (size_t)b * M * 3;

    auto load_ref = [&](int j0, float& cx, float& cy, float& cz) {     // +inf padding past M never wins a minimum
        const int j = j0 + lane;
        const bool in = j < M;
        const int jc = in ? j : M - 1;
        cx = in ? r[jc * 3 + 0] : __builtin_inff();
        cy = r[jc * 3 + 1];
        cz = r[jc * 3 + 2];
    };

    // ---- sweep A: 32 strided group minima
    float gm[32];
#pragma unroll
    for (int g = 0; g < 32; ++g) gm[g] = __builtin_inff();
    {
        float cx, cy, cz;
        load_ref(0, cx, cy, cz);
        for (int j0 = 0; j0 < M; j0 += 64) {
            float nx = cx, ny = cy, nz = cz;
            if (j0 + 64 < M) load_ref(j0 + 64, nx, ny, nz);          // prefetch the next 64 references
#pragma unroll
            for (int c = 0; c < 64; ++c)
                gm[c & 31] = fminf(gm[c & 31], sqdist(qx, qy, qz, bcast(cx, c), bcast(cy, c), bcast(cz, c)));
            cx = nx; cy = ny; cz = nz;
        }
    }
    // tau = K-th smallest of the 32 group minima: K-th smallest of two sorted halves = max_i min(A[i], B[K-1-i])
    float tau;
    {
        float ha[16], hb[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) { ha[g] = gm[g]; hb[g] = gm[16 + g]; }
        sort16(ha);
        sort16(hb);
        tau = fminf(ha[0], hb[K - 1]);
#pragma unroll
        for (int i = 1; i < K; ++i) tau = fmaxf(tau, fminf(ha[i], hb[K - 1 - i]));
    }

    // ---- sweep B: collect every candidate with d <= tau (in index order)
    int cnt = 0;
    {
        float cx, cy, cz;
        load_ref(0, cx, cy, cz);
        for (int j0 = 0; j0 < M; j0 += 64) {
            float nx = cx, ny = cy, nz = cz;
            if (j0 + 64 < M) load_ref(j0 + 64, nx, ny, nz);
#pragma unroll
            for (int c = 0; c < 64; ++c) {
                const float d = sqdist(qx, qy, qz, bcast(cx, c), bcast(cy, c), bcast(cz, c));
                if (d <= tau) {                                       // padding has d = +inf > tau unless tau = +inf (M < 32K.. never)
                    if (cnt < KNN2_CAP) lst[cnt][tid] = (unsigned short)(j0 + c);
                    ++cnt;
                }
            }
            cx = nx; cy = ny; cz = nz;
        }
    }

    float bd[K];
    int bi[K];
    if (__any(cnt > KNN2_CAP)) {
        exact_scan<K>(qx, qy, qz, r, M, bd, bi);      // heavy ties: redo this wave exactly (rare)
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i) { bd[i] = __builtin_inff(); bi[i] = -1; }
        for (int s = 0; __any(s < cnt); ++s) {
            if (s < cnt) {
                const int j = lst[s][tid];
                const float d = sqdist(qx, qy, qz, r[j * 3 + 0], r[j * 3 + 1], r[j * 3 + 2]);
                topk_insert<K>(bd, bi, d, j);
            }
        }
    }
    if (live) store_topk<K>(bd, bi, (size_t)b * N + n, idx_out, dist_out);
}

// ---- knn4_kernel: the two-sweep algorithm with the references split over the 4 waves of a workgroup ----
// A workgroup owns 64 queries (lane = query in every wave); wave w scans reference quarter w.  Same 32 group
// minima (8 strided groups per quarter), same tau, same candidates - but 4x the waves of knn2_kernel for the same
// work: at 32 x 2048 that kernel puts ONE wave on each SIMD, so nothing hides its readlane / LDS / dependent-issue stalls.
// Candidate lists are per (wave, lane).  4 slices: every wave reduces its own list to a sorted top K, wave 0 merges the four
// sorted lists of each query; 8 / 16 slices: all waves merge the lists by rank in (distance, index) order (a query at a time, a
// candidate per lane).  Overflow of any list -> exact scan of that query tile by wave 0.
// Where the time goes at 32 x 2048, K = 16 (tools/tune_knn.py, ablation builds): sweep A + threshold 23 us = the VALU bound of
// its 6.5 operations per pair (29 us with the exact 8-operation distance in the sweeps: they now run on a fused FILTER
// distance with a proven margin, see fdist below); sweep B 41 us (the same arithmetic + the divergent appends: almost every
// group of four references is a candidate for SOME lane of the wave); selection 14-20 us (42 us when wave 0 inserted all four
// lists alone).  Tried and dropped: a 3-fma filter |r|^2 - 2 q.r from a per-reference (x, y, z, |r|^2) table, widened by an
// absolute rounding bound (built twice, bit-exact on all tests): 4.5 operations per pair on paper, but slower than the fused
// filter at every chunk size tried (sweep A alone 33-39 us with 8 / 16 / 32 references per chunk against 22.5 us).
// KNN4_W = waves per workgroup = reference slices: 4 when the grid fills the chip anyway (32 x 2048: 1024 workgroups), 8 or
// 16 for small batches (4 x 2048: 128 workgroups of 16 waves instead of 4 - the kernel's latency is one wave's two sweeps
// over its slice, so more, shorter slices cut it almost proportionally).  Same results for every split.
#ifndef PF_KNN4_U
#define PF_KNN4_U 32
#endif
constexpr int KNN4_U = PF_KNN4_U;     // references per unrolled chunk (a multiple of every KNN4_G)

template <int K, int KNN4_W>
__global__ __launch_bounds__(KNN4_W * 64) void knn4_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                           int N, int M, int* __restrict__ idx_out,
                                                           float* __restrict__ dist_out) {
    static_assert(K <= 16, "threshold selection uses two 16-element sorted halves");
    constexpr int KNN4_G = 32 / KNN4_W;                       // strided minimum groups per slice
    constexpr int KNN4_CAP = 32;                              // per-(wave, lane) candidate list capacity (spatially ordered inputs put most candidates in one slice)
    __shared__ float gms[32][64];
    __shared__ unsigned short lst[KNN4_W][KNN4_CAP][64];
    __shared__ int cnts[KNN4_W][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);        // wave-uniform (SGPR) by construction
    const int n = blockIdx.x * 64 + lane;
    const bool live = n < N;
    const float* q = p1 + ((size_t)b * N + (live ? n : N - 1)) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float* __restrict__ r = p2 + (size_t)b * M * 3;
    // FILTER distance to reference j (wave-uniform j: scalar loads, SGPR operands) for the two sweeps: the same rounded
    // differences as the exact distance, summed with two fmas - 6 operations instead of 8.  Both are sums of three non-negative
    // terms with at most 3 roundings each, so filter = exact x (1 +- 7u), u = 2^-24.  Sweep A's tau (K-th smallest group minimum
    // of the filter) therefore bounds the exact K-th neighbour distance by tau (1 + 8u), and every reference of the exact answer,
    // ties included, has a filter distance <= tau (1 + 16u): sweep B keeps filter <= tau (1 + 2^-19) + 1e-37 (twice that; the
    // absolute term covers subnormal distances).  The answer is computed from EXACT distances of the survivors.
    auto fdist = [&](int j) {
        const float* rr = r + (size_t)j * 3;
        const float dx = qx - rr[0], dy = qy - rr[1], dz = qz - rr[2];
        return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    };
    const int mq = ((M + KNN4_W * 64 - 1) / (KNN4_W * 64)) * 64;      // slice length, multiple of 64
    const int jb = wave * mq, je = min(jb + mq, M);                   // this wave's references [jb, je)
    const int jfull = jb + ((je - jb) > 0 ? ((je - jb) / KNN4_U) * KNN4_U : 0);   // end of the unguarded chunks
    // References are wave-uniform: r[j] below compiles to scalar loads (s_load_dwordx*), the coordinates are SGPR
    // operands of the distance arithmetic - no VALU slot, no LDS, no cross-lane traffic is spent on them.

    // ---- sweep A: KNN4_G strided group minima of this slice
    {
        float gm[KNN4_G];
#pragma unroll
        for (int g = 0; g < KNN4_G; ++g) gm[g] = __builtin_inff();
        for (int j0 = jb; j0 < jfull; j0 += KNN4_U) {
#pragma unroll
            for (int c = 0; c < KNN4_U; ++c) gm[c % KNN4_G] = fminf(gm[c % KNN4_G], fdist(j0 + c));
        }
        for (int j0 = jfull; j0 < je; j0 += KNN4_G) {                   // tail: uniform guards, same group order
#pragma unroll
            for (int g = 0; g < KNN4_G; ++g)
                if (j0 + g < je) gm[g] = fminf(gm[g], fdist(j0 + g));
        }
#pragma unroll
        for (int g = 0; g < KNN4_G; ++g) gms[wave * KNN4_G + g][lane] = gm[g];
    }
    __syncthreads();
    // tau = K-th smallest of the 32 group minima (every wave computes it for its lanes: same value in all four)
    float tau;
    {
        float ha[16], hb[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) { ha[g] = gms[g][lane]; hb[g] = gms[16 + g][lane]; }
        sort16(ha);
        sort16(hb);
        tau = fminf(ha[0], hb[K - 1]);
#pragma unroll
        for (int i = 1; i < K; ++i) tau = fmaxf(tau, fminf(ha[i], hb[K - 1 - i]));
        tau = fmaf(tau, 1.9073486328125e-6f, tau) + 1e-37f;            // filter -> exact margin (see fdist)
    }

#if defined(PF_KNN_ABL) && PF_KNN_ABL == 2          // timing-only build: sweep A + threshold only
    if (wave == 0 && live) idx_out[((size_t)b * N + n) * K] = __float_as_int(tau);
    return;
#endif
    // ---- sweep B: collect this slice's candidates with filter distance <= tau (in index order)
    int cnt = 0;
    auto take = [&](float d, int j) {
        if (d <= tau) {
            if (cnt < KNN4_CAP) lst[wave][cnt][lane] = (unsigned short)j;
            ++cnt;
        }
    };
    for (int j0 = jb; j0 < jfull; j0 += KNN4_U) {
        // all distances of the chunk first (straight-line code: the scalar loads of the references are issued together),
        // then the rarely taken appends - with the branch right behind every distance each reference paid its own
        // scalar-load latency
        float dd[KNN4_U];
#pragma unroll
        for (int c = 0; c < KNN4_U; ++c) dd[c] = fdist(j0 + c);
#pragma unroll
        for (int c = 0; c < KNN4_U; c += 4) {                               // ~1 % of the references pass: test four at a time
            if (fminf(fminf(dd[c], dd[c + 1]), fminf(dd[c + 2], dd[c + 3])) <= tau) {
                take(dd[c], j0 + c); take(dd[c + 1], j0 + c + 1); take(dd[c + 2], j0 + c + 2); take(dd[c + 3], j0 + c + 3);
            }
        }
    }
    for (int j = jfull; j < je; ++j) take(fdist(j), j);
    cnts[wave][lane] = cnt;
#if defined(PF_KNN_ABL) && PF_KNN_ABL == 1          // timing-only build (tools/tune_knn.py): no merge
    __syncthreads();
    if (wave == 0 && live) idx_out[((size_t)b * N + n) * K] = cnt;
    return;
#endif
    if constexpr (KNN4_W == 4) {
        // 4 slices (the grid fills the chip by itself).  Every wave first reduces ITS OWN list to a sorted top K (exact
        // distances, inserted in list order = increasing index; the next candidate's coordinates are fetched while the current
        // one is inserted), then wave 0 merges the four sorted lists of every query (lane) - 16 steps of "smallest head".
        // Before: wave 0 alone inserted all four lists, list after list, while three waves idled: 42 of the kernel's 112 us at
        // 32 x 2048 (tools/tune_knn.py); one pass over each lane's concatenated lists: 31 us.  (The rank merge of the
        // 8 / 16-slice path measured slower here: 16 queries per wave, one after the other.)
        static_assert(KNN4_CAP * 2 >= K * 4, "a wave's list region is reused for its K sorted distances");
        __shared__ unsigned short sidx[4][K][64];
        float (*sdst)[64] = reinterpret_cast<float (*)[64]>(&lst[wave][0][0]);      // own region: only this wave reads lst[wave]
        {
            float bd[K];
            int bi[K];
#pragma unroll
            for (int i = 0; i < K; ++i) { bd[i] = __builtin_inff(); bi[i] = 0xffff; }
            const int cme = cnt < KNN4_CAP ? cnt : KNN4_CAP;
            int j = cme > 0 ? (int)lst[wave][0][lane] : 0;
            float rx = r[j * 3 + 0], ry = r[j * 3 + 1], rz = r[j * 3 + 2];
            for (int t = 0; __any(t < cme); ++t) {
                const int jn = t + 1 < cme ? (int)lst[wave][t + 1][lane] : 0;
                const float nx = r[jn * 3 + 0], ny = r[jn * 3 + 1], nz = r[jn * 3 + 2];
                if (t < cme) topk_insert<K>(bd, bi, sqdist(qx, qy, qz, rx, ry, rz), j);
                j = jn; rx = nx; ry = ny; rz = nz;
            }
            // all lanes of the wave are past their last list read (the loop above is wave-uniform): overwrite the region
#pragma unroll
            for (int i = 0; i < K; ++i) { sdst[i][lane] = bd[i]; sidx[wave][i][lane] = (unsigned short)bi[i]; }
        }
        __syncthreads();
        if (wave != 0) return;
        float bd[K];
        int bi[K];
        if (__any(cnts[0][lane] > KNN4_CAP || cnts[1][lane] > KNN4_CAP || cnts[2][lane] > KNN4_CAP || cnts[3][lane] > KNN4_CAP)) {
            exact_scan<K>(qx, qy, qz, r, M, bd, bi);      // heavy ties: redo this query tile exactly (rare)
        } else {
            // heads of the four lists (list w = slice w = lower indices first: on equal distances the lower list wins)
            float hd[4];
            int hi[4], hp[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                hd[w] = reinterpret_cast<const float (*)[64]>(&lst[w][0][0])[0][lane];
                hi[w] = sidx[w][0][lane];
                hp[w] = 1;
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                int wm = 0;
                float dm = hd[0];
#pragma unroll
                for (int w = 1; w < 4; ++w)
                    if (hd[w] < dm) { dm = hd[w]; wm = w; }      // strict: ties stay with the lower slice = lower index
                int im = hi[0];
#pragma unroll
                for (int w = 1; w < 4; ++w) im = wm == w ? hi[w] : im;
                bd[i] = dm; bi[i] = im;
                // advance list wm (its entries past the end are +inf / 0xffff: never chosen before a finite head)
                int pm = hp[0];
#pragma unroll
                for (int w = 1; w < 4; ++w) pm = wm == w ? hp[w] : pm;
                const int pc = pm < K ? pm : K - 1;
                const float nd = pm < K ? reinterpret_cast<const float (*)[64]>(&lst[wm][0][0])[pc][lane] : __builtin_inff();
                const int ni = sidx[wm][pc][lane];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const bool m = wm == w;
                    hd[w] = m ? nd : hd[w]; hi[w] = m ? ni : hi[w]; hp[w] = m ? pm + 1 : hp[w];
                }
            }
        }
        if (live) store_topk<K>(bd, bi, (size_t)b * N + n, idx_out, dist_out);
        return;
    }
    __syncthreads();
    // ---- 8 or 16 slices (small batches).  Every wave sees every list length: the overflow decision is the same in all of them
    bool over = false;
    int ctot = 0;
#pragma unroll
    for (int w = 0; w < KNN4_W; ++w) { over |= cnts[w][lane] > KNN4_CAP; ctot += cnts[w][lane]; }
    if (__any(over || ctot > 64)) {                  // heavy ties: wave 0 redoes this query tile exactly (rare)
        if (wave != 0) return;
        float bd[K];
        int bi[K];
        exact_scan<K>(qx, qy, qz, r, M, bd, bi);
        if (live) store_topk<K>(bd, bi, (size_t)b * N + n, idx_out, dist_out);
        return;
    }
    // merge by RANK, all waves: wave w takes queries [w 64/W, (w+1) 64/W), one query at a time with a candidate per lane
    // (more than 64 candidates of one query count as heavy ties: exact path above).  rank = number of candidates that precede
    // this one in (distance, index) order; the candidates of rank < K are the answer, written straight to their position.
    // (Wave 0 inserting 8 or 16 lists alone was most of the kernel there: 0.081 -> 0.039 ms at 4 x 2048.)
    constexpr int QPW = 64 / KNN4_W;
    // phase 1, all of the wave's queries: this lane's candidate of each query and its distance (the loads of the QPW queries
    // are independent: issued together, one memory latency instead of QPW)
    int jq[QPW], cq[QPW];
    float dq[QPW];
#pragma unroll
    for (int t = 0; t < QPW; ++t) {
        const int ql = wave * QPW + t;                                    // query (= lane index in every wave), uniform
        cq[t] = __builtin_amdgcn_readlane(ctot, ql);
        int j0 = 0x7fffffff, acc = 0;
#pragma unroll
        for (int w = 0; w < KNN4_W; ++w) {
            const int cw = cnts[w][ql];
            const int s0 = lane - acc;
            if (s0 >= 0 && s0 < cw) j0 = lst[w][s0][ql];
            acc += cw;
        }
        jq[t] = j0;
    }
#pragma unroll
    for (int t = 0; t < QPW; ++t) {
        const int ql = wave * QPW + t;
        const float ax = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qx), ql));
        const float ay = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qy), ql));
        const float az = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qz), ql));
        const int jj = lane < cq[t] ? jq[t] : 0;
        const float d = sqdist(ax, ay, az, r[(size_t)jj * 3 + 0], r[(size_t)jj * 3 + 1], r[(size_t)jj * 3 + 2]);
        dq[t] = lane < cq[t] ? d : __builtin_inff();
    }
    // phase 2: ranks
#pragma unroll
    for (int t = 0; t < QPW; ++t) {
        const int ql = wave * QPW + t;
        const int nq = blockIdx.x * 64 + ql;
        if (nq >= N) break;                                               // uniform
        const int C = cq[t];
        const float d0 = dq[t];
        const int j0 = jq[t];
        const size_t row = (size_t)b * N + nq;
        int r0 = 0;                                                       // C <= 64: a candidate per lane
        for (int c = 0; c < C; ++c) {
            const float dc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d0), c));
            const int jc = __builtin_amdgcn_readlane(j0, c);
            r0 += (dc < d0 || (dc == d0 && jc < j0)) ? 1 : 0;
        }
        if (lane < C && r0 < K) { idx_out[row * K + r0] = j0; if (dist_out) dist_out[row * K + r0] = d0; }
    }
}

// ---- knn5_kernel: the two sweeps on the MATRIX pipe ------------------------------------------------------------------
// knn4's sweeps are bound by the VALU issue rate of their 6.5 operations per (query, reference) pair.  Here the sweeps run on
// a filter the f32 MFMA computes for a 16 x 16 tile of pairs per instruction: with the reference table a_j = (-2x, -2y, -2z,
// |r_j|^2) and b = (qx, qy, qz, 1), v_mfma_f32_16x16x4_f32 gives f_j = |r_j|^2 - 2 q.r_j = d^2 - |q|^2 - the same ORDER as the
// distance for a fixed query.  A wave owns 16 queries (MFMA columns); lane (col, g) receives the 4 references 16 c + 4 g + r of
// every 16-reference chunk c for query col: 8 strided group minima per lane = 32 per query (as before), 1 VALU minimum per
// pair instead of 6.5 operations.  Error of the filter (4 roundings of sums bounded by (|r| + |q|)^2, plus 3 in |r|^2):
// |f_j - (d_j^2 - |q|^2)| <= 8u (|r_j| + |q|)^2, u = 2^-24.  With E = 16u (sqrt(max |r|^2) + |q|)^2: at least K references have
// f <= tau (the K-th smallest group minimum), so the exact K-th neighbour distance is <= (tau + E + |q|^2)(1 + 8u), and every
// reference of the exact answer, ties included, has f <= tau + 2E + 17u (tau + E + |q|^2) <= tau + 3.2 E.  Sweep B computes
// f - (tau + 4E) (the threshold enters as the MFMA's accumulator: one more rounding of the same magnitude, covered by the 0.8 E
// of slack) and keeps the NEGATIVE results - the sign bit is the flag, shifted into a mask by one v_alignbit per value; the
// answer is computed from EXACT distances of the survivors (same bits as every other kernel of this file).
// Badly scaled clouds (coordinates >> the neighbour distance) only make E large: more survivors, and past the list capacity
// the exact scan - never a wrong answer.  NaN / inf coordinates: the filter comparisons fail -> too few survivors -> exact scan.
// A workgroup is NW waves = 16 NW queries of one batch item and builds the item's table once (NW = 16 at 32 x 2048: one workgroup
// per CU, one round over the chip).  Per-wave scratch (6 KiB): the group minima, then the survivor lists, then the sorted lists.
constexpr int KNN5_CAP = 32;             // per-lane survivor list: ~22 survivors per query at K = 16, a quarter of them per lane
constexpr int KNN5_WB = 6144;            // bytes of LDS scratch per wave
template <int K, int NW>
__global__ __launch_bounds__(NW * 64) void knn5_kernel(const float* __restrict__ p1, const float* __restrict__ p2, int N, int M,
                                                       int Mpad, int* __restrict__ idx_out, float* __restrict__ dist_out) {
    static_assert(K <= 16, "threshold selection uses two 16-element sorted halves");
    extern __shared__ float4 k5lds[];                        // [Mpad] table, then NW x KNN5_WB bytes of per-wave scratch
    __shared__ float rmx[NW];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, g = lane >> 4;
    const int n = (blockIdx.x * NW + wave) * 16 + col;
    const bool live = n < N;
    const float* __restrict__ r = p2 + (size_t)b * M * 3;
    float4* tab = k5lds;
    unsigned char* wsm = reinterpret_cast<unsigned char*>(k5lds + Mpad) + wave * KNN5_WB;
    float (*gms)[64] = reinterpret_cast<float (*)[64]>(wsm);                                  // [8][64] floats
    unsigned short (*lst)[64] = reinterpret_cast<unsigned short (*)[64]>(wsm + 2048);       // [KNN5_CAP][64]
    // ---- the reference table of this batch item, and max |r|^2
    float rm = 0.f;
    for (int j = threadIdx.x; j < Mpad; j += NW * 64) {
        float4 e = make_float4(0.f, 0.f, 0.f, __builtin_inff());      // padding: f = +inf, never a minimum, never a survivor
        if (j < M) {
            const float x = r[j * 3 + 0], y = r[j * 3 + 1], z = r[j * 3 + 2];
            const float R = fmaf(z, z, fmaf(y, y, x * x));
            e = make_float4(-2.f * x, -2.f * y, -2.f * z, R);
            rm = fmaxf(rm, R);
        }
        tab[j] = e;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) rm = fmaxf(rm, __shfl_xor(rm, m));
    if (lane == 0) rmx[wave] = rm;
    __syncthreads();
    float Rmax = rmx[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) Rmax = fmaxf(Rmax, rmx[w]);
#if defined(PF_KNN_ABL) && PF_KNN_ABL == 3          // timing-only builds (tools/time_knn5.py): table only
    if (live && g == 0) idx_out[((size_t)b * N + n) * K] = __float_as_int(Rmax);
    return;
#endif
    if ((blockIdx.x * NW + wave) * 16 >= N) return;           // a wave without queries (no workgroup barrier below this line)
    const float* qp = p1 + ((size_t)b * N + (live ? n : N - 1)) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
    const float bq = g == 0 ? qx : (g == 1 ? qy : (g == 2 ? qz : 1.f));
    const float* tabf = reinterpret_cast<const float*>(tab) + col * 4 + g;       // A operand of chunk c: tabf[c * 64]
    const int nch = Mpad / 16;                                                    // a multiple of 8
    auto filtc = [&](int c, f4 acc) { return __builtin_amdgcn_mfma_f32_16x16x4f32(tabf[c * 64], bq, acc, 0, 0, 0); };
    auto filt = [&](int c) {
        const f4 z4 = {0.f, 0.f, 0.f, 0.f};
        return filtc(c, z4);
    };
    // ---- sweep A: 8 strided group minima per lane (group = (chunk parity, register))
    {
        float gm[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) gm[i] = __builtin_inff();
        // (measured: v_min3 instead of two v_min, and a software pipeline inside the wave - the next four MFMAs issued before the
        // minima of the previous four - leave the sweep's time unchanged)
        for (int c0 = 0; c0 < nch; c0 += 8) {
            f4 d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) d[u] = filt(c0 + u);
            __builtin_amdgcn_sched_barrier(0);                  // all eight MFMAs in flight before a minimum waits for the first (57.5 -> 56 us)
#pragma unroll
            for (int u = 0; u < 8; u += 4)
#pragma unroll
                for (int v = 0; v < 2; ++v)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        gm[v * 4 + i] = __builtin_fminf(gm[v * 4 + i], __builtin_fminf(d[u + v][i], d[u + v + 2][i]));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) gms[i][lane] = gm[i];
    }
    // (the exchanges below stay inside the wave: its lanes run in lockstep and its LDS operations complete in order; the wave
    // barriers only keep the compiler from moving the reads above the writes)
    __builtin_amdgcn_wave_barrier();
    float tau;
    {
        float ha[16], hb[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            ha[i] = gms[i][col]; ha[8 + i] = gms[i][col + 16];
            hb[i] = gms[i][col + 32]; hb[8 + i] = gms[i][col + 48];
        }
        if constexpr (K == 1) {                               // nearest neighbour (pf_nn1): the smallest group minimum
            tau = fminf(ha[0], hb[0]);
#pragma unroll
            for (int i = 1; i < 16; ++i) tau = fminf(tau, fminf(ha[i], hb[i]));
        } else {
            sort16(ha);
            sort16(hb);
            tau = fminf(ha[0], hb[K - 1]);
#pragma unroll
            for (int i = 1; i < K; ++i) tau = fmaxf(tau, fminf(ha[i], hb[K - 1 - i]));
        }
        const float s = __builtin_sqrtf(Rmax) + __builtin_sqrtf(fmaf(qz, qz, fmaf(qy, qy, qx * qx)));
        tau = fmaf(3.814697265625e-6f, s * s, tau) + 4e-37f;              // + 4 E, E = 16 u s^2 (header); the absolute term also
                                                                          // covers results flushed to zero near the subnormals
    }
#if defined(PF_KNN_ABL) && PF_KNN_ABL == 2          // + sweep A and the threshold
    if (live && g == 0) idx_out[((size_t)b * N + n) * K] = __float_as_int(tau);
    return;
#endif
    // ---- sweep B: survivors of this lane's references, in index order.  ~1 % of the values pass, but SOME lane of the wave
    // has one in almost every group of four: a branch or an exec-masked append per value costs ~10 instructions each.  Instead
    // the flags of a batch (8 MFMAs = 32 values per lane) are shifted into one 32-bit mask per lane - one v_alignbit per value -
    // and the few set bits are appended afterwards (the first value of the batch is the mask's top bit: count-leading-zeros
    // walks them in index order).  Where the time goes at 32 x 2048, K = 16 (tools/time_knn5.py with -DPF_KNN_ABL=3 / 2 / 6 /
    // 1 / 5 builds, one box): table + sweep A + threshold 23 us, the masks of sweep B 13.5, the appends 5, the per-lane sorted
    // lists 8, merge + store 7 = 57 us (knn4_kernel on the same box: 76-79).
    int cnt = 0;
    const f4 ntau = {-tau, -tau, -tau, -tau};                // the MFMA's accumulator input: results are f - tau
    for (int c0 = 0; c0 < nch; c0 += 8) {
        f4 d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) d[u] = filtc(c0 + u, ntau);          // f - tau: the sign bit is the survivor flag
        __builtin_amdgcn_sched_barrier(0);                   // all eight MFMAs in flight before the first result is waited for
        // four independent shift-in chains (MFMA pairs 0-1, 2-3, 4-5, 6-7), ONE instruction per value:
        // v_alignbit(m, x, 31) = (m << 1) | sign(x)
        unsigned m0 = 0, m1 = 0, m2 = 0, m3 = 0;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int u = s >> 2, i = s & 3;
            // (inline asm: with the builtin, hipcc 7.2 reads component 0 of every MFMA result for all four - a demanded-bits
            // simplification that only keeps the sign bit in view goes wrong on the accumulator vector)
            asm("v_alignbit_b32 %0, %0, %4, 31\n\tv_alignbit_b32 %1, %1, %5, 31\n\t"
                "v_alignbit_b32 %2, %2, %6, 31\n\tv_alignbit_b32 %3, %3, %7, 31"
                : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3)
                : "v"(d[u][i]), "v"(d[2 + u][i]), "v"(d[4 + u][i]), "v"(d[6 + u][i]));
        }
        unsigned mask = (m0 << 24) | (m1 << 16) | (m2 << 8) | m3;
        // append the set bits: two straight-line pops (the wave's busiest lane has ~2 survivors per batch), a loop for the rest
        auto pop = [&]() {
            if (mask != 0) {
                const int p = __clz((int)mask);                                  // value u * 4 + i of the batch
                mask &= ~(0x80000000u >> p);
                lst[cnt & (KNN5_CAP - 1)][lane] = (unsigned short)((c0 + (p >> 2)) * 16 + 4 * g + (p & 3));
                ++cnt;                                                           // (a list that wraps is caught by its count below)
            }
        };
#if defined(PF_KNN_ABL) && PF_KNN_ABL == 6          // timing-only: the masks without the appends
        cnt += __popc(mask);
#else
        pop();
        pop();
        while (__any(mask != 0)) pop();
#endif
    }
#if defined(PF_KNN_ABL) && (PF_KNN_ABL == 1 || PF_KNN_ABL == 6)          // + sweep B
    if (live && g == 0) idx_out[((size_t)b * N + n) * K] = cnt;
    return;
#endif
    // ---- this lane's survivors sorted by (exact distance, index); coordinates from the table (x = -0.5 (-2x), exact)
    float bd[K];
    int bi[K];
    {
#pragma unroll
        for (int i = 0; i < K; ++i) { bd[i] = __builtin_inff(); bi[i] = 0xffff; }
        const int cme = cnt < KNN5_CAP ? cnt : KNN5_CAP;
        // the t-th survivor can only land in the first t + 1 slots: the first K insertions are unrolled with an insert that
        // touches just those (5 (t + 1) operations instead of 5 K; a lane has ~K / 3 survivors, the wave's longest list ~K).
        // Straight-line code in groups of four (a lane past its list inserts +inf: a no-op), so that the LDS reads of the next
        // survivors - list entry, then its table row - are in flight while the current one is inserted.
        auto dist_of = [&](int t) {
            int j = lst[t][lane];
            j = j < Mpad ? j : 0;                                               // past the lane's count: any valid row
            const float4 e = tab[j];
            return t < cme ? sqdist(qx, qy, qz, -0.5f * e.x, -0.5f * e.y, -0.5f * e.z) : __builtin_inff();
        };
        pf_static_for_knn<0, (K + 3) / 4>([&](auto gc) {
            constexpr int G4 = decltype(gc)::value;
            if (G4 == 0 || __any(cme > 4 * G4)) {
                float dd[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) dd[k] = dist_of(4 * G4 + k);
                pf_static_for_knn<0, 4>([&](auto kc) {
                    constexpr int T = 4 * G4 + decltype(kc)::value;
                    if (T < K) topk_insert_head<K, T>(bd, bi, dd[T - 4 * G4], (int)lst[T][lane]);
                });
            }
        });
        for (int t = K; __any(t < cme); ++t) {
            if (t < cme) {
                const int j = lst[t][lane];
                const float4 e = tab[j];
                topk_insert<K>(bd, bi, sqdist(qx, qy, qz, -0.5f * e.x, -0.5f * e.y, -0.5f * e.z), j);
            }
        }
    }
#if defined(PF_KNN_ABL) && PF_KNN_ABL == 5          // + the per-lane sorted lists
    if (live && g == 0) idx_out[((size_t)b * N + n) * K] = bi[0] + bi[K - 1];
    return;
#endif
    // the lists are read: the wave's scratch takes the sorted lists ([K][64] floats, then [K][64] indices)
    __builtin_amdgcn_wave_barrier();
    float (*sd)[64] = reinterpret_cast<float (*)[64]>(wsm);
    unsigned short (*si)[64] = reinterpret_cast<unsigned short (*)[64]>(wsm + 4096);
    static_assert(K * 64 * 4 <= 4096 && 4096 + K * 64 * 2 <= KNN5_WB, "per-wave scratch holds the sorted lists");
#pragma unroll
    for (int i = 0; i < K; ++i) { sd[i][lane] = bd[i]; si[i][lane] = (unsigned short)bi[i]; }
    __builtin_amdgcn_wave_barrier();
    // survivors of the query = the four lanes of its column
    int ctot = cnt;
    ctot += __shfl_xor(ctot, 16);
    ctot += __shfl_xor(ctot, 32);
    int cmax = cnt;
    cmax = max(cmax, __shfl_xor(cmax, 16));
    cmax = max(cmax, __shfl_xor(cmax, 32));
#if defined(PF_KNN_ABL) && PF_KNN_ABL == 4          // timing-only: never the exact path
    if (false) {
#else
    if (__any(cmax > KNN5_CAP || ctot < K)) {                 // heavy ties / NaN / inf coordinates: this wave's queries exactly
#endif
        exact_scan<K>(qx, qy, qz, r, M, bd, bi);
        if (live && g == 0) store_topk<K>(bd, bi, (size_t)b * N + n, idx_out, dist_out);
        return;
    }
    if (g != 0) return;
    // merge the column's four sorted lists by (distance, index)
    float hd[4];
    int hi[4], hp[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) { hd[w] = sd[0][col + 16 * w]; hi[w] = si[0][col + 16 * w]; hp[w] = 1; }
#pragma unroll
    for (int i = 0; i < K; ++i) {
        int wm = 0;
        float dm = hd[0];
        int im = hi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (hd[w] < dm || (hd[w] == dm && hi[w] < im)) { dm = hd[w]; im = hi[w]; wm = w; }
        bd[i] = dm; bi[i] = im;
        int pm = hp[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) pm = wm == w ? hp[w] : pm;
        const int pc = pm < K ? pm : K - 1;
        const float nd = pm < K ? sd[pc][col + 16 * wm] : __builtin_inff();
        const int ni = pm < K ? (int)si[pc][col + 16 * wm] : 0xffff;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const bool m = wm == w;
            hd[w] = m ? nd : hd[w]; hi[w] = m ? ni : hi[w]; hp[w] = m ? pm + 1 : hp[w];
        }
    }
    if (live) store_topk<K>(bd, bi, (size_t)b * N + n, idx_out, dist_out);
}

// K = 1: nearest neighbour distance + index (first minimum wins ties).
__global__ __launch_bounds__(256) void nn1_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                                  int N, int M, float* __restrict__ dist_out,
                                                  int* __restrict__ idx_out) {
    const int b = blockIdx.y;
    const int n = blockIdx.x * 256 + threadIdx.x;
    const bool live = n < N;
    const float* q = p1 + ((size_t)b * N + (live ? n : N - 1)) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    const float* __restrict__ r = p2 + (size_t)b * M * 3;
    // references are wave-uniform: scalar loads, SGPR operands (see knn4_kernel)
    float best = __builtin_inff(), thr = __builtin_inff();
    int besti = 0;
    auto visit = [&](int j) {
        const float* rr = r + (size_t)j * 3;
        const float d = sqdist(qx, qy, qz, rr[0], rr[1], rr[2]);
        const bool lt = d < best;                               // strict: the first minimum is kept
        best = lt ? d : best;
        besti = lt ? j : besti;
    };
    // Four references at a time through the fused FILTER distance of knn4_kernel (same rounded differences, two fmas: 6
    // operations instead of 8, within a factor 1 +- 7 * 2^-24 of the exact value): a reference can only lower `best` if its
    // exact distance is below it, i.e. its filter distance below best (1 + 2^-20) - only then (rarely, once the scan has seen
    // a few near points) the group's exact distances are computed, in index order, with the strict comparison above.  Same
    // result bit for bit; 11 -> ~7 operations per pair.
    auto filt = [&](int j) {
        const float* rr = r + (size_t)j * 3;
        const float dx = qx - rr[0], dy = qy - rr[1], dz = qz - rr[2];
        return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    };
    const int mfull = (M / 32) * 32;
    for (int j0 = 0; j0 < mfull; j0 += 32) {
        float f[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) f[c] = filt(j0 + c);
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            if (fminf(fminf(f[c], f[c + 1]), fminf(f[c + 2], f[c + 3])) < thr) {
                visit(j0 + c); visit(j0 + c + 1); visit(j0 + c + 2); visit(j0 + c + 3);
                thr = fmaf(best, 9.5367431640625e-7f, best) + 1e-37f;
            }
        }
    }
    for (int j = mfull; j < M; ++j) visit(j);
    if (live) {
        dist_out[(size_t)b * N + n] = best;
        if (idx_out) idx_out[(size_t)b * N + n] = besti;
    }
}


// knn5_kernel with as many waves per workgroup as still give every CU a workgroup (the table is built once per workgroup)
template <int K>
static void launch_knn5(const float* p1, const float* p2, int B, int N, int M, int* idx_out, float* dist_out, hipStream_t s) {
    const int Mpad = (M + 127) / 128 * 128;
    int ncu = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    int NW = PF_KNN5_NWMAX;
    while (NW > 4 && ((long long)B * ((N + 16 * NW - 1) / (16 * NW)) < ncu || (size_t)Mpad * 16 + (size_t)NW * KNN5_WB > 150 * 1024)) NW >>= 1;
    const size_t lds = (size_t)Mpad * 16 + (size_t)NW * KNN5_WB;
    const dim3 g5((N + 16 * NW - 1) / (16 * NW), B);
#define PF_KNN5_LAUNCH(W)                                                                                                 \
    do { pf_allow_lds(reinterpret_cast<const void*>(knn5_kernel<K, W>), lds);                                            \
         hipLaunchKernelGGL((knn5_kernel<K, W>), g5, dim3(W * 64), lds, s, p1, p2, N, M, Mpad, idx_out, dist_out); } while (0)
    if (NW == 16) PF_KNN5_LAUNCH(16); else if (NW == 8) PF_KNN5_LAUNCH(8); else PF_KNN5_LAUNCH(4);
#undef PF_KNN5_LAUNCH
}

// which shapes take knn5_kernel: the table must fit in LDS; 64 query tiles and up; 256 <= M < 1024 (the training step's 32 x 256:
// 46 -> 16 us) only while the grid is small - there the kernel's latency is what counts; PU-GAN's 2 496 patches of 256 points
// stay on knn2_kernel (179 us against 219)
static bool use_knn5(int B, int N, int M) {
#if PF_KNN5
    const long long wgs = (long long)((N + 63) / 64) * B;
    return wgs >= PF_KNN5_MIN_WGS && (M >= 1024 || (M >= PF_KNN5_MIN_M && wgs <= 1024)) && M <= 4096;
#else
    (void)B; (void)N; (void)M;
    return false;
#endif
}

}  // namespace

extern "C" int pf_knn(const float* p1, const float* p2, int B, int N, int M, int K, int* idx_out,
                      float* dist_out, void* stream) {
    if (!p1 || !p2 || !idx_out) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || M <= 0 || K <= 0 || K > M || B > 65535) return PF_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    if (K <= 16 && M >= (PF_KNN5_MIN_M < 1024 ? PF_KNN5_MIN_M : 1024) && M <= 65536) {  // two-sweep kernels: sweeps as f32 MFMAs, or references split over 4 / 8 / 16 waves
        const dim3 g4((N + 63) / 64, B);
        const long long wgs = (long long)g4.x * B;
        if (use_knn5(B, N, M) && (K == 4 || K == 8 || K == 16)) {       // the sweeps as f32 MFMAs
            switch (K) {
                case 4:  launch_knn5<4>(p1, p2, B, N, M, idx_out, dist_out, s); break;
                case 8:  launch_knn5<8>(p1, p2, B, N, M, idx_out, dist_out, s); break;
                default: launch_knn5<16>(p1, p2, B, N, M, idx_out, dist_out, s); break;
            }
            return pf_last_launch_status();
        }
        if (M >= 1024) {
        const int W = wgs >= 1024 ? 4 : (wgs >= 384 ? 8 : 16);
#define PF_KNN4_LAUNCH(KK)                                                                                              \
        if (W == 4) hipLaunchKernelGGL((knn4_kernel<KK, 4>), g4, dim3(256), 0, s, p1, p2, N, M, idx_out, dist_out);         \
        else if (W == 8) hipLaunchKernelGGL((knn4_kernel<KK, 8>), g4, dim3(512), 0, s, p1, p2, N, M, idx_out, dist_out);    \
        else hipLaunchKernelGGL((knn4_kernel<KK, 16>), g4, dim3(1024), 0, s, p1, p2, N, M, idx_out, dist_out)
        switch (K) {
            case 4:  PF_KNN4_LAUNCH(4); break;
            case 8:  PF_KNN4_LAUNCH(8); break;
            case 16: PF_KNN4_LAUNCH(16); break;
            default: return PF_ERR_UNSUPPORTED;
        }
#undef PF_KNN4_LAUNCH
        return pf_last_launch_status();
        }
    }
    if (K <= 16 && M >= 256 && M <= 65536) {           // two-sweep kernel (u16 candidate lists)
        dim3 g2((N + KNN2_T - 1) / KNN2_T, B), b2(KNN2_T);
        switch (K) {
            case 4:  hipLaunchKernelGGL(knn2_kernel<4>, g2, b2, 0, s, p1, p2, N, M, idx_out, dist_out); break;
            case 8:  hipLaunchKernelGGL(knn2_kernel<8>, g2, b2, 0, s, p1, p2, N, M, idx_out, dist_out); break;
            case 16: hipLaunchKernelGGL(knn2_kernel<16>, g2, b2, 0, s, p1, p2, N, M, idx_out, dist_out); break;
            default: return PF_ERR_UNSUPPORTED;
        }
        return pf_last_launch_status();
    }
    dim3 grid((N + 63) / 64, B), block(64);
    switch (K) {
        case 8:  hipLaunchKernelGGL(knn_kernel<8>, grid, block, 0, s, p1, p2, N, M, idx_out, dist_out); break;
        case 16: hipLaunchKernelGGL(knn_kernel<16>, grid, block, 0, s, p1, p2, N, M, idx_out, dist_out); break;
        case 4:  hipLaunchKernelGGL(knn_kernel<4>, grid, block, 0, s, p1, p2, N, M, idx_out, dist_out); break;
        case 32: hipLaunchKernelGGL(knn_kernel<32>, grid, block, 0, s, p1, p2, N, M, idx_out, dist_out); break;
        default: return PF_ERR_UNSUPPORTED;
    }
    return pf_last_launch_status();
}

extern "C" int pf_nn1(const float* p1, const float* p2, int B, int N, int M, float* dist_out, int* idx_out,
                      void* stream) {
    if (!p1 || !p2 || !dist_out) return PF_ERR_NULL;
    if (B <= 0 || N <= 0 || M <= 0 || B > 65535) return PF_ERR_SHAPE;
    if (use_knn5(B, N, M)) {             // K = 1 of the MFMA-filter kernel: (distance, index) order = the first minimum
        launch_knn5<1>(p1, p2, B, N, M, idx_out, dist_out, (hipStream_t)stream);
        return pf_last_launch_status();
    }
    hipLaunchKernelGGL(nn1_kernel, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, p1, p2, N, M, dist_out,
                       idx_out);
    return pf_last_launch_status();
}
